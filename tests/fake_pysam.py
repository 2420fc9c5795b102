"""A stand-in for the `pysam` module (absent from this image) with the four calls the call_var path makes
(/root/reference/clair/call_var.py:80-99, 102-170, 230-231): AlignmentFile(path, mode="rb").pileup(...) yielding columns with
.reference_pos and .get_query_sequences(mark_matches=False, mark_ends=False, add_indels=True), FastaFile(filename=...).fetch(
reference=, start=, end=), and .close().  The "BAM" is a JSON file {contig: {reference_pos (0-based): [pileup tokens]}} and the
"FASTA" a JSON file {contig: sequence}.  Installed as sys.modules["pysam"] both when the goldens are minted from the real
reference (tools/make_pysam_goldens.py) and when clair_amd is tested against them, the way tests/fake_samtools.py stands in
for samtools."""
import json


class _Column(object):
    def __init__(self, reference_pos, tokens):
        self.reference_pos = reference_pos
        self._tokens = tokens

    def get_query_sequences(self, mark_matches=False, mark_ends=False, add_indels=False):
        assert add_indels and not mark_matches and not mark_ends
        return list(self._tokens)


class AlignmentFile(object):
    def __init__(self, path, mode="rb"):
        with open(path) as f:
            self._columns = {ctg: {int(p): toks for p, toks in cols.items()} for ctg, cols in json.load(f).items()}
        self.closed = False
        self.queries = 0

    def pileup(self, contig, start=None, stop=None, flag_filter=None, min_base_quality=None, max_depth=None, **kw):
        # like htslib, every column of every read overlapping [start, stop) comes back, not only the requested ones:
        # the caller filters on reference_pos (call_var.py:113-114)
        self.queries += 1
        assert flag_filter == 2316 and min_base_quality == 0 and max_depth == 250
        cols = self._columns.get(contig, {})
        for p in range(start - 3, stop + 3):
            if p in cols:
                yield _Column(p, cols[p])

    def close(self):
        self.closed = True


class FastaFile(object):
    def __init__(self, filename=None):
        with open(filename) as f:
            self._seq = json.load(f)
        self.closed = False

    def fetch(self, reference=None, start=None, end=None):
        return self._seq[reference][start:end]

    def close(self):
        self.closed = True
