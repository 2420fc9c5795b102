#!/usr/bin/env python3
"""A stand-in for the two `samtools` sub-commands the pileup front end spawns, over TEXT inputs (test support only).

The reference's dataPrepScripts (CreateTensor.py:113-170, ExtractVariantCandidates.py:131-156, 259-261) never link
htslib: they run `samtools faidx <fasta> <region>` and `samtools view -F <flags> <bam> <region>` and read the text that comes
back.  No samtools exists in this image, so the tests (and tools/make_pileup_goldens.py, which drives the REAL reference
scripts) pass `--samtools "python tests/fake_samtools.py"`: `<bam>` is then a SAM text file.

    faidx <fasta> [region ...]   region = ctg | ctg:start-end (1-based, inclusive, clamped); 60 columns per line
    view [-@ N] -F <int> <sam> [region] alignments with FLAG & int == 0 that overlap the region, header lines dropped
         [--no-PG] [-x TAG] [--keep-tag TAG[,TAG]]   as samtools: no effect on alignment lines / drop a tag / keep only these tags
"""
import re
import sys


def read_fasta(path):
    seqs, name = {}, None
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith(">"):
                name = line[1:].split()[0]
                seqs[name] = []
            elif name is not None:
                seqs[name].append(line)
    return {k: "".join(v) for k, v in seqs.items()}


def parse_region(region):
    m = re.fullmatch(r"(.+):(\d+)-(\d+)", region)
    if m:
        return m.group(1), int(m.group(2)), int(m.group(3))
    return region, None, None


def faidx(argv):
    seqs = read_fasta(argv[0])
    for region in argv[1:]:
        ctg, start, end = parse_region(region)
        if ctg not in seqs:
            sys.stderr.write("[faidx] region %s not found\n" % region)
            return 1
        s = seqs[ctg]
        if start is not None:
            s = s[max(start, 1) - 1:min(end, len(s))]
        sys.stdout.write(">%s\n" % region)
        for i in range(0, len(s), 60):
            sys.stdout.write(s[i:i + 60] + "\n")
    return 0


def reference_span(cigar):
    return sum(int(n) for n, op in re.findall(r"(\d+)([MIDNSHP=X])", cigar) if op in "MDN=X")


def view(argv):
    flags, drop, keep = 0, set(), None
    while argv and argv[0].startswith("-"):
        if argv[0] == "--no-PG":
            argv = argv[1:]
            continue
        if argv[0] == "-F":
            flags = int(argv[1])
        elif argv[0] == "-x":
            drop.add(argv[1])
        elif argv[0] == "--keep-tag":
            keep = set(argv[1].split(","))
        elif argv[0] != "-@":                # -@ N: threads, nothing to do here
            sys.stderr.write("[view] unknown option %s\n" % argv[0])
            return 1
        argv = argv[2:]
    path, regions = argv[0], [parse_region(r) for r in argv[1:]]
    with open(path) as f:
        for line in f:
            if line.startswith("@"):
                continue
            col = line.split("\t")
            if len(col) < 10 or int(col[1]) & flags:
                continue
            pos = int(col[3])
            end = pos + max(reference_span(col[5]), 1) - 1
            if regions and not any(col[2] == c and (s is None or (pos <= e and end >= s)) for c, s, e in regions):
                continue
            if (drop or keep is not None) and len(col) > 11:
                tags = [t for t in line.rstrip("\n").split("\t")[11:] if t[:2] not in drop and (keep is None or t[:2] in keep)]
                line = "\t".join(col[:11]).rstrip("\n") + "".join("\t" + t for t in tags) + "\n"
            sys.stdout.write(line)
    return 0


def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "faidx":
        return faidx(sys.argv[2:])
    if len(sys.argv) >= 3 and sys.argv[1] == "view":
        return view(sys.argv[2:])
    sys.stderr.write(__doc__)
    return 1


if __name__ == "__main__":
    sys.exit(main())
