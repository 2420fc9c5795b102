"""Synthetic long-read alignments for the front-end benchmarks, generated per EVENT instead of per base (tests/pileup_synth.py walks
every base in Python: fine for the 3 kb test contigs, minutes for a megabase at 50x).

    case = fast_reads.make(ref_len=2_000_000, depth=50, read_len=(2000, 9000), seed=1)
    -> dict(ctg, fasta, sam (bytes: what `samtools view` prints), ref_len, n_reads)

Reads are sorted, both strands, MAPQ 60, CIGARs of M / I / D only: substitutions 4 %, insertions and deletions 2 % of positions each
(lengths 1-4, occasionally up to 24), heterozygous SNPs every ~1 kb carried by every other read, and systematic-error sites (one
position in `noisy_every`, miscalled by a quarter of the reads) -- which is what leaves one candidate site per ~noisy_every bases at
the default 0.125 allele-frequency threshold, the density knob of the benchmarks.
"""
import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def make(ref_len=200000, depth=50, read_len=(2000, 9000), seed=1, ctg="chrS", sub_rate=0.04, ins_rate=0.02, del_rate=0.02, noisy_every=25):
    rng = np.random.default_rng(seed)
    ref = BASES[rng.integers(0, 4, ref_len)]
    het_pos = np.unique(rng.integers(0, ref_len, ref_len // 1000))
    het_base = BASES[(np.searchsorted(BASES, ref[het_pos]) + rng.integers(1, 4, len(het_pos))) % 4]
    alt = ref.copy()
    alt[het_pos] = het_base
    noisy = rng.random(ref_len) < 1.0 / noisy_every        # systematic-error sites: a quarter of the reads miscall them
    mean_len = (read_len[0] + read_len[1]) / 2.0
    n_reads = int(ref_len * depth / mean_len)
    starts = np.sort(rng.integers(0, max(1, ref_len - read_len[0]), n_reads))
    lines = []
    for ri in range(n_reads):
        start = int(starts[ri])
        span = min(int(rng.integers(read_len[0], read_len[1])), ref_len - start)
        hap = alt if ri & 1 else ref
        seg = hap[start:start + span].copy()
        # substitutions in place
        r = rng.random(span)
        m = np.nonzero((r < sub_rate) | (noisy[start:start + span] & (r < 0.25)))[0]
        seg[m] = BASES[(np.searchsorted(BASES, seg[m]) + rng.integers(1, 4, len(m))) % 4]
        # indel events at sorted reference offsets (never at offset 0, never adjacent)
        n_ev = rng.binomial(span, ins_rate + del_rate)
        at = np.unique(rng.integers(2, max(3, span - 30), n_ev)) if span > 40 else np.zeros(0, np.int64)
        at = at[np.concatenate([[True], np.diff(at) > 30])] if len(at) else at
        is_ins = rng.random(len(at)) < ins_rate / (ins_rate + del_rate)
        size = np.where(rng.random(len(at)) < 0.9, rng.integers(1, 5, len(at)), rng.integers(5, 25, len(at)))
        cigar, pieces, prev = [], [], 0
        for a, ins, n in zip(at.tolist(), is_ins.tolist(), size.tolist()):
            cigar.append("%dM" % (a - prev))
            pieces.append(seg[prev:a])
            if ins:
                cigar.append("%dI" % n)
                pieces.append(BASES[rng.integers(0, 4, n)])
                prev = a
            else:
                cigar.append("%dD" % n)
                prev = a + n
        if span - prev > 0:
            cigar.append("%dM" % (span - prev))
            pieces.append(seg[prev:span])
        seq = np.concatenate(pieces).tobytes()
        flag = 16 if rng.random() < 0.5 else 0
        lines.append(b"r%d\t%d\t%s\t%d\t60\t%s\t*\t0\t0\t%s\t%s" % (ri, flag, ctg.encode(), start + 1, "".join(cigar).encode(), seq, b"I" * len(seq)))
    ref_s = ref.tobytes().decode()
    fasta = ">%s\n" % ctg + "\n".join(ref_s[i:i + 60] for i in range(0, ref_len, 60)) + "\n"
    return {"ctg": ctg, "fasta": fasta, "ref": ref_s, "sam": b"\n".join(lines) + b"\n", "ref_len": ref_len, "n_reads": n_reads}
