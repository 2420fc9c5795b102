"""Synthetic inputs for the pileup front end (test support): a reference FASTA, sorted SAM text and a candidate list.

Used by tools/make_pileup_goldens.py (which feeds them to the REAL reference scripts in the build container to mint
tests/golden/pileup_*.json.gz) and by tests/test_pileup.py (native code vs the Python restatement on fresh seeds).
The reads exercise what dataPrepScripts/CreateTensor.py:251-373 branches on: M/=/X/I/D/S/H/N/P CIGAR operations, both
strands, mapping-quality filter, more than `dcov` reads starting at one position, IUPAC and lower-case bases in the
reference and in SEQ, insertions next to the window edge, candidates outside the region.
"""
import numpy as np

BASES = "ACGT"


def _mutate(rng, base):
    return BASES[(BASES.index(base) + int(rng.integers(1, 4))) % 4] if base in BASES else "A"


def synth_case(seed, ref_len=3000, n_reads=300, read_len=(40, 300), ctg="chrS", cand_step=(3, 40),
               sub_rate=0.04, ins_rate=0.02, del_rate=0.02, dup_burst=0, iupac=True, second_ctg=True, skip_ops=True,
               lead_indel=0.0, lead_indel_late=True):
    """lead_indel: share of reads whose CIGAR begins (after a clip) with an insertion or deletion -- what the candidate search tallies at
    POS - 1 (ExtractVariantCandidates.py:326-336) and the pileup hangs on no window; lead_indel_late=False gives one only to reads that
    are the first at their start position (the search then sees it together with the rest of POS - 1).  Drawn from a generator of its
    own, so that every other byte of a case is what it was without the option."""
    rng = np.random.default_rng(seed)
    lead_rng = np.random.default_rng(seed + 7919) if lead_indel > 0 else None
    ref = "".join(BASES[i] for i in rng.integers(0, 4, ref_len))
    ref = list(ref)
    if iupac:
        for p in rng.integers(0, ref_len, max(1, ref_len // 300)):
            ref[p] = "NRYSWKMBDHV"[int(rng.integers(0, 11))]
        a = int(rng.integers(0, ref_len - 60))
        for p in range(a, a + 50):       # a soft-masked stretch
            ref[p] = ref[p].lower()
    ref = "".join(ref)
    het = {int(p): _mutate(rng, ref[p].upper()) for p in rng.integers(0, ref_len, ref_len // 60)}

    starts = np.sort(rng.integers(0, ref_len - read_len[0], n_reads))
    if dup_burst:
        k = int(rng.integers(0, n_reads - dup_burst))
        starts[k:k + dup_burst] = starts[k]
    reads = []
    for ri, start in enumerate(starts):
        start = int(start)
        want = int(rng.integers(read_len[0], read_len[1]))
        ops, seq, rp = [], [], start
        if rng.random() < 0.15:
            n = int(rng.integers(1, 12))
            if rng.random() < 0.3:
                ops.append((n, "H"))
            else:
                ops.append((n, "S"))
                seq += [BASES[i] for i in rng.integers(0, 4, n)]
        use_eqx = rng.random() < 0.15
        if lead_rng is not None and lead_rng.random() < lead_indel and (lead_indel_late or ri == 0 or int(starts[ri - 1]) != start):
            n = int(lead_rng.integers(1, 4))
            if lead_rng.random() < 0.5:
                ops.append((n, "I"))
                seq += [BASES[i] for i in lead_rng.integers(0, 4, n)]
            elif rp + n < ref_len - 2:
                ops.append((n, "D"))
                rp += n
            if lead_rng.random() < 0.2 and ops and ops[-1][1] == "I" and rp + 2 < ref_len - 2:      # both before the first matched base
                ops.append((2, "D"))
                rp += 2
        while rp < min(start + want, ref_len):
            r = rng.random()
            if r < ins_rate and ops and ops[-1][1] in "M=X":
                n = int(rng.integers(1, 5)) if rng.random() < 0.9 else int(rng.integers(5, 25))
                ops.append((n, "I"))
                seq += [BASES[i] for i in rng.integers(0, 4, n)]
            elif r < ins_rate + del_rate and ops and ops[-1][1] in "M=X":
                n = int(rng.integers(1, 5)) if rng.random() < 0.9 else int(rng.integers(5, 40))
                n = min(n, ref_len - rp - 1)
                if n > 0:
                    # N moves neither cursor in the reference's loops: everything after it is walked n bases early (what real spliced
                    # alignments would get too); skip_ops=False keeps the reads aligned end to end
                    ops.append((n, "D" if rng.random() < 0.95 or not skip_ops else "N"))
                    rp += n
            else:
                rb = ref[rp].upper()
                qb = het[rp] if rp in het and (ri & 1) else (rb if rb in BASES else "A")
                mism = rng.random() < sub_rate
                if mism:
                    qb = _mutate(rng, qb)
                if rng.random() < 0.003:
                    qb = "N"
                op = ("X" if qb != rb else "=") if use_eqx else "M"
                if ops and ops[-1][1] == op:
                    ops[-1] = (ops[-1][0] + 1, op)
                else:
                    ops.append((1, op))
                seq.append(qb if rng.random() > 0.02 else qb.lower())
                rp += 1
        if rng.random() < 0.1:
            n = int(rng.integers(1, 12))
            ops.append((n, "S"))
            seq += [BASES[i] for i in rng.integers(0, 4, n)]
        if rng.random() < 0.02:
            ops.append((2, "P"))
        flag = 16 if rng.random() < 0.5 else 0
        r = rng.random()
        if r < 0.03:
            flag |= 256
        elif r < 0.05:
            flag |= 2048
        elif r < 0.06:
            flag |= 4
        mq = int(rng.integers(0, 61)) if rng.random() < 0.3 else 60
        cigar = "".join("%d%s" % o for o in ops)
        s = "".join(seq)
        reads.append("read%d\t%d\t%s\t%d\t%d\t%s\t*\t0\t0\t%s\t%s" % (ri, flag, ctg, start + 1, mq, cigar, s, "I" * len(s)))
    if second_ctg:
        reads.append("other0\t0\tchrOther\t5\t60\t20M\t*\t0\t0\t%s\t%s" % ("A" * 20, "I" * 20))

    cands, p = [], int(rng.integers(1, 30))
    while p <= ref_len:
        cands.append("%s\t%d\t%s\t%d\t0\t0\t0\t0\t0\t0" % (ctg, p, ref[p - 1].upper(), int(rng.integers(4, 60))))
        p += int(rng.integers(cand_step[0], cand_step[1]))
    fasta = ">%s\n" % ctg + "\n".join(ref[i:i + 60] for i in range(0, ref_len, 60)) + "\n"
    if second_ctg:
        fasta += ">chrOther\n" + "ACGT" * 30 + "\n"
    sam = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:%s\tLN:%d\n" % (ctg, ref_len) + "\n".join(reads) + "\n"
    return {"ctg": ctg, "fasta": fasta, "sam": sam, "candidates": "\n".join(cands) + "\n", "ref_len": ref_len}
