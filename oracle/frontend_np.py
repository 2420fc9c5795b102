"""CPU restatement (NumPy) of the BAM front end as the DEVICE computes it: per-position column tables, then window assembly.

TEST INFRASTRUCTURE ONLY.  Nothing under ``clair_amd/`` may import this module; only ``tests/`` use it, as the checker of
clair_amd/hostsrc/host_sampack.cpp (the packer) and of clair_amd/csrc/frontend.hip.h (the kernels).

What it restates (reference = /root/reference/dataPrepScripts):

  ExtractVariantCandidates.py:143-157, 259-345   read filter (RNAME, MQ, '*', 55 % aligned), per-position tallies A C G T I D N,
                                                 insertion / deletion counted once per operation at the base before it
  ExtractVariantCandidates.py:347-393            depth / allele-frequency filter, dict-order ties, range and bed restriction
  CreateTensor.py:251-373                        the read walk: which (read base, window) pairs exist
  CreateTensor.py:29-65                          generate_tensor: what a pair adds to the [33][8][4] window
  CreateTensor.py:181, 283-287, 369-373          the budget of outstanding tuples and when a window's tuples are released
  clair/utils.py:90-91                           windows whose centre base is not an IUPAC code are dropped

PARITY PIN: the sequential restatements of the same two scripts (clair_amd/create_tensor.py: PileupBuilderPy,
clair_amd/extract_variant_candidates.py and their C++ twins) are pinned byte for byte against records minted from the real
scripts (tests/golden/pileup_ct_*.json.gz, pileup_evc_*.json.gz); tests/test_frontend.py pins THIS formulation against those
golden records and against the sequential code on fresh synthetic alignments.

The reformulation.  For a sorted candidate list, left-edge windows and a tuple budget that never binds, what a read adds to the
window of centre c (1-based) depends only on the reference position rp (0-based) of the read base, not on c:

  M at rp            -> column idx = rp - c + 17, rows of the reference base and of the read base; counted for every idx in [0,33)
  D at rp (rp > POS) -> column idx, row of the reference base, channel 2; counted for idx in [1,33): a window opens AFTER a deleted
                        base has been offered (CreateTensor.py:343-361), so the deletion under a window's first column never counts
  I at rp (rp > POS), k-th base -> column min(idx + k, 32), channel 1; counted for idx in [1,33)

(the `rp > POS` conditions: a read opens its windows at its first M, or after its first D).  So the windows are assembled from
per-position tables -- M read-base rows, M and D counts per strand -- plus a scatter of the insertion bases, and the tuple budget
is verified afterwards from per-read and per-window tuple counts (budget_binds); where it binds, or where the input leaves the
regime above, the caller falls back to the sequential host code.
"""
import numpy as np

FLANK = 16
N_POS = 33
LOOKAHEAD = 100000
SLOTS = 5000000

IUPAC_KEYS = "ACGTURYSWKMBDHVN"
PILE_ROW = np.full(256, 255, np.uint8)        # IUPAC_base_to_num_dict (shared/utils.py:24-27)
EVC_IDX = np.full(256, 255, np.uint8)         # IUPAC_base_to_ACGT_base_dict through evc_base_from (shared/utils.py:19-22, 27-28): N stays N
for _k, _v, _a in zip(IUPAC_KEYS, (0, 1, 2, 3, 3, 0, 1, 1, 0, 2, 0, 1, 0, 0, 0, 0), "ACGTTACCAGACAAAA"):
    PILE_ROW[ord(_k)] = _v
    EVC_IDX[ord(_k)] = "ACGT".index(_a)
EVC_IDX[ord("N")] = 6

OP_M, OP_I, OP_D = 0, 1, 2
F_STRAND, F_EVC, F_PILE, F_FLUSH = 1, 2, 4, 8

A_UNSORTED, A_ZERO_INDEL, A_LONG_SPAN, A_SEQ_OVERRUN, A_BAD_BASE, A_BAD_REF, A_OVERFLOW, A_BUDGET, A_CANDIDATES, A_LEAD_INDEL = (1 << i for i in range(10))


def _is_space(c):
    return c == 32 or 9 <= c <= 13 or 0x1c <= c <= 0x1f


def pack_sam(sam, ctg_name, dcov=250, evc_min_mq=0, pile_min_mq=0, pile_region=None):
    """SAM text (bytes) -> packed reads, the Python twin of clair_host_sampack_*.  pile_region = (start, end) 1-based inclusive:
    the pileup takes only the alignments `samtools view ctg:start-end` would print (CreateTensor.py:163-170)."""
    if isinstance(sam, str):
        sam = sam.encode("latin-1")
    ctg = ctg_name.encode()
    pos0, flags, seq0, seq_len, op0, n_ops = [], [], [], [], [], []
    code, length, ref_off, q_off, op_read = [], [], [], [], []
    seq_parts, seq_at = [], 0
    anomalies = 0
    prev_pos, depth_cap = 0, 0
    last_pos = None
    evc_last_pos = None
    n_lines = 0
    lines = sam.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    for line in lines:
        col = line.split()
        if not col:
            raise ValueError("alignment line is empty")
        if col[0][:1] == b"@":
            continue
        n_lines += 1
        if len(col) < 10:
            raise ValueError("alignment line has %d columns (11 expected)" % len(col))
        flag, pos1, mq = int(col[1]), int(col[3]), int(col[4])
        cigar, seq = col[5], col[9].upper()
        pos = pos1 - 1
        # the operations both scripts walk: M/=/X, I, D; S moves the read cursor; anything else is skipped without moving either cursor
        ops, adv, rp, qp, soft, total, zero_indel, lead_indel = [], 0, 0, 0, 0, 0, False, False
        for ch in cigar:
            if 48 <= ch <= 57:
                adv = adv * 10 + (ch - 48)
                continue
            c = chr(ch)
            if c == "S":
                soft += adv
                qp += adv
            elif c in "M=X":
                if adv:
                    ops.append((OP_M, adv, rp, qp))
                rp += adv
                qp += adv
            elif c == "I":
                if adv:
                    ops.append((OP_I, adv, rp, qp))
                    lead_indel |= rp == 0
                else:
                    zero_indel = True
                qp += adv
            elif c == "D":
                if adv:
                    ops.append((OP_D, adv, rp, qp))
                    lead_indel |= rp == 0
                else:
                    zero_indel = True
                rp += adv
            total += adv
            adv = 0
        evc_ok = col[2] == ctg and mq >= evc_min_mq and cigar != b"*" and not (1.0 - soft / (total + 1.0) < 0.55)
        in_region = True
        if pile_region is not None:
            span = 0                      # bam_cigar2rlen: M D N = X; an alignment without reference length occupies one base
            adv = 0
            for ch in cigar:
                if 48 <= ch <= 57:
                    adv = adv * 10 + (ch - 48)
                    continue
                if chr(ch) in "MDN=X":
                    span += adv
                adv = 0
            end1 = pos1 + max(span, 1) - 1
            in_region = col[2] == ctg and pos1 <= pile_region[1] and end1 >= pile_region[0]
        pile_ok, flush = False, False
        if in_region and mq >= pile_min_mq:
            if prev_pos != pos:
                prev_pos, depth_cap = pos, 0
                pile_ok = True
            else:
                depth_cap += 1
                pile_ok = depth_cap < dcov
            flush = pile_ok and depth_cap == 0
        if not (evc_ok or pile_ok):
            continue
        if last_pos is not None and pos < last_pos:
            anomalies |= A_UNSORTED
        last_pos = pos
        if zero_indel and evc_ok:
            anomalies |= A_ZERO_INDEL
        if evc_ok:
            # ExtractVariantCandidates.py:345 flushes every position < POS after EACH accepted alignment: an I / D tallied at POS - 1
            # (:326-336, the cursor still at POS) by a later alignment of the same POS is evaluated on its own there
            if lead_indel and evc_last_pos == pos:
                anomalies |= A_LEAD_INDEL
            evc_last_pos = pos
        if rp > len(seq) + LOOKAHEAD - 64 or rp > 0x7fffff00 or qp > 0x7fffff00:
            anomalies |= A_LONG_SPAN
        r = len(pos0)
        pos0.append(pos)
        flags.append((F_STRAND if flag & 16 else 0) | (F_EVC if evc_ok else 0) | (F_PILE if pile_ok else 0) | (F_FLUSH if flush else 0))
        seq0.append(seq_at)
        seq_len.append(len(seq))
        seq_parts.append(seq)
        seq_at += len(seq)
        op0.append(len(code))
        n_ops.append(len(ops))
        for c, n, a, b in ops:
            code.append(c)
            length.append(n)
            ref_off.append(a)
            q_off.append(b)
            op_read.append(r)
    length = np.array(length, np.int64)
    return {
        "pos0": np.array(pos0, np.int64), "flags": np.array(flags, np.uint32), "seq0": np.array(seq0, np.int64),
        "seq_len": np.array(seq_len, np.int64), "op0": np.array(op0, np.int64), "n_ops": np.array(n_ops, np.int64),
        "op_code": np.array(code, np.uint8), "op_len": length, "op_ref": np.array(ref_off, np.int64), "op_q": np.array(q_off, np.int64),
        "op_read": np.array(op_read, np.int64), "op_elem": np.concatenate([[0], np.cumsum(length)]).astype(np.int64),
        "seq": np.frombuffer(b"".join(seq_parts), np.uint8).copy(), "anomalies": anomalies, "lines": n_lines,
    }


class Columns(object):
    """The device computation on NumPy arrays.  span = [lo, hi) 0-based reference positions the tables cover."""

    def __init__(self, ref, ref0, lo, hi):
        self.ref = np.frombuffer(ref.encode("latin-1") if isinstance(ref, str) else bytes(ref), np.uint8)
        self.ref0, self.lo, self.hi = int(ref0), int(lo), int(hi)
        n = self.hi - self.lo
        self.ev = np.zeros((n, 7), np.int64)          # A C G T I D N
        self.q = np.zeros((n, 8), np.int64)           # read-base row + strand, M
        self.mw = np.zeros((n, 2), np.int64)          # M per strand
        self.dw = np.zeros((n, 2), np.int64)          # D per strand, rp > POS
        self.d0 = np.zeros(n, np.int64)               # D at rp == POS (opens windows, counts nowhere)
        self.iw = np.zeros(n, np.int64)               # inserted bases at rp, rp > POS
        self.start_m = np.zeros(n, np.int64)          # reads whose M covers their own POS
        self.anomalies = 0
        self.slabs = []

    def ref_row(self, p):
        """row of the reference base at 0-based positions p (255: outside the sequence or not an IUPAC code)."""
        i = np.asarray(p, np.int64) - self.ref0
        ok = (i >= 0) & (i < len(self.ref))
        out = np.full(i.shape, 255, np.uint8)
        out[ok] = PILE_ROW[self.ref[i[ok]]]
        return out

    def _elements(self, s):
        n = s["op_len"]
        e_op = np.repeat(np.arange(len(n)), n)
        k = np.arange(int(n.sum())) - np.repeat(s["op_elem"][:-1], n)
        code = s["op_code"][e_op]
        read = s["op_read"][e_op]
        rp = s["pos0"][read] + s["op_ref"][e_op] + np.where(code == OP_I, 0, k)
        qp = s["op_q"][e_op] + np.where(code == OP_D, 0, k)
        return e_op, k, code, read, rp, qp

    def add_reads(self, s):
        """first pass over a slab of packed reads: the tables."""
        self.anomalies |= s["anomalies"]
        self.slabs.append(s)
        e_op, k, code, read, rp, qp = self._elements(s)
        fl = s["flags"][read]
        evc, pile, so = (fl & F_EVC) != 0, (fl & F_PILE) != 0, (fl & F_STRAND).astype(np.int64)
        pos0 = s["pos0"][read]
        has_base = code != OP_D
        over = has_base & (qp >= s["seq_len"][read]) & ((code == OP_M) | pile)
        if over.any():
            self.anomalies |= A_SEQ_OVERRUN
        base = np.zeros(len(rp), np.uint8)
        okb = has_base & ~over
        base[okb] = s["seq"][(s["seq0"][read] + qp)[okb]]
        inside = (rp >= self.lo) & (rp < self.hi)
        t = rp - self.lo
        # M
        m = (code == OP_M) & ~over
        if ((EVC_IDX[base] == 255) & m & (evc | pile)).any():
            self.anomalies |= A_BAD_BASE
        good = m & inside & (EVC_IDX[base] != 255)
        np.add.at(self.ev, (t[good & evc], EVC_IDX[base[good & evc]]), 1)
        g = good & pile
        np.add.at(self.q, (t[g], PILE_ROW[base[g]].astype(np.int64) + 4 * so[g]), 1)
        np.add.at(self.mw, (t[g], so[g]), 1)
        np.add.at(self.start_m, t[g & (rp == pos0)], 1)
        # I: once per operation for the candidate search, per base for the windows
        i = code == OP_I
        first = i & (k == 0) & evc & (rp - 1 >= self.lo) & (rp - 1 < self.hi)
        np.add.at(self.ev, (t[first] - 1, 4), 1)
        gi = i & pile & (rp > pos0) & ~over
        if ((PILE_ROW[base] == 255) & gi).any():
            self.anomalies |= A_BAD_BASE
        np.add.at(self.iw, t[gi & inside], 1)
        # D
        d = code == OP_D
        first = d & (k == 0) & evc & (rp - 1 >= self.lo) & (rp - 1 < self.hi)
        np.add.at(self.ev, (t[first] - 1, 5), 1)
        gd = d & pile & inside
        np.add.at(self.dw, (t[gd & (rp > pos0)], so[gd & (rp > pos0)]), 1)
        np.add.at(self.d0, t[gd & (rp == pos0)], 1)
        # the reference base under every base a window could see
        walked = (m | d) & pile & inside
        if (self.ref_row(rp[walked]) == 255).any():
            self.anomalies |= A_BAD_REF

    def candidates(self, min_depth=4.0, min_af=0.125, ctg_range=None, bed=None, ref_evc=None, ref0_evc=None):
        """ExtractVariantCandidates.py:347-393 over the tallies -> 1-based positions, ascending."""
        ref = self.ref if ref_evc is None else np.frombuffer(ref_evc.encode("latin-1") if isinstance(ref_evc, str) else bytes(ref_evc), np.uint8)
        ref0 = self.ref0 if ref0_evc is None else int(ref0_evc)
        t = self.ev
        p0 = np.nonzero(t.any(axis=1))[0] + self.lo
        if ctg_range is not None:
            p0 = p0[(p0 + 1 >= ctg_range[0]) & (p0 + 1 <= ctg_range[1])]
        if bed is not None:
            starts, ends = bed
            j = np.searchsorted(starts, p0, side="right") - 1
            p0 = p0[(j >= 0) & (p0 < ends[np.maximum(j, 0)])]
        i = p0 - ref0
        i = np.where(i < 0, i + len(ref), i)
        ok = (i >= 0) & (i < len(ref))
        p0, i = p0[ok], i[ok]
        rb = EVC_IDX[ref[i]]
        ok = rb != 255
        p0, rb = p0[ok], rb[ok].astype(np.int64)
        n = t[p0 - self.lo]
        depth = n.sum(axis=1) - n[:, 4] - n[:, 5]
        order = np.argsort(-n, axis=1, kind="stable")
        second = np.take_along_axis(n, order[:, 1:2], axis=1)[:, 0]
        denom = np.where(depth > 0, depth, 1)
        keep = (depth.astype(np.float64) >= min_depth) & ((order[:, 0] != rb) | (second.astype(np.float64) / denom.astype(np.float64) >= min_af))
        return (p0[keep] + 1).astype(np.int64)

    def windows(self, cands, min_cov=0, slots=SLOTS, left_edge=True):
        """second pass + assembly.  cands: 1-based, strictly ascending.  -> dict(centres, refseq [n,34] uint8, counts int32 [n,33,8,4],
        tuples per read (per slab), per-candidate totals, anomalies).

        left_edge=False (--stop_consider_left_edge, CreateTensor.py:103-104): a read opens the window of centre c only by walking
        position c - 17 itself, so a read that STARTS inside a window (c - 17 < POS) adds nothing to it.  The position tables hold such
        reads too: what they added -- only the first 32 walked positions of a read can be "late" for some window -- is collected per
        window and taken out again."""
        cands = np.asarray(cands, np.int64)
        if len(cands) > 1 and not (np.diff(cands) > 0).all():
            self.anomalies |= A_CANDIDATES
            return None
        lo, hi, n_t = self.lo, self.hi, self.hi - self.lo
        # cpre[i] = candidates with position <= lo + i - 1 (so a count over [a, b] is cpre[b - lo + 1] - cpre[a - lo])
        inside = cands[(cands >= lo) & (cands < hi)]
        cpre = np.concatenate([[0], np.cumsum(np.bincount(inside - lo, minlength=n_t))]).astype(np.int64)

        def n_between(a, b):       # candidates c with a <= c <= b (c is 1-based, compared as a plain integer with the bounds)
            a = np.clip(a, lo, hi)
            b = np.clip(b + 1, lo, hi)
            return np.where(b > a, cpre[b - lo] - cpre[a - lo], 0)

        ins = np.zeros((len(cands), N_POS, 8), np.int64)
        late_q = np.zeros((len(cands), N_POS, 8), np.int64)          # left_edge=False: what reads starting inside a window put into the tables
        late_mw = np.zeros((len(cands), N_POS, 2), np.int64)
        late_dw = np.zeros((len(cands), N_POS, 2), np.int64)
        diff = np.zeros(len(cands) + 1, np.int64)                    # left_edge=False: tuples per window as range additions over the candidate index
        tuples = []
        for s in self.slabs:
            e_op, k, code, read, rp, qp = self._elements(s)
            fl = s["flags"][read]
            pile, so = (fl & F_PILE) != 0, (fl & F_STRAND).astype(np.int64)
            pos0 = s["pos0"][read]
            m = (code == OP_M) & pile
            other = (code != OP_M) & pile & (rp > pos0)
            nc = np.zeros(len(rp), np.int64)
            first = np.where(rp > pos0, rp - 17, pos0 - 16) if left_edge else np.maximum(rp - 17, pos0 + 17)
            nc[m] = n_between(first[m], rp[m] + 17)
            nc[other] = n_between(first[other], rp[other] + 16)
            tuples.append(np.bincount(read, weights=nc, minlength=len(s["pos0"])).astype(np.int64))
            if not left_edge:
                sel = np.nonzero(nc > 0)[0]
                a = np.searchsorted(cands, first[sel])
                np.add.at(diff, a, 1)
                np.add.at(diff, a + nc[sel], -1)
                for e in np.nonzero(pile & (rp - pos0 <= 31) & ((code == OP_M) | ((code == OP_D) & (rp > pos0))))[0]:
                    a, b = np.searchsorted(cands, [rp[e] - 15, pos0[e] + 17])
                    if a >= b:
                        continue
                    ci = np.arange(a, b)
                    col = rp[e] - cands[a:b] + 17
                    if code[e] == OP_M:
                        if qp[e] < s["seq_len"][read[e]] and PILE_ROW[s["seq"][s["seq0"][read[e]] + qp[e]]] != 255:
                            late_q[ci, col, PILE_ROW[s["seq"][s["seq0"][read[e]] + qp[e]]] + 4 * so[e]] += 1
                            late_mw[ci, col, so[e]] += 1
                    else:
                        late_dw[ci, col, so[e]] += 1
            # insertion bases into the windows of centres rp-15 .. rp+16
            gi = np.nonzero((code == OP_I) & pile & (rp > pos0) & (qp < s["seq_len"][read]))[0]
            for e in gi:
                a, b = np.searchsorted(cands, [rp[e] - 15 if left_edge else max(rp[e] - 15, pos0[e] + 17), rp[e] + 17])
                if a == b:
                    continue
                row = PILE_ROW[s["seq"][s["seq0"][read[e]] + qp[e]]]
                if row == 255:
                    continue
                c = cands[a:b]
                col = np.minimum(rp[e] - c + 17 + k[e], N_POS - 1)
                ins[np.arange(a, b), col, row + 4 * so[e]] += 1

        def tab(table, p):         # table value at 0-based positions p, 0 outside the span
            p = np.asarray(p, np.int64)
            ok = (p >= lo) & (p < hi)
            out = np.zeros(p.shape + table.shape[1:], table.dtype)
            out[ok] = table[p[ok] - lo]
            return out

        idx = np.arange(N_POS)
        rp = cands[:, None] - 17 + idx[None, :]                       # [n,33]
        q = tab(self.q, rp) - late_q                                  # [n,33,8]
        mw, dw = tab(self.mw, rp) - late_mw, tab(self.dw, rp) - late_dw   # [n,33,2]
        rrow = self.ref_row(rp)                                       # [n,33]
        counts = np.zeros((len(cands), N_POS, 8, 4), np.int64)
        counts[..., 1] = q + ins
        counts[..., 3] = q
        for so in (0, 1):
            for b in range(4):
                hit = rrow == b
                counts[:, :, 4 * so + b, 0] = np.where(hit, mw[:, :, so], 0)
                counts[:, :, 4 * so + b, 2] = np.where(hit, mw[:, :, so] + np.where(idx[None, :] >= 1, dw[:, :, so], 0), 0)
        if counts.size and counts.max() > 32767:
            self.anomalies |= A_OVERFLOW
        walk = self.mw.sum(axis=1) + self.dw.sum(axis=1) + self.d0
        wpre = np.concatenate([[0], np.cumsum(walk)])
        mpre = np.concatenate([[0], np.cumsum(self.mw.sum(axis=1))])
        dipre = np.concatenate([[0], np.cumsum(self.dw.sum(axis=1) + self.iw)])

        def between(pre, a, b):    # sum of a table over 0-based positions a..b inclusive
            a = np.clip(a, lo, hi)
            b = np.clip(b + 1, lo, hi)
            return np.where(b > a, pre[b - lo] - pre[a - lo], 0)

        if left_edge:
            opened = between(wpre, cands - 17, cands + 16) > 0
            totals = between(mpre, cands - 17, cands + 17) - tab(self.start_m, cands + 17) + between(dipre, cands - 16, cands + 17)
        else:
            opened = tab(walk, cands - 17) > 0
            totals = np.cumsum(diff)[:len(cands)]
        depth_centre = tab(self.mw, cands - 1).sum(axis=1) - late_mw[:, 16, :].sum(axis=1)
        nrp = cands - self.ref0
        a = np.minimum(np.maximum(nrp - 17, 0), len(self.ref))
        b = np.minimum(np.maximum(nrp + 16, 0), len(self.ref))
        refseq = np.zeros((len(cands), 34), np.uint8)
        for i in range(len(cands)):
            if nrp[i] - 17 >= 0:
                refseq[i, :max(0, b[i] - a[i])] = self.ref[a[i]:b[i]]
        keep = opened & (nrp - 17 >= 0) & (depth_centre >= min_cov)
        centre_ok = PILE_ROW[refseq[:, 16]] != 255                    # clair/utils.py:90-91 (applied by the caller of the builder)
        # the budget, replayed (CreateTensor.py:283-287, 369-373)
        if budget_binds(self.slabs, tuples, cands, np.where(opened, totals, 0), slots):
            self.anomalies |= A_BUDGET
        return {"centres": cands[keep], "refseq": refseq[keep], "counts": counts[keep].astype(np.int32), "centre_ok": centre_ok[keep],
                "tuples": tuples, "totals": totals, "opened": opened, "keep": keep}


def budget_binds(slabs, tuples, cands, totals, slots=SLOTS):
    """True when the reference's count of free tuple slots would have reached zero at any point of the run: its results then depend
    on the order in which bases were offered.  Reads in stream order; a read with F_FLUSH releases the windows with centre + 17 < POS."""
    free = int(slots)
    ci = 0
    for s, t in zip(slabs, tuples):
        for r in range(len(s["pos0"])):
            if not s["flags"][r] & F_PILE:
                continue
            free -= int(t[r])
            if free < 1:
                return True
            if s["flags"][r] & F_FLUSH:
                pos = int(s["pos0"][r])
                while ci < len(cands) and cands[ci] + 17 < pos:
                    free += int(totals[ci])
                    ci += 1
    return False
