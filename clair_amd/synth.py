"""Synthetic pileup tensors shaped like dataPrepScripts/CreateTensor.py output.

There is no network, BAM, samtools or pysam in the build environment, so benchmarks
and parity tests run on synthetic candidates.  The generator follows the counting rules
of /root/reference/dataPrepScripts/CreateTensor.py:29-65 (``generate_tensor``):

  aligned base (ref R, query Q, strand s):  ch0[R]+=1, ch1[Q]+=1, ch2[R]+=1, ch3[Q]+=1
  inserted base Q (ref '-'):                ch1[Q]+=1 at position min(p+queryAdv, 32)
  deleted base (query '-'):                 ch2[R]+=1
  row = base(ACGT) + 4*strand ; tensor[position][row][channel], integers

and the text record ``ctg pos refseq33 v0..v1055`` (CreateTensor.py:60-65).
``to_model_input`` applies the ingest transform of clair/utils.py:96-98.
"""
import numpy as np

T, ROWS, CH = 33, 8, 4
CENTER = 16
BASES = "ACGT"

# (mean depth, per-base error, insertion rate, deletion rate) per BASELINE.json config
PLATFORM_PROFILES = {
    "ont": dict(cov=50, err=0.06, ins=0.03, dele=0.04),
    "pacbio_ccs": dict(cov=30, err=0.005, ins=0.004, dele=0.004),
    "illumina": dict(cov=300, err=0.002, ins=0.0005, dele=0.0005),
}
DEPTH_CAP = 250  # CreateTensor.py:431 (--dcov)


def synthetic_candidates(n, platform="ont", seed=20250928, contig="chr20", start=100000):
    """Return (raw int32 [n,33,8,4], infos [[ctg, pos_str, seq33], ...])."""
    prof = PLATFORM_PROFILES[platform]
    rng = np.random.default_rng(seed)
    ref = rng.integers(0, 4, size=(n, T))                                   # ref base per position
    depth = np.clip(rng.poisson(prof["cov"], size=n), 4, DEPTH_CAP)         # per candidate
    # per-position depth jitters around the candidate depth
    pos_depth = np.clip(depth[:, None] - rng.poisson(1.0, size=(n, T)), 1, DEPTH_CAP)
    fwd = rng.binomial(pos_depth, 0.5)
    reads = np.stack([fwd, pos_depth - fwd], axis=2)                        # [n,T,2]

    # query-base distribution per (candidate, position): mostly the ref base
    probs = np.full((n, T, 4), prof["err"] / 3.0)
    np.put_along_axis(probs, ref[:, :, None], 1.0 - prof["err"], axis=2)
    # centre-site genotype: 0 hom-ref, 1 het SNP, 2 hom SNP, 3 het ins, 4 het del, 5 hom del
    kind = rng.choice(6, size=n, p=[0.45, 0.2, 0.1, 0.1, 0.1, 0.05])
    alt = (ref[:, CENTER] + rng.integers(1, 4, size=n)) % 4
    af = np.where(kind == 1, 0.5, np.where(kind == 2, 1.0, 0.0))
    centre = probs[:, CENTER, :] * (1.0 - af)[:, None]
    centre[np.arange(n), alt] += af
    probs[:, CENTER, :] = centre / centre.sum(axis=1, keepdims=True)

    raw = np.zeros((n, T, ROWS, CH), dtype=np.int64)
    ref_onehot = np.eye(4, dtype=np.int64)[ref]                             # [n,T,4]
    for s in range(2):
        r = reads[:, :, s]
        dele_p = np.full((n, T), prof["dele"])
        dele_p[:, CENTER + 1] = np.where(kind == 4, 0.5, np.where(kind == 5, 0.95, prof["dele"]))
        n_del = rng.binomial(r, dele_p)
        aligned = r - n_del
        q = rng.multinomial(aligned.reshape(-1), probs.reshape(-1, 4)).reshape(n, T, 4)
        rows = slice(4 * s, 4 * s + 4)
        raw[:, :, rows, 0] += ref_onehot * aligned[:, :, None]
        raw[:, :, rows, 1] += q
        raw[:, :, rows, 2] += ref_onehot * (aligned + n_del)[:, :, None]
        raw[:, :, rows, 3] += q
        ins_p = np.full((n, T), prof["ins"])
        ins_p[:, CENTER] = np.where(kind == 3, 0.5, prof["ins"])
        n_ins = rng.binomial(aligned, ins_p)
        ins_base = rng.integers(0, 4, size=(n, T))
        # inserted bases land on position p+queryAdv (CreateTensor.py:51-53); queryAdv >= 1
        for adv in (1, 2):
            share = n_ins // 2 if adv == 2 else n_ins - n_ins // 2
            tgt = np.minimum(np.arange(T) + adv, T - 1)
            for b in range(4):
                np.add.at(raw[:, :, 4 * s + b, 1], (slice(None), tgt), share * (ins_base == b))
    seqs = ["".join(BASES[b] for b in row) for row in ref]
    infos = [[contig, str(start + 7 * i), seqs[i]] for i in range(n)]
    return raw.astype(np.int32), infos


def to_model_input(raw):
    """float32 [n,33,8,4] with channels 1..3 minus channel 0 (clair/utils.py:96-98)."""
    x = raw.astype(np.float32)
    x[:, :, :, 1:] -= x[:, :, :, 0:1]
    return x


def tensor_records(raw, infos):
    """Text lines in CreateTensor's record format (CreateTensor.py:60-65)."""
    flat = raw.reshape(raw.shape[0], -1)
    for (ctg, pos, seq), row in zip(infos, flat):
        yield "%s %s %s %s" % (ctg, pos, seq, " ".join("%d" % v for v in row))


def synthetic_input(n, platform="ont", seed=20250928):
    raw, infos = synthetic_candidates(n, platform, seed)
    return to_model_input(raw), infos
