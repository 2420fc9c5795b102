"""Crafted probability vectors for the decode goldens (shared by make_ref_goldens.py and make_pysam_goldens.py)."""
import numpy as np


def peaky(rng, size, hot, mass):
    """A probability vector with `mass` on index hot and the rest spread randomly (float32, sums ~1)."""
    v = rng.random(size) ** 3
    v[hot] = 0
    v = v / v.sum() * (1.0 - mass)
    v[hot] = mass
    return v.astype(np.float32)


def crafted_probabilities(rng, n, long_bias=False):
    """Probability sets aimed at every branch of output_from (call_var.py:693-947).  long_bias: half of the indel lengths are
    16 ("16 or more": the reference then consults the BAM, call_var.py:498-524, 540-565)."""
    kinds = ["ref", "homo_snp", "hetero_snp", "homo_ins", "acgt_ins", "insins", "homo_del", "acgt_del",
             "deldel", "insdel", "random", "flat"]
    P = np.zeros((n, 90), dtype=np.float32)
    tags = []
    for i in range(n):
        kind = kinds[i % len(kinds)]
        tags.append(kind)
        mass = float(rng.choice([0.5, 0.8, 0.97, 0.9999]))
        L = lambda idx: peaky(rng, 33, 16 + idx, mass)  # noqa: E731
        la = int(rng.integers(1, 17))
        lb = int(rng.integers(1, 17))
        if long_bias:
            la = 16 if rng.random() < 0.5 else la
            lb = 16 if rng.random() < 0.3 else lb
        if kind == "ref":
            g, z, l1, l2 = peaky(rng, 21, int(rng.choice([0, 4, 7, 9])), mass), peaky(rng, 3, 0, mass), L(0), L(0)
        elif kind == "homo_snp":
            g, z, l1, l2 = peaky(rng, 21, int(rng.choice([0, 4, 7, 9])), mass), peaky(rng, 3, 1, mass), L(0), L(0)
        elif kind == "hetero_snp":
            g, z, l1, l2 = peaky(rng, 21, int(rng.choice([1, 2, 3, 5, 6, 8])), mass), peaky(rng, 3, 2, mass), L(0), L(0)
        elif kind == "homo_ins":
            g, z, l1, l2 = peaky(rng, 21, 15, mass), peaky(rng, 3, 1, mass), L(la), L(la)
        elif kind == "acgt_ins":
            g, z = peaky(rng, 21, int(rng.integers(16, 20)), mass), peaky(rng, 3, 2, mass)
            l1, l2 = (L(0), L(la)) if rng.random() < 0.5 else (L(la), L(0))
        elif kind == "insins":
            g, z, l1, l2 = peaky(rng, 21, 15, mass), peaky(rng, 3, 2, mass), L(la), L(lb)
        elif kind == "homo_del":
            g, z, l1, l2 = peaky(rng, 21, 10, mass), peaky(rng, 3, 1, mass), L(-la), L(-la)
        elif kind == "acgt_del":
            g, z = peaky(rng, 21, int(rng.integers(11, 15)), mass), peaky(rng, 3, 2, mass)
            l1, l2 = (L(0), L(-la)) if rng.random() < 0.5 else (L(-la), L(0))
        elif kind == "deldel":
            g, z, l1, l2 = peaky(rng, 21, 10, mass), peaky(rng, 3, 2, mass), L(-la), L(-lb)
        elif kind == "insdel":
            g, z = peaky(rng, 21, 20, mass), peaky(rng, 3, 2, mass)
            l1, l2 = (L(la), L(-lb)) if rng.random() < 0.5 else (L(-la), L(lb))
        elif kind == "random":
            g, z, l1, l2 = [rng.dirichlet(np.full(s, 0.3)).astype(np.float32) for s in (21, 3, 33, 33)]
        else:  # flat: many exact ties
            g, z, l1, l2 = [np.full(s, 1.0 / s, dtype=np.float32) for s in (21, 3, 33, 33)]
        P[i] = np.concatenate([g, z, l1, l2])
    P = np.minimum(P, np.float32(0.99999))  # NumPy-2 reference raises at p == 1.0f (SURVEY.md 8c caveat)
    return P, tags
