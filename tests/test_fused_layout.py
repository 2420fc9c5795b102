"""Index arithmetic of the fused layer-2 launch (clair_amd/csrc/lstm2_fused.hip.h, gemm_split.hip.h FUSED, lstm32.hip.h FUSED),
restated in Python: who writes which ticket word, who waits for it, and that writer and reader carry the same XCD.  No GPU."""
import itertools

import pytest

T_POS = 33


def producer_items(ntiles, groups, block):
    """gemm_split_body<true>: the (xt, t, pair) items of logical workgroup `block`, in order."""
    xcd, local = block & 7, block >> 3
    gtile, group = local & 3, local >> 2
    npairs = ntiles >> 1
    nq = (npairs - xcd + 7) >> 3 if xcd < npairs else 0
    n_items = T_POS * nq
    out = []
    it = 0
    while nq and group + it * groups < n_items:
        i = group + it * groups
        s, k = divmod(i, nq)
        t = T_POS - 1 - s if gtile >> 1 else s
        q = xcd + 8 * k
        out.append((t * npairs + q, t, q, s))
        it += 1
    return gtile, out


def consumer_coords(c):
    x, j = c & 7, c >> 3
    r, q = j & 3, (j >> 2) * 8 + x
    return r & 1, 2 * q + (r >> 1)          # d, tile


@pytest.mark.parametrize("ntiles,groups", [(2, 4), (16, 4), (32, 4), (32, 3), (34, 4), (64, 4), (100, 5)])
def test_every_block_has_its_eight_writers_and_its_readers_share_their_xcd(ntiles, groups):
    npairs = ntiles // 2
    producers = 32 * groups
    written = {}
    for block in range(producers):
        gtile, items = producer_items(ntiles, groups, block)
        d = gtile >> 1
        steps = [s for _, _, _, s in items]
        assert steps == sorted(steps)                                   # every workgroup walks its direction's time forwards
        for xt, t, q, s in items:
            assert xt == t * npairs + q and q % 8 == block % 8 and q < npairs
            for wave in range(4):
                for ni in range(2):                                     # the two candidate tiles of the 64-row activation tile
                    word = (((d * ntiles + 2 * q + ni) * T_POS + t) * 8) + (gtile & 1) * 4 + wave
                    assert word not in written, "ticket word written twice"
                    written[word] = block % 8
    assert len(written) == 2 * ntiles * T_POS * 8                       # every (direction, tile, t) block has all eight words
    consumers = 32 * ((npairs + 7) // 8)
    seen = set()
    for c in range(consumers):
        d, tile = consumer_coords(c)
        if tile >= ntiles:
            continue
        assert (d, tile) not in seen
        seen.add((d, tile))
        assert (producers + c) % 8 == (tile // 2) % 8                   # its logical id sits on the XCD of its pair
        for t, w in itertools.product(range(T_POS), range(4)):
            base = ((d * ntiles + tile) * T_POS + t) * 8
            for word in (base + 2 * w, base + 2 * w + 1):               # lstm32_body: the 8-byte word pair of wave w
                assert written[word] == (tile // 2) % 8
            # ... and those two words belong to the producer that holds this wave's gate rows: gate tile 2d + (w >> 1)
            assert (2 * w) // 4 == w >> 1
    assert seen == {(d, tile) for d in range(2) for tile in range(ntiles)}
