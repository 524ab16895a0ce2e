"""The deal of the block backward's attention passes (cr_stack_bwd1.hip b1_deal_tiles, exported for this test as cr_stack_block_bwd_deal):
every tile of a pass goes to exactly one wave, at most two per wave, and the SIMD sums (waves w and w + 4 share a SIMD) are as even as
the tile costs allow -- the quantity the wave timelines showed to set a launch's pace.  Host code: no GPU."""
import ctypes
import math

import pytest

import castrec_amd  # noqa: F401
from castrec_amd import lib as L


def deal(nkt, query):
    pk = (ctypes.c_uint32 * 8)()
    assert L.lib.cr_stack_block_bwd_deal(nkt, int(query), pk) == 0
    return [[(pk[w] >> (5 * i)) & 31 for i in range(2) if ((pk[w] >> (5 * i)) & 31) != 31] for w in range(8)]


def cost(nkt, query, t):
    """pair iterations of the tile's loop + 3 (what a tile costs whatever its length)"""
    return ((t + 1 if query else nkt - t) + 1) // 2 + 3


@pytest.mark.parametrize("query", [False, True])
@pytest.mark.parametrize("nkt", list(range(1, 15)))
def test_every_tile_once_and_even_simds(nkt, query):
    waves = deal(nkt, query)
    tiles = sorted(t for ws in waves for t in ws)
    assert tiles == list(range(nkt)) and all(len(ws) <= 2 for ws in waves)
    loads = [sum(cost(nkt, query, t) for t in ws) for ws in waves]
    simd = [loads[i] + loads[i + 4] for i in range(4)]
    if nkt <= 8:                                                          # the smallest possible largest SIMD sum, by brute force
        import itertools
        best = min(max(sum(cost(nkt, query, t) for t in range(nkt) if a[t] == s) for s in range(4)) for a in itertools.product(range(4), repeat=nkt))
        assert max(simd) == best, (waves, simd, best)
    else:
        assert max(simd) <= math.ceil(sum(simd) / 4) + 1, (waves, simd)
    for i in range(4):                                                    # the heavier wave of a SIMD sits in the upper half (raised priority)
        if waves[i] and waves[i + 4]:
            assert loads[i] <= loads[i + 4], (i, waves)
    for ws in waves:                                                       # a wave's heavier tile goes first
        if len(ws) == 2:
            assert cost(nkt, query, ws[0]) >= cost(nkt, query, ws[1])


def test_the_headline_deal_is_perfectly_even():
    """13 tiles (T = 200): SIMD sums 22 / 22 / 22 / 22 in these units (19 / 19 / 19 / 18 pair iterations + 2 per tile), where the greedy deal of
    rounds 3-4 left 19 / 18 / 18 / 20: 0.3163 -> 0.3125 ms per step (tools/probes/deal_ab.sh)."""
    for query in (False, True):
        waves = deal(13, query)
        loads = [sum(cost(13, query, t) for t in ws) for ws in waves]
        assert [loads[i] + loads[i + 4] for i in range(4)] == [22, 22, 22, 22], waves
    assert L.lib.cr_stack_block_bwd_deal(0, 0, (ctypes.c_uint32 * 8)()) != 0 and L.lib.cr_stack_block_bwd_deal(15, 1, (ctypes.c_uint32 * 8)()) != 0
