"""lf_free's deal of cell chunks and flux bins to its 32 virtual workgroups (csrc/lfmcmc.hip: make_deal, behind lf_deal_table;
DESIGN.md section 3.4c) - host logic, no GPU: every chunk and every bin exactly once, whatever their numbers; the loads as the
cost model wants them; a source-sharded rank's foreign bins cost nothing."""
import numpy as np
import pytest

from lumfuncmcmc_amd import capi



@pytest.mark.parametrize("nc,nb", [(55, 17), (0, 17), (55, 0), (16, 16), (1, 1), (440, 6), (3, 40), (0, 0)])
def test_every_chunk_and_bin_exactly_once(nc, nb):
    cells, bins = capi.deal_table(nc, nb)
    assert len(cells) == 32 and len(bins) == 32
    assert sorted(c for r in cells for c in r) == list(range(nc))
    assert sorted(b for r in bins for b in r) == list(range(nb))
    for r in cells + bins:
        assert r == sorted(r)                       # a rank takes its items in rising order (its partial sum's order is fixed)
    # deterministic: the table depends on the two numbers only
    assert (cells, bins) == capi.deal_table(nc, nb)


def test_the_benchmark_context_is_balanced():
    """17 bins (8 units each), 55 cell chunks (3 each), the younger half of the ranks counted 8 units behind (the costs and the
    sweep they come from: lfmcmc.hip, make_deal): with that handicap counted in, no rank is more than a bin above another"""
    cells, bins = capi.deal_table(55, 17)
    load = np.array([8 * len(b) + 3 * len(c) for b, c in zip(bins, cells)])
    assert load.sum() == 17 * 8 + 55 * 3
    eff = load + np.where(np.arange(32) >= 16, 8, 0)
    assert eff.max() - eff.min() <= 8
    assert sum(len(b) for b in bins[:16]) > sum(len(b) for b in bins[16:])       # the bins go to the elders


def test_a_source_shard_counts_only_its_own_bins():
    cells, bins = capi.deal_table(55, 18, grid_part=1, grid_parts=3)
    assert sorted(b for r in bins for b in r) == list(range(18))
    own = [[b for b in r if b % 3 == 1] for r in bins]
    load = np.array([8 * len(o) + 3 * len(c) for o, c in zip(own, cells)])
    eff = load + np.where(np.arange(32) >= 16, 8, 0)
    assert eff.max() - eff.min() <= 8
    with pytest.raises(ValueError):
        capi.deal_table(5, 5, grid_part=3, grid_parts=3)
