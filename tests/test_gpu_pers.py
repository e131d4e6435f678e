"""lf_pers (csrc/lf_pers.h): the z-evolving and fixed-completeness lnprob in persistent workgroups, one launch per plain
evaluation - against lf_main (three launches, option persistent = 0), against its own three-launch form (bitwise) and against
the oracle.  Reference: lumfuncmcmc_z.py:364-392 (z-evolving), lumfuncmcmc.py:380-393, :411-424 (fixed completeness)."""
import numpy as np
import pytest

from lf_testlib import O, compare_rows, make_inputs, synth

pytestmark = pytest.mark.gpu


def _rows(variant, B, seed):
    th = synth.walkers(variant, B, seed=seed)
    if B >= 8:
        th[1, 0] = 40.2                            # underflow zone: -inf (NEGINF)
        th[2, 3 if variant == "zevol" else 1] = 6.0     # outside the prior
        # near the underflow boundary: bounds inconclusive -> the careful path over the sources
        if variant == "zevol":
            th[3, :3] = (40.75, 40.8, 40.85)
            th[4, :3] = (40.66, 40.7, 40.9)
        else:
            th[3, 0] = 40.75
            th[4, 0] = 40.66
    return th


@pytest.mark.parametrize("variant,n,B,zslices", [("zevol", 1000000, 128, 0), ("zevol", 800000, 256, 8), ("zevol", 100003, 130, 0),
                                                 ("zevol", 100003, 9, 0), ("fixcomp", 1000000, 128, 0), ("fixcomp", 5000, 597, 0),
                                                 ("fixcomp", 1000, 16, 0)])
def test_persistent_kernel_equals_lf_main(variant, n, B, zslices):
    from lumfuncmcmc_amd.capi import LFContext
    inp = make_inputs(variant, n, seed=171, zslices=zslices)
    ctx = LFContext(inp, max_batch=max(B, 64))
    th = _rows(variant, B, 172)
    lp1 = ctx.lnprob_batch(th)
    ll = ctx.last_launch()
    assert ll["kernel"] == "lf_pers" and ll["fused"], ll
    a1, b1 = ctx.lnprob_pieces(th)                 # (the diagnostics take three launches around the same kernel)
    assert ctx.last_launch()["kernel"] == "lf_pers" and not ctx.last_launch()["fused"]
    lp2 = ctx.lnprob_batch(th[::-1].copy())[::-1]  # the tile counters come back to zero; another tile for every walker
    ctx.set_option("fuse", 0)
    lp3 = ctx.lnprob_batch(th)
    assert not ctx.last_launch()["fused"]
    ctx.set_option("fuse", 1)
    ctx.set_option("persistent", 0)
    lp0 = ctx.lnprob_batch(th)
    assert ctx.last_launch()["kernel"] == "lf_main"
    a0, b0 = ctx.lnprob_pieces(th)
    ctx.close()
    assert not np.isnan(lp1).any()
    assert np.array_equal(np.isinf(lp1), np.isinf(lp0)) and np.array_equal(np.isnan(a1), np.isnan(a0))
    if B >= 8:
        assert np.isinf(lp1[1]) and np.isinf(lp1[2])
    np.testing.assert_array_equal(lp3, lp1)        # one launch = three launches: same slots, same order, same bits
    fin = np.isfinite(lp0)
    np.testing.assert_allclose(lp2[fin], lp1[fin], rtol=2e-15)      # (a row's bits depend on its place in its tile's launch geometry only)
    np.testing.assert_allclose(lp1[fin], lp0[fin], rtol=1e-13)
    ok = np.isfinite(a0) & np.isfinite(a1)
    np.testing.assert_allclose(a1[ok], a0[ok], rtol=1e-13)
    np.testing.assert_allclose(b1[ok], b0[ok], rtol=1e-13)
    nref = min(B, 10 if n <= 200000 else 6)
    with np.errstate(all="ignore"):
        ref = O.lnprob_batch(inp, th[:nref])
    compare_rows(lp1[:nref], ref, inp, th[:nref], 1e-12)


def test_careful_path_walkers_share_tiles_with_walkers_on_the_cells():
    """rows in the band where the bounds cannot rule out an underflow, scattered over the tiles: summed over the sources by
    the tile's workgroups together, -inf where the reference's product underflows"""
    from lumfuncmcmc_amd.capi import LFContext
    inp = make_inputs("zevol", 200003, seed=181)
    B = 64
    th = synth.walkers("zevol", B, seed=182)
    ls = np.linspace(40.55, 41.0, 16)
    for i, l in enumerate(ls):
        th[4 * i + 1, :3] = (l, l + 0.02, l + 0.05)
    with np.errstate(all="ignore"):
        ref = O.lnprob_batch(inp, th)
    ctx = LFContext(inp)
    got = ctx.lnprob_batch(th)
    assert ctx.last_launch()["kernel"] == "lf_pers"
    ctx.close()
    assert np.isinf(ref).any() and np.isfinite(ref[1::4]).any()
    compare_rows(got, ref, inp, th, 1e-12)
    # fixed completeness: the careful path only decides about -inf
    inp = make_inputs("fixcomp", 20000, seed=183)
    th = np.array([[l, -2.0, -1.49] for l in np.linspace(40.0, 41.2, 49)])
    with np.errstate(all="ignore"):
        ref = O.lnprob_batch(inp, th)
    ctx = LFContext(inp)
    got = ctx.lnprob_batch(th)
    assert ctx.last_launch()["kernel"] == "lf_pers"
    ctx.close()
    assert np.isinf(ref).any() and np.isfinite(ref).any()
    compare_rows(got, ref, inp, th, 1e-12)


def test_source_shards_split_the_grid_in_the_same_granules_in_both_kernels():
    """grid_share: chunks of 64 nodes c with c % parts == part, whichever kernel integrates them (a rank may run lf_main
    where another runs lf_pers): the parts of either kernel add up to the whole of the other"""
    from lumfuncmcmc_amd.capi import LFContext
    for variant in ("zevol", "fixcomp"):
        inp = make_inputs(variant, 50021, seed=191)
        ctx = LFContext(inp)
        th = synth.walkers(variant, 24, seed=192)
        _, bfull = ctx.lnprob_pieces(th)
        tot = np.zeros_like(bfull)
        for part in range(3):
            ctx.set_option("grid_share", part + 65536 * 3)
            ctx.set_option("persistent", part % 2)          # alternate the kernels
            tot += ctx.lnprob_pieces(th)[1]
        ctx.close()
        np.testing.assert_allclose(tot, bfull, rtol=1e-13)
