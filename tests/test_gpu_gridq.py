"""Piece B of the free variant over FLUX BINS (csrc/lf_gridbound.h, lf_free.h; option "grid_shortcut", default on for
separable grids) against the same kernel's sum over the S^2 lattice points and against the oracle.
Reference: lumfuncmcmc.py:373-377 (trapz of TrueLumFunc x Omega x dV/dz over logL and z, per field)."""
import numpy as np
import pytest

from lf_testlib import O, compare_rows, make_inputs, synth

pytestmark = pytest.mark.gpu


def _rows(n, seed, corners=True):
    """walkers all over the prior box of the completeness parameters, its corners included"""
    th = synth.walkers("free", n, seed=seed)
    rng = np.random.default_rng(seed + 1)
    th[:, 3:8] = rng.uniform(1.0, 6.0, (n, 5))
    th[:, 8] = rng.uniform(1.0, 7.0, n)
    if corners:
        for i, (a, f) in enumerate(((1.0, 1.0), (7.0, 1.0), (1.0, 6.0), (7.0, 6.0))):
            th[i, 3:8] = f
            th[i, 8] = a
        th[4, 3:8] = (1.0, 6.0, 1.0, 6.0, 3.3)
    return th


@pytest.mark.parametrize("n,rows", [(1000000, 128), (100003, 96), (20011, 40)])
def test_bins_equal_the_lattice(n, rows):
    from lumfuncmcmc_amd.capi import LFContext
    inp = make_inputs("free", n, seed=131)
    ctx = LFContext(inp)
    ctx.set_option("persistent", 2)                # lf_free whatever the size
    th = _rows(rows, 132)
    th[7, 0] = 40.2                                # underflow zone: -inf
    th[9, 1] = 6.0                                 # outside the prior
    a1, b1 = ctx.lnprob_pieces(th)
    lp1 = ctx.lnprob_batch(th)                     # the one-launch form
    ll = ctx.last_launch()
    assert ll["kernel"].startswith("lf_free") and ll["fused"] and ll["chunks_b"] <= 32, ll      # bins, not 160 node chunks
    ctx.set_option("grid_shortcut", 0)
    a0, b0 = ctx.lnprob_pieces(th)
    lp0 = ctx.lnprob_batch(th)
    assert ctx.last_launch()["chunks_b"] == (101 * 101 + 63) // 64
    ctx.close()
    assert np.array_equal(np.isnan(b1), np.isnan(b0)) and np.array_equal(np.isinf(lp1), np.isinf(lp0))
    assert np.isinf(lp1[7]) and np.isinf(lp1[9])
    fin = np.isfinite(b0)
    # piece A does not depend on the option - up to the order of its partial sums when the launch has another geometry
    # (fewer grid chunks: fewer workgroups for a small catalogue)
    np.testing.assert_allclose(a1[fin], a0[fin], rtol=2e-15)
    # proven: 1e-15 of piece B in exact arithmetic; in binary64 both sums also carry the rounding of their nodes' log flux
    # (values near -17 are 1.8e-15 apart, and d ln F / d log f reaches ~100 at the faint end)
    rel = np.abs(b1[fin] - b0[fin]) / np.abs(b0[fin])
    print("piece B, bins vs lattice, n=%d: worst rel %.2e (median %.2e)" % (n, rel.max(), np.median(rel)))
    assert rel.max() <= 2e-13
    fin = np.isfinite(lp0)
    np.testing.assert_allclose(lp1[fin], lp0[fin], rtol=1e-13)
    # ... and against the oracle (the reference's own arithmetic restated)
    nref = 24 if n <= 200000 else 8                # (the oracle takes ~0.5 s per row at 10^6 sources)
    with np.errstate(all="ignore"):
        ref, _, refB = O.lnprob_batch(inp, th[:nref], pieces=True)
    compare_rows(lp1[:nref], ref, inp, th[:nref], 1e-12)
    ok = np.isfinite(refB) & np.isfinite(b1[:nref])
    assert ok.sum() >= nref - 2
    np.testing.assert_allclose(b1[:nref][ok], refB[ok], rtol=1e-12)


def test_source_shards_split_the_bins():
    """grid_share: rank r integrates the bins c with c % parts == r; the parts add up to the whole"""
    from lumfuncmcmc_amd.capi import LFContext
    inp = make_inputs("free", 50021, seed=141)
    ctx = LFContext(inp)
    ctx.set_option("persistent", 2)
    th = _rows(24, 142)
    _, bfull = ctx.lnprob_pieces(th)
    tot = np.zeros_like(bfull)
    for part in range(3):
        ctx.set_option("grid_share", part + 65536 * 3)
        tot += ctx.lnprob_pieces(th)[1]
    ctx.close()
    np.testing.assert_allclose(tot, bfull, rtol=1e-14)


def test_a_grid_that_is_not_separable_keeps_the_lattice():
    from lumfuncmcmc_amd.capi import LFContext
    inp = make_inputs("free", 30011, seed=151)
    logL = inp["logL"].copy()
    logL[:, 50:] += 1e-3 * np.linspace(1, 0, logL.shape[0])[:, None]       # columns with their own luminosity nodes (min_comp_frac > 0)
    inp["logL"] = logL
    ctx = LFContext(inp)
    ctx.set_option("persistent", 2)
    th = _rows(16, 152)
    lp = ctx.lnprob_batch(th)
    assert ctx.last_launch()["chunks_b"] == (101 * 101 + 63) // 64
    ctx.close()
    with np.errstate(all="ignore"):
        ref = O.lnprob_batch(inp, th)
    compare_rows(lp, ref, inp, th, 1e-12)


def test_a_box_without_proven_bins_keeps_the_lattice():
    from lumfuncmcmc_amd.capi import LFContext
    inp = make_inputs("free", 30011, seed=161)
    inp["lims"] = dict(inp["lims"])
    inp["lims"]["alpha"] = [0.0, 7.0]              # alpha_C down to 0: no bound, no bins
    ctx = LFContext(inp)
    ctx.set_option("persistent", 2)
    th = _rows(16, 162)
    lp = ctx.lnprob_batch(th)
    assert ctx.last_launch()["chunks_b"] == (101 * 101 + 63) // 64
    ctx.close()
    with np.errstate(all="ignore"):
        ref = O.lnprob_batch(inp, th)
    compare_rows(lp, ref, inp, th, 1e-12)


@pytest.mark.parametrize("n,rows", [(1000000, 128), (1000000, 300), (100003, 40), (1000, 16)])
def test_the_dealt_chunks_equal_the_arithmetic_deal(n, rows, monkeypatch):
    """lf_free deals its cell chunks and flux bins to the virtual workgroups by a host-made table (lfmcmc.hip: ensure_deal;
    lf_free.h: DEAL_*), or arithmetically (LF_NO_DEAL: the A/B switch): the same chunks, each exactly once, in another order
    of summation - and with the table, too, one row at a time gives the batch's bits"""
    from lumfuncmcmc_amd.capi import LFContext
    inp = make_inputs("free", n, seed=141)
    ctx = LFContext(inp, max_batch=max(rows, 64))
    ctx.set_option("persistent", 2)
    th = _rows(rows, 142)
    th[7, 0] = 40.2
    th[9, 1] = 6.0
    a1, b1 = ctx.lnprob_pieces(th)
    lp1 = ctx.lnprob_batch(th)
    one = np.array([ctx.lnprob_batch(th[i:i + 1])[0] for i in range(0, rows, 7)])
    monkeypatch.setenv("LF_NO_DEAL", "1")
    a0, b0 = ctx.lnprob_pieces(th)
    lp0 = ctx.lnprob_batch(th)
    monkeypatch.delenv("LF_NO_DEAL")
    lp2 = ctx.lnprob_batch(th)                     # back on the table
    ctx.close()
    np.testing.assert_array_equal(lp2, lp1)
    np.testing.assert_array_equal(one, lp1[::7])
    assert np.array_equal(np.isinf(lp1), np.isinf(lp0))
    fin = np.isfinite(lp0)
    np.testing.assert_allclose(lp1[fin], lp0[fin], rtol=2e-15)
    ok = np.isfinite(a0) & np.isfinite(a1)
    np.testing.assert_allclose(a1[ok], a0[ok], rtol=2e-15)
    np.testing.assert_allclose(b1[ok], b0[ok], rtol=2e-15)
