"""Seeded random configurations (sizes, field layouts, grid sides, batch sizes, variants, forced launch
geometries) against the oracle: the ragged and degenerate shapes nobody writes by hand."""
import numpy as np
import pytest

from lf_testlib import O, compare_rows, make_inputs, synth

pytestmark = pytest.mark.gpu
RTOL = 1e-12


@pytest.mark.parametrize("seed", range(24))
def test_random_configuration(seed):
    from lumfuncmcmc_amd.capi import LFContext
    rng = np.random.default_rng(1000 + seed)
    variant = ("free", "fixcomp", "zevol")[seed % 3]
    nf = int(rng.integers(1, 9))
    n = int(rng.choice([1, 2, 7, 63, 255, 256, 257, 511, 2047, 2048, 2049, 4500]))
    S = int(rng.integers(4, 40))
    fsa = bool(rng.integers(0, 2))
    inp = make_inputs(variant, n, seed=seed, S=S, fix_sch_al=fsa, nf=nf,
                      pivots=(1.18, 1.36, 1.54) if seed % 2 else (1.20, 1.53, 1.86))
    # ragged fields: random cut points, some fields empty
    cuts = np.sort(rng.integers(0, n + 1, nf - 1)) if nf > 1 else np.array([], dtype=int)
    inp["field_ind"] = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    if variant != "free":
        # Om_arr was built for the equal split: rebuild it for the ragged one
        om0 = np.zeros(n, dtype=int); fl = np.zeros(n)
        for f in range(nf):
            om0[inp["field_ind"][f]:inp["field_ind"][f + 1]] = inp["Omega_0"][f]
            fl[inp["field_ind"][f]:inp["field_ind"][f + 1]] = inp["Flim0"][f]
        with np.errstate(all="ignore"):
            inp["Om_arr"] = O.omega(inp["lum"], inp["DLz"], om0, 1.0e-17 * fl, synth.ALPHA_C, synth.FCMIN)
    B = int(rng.integers(1, 70))
    th = synth.walkers(variant, B, seed=seed + 7, fix_sch_al=fsa, nf=nf)
    # a few rows outside the box / in the underflow zone / NaN
    if B > 3:
        th[0, 0] = 40.2
        th[1, 1] = 5.5
        th[2, -1] = np.nan
    ref = O.lnprob_batch(inp, th)
    ctx = LFContext(inp)
    for gi in (-1, int(rng.integers(0, 9))):
        ctx.set_option("geometry", gi)
        got = ctx.lnprob_batch(th)
        compare_rows(got, ref, inp, th, RTOL, "seed %d geo %d" % (seed, gi))
    # the compressed catalogue on the same ragged shapes (small fields stay uncompressed, n = 4500 does not)
    ctx.set_option("geometry", -1)
    ctx.set_option("compress", 1)
    compare_rows(ctx.lnprob_batch(th), ref, inp, th, RTOL, "seed %d compressed" % seed)
    ctx.close()


@pytest.mark.parametrize("seed", range(16))
def test_random_configuration_persistent_kernel(seed):
    """The same ragged shapes through lf_free (persistent workgroups, table-driven term; forced: it normally serves
    catalogues of 4e5 sources and more), every sources-per-lane instantiation, against the oracle."""
    from lumfuncmcmc_amd.capi import LFContext
    rng = np.random.default_rng(3000 + seed)
    nf = int(rng.integers(1, 9))
    n = int(rng.choice([1, 2, 7, 63, 511, 513, 2049, 4095, 4096, 4097, 9000, 33000]))
    S = int(rng.integers(4, 40))
    fsa = bool(rng.integers(0, 2))
    inp = make_inputs("free", n, seed=100 + seed, S=S, fix_sch_al=fsa, nf=nf)
    cuts = np.sort(rng.integers(0, n + 1, nf - 1)) if nf > 1 else np.array([], dtype=int)      # ragged fields, some empty
    inp["field_ind"] = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    if seed % 4 == 1:
        # steep number counts with a sparse bright tail: lanes too wide for the tables -> the general forms of pass 2
        lum = inp["lum"]
        inp["lum"] = np.minimum(lum.min() + rng.exponential(0.3, n), 43.5)
        inp["lum"][0] = lum.min()
    B = int(rng.choice([1, 7, 8, 9, 33, 64, 70, 130]))
    th = synth.walkers("free", B, seed=seed + 11, fix_sch_al=fsa, nf=nf)
    if B > 3:
        th[0, 0] = 40.2                     # underflow zone
        th[1, 1] = 5.5                      # outside the prior
        th[2, -1] = np.nan
    ref = O.lnprob_batch(inp, th)
    ctx = LFContext(inp)
    ctx.set_option("persistent", 2)
    for st in (0, int(rng.choice([2, 4]))):
        ctx.set_option("free_st", st)
        got = ctx.lnprob_batch(th)
        assert ctx.last_launch()["kernel"].startswith("lf_free"), ctx.last_launch()
        compare_rows(got, ref, inp, th, RTOL, "seed %d st %d" % (seed, st))
    ctx.close()


@pytest.mark.parametrize("seed", range(6))
def test_random_large_catalogue_compressed(seed):
    """Catalogues big enough for every bin to be compressed, with skewed source distributions: compressed
    against direct (the oracle is too slow at these sizes; direct is pinned to it above)."""
    from lumfuncmcmc_amd.capi import LFContext
    rng = np.random.default_rng(2000 + seed)
    variant = ("free", "zevol")[seed % 2]
    n = int(rng.integers(30000, 120000))
    nf = int(rng.integers(1, 9))
    inp = make_inputs(variant, n, seed=seed, nf=nf, fix_sch_al=bool(seed & 2), S=int(rng.integers(8, 30)),
                      zslices=(8 if variant == "zevol" and nf == 5 else 0),
                      pivots=(1.18, 1.36, 1.54) if seed % 3 == 0 else (1.20, 1.53, 1.86))
    if variant == "free":
        # steep number counts: most sources near the faint end, a bright tail
        lum = inp["lum"]
        inp["lum"] = np.minimum(lum.min() + rng.exponential(0.25, n), 43.5)
        inp["lum"][0] = lum.min()                     # the grid's lower edge stays where make_inputs put it
    B = int(rng.integers(20, 90))
    th = synth.walkers(variant, B, seed=seed + 17, fix_sch_al=bool(seed & 2), nf=nf)
    th[0, 0] = 40.3                                   # underflow zone: rescue workgroups
    ctx = LFContext(inp)
    direct = ctx.lnprob_batch(th)
    ctx.set_option("compress", 1)
    got = ctx.lnprob_batch(th)
    assert np.array_equal(np.isinf(got), np.isinf(direct)) and not np.isnan(got).any()
    fin = np.isfinite(direct)
    rel = np.max(np.abs(got[fin] - direct[fin]) / np.abs(direct[fin]))
    print("seed %d %s n=%d nf=%d: compressed vs direct %.2e" % (seed, variant, n, nf, rel))
    assert rel < 1e-13
    ctx.close()


@pytest.mark.parametrize("seed", range(10))
def test_random_configuration_cells(seed):
    """Catalogues big enough for cells (free completeness: in log-flux, through lf_free; z-evolving: in redshift, through
    lf_main), ragged fields some of them too small for any cell, random prior-box widths (they set the cells' width),
    rows outside the prior / in the underflow zone / NaN: against the sum over the sources of the same build ("cells" = 0,
    all rows), against the oracle (the first rows), and - free completeness - one launch against three, bit for bit."""
    from lumfuncmcmc_amd.capi import LFContext
    rng = np.random.default_rng(5000 + seed)
    variant = ("free", "zevol")[seed % 2]
    nf = int(rng.integers(1, 9))
    # (the z-evolving cells are narrow - their width covers the whole prior box of L1..L3 - so they need more sources)
    n = int(rng.choice([33000, 60000, 90001, 150000] if variant == "free" else [200000, 300001, 400000]))
    inp = make_inputs(variant, n, seed=300 + seed, S=int(rng.integers(6, 40)), fix_sch_al=bool(rng.integers(0, 2)), nf=nf,
                      pivots=(1.18, 1.36, 1.54) if seed % 4 < 2 else (1.20, 1.53, 1.86))
    if variant == "free":
        cuts = np.sort(rng.integers(0, n + 1, nf - 1)) if nf > 1 else np.array([], dtype=int)
        inp["field_ind"] = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    inp["lims"] = {k: list(v) for k, v in inp["lims"].items()}
    if variant == "free" and seed % 3 == 0:
        inp["lims"]["alpha"] = [1.0, float(rng.uniform(7.0, 30.0))]        # narrower cells
    B = int(rng.choice([7, 9, 33, 130]))
    th = synth.walkers(variant, B, seed=seed + 21, fix_sch_al=inp["fix_sch_al"], nf=nf)
    if B > 3:
        th[0, 0] = 40.2
        th[1, 1] = 5.5
        th[2, -1] = np.nan
    nref = min(B, 10)
    ref = O.lnprob_batch(inp, th[:nref])
    ctx = LFContext(inp, max_batch=max(B, 64))
    if variant == "free":
        ctx.set_option("persistent", 2)
    ctx.set_option("count_forms", 1)
    ctx.lnprob_pieces(th)
    ncell = ctx.form_counts()["cell"]
    ctx.set_option("count_forms", 0)
    lp1 = ctx.lnprob_batch(th)
    fused = ctx.last_launch()["fused"]
    ctx.set_option("fuse", 0)
    lp2 = ctx.lnprob_batch(th)
    ctx.set_option("fuse", 1)
    ctx.set_option("cells", 0)
    lp0 = ctx.lnprob_batch(th)
    ctx.close()
    # (z-evolving with the close pivots: the box allows slopes so steep that a cell would hold under four sources - none made)
    # (... or free completeness with the prior box widened in alpha_C: cells as narrow as 5e-4 dex, ragged fields)
    assert ncell > 0 or (variant == "zevol" and seed % 4 < 2) or (variant == "free" and seed % 3 == 0), "seed %d: no cells" % seed
    assert fused == (variant == "free")
    np.testing.assert_array_equal(lp1, lp2)
    assert np.array_equal(np.isinf(lp1), np.isinf(lp0)) and not np.isnan(lp1).any()
    fin = np.isfinite(lp0)
    np.testing.assert_allclose(lp1[fin], lp0[fin], rtol=1e-13)
    compare_rows(lp1[:nref], ref, inp, th[:nref], RTOL, "seed %d cells" % seed)
