"""The table-driven form of the free-completeness term (lf_kernels.h: srcsum_free / table_terms) against the general
form of the same kernel and against the oracle, on catalogues dense enough for it to apply; the census of forms
(lf_form_counts) tells which form actually ran.  Reference: lumfuncmcmc.py:370 (piece A), VmaxLumFunc.py:118-127."""
import numpy as np
import pytest

from lf_testlib import O, compare_rows, make_inputs, synth

pytestmark = pytest.mark.gpu


def _rows(n, seed, wide=False):
    th = synth.walkers("free", n, seed=seed)
    if wide:                                       # the whole prior box of the completeness parameters
        rng = np.random.default_rng(seed + 1)
        th[:, 3:8] = rng.uniform(1.0, 6.0, (n, 5))
        th[:, 8] = rng.uniform(1.0, 7.0, n)
    return th


@pytest.mark.parametrize("n,wide", [(400003, False), (400003, True), (1000000, False), (60011, False)])
def test_table_form_equals_general_form(n, wide):
    from lumfuncmcmc_amd.capi import LFContext
    inp = make_inputs("free", n, seed=91)
    ctx = LFContext(inp)
    ctx.set_option("persistent", 2)                # lf_free whatever the size (auto: only when it pays)
    ctx.set_option("cells", 0)                     # this test is about the per-source forms
    th = _rows(96, 92, wide)
    th[5, 0] = 40.2                                # underflow zone: -inf
    th[6, 1] = 6.0                                 # outside the prior
    ctx.set_option("count_forms", 1)
    a1, b1 = ctx.lnprob_pieces(th)
    lp1 = ctx.lnprob_batch(th)
    fc = ctx.form_counts()
    ctx.set_option("count_forms", 0)
    ctx.set_option("tables", 0)
    a0, b0 = ctx.lnprob_pieces(th)
    lp0 = ctx.lnprob_batch(th)
    ctx.close()
    tab = fc["table"] + fc["table_noexp"]
    gen = fc["general"] + fc["general_noexp"]
    assert fc["careful"] == 0
    if n >= 400000 and not wide:
        assert tab > 0.9 * (tab + gen), fc          # the bulk of a dense catalogue takes the table form
    if n < 100000:
        assert tab > 0, fc
    assert np.array_equal(np.isnan(a1), np.isnan(a0)) and np.array_equal(np.isinf(lp1), np.isinf(lp0))
    assert np.isinf(lp1[5]) and np.isinf(lp1[6])
    fin = np.isfinite(lp0)
    # piece A = closed-form Schechter part + the completeness sum: the two forms of the latter agree to the tables'
    # error (8e-15 per term, far less in the sum) plus rounding
    np.testing.assert_allclose(a1[fin], a0[fin], rtol=5e-14)
    np.testing.assert_array_equal(b1[fin], b0[fin])     # the grid integral does not depend on the option
    np.testing.assert_allclose(lp1[fin], lp0[fin], rtol=5e-14)


@pytest.mark.parametrize("st", [2, 4, 8])
def test_persistent_kernel_equals_lf_main(st):
    """lf_free (persistent workgroups, every sources-per-lane instantiation) against lf_main on the same context."""
    from lumfuncmcmc_amd.capi import LFContext
    inp = make_inputs("free", 150001, seed=97)
    ctx = LFContext(inp)
    th = _rows(50, 98, wide=True)
    th[7, 0] = 40.2
    th[9, 8] = 0.5
    ctx.set_option("persistent", 0)
    a0, b0 = ctx.lnprob_pieces(th)
    lp0 = ctx.lnprob_batch(th)
    assert ctx.last_launch()["kernel"] == "lf_main"
    ctx.set_option("persistent", 2)
    ctx.set_option("free_st", st)
    a1, b1 = ctx.lnprob_pieces(th)
    lp1 = ctx.lnprob_batch(th)
    assert ctx.last_launch()["kernel"] == "lf_free<%d>" % st
    ctx.close()
    assert np.array_equal(np.isnan(a1), np.isnan(a0)) and np.array_equal(np.isinf(lp1), np.isinf(lp0))
    fin = np.isfinite(lp0)
    np.testing.assert_allclose(a1[fin], a0[fin], rtol=5e-14)
    np.testing.assert_allclose(b1[fin], b0[fin], rtol=1e-14)
    np.testing.assert_allclose(lp1[fin], lp0[fin], rtol=5e-14)


def test_persistent_kernel_with_more_walker_tiles_than_workgroup_groups():
    """B = 600 rows = 75 tiles of 8 walkers for 64 groups of workgroups: a group serves several tiles in turn (the
    tile_stride loop of lf_free), the last tile ragged."""
    from lumfuncmcmc_amd.capi import LFContext
    inp = make_inputs("free", 60001, seed=41)
    ctx = LFContext(inp, max_batch=600)
    th = _rows(597, 42, wide=True)
    th[5, 0] = 40.2
    ctx.set_option("persistent", 0)
    lp0 = ctx.lnprob_batch(th)
    ctx.set_option("persistent", 2)
    lp1 = ctx.lnprob_batch(th)
    assert ctx.last_launch()["kernel"].startswith("lf_free") and ctx.last_launch()["rows"] == 597
    ctx.close()
    assert np.array_equal(np.isinf(lp1), np.isinf(lp0)) and not np.isnan(lp1).any()
    fin = np.isfinite(lp0)
    np.testing.assert_allclose(lp1[fin], lp0[fin], rtol=5e-14)


def test_table_form_against_the_oracle():
    from lumfuncmcmc_amd.capi import LFContext
    inp = make_inputs("free", 300007, seed=93)
    ctx = LFContext(inp)
    ctx.set_option("persistent", 2)
    ctx.set_option("cells", 0)
    th = _rows(6, 94)
    ctx.set_option("count_forms", 1)
    a, b = ctx.lnprob_pieces(th)
    got = ctx.lnprob_batch(th)
    fc = ctx.form_counts()
    ctx.close()
    assert fc["table"] + fc["table_noexp"] > 0.8 * 2 * 6 * 300007, fc      # two calls of 6 rows
    ref = O.lnprob_batch(inp, th)
    compare_rows(got, ref, inp, th, 1e-12)
    pa = np.array([O.lnprob(inp, t, pieces=True)[1] for t in th[:3]])
    np.testing.assert_allclose(a[:3], pa, rtol=1e-12)


def test_census_adds_up():
    from lumfuncmcmc_amd.capi import LFContext
    inp = make_inputs("free", 250000, seed=95)
    ctx = LFContext(inp)
    ctx.set_option("persistent", 2)
    ctx.set_option("cells", 0)
    th = _rows(40, 96, wide=True)
    th[3, 8] = 9.0                                 # outside the prior: its terms are counted as skipped
    ctx.set_option("count_forms", 1)
    ctx.lnprob_batch(th)
    fc = ctx.form_counts()
    nbins = ctx.last_launch()["chunks_b"]
    ctx.set_option("grid_shortcut", 0)
    ctx.set_option("count_forms", 1)
    ctx.lnprob_batch(th)
    fc0 = ctx.form_counts()
    ctx.close()
    src = sum(fc[k] for k in ("general", "general_noexp", "table", "table_noexp", "careful", "skipped"))
    assert src == 40 * 250000, fc
    assert fc["skipped"] == 250000
    # piece B: 64 nodes per flux bin (lf_gridbound.h) x 5 fields, or the lattice's 101 x 101 points x 5 fields
    assert nbins <= 32 and fc["node_general"] + fc["node_bright"] == 39 * nbins * 64 * 5, fc
    assert fc0["node_general"] + fc0["node_bright"] == 39 * 101 * 101 * 5, fc0


@pytest.mark.parametrize("n,B,cells", [(60001, 597, 1), (60001, 597, 0), (400003, 130, 1), (400003, 9, 0), (1000000, 128, 1)])
def test_one_launch_form_equals_the_three_launch_form(n, B, cells):
    """lf_free doing lf_prepare's and lf_finalize's work itself (lf_free.h: FUSED - the tile's walkers prepared by every
    workgroup of the tile, the partial sums written through to memory and added up by the last workgroup to finish the
    tile) against the three launches: the same partial sums in the same slots added in the same order, so the same
    bits.  Several tiles per group of workgroups, ragged tiles, -inf rows and rows outside the prior, walkers on the
    cells and on the sources, twice in a row (the tile counters must come back to zero)."""
    from lumfuncmcmc_amd.capi import LFContext
    inp = make_inputs("free", n, seed=61)
    ctx = LFContext(inp, max_batch=max(B, 64))
    ctx.set_option("persistent", 2)
    ctx.set_option("cells", cells)
    th = _rows(B, 62, wide=True)
    th[min(5, B - 1), 0] = 40.2
    th[min(7, B - 1), 1] = 6.0
    ctx.set_option("fuse", 0)
    lp0 = ctx.lnprob_batch(th)
    assert not ctx.last_launch()["fused"]
    ctx.set_option("fuse", 1)
    lp1 = ctx.lnprob_batch(th)
    assert ctx.last_launch()["fused"] and ctx.last_launch()["kernel"].startswith("lf_free")
    lp2 = ctx.lnprob_batch(th[::-1].copy())[::-1]
    lp3 = ctx.lnprob_batch(th)
    a, b = ctx.lnprob_pieces(th)                        # (the diagnostics keep the three launches)
    assert not ctx.last_launch()["fused"]
    ctx.close()
    assert not np.isnan(lp0).any() and np.isinf(lp0).sum() >= 2
    np.testing.assert_array_equal(lp1, lp0)
    np.testing.assert_array_equal(lp3, lp0)
    fin = np.isfinite(lp0)
    np.testing.assert_allclose(lp2[fin], lp0[fin], rtol=1e-13)      # (other tiles, other slots: another order of the sums)
    assert np.array_equal(np.isinf(lp2), np.isinf(lp0))


# ---------------------------------------------------------------------------------------------- cells
@pytest.mark.parametrize("n,nf,B", [(400003, 5, 96), (1000000, 5, 130), (60011, 3, 33), (33000, 8, 9), (6000, 8, 9)])
def test_cells_equal_the_sum_over_sources(n, nf, B):
    """lf_free summing walkers over the catalogue's cells (midpoint + power sums of a narrow flux interval, lf_kernels.h:
    CELL_M) against the same kernel summing them over the sources ("cells" = 0) and against the oracle: walkers over the
    whole prior box of the completeness parameters, -inf rows, a row outside the prior, ragged tiles."""
    from lumfuncmcmc_amd.capi import LFContext
    inp = make_inputs("free", n, seed=51, nf=nf)
    ctx = LFContext(inp, max_batch=max(B, 64))
    ctx.set_option("persistent", 2)
    th = synth.walkers("free", B, seed=52, nf=nf)
    rng = np.random.default_rng(53)
    k = B // 2
    th[k:, 3:3 + nf] = rng.uniform(1.0, 6.0, (B - k, nf))
    th[k:, 3 + nf] = rng.uniform(1.0, 7.0, B - k)
    th[1, 0] = 40.2                                # underflow zone: -inf
    th[2, 1] = 6.0                                 # outside the prior
    ctx.set_option("count_forms", 1)
    a1, b1 = ctx.lnprob_pieces(th)
    lp1 = ctx.lnprob_batch(th)
    fc = ctx.form_counts()
    ctx.set_option("count_forms", 0)
    ctx.set_option("cells", 0)
    a0, b0 = ctx.lnprob_pieces(th)
    lp0 = ctx.lnprob_batch(th)
    ctx.close()
    assert fc["cell"] > 0, fc                                               # (the cells did run)
    if n >= 33000:
        assert fc["cell"] < 0.2 * (B - 2) * n * 2, fc                       # ... and are far fewer than the sources
    # (750 sources per field: under four per cell - a small catalogue gets its cells all the same, they beat the per-source path's overhead)
    assert np.array_equal(np.isinf(lp1), np.isinf(lp0)) and not np.isnan(lp1).any()
    assert np.isinf(lp1[1]) and np.isinf(lp1[2])
    fin = np.isfinite(lp0)
    np.testing.assert_allclose(a1[fin], a0[fin], rtol=2e-14)                # the orders dropped: < 2e-18 per term
    np.testing.assert_array_equal(b1[fin], b0[fin])
    ref = O.lnprob_batch(inp, th[:8])
    compare_rows(lp1[:8], ref, inp, th[:8], 1e-12)


def test_cells_are_not_used_when_they_cannot_be():
    """A non-finite flux in the catalogue, or a prior box so wide in alpha_C that a cell would be a source: no cells, every
    walker over the sources, same results as ever."""
    from lumfuncmcmc_amd.capi import LFContext
    inp = make_inputs("free", 120000, seed=55)
    inp["lims"] = {k: list(v) for k, v in inp["lims"].items()}
    inp["lims"]["alpha"] = [1.0, 4000.0]
    th = _rows(16, 56)
    ref = O.lnprob_batch(inp, th)
    ctx = LFContext(inp)
    ctx.set_option("persistent", 2)
    ctx.set_option("count_forms", 1)
    got = ctx.lnprob_batch(th)
    fc = ctx.form_counts()
    ctx.close()
    assert fc["cell"] == 0 and fc["table"] + fc["table_noexp"] > 0, fc
    compare_rows(got, ref, inp, th, 1e-12)


@pytest.mark.parametrize("cells", [1, 0])
def test_source_shards_through_the_persistent_kernel_add_up(cells):
    """Source-sharded ranks (dist.SourceShardedLnProb) in one process: three contexts hold a third of every field's
    sources each and a third of the grid chunks ("grid_share"); their lnprob values add up to the unsharded one - through
    lf_free, with and without cells (the grid must be integrated once, not once per shard)."""
    from lumfuncmcmc_amd.capi import LFContext
    from lumfuncmcmc_amd.dist import shard_sources
    inp = make_inputs("free", 240000, seed=61)
    th = _rows(24, 62)
    th[2, 0] = 40.2
    full = LFContext(inp)
    full.set_option("persistent", 2)
    full.set_option("cells", cells)
    ref = full.lnprob_batch(th)
    full.close()
    tot = np.zeros(len(th))
    for part in range(3):
        ctx = LFContext(shard_sources(inp, part, 3))
        ctx.set_option("persistent", 2)
        ctx.set_option("cells", cells)
        ctx.set_option("grid_share", part + 65536 * 3)
        tot = tot + ctx.lnprob_batch(th)
        assert ctx.last_launch()["kernel"].startswith("lf_free")
        ctx.close()
    assert np.array_equal(np.isinf(tot), np.isinf(ref))
    fin = np.isfinite(ref)
    np.testing.assert_allclose(tot[fin], ref[fin], rtol=1e-13)
