"""Parity of the HIP path (through the C ABI) with the reference: golden vectors recorded from
the reference, the oracle on seeded inputs, size-independent properties at BASELINE sizes, and
the edge cases.  All of it needs a real MI355X: `pytest -m gpu`.

Tolerance: the kernels work in log space with hoisted factors (lf_kernels.h), the reference in
linear space; per term they agree to a few ulp, and the stated fp64 tolerance on lnprob and on
each piece is RTOL = 1e-12 relative (measured worst case is printed by each test)."""
import glob
import os

import numpy as np
import pytest

from lf_testlib import O, compare_rows, make_inputs, synth

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "*.npz"))
               if os.path.basename(f).split("_")[0] in ("free", "fixcomp", "zevol"))
RTOL = 1e-12


def ctx_of(inp, **kw):
    from lumfuncmcmc_amd.capi import LFContext
    return LFContext(inp, device=0, **kw)


@pytest.mark.parametrize("case", CASES)
def test_golden_vectors(case):
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    inp = O.inputs_from_golden(g, case.split("_")[0])
    ctx = ctx_of(inp)
    th = g["theta"]
    assert ctx.ndim == th.shape[1]
    got = ctx.lnprob_batch(th)
    w = compare_rows(got, g["lnprob"], inp, th, RTOL)
    A, B = ctx.lnprob_pieces(th)
    wa = compare_rows(A, g["A"], inp, th, RTOL, "A")
    okB = np.isfinite(g["B"])
    np.testing.assert_allclose(B[okB], g["B"][okB], rtol=RTOL, atol=0)
    assert np.array_equal(np.isnan(A), np.isnan(g["A"]))      # NaN pieces exactly where the prior fails
    # one row at a time (the reference's calling pattern) gives the same bits as the batch
    one = np.array([ctx.lnprob_batch(t)[0] for t in th[:5]])
    assert np.array_equal(one, got[:5])
    print("%s worst rel lnprob %.2e A %.2e" % (case, w, wa))
    ctx.close()


@pytest.mark.parametrize("variant,n,fsa", [("free", 20000, False), ("free", 4099, True),
                                           ("fixcomp", 20000, False), ("zevol", 20000, False),
                                           ("zevol", 7777, True)])
def test_oracle_seeded(variant, n, fsa):
    inp = make_inputs(variant, n, seed=123 + n, fix_sch_al=fsa, zslices=8 if variant == "zevol" else 0)
    th = synth.walkers(variant, 37, seed=5, fix_sch_al=fsa)       # 37: not a multiple of the walker tile
    ref, refA, refB = O.lnprob_batch(inp, th, pieces=True)
    ctx = ctx_of(inp)
    got = ctx.lnprob_batch(th)
    w = compare_rows(got, ref, inp, th, RTOL)
    A, B = ctx.lnprob_pieces(th)
    np.testing.assert_allclose(A, refA, rtol=RTOL)
    np.testing.assert_allclose(B, refB, rtol=RTOL)
    print("%s n=%d worst rel %.2e" % (variant, n, w))
    ctx.close()


def test_batch_shapes_and_growth():
    inp = make_inputs("free", 3000, seed=9)
    ctx = ctx_of(inp, max_batch=8)
    th = synth.walkers("free", 700, seed=11)
    full = ctx.lnprob_batch(th)                  # grows the workspace past max_batch
    for b in (1, 7, 8, 9, 64, 129):
        assert np.array_equal(ctx.lnprob_batch(th[:b]), full[:b])
    ref = O.lnprob_batch(inp, th[:12])
    np.testing.assert_allclose(full[:12], ref, rtol=RTOL)
    assert ctx.lnprob_batch(np.empty((0, ctx.ndim))).shape == (0,)
    with pytest.raises(ValueError):
        ctx.lnprob_batch(np.zeros((3, ctx.ndim + 1)))
    ctx.close()


def test_prior_and_nan_rows():
    inp = make_inputs("free", 1000, seed=2)
    ctx = ctx_of(inp)
    th = synth.walkers("free", 9, seed=3)
    th[1, 0] = 45.0 + 1e-9        # outside
    th[2, 8] = np.nan             # NaN never passes a comparison -> -inf, never NaN out
    th[3, 3] = 1.0                # inclusive edge stays finite
    th[4, :] = np.inf
    out = ctx.lnprob_batch(th)
    assert out[1] == -np.inf and out[2] == -np.inf and out[4] == -np.inf
    assert np.isfinite(out[[0, 3, 5, 6, 7, 8]]).all()
    assert not np.isnan(out).any()
    allbad = np.full((5, ctx.ndim), 99.0)
    assert (ctx.lnprob_batch(allbad) == -np.inf).all()
    ctx.close()


def test_underflow_convention():
    """log(product) = -inf as soon as one source's linear-space product is 0 (App. B-5)."""
    inp = make_inputs("fixcomp", 5000, seed=4)
    ctx = ctx_of(inp)
    th = np.array([[ls, -2.0, -1.49] for ls in np.linspace(40.0, 41.2, 49)])
    ref = O.lnprob_batch(inp, th)
    got = ctx.lnprob_batch(th)
    assert np.isinf(ref).any() and np.isfinite(ref).any()
    compare_rows(got, ref, inp, th, RTOL)
    ctx.close()


def test_ragged_and_empty_fields():
    inp = make_inputs("free", 2000, seed=6)
    inp["field_ind"] = np.array([0, 0, 3, 1200, 1200, 2000], dtype=np.int64)   # two empty fields, one of 3
    th = synth.walkers("free", 10, seed=7)
    ref = O.lnprob_batch(inp, th)
    ctx = ctx_of(inp)
    np.testing.assert_allclose(ctx.lnprob_batch(th), ref, rtol=RTOL)
    ctx.close()
    for n in (1, 5):
        inp = make_inputs("zevol", n, seed=8)
        th = synth.walkers("zevol", 4, seed=7)
        ctx = ctx_of(inp)
        np.testing.assert_allclose(ctx.lnprob_batch(th), O.lnprob_batch(inp, th), rtol=RTOL)
        ctx.close()
    # empty catalogue: lnlike = -integral
    inp = make_inputs("fixcomp", 0, seed=8)
    th = synth.walkers("fixcomp", 4, seed=7)
    ctx = ctx_of(inp)
    A, B = ctx.lnprob_pieces(th)
    assert (A == 0).all()
    np.testing.assert_allclose(ctx.lnprob_batch(th), -B, rtol=0)
    np.testing.assert_allclose(B, [O.piece_b(inp, O.split_theta(inp, t)) for t in th], rtol=RTOL)
    ctx.close()


def test_more_fields_and_other_grid_sizes():
    inp = make_inputs("free", 4000, seed=10, nf=8, S=37)
    th = synth.walkers("free", 6, seed=7, nf=8)
    ctx = ctx_of(inp)
    assert ctx.ndim == 12
    np.testing.assert_allclose(ctx.lnprob_batch(th), O.lnprob_batch(inp, th), rtol=RTOL)
    ctx.close()
    inp = make_inputs("fixcomp", 900, seed=10, nf=1, S=64)
    th = synth.walkers("fixcomp", 6, seed=7)
    ctx = ctx_of(inp)
    np.testing.assert_allclose(ctx.lnprob_batch(th), O.lnprob_batch(inp, th), rtol=RTOL)
    ctx.close()


def test_chunking_does_not_change_results():
    inp = make_inputs("free", 50000, seed=12)
    th = synth.walkers("free", 16, seed=13)
    ctx = ctx_of(inp)
    base = ctx.lnprob_batch(th)
    assert np.array_equal(base, ctx.lnprob_batch(th))            # bitwise reproducible
    for gi in (0, 1, 2, 3, 4, 5, 6, 7, 8, -1):
        ctx.set_option("geometry", gi)
        np.testing.assert_allclose(ctx.lnprob_batch(th), base, rtol=1e-14)
    ctx.close()


@pytest.mark.parametrize("variant,B", [("free", 37), ("free", 128), ("zevol", 131), ("fixcomp", 64)])
def test_tapered_tiling_is_bitwise_neutral(variant, B):
    # the walker tiling only decides which workgroup owns a (chunk, walker) pair, never the order
    # of a sum: quarter-size tail tiles on or off must give the same bits, for ragged B as well
    inp = make_inputs(variant, 30000, seed=21)
    th = synth.walkers(variant, B, seed=22)
    ctx = ctx_of(inp)
    ctx.set_option("persistent", 0)          # (lf_main's tiling is the subject: lf_free has none, and would serve this size)
    for gi in (-1, 0, 2):
        ctx.set_option("geometry", gi)
        ctx.set_option("taper", 1)
        on = ctx.lnprob_batch(th)
        ctx.set_option("taper", 0)
        off = ctx.lnprob_batch(th)
        assert np.array_equal(on, off, equal_nan=True)
    np.testing.assert_allclose(on, O.lnprob_batch(inp, th), rtol=RTOL)
    ctx.close()


def test_device_pointer_entry_matches_host_entry():
    import torch
    inp = make_inputs("zevol", 10000, seed=14, zslices=8)
    th = synth.walkers("zevol", 33, seed=15)
    ctx = ctx_of(inp)
    host = ctx.lnprob_batch(th)
    dth = torch.from_numpy(th).cuda()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        out = ctx.lnprob_torch(dth)
    side.synchronize()
    assert np.array_equal(out.cpu().numpy(), host)
    ctx.close()


@pytest.mark.parametrize("variant,n,W", [("free", 1000000, 256), ("free", 1000000, 512), ("fixcomp", 1000000, 256),
                                         ("zevol", 800000, 512)])        # (free, 512 walkers = BASELINE config 3)
def test_full_size_properties(variant, n, W):
    """BASELINE sizes, where the scalar oracle is too slow to sweep: size-independent properties.
    (1) phi* shift: A moves by N ln10 d, B scales by 10^d.  (2) catalogue additivity: A over the
    whole catalogue = sum of A over two halves (different launch geometry).  (3) a handful of rows
    against the oracle.  (4) FIXCOMP: the closed form of SURVEY App. A.4."""
    inp = make_inputs(variant, n, zslices=8 if variant == "zevol" else 0)
    B = W // 2
    th = synth.walkers(variant, B, seed=1)
    ctx = ctx_of(inp)
    A, Bi = ctx.lnprob_pieces(th)
    lnp = ctx.lnprob_batch(th)
    assert np.isfinite(lnp).all()
    np.testing.assert_allclose(lnp, A - Bi, rtol=1e-15)
    d = 0.37
    th2 = th.copy()
    if variant == "zevol":
        th2[:, 3:6] += d
    else:
        th2[:, 1] += d
    A2, B2 = ctx.lnprob_pieces(th2)
    np.testing.assert_allclose(A2 - A, n * np.log(10.0) * d, rtol=1e-11)
    np.testing.assert_allclose(B2 / Bi, 10.0 ** d, rtol=1e-12)
    ref = O.lnprob_batch(inp, th[:3])
    np.testing.assert_allclose(lnp[:3], ref, rtol=RTOL)
    ctx.close()
    # additivity over a split of every field in two
    fi = inp["field_ind"]
    halves = []
    for part in (0, 1):
        sel = np.concatenate([np.arange(fi[f], fi[f + 1])[part::2] for f in range(len(fi) - 1)])
        sub = dict(inp)
        for k in ("lum", "z", "DLz", "Om_arr"):
            sub[k] = inp[k][sel]
        sub["field_ind"] = np.concatenate([[0], np.cumsum([len(np.arange(fi[f], fi[f + 1])[part::2])
                                                            for f in range(len(fi) - 1)])]).astype(np.int64)
        c2 = ctx_of(sub)
        halves.append(c2.lnprob_pieces(th)[0])
        c2.close()
    np.testing.assert_allclose(halves[0] + halves[1], A, rtol=1e-13)
    if variant == "fixcomp":
        lum = inp["lum"]
        c1 = np.log(10.0) * (th[:, 2] + 1)
        cf = (n * (np.log(np.log(10.0)) + np.log(10.0) * th[:, 1]) + c1 * (lum.sum() - n * th[:, 0])
              - 10.0 ** (-th[:, 0]) * np.sum(10.0 ** lum) + np.log(inp["Om_arr"]).sum())
        np.testing.assert_allclose(A, cf, rtol=1e-12)


def test_faint_source_exercises_all_three_modes():
    """One source far below the flux limit: depending on (Flim, alpha_C) a walker is FAST (bounds
    clear), SLOW (bounds cannot exclude underflow: per-term checks decide) or really -inf."""
    inp = make_inputs("free", 3000, seed=21)
    inp["lum"] = inp["lum"].copy()
    inp["lum"][5] = 39.75                      # ~1.9 dex below the 50% completeness flux of field 0
    th = synth.walkers("free", 96, seed=22)
    th[:, 3] = np.linspace(1.2, 5.9, 96)       # Flim of field 0
    th[::3, 8] = 6.9                           # steep completeness: the faint source underflows
    ref = O.lnprob_batch(inp, th)
    assert np.isinf(ref).any() and np.isfinite(ref).sum() > 20
    ctx = ctx_of(inp)
    w = compare_rows(ctx.lnprob_batch(th), ref, inp, th, RTOL)
    print("faint-source worst rel %.2e, -inf rows %d" % (w, np.isinf(ref).sum()))
    ctx.close()


def test_class_surface_free_fit_model():
    """BASELINE config 1 shape: 1k sources, 32 walkers, 50 steps, through the class surface the
    reference drivers use (constructor keywords of run_lumfuncmcmc.py:245-256)."""
    from lumfuncmcmc_amd.model import LumFuncMCMC
    cat = synth.catalogue(1000, seed=0)
    fi = cat["field_ind"]
    o = LumFuncMCMC(synth.split_fields(cat["z"], fi), flux=None, flux_e=None,
                    lum=synth.split_fields(cat["lum"], fi), lum_e=synth.split_fields(cat["lum_e"], fi),
                    Flim=list(synth.FLIM), alpha=synth.ALPHA_C, line_name="OIII",
                    line_plot_name=r'[OIII] $\lambda 5007$', Omega_0=list(synth.OMEGA_0), nbins=50, nboot=100,
                    sch_al=synth.SCH_AL, sch_al_lims=synth.SCH_AL_LIMS, Lstar=synth.LSTAR,
                    Lstar_lims=synth.LSTAR_LIMS, phistar=synth.PHISTAR, phistar_lims=synth.PHISTAR_LIMS,
                    Lc=synth.LC, Lh=synth.LH, nwalkers=32, nsteps=50, fix_sch_al=False, fix_comp=False,
                    min_comp_frac=0.0, Flim_lims=synth.FLIM_LIMS, alpha_lims=synth.ALPHA_LIMS,
                    field_names=np.array(["AEGIS", "COSMOS", "GOODSN", "GOODSS", "UDS"]), field_ind=fi,
                    diff_rand=True)
    th0 = np.array([42.5, -2.0, -1.49, 2.72, 3.61, 2.55, 3.31, 3.30, 4.56])
    v = o.lnprob(th0)
    assert isinstance(v, float) and abs(v - (-46587.950223002365)) < 1e-12 * 46588 * 10
    g = np.load(os.path.join(GOLDEN, "free_n1000.npz"))
    got = o.lnprob(g["theta"])                                     # (B, ndim) in one device call
    inp = O.inputs_from_golden(g, "free")
    compare_rows(got, g["lnprob"], inp, g["theta"], RTOL)
    with pytest.raises(ValueError):
        o.lnprob_fix_comp(th0[:3])
    np.random.seed(5)
    o.fit_model()
    assert o.samples.shape[1] == 10 and o.samples.shape[0] % 32 == 0 and o.samples.shape[0] >= 32 * 25
    assert o.chain.shape == (32, 50, 9)
    fin = o.samples[np.isfinite(o.samples[:, -1])]
    chk = o.lnprob(fin[:5, :-1])
    np.testing.assert_allclose(chk, fin[:5, -1], rtol=1e-13)      # stored lnprob is what the path returns
    assert len(o.get_param_names()) == 9 and o.get_init_walker_values().shape == (32, 9)
    o.close()


def test_class_surface_fixcomp_and_z():
    from lumfuncmcmc_amd.model import LumFuncMCMC, LumFuncMCMCz
    g = np.load(os.path.join(GOLDEN, "fixcomp_n1000.npz"))
    fi = g["field_ind"]
    kw = dict(lum=synth.split_fields(g["lum"], fi), lum_e=synth.split_fields(g["lum_e"], fi),
              Flim=list(synth.FLIM), alpha=synth.ALPHA_C, Omega_0=list(synth.OMEGA_0), sch_al=synth.SCH_AL,
              sch_al_lims=synth.SCH_AL_LIMS, Lstar=synth.LSTAR, Lstar_lims=synth.LSTAR_LIMS,
              phistar=synth.PHISTAR, phistar_lims=synth.PHISTAR_LIMS, Lc=synth.LC, Lh=synth.LH, nwalkers=32,
              nsteps=20, min_comp_frac=0.0, field_ind=fi)
    o = LumFuncMCMC(synth.split_fields(g["z"], fi), fix_comp=True, Flim_lims=synth.FLIM_LIMS,
                    alpha_lims=synth.ALPHA_LIMS, **kw)
    inp = O.inputs_from_golden(g, "fixcomp")
    compare_rows(o.lnprob_fix_comp(g["theta"]), g["lnprob"], inp, g["theta"], RTOL)
    o.close()
    g = np.load(os.path.join(GOLDEN, "zevol_n1000.npz"))
    oz = LumFuncMCMCz(synth.split_fields(g["z"], fi), **kw)
    inp = O.inputs_from_golden(g, "zevol")
    compare_rows(oz.lnprob(g["theta"]), g["lnprob"], inp, g["theta"], RTOL)
    np.random.seed(1)
    oz.fit_model()
    assert oz.samples.shape[1] == 8
    oz.close()


@pytest.mark.parametrize("name", ["e2e_free_n100000", "e2e_free_n1000000", "e2e_fixcomp_n1000000", "e2e_zevol_n800000",
                                  "e2e_free_fsa_n100000", "e2e_free_mcf50_n100000"])
def test_end_to_end_against_the_reference_at_baseline_sizes(name):
    """BASELINE sizes, the reference itself as the oracle: the fixture holds only the generator arguments, 48 theta
    rows and the lnprob the reference returned (oracle/gen_golden.py --only e2e): 32 rows of the finite box, the
    underflow zone, the prior's edges (on them and 1e-9 outside), and - free variant, one source far below the flux
    limit - walkers on the careful path.  Here the catalogue is regenerated, the build's own host setup (cosmology,
    tables, splines) makes the kernel inputs, and the HIP path evaluates them: setup and kernels are compared with
    the reference together.  Also with fixed Schechter alpha, and with min_comp_frac = 0.5 (grid aliasing quirk)."""
    from lf_testlib import e2e_compare, e2e_model
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    o = e2e_model(g)
    worst, ninf = e2e_compare(o, g, RTOL)
    print("%s worst rel %.2e (%d rows -inf)" % (name, worst, ninf))
    assert ninf >= 4
    o.close()


@pytest.mark.parametrize("n,nf,zslices,pivots", [(400000, 5, 8, (1.20, 1.53, 1.86)), (250001, 3, 0, (1.18, 1.36, 1.54)),
                                                 (33333, 4, 0, (1.20, 1.76, 2.32))])
def test_zevol_local_form_equals_the_per_source_form(n, nf, zslices, pivots):
    """The z-evolving term with ONE exponential per (walker, lane of z-neighbours) (lf_kernels.h: srcsum_body, local
    form) against the per-source exponential of the same kernel ("specialise" = 0) and against the oracle: walkers over
    the whole prior box of L1..L3 (steep, curved L*(z)), ragged last chunks and lanes (n is not a multiple of anything),
    a catalogue small enough that wide lanes send some walkers back to the per-source form."""
    from lumfuncmcmc_amd.capi import LFContext
    inp = make_inputs("zevol", n, seed=31, nf=nf, zslices=zslices, pivots=pivots)
    th = synth.walkers("zevol", 40, seed=32, nf=nf)
    rng = np.random.default_rng(33)
    th[8:, 0:3] = rng.uniform(41.2, 44.8, (32, 3))
    ref = O.lnprob_batch(inp, th[:10])
    ctx = LFContext(inp)
    ctx.set_option("cells", 0)                               # (per source: the cells in redshift have their own test below)
    a1, b1 = ctx.lnprob_pieces(th)
    lp1 = ctx.lnprob_batch(th)
    ctx.set_option("specialise", 0)
    a0, b0 = ctx.lnprob_pieces(th)
    lp0 = ctx.lnprob_batch(th)
    ctx.close()
    assert np.array_equal(np.isinf(lp1), np.isinf(lp0)) and not np.isnan(lp1).any()
    fin = np.isfinite(lp0)
    np.testing.assert_allclose(a1[fin], a0[fin], rtol=2e-15)
    np.testing.assert_array_equal(b1[fin], b0[fin])
    f10 = np.isfinite(ref)
    np.testing.assert_allclose(lp1[:10][f10], ref[f10], rtol=RTOL)
    if n == 400000:
        assert not np.array_equal(a1[fin], a0[fin])          # (the local form did run: only FAST-mode walkers take it)


@pytest.mark.parametrize("n,nf,zslices,pivots,rows", [(400000, 5, 8, (1.20, 1.53, 1.86), 40), (250001, 3, 0, (1.18, 1.36, 1.54), 40),
                                                      (33333, 4, 0, (1.20, 1.76, 2.32), 21), (1000000, 5, 0, None, 128)])
def test_zevol_cells_equal_the_sum_over_sources(n, nf, zslices, pivots, rows):
    """The z-evolving variant summed over the catalogue's cells in REDSHIFT (lf_kernels.h: ZCELL_RHO, zcell_body) against
    the sum over the sources ("cells" = 0) and against the oracle.  Walkers over the whole prior box of L1..L3 (steep and
    curved L*(z): the cells' width was chosen for the box, so every walker inside it takes them) and beyond it (-inf
    without being evaluated), ragged chunks, redshift slices with gaps between them."""
    from lumfuncmcmc_amd.capi import LFContext
    kw = {} if pivots is None else {"pivots": pivots}
    inp = make_inputs("zevol", n, seed=41, nf=nf, zslices=zslices, **kw)
    th = synth.walkers("zevol", rows, seed=42, nf=nf)
    rng = np.random.default_rng(43)
    wild = np.arange(rows) % 4 == 3                         # every 4th walker anywhere in the box
    th[wild, 0:3] = rng.uniform(41.2, 44.8, (int(wild.sum()), 3))
    # ... and a few with L* so low that the brightest sources' exp(-10^(lum - L*)) is near underflow: inside the prior,
    # but on the careful path, i.e. summed over the SOURCES (by the workers that share the per-source items) while
    # their tile-mates take the cells
    low = np.arange(rows) % 16 == 5
    th[low, 0:3] = float(np.max(inp["lum"])) - np.log10(716.0)          # (10^(lum_max - L*) = 716: careful from 700, -inf from 745)
    ref = O.lnprob_batch(inp, th[:6]) if n <= 400000 else None
    ctx = LFContext(inp)
    ctx.set_option("count_forms", 1)
    a1, b1 = ctx.lnprob_pieces(th)
    fc = ctx.form_counts()
    ncell = fc["cell"]
    assert fc["careful"] > 0, fc                                  # (some walkers were summed over the sources, term by term)
    ctx.set_option("count_forms", 0)
    lp1 = ctx.lnprob_batch(th)
    ctx.set_option("cells", 0)
    a0, b0 = ctx.lnprob_pieces(th)
    lp0 = ctx.lnprob_batch(th)
    ctx.close()
    assert ncell > 0                                         # the cell workgroups ran
    assert np.array_equal(np.isinf(lp1), np.isinf(lp0)) and not np.isnan(lp1).any()
    fin = np.isfinite(lp0)
    assert fin.sum() >= rows // 2
    # piece A is (closed form) - sum_i v_i: the difference of the two sums of v against the sum itself (the closed form
    # is common to both and can be much larger or smaller than the sum)
    np.testing.assert_allclose(a1[fin], a0[fin], rtol=1e-13)
    np.testing.assert_array_equal(b1[fin], b0[fin])
    np.testing.assert_allclose(lp1[fin], lp0[fin], rtol=1e-13)
    assert not np.array_equal(a1[~wild & fin], a0[~wild & fin])          # (the gentle walkers did take the cells)
    if ref is not None:
        f6 = np.isfinite(ref)
        np.testing.assert_allclose(lp1[:6][f6], ref[f6], rtol=RTOL)


@pytest.mark.parametrize("sep", [True, False])
def test_fixcomp_grid_summed_over_its_rows(sep, monkeypatch):
    """Fixed completeness, every redshift column on the same luminosity nodes (what the reference's constructor makes
    with min_comp_frac = 0): the integrand is the Schechter function of the ROW times a weight, so lf_create sums the
    weights over the columns and the kernel integrates S nodes instead of S^2 - the same sums in another order.  When
    the columns differ (min_comp_frac > 0: their lower ends follow the completeness limit) the full lattice stays.
    Against the full lattice of the same build (LF_NO_COLLAPSE_GRID) and the oracle."""
    from lumfuncmcmc_amd.capi import LFContext
    inp = make_inputs("fixcomp", 30000, seed=71)
    S = int(np.asarray(inp["zarr"]).shape[0])
    if not sep:
        lo = np.asarray(inp["logL"])[0, 0]
        inp["logL"] = np.stack([np.linspace(lo + 0.3 * k / S, synth.LH, S) for k in range(S)], axis=1)
    th = synth.walkers("fixcomp", 40, seed=72)
    th[3, 0] = 40.2
    # a walker on the careful path (the brightest source's exp(-10^(lum - L*)) near underflow): the only kind whose piece A
    # is summed over the sources in this variant - by the workers that share the per-source items; its tile-mates' are not
    th[4, 0] = th[20, 0] = float(np.max(inp["lum"])) - np.log10(716.0)
    ctx = LFContext(inp)
    a1, b1 = ctx.lnprob_pieces(th)
    lp = ctx.lnprob_batch(th[:10])
    nodes1 = ctx.last_launch()["chunks_b"]
    ctx.close()
    monkeypatch.setenv("LF_NO_COLLAPSE_GRID", "1")
    ctx = LFContext(inp)
    a0, b0 = ctx.lnprob_pieces(th)
    nodes0 = ctx.last_launch()["chunks_b"]
    ctx.close()
    assert nodes0 == (S * S + 63) // 64 and nodes1 == ((S + 63) // 64 if sep else nodes0)      # (lf_pers: chunks of 64 nodes)
    np.testing.assert_array_equal(a1, a0)
    fin = np.isfinite(b0)
    np.testing.assert_allclose(b1[fin], b0[fin], rtol=1e-13)
    if not sep:
        np.testing.assert_array_equal(b1, b0)
    ref = O.lnprob_batch(inp, th[:10])
    f = np.isfinite(ref)
    assert np.array_equal(f, np.isfinite(lp))
    np.testing.assert_allclose(lp[f], ref[f], rtol=RTOL)


def test_zevol_grid_by_columns_equals_the_grid_by_nodes(monkeypatch):
    """The z-evolving grid integral with the walker's factors taken per redshift COLUMN (nodes stored column by column,
    one exponential per node) against the node-by-node form of the same build (LF_NO_ZGRID_COLS: two exponentials and two
    parabolas per node) and the oracle; walkers over the whole prior box, one outside it."""
    from lumfuncmcmc_amd.capi import LFContext
    inp = make_inputs("zevol", 60000, seed=81, zslices=8)
    th = synth.walkers("zevol", 37, seed=82)
    th[2, 3] = 9.0
    ctx = LFContext(inp)
    a1, b1 = ctx.lnprob_pieces(th)
    lp1 = ctx.lnprob_batch(th)
    ctx.close()
    monkeypatch.setenv("LF_NO_ZGRID_COLS", "1")
    ctx = LFContext(inp)
    a0, b0 = ctx.lnprob_pieces(th)
    ctx.close()
    fin = np.isfinite(b0)
    assert fin.sum() >= 30 and np.array_equal(fin, np.isfinite(b1))
    # (by columns: lf_pers, a wave per walker; by nodes: lf_main - the cells' sums in another order)
    np.testing.assert_allclose(a1[fin], a0[fin], rtol=1e-14)
    np.testing.assert_allclose(b1[fin], b0[fin], rtol=2e-14)
    assert not np.array_equal(b1[fin], b0[fin])              # (it is another route to the same numbers)
    ref = O.lnprob_batch(inp, th[:10])
    f = np.isfinite(ref)
    assert np.array_equal(f, np.isfinite(lp1[:10]))
    np.testing.assert_allclose(lp1[:10][f], ref[f], rtol=RTOL)
