"""The compressed-catalogue option (lf_set_option "compress", csrc/lf_compress.h) against the same references
as the direct path: golden vectors recorded from the reference, the oracle on seeded inputs, the reference's
own numbers at BASELINE sizes, and the direct path itself.  Same stated tolerance, RTOL = 1e-12; the measured
differences (printed) are at the 1e-15 level.  Needs a real MI355X."""
import glob
import os

import numpy as np
import pytest

from lf_testlib import O, compare_rows, make_inputs, synth

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "*.npz"))
               if os.path.basename(f).split("_")[0] in ("free", "zevol"))
RTOL = 1e-12


def ctx_of(inp, compress=True):
    from lumfuncmcmc_amd.capi import LFContext
    ctx = LFContext(inp, device=0)
    if compress:
        ctx.set_option("compress", 1)
    return ctx


@pytest.mark.parametrize("case", CASES)
def test_golden_vectors_compressed(case):
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    inp = O.inputs_from_golden(g, case.split("_")[0])
    ctx = ctx_of(inp)
    th = g["theta"]
    got = ctx.lnprob_batch(th)
    w = compare_rows(got, g["lnprob"], inp, th, RTOL)
    A, B = ctx.lnprob_pieces(th)
    wa = compare_rows(A, g["A"], inp, th, RTOL, "A")
    assert np.array_equal(np.isnan(A), np.isnan(g["A"]))
    print("%s compressed: worst rel lnprob %.2e A %.2e" % (case, w, wa))
    ctx.close()


@pytest.mark.parametrize("variant,n,fsa,B", [("free", 200000, False, 37), ("free", 4099, True, 64),
                                             ("zevol", 160000, False, 37), ("zevol", 7777, True, 131)])
def test_compressed_matches_direct_and_oracle(variant, n, fsa, B):
    inp = make_inputs(variant, n, seed=77 + n, fix_sch_al=fsa, zslices=8 if variant == "zevol" else 0)
    th = synth.walkers(variant, B, seed=6, fix_sch_al=fsa)
    ctx = ctx_of(inp, compress=False)
    direct = ctx.lnprob_batch(th)
    dA, dB = ctx.lnprob_pieces(th)
    ctx.set_option("compress", 1)
    ctx.set_option("compress_grid", 0)
    A, Bp = ctx.lnprob_pieces(th)
    # full grid: the integral is the direct one (to the summation order: the direct path of a large free-variant
    # catalogue runs in lf_free, 512-node chunks; the compressed path in lf_main, 256-node chunks)
    np.testing.assert_allclose(Bp, dB, rtol=1e-14)
    ctx.set_option("compress_grid", 1)                             # (default) FREE: the separable grid is compressed too
    got = ctx.lnprob_batch(th)
    A, Bp = ctx.lnprob_pieces(th)
    relA = np.max(np.abs(A - dA) / np.abs(dA))
    relB = np.max(np.abs(Bp - dB) / np.abs(dB))
    rel = np.max(np.abs(got - direct) / np.abs(direct))
    print("%s n=%d: compressed vs direct: lnprob %.2e piece A %.2e piece B %.2e" % (variant, n, rel, relA, relB))
    assert rel < 1e-13 and relA < 1e-13 and relB < 1e-13
    if variant == "zevol":
        # no grid compression for the z-evolving model: the same nodes (the direct path sums them in lf_pers, per wave;
        # the compressed path in lf_main, per workgroup: another order)
        np.testing.assert_allclose(Bp, dB, rtol=1e-14)
    if n <= 10000:
        compare_rows(got, O.lnprob_batch(inp, th), inp, th, RTOL)
    for gi in (1, 2, 4, 8):                                        # every instantiated geometry
        ctx.set_option("geometry", gi)
        np.testing.assert_allclose(ctx.lnprob_batch(th), got, rtol=1e-14)
    ctx.set_option("geometry", -1)
    ctx.set_option("compress", 0)                                  # and off again: the direct bits
    assert np.array_equal(ctx.lnprob_batch(th), direct)
    ctx.close()


@pytest.mark.parametrize("variant", ["free", "zevol"])
def test_walkers_that_need_the_per_source_checks_are_summed_over_the_real_catalogue(variant):
    """Rows of the underflow zone (App. B-5) and of the full prior: the compressed path must return the direct
    path's value for them - -inf where a source's product underflows, the careful sum elsewhere."""
    inp = make_inputs(variant, 20000, seed=31, zslices=8 if variant == "zevol" else 0)
    th = np.concatenate([synth.walkers(variant, 24, seed=8), _underflow_rows(variant, 40)])
    ctx = ctx_of(inp, compress=False)
    ctx.set_option("geometry", 1)                                  # the chunk size the rescue workgroups use
    direct = ctx.lnprob_batch(th)
    ctx.set_option("geometry", -1)
    ctx.set_option("compress", 1)
    got = ctx.lnprob_batch(th)
    ref = O.lnprob_batch(inp, th)
    compare_rows(got, ref, inp, th, RTOL)
    assert np.array_equal(np.isinf(got), np.isinf(direct))
    fin = np.isfinite(direct)
    np.testing.assert_allclose(got[fin], direct[fin], rtol=1e-13)
    nslow = int(np.sum(np.isinf(direct)))
    print("%s: %d of %d rows -inf, identical pattern" % (variant, nslow, len(th)))
    assert nslow > 0
    ctx.close()


def _underflow_rows(variant, n):
    """theta rows inside the prior whose brightest sources drive exp(-10^(lum - L*)) towards underflow"""
    rng = np.random.default_rng(99)
    th = synth.walkers(variant, n, seed=10)
    if variant == "zevol":
        th[:, 0:3] = rng.uniform(40.05, 41.2, (n, 3))
    else:
        th[:, 0] = rng.uniform(40.05, 41.2, n)
    return th


@pytest.mark.parametrize("name", ["e2e_free_n100000", "e2e_free_n1000000", "e2e_zevol_n800000"])
def test_end_to_end_against_the_reference_compressed(name):
    """BASELINE sizes, the reference's own lnprob as the oracle (see test_gpu_parity), compressed catalogue on."""
    from lf_testlib import e2e_compare, e2e_model
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    o = e2e_model(g)
    o.context().set_option("compress", 1)
    worst, ninf = e2e_compare(o, g, RTOL)
    print("%s compressed: worst rel vs the reference %.2e (%d rows -inf)" % (name, worst, ninf))
    o.close()


def test_device_sampler_on_the_compressed_catalogue():
    """The device-resident sampler goes through the same enqueue: its recorded lnprob must be the direct
    path's lnprob of the recorded positions."""
    from lumfuncmcmc_amd.sampler import DeviceEnsembleSampler
    inp = make_inputs("free", 50000, seed=41)
    ctx = ctx_of(inp)
    W = 32
    p0 = synth.walkers("free", W, seed=11)
    ds = DeviceEnsembleSampler(ctx, W, seed=3, capacity=12)
    ds.run_mcmc(p0, 12)
    chain, lnp = ds.chain, ds.lnprobability
    ds.close()
    ctx.set_option("compress", 0)
    last = ctx.lnprob_batch(chain[:, -1, :])
    np.testing.assert_allclose(lnp[:, -1], last, rtol=1e-13)
    assert 0.05 < ds.acceptance_fraction.mean() <= 1.0
    ctx.close()


@pytest.mark.parametrize("lims", [{"Flim": [0.5, 9.0], "alpha": [0.6, 10.0]}, {"Flim": [2.0, 4.0], "alpha": [3.0, 5.0]},
                                  {"Lstar": [41.0, 47.0], "phistar": [-9.0, 2.0]}])
def test_compress_with_a_non_default_prior_box(lims):
    """The bins are validated at sampled walkers of the context's own prior box and the finished tables must reproduce
    the direct path on 64 more walkers of that box at build time (lf_set_option "compress"): with another box the option
    is either refused (LFError) or - what is asserted then - as good as with the default one."""
    from lumfuncmcmc_amd.capi import LFError
    inp = make_inputs("free", 60000, seed=83)
    inp["lims"] = dict(inp["lims"])
    inp["lims"].update(lims)
    ctx = ctx_of(inp, compress=False)
    rng = np.random.default_rng(84)
    L = inp["lims"]
    th = synth.walkers("free", 80, seed=85)
    th[:, 3:8] = rng.uniform(L["Flim"][0], L["Flim"][1], (80, 5))
    th[:, 8] = rng.uniform(L["alpha"][0], L["alpha"][1], 80)
    th[:, 0] = rng.uniform(max(L["Lstar"][0], 41.5), L["Lstar"][1], 80)
    direct = ctx.lnprob_batch(th)
    try:
        ctx.set_option("compress", 1)
    except LFError as e:
        print("refused:", e)
        assert np.array_equal(ctx.lnprob_batch(th), direct)          # and the direct path is untouched
        ctx.close()
        return
    got = ctx.lnprob_batch(th)
    assert np.array_equal(np.isinf(got), np.isinf(direct))
    fin = np.isfinite(direct)
    assert fin.sum() > 40
    np.testing.assert_allclose(got[fin], direct[fin], rtol=1e-12)
    ctx.close()
