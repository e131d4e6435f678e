"""1/Veff estimator (lumfuncmcmc_amd/veff.py) against LumFuncMCMC.VeffLF of the reference, recorded by
oracle/gen_golden.py with a seeded global NumPy state (the bootstrap draws come from np.random)."""
import os

import numpy as np
import pytest

from lumfuncmcmc_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def build(g):
    from lumfuncmcmc_amd.model import LumFuncMCMC
    fi = g["field_ind"]
    return LumFuncMCMC(synth.split_fields(g["z"], fi), lum=synth.split_fields(g["lum"], fi),
                       lum_e=synth.split_fields(g["lum_e"], fi), Flim=list(synth.FLIM), alpha=synth.ALPHA_C,
                       Omega_0=list(synth.OMEGA_0), sch_al=synth.SCH_AL, sch_al_lims=synth.SCH_AL_LIMS,
                       Lstar=synth.LSTAR, Lstar_lims=synth.LSTAR_LIMS, phistar=synth.PHISTAR,
                       phistar_lims=synth.PHISTAR_LIMS, Lc=synth.LC, Lh=synth.LH, nwalkers=32, nsteps=50,
                       fix_comp=False, min_comp_frac=float(g["min_comp_frac"]), Flim_lims=synth.FLIM_LIMS,
                       alpha_lims=synth.ALPHA_LIMS, field_ind=fi, nboot=int(g["nboot"]), nbins=int(g["nbins"]))


def test_veff_shared_integral():
    g = np.load(os.path.join(GOLDEN, "veff_n1000.npz"))
    o = build(g)
    np.random.seed(int(g["rseed"]))
    o.VeffLF()
    # one scipy.quad of the same interpolant instead of 1000: same nodes, same value
    np.testing.assert_allclose(o.phifunc, g["phifunc"], rtol=1e-12)
    np.testing.assert_allclose(o.Lavg, g["Lavg"], rtol=1e-15)
    np.testing.assert_allclose(o.lfbinorig, g["lfbinorig"], rtol=1e-11)
    # the same randint sequence from the same seed: the bootstrap variances are the reference's
    np.testing.assert_allclose(o.var, g["var"], rtol=1e-9)
    assert o.lfbinorig.shape == (50,) and (o.var > 0).all()


def test_veff_per_source_zmax():
    """min_comp_frac = 0.5: each source has its own upper redshift (fsolve) and its own integral; here
    the exact integral of the interpolant, the reference an adaptive quadrature (tolerance ~1e-8)."""
    g = np.load(os.path.join(GOLDEN, "veff_n200_mcf50.npz"))
    o = build(g)
    np.random.seed(int(g["rseed"]))
    with np.errstate(all="ignore"):
        o.VeffLF()
    ref = g["phifunc"]
    assert np.array_equal(o.phifunc == 0, ref == 0)          # sources never observable: weight 0
    nz = ref != 0
    np.testing.assert_allclose(o.phifunc[nz], ref[nz], rtol=2e-7)
    np.testing.assert_allclose(o.lfbinorig, g["lfbinorig"], rtol=2e-7)
    np.testing.assert_allclose(o.var, g["var"], rtol=1e-6)


def test_bootstrap_bins_drop_the_brightest_source():
    from lumfuncmcmc_amd.veff import boot_err_log
    L = np.array([41.0, 41.5, 42.0, 42.5, 43.0])
    phi = np.ones(5)
    np.random.seed(0)
    Lavg, lf, var = boot_err_log(L, phi, nboot=10, nbin=4)
    dL = Lavg[1] - Lavg[0]
    # edges start at min(L)*1.001 > min(L) and the last bin is half-open: first and last source uncounted
    assert abs(lf.sum() * dL - 3.0) < 1e-12


def test_set_median_fit_and_table_flow():
    """The post-fit calls of the drivers (run_lumfuncmcmc.py:291-323) on a finished chain: no GPU
    involved - median LF over posterior draws, the 1/Veff estimate, percentiles into the table."""
    g = np.load(os.path.join(GOLDEN, "veff_n1000.npz"))
    o = build(g)
    rng = np.random.default_rng(3)
    nd = 9
    centre = np.array([42.5, -2.0, -1.49, 2.72, 3.61, 2.55, 3.31, 3.30, 4.56])
    o.samples = np.hstack([centre + 0.01 * rng.normal(size=(800, nd)), -rng.random((800, 1)) * 5 - 1e4])
    np.random.seed(1)
    o.set_median_fit(rndsamples=20)
    assert o.medianLF.shape == (1000,) and np.all(o.medianLF > 0)
    assert o.Lavg.shape == (50,) and o.lfbinorig.shape == (50,) and o.var.shape == (50,)
    assert len(o.Flim) == 5 and 4.0 < o.alpha < 5.0
    names = o.get_param_names()
    percentiles = [5, 16, 50, 84, 95]
    o.table = [[0.0] * (1 + len(names) * len(percentiles))]
    o.add_fitinfo_to_table(percentiles)
    row = o.table[-1]
    assert abs(row[3] - 42.5) < 0.01 and row[1] < row[3] < row[5]          # L*: 5th < 50th < 95th


def test_vectorised_max_redshift_equals_the_per_source_fsolve():
    """V.getMaxz (VmaxLumFunc.py:761-777) solves 4 pi DL(z)^2 fmin = L with one fsolve per source; the build
    inverts DL for all sources at once.  Same roots (the reference's to its own tolerance)."""
    from lumfuncmcmc_amd import veff
    from lumfuncmcmc_amd.cosmology import cosmo
    rng = np.random.default_rng(0)
    L = 10 ** rng.uniform(41.0, 43.5, 150)
    fm = 10 ** rng.uniform(-17.5, -16.3, 150)
    z0 = veff.max_redshift_fsolve(L, fm, cosmo)
    z1 = veff.max_redshift(L, fm, cosmo)
    np.testing.assert_allclose(z1, z0, rtol=1e-9)
    res = 4 * np.pi * (cosmo.luminosity_distance(z1) * veff.MPC_CM_EXACT) ** 2 * fm / L - 1
    assert np.max(np.abs(res)) < 1e-13
    assert np.isnan(veff.max_redshift(np.array([1e42, np.nan]), np.array([1e-17, 1e-17]), cosmo)[1])
