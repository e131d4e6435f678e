"""VeffLF on the device (lf_veff) against the host implementation and against the reference's own seeded results
(tests/golden/veff_*.npz, recorded from LumFuncMCMC.VeffLF: lumfuncmcmc.py:515-525, VmaxLumFunc.py:235-257, :304-378)."""
import os

import numpy as np
import pytest

from lf_testlib import synth

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _model(n, seed, mcf, nboot, nbins):
    from lumfuncmcmc_amd.model import LumFuncMCMC
    cat = synth.catalogue(n, seed=seed)
    fi = cat["field_ind"]
    return LumFuncMCMC(synth.split_fields(cat["z"], fi), lum=synth.split_fields(cat["lum"], fi),
                       lum_e=synth.split_fields(cat["lum_e"], fi), Flim=list(synth.FLIM), alpha=synth.ALPHA_C,
                       Omega_0=list(synth.OMEGA_0), sch_al=synth.SCH_AL, sch_al_lims=synth.SCH_AL_LIMS, Lstar=synth.LSTAR,
                       Lstar_lims=synth.LSTAR_LIMS, phistar=synth.PHISTAR, phistar_lims=synth.PHISTAR_LIMS, Lc=synth.LC,
                       Lh=synth.LH, nwalkers=32, nsteps=10, min_comp_frac=mcf, field_ind=fi, Flim_lims=synth.FLIM_LIMS,
                       alpha_lims=synth.ALPHA_LIMS, nboot=nboot, nbins=nbins)


@pytest.mark.parametrize("name", ["veff_n1000", "veff_n200_mcf50"])
def test_device_veff_reproduces_the_reference_under_its_own_seed(name):
    """Weights, binned LF and bootstrap variances of the reference, through the device kernels: the resample indices
    are the reference's own (np.random.seed(rseed); one randint(N, size=N) per resample) handed to lf_veff."""
    from lumfuncmcmc_amd import veff
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    n, nboot, nbins, mcf = len(g["lum"]), int(g["nboot"]), int(g["nbins"]), float(g["min_comp_frac"])
    o = _model(n, int(g["seed"]), mcf, nboot, nbins)
    np.testing.assert_array_equal(o.lum, g["lum"])
    o.getFlim()
    if mcf <= 0.001:
        zmaxval = o.zmax
    else:
        from lumfuncmcmc_amd.cosmology import cosmo
        zmaxval = np.minimum(o.zmax, veff.max_redshift(10 ** o.lum, o.rootsf.ev(o.Flims_arr, o.alpha), cosmo))
    vol = veff.comoving_volume(o.dVdzf, o.zmin, zmaxval)
    np.random.seed(int(g["rseed"]))
    idx = np.array([np.random.randint(n, size=n) for _ in range(nboot)])
    phi, Lavg, lfbin, var = veff.veff_device(o.lum, o.flux, 1.0e-17 * o.Flims_arr, vol, sum(o.Omega_0), o.alpha, o.fcmin,
                                             nboot=nboot, nbin=nbins, boot_idx=idx)
    # (per-source z_max: the reference's adaptive quadrature of an interpolant agrees with the exact integral to ~1e-8)
    tol = 1e-12 if mcf <= 0.001 else 5e-7
    np.testing.assert_allclose(phi, g["phifunc"], rtol=tol)
    np.testing.assert_allclose(Lavg, g["Lavg"], rtol=1e-14)
    np.testing.assert_allclose(lfbin, g["lfbinorig"], rtol=tol)
    np.testing.assert_allclose(var, g["var"], rtol=max(tol, 1e-9))
    o.close()


def test_device_veff_against_the_host_at_catalogue_size():
    from lumfuncmcmc_amd import veff
    n = 400000
    o = _model(n, 3, 0.0, 60, 40)
    o.VeffLF(device=False)
    host = (o.phifunc.copy(), o.lfbinorig.copy(), o.var.copy())
    np.random.seed(5)
    o.VeffLF(device=True)
    np.testing.assert_allclose(o.phifunc, host[0], rtol=1e-13)
    np.testing.assert_allclose(o.lfbinorig, host[1], rtol=1e-12)       # (atomic adds: summation order)
    # different random streams: the two bootstrap variances agree as two draws of the same estimator do
    ok = host[1] > 0
    ratio = o.var[ok] / host[2][ok]
    assert 0.5 < np.median(ratio) < 2.0 and np.all(ratio > 0.1) and np.all(ratio < 10.0), ratio
    o.close()


def test_resampling_indices_outside_the_catalogue_are_refused():
    """boot_idx addresses the weights on the device: an index outside [0, n) is an argument error, not a stray read"""
    from lumfuncmcmc_amd import capi
    n = 1000
    rng = np.random.default_rng(1)
    flux, flim = rng.uniform(2e-17, 9e-17, n), np.full(n, 2.7e-17)
    bins = rng.integers(0, 10, n)
    for bad in (-1, n, 2 ** 40):
        idx = rng.integers(0, n, (3, n))
        idx[1, 17] = bad
        with pytest.raises(capi.LFError):
            capi.veff_device(flux, flim, 1.0e6, 0.04, 4.56, 0.1, bin_of=bins, nbin=10, nboot=3, boot_idx=idx)
    phi, sums = capi.veff_device(flux, flim, 1.0e6, 0.04, 4.56, 0.1, bin_of=bins, nbin=10, nboot=3, boot_idx=rng.integers(0, n, (3, n)))
    assert np.isfinite(phi).all() and sums.shape == (4, 10)
