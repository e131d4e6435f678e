"""Host setup (own cosmology + tables) against arrays recorded from the reference's constructor.
These arrays are the kernels' inputs: if they match, the GPU result depends only on the kernels."""
import json
import os

import numpy as np
import pytest

from lumfuncmcmc_amd import hostsetup as hs, synth
from lumfuncmcmc_amd.cosmology import cosmo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def kwargs_from(g, fix_comp=False, mcf=0.0):
    fi = g["field_ind"]
    return dict(lum=synth.split_fields(g["lum"], fi), lum_e=synth.split_fields(g["lum_e"], fi),
                Flim=list(synth.FLIM), alpha=synth.ALPHA_C, Omega_0=list(synth.OMEGA_0),
                sch_al=synth.SCH_AL, sch_al_lims=synth.SCH_AL_LIMS, Lstar=synth.LSTAR,
                Lstar_lims=synth.LSTAR_LIMS, phistar=synth.PHISTAR, phistar_lims=synth.PHISTAR_LIMS,
                Lc=synth.LC, Lh=synth.LH, nwalkers=32, nsteps=50, fix_sch_al=bool(g["fix_sch_al"]),
                min_comp_frac=mcf, field_ind=fi)


def test_cosmology_known_answers():
    with open(os.path.join(GOLDEN, "cosmo_known.json")) as f:
        k = json.load(f)
    assert abs(cosmo.Ogamma0 / k["Ogamma0"] - 1) < 1e-14
    assert abs(cosmo.Onu0 / k["Onu0"] - 1) < 1e-14
    assert abs(cosmo.Ok0 / k["Ok0"] - 1) < 1e-13 and cosmo.Ok0 < 0          # closed: sin branch
    z = np.array(k["z"])
    np.testing.assert_allclose(cosmo.luminosity_distance(z), k["DL_Mpc"], rtol=5e-15)
    np.testing.assert_allclose(cosmo.differential_comoving_volume(z), k["dVc_dz_dOmega_Mpc3_sr"], rtol=5e-15)
    # table path (anchors + short steps) == direct path
    zz = np.linspace(1.0, 2.1, 20001)
    np.testing.assert_allclose(cosmo.luminosity_distance(zz)[::500], cosmo.luminosity_distance(zz[::500]), rtol=2e-15)


def test_linear_interp_is_interp1d():
    x = np.linspace(1.0, 2.0, 57)
    y = np.sin(x) * 1e4
    f = hs.LinearInterp(x, y)
    from scipy.interpolate import interp1d
    xn = np.random.default_rng(0).uniform(1.0, 2.0, 1000)
    assert np.array_equal(f(xn), interp1d(x, y)(xn))
    assert np.array_equal(f(x), y)
    with pytest.raises(ValueError):
        f(np.array([2.0000001]))


@pytest.mark.parametrize("name", ["free_n50", "free_n1000"])
def test_free_constructor_arrays(name):
    from lumfuncmcmc_amd.model import LumFuncMCMC
    g = load(name)
    kw = kwargs_from(g)
    o = LumFuncMCMC(synth.split_fields(g["z"], g["field_ind"]), fix_comp=False, Flim_lims=synth.FLIM_LIMS,
                    alpha_lims=synth.ALPHA_LIMS, **kw)
    np.testing.assert_allclose(o._DLarr, g["DLarr"], rtol=3e-15)
    np.testing.assert_allclose(o.dVdzf(g["zint"]), g["dVdzarr"], rtol=5e-15)
    np.testing.assert_allclose(o._DLz, g["DLz"], rtol=3e-15)
    np.testing.assert_allclose(o.DL, g["DL_exact"], rtol=3e-15)
    np.testing.assert_allclose(o.flux, g["flux"], rtol=1e-14)
    assert o.size_ln == 101 and np.array_equal(o.zarr, g["zarr"])
    np.testing.assert_allclose(o.DL_zarr, g["DL_zarr"], rtol=3e-15)
    np.testing.assert_allclose(o.volume_part, g["volume_part"], rtol=5e-15)
    assert np.array_equal(o.logL[0], g["logL"]) and all(o.logL[i] is o.logL[0] for i in range(5))
    assert o.Omega_0_arr.dtype.kind == "i" and np.array_equal(o.Omega_0_arr, g["Omega_0_arr"])
    np.testing.assert_allclose(o.Om_arr, g["Om_arr"], rtol=1e-12)
    inp = o.kernel_inputs()
    assert inp["variant"] == "free" and inp["integ_part"] is None


def test_fixcomp_integ_part_through_the_spline():
    from lumfuncmcmc_amd.model import LumFuncMCMC
    g = load("fixcomp_n1000")
    o = LumFuncMCMC(synth.split_fields(g["z"], g["field_ind"]), fix_comp=True, Flim_lims=synth.FLIM_LIMS,
                    alpha_lims=synth.ALPHA_LIMS, **kwargs_from(g))
    assert o.size_ln == 201
    ip = np.array(o.integ_part)
    ref = g["integ_part"]
    scale = np.abs(ref).max(axis=(1, 2), keepdims=True)
    assert np.max(np.abs(ip - ref) / scale) < 1e-12
    assert np.array_equal(o.logL[0], g["logL"])


def test_zevol_constructor():
    from lumfuncmcmc_amd.model import LumFuncMCMCz
    g = load("zevol_n800")
    kw = kwargs_from(g)
    piv = g["pivots"]
    o = LumFuncMCMCz(synth.split_fields(g["z"], g["field_ind"]), z1=piv[0], z2=piv[1], z3=piv[2], **kw)
    s = np.sum(np.array(o.integ_part), axis=0)
    assert np.max(np.abs(s - g["integ_sum"])) / np.abs(g["integ_sum"]).max() < 1e-12
    np.testing.assert_allclose(o.Om_arr, g["Om_arr"], rtol=1e-12)
    assert o.get_init_walker_values().shape == (32, 7)
    assert len(o.get_param_names()) == 7


def test_min_comp_frac_grid_aliasing():
    """min_comp_frac = 0.5: per-field L grids differ but all fields integrate on the last one."""
    from lumfuncmcmc_amd.model import LumFuncMCMC
    g = load("free_n300_mcf50")
    o = LumFuncMCMC(synth.split_fields(g["z"], g["field_ind"]), fix_comp=False, Flim_lims=synth.FLIM_LIMS,
                    alpha_lims=synth.ALPHA_LIMS, **kwargs_from(g, mcf=0.5))
    np.testing.assert_allclose(o.roots_ln, g["roots_setup"], rtol=1e-9)
    ml = np.array([o.minlumf[i](o.zarr) for i in range(5)])
    np.testing.assert_allclose(ml, g["minlum_zarr"], rtol=1e-11)
    np.testing.assert_allclose(o.logL[0], g["logL"], rtol=1e-11)      # = the LAST field's grid
    assert o.logL[0] is o.logL[4]


def test_flux_input_path():
    from lumfuncmcmc_amd.model import LumFuncMCMC
    g = load("free_n200_fluxin")
    fi = g["field_ind"]
    kw = kwargs_from(g)
    kw.update(lum=None, lum_e=None, flux=synth.split_fields(g["flux17"], fi),
              flux_e=synth.split_fields(g["flux17_e"], fi))
    o = LumFuncMCMC(synth.split_fields(g["z"], fi), fix_comp=False, Flim_lims=synth.FLIM_LIMS,
                    alpha_lims=synth.ALPHA_LIMS, **kw)
    np.testing.assert_allclose(o.lum, g["lum"], rtol=1e-15)
    np.testing.assert_allclose(o.lum_e, g["lum_e"], rtol=1e-12)
    np.testing.assert_allclose(o._DLz, g["DLz"], rtol=3e-15)
