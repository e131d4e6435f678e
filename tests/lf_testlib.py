"""Shared helpers for the parity tests: synthetic-but-consistent kernel inputs of any size.

The kernels and the oracle consume the same arrays (the outputs of the reference's setup), so
parity at sizes beyond the golden fixtures does not need the real cosmology: a smooth analytic
stand-in for DL(z) and dV/dz gives inputs with the right ranges.  Test infrastructure only.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import lf_oracle as O                       # noqa: E402
from lumfuncmcmc_amd import synth           # noqa: E402


def dl_standin(z):
    """Smooth, monotone stand-in for the luminosity distance in Mpc (about LCDM over 1 < z < 2.5)."""
    z = np.asarray(z, dtype=np.float64)
    return 7938.7 + 8733.0 * (z - 1.16) + 1180.0 * (z - 1.16) ** 2


def dvdz_standin(z):
    z = np.asarray(z, dtype=np.float64)
    return 2.9975e10 + 1.51e10 * (z - 1.16) - 5.2e9 * (z - 1.16) ** 2


def make_inputs(variant, n, seed=20241016, S=None, fix_sch_al=False, zslices=0, pivots=(1.20, 1.53, 1.86),
                nf=5):
    cat = synth.catalogue(n, seed=seed, nf=nf, zslices=zslices)
    z, lum, fi = cat["z"], cat["lum"], cat["field_ind"]
    if S is None:
        S = 101 if variant == "free" else 201
    Flim0 = np.array(synth.FLIM[:nf]) if nf <= 5 else np.linspace(2.2, 3.6, nf)
    om0 = np.array(synth.OMEGA_0[:nf]) if nf <= 5 else np.linspace(3.5e5, 4.5e5, nf) + 0.25
    zmin, zmax = (z.min(), z.max()) if n else (synth.ZLO, synth.ZHI)
    zarr = np.linspace(zmin, zmax, S)
    lmin = lum.min() if n else 41.0
    logL = np.repeat(np.linspace(lmin, synth.LH, S)[:, None], S, axis=1)
    DLz = dl_standin(z)
    DL_zarr = dl_standin(zarr)
    inp = {
        "variant": variant, "fix_sch_al": fix_sch_al, "sch_al0": synth.SCH_AL,
        "field_ind": fi, "lum": lum, "z": z, "DLz": DLz, "Omega_0": om0,
        "Flim0": Flim0, "alpha0": synth.ALPHA_C, "size_ln": S, "logL": logL, "zarr": zarr,
        "DL_zarr": DL_zarr, "volume_part": dvdz_standin(zarr), "fcmin": synth.FCMIN,
        "lims": dict(O.DEFAULT_LIMS), "pivots": tuple(pivots), "integ_part": None, "integ_sum": None,
    }
    om0_arr = np.zeros(n, dtype=int)
    fl_arr = np.zeros(n)
    for f in range(nf):
        om0_arr[fi[f]:fi[f + 1]] = om0[f]
        fl_arr[fi[f]:fi[f + 1]] = Flim0[f]
    with np.errstate(all="ignore"):
        inp["Om_arr"] = O.omega(lum, DLz, om0_arr, 1.0e-17 * fl_arr, synth.ALPHA_C, synth.FCMIN)
        if variant != "free":
            DLg = np.repeat(DL_zarr[None], S, axis=0)
            inp["integ_part"] = np.array([
                inp["volume_part"] * O.omega(logL, DLg, om0[f], 1.0e-17 * Flim0[f], synth.ALPHA_C, synth.FCMIN)
                for f in range(nf)])
    return inp


def min_linear_product(inp, theta):
    """Smallest per-source product TrueLumFunc*Omega (linear space) for one theta row: tells
    whether a row sits in the subnormal band where log(product) is quantised."""
    p = O.split_theta(inp, np.asarray(theta, dtype=float))
    lum = inp["lum"]
    with np.errstate(all="ignore"):
        if inp["variant"] == "free":
            fi = inp["field_ind"]
            om0_arr = O._field_scatter(inp["Omega_0"], fi, dtype=int)
            fl = O._field_scatter(p["Flim"], fi)
            prod = O.true_lum_func(lum, p["sch_al"], p["Lstar"], p["phistar"]) * \
                O.omega(lum, inp["DLz"], om0_arr, 1.0e-17 * fl, p["alpha"], inp["fcmin"])
            e = np.exp(-10 ** (lum - p["Lstar"]))
        elif inp["variant"] == "fixcomp":
            prod = O.true_lum_func(lum, p["sch_al"], p["Lstar"], p["phistar"]) * inp["Om_arr"]
            e = np.exp(-10 ** (lum - p["Lstar"]))
        else:
            z1, z2, z3 = inp["pivots"]
            prod = O.schechter_z(lum, inp["z"], p["sch_al"], *p["L"], *p["phi"], z1, z2, z3) * inp["Om_arr"]
            e = prod
    return float(min(np.min(prod), np.min(e))) if len(lum) else 1.0


def compare_rows(got, ref, inp, thetas, rtol, what="lnprob"):
    """-inf pattern identical; finite rows within rtol, except rows in the subnormal band, where
    the reference's own value is quantised: there both must be below -700 and within 1e-3."""
    got, ref = np.asarray(got), np.asarray(ref)
    assert not np.isnan(got[~np.isnan(ref)]).any(), "%s: NaN from the kernel" % what
    worst = 0.0
    for i in range(len(ref)):
        if np.isnan(ref[i]):
            continue
        if np.isinf(ref[i]):
            assert got[i] == ref[i], "%s row %d: expected -inf, got %r" % (what, i, got[i])
            continue
        tiny = min_linear_product(inp, thetas[i]) < 2.3e-308
        if tiny:
            assert got[i] == -np.inf or abs(got[i] - ref[i]) <= 1e-3 * abs(ref[i]), (what, i, got[i], ref[i])
            continue
        assert np.isfinite(got[i]), "%s row %d: expected %r, got %r" % (what, i, ref[i], got[i])
        rel = abs(got[i] - ref[i]) / max(abs(ref[i]), 1e-300)
        worst = max(worst, rel)
        assert rel <= rtol, "%s row %d: got %.17g ref %.17g rel %.3e" % (what, i, got[i], ref[i], rel)
    return worst


def e2e_model(g, **extra):
    """The model object of an end-to-end fixture (tests/golden/e2e_*.npz, oracle/gen_golden.py --only e2e): the catalogue
    is regenerated from the generator arguments the fixture stores and goes through the build's own host setup."""
    from lumfuncmcmc_amd.model import LumFuncMCMC, LumFuncMCMCz
    variant, n = str(g["variant"]), int(g["n"])
    cat = synth.catalogue(n, seed=int(g["seed"]), zslices=int(g["zslices"]))
    if "faint" in g.files and int(g["faint"]):
        cat["lum"][5] = 39.75                  # the fixture's one source far below the flux limit of field 0
    fi = cat["field_ind"]
    fsa = bool(g["fix_sch_al"]) if "fix_sch_al" in g.files else False
    mcf = float(g["min_comp_frac"]) if "min_comp_frac" in g.files else 0.0
    kw = dict(lum=synth.split_fields(cat["lum"], fi), lum_e=synth.split_fields(cat["lum_e"], fi),
              Flim=list(synth.FLIM), alpha=synth.ALPHA_C, Omega_0=list(synth.OMEGA_0), sch_al=synth.SCH_AL,
              sch_al_lims=synth.SCH_AL_LIMS, Lstar=synth.LSTAR, Lstar_lims=synth.LSTAR_LIMS,
              phistar=synth.PHISTAR, phistar_lims=synth.PHISTAR_LIMS, Lc=synth.LC, Lh=synth.LH, nwalkers=32,
              nsteps=10, min_comp_frac=mcf, field_ind=fi, fix_sch_al=fsa)
    kw.update(extra)
    zs = synth.split_fields(cat["z"], fi)
    if variant == "zevol":
        return LumFuncMCMCz(zs, **kw)
    return LumFuncMCMC(zs, fix_comp=(variant == "fixcomp"), Flim_lims=synth.FLIM_LIMS, alpha_lims=synth.ALPHA_LIMS, **kw)


def e2e_compare(o, g, rtol):
    """lnprob of the fixture's theta rows through the class surface against the reference's own values."""
    variant = str(g["variant"])
    th = g["theta"]
    got = o.lnprob_fix_comp(th) if variant == "fixcomp" else o.lnprob(th)
    ref = g["lnprob"]
    inp = o.kernel_inputs()
    inp["lims"] = {k: list(v) for k, v in inp["lims"].items()}
    return compare_rows(got, ref, inp, th, rtol), int(np.isinf(ref).sum())
