#!/opt/conda/bin/python3.9
"""Golden-vector generator: runs the REFERENCE itself and records what it returns.

TEST INFRASTRUCTURE - never imported by the product path.

Run in the build container only (the reference does not exist on the GPU box):

    /opt/conda/bin/python3.9 oracle/gen_golden.py            # writes tests/golden/*.npz, *.json

What it does (SURVEY.md App. C recipe):
  * imports /root/reference/lumfuncmcmc.py and lumfuncmcmc_z.py unmodified, under the
    conda interpreter (numpy 1.26.4 / scipy 1.7.1 / astropy 4.3.1), with in-process
    stand-ins for four packages that are absent on this machine and that do not take part
    in the lnprob arithmetic: emcee (the caller), corner / lmfit (plots, fits),
    uncertainties (flux<->lum error propagation at setup; first-order stub);
  * builds small synthetic catalogues with lumfuncmcmc_amd/synth.py;
  * records constructor-derived arrays (the kernel's inputs), theta batches, and the
    reference's lnprob for each theta, plus piece A (per-source log-term sum,
    lumfuncmcmc.py:370 / :388 / lumfuncmcmc_z.py:371) and piece B (trapz^2 integral,
    :373-377 / :389-392 / _z:373-375) separately.

The files written are DATA (inputs + expected outputs); no reference source text is stored.
"""
import importlib.util
import json
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
OUT = os.path.join(REPO, "tests", "golden")
REF = "/root/reference"


# --------------------------------------------------------------------------- shims
def install_shims():
    for name, fn in (("asscalar", lambda a: a.item()), ("alen", len), ("rank", np.ndim)):
        if not hasattr(np, name):
            setattr(np, name, fn)

    class UArr(object):
        """value/sigma pair with first-order propagation (stand-in for uncertainties)."""
        __array_ufunc__ = None

        def __init__(self, n, s):
            self.n = np.asarray(n, dtype=float)
            self.s = np.asarray(s, dtype=float)

        def __mul__(self, c):
            return UArr(self.n * c, self.s * np.abs(c))
        __rmul__ = __mul__

        def __truediv__(self, c):
            return UArr(self.n / c, self.s / np.abs(c))

        def __rpow__(self, base):
            v = base ** self.n
            return UArr(v, np.log(base) * v * self.s)

    unumpy = types.ModuleType("uncertainties.unumpy")
    unumpy.uarray = lambda n, s: UArr(n, s)
    unumpy.log10 = lambda u: UArr(np.log10(u.n), u.s / (np.abs(u.n) * np.log(10.0)))
    unumpy.exp = lambda u: UArr(np.exp(u.n), np.exp(u.n) * u.s)
    unumpy.nominal_values = lambda u: u.n
    unumpy.std_devs = lambda u: u.s
    unc = types.ModuleType("uncertainties")
    unc.unumpy = unumpy
    unc.ufloat = lambda n, s: UArr(n, s)
    sys.modules["uncertainties"] = unc
    sys.modules["uncertainties.unumpy"] = unumpy

    emcee = types.ModuleType("emcee")
    emcee.EnsembleSampler = object
    sys.modules["emcee"] = emcee
    corner = types.ModuleType("corner")
    corner.corner = None
    sys.modules["corner"] = corner
    lmfit = types.ModuleType("lmfit")
    lmfit.Model = object
    sys.modules["lmfit"] = lmfit
    sys.path.insert(0, REF)


def load_synth():
    spec = importlib.util.spec_from_file_location(
        "lf_synth", os.path.join(REPO, "lumfuncmcmc_amd", "synth.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# --------------------------------------------------------------------------- theta sets
def theta_free(S, nrows, fix_sch_al, seed):
    """Rows inside the finite box, the configLF point, the full prior box, rows outside the
    prior, rows exactly on the (inclusive) prior edges, and the underflow zone."""
    rng = np.random.default_rng(seed)
    nd = S.ndim_of("free", fix_sch_al)
    th = [S.walkers("free", nrows, seed=seed + 100, fix_sch_al=fix_sch_al)]
    base = [S.LSTAR, S.PHISTAR] + ([] if fix_sch_al else [S.SCH_AL]) + list(S.FLIM) + [S.ALPHA_C]
    th.append(np.array([base]))
    lims = [S.LSTAR_LIMS, S.PHISTAR_LIMS] + ([] if fix_sch_al else [S.SCH_AL_LIMS]) \
        + [S.FLIM_LIMS] * 5 + [S.ALPHA_LIMS]
    lims = np.array(lims, dtype=float)
    full = rng.random((10, nd)) * (lims[:, 1] - lims[:, 0]) + lims[:, 0]
    th.append(full)
    extra = []
    for j in range(nd):                       # one parameter just outside, low then high
        lo = np.array(base); lo[j] = lims[j, 0] - 1e-9; extra.append(lo)
        hi = np.array(base); hi[j] = lims[j, 1] + 1e-9; extra.append(hi)
    for j in range(nd):                       # exactly on the edges (inclusive prior)
        lo = np.array(base); lo[j] = lims[j, 0]; extra.append(lo)
        hi = np.array(base); hi[j] = lims[j, 1]; extra.append(hi)
    for ls in (40.3, 40.55, 40.6, 40.62, 40.64, 40.66, 40.7, 40.8, 41.0):   # underflow zone
        u = np.array(base); u[0] = ls; extra.append(u)
    th.append(np.array(extra))
    return np.concatenate(th, axis=0)


def theta_fixcomp(S, nrows, fix_sch_al, seed):
    rng = np.random.default_rng(seed)
    nd = S.ndim_of("fixcomp", fix_sch_al)
    th = [S.walkers("fixcomp", nrows, seed=seed + 100, fix_sch_al=fix_sch_al)]
    base = [S.LSTAR, S.PHISTAR] + ([] if fix_sch_al else [S.SCH_AL])
    th.append(np.array([base]))
    lims = np.array([S.LSTAR_LIMS, S.PHISTAR_LIMS] + ([] if fix_sch_al else [S.SCH_AL_LIMS]), dtype=float)
    th.append(rng.random((10, nd)) * (lims[:, 1] - lims[:, 0]) + lims[:, 0])
    extra = []
    for j in range(nd):
        for v in (lims[j, 0] - 1e-9, lims[j, 1] + 1e-9, lims[j, 0], lims[j, 1]):
            u = np.array(base); u[j] = v; extra.append(u)
    for ls in (40.3, 40.55, 40.6, 40.62, 40.64, 40.66, 40.7, 40.8, 41.0):
        u = np.array(base); u[0] = ls; extra.append(u)
    th.append(np.array(extra))
    return np.concatenate(th, axis=0)


def theta_zevol(S, nrows, fix_sch_al, seed):
    rng = np.random.default_rng(seed)
    nd = S.ndim_of("zevol", fix_sch_al)
    th = [S.walkers("zevol", nrows, seed=seed + 100, fix_sch_al=fix_sch_al)]
    base = [42.4, 42.5, 42.6, -2.1, -2.0, -1.9] + ([] if fix_sch_al else [S.SCH_AL])
    th.append(np.array([base]))
    lims = np.array([S.LSTAR_LIMS] * 3 + [S.PHISTAR_LIMS] * 3 + ([] if fix_sch_al else [S.SCH_AL_LIMS]), dtype=float)
    th.append(rng.random((10, nd)) * (lims[:, 1] - lims[:, 0]) + lims[:, 0])
    extra = []
    for j in range(nd):
        for v in (lims[j, 0] - 1e-9, lims[j, 1] + 1e-9, lims[j, 0], lims[j, 1]):   # L/phi edges are STRICT
            u = np.array(base); u[j] = v; extra.append(u)
    for ls in (40.3, 40.6, 40.64, 40.7, 41.0):
        u = np.array(base); u[0] = ls; u[1] = ls; u[2] = ls; extra.append(u)
    th.append(np.array(extra))
    return np.concatenate(th, axis=0)


# --------------------------------------------------------------------------- recording
class TrapzTap(object):
    """Wraps the reference module's `trapz` name to record the outer (scalar) integrals."""

    def __init__(self, mod):
        self.mod, self.real, self.vals = mod, mod.trapz, []
        mod.trapz = self

    def __call__(self, *a, **k):
        r = self.real(*a, **k)
        if np.ndim(r) == 0:
            self.vals.append(float(r))
        return r

    def take(self):
        v, self.vals = self.vals, []
        return v


def run_thetas(obj, func, thetas, tap, piece_a):
    """Call the reference's lnprob on each row; return lnprob, A, B (nan where the prior fails)."""
    n = len(thetas)
    lnp, A, B = np.empty(n), np.full(n, np.nan), np.full(n, np.nan)
    Bf = np.full((n, obj.nfields), np.nan)
    f = getattr(obj, func)
    for i, th in enumerate(thetas):
        tap.take()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with np.errstate(all="ignore"):
                lnp[i] = f(np.array(th))
                vals = tap.take()
                if len(vals) == obj.nfields:          # prior passed, lnlike ran
                    full = 0.0
                    for v in vals:
                        full += v
                    B[i] = full
                    Bf[i] = vals
                    A[i] = piece_a(obj)                # parameters are still set on obj
                    chk = A[i] - B[i]
                    assert (chk == lnp[i]) or (np.isnan(chk) and np.isnan(lnp[i])) \
                        or (np.isinf(chk) and chk == lnp[i]), (chk, lnp[i])
    return lnp, A, B, Bf


def common_arrays(obj, with_tables, with_integ):
    d = {
        "z": obj.z, "lum": obj.lum, "lum_e": obj.lum_e, "flux": obj.flux,
        "field_ind": np.asarray(obj.field_ind, dtype=np.int64),
        "DLz": obj.DLf(obj.z),                    # linear-interpolated DL at the sources (lumfuncmcmc.py:70)
        "DL_exact": obj.DL,
        "Omega_0_arr": np.asarray(obj.Omega_0_arr, dtype=np.int64),
        "Omega_0": np.asarray(obj.Omega_0, dtype=float),
        "Flim0": np.asarray(obj.Flim, dtype=float), "alpha0": float(obj.alpha),
        "Om_arr": obj.Om_arr,
        "size_ln": obj.size_ln, "zarr": obj.zarr, "DL_zarr": obj.DL_zarr,
        "volume_part": obj.volume_part, "logL": obj.logL[0],
        "logL_aliased": np.array([obj.logL[i] is obj.logL[0] for i in range(obj.nfields)]),
        "fcmin": obj.fcmin, "min_comp_frac": obj.min_comp_frac, "Lc": obj.Lc, "Lh": obj.Lh,
    }
    if with_tables:
        n = len(obj.z)
        zint = np.linspace(0.95 * obj.zmin, 1.05 * obj.zmax, n)
        d["zint"] = zint
        d["DLarr"] = obj.DLf(zint)        # the interpolant returns the node values at the nodes
        d["dVdzarr"] = obj.dVdzf(zint)
    if with_integ:
        d["integ_part"] = np.array(obj.integ_part)
    return d


def main():
    install_shims()
    S = load_synth()
    os.makedirs(OUT, exist_ok=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import lumfuncmcmc as R
        import lumfuncmcmc_z as RZ
        import VmaxLumFunc as V
    tapR, tapZ = TrapzTap(R), TrapzTap(RZ)

    def a_free(o):       # lumfuncmcmc.py:370, evaluated with the reference's own functions
        return np.log(R.TrueLumFunc(o.lum, o.sch_al, o.Lstar, o.phistar)
                      * R.Omega(o.lum, o.z, o.DLf, o.Omega_0_arr, 1.0e-17 * o.Flims_arr, o.alpha, o.fcmin)).sum()

    def a_fix(o):        # lumfuncmcmc.py:388
        return np.log(R.TrueLumFunc(o.lum, o.sch_al, o.Lstar, o.phistar) * o.Om_arr).sum()

    def a_z(o):          # lumfuncmcmc_z.py:371
        return np.log(RZ.schechter_z(o.lum, o.z, o.sch_al, o.L1, o.L2, o.L3, o.phi1, o.phi2, o.phi3,
                                     o.z1, o.z2, o.z3) * o.Om_arr).sum()

    def ctor_kwargs(cat, fix_sch_al, fix_comp, mcf, nw=32, ns=50):
        # exactly the keyword set of run_lumfuncmcmc.py:245-256, values from configLF.py
        return dict(flux=None, flux_e=None,
                    lum=S.split_fields(cat["lum"], cat["field_ind"]),
                    lum_e=S.split_fields(cat["lum_e"], cat["field_ind"]),
                    Flim=list(S.FLIM), alpha=S.ALPHA_C, line_name="OIII",
                    line_plot_name=r'[OIII] $\lambda 5007$', Omega_0=list(S.OMEGA_0),
                    nbins=50, nboot=100, sch_al=S.SCH_AL, sch_al_lims=S.SCH_AL_LIMS,
                    Lstar=S.LSTAR, Lstar_lims=S.LSTAR_LIMS, phistar=S.PHISTAR,
                    phistar_lims=S.PHISTAR_LIMS, Lc=S.LC, Lh=S.LH, nwalkers=nw, nsteps=ns,
                    fix_sch_al=fix_sch_al, fix_comp=fix_comp, min_comp_frac=mcf,
                    Flim_lims=S.FLIM_LIMS, alpha_lims=S.ALPHA_LIMS,
                    field_names=np.array(["AEGIS", "COSMOS", "GOODSN", "GOODSS", "UDS"]),
                    field_ind=cat["field_ind"], diff_rand=True)

    manifest = {}
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    if only:
        with open(os.path.join(OUT, "MANIFEST.json")) as f:
            manifest = json.load(f)

    def save(name, d):
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **d)
        manifest[name] = {"bytes": os.path.getsize(path), "keys": sorted(d.keys())}
        print("wrote", path, os.path.getsize(path))

    # ---- 1/Veff estimator (LumFuncMCMC.VeffLF, lumfuncmcmc.py:515-525): per-source weights from
    #      V.lumfunc (one scipy.quad each) and the bootstrap of V.getBootErrLog with a seeded global state
    def veff_case(name, n, seed, mcf, nboot, nbins, rseed):
        cat = S.catalogue(n, seed=seed)
        kw = ctor_kwargs(cat, False, False, mcf)
        kw.update(nboot=nboot, nbins=nbins)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with np.errstate(all="ignore"):
                o = R.LumFuncMCMC(S.split_fields(cat["z"], cat["field_ind"]), **kw)
                np.random.seed(rseed)
                import io, contextlib
                with contextlib.redirect_stdout(io.StringIO()):
                    o.VeffLF()
        save(name, dict(z=o.z, lum=o.lum, lum_e=o.lum_e, field_ind=np.asarray(o.field_ind, dtype=np.int64),
                        flux=o.flux, phifunc=o.phifunc, Lavg=o.Lavg, lfbinorig=o.lfbinorig, var=o.var,
                        min_comp_frac=mcf, nboot=nboot, nbins=nbins, rseed=rseed, seed=seed))

    if only in (None, "veff"):
        veff_case("veff_n1000", 1000, 0, 0.0, 100, 50, 12345)
        veff_case("veff_n200_mcf50", 200, 7, 0.5, 20, 10, 54321)
    # ---- table formats: a catalogue written by astropy, what the driver's read_input_file
    #      (run_lumfuncmcmc.py:165-228) makes of it, and astropy's fixed_width_two_line output
    if only in (None, "tableio"):
        import io, contextlib
        from astropy.table import Table
        with contextlib.redirect_stdout(io.StringIO()):
            import run_lumfuncmcmc as DRV
        rng = np.random.default_rng(77)
        n = 60
        fnames = np.array(["AEGIS", "COSMOS", "GOODSN", "GOODSS", "UDS"])
        fld = fnames[rng.integers(0, 5, n)]
        cat = Table([fld, np.arange(n), rng.uniform(S.ZLO, S.ZHI, n), rng.uniform(0.3, 30.0, n),
                     rng.uniform(0.1, 1.0, n)], names=["Field", "ID", "z", "OIII_flux", "OIII_flux_e"])
        cpath = os.path.join(OUT, "catalogue_n60.dat")
        cat.write(cpath, format="ascii", overwrite=True)
        for mcf, tag in ((0.0, "mcf0"), (0.5, "mcf50")):
            args = types.SimpleNamespace(filename=cpath, min_comp_frac=mcf, Flim=list(S.FLIM), alpha=S.ALPHA_C,
                                         fcmin=S.FCMIN, line_name="OIII")
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                z, flux, flux_e, lum, lum_e, field_names, field_ind = DRV.read_input_file(args, dust_fn=None)[:7]
            save("readinput_%s" % tag, dict(z=np.concatenate(z), flux=np.concatenate(flux), flux_e=np.concatenate(flux_e),
                                            field_names=np.array([str(x) for x in field_names]),
                                            field_ind=np.asarray(field_ind, dtype=np.int64), min_comp_frac=mcf))
        T = Table([rng.uniform(41, 43, 6), np.full(6, 0.05), rng.uniform(1e-4, 1e-2, 6)],
                  names=["Luminosity", "Luminosity_Err", "MedianLF"])
        T.write(os.path.join(OUT, "fwtl_plain.dat"), overwrite=True, format="ascii.fixed_width_two_line")
        np.save(os.path.join(OUT, "fwtl_plain_cols.npy"), np.array([T[c] for c in T.colnames]))
        names = ["Line", r"$\log L_*$_05", r"$\log L_*$_50"]
        T2 = Table(names=names, dtype=["S10", "f8", "f8"])
        T2.add_row(["OIII", 42.123456, 42.5])
        T2.write(os.path.join(OUT, "fwtl_formats.dat"), format="ascii.fixed_width_two_line", overwrite=True,
                 formats={"Line": "%s", names[1]: "%0.3f", names[2]: "%0.3f"})
    # ---- end to end at BASELINE sizes: only the generator arguments, theta and the reference's lnprob
    #      are stored (a few KB) - the test rebuilds the catalogue with synth.catalogue and goes through
    #      the build's own host setup, so setup + kernels are compared with the reference together
    if only in (None, "e2e"):
        # (name, variant, n, zslices, fix_sch_al, min_comp_frac, faint source)
        cases = (("e2e_free_n100000", "free", 100000, 0, False, 0.0, True),
                 ("e2e_free_n1000000", "free", 1000000, 0, False, 0.0, True),
                 ("e2e_fixcomp_n1000000", "fixcomp", 1000000, 0, False, 0.0, False),
                 ("e2e_zevol_n800000", "zevol", 800000, 8, False, 0.0, False),
                 ("e2e_free_fsa_n100000", "free", 100000, 0, True, 0.0, False),
                 ("e2e_free_mcf50_n100000", "free", 100000, 0, False, 0.5, False))
        for (name, variant, n, zsl, fsa, mcf, faint) in cases:
            seed = 20241016
            cat = S.catalogue(n, seed=seed, zslices=zsl)
            if faint:
                cat["lum"][5] = 39.75          # one source ~1.9 dex below the 50 % flux of field 0: SLOW-mode walkers
            kw = ctor_kwargs(cat, fsa, variant == "fixcomp", mcf)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                with np.errstate(all="ignore"):
                    if variant == "zevol":
                        for k in ("fix_comp", "Flim_lims", "alpha_lims", "diff_rand"):
                            kw.pop(k)
                        o = RZ.LumFuncMCMCz(S.split_fields(cat["z"], cat["field_ind"]), **kw)
                        f = o.lnprob
                    else:
                        o = R.LumFuncMCMC(S.split_fields(cat["z"], cat["field_ind"]), **kw)
                        f = o.lnprob_fix_comp if variant == "fixcomp" else o.lnprob
                    # 48 rows: 32 in the finite box; the underflow zone; the prior's edges (on them, 1e-9 outside);
                    # (free, faint source) walkers whose bounds cannot exclude underflow
                    th = S.walkers(variant, 48, seed=1, fix_sch_al=fsa)
                    nd = th.shape[1]
                    if variant == "zevol":
                        th[32, 0] = S.LSTAR_LIMS[0]                  # strict for L and phi (lumfuncmcmc_z.py:355-358): -inf
                        th[33, 0] = S.LSTAR_LIMS[0] + 1e-9
                        th[34, 2] = S.LSTAR_LIMS[1]
                        th[35, 2] = S.LSTAR_LIMS[1] - 1e-9
                        th[36, 3] = S.PHISTAR_LIMS[1]
                        th[37, 5] = S.PHISTAR_LIMS[0] + 1e-9
                        if not fsa:
                            th[38, 6] = S.SCH_AL_LIMS[0]             # inclusive for alpha (:351-353)
                            th[39, 6] = S.SCH_AL_LIMS[1] + 1e-9
                        th[40:44, 0:3] = np.array([[40.2, 40.3, 40.4], [40.6, 40.5, 40.4], [41.0, 40.2, 41.0], [40.9, 40.9, 40.9]])
                    else:
                        th[32, 0] = 40.3                             # underflow zone
                        th[33, 0] = S.LSTAR_LIMS[1]                  # inclusive box (lumfuncmcmc.py:346-358)
                        th[34, 0] = S.LSTAR_LIMS[1] + 1e-9
                        th[35, 1] = S.PHISTAR_LIMS[0]
                        th[36, 1] = S.PHISTAR_LIMS[0] - 1e-9
                        if not fsa:
                            th[37, 2] = S.SCH_AL_LIMS[1]
                            th[38, 2] = S.SCH_AL_LIMS[0] - 1e-9
                        if variant == "free":
                            th[39, nd - 1] = S.ALPHA_LIMS[1]
                            th[40, nd - 1] = S.ALPHA_LIMS[1] + 1e-9
                            th[41, nd - 6] = S.FLIM_LIMS[0]
                            th[42, nd - 2] = S.FLIM_LIMS[1] + 1e-9
                            th[43:48, nd - 6] = np.linspace(1.2, 5.9, 5)   # Flim of field 0 ...
                            th[43:48:2, nd - 1] = 6.9                      # ... under a steep completeness curve
                    lnp = np.array([f(np.array(t)) for t in th])
            save(name, dict(variant=variant, n=n, seed=seed, zslices=zsl, fix_sch_al=fsa, min_comp_frac=mcf,
                            faint=int(faint), theta=th, lnprob=lnp))
            print(name, lnp[:3], "-inf rows", int(np.isinf(lnp).sum()))
    if only:
        with open(os.path.join(OUT, "MANIFEST.json"), "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        print("done (only %s)" % only)
        return

    # ---- FREE variant (lnprob, S=101): N = 50, 1000, 10000, with and without fixed alpha
    for (n, seed, fsa, nrows, tables) in ((50, 0, False, 24, True), (1000, 0, False, 40, True),
                                          (1000, 0, True, 24, False), (10000, 3, False, 24, False)):
        cat = S.catalogue(n, seed=seed)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            o = R.LumFuncMCMC(S.split_fields(cat["z"], cat["field_ind"]), **ctor_kwargs(cat, fsa, False, 0.0))
        th = theta_free(S, nrows, fsa, seed + 11)
        lnp, A, B, Bf = run_thetas(o, "lnprob", th, tapR, a_free)
        d = common_arrays(o, tables, False)
        d.update(theta=th, lnprob=lnp, A=A, B=B, B_fields=Bf, fix_sch_al=fsa, seed=seed,
                 sch_al0=S.SCH_AL)
        save("free_n%d%s" % (n, "_fsa" if fsa else ""), d)

    # ---- FIXCOMP variant (lnprob_fix_comp, S=201)
    for (n, seed, fsa, nrows, integ) in ((1000, 0, False, 40, True), (1000, 0, True, 24, False),
                                         (50, 0, False, 24, False)):
        cat = S.catalogue(n, seed=seed)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            o = R.LumFuncMCMC(S.split_fields(cat["z"], cat["field_ind"]), **ctor_kwargs(cat, fsa, True, 0.0))
        th = theta_fixcomp(S, nrows, fsa, seed + 21)
        lnp, A, B, Bf = run_thetas(o, "lnprob_fix_comp", th, tapR, a_fix)
        d = common_arrays(o, False, integ)
        if not integ:                         # keep the fixture small: the field-summed table is enough
            d["integ_sum"] = np.sum(np.array(o.integ_part), axis=0)
        d.update(theta=th, lnprob=lnp, A=A, B=B, B_fields=Bf, fix_sch_al=fsa, seed=seed,
                 sch_al0=S.SCH_AL)
        save("fixcomp_n%d%s" % (n, "_fsa" if fsa else ""), d)

    # ---- ZEVOL variant (LumFuncMCMCz.lnprob, S=201), two pivot sets (run_lumfuncmcmc_z.py:123-128)
    for (n, seed, fsa, nrows, piv, zsl) in ((1000, 0, False, 40, (1.20, 1.53, 1.86), 0),
                                            (1000, 0, True, 24, (1.20, 1.76, 2.32), 0),
                                            (800, 5, False, 24, (1.18, 1.36, 1.54), 8)):
        cat = S.catalogue(n, seed=seed, zslices=zsl)
        kw = ctor_kwargs(cat, fsa, False, 0.0)
        for k in ("fix_comp", "Flim_lims", "alpha_lims", "diff_rand"):
            kw.pop(k)
        kw.update(z1=piv[0], z2=piv[1], z3=piv[2])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with np.errstate(all="ignore"):
                o = RZ.LumFuncMCMCz(S.split_fields(cat["z"], cat["field_ind"]), **kw)
        th = theta_zevol(S, nrows, fsa, seed + 31)
        lnp, A, B, Bf = run_thetas(o, "lnprob", th, tapZ, a_z)
        d = common_arrays(o, False, False)
        d["integ_sum"] = np.sum(np.array(o.integ_part), axis=0)
        d.update(theta=th, lnprob=lnp, A=A, B=B, B_fields=Bf, fix_sch_al=fsa, seed=seed,
                 pivots=np.array(piv), zslices=zsl, sch_al0=S.SCH_AL)
        save("zevol_n%d%s" % (n, "_fsa" if fsa else ""), d)

    # ---- min_comp_frac = 0.5: per-field L grids differ, and the reference integrates every
    #      field on the LAST field's grid.  FREE + FIXCOMP on one catalogue.
    cat = S.catalogue(300, seed=7)
    for fc in (False, True):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with np.errstate(all="ignore"):
                o = R.LumFuncMCMC(S.split_fields(cat["z"], cat["field_ind"]), **ctor_kwargs(cat, False, fc, 0.5))
        if fc:
            th = theta_fixcomp(S, 16, False, 41)
            lnp, A, B, Bf = run_thetas(o, "lnprob_fix_comp", th, tapR, a_fix)
        else:
            th = theta_free(S, 16, False, 42)
            lnp, A, B, Bf = run_thetas(o, "lnprob", th, tapR, a_free)
        d = common_arrays(o, not fc, fc)
        d["roots_setup"] = o.rootsf.ev(np.asarray(S.FLIM), S.ALPHA_C)
        d["minlum_zarr"] = np.array([o.minlumf[i](o.zarr) for i in range(o.nfields)])
        d.update(theta=th, lnprob=lnp, A=A, B=B, B_fields=Bf, fix_sch_al=False, seed=7, sch_al0=S.SCH_AL)
        save("%s_n300_mcf50" % ("fixcomp" if fc else "free"), d)

    # ---- flux-input constructor path (what the drivers actually pass: run_lumfuncmcmc.py:245)
    cat = S.catalogue(200, seed=9)
    zz = cat["z"]
    dl = V.cosmo.luminosity_distance(zz).value
    flux17 = 10 ** cat["lum"] / (4.0 * np.pi * (dl * 3.086e24) ** 2) / 1.0e-17
    flux17_e = 0.1 * flux17
    kw = ctor_kwargs(cat, False, False, 0.0)
    kw.update(flux=S.split_fields(flux17, cat["field_ind"]), flux_e=S.split_fields(flux17_e, cat["field_ind"]),
              lum=None, lum_e=None)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        o = R.LumFuncMCMC(S.split_fields(zz, cat["field_ind"]), **kw)
    th = theta_free(S, 8, False, 51)[:12]
    lnp, A, B, Bf = run_thetas(o, "lnprob", th, tapR, a_free)
    d = common_arrays(o, True, False)
    d.update(theta=th, lnprob=lnp, A=A, B=B, B_fields=Bf, flux17=flux17, flux17_e=flux17_e,
             fix_sch_al=False, seed=9, sch_al0=S.SCH_AL)
    save("free_n200_fluxin", d)

    # ---- cosmology known answers (VmaxLumFunc.py:16-17) + Fleming curve samples (:95-167)
    zk = np.array([0.1, 0.5, 1.0, 1.102, 1.16, 1.5, 1.9, 1.995, 2.5])
    fl = np.logspace(-18.5, -15.0, 29)
    cosmo = {
        "H0": 70.0, "Om0": 0.3, "Ode0": 0.7, "Tcmb0": 2.725,
        "Ogamma0": float(V.cosmo.Ogamma0), "Onu0": float(V.cosmo.Onu0), "Ok0": float(V.cosmo.Ok0),
        "Neff": float(V.cosmo.Neff),
        "z": zk.tolist(),
        "DL_Mpc": V.cosmo.luminosity_distance(zk).value.tolist(),
        "dVc_dz_dOmega_Mpc3_sr": V.cosmo.differential_comoving_volume(zk).value.tolist(),
        "sqarcsec": float(V.sqarcsec),
        "fleming_f": fl.tolist(),
        "fleming_Flim2.72e-17_a4.56_fc0.1": V.fleming(fl, 2.72e-17, 4.56, 0.1).tolist(),
        "fleming_Flim3.3e-17_a2.0_fc0.1": V.fleming(fl, 3.3e-17, 2.0, 0.1).tolist(),
        "inverse_fleming_2.72e-17_4.56": float(V.inverse_fleming(2.72e-17, 4.56, 0.1)),
        "versions": {"numpy": np.__version__, "scipy": __import__("scipy").__version__,
                     "astropy": __import__("astropy").__version__, "python": sys.version.split()[0]},
    }
    with open(os.path.join(OUT, "cosmo_known.json"), "w") as f:
        json.dump(cosmo, f, indent=1)
    with open(os.path.join(OUT, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("done")


if __name__ == "__main__":
    main()
