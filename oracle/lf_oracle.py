"""CPU oracle: NumPy restatement of the reference's per-step log-posterior.

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this module; the product path (lumfuncmcmc_amd/) never does and fails
loudly when the HIP library is missing.

Parity status: PINNED.  Every function below is checked in tests/test_oracle_golden.py
against vectors recorded from the reference itself (oracle/gen_golden.py ->
tests/golden/*.npz; numpy 1.26.4 / scipy 1.7.1 / astropy 4.3.1).

The arithmetic follows the reference statement by statement (same operation order, linear
space then log, the same -inf on underflow); the only liberty is that DLf(z_i) - which
the reference re-interpolates on every call (lumfuncmcmc.py:70 via :370) although it
does not depend on theta - is taken as a precomputed input array `DLz`.

Input dict `inp` (all float64 unless noted) - the same quantities the C-ABI descriptor
(include/lfmcmc.h) takes:
    variant            'free' | 'fixcomp' | 'zevol'
    fix_sch_al, sch_al0
    field_ind[nf+1]    int64
    lum[N], z[N], DLz[N]
    Omega_0[nf]        float effective areas (sq arcsec)
    Om_arr[N]          fixcomp / zevol
    Flim0[nf], alpha0  fixed completeness parameters (prior of fixcomp tests them too)
    size_ln, logL[S,S], zarr[S], DL_zarr[S], volume_part[S]
    integ_part[nf,S,S] or integ_sum[S,S]   fixcomp / zevol
    fcmin, lims{Lstar,phistar,sch_al,Flim,alpha}, pivots(z1,z2,z3)
"""
import numpy as np

LN10 = np.log(10.0)
SQARCSEC = (180. / np.pi * 3600.0) ** 2          # VmaxLumFunc.py:43
MPC_CM = 3.086e24                                # lumfuncmcmc.py:70


# ----------------------------------------------------------------------------- elementwise
def true_lum_func(logL, alpha, logLstar, logphistar):
    """lumfuncmcmc.py:44 (dup lumfuncmcmc_z.py:88)."""
    return LN10 * 10 ** logphistar * 10 ** ((logL - logLstar) * (alpha + 1)) * np.exp(-10 ** (logL - logLstar))


def inverse_fleming(f50, alpha, fcmin=0.1):
    """VmaxLumFunc.py:164-167."""
    a = (2 * fcmin - 1) ** 2.
    b = -1 * (abs(a / (1 - a)) * alpha ** -2.) ** 0.5
    return f50 * 10 ** b


def fleming(f, Flim, alpha, fcmin=0.1):
    """VmaxLumFunc.py:118-127 (fcmin truthy on this path) and :141."""
    numerator = alpha * np.log10(f / Flim)
    denominator = (1. + numerator ** 2.) ** 0.5
    fc = 0.5 * (1. + numerator / denominator)
    if not fcmin:
        return fc
    f_tau = inverse_fleming(Flim, alpha, fcmin)
    fc_decay = 1. - np.exp(-f / f_tau)
    return fc ** (1. / fc_decay)


def omega(logL, DL, Omega_0, Flim, alpha, fcmin=0.1):
    """lumfuncmcmc.py:69-70 with DL = DLf(z) already interpolated (Mpc)."""
    L = 10 ** logL
    return Omega_0 / SQARCSEC * fleming(L / (4.0 * np.pi * (MPC_CM * DL) ** 2), Flim, alpha, fcmin)


def get_quad_coef(y1, y2, y3, z1, z2, z3):
    """lumfuncmcmc_z.py:40-42."""
    a = ((y3 - y1) + (y2 - y1) * (z1 - z3) / (z2 - z1)) / (z3 ** 2 - z1 ** 2 + (z2 ** 2 - z1 ** 2) * (z1 - z3) / (z2 - z1))
    b = (y2 - y1 - a * (z2 ** 2 - z1 ** 2)) / (z2 - z1)
    c = y1 - a * z1 ** 2 - b * z1
    return a, b, c


def schechter_z(L, z, al, L1, L2, L3, phi1, phi2, phi3, z1, z2, z3):
    """lumfuncmcmc_z.py:63-67."""
    aphi, bphi, cphi = get_quad_coef(phi1, phi2, phi3, z1, z2, z3)
    alum, blum, clum = get_quad_coef(L1, L2, L3, z1, z2, z3)
    phistar = aphi * z ** 2 + bphi * z + cphi
    Lstar = alum * z ** 2 + blum * z + clum
    return true_lum_func(L, al, Lstar, phistar)


def trapz(y, x, axis=-1):
    """numpy/scipy composite trapezoid as scipy 1.7.1 `trapz` evaluates it
    (d = diff(x); sum(d * (y[1:] + y[:-1]) / 2)) - used at lumfuncmcmc.py:377."""
    y = np.asarray(y)
    x = np.asarray(x)
    nd = y.ndim
    if x.ndim == 1:
        d = np.diff(x)
        shape = [1] * nd
        shape[axis] = d.shape[0]
        d = d.reshape(shape)
    else:
        d = np.diff(x, axis=axis)
    s1 = [slice(None)] * nd
    s2 = [slice(None)] * nd
    s1[axis] = slice(1, None)
    s2[axis] = slice(None, -1)
    return (d * (y[tuple(s1)] + y[tuple(s2)]) / 2.0).sum(axis)


# ----------------------------------------------------------------------------- parameters
def split_theta(inp, theta):
    """set_parameters_from_list: lumfuncmcmc.py:327-337, lumfuncmcmc_z.py:339-341."""
    v = inp["variant"]
    fsa = bool(inp["fix_sch_al"])
    nf = len(inp["field_ind"]) - 1
    p = {}
    if v in ("free", "fixcomp"):
        p["Lstar"], p["phistar"] = theta[0], theta[1]
        k = 2
        if fsa:
            p["sch_al"] = inp["sch_al0"]
        else:
            p["sch_al"] = theta[2]
            k = 3
        if v == "free":
            p["Flim"], p["alpha"] = theta[k:k + nf], theta[k + nf]
        else:
            p["Flim"], p["alpha"] = np.asarray(inp["Flim0"], dtype=float), inp["alpha0"]
    else:
        p["L"] = theta[0:3]
        p["phi"] = theta[3:6]
        p["sch_al"] = inp["sch_al0"] if fsa else theta[6]
    return p


def lnprior(inp, p):
    """lumfuncmcmc.py:346-358 (inclusive, fixed parameters tested too);
    lumfuncmcmc_z.py:350-362 (alpha inclusive, L/phi strict)."""
    lims = inp["lims"]
    if inp["variant"] in ("free", "fixcomp"):
        flag = 1.0
        for name in ("Lstar", "phistar", "sch_al"):
            flag *= (p[name] >= lims[name][0]) * (p[name] <= lims[name][1])
        for F in p["Flim"]:
            flag *= (F >= lims["Flim"][0]) * (F <= lims["Flim"][1])
        flag *= (p["alpha"] >= lims["alpha"][0]) * (p["alpha"] <= lims["alpha"][1])
    else:
        if inp["fix_sch_al"]:
            flag = 1
        else:
            flag = (p["sch_al"] >= lims["sch_al"][0]) * (p["sch_al"] <= lims["sch_al"][1])
        for i in range(3):
            flag *= (p["L"][i] > lims["Lstar"][0]) * (p["L"][i] < lims["Lstar"][1])
            flag *= (p["phi"][i] > lims["phistar"][0]) * (p["phi"][i] < lims["phistar"][1])
    return 0.0 if flag else -np.inf


# ----------------------------------------------------------------------------- pieces
def _field_scatter(vals, field_ind, dtype=float):
    """defineFlimOmArr / getFlim: lumfuncmcmc.py:285-293 (Omega_0_arr is dtype=int)."""
    out = np.zeros(field_ind[-1], dtype=dtype)
    for ii in range(len(field_ind) - 1):
        out[field_ind[ii]:field_ind[ii + 1]] = vals[ii]
    return out


def piece_a(inp, p):
    """Per-source log-term sum: lumfuncmcmc.py:370, :388; lumfuncmcmc_z.py:371."""
    v = inp["variant"]
    lum = inp["lum"]
    with np.errstate(all="ignore"):
        if v == "free":
            fi = inp["field_ind"]
            om0_arr = _field_scatter(inp["Omega_0"], fi, dtype=int)      # int truncation (App. B-1)
            flims_arr = _field_scatter(p["Flim"], fi)
            om = omega(lum, inp["DLz"], om0_arr, 1.0e-17 * flims_arr, p["alpha"], inp["fcmin"])
            return np.log(true_lum_func(lum, p["sch_al"], p["Lstar"], p["phistar"]) * om).sum()
        if v == "fixcomp":
            return np.log(true_lum_func(lum, p["sch_al"], p["Lstar"], p["phistar"]) * inp["Om_arr"]).sum()
        z1, z2, z3 = inp["pivots"]
        return np.log(schechter_z(lum, inp["z"], p["sch_al"], p["L"][0], p["L"][1], p["L"][2],
                                  p["phi"][0], p["phi"][1], p["phi"][2], z1, z2, z3) * inp["Om_arr"]).sum()


def piece_b(inp, p):
    """Expected-count integral: lumfuncmcmc.py:373-377, :389-392; lumfuncmcmc_z.py:373-375.
    All fields integrate on the one aliased logL grid (App. B-2)."""
    v = inp["variant"]
    nf = len(inp["field_ind"]) - 1
    logL, zarr = inp["logL"], inp["zarr"]
    S = logL.shape[0]
    fullint = 0.0
    with np.errstate(all="ignore"):
        if v == "free":
            tlf = true_lum_func(logL, p["sch_al"], p["Lstar"], p["phistar"])
            DLg = np.repeat(inp["DL_zarr"][None], S, axis=0)       # DLf(zarr_rep)
            for ii in range(nf):
                integ_part = inp["volume_part"] * omega(logL, DLg, inp["Omega_0"][ii], 1.0e-17 * p["Flim"][ii],
                                                        p["alpha"], inp["fcmin"])
                fullint += trapz(trapz(tlf * integ_part, logL, axis=0), zarr)
            return fullint
        if v == "fixcomp":
            tlf = true_lum_func(logL, p["sch_al"], p["Lstar"], p["phistar"])
        else:
            z1, z2, z3 = inp["pivots"]
            zrep = np.repeat(zarr[None], S, axis=0)
            tlf = schechter_z(logL, zrep, p["sch_al"], p["L"][0], p["L"][1], p["L"][2],
                              p["phi"][0], p["phi"][1], p["phi"][2], z1, z2, z3)
        if "integ_part" in inp and inp["integ_part"] is not None:
            for ii in range(nf):
                fullint += trapz(trapz(tlf * inp["integ_part"][ii], logL, axis=0), zarr)
        else:   # fixtures that only keep the field-summed table
            fullint += trapz(trapz(tlf * inp["integ_sum"], logL, axis=0), zarr)
    return fullint


def lnprob(inp, theta, pieces=False):
    """lumfuncmcmc.py:395-424, lumfuncmcmc_z.py:378-392: one theta row -> float."""
    theta = np.asarray(theta, dtype=float)
    p = split_theta(inp, theta)
    lp = lnprior(inp, p)
    if not np.isfinite(lp):
        return (-np.inf, np.nan, np.nan) if pieces else -np.inf
    A = piece_a(inp, p)
    B = piece_b(inp, p)
    r = (A - B) + lp
    return (r, A, B) if pieces else r


def lnprob_batch(inp, thetas, pieces=False):
    """Scalar loop over rows, exactly the reference's execution model under emcee."""
    thetas = np.atleast_2d(np.asarray(thetas, dtype=float))
    if pieces:
        out = np.array([lnprob(inp, t, True) for t in thetas])
        return out[:, 0], out[:, 1], out[:, 2]
    return np.array([lnprob(inp, t) for t in thetas])


# ----------------------------------------------------------------------------- fixtures
DEFAULT_LIMS = {"Lstar": [40.0, 45.0], "phistar": [-8.0, 5.0], "sch_al": [-3.0, 1.0],
                "Flim": [1.0, 6.0], "alpha": [1.0, 7.0]}          # configLF.py:9,13,24-28


def inputs_from_golden(g, variant):
    """Build an oracle input dict from a tests/golden/*.npz record."""
    inp = {
        "variant": variant, "fix_sch_al": bool(g["fix_sch_al"]), "sch_al0": float(g["sch_al0"]),
        "field_ind": np.asarray(g["field_ind"], dtype=np.int64),
        "lum": g["lum"], "z": g["z"], "DLz": g["DLz"], "Omega_0": g["Omega_0"],
        "Om_arr": g["Om_arr"], "Flim0": g["Flim0"], "alpha0": float(g["alpha0"]),
        "size_ln": int(g["size_ln"]), "logL": g["logL"], "zarr": g["zarr"], "DL_zarr": g["DL_zarr"],
        "volume_part": g["volume_part"], "fcmin": float(g["fcmin"]),
        "lims": dict(DEFAULT_LIMS), "pivots": tuple(g["pivots"]) if "pivots" in g else (1.20, 1.53, 1.86),
        "integ_part": g["integ_part"] if "integ_part" in g else None,
        "integ_sum": g["integ_sum"] if "integ_sum" in g else None,
    }
    return inp
