"""1/Veff estimator of the binned luminosity function with bootstrap errors (post-fit diagnostic).

Reference: LumFuncMCMC.VeffLF (lumfuncmcmc.py:515-525, lumfuncmcmc_z.py:470-478) ->
V.lumfunc (VmaxLumFunc.py:235-257), V.getMaxz (:761-777), V.getBootErrLog (:304-378).

The reference calls scipy.quad once per source on an integrand whose only z dependence is the
interpolated dV/dz - the completeness is evaluated at the source's OBSERVED flux, a constant of
the integral (VmaxLumFunc.py:215-233).  So the weight of source i is

    phi_i = 1 / ( sum(Omega_0)/sqarcsec * fleming(F_i, Flim_i, alpha) * int_{zmin}^{zmax_i} dV/dz dz )

and with min_comp_frac <= 0.001 the integral is the same for every source: one quadrature instead
of N.  The bootstrap is a bincount per resample instead of nboot x nbins boolean masks over the
catalogue.  Host NumPy is O(N) (N = 10^6: ~1 s; the reference needs N quad calls plus 5000 N-long masks);
`veff_device` does the weights, the binning and the nboot resamples in two HIP kernels (lf_veff).
"""
import numpy as np

from . import hostsetup as hs


def _interp_integral(f, a, b):
    """Exact integral of the piecewise-linear interpolant f (hostsetup.LinearInterp) over [a, b_i]."""
    x, y = f.x, f.y
    seg = 0.5 * (y[1:] + y[:-1]) * np.diff(x)
    cum = np.concatenate([[0.0], np.cumsum(seg)])

    def F(t):
        t = np.asarray(t, dtype=np.float64)
        hi = np.searchsorted(x, t).clip(1, len(x) - 1)
        lo = hi - 1
        yt = f(t)
        return cum[lo] + 0.5 * (y[lo] + yt) * (t - x[lo])
    return F(b) - F(a)


MPC_CM_EXACT = 3.0856775814913673e24      # astropy's Mpc in cm: get_L_constF uses .to('cm'), not the 3.086e24 literal


def max_redshift_fsolve(lum_lin, fmin, cosmo, z0=1.5):
    """The reference's own procedure (V.getMaxz, VmaxLumFunc.py:761-777): one scipy fsolve of
    get_L_constF(fmin, z) - L per source, from z = 1.5.  Kept for the tests; 2 ms per source."""
    from scipy.optimize import fsolve
    out = np.empty(len(lum_lin))
    for i, (L, fm) in enumerate(zip(lum_lin, fmin)):
        out[i] = fsolve(lambda x: 4.0 * np.pi * (cosmo.luminosity_distance(x) * MPC_CM_EXACT) ** 2 * fm - L, z0)[0]
    return out


def max_redshift(lum_lin, fmin, cosmo, z0=1.5):
    """Redshift at which luminosity lum_lin (erg/s) is seen at flux fmin: the root of
    4 pi (DL(z) Mpc)^2 fmin = L that V.getMaxz finds with fsolve, for all sources at once.  DL is monotonic, so
    the root is DL^-1 of a target distance: a table gives the first guess, Newton steps on the exact DL (secant
    slope from the table) polish it to 1e-13 - the reference's fsolve stops at 1.5e-8."""
    lum_lin = np.asarray(lum_lin, dtype=np.float64)
    fmin = np.asarray(fmin, dtype=np.float64)
    target = np.sqrt(lum_lin / (4.0 * np.pi * fmin)) / MPC_CM_EXACT            # Mpc
    out = np.full(target.shape, np.nan)
    good = np.isfinite(target) & (target > 0)
    if not good.any():
        return out
    zhi = 4.0
    while cosmo.luminosity_distance(np.array([zhi]))[0] < target[good].max() and zhi < 1.0e4:
        zhi *= 2.0
    ztab = np.linspace(0.0, zhi, 8193)
    dtab = np.asarray(cosmo.luminosity_distance(ztab))
    slope = np.gradient(dtab, ztab)
    z = np.interp(target[good], dtab, ztab)
    for _ in range(6):
        step = (np.asarray(cosmo.luminosity_distance(z)) - target[good]) / np.interp(z, ztab, slope)
        z = np.clip(z - step, 0.0, zhi)
        if np.max(np.abs(step)) < 1e-13 * max(1.0, zhi):
            break
    out[good] = z
    return out


def comoving_volume(dVdzf, zmin, zmaxval):
    """int_{zmin}^{zmax} dV/dz dz of the docstring: one scipy.quad for a scalar zmax (as V.lumfunc evaluates it), the
    exact integral of the interpolant per source otherwise; 0 where zmax <= zmin."""
    if np.ndim(zmaxval) == 0:
        if not zmaxval > zmin:
            return 0.0
        from scipy.integrate import quad
        return quad(lambda z: float(dVdzf(z)), zmin, zmaxval)[0]
    ok = zmaxval > zmin
    return np.where(ok, _interp_integral(dVdzf, zmin, np.where(ok, zmaxval, zmin)), 0.0)


def lumfunc_weights(flux, dVdzf, sum_omega, zmin, zmaxval, flim, alpha, fcmin):
    """phi_i of the docstring.  zmaxval: scalar (shared integral, evaluated with scipy.quad exactly
    as V.lumfunc does) or per-source array (exact integral of the interpolant; the reference's
    adaptive quadrature agrees with it to its own tolerance, ~1e-8)."""
    comp = hs.fleming(flux, flim, alpha, fcmin)
    pref = sum_omega / hs.SQARCSEC * comp
    phi = np.zeros_like(flux)
    if np.ndim(zmaxval) == 0:
        if zmaxval > zmin:
            from scipy.integrate import quad
            vol, _ = quad(lambda z: float(dVdzf(z)), zmin, zmaxval)
            phi = 1.0 / (pref * vol)
        return phi
    ok = zmaxval > zmin
    vol = _interp_integral(dVdzf, zmin, np.where(ok, zmaxval, zmin))
    with np.errstate(divide="ignore"):
        phi[ok] = 1.0 / (pref[ok] * vol[ok])
    return phi


def luminosity_bins(L, nbin):
    """Bin edges, centres, width and the bin index of every source (nbin = no bin) of V.getBootErrLog."""
    L = np.asarray(L, dtype=np.float64)
    Larr = np.linspace(min(L) * 1.001, max(L), nbin + 1)
    Lavg = np.linspace((Larr[0] + Larr[1]) / 2.0, (Larr[-1] + Larr[-2]) / 2.0, len(Larr) - 1)
    idx = np.searchsorted(Larr, L, side="right") - 1
    idx[(L < Larr[0]) | (L >= Larr[-1])] = nbin
    return Larr, Lavg, Lavg[1] - Lavg[0], idx


def veff_device(L, flux, flim, vol, sum_omega, alpha, fcmin, nboot=100, nbin=25, seed=None, boot_idx=None, device=0):
    """VeffLF on the GPU (lf_veff of include/lfmcmc.h): the per-source weights, the binned LF and the bootstrap sums in
    two kernels.  vol: the comoving-volume integral, scalar (min_comp_frac <= 0.001) or per source.  The resamples are
    drawn on the device (Philox keyed by `seed`, default: one draw from numpy's global state) unless `boot_idx`
    (nboot, N) is given - e.g. the reference's own seeded np.random.randint stream, which then reproduces
    V.getBootErrLog's variances exactly.  Returns (phifunc, Lavg, lfbinorig, var) as lumfunc_weights + boot_err_log do."""
    from .capi import veff_device as _dev
    _, Lavg, dL, idx = luminosity_bins(L, nbin)
    if seed is None:
        seed = int(np.random.randint(0, 2 ** 31 - 1))
    phi, sums = _dev(flux, flim, vol, sum_omega / hs.SQARCSEC, alpha, fcmin, bin_of=idx, nbin=nbin, nboot=nboot,
                     boot_idx=boot_idx, seed=seed, device=device)
    lfbin = sums[1:] / dL
    binavg = np.average(lfbin, axis=0)
    var = 1. / (nboot - 1) * np.sum((lfbin - binavg) ** 2, axis=0)
    var[var <= 0.0] = min(var[var > 0.0])
    return phi, Lavg, sums[0] / dL, var


def boot_err_log(L, phi, nboot=100, nbin=25):
    """Binned LF dn/dlogL, and bootstrap variances (V.getBootErrLog with correct_low=False).
    Bin edges linspace(min(L)*1.001, max(L), nbin+1), half-open bins [e_j, e_j+1): the brightest
    source and anything below min(L)*1.001 fall in no bin, as in the reference.  One
    np.random.randint(N, size=N) per resample, in order - the reference's use of the global state."""
    L = np.asarray(L, dtype=np.float64)
    phi = np.asarray(phi, dtype=np.float64)
    Larr = np.linspace(min(L) * 1.001, max(L), nbin + 1)
    Lavg = np.linspace((Larr[0] + Larr[1]) / 2.0, (Larr[-1] + Larr[-2]) / 2.0, len(Larr) - 1)
    dL = Lavg[1] - Lavg[0]
    idx = np.searchsorted(Larr, L, side="right") - 1
    idx[(L < Larr[0]) | (L >= Larr[-1])] = nbin            # overflow slot, dropped below
    lfbinorig = np.bincount(idx, weights=phi, minlength=nbin + 1)[:nbin] / dL
    lfbin = np.zeros((nboot, nbin))
    for k in range(nboot):
        boot = np.random.randint(len(phi), size=len(phi))
        lfbin[k] = np.bincount(idx[boot], weights=phi[boot], minlength=nbin + 1)[:nbin] / dL
    binavg = np.average(lfbin, axis=0)
    var = 1. / (nboot - 1) * np.sum((lfbin - binavg) ** 2, axis=0)
    var[var <= 0.0] = min(var[var > 0.0])
    return Lavg, lfbinorig, var
