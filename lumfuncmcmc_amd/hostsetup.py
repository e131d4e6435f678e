"""Host-side, one-off setup: everything lnlike reads, produced once per catalogue.

These are the outputs of the reference's constructor helpers (setDLdVdz, setOmegaLz,
setlnsimple, defineFlimOmArr, getRoot: lumfuncmcmc.py:180-235, :272-288;
lumfuncmcmc_z.py:226-305); they are the *inputs* of the HIP kernels, so they have to come out
the same (tests/test_hostsetup.py checks them against arrays recorded from the reference).
NumPy/SciPy on the host; nothing here runs per MCMC step.

Reference quirks kept on purpose: the distance tables have len(z) nodes; DLf
is a *linear* interpolant; Omega_0_arr is integer-truncated; every field integrates on the last
field's luminosity grid; integ_part goes through a bicubic spline of a 501x501 table.
"""
import numpy as np

from .cosmology import cosmo as _default_cosmo

LN10 = np.log(10.0)
SQARCSEC = (180. / np.pi * 3600.0) ** 2        # VmaxLumFunc.py:43
MPC_CM = 3.086e24                              # literal of lumfuncmcmc.py:70


# ------------------------------------------------------------------------------ elementwise physics
def true_lum_func(logL, alpha, logLstar, logphistar):
    """Schechter function per dex (lumfuncmcmc.py:44)."""
    t = logL - logLstar
    return LN10 * 10 ** logphistar * 10 ** (t * (alpha + 1)) * np.exp(-10 ** t)


def inverse_fleming(f50, alpha, fcmin=0.1):
    """Flux where the Fleming curve equals fcmin (VmaxLumFunc.py:164-167)."""
    a = (2 * fcmin - 1) ** 2.
    return f50 * 10 ** (-1 * (abs(a / (1 - a)) * alpha ** -2.) ** 0.5)


def fleming(f, Flim=3.0e-17, alpha=3.5, fcmin=0.1):
    """(Modified) Fleming completeness (VmaxLumFunc.py:116-127)."""
    if alpha is None:
        return np.ones(len(list(f)))
    num = alpha * np.log10(f / Flim)
    fc = 0.5 * (1. + num / (1. + num ** 2.) ** 0.5)
    if not fcmin:
        return fc
    return fc ** (1. / (1. - np.exp(-f / inverse_fleming(Flim, alpha, fcmin))))


def omega(logL, z, dLzfunc, Omega_0, Flim, alpha, fcmin=0.1):
    """Effective area fraction at (logL, z) (lumfuncmcmc.py:69-70)."""
    flux = 10 ** logL / (4.0 * np.pi * (MPC_CM * dLzfunc(z)) ** 2)
    return Omega_0 / SQARCSEC * fleming(flux, Flim, alpha, fcmin)


def get_quad_coef(y1, y2, y3, z1, z2, z3):
    """Quadratic through three pivots (lumfuncmcmc_z.py:40-42)."""
    a = ((y3 - y1) + (y2 - y1) * (z1 - z3) / (z2 - z1)) / (z3 ** 2 - z1 ** 2 + (z2 ** 2 - z1 ** 2) * (z1 - z3) / (z2 - z1))
    b = (y2 - y1 - a * (z2 ** 2 - z1 ** 2)) / (z2 - z1)
    c = y1 - a * z1 ** 2 - b * z1
    return a, b, c


def schechter_z(L, z, al, L1, L2, L3, phi1, phi2, phi3, z1, z2, z3):
    """Schechter function with L*(z), phi*(z) quadratic through the pivots (lumfuncmcmc_z.py:63-67)."""
    aphi, bphi, cphi = get_quad_coef(phi1, phi2, phi3, z1, z2, z3)
    alum, blum, clum = get_quad_coef(L1, L2, L3, z1, z2, z3)
    return true_lum_func(L, al, alum * z ** 2 + blum * z + clum, aphi * z ** 2 + bphi * z + cphi)


# ------------------------------------------------------------------------------ interpolation
class LinearInterp(object):
    """y(x) by linear interpolation on sorted nodes, out-of-range is an error: what
    scipy.interpolate.interp1d(x, y) evaluates (lumfuncmcmc.py:196-197), same formula
    slope * (x_new - x_lo) + y_lo so that the values agree to the last bit."""

    def __init__(self, x, y):
        self.x = np.asarray(x, dtype=np.float64)
        self.y = np.asarray(y, dtype=np.float64)
        if self.x.ndim != 1 or self.x.shape != self.y.shape or self.x.size < 2:
            raise ValueError("x and y must be 1-D arrays of equal length >= 2")

    def _right_node(self, q):
        """np.searchsorted(self.x, q) clipped to [1, n-1] - the index interp1d uses.  A binary search of N random
        queries in an N-node table is cache-miss bound (1 s at N = 10^6); when the nodes are evenly spaced the index
        is guessed arithmetically and corrected by +-1 against the actual nodes, which gives the very same index."""
        x, n = self.x, self.x.size
        if n < 4096 or q.size < 4096:
            return np.searchsorted(x, q).clip(1, n - 1)
        h = (x[-1] - x[0]) / (n - 1)
        if not (h > 0) or np.max(np.abs(np.diff(x) - h)) > 1e-6 * h:
            return np.searchsorted(x, q).clip(1, n - 1)
        hi = np.ceil((q - x[0]) / h).astype(np.int64).clip(1, n - 1)
        for _ in range(3):                           # searchsorted(left): x[hi-1] < q <= x[hi]
            up = (hi < n - 1) & (x[hi] < q)
            dn = (hi > 1) & (x[hi - 1] >= q)
            if not (up.any() or dn.any()):
                break
            hi = hi + up - dn
        return hi

    def __call__(self, xn):
        xn = np.asarray(xn, dtype=np.float64)
        flat = xn.ravel()
        if flat.size and (np.min(flat) < self.x[0] or np.max(flat) > self.x[-1]):
            raise ValueError("A value in x_new is outside the interpolation range.")
        hi = self._right_node(flat)
        lo = hi - 1
        slope = (self.y[hi] - self.y[lo]) / (self.x[hi] - self.x[lo])
        return (slope * (flat - self.x[lo]) + self.y[lo]).reshape(xn.shape)


# ------------------------------------------------------------------------------ setup stages
def distance_tables(z, cosmo=None):
    """setDLdVdz (lumfuncmcmc.py:183-197): len(z) nodes over [0.95 zmin, 1.05 zmax]."""
    cosmo = cosmo or _default_cosmo
    z = np.asarray(z, dtype=np.float64)
    zint = np.linspace(0.95 * z.min(), 1.05 * z.max(), len(z))
    DLarr = cosmo.luminosity_distance(zint)
    dVdzarr = cosmo.differential_comoving_volume(zint)
    return {"zint": zint, "DLarr": DLarr, "dVdzarr": dVdzarr,
            "DLf": LinearInterp(zint, DLarr), "dVdzf": LinearInterp(zint, dVdzarr),
            "DL": cosmo.luminosity_distance(z)}


def field_arrays(Flim, Omega_0, field_ind):
    """defineFlimOmArr (lumfuncmcmc.py:285-288): Omega_0_arr has dtype int (truncation)."""
    n = int(field_ind[-1])
    Flims_arr, Omega_0_arr = np.zeros(n), np.zeros(n, dtype=int)
    for ii in range(len(field_ind) - 1):
        Flims_arr[field_ind[ii]:field_ind[ii + 1]] = Flim[ii]
        Omega_0_arr[field_ind[ii]:field_ind[ii + 1]] = Omega_0[ii]
    return Flims_arr, Omega_0_arr


class _ZeroSurface(object):
    """rootsf when min_comp_frac <= 0.001: the spline of an all-zero table (lumfuncmcmc.py:276-281)."""

    def ev(self, x, y):
        return np.zeros(np.broadcast(np.asarray(x), np.asarray(y)).shape)


def completeness_roots(Flim_lims, alpha_lims, fcmin, min_comp_frac, size=201):
    """getRoot (lumfuncmcmc.py:272-281): minimum-flux surface over (Flim, alpha)."""
    if min_comp_frac <= 0.001:
        return _ZeroSurface()
    from scipy.interpolate import RectBivariateSpline
    from scipy.optimize import fsolve
    flims = np.linspace(Flim_lims[0], Flim_lims[1], size)
    alphas = np.linspace(alpha_lims[0], alpha_lims[1], size)
    roots = np.zeros((size, size))
    for i in range(size):
        for j in range(size):
            roots[i, j] = fsolve(lambda x: fleming(x, 1.0e-17 * flims[i], alphas[j], fcmin) - min_comp_frac,
                                 [3.0e-17])[0]
    return RectBivariateSpline(flims, alphas, roots)


def omega_splines(DLf, zmin, zmax, Lc, Lh, Omega_0, Flim, alpha, fcmin, size=501):
    """setOmegaLz (lumfuncmcmc.py:206-215): one bicubic spline of Omega(logL, z) per field."""
    from scipy.interpolate import RectBivariateSpline
    logL = np.linspace(Lc, Lh, size)
    zarr = np.linspace(0.95 * zmin, 1.05 * zmax, size)
    area = 4.0 * np.pi * (MPC_CM * DLf(zarr)) ** 2
    flux = 10 ** logL[:, None] / area[None, :]
    out = []
    for ii in range(len(Flim)):
        tab = Omega_0[ii] / SQARCSEC * fleming(flux, 1.0e-17 * Flim[ii], alpha, fcmin)
        out.append(RectBivariateSpline(logL, zarr, tab))
    return out


def integration_grid(size_ln, zmin, zmax, lum_min, Lh, DLf, dVdzf, minlumf, Omegaf=None):
    """setlnsimple (lumfuncmcmc.py:219-234).  Returns zarr, DL_zarr, volume_part, zarr_rep, the
    list `logL` (nf references to ONE array, as in the reference) and integ_part (or None when
    no splines were given - the free-completeness likelihood never reads it)."""
    S = size_ln
    zarr = np.linspace(zmin, zmax, S)
    DL_zarr = DLf(zarr)
    volume_part = dVdzf(zarr)
    zarr_rep = np.repeat(zarr[None], S, axis=0)
    grid = np.empty((S, S))
    logL, integ_part = [], []
    for ii in range(len(minlumf)):
        lo = np.array(minlumf[ii](zarr), dtype=np.float64)
        lo[lo < lum_min] = lum_min
        for i in range(S):
            grid[:, i] = np.linspace(lo[i], Lh, S)
        logL.append(grid)                          # same object every time: the aliasing of App. B-2
        if Omegaf is not None:
            integ_part.append(volume_part * Omegaf[ii].ev(grid, zarr_rep))
    return {"zarr": zarr, "DL_zarr": DL_zarr, "volume_part": volume_part, "zarr_rep": zarr_rep,
            "logL": logL, "integ_part": integ_part if Omegaf is not None else None}


def lum_from_flux(flux, flux_e, DL):
    """getLumin (lumfuncmcmc.py:255-260); the error is first-order propagation, which is what the
    `uncertainties` package computes for log10."""
    area = 4.0 * np.pi * (DL * MPC_CM) ** 2
    lum = np.log10(area * flux)
    lum_e = None if flux_e is None else (area * flux_e) / (np.abs(area * flux) * LN10)
    return lum, lum_e


def flux_from_lum(lum, lum_e, DL):
    """getFluxes (lumfuncmcmc.py:264-270)."""
    area = 4.0 * np.pi * (DL * MPC_CM) ** 2
    L = 10 ** lum
    flux = L / area
    flux_e = None if lum_e is None else (LN10 * L * lum_e) / area
    return flux, flux_e
