"""Minimal LambdaCDM with photons and massless neutrinos: the two quantities the setup needs.

The reference takes them from astropy,
    cosmo = LambdaCDM(H0=70, Tcmb0=2.725, Om0=0.3, Ode0=0.7)              VmaxLumFunc.py:16-17
    cosmo.luminosity_distance(z), cosmo.differential_comoving_volume(z)   lumfuncmcmc.py:186-188
astropy is not a dependency here (absent from the system interpreter and from the GPU box), so
the same model is restated from its published definition: E(z)^2 = Or (1+z)^4 + Om (1+z)^3 +
Ok (1+z)^2 + Ode with Or = Ogamma0 (1 + 0.2271 Neff), Ogamma0 from Tcmb0 (CODATA 2018), Ok0 the
remainder (slightly closed for these inputs -> the sin branch).  Known answers recorded from
astropy 4.3.1 are in tests/golden/cosmo_known.json.

Integration: Gauss-Legendre, which is exact to rounding for this smooth integrand, instead of one
adaptive QUADPACK call per redshift - a table of 10^6 nodes takes a fraction of a second.
"""
import numpy as np

C_KMS = 299792.458
C_SI = 299792458.0
G_SI = 6.6743e-11                      # CODATA 2018
SIGMA_SB = 5.6703744191844314e-08      # CODATA 2018, W m^-2 K^-4
MPC_M = 3.085677581491367e+22


class LambdaCDM(object):
    def __init__(self, H0=70.0, Om0=0.3, Ode0=0.7, Tcmb0=2.725, Neff=3.04):
        self.H0, self.Om0, self.Ode0, self.Tcmb0, self.Neff = float(H0), float(Om0), float(Ode0), float(Tcmb0), float(Neff)
        H0_s = self.H0 * 1000.0 / MPC_M                            # 1/s
        self.critical_density0 = 3.0 * H0_s ** 2 / (8.0 * np.pi * G_SI)   # kg/m^3
        self.Ogamma0 = 4.0 * SIGMA_SB / C_SI ** 3 * self.Tcmb0 ** 4 / self.critical_density0
        self.Onu0 = 0.22710731766 * self.Neff * self.Ogamma0      # 7/8 (4/11)^(4/3), massless neutrinos
        self.Ok0 = 1.0 - self.Om0 - self.Ode0 - self.Ogamma0 - self.Onu0
        self.hubble_distance = C_KMS / self.H0                     # Mpc
        self._xg, self._wg = np.polynomial.legendre.leggauss(48)
        self._xs, self._ws = np.polynomial.legendre.leggauss(8)

    def inv_efunc(self, z):
        zp1 = 1.0 + np.asarray(z, dtype=np.float64)
        Or = self.Ogamma0 + self.Onu0
        return (zp1 ** 2 * ((Or * zp1 + self.Om0) * zp1 + self.Ok0) + self.Ode0) ** (-0.5)

    def efunc(self, z):
        return 1.0 / self.inv_efunc(z)

    def _gl(self, a, b, x, w):
        """Gauss-Legendre integral of inv_efunc over [a, b] (arrays)."""
        a = np.asarray(a, dtype=np.float64)[..., None]
        b = np.asarray(b, dtype=np.float64)[..., None]
        h = 0.5 * (b - a)
        return (h * self.inv_efunc(a + h * (x + 1.0)) * w).sum(axis=-1)

    def comoving_integral(self, z):
        """int_0^z dz'/E(z') for an array of redshifts, to rounding."""
        z = np.asarray(z, dtype=np.float64)
        flat = z.ravel()
        if flat.size <= 4096:
            # split [0, z] in unit-length pieces so that 48 points are far more than enough
            out = np.zeros_like(flat)
            nseg = np.maximum(1, np.ceil(np.abs(flat)).astype(int))
            for k in range(int(nseg.max())):
                m = nseg > k
                a = flat[m] * k / nseg[m]
                b = flat[m] * (k + 1) / nseg[m]
                out[m] += self._gl(a, b, self._xg, self._wg)
            return out.reshape(z.shape)
        # large tables: exact anchors every 1024 sorted points, short 8-point steps in between
        order = np.argsort(flat, kind="stable")
        zs = flat[order]
        res = np.empty_like(zs)
        anchors = np.arange(0, zs.size, 1024)
        base = self.comoving_integral(zs[anchors])
        for ai, start in enumerate(anchors):
            stop = min(start + 1024, zs.size)
            seg = self._gl(zs[start:stop - 1], zs[start + 1:stop], self._xs, self._ws)
            res[start] = base[ai]
            res[start + 1:stop] = base[ai] + np.cumsum(seg)
        out = np.empty_like(flat)
        out[order] = res
        return out.reshape(z.shape)

    def comoving_transverse_distance(self, z):
        dc = self.hubble_distance * self.comoving_integral(z)
        if self.Ok0 == 0:
            return dc
        s = np.sqrt(abs(self.Ok0))
        dh = self.hubble_distance
        if self.Ok0 > 0:
            return dh / s * np.sinh(s * dc / dh)
        return dh / s * np.sin(s * dc / dh)

    def luminosity_distance(self, z):
        """Mpc."""
        z = np.asarray(z, dtype=np.float64)
        return (1.0 + z) * self.comoving_transverse_distance(z)

    def differential_comoving_volume(self, z):
        """dVc/dz/dOmega in Mpc^3/sr."""
        dm = self.comoving_transverse_distance(z)
        return self.hubble_distance * dm ** 2 / self.efunc(z)


# the reference's module-level instance, VmaxLumFunc.py:16-17
cosmo = LambdaCDM(H0=70.0, Om0=0.3, Ode0=0.7, Tcmb0=2.725)
