"""Drop-in module for `import VmaxLumFunc as V` (run_lumfuncmcmc.py:8, run_lumfuncmcmc_z.py:8): what the drivers and the
model classes read from it - the completeness curve (`V.fleming` in read_input_file, run_lumfuncmcmc.py:173), its
helpers, the cosmology object and the steradian -> square-arcsecond factor.  The 1/Veff estimator behind
LumFuncMCMC.VeffLF lives in lumfuncmcmc_amd.veff; the lmfit / matplotlib scripts of the reference's VmaxLumFunc.py
(:451-823) are out of scope."""
import numpy as np

from lumfuncmcmc_amd import hostsetup as _hs
from lumfuncmcmc_amd.cosmology import cosmo  # noqa: F401   (VmaxLumFunc.py:14-17)

sqarcsec = _hs.SQARCSEC                      # VmaxLumFunc.py:43
fleming = _hs.fleming                        # VmaxLumFunc.py:95-127
inverse_fleming = _hs.inverse_fleming        # VmaxLumFunc.py:143-167
TrueLumFunc = _hs.true_lum_func              # VmaxLumFunc.py:54-56 (same formula as lumfuncmcmc.py:25)


def expdecay(f, f_tau):
    """1 - exp(-f / f_tau), VmaxLumFunc.py:136-141."""
    return 1.0 - np.exp(-np.asarray(f, dtype=np.float64) / f_tau)
