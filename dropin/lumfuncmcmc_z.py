"""Drop-in module for `from lumfuncmcmc_z import LumFuncMCMCz` (run_lumfuncmcmc_z.py:7)."""
from lumfuncmcmc_amd.model import LumFuncMCMCz, Omega, TrueLumFunc, getQuadCoef, schechter_z  # noqa: F401
