"""Drop-in module: put this directory ahead of the reference checkout on PYTHONPATH and
`from lumfuncmcmc import LumFuncMCMC` (run_lumfuncmcmc.py:7) resolves to the MI355X path.
See INTEGRATION.md."""
from lumfuncmcmc_amd.model import LumFuncMCMC, Omega, TrueLumFunc  # noqa: F401
