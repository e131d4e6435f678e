"""lumfuncmcmc_amd - MI355X-native per-step log-posterior of LumFuncMCMC.

The hot path (LumFuncMCMC.lnprob / lnprob_fix_comp, LumFuncMCMCz.lnprob of the reference)
runs in hand-written HIP kernels for gfx950 behind a C ABI (include/lfmcmc.h, liblfmcmc.so);
this package is the Python host side: the ctypes binding, the reference's class surface and
the ensemble sampler that calls the batched boundary.
"""
__version__ = "0.1.0"

__all__ = ["capi", "synth", "build"]
