// lf_math.h - fp64 device math for the lnprob kernels (gfx950).
//
// Everything on the path is fp64 VALU work: there is no contraction, so no MFMA.
// Round-1 baseline: the ROCm device-library (ocml) routines, which are <= 1 ulp.
// The wrappers exist so that a tuned range-reduction + FMA-polynomial version can be
// swapped in kernel by kernel and A/B-checked against these.
#pragma once
#include <hip/hip_runtime.h>

#define LF_LN10 2.302585092994045684
#define LF_LNLN10 0.834032445247955959       // ln(ln 10)
#define LF_LOG10E 0.434294481903251828
// exp(-v) rounds to +0 in binary64 for v > ln(2^1075); log of a product below 2^-1075 is -inf.
#define LF_UNDERFLOW 745.13321910194122
#define LF_LREF 42.0        // P_i = 10^(lum_i - LF_LREF),  Q_w = 10^(LF_LREF - L*_w)
#define LF_FREF (-17.0)     // U_i = 10^(logf_i - LF_FREF)
#define LF_SQARCSEC 42545170296.152206       // (180/pi*3600)^2, VmaxLumFunc.py:43
#define LF_MPC_CM 3.086e24                   // lumfuncmcmc.py:70

namespace lf {

__device__ __forceinline__ double dexp(double x) { return exp(x); }
__device__ __forceinline__ double dlog(double x) { return log(x); }
__device__ __forceinline__ double drsqrt(double x) { return rsqrt(x); }
__device__ __forceinline__ double ddiv(double a, double b) { return a / b; }

// 64-lane wavefront sum (no masks on CDNA: every lane takes part).
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

}  // namespace lf
