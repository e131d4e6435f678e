// lf_math.h - fp64 device math for the lnprob kernels (gfx950).
//
// Everything on the path is fp64 VALU work: there is no contraction, so no MFMA.  The terms are
// dominated by exp / log / rsqrt / reciprocal, so the kernels carry their own versions, sized
// for this path (known argument ranges, ~1 ulp, no special-case branches):
//   fexp_t     table-driven: 256-entry 2^(j/256) table in LDS, magic-number rounding (no cvt),
//              degree-4 polynomial on |r| <= ln2/512, ldexp
//   flog_half  table-driven: 256 x {1/c, log c} in LDS, degree-5 log1p on |r| < 2^-9
//   frsqrt     v_rsq_f64 seed (2^-24 measured on gfx950) + one cubic step
//   frcp       v_rcp_f64 seed (2^-24) + one cubic Newton step
//   TermTables piecewise degree-7 polynomials of the two univariate factors of the free-completeness term,
//              g(num) = ln fc and h(y) = 1 / (1 - e^(-10^y)) (lf_tables.h, gen_tables.py): the per-source fast path
// The "careful" path (rare walkers that may underflow, see lf_kernels.h) uses the ROCm device
// library (ocml) versions; tests compare both against the oracle.
#pragma once
#include <hip/hip_runtime.h>

#include "lf_tables.h"

#define LF_LN10 2.302585092994045684
#define LF_LNLN10 0.834032445247955959       // ln(ln 10)
#define LF_LN2 0.693147180559945309
// exp(-v) rounds to +0 in binary64 for v > ln(2^1075); log of a product below 2^-1075 is -inf.
#define LF_UNDERFLOW 745.13321910194122
#define LF_LREF 42.0        // P_i = 10^(lum_i - LF_LREF),  Q_w = 10^(LF_LREF - L*_w)
#define LF_FREF (-17.0)     // U_i = 10^(logf_i - LF_FREF)
#define LF_SQARCSEC 42545170296.152206       // (180/pi*3600)^2, VmaxLumFunc.py:43
#define LF_MPC_CM 3.086e24                   // lumfuncmcmc.py:70

namespace lf {

// ---- careful versions (device library)
__device__ __forceinline__ double dexp(double x) { return exp(x); }
__device__ __forceinline__ double dlog(double x) { return log(x); }
__device__ __forceinline__ double drsqrt(double x) { return rsqrt(x); }
__device__ __forceinline__ double ddiv(double a, double b) { return a / b; }

// ---- fast versions
// LDS image of the tables of lf_tables.h
struct MathTables {
    double2 logt[256];   // {1/c_j, log c_j}
    double expt[256];    // 2^(j/256)
};

// e^x, table-driven: 256 x / ln2 = 256 n + j + (256 / ln2) r, |r| <= ln2/512, e^x = 2^n 2^(j/256) e^r.
// Valid for |x| < 2^22 (results below 2^-1075 come out as 0 through ldexp); no clamps, NaN in -> NaN out.
__device__ __forceinline__ double fexp_t(double x, const MathTables* __restrict__ mt) {
    const double MAGIC = 6755399441055744.0;                     // 1.5 * 2^52
    const double t = fma(x, 369.32993046757463, MAGIC);          // 256 / ln2; low word of t = round(256 x / ln2)
    const double kd = t - MAGIC;
    double r = fma(kd, -0.00270760617331689, x);          // ln2_hi / 256 (kd * hi exact)
    r = fma(kd, -7.453964567463233e-13, r);                 // ln2_lo / 256
    const int k = __double2loint(t);
    const double T = mt->expt[k & 255];
    double q = fma(r, 4.16666666666666666667e-02, 1.66666666666666666667e-01);
    q = fma(q, r, 0.5);
    const double p = fma(r * r, q, r);                           // e^r - 1, |r|^5/120 < 4e-17
    return ldexp(fma(T, p, T), k >> 8);
}

// e^(-u) for the completeness decay 1 - e^(-u), u >= 0.  One-constant reduction: the error of
// ln2/256 as a double adds u * 1.1e-16 relative to e^(-u), i.e. at most 4e-17 absolute to 1 - e^(-u).
__device__ __forceinline__ double fexp_neg(double u, const MathTables* __restrict__ mt) {
    const double MAGIC = 6755399441055744.0;
    const double t = fma(u, -369.32993046757463, MAGIC);
    const double kd = t - MAGIC;
    const double r = fma(kd, -0.0027076061740622863, -u);   // ln2 / 256
    const int k = __double2loint(t);
    const double T = mt->expt[k & 255];
    double q = fma(r, 4.16666666666666666667e-02, 1.66666666666666666667e-01);
    q = fma(q, r, 0.5);
    const double p = fma(r * r, q, r);
    return ldexp(fma(T, p, T), k >> 8);
}

// the same with the argument clamped to [-750, 709] (grid kernels: arguments are not pre-screened)
__device__ __forceinline__ double fexp_c(double x, const MathTables* __restrict__ mt) {
    return fexp_t(fmin(fmax(x, -750.0), 709.0), mt);
}

// ln(w / 2) for w in (0, 2]; w = 0 returns about -710 (finite).
__device__ __forceinline__ double flog_half(double w, const MathTables* __restrict__ mt) {
    const int hi = __double2hiint(w), lo = __double2loint(w);
    const int e = ((hi >> 20) & 0x7ff) - 1024;                       // exponent of w/2
    const int j = (hi >> 12) & 0xff;
    const double m = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, lo);   // mantissa in [1, 2)
    const double2 t = mt->logt[j];
    const double r = fma(m, t.x, -1.0);
    double p = fma(r, 0.2, -0.25);                                   // |r| < 2^-9: r^6/6 < 1e-17
    // (a scalar-register constant made HERE: left to the compiler it is hoisted out of lf_free's item loop into a vector
    // register pair and spilled - the one scratch slot of that kernel)
    double third = 1.0 / 3.0;
    asm volatile("" : "+s"(third));
    p = fma(p, r, third);
    p = fma(p, r, -0.5);
    const double r2 = r * r;
    const double hi_part = fma((double)e, LF_LN2, t.y);
    return hi_part + fma(p, r2, r);
}

// ln(w / 2) for w in [1, 2] (flux at or above the 50 % limit): the exponent of w/2 is known, so no exponent
// extraction, no int -> double conversion, and the argument is its own mantissa.  w = 2 exactly (fc = 1 to
// rounding) is nudged one ulp down: ln gives -1.1e-16 instead of 0.
__device__ __forceinline__ double flog_half_upper(double w, const MathTables* __restrict__ mt) {
    const double m = fmin(w, 1.9999999999999998);
    const int j = (__double2hiint(m) >> 12) & 0xff;
    const double2 t = mt->logt[j];
    const double r = fma(m, t.x, -1.0);
    double p = fma(r, 0.2, -0.25);
    p = fma(p, r, 1.0 / 3.0);
    p = fma(p, r, -0.5);
    const double r2 = r * r;
    return (t.y - LF_LN2) + fma(p, r2, r);
}

// 1/sqrt(s), s >= 1: v_rsq_f64 seed (2^-24, measured) + one cubic step (error ~ e^3)
__device__ __forceinline__ double frsqrt(double s) {
    const double z0 = __builtin_amdgcn_rsq(s);
    const double e = fma(-(s * z0), z0, 1.0);
    const double p = fma(0.375, e, 0.5);
    return fma(z0, p * e, z0);
}

// 1/d for normal d: v_rcp_f64 seed (2^-24) + one cubic Newton step
__device__ __forceinline__ double frcp(double d) {
    const double y0 = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, y0, 1.0);
    return fma(y0, fma(e, e, e), y0);
}

// LDS image of G_TABLE / H_TABLE (lf_tables.h): 8 coefficients (64 B) per piece
struct TermTables {
    double g[G_N * 8];
    double h[H_N * 8];
};
// All loads are issued before the first LDS store (fixed trip counts, NT threads): the prologue of a workgroup pays ONE
// round trip to L2 for the 27 KB, not one per loop iteration.
// (t: this thread's rank among the NT that take part; default: the whole workgroup)
template <int NT>
__device__ __forceinline__ void load_term_tables(TermTables* tt, int t = -1) {
    double2* g2 = reinterpret_cast<double2*>(tt->g);
    double2* h2 = reinterpret_cast<double2*>(tt->h);
    const double2* G2 = reinterpret_cast<const double2*>(G_TABLE);
    const double2* H2 = reinterpret_cast<const double2*>(H_TABLE);
    constexpr int NG = (G_N * 4 + NT - 1) / NT, NH = (H_N * 4 + NT - 1) / NT;
    double2 vg[NG], vh[NH];
    if (t < 0) t = threadIdx.x;
#pragma unroll
    for (int q = 0; q < NG; ++q) vg[q] = G2[min(t + q * NT, G_N * 4 - 1)];
#pragma unroll
    for (int q = 0; q < NH; ++q) vh[q] = H2[min(t + q * NT, H_N * 4 - 1)];
    // (every load is issued before the first is waited for: without this the compiler sinks each load into the guarded
    // block of its store - seven dependent round trips to a cold L2 for the workgroup's tables, 3.8 us of a 19-us launch)
#pragma unroll
    for (int q = 0; q < NG; ++q) asm volatile("" : "+v"(vg[q].x), "+v"(vg[q].y));
#pragma unroll
    for (int q = 0; q < NH; ++q) asm volatile("" : "+v"(vh[q].x), "+v"(vh[q].y));
#pragma unroll
    for (int q = 0; q < NG; ++q)
        if (t + q * NT < G_N * 4) g2[t + q * NT] = vg[q];
#pragma unroll
    for (int q = 0; q < NH; ++q)
        if (t + q * NT < H_N * 4) h2[t + q * NT] = vh[q];
}

// degree-7 Horner from 8 coefficients held in registers
__device__ __forceinline__ double horner7(const double (&c)[8], double t) {
    double p = fma(c[7], t, c[6]);
    p = fma(p, t, c[5]);
    p = fma(p, t, c[4]);
    p = fma(p, t, c[3]);
    p = fma(p, t, c[2]);
    p = fma(p, t, c[1]);
    return fma(p, t, c[0]);
}

// copy the tables to LDS (call with all threads, then __syncthreads())
__device__ __forceinline__ void load_tables(MathTables* mt) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        mt->logt[i] = make_double2(LOG_TABLE[2 * i], LOG_TABLE[2 * i + 1]);
        mt->expt[i] = EXP_TABLE[i];
    }
}
// the same for workgroups of exactly 256 threads: one straight-line pass
__device__ __forceinline__ void load_tables_256(MathTables* mt) {
    const int i = threadIdx.x;
    const double2 l = *reinterpret_cast<const double2*>(LOG_TABLE + 2 * i);
    const double e = EXP_TABLE[i];
    mt->logt[i] = l;
    mt->expt[i] = e;
}

// 64-lane wavefront sum on the DPP network (no LDS crossbar): row_shr 1, 2, 4, 8 inside the 16-lane rows, then
// row_bcast15 / row_bcast31 across them; the total arrives in lane 63.  The __shfl_down tree of wave_sum goes through
// ds_bpermute - twelve LDS round trips per double - and took 3.9k cycles at the end of every workgroup-wide item.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_shift_add(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    // lanes that receive nothing (bound_ctrl off, old = 0) add 0: each step is a valid partial sum in the upper lanes
    v = dpp_shift_add<0x111, 0xf>(v);     // row_shr:1
    v = dpp_shift_add<0x112, 0xf>(v);     // row_shr:2
    v = dpp_shift_add<0x114, 0xf>(v);     // row_shr:4
    v = dpp_shift_add<0x118, 0xf>(v);     // row_shr:8   -> lane 15 of every row holds the row's sum
    v = dpp_shift_add<0x142, 0xa>(v);     // row_bcast:15 into rows 1 and 3
    v = dpp_shift_add<0x143, 0xc>(v);     // row_bcast:31 into rows 2 and 3  -> lane 63 holds the total
    return v;
}

// 64-lane wavefront sum (no masks on CDNA: every lane takes part).
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// ----------------------------------------------------------------------------------------------
// Kernel arguments: one round trip instead of a dozen.  A launch's kernarg segment is freshly written memory - nothing of
// it is in the scalar cache or in L2 - and these kernels take ~1.5 KB of arguments (KConst alone is 1360 bytes) which the
// compiler fetches with s_load instructions hoisted to the top of the kernel, one s_waitcnt after the other because it
// spills them to VGPR lanes as they arrive: twelve dependent misses in lf_free's preamble, ~0.6 us each from device
// memory (measured: 19.0 us per evaluation with the kernarg ring in device memory, 32.2 us with it in host memory,
// HIP_FORCE_DEV_KERNARG=0: thirteen misses' worth of PCIe).  warm_kernarg issues loads of the whole segment back to
// back, the results discarded, and waits once: afterwards every line is in the scalar cache (and in L2 for the vector
// loads of fields indexed per lane).  BYTES = the size of the explicit arguments; the over-read up to the next multiple
// of 256 stays inside the segment (the hidden arguments that follow are 256 bytes).
// ----------------------------------------------------------------------------------------------
#define LF_KW1(o) "s_load_dword %0, %1, " #o "\n\t"
#define LF_KW4(b) LF_KW1(b + 0x00) LF_KW1(b + 0x40) LF_KW1(b + 0x80) LF_KW1(b + 0xc0)
template <int BYTES>
__device__ __forceinline__ void warm_kernarg() {
    constexpr int N4 = (BYTES + 255) / 256;
    static_assert(N4 >= 1 && N4 <= 10, "extend the ladder");
    auto p = __builtin_amdgcn_kernarg_segment_ptr();
    int t;      // (one dword per 64-byte line: the line comes into the cache whatever the size of the load - and the cache's return
                // path, shared by the 32 waves of two CUs that all do this at once, carries 4 bytes per line instead of 64)
    if constexpr (N4 == 1) asm volatile(LF_KW4(0x000) "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(p) : "memory");
    if constexpr (N4 == 2) asm volatile(LF_KW4(0x000) LF_KW4(0x100) "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(p) : "memory");
    if constexpr (N4 == 3) asm volatile(LF_KW4(0x000) LF_KW4(0x100) LF_KW4(0x200) "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(p) : "memory");
    if constexpr (N4 == 4)
        asm volatile(LF_KW4(0x000) LF_KW4(0x100) LF_KW4(0x200) LF_KW4(0x300) "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(p) : "memory");
    if constexpr (N4 == 5)
        asm volatile(LF_KW4(0x000) LF_KW4(0x100) LF_KW4(0x200) LF_KW4(0x300) LF_KW4(0x400) "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(p) : "memory");
    if constexpr (N4 == 6)
        asm volatile(LF_KW4(0x000) LF_KW4(0x100) LF_KW4(0x200) LF_KW4(0x300) LF_KW4(0x400) LF_KW4(0x500) "s_waitcnt lgkmcnt(0)"
                     : "=&s"(t) : "s"(p) : "memory");
    if constexpr (N4 == 7)
        asm volatile(LF_KW4(0x000) LF_KW4(0x100) LF_KW4(0x200) LF_KW4(0x300) LF_KW4(0x400) LF_KW4(0x500) LF_KW4(0x600) "s_waitcnt lgkmcnt(0)"
                     : "=&s"(t) : "s"(p) : "memory");
    if constexpr (N4 == 8)
        asm volatile(LF_KW4(0x000) LF_KW4(0x100) LF_KW4(0x200) LF_KW4(0x300) LF_KW4(0x400) LF_KW4(0x500) LF_KW4(0x600) LF_KW4(0x700)
                     "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(p) : "memory");
    if constexpr (N4 == 9)
        asm volatile(LF_KW4(0x000) LF_KW4(0x100) LF_KW4(0x200) LF_KW4(0x300) LF_KW4(0x400) LF_KW4(0x500) LF_KW4(0x600) LF_KW4(0x700)
                     LF_KW4(0x800) "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(p) : "memory");
    if constexpr (N4 == 10)
        asm volatile(LF_KW4(0x000) LF_KW4(0x100) LF_KW4(0x200) LF_KW4(0x300) LF_KW4(0x400) LF_KW4(0x500) LF_KW4(0x600) LF_KW4(0x700)
                     LF_KW4(0x800) LF_KW4(0x900) "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(p) : "memory");
}

}  // namespace lf
