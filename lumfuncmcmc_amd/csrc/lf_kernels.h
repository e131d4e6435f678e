// lf_kernels.h - HIP kernels of the per-step log-posterior (gfx950, fp64 VALU).
//
// Work decomposition (both pieces): a workgroup owns (one chunk of items) x (one tile of TW
// walkers).  Items - catalogue sources for piece A, integration-grid nodes for piece B - sit on
// the 64 lanes of each wave and are loaded once, coalesced, into registers; the walker loop is
// unrolled and every walker-only quantity is wave-uniform, so the compiler keeps it in SGPRs
// (scalar loads, no LDS traffic, no VGPRs).  Each lane keeps TW running sums; a wave
// __shfl_down tree, then a 4-wave LDS step, gives one partial per (chunk, walker), written to
// HBM in a fixed slot; lf_finalize adds the partials in a fixed order, so results are bitwise
// reproducible for a given launch geometry.
//
// The arithmetic is that of SURVEY.md App. A (reference: lumfuncmcmc.py:44, :69-70, :370-377,
// :388-392; lumfuncmcmc_z.py:40-42, :63-67, :371-375; VmaxLumFunc.py:118-127, :141, :164-167),
// evaluated in log space with everything source-only or walker-only hoisted:
//     10^(lum_i - L*)   = P_i * Q_w        P_i = 10^(lum_i - 42),  Q_w = 10^(42 - L*_w)
//     f_i / f_tau(w,f)  = U_i * V_wf       U_i = 10^(logf_i + 17), V_wf = 1 / (Flim_f 10^b_w)
// The reference's -inf (log of a product that underflowed to 0) is reproduced by poisoning the
// running sum with -inf whenever a factor or the product would round to zero in binary64.
#pragma once
#include "lf_math.h"

namespace lf {

constexpr int TW = 8;        // walkers per workgroup tile
constexpr int BLOCK = 256;   // 4 waves
constexpr int REC = 24;      // doubles per walker record
constexpr int MAXF = 8;

// walker record, FREE / FIXCOMP
enum { R_LSTAR = 0, R_C0 = 1, R_C1 = 2, R_Q = 3, R_ALPHAC = 4, R_LF = 8, R_V = 16 };
// walker record, ZEVOL
enum { Z_AL = 0, Z_BL = 1, Z_CL = 2, Z_AP = 3, Z_BP = 4, Z_CP = 5, Z_C1 = 6 };

struct KConst {
    int variant, fix_sch_al, nf, S, ndim;
    double lnom0_src[MAXF];   // ln(trunc(Omega_0[f]) / sqarcsec)   (int-truncated, lumfuncmcmc.py:285)
    double om0_grid[MAXF];    // Omega_0[f] / sqarcsec              (float, lumfuncmcmc.py:375)
    double fc_ratio;          // |a / (1 - a)|, a = (2 fcmin - 1)^2 (VmaxLumFunc.py:164-165)
    double lims[5][2];
    double pivots[3];
    double sch_al0, alpha0;
    double flim0[MAXF];
};

// getQuadCoef, lumfuncmcmc_z.py:40-42, with the reference's operation order and no FMA contraction
__device__ inline void quad_coef(double y1, double y2, double y3, double z1, double z2, double z3,
                                 double& a, double& b, double& c) {
#pragma clang fp contract(off)
    const double z1s = z1 * z1, z2s = z2 * z2, z3s = z3 * z3;
    a = ((y3 - y1) + (y2 - y1) * (z1 - z3) / (z2 - z1)) / (z3s - z1s + (z2s - z1s) * (z1 - z3) / (z2 - z1));
    b = (y2 - y1 - a * (z2s - z1s)) / (z2 - z1);
    c = y1 - a * z1s - b * z1;
}

// ----------------------------------------------------------------------------------------------
// prepare: theta rows -> walker records + prior flag.  One thread per (padded) walker.
// set_parameters_from_list + lnprior: lumfuncmcmc.py:327-358, lumfuncmcmc_z.py:339-362.
// ----------------------------------------------------------------------------------------------
__global__ void lf_prepare(KConst kc, const double* __restrict__ theta, int B, int Bpad,
                           double* __restrict__ wrec, int* __restrict__ prior_ok) {
    int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= Bpad) return;
    int ws = w < B ? w : B - 1;                 // padding walkers replay the last real one
    const double* th = theta + (size_t)ws * kc.ndim;
    double* r = wrec + (size_t)w * REC;
    for (int i = 0; i < REC; ++i) r[i] = 0.0;
    bool ok = true;
    if (kc.variant == LF_ZEVOL) {
        double L1 = th[0], L2 = th[1], L3 = th[2], p1 = th[3], p2 = th[4], p3 = th[5];
        double al = kc.fix_sch_al ? kc.sch_al0 : th[6];
        if (!kc.fix_sch_al) ok = ok && (al >= kc.lims[LF_LIM_SCH_AL][0]) && (al <= kc.lims[LF_LIM_SCH_AL][1]);
        const double Ls[3] = {L1, L2, L3}, ps[3] = {p1, p2, p3};
        for (int i = 0; i < 3; ++i) {           // strict for L and phi (lumfuncmcmc_z.py:355-358)
            ok = ok && (Ls[i] > kc.lims[LF_LIM_LSTAR][0]) && (Ls[i] < kc.lims[LF_LIM_LSTAR][1]);
            ok = ok && (ps[i] > kc.lims[LF_LIM_PHISTAR][0]) && (ps[i] < kc.lims[LF_LIM_PHISTAR][1]);
        }
        double z1 = kc.pivots[0], z2 = kc.pivots[1], z3 = kc.pivots[2];
        // getQuadCoef, lumfuncmcmc_z.py:40-42
        double aL, bL, cL, aP, bP, cP;
        quad_coef(L1, L2, L3, z1, z2, z3, aL, bL, cL);
        quad_coef(p1, p2, p3, z1, z2, z3, aP, bP, cP);
        r[Z_AL] = aL; r[Z_BL] = bL; r[Z_CL] = cL;
        r[Z_AP] = aP; r[Z_BP] = bP; r[Z_CP] = cP;
        r[Z_C1] = LF_LN10 * (al + 1.0);
    } else {
        double Lstar = th[0], phistar = th[1];
        int k = 2;
        double al = kc.fix_sch_al ? kc.sch_al0 : th[k++];
        double alphaC = kc.alpha0;
        double Flim[MAXF];
        for (int f = 0; f < kc.nf; ++f) Flim[f] = kc.flim0[f];
        if (kc.variant == LF_FREE) {
            for (int f = 0; f < kc.nf; ++f) Flim[f] = th[k + f];
            alphaC = th[k + kc.nf];
        }
        // inclusive box on all five named parameters, fixed ones too (lumfuncmcmc.py:346-354)
        ok = ok && (Lstar >= kc.lims[LF_LIM_LSTAR][0]) && (Lstar <= kc.lims[LF_LIM_LSTAR][1]);
        ok = ok && (phistar >= kc.lims[LF_LIM_PHISTAR][0]) && (phistar <= kc.lims[LF_LIM_PHISTAR][1]);
        ok = ok && (al >= kc.lims[LF_LIM_SCH_AL][0]) && (al <= kc.lims[LF_LIM_SCH_AL][1]);
        for (int f = 0; f < kc.nf; ++f)
            ok = ok && (Flim[f] >= kc.lims[LF_LIM_FLIM][0]) && (Flim[f] <= kc.lims[LF_LIM_FLIM][1]);
        ok = ok && (alphaC >= kc.lims[LF_LIM_ALPHA][0]) && (alphaC <= kc.lims[LF_LIM_ALPHA][1]);
        r[R_LSTAR] = Lstar;
        r[R_C0] = LF_LNLN10 + LF_LN10 * phistar;
        r[R_C1] = LF_LN10 * (al + 1.0);
        r[R_Q] = pow(10.0, LF_LREF - Lstar);
        if (kc.variant == LF_FREE) {
            r[R_ALPHAC] = alphaC;
            double b = -sqrt(kc.fc_ratio / (alphaC * alphaC));     // VmaxLumFunc.py:165
            double tenb = pow(10.0, b);
            for (int f = 0; f < kc.nf; ++f) {
                r[R_LF + f] = log10(1.0e-17 * Flim[f]);
                r[R_V + f] = 1.0 / (Flim[f] * tenb);
            }
        }
    }
    prior_ok[w] = ok ? 1 : 0;
}

// ----------------------------------------------------------------------------------------------
// shared pieces
// ----------------------------------------------------------------------------------------------
// a z^2 + b z + c with the reference's roundings (lumfuncmcmc_z.py:65-66): with close pivots the
// three terms cancel by two or three digits, so an FMA-contracted form drifts by ~1e-13 relative.
__device__ __forceinline__ double quad_nofma(double a, double b, double c, double z, double z2) {
    return __dadd_rn(__dadd_rn(__dmul_rn(a, z2), __dmul_rn(b, z)), c);
}

// ln of the Fleming completeness fc = 1/2 (1 + num / sqrt(1 + num^2)), VmaxLumFunc.py:118-120
__device__ __forceinline__ double ln_fc(double num) {
    double s = fma(num, num, 1.0);
    double q = num * drsqrt(s);
    return dlog(0.5 * (1.0 + q));
}

// block reduction of TW per-lane sums -> out[(tile*TW + w) * stride + chunk]
__device__ __forceinline__ void block_reduce_store(double (&acc)[TW], double* __restrict__ out,
                                                   size_t stride, int tile, int chunk) {
    __shared__ double red[BLOCK / 64][TW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int w = 0; w < TW; ++w) {
        double s = wave_sum(acc[w]);
        if (lane == 0) red[wave][w] = s;
    }
    __syncthreads();
    if (threadIdx.x < TW) {
        double s = red[0][threadIdx.x];
#pragma unroll
        for (int k = 1; k < BLOCK / 64; ++k) s += red[k][threadIdx.x];
        out[(size_t)(tile * TW + threadIdx.x) * stride + chunk] = s;
    }
}

// ----------------------------------------------------------------------------------------------
// piece A: per-source log-term sum.  grid = (chunks, walker tiles)
// lumfuncmcmc.py:370 (FREE), :388 (FIXCOMP), lumfuncmcmc_z.py:371 (ZEVOL)
// ----------------------------------------------------------------------------------------------
struct SrcArrays {
    const double* lum;    // [N]
    const double* a1;     // FREE: logf      FIXCOMP: ln(Om_arr)   ZEVOL: z
    const double* P;      // FREE/FIXCOMP: 10^(lum-42)             ZEVOL: ln(Om_arr)
    const double* U;      // FREE: 10^(logf+17)                    ZEVOL: z^2
    const int* chunk_start;
    const int* chunk_len;
    const int* chunk_field;
};

template <int VARIANT>
__global__ __launch_bounds__(BLOCK) void lf_srcsum(KConst kc, SrcArrays sa,
                                                   const double* __restrict__ wrec,
                                                   double* __restrict__ partial, int pstride) {
    const int c = blockIdx.x, tile = blockIdx.y;
    const int s0 = sa.chunk_start[c], n = sa.chunk_len[c], fld = sa.chunk_field[c];
    const double* __restrict__ wr = wrec + (size_t)tile * TW * REC;
    const double NEG_INF = -__builtin_huge_val();
    double acc[TW];
#pragma unroll
    for (int w = 0; w < TW; ++w) acc[w] = 0.0;

    for (int i = threadIdx.x; i < n; i += BLOCK) {
        const size_t g = (size_t)s0 + i;
        const double lum = sa.lum[g];
        if (VARIANT == LF_FREE) {
            const double logf = sa.a1[g], P = sa.P[g], U = sa.U[g];
            const double lnom0 = kc.lnom0_src[fld];
#pragma unroll
            for (int w = 0; w < TW; ++w) {
                const double* r = wr + w * REC;
                const double t = lum - r[R_LSTAR];
                const double v = P * r[R_Q];                       // 10^(lum - L*)
                const double lnT = fma(r[R_C1], t, r[R_C0]) - v;   // ln TrueLumFunc
                const double x = logf - r[R_LF + fld];             // log10(f / Flim)
                const double lnfc = ln_fc(r[R_ALPHAC] * x);
                const double u = U * r[R_V + fld];                 // f / f_tau
                const double d = 1.0 - dexp(-u);                   // expdecay, VmaxLumFunc.py:141
                const double lnOm = lnom0 + ddiv(lnfc, d);         // ln Omega
                const double term = lnT + lnOm;
                const bool bad = (v > LF_UNDERFLOW) | (lnT < -LF_UNDERFLOW) | (lnOm < -LF_UNDERFLOW) |
                                 (term < -LF_UNDERFLOW);
                acc[w] += bad ? NEG_INF : term;
            }
        } else if (VARIANT == LF_FIXCOMP) {
            const double lnOm = sa.a1[g], P = sa.P[g];
#pragma unroll
            for (int w = 0; w < TW; ++w) {
                const double* r = wr + w * REC;
                const double t = lum - r[R_LSTAR];
                const double v = P * r[R_Q];
                const double lnT = fma(r[R_C1], t, r[R_C0]) - v;
                const double term = lnT + lnOm;
                const bool bad = (v > LF_UNDERFLOW) | (lnT < -LF_UNDERFLOW) | (term < -LF_UNDERFLOW);
                acc[w] += bad ? NEG_INF : term;
            }
        } else {
            const double z = sa.a1[g], lnOm = sa.P[g], z2 = sa.U[g];
#pragma unroll
            for (int w = 0; w < TW; ++w) {
                const double* r = wr + w * REC;
                const double Ls = quad_nofma(r[Z_AL], r[Z_BL], r[Z_CL], z, z2);   // lumfuncmcmc_z.py:66
                const double ph = quad_nofma(r[Z_AP], r[Z_BP], r[Z_CP], z, z2);   // :65
                const double t = lum - Ls;
                const double v = dexp(LF_LN10 * t);
                const double lnT = fma(r[Z_C1], t, fma(LF_LN10, ph, LF_LNLN10)) - v;
                const double term = lnT + lnOm;
                const bool bad = (v > LF_UNDERFLOW) | (lnT < -LF_UNDERFLOW) | (term < -LF_UNDERFLOW);
                acc[w] += bad ? NEG_INF : term;
            }
        }
    }
    block_reduce_store(acc, partial, (size_t)pstride, tile, c);
}

// ----------------------------------------------------------------------------------------------
// piece B: expected-count integral on the S x S grid.  grid = (node chunks, walker tiles)
// trapz(trapz(I, logL, axis=0), zarr) = sum_jk W_jk I_jk with W from the actual grid spacings.
// lumfuncmcmc.py:373-377 (FREE), :389-392 (FIXCOMP), lumfuncmcmc_z.py:373-375 (ZEVOL)
// ----------------------------------------------------------------------------------------------
struct NodeArrays {
    const double* G;      // [S*S] logL
    const double* PG;     // 10^(G - 42)
    const double* W;      // FREE: trapz weight * volume_part[k]   else: trapz weight * sum_f integ_part[f]
    const double* a3;     // FREE: logf on the grid                ZEVOL: zarr[k]
    const double* a4;     // FREE: 10^(logf_grid + 17)             ZEVOL: zarr[k]^2
    int nnodes;
};

template <int VARIANT>
__global__ __launch_bounds__(BLOCK) void lf_gridsum(KConst kc, NodeArrays na,
                                                    const double* __restrict__ wrec,
                                                    double* __restrict__ partial, int pstride) {
    const int c = blockIdx.x, tile = blockIdx.y;
    const double* __restrict__ wr = wrec + (size_t)tile * TW * REC;
    double acc[TW];
#pragma unroll
    for (int w = 0; w < TW; ++w) acc[w] = 0.0;
    const int g = c * BLOCK + threadIdx.x;
    if (g < na.nnodes) {
        const double G = na.G[g], PG = na.PG[g], W = na.W[g];
        if (VARIANT == LF_FREE) {
            const double logf = na.a3[g], UG = na.a4[g];
#pragma unroll
            for (int w = 0; w < TW; ++w) {
                const double* r = wr + w * REC;
                const double t = G - r[R_LSTAR];
                const double T = dexp(fma(r[R_C1], t, r[R_C0]) - PG * r[R_Q]);
                double s = 0.0;
                for (int f = 0; f < kc.nf; ++f) {
                    const double lnfc = ln_fc(r[R_ALPHAC] * (logf - r[R_LF + f]));
                    const double d = 1.0 - dexp(-UG * r[R_V + f]);
                    s = fma(kc.om0_grid[f], dexp(ddiv(lnfc, d)), s);     // fc ** (1 / fc_decay)
                }
                acc[w] = fma(W * T, s, acc[w]);
            }
        } else if (VARIANT == LF_FIXCOMP) {
#pragma unroll
            for (int w = 0; w < TW; ++w) {
                const double* r = wr + w * REC;
                const double t = G - r[R_LSTAR];
                acc[w] = fma(W, dexp(fma(r[R_C1], t, r[R_C0]) - PG * r[R_Q]), acc[w]);
            }
        } else {
            const double z = na.a3[g], z2 = na.a4[g];
#pragma unroll
            for (int w = 0; w < TW; ++w) {
                const double* r = wr + w * REC;
                const double Ls = quad_nofma(r[Z_AL], r[Z_BL], r[Z_CL], z, z2);
                const double ph = quad_nofma(r[Z_AP], r[Z_BP], r[Z_CP], z, z2);
                const double t = G - Ls;
                const double lnT = fma(r[Z_C1], t, fma(LF_LN10, ph, LF_LNLN10)) - dexp(LF_LN10 * t);
                acc[w] = fma(W, dexp(lnT), acc[w]);
            }
        }
    }
    block_reduce_store(acc, partial, (size_t)pstride, tile, c);
}

// ----------------------------------------------------------------------------------------------
// finalize: one wave per walker; fixed-order sum of the partials; lnprob = lnprior + A - B.
// lumfuncmcmc.py:378, :403-409.  Never NaN (emcee raises on NaN): NaN -> -inf.
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void lf_finalize(const double* __restrict__ partA, int nchA, int strideA,
                                                  const double* __restrict__ partB, int nchB, int strideB,
                                                  const int* __restrict__ prior_ok, int B,
                                                  double* __restrict__ out, double* __restrict__ outA,
                                                  double* __restrict__ outB) {
    const int w = blockIdx.x;
    if (w >= B) return;
    const int lane = threadIdx.x;
    double a = 0.0, b = 0.0;
    const double* pa = partA + (size_t)w * strideA;
    const double* pb = partB + (size_t)w * strideB;
    for (int c = lane; c < nchA; c += 64) a += pa[c];
    for (int c = lane; c < nchB; c += 64) b += pb[c];
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane == 0) {
        const bool ok = prior_ok[w] != 0;
        double r = a - b;
        if (!ok || r != r) r = -__builtin_huge_val();
        if (out) out[w] = r;
        if (outA) outA[w] = ok ? a : __builtin_nan("");
        if (outB) outB[w] = ok ? b : __builtin_nan("");
    }
}

}  // namespace lf
