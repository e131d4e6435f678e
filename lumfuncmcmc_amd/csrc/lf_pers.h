// lf_pers.h - the z-evolving and the fixed-completeness lnprob (lumfuncmcmc_z.py:364-392, lumfuncmcmc.py:380-393,
// :411-424) in PERSISTENT 512-thread workgroups, one launch per plain evaluation (gfx950).
//
// Shaped like lf_free (lf_free.h), for the same reasons: after the cells in redshift (z-evolving)
// and the closed form of piece A (fixed completeness) an evaluation is a latency chain, and lf_prepare -> lf_main ->
// lf_finalize paid three kernel boundaries, three sets of cold kernel arguments and workgroup-wide LDS reductions for it.
// Here a tile of 8 walkers is served by a group of workgroups (one of rank r on each of ... every CU holds two):
//   prologue   wave 0 prepares the tile's walkers (lf_prepare's body: prepare_lane, records straight to LDS), the other
//              waves bring the exp table in;
//   columns    (z-evolving) wave v = walker v makes Q_wk = 10^(42 - L*_w(z_k)) and E_wk for all S redshift columns of the
//              grid (the grid integrand depends on the walker through its column only, lf_kernels.h: gridsum_body);
//   cells      (z-evolving) chunks of 64 cells in redshift, lane = cell, dealt statically over the tile's workgroups;
//   grid       chunks of 64 nodes, lane = node, ONE exponential per node; a wave adds up its lanes over all its chunks and
//              reduces once (DPP) - one partial per (walker, workgroup), written through to memory;
//   sources    only for walkers that need the per-source checks (careful path) - rare; the tile's workgroups stride over
//              the catalogue together;
//   epilogue   the last workgroup to finish the tile adds the partials up (finalize_wave: lf_finalize's body).
// FUSED = false keeps lf_prepare and lf_finalize as launches of their own around the same kernel (the sampler's propose /
// accept steps, lf_lnprob_pieces): the same partial sums in the same slots, so the same bits as the one-launch form.
#pragma once
#include "lf_free.h"

namespace lf {

constexpr int PERS_MAXS = 256;        // z-evolving: redshift columns whose (Q, E) fit the LDS table (the reference's S is 201)

struct PersArgs {
    int B, ntiles;
    int nchB;                 // node chunks (64 nodes)
    int nchC, ncell;          // z-evolving: chunks of 64 cells / cells (0: no cells)
    int nslot;                // partial sums per walker and piece: one per workgroup serving the walker's tile
    int tile_stride;          // workgroup g serves tiles (g / 8) % ntiles, + tile_stride, ...
    int* queues;              // [ntiles][QSTRIDE]: [0] = workgroups that have finished the tile (FUSED)
    double* partA;            // [B][nslot] per-source sums (walkers on the careful path only)
    double* partB;            // [B][nslot] the grid integral
    double* partC;            // [B][nslot] z-evolving: the cells' sums
    const double* nodes4;     // [nchB * 64][4] {G, 10^(G - 42), W, redshift column}: the grid's nodes, padded to whole chunks (pads: W = 0)
    const double* zcol;       // [S][2] z-evolving: {z_k, z_k^2}
    const double* cells;      // [ncell][8] z-evolving: {z_c, S_0 .. S_6}
    const double* lum;        // the catalogue (careful path): [N] each, sources of a field contiguous
    const double* a1;         //   FIXCOMP: ln(Om_arr)   ZEVOL: z
    const double* P;          //   FIXCOMP: 10^(lum-42)  ZEVOL: ln(Om_arr)
    const double* U;          //                         ZEVOL: z^2
    const double* theta;      // FUSED: [B][ndim]
    double* out;              // FUSED: [B] lnprob
    const double* wrec;       // !FUSED: lf_prepare's records
    const int* wmode;
    const int* wstat;
    int poll;                 // 1: the slots of partB / partC are PART_EMPTY: tiles without a careful path hand over by polling (lf_free.h)
    int* err;
};

// STEP (with FUSED): the launch is a half-step of the device-resident sampler (lf_free.h: lf_free_body)
template <int VARIANT, bool FUSED, bool STEP>
__device__ __forceinline__ void lf_pers_body(const KConst& kc, const PersArgs& pa, const StepArgs& sp, const AcceptArgs& ap) {
    static_assert(VARIANT == LF_ZEVOL || VARIANT == LF_FIXCOMP, "the free variant has lf_free");
    auto pstore = [](double* p, double v) {                 // (see lf_free: partial sums are written THROUGH in the one-launch form)
        if (FUSED) __hip_atomic_store(p, v == v ? v : __builtin_nan(""), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (never PART_EMPTY)
        else *p = v;
    };
    __shared__ MathTables tab;
    __shared__ __attribute__((aligned(16))) double wsc[PTW * 8];              // per walker: ZEVOL aL bL cL aP bP cP c1 / FIXCOMP L* c0 c1 Q
    __shared__ __attribute__((aligned(16))) double wfc[PTW * MAXF * 8];       // per (walker, field): [0] ZEVOL slope bound, ints at [4]: mode
    __shared__ __attribute__((aligned(16))) double qe[VARIANT == LF_ZEVOL ? PTW * PERS_MAXS * 2 : 2];   // per (walker, column): {Q, E}
    __shared__ __attribute__((aligned(16))) double red[PB];                   // prepare's staging; the careful path's reduction
    __shared__ __attribute__((aligned(16))) double zcl[VARIANT == LF_ZEVOL ? PERS_MAXS * 2 : 2];      // the columns' {z, z^2}
    __shared__ int sstat[PTW];
    __shared__ double sbase[PTW];
    __shared__ double sprop[PTW * 16];     // STEP: the tile's proposals and stretch factors (the accept step's)
    __shared__ double szz[PTW];
    __shared__ double spre[PTW * 2];       //       ... and the accept step's two logarithms per walker, made ahead (lf_free.h)
    __shared__ int smask[4];               // bit w: [0] walker on the cells, [1] needs the sources, [2] outside the prior (no grid)
    __shared__ int sdone;
    const int tid = threadIdx.x;
    const int wave_base = __builtin_amdgcn_readfirstlane(tid & ~63);
    const int lane = tid & 63, v = wave_base >> 6;
    bool tables_loaded = false;
#ifdef LF_STAMPS
    unsigned long long* stamp = kc.stamps ? kc.stamps + (size_t)blockIdx.x * 8 : nullptr;
    unsigned long long t_prep = 0, t_cols = 0, t_cells = 0, t_grid = 0;
    if (stamp && tid == 0) {
        stamp[0] = __builtin_amdgcn_s_memtime();
        stamp[5] = __builtin_amdgcn_s_memrealtime();
    }
#endif

#pragma unroll 1
    for (int tile = ((int)blockIdx.x >> 3) % pa.ntiles; tile < pa.ntiles; tile += pa.tile_stride) {
        const int w0 = tile * PTW;
        const int nw = min(PTW, pa.B - w0);
        __builtin_assume(nw >= 1 && nw <= PTW);
        int* __restrict__ q = pa.queues + tile * QSTRIDE;
        // rank of this workgroup among the fgroup that serve the tile (lf_free.h)
        int fgroup = 8, frank = (int)blockIdx.x & 7;
        if (pa.ntiles <= pa.tile_stride) {
            const int k = (int)blockIdx.x >> 3;
            fgroup = 8 * ((pa.tile_stride - tile + pa.ntiles - 1) / pa.ntiles);
            frank += 8 * ((k - tile) / pa.ntiles);
        }
        __syncthreads();                          // the previous tile's last reads of the LDS records are done
        // ---- prologue: the tile's records in LDS; the exp table (first tile)
        if (FUSED) {
            // (the preparing wave: wave 0 in the launch's first 256 workgroups, wave 1 in the rest - the two workgroups of a CU
            // keep their lone dependent chains on different SIMDs, lf_free.h)
            const int pw = (int)(blockIdx.x >> 8) & 1;
            const int up = tid - 64 * pw;
            if (up >= 0 && up < 64) {
                prepare_lane<false, true, STEP, VARIANT>(kc, sp, pa.theta, pa.B, nullptr, nullptr, nullptr, nullptr, nullptr, 0, w0 + (up >> 3), up & 7,
                                                up >> 3, reinterpret_cast<double(*)[16]>(red), wfc, wsc, sstat, sbase, nullptr, nullptr, sprop, szz);
            } else if (!tables_loaded && tid >= 128 && tid < 128 + 256) {
                // (the thread number made anew: from `tid` the compiler lifts the tables' addresses out of the tile loop, up into the
                // kernel's preamble - and in the step kernel, short of registers, spills them there and fetches them back here one
                // by one, each load of a table behind its own trip to scratch: three dependent round trips instead of one)
                int fm = -1;
                asm volatile("" : "+s"(fm));
                const int t = wave_base + __builtin_amdgcn_mbcnt_hi(fm, __builtin_amdgcn_mbcnt_lo(fm, 0)) - 128;
                double2 lt = *reinterpret_cast<const double2*>(LOG_TABLE + 2 * t);
                double et = EXP_TABLE[t];
                double2 zz = VARIANT == LF_ZEVOL ? *reinterpret_cast<const double2*>(pa.zcol + 2 * min(t, kc.S - 1)) : double2{0.0, 0.0};
                asm volatile("" : "+v"(lt.x), "+v"(lt.y), "+v"(et), "+v"(zz.x), "+v"(zz.y));      // (one round trip for all of them)
                tab.logt[t] = lt;
                tab.expt[t] = et;
                if (VARIANT == LF_ZEVOL) *reinterpret_cast<double2*>(zcl + 2 * t) = zz;
            }
        } else {
            if (!tables_loaded && tid >= 256) {
                const int t = tid - 256;
                double2 lt = *reinterpret_cast<const double2*>(LOG_TABLE + 2 * t);
                double et = EXP_TABLE[t];
                double2 zz = VARIANT == LF_ZEVOL ? *reinterpret_cast<const double2*>(pa.zcol + 2 * min(t, kc.S - 1)) : double2{0.0, 0.0};
                asm volatile("" : "+v"(lt.x), "+v"(lt.y), "+v"(et), "+v"(zz.x), "+v"(zz.y));
                tab.logt[t] = lt;
                tab.expt[t] = et;
                if (VARIANT == LF_ZEVOL) *reinterpret_cast<double2*>(zcl + 2 * t) = zz;
            }
            if (tid < nw * MAXF) {                // (walker, field): slope bound and mode
                const int w = tid / MAXF, f = tid - w * MAXF;
                wfc[tid * 8] = pa.wrec[(size_t)(w0 + w) * REC + RF(f, 0)];
                reinterpret_cast<int*>(wfc + tid * 8 + 4)[M_MODE] = pa.wmode[((size_t)(w0 + w) * MAXF + f) * WM + M_MODE];
            }
            if (tid >= 64 && tid < 128) {         // walker scalars
                const int t = tid - 64, w = t >> 3, j = t & 7;
                if (w < nw) wsc[t] = pa.wrec[(size_t)(w0 + w) * REC + j];
            }
            if (tid >= 128 && tid < 128 + nw) sstat[tid - 128] = pa.wstat[w0 + tid - 128];
        }
        if (FUSED && STEP && tid >= PB - 64 && tid < PB - 64 + nw) {
            const int wl = tid - (PB - 64);
            unsigned int rr[4];
            sampler_draw(sp.step, sp.half, w0 + wl, 0, sp.seed, rr);
            accept_terms(ap, w0 + wl, stretch_z(sp.a, u53(rr[0], rr[1])), spre[2 * wl], spre[2 * wl + 1]);
        }
        tables_loaded = true;
        __syncthreads();
        if (tid < 64) {
            const int st = sstat[min(tid, nw - 1)];
            const bool in = tid < nw;
            const bool live = in && (st & STAT_PRIOR_OK) && !(st & STAT_NEGINF);
            const bool on = VARIANT == LF_ZEVOL && pa.nchC > 0 && in && (st & STAT_CELLS);
            const bool need = live && (VARIANT == LF_FIXCOMP ? (st & STAT_SLOW) != 0 : !on);
            const int mc = (int)__ballot(on), mn = (int)__ballot(need), mo = (int)__ballot(in && !(st & STAT_PRIOR_OK));
            if (tid == 0) {
                smask[0] = mc;
                smask[1] = mn;
                smask[2] = mo;
            }
        }
        __syncthreads();
#ifdef LF_STAMPS
        t_prep = __builtin_amdgcn_s_memtime();
#endif
        const int cellmask = uni(smask[0]), needmask = uni(smask[1]), outmask = uni(smask[2]);
        const bool mine_w = v < nw;               // this wave's walker exists
        const bool grid_w = mine_w && !((outmask >> v) & 1);      // (outside the prior: the grid is not evaluated, lumfuncmcmc.py:408)
        const double* __restrict__ sc = wsc + v * 8;

#ifdef LF_STAMPS
        t_cols = __builtin_amdgcn_s_memtime();
#endif
        // ---- z-evolving: the cells in redshift (lf_kernels.h: ZCELL_M), for walkers flagged STAT_CELLS.  Dealt from the
        // middle rank up, the grid's chunks from rank 0 (lf_free.h: the younger workgroups of a CU run behind the elders)
        if (VARIANT == LF_ZEVOL && pa.nchC > 0 && mine_w) {
#pragma unroll 1
            for (int vr = frank; vr < VF; vr += fgroup) {         // (virtual ranks: lf_free.h)
            double acc = 0.0;
            const int cfirst = (vr - VF / 2 + VF) % VF;
            if ((cellmask >> v) & 1) {
                const double aL = uni(sc[Z_AL]), bL = uni(sc[Z_BL]), cL = uni(sc[Z_CL]);
                auto load_cell = [&](double (&d)[8], int cc) {
                    const int i = cc * 64 + lane;
                    const double2* __restrict__ src = reinterpret_cast<const double2*>(pa.cells + (size_t)min(i, pa.ncell - 1) * 8);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double2 a = src[k];
                        d[2 * k] = i < pa.ncell || k == 0 ? a.x : 0.0;      // (z_c always; a lane past the end: every power sum 0)
                        d[2 * k + 1] = i < pa.ncell ? a.y : 0.0;
                    }
                };
                // (one chunk ahead; two were measured and gained nothing - the phase is bound by issue, not by the loads - and
                // their 16 registers pushed the kernel into scratch)
                double nx[8];
                if (cfirst < pa.nchC) load_cell(nx, cfirst);
#pragma unroll 1
                for (int cc = cfirst; cc < pa.nchC; cc += VF) {
                    double cd[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) cd[k] = nx[k];
                    if (cc + VF < pa.nchC) load_cell(nx, cc + VF);
                    asm volatile("; LF_BEGIN pzcell items=1");
                    const double zc2 = cd[0] * cd[0];
                    const double Lc = quad_nofma(aL, bL, cL, cd[0], zc2);
                    const double Hc = fexp_c(LF_LN10 * (LF_LREF - Lc), &tab);
                    const double b2 = -2.0 * LF_LN10 * aL;                              // 2 b
                    const double a = fma(b2, cd[0], -LF_LN10 * bL);                     // -ln10 L*'(z_c)
                    double cm2 = 1.0, cm1 = a, tail = 0.0;
                    const double head = fma(cm1, cd[2], cd[1]);                         // S_0 + c_1 S_1
#pragma unroll
                    for (int j = 2; j <= ZCELL_M; ++j) {
                        const double cj = fma(a, cm1, b2 * cm2) * (1.0 / j);
                        tail = fma(cj, cd[1 + j], tail);
                        cm2 = cm1;
                        cm1 = cj;
                    }
                    acc = fma(-Hc, head + tail, acc);                                   // everything else of the term is in wbase
                    asm volatile("; LF_END pzcell");
                }
            }
            acc = wave_sum_dpp(acc);              // lane 63: the wave's total
            double* __restrict__ row = pa.partC + (size_t)(w0 + v) * pa.nslot;
            if (lane == 63) pstore(row + vr, acc);
            }
        }
#ifdef LF_STAMPS
        t_cells = __builtin_amdgcn_s_memtime();
#endif
        // ---- the grid integral (piece B): 64 nodes per chunk, lane = node, one exponential per node
        // The chunks are dealt to the virtual ranks in CONTIGUOUS runs (rank vr: chunks [vr cpr, (vr + 1) cpr)): the z-evolving
        // lattice is stored column by column, so a run touches only a few of the S redshift columns, and what the integrand
        // takes from the walker depends on the column only (lf_kernels.h: gridsum_body) - Q_wk = 10^(42 - L*_w(z_k)) and E_wk
        // are made here for the run's columns, one lane each (dealt round-robin every workgroup of a tile needed all S
        // columns of its walkers: 32 times the exponentials, 2.8k cycles of a 36k-cycle launch).
        if (mine_w) {
            const int cpr = (pa.nchB + VF - 1) / VF;
#pragma unroll 1
            for (int vr = frank; vr < VF; vr += fgroup) {
            double bsum = 0.0;
            const int clo = vr * cpr, chi = min(clo + cpr, pa.nchB);
            if (VARIANT == LF_ZEVOL && grid_w && clo < chi) {
                const double aL = uni(sc[Z_AL]), bL = uni(sc[Z_BL]), cL = uni(sc[Z_CL]);
                const double aP = uni(sc[Z_AP]), bP = uni(sc[Z_BP]), cP = uni(sc[Z_CP]), c1z = uni(sc[Z_C1]);
                const int k_lo = (clo * 64) / kc.S, k_hi = min((chi * 64 - 1) / kc.S, kc.S - 1);
                for (int k = k_lo + lane; k <= k_hi; k += 64) {
                    const double2 zz = *reinterpret_cast<const double2*>(zcl + 2 * k);
                    const double Lsz = quad_nofma(aL, bL, cL, zz.x, zz.y);           // lumfuncmcmc_z.py:66
                    const double ph = quad_nofma(aP, bP, cP, zz.x, zz.y);            // :65
                    double2 o;
                    o.x = fexp_c(LF_LN10 * (LF_LREF - Lsz), &tab);
                    o.y = fma(-c1z, Lsz - LF_LREF, fma(LF_LN10, ph, LF_LNLN10));
                    *reinterpret_cast<double2*>(qe + (v * PERS_MAXS + k) * 2) = o;
                }
                __builtin_amdgcn_wave_barrier();      // (this wave's LDS writes are in order before its reads)
            }
            if (grid_w && clo < chi) {
                struct Node {
                    double G, PG, W, col;
                };
                auto load_nodes = [&](int c) -> Node {
                    const double2* __restrict__ src = reinterpret_cast<const double2*>(pa.nodes4 + ((size_t)c * 64 + lane) * 4);
                    const double2 a = src[0], b = src[1];
                    return Node{a.x, a.y, b.x, b.y};
                };
                // (source-sharded ranks split the grid by chunks of 64 nodes: the same granule in every kernel)
                auto mine = [&](int c) { return !(kc.grid_parts > 1 && c % kc.grid_parts != kc.grid_part); };
                const double c1 = uni(sc[VARIANT == LF_ZEVOL ? Z_C1 : R_C1]);
                const double Ls = uni(sc[R_LSTAR]), c0 = uni(sc[R_C0]), Qf = uni(sc[R_Q]);      // (FIXCOMP)
                auto node = [&](const Node& nd, int c) {
                    if (!mine(c)) return;
                    asm volatile("; LF_BEGIN pznode items=1");
                    double e;
                    if (VARIANT == LF_ZEVOL) {
                        const int k = min((int)nd.col, PERS_MAXS - 1);
                        const double2 QE = *reinterpret_cast<const double2*>(qe + (v * PERS_MAXS + k) * 2);
                        e = fma(c1, nd.G - LF_LREF, QE.y) - nd.PG * QE.x;
                    } else {
                        e = fma(c1, nd.G - Ls, c0) - nd.PG * Qf;
                    }
                    // (a chunk whose every node has underflowed - the bright end of the grid, 10^(L - L*) > 750 - adds exact zeros)
                    if (__ballot(e > -750.0) != 0ull) bsum = fma(nd.W, fexp_c(e, &tab), bsum);
                    asm volatile("; LF_END pznode");
                };
                // One chunk ahead, in two sets of registers used in turn (the loop does two chunks per trip).  Written with one
                // set and a copy - nd = nx; nx = the next chunk - the compiler waited for the next chunk's loads at the END of
                // the trip that issued them, to make the copy: nothing was ahead, and a trip took a trip to L2 (720 cycles
                // for 90 of issue, four waves to a SIMD: tools/stamps_fused.py; 14.4k -> 12.4k cycles for the phase).  The loads
                // unconditional - past the run's end the last chunk again: behind a branch the compiler cannot count the loads in
                // flight, and waits for all of them.  (Three sets, two chunks ahead: measured the same.  The cells' loop above
                // has the same shape, but written this way its second set of 16 registers cost the grid's loop more than the
                // cells gained: 17.95 against 17.5 us per evaluation; so did touching the cells ahead, from the prologue or from
                // here with the cells' phase moved behind the grid's.)
                Node na = load_nodes(clo), nb;
#pragma unroll 1
                for (int c = clo; c < chi; c += 2) {
                    nb = load_nodes(min(c + 1, chi - 1));
                    node(na, c);
                    na = load_nodes(min(c + 2, chi - 1));
                    if (c + 1 < chi) node(nb, c + 1);
                }
            }
            bsum = wave_sum_dpp(bsum);
            double* __restrict__ row = pa.partB + (size_t)(w0 + v) * pa.nslot;
            if (lane == 63) pstore(row + vr, bsum);
            }
        }
#ifdef LF_STAMPS
        t_grid = __builtin_amdgcn_s_memtime();
#endif
        // ---- the careful path (rare): walkers whose bounds do not rule out an underflow are summed over the sources with
        // the reference's own -inf convention (per-term checks, device-library math; lf_kernels.h: srcsum_body's careful
        // branch, same arithmetic).  All the tile's workgroups stride over the catalogue together, field by field.
        if (needmask) {
            const double NEG_INF = -__builtin_huge_val();
#pragma unroll 1
            for (int w = 0; w < nw; ++w) {
                if (!((needmask >> w) & 1)) continue;
                const double* __restrict__ r = wsc + w * 8;
#pragma unroll 1
                for (int vr = frank; vr < VF; vr += fgroup) {
                double acc = 0.0;
                long long start = 0;
#pragma unroll 1
                for (int f = 0; f < kc.nf; ++f) {
                    const int n = kc.nsrc[f];
                    const int mode = uni(reinterpret_cast<const int*>(wfc + (w * MAXF + f) * 8 + 4)[M_MODE]);
                    if (VARIANT == LF_FIXCOMP ? mode == MODE_SLOW : mode <= MODE_SLOW) {
#pragma unroll 1
                        for (int i = vr * PB + tid; i < n; i += VF * PB) {
                            const size_t g = (size_t)(start + i);
                            const double clum = pa.lum[g], ca1 = pa.a1[g], cpp = pa.P[g];
                            double term;
                            if (VARIANT == LF_FIXCOMP) {
                                const double vv = cpp * r[R_Q];
                                const double lnT = fma(r[R_C1], clum - r[R_LSTAR], r[R_C0]) - vv;
                                term = lnT + ca1;
                                const bool bad = (vv > LF_UNDERFLOW) | (lnT < -LF_UNDERFLOW) | (term < -LF_UNDERFLOW) | (term != term);
                                term = bad ? NEG_INF : 0.0;             // the value itself is in wbase
                            } else {
                                const WZ wz{r[Z_AL], r[Z_BL], r[Z_CL], r[Z_AP], r[Z_BP], r[Z_CP], r[Z_C1], 0.0};
                                double vv;
                                if (mode == MODE_SLOW) {
                                    const double lnT = lnT_zevol<false>(wz, clum, ca1, pa.U[g], vv, &tab);
                                    term = lnT + cpp;
                                    const bool bad = (vv > LF_UNDERFLOW) | (lnT < -LF_UNDERFLOW) | (term < -LF_UNDERFLOW) | (term != term);
                                    term = bad ? NEG_INF : -vv;         // the rest of the term is in wbase
                                } else {
                                    // (a field that is safe, of a walker another field keeps off the cells: the plain exponential)
                                    const double Lsz = quad_nofma(wz.aL, wz.bL, wz.cL, ca1, pa.U[g]);
                                    term = -fexp_t(LF_LN10 * (clum - Lsz), &tab);
                                }
                            }
                            acc += term;
                        }
                    }
                    start += n;
                }
                __syncthreads();                  // `red` is free (the previous walker's reduction has been read)
                red[tid] = acc;
                __syncthreads();
                if (tid < 64) {
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < PB / 64; ++k) s += red[tid + 64 * k];
                    s = wave_sum_dpp(s);
                    double* __restrict__ row = pa.partA + (size_t)(w0 + w) * pa.nslot;
                    if (tid == 63) pstore(row + vr, s);
                }
                }
            }
        }
        if (FUSED && pa.poll && needmask == 0) {
            // hand-over by polling (lf_free.h: PART_EMPTY): no careful path in this tile, so every slot has its writer; this
            // workgroup's sums are on their way, and only the tile's finisher - the last physical rank - has more to do
            if (frank == fgroup - 1 && v < nw) {
                const int nB = pa.nchB > 0 ? pa.nslot : 0, nC = VARIANT == LF_ZEVOL && pa.nchC > 0 ? pa.nslot : 0;
                double* __restrict__ pb = pa.partB + (size_t)(w0 + v) * pa.nslot;
                double* __restrict__ pc = pa.partC + (size_t)(w0 + v) * pa.nslot;
                double pre[2] = {0.0, 0.0};
                int tries = 0;
                bool have;
                do {
                    if (lane < nC) pre[0] = __hip_atomic_load(pc + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (lane < nB) pre[1] = __hip_atomic_load(pb + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    have = !((lane < nC && (unsigned long long)__double_as_longlong(pre[0]) == PART_EMPTY) ||
                             (lane < nB && (unsigned long long)__double_as_longlong(pre[1]) == PART_EMPTY));
                } while (!__all(have) && ++tries < PART_POLLS);
                if (tries >= PART_POLLS) {
                    pre[0] = pre[1] = __builtin_nan("");
                    if (lane == 0) atomicExch(pa.err, 1);
                }
                if (lane < nC) __hip_atomic_store(pc + lane, __longlong_as_double((long long)PART_EMPTY), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane < nB) __hip_atomic_store(pb + lane, __longlong_as_double((long long)PART_EMPTY), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                finalize_wave<true>(pa.partA, 0, pa.nslot, pa.partB, nB, pa.nslot, nC > 0 ? pa.partC : nullptr, nC, (int)STAT_CELLS,
                                    sstat - w0, sbase - w0, w0 + v, lane, ap, pa.out, nullptr, nullptr,
                                    VARIANT == LF_FIXCOMP ? (int)STAT_SLOW : 0,
                                    STEP ? sprop + v * 16 : nullptr, STEP ? szz + v : nullptr, STEP ? spre + 2 * v : nullptr, pre);
            }
        } else if (FUSED) {
            // as in lf_free: this workgroup's partial sums have been acknowledged, then it counts itself; the last of the
            // tile's workgroups to count adds the partials up (finalize_wave: lf_finalize's body, the same slots and order)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __threadfence_block();
            __syncthreads();
            if (tid == 0) sdone = atomicAdd(q, 1);
            __syncthreads();
            if (sdone == fgroup - 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const int nC = VARIANT == LF_ZEVOL && pa.nchC > 0 ? pa.nslot : 0;
                if (v < nw) {
                    finalize_wave<true>(pa.partA, pa.nslot, pa.nslot, pa.partB, pa.nslot, pa.nslot, nC > 0 ? pa.partC : nullptr, nC, (int)STAT_CELLS,
                                        sstat - w0, sbase - w0, w0 + v, lane, ap, pa.out, nullptr, nullptr,
                                        VARIANT == LF_FIXCOMP ? (int)STAT_SLOW : 0,       // (FIXCOMP: per-source partials exist for SLOW walkers only)
                                        STEP ? sprop + v * 16 : nullptr, STEP ? szz + v : nullptr, STEP ? spre + 2 * v : nullptr);
                    if (pa.poll) {                // (the slots empty again: the next launch's tiles may poll)
                        if (lane < nC) __hip_atomic_store(pa.partC + (size_t)(w0 + v) * pa.nslot + lane, __longlong_as_double((long long)PART_EMPTY), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (lane < pa.nslot) __hip_atomic_store(pa.partB + (size_t)(w0 + v) * pa.nslot + lane, __longlong_as_double((long long)PART_EMPTY), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (tid < QSTRIDE) q[tid] = 0;    // the tile's counters, for the next launch
            }
        }
    }
#ifdef LF_STAMPS
    if (stamp && tid == 0) {                      // (tools/stamps_fused.py: the slots of lf_free's time line; [2] = the columns' end)
        stamp[1] = __builtin_amdgcn_s_memtime();
        stamp[2] = 0;
        stamp[3] = t_prep - stamp[0];
        stamp[4] = t_cells - stamp[0];
        stamp[7] = t_grid - stamp[0];
        stamp[6] = __builtin_amdgcn_s_memrealtime();
        kc.stamps[(size_t)gridDim.x * 8 + blockIdx.x] = ((t_cols - stamp[0]) << 32) | ((t_prep - stamp[0]) & 0xffffffffull);
    }
#endif
}

template <int VARIANT, bool FUSED>
__global__ __launch_bounds__(PB, 4) void lf_pers(KConst kc, PersArgs pa) {
    warm_kernarg<sizeof(KConst) + sizeof(PersArgs)>();      // (lf_math.h: the arguments in one round trip)
    lf_pers_body<VARIANT, FUSED, false>(kc, pa, StepArgs{}, AcceptArgs{});
}

template <int VARIANT>
__global__ __launch_bounds__(PB, 4) void lf_pers_step(KConst kc, PersArgs pa, StepArgs sp, AcceptArgs ap) {
    warm_kernarg<sizeof(KConst) + sizeof(PersArgs) + sizeof(StepArgs) + sizeof(AcceptArgs)>();
    lf_pers_body<VARIANT, true, true>(kc, pa, sp, ap);
}

}  // namespace lf
