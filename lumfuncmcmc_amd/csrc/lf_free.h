// lf_free.h - the FREE variant's lnprob work (pieces A and B) in PERSISTENT 512-thread workgroups (gfx950).
//
// Why this shape (all measured on MI355X):
//  * One wave issues one fp64 VALU instruction per ~8.5 cycles however many independent chains it has; a SIMD needs
//    4 resident waves to reach its 3.4 cycles per instruction (profiles/r01_ubench.txt).  So: <= 128 VGPRs, 16 waves
//    per CU.
//  * The table-driven form of the term (lf_kernels.h: table_lookup / table_terms) needs 33 KB of tables in LDS.  Eight
//    waves share one copy: 512-thread workgroups, two per CU (70 KB of LDS each).
//  * A workgroup that lives for one item pays its prologue (tables, items, walker constants: 5k cycles) and its
//    reduction every ~35k cycles (tools/stamps.py).  Persistent workgroups load the tables ONCE, serve ONE tile of PTW
//    walkers for their whole life (its constants for all fields: 4.5 KB of LDS, loaded once), and pull items from the
//    tile's queues; the next item's sources are already on their way into registers while the current one is summed.
//
// Items of a tile: node chunks of the integration grid (piece B, 512 nodes each; longer, so served first) and catalogue
// chunks (piece A, 512 ST flux-neighbouring sources of one field).  Catalogue chunks are dealt to eight queues per
// tile, one per XCD (workgroups ask for their XCC id), so that the sixteen tiles' reads of a chunk meet in one L2; an
// empty queue steals from the next.  lf_prepare zeroes the counters of the launch that follows it.
//
// Results: every (walker, chunk) partial sum goes to its own slot of partA / partB and lf_finalize adds them in a fixed
// order, as in lf_main - the bits depend on the launch geometry (ST), not on which workgroup served which item.
#pragma once
#include "lf_kernels.h"

namespace lf {

constexpr int PB = 512;      // threads per persistent workgroup: 8 waves
constexpr int PTW = 8;       // walkers per tile
constexpr int QSTRIDE = 9;   // counters per tile: [0] grid queue, [1..8] catalogue queues of XCD 0..7
// The cells' and the grid's chunks of a tile are dealt to VF VIRTUAL workgroups, and partB / partC hold one partial sum per
// (walker, virtual workgroup): the workgroups that actually serve the tile (at most VF: 32 at 128 rows, 16 at 256, 8 when a
// group serves several tiles in turn) take the virtual ranks r, r + fgroup, ... and keep their sums apart.  So a walker's
// partial sums - and with them the bits of its lnprob - do not depend on how many rows share its call, on its place in
// the batch, or on how a batch is sharded over GPUs.
constexpr int VF = 32;
// The deal table: [0 .. VF] where rank vr's cell chunks start in the list, [VF + 1 .. 2 VF + 1] the same for its bins, then the
// list (cell chunks rank by rank, then bins rank by rank).  Who gets what is decided by COST: a flux bin costs a wave about
// 2.7 cell chunks, and the workgroups of ranks >= VF / 2 are the younger ones of their CUs, behind their elders when the sums
// begin (tools/stamps_fused.py) - the host deals bins, then cells, each to the rank that would be done first (lfmcmc.hip:
// ensure_deal has the costs and the sweep they come from).  With the arithmetic deal (bin c to rank c mod VF, cell chunk cc to rank (cc + VF / 2) mod VF) the busiest rank of the
// benchmark's context had a bin and two cell chunks (10.4k cycles), the average being 6.6k, and the 17th bin sat on a younger
// rank with two cell chunks of its own.  A context's table depends on its numbers of bins and cell chunks only: a row's
// partial sums (one per virtual rank) are the same whatever the batch.
constexpr int DEAL_BINS = VF + 1, DEAL_LIST = 2 * (VF + 1), DEAL_MAX = 512;
// The one-launch form's hand-over by POLLING (tiles whose walkers are all on the cells: the normal case).  The slots of partB /
// partC hold PART_EMPTY between launches (the host fills them, every finisher leaves them so); a workgroup writes its partial
// sums through and is done; the tile's FINISHER - the workgroup of the last physical rank, the lightest of the deal - reads the
// slots past its caches until none is empty, adds them up and empties them again.  Against the counter (every workgroup:
// wait for the stores' acknowledgements, count, wait for the count; the last one: load, add) the launch's critical path
// loses two of its three trips to memory.  A partial sum is never PART_EMPTY (a NaN is made canonical before it is stored);
// a finisher that has polled PART_POLLS times without success writes NaN (emcee raises on NaN) and sets the error word.
constexpr unsigned long long PART_EMPTY = 0x7ff8dead7ff8deadull;
constexpr int PART_POLLS = 1 << 19;

// 512-thread block reduction: red[nw][512] -> out[(w0 + w) * stride + chunk], nw <= 8: wave w adds walker w's row
// (eight columns per lane, stride 64) and runs one 64-lane sum on the DPP network.  Fixed order.
// THROUGH: the partial sum is written through to memory (see pstore in lf_free).
template <bool THROUGH = false>
__device__ __forceinline__ void reduce_store512(const double* __restrict__ red, int wlo, int whi, double* __restrict__ out,
                                                size_t stride, int w0, int chunk, int tid) {
    const int w = tid >> 6, lane = tid & 63;
    if (w >= wlo && w < whi) {             // (wave w sums walker w's 512 lane sums)
        const double* row = red + w * PB + lane;
        double s0 = (row[0] + row[64]) + (row[128] + row[192]);
        double s1 = (row[256] + row[320]) + (row[384] + row[448]);
        const double s = wave_sum_dpp(s0 + s1);
        if (lane == 63) {
            if (THROUGH) __hip_atomic_store(out + (size_t)(w0 + w) * stride + chunk, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else out[(size_t)(w0 + w) * stride + chunk] = s;
        }
    }
}

struct FreeArgs {
    int B, ntiles;            // theta rows, tiles of PTW walkers
    int nchA, nchB;           // catalogue chunks (512 ST sources), node chunks (64 nodes)
    int nslot;                // slots per walker in partB / partC: one per workgroup serving the walker's tile (the largest
                              // such number over the tiles; a tile with fewer workgroups zeroes the rest)
    int tile_stride;          // workgroup g serves tiles (g / 8) % ntiles, + tile_stride, ... (normally just one)
    int skip_grid;
    int* queues;              // [ntiles][QSTRIDE]
    double* partA;            // [B][nchA]
    double* partB;            // [B][nslot]: the grid integral, one partial per (walker, workgroup of its tile)
    // cells (lf_kernels.h: CELL_M): walkers flagged STAT_CELLS by lf_prepare are summed over them instead of the sources
    const double* cells;      // [nchC * 64][CELL_REC] {x_c, S_0 .. S_8}: every field's cells padded to whole chunks of 64 (pads: all sums 0)
    const double* nodes8;     // [nchB * 64][8] {G, PG, W, a3, a4, the chunk's smallest a4, -, -}: the grid's nodes, one 64-byte
                              // record per node, padded to whole chunks (pads: W = 0)
    const int* deal;          // the static deal of cell chunks and flux bins to the VF virtual workgroups (DEAL_* below; made by the
                              // host from the chunks' costs: lfmcmc.hip, ensure_deal) or NULL = the arithmetic deal
    const int* cc_len;        // [nchC] its cells (<= 64: one per lane of a wave; the waves take one walker each)
    const int* cc_field;      // [nchC]
    int nchC;                 // cell chunks (64 cells; 0: no cells)
    double* partC;            // [B][nslot]: the cells' sums, likewise
    const int* wstat;         // [B]
    // FUSED instantiation: lf_prepare's and lf_finalize's work is done here, one launch per evaluation (plain lnprob
    // calls; the sampler's propose / accept steps and the diagnostics keep the three launches)
    const double* theta;      // [B][ndim]
    double* out;              // [B] lnprob
    double* wrec_w;           // the same buffers as the kernel's wrec / wmode arguments, wstat above, wbase: written by the
    int* wmode_w;             // prologue, read back behind a barrier - through THESE pointers only (the arguments are
    int* wstat_w;             // const __restrict__: the FUSED instantiation never dereferences them)
    double* wbase_w;
    // piece B over flux bins instead of lattice points (lf_gridbound.h; separable grids): nbq > 0 replaces the lattice loop
    const double* gq_rec;     // [nbq * 64][4] {x_n, 10^(x_n + 17), L of row row0 + lane, 10^(that - 42)}: one bin per wave, lane = node
    const double* gq_omega;   // per bin [row][64]: the rows' quadrature weights, trapezoid weights folded in
    const int* gq_rows;       // [nbq][4] {row0, nrows (<= 64), offset into gq_omega, -}
    int nbq;
    int poll;                 // 1: the slots of partB / partC are PART_EMPTY (the host's word): tiles without source work hand over by polling
    int* err;                 // error word (a finisher gave up polling)
};

// CENSUS: the instantiation that counts which form of the term ran (lf_form_counts); the product one has no trace of it
// FUSED: one launch per evaluation.  Every workgroup of a tile first does lf_prepare's work for the tile's 8 walkers
// (one wave: 8 lanes per walker, as there; all of the tile's workgroups write the same values to the same records, which
// spares them a hand-over), and the LAST workgroup to finish a tile (a counter per tile, q[0]) does lf_finalize's for
// them and leaves the tile's counters at zero for the next launch.  Two kernel boundaries fewer per evaluation:
// measured 7 + 4.5 us of the 33 an evaluation of 128 rows took.
// STEP (with FUSED): the launch is a half-step of the device-resident sampler - the tile's walkers are the stretch-move
// PROPOSALS of the active half (made in the prologue by prepare_lane, as lf_prepare makes them: Philox keyed by (step, half,
// walker), every workgroup of the tile the same values), and the tile's finishing workgroup accepts or rejects them and
// writes the chain's row (accept_walker, as lf_finalize does).  An instantiation of its own behind a kernel of its own
// (lf_free_step): the sampler's arguments are not in everybody's argument block.
template <int ST, bool CENSUS, bool FUSED, bool STEP>
__device__ __forceinline__ void lf_free_body(const KConst& kc, const SrcArrays& sa, const NodeArrays& na, const double* __restrict__ wrec_arg,
                                             const int* __restrict__ wmode_arg, const FreeArgs& fa, const StepArgs& sp, const AcceptArgs& ap,
                                             unsigned long long t_pre = 0) {
    const double* wrec = FUSED ? fa.wrec_w : wrec_arg;
    const int* wmode = FUSED ? fa.wmode_w : wmode_arg;
    // A partial sum: in the fused form it is read by a workgroup on another XCD while the launch is still running, so it
    // is written THROUGH this XCD's L2 (a relaxed store of agent scope: scope bits on the one store - no cache-wide
    // write-back or invalidate, which is what a fence of that scope costs: measured 154 us per evaluation instead of 30)
    auto pstore = [](double* p, double v) {
        if (FUSED) __hip_atomic_store(p, v == v ? v : __builtin_nan(""), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (never PART_EMPTY)
        else *p = v;
    };
    __shared__ MathTables tab;
    __shared__ TermTables tt;
    __shared__ __attribute__((aligned(16))) double red[PTW * PB];         // per-walker lane sums of the item in hand
    __shared__ __attribute__((aligned(16))) double wfc[PTW * MAXF * 8];   // per (walker, field): aC, V, cA, cYs, {mode, klo, khi, kne, kaC}, cYH
    __shared__ __attribute__((aligned(16))) double wsc[PTW * 8];          // per walker: L*, c0, c1, Q, alpha_C
    __shared__ int sitem[2];
    __shared__ int sdeal[DEAL_MAX];        // the deal table (fa.deal), with the tables
    __shared__ int scell;                  // bit w: walker w of the tile is summed over the cells
    __shared__ int snosrc;                 // the tile has no source work at all (every live walker on the cells)
    const int tid = threadIdx.x;
    __shared__ int sstat[PTW];             // FUSED: the tile's status words, straight from the preparation
    __shared__ double sbase[PTW];          //        ... the closed-form part of piece A
    __shared__ double wlf[PTW * MAXF];     //        ... lF per (walker, field) (the careful path's)
    __shared__ double sprop[PTW * 16];     // STEP:  ... the tile's proposals and stretch factors (the accept step's)
    __shared__ double szz[PTW];
    __shared__ double spre[PTW * 2];       //        ... and the accept step's two logarithms per walker, made ahead
    // ---- once per workgroup: the tables, and which XCD we are on.  (FUSED: waves 1..7 load them while wave 0 prepares the
    // first tile's walkers - see the top of the tile loop.)
    if (!FUSED) {
        for (int i = tid; i < 256; i += PB) {
            tab.logt[i] = *reinterpret_cast<const double2*>(LOG_TABLE + 2 * i);
            tab.expt[i] = EXP_TABLE[i];
        }
        load_term_tables<PB>(&tt);
        if (fa.deal && tid < DEAL_LIST + fa.nchC + fa.nbq) sdeal[tid] = fa.deal[tid];      // (<= DEAL_MAX = PB entries)
    }
    bool tables_loaded = !FUSED;
    // The thread number, made anew wherever it is needed (wave index from a scalar register, lane from mbcnt on an opaque
    // mask): what is derived from it - indices, addresses - is then computed where it is used.  Carried across the item
    // loop such values were spilled - 30 MB of scratch stores per launch from the prologue alone - and a reload from
    // scratch between two loads waits for every load issued before it: the eight loads of a switch-in took eight round
    // trips (tools/stamps.py: 10k of an item's 31k cycles).
    const int wave_base = __builtin_amdgcn_readfirstlane(tid & ~63);
    auto fresh_tid = [&]() -> int {
        int m = -1;
        asm volatile("" : "+s"(m));
        return wave_base + __builtin_amdgcn_mbcnt_hi(m, __builtin_amdgcn_mbcnt_lo(m, 0));
    };
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int myq = (int)(xcc & 7u);
#ifdef LF_STAMPS
    __shared__ unsigned long long s_targs;                       // arguments in (wave 0)
    unsigned long long* stamp = kc.stamps ? kc.stamps + (size_t)blockIdx.x * 8 : nullptr;
    if (stamp && tid == 0) {
        stamp[0] = t_pre ? t_pre : __builtin_amdgcn_s_memtime();       // (the wave's first instruction, before its arguments)
        stamp[5] = __builtin_amdgcn_s_memrealtime();
        s_targs = __builtin_amdgcn_s_memtime();
    }
    int nitems_done = 0;
    unsigned long long t_first = 0, t_sw = 0, t_loop = 0, t_red = 0;
    unsigned long long t_prep = 0, t_cells = 0, t_grid = 0;      // (the last tile's: after the records are in LDS, the cells, the grid)
    __shared__ unsigned long long s_ttab;                        // wave 1 done with its share of the tables
    __shared__ unsigned long long s_tp0;                         // the preparing wave back from prepare_lane
    __shared__ unsigned long long s_tprep[8];                    // inside the preparation: entry, theta in LDS, Q made, keys made, before the combine
#endif

#pragma unroll 1
    for (int tile = ((int)blockIdx.x >> 3) % fa.ntiles; tile < fa.ntiles; tile += fa.tile_stride) {
        const int w0 = tile * PTW;
        const int nw = min(PTW, fa.B - w0);
        __builtin_assume(nw >= 1 && nw <= PTW);
        int* __restrict__ q = fa.queues + tile * QSTRIDE;
        // next item of this tile: a node chunk while there are any (item = chunk), then a catalogue chunk of our XCD's
        // queue, or of the next queue that still has some; -1 = none left
        // Cell chunks and grid chunks all cost about the same (and with every walker on cells they are all there is): they
        // are dealt STATICALLY, chunk i to the workgroup of rank i mod fgroup among those serving the tile - no claims.
        // Only the source chunks - unequal, and only needed by walkers that cannot use the cells - are claimed from queues.
        int fgroup = 8, frank = (int)blockIdx.x & 7;
        if (fa.ntiles <= fa.tile_stride) {
            const int k = (int)blockIdx.x >> 3;   // (here tile = k mod ntiles, and the groups k, k + ntiles, ... share it)
            fgroup = 8 * ((fa.tile_stride - tile + fa.ntiles - 1) / fa.ntiles);
            frank += 8 * ((k - tile) / fa.ntiles);
        }
        bool no_src = false;                       // (thread 0's view) every walker of the tile is summed over the cells
        auto grab = [&]() -> int {
            if (no_src) return -1;
            for (int d = 0; d < 8; ++d) {
                const int qq = (myq + d) & 7;
                const int lo = (int)(((long long)qq * fa.nchA) >> 3), hi = (int)(((long long)(qq + 1) * fa.nchA) >> 3);
                if (hi <= lo) continue;
                const int i = atomicAdd(q + 1 + qq, 1);
                if (i < hi - lo) return lo + i;
            }
            return -1;
        };
        // The claim of the item after the current one rides on the current item's loads: thread 0 takes a TICKET of the
        // queue it expects to serve it - the grid queue until that has run dry once, then its XCD's catalogue queue -
        // just before the loads are issued, and turns it into an item (or falls back to the full search) when they have
        // arrived: one round trip for both.  The result goes to LDS at once: kept in a register across the walker loop
        // it is spilled, and the spill waits for the atomic on the spot.  (build.py keeps the compiler's atomic
        // optimizer from reading the one-lane atomic's result where it is issued.)
        int ticket = 0;
        auto take_ticket = [&]() {
            if (!no_src) ticket = atomicAdd(q + 1 + myq, 1);
        };
        auto redeem = [&]() -> int {
            if (no_src) return -1;
            const int lo = (int)(((long long)myq * fa.nchA) >> 3), hi = (int)(((long long)(myq + 1) * fa.nchA) >> 3);
            if (ticket < hi - lo) return lo + ticket;
            return grab();                        // our queue is empty: steal (the grid queue and ours just hand out misses)
        };
        if (tile != ((int)blockIdx.x >> 3) % fa.ntiles) __syncthreads();      // the previous tile's last reads of wfc / wsc / sitem are done
        const int u = fresh_tid();
        if (FUSED) {
            // Wave 0 prepares the tile's 8 walkers (lf_prepare's body) and puts their records straight into LDS (wfc, wsc,
            // wlf, sstat, sbase: the one-launch form writes no records to memory - the launch that makes them is the only
            // one that reads them); the other seven waves meanwhile bring the tables in (first tile only).  Nobody waits for a trip through memory:
            // the preparation's dependent chain (theta from memory, device-library exp10 / log10 / sqrt / log) and the
            // tables' one round trip to L2 overlap, one barrier ends both.  (Records written to memory, acknowledged, read
            // back by every wave, behind the tables' load: 8.0 of the launch's 19 us, tools/stamps_fused.py.)
            // Which wave prepares: wave 0 in the workgroups of the launch's first half, wave 1 in those of the second.  A CU
            // holds workgroups i and i + 256 (the launch fills one slot of every CU before the second) and waves of equal index
            // share a SIMD: two lone dependent chains of quarter-rate fp64 (divisions, sqrt, the library's exp10 / log10) on one
            // SIMD took the younger workgroup 5k cycles longer than its elder (tools/stamps_fused.py) - and the launch ends
            // with its slowest workgroup.
            const int pw = (int)(blockIdx.x >> 8) & 1;
            const int up = u - 64 * pw;           // the preparing wave's threads: 0 .. 63
            if (up >= 0 && up < 64) {
                prepare_lane<false, true, STEP, LF_FREE>(kc, sp, fa.theta, fa.B, nullptr, nullptr, nullptr, nullptr, nullptr, 1,
                                          w0 + (up >> 3), up & 7, up >> 3, reinterpret_cast<double(*)[16]>(red), wfc, wsc, sstat, sbase, wlf,
#ifdef LF_STAMPS
                                          s_tprep,
#else
                                          nullptr,
#endif
                                          sprop, szz);
#ifdef LF_STAMPS
                if (up == 0) s_tp0 = __builtin_amdgcn_s_memtime();
#endif
            } else if (!tables_loaded) {
                const int t7 = u < 64 * pw ? u : u - 64;      // the other seven waves' threads: 0 .. 447
                // (all of a thread's loads in flight together: one round trip to the cold L2)
                double2 lt = *reinterpret_cast<const double2*>(LOG_TABLE + 2 * min(t7, 255));
                double et = EXP_TABLE[min(t7, 255)];
                const int ndeal = fa.deal ? DEAL_LIST + fa.nchC + fa.nbq : 0;     // (<= DEAL_MAX: two entries per thread of the 448)
                int d0 = ndeal ? fa.deal[min(t7, ndeal - 1)] : 0, d1 = ndeal ? fa.deal[min(t7 + 256, ndeal - 1)] : 0;
                asm volatile("" : "+v"(lt.x), "+v"(lt.y), "+v"(et), "+v"(d0), "+v"(d1));
                load_term_tables<PB - 64>(&tt, t7);
                if (t7 < 256) {
                    tab.logt[t7] = lt;
                    tab.expt[t7] = et;
                    if (t7 < ndeal) sdeal[t7] = d0;
                    if (t7 + 256 < ndeal) sdeal[t7 + 256] = d1;
                }
#ifdef LF_STAMPS
                if (t7 == 0) s_ttab = __builtin_amdgcn_s_memtime();
#endif
            }
            if (STEP && u >= PB - 64 && u < PB - 64 + nw) {
                // (the last wave, idle once its share of the tables is on its way: the accept step's logarithms, from the same
                // Philox draws as the proposal's stretch factor and the accept step's uniform - bit for bit what
                // accept_walker would make in the epilogue)
                const int wl = u - (PB - 64);
                unsigned int rr[4];
                sampler_draw(sp.step, sp.half, w0 + wl, 0, sp.seed, rr);
                accept_terms(ap, w0 + wl, stretch_z(sp.a, u53(rr[0], rr[1])), spre[2 * wl], spre[2 * wl + 1]);
            }
            tables_loaded = true;
            __syncthreads();
        }
        if (u < 64) {
            // which walkers of the tile lf_prepare put on the cells (one load per lane); when all of them are, the sources
            // are not touched
            const int st = FUSED ? sstat[min(u, nw - 1)] : fa.wstat[w0 + min(u, nw - 1)];
            const bool on = fa.nchC > 0 && u < nw && (st & STAT_CELLS);
            // ... and which need the sources at all: not the ones outside the prior or already known to be -inf (their
            // lnprob is -inf whatever the sums are: lf_finalize).  A stretch-move ensemble puts such a proposal into nearly
            // every tile of 8, and each used to cost its tile a pass over the whole catalogue with every term skipped.
            const bool need = u < nw && !on && (st & STAT_PRIOR_OK) && !(st & STAT_NEGINF);
            const int m = (int)__ballot(on);
            const unsigned long long mneed = __ballot(need);
            if (u == 0) {
                scell = m;
                no_src = mneed == 0ull;
                snosrc = no_src ? 1 : 0;
                sitem[0] = grab();
            }
        }
        // the tile's walker constants, all fields (64 B per (walker, field)), once  (FUSED: they are in LDS already)
        if (!FUSED && u < nw * MAXF) {
            const int w = u / MAXF, f = u - w * MAXF;
            double* d = wfc + u * 8;
            int* di = reinterpret_cast<int*>(d + 4);
            if (f < kc.nf) {
                // slot order of the record's own field block (F_V = 1, F_CA = 2, F_CY = 3), alpha_C in the slot of lF
                const double* __restrict__ r = wrec + (size_t)(w0 + w) * REC;
                const int* __restrict__ km = wmode + ((size_t)(w0 + w) * MAXF + f) * WM;
                d[0] = r[R_ALPHAC];
                d[F_V] = r[RF(f, F_V)];
                d[F_CA] = r[RF(f, F_CA)];
                // (the h table's index arithmetic wants y - H_LO, and that minus half a piece: table_lookup)
                d[F_CY] = r[RF(f, F_CY)] - H_LO - 0.5 / H_INV;
                d[7] = r[RF(f, F_CY)] - H_LO;
#pragma unroll
                for (int i = 0; i < 5; ++i) di[i] = km[i];
            }
        }
        if (!FUSED && u >= PB - PTW * 8) {      // (the last 64 threads: walker w, scalar slot j)
            const int t = u - (PB - PTW * 8), w = t >> 3, j = t & 7;
            if (w < nw) wsc[t] = wrec[(size_t)(w0 + w) * REC + j];
        }
        __syncthreads();
#ifdef LF_STAMPS
        t_prep = __builtin_amdgcn_s_memtime();
#endif
        const int cellmask = __builtin_amdgcn_readfirstlane(scell);
        // (No register prefetch of the next item: it would cost 16 VGPRs across the whole walker loop, and with 128 per
        // wave that means scratch traffic inside the loop - measured 4x slower.  The other workgroup of the CU computes
        // while this one waits for its item at switch-in.)
        auto fetch = [&](int w, int fld) {        // uniform LDS address: broadcast reads
            const double2* __restrict__ p = reinterpret_cast<const double2*>(wfc + (w * MAXF + fld) * 8);
            const double2 a = p[0], b = p[1], c = p[3];
            // mode and keys: scalar loads straight from lf_prepare's table (wave-uniform address; five readfirstlanes
            // per walker fewer than through the LDS copy)
            if (FUSED) {                            // (the keys are in the LDS copy; nothing of the records is read from memory)
                const int* __restrict__ ki = reinterpret_cast<const int*>(wfc + (w * MAXF + fld) * 8 + 4);
                return WalkerK{a.x, b.x, b.y, c.y, uni(ki[M_MODE]), uni(ki[M_KLO]), uni(ki[M_KHI]), uni(ki[M_KNE]), uni(ki[M_KAC])};
            }
            const int* __restrict__ km = wmode + ((size_t)(w0 + w) * MAXF + fld) * WM;
            return WalkerK{a.x, b.x, b.y, c.y, km[M_MODE], km[M_KLO], km[M_KHI], km[M_KNE], km[M_KAC]};
        };

        // ---- cells: piece A of the walkers on cells.  64 cells per chunk, one WALKER PER WAVE (lane = cell): a wave sums its
        // walker over this workgroup's share of the tile's cell chunks on its own - no LDS, no barrier, the next chunk's
        // cells in flight while the current one is summed - and writes one partial per chunk from the DPP network.
        {
            const int v = wave_base >> 6;
            // A wave adds up its lanes' values over ALL the chunks of a virtual rank and reduces them once: one partial per
            // (walker, virtual workgroup) (the chunks are dealt statically: the order of the sums is fixed by the context).
#pragma unroll 1
            for (int vr = frank; vr < VF; vr += fgroup) {
            double acc = 0.0;
            // Who gets which chunk.  The workgroups of ranks >= fgroup / 2 are the YOUNGER ones of their CUs (the launch fills one
            // slot of every CU before the second) and run ~2 us behind their elders in every phase (tools/stamps_fused.py), and
            // a bin of the grid costs three cell chunks: so the bins go to the elders (rank c mod VF, from 0 up) and the
            // cell chunks are dealt from the middle (chunk cc to rank (cc + VF / 2) mod VF): the younger half gets cells
            // first and no bins.
            // (the host's deal - DEAL_* above - where there is one: this rank's chunks are entries cb .. cb + cn - 1 of the list)
            const bool dealt = fa.deal != nullptr;
            const int cfirst = (vr - VF / 2 + VF) % VF;
            const int cb = dealt ? DEAL_LIST + uni(sdeal[vr]) : cfirst;
            const int cn = dealt ? uni(sdeal[vr + 1]) - uni(sdeal[vr]) : (cfirst < fa.nchC ? (fa.nchC - cfirst + VF - 1) / VF : 0);
            auto chunk_at = [&](int i) { return dealt ? uni(sdeal[cb + i]) : cb + i * VF; };
            if (fa.nchC > 0 && v < nw && ((cellmask >> v) & 1) && cn > 0) {       // (wave-uniform)
                // (Nothing but these loads goes through the vector memory counter inside the loop - chunk cc starts at cell
                // 64 cc, its field comes from KConst by scalar compares, pads need no masking - so the next chunk's cells
                // really are in flight while the current chunk is summed.  With the chunk table read from memory and the
                // lanes past the end masked, every chunk waited for its own loads twice over.)
                auto load_cells = [&](double (&d)[CELL_REC], int cc) {
                    const int lane = fresh_tid() & 63;      // (made here: carried through the loop it is spilled in one instantiation)
                    const double2* __restrict__ src = reinterpret_cast<const double2*>(fa.cells + ((size_t)cc * 64 + lane) * CELL_REC);
                    static_assert(CELL_REC % 2 == 0, "16-byte loads");
#pragma unroll
                    for (int k = 0; k < CELL_REC / 2; ++k) {
                        const double2 a = src[k];
                        d[2 * k] = a.x;
                        d[2 * k + 1] = a.y;
                    }
                };
                auto field_of = [&](int cc) {               // (scalar: cc and the table are wave-uniform)
                    int f = 0;
#pragma unroll
                    for (int k = 1; k < MAXF; ++k) f += cc >= kc.cc_fstart[k] ? 1 : 0;
                    return f;
                };
                double nx[CELL_REC];
                load_cells(nx, chunk_at(0));
#pragma unroll 1
                for (int ci = 0; ci < cn; ++ci) {
                    const int cc = chunk_at(ci);
                    double cd[CELL_REC];
#pragma unroll
                    for (int k = 0; k < CELL_REC; ++k) cd[k] = nx[k];
                    if (ci + 1 < cn) load_cells(nx, chunk_at(ci + 1));
                    const WalkerK p = fetch(v, field_of(cc));
                    asm volatile("; LF_BEGIN cell items=1");
                    acc += cell_sum(cd, p, &tt);
                    asm volatile("; LF_END cell");
                    if (CENSUS && kc.forms && (fresh_tid() & 63) == 0) atomicAdd(kc.forms + FORM_CELL, (unsigned long long)uni(fa.cc_len[cc]));
                }
            }
            if (fa.nchC > 0 && v < nw) {
                acc = wave_sum_dpp(acc);          // lane 63: the wave's total
                const int ln = fresh_tid() & 63;
                double* __restrict__ row = fa.partC + (size_t)(w0 + v) * fa.nslot;
                if (ln == 63) pstore(row + vr, acc);
            }
            }
        }
#ifdef LF_STAMPS
        t_cells = __builtin_amdgcn_s_memtime();
#endif
        // ---- the grid integral (piece B).  Separable grid (the default): over FLUX BINS (lf_gridbound.h) - one bin of 64
        // Chebyshev nodes per wave-chunk, lane = node: the completeness sum over the fields at the node, times the dot product
        // of the Schechter function at the rows that cross the bin (one row per lane, handed round through LDS) with the
        // rows' quadrature weights.  16-20 chunks per walker instead of 160, error proven <= 1e-15 of piece B.  The bins
        // continue the static deal of the cell chunks (bin c is item nchC + c of the tile).
        if (fa.nbq > 0) {
            const int v = wave_base >> 6;
            const int nq = fa.nbq;
#pragma unroll 1
            for (int vr = frank; vr < VF; vr += fgroup) {
            double bsum = 0.0;
            // (bin c goes to the virtual workgroup of rank c mod VF: see the cells' deal above)
            const bool dealt = fa.deal != nullptr;
            const int qb = dealt ? DEAL_LIST + fa.nchC + uni(sdeal[DEAL_BINS + vr]) : vr;
            const int qn = dealt ? uni(sdeal[DEAL_BINS + vr + 1]) - uni(sdeal[DEAL_BINS + vr]) : (vr < nq ? (nq - vr + VF - 1) / VF : 0);
            auto bin_at = [&](int i) { return dealt ? uni(sdeal[qb + i]) : qb + i * VF; };
            if (v < nw && qn > 0) {
                const double* __restrict__ sc = wsc + v * 8;
                const int mode = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(wfc + (v * MAXF) * 8 + 4));
                const double* __restrict__ r = wfc + v * MAXF * 8 - 8;   // RF(f, slot) = 8 + 8 f + slot
                double vmin = r[RF(0, F_V)];
                for (int f = 1; f < kc.nf; ++f) vmin = fmin(vmin, r[RF(f, F_V)]);
                vmin = uni(vmin);
                const double alphaC = uni(sc[R_ALPHAC]);
                auto mine = [&](int c) { return mode != MODE_SKIP && !(kc.grid_parts > 1 && c % kc.grid_parts != kc.grid_part); };
                struct QRec {
                    double x, a4, L, PGL;
                };
                auto load_rec = [&](int c) -> QRec {        // (one 32-byte record per lane)
                    const int lane = fresh_tid() & 63;
                    const double2* __restrict__ src = reinterpret_cast<const double2*>(fa.gq_rec + ((size_t)c * 64 + lane) * 4);
                    const double2 a = src[0], b = src[1];
                    return QRec{a.x, a.y, b.x, b.y};
                };
                double* __restrict__ Tl = red + v * 64;      // this wave's Schechter values, one row per lane
                QRec nx = load_rec(bin_at(0));
#pragma unroll 1
                for (int qi = 0; qi < qn; ++qi) {
                    const int c = bin_at(qi);
                    const QRec nd = nx;
                    const int nr = uni(fa.gq_rows[4 * c + 1]), off = uni(fa.gq_rows[4 * c + 2]);
                    if (qi + 1 < qn) nx = load_rec(bin_at(qi + 1));
                    if (mine(c)) {
                        const int lane = fresh_tid() & 63;
                        const double* __restrict__ om = fa.gq_omega + off + lane;
                        // the first rows' weights are on their way while the node's completeness sum is made
                        double o[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) o[i] = i < nr ? om[i * 64] : 0.0;
                        asm volatile("; LF_BEGIN qT items=1");
                        Tl[lane] = fexp_c(fma(uni(sc[R_C1]), nd.L - uni(sc[R_LSTAR]), uni(sc[R_C0])) - nd.PGL * uni(sc[R_Q]), &tab);
                        asm volatile("; LF_END qT");
                        // the bin's faintest node is its last (the nodes descend), for the bright form of the field sum
                        const double a4min = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(nd.a4), 63),
                                                              __builtin_amdgcn_readlane(__double2loint(nd.a4), 63));
                        const bool bright = kc.specialise && alphaC > 0.0 && a4min * vmin > 37.5;
                        if (CENSUS && kc.forms && lane == 0)
                            atomicAdd(kc.forms + (bright ? FORM_NODE_BRIGHT : FORM_NODE_GENERAL), (unsigned long long)(64 * kc.nf));
                        const double s = field_sum_nf(kc, r, alphaC, nd.x, nd.a4, &tab, bright);
                        __builtin_amdgcn_wave_barrier();      // (this wave's LDS writes are in order before its reads; the compiler keeps them so)
                        asm volatile("; LF_BEGIN qdot items=1");
                        double R = 0.0;
#pragma unroll 1
                        for (int k0 = 0; k0 < nr; k0 += 8) {
                            double on[8];
#pragma unroll
                            for (int i = 0; i < 8; ++i) on[i] = k0 + 8 + i < nr ? om[(k0 + 8 + i) * 64] : 0.0;
#pragma unroll
                            for (int i = 0; i < 8; ++i) R = fma(Tl[min(k0 + i, 63)], o[i], R);
#pragma unroll
                            for (int i = 0; i < 8; ++i) o[i] = on[i];
                        }
                        asm volatile("; LF_END qdot");
                        bsum = fma(R, s, bsum);
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
            if (v < nw) {
                bsum = wave_sum_dpp(bsum);        // lane 63: the wave's total
                const int ln = fresh_tid() & 63;
                double* __restrict__ row = fa.partB + (size_t)(w0 + v) * fa.nslot;
                if (ln == 63) pstore(row + vr, bsum);
            }
            }
        } else
        // ---- ... or over the lattice: 64 nodes per chunk, one WALKER PER WAVE (lane = node), every wave
        // on its own through this workgroup's share of the chunks - no LDS, no barrier, one partial per (walker, chunk) from
        // the DPP network.  (As workgroup-wide items of 512 nodes x 4 walkers, with two barriers and an LDS reduction each,
        // the grid took 16 us of a 34-us launch for 6 us worth of instructions.)
        {
            const int v = wave_base >> 6;
            const int nch64 = fa.nchB;            // chunks of 64 nodes
#pragma unroll 1
            for (int vr = frank; vr < VF; vr += fgroup) {
            double bsum = 0.0;
            if (v < nw && vr < nch64) {
                const double* __restrict__ sc = wsc + v * 8;
                const int mode = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(wfc + (v * MAXF) * 8 + 4));
                // per-field constants of this walker as the grid forms expect them: r[RF(f, F_CA)], r[RF(f, F_V)]
                const double* __restrict__ r = wfc + v * MAXF * 8 - 8;   // RF(f, slot) = 8 + 8 f + slot
                double vmin = r[RF(0, F_V)];
                for (int f = 1; f < kc.nf; ++f) vmin = fmin(vmin, r[RF(f, F_V)]);
                vmin = uni(vmin);
                const double alphaC = uni(sc[R_ALPHAC]);
                // (source-sharded ranks split the grid: this context integrates the chunks c with c % parts == part)
                auto mine = [&](int c) { return mode != MODE_SKIP && !(kc.grid_parts > 1 && c % kc.grid_parts != kc.grid_part); };
                struct Node {
                    double G, PG, W, a3, a4, a4min;
                };
                auto load_nodes = [&](int c) -> Node {      // (one 64-byte record per lane; see load_cells)
                    const int lane = fresh_tid() & 63;
                    const double2* __restrict__ src = reinterpret_cast<const double2*>(fa.nodes8 + ((size_t)c * 64 + lane) * 8);
                    const double2 a = src[0], b = src[1], d = src[2];
                    return Node{a.x, a.y, b.x, b.y, d.x, d.y};
                };
                Node nx = load_nodes(vr);      // the next chunk's nodes are in flight while the current one is summed
#pragma unroll 1
                for (int c = vr; c < nch64; c += VF) {
                    const Node nd = nx;
                    if (c + VF < nch64) nx = load_nodes(c + VF);
                    if (mine(c)) {
                        const double a4min = uni(nd.a4min);            // the chunk's faintest node, for the bright form of the field sum
                        const double T = fexp_c(fma(uni(sc[R_C1]), nd.G - uni(sc[R_LSTAR]), uni(sc[R_C0])) - nd.PG * uni(sc[R_Q]), &tab);
                        const bool bright = kc.specialise && alphaC > 0.0 && a4min * vmin > 37.5;
                        if (CENSUS && kc.forms && (fresh_tid() & 63) == 0)
                            atomicAdd(kc.forms + (bright ? FORM_NODE_BRIGHT : FORM_NODE_GENERAL),
                                      (unsigned long long)(min(64, na.nnodes - c * 64) * kc.nf));
                        const double s = field_sum_nf(kc, r, alphaC, nd.a3, nd.a4, &tab, bright);
                        bsum = fma(nd.W * T, s, bsum);
                    }
                }
            }
            if (nch64 > 0 && v < nw) {
                bsum = wave_sum_dpp(bsum);        // lane 63: the wave's total
                const int ln = fresh_tid() & 63;
                double* __restrict__ row = fa.partB + (size_t)(w0 + v) * fa.nslot;
                if (ln == 63) pstore(row + vr, bsum);
            }
            }
        }
#ifdef LF_STAMPS
        t_grid = __builtin_amdgcn_s_memtime();
#endif
        // ---- what is left are the source chunks, for walkers that cannot use the cells: claimed from the per-XCD queues
        int item = sitem[0];
#pragma unroll 1
        while (item >= 0) {
#ifdef LF_STAMPS
            const unsigned long long ta = __builtin_amdgcn_s_memtime();
            if (t_first == 0) t_first = ta;
            unsigned long long tb = ta, tc = ta;
#endif
            __syncthreads();                      // [D] the previous item's reduction has read `red` (and sitem)
            int t = fresh_tid();
            {
                // ================= catalogue chunk: piece A =================
                const int c = item;
                // (wave-uniform by construction; said so, or they sit - and are spilled - in vector registers)
                const int s0 = uni(sa.chunk_start[c]), n = uni(sa.chunk_len[c]), fld = uni(sa.chunk_field[c]);
                // (kamax: the largest alpha_C key for which THIS WAVE's lanes fit one table piece each)
                const int kfirst = uni(sa.chunk_keys[KEY_STRIDE * c]), klast = uni(sa.chunk_keys[KEY_STRIDE * c + 1]),
                          kamax = uni(sa.chunk_keys[KEY_STRIDE * c + 4 + (wave_base >> 6)]);
                // switch in: every lane loads its own ST flux-neighbours, 64 contiguous bytes (four 16-byte loads; the four
                // touch the same cache lines).  Not the coalesced pattern - a wave instruction spans 4 KB - but the kernel
                // streams 8 MB per launch against hundreds of microseconds of arithmetic: what counted was the staging
                // through LDS it replaces (a transposition and two more workgroup barriers per item, 17 % of a
                // workgroup's life in tools/stamps.py).  Slots past the end of a ragged chunk hold copies of its last source.
                double x[ST];
                if (t == 0) take_ticket();        // in front of the loads: back when they are
                if (n == PB * ST) {               // (wave-uniform; all but the last chunk of a field)
                    typedef double __attribute__((ext_vector_type(2), aligned(8))) double2_a8;      // (chunks start at any source)
                    const double2_a8* __restrict__ src = reinterpret_cast<const double2_a8*>(sa.a1 + s0 + t * ST);
#pragma unroll
                    for (int k = 0; k < ST / 2; ++k) {
                        const double2_a8 a = src[k];
                        x[2 * k] = a.x;
                        x[2 * k + 1] = a.y;
                    }
                } else {
                    const double* __restrict__ src = sa.a1 + s0;
#pragma unroll
                    for (int k = 0; k < ST; ++k) x[k] = src[min(t * ST + k, n - 1)];
                }
                if (t == 0) {
                    asm volatile("" ::"v"(x[ST - 1]));              // (the loads - and the claim issued before them - have arrived)
                    sitem[0] = redeem();
                }
#ifdef LF_STAMPS
                tb = __builtin_amdgcn_s_memtime();
#endif
                // slots past the end of the chunk hold copies of its last source
                const int npad = ST - min(max(n - t * ST, 0), ST);
                const int nwave = min(max(n - wave_base * ST, 0), 64 * ST);      // real sources of this wave (census)
                // ---- pass 1: the walkers whose (walker, chunk) pair takes the table-driven form (the bulk)
                int rest = 0;                     // bit w: walker w needs pass 2 (wave-uniform)
                WalkerK pn = fetch(0, fld);       // the next walker's constants are read while this one's terms run
#pragma unroll 1
                for (int w = 0; w < nw; ++w) {
                    const WalkerK p = pn;
                    pn = fetch(min(w + 1, nw - 1), fld);
                    double acc = 0.0;
                    if ((cellmask >> w) & 1) {
                        // summed over the cells (its partials for source chunks are never read)
                    } else if (p.mode < MODE_SKIP && p.mode != MODE_SLOW && kc.tables && kfirst >= p.klo && klast <= p.khi && p.kac <= kamax) {
                        TabCoef C;
                        if (kc.specialise && kfirst >= p.kne) {
                            asm volatile("; LF_BEGIN table_noexp items=%0" ::"n"(ST));
                            table_lookup<ST>(C, x, p, true, &tt);
                            acc = table_terms<ST, true>(C, x, npad);
                            asm volatile("; LF_END table_noexp");
                            if (CENSUS && kc.forms && (t & 63) == 0) atomicAdd(kc.forms + FORM_TABLE_NOEXP, (unsigned long long)nwave);
                        } else {
                            asm volatile("; LF_BEGIN table items=%0" ::"n"(ST));
                            table_lookup<ST>(C, x, p, false, &tt);
                            acc = table_terms<ST, false>(C, x, npad);
                            asm volatile("; LF_END table");
                            if (CENSUS && kc.forms && (t & 63) == 0) atomicAdd(kc.forms + FORM_TABLE, (unsigned long long)nwave);
                        }
                    } else if (p.mode < MODE_SKIP) {
                        rest |= 1 << w;
                    } else if (CENSUS && kc.forms && (t & 63) == 0) {
                        // -inf already (outside the prior, or the brightest source underflows): nothing to sum
                        atomicAdd(kc.forms + FORM_SKIPPED, (unsigned long long)nwave);
                    }
                    red[w * PB + t] = acc;
                }
                // ---- pass 2 (rare at the catalogue sizes this kernel serves): pairs outside the tables' reach - the
                // sparse tails of a field, extreme walkers - in the general form (lf_math.h: table exp / log, one rsqrt
                // seed), and walkers that may underflow on the careful path (device-library math, per-term checks, -inf
                // poisoning).  Sources re-read from memory in a rolled loop: none of this may claim registers next to
                // the table form's.
                if (rest) {
                    t = fresh_tid();              // (pass 2's addresses are made here, not carried through pass 1)
                    // The general forms run on the wave's own lanes (its sources are in registers; whether a pair takes the
                    // table form is decided per wave) and also need 10^(logf + 17).  Only the careful path, which every
                    // wave of the chunk takes together (the mode is the walker's), strides over the whole chunk.
                    double u[ST];
                    {
                        const double* __restrict__ up = sa.U + s0;
#pragma unroll
                        for (int k = 0; k < ST; ++k) u[k] = up[min(t * ST + k, n - 1)];
                    }
                    const double a1_first = uni(x[0]), u_first = uni(u[0]);     // sorted by flux: the wave's faintest source
                    const int nmine = ST - npad;                                // real sources of this lane
#pragma unroll 1
                    for (int w = 0; w < nw; ++w) {
                        if (!((rest >> w) & 1)) continue;
                        // (the walker's record: in memory, or - one-launch form - in LDS: scalars in wsc, per-field values in wfc / wlf)
                        const double* __restrict__ r = FUSED ? wsc + w * 8 : wrec + (size_t)(w0 + w) * REC;
                        const double* __restrict__ rf = wfc + (w * MAXF + fld) * 8;
                        const int mode = FUSED ? uni(*reinterpret_cast<const int*>(rf + 4)) : wmode[((size_t)(w0 + w) * MAXF + fld) * WM];
                        const double r_lf = FUSED ? wlf[w * MAXF + fld] : r[RF(fld, F_LF)];
                        const double r_v = FUSED ? rf[F_V] : r[RF(fld, F_V)], r_ca = FUSED ? rf[F_CA] : r[RF(fld, F_CA)];
                        double acc = 0.0;
                        int form;
                        if (mode == MODE_SLOW) {
                            form = FORM_CAREFUL;
                            const WFree wf{r[R_LSTAR], r[R_C0] + kc.lnom0_src[fld], r[R_C1], r[R_Q], r[R_ALPHAC],
                                           r_lf, r_v, kc.lnom0_src[fld], 0.0};
#pragma unroll 1
                            for (int i = t; i < n; i += PB) {
                                const size_t g = (size_t)s0 + i;
                                acc += term_free_careful(wf, sa.lum[g], sa.a1[g], sa.P[g], sa.U[g]);
                            }
                        } else {
                            WFree wf{};
                            wf.alphaC = r[R_ALPHAC];
                            wf.cA = r_ca;
                            wf.V = r_v;
                            const bool upper = kc.specialise && wf.alphaC > 0.0 && fma(wf.alphaC, a1_first, wf.cA) >= 0.0;
                            if (upper && u_first * wf.V > 37.5) {
                                form = FORM_GENERAL_NOEXP;
#pragma unroll
                                for (int k = 0; k < ST; ++k) {
                                    asm volatile("; LF_BEGIN general_noexp items=1");
                                    const double term = term_free_noexp(wf, x[k], &tab);
                                    asm volatile("; LF_END general_noexp");
                                    acc += k < nmine ? term : 0.0;
                                    __builtin_amdgcn_sched_barrier(0);      // one term at a time (registers)
                                }
                            } else {
                                form = FORM_GENERAL;
#pragma unroll
                                for (int k = 0; k < ST; ++k) {
                                    asm volatile("; LF_BEGIN general items=1");
                                    const double term = term_free_fast(wf, x[k], u[k], &tab);
                                    asm volatile("; LF_END general");
                                    acc += k < nmine ? term : 0.0;
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            }
                        }
                        if (CENSUS && kc.forms && (t & 63) == 0) atomicAdd(kc.forms + form, (unsigned long long)nwave);
                        red[w * PB + t] = acc;
                    }
                }
#ifdef LF_STAMPS
                tc = __builtin_amdgcn_s_memtime();
#endif
                __syncthreads();                  // [C]
                t = fresh_tid();
                reduce_store512<FUSED>(red, 0, nw, fa.partA, (size_t)fa.nchA, w0, c, t);
                item = sitem[0];
            }
#ifdef LF_STAMPS
            ++nitems_done;
            if (tb != ta) {
                t_sw += tb - ta;
                t_loop += tc - tb;
                t_red += __builtin_amdgcn_s_memtime() - tc;
            }
#endif
        }
        if (FUSED && fa.poll && uni(snosrc)) {
            // Hand-over by polling (PART_EMPTY above): this workgroup's partial sums are on their way, written through; only
            // the tile's finisher has more to do.
            if (frank == fgroup - 1) {
                const int t = fresh_tid(), v = t >> 6, ln = t & 63;
                const int nB = fa.nchB > 0 ? fa.nslot : 0, nC = fa.nchC > 0 ? fa.nslot : 0;
                if (v < nw) {
                    double* __restrict__ pb = fa.partB + (size_t)(w0 + v) * fa.nslot;
                    double* __restrict__ pc = fa.partC + (size_t)(w0 + v) * fa.nslot;
                    double pre[2] = {0.0, 0.0};
                    int tries = 0;
                    bool have;
                    do {
                        if (ln < nC) pre[0] = __hip_atomic_load(pc + ln, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (ln < nB) pre[1] = __hip_atomic_load(pb + ln, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        have = !((ln < nC && (unsigned long long)__double_as_longlong(pre[0]) == PART_EMPTY) ||
                                 (ln < nB && (unsigned long long)__double_as_longlong(pre[1]) == PART_EMPTY));
                    } while (!__all(have) && ++tries < PART_POLLS);
                    if (tries >= PART_POLLS) {        // (cannot happen while the device runs the launch's other workgroups)
                        pre[0] = pre[1] = __builtin_nan("");
                        if (ln == 0) atomicExch(fa.err, 1);
                    }
                    // the slots empty again for the next launch (visible to it: a kernel boundary lies between)
                    if (ln < nC) __hip_atomic_store(pc + ln, __longlong_as_double((long long)PART_EMPTY), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (ln < nB) __hip_atomic_store(pb + ln, __longlong_as_double((long long)PART_EMPTY), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    finalize_wave<true>(fa.partA, 0, fa.nchA, fa.partB, nB, nB, nC > 0 ? fa.partC : nullptr, nC, (int)STAT_CELLS,
                                        sstat - w0, sbase - w0, w0 + v, ln, ap, fa.out, nullptr, nullptr, 0,
                                        STEP ? sprop + v * 16 : nullptr, STEP ? szz + v : nullptr, STEP ? spre + 2 * v : nullptr, pre);
                }
            }
        } else if (FUSED) {
            // This workgroup's partial sums are out - written through, and complete once its waves have waited for their
            // stores' acknowledgements (the explicit s_waitcnt: the compiler does not emit one for a workgroup-scope fence,
            // and the count must not overtake a partial sum on its way to memory); the count (an atomic of agent scope)
            // comes after the barrier.  The last of the tile's workgroups to count adds the partials up,
            // reading them from memory (finalize_wave<true>); the walkers' records it needs are its own copies.
            // (Tiles with source work: their per-chunk sums have no fixed writer to poll for.)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (every wave: its write-through stores have been acknowledged)
            __threadfence_block();
            __syncthreads();
            if (fresh_tid() == 0) sitem[1] = atomicAdd(q, 1);
            __syncthreads();
            if (sitem[1] == fgroup - 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const int t = fresh_tid(), v = t >> 6;
                const int nB = fa.nchB > 0 ? fa.nslot : 0, nC = fa.nchC > 0 ? fa.nslot : 0;
                if (v < nw) {
                    finalize_wave<true>(fa.partA, fa.nchA, fa.nchA, fa.partB, nB, nB, nC > 0 ? fa.partC : nullptr, nC, (int)STAT_CELLS,
                                        sstat - w0, sbase - w0, w0 + v, t & 63, ap, fa.out, nullptr, nullptr, 0,
                                        STEP ? sprop + v * 16 : nullptr, STEP ? szz + v : nullptr, STEP ? spre + 2 * v : nullptr);
                    if (fa.poll) {                // (the slots empty again: the next launch's tiles may poll)
                        const int ln = t & 63;
                        if (ln < nC) __hip_atomic_store(fa.partC + (size_t)(w0 + v) * fa.nslot + ln, __longlong_as_double((long long)PART_EMPTY), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (ln < nB) __hip_atomic_store(fa.partB + (size_t)(w0 + v) * fa.nslot + ln, __longlong_as_double((long long)PART_EMPTY), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (t < QSTRIDE) q[t] = 0;        // the tile's counters, for the next launch
            }
        }
    }
#ifdef LF_STAMPS
    if (stamp && tid == 0) {
        stamp[1] = __builtin_amdgcn_s_memtime();
        const bool noitems = nitems_done == 0;
        stamp[2] = (unsigned long long)nitems_done;
        if (noitems) {
            // no source items (every walker on the cells: the normal case): the time line of the one launch instead - tables +
            // preparation + records up to t_prep, the cells up to t_cells, the grid up to t_grid, count and final sums after
            stamp[3] = t_prep - stamp[0];
            stamp[4] = t_cells - stamp[0];
            stamp[7] = t_grid - stamp[0];
            // (the preparation's inner time line goes where the items count would be: 4 x 16 bits, units of 16 cycles)
            unsigned long long pk = 0;
            pk = ((s_targs - stamp[0]) >> 4) & 0xfffull;
            for (int i = 0; i < 4; ++i) pk |= (((s_tprep[i] - stamp[0]) >> 4) & 0xfffull) << (12 * (i + 1));
            stamp[2] = pk;
        } else {
            stamp[3] = t_first ? t_first - stamp[0] : 0;      // prologue: tables, tile constants, first claim
            stamp[4] = t_sw;                                  // catalogue items: barrier D, loads, LDS transposition (wave 0's view)
            stamp[7] = t_loop;                                //                  walker loops
        }
        stamp[6] = __builtin_amdgcn_s_memrealtime();
        kc.stamps[(size_t)gridDim.x * 8 + blockIdx.x] =             // barrier C, reduction, stores  (second table behind the first)
            !noitems ? t_red : (((s_ttab - stamp[0]) << 32) | ((s_tp0 - stamp[0]) & 0xffffffffull));
    }
#endif
}

template <int ST, bool CENSUS, bool FUSED = false>
__global__ __launch_bounds__(PB, 4) void lf_free(KConst kc, SrcArrays sa, NodeArrays na, const double* __restrict__ wrec_arg,
                                                 const int* __restrict__ wmode_arg, FreeArgs fa) {
#ifdef LF_STAMPS
    const unsigned long long t_pre = __builtin_amdgcn_s_memtime();
#else
    const unsigned long long t_pre = 0;
#endif
    warm_kernarg<sizeof(KConst) + sizeof(SrcArrays) + sizeof(NodeArrays) + 2 * 8 + sizeof(FreeArgs)>();      // (lf_math.h: one round trip)
    lf_free_body<ST, CENSUS, FUSED, false>(kc, sa, na, wrec_arg, wmode_arg, fa, StepArgs{}, AcceptArgs{}, t_pre);
}

template <int ST>
__global__ __launch_bounds__(PB, 4) void lf_free_step(KConst kc, SrcArrays sa, NodeArrays na, FreeArgs fa, StepArgs sp, AcceptArgs ap) {
    warm_kernarg<sizeof(KConst) + sizeof(SrcArrays) + sizeof(NodeArrays) + sizeof(FreeArgs) + sizeof(StepArgs) + sizeof(AcceptArgs)>();
    lf_free_body<ST, false, true, true>(kc, sa, na, nullptr, nullptr, fa, sp, ap);
}

}  // namespace lf
