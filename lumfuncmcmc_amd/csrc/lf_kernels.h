// lf_kernels.h - HIP kernels of the per-step log-posterior (gfx950, fp64 VALU).
//
// Work decomposition (both pieces): a workgroup of 256 threads owns (one chunk of items) x (one
// tile of up to TW walkers).  Items - catalogue sources for piece A, integration-grid nodes for
// piece B - are loaded ONCE, coalesced, into registers (ST per lane) and stay there while the
// workgroup walks over its walkers.  Everything walker-only is wave-uniform, so it lives in SGPRs
// (scalar loads from the walker record; no LDS traffic, no VGPRs).  Per walker each lane adds its
// ST terms and parks the sum in LDS; after the walker loop the four waves reduce the TW x 256
// sums with 64-lane __shfl_down trees and write one partial per (chunk, walker) to a fixed slot.
// lf_finalize adds the partials in a fixed order: results are bitwise reproducible for a given
// launch geometry.
//
// The arithmetic is that of SURVEY.md App. A (reference: lumfuncmcmc.py:44, :69-70, :370-377,
// :388-392; lumfuncmcmc_z.py:40-42, :63-67, :371-375; VmaxLumFunc.py:118-127, :141, :164-167),
// evaluated in log space with everything source-only or walker-only hoisted:
//     10^(lum_i - L*)   = P_i * Q_w        P_i = 10^(lum_i - 42),  Q_w = 10^(42 - L*_w)
//     f_i / f_tau(w,f)  = U_i * V_wf       U_i = 10^(logf_i + 17), V_wf = 1 / (Flim_f 10^b_w)
//
// Underflow convention: the reference takes log(product) in linear space and
// returns -inf as soon as one source's product rounds to 0.  lf_prepare classifies every
// (walker, field) from per-field extremes of the catalogue:
//     NEGINF  the brightest source already underflows exp(-10^(lum-L*))   -> lnprob = -inf, exactly
//     FAST    a lower bound of every factor is far above the underflow threshold (the normal case)
//             -> branch-free terms with the tuned math of lf_math.h, no per-term checks
//     SLOW    anything else -> per-term checks with the device-library math, -inf poisoning
// The mode is per (walker, field) and wave-uniform inside the walker loop: one scalar branch per walker,
// and a walker's result never depends on which other walkers share its tile.
#pragma once
#include "lf_math.h"

namespace lf {

constexpr int BLOCK = 256;   // 4 waves
constexpr int MAXF = 8;
constexpr int REC = 8 + 8 * MAXF;   // doubles per walker record: 8 walker scalars, then one block of 8 per field
constexpr int KEY_STRIDE = 12;   // ints per chunk in SrcArrays::chunk_keys
constexpr int WM = 8;        // ints per (walker, field) in wmode: {mode, klo, khi, kne, kaC, -, -, -}
// SKIP: the walker failed the prior - lnprob is -inf whatever the sums are (the reference returns before lnlike,
// lumfuncmcmc.py:408): neither its terms nor its grid nodes are evaluated.  SKIPSRC: piece A is already known to be
// -inf (NEGINF); the grid integral is still computed (lf_lnprob_pieces reports it).
enum { MODE_FAST = 0, MODE_SLOW = 1, MODE_NEGINF = 2, MODE_SKIP = 3, MODE_SKIPSRC = 4 };
enum { STAT_PRIOR_OK = 1, STAT_NEGINF = 2, STAT_SLOW = 4,      // SLOW: some field of the walker takes the careful path
       STAT_CELLS = 8 };   // FREE: piece A of this walker is summed over the catalogue's CELLS (lf_free.h), not its sources
// Cells (FREE, real catalogue): a cell is a run of flux-neighbouring sources of one field no wider than 2 rho, stored as
// {x_c, S_0 .. S_8}: its midpoint and the power sums S_j = sum_k (x_k - x_c)^j.  On a table piece the term is the
// product of two degree-7 polynomials, g(t_c + sa d) h(u_c + d) = sum_j c_j d^j, so the cell's sum over its sources
// is sum_j c_j S_j - whatever the number of sources.  Orders above CELL_M are dropped: with alpha_C rho <= CELL_RHO_G
// and rho <= CELL_RHO_H their share of any term is below 2e-18 for every piece of both tables
// (tests/test_tables_cpu.py recomputes that bound from the tables' coefficients).  Both radii stay inside the tables'
// margins (a cell lies within the margins of its midpoint's pieces).  (With CELL_M = 6 the same bound needed cells a
// third as wide: three times the cells at 0.82 of the cost each.)
constexpr int CELL_M = 8;
constexpr int CELL_REC = CELL_M + 2;         // doubles per cell record
constexpr double CELL_RHO_G = 1.5e-2, CELL_RHO_H = 3.0e-3;
// ZEVOL has cells too, in REDSHIFT.  What is left of the term per source is v_i = 10^(lum_i - L*(z_i)) with L* a
// quadratic aL z^2 + bL z + cL of the walker (lumfuncmcmc_z.py:66): about a cell's midpoint z_c, with d = z_i - z_c,
//     v_i = 10^(lum_i - LREF) * 10^(LREF - L*(z_c)) * exp(a d + b d^2),   a = -ln10 L*'(z_c),  b = -ln10 aL
// and exp(a d + b d^2) = sum_j c_j d^j with c_0 = 1, j c_j = a c_(j-1) + 2 b c_(j-2): an entire function, so the sum over
// the cell is 10^(LREF - L*(z_c)) sum_j c_j S_j with the WEIGHTED power sums S_j = sum_i 10^(lum_i - LREF) d_i^j - all
// weights positive, so a relative error of the series is a relative error of the cell's sum.  Cells are at most
// 2 rho wide, and a walker takes them when |a| rho <= ZCELL_X1 over the field's redshifts and |b| rho^2 <= ZCELL_X2
// (lf_prepare): the orders above CELL_M then sum to less than 1e-17 (tests/test_tables_cpu.py recomputes the bound from
// the majorant series exp(X1 t + X2 t^2)).  Records {z_c, S_0 .. S_6}: ZCELL_M = 6.  rho (KConst::zcell_rho) is chosen when the context is made, at most
// ZCELL_RHO and small enough that EVERY walker inside the prior box of L1..L3 passes (lfmcmc.hip: zcell_rho_for_box).
constexpr double ZCELL_RHO = 1.0e-3, ZCELL_X1 = 6.0e-3, ZCELL_X2 = 1.0e-5;
constexpr int ZCOLS = 3;                  // redshift columns a chunk of the column-major z-evolving grid may touch
constexpr int ZCELL_M = 6;
constexpr int ZCELL_REC = ZCELL_M + 2;

// walker record, FREE / FIXCOMP: walker scalars ...
enum { R_LSTAR = 0, R_C0 = 1, R_C1 = 2, R_Q = 3, R_ALPHAC = 4 };
// ... and per field f the block r[RF(f, .)]: everything the per-source loop needs for one (walker, field) sits in
// one 64-B line (one batch of scalar loads)
enum { F_LF = 0, F_V = 1, F_CA = 2, F_CY = 3 };   // lF = log10(1e-17 Flim); V = 1 / f_tau; cA = -alpha_C lF; cY = -(lF + b)
__host__ __device__ constexpr int RF(int f, int slot) { return 8 + 8 * f + slot; }
// integer keys of wmode[(w * MAXF + f) * WM + .], see lf_prepare: the per-(walker, chunk) choice of the term's form is
// made with scalar integer compares only
enum { M_MODE = 0, M_KLO = 1, M_KHI = 2, M_KNE = 3, M_KAC = 4 };
// slots of the census KConst::forms (lf_form_counts)
enum { FORM_GENERAL = 0, FORM_GENERAL_NOEXP = 1, FORM_TABLE = 2, FORM_TABLE_NOEXP = 3, FORM_CAREFUL = 4, FORM_SKIPPED = 5,
       FORM_NODE_GENERAL = 6, FORM_NODE_BRIGHT = 7, FORM_CELL = 8, FORM_COUNT = 9 };
constexpr double KEY_SCALE = 1048576.0;     // keys of log-flux: (x - x0) * 2^20, 1e-6 dex
constexpr double KEY_ASCALE = 65536.0;      // keys of alpha_C
constexpr int KEY_MAX = 2147483000;
// walker record, ZEVOL
enum { Z_AL = 0, Z_BL = 1, Z_CL = 2, Z_AP = 3, Z_BP = 4, Z_CP = 5, Z_C1 = 6 };

struct KConst {
    int variant, fix_sch_al, nf, S, ndim;
    int specialise;           // 1: chunk-level term specialisation (term_free_noexp); 0 for A/B runs
    int cells;                // FREE, ZEVOL: 1 = the catalogue's cells exist and lf_prepare may flag walkers STAT_CELLS
    int cc_fstart[MAXF + 1];  // FREE: cell chunks (64 cells) of field f are [cc_fstart[f], cc_fstart[f + 1])
    int zgrid_cols;           // ZEVOL: 1 = the grid's nodes are stored column by column (node = k S + j) and S >= BLOCK / (ZCOLS - 1),
                              // so that a chunk of BLOCK nodes touches at most ZCOLS redshift columns (gridsum_body)
    double zcell_rho;         // ZEVOL: half the largest width of a cell in redshift (ZCELL_RHO below)
    int kf_first[MAXF], kf_last[MAXF];   // FREE: keys (floor / ceil) of each field's faintest / brightest source
    int grid_part, grid_parts; // source-sharded ranks split piece B too: this context integrates the node chunks c with
                              // c % grid_parts == grid_part (the others contribute 0); 0 / 1 = the whole grid
    double lnom0_src[MAXF];   // ln(trunc(Omega_0[f]) / sqarcsec)   (int-truncated, lumfuncmcmc.py:285)
    double om0_grid[MAXF];    // Omega_0[f] / sqarcsec              (float, lumfuncmcmc.py:375)
    double fc_ratio;          // |a / (1 - a)|, a = (2 fcmin - 1)^2 (VmaxLumFunc.py:164-165)
    double lims[5][2];
    double pivots[3];
    double sch_al0, alpha0;
    double flim0[MAXF];
    // per-field extremes of the catalogue, for the mode classification
    int nsrc[MAXF];
    double pmax[MAXF];        // max 10^(lum-42)            (FREE, FIXCOMP)
    double lum_min[MAXF], lum_max[MAXF];
    double a_min[MAXF];       // FREE: min logf             FIXCOMP/ZEVOL: min ln(Om_arr)
    double u_min[MAXF];       // FREE: 10^(min logf + 17)
    double u_max[MAXF];       // FREE: 10^(max logf + 17)
    double z_lo[MAXF], z_hi[MAXF];   // ZEVOL
    double key_x0;            // FREE: origin of the integer keys of log-flux (the catalogue's smallest logf)
    int tables;               // FREE: 1 = table-driven form of the term where it applies (default), 0 = general form only
    // optional census of which form of the term / node ran (bench.py's flop accounting, tests): terms or node-fields
    // added per (walker, chunk) by one lane; NULL = off (the default: no atomics on the path)
    unsigned long long* forms;
#ifdef LF_STAMPS
    // diagnostic build (tools/stamps.py): s_memtime at four points of every source workgroup; never in the product
    unsigned long long* stamps;
#endif
    // per-field sums for the closed-form part of piece A:
    //   sum_i ln TrueLumFunc_i = n (ln ln10 + ln10 phi*) + c1 (sum(lum_i - 42) - n (L* - 42)) - Q sum P_i
    double slc[MAXF];         // sum (lum_i - 42)
    double sp[MAXF];          // sum 10^(lum_i - 42)
    double som[MAXF];         // FIXCOMP/ZEVOL: sum ln(Om_arr_i)
    double sz[MAXF], sz2[MAXF];   // ZEVOL: sum z_i, sum z_i^2  (L*(z), phi*(z) enter the log-terms linearly)
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

// a z^2 + b z + c with the reference's roundings (lumfuncmcmc_z.py:65-66): with close pivots the
// three terms cancel by two or three digits, so an FMA-contracted form drifts by ~1e-13 relative.
// (HIP's __dmul_rn / __dadd_rn are plain * and + and may be contracted: the pragma is what holds.)
__device__ __forceinline__ double quad_nofma(double a, double b, double c, double z, double z2) {
#pragma clang fp contract(off)
    const double t2 = a * z2;
    const double t1 = b * z;
    const double s = t2 + t1;
    return s + c;
}

// a z^2 + b z + c to ~1 ulp (error-free products and sums, z^2 exact): the compressed catalogue evaluates the
// quadratic at a few hundred nodes that stand for ~1e6 sources, so it wants the value the reference's per-source
// roundings scatter around, not one draw of that scatter.
__device__ __forceinline__ double quad_comp(double a, double b, double c, double z) {
#pragma clang fp contract(off)
    const double z2 = z * z, e2 = fma(z, z, -z2);
    const double p = a * z2, ep = fma(a, z2, -p) + a * e2;
    const double q = b * z, eq = fma(b, z, -q);
    const double s = p + q, bs = s - p, es = (p - (s - bs)) + (q - bs);
    const double r = s + c, br = r - s, er = (s - (r - br)) + (c - br);
    return r + ((ep + eq) + (es + er));
}

__device__ inline void quad_range(double a, double b, double c, double lo, double hi, double& mn, double& mx) {
    const double v0 = quad_nofma(a, b, c, lo, lo * lo), v1 = quad_nofma(a, b, c, hi, hi * hi);
    mn = fmin(v0, v1);
    mx = fmax(v0, v1);
    if (a != 0.0) {
        const double zv = -b / (2.0 * a);
        if (zv > lo && zv < hi) {
            const double vv = quad_nofma(a, b, c, zv, zv * zv);
            mn = fmin(mn, vv);
            mx = fmax(mx, vv);
        }
    }
}

// ----------------------------------------------------------------------------------------------
// device-resident ensemble sampler (Goodman & Weare stretch move, the parallel form with two fixed
// half-ensembles that emcee 2.x - the API the reference calls, lumfuncmcmc.py:489-491 - uses).
// One half-step = propose (inside lf_prepare) -> lf_main -> accept (inside lf_finalize): theta never
// leaves HBM and the host only enqueues.  Random numbers: Philox4x32-10, counter = (step, half,
// walker, stream), key = seed, so a chain is a pure function of (seed, start) - tests replay it on
// the host with the same generator.
// ----------------------------------------------------------------------------------------------
struct StepArgs {
    int enabled;                 // 0: plain lnprob call (theta rows given)
    int half, halfW, ndim;
    unsigned long long step, seed;
    double a;                    // stretch scale (2.0)
    const double* pos;           // [W][ndim] current positions
    double* prop;                // [halfW][ndim] proposals of the active half
    double* zz;                  // [halfW] stretch factors
};
struct AcceptArgs {
    int enabled;
    int half, halfW, ndim;
    unsigned long long step, seed;
    long long t, cap;            // chain slot of this step, chain capacity (steps)
    double* pos;                 // [W][ndim]
    double* lnp;                 // [W]
    const double* prop;          // [halfW][ndim]
    const double* zz;            // [halfW]
    long long* nacc;             // [W]
    double* chain;               // [W][cap][ndim]
    double* chain_lnp;           // [W][cap]
};

__device__ __forceinline__ void philox4x32(unsigned int c0, unsigned int c1, unsigned int c2, unsigned int c3,
                                           unsigned int k0, unsigned int k1, unsigned int (&out)[4]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c1 ^ k0;
        const unsigned int n1 = (unsigned int)p1;
        const unsigned int n2 = (unsigned int)(p0 >> 32) ^ c3 ^ k1;
        const unsigned int n3 = (unsigned int)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// 53-bit uniform in [0, 1) from two words
__device__ __forceinline__ double u53(unsigned int hi, unsigned int lo) {
    return (double)((((unsigned long long)hi << 32) | lo) >> 11) * 1.1102230246251565e-16;
}
// z = ((a - 1) u + 1)^2 / a and y = x_j - (x_j - x_k) z, each operation rounded on its own (no FMA
// contraction) so that a host replay with the same random numbers gives the same bits
__device__ __forceinline__ double stretch_z(double a, double u) {
#pragma clang fp contract(off)
    const double t = (a - 1.0) * u;
    const double g = t + 1.0;
    const double g2 = g * g;
    return g2 / a;
}
__device__ __forceinline__ double stretch_point(double xj, double xk, double z) {
#pragma clang fp contract(off)
    const double d = xj - xk;
    const double dz = d * z;
    return xj - dz;
}
__device__ __forceinline__ void sampler_draw(unsigned long long step, int half, int w, int stream,
                                             unsigned long long seed, unsigned int (&r)[4]) {
    philox4x32((unsigned int)step, (unsigned int)(step >> 32) ^ ((unsigned int)half << 31), (unsigned int)w,
               (unsigned int)stream, (unsigned int)seed, (unsigned int)(seed >> 32), r);
}

// ----------------------------------------------------------------------------------------------
// prepare: theta rows -> walker records, prior flag, closed-form base, per-(walker, field) mode.
// MAXF = 8 lanes per walker: lane f does field f (the expensive part: one careful Fleming evaluation
// per field for the mode bound), every lane repeats the short walker-common part; the 8-lane groups
// are combined with __shfl_xor.  The kernel is pure latency (B is a few hundred), so the point is a
// short dependent chain.  set_parameters_from_list + lnprior: lumfuncmcmc.py:327-358,
// lumfuncmcmc_z.py:339-362.
// ----------------------------------------------------------------------------------------------
// Butterfly over the 8 lanes of a walker on the DPP network: quad_perm [1,0,3,2], quad_perm [2,3,0,1], then the mirrored
// lane of the 8 (row_half_mirror; by then every lane of a quad holds the quad's total, so any lane of the other quad will
// do).  The same sums in the same association as the __shfl_xor(1), (2), (4) butterfly it replaces - whose ds_bpermutes
// (two per step for a double, an LDS round trip each) were 1.4k cycles of the lone wave that prepares a tile.
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    return __hiloint2double(dpp_mov<CTRL>(__double2hiint(v)), dpp_mov<CTRL>(__double2loint(v)));
}
__device__ __forceinline__ double group8_sum(double v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    return v;
}
__device__ __forceinline__ int group8_or(int v) {
    v |= dpp_mov<0xB1>(v);
    v |= dpp_mov<0x4E>(v);
    v |= dpp_mov<0x141>(v);
    return v;
}

// conservative integer keys (NaN and out-of-range values give keys that fail every test)
__device__ __forceinline__ int key_ceil(double v) {
    if (!(v == v)) return KEY_MAX;
    return (int)fmin(fmax(ceil(v), 0.0), (double)KEY_MAX);
}
__device__ __forceinline__ int key_floor(double v) {
    if (!(v == v)) return -1;
    return (int)fmin(fmax(floor(v), -1.0), (double)(KEY_MAX - 1));
}

// slow: {count, list...} of the walkers flagged STAT_SLOW, for the rescue workgroups of the compressed-catalogue
// launch (order = arrival order of the atomics; every walker's sums go to its own slots, so the order is immaterial);
// lf_finalize resets the count.
// The work of one lane of lf_prepare: walker wq (8 lanes per walker, lane f = field f), `grp` = the walker's row of the
// 8 x 16 LDS staging area `sth`.  BLOCK_SYNC: the 64 lanes are a workgroup of their own (lf_prepare) and meet at a
// barrier; otherwise they are one wave of a larger workgroup (lf_free's fused prologue), in lockstep anyway.
// nqueue > 0: the launch is lf_free's (the table keys and the cells' flag are wanted).
template <bool BLOCK_SYNC, bool TOLDS = false, bool STEP = BLOCK_SYNC, int VAR = -1>
__device__ __forceinline__ void prepare_lane(const KConst& kc, const StepArgs& sp, const double* __restrict__ theta, int B,
                                             double* __restrict__ wrec, int* __restrict__ wstat,
                                             int* __restrict__ wmode, double* __restrict__ wbase, int* __restrict__ slow_list,
                                             int nqueue, int wq, int f, int grp, double (*sth)[16],
                                             double* __restrict__ l_fc = nullptr, double* __restrict__ l_sc = nullptr,
                                             int* __restrict__ l_stat = nullptr, double* __restrict__ l_base = nullptr,
                                             double* __restrict__ l_lf = nullptr, unsigned long long* tst = nullptr,
                                             double* __restrict__ l_prop = nullptr, double* __restrict__ l_zz = nullptr) {
#ifdef LF_STAMPS
#define LF_TST(i) do { if (tst && wq % 8 == 0 && f == 0) tst[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LF_TST(i) do { } while (0)
#endif
    LF_TST(0);
    const int variant = VAR >= 0 ? VAR : kc.variant;      // (the persistent kernels know theirs at compile time)
    // TOLDS (lf_free's one-launch form, FREE): the records go to the tile's LDS arrays INSTEAD of memory - l_fc per (walker,
    // field) 8 doubles {alpha_C, V, cA, cY - H_LO - 1 / (2 H_INV), {mode, klo, khi, kne, kaC} as ints, cY - H_LO}, l_lf per
    // (walker, field) lF, l_sc per walker the 5 scalars, l_stat the status word, l_base the closed-form part of piece A: the
    // launch that makes them is the only one that reads them, and nobody waits for a trip through memory
    const bool live = wq < B;
    const int w = live ? wq : B - 1;                 // idle groups replay the last walker, write nothing
    const bool has_f = f < kc.nf;
    // This lane's field's constants.  The per-field arrays of the kernel's arguments are indexed by a lane number, so they are
    // read from the argument segment by VECTOR loads (L2 hits: the scalar loads of warm_kernarg brought the lines in).  Left
    // where they are used, each sits behind its own guard with its own wait - thirteen L2 round trips one after the other,
    // ~6k cycles of the lone wave that prepares a tile (r03 stamps: 5k cycles between Q and the 8-lane combines for ~300
    // instructions).  Here they are all issued together, in front of the row of theta, and waited for once.
    struct {
        double pmax, lum_min, lum_max, a_min, u_min, u_max, lnom0, slc, sp, som, flim0, z_lo, z_hi, sz, sz2;
        int nsrc, kf_first, kf_last;
    } kf;
    // The prior box and the other scalars only the preparation reads, the same way (an index the compiler cannot see is zero
    // makes the loads vector loads): as scalar arguments they were fetched by every wave in the kernel's preamble, parked in
    // VGPR lanes and read back here lane by lane - ~100 of the lone wave's instructions, and two of the preamble's waits.
    struct {
        double lims[5][2], pivots[3], sch_al0, alpha0, fc_ratio, key_x0, zcell_rho;
    } pc;
    {
        int vz = 0;
        asm volatile("" : "+v"(vz));
        const double* __restrict__ lp = &kc.lims[0][0];
        const bool zv = variant == LF_ZEVOL;     // (what the other variant does not read stays 0 and costs no register)
#pragma unroll
        for (int i = 0; i < 10; ++i) pc.lims[i >> 1][i & 1] = zv && (i >> 1) >= LF_LIM_FLIM ? 0.0 : lp[i + vz];
        static_assert(LF_LIM_LSTAR < LF_LIM_FLIM && LF_LIM_PHISTAR < LF_LIM_FLIM && LF_LIM_SCH_AL < LF_LIM_FLIM && LF_LIM_ALPHA >= LF_LIM_FLIM,
                      "the z-evolving variant's limits come first");
#pragma unroll
        for (int i = 0; i < 3; ++i) pc.pivots[i] = zv ? (&kc.pivots[0])[i + vz] : 0.0;
        pc.sch_al0 = (&kc.sch_al0)[vz];
        pc.alpha0 = zv ? 0.0 : (&kc.alpha0)[vz];
        pc.fc_ratio = zv ? 0.0 : (&kc.fc_ratio)[vz];
        pc.key_x0 = zv ? 0.0 : (&kc.key_x0)[vz];
        pc.zcell_rho = zv ? (&kc.zcell_rho)[vz] : 0.0;
    }
    // (ndim <= 16: two elements of the theta row per lane; a sampler's half-step makes its own row below)
    const bool stepping = STEP && sp.enabled;
    double t0 = 0.0, t1 = 0.0;
    if (!stepping) {
        const double* __restrict__ row = theta + (size_t)w * kc.ndim;
        t0 = row[min(f, kc.ndim - 1)];
        t1 = row[min(f + 8, kc.ndim - 1)];
    }
    kf.lum_min = kc.lum_min[f]; kf.lum_max = kc.lum_max[f]; kf.a_min = kc.a_min[f]; kf.slc = kc.slc[f]; kf.som = kc.som[f];
    kf.nsrc = kc.nsrc[f];
    if (variant == LF_ZEVOL) {
        kf.z_lo = kc.z_lo[f]; kf.z_hi = kc.z_hi[f]; kf.sz = kc.sz[f]; kf.sz2 = kc.sz2[f];
        kf.pmax = kf.u_min = kf.u_max = kf.lnom0 = kf.sp = kf.flim0 = 0.0; kf.kf_first = kf.kf_last = 0;
        asm volatile("" : "+v"(t0), "+v"(t1), "+v"(kf.lum_min), "+v"(kf.lum_max), "+v"(kf.a_min), "+v"(kf.slc), "+v"(kf.som), "+v"(kf.nsrc),
                          "+v"(kf.z_lo), "+v"(kf.z_hi), "+v"(kf.sz), "+v"(kf.sz2));
    } else {
        kf.pmax = kc.pmax[f]; kf.u_min = kc.u_min[f]; kf.u_max = kc.u_max[f]; kf.lnom0 = kc.lnom0_src[f]; kf.sp = kc.sp[f];
        kf.flim0 = kc.flim0[f]; kf.kf_first = kc.kf_first[f]; kf.kf_last = kc.kf_last[f];
        kf.z_lo = kf.z_hi = kf.sz = kf.sz2 = 0.0;
        asm volatile("" : "+v"(t0), "+v"(t1), "+v"(kf.lum_min), "+v"(kf.lum_max), "+v"(kf.a_min), "+v"(kf.slc), "+v"(kf.som), "+v"(kf.nsrc),
                          "+v"(kf.pmax), "+v"(kf.u_min), "+v"(kf.u_max), "+v"(kf.lnom0), "+v"(kf.sp), "+v"(kf.flim0), "+v"(kf.kf_first),
                          "+v"(kf.kf_last));
    }
    asm volatile("" : "+v"(pc.lims[0][0]), "+v"(pc.lims[0][1]), "+v"(pc.lims[1][0]), "+v"(pc.lims[1][1]), "+v"(pc.lims[2][0]), "+v"(pc.lims[2][1]),
                      "+v"(pc.lims[3][0]), "+v"(pc.lims[3][1]), "+v"(pc.lims[4][0]), "+v"(pc.lims[4][1]), "+v"(pc.pivots[0]), "+v"(pc.pivots[1]),
                      "+v"(pc.pivots[2]), "+v"(pc.sch_al0), "+v"(pc.alpha0), "+v"(pc.fc_ratio), "+v"(pc.key_x0), "+v"(pc.zcell_rho));
    // theta row of this walker -> LDS: either the given row, or the stretch-move proposal
    //   y = x_j - (x_j - x_k) z,   z = ((a - 1) u + 1)^2 / a,   j uniform in the other half
    if (stepping) {                                // (STEP: this instantiation may be handed the sampler's half-step)
        unsigned int rr[4];
        sampler_draw(sp.step, sp.half, w, 0, sp.seed, rr);
        const double z = stretch_z(sp.a, u53(rr[0], rr[1]));
        const int j = (1 - sp.half) * sp.halfW + (int)(((unsigned long long)rr[2] * (unsigned long long)sp.halfW) >> 32);
        const int k = sp.half * sp.halfW + w;
        for (int i = f; i < sp.ndim; i += 8) {
            const double xj = sp.pos[(size_t)j * sp.ndim + i], xk = sp.pos[(size_t)k * sp.ndim + i];
            const double y = stretch_point(xj, xk, z);
            sth[grp][i] = y;
            // (TOLDS: the launch that proposes is the one that accepts - proposal and stretch factor stay in LDS: l_prop, l_zz)
            if (TOLDS) l_prop[grp * 16 + i] = y;
            else if (live) sp.prop[(size_t)w * sp.ndim + i] = y;
        }
        if (TOLDS) {
            if (f == 0) l_zz[grp] = z;
        } else if (live && f == 0) sp.zz[w] = z;
    } else {
        if (f < kc.ndim) sth[grp][f] = t0;
        if (f + 8 < kc.ndim) sth[grp][f + 8] = t1;
    }
    if (BLOCK_SYNC) __syncthreads();
    LF_TST(1);
    const double* th = sth[grp];
    double* r = wrec + (size_t)w * REC;
    const double SAFE = -700.0;
    bool ok = true;
    int neginf = 0, m = MODE_FAST;
    int cell_ok = 1;       // FREE: this lane's field may be summed over its cells (see below)
    double base = 0.0;     // this lane's share of the walker-only part of piece A (closed form)
    if (variant == LF_ZEVOL) {
        const double L1 = th[0], L2 = th[1], L3 = th[2], p1 = th[3], p2 = th[4], p3 = th[5];
        const double al = kc.fix_sch_al ? pc.sch_al0 : th[6];
        if (!kc.fix_sch_al) ok = ok && (al >= pc.lims[LF_LIM_SCH_AL][0]) && (al <= pc.lims[LF_LIM_SCH_AL][1]);
        const double Ls[3] = {L1, L2, L3}, ps[3] = {p1, p2, p3};
        for (int i = 0; i < 3; ++i) {           // strict for L and phi (lumfuncmcmc_z.py:355-358)
            ok = ok && (Ls[i] > pc.lims[LF_LIM_LSTAR][0]) && (Ls[i] < pc.lims[LF_LIM_LSTAR][1]);
            ok = ok && (ps[i] > pc.lims[LF_LIM_PHISTAR][0]) && (ps[i] < pc.lims[LF_LIM_PHISTAR][1]);
        }
        double aL, bL, cL, aP, bP, cP;
        quad_coef(L1, L2, L3, pc.pivots[0], pc.pivots[1], pc.pivots[2], aL, bL, cL);
        quad_coef(p1, p2, p3, pc.pivots[0], pc.pivots[1], pc.pivots[2], aP, bP, cP);
        const double c1 = LF_LN10 * (al + 1.0);
        if (has_f && kf.nsrc > 0) {
            // closed-form part: sum_i [ln Om_i + ln ln10 + ln10 phi*(z_i) + c1 (lum_i - L*(z_i))]; only
            // -10^(lum_i - L*(z_i)) is left per source.  (First thing: the field's sums are not held in registers any longer.)
            const double n = (double)kf.nsrc;
            const double sph = aP * kf.sz2 + bP * kf.sz + n * cP;
            const double sls = aL * kf.sz2 + bL * kf.sz + n * (cL - LF_LREF);
            base = kf.som + n * LF_LNLN10 + LF_LN10 * sph + c1 * (kf.slc - sls);
        }
        if (live && f == 0) {
            double* d = TOLDS ? l_sc + grp * 8 : r;
            d[Z_AL] = aL; d[Z_BL] = bL; d[Z_CL] = cL;
            d[Z_AP] = aP; d[Z_BP] = bP; d[Z_CP] = cP;
            d[Z_C1] = c1;
        }
        if (live && has_f) {
            // the local form of the term (srcsum_body) expands L*(z) about a lane's middle source: the size of its
            // exponent is bounded by this slope times the lane's width in z (a lane is narrower than 1 / 128 or the
            // chunk's key is 0 and the form is not taken)
            const double s0 = fabs(fma(2.0 * aL, kf.z_lo, bL)), s1 = fabs(fma(2.0 * aL, kf.z_hi, bL));
            (TOLDS ? l_fc + (grp * MAXF + f) * 8 : r + RF(f, 0))[0] = LF_LN10 * (fmax(s0, s1) + fabs(aL) * (1.0 / 128.0));
            // the field's cells in redshift can stand for its sources (ZCELL_RHO above; NaN coefficients fail the test)
            cell_ok = kf.nsrc == 0 || (LF_LN10 * fmax(s0, s1) * pc.zcell_rho <= ZCELL_X1 &&
                                          LF_LN10 * fabs(aL) * pc.zcell_rho * pc.zcell_rho <= ZCELL_X2);
        }
        if (has_f && kf.nsrc > 0) {
            double lsmn, lsmx, phmn, phmx;
            quad_range(aL, bL, cL, kf.z_lo, kf.z_hi, lsmn, lsmx);
            quad_range(aP, bP, cP, kf.z_lo, kf.z_hi, phmn, phmx);
            const double tmax = kf.lum_max - lsmn, tmin = kf.lum_min - lsmx;
            // an UPPER bound of 10^tmax is all the test needs (single-precision hardware exp2, rounded up; NaN fails the test)
            const double vb = tmax < 2.9 ? (double)(__builtin_amdgcn_exp2f((float)tmax * 3.3219285f) * 1.0001f) + 1.0e-30 : 1.0e300;
            const double lb = LF_LNLN10 + LF_LN10 * phmn + fmin(c1 * tmin, c1 * tmax) - vb;
            m = (vb < 700.0 && lb > SAFE && lb + kf.a_min > SAFE) ? MODE_FAST : MODE_SLOW;
        }
    } else {
        const double Lstar = th[0], phistar = th[1];
        int k = 2;
        const double al = kc.fix_sch_al ? pc.sch_al0 : th[k++];
        const double alphaC = variant == LF_FREE ? th[k + kc.nf] : pc.alpha0;
        // inclusive box on all five named parameters, fixed ones too (lumfuncmcmc.py:346-354)
        ok = ok && (Lstar >= pc.lims[LF_LIM_LSTAR][0]) && (Lstar <= pc.lims[LF_LIM_LSTAR][1]);
        ok = ok && (phistar >= pc.lims[LF_LIM_PHISTAR][0]) && (phistar <= pc.lims[LF_LIM_PHISTAR][1]);
        ok = ok && (al >= pc.lims[LF_LIM_SCH_AL][0]) && (al <= pc.lims[LF_LIM_SCH_AL][1]);
        ok = ok && (alphaC >= pc.lims[LF_LIM_ALPHA][0]) && (alphaC <= pc.lims[LF_LIM_ALPHA][1]);
        const double c0 = LF_LNLN10 + LF_LN10 * phistar, c1 = LF_LN10 * (al + 1.0);
        const double Q = exp10(LF_LREF - Lstar);
        asm volatile("" : "+v"(const_cast<double&>(Q)));
        LF_TST(2);
        if (has_f && kf.nsrc > 0) {                 // (first thing: the field's sums are not held in registers any longer)
            const double n = (double)kf.nsrc;
            const double c0f = c0 + (variant == LF_FREE ? kf.lnom0 : 0.0);
            base = n * c0f + c1 * (kf.slc - n * (Lstar - LF_LREF)) - Q * kf.sp +
                   (variant == LF_FIXCOMP ? kf.som : 0.0);
        }
        if (live && f == 0) {
            double* d = TOLDS ? l_sc + grp * 8 : r;
            d[R_LSTAR] = Lstar;
            d[R_C0] = c0;
            d[R_C1] = c1;
            d[R_Q] = Q;
            d[R_ALPHAC] = alphaC;
        }
        if (has_f) {
            const double Flim = variant == LF_FREE ? th[k + f] : kf.flim0;
            ok = ok && (Flim >= pc.lims[LF_LIM_FLIM][0]) && (Flim <= pc.lims[LF_LIM_FLIM][1]);
            double lF = 0.0, V = 0.0;
            if (variant == LF_FREE) {
                const double b = -sqrt(pc.fc_ratio / (alphaC * alphaC));     // VmaxLumFunc.py:165
                lF = log10(1.0e-17 * Flim);
                V = 1.0 / (Flim * exp10(b));
                const double cA = -alphaC * lF, cY = -(lF + b);
                // Where the table-driven form of the term applies, as integer keys of log-flux x (conservatively
                // rounded): num = alpha_C x + cA inside the g table, y = x + cY inside the h table, both with
                // room for the margins; and from where on h = 1 (f / f_tau > 37.5: decay factor exactly 1.0).
                int klo = KEY_MAX, khi = -1, kne = KEY_MAX, kac = KEY_MAX;
                if (nqueue > 0 && alphaC > 0.0 && alphaC < 1.0e4) {      // (only lf_free reads the keys)
                    const double xlo = fmax((G_NUM_LO + 2.0 * G_MARGIN - cA) / alphaC, H_LO + 2.0 * H_MARGIN - cY);
                    const double xhi = (G_NUM_HI - 2.0 * G_MARGIN - cA) / alphaC;
                    const double xne = 1.5740312677277188 - cY;            // log10(37.5)
                    klo = key_ceil((xlo - pc.key_x0) * KEY_SCALE);
                    khi = key_floor((xhi - pc.key_x0) * KEY_SCALE);
                    kne = key_ceil((xne - pc.key_x0) * KEY_SCALE);
                    kac = key_ceil(alphaC * KEY_ASCALE);
                }
                // the field's cells can stand for its sources when all of them lie inside the tables for this walker
                // (their width was chosen for the prior box's largest alpha_C: lfmcmc.hip, build_cells)
                cell_ok = kf.nsrc == 0 || (klo <= kf.kf_first && kf.kf_last <= khi);
                if (live && TOLDS) {
                    double* d = l_fc + (grp * MAXF + f) * 8;
                    int* di = reinterpret_cast<int*>(d + 4);
                    d[0] = alphaC;
                    d[F_V] = V;
                    d[F_CA] = cA;
                    d[F_CY] = cY - H_LO - 0.5 / H_INV;      // (the h table's index arithmetic wants y - H_LO, and that minus half a piece)
                    d[7] = cY - H_LO;
                    di[M_KLO] = klo;
                    di[M_KHI] = khi;
                    di[M_KNE] = kne;
                    di[M_KAC] = kac;
                    l_lf[grp * MAXF + f] = lF;
                } else if (live) {
                    r[RF(f, F_LF)] = lF;
                    r[RF(f, F_V)] = V;
                    r[RF(f, F_CA)] = cA;
                    r[RF(f, F_CY)] = cY;
                    int* km = wmode + ((size_t)w * MAXF + f) * WM;
                    km[M_KLO] = klo;
                    km[M_KHI] = khi;
                    km[M_KNE] = kne;
                    km[M_KAC] = kac;
                }
            }
            LF_TST(3);
            if (kf.nsrc > 0) {
                const double vmax = kf.pmax * Q;       // the very product the kernels form for that source
                const double tlo = kf.lum_min - Lstar, thi = kf.lum_max - Lstar;
                const double lbT = c0 + fmin(c1 * tlo, c1 * thi) - vmax;
                if (vmax > LF_UNDERFLOW) {
                    m = MODE_NEGINF;
                    neginf = 1;
                } else if (variant == LF_FREE) {
                    // A LOWER bound of ln Omega of the field's faintest source is all the test needs, and the threshold is
                    // -700: elementary inequalities and single-precision hardware log instead of the device library's log, rsqrt
                    // and exp (this chain, on one wave, was a good part of the 7 us a tile's preparation took inside lf_free):
                    //   fc = 1 / (2 s (s - num)), s = sqrt(1 + num^2) <= 1 + |num|   =>   ln fc >= -2 ln(2 (1 + |num|))   (num < 0)
                    //   fc >= 1/2                                                                                        (num >= 0)
                    //   1 - e^-x >= x / (1 + x)                                   =>   ln fc / (1 - e^-x) >= ln fc (1 + 1 / x)
                    // (a NaN anywhere fails the comparisons: careful path.)  Looser than the exact value by at most ln 2 and a
                    // factor 1.3 - immaterial 700 e-folds away, and erring towards the careful path.
                    const double num = alphaC * (kf.a_min - lF);
                    const float an = (float)fmax(-num, 0.0);
                    const float lnfc_lo = num >= 0.0 ? -0.6932f
                                                     : -2.00002f * 0.69314724f * __builtin_amdgcn_logf(2.0f * (1.0f + an) * 1.000001f) - 0.01f;
                    const float x = (float)(kf.u_min * V) * 0.999999f;
                    const double lnOm = kf.lnom0 + (double)(lnfc_lo * (1.0f + 1.000001f / x));
                    // fexp_neg takes |x| < 2^24 unclamped: screen the largest f / f_tau of the field as well
                    m = (lbT > SAFE && lnOm > SAFE && lbT + lnOm > SAFE && kf.u_max * V < 1.0e6) ? MODE_FAST : MODE_SLOW;
                } else {
                    m = (lbT > SAFE && lbT + kf.a_min > SAFE) ? MODE_FAST : MODE_SLOW;
                }
            }
        }
    }
    LF_TST(4);
    base = group8_sum(base);
    const int bad = group8_or(ok ? 0 : 1);
    neginf = group8_or(neginf);
    // a walker outside the prior, or already known to be -inf, gets lnprob = -inf whatever its sums are:
    // it must not drag its tile onto the careful path (stretch-move proposals leave the box often)
    if (bad) m = MODE_SKIP;
    else if (neginf) m = MODE_SKIPSRC;
    const int slow = group8_or(has_f && m == MODE_SLOW ? 1 : 0);
    // cells: only walkers whose every field is FAST and inside the tables (all the others are rare, and summed per source)
    const int nocell = group8_or(has_f && !(cell_ok && (m == MODE_FAST || kf.nsrc == 0)) ? 1 : 0);
    const int cells = kc.cells && (variant == LF_FREE ? nqueue > 0 : variant == LF_ZEVOL) && !bad && !neginf && !nocell;
    if (live && has_f) {
        if (TOLDS) reinterpret_cast<int*>(l_fc + (grp * MAXF + f) * 8 + 4)[M_MODE] = m;
        else wmode[((size_t)w * MAXF + f) * WM + M_MODE] = m;
    }
    if (live && f == 0) {
        const int st = (bad ? 0 : STAT_PRIOR_OK) | (neginf ? STAT_NEGINF : 0) | (slow ? STAT_SLOW : 0) | (cells ? STAT_CELLS : 0);
        if (TOLDS) {
            l_base[grp] = base;
            l_stat[grp] = st;
        } else {
            wbase[w] = base;
            wstat[w] = st;
            if (slow && slow_list) slow_list[1 + atomicAdd(slow_list, 1)] = w;
        }
    }
}

__global__ __launch_bounds__(64) void lf_prepare(KConst kc, StepArgs sp, const double* __restrict__ theta, int B,
                                                 double* __restrict__ wrec, int* __restrict__ wstat,
                                                 int* __restrict__ wmode, double* __restrict__ wbase, int* __restrict__ slow_list,
                                                 int* __restrict__ queue, int nqueue) {
    warm_kernarg<sizeof(KConst) + sizeof(StepArgs) + 7 * 8 + 8>();      // (lf_math.h: the arguments in one round trip)
    __shared__ double sth[8][16];
    const int gt = blockIdx.x * blockDim.x + threadIdx.x;
    if (gt < nqueue) queue[gt] = 0;                  // item counters of the lf_free launch that follows (8 B threads >= 9 B / 8)
    prepare_lane<true>(kc, sp, theta, B, wrec, wstat, wmode, wbase, slow_list, nqueue, gt >> 3, gt & 7, (int)threadIdx.x >> 3, sth);
}

// ----------------------------------------------------------------------------------------------
// per-term arithmetic
// ----------------------------------------------------------------------------------------------
// ln of the Fleming completeness fc = 1/2 (1 + num / sqrt(1 + num^2)), VmaxLumFunc.py:118-120
__device__ __forceinline__ double ln_fc_careful(double num) {
    const double s = fma(num, num, 1.0);
    return dlog(0.5 * (1.0 + num * drsqrt(s)));
}
__device__ __forceinline__ double ln_fc_fast(double num, const MathTables* __restrict__ tab) {
    const double s = fma(num, num, 1.0);
    return flog_half(fma(num, frsqrt(s), 1.0), tab);
}

// a wave-uniform double that arrived in a VGPR (broadcast LDS read) -> SGPR pair: VALU instructions take it as a scalar
// operand and it stops occupying a vector register per lane
__device__ __forceinline__ double uni(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

struct WFree {   // wave-uniform walker constants of one (walker, field)
    double Lstar, c0f, c1, Q, alphaC, lF, V, lnom0, cA;
};

// completeness part of the term, ln(fc) / (1 - e^(-f/f_tau)); the Schechter part and ln Omega_0 are
// summed in closed form per walker (lf_prepare -> wbase)
__device__ __forceinline__ double term_free_fast(const WFree& w, double logf, double U,
                                                 const MathTables* __restrict__ tab) {
    // ln(fc) / d with fc = (1 + num / sqrt(s)) / 2, s = 1 + num^2, d = 1 - e^(-U V)  (VmaxLumFunc.py:118-127, :141)
    // needs 1/sqrt(s) and 1/d: ONE v_rsq_f64 seed serves both, Z = rsqrt(s d^2) = 1 / (sqrt(s) d), then
    // 1/sqrt(s) = Z d and 1/d = Z^2 s d  (a quarter-rate seed costs four FMA slots)
    const double num = fma(w.alphaC, logf, w.cA);            // alpha_C (logf - lF), one FMA
    const double s = fma(num, num, 1.0);
    const double d = 1.0 - fexp_neg(U * w.V, tab);
    const double sd = s * d;
    const double Z = frsqrt(sd * d);
    const double lnfc = flog_half(fma(num, Z * d, 1.0), tab);
    return lnfc * ((Z * sd) * Z);
}

// The catalogue is sorted by flux inside each field (lfmcmc.hip: build), so a chunk's first source is its
// faintest, and one wave-uniform fact about a (walker, chunk) pair selects a cheaper form of the same term:
//   U_min V > 37.5 = 54.1 ln 2  =>  e^(-f/f_tau) < 2^-54 for every source of the chunk, so 1 - e^(-f/f_tau) is
//   exactly 1.0 in binary64 - the value the general form computes too - and the term is ln(fc): no exp, no 1/d.
// Such sources are also above the walker's 50 % flux (alpha_C (logf - lF) >= 0 is tested with it), fc in [1/2, 1],
// and the log needs no exponent handling (flog_half_upper).  (A third form - exponent-free log, exponential kept -
// raised the kernel to 145 VGPRs and 3 waves/SIMD: dropped.)
__device__ __forceinline__ double term_free_noexp(const WFree& w, double logf, const MathTables* __restrict__ tab) {
    const double num = fma(w.alphaC, logf, w.cA);
    const double s = fma(num, num, 1.0);
    return flog_half_upper(fma(num, frsqrt(s), 1.0), tab);
}

__device__ __forceinline__ double term_free_careful(const WFree& w, double lum, double logf, double P, double U) {
    const double NEG_INF = -__builtin_huge_val();
    const double v = P * w.Q;
    const double lnT = fma(w.c1, lum - w.Lstar, w.c0f) - v;
    const double lnfc = ln_fc_careful(w.alphaC * (logf - w.lF));
    const double d = 1.0 - dexp(-(U * w.V));
    const double lnOm = w.lnom0 + ddiv(lnfc, d);
    const double term = (lnT - w.lnom0) + lnOm;
    const bool bad = (v > LF_UNDERFLOW) | (lnT - w.lnom0 < -LF_UNDERFLOW) | (lnOm < -LF_UNDERFLOW) |
                     (term < -LF_UNDERFLOW) | (term != term);
    return bad ? NEG_INF : lnOm - w.lnom0;      // the rest of the term is in wbase
}

struct WZ {
    double aL, bL, cL, aP, bP, cP, c1;
    double mslope;     // ln10 * (max |dL*/dz| over the field's redshifts + |aL| / 128): bounds the exponent of the local form
};

template <bool FAST>
__device__ __forceinline__ double lnT_zevol(const WZ& w, double lum, double z, double z2, double& v,
                                            const MathTables* __restrict__ tab) {
    const double Ls = quad_nofma(w.aL, w.bL, w.cL, z, z2);        // lumfuncmcmc_z.py:66
    const double ph = quad_nofma(w.aP, w.bP, w.cP, z, z2);        // :65
    const double t = lum - Ls;
    v = FAST ? fexp_c(LF_LN10 * t, tab) : dexp(LF_LN10 * t);
    return fma(w.c1, t, fma(LF_LN10, ph, LF_LNLN10)) - v;
}

// ----------------------------------------------------------------------------------------------
// block reduction: red[nw][256] (LDS) -> out[(w0 + w) * stride + chunk], nw <= 16.
// All walkers at once: thread t = (walker t >> 4, column group t & 15) adds its 16 columns (stride 16: consecutive
// lanes read consecutive doubles), then the 16 lanes of a walker combine with four DPP row shifts inside their
// 16-lane row.  Fixed order: the bits depend on the launch geometry only.  (The first version gave each wave four
// walkers in turn, a 64-lane shuffle tree each: 3.3k cycles of dependent latency at the end of every workgroup,
// 8 % of its lifetime, measured with tools/stamps.py.)
// ----------------------------------------------------------------------------------------------
// widx: optional list of walker indices (the tile is widx[0 .. nw-1] instead of w0 .. w0+nw-1)
__device__ __forceinline__ void reduce_store(const double* __restrict__ red, int nw, double* __restrict__ out,
                                             size_t stride, int w0, int chunk, const int* __restrict__ widx = nullptr) {
    static_assert(BLOCK == 256, "16 walkers x 16 column groups");
    for (int wb = 0; wb < nw; wb += 16) {            // (tiles are at most 16 walkers: one pass)
        const int w = wb + (threadIdx.x >> 4), j = threadIdx.x & 15;
        const double* row = red + (w < nw ? w : 0) * BLOCK + j;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            s0 += row[16 * i];
            s1 += row[16 * (i + 1)];
            s2 += row[16 * (i + 2)];
            s3 += row[16 * (i + 3)];
        }
        double sum = (s0 + s1) + (s2 + s3);
        // the 16 lanes of a walker are one DPP row: row_shr 1, 2, 4, 8 leave the row's sum in its lane 15
        sum = dpp_shift_add<0x111, 0xf>(sum);
        sum = dpp_shift_add<0x112, 0xf>(sum);
        sum = dpp_shift_add<0x114, 0xf>(sum);
        sum = dpp_shift_add<0x118, 0xf>(sum);
        if (j == 15 && w < nw) out[(size_t)(widx ? widx[w] : w0 + w) * stride + chunk] = sum;
    }
}

// ----------------------------------------------------------------------------------------------
// piece A: per-source log-term sum.  grid = (chunks, walker tiles); chunk = up to 256*ST sources of
// ONE field.  lumfuncmcmc.py:370 (FREE), :388 (FIXCOMP), lumfuncmcmc_z.py:371 (ZEVOL)
// ----------------------------------------------------------------------------------------------
struct SrcArrays {
    const double* lum;    // [N]
    const double* a1;     // FREE: logf      FIXCOMP: ln(Om_arr)   ZEVOL: z
    const double* P;      // FREE/FIXCOMP: 10^(lum-42)             ZEVOL: ln(Om_arr)
    const double* U;      // FREE: 10^(logf+17)                    ZEVOL: z^2
    const double* W;      // compressed catalogue only: weight of the pseudo-source
    const int* chunk_start;
    const int* chunk_len;
    const int* chunk_field;
    const int* chunk_keys;   // FREE, real catalogue: KEY_STRIDE ints per chunk {kfirst, klast, kamax, -, kamax of waves 0..7}
                             // (lfmcmc.hip: get_chunks), else NULL
    int* queue;              // FREE, real catalogue: the eight per-XCD item counters of the persistent workgroups
};

// CMP = the items are the pseudo-sources of the compressed catalogue (lfmcmc.hip: build_compressed): each
// carries a weight, and walkers that need the per-source underflow checks are left out (they are summed over
// the real catalogue by the rescue workgroups of the same launch).
// IDX = the walker tile is a list, widx[0 .. nw-1] (the rescue workgroups' tiles of flagged walkers), instead of
// the contiguous w0 .. w0+nw-1.
template <int VARIANT, int ST, int TW, bool CMP, bool IDX = false>
__device__ __forceinline__ void srcsum_body(const KConst& kc, const SrcArrays& sa, const double* __restrict__ wrec,
                                            const int* __restrict__ wmode, int c, int w0, int nw,
                                            double* __restrict__ partial, int pstride,
                                            const MathTables& tab, double* __restrict__ red,
                                            const int* __restrict__ widx = nullptr) {
    auto wi = [&](int w) -> size_t { return (size_t)(IDX ? widx[w] : w0 + w); };
    // one item: catalogue chunk c x walkers w0 .. w0+nw-1 (nw <= TW, which sizes the LDS buffer)
    const int tid = threadIdx.x;
    const int s0 = sa.chunk_start[c], n = sa.chunk_len[c], fld = sa.chunk_field[c];
    // FREE: sources are sorted by flux inside a field, the chunk's first is its faintest (NaN fluxes sort last and
    // put the field on the careful path anyway)
    const double a1_first = VARIANT == LF_FREE ? sa.a1[s0] : 0.0, u_first = VARIANT == LF_FREE ? sa.U[s0] : 0.0;

    if (VARIANT == LF_FIXCOMP) {
        // piece A is closed-form (wbase): unless one of the tile's walkers needs the per-term underflow
        // checks there is nothing to do here - do not even read the catalogue
        int any_slow = 0;
        for (int w = 0; w < nw; ++w) any_slow |= (wmode[(wi(w) * MAXF + fld) * WM] == MODE_SLOW);
        if (!__builtin_amdgcn_readfirstlane(any_slow)) {
            if (tid < nw) partial[wi(tid) * pstride + c] = 0.0;
            return;
        }
    }
    // ZEVOL on the real catalogue: a lane holds ST NEIGHBOURS in redshift (the catalogue is sorted by z inside a field),
    // not every 256th source - see the local form of the term below.  Any assignment of sources to lanes gives the same
    // sums up to rounding; the careful loop further down re-reads by index and covers every source once either way.
    constexpr bool ZLOCAL = VARIANT == LF_ZEVOL && !CMP;
    // items -> registers (lanes past the end replay the chunk's first source with weight 0)
    double lum[ST], a1[ST], pp[ST], uu[ST], wgt[ST];
#pragma unroll
    for (int k = 0; k < ST; ++k) {
        const int i = ZLOCAL ? tid * ST + k : k * BLOCK + tid;
        const size_t g = (size_t)s0 + (i < n ? i : 0);
        wgt[k] = i < n ? (CMP ? sa.W[g] : 1.0) : 0.0;
        lum[k] = sa.lum[g];
        a1[k] = sa.a1[g];
        pp[k] = ZLOCAL ? 0.0 : sa.P[g];
        uu[k] = VARIANT == LF_FIXCOMP ? 0.0 : sa.U[g];
        if (VARIANT == LF_FREE && i >= n) {
            // padding lanes of the FAST loop: a source so bright that fc = 1 and the decay is 1, i.e.
            // ln(fc)/decay = 0 to within 2e-16 - cheaper than a weight multiply on every term
            a1[k] = 1.0e30;
            uu[k] = 1.0e4;
        }
    }
    // ZEVOL, local form: with z_c the lane's first source and d_k = z_k - z_c (L* is a quadratic: the expansion is exact),
    //     10^(lum_k - L*(z_k)) = 10^(42 - L*(z_c)) * 10^(lum_k - 42) * exp(e_k),   e_k = -ln10 (L*'(z_c) + aL d_k) d_k
    // - ONE exponential per (walker, lane) instead of one per (walker, source): the lane's sources are neighbours in z,
    // so e_k is tiny and exp(e_k) is a degree-5 polynomial (1e-16 for |e_k| <= 6e-3).  10^(lum_k - 42) does not depend
    // on the walker and is made here, once per item (in the registers of ln Om, which this path does not read).
    // Measured against the per-source form on 4e5 sources, walkers over the whole prior box, three pivot sets: 2-3e-16
    // on piece A, the same 6e-16 .. 8e-15 from the oracle as the per-source form.  Whether |e_k| <= 6e-3 holds for a
    // (walker, chunk) pair is one compare: the walker's bound on ln10 |dL*/dz| over the field (lf_prepare) times the
    // chunk's widest lane, which get_chunks folded into the chunk's third key (kamax = floor(2^16 G_MARGIN / width),
    // 0 if the width exceeds 1 / 128 or is unknown).  Pairs that fail take the per-source exponential as before.
    double zc = 0.0, zwidth_inv = 0.0;
    if (ZLOCAL) {
        zc = a1[0];          // (the lane's first source: real whenever any of the lane's sources is - a ragged chunk's pads replay
                             // the chunk's first source, which is not a neighbour)
#pragma unroll
        for (int k = 0; k < ST; ++k) pp[k] = wgt[k] * fexp_t(LF_LN10 * (lum[k] - LF_LREF), &tab);
        const int kamax = sa.chunk_keys ? __builtin_amdgcn_readfirstlane(sa.chunk_keys[KEY_STRIDE * c + 2]) : 0;
        zwidth_inv = (double)kamax * (1.0 / (G_MARGIN * KEY_ASCALE));      // <= 1 / (widest lane); 0: unknown
    }
    const double NEG_INF = -__builtin_huge_val();
    // walker constants (and the walker's mode) of the NEXT walker are fetched with scalar loads while the
    // current one computes.  The mode branch is per walker and wave-uniform: a walker's sum never
    // depends on which other walkers share its tile (lnprob is a function of its theta row alone).
    double nxA = 0.0, nxC = 0.0, nxV = 0.0;
    WZ nz{};
    int nxm = wmode[(wi(0) * MAXF + fld) * WM];
    {
        const double* __restrict__ r0 = wrec + wi(0) * REC;
        if (VARIANT == LF_FREE) {
            nxA = r0[R_ALPHAC];
            nxC = r0[RF(fld, F_CA)];
            nxV = r0[RF(fld, F_V)];
        }
        if (VARIANT == LF_ZEVOL) nz = WZ{r0[Z_AL], r0[Z_BL], r0[Z_CL], r0[Z_AP], r0[Z_BP], r0[Z_CP], r0[Z_C1], r0[RF(fld, 0)]};
    }
#pragma unroll 1
    for (int w = 0; w < nw; ++w) {
        const int mode = __builtin_amdgcn_readfirstlane(nxm);
        const double* __restrict__ rn = wrec + wi(min(w + 1, nw - 1)) * REC;
        nxm = wmode[(wi(min(w + 1, nw - 1)) * MAXF + fld) * WM];
        double acc = 0.0;
        if (mode != MODE_SLOW) {
            if (VARIANT == LF_FREE) {
                WFree wf{};
                wf.alphaC = nxA;
                wf.cA = nxC;
                wf.V = nxV;
                nxA = rn[R_ALPHAC];
                nxC = rn[RF(fld, F_CA)];
                nxV = rn[RF(fld, F_V)];
                // chunk-level facts (the first source of a chunk is its faintest), see term_free_noexp
                const bool upper = kc.specialise && wf.alphaC > 0.0 && fma(wf.alphaC, a1_first, wf.cA) >= 0.0;
                const bool noexp = upper && u_first * wf.V > 37.5;
                if (mode >= MODE_SKIP) {
                    // -inf already: nothing to sum
                } else if (noexp) {
                    if (!CMP && !IDX) asm volatile("; LF_BEGIN general_noexp items=%0" ::"n"(ST));
#pragma unroll
                    for (int k = 0; k < ST; ++k) {
                        const double term = term_free_noexp(wf, a1[k], &tab);
                        acc = CMP ? fma(term, wgt[k], acc) : acc + term;
                    }
                    if (!CMP && !IDX) asm volatile("; LF_END general_noexp");
                } else {
                    if (!CMP && !IDX) asm volatile("; LF_BEGIN general items=%0" ::"n"(ST));
#pragma unroll
                    for (int k = 0; k < ST; ++k) {
                        const double term = term_free_fast(wf, a1[k], uu[k], &tab);
                        acc = CMP ? fma(term, wgt[k], acc) : acc + term;
                    }
                    if (!CMP && !IDX) asm volatile("; LF_END general");
                }
            } else if (VARIANT == LF_FIXCOMP) {
                // nothing left per source: piece A is the closed form in wbase
            } else {
                const WZ wz = nz;
                nz = WZ{rn[Z_AL], rn[Z_BL], rn[Z_CL], rn[Z_AP], rn[Z_BP], rn[Z_CP], rn[Z_C1], rn[RF(fld, 0)]};
                if (mode < MODE_SKIP) {
                const bool zlocal = ZLOCAL && kc.specialise && wz.mslope <= 6.0e-3 * zwidth_inv;      // (wave-uniform)
                // (census, ZEVOL: "table" counts the terms of the local form, "general" the per-source exponentials)
                if (!CMP && kc.forms && tid == 0) atomicAdd(kc.forms + (zlocal ? FORM_TABLE : FORM_GENERAL), (unsigned long long)n);
                if (zlocal) {
                    asm volatile("; LF_BEGIN zevol items=%0" ::"n"(ST));
                    const double Lc = quad_nofma(wz.aL, wz.bL, wz.cL, zc, uu[0]);
                    const double Hc = fexp_t(LF_LN10 * (LF_LREF - Lc), &tab);
                    const double a2 = -LF_LN10 * wz.aL;
                    const double s1 = fma(2.0 * a2, zc, -LF_LN10 * wz.bL);          // -ln10 L*'(z_c)
                    double sum4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int k = 0; k < ST; ++k) {
                        const double dz = a1[k] - zc;
                        const double e = fma(a2, dz, s1) * dz;
                        double p = fma(e, 1.0 / 120.0, 1.0 / 24.0);
                        p = fma(p, e, 1.0 / 6.0);
                        p = fma(p, e, 0.5);
                        p = fma(p, e, 1.0);
                        p = fma(p, e, 1.0);
                        sum4[k & 3] = k < 4 ? pp[k] * p : fma(pp[k], p, sum4[k & 3]);
                    }
                    acc = -Hc * ((sum4[0] + sum4[1]) + (sum4[2] + sum4[3]));      // everything else of the term is in wbase
                    asm volatile("; LF_END zevol");
                } else {
                if (!CMP) asm volatile("; LF_BEGIN zevol_direct items=%0" ::"n"(ST));
#pragma unroll
                for (int k = 0; k < ST; ++k) {
                    const double Ls = CMP ? quad_comp(wz.aL, wz.bL, wz.cL, a1[k])
                                          : quad_nofma(wz.aL, wz.bL, wz.cL, a1[k], uu[k]);   // L*(z_i), lumfuncmcmc_z.py:66
                    // 10^(lum_i - L*(z_i)): FAST mode has bounded it by 700 from above (lf_prepare: vb), and far below
                    // fexp_t underflows to 0 by itself - no clamps
                    const double v = fexp_t(LF_LN10 * (lum[k] - Ls), &tab);
                    acc = fma(-v, wgt[k], acc);             // everything else of the term is in wbase
                }
                if (!CMP) asm volatile("; LF_END zevol_direct");
                }
                }
            }
        } else if (CMP) {
            // left to the rescue workgroups (real catalogue); only keep the prefetch chain going
            if (VARIANT == LF_FREE) {
                nxA = rn[R_ALPHAC];
                nxC = rn[RF(fld, F_CA)];
                nxV = rn[RF(fld, F_V)];
            }
            if (VARIANT == LF_ZEVOL) nz = WZ{rn[Z_AL], rn[Z_BL], rn[Z_CL], rn[Z_AP], rn[Z_BP], rn[Z_CP], rn[Z_C1], rn[RF(fld, 0)]};
        } else {
            // careful path (rare): device-library math, per-term underflow checks, -inf poisoning.  Items
            // are re-read from memory inside a rolled loop so that this path adds no register pressure
            // to the fast one.
            const double* __restrict__ r = wrec + wi(w) * REC;
            if (VARIANT == LF_ZEVOL && !CMP && kc.forms && tid == 0) atomicAdd(kc.forms + FORM_CAREFUL, (unsigned long long)n);
            if (VARIANT == LF_FREE) {
                nxA = rn[R_ALPHAC];
                nxC = rn[RF(fld, F_CA)];
                nxV = rn[RF(fld, F_V)];
            }
            if (VARIANT == LF_ZEVOL) nz = WZ{rn[Z_AL], rn[Z_BL], rn[Z_CL], rn[Z_AP], rn[Z_BP], rn[Z_CP], rn[Z_C1], rn[RF(fld, 0)]};
#pragma unroll 1
            for (int k = 0; k < ST; ++k) {
                const int i = k * BLOCK + tid;
                if (i >= n) break;
                const size_t g = (size_t)s0 + i;
                const double clum = sa.lum[g], ca1 = sa.a1[g], cpp = sa.P[g];
                double term;
                if (VARIANT == LF_FREE) {
                    const WFree wf{r[R_LSTAR], r[R_C0] + kc.lnom0_src[fld], r[R_C1], r[R_Q], r[R_ALPHAC],
                                   r[RF(fld, F_LF)], r[RF(fld, F_V)], kc.lnom0_src[fld], 0.0};
                    term = term_free_careful(wf, clum, ca1, cpp, sa.U[g]);
                } else if (VARIANT == LF_FIXCOMP) {
                    const double v = cpp * r[R_Q];
                    const double lnT = fma(r[R_C1], clum - r[R_LSTAR], r[R_C0]) - v;
                    term = lnT + ca1;
                    const bool bad = (v > LF_UNDERFLOW) | (lnT < -LF_UNDERFLOW) | (term < -LF_UNDERFLOW) | (term != term);
                    term = bad ? NEG_INF : 0.0;             // the value itself is in wbase
                } else {
                    const WZ wz{r[Z_AL], r[Z_BL], r[Z_CL], r[Z_AP], r[Z_BP], r[Z_CP], r[Z_C1], 0.0};
                    double v;
                    const double lnT = lnT_zevol<false>(wz, clum, ca1, sa.U[g], v, &tab);
                    term = lnT + cpp;
                    const bool bad = (v > LF_UNDERFLOW) | (lnT < -LF_UNDERFLOW) | (term < -LF_UNDERFLOW) | (term != term);
                    term = bad ? NEG_INF : -v;              // the rest of the term is in wbase
                }
                acc += term;
            }
        }
        red[w * BLOCK + tid] = acc;
    }
    __syncthreads();
    reduce_store(red, nw, partial, (size_t)pstride, w0, c, IDX ? widx : nullptr);
}

// ----------------------------------------------------------------------------------------------
// piece A, FREE variant, real catalogue: the table-driven form of the term (used by lf_free.h).
//
// After the hoisting of the header comment the term of (walker w, source i of field f) is
//     ln fc(num) / (1 - e^(-f/f_tau)) = g(num_i) h(y_i),   num_i = alpha_C x_i + cA_wf,   y_i = x_i + cY_wf,   x_i = logf_i
// with g and h UNIVARIATE (lf_tables.h: piecewise degree-7 polynomials, generated and error-checked offline over their
// whole domain: 8e-15 / 3e-15 relative).  The catalogue is sorted by flux inside a field and a lane holds ST
// NEIGHBOURS of that order, so all its sources fall on one piece of each table, or just across its end - every
// piece is fitted G_MARGIN / H_MARGIN beyond its ends for that.  Per (walker, lane): one piece index from the
// lane's middle source, 2 x 4 ds_read_b128 for the coefficients; per term: two FMAs for the local coordinates and
// two 7-FMA Horner chains (17 fp64 instructions, 16 of them FMAs) instead of rsqrt + log + exp + reciprocal.
// Whether a (walker, chunk) pair may take it is decided with scalar integer compares on keys that lf_prepare
// (walker, field) and get_chunks (chunk) have rounded conservatively: the chunk's fluxes inside both tables with
// room for the margins, and alpha_C times the widest lane of the chunk within the margin.  Pairs that may not
// (the sparse bright and faint tails of a field, tiny catalogues, extreme walkers) take the general form.
//
// ----------------------------------------------------------------------------------------------
struct TabCoef {       // one lane's pieces of g and h for one walker, and the affine maps into their local coordinates
    double cg[8], ch[8];
    double sa, sc, dy;
};
struct WalkerK {       // wave-uniform constants of one (walker, field), read from the LDS copy
    double aC, cA;          // num = aC x + cA
    double cYs, cYH;        // y - H_LO = x + cYH;  cYs = cYH - 1 / (2 H_INV): round(H_INV (x + cYs)) = floor(H_INV (y - H_LO))
    int mode, klo, khi, kne, kac;
};

template <int ST>
__device__ __forceinline__ void table_lookup(TabCoef& C, const double (&x)[ST], const WalkerK& p, bool noexp,
                                             const TermTables* __restrict__ tt) {
    const double xc = x[ST / 2];
    // g: binade of v = |num| + 1 and its top G_BITS mantissa bits.  The sign of num selects the half of the table and
    // flips the affine map - by bit operations on the high words, no compare / select pairs.
    const double numc = fma(p.aC, xc, p.cA);
    const int nh = __double2hiint(numc);
    const int sgn = nh & (int)0x80000000u;
    const double v = fabs(numc) + 1.0;
    const int hi = __double2hiint(v) & (int)(0xffffffffu << (20 - G_BITS));
    const double vlo = __hiloint2double(hi, 0);
    int pg = (hi >> (20 - G_BITS)) + ((nh >> 31) & G_NPOS) - (0x3ff << G_BITS);
    pg = min(max(pg, 0), G_N - 1);
    C.sa = __hiloint2double(__double2hiint(p.aC) ^ sgn, __double2loint(p.aC));            // t_i = |num_i| + 1 - v_lo = sa x_i + sc
    C.sc = __hiloint2double(__double2hiint(p.cA) ^ sgn, __double2loint(p.cA)) + (1.0 - vlo);
    const double2* __restrict__ g2 = reinterpret_cast<const double2*>(tt->g) + pg * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const double2 t = g2[q];
        C.cg[2 * q] = t.x;
        C.cg[2 * q + 1] = t.y;
    }
    C.dy = 0.0;
    if (!noexp) {                                             // (wave-uniform)
        // h: piece floor(H_INV (y - H_LO)) by magic-number rounding (the index is the low word of the sum: no
        // conversions); a tie goes to either neighbour, both valid there (H_MARGIN).  The clamp only matters for
        // values the key tests never let through (NaN).
        constexpr double MAGIC = 6755399441055744.0;          // 1.5 * 2^52
        const double tq = fma(xc + p.cYs, (double)H_INV, MAGIC);
        const int ph = min(max(__double2loint(tq), 0), H_N - 1);
        C.dy = fma(tq - MAGIC, -1.0 / H_INV, p.cYH);          // t_i = y_i - y_lo = x_i + dy
        const double2* __restrict__ h2 = reinterpret_cast<const double2*>(tt->h) + ph * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double2 t = h2[q];
            C.ch[2 * q] = t.x;
            C.ch[2 * q + 1] = t.y;
        }
    }
}

// Sum of the lane's ST terms.  Horner steps go coefficient by coefficient ACROSS the ST sources: 2 ST independent FMA
// chains in flight (a source's own chain is 7 dependent FMAs).  npad = slots of this lane past the end of the chunk:
// they hold copies of the chunk's last source, so they all evaluate to the value of slot ST - 1, which is taken
// out again (exact for whole lanes: 8 t - 8 t; one rounding otherwise).
template <int ST, bool NOEXP>
__device__ __forceinline__ double table_terms(const TabCoef& C, const double (&x)[ST], int npad) {
    // Sources in batches of CH: g and h of a batch together, 2 CH independent FMA chains in flight (a source's own
    // chain is 7 dependent FMAs).  CH = 4: with the 4 waves per SIMD the kernel is sized for (<= 128 VGPRs) a wave
    // issues every fourth slot, so two steps of one chain are ~100 cycles apart anyway; all ST sources at once would
    // cost 24 more VGPRs.
    constexpr int CH = ST < 4 ? ST : 4;
    double acc4[4] = {0.0, 0.0, 0.0, 0.0};
    double last = 0.0;
#pragma unroll
    for (int k0 = 0; k0 < ST; k0 += CH) {
        double t[CH], u[CH], p[CH], q[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            t[k] = fma(C.sa, x[k0 + k], C.sc);
            if (!NOEXP) u[k] = x[k0 + k] + C.dy;
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            p[k] = fma(C.cg[7], t[k], C.cg[6]);
            if (!NOEXP) q[k] = fma(C.ch[7], u[k], C.ch[6]);
        }
#pragma unroll
        for (int j = 5; j >= 0; --j) {
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                p[k] = fma(p[k], t[k], C.cg[j]);
                if (!NOEXP) q[k] = fma(q[k], u[k], C.ch[j]);
            }
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const bool first = k0 + k < 4;        // (the accumulator's first term: no add of 0.0, which the compiler keeps)
            if (NOEXP) {
                acc4[k & 3] = first ? p[k] : acc4[k & 3] + p[k];
                if (k0 + k == ST - 1) last = p[k];
            } else if (k0 + k == ST - 1) {
                last = p[k] * q[k];
                acc4[k & 3] = first ? last : acc4[k & 3] + last;
            } else {
                acc4[k & 3] = first ? p[k] * q[k] : fma(p[k], q[k], acc4[k & 3]);
            }
        }
        if (k0 + CH < ST) __builtin_amdgcn_sched_barrier(0);      // keep the batches apart (registers)
    }
    const double sum = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
    return fma(-(double)npad, last, sum);
}

// One cell for one walker: sum over the cell's sources of g(num) h(y) = sum_j c_j S_j (see CELL_M above).  cd = {x_c,
// S_0 .. S_8}.  Per (walker, cell): the piece lookup at x_c, the Taylor shifts of both piece polynomials to the cell's
// midpoint (28 FMAs each), the powers of the slope, the product series up to order CELL_M (the polynomials have no
// order above 7) and the dot product with the power sums: ~150 fp64 instructions whatever the number of sources in
// the cell.
__device__ __forceinline__ double cell_sum(const double (&cd)[CELL_REC], const WalkerK& p, const TermTables* __restrict__ tt) {
    static_assert(CELL_M >= 7 && CELL_M <= 14, "the product of two degree-7 polynomials");
    TabCoef C;
    const double xs[1] = {cd[0]};
    table_lookup<1>(C, xs, p, false, tt);          // (bright cells land on the h table's last piece, the constant 1)
    const double tc = fma(C.sa, cd[0], C.sc), uc = cd[0] + C.dy;
#pragma unroll
    for (int j = 0; j <= 6; ++j) {
#pragma unroll
        for (int i = 6; i >= j; --i) {
            C.cg[i] = fma(tc, C.cg[i + 1], C.cg[i]);
            C.ch[i] = fma(uc, C.ch[i + 1], C.ch[i]);
        }
    }
    // g in powers of d = x - x_c: (sa d)^j
    double pw = C.sa;
#pragma unroll
    for (int j = 1; j <= 7; ++j) {
        C.cg[j] *= pw;
        if (j < 7) pw *= C.sa;
    }
    double acc = 0.0;
#pragma unroll
    for (int j = CELL_M; j >= 0; --j) {             // (small terms first)
        const int i0 = j > 7 ? j - 7 : 0, i1 = j < 7 ? j : 7;
        double cj = C.cg[i0] * C.ch[j - i0];
#pragma unroll
        for (int i = i0 + 1; i <= i1; ++i) cj = fma(C.cg[i], C.ch[j - i], cj);
        acc = fma(cj, cd[1 + j], acc);
    }
    return acc;
}

// ----------------------------------------------------------------------------------------------
// piece B: expected-count integral on the S x S grid.  grid = (node chunks of 256, walker tiles)
// trapz(trapz(I, logL, axis=0), zarr) = sum_jk W_jk I_jk with W from the actual grid spacings.
// lumfuncmcmc.py:373-377 (FREE), :389-392 (FIXCOMP), lumfuncmcmc_z.py:373-375 (ZEVOL)
// ----------------------------------------------------------------------------------------------
struct NodeArrays {
    const double* G;      // [S*S] logL
    const double* PG;     // 10^(G - 42)
    const double* W;      // FREE: trapz weight * volume_part[k]   else: trapz weight * sum_f integ_part[f]
    const double* a3;     // FREE: logf on the grid                ZEVOL: zarr[k]
    const double* a4;     // FREE: 10^(logf_grid + 17)             ZEVOL: zarr[k]^2
    const double* a4min;  // FREE: per chunk of 256 nodes, the smallest a4 (for the bright form, field_sum_bright)
    int nnodes;
};

// FREE: sum over the fields of Omega_0[f] fc^(1 / decay) at one node, for one walker.  NF is a template parameter
// so that the loop is straight-line code: the NF chains (rsqrt -> log -> exp -> rcp -> exp) are independent and
// the scheduler interleaves them, and all the walker constants are fetched by one batch of scalar loads.
template <int NF>
__device__ __forceinline__ double field_sum(const KConst& kc, const double* __restrict__ r, double alphaC, double a3,
                                            double a4, const MathTables* __restrict__ tab) {
    double lF[NF], V[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        lF[f] = uni(r[RF(f, F_CA)]);
        V[f] = uni(r[RF(f, F_V)]);
    }
    asm volatile("; LF_BEGIN node_general items=%0" ::"n"(NF));
    double s = 0.0;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const double num = fma(alphaC, a3, lF[f]);                    // alpha_C (logf - lF)
        const double sq = fma(num, num, 1.0);
        // arguments of both exponentials are <= 0: one-sided clamps
        const double d = fmax(1.0 - fexp_t(fmax(-(a4 * V[f]), -750.0), tab), 1e-100);   // (sd d and Z^2 sd stay finite)
        // one rsqrt seed for 1/sqrt(sq) and 1/d, as in term_free_fast: Z = 1 / (sqrt(sq) d)
        const double sd = sq * d;
        const double Z = frsqrt(sd * d);
        const double lnfc = flog_half(fma(num, Z * d, 1.0), tab);
        s = fma(kc.om0_grid[f], fexp_t(fmax(lnfc * ((Z * sd) * Z), -750.0), tab), s);   // fc ** (1 / fc_decay)
    }
    asm volatile("; LF_END node_general");
    return s;
}

// The same sum when every node of the workgroup has f / f_tau > 37.5 for every field of this walker (wave-uniform
// test on the workgroup's faintest node and the walker's smallest V): the decay factor is exactly 1.0 in binary64,
// so fc^(1/decay) = fc = (1 + num / sqrt(1 + num^2)) / 2 - no exp, no log, no reciprocal.  (The reference's
// fc ** 1.0 is fc itself: this form is the closer one.)
template <int NF>
__device__ __forceinline__ double field_sum_bright(const KConst& kc, const double* __restrict__ r, double alphaC, double a3) {
    asm volatile("; LF_BEGIN node_bright items=%0" ::"n"(NF));
    double s = 0.0;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const double num = fma(alphaC, a3, uni(r[RF(f, F_CA)]));
        const double w = fma(num, frsqrt(fma(num, num, 1.0)), 1.0);
        s = fma(kc.om0_grid[f], w, s);
    }
    asm volatile("; LF_END node_bright");
    return 0.5 * s;      // (a power of two: the same bits as scaling every weight, and no scaled weights to keep in registers)
}

__device__ __forceinline__ double field_sum_nf(const KConst& kc, const double* __restrict__ r, double alphaC, double a3,
                                               double a4, const MathTables* __restrict__ tab, bool bright = false) {
    if (bright) switch (kc.nf) {
        case 1: return field_sum_bright<1>(kc, r, alphaC, a3);
        case 2: return field_sum_bright<2>(kc, r, alphaC, a3);
        case 3: return field_sum_bright<3>(kc, r, alphaC, a3);
        case 4: return field_sum_bright<4>(kc, r, alphaC, a3);
        case 5: return field_sum_bright<5>(kc, r, alphaC, a3);
        case 6: return field_sum_bright<6>(kc, r, alphaC, a3);
        case 7: return field_sum_bright<7>(kc, r, alphaC, a3);
        default: return field_sum_bright<8>(kc, r, alphaC, a3);
    }
    switch (kc.nf) {
        case 1: return field_sum<1>(kc, r, alphaC, a3, a4, tab);
        case 2: return field_sum<2>(kc, r, alphaC, a3, a4, tab);
        case 3: return field_sum<3>(kc, r, alphaC, a3, a4, tab);
        case 4: return field_sum<4>(kc, r, alphaC, a3, a4, tab);
        case 5: return field_sum<5>(kc, r, alphaC, a3, a4, tab);
        case 6: return field_sum<6>(kc, r, alphaC, a3, a4, tab);
        case 7: return field_sum<7>(kc, r, alphaC, a3, a4, tab);
        default: return field_sum<8>(kc, r, alphaC, a3, a4, tab);
    }
}

// smallest V = 1 / f_tau over the walker's fields (wave-uniform)
__device__ __forceinline__ double walker_vmin(const KConst& kc, const double* __restrict__ r) {
    double v = r[RF(0, F_V)];
    for (int f = 1; f < kc.nf; ++f) v = fmin(v, r[RF(f, F_V)]);
    return uni(v);
}

template <int VARIANT, int TW>
__device__ __forceinline__ void gridsum_body(const KConst& kc, const NodeArrays& na, const double* __restrict__ wrec,
                                             const int* __restrict__ wmode, int B, int ntiles, int tw, int id, double* __restrict__ partial,
                                             int pstride, const MathTables& tab, double* __restrict__ red) {
    const int tid = threadIdx.x;
    const int c = id / ntiles, tile = id - c * ntiles;
    const int w0 = tile * tw;
    const int nw = min(tw, B - w0);
    const int gi = c * BLOCK + tid;
    // (source-sharded ranks split piece B by GRANULES of 64 nodes - g with g % parts == part - whichever kernel integrates
    // them: lf_free's and lf_pers's chunks are granules, a workgroup here holds four; a node of another rank's weighs 0)
    const bool valid = gi < na.nnodes && !(kc.grid_parts > 1 && (gi >> 6) % kc.grid_parts != kc.grid_part);
    const int g = gi < na.nnodes ? gi : 0;
    const double G = na.G[g], PG = na.PG[g], W = valid ? na.W[g] : 0.0;
    const double a3 = na.a3[g], a4 = na.a4[g];
    const double a4min = VARIANT == LF_FREE ? na.a4min[c] : 0.0;      // wave-uniform
    if (VARIANT == LF_ZEVOL && kc.zgrid_cols) {
        // The z-evolving integrand at node (j, k) is W exp(c1 (G - L*(z_k)) + ln10 phi*(z_k) + ln ln10 - 10^(G - L*(z_k))),
        // and everything of the walker in it depends on the COLUMN k only:
        //     10^(G - L*(z_k)) = PG * Q_wk,   Q_wk = 10^(42 - L*(z_k));      the rest = c1 (G - 42) + E_wk.
        // The nodes are stored column by column (lfmcmc.hip: build), so a chunk of 256 touches at most ZCOLS columns: a few
        // lanes make Q and E of the tile's walkers for them (one exponential each), and a node costs ONE exponential and no
        // parabola instead of two and two.
        __shared__ double qe[TW * ZCOLS * 2];
        const int S = kc.S, k0 = (c * BLOCK) / S;
        if (tid < nw * ZCOLS) {
            const int w = tid / ZCOLS, cc = tid - w * ZCOLS;
            const int k = min(k0 + cc, S - 1);
            const double* __restrict__ r = wrec + (size_t)(w0 + w) * REC;
            const double z = na.a3[(size_t)k * S], z2 = na.a4[(size_t)k * S];
            const double Ls = quad_nofma(r[Z_AL], r[Z_BL], r[Z_CL], z, z2);        // lumfuncmcmc_z.py:66
            const double ph = quad_nofma(r[Z_AP], r[Z_BP], r[Z_CP], z, z2);        // :65
            qe[tid * 2] = fexp_c(LF_LN10 * (LF_LREF - Ls), &tab);
            qe[tid * 2 + 1] = fma(-r[Z_C1], Ls - LF_LREF, fma(LF_LN10, ph, LF_LNLN10));
        }
        __syncthreads();
        const int cc = min(gi, na.nnodes - 1) / S - k0;      // this node's column among the chunk's (threads past the end: the last)
        const double Gm = G - LF_LREF;
        if (kc.forms && tid == 0) atomicAdd(kc.forms + FORM_NODE_GENERAL, (unsigned long long)min(BLOCK, na.nnodes - c * BLOCK) * nw);
#pragma unroll 1
        for (int w = 0; w < nw; ++w) {
            double val = 0.0;
            if (wmode[(size_t)(w0 + w) * MAXF * WM] != MODE_SKIP) {      // (outside the prior: not evaluated)
                const double c1 = uni(wrec[(size_t)(w0 + w) * REC + Z_C1]);
                const double Q = qe[(w * ZCOLS + cc) * 2], E = qe[(w * ZCOLS + cc) * 2 + 1];
                asm volatile("; LF_BEGIN znode items=1");
                val = W * fexp_c(fma(c1, Gm, E) - PG * Q, &tab);
                asm volatile("; LF_END znode");
            }
            red[w * BLOCK + tid] = val;
        }
        __syncthreads();
        reduce_store(red, nw, partial, (size_t)pstride, w0, c);
        return;
    }
#pragma unroll 1
    for (int w = 0; w < nw; ++w) {
        const double* __restrict__ r = wrec + (size_t)(w0 + w) * REC;
        double val;
        if (wmode[(size_t)(w0 + w) * MAXF * WM] == MODE_SKIP) {
            val = 0.0;                                  // outside the prior: not evaluated
        } else if (VARIANT == LF_FREE) {
            const double T = fexp_c(fma(uni(r[R_C1]), G - uni(r[R_LSTAR]), uni(r[R_C0])) - PG * uni(r[R_Q]), &tab);
            const double alphaC = uni(r[R_ALPHAC]);
            const bool bright = kc.specialise && alphaC > 0.0 && a4min * walker_vmin(kc, r) > 37.5;
            if (kc.forms && tid == 0)
                atomicAdd(kc.forms + (bright ? FORM_NODE_BRIGHT : FORM_NODE_GENERAL),
                          (unsigned long long)(min(BLOCK, na.nnodes - c * BLOCK) * kc.nf));
            const double s = field_sum_nf(kc, r, alphaC, a3, a4, &tab, bright);
            val = W * T * s;
        } else if (VARIANT == LF_FIXCOMP) {
            val = W * fexp_c(fma(r[R_C1], G - r[R_LSTAR], r[R_C0]) - PG * r[R_Q], &tab);
        } else {
            const WZ wz{r[Z_AL], r[Z_BL], r[Z_CL], r[Z_AP], r[Z_BP], r[Z_CP], r[Z_C1], 0.0};
            double v;
            if (kc.forms && tid == 0) atomicAdd(kc.forms + FORM_NODE_GENERAL, (unsigned long long)min(BLOCK, na.nnodes - c * BLOCK));
            asm volatile("; LF_BEGIN znode_rows items=1");
            val = W * fexp_c(lnT_zevol<true>(wz, G, a3, a4, v, &tab), &tab);
            asm volatile("; LF_END znode_rows");
        }
        red[w * BLOCK + tid] = val;
    }
    __syncthreads();
    reduce_store(red, nw, partial, (size_t)pstride, w0, c);
}

// ----------------------------------------------------------------------------------------------
// piece B on the COMPRESSED grid (FREE, separable grid; csrc/lf_compress.h: compress_grid).  The S^2 lattice
// points are replaced by 16 nodes per flux bin, shared by all rows:
//     B_w ~ sum_b sum_n [ sum_f om_f F_wf(u_bn) ] * [ sum_r T_w(L_{row0_b + r}) omega_b[r][n] ]
// A workgroup owns 16 bins (thread = (bin, node)) x a walker tile.  Per walker it first puts the S Schechter
// values T_w(L_j) in LDS (one exponential per thread), then every thread takes its node's field sum (the same
// field_sum as the full grid) times its short dot product with T_w.
// ----------------------------------------------------------------------------------------------
constexpr int GRIDC_MAX_S = 512;
struct GridC {
    const double* U;       // [nb * 16] node log-flux
    const double* A4;      // [nb * 16] 10^(U + 17)
    const int* row0;       // [nb]
    const int* nrows;      // [nb]
    const int* off;        // [nb] offset of omega_b
    const double* omega;   // per bin [row][node], trapezoid weights folded in
    const double* L;       // [S] luminosity nodes
    const double* PGL;     // [S] 10^(L - 42)
    int nb, S;
};

template <int TW>
__device__ __forceinline__ void gridc_body(const KConst& kc, const GridC& gc, const double* __restrict__ wrec,
                                           const int* __restrict__ wmode, int B,
                                           int ntiles, int tw, int id, double* __restrict__ partial, int pstride,
                                           const MathTables& tab, double* __restrict__ red, double* __restrict__ Tw) {
    const int tid = threadIdx.x;
    const int c = id / ntiles, tile = id - c * ntiles;
    const int w0 = tile * tw;
    const int nw = min(tw, B - w0);
    const int b = c * 16 + (tid >> 4), n = tid & 15;
    const bool valid = b < gc.nb;
    const int bb = valid ? b : 0;
    const double u = gc.U[bb * 16 + n], a4 = gc.A4[bb * 16 + n];
    // faintest node of the workgroup: bins ascend in flux, the nodes of a bin descend -> last node of the first bin
    const double a4min_g = gc.A4[min(c * 16, gc.nb - 1) * 16 + 15];
    const int j0 = gc.row0[bb], nr = valid ? gc.nrows[bb] : 0;
    const double* __restrict__ om = gc.omega + gc.off[bb] + n;
    double Lj[GRIDC_MAX_S / BLOCK], PGj[GRIDC_MAX_S / BLOCK];
#pragma unroll
    for (int q = 0; q < GRIDC_MAX_S / BLOCK; ++q) {
        const int j = min(tid + q * BLOCK, gc.S - 1);
        Lj[q] = gc.L[j];
        PGj[q] = gc.PGL[j];
    }
#pragma unroll 1
    for (int w = 0; w < nw; ++w) {
        const double* __restrict__ r = wrec + (size_t)(w0 + w) * REC;
        if (wmode[(size_t)(w0 + w) * MAXF * WM] == MODE_SKIP) {   // outside the prior (block-uniform)
            red[w * BLOCK + tid] = 0.0;
            continue;
        }
        __syncthreads();                                   // the previous walker's T_w has been read
#pragma unroll
        for (int q = 0; q < GRIDC_MAX_S / BLOCK; ++q) {
            const int j = tid + q * BLOCK;
            if (j < gc.S) Tw[j] = fexp_c(fma(r[R_C1], Lj[q] - r[R_LSTAR], r[R_C0]) - PGj[q] * r[R_Q], &tab);
        }
        __syncthreads();
        // dot product with T_w over the bin's rows, eight weights in flight at a time (a rolled loop would pay one
        // L2 latency per row)
        double R = 0.0;
        for (int k0 = 0; k0 < nr; k0 += 8) {
            double o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = k0 + i < nr ? om[(k0 + i) * 16] : 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i) R = fma(Tw[min(j0 + k0 + i, gc.S - 1)], o[i], R);
        }
        const bool bright = kc.specialise && r[R_ALPHAC] > 0.0 && a4min_g * walker_vmin(kc, r) > 37.5;
        const double Fs = field_sum_nf(kc, r, r[R_ALPHAC], u, a4, &tab, bright);
        red[w * BLOCK + tid] = R * Fs;
    }
    __syncthreads();
    reduce_store(red, nw, partial, (size_t)pstride, w0, c);
}

// ----------------------------------------------------------------------------------------------
// lf_main: pieces A and B in ONE launch.  Workgroups [0, nblkB) integrate the grid (piece B),
// the rest sum the catalogue (piece A); both kinds take about the same time per workgroup
// at the big geometry (TW walkers x 256 items); the small geometry gives the grid part TWB = 2 walkers
// per workgroup so that a small batch still spreads over the chip.  B first: it never forms the tail.
// ----------------------------------------------------------------------------------------------
// XCD-aware renumbering of a run of n workgroup ids: workgroups are dealt round-robin over the 8 XCDs
// (id % 8), each with its own L2; the result is an index such that consecutive indices sit on ONE XCD,
// in dispatch order.  Used so that the tiles which read the SAME catalogue chunk run side by side on
// one XCD: the chunk comes from HBM once and is served to the others from L2 (speed only).
__device__ __forceinline__ int xcd_renumber(int id, int n) {
    const int q = n >> 3, rem = n & 7, xcd = id & 7;
    return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (id >> 3);
}

// Tapered walker tiling of the per-source part: walkers [0, B1) in tiles of `tw`, walkers [B1, B) in tiles
// of `tws` << tw, and the small tiles are dispatched LAST (largest-first): the launch tail is made of
// quarter-size workgroups, which cuts the drain time of a grid that is only ~4 residency rounds long.
struct Tiling {
    int tw, ntiles;        // big tiles
    int B1, tws, ntiles_s; // small tiles over walkers B1 .. B-1
};

// Compressed-catalogue launches (CMP) append `nresc` rescue workgroups: they do nothing unless lf_prepare
// flagged a walker STAT_SLOW, and then sum that walker over the REAL catalogue `sd` (chunk r, r + nresc, ...)
// into partR[w][chunk] with the per-source path - such a walker's result is the direct path's (same code, same
// per-term checks; bitwise when the direct path runs with the same chunk size).
struct Rescue {
    SrcArrays sd;
    const int* slow_list;       // walkers flagged STAT_SLOW by lf_prepare, in arrival order
    const int* slow_count;      // how many
    double* partR;
    int nchD, nresc;
};

// ZEVOL on the real catalogue: `nchC` chunks of up to BLOCK cells in redshift (ZCELL_RHO above; one field per chunk).
// Walkers flagged STAT_CELLS by lf_prepare are summed over them by workgroups appended to the launch, into
// partC[w][chunk], which lf_finalize then takes instead of partA; a per-source workgroup whose walkers all carry the
// flag leaves at once.
struct ZCells {
    const double* cells;        // [ncell][8] {z_c, S_0 .. S_6}
    const int* cc_start;        // [nchC]
    const int* cc_len;
    const int* cc_field;
    int nchC;                   // 0: no cells
    double* partC;              // [B][nchC]
    const int* wstat;           // [B]
    int nwork;                  // workers that share the per-source items (lf_main)
};
constexpr int ZCELL_MAXB = 16384;         // walkers whose "needs the sources" bits have a bit of their own in a worker's LDS

// one item: cell chunk cc x walkers w0 .. w0+nw-1 (a thread = a cell)
template <int TW>
__device__ __forceinline__ void zcell_body(const KConst& kc, const ZCells& zc, const double* __restrict__ wrec, int cc, int w0, int nw,
                                           const MathTables& tab, double* __restrict__ red) {
    const int tid = threadIdx.x;
    const int c0 = zc.cc_start[cc], ncl = zc.cc_len[cc];
    const double2* __restrict__ src = reinterpret_cast<const double2*>(zc.cells + (size_t)(c0 + min(tid, ncl - 1)) * 8);
    double cd[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double2 a = src[k];
        cd[2 * k] = tid < ncl || k == 0 ? a.x : 0.0;       // (z_c always; a thread past the end: every power sum 0)
        cd[2 * k + 1] = tid < ncl ? a.y : 0.0;
    }
    const double zc2 = cd[0] * cd[0];
    if (kc.forms && tid == 0) atomicAdd(kc.forms + FORM_CELL, (unsigned long long)ncl * nw);
#pragma unroll 1
    for (int w = 0; w < nw; ++w) {
        const double* __restrict__ r = wrec + (size_t)(w0 + w) * REC;
        const double aL = uni(r[Z_AL]), bL = uni(r[Z_BL]), cL = uni(r[Z_CL]);
        asm volatile("; LF_BEGIN zcell items=1");
        const double Lc = quad_nofma(aL, bL, cL, cd[0], zc2);
        const double Hc = fexp_c(LF_LN10 * (LF_LREF - Lc), &tab);
        const double b2 = -2.0 * LF_LN10 * aL;                              // 2 b
        const double a = fma(b2, cd[0], -LF_LN10 * bL);                     // -ln10 L*'(z_c)
        // c_j by the recurrence, the sum with the small terms kept apart from S_0 (the cell's total weight)
        double cm2 = 1.0, cm1 = a;
        double tail = 0.0;
        double head = fma(cm1, cd[2], cd[1]);                               // S_0 + c_1 S_1
#pragma unroll
        for (int j = 2; j <= ZCELL_M; ++j) {
            const double cj = fma(a, cm1, b2 * cm2) * (1.0 / j);
            tail = fma(cj, cd[1 + j], tail);
            cm2 = cm1;
            cm1 = cj;
        }
        red[w * BLOCK + tid] = -Hc * (head + tail);                         // everything else of the term is in wbase
        asm volatile("; LF_END zcell");
    }
    __syncthreads();
    reduce_store(red, nw, zc.partC, (size_t)zc.nchC, w0, cc);
}

template <int VARIANT, int ST, int TW, int TWB, bool CMP>
__global__ __launch_bounds__(BLOCK) void lf_main(KConst kc, SrcArrays sa, NodeArrays na,
                                                 const double* __restrict__ wrec, const int* __restrict__ wmode,
                                                 int B, Tiling tl, int nchA, int ntilesB, int twb, int nblkB,
                                                 double* __restrict__ partA, int strideA,
                                                 double* __restrict__ partB, int strideB, Rescue rs, GridC gc, ZCells zc) {
    warm_kernarg<sizeof(KConst) + sizeof(SrcArrays) + sizeof(NodeArrays) + 2 * 8 + 4 + sizeof(Tiling) + 4 * 4 + 16 + 16 + sizeof(Rescue) +
                 sizeof(GridC) + sizeof(ZCells)>();      // (lf_math.h: the arguments in one round trip)
    __shared__ MathTables tab;
    __shared__ __attribute__((aligned(16))) double red[(TW > TWB ? TW : TWB) * BLOCK];
    __shared__ double Tw[CMP ? GRIDC_MAX_S : 1];
    if (CMP) {
        // rescue workgroups have nothing to do unless lf_prepare listed a walker: leave before the table prologue
        const int first_resc = nblkB + nchA * (tl.ntiles + tl.ntiles_s);
        if ((int)blockIdx.x >= first_resc && *rs.slow_count == 0) return;
    }
    int id = blockIdx.x;
    constexpr bool ZC = VARIANT == LF_ZEVOL && !CMP;
    const int nsrc_wg = nchA * (tl.ntiles + tl.ntiles_s);
    // item `sid` of the per-source part -> chunk, first walker, walkers
    auto src_item = [&](int sid, int& c, int& w0, int& nw) {
        const int nbig = nchA * tl.ntiles;
        if (sid < nbig) {
            const int wg = xcd_renumber(sid, nbig);
            c = wg / tl.ntiles;
            w0 = (wg - c * tl.ntiles) * tl.tw;
            nw = min(tl.tw, tl.B1 - w0);
        } else {
            const int wg = xcd_renumber(sid - nbig, nchA * tl.ntiles_s);
            c = wg / tl.ntiles_s;
            w0 = tl.B1 + (wg - c * tl.ntiles_s) * tl.tws;
            nw = min(tl.tws, B - w0);
        }
    };
    if constexpr (VARIANT != LF_FREE && !CMP) if (zc.nwork > 0) {
        // Z-evolving with cells, fixed completeness: the per-source items are normally all idle (every walker inside the
        // prior is summed over the cells; the fixed-completeness sum is closed-form unless a walker is on the careful path).
        // Instead of one workgroup per item - 3920 of them at 10^6 sources and 128 rows, each dispatched only to leave,
        // 0.8 ns apiece - `zc.nwork` workers share the items, and a worker first looks at the walkers' flags once (a bit per
        // walker in LDS) and leaves at once when no walker needs the sources.  (lf_finalize ignores the per-source
        // partials of the walkers that did not need them.)
        __shared__ unsigned needbits[ZCELL_MAXB / 32];
        __shared__ int anyneed;
        if (id >= nblkB && id < nblkB + zc.nwork) {
            const int tid = threadIdx.x;
            if (tid == 0) anyneed = 0;
            for (int i = tid; i < ZCELL_MAXB / 32; i += BLOCK) needbits[i] = 0u;
            __syncthreads();
            for (int w = tid; w < B; w += BLOCK) {
                // (walkers outside the prior or already known to be -inf need no sums either: lf_finalize)
                const int st_ = zc.wstat[w];
                const bool wants = VARIANT == LF_FIXCOMP ? (st_ & STAT_SLOW) != 0 : !(st_ & STAT_CELLS);
                if (wants && (st_ & STAT_PRIOR_OK) && !(st_ & STAT_NEGINF)) {
                    atomicOr(&needbits[(w >> 5) & (ZCELL_MAXB / 32 - 1)], 1u << (w & 31));      // (B > ZCELL_MAXB: bits shared, only ever too many set)
                    anyneed = 1;
                }
            }
            __syncthreads();
            if (!anyneed) return;
            load_tables_256(&tab);
            __syncthreads();
            for (int sid = id - nblkB; sid < nsrc_wg; sid += zc.nwork) {
                int c, w0, nw;
                src_item(sid, c, w0, nw);
                int need = 0;
                for (int w = w0; w < w0 + nw; ++w) need |= (needbits[(w >> 5) & (ZCELL_MAXB / 32 - 1)] >> (w & 31)) & 1u;
                if (!need) continue;                   // (uniform: LDS broadcast reads)
                srcsum_body<VARIANT, ST, TW, CMP>(kc, sa, wrec, wmode, c, w0, nw, partA, strideA, tab, red);
                __syncthreads();
            }
            return;
        }
    }
    load_tables_256(&tab);
    __syncthreads();
    if (ZC && zc.nchC > 0 && zc.nwork > 0 && id >= nblkB + zc.nwork) {
        const int k = id - (nblkB + zc.nwork);          // cell workgroups: (chunk, tile of TW walkers)
        const int ntc = (B + TW - 1) / TW;
        const int cc = k / ntc, w0c = (k - cc * ntc) * TW;
        zcell_body<TW>(kc, zc, wrec, cc, w0c, min(TW, B - w0c), tab, red);
        return;
    }
    if (id < nblkB) {
        if (CMP && VARIANT == LF_FREE && gc.nb > 0) gridc_body<TWB>(kc, gc, wrec, wmode, B, ntilesB, twb, id, partB, strideB, tab, red, Tw);
        else gridsum_body<VARIANT, TWB>(kc, na, wrec, wmode, B, ntilesB, twb, id, partB, strideB, tab, red);
        return;
    }
    id -= nblkB;
    const int nbig = nchA * tl.ntiles;
    if (CMP && id >= nbig + nchA * tl.ntiles_s) {
        // items = (chunk of the real catalogue) x (tile of up to TW listed walkers), dealt round-robin
        const int r = id - (nbig + nchA * tl.ntiles_s);
        const int nslow = *rs.slow_count;
        const int nt = (nslow + TW - 1) / TW;
        for (long long it = r; it < (long long)rs.nchD * nt; it += rs.nresc) {
            const int c = (int)(it / nt), t = (int)(it - (long long)c * nt);
            srcsum_body<VARIANT, ST, TW, false, true>(kc, rs.sd, wrec, wmode, c, 0, min(TW, nslow - t * TW), rs.partR, rs.nchD, tab, red,
                                                      rs.slow_list + t * TW);
            __syncthreads();
        }
        return;
    }
    int c, w0, nw;
    src_item(id, c, w0, nw);
    srcsum_body<VARIANT, ST, TW, CMP>(kc, sa, wrec, wmode, c, w0, nw, partA, strideA, tab, red);
}

// accept / reject walker k = half*halfW + w with the new lnprob `newlp`, and record it in the chain
// (emcee keeps the state after the full step; a walker only changes in its own half-step).  One wave
// per walker; lanes < ndim move the coordinates.
// (oldlp, zz, propv, posv: the walker's current lnprob, its stretch factor, and this lane's coordinate of the proposal and of the
// current position - loaded by the caller, ahead of the sums the new lnprob comes from)
// The accept step's two logarithms do not depend on the new lnprob: accept_terms makes them (the one-launch form: a wave that is
// idle during the tile's preparation, so that the epilogue - the tail of the launch - is left with two adds and a compare).
__device__ __forceinline__ void accept_terms(const AcceptArgs& ap, int w, double zz, double& lnz_term, double& logu) {
    unsigned int rr[4];
    sampler_draw(ap.step, ap.half, w, 1, ap.seed, rr);
    lnz_term = (ap.ndim - 1.0) * log(zz);
    logu = log(u53(rr[0], rr[1]));
}
__device__ __forceinline__ void accept_walker(const AcceptArgs& ap, int w, double newlp, int lane, double oldlp, double zz, double propv,
                                              double posv, const double* pre = nullptr) {
    const int k = ap.half * ap.halfW + w;
    const long long t = ap.t;
    double lnz_term, logu;
    if (pre) {
        lnz_term = pre[0];
        logu = pre[1];
    } else {
        accept_terms(ap, w, zz, lnz_term, logu);
    }
    const double lnq = lnz_term + newlp - oldlp;
    const bool acc = (logu < lnq) && (newlp > -__builtin_huge_val());
    if (lane < ap.ndim) {
        const double v = acc ? propv : posv;
        if (acc) ap.pos[(size_t)k * ap.ndim + lane] = v;
        ap.chain[((size_t)k * ap.cap + t) * ap.ndim + lane] = v;
    }
    if (lane == 0) {
        if (acc) {
            ap.lnp[k] = newlp;
            ap.nacc[k] += 1;
        }
        ap.chain_lnp[(size_t)k * ap.cap + t] = acc ? newlp : oldlp;
    }
}

// the two halves of a sampler half-step as separate launches, for the walker-sharded (multi-GPU)
// sampler: every rank proposes for the whole half, evaluates its slice with the plain lnprob path,
// all-gathers, and accepts for the whole half - same arithmetic as the fused path, same chain.
__global__ __launch_bounds__(64) void lf_propose(StepArgs sp) {
    const int gt = blockIdx.x * blockDim.x + threadIdx.x;
    const int w = gt >> 3, f = gt & 7;
    if (w >= sp.halfW) return;
    unsigned int rr[4];
    sampler_draw(sp.step, sp.half, w, 0, sp.seed, rr);
    const double z = stretch_z(sp.a, u53(rr[0], rr[1]));
    const int j = (1 - sp.half) * sp.halfW + (int)(((unsigned long long)rr[2] * (unsigned long long)sp.halfW) >> 32);
    const int k = sp.half * sp.halfW + w;
    for (int i = f; i < sp.ndim; i += 8)
        sp.prop[(size_t)w * sp.ndim + i] = stretch_point(sp.pos[(size_t)j * sp.ndim + i], sp.pos[(size_t)k * sp.ndim + i], z);
    if (f == 0) sp.zz[w] = z;
}

__global__ __launch_bounds__(64) void lf_accept(AcceptArgs ap, const double* __restrict__ newlp) {
    const int w = blockIdx.x;
    if (w >= ap.halfW) return;
    const int k = ap.half * ap.halfW + w, i = min((int)threadIdx.x, ap.ndim - 1);
    accept_walker(ap, w, newlp[w], threadIdx.x, ap.lnp[k], ap.zz[w], ap.prop[(size_t)w * ap.ndim + i], ap.pos[(size_t)k * ap.ndim + i]);
}

// ----------------------------------------------------------------------------------------------
// finalize: one wave per walker; fixed-order sum of the partials; lnprob = lnprior + A - B.
// lumfuncmcmc.py:378, :403-409.  Never NaN (emcee raises on NaN): NaN -> -inf.
// ----------------------------------------------------------------------------------------------
// One wave finishes walker w.  COHERENT: the partials were written by OTHER workgroups of the launch that is still
// running, on other XCDs (lf_free's fused epilogue: written through to memory, see there) - read them from there, past
// this XCD's caches.
template <bool COHERENT>
__device__ __forceinline__ void finalize_wave(const double* partA, int nchA, int strideA, const double* partB, int nchB, int strideB,
                                              const double* partR, int nchR, int alt_flag, const int* wstat, const double* wbase,
                                              int w, int lane, const AcceptArgs& ap, double* out, double* outA, double* outB,
                                              int a_flag = 0, const double* l_prop = nullptr, const double* l_zz = nullptr,
                                              const double* l_pre = nullptr, const double* polled = nullptr) {
    auto ld = [](const double* p) -> double {
        if (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return *p;
    };
    // the sampler's half-step: what the accept step reads of the walker's state is on its way while the partial sums are added
    // (l_prop / l_zz: proposal and stretch factor of the tile in LDS - the one-launch form; else in memory, lf_prepare's)
    double s_old = 0.0, s_zz = 1.0, s_prop = 0.0, s_pos = 0.0;
    if (ap.enabled) {
        const int k = ap.half * ap.halfW + w;
        const int i = min(lane, ap.ndim - 1);
        s_old = ap.lnp[k];
        s_pos = ap.pos[(size_t)k * ap.ndim + i];
        s_zz = l_zz ? l_zz[0] : ap.zz[w];
        s_prop = l_prop ? l_prop[i] : ap.prop[(size_t)w * ap.ndim + i];
    }
    double a = 0.0, b = 0.0;
    // piece A of a walker with the alt_flag bit lies in partR: the compressed catalogue's SLOW walkers, summed over the
    // real catalogue by the rescue workgroups (alt_flag = STAT_SLOW); lf_free's walkers summed over cells (STAT_CELLS)
    const int st = wstat[w];                          // (COHERENT: the workgroup's own copy, see lf_free.h)
    const bool resc = partR != nullptr && (st & alt_flag);
    const double* pa = resc ? partR + (size_t)w * nchR : partA + (size_t)w * strideA;
    const double* pb = partB + (size_t)w * strideB;
    if (resc) nchA = nchR;
    // (a_flag: per-source partials were only made for walkers with this flag - fixed completeness, careful path)
    else if (a_flag && !(st & a_flag)) nchA = 0;
    if (polled) {
        // (lf_free's polling finisher: the lane's slot of partR and of partB is in hand - at most 64 slots apiece, so the sums
        // below are the ones the loops would make; a walker that is not on the cells has no sources either, and is -inf)
        a = resc && lane < nchR ? polled[0] : 0.0;
        b = lane < nchB ? polled[1] : 0.0;
    } else {
    // four independent running sums per lane so that the loads are in flight together (latency kernel)
    double a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int c = lane;
    for (; c + 192 < nchA; c += 256) {
        a += ld(pa + c);
        a1 += ld(pa + c + 64);
        a2 += ld(pa + c + 128);
        a3 += ld(pa + c + 192);
    }
    for (; c < nchA; c += 64) a += ld(pa + c);
    a = (a + a1) + (a2 + a3);
    for (c = lane; c < nchB; c += 64) b += ld(pb + c);
    }
    a = wave_sum_dpp(a);                            // totals in lane 63
    b = wave_sum_dpp(b);
    if (lane == 63) {
        const bool ok = (st & STAT_PRIOR_OK) != 0;
        a += wbase[w];
        if ((st & STAT_NEGINF) || a != a) a = -__builtin_huge_val();
        double r = a - b;
        if (!ok || r != r) r = -__builtin_huge_val();
        if (out) out[w] = r;
        if (outA) outA[w] = ok ? a : __builtin_nan("");
        if (outB) outB[w] = ok ? b : __builtin_nan("");
        a = r;                                   // lane 63 keeps the new lnprob for the accept step
    }
    if (ap.enabled)                                  // (the sampler's half-step: accept / reject and the chain's row)
        accept_walker(ap, w, __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a), 63), __builtin_amdgcn_readlane(__double2loint(a), 63)), lane,
                      s_old, s_zz, s_prop, s_pos, l_pre);
}

__global__ __launch_bounds__(64) void lf_finalize(const double* __restrict__ partA, int nchA, int strideA,
                                                  const double* __restrict__ partB, int nchB, int strideB,
                                                  const double* __restrict__ partR, int nchR, int alt_flag,
                                                  const int* __restrict__ wstat,
                                                  const double* __restrict__ wbase, int B, AcceptArgs ap,
                                                  double* __restrict__ out, double* __restrict__ outA,
                                                  double* __restrict__ outB, int* __restrict__ slow_list, int a_flag) {
    warm_kernarg<12 * 8 + 6 * 4 + sizeof(AcceptArgs)>();      // (lf_math.h: the arguments in one round trip)
    const int w = blockIdx.x;
    if (w >= B) return;
    if (slow_list && w == 0 && threadIdx.x == 0) slow_list[0] = 0;     // lf_main has consumed the list
    finalize_wave<false>(partA, nchA, strideA, partB, nchB, strideB, partR, nchR, alt_flag, wstat, wbase, w, (int)threadIdx.x, ap, out,
                         outA, outB, a_flag);
}

// ----------------------------------------------------------------------------------------------
// 1/Veff estimator (post-fit diagnostic): LumFuncMCMC.VeffLF, lumfuncmcmc.py:515-525 ->
// V.lumfunc (VmaxLumFunc.py:235-257) and V.getBootErrLog (:304-378).
//   veff_weights: phi_i = 1 / (sum(Omega_0)/sqarcsec * fleming(F_i, Flim_i, alpha, fcmin) * vol_i)   (device-library math)
//   veff_bins:    for resample k (blockIdx.y) the binned sums sum_{j} phi[idx_kj] [bin(idx_kj) == b]; k = 0 is the
//                 catalogue itself, k >= 1 draws idx_kj = floor(u N) from Philox4x32-10 (counter = (j, k), key = seed),
//                 or reads it from `boot_idx` when the caller supplies the indices (tests replay the reference's seeded
//                 numpy stream through the same kernel).  LDS bins per workgroup, one global atomic per bin.
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void veff_weights(const double* __restrict__ flux, const double* __restrict__ flim, const double* __restrict__ vol,
                                                    double vol_all, double pref0, double alpha, double fc_ratio, int use_fcmin,
                                                    long long n, double* __restrict__ phi) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double f = flux[i], fl = flim[i];
    const double num = alpha * log10(f / fl);                               // VmaxLumFunc.py:118-120
    double fc = 0.5 * (1.0 + num / sqrt(1.0 + num * num));
    if (use_fcmin) {
        const double ftau = fl * exp10(-sqrt(fc_ratio / (alpha * alpha)));   // :164-167
        fc = pow(fc, 1.0 / (1.0 - exp(-f / ftau)));                          // :141, :125
    }
    const double v = vol ? vol[i] : vol_all;
    phi[i] = v > 0.0 ? 1.0 / (pref0 * fc * v) : 0.0;
}

constexpr int VEFF_MAXBIN = 1024;
__global__ __launch_bounds__(256) void veff_bins(const double* __restrict__ phi, const int* __restrict__ bin_of, long long n, int nbin,
                                                 const long long* __restrict__ boot_idx, unsigned long long seed,
                                                 double* __restrict__ sums /* [nres][nbin] */) {
    __shared__ double lb[VEFF_MAXBIN];
    const int k = blockIdx.y;
    for (int b = threadIdx.x; b < nbin; b += blockDim.x) lb[b] = 0.0;
    __syncthreads();
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (long long)gridDim.x * blockDim.x) {
        long long idx = j;
        if (k > 0) {
            if (boot_idx) {
                idx = boot_idx[(size_t)(k - 1) * (size_t)n + (size_t)j];
            } else {
                unsigned int r[4];
                philox4x32((unsigned int)j, (unsigned int)((unsigned long long)j >> 32), (unsigned int)k, 0x5eedu, (unsigned int)seed,
                           (unsigned int)(seed >> 32), r);
                idx = (long long)(u53(r[0], r[1]) * (double)n);
                if (idx >= n) idx = n - 1;
            }
        }
        const int b = bin_of[idx];
        if (b >= 0 && b < nbin) atomicAdd(&lb[b], phi[idx]);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nbin; b += blockDim.x)
        if (lb[b] != 0.0) atomicAdd(&sums[(size_t)k * nbin + b], lb[b]);
}

}  // namespace lf
