// lfmcmc.hip - C ABI (include/lfmcmc.h) over the gfx950 kernels in lf_kernels.h.
//
// Host side of the boundary: copies the catalogue and grids to HBM once, derives the
// parameter-independent tables (P_i, U_i, trapezoid weights, field-summed integrand), and per
// call enqueues prepare -> per-source sum -> grid integral -> finalize on one HIP stream.
#include "../../include/lfmcmc.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "lf_compress.h"
#include "lf_gridbound.h"
#include "lf_kernels.h"
#include "lf_free.h"
#include "lf_pers.h"

namespace {

thread_local std::string g_create_error = "";

struct ChunkTable {
    int n = 0;
    int* d_start = nullptr;
    int* d_len = nullptr;
    int* d_field = nullptr;
    int* d_keys = nullptr;     // FREE, real catalogue: lf::KEY_STRIDE ints per chunk (get_chunks)
};

struct EventPair {
    hipEvent_t a, b;
    int kind;
    int count = 1;        // launches between the two events (option "profile_span")
};

// compressed catalogue (lf_compress.h): weighted pseudo-sources, sources of a field contiguous
struct CompressedCat {
    bool built = false;
    int64_t n = 0;
    std::vector<int64_t> field_ind;
    double *d_lum = nullptr, *d_a1 = nullptr, *d_U = nullptr, *d_W = nullptr;
    std::map<int, ChunkTable> chunks;
    int nbins = 0;
    double bound = 0.0;
};

}  // namespace

struct lf_ctx {
    lf::KConst kc{};
    int device = 0;
    int64_t N = 0;
    int nnodes = 0;
    std::vector<int64_t> field_ind;
    std::vector<double> h_x;            // FREE: host copy of the flux-sorted logf (chunk keys are derived from it)
    unsigned long long* d_forms = nullptr;   // census of the term forms (option "count_forms"), FORM_COUNT slots
    int last_launch[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // lf_last_launch
    int* d_queue = nullptr;              // FREE: item counters of the persistent workgroups, [tiles][lf::QSTRIDE]
    // the catalogue's cells (lf_kernels.h: CELL_M, ZCELL_M) - {x_c, S_0 .. S_M} per cell, chunks of 64 (FREE) / 256 (ZEVOL) cells of one field
    double* d_cells = nullptr;           // [ncell][8]
    int* d_cc_start = nullptr;           // [ncchunk] first cell of the chunk
    int* d_deal = nullptr;               // lf_free's deal of cell chunks and flux bins to its virtual workgroups (lf_free.h: DEAL_*)
    int64_t deal_key = -1;               // ... made for this (cell chunks, bins, grid share)
    int* d_cc_len = nullptr;             // [ncchunk] cells in the chunk (<= 64)
    int* d_cc_field = nullptr;           // [ncchunk]
    int ncell = 0, ncchunk = 0;
    int64_t opt_cells = 1;               // 0: sum every walker over the sources (A/B runs)
    int cap_queue = 0;
    int slots_free[3] = {0, 0, 0};       // workgroups of lf_free<2 / 4 / 8> the chip holds at once (0 = not asked yet)
    int num_cu = 0;
    // device tables
    double *d_lum = nullptr, *d_a1 = nullptr, *d_P = nullptr, *d_U = nullptr;
    double *d_G = nullptr, *d_PG = nullptr, *d_W = nullptr, *d_a3 = nullptr, *d_a4 = nullptr, *d_a4min = nullptr, *d_nodes8 = nullptr;
    std::map<int, ChunkTable> chunks;   // keyed by sources-per-chunk
    std::map<int, ChunkTable> chunks_free;   // the persistent FREE kernel's (lf_free.h): 512 ST sources per chunk, lanes of ST
    int64_t opt_persistent = 1;         // FREE: 1 = lf_free (persistent 512-thread workgroups) for catalogues that fill it
    int64_t opt_fuse = 1;               // lf_free: prepare and finalize inside the one launch (plain evaluations)
    int64_t opt_fuse_step = 1;          // ... and the sampler's half-step too (0: three launches per half-step, A/B runs)
    bool queue_zero = false;            // d_queue is all zeros (what a fused launch needs and leaves behind)
    bool parts_empty = false;           // every slot of d_partB / d_partR is lf::PART_EMPTY (what lf_free's polling hand-over needs and leaves behind)
    int* d_err = nullptr;               // device error word (a finisher gave up polling)
    int64_t opt_poll = 1;               // 0: the one-launch form hands over through the tile's counter only (A/B runs)
    int64_t opt_free_st = 0;            // lf_free: sources per lane, 0 = chosen from N and B, else 2 / 4 / 8 (tuning runs)
    int64_t opt_geometry = -1;          // index into GEOS, -1 = auto
    int64_t opt_walker_tile = 0;        // walkers per workgroup (<= the geometry's maximum), 0 = auto
    int64_t opt_taper = 0;              // 1: quarter-size walker tiles for the last ~1/8 of the walkers (second pass over the catalogue)
    int64_t opt_skip_grid = 0;          // 1: leave piece B out (source-sharded ranks other than the first)
    int64_t opt_compress = 0;           // 1: piece A from the compressed catalogue (FREE, ZEVOL)
    int64_t opt_compress_grid = 1;      // with compress: also the FREE integration grid, when it is separable
    CompressedCat cmp;
    // FREE: the factors of a separable integration grid (every redshift column has the same luminosity nodes),
    // kept on the host for the compressed grid; empty when the grid is not separable
    std::vector<double> h_L, h_wL, h_ck, h_Dk;
    struct {
        bool built = false;
        int nb = 0;
        double bound = 0.0;
        double *d_U = nullptr, *d_A4 = nullptr, *d_omega = nullptr, *d_L = nullptr, *d_PGL = nullptr;
        int *d_row0 = nullptr, *d_nrows = nullptr, *d_off = nullptr;
    } gridc;
    // FREE, separable grid: piece B over flux bins with a proven bound (lf_gridbound.h); lf_free's default when built
    struct {
        bool built = false;
        int nb = 0;
        double margin = 0.0;
        double *d_rec = nullptr, *d_omega = nullptr;
        int* d_rows = nullptr;
    } gridq;
    int64_t opt_grid_shortcut = 1;      // 0: lf_free integrates the lattice (A/B runs)
    // FIXCOMP, ZEVOL: the grid's nodes as 32-byte records {G, PG, W, column} padded to chunks of 64, and the columns' redshifts
    // (lf_pers.h: the persistent kernel of these variants)
    double *d_nodes4 = nullptr, *d_zcol = nullptr;
    int nch4 = 0;
    int slots_pers = 0;                 // workgroups of lf_pers the chip holds at once (0 = not asked yet)
    double* d_partR = nullptr;          // rescue partials [B][chunks of the real catalogue]
    int* d_slow = nullptr;              // {count, walker indices...} of the walkers lf_prepare flagged SLOW (compressed mode)
    int cap_slow = 0;
    size_t cap_partR = 0;
    // workspace
    int cap_B = 0;                      // padded walker capacity
    size_t cap_partA = 0, cap_partB = 0;
    double *d_theta = nullptr, *d_out = nullptr, *d_outA = nullptr, *d_outB = nullptr;
    double *d_wrec = nullptr, *d_partA = nullptr, *d_partB = nullptr;
    int *d_wstat = nullptr, *d_wmode = nullptr;
    double* d_wbase = nullptr;
    double *h_theta = nullptr, *h_out = nullptr;   // pinned staging
    hipStream_t stream = nullptr;
    hipStream_t last_stream = nullptr;   // stream of the previous enqueue (workspace is shared)
    bool any_enqueued = false;
    // profiling
    int profiling = 0;   // 0 off, 1 lf_main only, 2 every launch
    lf::ZCells zcells{};                // ZEVOL, real catalogue: the cell workgroups' arguments for the launch being enqueued (nchC = 0: none)
    int64_t opt_profile_every = 1;      // ... of every n-th evaluation only (an event pair costs the stream ~4 us: it drains the queue)
    int64_t prof_tick = 0;
    int64_t opt_profile_span = 1;       // one event pair around this many CONSECUTIVE one-launch evaluations (their average: a pair of
                                        // barrier packets around every single launch adds the dispatch to it)
    int64_t prof_pos = 0;               // this evaluation's place in its period of opt_profile_every
    bool prof_span_ok = false;          // this evaluation is one launch (spans make sense)
    bool span_open = false;
    EventPair span_ep{};
    bool prof_this = true;              // (this evaluation is one of them)
    std::vector<EventPair> events;
    double acc_ms[4] = {0, 0, 0, 0};
    int64_t acc_n[4] = {0, 0, 0, 0};
    std::string err;
};

namespace {

#define LF_HIP(ctx, call)                                                                  \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                \
            return LF_ERR_HIP;                                                             \
        }                                                                                  \
    } while (0)

template <typename T>
int upload(lf_ctx* c, T** dst, const T* src, size_t n) {
    LF_HIP(c, hipMalloc((void**)dst, std::max<size_t>(n, 1) * sizeof(T)));
    if (n) LF_HIP(c, hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return LF_OK;
}

// The cells of a FREE catalogue (lf_kernels.h: CELL_M): runs of flux-neighbouring sources of one field no wider than
// 2 rho, rho = min(CELL_RHO_H, CELL_RHO_G / alpha_hi) with alpha_hi the prior box's largest alpha_C (walkers outside the
// box are -inf before any sum is looked at).  x = the flux-sorted logf.  Walker-independent: built once.  A field with a
// non-finite flux, or a prior box so wide in alpha_C that cells would hold fewer than four sources on average, gets
// none (kc.cells = 0: every walker is summed over the sources, as before).
// ZEVOL: half the width of a cell in redshift such that every walker inside the prior box of (L1, L2, L3) may be summed
// over the cells (lf_kernels.h: ZCELL_X1, ZCELL_X2).  L*(z) is the parabola through (pivot_i, L_i): its slope at a given
// z and its curvature are linear in (L1, L2, L3), so their largest magnitudes over the box are taken at its corners, and
// the slope's over the catalogue's redshifts at their ends.  0: no cells (an unbounded box, coinciding pivots).
double zcell_rho_for_box(const lf::KConst& kc, int nf) {
    using namespace lf;
    const double lo = kc.lims[LF_LIM_LSTAR][0], hi = kc.lims[LF_LIM_LSTAR][1];
    double zmin = HUGE_VAL, zmax = -HUGE_VAL;
    for (int f = 0; f < nf; ++f)
        if (kc.nsrc[f] > 0) {
            zmin = std::fmin(zmin, kc.z_lo[f]);
            zmax = std::fmax(zmax, kc.z_hi[f]);
        }
    if (!(std::isfinite(lo) && std::isfinite(hi) && std::isfinite(zmin) && std::isfinite(zmax))) return 0.0;
    const double z1 = kc.pivots[0], z2 = kc.pivots[1], z3 = kc.pivots[2];
    if (!(z1 != z2 && z2 != z3 && z1 != z3)) return 0.0;
    double smax = 0.0, amax = 0.0;
    for (int corner = 0; corner < 8; ++corner) {
        const double L1 = corner & 1 ? hi : lo, L2 = corner & 2 ? hi : lo, L3 = corner & 4 ? hi : lo;
        const double d12 = (L2 - L1) / (z2 - z1), d23 = (L3 - L2) / (z3 - z2);
        const double a = (d23 - d12) / (z3 - z1);                     // divided differences: L* = L1 + d12 (z - z1) + a (z - z1)(z - z2)
        for (double z : {zmin, zmax}) smax = std::fmax(smax, std::fabs(d12 + a * (2.0 * z - z1 - z2)));
        amax = std::fmax(amax, std::fabs(a));
    }
    double rho = ZCELL_RHO;
    // (a little inside the limits: lf_prepare evaluates the same quantities from its own rounded coefficients)
    if (smax > 0.0) rho = std::fmin(rho, 0.98 * ZCELL_X1 / (LF_LN10 * smax));
    if (amax > 0.0) rho = std::fmin(rho, std::sqrt(0.98 * ZCELL_X2 / (LF_LN10 * amax)));
    return std::isfinite(rho) ? rho : 0.0;
}

// wts: NULL (FREE: cells in log-flux, plain power sums, chunks of 64 cells) or the sources' weights (ZEVOL: cells in
// redshift, S_j = sum_i wts_i d_i^j, chunks of BLOCK cells; lf_kernels.h: ZCELL_RHO)
int build_cells(lf_ctx* c, lf::KConst& kc, const std::vector<double>& x, int nf, const double* wts = nullptr) {
    using namespace lf;
    const double ahi = kc.lims[LF_LIM_ALPHA][1];
    if (!wts && (!(ahi > 0.0) || !std::isfinite(ahi))) return LF_OK;
    const double rho = wts ? kc.zcell_rho : std::fmin(CELL_RHO_H, CELL_RHO_G / ahi);
    if (!(rho > 0.0)) return LF_OK;
    const size_t per_chunk = wts ? (size_t)BLOCK : 64;
    const int M = wts ? ZCELL_M : CELL_M;                 // orders kept
    const size_t rec = (size_t)M + 2;                     // doubles per cell: midpoint, S_0 .. S_M
    std::vector<double> cd;
    std::vector<int> cst, cln, cfl;
    size_t nreal = 0;                    // cells with sources
    for (int f = 0; f < nf; ++f) {
        const int64_t lo = c->field_ind[f], hi = c->field_ind[f + 1];
        if (hi <= lo) continue;
        for (int64_t i = lo; i < hi; ++i)
            if (!std::isfinite(x[(size_t)i]) || (wts && !(std::isfinite(wts[(size_t)i]) && wts[(size_t)i] > 0.0))) return LF_OK;
        if (!wts) {
            const double k0 = std::floor((x[(size_t)lo] - kc.key_x0) * KEY_SCALE), k1 = std::ceil((x[(size_t)hi - 1] - kc.key_x0) * KEY_SCALE);
            if (!(k0 >= 0.0 && k1 < (double)KEY_MAX)) return LF_OK;
            kc.kf_first[f] = (int)k0;
            kc.kf_last[f] = (int)k1;
        }
        const size_t first_cell = cd.size() / rec;
        for (int64_t i = lo; i < hi;) {
            int64_t j = i + 1;
            while (j < hi && x[(size_t)j] - x[(size_t)i] <= 2.0 * rho) ++j;
            const double xc = 0.5 * (x[(size_t)i] + x[(size_t)j - 1]);
            long double S[CELL_M + 1] = {0};
            static_assert(ZCELL_M <= CELL_M, "S is sized for the larger");
            for (int64_t k = i; k < j; ++k) {
                const long double dlt = (long double)x[(size_t)k] - (long double)xc;
                long double pw = wts ? (long double)wts[(size_t)k] : 1.0L;
                for (int m = 0; m <= M; ++m) {
                    S[m] += pw;
                    pw *= dlt;
                }
            }
            cd.push_back(xc);
            for (int m = 0; m <= M; ++m) cd.push_back((double)S[m]);
            i = j;
        }
        size_t ncf = cd.size() / rec - first_cell;
        nreal += ncf;
        if (!wts) {
            // lf_free addresses chunk cc at cell 64 cc and masks nothing: pad the field to whole chunks with cells of no
            // sources (all sums 0) at the last real midpoint (inside the tables wherever the real cell is)
            kc.cc_fstart[f] = (int)cst.size();
            const double xlast = cd[cd.size() - rec];
            while (ncf % 64) {
                cd.push_back(xlast);
                for (int m = 0; m <= M; ++m) cd.push_back(0.0);
                ++ncf;
            }
        }
        for (size_t s0 = 0; s0 < ncf; s0 += per_chunk) {                // a cell chunk = one wave's lanes (lf_free.h) / one workgroup's threads
            cst.push_back((int)(first_cell + s0));
            cln.push_back((int)std::min<size_t>(per_chunk, ncf - s0));        // (FREE: pads included; they add 0)
            cfl.push_back(f);
        }
    }
    const size_t ncell = cd.size() / rec;
    if (!wts) {                                                       // (a field without cells starts where the next one does)
        for (int f = nf; f <= MAXF; ++f) kc.cc_fstart[f] = (int)cst.size();
        for (int f = nf - 1; f >= 0; --f)
            if (c->field_ind[f + 1] <= c->field_ind[f]) kc.cc_fstart[f] = kc.cc_fstart[f + 1];
    }
    // (too few sources per cell to pay - for a big catalogue: for a small one even cells of one source apiece beat the
    // per-source path, whose cost is its per-item overhead: 10^3 sources, 16 rows: 23.5 us per evaluation over the sources,
    // 17.5 in lf_main's three launches, 12 over cells)
    if (nreal == 0 || ((size_t)c->N < 4 * nreal && c->N > 65536)) return LF_OK;
    int rc;
    if ((rc = upload(c, &c->d_cells, cd.data(), cd.size())) != LF_OK) return rc;
    if ((rc = upload(c, &c->d_cc_start, cst.data(), cst.size())) != LF_OK) return rc;
    if ((rc = upload(c, &c->d_cc_len, cln.data(), cln.size())) != LF_OK) return rc;
    if ((rc = upload(c, &c->d_cc_field, cfl.data(), cfl.size())) != LF_OK) return rc;
    c->ncell = (int)ncell;
    c->ncchunk = (int)cst.size();
    kc.cells = c->opt_cells ? 1 : 0;
    return LF_OK;
}

// hx: the flux-sorted logf of the REAL catalogue (FREE), or NULL (no keys: the chunks never take the table form)
int get_chunks(lf_ctx* c, std::map<int, ChunkTable>& tables, const std::vector<int64_t>& field_ind, int ch,
               ChunkTable** out, const double* hx = nullptr, int lane_w = 0) {
    auto it = tables.find(ch);
    if (it != tables.end()) {
        *out = &it->second;
        return LF_OK;
    }
    std::vector<int> st, ln, fl;
    for (int f = 0; f < c->kc.nf; ++f) {
        for (int64_t s = field_ind[f]; s < field_ind[f + 1]; s += ch) {
            st.push_back((int)s);
            ln.push_back((int)std::min<int64_t>(ch, field_ind[f + 1] - s));
            fl.push_back(f);
        }
    }
    // Keys for the table-driven form of the FREE term (lf_free.h), rounded so that a key test that
    // passes implies the real-valued condition: kfirst = floor, klast = ceil of (x - x0) 2^20 for the chunk's
    // faintest / brightest source; kamax = the largest alpha_C (x 2^16, floor) for which alpha_C times the widest
    // lane of the chunk (a lane = lane_w neighbours in flux) stays within the g table's margin - 0 when that
    // width already exceeds the h table's margin.  A chunk with a non-finite flux gets keys that fail every test.
    // KS ints per chunk: {kfirst, klast, kamax of the whole chunk, -, kamax of each of its 8 waves}: with lanes of
    // flux-neighbours a wave is 64 lane_w consecutive sources, and a chunk's widest lanes cluster in one or two waves (the
    // sparse end of a field): decided per wave, 0.4 % of the (walker, wave) pairs of the bench workload miss the table
    // form instead of 3.4 %.
    constexpr int KS = lf::KEY_STRIDE;
    std::vector<int> keys((size_t)KS * st.size(), 0);
    for (size_t i = 0; i < st.size(); ++i) {
        keys[KS * i] = -1;
        keys[KS * i + 1] = lf::KEY_MAX;
        if (!hx) continue;
        const int64_t s = st[i], n = ln[i];
        if (lane_w <= 0) continue;                // (a kernel that holds no lanes of flux-neighbours: no keys)
        bool finite = true;
        for (int64_t j = 0; j < n; ++j) finite = finite && std::isfinite(hx[s + j]);
        if (!finite) continue;
        const double k0 = std::floor((hx[s] - c->kc.key_x0) * lf::KEY_SCALE), k1 = std::ceil((hx[s + n - 1] - c->kc.key_x0) * lf::KEY_SCALE);
        if (!(k0 >= 0.0 && k1 < (double)lf::KEY_MAX)) continue;
        auto amax_key = [&](double spread) {
            double amax = spread > 0.0 ? lf::G_MARGIN / spread : 3.0e4;
            if (spread > lf::H_MARGIN) amax = 0.0;
            return (int)std::floor(std::fmin(amax, 3.0e4) * lf::KEY_ASCALE);
        };
        double spread = 0.0, wspread[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int64_t per_wave = 64 * (int64_t)lane_w;
        for (int64_t j = 0; j < n; j += lane_w) {
            const double sp = hx[s + std::min<int64_t>(j + lane_w, n) - 1] - hx[s + j];
            spread = std::fmax(spread, sp);
            const int64_t wv = std::min<int64_t>(j / per_wave, 7);
            wspread[wv] = std::fmax(wspread[wv], sp);
        }
        keys[KS * i] = (int)k0;
        keys[KS * i + 1] = (int)k1;
        keys[KS * i + 2] = amax_key(spread);
        for (int wv = 0; wv < 8; ++wv) keys[KS * i + 4 + wv] = amax_key(wspread[wv]);
    }
    // Chunk order.  The kernels deal contiguous runs of chunk indices to the 8 XCDs (lf_main: in dispatch order inside
    // each run; lf_free: one queue per run), and with the catalogue sorted by flux a chunk's cost depends on its rank
    // in its field (bright chunks run the forms without the exponential for most walkers).  So: deal the chunks
    // round-robin into 8 groups (every group gets the same mix of ranks and fields: natural order would hand one XCD
    // only full-cost chunks), and inside a group put the expensive chunks first - longest first keeps the drain of
    // the launch short: the faint ones before the bright ones, and, with lanes of flux-neighbours (lf_free), before
    // both the chunks whose lanes are too wide for the tables at ordinary alpha_C (the sparse ends of a field: the
    // general form, twice the cost per term).
    {
        const size_t n = st.size();
        if (n > 8) {
            std::vector<size_t> rank(n);                  // rank of the chunk inside its field (natural order is field-major)
            for (size_t i = 0, r = 0; i < n; ++i) {
                r = (i > 0 && fl[i] == fl[i - 1]) ? r + 1 : 0;
                rank[i] = r;
            }
            const int wide = (int)(32.0 * lf::KEY_ASCALE);
            auto cls = [&](size_t i) { return (lane_w > 0 && hx) ? std::min(keys[KS * i + 2], wide) : wide; };
            std::vector<size_t> order;
            order.reserve(n);
            for (size_t g = 0; g < 8; ++g) {
                std::vector<size_t> grp;
                for (size_t i = g; i < n; i += 8) grp.push_back(i);
                std::stable_sort(grp.begin(), grp.end(), [&](size_t a, size_t b) {
                    const int ca = cls(a), cb = cls(b);
                    return ca != cb ? ca < cb : rank[a] < rank[b];
                });
                order.insert(order.end(), grp.begin(), grp.end());
            }
            std::vector<int> st2(n), ln2(n), fl2(n), keys2((size_t)KS * n);
            for (size_t i = 0; i < n; ++i) {
                st2[i] = st[order[i]];
                ln2[i] = ln[order[i]];
                fl2[i] = fl[order[i]];
                for (int j = 0; j < KS; ++j) keys2[KS * i + j] = keys[KS * order[i] + j];
            }
            st.swap(st2);
            ln.swap(ln2);
            fl.swap(fl2);
            keys.swap(keys2);
        }
    }
    ChunkTable t;
    t.n = (int)st.size();
    int rc;
    if ((rc = upload(c, &t.d_start, st.data(), st.size())) != LF_OK) return rc;
    if ((rc = upload(c, &t.d_len, ln.data(), ln.size())) != LF_OK) return rc;
    if ((rc = upload(c, &t.d_field, fl.data(), fl.size())) != LF_OK) return rc;
    if ((rc = upload(c, &t.d_keys, keys.data(), keys.size())) != LF_OK) return rc;
    tables[ch] = t;
    *out = &tables[ch];
    return LF_OK;
}

// launch geometries of the per-source kernel: sources per lane (ST) x walkers per workgroup (TW)
struct Geo {
    int st, tw, twb;     // sources per lane, walkers per source workgroup, walkers per grid workgroup
};
// instantiated geometries; [0] and [1] are the defaults for large and small problems
constexpr Geo GEOS[] = {{8, 16, 16}, {2, 8, 2}, {8, 8, 8}, {8, 4, 4}, {4, 8, 4}, {4, 4, 4}, {6, 16, 16}, {4, 16, 16}, {2, 8, 16}};
constexpr int NGEO = sizeof(GEOS) / sizeof(GEOS[0]);

int pick_geometry(const lf_ctx* c, int B) {
    if (c->opt_geometry >= 0 && c->opt_geometry < NGEO) return (int)c->opt_geometry;
    // The big tile (2048 sources x 16 walkers) amortises loads and reductions best, but the launch wants >= ~1024
    // workgroups (4 per CU).  Below that keep the 2048-source chunks and shrink the walker tile (16 -> 8 -> 4);
    // tiny catalogues take the 512-source chunks.  (Measured at B = 128, lf_main in us, geometries 0 / 2 / 3 / 1:
    // N = 4e5: 98 / 95 / 100 / 122;  2e5: 67 / 60 / 62 / 72;  1e5: 49 / 43 / 42 / 49;  3e4: 39 / 30 / 30 / 30.)
    // Fixed completeness on a grid summed over its rows (build(): S nodes, one or two chunks): the per-source part is idle
    // and the grid part is a serial loop over a workgroup's walkers - two per workgroup instead of 16 (lf_main 14.2 -> 6.5 us
    // at 128 rows).
    if (c->kc.variant == LF_FIXCOMP && c->nnodes <= 2 * lf::BLOCK) return 1;
    const int64_t chunks = (c->N + GEOS[0].st * lf::BLOCK - 1) / (GEOS[0].st * lf::BLOCK);
    if (chunks * ((B + 15) / 16) >= 1024) return 0;
    if (chunks * ((B + 7) / 8) >= 1024) return 2;
    if (chunks * ((B + 3) / 4) >= 384) return 3;
    return 1;
}

// free a device / pinned buffer and clear the pointer: a failed re-allocation below must leave nothing dangling
template <typename T>
void release(T*& p) {
    if (p) hipFree(p);
    p = nullptr;
}
template <typename T>
void release_host(T*& p) {
    if (p) hipHostFree(p);
    p = nullptr;
}

int ensure_workspace(lf_ctx* c, int Bpad, size_t partA, size_t partB, size_t partR = 0) {
    if (Bpad > c->cap_B) {
        int nb = std::max(Bpad, c->cap_B * 2);
        LF_HIP(c, hipDeviceSynchronize());
        c->cap_B = 0;
        release(c->d_theta); release(c->d_out); release(c->d_outA); release(c->d_outB);
        release(c->d_wrec); release(c->d_wstat); release(c->d_wmode); release(c->d_wbase); release(c->d_slow);
        release_host(c->h_theta); release_host(c->h_out);
        LF_HIP(c, hipMalloc((void**)&c->d_theta, (size_t)nb * 16 * sizeof(double)));
        LF_HIP(c, hipMalloc((void**)&c->d_out, (size_t)nb * sizeof(double)));
        LF_HIP(c, hipMalloc((void**)&c->d_outA, (size_t)nb * sizeof(double)));
        LF_HIP(c, hipMalloc((void**)&c->d_outB, (size_t)nb * sizeof(double)));
        LF_HIP(c, hipMalloc((void**)&c->d_wrec, (size_t)nb * lf::REC * sizeof(double)));
        LF_HIP(c, hipMalloc((void**)&c->d_wstat, (size_t)nb * sizeof(int)));
        LF_HIP(c, hipMalloc((void**)&c->d_wmode, (size_t)nb * lf::MAXF * lf::WM * sizeof(int)));
        LF_HIP(c, hipMalloc((void**)&c->d_wbase, (size_t)nb * sizeof(double)));
        LF_HIP(c, hipMalloc((void**)&c->d_slow, ((size_t)nb + 1) * sizeof(int)));
        LF_HIP(c, hipMemset(c->d_slow, 0, ((size_t)nb + 1) * sizeof(int)));
        LF_HIP(c, hipHostMalloc((void**)&c->h_theta, (size_t)nb * 16 * sizeof(double), hipHostMallocDefault));
        LF_HIP(c, hipHostMalloc((void**)&c->h_out, (size_t)nb * 3 * sizeof(double), hipHostMallocDefault));
        c->cap_B = nb;
    }
    if (partA > c->cap_partA) {
        LF_HIP(c, hipDeviceSynchronize());
        c->cap_partA = 0;
        release(c->d_partA);
        LF_HIP(c, hipMalloc((void**)&c->d_partA, partA * sizeof(double)));
        c->cap_partA = partA;
    }
    if (partR > c->cap_partR) {
        LF_HIP(c, hipDeviceSynchronize());
        c->cap_partR = 0;
        release(c->d_partR);
        LF_HIP(c, hipMalloc((void**)&c->d_partR, partR * sizeof(double)));
        c->cap_partR = partR;
        c->parts_empty = false;
    }
    if (partB > c->cap_partB) {
        LF_HIP(c, hipDeviceSynchronize());
        c->cap_partB = 0;
        release(c->d_partB);
        LF_HIP(c, hipMalloc((void**)&c->d_partB, partB * sizeof(double)));
        c->cap_partB = partB;
        c->parts_empty = false;
    }
    return LF_OK;
}

struct Prof {
    lf_ctx* c;
    hipStream_t s;
    int kind;
    EventPair ep{};
    bool on, span;
    Prof(lf_ctx* c_, hipStream_t s_, int k) : c(c_), s(s_), kind(k) {
        // (option "profile_span" = n > 1, profiling level 1, one-launch evaluations: ONE pair around n consecutive launches)
        span = k == 1 && c->profiling == 1 && c->opt_profile_span > 1 && c->prof_span_ok;
        on = !span && c->prof_this && (c->profiling >= 2 || (c->profiling == 1 && k == 1));
        if (on) {
            hipEventCreate(&ep.a);
            hipEventCreate(&ep.b);
            ep.kind = kind;
            hipEventRecord(ep.a, s);
        }
        if (span && c->prof_pos == 0 && !c->span_open) {
            hipEventCreate(&c->span_ep.a);
            hipEventCreate(&c->span_ep.b);
            c->span_ep.kind = kind;
            c->span_ep.count = 0;
            hipEventRecord(c->span_ep.a, s);
            c->span_open = true;
        }
    }
    ~Prof() {
        if (on) {
            hipEventRecord(ep.b, s);
            c->events.push_back(ep);
        }
        if (span && c->span_open) {
            ++c->span_ep.count;
            if (c->span_ep.count >= std::min(c->opt_profile_span, c->opt_profile_every)) {
                hipEventRecord(c->span_ep.b, s);
                c->events.push_back(c->span_ep);
                c->span_open = false;
            }
        }
    }
};

// enqueue the three launches of one batched evaluation on `s`: prepare -> main (A and B) -> finalize
template <int VARIANT, int GI, bool CMP = false>
void launch_geo(lf_ctx* c, dim3 grid, lf::Tiling tl, int ntilesB, int twb, int nblkB, hipStream_t s,
                const lf::SrcArrays& sa, const lf::NodeArrays& na, int B, int nchA, int nchB,
                const lf::Rescue& rs = lf::Rescue{}, const lf::GridC& gc = lf::GridC{}) {
    using namespace lf;
    if (std::getenv("LF_DEBUG_OCC")) {
        int nb = -1;
        hipFuncAttributes fa{};
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, lf_main<VARIANT, GEOS[GI].st, GEOS[GI].tw, GEOS[GI].twb, CMP>, BLOCK, 0);
        hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&lf_main<VARIANT, GEOS[GI].st, GEOS[GI].tw, GEOS[GI].twb, CMP>));
        std::fprintf(stderr, "lf_main<%d,%d,%d,%d,%d>: %d workgroups per CU (occupancy API), %d VGPRs, %zu B LDS, grid %u\n", VARIANT,
                     GEOS[GI].st, GEOS[GI].tw, GEOS[GI].twb, (int)CMP, nb, fa.numRegs, fa.sharedSizeBytes, grid.x);
    }
    ZCells zc{};
    if (!CMP && ((VARIANT == LF_ZEVOL && c->zcells.nchC > 0) || VARIANT == LF_FIXCOMP)) {
        zc = c->zcells;
        // the per-source items are shared by a bounded number of workers (lf_kernels.h: lf_main), the cell workgroups come
        // after them: (chunk, tile of TW walkers)
        const int nsrc_wg = nchA * (tl.ntiles + tl.ntiles_s);
        // (one per CU: idle workers still cost ~0.6 us per 256 of them - 20.7 / 19.8 / 18.9 us with 4 / 2 / 1 per CU at 128
        // rows - and the rare walker that needs the sources has 490 items for 256 workers)
        zc.nwork = std::min(nsrc_wg, std::max(c->num_cu, 1));
        if (zc.nwork > 0) {
            grid.x = (unsigned)(nblkB + zc.nwork + zc.nchC * ((B + GEOS[GI].tw - 1) / GEOS[GI].tw));
            c->last_launch[4] = (int)grid.x;
        }
    }
    hipLaunchKernelGGL((lf_main<VARIANT, GEOS[GI].st, GEOS[GI].tw, GEOS[GI].twb, CMP>), grid, dim3(BLOCK), 0, s, c->kc, sa,
                       na, c->d_wrec, c->d_wmode, B, tl, nchA, ntilesB, twb, nblkB, c->d_partA, nchA, c->d_partB, nchB, rs, gc, zc);
}

// compressed-catalogue launches: the pseudo-sources are few, so only the small-tile geometries are instantiated
constexpr int CMP_GEOS[] = {8, 1, 4, 2};     // [0] is the default: few sources per lane, 16 walkers per grid workgroup
template <int VARIANT>
void launch_main_cmp(lf_ctx* c, int gi, dim3 grid, lf::Tiling tl, int ntilesB, int twb, int nblkB, hipStream_t s,
                     const lf::SrcArrays& sa, const lf::NodeArrays& na, int B, int nchA, int nchB, const lf::Rescue& rs,
                     const lf::GridC& gc) {
    switch (gi) {
        case 1: launch_geo<VARIANT, 1, true>(c, grid, tl, ntilesB, twb, nblkB, s, sa, na, B, nchA, nchB, rs, gc); break;
        case 4: launch_geo<VARIANT, 4, true>(c, grid, tl, ntilesB, twb, nblkB, s, sa, na, B, nchA, nchB, rs, gc); break;
        case 2: launch_geo<VARIANT, 2, true>(c, grid, tl, ntilesB, twb, nblkB, s, sa, na, B, nchA, nchB, rs, gc); break;
        default: launch_geo<VARIANT, 8, true>(c, grid, tl, ntilesB, twb, nblkB, s, sa, na, B, nchA, nchB, rs, gc);
    }
}

template <int VARIANT>
void launch_main(lf_ctx* c, int gi, dim3 grid, lf::Tiling tl, int ntilesB, int twb, int nblkB, hipStream_t s,
                 const lf::SrcArrays& sa, const lf::NodeArrays& na, int B, int nchA, int nchB, const lf::Rescue& rs) {
    switch (gi) {
        case 0: launch_geo<VARIANT, 0>(c, grid, tl, ntilesB, twb, nblkB, s, sa, na, B, nchA, nchB, rs); break;
        case 1: launch_geo<VARIANT, 1>(c, grid, tl, ntilesB, twb, nblkB, s, sa, na, B, nchA, nchB, rs); break;
        case 2: launch_geo<VARIANT, 2>(c, grid, tl, ntilesB, twb, nblkB, s, sa, na, B, nchA, nchB, rs); break;
        case 3: launch_geo<VARIANT, 3>(c, grid, tl, ntilesB, twb, nblkB, s, sa, na, B, nchA, nchB, rs); break;
        case 4: launch_geo<VARIANT, 4>(c, grid, tl, ntilesB, twb, nblkB, s, sa, na, B, nchA, nchB, rs); break;
        case 5: launch_geo<VARIANT, 5>(c, grid, tl, ntilesB, twb, nblkB, s, sa, na, B, nchA, nchB, rs); break;
        case 6: launch_geo<VARIANT, 6>(c, grid, tl, ntilesB, twb, nblkB, s, sa, na, B, nchA, nchB, rs); break;
        case 7: launch_geo<VARIANT, 7>(c, grid, tl, ntilesB, twb, nblkB, s, sa, na, B, nchA, nchB, rs); break;
        default: launch_geo<VARIANT, 8>(c, grid, tl, ntilesB, twb, nblkB, s, sa, na, B, nchA, nchB, rs);
    }
}

// FREE variant, real catalogue, catalogues large enough to fill it: prepare -> lf_free (persistent 512-thread workgroups,
// pieces A and B, lf_free.h) -> finalize
// Source-chunk size of the persistent FREE kernel (lf_free.h), for walkers that are summed over the sources: sources per
// lane 8 (the table lookup is shared by 8 terms) once the catalogue gives 4096-source chunks enough to go round, else 4
// (measured on a warmed-up device, cells off, 128 rows, us: N = 2.5e5 58.1 -> 55.7, 1e5 45.7 -> 40.2 going from ST = 8
// to 4).  free_st overrides (tuning runs).
struct FreeShape {
    int st;
    int64_t items_per_tile;      // source chunks + grid chunks of 512 nodes' worth (the old item count: the crossover rule
                                 // for contexts without cells was measured in these units)
};
FreeShape free_shape(const lf_ctx* c) {
    using namespace lf;
    FreeShape fs;
    const int64_t chunks8 = (c->N + 8 * (int64_t)PB - 1) / (8 * (int64_t)PB);
    fs.st = c->opt_free_st ? (int)c->opt_free_st : (chunks8 >= 74 ? 8 : 4);
    const int64_t nchB = c->opt_skip_grid ? 0 : (c->nnodes + PB - 1) / PB;
    fs.items_per_tile = (c->N + (int64_t)PB * fs.st - 1) / ((int64_t)PB * fs.st) + nchB * 2;
    return fs;
}

// The number of groups of 8 workgroups (one per XCD under round-robin placement) lf_free<ST> is launched with: no more
// than the chip holds at once, no more than there are items.
template <int ST>
int free_groups(lf_ctx* c, int slot, int ntiles, int nchA, int nchB, int nchC) {
    using namespace lf;
    if (c->slots_free[slot] == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, lf_free<ST, false>, PB, 0) != hipSuccess || nb < 1) nb = 1;
        c->slots_free[slot] = nb * std::max(c->num_cu, 1);
        if (std::getenv("LF_DEBUG_OCC")) {
            hipFuncAttributes at{};
            hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&lf_free<ST, false>));
            std::fprintf(stderr, "lf_free<%d>: %d workgroups per CU (occupancy API), %d VGPRs, %zu B LDS\n", ST, nb, at.numRegs, at.sharedSizeBytes);
        }
    }
    // (at most VF workgroups per tile: the cells' and the grid's chunks are dealt to VF virtual workgroups, lf_free.h)
    const int64_t per_tile = std::min<int64_t>(VF / 8, ((int64_t)nchA + nchB + nchC + 7) / 8);
    return (int)std::max<int64_t>(1, std::min<int64_t>(c->slots_free[slot] / 8, (int64_t)ntiles * std::max<int64_t>(per_tile, 1)));
}

// lf_free's static deal (lf_free.h: DEAL_*): flux bins, then cell chunks, each to the virtual workgroup that would be done
// first - a bin costs a wave 8 units, a cell chunk 3 (tools/stamps_fused.py), and the ranks of the younger half are
// counted 8 units behind (swept on one box, tools/deal_sweep.sh: 13.4 us per 128-row evaluation at 8-10, 13.75 at 0-6,
// 14.0 at 16; the arithmetic deal 15.25; at 256 rows, where a workgroup serves an elder and a younger rank, all within 2 %).  Bins that a source-sharded rank does not integrate (grid_share) cost nothing.  The table depends
// on the context (numbers of cell chunks and bins, grid share) only - never on the batch.  No table (the arithmetic deal):
// no bins, or more entries than the kernel keeps in LDS.
// (host only; also behind lf_deal_table for the CPU tests)
static std::vector<int> make_deal(int nchC, int nbq, int grid_part, int grid_parts) {
    using namespace lf;
    std::vector<int> load(VF), cnt_c(VF, 0), cnt_b(VF, 0), own_c(nchC), own_b(nbq);
    const int cost_h = std::getenv("LF_DEAL_H") ? std::atoi(std::getenv("LF_DEAL_H")) : 8;       // (tuning runs: tools/deal_sweep.sh)
    const int cost_b = std::getenv("LF_DEAL_B") ? std::atoi(std::getenv("LF_DEAL_B")) : 8;
    for (int v = 0; v < VF; ++v) load[v] = v >= VF / 2 ? cost_h : 0;
    auto next = [&]() { return (int)(std::min_element(load.begin(), load.end()) - load.begin()); };      // (ties: the lowest rank)
    for (int b = 0; b < nbq; ++b) {
        const int v = next();
        own_b[b] = v;
        ++cnt_b[v];
        load[v] += grid_parts > 1 && b % grid_parts != grid_part ? 0 : cost_b;
    }
    for (int i = 0; i < nchC; ++i) {
        const int v = next();
        own_c[i] = v;
        ++cnt_c[v];
        load[v] += 3;
    }
    std::vector<int> t(DEAL_LIST + nchC + nbq);
    t[0] = 0;
    t[DEAL_BINS] = 0;
    for (int v = 0; v < VF; ++v) {
        t[v + 1] = t[v] + cnt_c[v];
        t[DEAL_BINS + v + 1] = t[DEAL_BINS + v] + cnt_b[v];
    }
    std::vector<int> at_c(t.begin(), t.begin() + VF), at_b(t.begin() + DEAL_BINS, t.begin() + DEAL_BINS + VF);
    for (int i = 0; i < nchC; ++i) t[DEAL_LIST + at_c[own_c[i]]++] = i;
    for (int b = 0; b < nbq; ++b) t[DEAL_LIST + nchC + at_b[own_b[b]]++] = b;
    return t;
}

int ensure_deal(lf_ctx* c, int nchC, int nbq, hipStream_t s) {
    using namespace lf;
    const int64_t key = nbq <= 0 || DEAL_LIST + nchC + nbq > DEAL_MAX || std::getenv("LF_NO_DEAL")      // (the variable: A/B runs)
                            ? 0 : 1 + nchC + 4096ll * nbq + (4096ll * 4096) * (c->kc.grid_part + 4096ll * c->kc.grid_parts);
    if (key == c->deal_key) return LF_OK;
    if (key > 0) {
        const std::vector<int> t = make_deal(nchC, nbq, c->kc.grid_part, c->kc.grid_parts);
        if (!c->d_deal) LF_HIP(c, hipMalloc((void**)&c->d_deal, (size_t)DEAL_MAX * sizeof(int)));
        if (c->any_enqueued) LF_HIP(c, hipStreamSynchronize(c->last_stream));        // (a launch may still be reading the old table)
        LF_HIP(c, hipMemcpy(c->d_deal, t.data(), t.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    c->deal_key = key;
    return LF_OK;
}

template <int ST>
void launch_free(lf_ctx* c, int slot, int B, int ntiles, const lf::SrcArrays& sa, const lf::NodeArrays& na, lf::FreeArgs fa, hipStream_t s,
                 bool fused, const lf::StepArgs* sp = nullptr, const lf::AcceptArgs* ap = nullptr) {
    // (fa.nslot is set by the caller from free_groups(), which is also what the grid is made of here)
    using namespace lf;
    const dim3 grid((unsigned)(8 * fa.tile_stride));
    const int info[8] = {ST, PTW, PTW, fused ? 3 : 2, (int)grid.x, fa.nchA, fa.nchB, B};
    std::memcpy(c->last_launch, info, sizeof(info));
    if (fused && sp) hipLaunchKernelGGL((lf_free_step<ST>), grid, dim3(PB), 0, s, c->kc, sa, na, fa, *sp, *ap);      // the sampler's half-step
    else if (fused) hipLaunchKernelGGL((lf_free<ST, false, true>), grid, dim3(PB), 0, s, c->kc, sa, na, c->d_wrec, c->d_wmode, fa);
    else if (c->kc.forms) hipLaunchKernelGGL((lf_free<ST, true>), grid, dim3(PB), 0, s, c->kc, sa, na, c->d_wrec, c->d_wmode, fa);
    else hipLaunchKernelGGL((lf_free<ST, false>), grid, dim3(PB), 0, s, c->kc, sa, na, c->d_wrec, c->d_wmode, fa);
}

int enqueue_free(lf_ctx* c, const double* d_theta, int B, double* d_out, double* d_outA, double* d_outB, hipStream_t s,
                 const lf::StepArgs& sp, const lf::AcceptArgs& ap) {
    using namespace lf;
    const int ntiles = (B + PTW - 1) / PTW;
    const FreeShape fs = free_shape(c);
    const int st = fs.st, slot = fs.st == 8 ? 2 : (fs.st == 4 ? 1 : 0);
    ChunkTable* ct = nullptr;
    int rc = get_chunks(c, c->chunks_free, c->field_ind, PB * st, &ct, c->h_x.data(), st);
    if (rc != LF_OK) return rc;
    const int nchA = ct->n;
    const bool gq = c->gridq.built && c->opt_grid_shortcut && !c->opt_skip_grid;      // piece B over flux bins (one bin per chunk)
    const int nchB = c->opt_skip_grid ? 0 : (gq ? c->gridq.nb : (c->nnodes + 63) / 64);        // chunks of 64 nodes: a wave's lanes
    const int nchC = c->kc.cells ? c->ncchunk : 0;
    const int g8 = st == 8 ? free_groups<8>(c, slot, ntiles, nchA, nchB, nchC)
                 : st == 4 ? free_groups<4>(c, slot, ntiles, nchA, nchB, nchC) : free_groups<2>(c, slot, ntiles, nchA, nchB, nchC);
    // the grid's and the cells' partial sums: one per (walker, workgroup serving the walker's tile)
    const int nslot = VF;                  // one per (walker, VIRTUAL workgroup of its tile): independent of the batch
    rc = ensure_workspace(c, B, (size_t)B * std::max(nchA, 1), (size_t)B * nslot, (size_t)B * nslot);
    if (rc != LF_OK) return rc;
    if ((rc = ensure_deal(c, nchC, gq ? c->gridq.nb : 0, s)) != LF_OK) return rc;
    if (ntiles * QSTRIDE > c->cap_queue) {
        LF_HIP(c, hipDeviceSynchronize());
        release(c->d_queue);
        c->cap_queue = 0;
        const int cap = std::max(2 * ntiles * QSTRIDE, 1024);
        LF_HIP(c, hipMalloc((void**)&c->d_queue, (size_t)cap * sizeof(int)));
        c->cap_queue = cap;
        c->queue_zero = false;
    }
    // One launch instead of three (lf_free.h: FUSED) for the plain evaluation; the sampler's propose / accept steps, the
    // two-piece diagnostics, the census and the profile of every launch keep lf_prepare and lf_finalize.
    // ... and so is the sampler's half-step (proposal in the prologue, accept / reject by the tile's finishing workgroup)
    const bool stepf = sp.enabled && ap.enabled && c->opt_fuse_step;
    const bool fused = c->opt_fuse && (stepf || (!sp.enabled && !ap.enabled)) && !d_outA && !d_outB && d_out && !c->kc.forms && c->profiling < 2 &&
                       nchA + nchB > 0;
    if (c->any_enqueued && c->last_stream != s) LF_HIP(c, hipStreamSynchronize(c->last_stream));
    c->last_stream = s;
    c->any_enqueued = true;
    c->prof_pos = c->prof_tick++ % c->opt_profile_every;
    c->prof_this = c->profiling > 0 && c->prof_pos == 0;
    c->prof_span_ok = fused;
    if (fused && !c->queue_zero) {
        // the tiles' counters start at zero; a fused launch leaves them so, lf_prepare zeroes them for the three-launch form
        LF_HIP(c, hipMemsetAsync(c->d_queue, 0, (size_t)c->cap_queue * sizeof(int), s));
        c->queue_zero = true;
    }
    // the polling hand-over (lf_free.h: PART_EMPTY) wants every slot of the two partial-sum buffers empty; a fused launch
    // leaves them so, every other form leaves sums behind
    const bool poll = fused && c->opt_poll && nslot <= 64;
    if (poll && !c->parts_empty) {
        static_assert((PART_EMPTY >> 32) == (PART_EMPTY & 0xffffffffull), "filled by 32-bit words");
        LF_HIP(c, hipMemsetD32Async((hipDeviceptr_t)c->d_partB, (int)(PART_EMPTY & 0xffffffffull), c->cap_partB * 2, s));
        LF_HIP(c, hipMemsetD32Async((hipDeviceptr_t)c->d_partR, (int)(PART_EMPTY & 0xffffffffull), c->cap_partR * 2, s));
        if (!c->d_err) {
            LF_HIP(c, hipMalloc((void**)&c->d_err, sizeof(int)));
            LF_HIP(c, hipMemsetAsync(c->d_err, 0, sizeof(int), s));
        }
        c->parts_empty = true;
    }
    if (!poll) c->parts_empty = false;
    if (!fused) {
        c->queue_zero = false;
        Prof p(c, s, 0);
        hipLaunchKernelGGL(lf_prepare, dim3((B + 7) / 8), dim3(64), 0, s, c->kc, sp, d_theta, B, c->d_wrec,
                           c->d_wstat, c->d_wmode, c->d_wbase, (int*)nullptr, c->d_queue, ntiles * QSTRIDE);
    }
    const SrcArrays sa{c->d_lum, c->d_a1, c->d_P, c->d_U, nullptr, ct->d_start, ct->d_len, ct->d_field, ct->d_keys, nullptr};
    const NodeArrays na{c->d_G, c->d_PG, c->d_W, c->d_a3, c->d_a4, c->d_a4min, c->nnodes};
    FreeArgs fa{B, ntiles, nchA, nchB, nslot, g8, (int)c->opt_skip_grid, c->d_queue, c->d_partA, c->d_partB,
                c->d_cells, c->d_nodes8, c->deal_key > 0 ? c->d_deal : nullptr, c->d_cc_len, c->d_cc_field, nchC, c->d_partR, c->d_wstat,
                d_theta, d_out, c->d_wrec, c->d_wmode, c->d_wstat, c->d_wbase,
                gq ? c->gridq.d_rec : nullptr, gq ? c->gridq.d_omega : nullptr, gq ? c->gridq.d_rows : nullptr, gq ? c->gridq.nb : 0,
                poll ? 1 : 0, c->d_err};
    {
        Prof p(c, s, 1);
        if (nchA + nchB > 0) {
            const StepArgs* psp = fused && stepf ? &sp : nullptr;
            if (st == 8) launch_free<8>(c, slot, B, ntiles, sa, na, fa, s, fused, psp, &ap);
            else if (st == 4) launch_free<4>(c, slot, B, ntiles, sa, na, fa, s, fused, psp, &ap);
            else launch_free<2>(c, slot, B, ntiles, sa, na, fa, s, fused, psp, &ap);
        }
    }
    if (!fused) {
        Prof p(c, s, 3);
        const int nB = nchB > 0 ? nslot : 0, nC = nchC > 0 ? nslot : 0;
        hipLaunchKernelGGL(lf_finalize, dim3(B), dim3(64), 0, s, c->d_partA, nchA, nchA, c->d_partB, nB, nB,
                           nC > 0 ? (const double*)c->d_partR : (const double*)nullptr, nC, (int)STAT_CELLS, c->d_wstat, c->d_wbase, B, ap,
                           d_out, d_outA, d_outB, (int*)nullptr, 0);
    }
    LF_HIP(c, hipGetLastError());
    return LF_OK;
}

// The z-evolving and fixed-completeness variants in persistent workgroups (lf_pers.h): one launch for a plain evaluation,
// lf_prepare / lf_pers / lf_finalize for the sampler's steps and the two-piece diagnostics.
template <int VARIANT>
int enqueue_pers_v(lf_ctx* c, const double* d_theta, int B, double* d_out, double* d_outA, double* d_outB, hipStream_t s,
                   const lf::StepArgs& sp, const lf::AcceptArgs& ap) {
    using namespace lf;
    const int ntiles = (B + PTW - 1) / PTW;
    if (c->slots_pers == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, lf_pers<VARIANT, true>, PB, 0) != hipSuccess || nb < 1) nb = 1;
        c->slots_pers = std::min(nb, 2) * std::max(c->num_cu, 1);      // (two per CU: what the kernel is sized for)
    }
    const int nchB = c->opt_skip_grid ? 0 : c->nch4;
    const int nchC = VARIANT == LF_ZEVOL && c->kc.cells ? (c->ncell + 63) / 64 : 0;
    // groups of 8 workgroups: no more than the chip holds at once, no more than there is work for
    const int64_t per_tile = std::max<int64_t>(1, std::min<int64_t>(VF / 8, ((int64_t)nchB + nchC + 7) / 8));
    const int g8 = (int)std::max<int64_t>(1, std::min<int64_t>(c->slots_pers / 8, (int64_t)ntiles * per_tile));
    const int nslot = VF;                  // one per (walker, VIRTUAL workgroup of its tile), lf_free.h
    int rc = ensure_workspace(c, B, (size_t)B * nslot, (size_t)B * nslot, (size_t)B * nslot);
    if (rc != LF_OK) return rc;
    if (ntiles * QSTRIDE > c->cap_queue) {
        LF_HIP(c, hipDeviceSynchronize());
        release(c->d_queue);
        c->cap_queue = 0;
        const int cap = std::max(2 * ntiles * QSTRIDE, 1024);
        LF_HIP(c, hipMalloc((void**)&c->d_queue, (size_t)cap * sizeof(int)));
        c->cap_queue = cap;
        c->queue_zero = false;
    }
    const bool stepf = sp.enabled && ap.enabled && c->opt_fuse_step;       // the sampler's half-step: one launch too
    const bool fused = c->opt_fuse && (stepf || (!sp.enabled && !ap.enabled)) && !d_outA && !d_outB && d_out && c->profiling < 2;
    if (c->any_enqueued && c->last_stream != s) LF_HIP(c, hipStreamSynchronize(c->last_stream));
    c->last_stream = s;
    c->any_enqueued = true;
    c->prof_pos = c->prof_tick++ % c->opt_profile_every;
    c->prof_this = c->profiling > 0 && c->prof_pos == 0;
    c->prof_span_ok = fused;
    if (fused && !c->queue_zero) {
        LF_HIP(c, hipMemsetAsync(c->d_queue, 0, (size_t)c->cap_queue * sizeof(int), s));
        c->queue_zero = true;
    }
    const bool poll = fused && c->opt_poll;       // (lf_free.h: PART_EMPTY; see enqueue_free)
    if (poll && !c->parts_empty) {
        LF_HIP(c, hipMemsetD32Async((hipDeviceptr_t)c->d_partB, (int)(PART_EMPTY & 0xffffffffull), c->cap_partB * 2, s));
        LF_HIP(c, hipMemsetD32Async((hipDeviceptr_t)c->d_partR, (int)(PART_EMPTY & 0xffffffffull), c->cap_partR * 2, s));
        if (!c->d_err) {
            LF_HIP(c, hipMalloc((void**)&c->d_err, sizeof(int)));
            LF_HIP(c, hipMemsetAsync(c->d_err, 0, sizeof(int), s));
        }
        c->parts_empty = true;
    }
    if (!poll) c->parts_empty = false;
    if (!fused) {
        Prof p(c, s, 0);
        hipLaunchKernelGGL(lf_prepare, dim3((B + 7) / 8), dim3(64), 0, s, c->kc, sp, d_theta, B, c->d_wrec,
                           c->d_wstat, c->d_wmode, c->d_wbase, (int*)nullptr, c->d_queue, 0);
    }
    const PersArgs pa{B, ntiles, nchB, nchC, c->ncell, nslot, g8, c->d_queue, c->d_partA, c->d_partB, c->d_partR, c->d_nodes4, c->d_zcol,
                      c->d_cells, c->d_lum, c->d_a1, c->d_P, c->d_U, d_theta, d_out, c->d_wrec, c->d_wmode, c->d_wstat, poll ? 1 : 0, c->d_err};
    {
        Prof p(c, s, 1);
        const dim3 grid((unsigned)(8 * g8));
        const int info[8] = {0, PTW, PTW, fused ? 5 : 4, (int)grid.x, nchC, nchB, B};
        std::memcpy(c->last_launch, info, sizeof(info));
        if (fused && stepf) hipLaunchKernelGGL((lf_pers_step<VARIANT>), grid, dim3(PB), 0, s, c->kc, pa, sp, ap);
        else if (fused) hipLaunchKernelGGL((lf_pers<VARIANT, true>), grid, dim3(PB), 0, s, c->kc, pa);
        else hipLaunchKernelGGL((lf_pers<VARIANT, false>), grid, dim3(PB), 0, s, c->kc, pa);
    }
    if (!fused) {
        Prof p(c, s, 3);
        hipLaunchKernelGGL(lf_finalize, dim3(B), dim3(64), 0, s, c->d_partA, nslot, nslot, c->d_partB, nchB > 0 ? nslot : 0, nslot,
                           nchC > 0 ? (const double*)c->d_partR : (const double*)nullptr, nchC > 0 ? nslot : 0, (int)STAT_CELLS, c->d_wstat,
                           c->d_wbase, B, ap, d_out, d_outA, d_outB, (int*)nullptr, VARIANT == LF_FIXCOMP ? (int)STAT_SLOW : 0);
    }
    LF_HIP(c, hipGetLastError());
    return LF_OK;
}

int enqueue(lf_ctx* c, const double* d_theta, int B, double* d_out, double* d_outA, double* d_outB,
            hipStream_t s, const lf::StepArgs* step = nullptr, const lf::AcceptArgs* accept = nullptr) {
    using namespace lf;
    StepArgs sp{};
    AcceptArgs ap{};
    if (step) sp = *step;
    if (accept) ap = *accept;
    // the persistent kernel takes the free variant's direct path whenever the catalogue can fill it and no launch
    // geometry of lf_main was asked for explicitly
    // (source-sharded ranks whose piece B goes over flux bins split the BINS: every rank must then be in lf_free, whatever
    // the size of its shard - lf_main has no bins)
    const bool shared_bins = c->kc.variant == LF_FREE && c->kc.grid_parts > 1 && c->gridq.built && c->opt_grid_shortcut && !c->opt_skip_grid;
    if (c->kc.variant == LF_FREE && c->opt_persistent && c->opt_geometry < 0 && c->opt_walker_tile == 0 && !c->opt_taper &&
        !(c->opt_compress && c->cmp.built) && (c->N >= 8192 || c->kc.cells || c->opt_persistent == 2 || shared_bins)) {
        // Measured crossover (tools/time_parts.py on a warmed-up device, lf_main / lf_free in us; 128 rows: N = 5e4 33 / 37,
        // 7e4 36 / 38, 1e5 40 / 40, 1.8e5 49 / 49, 2.5e5 60 / 56, 5e5 95 / 72, 1e6 168 / 112; N = 1e6 with 16 / 32 / 64 /
        // 256 rows: 29 / 79, 50 / 56, 91 / 72, 327 / 205; N = 1e5 with 512 rows: 127 / 120): the persistent kernel wins
        // once every one of its ~512 workgroups gets about four items or more; below that its coarse items cost more
        // than its tables save.  opt_persistent = 2 or an explicit free_st force it (tests, tuning runs).
        // With the catalogue's cells (the normal case) piece A costs next to nothing and the kernel takes 22-34 us up to
        // 128 rows whatever N is (the grid integral): it wins from N x rows ~ 1e7 (lf_main / lf_free, 128 rows: N = 5e4
        // 32 / 34, 1e5 40 / 33, 2.5e5 60 / 33; N = 1e6 with 16 / 32 / 64 rows: 30 / 24, 50 / 25, 94 / 30; N = 1e5 with
        // 32 / 512 rows: 16 / 25, 126 / 88).  A plain evaluation is ONE launch in lf_free against three: the period of
        // back-to-back evaluations (tools/sweep_small.sh, lf_main / lf_free in us, 128 rows: N = 1e4 29.3 / 26.1, 3e4 37.0 /
        // 26.4, 7.8e4 44.4 / 26.9; 16 rows, where the Python caller binds both: 17-18 / 18.5-18.9) moves the crossover
        // down to N x rows ~ 1.2e6 from 32 rows.
        const int64_t ntiles = (B + PTW - 1) / PTW;
        const int64_t items = free_shape(c).items_per_tile * ntiles;
        const bool plain = !sp.enabled && !ap.enabled && !d_outA && !d_outB && c->opt_fuse;
        // (an evaluation over cells costs 12-18 us whatever N and B are: always - the sampler's steps and the diagnostics
        // included, so that a row has the same bits whichever entry point evaluates it)
        const bool wins = c->kc.cells ? true : items >= 4 * 2 * (int64_t)std::max(c->num_cu, 1);
        (void)plain;
        if (wins || shared_bins || c->opt_persistent == 2 || c->opt_free_st)
            return enqueue_free(c, d_theta, B, d_out, d_outA, d_outB, s, sp, ap);
    }
    // z-evolving (with its cells, grid by columns) and fixed completeness: the persistent kernel of lf_pers.h, unless a launch
    // geometry of lf_main was asked for, the census is on (its counters live in lf_main), or the catalogue is compressed
    if (c->kc.variant != LF_FREE && c->opt_persistent && c->opt_geometry < 0 && c->opt_walker_tile == 0 && !c->opt_taper &&
        !(c->opt_compress && c->cmp.built) && !c->kc.forms && c->d_nodes4 &&
        (c->kc.variant == LF_FIXCOMP || (c->kc.cells && c->kc.zgrid_cols && c->kc.S <= PERS_MAXS))) {
        return c->kc.variant == LF_FIXCOMP ? enqueue_pers_v<LF_FIXCOMP>(c, d_theta, B, d_out, d_outA, d_outB, s, sp, ap)
                                           : enqueue_pers_v<LF_ZEVOL>(c, d_theta, B, d_out, d_outA, d_outB, s, sp, ap);
    }
    c->parts_empty = false;           // (lf_main leaves sums in the partial-sum buffers: the polling hand-over refills them)
    // compressed catalogue: piece A over the weighted pseudo-sources, plus rescue workgroups over the real one
    const bool cmp = c->opt_compress && c->cmp.built && c->kc.variant != LF_FIXCOMP;
    int gi = pick_geometry(c, B);
    if (cmp) {
        bool ok = false;
        for (int g : CMP_GEOS) ok = ok || g == gi;
        if (!ok || c->opt_geometry < 0) gi = CMP_GEOS[0];
    }
    const Geo geo = GEOS[gi];
    ChunkTable *ct = nullptr, *ctd = nullptr;
    // (ZEVOL, real catalogue: the chunk keys carry the widest lane of z-neighbours, lane = geo.st consecutive sources -
    // what the local form of the term needs to know, lf_kernels.h: srcsum_body)
    const double* hz = c->kc.variant == LF_ZEVOL && !c->h_x.empty() ? c->h_x.data() : nullptr;
    int rc = cmp ? get_chunks(c, c->cmp.chunks, c->cmp.field_ind, geo.st * BLOCK, &ct)
                 : get_chunks(c, c->chunks, c->field_ind, geo.st * BLOCK, &ct, hz, hz ? geo.st : 0);
    if (rc != LF_OK) return rc;
    if (cmp && (rc = get_chunks(c, c->chunks, c->field_ind, geo.st * BLOCK, &ctd, hz, hz ? geo.st : 0)) != LF_OK) return rc;
    const int nchA = ct->n;
    const int nchD = cmp ? ctd->n : 0;
    // rescue workgroups leave at once unless a walker was flagged; still, each costs a dispatch slot: scale with B
    const int nresc = cmp ? std::min(nchD, std::min(1024, std::max(128, 2 * B))) : 0;
    // (the compressed grid's chunks are bins of its own: ranks that split piece B keep to the lattice's granules)
    const bool cgrid = cmp && c->gridc.built && c->opt_compress_grid && !c->opt_skip_grid && c->kc.grid_parts <= 1;
    const int nchB = c->opt_skip_grid ? 0 : (cgrid ? (c->gridc.nb + 15) / 16 : (c->nnodes + BLOCK - 1) / BLOCK);
    // ZEVOL on the real catalogue: walkers lf_prepare flags STAT_CELLS are summed over the cells in redshift (partR)
    const int nchC = !cmp && c->kc.variant == LF_ZEVOL && c->kc.cells ? c->ncchunk : 0;
    rc = ensure_workspace(c, B, (size_t)B * std::max(nchA, 1), (size_t)B * std::max(nchB, 1), (size_t)B * std::max(nchD, nchC));
    if (rc != LF_OK) return rc;
    c->zcells = ZCells{c->d_cells, c->d_cc_start, c->d_cc_len, c->d_cc_field, nchC, c->d_partR, c->d_wstat, 0};
    // the workspace is shared by consecutive calls: order a stream switch behind the previous work
    if (c->any_enqueued && c->last_stream != s) LF_HIP(c, hipStreamSynchronize(c->last_stream));
    c->last_stream = s;
    c->any_enqueued = true;
    c->prof_pos = c->prof_tick++ % c->opt_profile_every;
    c->prof_this = c->profiling > 0 && c->prof_pos == 0;
    c->prof_span_ok = false;

    {
        Prof p(c, s, 0);
        hipLaunchKernelGGL(lf_prepare, dim3((B + 7) / 8), dim3(64), 0, s, c->kc, sp, d_theta, B, c->d_wrec,
                           c->d_wstat, c->d_wmode, c->d_wbase, cmp ? c->d_slow : nullptr, c->d_queue, 0);
    }
    const SrcArrays sd{c->d_lum, c->d_a1, c->d_P, c->d_U, nullptr, ct->d_start, ct->d_len, ct->d_field, ct->d_keys, c->d_queue};
    SrcArrays sa = sd;
    Rescue rs{};
    GridC gc{};
    if (cgrid) {
        const auto& g = c->gridc;
        gc = GridC{g.d_U, g.d_A4, g.d_row0, g.d_nrows, g.d_off, g.d_omega, g.d_L, g.d_PGL, g.nb, c->kc.S};
    }
    if (cmp) {
        sa = SrcArrays{c->cmp.d_lum, c->cmp.d_a1, c->cmp.d_lum, c->cmp.d_U, c->cmp.d_W, ct->d_start, ct->d_len, ct->d_field, ct->d_keys, nullptr};
        rs.sd = sd;
        rs.sd.chunk_start = ctd->d_start;
        rs.sd.chunk_len = ctd->d_len;
        rs.sd.chunk_field = ctd->d_field;
        rs.sd.chunk_keys = ctd->d_keys;
        rs.slow_count = c->d_slow;
        rs.slow_list = c->d_slow + 1;
        rs.partR = c->d_partR;
        rs.nchD = nchD;
        rs.nresc = nresc;
    }
    NodeArrays na{c->d_G, c->d_PG, c->d_W, c->d_a3, c->d_a4, c->d_a4min, c->nnodes};
    {
        Prof p(c, s, 1);
        int tw = geo.tw, twb = geo.twb;
        if (cmp && nchB > 0) {
            // the grid integral is all the work there is: walkers per grid workgroup such that the launch
            // still has ~2000 workgroups (8 per CU), as many as the instantiation allows otherwise
            // (the compressed grid has only a few bin groups, each workgroup little work per walker: fewer, fatter ones)
            const int64_t want = cgrid ? 768 : 2048;
            int t = 1;
            while (t < geo.twb && (int64_t)nchB * ((B + 2 * t - 1) / (2 * t)) >= want) t *= 2;
            twb = t;
        }
        if (cmp && nchA > 0) {
            // the pseudo-sources are a handful of chunks: per workgroup the walker loop is a chain of dependent
            // terms (latency, not issue), so small batches get few walkers per workgroup and many workgroups
            int t = 1;
            while (t < geo.tw && (int64_t)nchA * ((B + 2 * t - 1) / (2 * t)) >= 1024) t *= 2;
            tw = t;
        }
        if (c->opt_walker_tile > 0) {
            tw = (int)std::min<int64_t>(c->opt_walker_tile, geo.tw);
            twb = (int)std::min<int64_t>(c->opt_walker_tile, geo.twb);
        }
        // tapered tiling: the last ~1/8 of the walkers go in quarter-size tiles that are dispatched last
        Tiling tl{tw, 0, B, std::max(1, tw / 4), 0};
        if (c->opt_taper && !cmp && B >= 2 * tw && tw >= 4) {
            const int tail = std::max(tw, ((B / 8 + tw - 1) / tw) * tw);       // whole big tiles' worth of walkers
            tl.B1 = ((B - tail) / tw) * tw;
        }
        tl.ntiles = (tl.B1 + tw - 1) / tw;
        tl.ntiles_s = (B - tl.B1 + tl.tws - 1) / tl.tws;
        const int ntilesB = (B + twb - 1) / twb;
        const int nblkB = nchB * ntilesB;
        // 1-D grid: B items, big A items, small A items, rescue workgroups
        dim3 grid((unsigned)(nblkB + nchA * (tl.ntiles + tl.ntiles_s) + nresc));
        const int info[8] = {geo.st, geo.tw, geo.twb, cmp ? 1 : 0, (int)grid.x, nchA, nchB, B};
        std::memcpy(c->last_launch, info, sizeof(info));
        if (grid.x > 0) {
            if (cmp) {
                if (c->kc.variant == LF_FREE) launch_main_cmp<LF_FREE>(c, gi, grid, tl, ntilesB, twb, nblkB, s, sa, na, B, nchA, nchB, rs, gc);
                else launch_main_cmp<LF_ZEVOL>(c, gi, grid, tl, ntilesB, twb, nblkB, s, sa, na, B, nchA, nchB, rs, gc);
            } else switch (c->kc.variant) {
                case LF_FREE: launch_main<LF_FREE>(c, gi, grid, tl, ntilesB, twb, nblkB, s, sa, na, B, nchA, nchB, rs); break;
                case LF_FIXCOMP: launch_main<LF_FIXCOMP>(c, gi, grid, tl, ntilesB, twb, nblkB, s, sa, na, B, nchA, nchB, rs); break;
                default: launch_main<LF_ZEVOL>(c, gi, grid, tl, ntilesB, twb, nblkB, s, sa, na, B, nchA, nchB, rs);
            }
        }
    }
    {
        Prof p(c, s, 3);
        hipLaunchKernelGGL(lf_finalize, dim3(B), dim3(64), 0, s, c->d_partA, nchA, nchA, c->d_partB, nchB, nchB,
                           cmp || nchC > 0 ? c->d_partR : nullptr, cmp ? nchD : nchC, (int)(cmp ? STAT_SLOW : STAT_CELLS), c->d_wstat,
                           c->d_wbase, B, ap, d_out, d_outA, d_outB, cmp ? c->d_slow : nullptr,
                           !cmp && c->kc.variant == LF_FIXCOMP && nchA > 0 ? (int)STAT_SLOW : 0);
    }
    LF_HIP(c, hipGetLastError());
    return LF_OK;
}

void free_cmp(CompressedCat& cc) {
    for (auto& kv : cc.chunks) {
        hipFree(kv.second.d_start);
        hipFree(kv.second.d_len);
        hipFree(kv.second.d_field);
        hipFree(kv.second.d_keys);
    }
    cc.chunks.clear();
    double* bufs[] = {cc.d_lum, cc.d_a1, cc.d_U, cc.d_W};
    for (double* b : bufs)
        if (b) hipFree(b);
    cc = CompressedCat{};
}

// Build the compressed catalogue from the per-source tables already in HBM (lf_compress.h).  FREE: key = logf_i,
// weight 1; ZEVOL: key = z_i, weight 10^(lum_i - 42).
int build_compressed(lf_ctx* c) {
    using namespace lf;
    if (c->cmp.built) return LF_OK;
    if (c->kc.variant == LF_FIXCOMP) return LF_OK;          // piece A is closed-form already
    const int64_t N = c->N;
    std::vector<double> key((size_t)N), wt;
    if (N) LF_HIP(c, hipMemcpy(key.data(), c->d_a1, (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
    lfc::Model m{};
    if (c->kc.variant == LF_FREE) {
        m.kind = 0;
        m.fc_ratio = c->kc.fc_ratio;
        m.alpha_lo = c->kc.lims[LF_LIM_ALPHA][0];
        m.alpha_hi = c->kc.lims[LF_LIM_ALPHA][1];
        m.flim_lo = c->kc.lims[LF_LIM_FLIM][0];
        m.flim_hi = c->kc.lims[LF_LIM_FLIM][1];
    } else {
        m.kind = 1;
        m.L_lo = c->kc.lims[LF_LIM_LSTAR][0];
        m.L_hi = c->kc.lims[LF_LIM_LSTAR][1];
        for (int i = 0; i < 3; ++i) m.piv[i] = c->kc.pivots[i];
        wt.resize((size_t)N);
        if (N) LF_HIP(c, hipMemcpy(wt.data(), c->d_lum, (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < N; ++i) wt[(size_t)i] = std::pow(10.0, wt[(size_t)i] - LF_LREF);
    }
    lfc::Out out;
    CompressedCat cc;
    cc.field_ind.assign(1, 0);
    // one validated set of bins for the whole catalogue's coordinate range, shared by the fields
    double klo = HUGE_VAL, khi = -HUGE_VAL;
    for (int64_t i = 0; i < N; ++i) {
        klo = std::fmin(klo, key[(size_t)i]);
        khi = std::fmax(khi, key[(size_t)i]);
    }
    const lfc::Bins bins = lfc::shared_bins(m, klo, khi);
    for (int f = 0; f < c->kc.nf; ++f) {
        const int64_t lo = c->field_ind[f], hi = c->field_ind[f + 1];
        if (!lfc::compress_field(m, key.data() + lo, wt.empty() ? nullptr : wt.data() + lo, hi - lo, out, &bins)) {
            c->err = "compress: the catalogue of field " + std::to_string(f) + " cannot be compressed to the error bound "
                     "(non-finite coordinate, or a prior box the bins cannot resolve)";
            return LF_ERR_ARG;
        }
        // in order of the coordinate inside the field, like the real catalogue (bins that kept their sources hold
        // them in catalogue order): a chunk's first pseudo-source is its faintest
        {
            const size_t a = (size_t)cc.field_ind.back(), b = out.node.size();
            std::vector<size_t> idx(b - a);
            for (size_t i = 0; i < idx.size(); ++i) idx[i] = a + i;
            std::stable_sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return out.node[x] < out.node[y]; });
            std::vector<double> nn(idx.size()), ww(idx.size());
            for (size_t i = 0; i < idx.size(); ++i) {
                nn[i] = out.node[idx[i]];
                ww[i] = out.weight[idx[i]];
            }
            std::copy(nn.begin(), nn.end(), out.node.begin() + (std::ptrdiff_t)a);
            std::copy(ww.begin(), ww.end(), out.weight.begin() + (std::ptrdiff_t)a);
        }
        cc.field_ind.push_back((int64_t)out.node.size());
    }
    cc.n = (int64_t)out.node.size();
    cc.nbins = out.nbins;
    cc.bound = out.bound;
    std::vector<double> lumc((size_t)cc.n, c->kc.variant == LF_ZEVOL ? LF_LREF : 0.0), U((size_t)cc.n);
    for (int64_t i = 0; i < cc.n; ++i)
        U[(size_t)i] = c->kc.variant == LF_FREE ? std::pow(10.0, out.node[(size_t)i] - LF_FREF) : out.node[(size_t)i] * out.node[(size_t)i];
    int rc;
    if ((rc = upload(c, &cc.d_lum, lumc.data(), (size_t)cc.n)) != LF_OK || (rc = upload(c, &cc.d_a1, out.node.data(), (size_t)cc.n)) != LF_OK ||
        (rc = upload(c, &cc.d_U, U.data(), (size_t)cc.n)) != LF_OK || (rc = upload(c, &cc.d_W, out.weight.data(), (size_t)cc.n)) != LF_OK) {
        free_cmp(cc);                                        // nothing half-built stays behind
        return rc;
    }
    cc.built = true;
    c->cmp = cc;
    // the integration grid, when it is separable (a failure here only leaves the full grid in use)
    if (c->kc.variant == LF_FREE && !c->h_L.empty() && !c->gridc.built) {
        lfc::Model mg = m;
        mg.kind = 2;
        lfc::GridOut go;
        const int S = c->kc.S;
        if (lfc::compress_grid(mg, S, c->h_L.data(), c->h_wL.data(), c->h_ck.data(), c->h_Dk.data(), go)) {
            std::vector<double> A4(go.u.size()), PGL((size_t)S);
            for (size_t i = 0; i < go.u.size(); ++i) A4[i] = std::pow(10.0, go.u[i] - LF_FREF);
            for (int j = 0; j < S; ++j) PGL[(size_t)j] = std::pow(10.0, c->h_L[(size_t)j] - LF_LREF);
            auto& g = c->gridc;
            if ((rc = upload(c, &g.d_U, go.u.data(), go.u.size())) != LF_OK) return rc;
            if ((rc = upload(c, &g.d_A4, A4.data(), A4.size())) != LF_OK) return rc;
            if ((rc = upload(c, &g.d_omega, go.omega.data(), go.omega.size())) != LF_OK) return rc;
            if ((rc = upload(c, &g.d_L, c->h_L.data(), (size_t)S)) != LF_OK) return rc;
            if ((rc = upload(c, &g.d_PGL, PGL.data(), (size_t)S)) != LF_OK) return rc;
            if ((rc = upload(c, &g.d_row0, go.row0.data(), go.row0.size())) != LF_OK) return rc;
            if ((rc = upload(c, &g.d_nrows, go.nrows.data(), go.nrows.size())) != LF_OK) return rc;
            if ((rc = upload(c, &g.d_off, go.off.data(), go.off.size())) != LF_OK) return rc;
            g.nb = go.nb;
            g.bound = go.bound;
            g.built = true;
        }
    }
    return LF_OK;
}

// see lf_set_option("compress"): direct vs compressed lnprob on 64 in-prior theta rows of this context
int compress_selfcheck(lf_ctx* c) {
    using namespace lf;
    const int B = 64, nd = c->kc.ndim, nf = c->kc.nf;
    std::vector<double> th((size_t)B * nd), direct(B), comp(B);
    uint64_t st = 0x9e3779b97f4a7c15ull;
    auto uni = [&]() {                                  // splitmix64 -> [0, 1)
        uint64_t z = (st += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return (double)((z ^ (z >> 31)) >> 11) * 1.1102230246251565e-16;
    };
    auto in = [&](int lim, double lo_frac) {            // a point of the prior interval `lim` (upper part for L*)
        const double lo = c->kc.lims[lim][0], hi = c->kc.lims[lim][1];
        return lo + (hi - lo) * (lo_frac + (1.0 - lo_frac) * uni());
    };
    for (int b = 0; b < B; ++b) {
        double* t = th.data() + (size_t)b * nd;
        int k = 0;
        if (c->kc.variant == LF_ZEVOL) {
            for (int i = 0; i < 3; ++i) t[k++] = in(LF_LIM_LSTAR, 0.3);     // (the faint end of the L* box is the underflow zone)
            for (int i = 0; i < 3; ++i) t[k++] = in(LF_LIM_PHISTAR, 0.0);
            if (!c->kc.fix_sch_al) t[k++] = in(LF_LIM_SCH_AL, 0.0);
        } else {
            t[k++] = in(LF_LIM_LSTAR, 0.3);
            t[k++] = in(LF_LIM_PHISTAR, 0.0);
            if (!c->kc.fix_sch_al) t[k++] = in(LF_LIM_SCH_AL, 0.0);
            if (c->kc.variant == LF_FREE) {
                for (int f = 0; f < nf; ++f) t[k++] = in(LF_LIM_FLIM, 0.0);
                t[k++] = in(LF_LIM_ALPHA, 0.0);
            }
        }
    }
    const int64_t was = c->opt_compress;
    int rc = ensure_workspace(c, B, 0, 0);
    for (int pass = 0; pass < 2 && rc == LF_OK; ++pass) {
        c->opt_compress = pass;
        LF_HIP(c, hipMemcpy(c->d_theta, th.data(), th.size() * sizeof(double), hipMemcpyHostToDevice));
        rc = enqueue(c, c->d_theta, B, c->d_out, nullptr, nullptr, c->stream);
        if (rc != LF_OK) break;
        LF_HIP(c, hipStreamSynchronize(c->stream));
        LF_HIP(c, hipMemcpy(pass ? comp.data() : direct.data(), c->d_out, B * sizeof(double), hipMemcpyDeviceToHost));
    }
    c->opt_compress = was;
    if (rc != LF_OK) return rc;
    double worst = 0.0;
    int nfin = 0;
    for (int b = 0; b < B; ++b) {
        if (std::isinf(direct[b]) || std::isinf(comp[b])) {
            if (direct[b] != comp[b]) worst = HUGE_VAL;
            continue;
        }
        ++nfin;
        worst = std::fmax(worst, std::fabs(comp[b] - direct[b]) / std::fmax(std::fabs(direct[b]), 1e-300));
    }
    if (!(worst <= 1.0e-12)) {
        char msg[200];
        std::snprintf(msg, sizeof(msg), "compress: the compressed catalogue differs from the direct path by %.3g (relative) on "
                      "%d of 64 self-check walkers of this prior box: option refused", worst, nfin);
        c->err = msg;
        return LF_ERR_ARG;
    }
    return LF_OK;
}

void free_ctx(lf_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (auto& e : c->events) {
        hipEventDestroy(e.a);
        hipEventDestroy(e.b);
    }
    for (auto& kv : c->chunks) {
        hipFree(kv.second.d_start);
        hipFree(kv.second.d_len);
        hipFree(kv.second.d_field);
        hipFree(kv.second.d_keys);
    }
    for (auto& kv : c->chunks_free) {
        hipFree(kv.second.d_start);
        hipFree(kv.second.d_len);
        hipFree(kv.second.d_field);
        hipFree(kv.second.d_keys);
    }
    free_cmp(c->cmp);
    if (c->d_partR) hipFree(c->d_partR);
    if (c->d_cells) hipFree(c->d_cells);
    if (c->d_cc_start) hipFree(c->d_cc_start);
    if (c->d_deal) hipFree(c->d_deal);
    if (c->d_err) hipFree(c->d_err);
    if (c->d_cc_len) hipFree(c->d_cc_len);
    if (c->d_cc_field) hipFree(c->d_cc_field);
    {
        auto& g = c->gridc;
        double* gb[] = {g.d_U, g.d_A4, g.d_omega, g.d_L, g.d_PGL};
        for (double* b : gb)
            if (b) hipFree(b);
        int* gi_[] = {g.d_row0, g.d_nrows, g.d_off};
        for (int* b : gi_)
            if (b) hipFree(b);
        if (c->d_nodes4) hipFree(c->d_nodes4);
        if (c->d_zcol) hipFree(c->d_zcol);
        if (c->gridq.d_rec) hipFree(c->gridq.d_rec);
        if (c->gridq.d_omega) hipFree(c->gridq.d_omega);
        if (c->gridq.d_rows) hipFree(c->gridq.d_rows);
    }
    double* bufs[] = {c->d_lum, c->d_a1, c->d_P, c->d_U, c->d_G, c->d_PG, c->d_W, c->d_a3, c->d_a4, c->d_a4min, c->d_nodes8,
                      c->d_theta, c->d_out, c->d_outA, c->d_outB, c->d_wrec, c->d_partA, c->d_partB};
    for (double* b : bufs)
        if (b) hipFree(b);
    if (c->d_wstat) hipFree(c->d_wstat);
    if (c->d_wmode) hipFree(c->d_wmode);
    if (c->d_wbase) hipFree(c->d_wbase);
    if (c->d_slow) hipFree(c->d_slow);
    if (c->d_forms) hipFree(c->d_forms);
    if (c->d_queue) hipFree(c->d_queue);
    if (c->h_theta) hipHostFree(c->h_theta);
    if (c->h_out) hipHostFree(c->h_out);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int build(lf_ctx* c, const lf_desc* d) {
    using namespace lf;
    const int nf = d->nf, S = d->S;
    const int64_t N = d->N;
    KConst& kc = c->kc;
    kc.variant = d->variant;
    kc.fix_sch_al = d->fix_sch_al ? 1 : 0;
    kc.specialise = 1;
    kc.grid_part = 0;
    kc.grid_parts = 1;
    kc.tables = 1;
    kc.key_x0 = 0.0;
    kc.forms = nullptr;
#ifdef LF_STAMPS
    kc.stamps = nullptr;
#endif
    kc.nf = nf;
    kc.S = S;
    if (d->variant == LF_FREE) kc.ndim = 2 + (kc.fix_sch_al ? 0 : 1) + nf + 1;
    else if (d->variant == LF_FIXCOMP) kc.ndim = 2 + (kc.fix_sch_al ? 0 : 1);
    else kc.ndim = 6 + (kc.fix_sch_al ? 0 : 1);
    for (int f = 0; f < MAXF; ++f) {
        kc.lnom0_src[f] = 0.0;
        kc.om0_grid[f] = 0.0;
        kc.flim0[f] = 0.0;
    }
    for (int f = 0; f < nf; ++f) {
        // per-source term: Omega_0_arr is dtype=int (lumfuncmcmc.py:285) -> truncation toward zero
        kc.lnom0_src[f] = std::log(std::trunc(d->omega0[f]) / LF_SQARCSEC);
        kc.om0_grid[f] = d->omega0[f] / LF_SQARCSEC;           // integral: float (lumfuncmcmc.py:375)
        if (d->flim0) kc.flim0[f] = d->flim0[f];
    }
    {
        const double a = (2.0 * d->fcmin - 1.0) * (2.0 * d->fcmin - 1.0);   // VmaxLumFunc.py:164
        kc.fc_ratio = std::fabs(a / (1.0 - a));
    }
    std::memcpy(kc.lims, d->lims, sizeof(kc.lims));
    std::memcpy(kc.pivots, d->pivots, sizeof(kc.pivots));
    kc.sch_al0 = d->sch_al0;
    kc.alpha0 = d->alpha0;
    c->N = N;
    c->nnodes = S * S;
    c->field_ind.assign(d->field_ind, d->field_ind + nf + 1);

    // ---- per-source tables.  FREE: the sources of a field are put in order of flux (the layout is ours to choose; a
    // sum over sources does not care), so that a chunk's first source is its faintest and the kernels can pick
    // a cheaper form of the term per (walker, chunk) - see term_free_noexp.  NaN fluxes go last.
    std::vector<int64_t> perm((size_t)N);
    for (int64_t i = 0; i < N; ++i) perm[(size_t)i] = i;
    // ZEVOL: in order of redshift, so that a lane's ST sources are neighbours in z and 10^(-L*(z)) of all of them follows
    // from ONE exponential at the lane's middle source (lf_kernels.h: the local form of the z-evolving term).
    if (d->variant == LF_FREE || d->variant == LF_ZEVOL) {
        const double* key = d->variant == LF_FREE ? d->logf : d->z;
        for (int f = 0; f < nf; ++f)
            std::stable_sort(perm.begin() + d->field_ind[f], perm.begin() + d->field_ind[f + 1], [&](int64_t a, int64_t b) {
                const double x = key[a], y = key[b];
                return std::isnan(y) ? !std::isnan(x) : x < y;
            });
    }
    std::vector<double> lumv(N), a1(N), P(N), U(N);
    for (int64_t i = 0; i < N; ++i) lumv[(size_t)i] = d->lum[perm[(size_t)i]];
    for (int64_t i = 0; i < N; ++i) {
        const double lum = lumv[(size_t)i];
        if (d->variant == LF_FREE) {
            a1[i] = d->logf[perm[(size_t)i]];
            P[i] = std::pow(10.0, lum - LF_LREF);
            U[i] = std::pow(10.0, d->logf[perm[(size_t)i]] - LF_FREF);
        } else if (d->variant == LF_FIXCOMP) {
            a1[i] = std::log(d->om_arr[perm[(size_t)i]]);
            P[i] = std::pow(10.0, lum - LF_LREF);
            U[i] = 0.0;
        } else {
            a1[i] = d->z[perm[(size_t)i]];
            P[i] = std::log(d->om_arr[perm[(size_t)i]]);
            U[i] = d->z[perm[(size_t)i]] * d->z[perm[(size_t)i]];
        }
    }
    // per-field extremes for the mode classification in lf_prepare
    for (int f = 0; f < MAXF; ++f) {
        kc.nsrc[f] = 0;
        kc.pmax[f] = kc.lum_min[f] = kc.lum_max[f] = kc.a_min[f] = kc.u_min[f] = kc.u_max[f] = kc.z_lo[f] = kc.z_hi[f] = kc.slc[f] = kc.sp[f] = kc.som[f] = kc.sz[f] = kc.sz2[f] = 0.0;
    }
    for (int f = 0; f < nf; ++f) {
        const int64_t lo = d->field_ind[f], hi = d->field_ind[f + 1];
        kc.nsrc[f] = (int)(hi - lo);
        if (hi <= lo) continue;
        double pmax = -HUGE_VAL, lmin = HUGE_VAL, lmax = -HUGE_VAL, amin = HUGE_VAL, amax = -HUGE_VAL, zlo = HUGE_VAL, zhi = -HUGE_VAL;
        bool nan = false;
        long double slc = 0.0L, sp = 0.0L, som = 0.0L, sz = 0.0L, sz2 = 0.0L;
        for (int64_t i = lo; i < hi; ++i) {
            const double lum = lumv[(size_t)i];
            slc += (long double)(lum - LF_LREF);
            if (d->variant != LF_ZEVOL) sp += (long double)P[i];
            if (d->variant == LF_FIXCOMP) som += (long double)a1[i];
            if (d->variant == LF_ZEVOL) {
                som += (long double)P[i];
                sz += (long double)d->z[perm[(size_t)i]];
                sz2 += (long double)U[i];             // the rounded z_i^2 the kernels use
            }
            lmin = std::fmin(lmin, lum);
            lmax = std::fmax(lmax, lum);
            nan = nan || std::isnan(lum);
            if (d->variant != LF_ZEVOL) pmax = std::fmax(pmax, P[i]);
            const double a = d->variant == LF_FREE ? a1[i] : (d->variant == LF_FIXCOMP ? a1[i] : P[i]);
            amin = std::fmin(amin, a);
            amax = std::fmax(amax, a);
            nan = nan || std::isnan(a);
            if (d->variant == LF_ZEVOL) {
                zlo = std::fmin(zlo, d->z[perm[(size_t)i]]);
                zhi = std::fmax(zhi, d->z[perm[(size_t)i]]);
                nan = nan || std::isnan(d->z[perm[(size_t)i]]);
            }
        }
        if (nan) amin = -HUGE_VAL;                 // NaN input: force the careful path
        kc.pmax[f] = pmax;
        kc.lum_min[f] = lmin;
        kc.lum_max[f] = lmax;
        kc.a_min[f] = amin;
        kc.u_min[f] = d->variant == LF_FREE ? std::pow(10.0, amin - LF_FREF) : 0.0;
        kc.u_max[f] = d->variant == LF_FREE ? (nan ? HUGE_VAL : std::pow(10.0, amax - LF_FREF)) : 0.0;
        kc.z_lo[f] = zlo;
        kc.z_hi[f] = zhi;
        kc.slc[f] = (double)slc;
        kc.sp[f] = (double)sp;
        kc.som[f] = (double)som;
        kc.sz[f] = (double)sz;
        kc.sz2[f] = (double)sz2;
    }
    if (d->variant == LF_FREE || d->variant == LF_ZEVOL) {
        // origin of the integer keys of log-flux (ZEVOL: of redshift), and the host copy of the sorted values the chunk
        // keys come from
        double x0 = HUGE_VAL;
        for (int64_t i = 0; i < N; ++i)
            if (std::isfinite(a1[(size_t)i])) x0 = std::fmin(x0, a1[(size_t)i]);
        kc.key_x0 = std::isfinite(x0) ? x0 : 0.0;
        c->h_x = a1;
    }
    int rc;
    kc.cells = 0;
    kc.zcell_rho = 0.0;
    for (int f = 0; f < MAXF; ++f) {
        kc.kf_first[f] = 0;
        kc.kf_last[f] = lf::KEY_MAX;
    }
    for (int f = 0; f <= MAXF; ++f) kc.cc_fstart[f] = 0;
    if (d->variant == LF_FREE && (rc = build_cells(c, kc, a1, nf)) != LF_OK) return rc;
    if (d->variant == LF_ZEVOL) {
        std::vector<double> wts((size_t)N);
        for (int64_t i = 0; i < N; ++i) wts[(size_t)i] = std::pow(10.0, lumv[(size_t)i] - LF_LREF);
        kc.zcell_rho = zcell_rho_for_box(kc, nf);
        if ((rc = build_cells(c, kc, a1, nf, wts.data())) != LF_OK) return rc;
    }
    if ((rc = upload(c, &c->d_lum, lumv.data(), (size_t)N)) != LF_OK) return rc;
    if ((rc = upload(c, &c->d_a1, a1.data(), (size_t)N)) != LF_OK) return rc;
    if ((rc = upload(c, &c->d_P, P.data(), (size_t)N)) != LF_OK) return rc;
    if ((rc = upload(c, &c->d_U, U.data(), (size_t)N)) != LF_OK) return rc;

    // ---- grid-node tables.  trapz weights from the actual spacings (scipy trapz = sum d*(y1+y0)/2)
    size_t nn = (size_t)S * S;
    const size_t nn2 = nn;               // (the lattice; nn becomes S below when the fixed-completeness grid collapses to its rows)
    std::vector<double> G(nn), PG(nn), W(nn), a3(nn, 0.0), a4(nn, 0.0), wz(S);
    for (int k = 0; k < S; ++k) {
        const double dl = k > 0 ? d->zarr[k] - d->zarr[k - 1] : 0.0;
        const double dr = k < S - 1 ? d->zarr[k + 1] - d->zarr[k] : 0.0;
        wz[k] = 0.5 * (dl + dr);
    }
    for (int j = 0; j < S; ++j) {
        for (int k = 0; k < S; ++k) {
            const size_t g = (size_t)j * S + k;
            const double x = d->logL[g];
            const double dl = j > 0 ? x - d->logL[g - S] : 0.0;
            const double dr = j < S - 1 ? d->logL[g + S] - x : 0.0;
            const double w = 0.5 * (dl + dr) * wz[k];
            G[g] = x;
            PG[g] = std::pow(10.0, x - LF_LREF);
            if (d->variant == LF_FREE) {
                const double dlcm = LF_MPC_CM * d->dl_zarr[k];
                const double lf = x - std::log10(4.0 * M_PI * dlcm * dlcm);
                a3[g] = lf;
                a4[g] = std::pow(10.0, lf - LF_FREF);
                W[g] = w * d->volume_part[k];
            } else {
                double s = 0.0;
                for (int f = 0; f < nf; ++f) s += d->integ_part[(size_t)f * nn2 + g];
                W[g] = w * s;
                if (d->variant == LF_ZEVOL) {
                    a3[g] = d->zarr[k];
                    a4[g] = d->zarr[k] * d->zarr[k];
                }
            }
        }
    }
    if (d->variant == LF_FREE && S <= GRIDC_MAX_S) {
        bool sep = true;
        for (int j = 0; j < S && sep; ++j)
            for (int k = 1; k < S; ++k)
                if (d->logL[(size_t)j * S + k] != d->logL[(size_t)j * S]) {
                    sep = false;
                    break;
                }
        if (sep) {
            c->h_L.resize(S); c->h_wL.resize(S); c->h_ck.resize(S); c->h_Dk.resize(S);
            for (int j = 0; j < S; ++j) {
                const double x = d->logL[(size_t)j * S];
                const double dl = j > 0 ? x - d->logL[(size_t)(j - 1) * S] : 0.0;
                const double dr = j < S - 1 ? d->logL[(size_t)(j + 1) * S] - x : 0.0;
                c->h_L[j] = x;
                c->h_wL[j] = 0.5 * (dl + dr);
            }
            for (int k = 0; k < S; ++k) {
                const double dlcm = LF_MPC_CM * d->dl_zarr[k];
                c->h_Dk[k] = std::log10(4.0 * M_PI * dlcm * dlcm);
                c->h_ck[k] = wz[k] * d->volume_part[k];
            }
            // piece B over flux bins (lf_gridbound.h): the bins are proven for the context's whole prior box of (alpha_C, Flim),
            // or the lattice stays.  Every rank of a sharded run derives the same bins from the same grid and box.
            const double alo = kc.lims[LF_LIM_ALPHA][0], ahi = kc.lims[LF_LIM_ALPHA][1];
            const double flo = kc.lims[LF_LIM_FLIM][0], fhi = kc.lims[LF_LIM_FLIM][1];
            if (alo > 0.0 && ahi >= alo && flo > 0.0 && fhi >= flo && std::isfinite(ahi) && std::isfinite(fhi) && !std::getenv("LF_NO_GRIDQ")) {
                const lfq::Box bx{std::sqrt(kc.fc_ratio), alo, ahi, std::log10(flo) + LF_FREF, std::log10(fhi) + LF_FREF};
                lfq::GridQ gq;
                if (lfq::build_gridq(bx, S, c->h_L.data(), c->h_wL.data(), c->h_ck.data(), c->h_Dk.data(), LF_FREF, LF_LREF, gq)) {
                    auto& g = c->gridq;
                    if ((rc = upload(c, &g.d_rec, gq.rec.data(), gq.rec.size())) != LF_OK) return rc;
                    if ((rc = upload(c, &g.d_omega, gq.omega.data(), gq.omega.size())) != LF_OK) return rc;
                    if ((rc = upload(c, &g.d_rows, gq.rows.data(), gq.rows.size())) != LF_OK) return rc;
                    g.nb = gq.nb;
                    g.margin = gq.margin;
                    g.built = true;
                }
            }
        }
    }
    kc.zgrid_cols = 0;
    if (d->variant == LF_ZEVOL && S >= lf::BLOCK / (lf::ZCOLS - 1) && !std::getenv("LF_NO_ZGRID_COLS")) {
        // z-evolving: store the lattice column by column (a sum does not care; lf_kernels.h: gridsum_body takes what depends
        // on the walker per COLUMN).  S >= 128: a chunk of 256 nodes then touches at most 3 columns.
        std::vector<double> t(nn);
        for (std::vector<double>* arr : {&G, &PG, &W, &a3, &a4}) {
            for (int j = 0; j < S; ++j)
                for (int k = 0; k < S; ++k) t[(size_t)k * S + j] = (*arr)[(size_t)j * S + k];
            arr->swap(t);
        }
        kc.zgrid_cols = 1;
    }
    if (d->variant == LF_FIXCOMP && !std::getenv("LF_NO_COLLAPSE_GRID")) {        // (the variable: A/B runs and the test of this step)
        // Fixed completeness: the integrand at node (j, k) is T_w(L_jk) W_jk with everything but the Schechter function
        // T folded into W.  When every redshift column has the same luminosity nodes (the constructor clips the columns'
        // lower ends to the catalogue's faintest luminosity: with min_comp_frac = 0 all of them) T depends on the row only
        // and the double sum is sum_j T_w(L_j) (sum_k W_jk): S nodes instead of S^2, exactly - the trapezoid rule's sums
        // in another order.  Row sums in extended precision.
        bool sep = true;
        for (int j = 0; j < S && sep; ++j)
            for (int k = 1; k < S; ++k)
                if (d->logL[(size_t)j * S + k] != d->logL[(size_t)j * S]) {
                    sep = false;
                    break;
                }
        if (sep) {
            for (int j = 0; j < S; ++j) {
                long double rs = 0.0L;
                for (int k = 0; k < S; ++k) rs += (long double)W[(size_t)j * S + k];
                G[(size_t)j] = G[(size_t)j * S];
                PG[(size_t)j] = PG[(size_t)j * S];
                W[(size_t)j] = (double)rs;
                a3[(size_t)j] = a4[(size_t)j] = 0.0;
            }
            nn = (size_t)S;
            c->nnodes = S;
        }
    }
    if ((rc = upload(c, &c->d_G, G.data(), nn)) != LF_OK) return rc;
    if ((rc = upload(c, &c->d_PG, PG.data(), nn)) != LF_OK) return rc;
    if ((rc = upload(c, &c->d_W, W.data(), nn)) != LF_OK) return rc;
    if ((rc = upload(c, &c->d_a3, a3.data(), nn)) != LF_OK) return rc;
    if ((rc = upload(c, &c->d_a4, a4.data(), nn)) != LF_OK) return rc;
    if (d->variant != LF_FREE) {
        // lf_pers reads the nodes as 32-byte records {G, PG, W, redshift column}, padded to whole chunks of 64 (pads: W = 0)
        const size_t nch = (nn + 63) / 64;
        std::vector<double> n4(nch * 64 * 4, 0.0);
        for (size_t g = 0; g < nch * 64; ++g) {
            const size_t gg = std::min(g, nn - 1);
            double* r = &n4[g * 4];
            r[0] = G[gg];
            r[1] = PG[gg];
            r[2] = g < nn ? W[gg] : 0.0;
            r[3] = kc.zgrid_cols ? (double)(gg / (size_t)S) : 0.0;      // (column-major lattice: node = k S + j)
        }
        if ((rc = upload(c, &c->d_nodes4, n4.data(), n4.size())) != LF_OK) return rc;
        c->nch4 = (int)nch;
        std::vector<double> zc((size_t)S * 2);
        for (int k = 0; k < S; ++k) {
            zc[(size_t)2 * k] = d->zarr[k];
            zc[(size_t)2 * k + 1] = d->zarr[k] * d->zarr[k];
        }
        if ((rc = upload(c, &c->d_zcol, zc.data(), zc.size())) != LF_OK) return rc;
    }
    {
        // per chunk of 256 nodes the smallest a4 (FREE; NaN-safe: a NaN node keeps the general form)
        std::vector<double> a4min((nn + lf::BLOCK - 1) / lf::BLOCK, 0.0);
        for (size_t ch = 0; ch < a4min.size(); ++ch) {
            double m = HUGE_VAL;
            for (size_t g = ch * lf::BLOCK; g < std::min(nn, (ch + 1) * lf::BLOCK); ++g) m = std::isnan(a4[g]) ? 0.0 : std::fmin(m, a4[g]);
            a4min[ch] = d->variant == LF_FREE ? m : 0.0;
        }
        if ((rc = upload(c, &c->d_a4min, a4min.data(), a4min.size())) != LF_OK) return rc;
        // lf_free reads the nodes as 64-byte records {G, PG, W, a3, a4, smallest a4 of the node's chunk of 64, -, -}, padded
        // to whole chunks (pads: the last node again with W = 0): one contiguous load per lane, nothing to mask
        if (d->variant == LF_FREE) {
            const size_t nch64 = (nn + 63) / 64;
            std::vector<double> n8(nch64 * 64 * 8, 0.0);
            for (size_t ch = 0; ch < nch64; ++ch) {
                double m = HUGE_VAL;
                for (size_t g = ch * 64; g < std::min(nn, (ch + 1) * 64); ++g) m = std::isnan(a4[g]) ? 0.0 : std::fmin(m, a4[g]);
                for (size_t l = 0; l < 64; ++l) {
                    const size_t g = std::min(ch * 64 + l, nn - 1);
                    double* r = &n8[(ch * 64 + l) * 8];
                    r[0] = G[g];
                    r[1] = PG[g];
                    r[2] = ch * 64 + l < nn ? W[g] : 0.0;
                    r[3] = a3[g];
                    r[4] = a4[g];
                    r[5] = m;
                }
            }
            if ((rc = upload(c, &c->d_nodes8, n8.data(), n8.size())) != LF_OK) return rc;
        }
    }
    {
        hipDeviceProp_t prop;
        LF_HIP(c, hipGetDeviceProperties(&prop, c->device));
        c->num_cu = prop.multiProcessorCount;
        LF_HIP(c, hipMalloc((void**)&c->d_queue, 1024 * sizeof(int)));
        LF_HIP(c, hipMemset(c->d_queue, 0, 1024 * sizeof(int)));
        c->cap_queue = 1024;
    }
    LF_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    const int mb = d->max_batch > 0 ? d->max_batch : 1024;
    return ensure_workspace(c, mb, 0, 0);
}

}  // namespace

extern "C" {

int lf_deal_table(int n_cell_chunks, int n_bins, int grid_part, int grid_parts, int32_t* table, int64_t cap) {
    using namespace lf;
    if (n_cell_chunks < 0 || n_bins < 0 || grid_parts < 0 || grid_part < 0 || (grid_parts > 0 && grid_part >= grid_parts)) return LF_ERR_ARG;
    const int64_t n = (int64_t)DEAL_LIST + n_cell_chunks + n_bins;
    if (!table) return (int)n;
    if (cap < n) return LF_ERR_ARG;
    const std::vector<int> t = make_deal(n_cell_chunks, n_bins, grid_part, grid_parts);
    for (int64_t i = 0; i < n; ++i) table[i] = t[(size_t)i];
    return (int)n;
}

int lf_abi_version(void) { return LF_ABI_VERSION; }

const char* lf_last_error(const lf_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

lf_ctx* lf_create(const lf_desc* d) {
    g_create_error.clear();
    if (!d) {
        g_create_error = "lf_create: NULL descriptor";
        return nullptr;
    }
    auto bad = [&](const char* m) {
        g_create_error = std::string("lf_create: ") + m;
        return (lf_ctx*)nullptr;
    };
    if (d->variant < LF_FREE || d->variant > LF_ZEVOL) return bad("unknown variant");
    if (d->nf < 1 || d->nf > LF_MAX_FIELDS) return bad("nf out of range (1..LF_MAX_FIELDS)");
    if (d->S < 2 || d->S > 4096) return bad("S out of range");
    if (d->N < 0 || d->N > 2000000000LL) return bad("N out of range");
    if (!d->field_ind || !d->omega0 || !d->logL || !d->zarr) return bad("NULL field_ind/omega0/logL/zarr");
    if (d->N > 0 && !d->lum) return bad("NULL lum");
    if (d->field_ind[0] != 0 || d->field_ind[d->nf] != d->N) return bad("field_ind must span [0, N]");
    for (int f = 0; f < d->nf; ++f)
        if (d->field_ind[f + 1] < d->field_ind[f]) return bad("field_ind must be non-decreasing");
    if (!(d->fcmin > 0.0 && d->fcmin < 1.0) || d->fcmin == 0.5) return bad("fcmin must be in (0,1), != 0.5");
    if (d->variant == LF_FREE) {
        if ((d->N > 0 && !d->logf) || !d->volume_part || !d->dl_zarr) return bad("FREE needs logf, volume_part, dl_zarr");
    } else {
        if ((d->N > 0 && !d->om_arr) || !d->integ_part) return bad("FIXCOMP/ZEVOL need om_arr, integ_part");
        if (d->variant == LF_FIXCOMP && !d->flim0) return bad("FIXCOMP needs flim0");
        if (d->variant == LF_ZEVOL && d->N > 0 && !d->z) return bad("ZEVOL needs z");
        if (d->variant == LF_ZEVOL && (d->pivots[0] == d->pivots[1] || d->pivots[0] == d->pivots[2] ||
                                       d->pivots[1] == d->pivots[2]))
            return bad("ZEVOL pivots must be distinct");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return bad("no HIP device visible");
    if (d->device < 0 || d->device >= ndev) return bad("device ordinal out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d->device) != hipSuccess) return bad("hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("lf_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        return nullptr;
    }
    if (hipSetDevice(d->device) != hipSuccess) return bad("hipSetDevice failed");
    lf_ctx* c = new (std::nothrow) lf_ctx();
    if (!c) return bad("out of host memory");
    c->device = d->device;
    int rc = LF_OK;
    try {
        rc = build(c, d);
    } catch (const std::bad_alloc&) {
        c->err = "out of host memory";
        rc = LF_ERR_NOMEM;
    }
    if (rc != LF_OK) {
        g_create_error = "lf_create: " + c->err;
        free_ctx(c);
        return nullptr;
    }
    return c;
}

void lf_destroy(lf_ctx* ctx) { free_ctx(ctx); }

int lf_ndim(const lf_ctx* ctx) { return ctx ? ctx->kc.ndim : LF_ERR_ARG; }

int lf_lnprob_batch_device(lf_ctx* c, const double* d_theta, int B, double* d_out, void* hip_stream) {
    if (!c) return LF_ERR_ARG;
    if (!d_theta || !d_out || B <= 0) {
        c->err = "lf_lnprob_batch_device: NULL pointer or B <= 0";
        return LF_ERR_ARG;
    }
    LF_HIP(c, hipSetDevice(c->device));
    return enqueue(c, d_theta, B, d_out, nullptr, nullptr, (hipStream_t)hip_stream);
}

int lf_lnprob_batch_device_n(lf_ctx* c, const double* d_theta, int B, int K, double* d_out, void* hip_stream) {
    if (!c) return LF_ERR_ARG;
    if (!d_theta || !d_out || B <= 0 || K <= 0) {
        c->err = "lf_lnprob_batch_device_n: NULL pointer, B <= 0 or K <= 0";
        return LF_ERR_ARG;
    }
    LF_HIP(c, hipSetDevice(c->device));
    const size_t nd = (size_t)c->kc.ndim;
    for (int k = 0; k < K; ++k) {
        const int rc = enqueue(c, d_theta + (size_t)k * B * nd, B, d_out + (size_t)k * B, nullptr, nullptr, (hipStream_t)hip_stream);
        if (rc != LF_OK) return rc;
    }
    return LF_OK;
}

static int host_eval(lf_ctx* c, const double* theta, int B, double* out, double* outA, double* outB) {
    using namespace lf;
    if (!c) return LF_ERR_ARG;
    if (!theta || B <= 0 || (!out && !(outA && outB))) {
        c->err = "lf_lnprob_batch: NULL pointer or B <= 0";
        return LF_ERR_ARG;
    }
    LF_HIP(c, hipSetDevice(c->device));
    int rc = ensure_workspace(c, B, 0, 0);
    if (rc != LF_OK) return rc;
    const size_t tb = (size_t)B * c->kc.ndim * sizeof(double);
    std::memcpy(c->h_theta, theta, tb);
    LF_HIP(c, hipMemcpyAsync(c->d_theta, c->h_theta, tb, hipMemcpyHostToDevice, c->stream));
    // (the two pieces only when they are asked for: a plain evaluation then takes lf_free's one-launch form)
    rc = enqueue(c, c->d_theta, B, c->d_out, outA ? c->d_outA : nullptr, outB ? c->d_outB : nullptr, c->stream);
    if (rc != LF_OK) return rc;
    const size_t ob = (size_t)B * sizeof(double);
    LF_HIP(c, hipMemcpyAsync(c->h_out, c->d_out, ob, hipMemcpyDeviceToHost, c->stream));
    if (outA) {
        LF_HIP(c, hipMemcpyAsync(c->h_out + B, c->d_outA, ob, hipMemcpyDeviceToHost, c->stream));
        LF_HIP(c, hipMemcpyAsync(c->h_out + 2 * (size_t)B, c->d_outB, ob, hipMemcpyDeviceToHost, c->stream));
    }
    LF_HIP(c, hipStreamSynchronize(c->stream));
    if (out && c->d_err) {
        // lnprob is never NaN - unless a polling finisher gave up (lf_free.h: PART_POLLS): then say so instead of handing NaN on
        bool nan = false;
        for (int i = 0; i < B; ++i) nan = nan || c->h_out[i] != c->h_out[i];
        if (nan) {
            int e = 0;
            LF_HIP(c, hipMemcpy(&e, c->d_err, sizeof(int), hipMemcpyDeviceToHost));
            if (e) {
                c->err = "lf_lnprob_batch: a tile's finishing workgroup waited in vain for the tile's partial sums (device stalled?)";
                return LF_ERR_HIP;
            }
        }
    }
    if (out) std::memcpy(out, c->h_out, ob);
    if (outA) {
        std::memcpy(outA, c->h_out + B, ob);
        std::memcpy(outB, c->h_out + 2 * (size_t)B, ob);
    }
    return LF_OK;
}

int lf_lnprob_batch(lf_ctx* c, const double* theta, int B, double* out) {
    if (c && !out) {
        c->err = "lf_lnprob_batch: NULL out";
        return LF_ERR_ARG;
    }
    return host_eval(c, theta, B, out, nullptr, nullptr);
}

int lf_lnprob_pieces(lf_ctx* c, const double* theta, int B, double* outA, double* outB) {
    if (c && (!outA || !outB)) {
        c->err = "lf_lnprob_pieces: NULL output";
        return LF_ERR_ARG;
    }
    return host_eval(c, theta, B, nullptr, outA, outB);
}

int lf_set_profiling(lf_ctx* c, int enabled) {
    if (!c) return LF_ERR_ARG;
    c->profiling = enabled < 0 ? 0 : (enabled > 2 ? 2 : enabled);
    return LF_OK;
}

int lf_kernel_times(lf_ctx* c, double ms[4], int64_t launches[4]) {
    if (!c || !ms || !launches) return LF_ERR_ARG;
    LF_HIP(c, hipSetDevice(c->device));
    if (c->span_open) {                  // (a span cut short: what it holds so far)
        if (c->span_ep.count > 0) {
            hipEventRecord(c->span_ep.b, c->last_stream);
            c->events.push_back(c->span_ep);
        } else {
            hipEventDestroy(c->span_ep.a);
            hipEventDestroy(c->span_ep.b);
        }
        c->span_open = false;
    }
    for (auto& e : c->events) {
        LF_HIP(c, hipEventSynchronize(e.b));
        float t = 0.f;
        LF_HIP(c, hipEventElapsedTime(&t, e.a, e.b));
        c->acc_ms[e.kind] += t;
        c->acc_n[e.kind] += e.count;
        hipEventDestroy(e.a);
        hipEventDestroy(e.b);
    }
    c->events.clear();
    for (int i = 0; i < 4; ++i) {
        ms[i] = c->acc_ms[i];
        launches[i] = c->acc_n[i];
        c->acc_ms[i] = 0;
        c->acc_n[i] = 0;
    }
    return LF_OK;
}

#ifdef LF_STAMPS
// diagnostic build only (tools/stamps.py): arm / read the per-workgroup time stamps of lf_main's source workgroups
int lf_debug_stamps(lf_ctx* c, uint64_t* out, int64_t nblocks) {
    if (!c) return LF_ERR_ARG;
    LF_HIP(c, hipSetDevice(c->device));
    LF_HIP(c, hipDeviceSynchronize());
    if (!out) {                                   // arm for nblocks workgroups
        if (c->kc.stamps) hipFree(c->kc.stamps);
        c->kc.stamps = nullptr;
        if (nblocks > 0) {
            LF_HIP(c, hipMalloc((void**)&c->kc.stamps, (size_t)nblocks * 8 * sizeof(uint64_t)));
            LF_HIP(c, hipMemset(c->kc.stamps, 0, (size_t)nblocks * 8 * sizeof(uint64_t)));
        }
        return LF_OK;
    }
    if (!c->kc.stamps) return LF_ERR_ARG;
    LF_HIP(c, hipMemcpy(out, c->kc.stamps, (size_t)nblocks * 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return LF_OK;
}
#endif

int lf_veff(int device, int64_t n, const double* flux, const double* flim, const double* vol, double vol_all, double pref0,
            double alpha, double fcmin, const int32_t* bin_of, int32_t nbin, int32_t nboot, const int64_t* boot_idx, uint64_t seed,
            double* phi, double* sums) {
    if (n <= 0 || !flux || !flim || !phi || pref0 <= 0.0 || nbin < 0 || nbin > lf::VEFF_MAXBIN || nboot < 0 || (nbin > 0 && (!bin_of || !sums)))
        return LF_ERR_ARG;
    // the caller's resampling indices address phi and bin_of on the device: outside [0, n) they are refused here
    if (boot_idx && nbin > 0)
        for (int64_t i = 0; i < n * (int64_t)nboot; ++i)
            if (boot_idx[i] < 0 || boot_idx[i] >= n) return LF_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return LF_ERR_NODEV;
    double *d_flux = nullptr, *d_flim = nullptr, *d_vol = nullptr, *d_phi = nullptr, *d_sums = nullptr;
    int* d_bin = nullptr;
    long long* d_idx = nullptr;
    const size_t nb = (size_t)n * sizeof(double);
    int rc = LF_OK;
    auto ok = [&](hipError_t e) {
        if (e != hipSuccess && rc == LF_OK) rc = LF_ERR_HIP;
        return e == hipSuccess;
    };
    if (ok(hipMalloc((void**)&d_flux, nb)) && ok(hipMalloc((void**)&d_flim, nb)) && ok(hipMalloc((void**)&d_phi, nb)) &&
        (!vol || ok(hipMalloc((void**)&d_vol, nb)))) {
        ok(hipMemcpy(d_flux, flux, nb, hipMemcpyHostToDevice));
        ok(hipMemcpy(d_flim, flim, nb, hipMemcpyHostToDevice));
        if (vol) ok(hipMemcpy(d_vol, vol, nb, hipMemcpyHostToDevice));
        double ratio = 0.0;
        if (fcmin > 0.0) {
            const double a = (2.0 * fcmin - 1.0) * (2.0 * fcmin - 1.0);      // VmaxLumFunc.py:164
            ratio = std::fabs(a / (1.0 - a));
        }
        if (rc == LF_OK) {
            hipLaunchKernelGGL(lf::veff_weights, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_flux, d_flim, d_vol, vol_all, pref0,
                               alpha, ratio, fcmin > 0.0 ? 1 : 0, (long long)n, d_phi);
            ok(hipGetLastError());
        }
        if (rc == LF_OK && nbin > 0) {
            const size_t sb = (size_t)(nboot + 1) * nbin * sizeof(double);
            if (ok(hipMalloc((void**)&d_bin, (size_t)n * sizeof(int))) && ok(hipMalloc((void**)&d_sums, sb)) &&
                (!boot_idx || ok(hipMalloc((void**)&d_idx, (size_t)n * nboot * sizeof(long long))))) {
                ok(hipMemcpy(d_bin, bin_of, (size_t)n * sizeof(int), hipMemcpyHostToDevice));
                ok(hipMemset(d_sums, 0, sb));
                if (boot_idx) ok(hipMemcpy(d_idx, boot_idx, (size_t)n * nboot * sizeof(long long), hipMemcpyHostToDevice));
                if (rc == LF_OK) {
                    const unsigned gx = (unsigned)std::min<int64_t>((n + 255) / 256, 1024);
                    hipLaunchKernelGGL(lf::veff_bins, dim3(gx, (unsigned)(nboot + 1)), dim3(256), 0, 0, d_phi, d_bin, (long long)n, nbin,
                                       d_idx, (unsigned long long)seed, d_sums);
                    ok(hipGetLastError());
                    ok(hipMemcpy(sums, d_sums, sb, hipMemcpyDeviceToHost));
                }
            }
        }
        if (rc == LF_OK) ok(hipMemcpy(phi, d_phi, nb, hipMemcpyDeviceToHost));
    }
    hipFree(d_flux); hipFree(d_flim); hipFree(d_vol); hipFree(d_phi); hipFree(d_sums); hipFree(d_bin); hipFree(d_idx);
    return rc;
}

int lf_last_launch(const lf_ctx* c, int32_t info[8]) {
    if (!c || !info) return LF_ERR_ARG;
    for (int i = 0; i < 8; ++i) info[i] = c->last_launch[i];
    return LF_OK;
}

int lf_form_counts(lf_ctx* c, int64_t counts[9]) {
    if (!c || !counts) return LF_ERR_ARG;
    for (int i = 0; i < lf::FORM_COUNT; ++i) counts[i] = 0;
    if (!c->d_forms) return LF_OK;
    LF_HIP(c, hipSetDevice(c->device));
    LF_HIP(c, hipDeviceSynchronize());
    unsigned long long h[lf::FORM_COUNT];
    LF_HIP(c, hipMemcpy(h, c->d_forms, sizeof(h), hipMemcpyDeviceToHost));
    for (int i = 0; i < lf::FORM_COUNT; ++i) counts[i] = (int64_t)h[i];
    return LF_OK;
}

int lf_set_option(lf_ctx* c, const char* key, int64_t value) {
    if (!c || !key) return LF_ERR_ARG;
    if (std::strcmp(key, "geometry") == 0) {
        if (value < -1 || value >= NGEO) {
            c->err = "geometry must be -1 (auto) or an index below " + std::to_string(NGEO);
            return LF_ERR_ARG;
        }
        c->opt_geometry = value;
        return LF_OK;
    }
    if (std::strcmp(key, "taper") == 0) {
        c->opt_taper = value != 0;
        return LF_OK;
    }
    if (std::strcmp(key, "compress") == 0) {
        if (value != 0) {
            hipSetDevice(c->device);
            const bool fresh = !c->cmp.built;
            int rc = build_compressed(c);
            if (rc != LF_OK) return rc;
            // The bins were accepted on a SAMPLED estimate of the interpolation error (lf_compress.h).  Before a freshly
            // built compressed catalogue may serve, it must also reproduce the direct path on walkers drawn from THIS
            // context's prior box: 64 theta rows (deterministic stream), both paths, lnprob to 1e-12 and the same -inf
            // pattern - otherwise the option is refused and the compressed tables are dropped.
            if (fresh && c->cmp.built && (rc = compress_selfcheck(c)) != LF_OK) {
                free_cmp(c->cmp);
                return rc;
            }
        }
        c->opt_compress = value != 0;
        return LF_OK;
    }
    if (std::strcmp(key, "count_forms") == 0) {
        hipSetDevice(c->device);
        if (value != 0 && !c->d_forms) LF_HIP(c, hipMalloc((void**)&c->d_forms, lf::FORM_COUNT * sizeof(unsigned long long)));
        if (c->d_forms) {
            LF_HIP(c, hipDeviceSynchronize());
            LF_HIP(c, hipMemset(c->d_forms, 0, lf::FORM_COUNT * sizeof(unsigned long long)));
        }
        c->kc.forms = value != 0 ? c->d_forms : nullptr;
        return LF_OK;
    }
    if (std::strcmp(key, "fuse") == 0) {
        c->opt_fuse = value != 0;
        return LF_OK;
    }
    if (std::strcmp(key, "fuse_step") == 0) {
        c->opt_fuse_step = value != 0;
        return LF_OK;
    }
    if (std::strcmp(key, "poll") == 0) {
        c->opt_poll = value != 0;
        return LF_OK;
    }
    if (std::strcmp(key, "profile_every") == 0) {
        c->opt_profile_every = value < 1 ? 1 : value;
        c->prof_tick = 0;
        return LF_OK;
    }
    if (std::strcmp(key, "profile_span") == 0) {
        c->opt_profile_span = value < 1 ? 1 : value;
        c->prof_tick = 0;
        return LF_OK;
    }
    if (std::strcmp(key, "free_st") == 0) {
        if (value != 0 && value != 2 && value != 4 && value != 8) {
            c->err = "free_st must be 0 (auto), 2, 4 or 8";
            return LF_ERR_ARG;
        }
        c->opt_free_st = value;
        return LF_OK;
    }
    if (std::strcmp(key, "cells") == 0) {
        c->opt_cells = value != 0;
        c->kc.cells = c->opt_cells && c->ncell > 0;
        return LF_OK;
    }
    if (std::strcmp(key, "persistent") == 0) {
        c->opt_persistent = value < 0 ? 0 : (value > 2 ? 2 : value);
        return LF_OK;
    }
    if (std::strcmp(key, "tables") == 0) {
        c->kc.tables = value != 0;
        return LF_OK;
    }
    if (std::strcmp(key, "specialise") == 0) {
        c->kc.specialise = value != 0;
        return LF_OK;
    }
    if (std::strcmp(key, "grid_shortcut") == 0) {
        c->opt_grid_shortcut = value != 0;
        return LF_OK;
    }
    if (std::strcmp(key, "compress_grid") == 0) {
        c->opt_compress_grid = value != 0;
        return LF_OK;
    }
    if (std::strcmp(key, "grid_share") == 0) {
        // value = part + parts * 65536: integrate only the node chunks c with c % parts == part (source-sharded ranks,
        // whose lnprob values are summed); 0 or parts <= 1 = the whole grid
        const int64_t parts = value >> 16, part = value & 0xffff;
        if (value < 0 || (parts > 1 && part >= parts)) {
            c->err = "grid_share must be part + 65536 * parts with part < parts";
            return LF_ERR_ARG;
        }
        c->kc.grid_parts = parts > 1 ? (int)parts : 1;
        c->kc.grid_part = parts > 1 ? (int)part : 0;
        return LF_OK;
    }
    if (std::strcmp(key, "skip_grid") == 0) {
        c->opt_skip_grid = value != 0;
        return LF_OK;
    }
    if (std::strcmp(key, "walker_tile") == 0) {
        if (value < 0 || value > 64) {
            c->err = "walker_tile must be 0 (auto) .. 64";
            return LF_ERR_ARG;
        }
        c->opt_walker_tile = value;
        return LF_OK;
    }
    c->err = std::string("unknown option ") + key;
    return LF_ERR_ARG;
}

int64_t lf_compress_keys(int kind, const double* params, const double* key, const double* wt, int64_t n,
                         double* node, double* weight, int64_t cap, double* bound) {
    if (!params || (!key && n > 0) || n < 0 || (kind != 0 && kind != 1)) return LF_ERR_ARG;
    lfc::Model m{};
    m.kind = kind;
    if (kind == 0) {
        m.fc_ratio = params[0];
        m.alpha_lo = params[1];
        m.alpha_hi = params[2];
        m.flim_lo = params[3];
        m.flim_hi = params[4];
    } else {
        m.L_lo = params[0];
        m.L_hi = params[1];
        for (int i = 0; i < 3; ++i) m.piv[i] = params[2 + i];
    }
    lfc::Out out;
    if (!lfc::compress_field(m, key, wt, n, out)) return LF_ERR_ARG;
    if (bound) *bound = out.bound;
    const int64_t cnt = (int64_t)out.node.size();
    if (cnt > cap || !node || !weight) return cnt;
    std::copy(out.node.begin(), out.node.end(), node);
    std::copy(out.weight.begin(), out.weight.end(), weight);
    return cnt;
}

int64_t lf_compress_grid(const double* params, int S, const double* L, const double* wL, const double* ck, const double* Dk,
                         double* u, int32_t* row0, int32_t* nrows, int32_t* off, double* omega, int64_t cap_bins,
                         int64_t cap_omega, double* bound) {
    if (!params || !L || !wL || !ck || !Dk || S < 2) return LF_ERR_ARG;
    lfc::Model m{};
    m.kind = 2;
    m.fc_ratio = params[0];
    m.alpha_lo = params[1];
    m.alpha_hi = params[2];
    m.flim_lo = params[3];
    m.flim_hi = params[4];
    lfc::GridOut g;
    if (!lfc::compress_grid(m, S, L, wL, ck, Dk, g)) return LF_ERR_ARG;
    if (bound) *bound = g.bound;
    if (g.nb > cap_bins || (int64_t)g.omega.size() > cap_omega || !u || !row0 || !nrows || !off || !omega) return g.nb;
    std::copy(g.u.begin(), g.u.end(), u);
    std::copy(g.row0.begin(), g.row0.end(), row0);
    std::copy(g.nrows.begin(), g.nrows.end(), nrows);
    std::copy(g.off.begin(), g.off.end(), off);
    std::copy(g.omega.begin(), g.omega.end(), omega);
    return g.nb;
}

int64_t lf_grid_bins(const double* params, int S, const double* L, const double* wL, const double* ck, const double* Dk,
                     double* edges, double* rec, int32_t* rows, double* omega, int64_t cap_bins, int64_t cap_omega, double* margin) {
    if (!params || !L || !wL || !ck || !Dk || S < 2) return LF_ERR_ARG;
    if (!(params[1] > 0.0) || !(params[3] > 0.0)) return LF_ERR_ARG;
    const lfq::Box bx{std::sqrt(params[0]), params[1], params[2], std::log10(params[3]) + LF_FREF, std::log10(params[4]) + LF_FREF};
    lfq::GridQ g;
    if (!lfq::build_gridq(bx, S, L, wL, ck, Dk, LF_FREF, LF_LREF, g)) return LF_ERR_ARG;
    if (margin) *margin = g.margin;
    if (g.nb > cap_bins || (int64_t)g.omega.size() > cap_omega || !edges || !rec || !rows || !omega) return g.nb;
    std::copy(g.edges.begin(), g.edges.end(), edges);
    std::copy(g.rec.begin(), g.rec.end(), rec);
    std::copy(g.rows.begin(), g.rows.end(), rows);
    std::copy(g.omega.begin(), g.omega.end(), omega);
    return g.nb;
}

/* ---------------------------------------------------------------------------------------------
 * device-resident ensemble sampler
 * ------------------------------------------------------------------------------------------- */
struct lf_sampler {
    lf_ctx* ctx = nullptr;
    int W = 0, ndim = 0;
    double a = 2.0;
    uint64_t seed = 0, step = 0;
    int64_t cap = 0, t = 0;
    double *d_pos = nullptr, *d_lnp = nullptr, *d_prop = nullptr, *d_zz = nullptr, *d_newlp = nullptr;
    double *d_chain = nullptr, *d_chain_lnp = nullptr;
    long long* d_nacc = nullptr;
    bool started = false;
};

lf_sampler* lf_sampler_create(lf_ctx* c, int nwalkers, double a, uint64_t seed, int64_t capacity_steps) {
    if (!c) return nullptr;
    if (nwalkers < 2 || (nwalkers & 1) || capacity_steps < 1 || !(a > 1.0)) {
        c->err = "lf_sampler_create: nwalkers must be even and >= 2, a > 1, capacity_steps >= 1";
        return nullptr;
    }
    if (hipSetDevice(c->device) != hipSuccess) return nullptr;
    lf_sampler* sm = new (std::nothrow) lf_sampler();
    if (!sm) return nullptr;
    sm->ctx = c;
    sm->W = nwalkers;
    sm->ndim = c->kc.ndim;
    sm->a = a;
    sm->seed = seed;
    sm->cap = capacity_steps;
    const size_t W = (size_t)nwalkers, nd = (size_t)sm->ndim, cap = (size_t)capacity_steps;
    bool ok = hipMalloc((void**)&sm->d_pos, W * nd * 8) == hipSuccess && hipMalloc((void**)&sm->d_lnp, W * 8) == hipSuccess &&
              hipMalloc((void**)&sm->d_prop, W * nd * 8) == hipSuccess && hipMalloc((void**)&sm->d_zz, W * 8) == hipSuccess &&
              hipMalloc((void**)&sm->d_newlp, W * 8) == hipSuccess &&
              hipMalloc((void**)&sm->d_chain, W * cap * nd * 8) == hipSuccess &&
              hipMalloc((void**)&sm->d_chain_lnp, W * cap * 8) == hipSuccess &&
              hipMalloc((void**)&sm->d_nacc, W * sizeof(long long)) == hipSuccess;
    if (!ok) {
        c->err = "lf_sampler_create: device allocation failed";
        lf_sampler_destroy(sm);
        return nullptr;
    }
    return sm;
}

void lf_sampler_destroy(lf_sampler* sm) {
    if (!sm) return;
    if (sm->ctx) {
        hipSetDevice(sm->ctx->device);
        hipDeviceSynchronize();
    }
    hipFree(sm->d_pos); hipFree(sm->d_lnp); hipFree(sm->d_prop); hipFree(sm->d_zz); hipFree(sm->d_newlp);
    hipFree(sm->d_chain); hipFree(sm->d_chain_lnp); hipFree(sm->d_nacc);
    delete sm;
}

int lf_sampler_start(lf_sampler* sm, const double* pos, const double* lnprob0) {
    if (!sm || !pos) return LF_ERR_ARG;
    lf_ctx* c = sm->ctx;
    LF_HIP(c, hipSetDevice(c->device));
    const size_t W = (size_t)sm->W, nd = (size_t)sm->ndim;
    LF_HIP(c, hipMemcpy(sm->d_pos, pos, W * nd * 8, hipMemcpyHostToDevice));
    LF_HIP(c, hipMemset(sm->d_nacc, 0, W * sizeof(long long)));
    if (lnprob0) {
        LF_HIP(c, hipMemcpy(sm->d_lnp, lnprob0, W * 8, hipMemcpyHostToDevice));
    } else {
        int rc = enqueue(c, sm->d_pos, sm->W, sm->d_lnp, nullptr, nullptr, c->stream);
        if (rc != LF_OK) return rc;
        LF_HIP(c, hipStreamSynchronize(c->stream));
    }
    sm->step = 0;
    sm->t = 0;
    sm->started = true;
    return LF_OK;
}

int lf_sampler_run(lf_sampler* sm, int64_t nsteps, void* hip_stream) {
    if (!sm || nsteps < 0) return LF_ERR_ARG;
    lf_ctx* c = sm->ctx;
    if (!sm->started) {
        c->err = "lf_sampler_run: call lf_sampler_start first";
        return LF_ERR_ARG;
    }
    if (sm->t + nsteps > sm->cap) {
        c->err = "lf_sampler_run: chain capacity exceeded";
        return LF_ERR_ARG;
    }
    LF_HIP(c, hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    const int halfW = sm->W / 2;
    for (int64_t it = 0; it < nsteps; ++it) {
        for (int half = 0; half < 2; ++half) {
            lf::StepArgs sp{1, half, halfW, sm->ndim, sm->step, sm->seed, sm->a, sm->d_pos, sm->d_prop, sm->d_zz};
            lf::AcceptArgs ap{1, half, halfW, sm->ndim, sm->step, sm->seed, (long long)sm->t, (long long)sm->cap,
                              sm->d_pos, sm->d_lnp, sm->d_prop, sm->d_zz, sm->d_nacc, sm->d_chain, sm->d_chain_lnp};
            int rc = enqueue(c, nullptr, halfW, sm->d_newlp, nullptr, nullptr, s, &sp, &ap);
            if (rc != LF_OK) return rc;
        }
        sm->step += 1;
        sm->t += 1;
    }
    return LF_OK;
}

int lf_sampler_read(lf_sampler* sm, double* chain, double* chain_lnprob, int64_t* naccepted, double* pos, double* lnprob) {
    if (!sm) return LF_ERR_ARG;
    lf_ctx* c = sm->ctx;
    LF_HIP(c, hipSetDevice(c->device));
    LF_HIP(c, hipDeviceSynchronize());
    const size_t W = (size_t)sm->W, nd = (size_t)sm->ndim, cap = (size_t)sm->cap, t = (size_t)sm->t;
    // device chain is [W][cap][ndim]; the caller's is [W][t][ndim]: one strided copy each
    // (packed on the device first, then ONE copy to the host: a strided copy into pageable host memory goes row by row)
    if (t > 0 && (chain || chain_lnprob)) {
        double* tmp = nullptr;
        LF_HIP(c, hipMalloc((void**)&tmp, W * t * nd * 8));
        int rc = LF_OK;
        if (chain) {
            if (hipMemcpy2D(tmp, t * nd * 8, sm->d_chain, cap * nd * 8, t * nd * 8, W, hipMemcpyDeviceToDevice) != hipSuccess ||
                hipMemcpy(chain, tmp, W * t * nd * 8, hipMemcpyDeviceToHost) != hipSuccess)
                rc = LF_ERR_HIP;
        }
        if (rc == LF_OK && chain_lnprob) {
            if (hipMemcpy2D(tmp, t * 8, sm->d_chain_lnp, cap * 8, t * 8, W, hipMemcpyDeviceToDevice) != hipSuccess ||
                hipMemcpy(chain_lnprob, tmp, W * t * 8, hipMemcpyDeviceToHost) != hipSuccess)
                rc = LF_ERR_HIP;
        }
        hipFree(tmp);
        if (rc != LF_OK) {
            c->err = "lf_sampler_read: copy failed";
            return rc;
        }
    }
    if (naccepted) LF_HIP(c, hipMemcpy(naccepted, sm->d_nacc, W * sizeof(long long), hipMemcpyDeviceToHost));
    if (pos) LF_HIP(c, hipMemcpy(pos, sm->d_pos, W * nd * 8, hipMemcpyDeviceToHost));
    if (lnprob) LF_HIP(c, hipMemcpy(lnprob, sm->d_lnp, W * 8, hipMemcpyDeviceToHost));
    return LF_OK;
}

int64_t lf_sampler_steps(const lf_sampler* sm) { return sm ? sm->t : LF_ERR_ARG; }

int lf_sampler_half_eval(lf_sampler* sm, int half, int lo, int hi, double* d_newlp, void* hip_stream) {
    if (!sm || !d_newlp || half < 0 || half > 1 || lo < 0 || hi < lo || hi > sm->W / 2) return LF_ERR_ARG;
    lf_ctx* c = sm->ctx;
    if (!sm->started || sm->t >= sm->cap) {
        c->err = "lf_sampler_half_eval: not started, or chain capacity exceeded";
        return LF_ERR_ARG;
    }
    LF_HIP(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)hip_stream;          // as given: NULL is the default stream (cf. lf_lnprob_batch_device)
    const int halfW = sm->W / 2;
    lf::StepArgs sp{1, half, halfW, sm->ndim, sm->step, sm->seed, sm->a, sm->d_pos, sm->d_prop, sm->d_zz};
    hipLaunchKernelGGL(lf::lf_propose, dim3((halfW + 7) / 8), dim3(64), 0, s, sp);
    LF_HIP(c, hipGetLastError());
    if (hi > lo)
        return enqueue(c, sm->d_prop + (size_t)lo * sm->ndim, hi - lo, d_newlp + lo, nullptr, nullptr, s);
    return LF_OK;
}

int lf_sampler_half_accept(lf_sampler* sm, int half, const double* d_newlp, void* hip_stream) {
    if (!sm || !d_newlp || half < 0 || half > 1) return LF_ERR_ARG;
    lf_ctx* c = sm->ctx;
    if (!sm->started || sm->t >= sm->cap) {
        c->err = "lf_sampler_half_accept: not started, or chain capacity exceeded";
        return LF_ERR_ARG;
    }
    LF_HIP(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)hip_stream;          // as given: NULL is the default stream (cf. lf_lnprob_batch_device)
    const int halfW = sm->W / 2;
    lf::AcceptArgs ap{1, half, halfW, sm->ndim, sm->step, sm->seed, (long long)sm->t, (long long)sm->cap,
                      sm->d_pos, sm->d_lnp, sm->d_prop, sm->d_zz, sm->d_nacc, sm->d_chain, sm->d_chain_lnp};
    hipLaunchKernelGGL(lf::lf_accept, dim3(halfW), dim3(64), 0, s, ap, d_newlp);
    LF_HIP(c, hipGetLastError());
    if (half == 1) {
        sm->step += 1;
        sm->t += 1;
    }
    return LF_OK;
}

}  // extern "C"
