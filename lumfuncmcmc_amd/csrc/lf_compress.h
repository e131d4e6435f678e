// lf_compress.h - host side of the compressed-catalogue path (opt-in, lf_set_option "compress").
//
// What is left per (walker, source) after the hoisting of lf_kernels.h is a walker-dependent function of ONE
// source coordinate:
//     FREE    g_w(logf_i - lF_wf) = ln fc(alpha_C x) / (1 - e^(-10^(x - b)))     (VmaxLumFunc.py:118-127, :141)
//     ZEVOL   P_i h_w(z_i),  h_w(z) = 10^(42 - L*_w(z)),  P_i = 10^(lum_i - 42)  (lumfuncmcmc_z.py:66, :45-67)
// Both are analytic in that coordinate.  On a bin [a, b] of the coordinate, let p be the polynomial of degree
// K-1 that interpolates the function at the K Chebyshev nodes x_n of the bin.  For any polynomial q of degree < K
//     sum_{i in bin} P_i q(x_i) = sum_n omega_n q(x_n),   omega_n = (1/K) (M_0 + 2 sum_{k>=1} T_k(t_n) M_k),
//     M_k = sum_i P_i T_k(t_i)           (Chebyshev moments of the bin's sources: walker-INDEPENDENT)
// so  | sum_i P_i f(x_i) - sum_n omega_n f(x_n) | <= sum_i P_i |f - p|(x_i) <= (sum_i P_i) max_bin |f - p|:
// the bin's sources can be replaced by K weighted pseudo-sources with a relative error of the bin sum of at
// most max|f - p| / min|f|, whatever the distribution of the sources inside the bin.  The bins are chosen here
// (recursive halving) until that bound, evaluated in long double over the corners of the prior box - the
// walkers for which f varies fastest - is below COMPRESS_TOL, i.e. below the rounding of the direct sum.
// A walker outside the prior box returns -inf before any sum is looked at; walkers that need the per-source
// underflow checks (App. B-5) are summed over the real catalogue (rescue workgroups, lf_kernels.h).
//
// Pure host C++ (no HIP): also exported as lf_compress_keys() so that the CPU tests can check it.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace lfc {

constexpr int K = 16;                      // pseudo-sources per bin
constexpr double COMPRESS_TOL = 1.0e-16;   // bound on the relative error of a bin sum
constexpr int MAX_DEPTH = 12;
constexpr double GRID_FLOOR = 1.0e-25;     // kind 2: completeness values below this are bounded absolutely

struct Model {
    int kind;                              // 0 FREE sources, 1 ZEVOL sources, 2 FREE integration grid (see compress_grid)
    double fc_ratio;                       // FREE: |a / (1 - a)|, a = (2 fcmin - 1)^2
    double alpha_lo, alpha_hi;             // FREE: prior box of the completeness slope
    double flim_lo, flim_hi;               // FREE: prior box of Flim (units of 1e-17)
    double L_lo, L_hi;                     // ZEVOL: prior box of L*(z_pivot)
    double piv[3];                         // ZEVOL: pivots
};

// Walkers the bound is evaluated for.  ZEVOL: the 8 corners of the (L1, L2, L3) box - the exponent's slope is
// linear in them, so its extremes sit on corners.  FREE (sources and grid): NALPHA slopes from alpha_lo to
// alpha_hi times `nshift` values of lF = log10(1e-17 Flim) from flim_lo to flim_hi; lF only translates the
// function along the coordinate, so the shifts are spaced at most half a bin width apart (every stretch of the
// function is seen by a window of this bin's width, up to half a window).
constexpr int NALPHA = 7;
constexpr int MAX_SHIFTS = 96;
inline int nshifts(const Model& m, double width) {
    const double span = std::log10(std::max(m.flim_hi, 1e-6) / std::max(std::min(m.flim_lo, m.flim_hi), 1e-6));
    return std::max(2, std::min(MAX_SHIFTS, (int)std::ceil(span / (0.5 * width)) + 1));
}
inline int ncorners(const Model& m, double width) { return m.kind == 1 ? 8 : NALPHA * nshifts(m, width); }

// the function of the source coordinate for prior-box corner `corner`, in long double
inline long double feval(const Model& m, int corner, long double x, int nshift = 5) {
    if (m.kind != 1) {
        const int ia = corner / nshift, il = corner % nshift;
        const long double alo = std::max(m.alpha_lo, 1e-3);
        const long double aC = alo + (std::max<long double>(m.alpha_hi, alo) - alo) * ia / (long double)(NALPHA - 1);
        const long double flo = std::max(m.flim_lo, 1e-6), fhi = std::max<long double>(m.flim_hi, flo);
        const long double lF = log10l(flo) + (log10l(fhi) - log10l(flo)) * il / (long double)(nshift - 1) - 17.0L;
        const long double xs = x - lF;
        const long double num = aC * xs;
        // ln fc, fc = (1 + num / sqrt(1 + num^2)) / 2, without the cancellations of the literal form:
        // fc = 1 / (2 s (s - num)) for num < 0,  1 - fc = 1 / (2 s (s + num)) for num >= 0
        const long double s = sqrtl(1.0L + num * num);
        const long double lnfc = num >= 0 ? log1pl(-0.5L / (s * (s + num))) : -logl(2.0L * s * (s - num));
        const long double b = -sqrtl((long double)m.fc_ratio / (aC * aC));
        const long double u = powl(10.0L, xs - b);
        const long double g = lnfc / -expm1l(-u);
        return m.kind == 2 ? expl(g) : g;               // the grid integrates fc^(1/decay) itself
    }
    // Lagrange form of the quadratic through the pivots (lumfuncmcmc_z.py:26-43 solves the same system), about
    // the middle of the box so that the extrapolated terms cancel as little as possible
    const long double ymid = 0.5L * ((long double)m.L_lo + m.L_hi), yh = 0.5L * ((long double)m.L_hi - m.L_lo);
    const long double d[3] = {(corner & 1) ? yh : -yh, (corner & 2) ? yh : -yh, (corner & 4) ? yh : -yh};
    const long double z1 = m.piv[0], z2 = m.piv[1], z3 = m.piv[2];
    const long double dL = d[0] * (x - z2) * (x - z3) / ((z1 - z2) * (z1 - z3)) +
                           d[1] * (x - z1) * (x - z3) / ((z2 - z1) * (z2 - z3)) +
                           d[2] * (x - z1) * (x - z2) / ((z3 - z1) * (z3 - z2));
    return powl(10.0L, (42.0L - ymid) - dL);
}

// bound on the relative error of a bin sum on [a, b]: max over corners of max|f - p| / min|f|
inline double bin_error(const Model& m, double a, double b) {
    const long double PI = 3.141592653589793238462643383279502884L;
    const long double mid = 0.5L * ((long double)a + b), half = 0.5L * ((long double)b - a);
    long double ct[K][K];                  // cos(k theta_n)
    long double xn[K];
    for (int n = 0; n < K; ++n) {
        const long double th = PI * (n + 0.5L) / K;
        xn[n] = mid + half * cosl(th);
        for (int k = 0; k < K; ++k) ct[k][n] = cosl(k * th);
    }
    const int M = 2 * K + 1;
    long double worst = 0.0L;
    const int nsh = nshifts(m, b - a);
    for (int cr = 0; cr < ncorners(m, b - a); ++cr) {
        long double f[K], c[K];
        for (int n = 0; n < K; ++n) f[n] = feval(m, cr, xn[n], nsh);
        for (int k = 0; k < K; ++k) {
            long double s = 0.0L;
            for (int n = 0; n < K; ++n) s += f[n] * ct[k][n];
            c[k] = s * (k == 0 ? 1.0L : 2.0L) / K;
        }
        long double emax = 0.0L, fmin = HUGE_VALL;
        for (int j = 0; j < M; ++j) {
            const long double t = -1.0L + 2.0L * j / (M - 1);
            long double b1 = 0.0L, b2 = 0.0L;      // Clenshaw
            for (int k = K - 1; k >= 1; --k) {
                const long double b0 = 2.0L * t * b1 - b2 + c[k];
                b2 = b1;
                b1 = b0;
            }
            const long double p = t * b1 - b2 + c[0];
            const long double fx = feval(m, cr, mid + half * t, nsh);
            emax = std::max(emax, fabsl(p - fx));
            fmin = std::min(fmin, fabsl(fx));
        }
        if (m.kind == 2) {
            // F = fc^(1/decay) in (0, 1]: a double evaluation of F = e^g carries |g| ulp of relative noise, so that is
            // the yardstick; below GRID_FLOOR the bound is absolute (such nodes cannot matter)
            const long double fm = std::max(fmin, (long double)GRID_FLOOR);
            if (!std::isfinite((double)emax)) return HUGE_VAL;
            worst = std::max(worst, emax / (fm * (1.0L + fabsl(logl(fm)))));
            if (worst > COMPRESS_TOL) return (double)worst;
            continue;
        }
        if (!(fmin > 0.0L) || !std::isfinite((double)emax)) return HUGE_VAL;
        worst = std::max(worst, emax / fmin);
        if (worst > COMPRESS_TOL) return (double)worst;      // the bin will be halved: no need for the other walkers
    }
    return (double)worst;
}

// bins of [lo, hi]: start from width0, halve a bin until its error bound passes.  Returns the worst accepted
// bound, or HUGE_VAL if some bin still fails at MAX_DEPTH.
inline double make_bins(const Model& m, double lo, double hi, double width0, std::vector<double>& edges) {
    edges.clear();
    edges.push_back(lo);
    if (!(hi > lo)) return 0.0;
    const int n0 = std::max(1, (int)std::ceil((hi - lo) / width0));
    double worst = 0.0;
    struct Item { double a, b; int depth; };
    for (int i = 0; i < n0; ++i) {
        std::vector<Item> stack;
        stack.push_back({lo + (hi - lo) * i / n0, i + 1 == n0 ? hi : lo + (hi - lo) * (i + 1) / n0, 0});
        while (!stack.empty()) {
            const Item it = stack.back();
            stack.pop_back();
            const double e = bin_error(m, it.a, it.b);
            if (e <= COMPRESS_TOL) {
                worst = std::max(worst, e);
                edges.push_back(it.b);
            } else if (it.depth >= MAX_DEPTH) {
                return HUGE_VAL;
            } else {
                const double mid = 0.5 * (it.a + it.b);
                stack.push_back({mid, it.b, it.depth + 1});     // popped second: edges stay ascending
                stack.push_back({it.a, mid, it.depth + 1});
            }
        }
    }
    return worst;
}

struct Out {
    std::vector<double> node, weight;
    int nbins = 0;
    double bound = 0.0;                    // worst accepted bin_error
};

// Compress one field: keys[n] (finite), wt[n] or NULL (= 1).  Appends to out.  Returns false when the bins
// cannot be made accurate enough (the caller then leaves the option off).
// `shared` (optional): bin edges validated once for a range that contains every field's keys (Bins below), so that
// the fields of a catalogue do not each pay for the validation.
struct Bins {
    std::vector<double> edges;
    double bound = 0.0;
    bool ok = false;
};
inline Bins shared_bins(const Model& m, double lo, double hi) {
    Bins b;
    if (!(hi > lo) || !std::isfinite(lo) || !std::isfinite(hi)) return b;
    b.bound = make_bins(m, lo, hi, m.kind == 0 ? 0.2 : 0.04, b.edges);
    b.ok = b.bound < HUGE_VAL;
    return b;
}

inline bool compress_field(const Model& m, const double* key, const double* wt, int64_t n, Out& out, const Bins* shared = nullptr) {
    if (n <= 0) return true;
    double lo = key[0], hi = key[0];
    for (int64_t i = 0; i < n; ++i) {
        if (!std::isfinite(key[i])) return false;
        lo = std::min(lo, key[i]);
        hi = std::max(hi, key[i]);
    }
    if (n <= K || !(hi > lo)) {            // nothing to gain: keep the sources as they are
        for (int64_t i = 0; i < n; ++i) {
            out.node.push_back(key[i]);
            out.weight.push_back(wt ? wt[i] : 1.0);
        }
        return true;
    }
    std::vector<double> edges;
    double bound;
    if (shared && shared->ok && shared->edges.front() <= lo && shared->edges.back() >= hi) {
        edges = shared->edges;
        bound = shared->bound;
    } else {
        const double width0 = m.kind == 0 ? 0.2 : 0.04;   // refined by halving where the bound asks for it
        bound = make_bins(m, lo, hi, width0, edges);
    }
    if (!(bound < HUGE_VAL)) return false;
    out.bound = std::max(out.bound, bound);
    const int nb = (int)edges.size() - 1;
    std::vector<int> bin((size_t)n);
    std::vector<int64_t> cnt((size_t)nb, 0);
    for (int64_t i = 0; i < n; ++i) {
        int b = (int)(std::upper_bound(edges.begin(), edges.end(), key[i]) - edges.begin()) - 1;
        b = std::min(std::max(b, 0), nb - 1);
        bin[(size_t)i] = b;
        ++cnt[(size_t)b];
    }
    std::vector<long double> mom((size_t)nb * K, 0.0L);
    for (int64_t i = 0; i < n; ++i) {
        const int b = bin[(size_t)i];
        if (cnt[(size_t)b] <= K) continue;
        const long double mid = 0.5L * ((long double)edges[b] + edges[b + 1]), half = 0.5L * ((long double)edges[b + 1] - edges[b]);
        const long double t = std::min(1.0L, std::max(-1.0L, ((long double)key[i] - mid) / half));
        const long double p = wt ? (long double)wt[i] : 1.0L;
        long double* M = &mom[(size_t)b * K];
        long double t0 = 1.0L, t1 = t;
        M[0] += p;
        M[1] += p * t;
        for (int k = 2; k < K; ++k) {
            const long double t2 = 2.0L * t * t1 - t0;
            M[k] += p * t2;
            t0 = t1;
            t1 = t2;
        }
    }
    // small bins keep their sources; they are gathered per bin so that the output stays sorted by bin
    std::vector<std::vector<int64_t>> raw((size_t)nb);
    for (int64_t i = 0; i < n; ++i)
        if (cnt[(size_t)bin[(size_t)i]] <= K) raw[(size_t)bin[(size_t)i]].push_back(i);
    const long double PI = 3.141592653589793238462643383279502884L;
    for (int b = 0; b < nb; ++b) {
        if (cnt[(size_t)b] == 0) continue;
        ++out.nbins;
        if (cnt[(size_t)b] <= K) {
            for (int64_t i : raw[(size_t)b]) {
                out.node.push_back(key[i]);
                out.weight.push_back(wt ? wt[i] : 1.0);
            }
            continue;
        }
        const long double mid = 0.5L * ((long double)edges[b] + edges[b + 1]), half = 0.5L * ((long double)edges[b + 1] - edges[b]);
        const long double* M = &mom[(size_t)b * K];
        for (int nn = 0; nn < K; ++nn) {
            const long double th = PI * (nn + 0.5L) / K;
            long double w = M[0];
            for (int k = 1; k < K; ++k) w += 2.0L * cosl(k * th) * M[k];
            out.node.push_back((double)(mid + half * cosl(th)));
            out.weight.push_back((double)(w / K));
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------
// The FREE expected-count integral on a SEPARABLE grid (every redshift column has the same luminosity nodes:
// G[j][k] = L_j, the default min_comp_frac = 0 case):
//     B_w = sum_j wL_j T_w(L_j) sum_k c_k sum_f om_f F_wf(L_j - D_k),      F_wf(u) = fc^(1/decay) at flux 10^u,
// c_k = trapezoid weight x dV/dz, D_k = log10(4 pi DL(z_k)^2).  The walker-dependent completeness depends on the
// lattice point only through u_jk = L_j - D_k, so the S^2 lattice points are binned in u and, ROW BY ROW, replaced
// by the K Chebyshev nodes of their bin (weights from the row's moments in the bin, exactly as for the sources):
//     sum_k c_k F(u_jk) ~ sum_b sum_n Omega[b][n][j] F(u_bn),   B_w ~ sum_b sum_n F-sum(u_bn) sum_j T_w(L_j) Omega'[b][n][j]
// (Omega' = wL_j Omega).  All rows share the nodes of a bin: 16 x (number of bins) completeness evaluations per
// field instead of S^2, plus a short dot product with T_w over the ~20 rows that cross the bin.
struct GridOut {
    std::vector<double> u;              // [nb * K] node positions (log10 flux)
    std::vector<int> row0, nrows, off;  // per bin: first row, row count, offset into omega
    std::vector<double> omega;          // per bin [row][node]
    double bound = 0.0;
    int nb = 0;
};

inline bool compress_grid(const Model& m, int S, const double* L, const double* wL, const double* ck, const double* Dk, GridOut& out) {
    if (S < 2) return false;
    double dmin = Dk[0], dmax = Dk[0];
    for (int k = 0; k < S; ++k) {
        if (!std::isfinite(Dk[k]) || !std::isfinite(ck[k])) return false;
        dmin = std::min(dmin, Dk[k]);
        dmax = std::max(dmax, Dk[k]);
    }
    double lmin = L[0], lmax = L[0];
    for (int j = 0; j < S; ++j) {
        if (!std::isfinite(L[j]) || !std::isfinite(wL[j])) return false;
        lmin = std::min(lmin, L[j]);
        lmax = std::max(lmax, L[j]);
    }
    const double lo = lmin - dmax, hi = lmax - dmin;
    if (!(hi > lo)) return false;
    std::vector<double> edges;
    const double bound = make_bins(m, lo, hi, 0.2, edges);
    if (!(bound < HUGE_VAL)) return false;
    out.bound = bound;
    const int nb = (int)edges.size() - 1;
    const long double PI = 3.141592653589793238462643383279502884L;
    for (int b = 0; b < nb; ++b) {
        const long double mid = 0.5L * ((long double)edges[b] + edges[b + 1]), half = 0.5L * ((long double)edges[b + 1] - edges[b]);
        const bool last = b == nb - 1;
        int j0 = S, j1 = -1;
        std::vector<long double> mom((size_t)S * K, 0.0L);
        for (int j = 0; j < S; ++j)
            for (int k = 0; k < S; ++k) {
                const double u = L[j] - Dk[k];                       // the very rounding the device tables use
                if (!(u >= edges[b] && (u < edges[b + 1] || (last && u <= edges[b + 1])))) continue;
                j0 = std::min(j0, j);
                j1 = std::max(j1, j);
                const long double t = std::min(1.0L, std::max(-1.0L, ((long double)u - mid) / half));
                long double* M = &mom[(size_t)j * K];
                long double t0 = 1.0L, t1 = t;
                M[0] += ck[k];
                M[1] += ck[k] * t;
                for (int q = 2; q < K; ++q) {
                    const long double t2 = 2.0L * t * t1 - t0;
                    M[q] += ck[k] * t2;
                    t0 = t1;
                    t1 = t2;
                }
            }
        if (j1 < j0) continue;                                        // no lattice point in this bin
        out.row0.push_back(j0);
        out.nrows.push_back(j1 - j0 + 1);
        out.off.push_back((int)out.omega.size());
        for (int n = 0; n < K; ++n) out.u.push_back((double)(mid + half * cosl(PI * (n + 0.5L) / K)));
        for (int j = j0; j <= j1; ++j)
            for (int n = 0; n < K; ++n) {
                const long double th = PI * (n + 0.5L) / K;
                const long double* M = &mom[(size_t)j * K];
                long double w = M[0];
                for (int q = 1; q < K; ++q) w += 2.0L * cosl(q * th) * M[q];
                out.omega.push_back((double)(wL[j] * w / K));
            }
        ++out.nb;
    }
    return out.nb > 0;
}

}  // namespace lfc
