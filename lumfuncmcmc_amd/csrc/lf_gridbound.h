// lf_gridbound.h - the FREE expected-count integral (piece B, lumfuncmcmc.py:373-377) on a SEPARABLE grid, summed over
// flux bins instead of lattice points, with a PROVEN error bound.  Pure host C++ (no HIP); exported as lf_grid_bins() so
// that tests/test_gridbound_cpu.py can recompute the bound independently.
//
// The sum.  On a separable grid (every redshift column has the same luminosity nodes: the default min_comp_frac = 0)
//     B_w = sum_j wL_j T_w(L_j) sum_k c_k sum_f om_f F_wf(L_j - D_k),    F_wf(x) = F_alpha(x - lF_wf),
//     F_alpha(u) = exp(Q(u; alpha)),   Q = g(alpha u) h(u + kappa / alpha),
//     g(v) = ln((1 + v / sqrt(1 + v^2)) / 2),  h(y) = 1 / (1 - exp(-10^y)),  kappa = sqrt(|a / (1 - a)|), a = (2 fcmin - 1)^2
// (VmaxLumFunc.py:118-127, :141, :164-167; c_k = trapezoid weight x dV/dz, D_k = log10(4 pi DL(z_k)^2), wL_j = trapezoid
// weight in log L, T_w the Schechter function).  The completeness depends on the lattice point only through the log flux
// x_jk = L_j - D_k.  The x axis is cut into bins; in a bin, ROW BY ROW, the lattice points are replaced by the bin's KQ
// Chebyshev nodes x_n with weights from the row's Chebyshev moments (walker-independent, made once in long double):
//     sum_{k: x_jk in bin} c_k q(x_jk) = sum_n omega_j[n] q(x_n)       EXACTLY for every polynomial q of degree < KQ,
// so with p the polynomial that interpolates F_wf at the nodes,
//     | sum_k c_k F_wf(x_jk) - sum_n omega_j[n] F_wf(x_n) | = | sum_k c_k (F_wf - p)(x_jk) | <= (sum_k c_k) max_bin |F_wf - p|
// (c_k >= 0), and since every term of B_w is positive:
//     if   max_bin |F_wf - p| <= EPS_REL min_bin F_wf + EPS_ABS   for every bin and every walker of the prior box,
//     then |B_w(bins) - B_w(lattice)| <= EPS_REL B_w + EPS_ABS B_w(completeness = 1).                                  (*)
//
// The bound on max|F - p| (what this file computes; not sampled over walkers - proven for the whole box).  F_alpha is
// analytic around the real axis (g: branch points at v = +-i, i.e. u = +-i / alpha; h: poles where 10^y = 2 pi i m, i.e.
// |Im y| = pi / (2 ln 10) = 0.68).  For a function analytic with |F| <= M inside the Bernstein ellipse E_rho of an interval
// (foci at its ends, semi-axes a (rho +- 1/rho) / 2, a = half width) its Chebyshev coefficients obey |a_k| <= 2 M rho^-k,
// and the interpolant at the KQ Chebyshev ROOTS differs from F by at most the tail plus its aliases
// (T_{2mK +- k}(x_n) = (-1)^m T_k(x_n)):
//     max |F - p| <= 2 sum_{k >= KQ} |a_k| <= 4 M rho^(1-KQ) / (rho - 1).                                              (**)
// M = exp(max over the ellipse of Re Q), by the maximum principle the max over its BOUNDARY, which is sampled at MTH
// points per half (Q(conj z) = conj Q(z)) and padded by a Lipschitz term: any boundary point is within r pi / (2 MTH) of
// a sample (|dz/dtheta| <= r = a (rho + 1/rho) / 2), so Re Q <= max_samples Re Q + Lz r pi / (2 MTH) with Lz >= |dQ/dz| on
// the region.  The walkers of the box are covered the same way: a bin [xa, xb] is the window of centre c = (xa + xb) / 2 -
// lF for the walker, so (alpha, c) ranges over a rectangle, which is cut into cells; a cell is checked at its centre and
// padded by Lz dc/2 + La da/2 (La >= |dQ/dalpha|); a cell that fails is quartered (to MAXCELL levels) before the bin is
// given up and halved.  Lz and La come from closed-form sup bounds over the cell's region {Re z in [p1, p2], |Im z| <= q0,
// alpha in [a1, a2]} (crude() below; every inequality is elementary and is spelled out there).  min_bin F is F at the
// bin's faint end (Q is increasing in u: |g| and h both decrease), lowered by the same pads.  rho is free: the best of a
// ladder of values is taken (any rho gives a valid bound).
//
// Result at the reference's configuration (alpha_C in [1, 7], Flim in [1, 6], L up to 46): 16-20 bins of 64 nodes
// instead of 10 201 lattice points per field.
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <map>
#include <utility>
#include <vector>

namespace lfq {

constexpr int KQ = 64;                  // nodes per bin: one per lane of a wave
constexpr double EPS_REL = 1.0e-15;     // see (*)
constexpr double EPS_ABS = 1.0e-30;
constexpr int MTH = 16;                 // boundary samples per half ellipse
constexpr int MAXCELL = 7;              // levels of quartering of an (alpha, c) cell
constexpr int NA0 = 6;                  // alpha cells at level 0 (more for a wide box: at most 1 apiece)
constexpr int MAXHALVE = 8;             // halvings of a bin
constexpr double LN10 = 2.302585092994045684;

struct Box {
    double kappa;                       // sqrt(fc_ratio)
    double a_lo, a_hi;                  // prior box of the completeness slope alpha_C (> 0)
    double lF_lo, lF_hi;                // prior box of lF = log10(1e-17 Flim), in the units of the grid's log flux
};

typedef std::complex<double> cplx;

// ln fc continued off the real axis.  With w = sqrt(1 + v^2) (principal: Re w > 0; analytic where 1 + v^2 avoids the
// negative reals, i.e. unless Re v = 0 and |Im v| >= 1): fc = (w + v) / (2 w).  For Re v >= 0, w and w + v lie in the right
// half plane, for Re v < 0, w and w - v do (and fc = 1 / (2 w (w - v))): in either case the principal log of the quotient
// / product is the difference / sum of the principal logs of its factors, each analytic - so these are THE continuation.
inline cplx g_c(cplx v) {
    const cplx w = std::sqrt(1.0 + v * v);
    if (v.real() >= 0.0) return std::log(1.0 - 0.5 / (w * (w + v)));
    return -std::log(2.0 * w * (w - v));
}
inline cplx h_c(cplx y) {
    const cplx t = std::exp(LN10 * y);
    return 1.0 / (1.0 - std::exp(-t));
}
inline cplx Q_c(cplx z, double al, double kappa) { return g_c(al * z) * h_c(z + kappa / al); }
inline double g_real(double v) {
    const double s = std::sqrt(1.0 + v * v);
    return v >= 0.0 ? std::log1p(-0.5 / (s * (s + v))) : -std::log(2.0 * s * (s - v));
}
inline double Q_real(double u, double al, double kappa) {
    return g_real(al * u) / -std::expm1(-std::pow(10.0, u + kappa / al));
}

// sup bounds of |dQ/dz| and |dQ/dalpha| over  Re z in [p1, p2], |Im z| <= q0, alpha in [a1, a2]  (0 < a1 <= a2).
//   v = alpha z:  Re v in [vp1, vp2], |Im v| <= qv = a2 q0.   P2min / P2max = min / max of (Re v)^2 over that range.
//   |1 + v^2| = |v - i| |v + i|,  |v -+ i|^2 = (Re v)^2 + (Im v -+ 1)^2   =>   D1 <= |1 + v^2| <= D2,
//        D1 = P2min + max(0, 1 - qv)^2,   D2 = P2max + (1 + qv)^2;   |w|^2 = |1 + v^2|.
//   g'(v) = 1 / (w^2 (w + v)) = (w - v) / w^2   =>   |g'| <= (sqrt(D2) + sqrt(P2max + qv^2)) / D1;
//        for Re v >= 0 also |w + v| >= Re w + Re v,  Re w >= sqrt(max(0, Re(1 + v^2))) >= sqrt(max(0, 1 + P2min - qv^2)).
//   |g(v)| <= |g(Re v)| + qv sup|g'|  (vertical segment inside the region)  <= -g(vp1) + qv Gp   (g real: increasing, < 0).
//   y = z + kappa / alpha:  Re y >= ymin = p1 + kappa / a2,  t = 10^y:  |arg t| <= th0 = q0 ln 10 (< pi/2 required),
//        s = Re t >= smin = 10^ymin cos th0,  |e^-t| = e^-s:
//   |h| = 1 / |1 - e^-t| <= 1 / (1 - e^-smin)                                              = Hh
//   h' = -e^-t t ln10 / (1 - e^-t)^2:  |h'| <= (ln10 / cos th0) s e^-s / (1 - e^-s)^2,  and phi(s) = s / (4 sinh^2(s/2)) is
//        decreasing (d ln phi / ds = 1/s - coth(s/2) < 0)                                 =>  Hp = (ln10 / cos th0) phi(smin)
//   dQ/dz = alpha g'(alpha z) h + g h';      dQ/dalpha = z g'(alpha z) h - g h' kappa / alpha^2.
struct Crude {
    bool ok;
    double Lz, La;
};
inline Crude crude(double p1, double p2, double q0, double a1, double a2, double kappa) {
    Crude out{false, 0.0, 0.0};
    const double qv = a2 * q0, th0 = q0 * LN10;
    if (!(a1 > 0.0) || !(th0 <= 1.2)) return out;
    const double vp1 = std::min(a1 * p1, a2 * p1), vp2 = std::max(a1 * p2, a2 * p2);
    const double P2min = (vp1 <= 0.0 && vp2 >= 0.0) ? 0.0 : std::min(vp1 * vp1, vp2 * vp2);
    const double P2max = std::max(vp1 * vp1, vp2 * vp2);
    if (qv > 0.85 && qv > 0.5 * std::sqrt(P2min)) return out;      // too close to the branch points to say anything useful
    const double om = std::max(0.0, 1.0 - qv);
    const double D1 = P2min + om * om, D2 = P2max + (1.0 + qv) * (1.0 + qv);
    if (!(D1 > 0.0)) return out;
    double Gp = (std::sqrt(D2) + std::sqrt(P2max + qv * qv)) / D1;
    if (vp1 >= 0.0) Gp = std::min(Gp, 1.0 / (D1 * (std::sqrt(std::max(0.0, 1.0 + P2min - qv * qv)) + vp1)));
    const double Gg = -g_real(vp1) + qv * Gp;
    const double ymin = p1 + kappa / a2;
    const double c0 = std::cos(th0);
    const double smin = std::pow(10.0, ymin) * c0;
    if (!(smin > 0.0)) return out;
    const double Hh = 1.0 / -std::expm1(-smin);
    const double sh = smin < 700.0 ? std::sinh(0.5 * smin) : HUGE_VAL;
    const double Hp = LN10 / c0 * (smin < 700.0 ? smin / (4.0 * sh * sh) : 0.0);
    const double zmax = std::hypot(std::max(std::fabs(p1), std::fabs(p2)), q0);
    out.Lz = a2 * Gp * Hh + Gg * Hp;
    out.La = zmax * Gp * Hh + Gg * Hp * kappa / (a1 * a1);
    out.ok = std::isfinite(out.Lz) && std::isfinite(out.La);
    return out;
}

inline double logaddexp(double a, double b) {
    const double m = std::max(a, b);
    return m + std::log1p(std::exp(std::min(a, b) - m));
}

// ln(bound of (**)) - ln(EPS_REL min_bin F + EPS_ABS) for every walker of the cell (alpha in al +- da/2, window centre in
// c +- dc/2), windows of half width a, KQ nodes: <= 0 means the cell is proven.  +inf when nothing can be said.
constexpr double RHOS[] = {1.4, 1.6, 2.0, 2.5, 3.2, 4.0, 5.0, 7.0, 10.0, 14.0, 20.0, 28.0, 40.0};
// *floor (optional): the same with the cell's pads left out - what quartering the cell for ever would tend to.
inline double cell_margin(const Box& bx, double a, double al, double da, double c, double dc, int K = KQ, double* floor = nullptr) {
    double best = HUGE_VAL;
    if (floor) *floor = HUGE_VAL;
    const double slack = 1.0e-9;         // rounding of the evaluations below (relative 1e-13 of |Q| <= 1e3 at most)
    for (double rho : RHOS) {
        const double r = 0.5 * a * (rho + 1.0 / rho), rim = 0.5 * a * (rho - 1.0 / rho);
        const Crude cr = crude(c - 0.5 * dc - r, c + 0.5 * dc + r, rim, al - 0.5 * da, al + 0.5 * da, bx.kappa);
        if (!cr.ok) continue;
        double q = -HUGE_VAL;
        for (int m = 0; m < MTH; ++m) {
            const double th = M_PI * (m + 0.5) / MTH;
            const cplx e(std::cos(th), std::sin(th));
            const cplx zeta = 0.5 * a * (rho * e + std::conj(e) / rho);
            q = std::max(q, Q_c(cplx(c, 0.0) + zeta, al, bx.kappa).real());
        }
        if (!std::isfinite(q)) continue;
        const double pad_p = cr.Lz * 0.5 * dc + cr.La * 0.5 * da;       // the cell's other walkers
        const double lnM = q + cr.Lz * r * M_PI / (2.0 * MTH) + pad_p + slack;
        const double lnE = std::log(4.0) + lnM - (K - 1) * std::log(rho) - std::log(rho - 1.0);
        const double lnFmin = Q_real(c - a, al, bx.kappa) - pad_p - slack;
        if (!std::isfinite(lnFmin)) continue;
        best = std::min(best, lnE - logaddexp(std::log(EPS_REL) + lnFmin, std::log(EPS_ABS)));
        if (floor) *floor = std::min(*floor, lnE - pad_p - logaddexp(std::log(EPS_REL) + lnFmin + pad_p, std::log(EPS_ABS)));
    }
    return best;
}

inline bool cell_ok(const Box& bx, double a, double a1, double a2, double c1, double c2, int depth, double* worst, int K = KQ) {
    double fl = 0.0;
    const double m = cell_margin(bx, a, 0.5 * (a1 + a2), a2 - a1, 0.5 * (c1 + c2), c2 - c1, K, &fl);
    if (m <= 0.0) {
        if (worst) *worst = std::max(*worst, m);
        return true;
    }
    if (depth >= MAXCELL || fl > -0.05) return false;       // (no quartering helps a cell that fails without its pads)
    const double am = 0.5 * (a1 + a2), cm = 0.5 * (c1 + c2);
    return cell_ok(bx, a, a1, am, c1, cm, depth + 1, worst, K) && cell_ok(bx, a, a1, am, cm, c2, depth + 1, worst, K) &&
           cell_ok(bx, a, am, a2, c1, cm, depth + 1, worst, K) && cell_ok(bx, a, am, a2, cm, c2, depth + 1, worst, K);
}

// Is the bin [xa, xb] proven for every walker of the box?  *worst (optional): the largest (least negative) margin met.
inline bool bin_ok(const Box& bx, double xa, double xb, double* worst = nullptr, int K = KQ) {
    const double a = 0.5 * (xb - xa), xm = 0.5 * (xa + xb);
    if (!(a > 0.0) || !(bx.a_lo > 0.0) || !(bx.a_hi >= bx.a_lo) || !(bx.lF_hi >= bx.lF_lo)) return false;
    const double clo = xm - bx.lF_hi, chi = xm - bx.lF_lo;
    const int nc = std::max(1, (int)std::ceil((chi - clo) / (2.0 * a)));
    const int na = std::max(NA0, (int)std::ceil(bx.a_hi - bx.a_lo));
    const double dc = (chi - clo) / nc, da = (bx.a_hi - bx.a_lo) / na;
    for (int ia = 0; ia < na; ++ia)
        for (int jc = 0; jc < nc; ++jc)
            if (!cell_ok(bx, a, bx.a_lo + ia * da, bx.a_lo + (ia + 1) * da, clo + jc * dc, clo + (jc + 1) * dc, 0, worst, K)) return false;
    return true;
}

// Bins of [lo, hi]: pieces of width <= width0, each halved until it is proven.  Returns false if a piece still fails
// after MAXHALVE halvings (the caller keeps the full lattice).  *margin: the largest margin of the accepted bins.
inline bool make_bins(const Box& bx, double lo, double hi, double width0, std::vector<double>& edges, double* margin = nullptr, int K = KQ) {
    edges.clear();
    if (!(hi > lo) || !std::isfinite(lo) || !std::isfinite(hi)) return false;
    edges.push_back(lo);
    const int n0 = std::max(1, (int)std::ceil((hi - lo) / width0));
    double worst = -HUGE_VAL;
    struct Item { double a, b; int depth; };
    for (int i = 0; i < n0; ++i) {
        std::vector<Item> stack;
        stack.push_back({lo + (hi - lo) * i / n0, i + 1 == n0 ? hi : lo + (hi - lo) * (i + 1) / n0, 0});
        while (!stack.empty()) {
            const Item it = stack.back();
            stack.pop_back();
            double w = -HUGE_VAL;
            if (bin_ok(bx, it.a, it.b, &w, K)) {
                worst = std::max(worst, w);
                edges.push_back(it.b);
            } else if (it.depth >= MAXHALVE) {
                return false;
            } else {
                const double mid = 0.5 * (it.a + it.b);
                stack.push_back({mid, it.b, it.depth + 1});     // popped second: edges stay ascending
                stack.push_back({it.a, mid, it.depth + 1});
            }
        }
    }
    if (margin) *margin = worst;
    return true;
}

struct GridQ {
    std::vector<double> edges;          // [nb + 1] of the bins that hold lattice points... (all bins; empty ones are dropped below)
    std::vector<double> rec;            // [nb * KQ][4] {x_n, 10^(x_n + 17), L of row row0 + n, 10^(that - 42)} (rows past the bin's last: the last again)
    std::vector<int> rows;              // [nb][4] {row0, nrows, offset into omega (doubles), 0}
    std::vector<double> omega;          // per bin [row][KQ], trapezoid weights in log L folded in
    double margin = 0.0;                // ln of the worst proven ratio bound / allowance (<= 0)
    int nb = 0;
};

// Builds the bins and the rows' weights for a separable grid: L[S], wL[S] (trapezoid weights in log L), ck[S] (trapezoid
// weight in z times dV/dz), Dk[S].  fref / lref: the offsets of the device's flux and luminosity tables (17, 42).
inline bool build_gridq(const Box& bx, int S, const double* L, const double* wL, const double* ck, const double* Dk, double fref, double lref,
                        GridQ& out) {
    if (S < 2) return false;
    double dmin = Dk[0], dmax = Dk[0], lmin = L[0], lmax = L[0];
    for (int k = 0; k < S; ++k) {
        if (!std::isfinite(Dk[k]) || !std::isfinite(ck[k]) || !(ck[k] >= 0.0)) return false;      // (c_k >= 0 is used by (*))
        dmin = std::min(dmin, Dk[k]);
        dmax = std::max(dmax, Dk[k]);
    }
    for (int j = 0; j < S; ++j) {
        if (!std::isfinite(L[j]) || !std::isfinite(wL[j]) || !(wL[j] >= 0.0)) return false;
        lmin = std::min(lmin, L[j]);
        lmax = std::max(lmax, L[j]);
    }
    const double lo = lmin - dmax, hi = lmax - dmin;
    std::vector<double> edges;
    double margin = 0.0;
    if (!make_bins(bx, lo, hi, 1.2, edges, &margin)) return false;
    // a bin's rows must fit a wave (one Schechter value per lane): halve bins that more than KQ rows cross (a proven bin's
    // halves are proven a fortiori only in practice, not in logic - so they are checked as well)
    for (size_t b = 0; b + 1 < edges.size();) {
        int j0 = S, j1 = -1;
        for (int j = 0; j < S; ++j)
            for (int k = 0; k < S; ++k) {
                const double x = L[j] - Dk[k];
                if (x >= edges[b] && x <= edges[b + 1]) {
                    j0 = std::min(j0, j);
                    j1 = std::max(j1, j);
                }
            }
        if (j1 - j0 + 1 > KQ) {
            const double mid = 0.5 * (edges[b] + edges[b + 1]);
            double w = -HUGE_VAL;
            if (edges[b + 1] - edges[b] < 1e-3 || !bin_ok(bx, edges[b], mid, &w) || !bin_ok(bx, mid, edges[b + 1], &w)) return false;
            margin = std::max(margin, w);
            edges.insert(edges.begin() + (std::ptrdiff_t)b + 1, mid);
            continue;
        }
        ++b;
    }
    out = GridQ{};
    out.margin = margin;
    const int nb = (int)edges.size() - 1;
    const long double PI = 3.141592653589793238462643383279502884L;
    std::vector<long double> ctab((size_t)KQ * KQ);          // cos(q theta_n)
    for (int n = 0; n < KQ; ++n)
        for (int q = 0; q < KQ; ++q) ctab[(size_t)n * KQ + q] = cosl(q * PI * (n + 0.5L) / KQ);
    for (int b = 0; b < nb; ++b) {
        const long double mid = 0.5L * ((long double)edges[b] + edges[b + 1]), half = 0.5L * ((long double)edges[b + 1] - edges[b]);
        const bool last = b == nb - 1;
        int j0 = S, j1 = -1;
        std::vector<long double> mom((size_t)S * KQ, 0.0L);
        for (int j = 0; j < S; ++j)
            for (int k = 0; k < S; ++k) {
                const double x = L[j] - Dk[k];                       // the very rounding the lattice tables use
                if (!(x >= edges[b] && (x < edges[b + 1] || (last && x <= edges[b + 1])))) continue;
                j0 = std::min(j0, j);
                j1 = std::max(j1, j);
                const long double t = std::min(1.0L, std::max(-1.0L, ((long double)x - mid) / half));
                long double* M = &mom[(size_t)j * KQ];
                long double t0 = 1.0L, t1 = t;
                M[0] += ck[k];
                M[1] += ck[k] * t;
                for (int q = 2; q < KQ; ++q) {
                    const long double t2 = 2.0L * t * t1 - t0;
                    M[q] += ck[k] * t2;
                    t0 = t1;
                    t1 = t2;
                }
            }
        if (j1 < j0) continue;                                        // no lattice point in this bin
        const int nr = j1 - j0 + 1;
        out.edges.push_back(edges[b]);
        out.edges.push_back(edges[b + 1]);
        out.rows.push_back(j0);
        out.rows.push_back(nr);
        out.rows.push_back((int)out.omega.size());
        out.rows.push_back(0);
        for (int n = 0; n < KQ; ++n) {
            const double xn = (double)(mid + half * cosl(PI * (n + 0.5L) / KQ));
            const int j = std::min(j0 + n, j1);
            out.rec.push_back(xn);
            out.rec.push_back(std::pow(10.0, xn - fref));
            out.rec.push_back(L[j]);
            out.rec.push_back(std::pow(10.0, L[j] - lref));
        }
        for (int j = j0; j <= j1; ++j)
            for (int n = 0; n < KQ; ++n) {
                const long double* M = &mom[(size_t)j * KQ];
                const long double* ct = &ctab[(size_t)n * KQ];
                long double w = M[0];
                for (int q = 1; q < KQ; ++q) w += 2.0L * ct[q] * M[q];
                out.omega.push_back((double)(wL[j] * w / KQ));
            }
        ++out.nb;
    }
    return out.nb > 0;
}

}  // namespace lfq
