/* lf_oracle.c - see lf_oracle.h.  Each function cites the reference lines it restates. */
#include "lf_oracle.h"

#include <math.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LN10 2.302585092994045684
static const double MPC_CM = 3.086e24; /* lumfuncmcmc.py:70 */

static double sqarcsec(void) { /* VmaxLumFunc.py:43 */
    const double a = 180. / M_PI * 3600.0;
    return a * a;
}

/* lumfuncmcmc.py:44 */
static double true_lum_func(double logL, double alpha, double logLstar, double logphistar) {
    return LN10 * pow(10.0, logphistar) * pow(10.0, (logL - logLstar) * (alpha + 1)) * exp(-pow(10.0, logL - logLstar));
}

/* VmaxLumFunc.py:118-127, :141, :164-167 (fcmin truthy) */
static double fleming(double f, double Flim, double alpha, double fcmin) {
    const double num = alpha * log10(f / Flim);
    const double fc = 0.5 * (1. + num / sqrt(1. + num * num));
    const double a = (2 * fcmin - 1) * (2 * fcmin - 1);
    const double b = -1 * sqrt(fabs(a / (1 - a)) * pow(alpha, -2.));
    const double f_tau = Flim * pow(10.0, b);
    return pow(fc, 1. / (1. - exp(-f / f_tau)));
}

/* lumfuncmcmc.py:69-70 with DL = DLf(z) */
static double omega(double logL, double DL, double Omega_0, double Flim, double alpha, double fcmin) {
    const double L = pow(10.0, logL);
    const double d = MPC_CM * DL;
    return Omega_0 / sqarcsec() * fleming(L / (4.0 * M_PI * (d * d)), Flim, alpha, fcmin);
}

/* lumfuncmcmc_z.py:40-42 */
static void quad_coef(double y1, double y2, double y3, double z1, double z2, double z3, double *a, double *b, double *c) {
    *a = ((y3 - y1) + (y2 - y1) * (z1 - z3) / (z2 - z1)) / (z3 * z3 - z1 * z1 + (z2 * z2 - z1 * z1) * (z1 - z3) / (z2 - z1));
    *b = (y2 - y1 - *a * (z2 * z2 - z1 * z1)) / (z2 - z1);
    *c = y1 - *a * z1 * z1 - *b * z1;
}

int lfo_ndim(const lfo_inputs *in) {
    if (in->variant == 0) return 2 + (in->fix_sch_al ? 0 : 1) + in->nf + 1;
    if (in->variant == 1) return 2 + (in->fix_sch_al ? 0 : 1);
    return 6 + (in->fix_sch_al ? 0 : 1);
}

static int inside(double v, const double lim[2]) { return (v >= lim[0]) && (v <= lim[1]); }
static int inside_strict(double v, const double lim[2]) { return (v > lim[0]) && (v < lim[1]); }

/* trapz(trapz(I, logL, axis=0), zarr): numpy's d * (y1 + y0) / 2 form (lumfuncmcmc.py:377) */
static double trapz2(const double *I, const double *logL, const double *zarr, int S, double *col) {
    for (int k = 0; k < S; ++k) {
        double s = 0.0;
        for (int j = 0; j + 1 < S; ++j)
            s += (logL[(size_t)(j + 1) * S + k] - logL[(size_t)j * S + k]) * (I[(size_t)(j + 1) * S + k] + I[(size_t)j * S + k]) / 2.0;
        col[k] = s;
    }
    double r = 0.0;
    for (int k = 0; k + 1 < S; ++k) r += (zarr[k + 1] - zarr[k]) * (col[k + 1] + col[k]) / 2.0;
    return r;
}

static double one(const lfo_inputs *in, const double *th, double *pA, double *pB, double *work) {
    const int nf = in->nf, S = in->S;
    const int64_t N = in->N;
    const size_t nn = (size_t)S * S;
    double *I = work, *col = work + nn;
    double A = 0.0, Bint = 0.0;
    *pA = NAN;
    *pB = NAN;
    if (in->variant == 2) { /* lumfuncmcmc_z.py:332-376 */
        const double *L = th, *P = th + 3;
        const double al = in->fix_sch_al ? in->sch_al0 : th[6];
        int ok = in->fix_sch_al ? 1 : inside(al, in->lims[2]);
        for (int i = 0; i < 3; ++i) ok = ok && inside_strict(L[i], in->lims[0]) && inside_strict(P[i], in->lims[1]);
        if (!ok) return -INFINITY;
        double aL, bL, cL, aP, bP, cP;
        quad_coef(L[0], L[1], L[2], in->pivots[0], in->pivots[1], in->pivots[2], &aL, &bL, &cL);
        quad_coef(P[0], P[1], P[2], in->pivots[0], in->pivots[1], in->pivots[2], &aP, &bP, &cP);
        for (int64_t i = 0; i < N; ++i) {
            const double z = in->z[i];
            const double ph = aP * (z * z) + bP * z + cP, Ls = aL * (z * z) + bL * z + cL;
            A += log(true_lum_func(in->lum[i], al, Ls, ph) * in->om_arr[i]);
        }
        for (int f = 0; f < nf; ++f) {
            for (int j = 0; j < S; ++j)
                for (int k = 0; k < S; ++k) {
                    const double z = in->zarr[k];
                    const double ph = aP * (z * z) + bP * z + cP, Ls = aL * (z * z) + bL * z + cL;
                    I[(size_t)j * S + k] = true_lum_func(in->logL[(size_t)j * S + k], al, Ls, ph) * in->integ_part[(size_t)f * nn + (size_t)j * S + k];
                }
            Bint += trapz2(I, in->logL, in->zarr, S, col);
        }
    } else { /* lumfuncmcmc.py:320-393 */
        const double Lstar = th[0], phistar = th[1];
        int k0 = 2;
        const double al = in->fix_sch_al ? in->sch_al0 : th[k0++];
        double Flim[16], alphaC = in->alpha0;
        for (int f = 0; f < nf; ++f) Flim[f] = in->flim0 ? in->flim0[f] : 0.0;
        if (in->variant == 0) {
            for (int f = 0; f < nf; ++f) Flim[f] = th[k0 + f];
            alphaC = th[k0 + nf];
        }
        int ok = inside(Lstar, in->lims[0]) && inside(phistar, in->lims[1]) && inside(al, in->lims[2]) && inside(alphaC, in->lims[4]);
        for (int f = 0; f < nf; ++f) ok = ok && inside(Flim[f], in->lims[3]);
        if (!ok) return -INFINITY;
        for (int f = 0; f < nf; ++f) {
            const double om0_int = (double)(long long)in->omega0[f]; /* dtype=int truncation, :285 */
            for (int64_t i = in->field_ind[f]; i < in->field_ind[f + 1]; ++i) {
                const double T = true_lum_func(in->lum[i], al, Lstar, phistar);
                const double Om = in->variant == 0 ? omega(in->lum[i], in->dl_src[i], om0_int, 1.0e-17 * Flim[f], alphaC, in->fcmin)
                                                   : in->om_arr[i];
                A += log(T * Om);
            }
        }
        for (int f = 0; f < nf; ++f) {
            for (int j = 0; j < S; ++j)
                for (int k = 0; k < S; ++k) {
                    const size_t g = (size_t)j * S + k;
                    const double T = true_lum_func(in->logL[g], al, Lstar, phistar);
                    const double part = in->variant == 0
                        ? in->volume_part[k] * omega(in->logL[g], in->dl_zarr[k], in->omega0[f], 1.0e-17 * Flim[f], alphaC, in->fcmin)
                        : in->integ_part[(size_t)f * nn + g];
                    I[g] = T * part;
                }
            Bint += trapz2(I, in->logL, in->zarr, S, col);
        }
    }
    *pA = A;
    *pB = Bint;
    return A - Bint;
}

int lfo_lnprob_batch(const lfo_inputs *in, const double *theta, int B, double *out, double *outA, double *outB, int nthreads) {
    const int nd = lfo_ndim(in);
    const size_t wsz = (size_t)in->S * in->S + in->S;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
    {
        double *work = (double *)malloc(wsz * sizeof(double));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int b = 0; b < B; ++b) {
            double a, bb;
            const double r = one(in, theta + (size_t)b * nd, &a, &bb, work);
            out[b] = r;
            if (outA) outA[b] = a;
            if (outB) outB[b] = bb;
        }
        free(work);
    }
    (void)nthreads;
    return 0;
}
