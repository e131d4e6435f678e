/* lf_oracle.h - plain-C restatement of the reference's per-step log-posterior (CPU oracle).
 *
 * TEST INFRASTRUCTURE: linked or loaded only by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg; never by the product path (lumfuncmcmc_amd/).
 * Parity status: PINNED - tests/test_oracle_c.py checks it against the vectors recorded from the
 * reference itself (tests/golden/, oracle/gen_golden.py).
 * Same arithmetic as oracle/lf_oracle.py: linear space, then log; -inf on underflow.
 */
#ifndef LF_ORACLE_H
#define LF_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct lfo_inputs {
    int32_t variant;            /* 0 free (lumfuncmcmc.py:395), 1 fixcomp (:411), 2 zevol (lumfuncmcmc_z.py:378) */
    int32_t fix_sch_al, nf, S;
    int64_t N;
    const int64_t *field_ind;   /* [nf+1] */
    const double *lum, *z, *dl_src, *om_arr;   /* [N]; dl_src = DLf(z_i) in Mpc */
    const double *omega0;       /* [nf] */
    const double *logL;         /* [S*S] */
    const double *zarr, *volume_part, *dl_zarr;   /* [S] */
    const double *integ_part;   /* [nf*S*S] or NULL */
    const double *flim0;        /* [nf] */
    double alpha0, sch_al0, fcmin;
    double lims[5][2];          /* Lstar, phistar, sch_al, Flim, alpha */
    double pivots[3];
} lfo_inputs;

/* out[b] = lnprob(theta[b]); outA/outB (optional) the two pieces, NaN where the prior fails.
 * nthreads > 1 spreads the theta rows over OpenMP threads.  Returns 0. */
int lfo_lnprob_batch(const lfo_inputs *in, const double *theta, int B, double *out, double *outA,
                     double *outB, int nthreads);
int lfo_ndim(const lfo_inputs *in);

#ifdef __cplusplus
}
#endif
#endif
