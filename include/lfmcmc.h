/*
 * lfmcmc.h - C ABI of liblfmcmc.so: the per-step log-posterior of LumFuncMCMC on MI355X (gfx950).
 *
 * The reference has no FFI: its hot path is a bound Python method handed to emcee,
 *     emcee.EnsembleSampler(nwalkers, ndim, self.lnprob)          lumfuncmcmc.py:487-489
 *     emcee.EnsembleSampler(nwalkers, ndim, self.lnprob)          lumfuncmcmc_z.py:444
 * with signature  f(theta: float64[ndim]) -> float  (-inf outside the prior or on underflow,
 * never NaN).  This header is the boundary a maintainer would bind instead (ctypes stub in
 * INTEGRATION.md): plain pointers and sizes, no torch / numpy types.
 *
 * Each entry point names the reference interface it replaces (file:line in the reference).
 *
 * Threading: one in-flight call per context; lf_create / lf_destroy are not thread-safe.
 * Ownership: the caller owns every host buffer it passes (they are copied during lf_create);
 * the library owns all device memory.  No entry point throws; errors are negative return codes
 * plus a message from lf_last_error().
 */
#ifndef LFMCMC_H
#define LFMCMC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LF_ABI_VERSION 2
#define LF_MAX_FIELDS 8

/* model variants = the three log-posterior callables of the reference */
enum {
    LF_FREE = 0,    /* LumFuncMCMC.lnprob          lumfuncmcmc.py:395-409  (completeness free)  */
    LF_FIXCOMP = 1, /* LumFuncMCMC.lnprob_fix_comp lumfuncmcmc.py:411-424  (completeness fixed) */
    LF_ZEVOL = 2    /* LumFuncMCMCz.lnprob         lumfuncmcmc_z.py:378-392 (z-evolving L*, phi*) */
};

/* return codes */
enum {
    LF_OK = 0,
    LF_ERR_ARG = -1,    /* bad descriptor / NULL pointer / B <= 0                       */
    LF_ERR_HIP = -2,    /* a HIP runtime call failed (message has the hipError string)   */
    LF_ERR_NOMEM = -3,  /* host or device allocation failed                              */
    LF_ERR_NODEV = -4   /* no gfx950 device visible                                      */
};

/* indices into lf_desc.lims: the *_lims constructor arguments, lumfuncmcmc.py:73-79 */
enum { LF_LIM_LSTAR = 0, LF_LIM_PHISTAR = 1, LF_LIM_SCH_AL = 2, LF_LIM_FLIM = 3, LF_LIM_ALPHA = 4 };

/*
 * Everything lnlike reads, i.e. the outputs of the reference's one-off setup
 * (setDLdVdz / setOmegaLz / setlnsimple / defineFlimOmArr, lumfuncmcmc.py:180-235, :283-288;
 * lumfuncmcmc_z.py:226-305).  Arrays are float64, C-contiguous.
 */
typedef struct lf_desc {
    int32_t variant;       /* LF_FREE | LF_FIXCOMP | LF_ZEVOL                                   */
    int32_t fix_sch_al;    /* 1: Schechter alpha is not in theta (lumfuncmcmc.py:330,333)       */
    int32_t nf;            /* number of fields, 1..LF_MAX_FIELDS (len(Flim), :147)              */
    int32_t S;             /* size_ln: side of the integration grid (101 free, 201 else; :219)  */
    int64_t N;             /* number of sources                                                 */
    const int64_t *field_ind; /* [nf+1] field f = sources field_ind[f] .. field_ind[f+1]-1 (:148) */
    const double *lum;     /* [N] log10 L (erg/s)                                    self.lum   */
    const double *logf;    /* [N] FREE: lum - log10(4 pi (3.086e24 DLf(z))^2)   (the flux of :70) */
    const double *z;       /* [N] ZEVOL: source redshifts                              self.z   */
    const double *om_arr;  /* [N] FIXCOMP, ZEVOL: Omega(lum_i, z_i) at fixed completeness (:235)  */
    const double *omega0;  /* [nf] effective areas in sq arcsec, as configured (float).  The
                              library applies the reference's int truncation for the per-source
                              term (:285) and uses the float value in the integral (:375).       */
    const double *logL;    /* [S*S] row-major [j][k]: j = luminosity index, k = redshift index;
                              the ONE grid every field integrates on (self.logL[-1], :225-232)    */
    const double *zarr;    /* [S]                                                    self.zarr  */
    const double *volume_part; /* [S] FREE: dVc/dz/dOmega at zarr                   (:223)      */
    const double *dl_zarr; /* [S] FREE: DLf(zarr) in Mpc                             (:222)      */
    const double *integ_part;  /* [nf*S*S] FIXCOMP, ZEVOL: volume_part * Omegaf[f].ev (:233-234) */
    const double *flim0;   /* [nf] FIXCOMP: the fixed Flim (prior still tests them, :346-351)   */
    double alpha0;         /* FIXCOMP: the fixed completeness slope                             */
    double sch_al0;        /* value of alpha when fix_sch_al                                    */
    double fcmin;          /* modified-Fleming knee, VmaxLumFunc.py:95 (0 < fcmin < 1)          */
    double lims[5][2];     /* prior boxes, rows LF_LIM_*                                        */
    double pivots[3];      /* ZEVOL: z1, z2, z3                      lumfuncmcmc_z.py:125       */
    int32_t device;        /* HIP device ordinal                                                */
    int32_t max_batch;     /* rows the workspace is sized for up front (grown on demand); 0 = 1024 */
} lf_desc;

typedef struct lf_ctx lf_ctx;

/* ABI version of the loaded library (== LF_ABI_VERSION of the header it was built from). */
/* Host-only helper behind lf_free's deal of its chunks to its 32 virtual workgroups, exported so that
 * the CPU tests can check it: table[0 .. 32] = where rank r's cell chunks start in the list, table[33 .. 65] the same for its
 * flux bins, then the list (cell chunks rank by rank, then bins rank by rank).  table = NULL: returns the number of ints
 * (66 + n_cell_chunks + n_bins); else fills `table` (cap ints) and returns that number.  grid_part / grid_parts as in the
 * "grid_share" option (0, 0: the whole grid). */
int lf_deal_table(int n_cell_chunks, int n_bins, int grid_part, int grid_parts, int32_t *table, int64_t cap);

int lf_abi_version(void);

/* Replaces the read side of LumFuncMCMC.__init__ / LumFuncMCMCz.__init__ (lumfuncmcmc.py:162-177):
 * copies the catalogue and grids to HBM, derives the parameter-independent per-source and
 * per-node tables.  Returns NULL on failure; lf_last_error(NULL) then holds the reason.
 * What is derived here, beyond the reference's arrays: the catalogue's cells (option "cells" of lf_set_option); for
 * fixed completeness the integration grid summed over its rows when every redshift column has the same luminosity
 * nodes; for the z-evolving model the grid stored column by column.  Two environment variables, read here, switch
 * the last two off for A/B tests - LF_NO_COLLAPSE_GRID, LF_NO_ZGRID_COLS (results agree to rounding either way);
 * LF_DEBUG_OCC prints the occupancy of a kernel instantiation at its first launch. */
lf_ctx *lf_create(const lf_desc *desc);

/* Releases device and host memory of the context. */
void lf_destroy(lf_ctx *ctx);

/* Number of free parameters of the variant (pos.shape[1] at lumfuncmcmc.py:485). */
int lf_ndim(const lf_ctx *ctx);

/* Replaces B serial calls of lnprob / lnprob_fix_comp (lumfuncmcmc.py:395, :411;
 * lumfuncmcmc_z.py:378).  theta: host, [B][ndim] row-major.  out: host, [B].
 * out[i] = -INFINITY where the prior fails or the likelihood underflows; never NaN.
 * Synchronous.  Returns LF_OK or a negative code. */
int lf_lnprob_batch(lf_ctx *ctx, const double *theta, int B, double *out);

/* Same, with theta and out in device memory of ctx's device and the launches enqueued on
 * `hip_stream` (a hipStream_t, NULL = the default stream).  Asynchronous: `out` is complete when
 * the stream reaches this point.  This is the form the multi-GPU path uses (the per-rank slice of
 * lnprob feeds the RCCL all-gather without a host round trip). */
int lf_lnprob_batch_device(lf_ctx *ctx, const double *d_theta, int B, double *d_out, void *hip_stream);

/* K evaluations in one call: block k is theta rows d_theta[k B ndim ..] -> d_out[k B ..], k = 0 .. K-1, enqueued back to back
 * on `hip_stream` (what K calls of lf_lnprob_batch_device enqueue, without K trips through the caller's language: a Python
 * caller spends 7 us per call, this loop 2-3 us per evaluation; the evaluations themselves take 8-21 us each).  For callers
 * with many independent half-ensembles to evaluate (several chains, a tempering ladder); emcee's stretch move itself
 * (lumfuncmcmc.py:489-491) needs block k's result before it can propose block k + 1 - that loop is lf_sampler_run's. */
int lf_lnprob_batch_device_n(lf_ctx *ctx, const double *d_theta, int B, int K, double *d_out, void *hip_stream);

/* Diagnostics for parity tests: the two pieces of lnlike separately (host buffers).
 * outA[i] = per-source log-term sum (lumfuncmcmc.py:370 / :388 / lumfuncmcmc_z.py:371),
 * outB[i] = expected-count integral (:373-377 / :389-392 / _z:373-375).  Rows failing the prior
 * get NaN in both.  Synchronous. */
int lf_lnprob_pieces(lf_ctx *ctx, const double *theta, int B, double *outA, double *outB);

/* Kernel timing for bench.py: level 1 brackets lf_main, level 2 every launch, with hipEvents on
 * the stream the launch runs on (0 = off; each event pair costs a few microseconds of stream time: the events are
 * barrier packets, and the next launch no longer overlaps the previous one's tail - measured 7.6 us per evaluation
 * of 40 at level 1.  Option "profile_every" = n brackets only every n-th evaluation).  lf_kernel_times reads and clears the accumulated totals:
 * ms[0..3] = {lf_prepare, lf_main (per-source sum + grid integral, one launch), unused (0), lf_finalize},
 * launches[0..3] the launch counts.  Synchronises the recorded events. */
int lf_set_profiling(lf_ctx *ctx, int level);
int lf_kernel_times(lf_ctx *ctx, double ms[4], int64_t launches[4]);

/* Tuning knobs (performance only, never results beyond summation order):
 * key "geometry": index of the launch geometry (sources per lane x walkers per workgroup),
 * -1 = chosen from N and B; "walker_tile": walkers per workgroup, 0 = the geometry's;
 * "taper": 1 gives the last ~B/8 walkers quarter-size tiles, dispatched last, so that the launch drains evenly
 * (bitwise neutral: a tile only decides which workgroup owns a (chunk, walker) sum); +1 % at N = 1e6 but the tail
 * tiles read the catalogue a second time (L2 -> fabric traffic 37 MB instead of 21 MB per launch): off by default.
 * "compress": 1 = take piece A from the COMPRESSED CATALOGUE (FREE, ZEVOL; off by default; never the headline path): the
 * ~N sources of a field are replaced by K = 16 weighted pseudo-sources per bin of the one coordinate the walker-dependent
 * factor of a term depends on (log flux; redshift).  Bins are halved until an ESTIMATE of the relative error of every
 * bin sum is below 1e-16; the estimate is evaluated at sampled walkers of the prior box (FREE: 7 slopes alpha_C x up to
 * 96 shifts of the flux limit, 33 points per bin; ZEVOL: the 8 corners of the (L1, L2, L3) box) - a sampled validation,
 * NOT a bound proven over the whole box (csrc/lf_compress.h).  In addition a freshly built compressed catalogue must
 * reproduce the direct path to 1e-12 on 64 walkers drawn from this context's prior box, or the option is refused
 * (LF_ERR_ARG).  lnprob then costs the grid integral plus a few hundred terms, whatever N is.  Walkers that need the
 * per-source underflow checks are still summed over the real catalogue.  Built at the first call with value 1.
 * With it, FREE contexts whose integration grid is separable (every redshift column has the same luminosity nodes:
 * min_comp_frac = 0) also take piece B over ~40 x 16 shared flux nodes per field instead of the S^2 lattice points
 * (same sampled validation; "compress_grid" = 0 keeps the full grid).
 * "tables": 1 (default) lets the free variant evaluate the term as the product of two tabulated univariate factors,
 * g(num) = ln fc and h(y) = 1 / (1 - e^(-10^y)) (piecewise degree-7 polynomials, relative error <= 8e-15 over their
 * whole domain, lf_tables.h), for (walker, chunk) pairs whose fluxes lie inside the tables and whose lanes of
 * flux-neighbours are narrower than the tables' margins; 0 = always the general form (A/B runs).
 * "persistent": 1 (default) runs the direct path in the persistent kernels - 512-thread workgroups, two per CU, that
 * hold the tables in LDS and serve one tile of 8 walkers per group of workgroups: lf_free (free completeness: whenever
 * the context has cells - every catalogue up to 65 536 sources and every larger one with at least four sources per
 * cell - or else when catalogue and grid give every workgroup about four chunks) and lf_pers (fixed completeness
 * always; z-evolving when the context has cells in redshift, one set of L nodes per redshift column and at most 256
 * columns); 0 = always lf_main; 2 = always the persistent kernel (tests).  "free_st": sources per lane of lf_free, 0 (auto: 8
 * for N >= 3e5, else 4), 2, 4 or 8 (tuning runs).  "cells": 1 (default) lets lf_free sum a walker whose every field lies inside the tables over the
 * catalogue's CELLS instead of its sources: runs of flux-neighbouring sources no wider than 2 rho, kept as their
 * midpoint and power sums S_0 .. S_8; on a table piece the term is a polynomial in the flux offset, so a cell's sum is
 * a dot product of its Taylor coefficients with the power sums, exact up to the orders above 8, whose share is below
 * 2e-18 by the choice of rho (from the prior box's largest alpha_C).  The z-evolving variant has cells too, in
 * redshift: what is left of its term per source is a weight times exp of a parabola in z, summed as a series in the
 * weighted power sums of a run of redshift neighbours (orders above 6 below 1e-17; the cells' width is chosen from the
 * prior box of L1..L3 so that every walker inside it qualifies).  0 = every walker over the sources (A/B runs).
 * "fuse": 1 (default) lets lf_free / lf_pers do lf_prepare's and lf_finalize's work themselves for plain evaluations -
 * one launch instead of three, same bits; 0 = three launches (A/B runs; lf_lnprob_pieces, the census, profiling level 2
 * and the walker-sharded sampler's halves always take three).  "fuse_step": 1 (default) makes a half-step of the
 * device-resident sampler ONE launch of the same kernels (proposal in the prologue, accept / reject and the chain's row
 * by the tile's finishing workgroup); 0 = lf_propose+prepare, kernel, lf_finalize+accept (same chain, bit for bit).
 * "poll": 1 (default) lets a tile of the one-launch form whose walkers need no per-source sums (the normal case) hand its
 * partial sums over by POLLING: every slot of the partial-sum buffers is kept at a reserved NaN pattern between launches,
 * a workgroup writes its sums through and is done, and the tile's finishing workgroup reads the slots until none is
 * empty (at most 2^19 times: then it writes NaN and lf_lnprob_batch returns LF_ERR_HIP); 0 = every tile counts its
 * workgroups (store, wait for the acknowledgement, atomic counter; A/B runs).  Same sums in the same order: same bits.
 * "profile_every": see lf_set_profiling.  "profile_span": n > 1 puts ONE event pair around n consecutive one-launch
 * evaluations (profiling level 1) instead of a pair around a single launch, and lf_kernel_times counts n launches for it:
 * a pair of barrier packets around one 13-us launch adds its dispatch to the figure (16.6 us where the profiler says 13.7).
 * "grid_shortcut": 1 (default) lets lf_free take piece B of a FREE context whose integration grid is separable (every
 * redshift column has the same luminosity nodes: min_comp_frac = 0) over FLUX BINS instead of the S^2 lattice points: the
 * completeness depends on a lattice point only through its log flux L_j - D_k, so per bin the lattice points of a row are
 * replaced by the bin's 64 Chebyshev nodes with weights from the row's moments (exact for polynomials of degree < 64),
 * and the Schechter factor of the rows enters by a dot product.  The bins are made at lf_create such that the
 * interpolation error of the completeness curve is PROVEN (Bernstein-ellipse bound, evaluated for the whole prior box
 * of alpha_C and Flim - csrc/lf_gridbound.h, tests/test_gridbound_cpu.py) to be below 1e-15 of the curve's smallest
 * value in the bin + 1e-30: |piece B (bins) - piece B (lattice)| <= 1e-15 piece B + 1e-30 x (piece B at completeness 1).
 * 16-20 bins x 64 nodes per field instead of 10 201 lattice points.  No proven set of bins (a prior box that reaches
 * alpha_C <= 0, a grid that is not separable) = the lattice; 0 = the lattice (A/B runs).
 * "specialise": 1 (default) lets the free variant take the cheaper form of the term for (walker, chunk) pairs whose
 * every source has f / f_tau > 37.5 (decay factor exactly 1.0 in binary64); 0 = always the general form (A/B runs).
 * Two keys change what is computed, for SOURCE-SHARDED ranks whose lnprob values are summed (all-reduce): "skip_grid" = 1
 * leaves the expected-count integral (piece B) out of lnprob altogether; "grid_share" = part + 65536 * parts makes this
 * context integrate only the node chunks c with c % parts == part, so that the ranks split piece B as well as piece A. */
int lf_set_option(lf_ctx *ctx, const char *key, int64_t value);

/* Census of which form of the per-source term / grid node ran since lf_set_option(ctx, "count_forms", 1) (which also
 * clears it; 0 switches it off again - it costs one atomic per (walker, chunk), so it is off by default).  For
 * bench.py's executed-flop accounting and the tests.  counts[0..8] = (walker, source) terms evaluated in the general
 * form, the general form without the exponential, the table-driven form, the table-driven form without the
 * exponential, the careful (checked) form, terms of walkers that were not evaluated (outside the prior / already
 * -inf); then (walker, node, field) terms of the grid integral in the general and in the bright form; then
 * (walker, cell) evaluations (option "cells": a cell stands for all the sources of a narrow flux interval).  FREE variant,
 * real catalogue.  The z-evolving variant counts its own: [8] (walker, cell in redshift) evaluations, [2] terms of its
 * local form (one exponential per lane of redshift neighbours), [0] per-source exponentials, [4] terms of the careful
 * form, [6] (walker, node) terms of its grid.  (The fixed-completeness variant and the compressed catalogue leave it at 0.)  Synchronises the device. */
int lf_form_counts(lf_ctx *ctx, int64_t counts[9]);

/* Shape of the most recent lf_main launch of this context (measurement only): info[0..7] = sources per lane, walkers
 * per source workgroup and per grid workgroup of the instantiation (template parameters ST, TW, TWB); which kernel ran
 * (0 lf_main, 1 its compressed-catalogue instantiation, 2 lf_free: the persistent kernel of the free variant, 3
 * lf_free in its one-launch form, see "fuse"; 4 lf_pers: the persistent kernel of the other two variants, 5 lf_pers in
 * its one-launch form);
 * workgroups in the launch, catalogue chunks, grid chunks, theta rows. */
int lf_last_launch(const lf_ctx *ctx, int32_t info[8]);

/*
 * Device-resident ensemble sampler: the Goodman & Weare stretch move in its parallel form (two fixed
 * half-ensembles, as emcee 2.x - the API the reference calls - implements it), with theta, lnprob
 * and the chain kept in HBM.  Replaces
 *     sampler = emcee.EnsembleSampler(nwalkers, ndim, lnprob); sampler.run_mcmc(pos, nsteps)
 *     sampler.chain, sampler.lnprobability, sampler.acceptance_fraction      lumfuncmcmc.py:489-513
 * One step = 2 half-steps of ONE launch each where the persistent kernels serve the context (option "fuse_step"),
 * else 2 x (propose+prepare, main, finalize+accept) = 6 launches; no host round trip either way.
 * Random numbers are Philox4x32-10 keyed by `seed`: the chain is a pure function of (seed, start).
 */
typedef struct lf_sampler lf_sampler;

/* `a` is the stretch scale (emcee default 2.0); capacity_steps sizes the on-device chain. */
lf_sampler *lf_sampler_create(lf_ctx *ctx, int nwalkers, double a, uint64_t seed, int64_t capacity_steps);
void lf_sampler_destroy(lf_sampler *s);
/* pos: host [nwalkers][ndim].  lnprob0: host [nwalkers] or NULL (then evaluated here).  Resets the chain. */
int lf_sampler_start(lf_sampler *s, const double *pos, const double *lnprob0);
/* Enqueue nsteps full ensemble steps on hip_stream (NULL = the context's stream).  Asynchronous. */
int lf_sampler_run(lf_sampler *s, int64_t nsteps, void *hip_stream);
/* Synchronise and copy out; any pointer may be NULL.  chain: [nwalkers][steps][ndim] (emcee's
 * sampler.chain layout), chain_lnprob: [nwalkers][steps], naccepted: [nwalkers], pos / lnprob: current state. */
int lf_sampler_read(lf_sampler *s, double *chain, double *chain_lnprob, int64_t *naccepted, double *pos, double *lnprob);
/* Steps recorded so far. */
int64_t lf_sampler_steps(const lf_sampler *s);
/* Walker-sharded form of one half-step (multi-GPU: one process per GPU, every rank holds the whole
 * ensemble).  half_eval proposes for the whole half (half = 0 or 1) and evaluates lnprob of the
 * proposals lo..hi-1 (indices within the half) into d_newlp[lo..hi-1] (device); the caller all-gathers
 * d_newlp over the ranks (RCCL); half_accept then accepts/rejects the whole half and records the chain
 * (after half 1 the step counter advances).  Both enqueue on hip_stream as given (NULL = the default
 * stream, as in lf_lnprob_batch_device), so that the collective in between is stream-ordered with them.
 * Same random numbers and arithmetic as lf_sampler_run: the chain is identical for any sharding. */
int lf_sampler_half_eval(lf_sampler *s, int half, int lo, int hi, double *d_newlp, void *hip_stream);
int lf_sampler_half_accept(lf_sampler *s, int half, const double *d_newlp, void *hip_stream);

/* Host-only helper behind "compress", exported for tests (touches no GPU): compress n coordinates `key` with
 * weights `wt` (NULL = 1) into pseudo-sources.  kind 0 (FREE): params = {|a/(1-a)|, alpha_lo, alpha_hi, flim_lo,
 * flim_hi}; kind 1 (ZEVOL): params = {L_lo, L_hi, z1, z2, z3}.  Returns the number of pseudo-sources (written to
 * node / weight when it is <= cap), or a negative code; *bound = the largest accepted (sampled) error estimate. */
int64_t lf_compress_keys(int kind, const double *params, const double *key, const double *wt, int64_t n,
                         double *node, double *weight, int64_t cap, double *bound);

/* Host-only helper behind the compressed grid, exported for tests: S luminosity nodes L with trapezoid weights wL,
 * S redshift weights ck (trapezoid x dV/dz) and D_k = log10(4 pi DL_k^2); params as for kind 0 of lf_compress_keys.
 * Outputs per bin: 16 node positions u, first row row0, row count nrows, offset off into omega ([row][node], wL
 * folded in).  Returns the number of bins (outputs written when the capacities suffice), or a negative code. */
int64_t lf_compress_grid(const double *params, int S, const double *L, const double *wL, const double *ck,
                         const double *Dk, double *u, int32_t *row0, int32_t *nrows, int32_t *off, double *omega,
                         int64_t cap_bins, int64_t cap_omega, double *bound);

/* Host-only helper behind "grid_shortcut", exported for tests (touches no GPU).  Arguments as lf_compress_grid.  Outputs
 * per bin: edges[2], rec[64][4] = {node x_n, 10^(x_n + 17), L of row row0 + n (the last row again past the bin's rows),
 * 10^(that - 42)}, rows[4] = {row0, nrows, offset into omega, 0}, omega [row][64].  *margin = ln(proven error bound /
 * allowance) of the worst bin, <= 0.  Returns the number of bins (outputs written when the capacities suffice),
 * or a negative code when no proven set of bins exists. */
int64_t lf_grid_bins(const double *params, int S, const double *L, const double *wL, const double *ck, const double *Dk,
                     double *edges, double *rec, int32_t *rows, double *omega, int64_t cap_bins, int64_t cap_omega,
                     double *margin);

/* 1/Veff estimator on the device (post-fit diagnostic, no context needed).  Replaces the per-source loop of
 * LumFuncMCMC.VeffLF (lumfuncmcmc.py:515-525: V.lumfunc, one scipy.quad per source, VmaxLumFunc.py:235-257) and the
 * nboot x nbins masked sums of V.getBootErrLog (VmaxLumFunc.py:304-378).  All arrays host, length n unless noted.
 *   phi[i] = 1 / (pref0 * fleming(flux[i], flim[i], alpha, fcmin) * (vol ? vol[i] : vol_all)),  0 where the volume is <= 0
 *   (pref0 = sum(Omega_0) / sqarcsec; fcmin <= 0 = the unmodified Fleming curve);
 *   sums[(nboot + 1) * nbin]: row 0 = sum of phi per luminosity bin (bin_of[i] in [0, nbin), anything else = no bin),
 *   rows 1..nboot = the same over bootstrap resamples of the catalogue: indices from boot_idx[nboot * n] when given
 *   (a caller replaying a seeded host stream; every index must lie in [0, n): LF_ERR_ARG otherwise), else drawn on the
 *   device (Philox4x32-10 keyed by seed).
 * nbin = 0 skips the binning (bin_of, sums may be NULL).  nbin <= 1024.  Synchronous. */
int lf_veff(int device, int64_t n, const double *flux, const double *flim, const double *vol, double vol_all, double pref0,
            double alpha, double fcmin, const int32_t *bin_of, int32_t nbin, int32_t nboot, const int64_t *boot_idx, uint64_t seed,
            double *phi, double *sums);

/* Last error message of this context (or of lf_create when ctx == NULL).  Never NULL. */
const char *lf_last_error(const lf_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* LFMCMC_H */
