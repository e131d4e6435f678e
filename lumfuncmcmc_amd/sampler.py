"""Affine-invariant ensemble sampler (Goodman & Weare stretch move) on the batched boundary.

The reference hands a scalar callable to emcee (lumfuncmcmc.py:489-491) and reads back
`sampler.chain`, `.lnprobability`, `.acor`, `.acceptance_fraction` (:499-513).  emcee is not
installed here, and its per-walker Python call is exactly the overhead the batched C ABI removes,
so this is an own implementation with that read-back surface.  One step = two half-ensemble
updates; each half is ONE call of `log_prob_fn` with a (W/2, ndim) block - the shape
lf_lnprob_batch is built for.

With several ranks (torch.distributed initialised), every rank runs the same sampler with the
same random stream; `log_prob_fn` (lumfuncmcmc_amd.dist.ShardedLnProb) evaluates only its slice
of the block and all-gathers, so the ensembles stay bitwise identical across ranks.
"""
import numpy as np


def integrated_time(x, c=5.0):
    """Integrated autocorrelation time of a (nsteps, nwalkers) series, Sokal windowing; finite for
    any input (short chains return a positive estimate instead of raising)."""
    x = np.atleast_2d(np.asarray(x, dtype=np.float64))
    n = x.shape[0]
    if n < 4:
        return 1.0
    acf = np.zeros(n)
    size = 1 << int(np.ceil(np.log2(2 * n)))
    for k in range(x.shape[1]):
        y = x[:, k] - x[:, k].mean()
        f = np.fft.rfft(y, n=size)
        a = np.fft.irfft(f * np.conjugate(f))[:n]
        if a[0] > 0:
            acf += a / a[0]
    acf /= max(x.shape[1], 1)
    taus = 2.0 * np.cumsum(acf) - 1.0
    m = np.arange(n) < c * taus
    win = int(np.argmin(m)) if not m.all() else n - 1
    tau = float(taus[win])
    return tau if np.isfinite(tau) and tau > 0 else 1.0


class EnsembleSampler(object):
    def __init__(self, nwalkers, ndim, log_prob_fn, a=2.0, vectorize=True, seed=None):
        if nwalkers < 2 * ndim or nwalkers % 2:
            raise ValueError("nwalkers must be even and at least 2*ndim (got %d for ndim %d)" % (nwalkers, ndim))
        self.nwalkers, self.ndim, self.a = int(nwalkers), int(ndim), float(a)
        self.log_prob_fn, self.vectorize = log_prob_fn, vectorize
        self.random = np.random.RandomState(seed)
        self.chain = np.empty((self.nwalkers, 0, self.ndim))
        self.lnprobability = np.empty((self.nwalkers, 0))
        self.naccepted = np.zeros(self.nwalkers)
        self.iterations = 0
        self.nevals = 0

    def _lnprob(self, block):
        self.nevals += len(block)
        if self.vectorize:
            lp = np.asarray(self.log_prob_fn(block), dtype=np.float64)
        else:
            lp = np.array([self.log_prob_fn(row) for row in block], dtype=np.float64)
        if np.isnan(lp).any():
            raise ValueError("log_prob_fn returned NaN")
        return lp

    def run_mcmc(self, pos, nsteps, rstate0=None, lnprob0=None):
        if rstate0 is not None:
            self.random.set_state(rstate0)
        p = np.array(pos, dtype=np.float64)
        if p.shape != (self.nwalkers, self.ndim):
            raise ValueError("pos must be (nwalkers, ndim)")
        lp = self._lnprob(p) if lnprob0 is None else np.array(lnprob0, dtype=np.float64)
        W, half = self.nwalkers, self.nwalkers // 2
        chain = np.empty((W, nsteps, self.ndim))
        lnps = np.empty((W, nsteps))
        for it in range(nsteps):
            perm = self.random.permutation(W)
            sets = (perm[:half], perm[half:])
            for s in (0, 1):
                act, oth = sets[s], sets[1 - s]
                zz = ((self.a - 1.0) * self.random.rand(half) + 1.0) ** 2 / self.a
                partner = oth[self.random.randint(len(oth), size=half)]
                prop = p[partner] - (p[partner] - p[act]) * zz[:, None]
                newlp = self._lnprob(prop)
                with np.errstate(invalid="ignore"):          # -inf - -inf = nan: never accepted
                    lnq = (self.ndim - 1.0) * np.log(zz) + newlp - lp[act]
                    acc = np.log(self.random.rand(half)) < lnq
                acc &= np.isfinite(newlp)
                idx = act[acc]
                p[idx] = prop[acc]
                lp[idx] = newlp[acc]
                self.naccepted[idx] += 1
            chain[:, it] = p
            lnps[:, it] = lp
        self.chain = np.concatenate([self.chain, chain], axis=1)
        self.lnprobability = np.concatenate([self.lnprobability, lnps], axis=1)
        self.iterations += nsteps
        return p, lp, self.random.get_state()

    @property
    def acceptance_fraction(self):
        return self.naccepted / max(self.iterations, 1)

    @property
    def flatchain(self):
        return self.chain.reshape(-1, self.ndim)

    def get_autocorr_time(self, c=5.0):
        return np.array([integrated_time(self.chain[:, :, d].T, c=c) for d in range(self.ndim)])

    @property
    def acor(self):
        return self.get_autocorr_time()
