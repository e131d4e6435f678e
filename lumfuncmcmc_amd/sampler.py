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


class DeviceEnsembleSampler(object):
    """The same read-back surface, with the whole stretch move on the GPU (lf_sampler_* of
    include/lfmcmc.h): positions, lnprob and the chain stay in HBM, a step is six kernel launches
    and no host round trip.  Parallel stretch move with two fixed half-ensembles (emcee 2.x form);
    Philox4x32-10 random numbers keyed by `seed`, so (seed, start) determines the chain
    (tests/test_gpu_sampler.py replays it on the host)."""

    def __init__(self, ctx, nwalkers, a=2.0, seed=0, capacity=1000):
        import ctypes
        if nwalkers < 2 * ctx.ndim or nwalkers % 2:
            raise ValueError("nwalkers must be even and at least 2*ndim (got %d for ndim %d)" % (nwalkers, ctx.ndim))
        self._ct = ctypes
        self.ctx, self.nwalkers, self.ndim, self.a, self.seed = ctx, int(nwalkers), ctx.ndim, float(a), int(seed)
        self.capacity = int(capacity)
        h = ctx._lib.lf_sampler_create(ctx._h, self.nwalkers, self.a, ctypes.c_uint64(self.seed), self.capacity)
        if not h:
            raise RuntimeError("lf_sampler_create failed: %s" % ctx._lib.lf_last_error(ctx._h).decode())
        self._h = ctypes.c_void_p(h)
        self._started = False

    def _p(self, a):
        return a.ctypes.data_as(self._ct.POINTER(self._ct.c_double)) if a is not None else None

    def run_mcmc(self, pos, nsteps, rstate0=None, lnprob0=None):
        """pos=None continues from the current state.  Enqueues and returns after the last step has
        been read back (use `enqueue` + `sync` to overlap with host work)."""
        self.enqueue(pos, nsteps, lnprob0)
        return self.sync()

    def enqueue(self, pos, nsteps, lnprob0=None):
        lib = self.ctx._lib
        if pos is not None or not self._started:
            p = np.ascontiguousarray(pos, dtype=np.float64)
            if p.shape != (self.nwalkers, self.ndim):
                raise ValueError("pos must be (nwalkers, ndim)")
            l0 = None if lnprob0 is None else np.ascontiguousarray(lnprob0, dtype=np.float64)
            self.ctx._check(lib.lf_sampler_start(self._h, self._p(p), self._p(l0)))
            self._started = True
        self.ctx._check(lib.lf_sampler_run(self._h, int(nsteps), None))

    def enqueue_sharded(self, pos, nsteps, group=None, lnprob0=None, force_collective=False, shard="walkers"):
        """The same chain with every half-step sharded over the ranks of a torch.distributed group (one process
        per GPU).

        shard="walkers": every rank holds the whole catalogue; propose everywhere, evaluate the local slice of the
        half, all-gather the slices' lnprob (RCCL, in stream order), accept everywhere.  The chain is bit-identical
        to the one-GPU chain.
        shard="sources": this sampler's context holds 1/world of the catalogue (dist.shard_sources) and its share of
        the grid ("grid_share"); every rank evaluates the WHOLE half on its shard, one all-reduce(SUM) of the half's
        lnprob, accept everywhere.  Only the summation order differs from the one-GPU chain (1e-13)."""
        import torch
        import torch.distributed as dist
        from .dist import slice_bounds
        lib, ct = self.ctx._lib, self._ct
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        dev = torch.device("cuda", self.ctx.device)
        half = self.nwalkers // 2
        # force_collective: keep the collective with a one-rank group (one-GPU rehearsal of the RCCL path)
        collective = world > 1 or (force_collective and dist.is_initialized())
        nccl = collective and dist.get_backend(group) == "nccl"
        stream = torch.cuda.current_stream(dev).cuda_stream

        def fence_before():           # gloo (rehearsal) does not order with our launches
            if collective and not nccl:
                torch.cuda.current_stream(dev).synchronize()

        def fence_after():            # ... nor its result with the next launch
            if collective and not nccl:
                torch.cuda.synchronize(dev)

        by_source = shard == "sources"
        if shard not in ("walkers", "sources"):
            raise ValueError("shard must be 'walkers' or 'sources'")
        if pos is not None or not self._started:
            p = np.ascontiguousarray(pos, dtype=np.float64)
            l0 = None if lnprob0 is None else np.ascontiguousarray(lnprob0, dtype=np.float64)
            if by_source and l0 is None and collective:
                # the start's lnprob is a sum over the shards too
                t0 = self.ctx.lnprob_torch(torch.from_numpy(p).to(dev))
                fence_before()
                dist.all_reduce(t0, op=dist.ReduceOp.SUM, group=group)
                fence_after()
                l0 = np.ascontiguousarray(t0.cpu().numpy())
            self.ctx._check(lib.lf_sampler_start(self._h, self._p(p), self._p(l0)))
            self._started = True
        if by_source:
            buf = torch.empty((half,), dtype=torch.float64, device=dev)
            for _ in range(int(nsteps)):
                for h in (0, 1):
                    self.ctx._check(lib.lf_sampler_half_eval(self._h, h, 0, half, ct.c_void_p(buf.data_ptr()), ct.c_void_p(stream)))
                    if collective:
                        fence_before()
                        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
                        fence_after()
                    self.ctx._check(lib.lf_sampler_half_accept(self._h, h, ct.c_void_p(buf.data_ptr()), ct.c_void_p(stream)))
            self._keep = buf
            return
        bounds, per = slice_bounds(half, world)
        lo, hi = bounds[rank]
        # the gather buffer is [world][per]; rank r's rows start at r * per in the half as well, so its first `half`
        # entries are the half's lnprob in walker order (no re-packing for ragged splits)
        buf = torch.full((per * world,), float("-inf"), dtype=torch.float64, device=dev)
        sep = None if (nccl or not collective) else torch.full((per,), float("-inf"), dtype=torch.float64, device=dev)   # RCCL gathers in place
        mine = buf[rank * per:(rank + 1) * per] if sep is None else sep
        for _ in range(int(nsteps)):
            for h in (0, 1):
                self.ctx._check(lib.lf_sampler_half_eval(self._h, h, lo, hi, ct.c_void_p(mine.data_ptr() - lo * 8),
                                                         ct.c_void_p(stream)))
                if collective:
                    fence_before()
                    dist.all_gather_into_tensor(buf, mine, group=group)
                    fence_after()
                self.ctx._check(lib.lf_sampler_half_accept(self._h, h, ct.c_void_p(buf.data_ptr()), ct.c_void_p(stream)))
        self._keep = buf

    def sync(self):
        lib = self.ctx._lib
        t = int(lib.lf_sampler_steps(self._h))
        W, nd = self.nwalkers, self.ndim
        self.chain = np.empty((W, t, nd))
        self.lnprobability = np.empty((W, t))
        self.naccepted = np.empty(W, dtype=np.int64)
        pos, lp = np.empty((W, nd)), np.empty(W)
        self.ctx._check(lib.lf_sampler_read(self._h, self._p(self.chain), self._p(self.lnprobability),
                                            self.naccepted.ctypes.data_as(self._ct.POINTER(self._ct.c_int64)),
                                            self._p(pos), self._p(lp)))
        self.iterations = t
        return pos, lp, None

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

    def close(self):
        if getattr(self, "_h", None) is not None:
            self.ctx._lib.lf_sampler_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
