"""Walker sharding across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path shards by walker: the catalogue and grids are replicated on every GPU (32 MB at 10^6
sources), each rank evaluates a contiguous slice of every (B, ndim) block and one all-gather of
B/G doubles per rank returns the whole block to every rank before the stretch move - the only
exchange step of the path (no data-path collective on the catalogue).  The payload is a few
hundred bytes, so the collective is latency-bound; it is issued on the same stream as the
kernels and needs no host round trip.
"""
import numpy as np


def slice_bounds(B, world):
    """Equal contiguous slices, the last ones padded: rows [lo, hi) of rank r, and the common
    padded slice length."""
    per = (B + world - 1) // world
    return [(min(r * per, B), min((r + 1) * per, B)) for r in range(world)], per


class ShardedLnProb(object):
    """Callable (B, ndim) block -> (B,) log-posterior, evaluated 1/world per rank.

    local_eval(theta_slice: tensor (b, ndim) on `device`) -> tensor (b,) on `device`; in the
    product this is LFContext.lnprob_torch.  Every rank must call with the same block.
    """

    def __init__(self, local_eval, ndim, device, group=None, force_collective=False, check_theta=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.local_eval, self.ndim, self.device, self.group = local_eval, ndim, device, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._buffers = {}
        self._flip = {}
        self._pending = {}                  # (B, flip) -> work handle of a deferred gather into that buffer
        # check_theta: every call first verifies (one small all-reduce) that all ranks passed the SAME block - the
        # silent failure of this scheme is ranks whose samplers drifted apart (different seeds / starts)
        self.check_theta = bool(check_theta)
        # a one-rank group normally skips the collective; force_collective keeps it (one-GPU rehearsal of the
        # RCCL path: same calls, same stream ordering, a degenerate gather)
        self._collective = self.world > 1 or (force_collective and dist.is_initialized())
        # RCCL/NCCL gathers in place (the input is the rank's slice of the output); other backends
        # (gloo in the CPU tests and one-GPU rehearsals) get a separate input buffer
        self._inplace = dist.is_initialized() and dist.get_backend(group) == "nccl"
        try:
            import inspect
            self._eval_takes_out = len(inspect.signature(local_eval).parameters) >= 2
        except (TypeError, ValueError):
            self._eval_takes_out = False

    def _assert_same_theta(self, theta):
        torch, dist = self.torch, self.dist
        t = theta.detach().to(torch.float64)
        # min and max over ranks of two position-weighted checksums must coincide
        wts = torch.arange(1, t.numel() + 1, dtype=torch.float64, device=t.device)
        flat = torch.nan_to_num(t.reshape(-1), nan=1.0e300, posinf=1.0e301, neginf=-1.0e301)
        cs = torch.stack([flat.sum(), (flat * wts).sum()])
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        if not bool((lo == hi).all().item()):
            raise RuntimeError("ShardedLnProb: the ranks passed different theta blocks (rank %d): their samplers have "
                               "diverged - broadcast the start positions and the sampler seed from rank 0" % self.rank)

    def flush(self):
        """Wait (on the current stream; the host does not block for RCCL) for every deferred gather."""
        for k in list(self._pending):
            self._pending.pop(k).wait()

    def evaluate_tensor(self, theta, defer=False):
        """theta: (B, ndim) float64 tensor on self.device -> (B,) tensor on self.device.  The local slice is
        written straight into the gather buffer (in-place all-gather: no staging copy, no allocation per call).
        Two gather buffers per B are used in turn, so the result of a call stays valid while the NEXT call with the
        same B runs (a sampler holds half 0's lnprob while half 1 is evaluated); it is overwritten by the call after.

        defer = True: the gather is issued asynchronously (torch.distributed runs collectives on a stream of its own) and
        NOT waited for here - the next evaluation's kernel overlaps it; the returned tensor is complete after flush() (or
        after the second next call with the same B, which waits before it reuses the buffer).  For callers with independent
        blocks to evaluate (bench.py's plain loop); a sampler needs block k before it can propose block k + 1 and keeps the
        default."""
        torch, dist = self.torch, self.dist
        B = theta.shape[0]
        if not self._collective:
            return self.local_eval(theta)
        if self.check_theta:
            self._assert_same_theta(theta)
        bounds, per = slice_bounds(B, self.world)
        lo, hi = bounds[self.rank]
        flip = self._flip[B] = 1 - self._flip.get(B, 1)
        prev = self._pending.pop((B, flip), None)
        if prev is not None:
            prev.wait()                         # the gather that last used this buffer (two calls ago)
        full = self._buffers.get((B, flip))
        if full is None:
            full = torch.full((per * self.world,), float("-inf"), dtype=torch.float64, device=self.device)
            self._buffers[(B, flip)] = full
        mine = full[self.rank * per:(self.rank + 1) * per]
        if not self._inplace:
            mine = self._buffers.setdefault(("in", B), torch.full((per,), float("-inf"), dtype=torch.float64,
                                                                  device=self.device))
        if hi > lo:
            out = mine[:hi - lo]
            res = self.local_eval(theta[lo:hi].contiguous(), out) if self._eval_takes_out else self.local_eval(theta[lo:hi].contiguous())
            if res.data_ptr() != out.data_ptr():
                out.copy_(res)
        if not self._inplace and self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()        # gloo (rehearsal) does not order with our launches
        if defer:
            self._pending[(B, flip)] = dist.all_gather_into_tensor(full, mine, group=self.group, async_op=True)
            return full[:B]
        try:
            dist.all_gather_into_tensor(full, mine, group=self.group)
        except RuntimeError:
            if not self._inplace:
                raise
            # a backend build that refuses the in-place form: gather from a separate input from now on
            self._inplace = False
            sep = self._buffers.setdefault(("in", B), torch.empty((per,), dtype=torch.float64, device=self.device))
            sep.copy_(mine)
            dist.all_gather_into_tensor(full, sep, group=self.group)
        if not self._inplace and self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        # rank r's rows start at r * per in the block as well as in the gather buffer: the first B entries ARE the
        # block's lnprob in row order (only the tail is padding)
        return full[:B]

    def __call__(self, theta):
        t = self.torch.as_tensor(np.ascontiguousarray(theta, dtype=np.float64)).to(self.device)
        # (a copy: on a CPU device .numpy() would be a view of the gather buffer, which the call after next overwrites)
        return np.array(self.evaluate_tensor(t.reshape(-1, self.ndim)).cpu().numpy(), copy=True)


def shard_sources(inp, rank, world):
    """Kernel inputs (LFContext's dict) of rank `rank` when the CATALOGUE is sharded: every field's
    sources are cut in `world` contiguous pieces and the rank keeps its piece of each field, so field
    membership survives; grids and parameters are replicated."""
    fi = np.asarray(inp["field_ind"], dtype=np.int64)
    sel, new_fi = [], [0]
    for f in range(len(fi) - 1):
        n = int(fi[f + 1] - fi[f])
        b, _ = slice_bounds(n, world)
        lo, hi = b[rank]
        sel.append(np.arange(fi[f] + lo, fi[f] + hi))
        new_fi.append(new_fi[-1] + (hi - lo))
    sel = np.concatenate(sel) if sel else np.zeros(0, dtype=np.int64)
    out = dict(inp)
    for k in ("lum", "z", "DLz", "Om_arr", "logf"):
        if out.get(k) is not None:
            out[k] = np.asarray(out[k])[sel]
    out["field_ind"] = np.array(new_fi, dtype=np.int64)
    return out


class SourceShardedLnProb(object):
    """The other way to use several GPUs (for batches smaller than a walker tile per GPU):
    every rank holds 1/world of the catalogue and evaluates ALL theta rows on it; the per-source sums
    (and the closed-form walker part, itself a sum over sources) add up, the expected-count integral is
    split by node chunks over the ranks in the same way ("grid_share"), and one all-reduce(SUM) of B doubles
    gives lnprob on every rank.  A row
    outside the prior or underflowing on any rank is -inf on that rank, hence in the sum.  Only the
    summation order differs from the single-GPU result."""

    def __init__(self, inp, device, group=None):
        import torch
        import torch.distributed as dist
        from .capi import LFContext
        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.device("cuda", device)
        self.ctx = LFContext(shard_sources(inp, self.rank, self.world), device=device)
        if self.world > 1:
            # piece B is split like piece A: rank r integrates the node chunks c with c % world == r
            self.ctx.set_option("grid_share", self.rank + 65536 * self.world)
        self.ndim = self.ctx.ndim
        self._sync = dist.is_initialized() and dist.get_backend(group) != "nccl"

    def evaluate_tensor(self, theta):
        out = self.ctx.lnprob_torch(theta)
        if self.world > 1:
            if self._sync:
                self.torch.cuda.current_stream(self.device).synchronize()     # gloo rehearsal, see ShardedLnProb
            self.dist.all_reduce(out, op=self.dist.ReduceOp.SUM, group=self.group)
            if self._sync:
                self.torch.cuda.synchronize(self.device)
        return out

    def __call__(self, theta):
        t = self.torch.as_tensor(np.ascontiguousarray(theta, dtype=np.float64)).to(self.device)
        return self.evaluate_tensor(t.reshape(-1, self.ndim)).cpu().numpy()

    def close(self):
        self.ctx.close()
