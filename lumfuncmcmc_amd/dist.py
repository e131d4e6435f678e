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

    def __init__(self, local_eval, ndim, device, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.local_eval, self.ndim, self.device, self.group = local_eval, ndim, device, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._buffers = {}
        # RCCL/NCCL gathers in place (the input is the rank's slice of the output); other backends
        # (gloo in the CPU tests and one-GPU rehearsals) get a separate input buffer
        self._inplace = dist.is_initialized() and dist.get_backend(group) == "nccl"
        try:
            import inspect
            self._eval_takes_out = len(inspect.signature(local_eval).parameters) >= 2
        except (TypeError, ValueError):
            self._eval_takes_out = False

    def evaluate_tensor(self, theta):
        """theta: (B, ndim) float64 tensor on self.device -> (B,) tensor on self.device.  The gather
        buffer is kept per B and the local slice is written straight into it (in-place all-gather:
        no staging copy, no allocation per call)."""
        torch, dist = self.torch, self.dist
        B = theta.shape[0]
        if self.world == 1:
            return self.local_eval(theta)
        bounds, per = slice_bounds(B, self.world)
        lo, hi = bounds[self.rank]
        full = self._buffers.get(B)
        if full is None:
            full = torch.full((per * self.world,), float("-inf"), dtype=torch.float64, device=self.device)
            self._buffers[B] = full
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
        dist.all_gather_into_tensor(full, mine, group=self.group)
        if not self._inplace and self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        if per * self.world == B:
            return full
        return torch.cat([full[r * per:r * per + (b - a)] for r, (a, b) in enumerate(bounds)])

    def __call__(self, theta):
        t = self.torch.as_tensor(np.ascontiguousarray(theta, dtype=np.float64)).to(self.device)
        return self.evaluate_tensor(t.reshape(-1, self.ndim)).cpu().numpy()
