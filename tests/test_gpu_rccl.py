"""The RCCL path on the one GPU a test box has: a one-rank "nccl" process group with the collectives forced
on (the calls, buffers and stream ordering of the multi-GPU path, a degenerate gather).  The two-rank forms
run on gloo (tests/test_gpu_sampler.py, tests/test_dist_gloo.py); the N > 1 RCCL runs are the driver's."""
import socket

import numpy as np
import pytest

from lf_testlib import make_inputs, synth

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_walker_sharded_paths_over_rccl_with_one_rank():
    import torch
    import torch.distributed as dist
    from lumfuncmcmc_amd.capi import LFContext
    from lumfuncmcmc_amd.dist import ShardedLnProb
    from lumfuncmcmc_amd.sampler import DeviceEnsembleSampler
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1, device_id=dev)
    try:
        inp = make_inputs("free", 20000, seed=51)
        ctx = LFContext(inp, device=0)
        th = synth.walkers("free", 48, seed=52)
        direct = ctx.lnprob_batch(th)
        side = torch.cuda.Stream(device=dev)                      # what bench.py does: launches on a side stream
        with torch.cuda.stream(side):
            sh = ShardedLnProb(ctx.lnprob_torch, ctx.ndim, dev, force_collective=True)
            t = torch.from_numpy(th).to(dev)
            got = [sh.evaluate_tensor(t).clone() for _ in range(3)]
            side.synchronize()
        assert sh._inplace                                        # RCCL took the in-place gather
        for g in got:
            assert np.array_equal(g.cpu().numpy(), direct)
        # the sharded device sampler, same group: chain identical to the fused single-GPU sampler
        W = 32
        p0 = synth.walkers("free", W, seed=53)
        a = DeviceEnsembleSampler(ctx, W, seed=9, capacity=6)
        a.run_mcmc(p0, 6)
        b = DeviceEnsembleSampler(ctx, W, seed=9, capacity=6)
        b.enqueue_sharded(p0, 6, force_collective=True)
        b.sync()
        assert np.array_equal(a.chain, b.chain) and np.array_equal(a.lnprobability, b.lnprobability)
        a.close()
        b.close()
        ctx.close()
    finally:
        dist.destroy_process_group()
