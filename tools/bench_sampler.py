"""Ensemble steps per second: host-side stretch move on the batched boundary vs the device-resident
sampler (theta never leaves HBM).  python tools/bench_sampler.py [nsrc walkers nsteps]..."""
import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
import bench
from lumfuncmcmc_amd import synth
from lumfuncmcmc_amd.sampler import DeviceEnsembleSampler, EnsembleSampler

cfgs = [(1000, 32, 200), (100000, 256, 100), (1000000, 256, 50)]
for nsrc, W, nsteps in cfgs:
    m = bench.build_model("free", nsrc, W, 0)
    ctx = m.context()
    pos = synth.walkers("free", W, seed=3)
    hs = EnsembleSampler(W, ctx.ndim, ctx.lnprob_batch, seed=1)
    hs.run_mcmc(pos, 3)
    t = time.perf_counter(); hs.run_mcmc(pos, nsteps); th = time.perf_counter() - t
    ds = DeviceEnsembleSampler(ctx, W, seed=1, capacity=nsteps + 3)
    ds.run_mcmc(pos, 3)
    t = time.perf_counter(); ds.run_mcmc(None, nsteps); td = time.perf_counter() - t
    print("N=%d W=%d: host sampler %.3f ms/step (%.0f evals/s), device sampler %.3f ms/step (%.0f evals/s), acc %.2f / %.2f"
          % (nsrc, W, th / nsteps * 1e3, W * nsteps / th, td / nsteps * 1e3, W * nsteps / td,
             hs.acceptance_fraction.mean(), ds.acceptance_fraction.mean()))
    ds.close(); m.close()
