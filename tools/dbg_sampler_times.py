import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
import bench
from lumfuncmcmc_amd import synth
from lumfuncmcmc_amd.sampler import DeviceEnsembleSampler, EnsembleSampler
for nsrc, W, nsteps in [(1000000, 256, 50), (100000, 256, 100)]:
    m = bench.build_model("free", nsrc, W, 0)
    ctx = m.context()
    pos = synth.walkers("free", W, seed=3)
    ds = DeviceEnsembleSampler(ctx, W, seed=1, capacity=2 * nsteps + 3)
    ds.run_mcmc(pos, 3)
    ctx.kernel_times(); ctx.set_profiling(2)
    t = time.perf_counter(); ds.run_mcmc(None, nsteps); td = time.perf_counter() - t
    kt = ctx.kernel_times(); ctx.set_profiling(0)
    print("device", nsrc, td / nsteps * 1e3, {k: (round(v["ms"] / max(v["launches"], 1), 4), v["launches"]) for k, v in kt.items()})
    t = time.perf_counter(); ds.run_mcmc(None, nsteps); td = time.perf_counter() - t
    print("device no-prof", nsrc, td / nsteps * 1e3)
    hs = EnsembleSampler(W, ctx.ndim, ctx.lnprob_batch, seed=1)
    hs.run_mcmc(pos, 3)
    ctx.kernel_times(); ctx.set_profiling(2)
    t = time.perf_counter(); hs.run_mcmc(pos, nsteps); th = time.perf_counter() - t
    kt = ctx.kernel_times(); ctx.set_profiling(0)
    print("host  ", nsrc, th / nsteps * 1e3, {k: (round(v["ms"] / max(v["launches"], 1), 4), v["launches"]) for k, v in kt.items()})
