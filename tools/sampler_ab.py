import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from lumfuncmcmc_amd import synth
from lumfuncmcmc_amd.sampler import DeviceEnsembleSampler
variant = sys.argv[1]
model = bench.build_model(variant, 1000000, 256, 0)
ctx = model.context()
for opts in ({}, {"fuse_step": 0}, {"persistent": 0}):
    for k, v in opts.items():
        ctx.set_option(k, v)
    W = 256
    ds = DeviceEnsembleSampler(ctx, W, seed=1, capacity=300)
    ds.run_mcmc(synth.walkers(variant, W, seed=1), 20)
    torch.cuda.synchronize()
    for rep in range(2):
        t = time.perf_counter()
        ds.enqueue(None, 100)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
    print(variant, opts, "%.1f us per half-step, %.3e evals/s" % (dt / 200 * 1e6, W * 100 / dt), ctx.last_launch()["kernel"], ctx.last_launch()["fused"], "acc %.2f" % ds.acceptance_fraction.mean())
    ds.close()
    for k in opts:
        ctx.set_option(k, 1)
