"""A/B of a device-sampler half-step between two builds of the library on one box:
   python tools/sampler_lib_ab.py <variant> [other_lib.so]"""
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
variant = sys.argv[1]
if len(sys.argv) > 2:
    from lumfuncmcmc_amd import capi
    capi.LIB_PATH = os.path.abspath(sys.argv[2])
import bench
from lumfuncmcmc_amd import synth
from lumfuncmcmc_amd.sampler import DeviceEnsembleSampler
model = bench.build_model(variant, 1000000, 256, 0)
ctx = model.context()
W = 256
ds = DeviceEnsembleSampler(ctx, W, seed=1, capacity=300)
ds.run_mcmc(synth.walkers(variant, W, seed=1), 20)
torch.cuda.synchronize()
for rep in range(2):
    t = time.perf_counter()
    ds.enqueue(None, 100)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
ds.sync()
lp = ds.lnprobability
print(variant, sys.argv[2:] or "this build", "%.1f us per half-step, %.3e evals/s" % (dt / 200 * 1e6, W * 100 / dt), "chain checksum %.17g" % float(lp[np.isfinite(lp)].sum()), "acc %.3f" % ds.acceptance_fraction.mean())
