"""Why do proposals of a z-evolving chain miss the cells?  Evaluates a running chain's proposals-like positions through lf_main with
the census on: careful terms = fields whose bounds are inconclusive; per-source exponentials = safe fields of walkers off the cells."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from lumfuncmcmc_amd import synth
from lumfuncmcmc_amd.sampler import DeviceEnsembleSampler
model = bench.build_model("zevol", 1000000, 256, 0)
ctx = model.context()
W = 256
ds = DeviceEnsembleSampler(ctx, W, seed=1, capacity=300)
ds.run_mcmc(synth.walkers("zevol", W, seed=1), 120)
pos, lp, _ = ds.sync()
rng = np.random.default_rng(0)
# stretch-move proposals from the current ensemble
z = ((2.0 - 1) * rng.random(W // 2) + 1) ** 2 / 2.0
j = rng.integers(W // 2, W, W // 2)
prop = pos[j] - (pos[j] - pos[:W // 2]) * z[:, None]
ctx.set_option("persistent", 0)
ctx.set_option("count_forms", 1)
out = ctx.lnprob_batch(prop)
fc = ctx.form_counts()
print("proposals: %d, -inf %d" % (len(out), np.isinf(out).sum()))
print({k: v for k, v in fc.items() if v})
print("careful terms / N = %.2f walker-equivalents; plain per-source exponentials / N = %.2f" % (fc["careful"] / 1e6, (fc["general"] + fc["table"]) / 1e6))
