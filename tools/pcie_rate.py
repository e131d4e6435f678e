import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
import bench
from lumfuncmcmc_amd import synth
m = bench.build_model("free", 1000000, 256, 0)
ctx = m.context()
th = synth.walkers("free", 128, seed=1)
for _ in range(200): ctx.lnprob_batch(th)
t0=time.perf_counter()
n=3000
for _ in range(n): ctx.lnprob_batch(th)
dt=(time.perf_counter()-t0)/n
print("host-pointer entry lf_lnprob_batch, 128 rows: %.1f us per call -> %.3g evals/s" % (dt*1e6, 128/dt))
