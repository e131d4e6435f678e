"""Where the compressed-mode lf_main time goes: full launch vs the launch without the grid part."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from lumfuncmcmc_amd import synth
m = bench.build_model("free", 1000000, 256, 0)
ctx = m.context()
ctx.set_option("compress", 1)
for B in (128, 1024, 4096):
    th = torch.from_numpy(synth.walkers("free", B, seed=3)).cuda()
    for skip in (0, 1):
        ctx.set_option("skip_grid", skip)
        for _ in range(5): ctx.lnprob_torch(th)
        torch.cuda.synchronize(); ctx.kernel_times(); ctx.set_profiling(1)
        for _ in range(20): ctx.lnprob_torch(th)
        torch.cuda.synchronize(); ctx.set_profiling(0)
        kt = ctx.kernel_times()
        print("B=%d skip_grid=%d: lf_main %.1f us" % (B, skip, kt["main"]["ms"] / kt["main"]["launches"] * 1e3), flush=True)
m.close()
