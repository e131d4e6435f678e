"""Which forms of the term do the device sampler's proposals take?  (Walkers that cannot be summed over the cells cost
their tile a pass over the whole catalogue.)   python tools/sampler_census.py [--nsrc N] [--walkers W] [--steps K]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from lumfuncmcmc_amd import synth  # noqa: E402
from lumfuncmcmc_amd.sampler import DeviceEnsembleSampler  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nsrc", type=int, default=1000000)
    ap.add_argument("--walkers", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--opts", default="", help="context options for the timed part, k=v,k=v")
    ap.add_argument("--variant", default="free")
    ap.add_argument("--ball", type=float, default=0.0, help="start in a ball of this relative size around the box centre instead of box-uniform")
    a = ap.parse_args()
    model = bench.build_model(a.variant, a.nsrc, a.walkers, 0)
    ctx = model.context()
    th = synth.walkers(a.variant, a.walkers, seed=1)
    if a.ball > 0:
        lims = np.array([th.min(axis=0), th.max(axis=0)])
        mid = lims.mean(axis=0)
        th = mid + (th - mid) * a.ball
    ds = DeviceEnsembleSampler(ctx, a.walkers, seed=3)
    ds.run_mcmc(th, 3)
    ctx.set_option("count_forms", 1)
    ds.run_mcmc(None, a.steps)
    fc = ctx.form_counts()
    ctx.set_option("count_forms", 0)
    calls = 2 * a.steps
    rows = a.walkers // 2
    print("per half-step of %d proposals: %s" % (rows, {k: round(v / calls / (1 if k.startswith("node") else 1), 1) for k, v in fc.items()}))
    print("(walker, source) terms per half-step if every proposal were summed over the sources: %d" % (rows * a.nsrc))
    import time
    import torch
    for kv in filter(None, a.opts.split(",")):
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ds.enqueue(None, a.steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ds.sync()
        t_read = time.perf_counter() - t0 - dt
    print("read-back of the chain so far (%d steps): %.2f ms" % (ds.iterations, 1e3 * t_read))
    ll = ctx.last_launch()
    print("%.1f us per half-step (%s%s); acceptance %.3f" % (1e6 * dt / calls, ll["kernel"], ", one launch" if ll["fused"] else "",
                                                          float(np.mean(ds.acceptance_fraction))))


if __name__ == "__main__":
    main()
