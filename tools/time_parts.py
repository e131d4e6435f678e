"""lf_main's time by part (HIP events around the launch): the whole launch, the per-source part alone (skip_grid), and
what options change.  python tools/time_parts.py [--nsrc N] [--rows B]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from lumfuncmcmc_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nsrc", type=int, default=1000000)
    ap.add_argument("--rows", type=int, default=128)
    ap.add_argument("--variant", default="free")
    ap.add_argument("--sets", default="default;skip_grid=1;tables=0;tables=0,skip_grid=1;specialise=0")
    ap.add_argument("--lib", default="", help="A/B: another build of the library (path to a .so)")
    a = ap.parse_args()
    if a.lib:
        from lumfuncmcmc_amd import capi
        capi.LIB_PATH = os.path.abspath(a.lib)
    model = bench.build_model(a.variant, a.nsrc, 2 * a.rows, 0)
    ctx = model.context()
    th = [torch.from_numpy(synth.walkers(a.variant, a.rows, seed=s)).cuda() for s in (1, 2, 3, 4)]
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    # the GPU's clocks take tens of milliseconds of sustained load to settle (the same launch: 127 us in the first 10 ms,
    # 112 us after 30 ms): load it first, or the order of the sets decides the comparison
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        for i in range(16):
            ctx.lnprob_torch(th[i % 4])
        torch.cuda.synchronize()
    for spec in a.sets.split(";"):
        opts = [] if spec == "default" else [kv.split("=") for kv in spec.split(",")]
        for k, v in opts:
            ctx.set_option(k, int(v))
        for i in range(10):
            ctx.lnprob_torch(th[i % 4])
        torch.cuda.synchronize()
        ctx.kernel_times()
        ctx.set_profiling(2)
        for i in range(40):
            ctx.lnprob_torch(th[i % 4])
        torch.cuda.synchronize()
        ctx.set_profiling(0)
        kt = ctx.kernel_times()
        print("%-28s lf_main %.1f us  prepare %.1f us  finalize %.1f us  launch %s" % (
            spec, 1e3 * kt["main"]["ms"] / max(kt["main"]["launches"], 1), 1e3 * kt["prepare"]["ms"] / max(kt["prepare"]["launches"], 1),
            1e3 * kt["finalize"]["ms"] / max(kt["finalize"]["launches"], 1), ctx.last_launch()), flush=True)
        for k, v in opts:
            ctx.set_option(k, {"tables": 1, "specialise": 1, "persistent": 1, "cells": 1}.get(k, 0))      # back to the defaults


if __name__ == "__main__":
    main()
