"""Wall-clock period of back-to-back lnprob calls (the host enqueues ahead), by profiling level.
    python tools/call_period.py [--nsrc N] [--rows B] [--calls K]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from lumfuncmcmc_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nsrc", type=int, default=1000000)
    ap.add_argument("--rows", type=int, default=128)
    ap.add_argument("--calls", type=int, default=4000)
    ap.add_argument("--variant", default="free")
    ap.add_argument("--opts", default="", help="context options, k=v,k=v")
    ap.add_argument("--levels", default="0,1,2,0", help="profiling levels to run")
    ap.add_argument("--lib", default="", help="A/B: another build of the library (path to a .so)")
    ap.add_argument("--default-stream", action="store_true", help="launch on the null stream instead of a stream of our own")
    a = ap.parse_args()
    if a.lib:
        from lumfuncmcmc_amd import capi
        capi.LIB_PATH = os.path.abspath(a.lib)
    model = bench.build_model(a.variant, a.nsrc, 2 * a.rows, 0)
    ctx = model.context()
    th = [torch.from_numpy(synth.walkers(a.variant, a.rows, seed=s)).cuda() for s in (1, 2, 3, 4)]
    if not a.default_stream:
        st = torch.cuda.Stream()
        torch.cuda.set_stream(st)
    for kv in filter(None, a.opts.split(",")):
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    for level in [int(x) for x in a.levels.split(",")]:
        ctx.set_profiling(level)
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(a.calls):
                ctx.lnprob_torch(th[i % 4])
            th_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ctx.kernel_times()
        print("profiling %d: %.2f us per call (host enqueue %.2f us per call) %s" % (level, 1e6 * dt / a.calls, 1e6 * th_host / a.calls,
                                                                                   ctx.last_launch()["kernel"] + (" fused" if ctx.last_launch()["fused"] else "")), flush=True)
    ctx.set_profiling(0)


if __name__ == "__main__":
    main()
