"""Host cost of an evaluation call, apart from the queue's back-pressure: bursts of K calls enqueued on an idle stream
(K small enough that the queue never fills), timed on the host alone and up to the device's completion.
    python tools/host_burst.py [--nsrc N] [--rows B] [--burst K] [--variant free]"""
import argparse
import os
import sys
import time

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
    ap.add_argument("--burst", type=int, default=64)
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--variant", default="free")
    a = ap.parse_args()
    model = bench.build_model(a.variant, a.nsrc, 2 * a.rows, 0)
    ctx = model.context()
    th = torch.from_numpy(synth.walkers(a.variant, a.rows, seed=1)).cuda()
    out = torch.empty(a.rows, dtype=torch.float64, device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    f, h, tp, op, sp = ctx._lib.lf_lnprob_batch_device, ctx._h, th.data_ptr(), out.data_ptr(), st.cuda_stream
    import gc
    gc.disable()
    for name, call in (("lnprob_torch(theta, out=out)", lambda: ctx.lnprob_torch(th, out=out)),
                       ("the C entry point through ctypes", lambda: f(h, tp, a.rows, op, sp))):
        host, total = [], []
        for _ in range(a.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.burst):
                call()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            host.append(1e6 * (t1 - t0) / a.burst)
            total.append(1e6 * (t2 - t0) / a.burst)
        host, total = np.array(host[5:]), np.array(total[5:])
        print("%-34s bursts of %d: host %.2f us per call (median; p10 %.2f, p90 %.2f); enqueue + device %.2f us per call  [%s]" % (
            name, a.burst, np.median(host), np.percentile(host, 10), np.percentile(host, 90), np.median(total),
            ctx.last_launch()["kernel"] + (" fused" if ctx.last_launch()["fused"] else "")))
    t0 = time.perf_counter()
    for _ in range(100000):
        pass
    print("an empty loop iteration: %.3f us" % (1e6 * (time.perf_counter() - t0) / 100000))


if __name__ == "__main__":
    main()
