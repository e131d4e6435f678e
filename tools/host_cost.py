"""Host-side cost of one evaluation call, where the GPU is not the bottleneck (small catalogue, 16 rows): the whole
Python path, the ctypes call alone, and an empty Python loop for scale.   python tools/host_cost.py"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from lumfuncmcmc_amd import synth  # noqa: E402


def per_call(fn, n=20000):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return 1e6 * (t1 - t0) / n, 1e6 * (time.perf_counter() - t0) / n


def main():
    for nsrc, rows, opts in ((1000, 16, {}), (100000, 16, {}), (100000, 16, {"persistent": 2})):
        model = bench.build_model("free", nsrc, 2 * rows, 0)
        ctx = model.context()
        for k, v in opts.items():
            ctx.set_option(k, v)
        th = torch.from_numpy(synth.walkers("free", rows, seed=1)).cuda()
        out = torch.empty(rows, dtype=torch.float64, device="cuda")
        st = torch.cuda.Stream()
        torch.cuda.set_stream(st)
        print("N = %d, %d rows, %s" % (nsrc, rows, opts))
        print("  lnprob_torch(theta)            host %.2f us, with the device %.2f us per call" % per_call(lambda: ctx.lnprob_torch(th)))
        print("  lnprob_torch(theta, out=out)   host %.2f us, with the device %.2f us per call" % per_call(lambda: ctx.lnprob_torch(th, out=out)))
        L, h, tp, op, sp = ctx._lib, ctx._h, th.data_ptr(), out.data_ptr(), st.cuda_stream
        f = L.lf_lnprob_batch_device
        print("  the C entry point alone        host %.2f us, with the device %.2f us per call   (%s)" % (
            per_call(lambda: f(h, tp, rows, op, sp)) + (ctx.last_launch()["kernel"] + (" fused" if ctx.last_launch()["fused"] else ""),)))
        torch.cuda.set_stream(torch.cuda.default_stream())
        ctx.close()
    print("  an empty lambda                host %.2f us" % per_call(lambda: None)[0])


if __name__ == "__main__":
    main()
