"""Experiment: does replaying the three launches of an evaluation as a captured graph shorten the period of back-to-back
evaluations?  (torch's graph capture as the harness; the library launches on the capturing stream.)
    python tools/graph_period.py [--nsrc N] [--rows B] [--calls K]"""
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
    a = ap.parse_args()
    model = bench.build_model("free", a.nsrc, 2 * a.rows, 0)
    ctx = model.context()
    th = [torch.from_numpy(synth.walkers("free", a.rows, seed=s)).cuda() for s in (1, 2, 3, 4)]
    outs = [torch.empty(a.rows, dtype=torch.float64, device="cuda") for _ in range(4)]
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    for i in range(200):
        ctx.lnprob_torch(th[i % 4], out=outs[i % 4])
    torch.cuda.synchronize()
    ref = [o.clone() for o in outs]

    def period(fn, label):
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(a.calls):
                fn(i)
            th_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print("%-34s %.2f us per evaluation (host %.2f)" % (label, 1e6 * dt / a.calls, 1e6 * th_host / a.calls), flush=True)

    period(lambda i: ctx.lnprob_torch(th[i % 4], out=outs[i % 4]), "plain launches")
    graphs = []
    for j in range(4):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            ctx.lnprob_torch(th[j], out=outs[j])
        graphs.append(g)
    torch.cuda.set_stream(st)
    for o in outs:
        o.zero_()
    for j in range(4):
        graphs[j].replay()
    torch.cuda.synchronize()
    print("graph replay equals plain launches:", all(torch.equal(outs[j], ref[j]) for j in range(4)))
    period(lambda i: graphs[i % 4].replay(), "one graph per evaluation")
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=st):
        for j in range(4):
            ctx.lnprob_torch(th[j], out=outs[j])
    torch.cuda.set_stream(st)
    a.calls //= 4
    period(lambda i: g2.replay(), "one graph per 4 evaluations (x4)")


if __name__ == "__main__":
    main()
