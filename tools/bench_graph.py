"""Device sampler, plain launches vs hipGraph replay, direct and compressed catalogue.
python tools/bench_graph.py"""
import sys, time
sys.path.insert(0, '/root/repo')
import bench
from lumfuncmcmc_amd import synth
from lumfuncmcmc_amd.sampler import DeviceEnsembleSampler

for variant, nsrc, W, nsteps in [("free", 1000, 32, 400), ("free", 1000, 256, 400), ("free", 100000, 256, 200),
                                 ("free", 1000000, 256, 60), ("zevol", 800000, 512, 60)]:
    m = bench.build_model(variant, nsrc, W, 0)
    ctx = m.context()
    pos = synth.walkers(variant, W, seed=3)
    row = []
    for compress in (0, 1):
        ctx.set_option("compress", compress)
        for graph in (0, 1):
            ctx.set_option("graph", graph)
            ds = DeviceEnsembleSampler(ctx, W, seed=1, capacity=nsteps + 5)
            ds.run_mcmc(pos, 5)
            t = time.perf_counter(); ds.run_mcmc(None, nsteps); td = time.perf_counter() - t
            row.append("%s%s %.1f us/step (%.3g evals/s)" % ("cmp " if compress else "", "graph" if graph else "plain",
                                                             td / nsteps * 1e6, W * nsteps / td))
            ds.close()
    print("%s N=%d W=%d: " % (variant, nsrc, W) + " | ".join(row), flush=True)
    m.close()
