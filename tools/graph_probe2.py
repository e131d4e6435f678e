import sys, time
sys.path.insert(0, '/root/repo')
import bench
from lumfuncmcmc_amd import synth
from lumfuncmcmc_amd.sampler import DeviceEnsembleSampler
m = bench.build_model("free", 1000, 256, 0)
ctx = m.context()
pos = synth.walkers("free", 256, seed=3)
def run(label, cap, taper=1, graph=1, seed=1, reps=3, n=200):
    ctx.set_option("graph", graph); ctx.set_option("taper", taper)
    ds = DeviceEnsembleSampler(ctx, 256, seed=seed, capacity=cap)
    ds.run_mcmc(pos, 5)
    out = []
    for rep in range(reps):
        t = time.perf_counter(); ds.run_mcmc(None, n); out.append((time.perf_counter() - t) / n * 1e6)
    print(label, " ".join("%.1f" % x for x in out), "us/step", flush=True)
    ds.close()
run("graph cap=605 (exact)", 605)
run("graph cap=5000", 5000)
run("graph cap=605 no taper", 605, taper=0)
run("plain cap=605", 605, graph=0)
run("graph cap=605 seed 2", 605, seed=2)
run("graph cap=605 again", 605)
m.close()
