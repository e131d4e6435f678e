"""When do the sporadic tens-of-ms stalls happen?  Calls lnprob back to back on a side stream for a few seconds and
prints every call slower than 1 ms with its time since the first launch."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from lumfuncmcmc_amd import synth
t00 = time.perf_counter()
m = bench.build_model("free", 1000000, 256, 0)
ctx = m.context()
if "nospec" in sys.argv: ctx.set_option("specialise", 0)
side = torch.cuda.Stream() if len(sys.argv) < 2 or sys.argv[1] != "default" else None
if side is not None:
    torch.cuda.set_stream(side)
th = [torch.from_numpy(synth.walkers("free", 128 * 8, seed=1).reshape(8, 128, -1)[i].copy()).cuda() for i in range(8)]
print("setup done at %.2f s" % (time.perf_counter() - t00), flush=True)
t0 = time.perf_counter(); n = 0; slow = []
while time.perf_counter() - t0 < 6.0:
    t = time.perf_counter(); ctx.lnprob_torch(th[n % 8]); torch.cuda.synchronize(); dt = time.perf_counter() - t
    if dt > 1e-3: slow.append((round(t - t0, 3), round(dt * 1e3, 1), n))
    n += 1
print("calls", n, "slow calls (t since first call s, ms, index):", slow, flush=True)
m.close()
