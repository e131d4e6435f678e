"""Large shapes outside the test suite: N = 10^7 sources, 2048 theta rows - direct vs compressed catalogue vs
the NumPy oracle (3 rows).  python tools/big_case.py [nsrc rows]"""
import sys, time
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle'); sys.path.insert(0, '/root/repo/tests')
import bench
from lumfuncmcmc_amd import synth
import lf_oracle as O
nsrc = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
for variant in ("free", "zevol"):
    t = time.perf_counter(); m = bench.build_model(variant, nsrc, 256, 0); ctx = m.context(); ts = time.perf_counter() - t
    th = synth.walkers(variant, B, seed=5)
    ctx.lnprob_batch(th[:8])
    t = time.perf_counter(); direct = ctx.lnprob_batch(th); td = time.perf_counter() - t
    t = time.perf_counter(); ctx.set_option("compress", 1); tb = time.perf_counter() - t
    ctx.lnprob_batch(th[:8])
    t = time.perf_counter(); comp = ctx.lnprob_batch(th); tc = time.perf_counter() - t
    rel = np.max(np.abs(comp - direct) / np.abs(direct))
    t = time.perf_counter(); ref = O.lnprob_batch(m.kernel_inputs(), th[:3]); to = time.perf_counter() - t
    relo = np.max(np.abs(direct[:3] - ref) / np.abs(ref))
    print("%s N=%d B=%d: setup %.1f s | direct %.3f s (%.3g evals/s) | compress build %.2f s, eval %.4f s (%.3g evals/s) | "
          "compressed vs direct %.2e | direct vs oracle (3 rows, %.1f s) %.2e" % (variant, nsrc, B, ts, td, B / td, tb, tc, B / tc, rel, to, relo), flush=True)
    m.close()
