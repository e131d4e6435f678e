"""State carried between calls of one context: batch sizes going up and down, the one-launch and three-launch forms,
cells on and off, the pieces entry, the census, the device sampler in between - every result against the first
evaluation of the same rows by the plain three-launch path.   python tools/mix_paths.py"""
import sys
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
from lf_testlib import make_inputs, synth
from lumfuncmcmc_amd.capi import LFContext
from lumfuncmcmc_amd.sampler import DeviceEnsembleSampler

rng = np.random.default_rng(7)
for variant, n in (("free", 300000), ("zevol", 300000), ("fixcomp", 100000), ("free", 3000)):
    inp = make_inputs(variant, n, seed=5)
    ctx = LFContext(inp, max_batch=64)
    th = synth.walkers(variant, 1500, seed=6)
    th[::37, 0] = 40.2
    th[5::41, 1] = 9.0
    ctx.set_option("fuse", 0)
    ref = ctx.lnprob_batch(th)
    ctx.set_option("fuse", 1)
    bad = 0
    ds = DeviceEnsembleSampler(ctx, 64, seed=1, capacity=200)
    ds.run_mcmc(synth.walkers(variant, 64, seed=9), 2)
    for it in range(300):
        B = int(rng.choice([1, 7, 8, 9, 64, 130, 600, 1500]))
        lo = int(rng.integers(0, 1500 - B + 1))
        if variant == "free":
            ctx.set_option("fuse", int(rng.integers(0, 2)))
            ctx.set_option("cells", int(rng.integers(0, 2)))
        what = int(rng.integers(0, 6))
        if what == 0:
            a, b = ctx.lnprob_pieces(th[lo:lo + B])
            got = None
        elif what == 1:
            ctx.set_option("count_forms", 1)
            got = ctx.lnprob_batch(th[lo:lo + B])
            ctx.set_option("count_forms", 0)
        elif what == 2:
            ds.run_mcmc(None, 1)
            got = None
        else:
            got = ctx.lnprob_batch(th[lo:lo + B])
        if got is not None:
            r = ref[lo:lo + B]
            fin = np.isfinite(r)
            ok = np.array_equal(fin, np.isfinite(got)) and not np.isnan(got).any() and \
                (not fin.any() or np.max(np.abs(got[fin] / r[fin] - 1)) < 1e-12)
            bad += 0 if ok else 1
    ds.close()
    ctx.close()
    print("%-8s N=%-7d 300 mixed calls: %d mismatches" % (variant, n, bad), flush=True)
    assert bad == 0
