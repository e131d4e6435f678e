import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
from lf_testlib import O, make_inputs, synth
from lumfuncmcmc_amd.capi import LFContext
for n in (400000, 60000):
    inp = make_inputs("free", n, seed=3)
    th = synth.walkers("free", 24, seed=5)
    th[3, 0] = 40.2
    ref = O.lnprob_batch(inp, th)
    fin = np.isfinite(ref)
    ctx = LFContext(inp)
    ctx.set_option("persistent", 2)
    ctx.set_option("count_forms", 1)
    a1, b1 = ctx.lnprob_pieces(th)
    g1 = ctx.lnprob_batch(th)
    fc = ctx.form_counts()
    ctx.set_option("count_forms", 0)
    ctx.set_option("cells", 0)
    a0, b0 = ctx.lnprob_pieces(th)
    g0 = ctx.lnprob_batch(th)
    print(n, ctx.last_launch()["kernel"], fc)
    print("   cells vs oracle %.2e  sources vs oracle %.2e  pieceA cells vs sources %.2e  inf pattern %s" % (
        np.max(np.abs(g1[fin] - ref[fin]) / np.abs(ref[fin])), np.max(np.abs(g0[fin] - ref[fin]) / np.abs(ref[fin])),
        np.nanmax(np.abs(a1[fin] - a0[fin]) / np.abs(a0[fin])), np.array_equal(np.isinf(g1), np.isinf(ref))))
    ctx.close()
