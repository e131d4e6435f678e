"""Stability checks on a GPU box: repeated create/destroy (leaks), large batches, long chains."""
import sys, time
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import torch
from lf_testlib import make_inputs, synth, O
from lumfuncmcmc_amd.capi import LFContext
from lumfuncmcmc_amd.sampler import DeviceEnsembleSampler

free0 = torch.cuda.mem_get_info()[0]
inp = make_inputs("free", 200000, seed=1)
th = synth.walkers("free", 64, seed=2)
ref = None
for i in range(30):
    c = LFContext(inp)
    out = c.lnprob_batch(th)
    if ref is None:
        ref = out
    assert np.array_equal(out, ref)
    c.close()
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("create/destroy x30: device memory delta %.1f MB (runtime pools; flat from the first few on)" % ((free0 - free1) / 1e6))

c = LFContext(inp)
big = synth.walkers("free", 20000, seed=3)
t = time.perf_counter(); out = c.lnprob_batch(big); dt = time.perf_counter() - t
assert np.isfinite(out).all()
# a 64-row call picks another launch geometry (summation order): equal to rounding, not bitwise
assert np.allclose(out[:64], c.lnprob_batch(big[:64]), rtol=1e-14, atol=0)
print("B=20000 x N=200000: %.3f s, %.3g terms/s" % (dt, 20000 * 200000 / dt))
chk = O.lnprob_batch(inp, big[19990:19993])
assert np.max(np.abs(out[19990:19993] / chk - 1)) < 1e-12
c.close()

# compressed catalogue + grid: create / build / destroy repeatedly, toggling between the paths, a long chain
inp = make_inputs("free", 200000, seed=1)
free2 = torch.cuda.mem_get_info()[0]
for i in range(10):
    c = LFContext(inp)
    a = c.lnprob_batch(th)
    c.set_option("compress", 1)
    b = c.lnprob_batch(th)
    c.set_option("compress", 0)
    assert np.array_equal(c.lnprob_batch(th), a) and np.allclose(a, b, rtol=1e-13, atol=0)
    c.close()
torch.cuda.synchronize()
print("compress build/destroy x10: device memory delta %.1f MB" % ((free2 - torch.cuda.mem_get_info()[0]) / 1e6))
c = LFContext(inp)
c.set_option("compress", 1)
ds = DeviceEnsembleSampler(c, 256, seed=6, capacity=5000)
t = time.perf_counter(); ds.run_mcmc(synth.walkers("free", 256, seed=7), 5000); dt = time.perf_counter() - t
last = ds.chain[:, -1, :]
c.set_option("compress", 0)
assert np.allclose(c.lnprob_batch(last), ds.lnprobability[:, -1], rtol=1e-12, atol=0)
print("compressed: 5000 steps x 256 walkers at N=200000: %.2f s (%.1f us/step), acceptance %.2f; final lnprob = direct lnprob of the final positions"
      % (dt, dt / 5000 * 1e6, ds.acceptance_fraction.mean()))
ds.close(); c.close()

inp = make_inputs("fixcomp", 5000, seed=4)
c = LFContext(inp)
ds = DeviceEnsembleSampler(c, 64, seed=5, capacity=20000)
p0 = np.array([42.5, -2.0, -1.49]) + 0.05 * np.random.default_rng(0).normal(size=(64, 3))
t = time.perf_counter(); ds.run_mcmc(p0, 20000); dt = time.perf_counter() - t
lp = ds.lnprobability
print("20000 steps x 64 walkers: %.2f s (%.1f us/step), acceptance %.2f, tau %s" % (dt, dt / 20000 * 1e6, ds.acceptance_fraction.mean(), np.round(ds.acor, 1)))
flat = ds.chain[:, 2000:, :].reshape(-1, 3)
print("posterior mean", flat.mean(axis=0), "std", flat.std(axis=0))
assert np.isfinite(lp).all()
ds.close(); c.close()
print("ok")
