import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
from lf_testlib import make_inputs, synth
from test_gpu_sampler import host_replay
from lumfuncmcmc_amd.capi import LFContext
from lumfuncmcmc_amd.sampler import DeviceEnsembleSampler
inp = make_inputs("fixcomp", 3000, seed=31)
ctx = LFContext(inp)
W=16
pos = synth.walkers("fixcomp", W, seed=32); pos[1] = 99.0
nsteps, seed = 25, 0x1234567890ABCDEF
ds = DeviceEnsembleSampler(ctx, W, seed=seed, capacity=nsteps)
ds.run_mcmc(pos, nsteps)
chain, lnps, nacc = host_replay(ctx, pos, nsteps, seed)
d = np.abs(ds.chain - chain)
print("max diff", d.max(), "first differing (walker, step, dim):", np.argwhere(d > 0)[:5])
print("lnp diff", np.nanmax(np.abs(ds.lnprobability - lnps)))
k,t,i = np.argwhere(d>0)[0]
print(ds.chain[k,t], chain[k,t], ds.chain[k,max(t-1,0)], chain[k,max(t-1,0)])
