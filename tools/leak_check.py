import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import torch, numpy as np
from lf_testlib import make_inputs, synth
from lumfuncmcmc_amd.capi import LFContext
inp = make_inputs("free", 200000, seed=1)
th = synth.walkers("free", 64, seed=2)
torch.cuda.init(); torch.zeros(1).cuda()
f0 = torch.cuda.mem_get_info()[0]
for rnd in range(4):
    for i in range(20):
        c = LFContext(inp); c.lnprob_batch(th); c.set_option("compress", 1); c.lnprob_batch(th); c.close()
    torch.cuda.synchronize()
    print("after %d cycles: delta %.1f MB" % ((rnd + 1) * 20, (f0 - torch.cuda.mem_get_info()[0]) / 1e6), flush=True)
