import ctypes, os, sys
order = sys.argv[1]
def maps():
    return sorted(set(x.split()[-1] for x in open('/proc/self/maps') if 'amdhip' in x or 'hsa-runtime' in x))
if order == "torch_first":
    import torch
    print("torch avail", torch.cuda.is_available(), torch.cuda.device_count())
    x = torch.zeros(4, device="cuda"); print("torch tensor ok", x.sum().item())
lib = ctypes.CDLL('/root/repo/lumfuncmcmc_amd/liblfmcmc.so')
print(maps())
hip = ctypes.CDLL('libamdhip64.so.7')
n = ctypes.c_int(-1)
rc = hip.hipGetDeviceCount(ctypes.byref(n))
hip.hipGetErrorString.restype = ctypes.c_char_p
print("hipGetDeviceCount rc", rc, hip.hipGetErrorString(rc), "n", n.value)
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
from lumfuncmcmc_amd import capi, synth
from lf_testlib import make_inputs
import lf_oracle as O, numpy as np
inp = make_inputs("fixcomp", 1000, seed=1)
th = synth.walkers("fixcomp", 8, seed=2)
try:
    ctx = capi.LFContext(inp)
    print("ctx ok", np.max(np.abs(ctx.lnprob_batch(th) / O.lnprob_batch(inp, th) - 1)))
except Exception as e:
    print("ctx failed:", e)
if order == "lib_first":
    import torch
    print(maps())
    print("torch avail", torch.cuda.is_available())
    try:
        x = torch.zeros(4, device="cuda"); print("torch tensor ok", x.sum().item())
        print(ctx.lnprob_torch(torch.from_numpy(th).cuda()).cpu().numpy()[:2])
    except Exception as e:
        print("torch failed:", e)
