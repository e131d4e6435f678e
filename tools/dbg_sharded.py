import os, sys, socket
import numpy as np
root = '/root/repo'
for p in (root, root + '/oracle', root + '/tests'):
    sys.path.insert(0, p)

def worker(rank, world, port, q):
    import torch, torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from lf_testlib import make_inputs, synth
    from lumfuncmcmc_amd.capi import LFContext
    from lumfuncmcmc_amd.sampler import DeviceEnsembleSampler
    from lumfuncmcmc_amd.dist import slice_bounds
    import ctypes as ct
    inp = make_inputs("zevol", 3000, seed=61)
    ctx = LFContext(inp)
    W, nsteps, seed = 22, 8, 1234
    pos = synth.walkers("zevol", W, seed=62)
    s = DeviceEnsembleSampler(ctx, W, seed=seed, capacity=nsteps)
    lib = ctx._lib
    ctx._check(lib.lf_sampler_start(s._h, s._p(np.ascontiguousarray(pos)), None)); s._started = True
    half = W // 2
    bounds, per = slice_bounds(half, world); lo, hi = bounds[rank]
    dev = torch.device("cuda", 0)
    buf = torch.full((per * world,), float("-inf"), dtype=torch.float64, device=dev)
    sep = torch.full((per,), float("-inf"), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    log = []
    for it in range(nsteps):
        for h in (0, 1):
            ctx._check(lib.lf_sampler_half_eval(s._h, h, lo, hi, ct.c_void_p(sep.data_ptr() - lo * 8), ct.c_void_p(stream)))
            log.append(("sep", it, h, sep.cpu().numpy().copy()))
            dist.all_gather_into_tensor(buf, sep)
            full = torch.cat([buf[r * per:r * per + (b - a)] for r, (a, b) in enumerate(bounds)])
            log.append(("full", it, h, full.cpu().numpy().copy()))
            ctx._check(lib.lf_sampler_half_accept(s._h, h, ct.c_void_p(full.data_ptr()), ct.c_void_p(stream)))
    s.sync()
    q.put((rank, log, s.chain))
    dist.barrier()

if __name__ == "__main__":
    import torch.multiprocessing as mp
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    c = mp.get_context("spawn"); q = c.Queue()
    ps = [c.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=200) for _ in range(2)], key=lambda r: r[0])
    [p.join() for p in ps]
    l0, l1 = res[0][1], res[1][1]
    for a, b in zip(l0, l1):
        if a[0] == "full":
            print(a[0], a[1], a[2], "equal" if np.array_equal(a[3], b[3]) else "DIFF")
            if not np.array_equal(a[3], b[3]): print(a[3]); print(b[3])
    print("chain equal", np.array_equal(res[0][2], res[1][2]))
