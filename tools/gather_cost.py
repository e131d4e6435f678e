"""Host cost of torch.distributed's in-place all-gather per call, one-rank RCCL group on one GPU (what the walker-sharded loop
pays per block before any wire is involved): the public call against the process group's own method.
   MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 python tools/gather_cost.py"""
import os, time
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
full = torch.zeros(128, dtype=torch.float64, device=dev)
mine = full[:128]
pg = dist.group.WORLD
x = torch.zeros(1 << 20, device=dev)
def t(fn, n=2000):
    for _ in range(200): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    h = time.perf_counter() - t0
    torch.cuda.synchronize(); return h / n * 1e6, (time.perf_counter() - t0) / n * 1e6
print("dist.all_gather_into_tensor      : host %.1f us per call, with device %.1f" % t(lambda: dist.all_gather_into_tensor(full, mine)))
print("pg._allgather_base(...).wait()    : host %.1f us per call, with device %.1f" % t(lambda: pg._allgather_base(full, mine).wait()))
try:
    from torch.distributed import _functional_collectives as fc
    print("functional all_gather_tensor      : host %.1f us per call, with device %.1f" % t(lambda: fc.all_gather_tensor(mine, 0, pg)))
except Exception as e:
    print("functional: ", e)
dist.destroy_process_group()
