"""Time line of the ONE launch of a plain evaluation (lf_free in its fused form, every walker on the cells): a DIAGNOSTIC
build of the library (-DLF_STAMPS) stamps, per workgroup (wave 0's view): start, tile prepared (tables in LDS, walker
records made and read back), cells done, grid done, end.  Never part of the product.
    python tools/stamps_fused.py [--nsrc N] [--rows B] [--opts k=v,...]"""
import argparse
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nsrc", type=int, default=1000000)
    ap.add_argument("--rows", type=int, default=128)
    ap.add_argument("--variant", default="free")
    ap.add_argument("--opts", default="")
    ap.add_argument("--sampler", action="store_true", help="the time line of a half-step of the device sampler instead of a plain evaluation")
    a = ap.parse_args()
    lib = os.path.join(ROOT, "tools", "liblfmcmc_stamps.so")
    src = os.path.join(ROOT, "lumfuncmcmc_amd", "csrc", "lfmcmc.hip")
    csrc = os.path.join(ROOT, "lumfuncmcmc_amd", "csrc")
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc)):
        from lumfuncmcmc_amd import build
        subprocess.run([build.hipcc()] + build.CXXFLAGS + ["-fPIC", "-shared", "-DLF_STAMPS", "-o", lib, src], check=True)
    from lumfuncmcmc_amd import capi
    capi.LIB_PATH = lib
    import torch
    import bench
    from lumfuncmcmc_amd import synth
    model = bench.build_model(a.variant, a.nsrc, 2 * a.rows, 0)
    ctx = model.context()
    for kv in filter(None, a.opts.split(",")):
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    L = ctx._lib
    L.lf_debug_stamps.restype = ctypes.c_int
    L.lf_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    th = torch.from_numpy(synth.walkers(a.variant, a.rows, seed=1)).cuda()
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:              # settle the clocks
        for _ in range(16):
            ctx.lnprob_torch(th)
        torch.cuda.synchronize()
    nb = ctx.last_launch()["workgroups"]
    nrow = nb + (nb + 7) // 8
    ds = None
    if a.sampler:
        from lumfuncmcmc_amd.sampler import DeviceEnsembleSampler
        ds = DeviceEnsembleSampler(ctx, 2 * a.rows, seed=1, capacity=400)
        ds.run_mcmc(synth.walkers(a.variant, 2 * a.rows, seed=1), 100)
    assert L.lf_debug_stamps(ctx._h, None, nrow) == 0
    if ds is not None:
        ds.enqueue(None, 3)
    else:
        for _ in range(3):
            ctx.lnprob_torch(th)
    torch.cuda.synchronize()
    out = np.zeros((nrow, 8), dtype=np.uint64)
    assert L.lf_debug_stamps(ctx._h, out.ctypes.data_as(ctypes.c_void_p), nrow) == 0
    second = out.reshape(-1)[nb * 8: nb * 8 + nb][out[:nb, 0] > 0]
    s = out[:nb][out[:nb, 0] > 0].astype(np.int64)
    t_p0, t_tab = (second & np.uint64(0xffffffff)).astype(np.int64), (second >> np.uint64(32)).astype(np.int64)
    print("launch:", ctx.last_launch())
    print("workgroups stamped: %d of %d" % (len(s), nb))
    life = s[:, 1] - s[:, 0]
    rt0 = s[:, 5].min()

    def q(x):
        return "median %7.0f  p10 %7.0f  p90 %7.0f  max %7.0f" % (np.median(x), np.percentile(x, 10), np.percentile(x, 90), x.max())
    print("shader cycles (wave 0 of each workgroup), from the wave's first instruction (lf_free) / after its arguments (lf_pers):")
    if ctx.last_launch()["kernel"].startswith("lf_free"):
        pk = out[:nb][out[:nb, 0] > 0][:, 2]
        for i, name in enumerate(("kernel arguments in", "entered (after the barrier)", "theta + field constants in", "Q = 10^(42 - L*) made",
                                  "records + keys written")):
            print("      preparation: %-26s" % name, q(((pk >> np.uint64(12 * i)) & np.uint64(0xfff)).astype(np.int64) * 16))
        print("    the preparing wave back         ", q(t_p0))
        print("    a table-loading wave done       ", q(t_tab))
    print("  tables + preparation + records  ", q(s[:, 3]))
    print("  cells (lf_pers: after the prologue)", q(s[:, 4] - s[:, 3]))
    print("  grid                            ", q(s[:, 7] - s[:, 4]))
    print("  count + final sums + exit       ", q(life - s[:, 7]))
    print("  life                            ", q(life))
    print("real time (100 MHz counter), us after the first workgroup's start:")
    print("  starts  ", q((s[:, 5] - rt0) / 100.0))
    print("  ends    ", q((s[:, 6] - rt0) / 100.0))
    ids = np.nonzero(out[:nb, 0] > 0)[0]
    print("median prologue by XCD:", " ".join("%d:%.0f" % (x, np.median(s[(ids & 7) == x, 3])) for x in range(8)))
    slow = s[:, 3] > np.percentile(s[:, 3], 90)
    print("workgroups with the slowest 10%% of prologues: ids %s" % ids[slow][:40].tolist())
    print("  first start to last end %.2f us; cycles per us of life: %.0f" % ((s[:, 6].max() - rt0) / 100.0,
                                                                        np.median(life / np.maximum((s[:, 6] - s[:, 5]) / 100.0, 0.01))))


if __name__ == "__main__":
    main()
