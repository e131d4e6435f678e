"""Where does a source workgroup of lf_main spend its time?  Builds a DIAGNOSTIC copy of the library (-DLF_STAMPS:
s_memtime at the start of the workgroup, after its prologue, after walkers 1 and 8, after the walker loop and at the
end) into tools/, runs one launch of the bench workload and prints the distribution.  Never part of the product.
    python tools/stamps.py [--nsrc N] [--rows B]"""
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
    ap.add_argument("--opts", default="")
    a = ap.parse_args()
    lib = os.path.join(ROOT, "tools", "liblfmcmc_stamps.so")
    src = os.path.join(ROOT, "lumfuncmcmc_amd", "csrc", "lfmcmc.hip")
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(os.path.join(ROOT, "lumfuncmcmc_amd", "csrc", f))
                                                              for f in os.listdir(os.path.join(ROOT, "lumfuncmcmc_amd", "csrc"))):
        from lumfuncmcmc_amd import build
        subprocess.run([build.hipcc()] + build.CXXFLAGS + ["-fPIC", "-shared", "-DLF_STAMPS", "-o", lib, src], check=True)
    from lumfuncmcmc_amd import capi
    capi.LIB_PATH = lib
    import torch
    import bench
    from lumfuncmcmc_amd import synth
    model = bench.build_model("free", a.nsrc, 2 * a.rows, 0)
    ctx = model.context()
    for kv in filter(None, a.opts.split(",")):
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    L = ctx._lib
    L.lf_debug_stamps.restype = ctypes.c_int
    L.lf_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    th = torch.from_numpy(synth.walkers("free", a.rows, seed=1)).cuda()
    for _ in range(10):
        ctx.lnprob_torch(th)
    torch.cuda.synchronize()
    nb = ctx.last_launch()["workgroups"]
    assert L.lf_debug_stamps(ctx._h, None, nb + (nb + 7) // 8) == 0      # 8 slots per workgroup + one more table of nb values
    for _ in range(3):
        ctx.lnprob_torch(th)
    torch.cuda.synchronize()
    nrow = nb + (nb + 7) // 8
    out = np.zeros((nrow, 8), dtype=np.uint64)
    assert L.lf_debug_stamps(ctx._h, out.ctypes.data_as(ctypes.c_void_p), nrow) == 0
    t_red = out.reshape(-1)[nb * 8: nb * 8 + nb].astype(np.int64)
    out = out[:nb]
    t_red = t_red[out[:, 0] > 0]
    s = out[out[:, 0] > 0].astype(np.int64)
    print("workgroups stamped: %d of %d in the launch" % (len(s), nb))
    life = s[:, 1] - s[:, 0]
    items = s[:, 2]

    def q(x):
        return "median %8.0f  p10 %8.0f  p90 %8.0f" % (np.median(x), np.percentile(x, 10), np.percentile(x, 90))
    print("shader cycles per workgroup (thread 0): life", q(life))
    print("items per workgroup                        ", q(items), " total", items.sum())
    print("cycles per item                            ", q(life / np.maximum(items, 1)))
    pm = lambda x: q(x / np.maximum(life, 1) * 1000)
    print("per mille of the workgroup's life (wave 0): prologue up to the first item   ", pm(s[:, 3]))
    print("   catalogue items: barrier D + loads + LDS transposition + barriers A, B    ", pm(s[:, 4]))
    print("                    walker loops (passes 1 and 2)                            ", pm(s[:, 7]))
    print("                    barrier C + reduction + stores                           ", pm(t_red))
    print("   the rest (node chunks, tile set-up, end)                                  ", pm(life - s[:, 3] - s[:, 4] - s[:, 7] - t_red))
    # who is slow?  by XCD (workgroup index mod 8 under round-robin placement) and by the group of 8 the workgroup is in
    ids = np.nonzero(out[:, 0] > 0)[0]
    for name, key in (("XCD", ids & 7), ("group of 8", ids >> 3)):
        ks = sorted(set(key.tolist()))[:64]
        print("median life by %s: %s" % (name, " ".join("%d:%.0f" % (k, np.median(life[key == k])) for k in ks)))
        print("median END (us after the first start) by %s: %s" % (name, " ".join("%d:%.1f" % (k, np.median(s[key == k, 6] - s[:, 5].min()) / 100.0) for k in ks)))
    print("start spread %.1f us; first start to last end %.1f us" % ((s[:, 5].max() - s[:, 5].min()) / 100.0, (s[:, 6].max() - s[:, 5].min()) / 100.0))
    rt = s[:, 6]
    print("spread of the workgroups' end times: %.1f us" % ((rt.max() - rt.min()) / 100.0))


if __name__ == "__main__":
    main()
