"""Builds liblfmcmc.so (HIP, gfx950 only) in-tree with hipcc.  No JIT cache, no torch extension:
the shared object sits next to the sources so it travels with the repo snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liblfmcmc.so")
SOURCES = [os.path.join(CSRC, "lfmcmc.hip")]
HEADERS = [os.path.join(CSRC, h) for h in ("lf_kernels.h", "lf_free.h", "lf_pers.h", "lf_math.h", "lf_tables.h", "lf_compress.h", "lf_gridbound.h")] + \
          [os.path.join(os.path.dirname(HERE), "include", "lfmcmc.h")]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS)


# -amdgpu-atomic-optimizer-strategy=None: lf_free's one-lane queue claims must stay plain returning atomics whose
# result is waited for where it is USED (half an item later); the optimizer's wave-aggregated form reads it at once.
# -disable-machine-licm: left on, the pass lifts the materialisation of every fp64 literal of every inlined routine (the
# device library's exp and log of the careful path among them: 14 coefficients) to the top of the kernel, where they
# hold vector registers for the kernel's whole life: lf_free sat at its 128-register cap and spilled, with it off it
# needs 86 and nothing spills.  Measured on the MI355X: lf_free 31.6 -> 29.5 us (10^6 sources, 128 rows), 31.8 -> 28.9
# (10^5), lf_main 24.0 -> 20.8 (10^3); only the z-evolving lf_main pays (82.3 -> 85.0 us).
CXXFLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
            "-Wno-unused-value", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-mllvm", "-disable-machine-licm"]


def build_library(force=False, verbose=True, extra_flags=()):
    """Compile the C-ABI library for gfx950.  Raises on failure."""
    if not force and not is_stale():
        return LIB
    cmd = [hipcc()] + CXXFLAGS + ["-fPIC", "-shared"] + list(extra_flags) + ["-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=CSRC)
    # the per-form instruction counts bench.py prices the kernel with come from the same sources (profiles/isa_mix.py)
    try:
        sys.path.insert(0, os.path.join(os.path.dirname(HERE), "profiles"))
        import isa_mix
        isa_mix.write_json(hipcc=hipcc())
    except Exception as e:                          # measurement aid: never fails the build
        print("build: profiles/isa_counts.json not refreshed: %s" % e, file=sys.stderr)
    finally:
        sys.path.pop(0)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(LIB)
