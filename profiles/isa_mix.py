"""Instruction mix of the per-term loops, from the compiler's assembly (hipcc -S): how many fp64
VALU operations ONE (walker, source) term executes.  This is the count bench.py uses for the
fp64-VALU roofline (FMA = 2 flops, every other fp64 VALU instruction = 1).

    python profiles/isa_mix.py > profiles/r01_isa_mix.txt
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "lumfuncmcmc_amd", "csrc", "lfmcmc.hip")


def assembly():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "lf.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc",
                        "-Wno-unused-value", "-S", "--cuda-device-only", "-o", out, SRC], check=True,
                       stderr=subprocess.DEVNULL)
        return open(out).read()


def kernels(s):
    for m in re.finditer(r"^(_ZN2lf\w+):[^\n]*\n(.*?)s_endpgm", s, re.S | re.M):
        yield m.group(1), m.group(2).split("\n")


def loops(lines):
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    for i, l in enumerate(lines):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            yield labels[m.group(1)], i


def mix(lines):
    c = collections.Counter()
    for l in lines:
        m = re.match(r"^\s+([a-z_0-9]+)", l)
        if m and not l.strip().startswith((";", ".")):
            c[m.group(1)] += 1
    return c


def main():
    s = assembly()
    for name, lines in kernels(s):
        m = re.search(r"lf_(?:srcsum|main)ILi(\d)ELi(\d+)ELi(\d+)", name)
        if not m:
            continue
        variant, st = int(m.group(1)), int(m.group(2))
        for a, b in loops(lines):
            c = mix(lines[a:b + 1])
            fp64 = {k: v for k, v in c.items() if k.startswith("v_") and "f64" in k}
            if variant == 0 and c.get("v_rsq_f64_e32", 0) >= 30 and not c.get("v_div_scale_f64", 0):
                # the walker loop of the grid integral (piece B): the switch over nf = 1..8 is unrolled inline, i.e. the
                # loop body holds 36 field terms (exp, rsqrt, log, exp each) and 8 copies of the per-node tail
                fma = c["v_fma_f64"] + c["v_fmac_f64_e32"]
                other = sum(fp64.values()) - fma
                print("lf_main<variant 0, ST %d>  grid walker loop, all nf = 1..8 cases inline (36 field terms + 8 per-node tails): "
                      "fp64 fma %d, other fp64 %d -> %d executed flops = 36 x %.1f + 8 x 23 (bench.py counts 70 per node-field + 23 per node)"
                      % (st, fma, other, 2 * fma + other, (2 * fma + other - 8 * 23) / 36.0))
                continue
            careful = c.get("v_div_scale_f64", 0) or (c.get("v_cndmask_b32_e64", 0) + c.get("v_cndmask_b32_e32", 0) >= st)
            if b - a < 20 * st or careful:           # skip small loops and the careful (checked) path
                continue
            fma = c["v_fma_f64"] + c["v_fmac_f64_e32"]
            other = sum(fp64.values()) - fma
            ints = sum(v for k, v in c.items() if k.startswith("v_") and "f64" not in k)
            lds = sum(v for k, v in c.items() if k.startswith("ds_read"))
            if variant == 0 and c.get("v_rsq_f64_e32", 0) == 2 * st:
                print("   (next loop: the walker loop of the free variant holds BOTH forms of the term - the general one, "
                      "54 flops / 174 cycles, and term_free_noexp: the rest)")
            print("lf_main<variant %d, ST %d>  loop of %d lines, per item (term or grid node-field):" % (variant, st, b - a))
            print("   fp64 fma %.2f   other fp64 VALU %.2f   32-bit VALU %.2f   LDS reads %.2f" % (fma / st, other / st, ints / st, lds / st))
            print("   executed fp64 flops/term (fma = 2, other = 1): %.1f" % ((2 * fma + other) / st))
            print("   issue cycles/term-wave (fp64 4, rcp/rsq 16, 32-bit ~2.5): %.0f" % (
                (4 * (fma + other - c.get("v_rcp_f64_e32", 0) - c.get("v_rsq_f64_e32", 0)) + 16 * (c.get("v_rcp_f64_e32", 0) + c.get("v_rsq_f64_e32", 0)) + 2.5 * ints) / st))
            print("   mix:", ", ".join("%s %d" % kv for kv in c.most_common(24)))


if __name__ == "__main__":
    main()
