"""Instruction mix of the per-term loops, from the compiler's assembly (hipcc -S): how many fp64 VALU operations ONE
(walker, source) term or ONE (walker, node, field) term of the grid integral executes, per FORM of the term.

The kernels bracket each form's code with assembly comments (`; LF_BEGIN <name> items=N` ... `; LF_END <name>`,
lf_kernels.h / lf_free.h): what the compiler laid out between them - for the table-driven form that includes the
per-lane table lookup - divided by the N items it handles (sources per lane, fields per node, or 1 for a rolled loop)
are the per-item figures.  bench.py reads them from the JSON this writes,
so `roofline.achieved` (executed flops: FMA = 2, any other fp64 VALU instruction = 1) follows the binary.

    python profiles/isa_mix.py                         -> profiles/isa_counts.json (also done by lumfuncmcmc_amd.build)
    python profiles/isa_mix.py --report > profiles/r02_isa_mix.txt
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "lumfuncmcmc_amd", "csrc", "lfmcmc.hip")
OUT = os.path.join(ROOT, "profiles", "isa_counts.json")
NF_REPORTED = 5                   # the grid forms are instantiated for nf = 1..8: report the configLF case
# issue cycles one wave spends (measured rates, profiles/r01_ubench.txt): fp64 VALU 4, v_rcp/v_rsq/v_sqrt_f64 16, 32-bit VALU ~2.5
CYC_FP64, CYC_TRANS64, CYC_VALU32 = 4.0, 16.0, 2.5


def assembly(hipcc="/opt/rocm/bin/hipcc"):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "lf.s")
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from lumfuncmcmc_amd.build import CXXFLAGS          # the library's own flags
        subprocess.run([hipcc] + CXXFLAGS + ["-S", "--cuda-device-only", "-o", out, SRC], check=True,
                       stderr=subprocess.DEVNULL)
        return open(out).read()


def kernels(s):
    for m in re.finditer(r"^(_ZN2lf\w+):[^\n]*\n(.*?)s_endpgm", s, re.S | re.M):
        yield m.group(1), m.group(2).split("\n")


def regions(lines):
    """(form, items, instruction lines) of every `; LF_BEGIN form items=N` ... `; LF_END form` bracket, in layout order
    (a bracket may span several basic blocks: everything the compiler laid out between the two markers)."""
    cur = None
    for l in lines:
        m = re.search(r"; LF_BEGIN (\w+) items=(\d+)", l)
        if m:
            cur = (m.group(1), int(m.group(2)), [])
            continue
        m = re.search(r"; LF_END (\w+)", l)
        if m and cur and cur[0] == m.group(1):
            yield cur
            cur = None
            continue
        if cur is not None:
            cur[2].append(l)


def mix(lines):
    c = collections.Counter()
    for l in lines:
        m = re.match(r"^\s+([a-z_0-9]+)", l)
        if m and not l.strip().startswith((";", ".")):
            c[m.group(1)] += 1
    return c


def classify(c):
    fp64 = {k: v for k, v in c.items() if k.startswith("v_") and "f64" in k}
    trans = sum(v for k, v in fp64.items() if re.match(r"v_(rcp|rsq|sqrt)_f64", k))
    fma = sum(v for k, v in fp64.items() if re.match(r"v_fma(c)?_f64", k))
    cvt = sum(v for k, v in fp64.items() if k.startswith("v_cvt") or k.startswith("v_cmp") or k.startswith("v_ldexp") or k.startswith("v_mov"))
    other = sum(fp64.values()) - fma - trans
    valu32 = sum(v for k, v in c.items() if k.startswith("v_") and "f64" not in k)
    lds = sum(v for k, v in c.items() if k.startswith("ds_read"))
    return {"fma": fma, "fp64_other": other, "trans64": trans, "valu32": valu32, "lds_reads": lds,
            "salu_smem": sum(v for k, v in c.items() if k.startswith("s_"))}


def analyse(s):
    """{kernel instantiation: {form: counts}} for every marked block."""
    out = {}
    for name, lines in kernels(s):
        m = re.search(r"lf_mainILi(\d)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)", name)
        mf = re.search(r"lf_freeILi(\d+)ELb0ELb1E", name)     # the persistent FREE kernel, product (fused) instantiation
        mp = re.search(r"lf_persILi(\d)ELb1E", name)           # the persistent kernel of the other two variants, one-launch form
        if mf:
            st = int(mf.group(1))
            key = "lf_free<%d>" % st
        elif mp:
            key = "lf_pers<%d>" % int(mp.group(1))
        elif m:
            variant, st, tw, twb, cmp_ = (int(x) for x in m.groups())
            key = "lf_main<%d,%d,%d,%d,%s>" % (variant, st, tw, twb, "true" if cmp_ else "false")
        else:
            continue
        for form, items, body in regions(lines):
            if form.startswith("node_") and items != NF_REPORTED:
                continue
            cl = classify(mix(body))
            cl["items"] = items
            cl["flops_per_item"] = (2.0 * cl["fma"] + cl["fp64_other"] + cl["trans64"]) / items
            cl["cycles_per_item"] = (CYC_FP64 * (cl["fma"] + cl["fp64_other"]) + CYC_TRANS64 * cl["trans64"] +
                                     CYC_VALU32 * cl["valu32"]) / items
            cl["instructions_in_region"] = sum(mix(body).values())
            if form not in out.setdefault(key, {}):        # (a form inlined twice: the first copy)
                out[key][form] = cl
    return out


def write_json(path=OUT, hipcc="/opt/rocm/bin/hipcc"):
    a = analyse(assembly(hipcc))
    doc = {"source": "profiles/isa_mix.py over lumfuncmcmc_amd/csrc/lfmcmc.hip (hipcc -O3 --offload-arch=gfx950 -S)",
           "units": "flops: FMA = 2, other fp64 VALU = 1; cycles: issue cycles of one wave (fp64 4, rcp/rsq 16, 32-bit VALU 2.5)",
           "nf_of_grid_forms": NF_REPORTED, "kernels": a}
    with open(path, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
        f.write("\n")
    return doc


def report(doc):
    for k in sorted(doc["kernels"]):
        print(k)
        for form, c in sorted(doc["kernels"][k].items()):
            print("   %-20s per item: %5.1f executed flops, %5.1f issue cycles | region of %d instructions for %d items: "
                  "fp64 fma %d, other fp64 %d, rcp/rsq %d, 32-bit VALU %d, LDS reads %d, scalar %d" % (
                      form, c["flops_per_item"], c["cycles_per_item"], c["instructions_in_region"], c["items"], c["fma"],
                      c["fp64_other"], c["trans64"], c["valu32"], c["lds_reads"], c["salu_smem"]))


if __name__ == "__main__":
    d = write_json()
    if "--report" in sys.argv:
        report(d)
    else:
        print(OUT)
