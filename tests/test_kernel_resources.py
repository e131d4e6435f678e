"""What the compiler made of the dominant kernels (hipcc -Rpass-analysis=kernel-resource-usage, no GPU needed): the
persistent FREE kernel is sized for four waves per SIMD (<= 128 VGPRs) and must not touch scratch - a spilled value in
its item loop costs a `s_waitcnt vmcnt(0)` on every reload, and prologue spills were 30 MB of
stores per launch before they were hunted down."""
import os
import re
import subprocess
import tempfile

import pytest

from lumfuncmcmc_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "lumfuncmcmc_amd", "csrc", "lfmcmc.hip")


@pytest.fixture(scope="module")
def remarks():
    hipcc = build.hipcc()
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run([hipcc] + build.CXXFLAGS + ["--cuda-device-only", "-c", "-o", os.path.join(d, "lf.o"), SRC,
                            "-Rpass-analysis=kernel-resource-usage"], stderr=subprocess.PIPE, stdout=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = {}
    name = None
    for line in r.stderr.decode().splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            out[name] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and name:
            out[name][m.group(1).strip()] = int(m.group(2))
    return out


def _find(remarks, prefix):
    hits = {k: v for k, v in remarks.items() if k.startswith(prefix)}
    assert hits, prefix
    return hits


@pytest.mark.parametrize("st", [2, 4, 8])
def test_persistent_free_kernel_fits_four_waves_per_simd_without_scratch(remarks, st):
    for census, fused in ((0, 0), (1, 0), (0, 1)):      # the three-launch form, its census twin, the one-launch form
        for name, r in _find(remarks, "_ZN2lf7lf_freeILi%dELb%dELb%dEEE" % (st, census, fused)).items():
            assert r["VGPRs"] <= 128, (name, r)
            if not census:                         # (the census instantiation is a measurement aid)
                assert r["VGPRs Spill"] == 0, (name, r)
                # The one-launch form holds more scalars (lf_prepare's and lf_finalize's arguments ride along): the compiler
                # parks some in VGPR lanes, and for two of the three instantiations it also reserves 36 bytes of private
                # segment that no instruction of the kernel touches (its assembly has no scratch_ / buffer_ access).
                assert r["ScratchSize"] <= (36 if fused else 0), (name, r)
            assert r["LDS Size"] <= 80 * 1024, (name, r)      # two workgroups per CU (160 KB)


@pytest.mark.parametrize("variant", [1, 2])
def test_persistent_kernel_of_the_other_variants_fits_two_workgroups_per_cu(remarks, variant):
    for prefix in ("_ZN2lf7lf_persILi%dELb0EEE" % variant, "_ZN2lf7lf_persILi%dELb1EEE" % variant, "_ZN2lf12lf_pers_stepILi%dEEE" % variant):
        for name, r in _find(remarks, prefix).items():
            # (the sampler's half-step keeps the accept step's arguments alive across the whole kernel: a few values are parked
            # in scratch in its preamble and fetched back in its epilogue - never inside the loops)
            assert r["VGPRs"] <= 128 and r["ScratchSize"] <= (192 if "_step" in prefix else 0), (name, r)
            assert r["LDS Size"] <= 80 * 1024, (name, r)


def test_big_geometry_of_lf_main_has_no_scratch(remarks):
    for name, r in _find(remarks, "_ZN2lf7lf_mainILi0ELi8ELi16ELi16ELb0EEE").items():
        assert r["ScratchSize"] == 0 and r["VGPRs"] <= 128, (name, r)


# ---------------------------------------------------------------------------------------------- the one-launch form's ISA
@pytest.fixture(scope="module")
def asm():
    hipcc = build.hipcc()
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "lf.s")
        r = subprocess.run([hipcc] + build.CXXFLAGS + ["--cuda-device-only", "-S", "-o", out, SRC], stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        text = open(out).read()
    kern = {}
    for m in re.finditer(r"^(_ZN2lf\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        kern[m.group(1)] = [l.strip() for l in m.group(2).split("\n")]
    return kern


def _instr(lines):
    """instruction lines only (no labels, comments, directives, asm-block markers)"""
    return [l for l in lines if l and not l.startswith((";", ".", "//")) and not l.endswith(":")]


FUSED = [("_ZN2lf7lf_freeILi%dELb0ELb1EEE" % st) for st in (2, 4, 8)] + ["_ZN2lf7lf_persILi1ELb1EEE", "_ZN2lf7lf_persILi2ELb1EEE"] + \
        [("_ZN2lf12lf_free_stepILi%dEEE" % st) for st in (2, 4, 8)] + ["_ZN2lf12lf_pers_stepILi1EEE", "_ZN2lf12lf_pers_stepILi2EEE"]


@pytest.mark.parametrize("prefix", FUSED)
def test_one_launch_form_hands_its_partial_sums_over_through_memory(asm, prefix):
    """The hand-over between the workgroups of a tile (lf_free.h / lf_pers.h, FUSED) rests on four facts
    of the generated code, pinned here so that a compiler change cannot silently remove one:
      1. every partial sum is stored with the agent-scope cache policy (sc1: written through this XCD's L2);
      2. each wave waits for its stores' acknowledgements (s_waitcnt vmcnt(0)) directly in front of the workgroup barrier
         that precedes the count;
      3. the count is ONE returning global atomic add of the constant 1 by one lane (not a wave-aggregated add, whose
         result the atomic optimizer would read where it is issued) - and the only atomic behind that barrier;
      4. the finishing workgroup reads the partial sums with the same cache policy (sc1: past its own L2's stale lines)."""
    names = [k for k in asm if k.startswith(prefix)]
    assert len(names) == 1, names
    ins = _instr(asm[names[0]])
    stores = [l for l in ins if l.startswith("global_store_dwordx2")]
    sc1_stores = [l for l in stores if l.endswith(" sc1")]
    # (the kernel's only 8-byte stores to memory are the partial sums, written through, lnprob itself and - the sampler's
    # half-step - the accept step's few: position, chain row, lnprob, counter)
    # (there are two finishers - the counter's and the polling one, lf_free.h: PART_EMPTY - and so two copies of those)
    two = 2
    assert len(sc1_stores) >= 2 and len(stores) - len(sc1_stores) <= two * (8 if "_step" in prefix else 1), stores
    assert not any(l.startswith(("buffer_wbl2", "buffer_inv")) for l in ins), "a cache-wide write-back / invalidate crept in (154 us per evaluation)"
    # the count: wait - barrier - one-lane returning atomic - barrier
    at = [i for i, l in enumerate(ins) if l.startswith("global_atomic_add") and l.endswith(" sc0") and "offset" not in l]
    assert len(at) == 1, [ins[i] for i in at]            # (the queue claims of lf_free's source path carry an offset)
    i = at[0]
    before = ins[max(0, i - 40):i]
    bar = max(j for j, l in enumerate(before) if l == "s_barrier")
    w = max((j for j, l in enumerate(before[:bar]) if l == "s_waitcnt vmcnt(0)"), default=-1)
    # (scalar bookkeeping may sit between the wait and the barrier; no memory instruction may)
    assert w >= 0 and bar - w <= 10 and not any(l.startswith(("global_", "flat_", "buffer_", "scratch_")) for l in before[w:bar]), before[max(0, bar - 12):bar + 1]
    assert any(re.match(r"v_mov_b32_e32 v\d+, 1$", l) for l in before[bar:]), before[bar:]
    assert not any(l.startswith(("s_bcnt1", "v_readfirstlane_b32")) and "exec" in l for l in before[bar:]), before[bar:]
    after = ins[i + 1:i + 12]
    assert "s_barrier" in after and after[0].startswith(("v_mov", "s_waitcnt")), after
    # the finishing workgroup's loads of the partial sums
    tail = ins[i:]
    loads = [l for l in tail if l.startswith("global_load_dwordx2")]
    plain = [l for l in loads if not l.endswith(" sc1")]
    # (the half-step's accept reads the walker's current lnprob and position: data of earlier launches, plain loads)
    # (... once per finisher where the compiler has laid the polling one out behind the count)
    assert len(loads) - len(plain) >= 2 and len(plain) <= (6 if "_step" in prefix else 0), loads
    # 5. the polling finisher (tiles without source work; lf_free.h: PART_EMPTY) reads the slots with the same cache policy, in a
    #    loop that ends: the bound of PART_POLLS = 2^19 polls is compared against somewhere in the kernel
    allsc1 = [l for l in ins if l.startswith("global_load_dwordx2") and l.endswith(" sc1")]
    assert len(allsc1) >= 4, allsc1
    assert any(l.startswith("s_cmp") and l.endswith(", 0x80000") for l in ins), "the polling loop has lost its bound"
