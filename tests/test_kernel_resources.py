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


def test_big_geometry_of_lf_main_has_no_scratch(remarks):
    for name, r in _find(remarks, "_ZN2lf7lf_mainILi0ELi8ELi16ELi16ELb0EEE").items():
        assert r["ScratchSize"] == 0 and r["VGPRs"] <= 128, (name, r)
