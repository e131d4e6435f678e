"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol the header
declares, and its ctypes descriptor matches the C struct.  No compute calls (no GPU here)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lfmcmc.h")


@pytest.fixture(scope="module")
def lib():
    from lumfuncmcmc_amd import build, capi
    build.build_library(verbose=False)
    return capi.load()


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lf_[a-z_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    from lumfuncmcmc_amd import capi
    names = header_functions()
    assert names == sorted(capi.EXPORTS)
    for n in names:
        assert hasattr(lib, n), n
    assert lib.lf_abi_version() == capi.LF_ABI_VERSION


def test_descriptor_layout_matches_the_header(tmp_path):
    """sizeof / offsetof of lf_desc as gcc sees the header == the ctypes mirror."""
    from lumfuncmcmc_amd import capi
    fields = [f[0] for f in capi.LfDesc._fields_]
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "lfmcmc.h"', 'int main(void){',
            'printf("%zu\\n", sizeof(lf_desc));']
    prog += ['printf("%%zu\\n", offsetof(lf_desc, %s));' % f for f in fields]
    prog += ['return 0;}']
    c = tmp_path / "lay.c"
    c.write_text("\n".join(prog))
    exe = tmp_path / "lay"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)], check=True)
    vals = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert vals[0] == ctypes.sizeof(capi.LfDesc)
    for f, off in zip(fields, vals[1:]):
        assert getattr(capi.LfDesc, f).offset == off, f


def test_create_without_a_gpu_fails_loudly(lib):
    """No silent CPU fallback: with no device visible lf_create returns NULL and says why."""
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from lumfuncmcmc_amd import capi
    from lf_testlib import make_inputs
    with pytest.raises(capi.LFError) as e:
        capi.LFContext(make_inputs("fixcomp", 10, S=8))
    assert "device" in str(e.value).lower() or "hip" in str(e.value).lower()
    assert lib.lf_lnprob_batch(None, None, 1, None) == -1
    assert lib.lf_ndim(None) == -1


def test_missing_library_is_an_error(monkeypatch):
    from lumfuncmcmc_amd import capi
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", "/nonexistent/liblfmcmc.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        capi.load()
