"""CPU-side checks of the boundary: the product library loads and exports every symbol include/brc.h declares;
without a GPU it refuses to create an engine (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from bam_readcount_amd import capi
from conftest import ROOT


def declared_symbols():
    h = open(os.path.join(ROOT, "include", "brc.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(brc_[a-z_0-9]+)\s*\(", h)))


def test_header_symbols_listed_in_binding():
    assert set(declared_symbols()) == set(capi.EXPORTS)


def test_product_library_builds_and_exports_abi():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "bam_readcount_amd", "csrc")])
    lib = capi.load_product()
    for s in declared_symbols():
        assert hasattr(lib.lib, s), s
    assert lib.kind() == "hip-gfx950"
    names = lib.kernel_names()
    assert "k_pileup" in names and "k_annotate" in names


def test_product_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu tests")
    lib = capi.load_product()
    with pytest.raises(capi.BrcError) as ei:
        capi.Engine(lib)
    assert "-2" in str(ei.value) or "no HIP device" in str(ei.value)


def test_oracle_is_not_linked_into_product():
    out = subprocess.check_output(["nm", "-D", "--defined-only", capi.PRODUCT_LIB]).decode()
    assert "oracle" not in out.lower()
    ldd = subprocess.check_output(["ldd", capi.PRODUCT_LIB]).decode()
    assert "oracle" not in ldd and "brc_sim" not in ldd
