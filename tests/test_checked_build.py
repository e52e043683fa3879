"""The bounds-checked instantiation of the device pipeline (brc_core.h: BRC_CHECKED; `make -C bam_readcount_amd/csrc checked` ->
libbrc_hip_checked.so; a fuzzing build, never shipped): every data-dependent device address of K1 and k_pileup2 is compared with the
extent its buffer was allocated for; a violation fails the call with {kernel, site, buffer, address, tile / read, piece}.

CPU: it builds, exports the C-ABI, names itself, and the product's machine code does not contain the checks.  GPU: one scenario of every
extreme kind and every fuzz family runs through it with results equal to the oracle's and no fault; the checker itself is proven live
by shrinking an extent (BRC_CHECKED_SHRINK) — the same scenario then fails with a fault record that names the buffer."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

import parity
from conftest import ROOT

CSRC = os.path.join(ROOT, "bam_readcount_amd", "csrc")
CHECKED = os.path.join(CSRC, "libbrc_hip_checked.so")
sys.path.insert(0, os.path.join(ROOT, "tools", "fuzz"))


def build_checked():
    subprocess.check_call(["make", "-s", "-C", CSRC, "checked"])
    return CHECKED


def test_checked_library_builds_and_is_not_the_product():
    from bam_readcount_amd import capi
    lib = ctypes.CDLL(build_checked())
    for sym in ("brc_create", "brc_push_reads", "brc_upload", "brc_compute", "brc_fetch_window", "brc_region_windows", "brc_engine_kind"):
        assert hasattr(lib, sym)
    lib.brc_engine_kind.restype = ctypes.c_char_p
    assert lib.brc_engine_kind() == b"hip-gfx950-checked"
    assert capi.load_product().kind() == "hip-gfx950"
    # the product carries neither the checker's messages nor its knob
    blob = open(os.path.join(CSRC, "libbrc_hip.so"), "rb").read()
    assert b"BRC_CHECKED" not in blob and b"out-of-bounds device access" not in blob
    assert b"BRC_CHECKED_SHRINK" in open(CHECKED, "rb").read()


@pytest.fixture(scope="module")
def checked_lib():
    from bam_readcount_amd import capi
    return capi.Library(build_checked())


EXTREME = {"deep": 5001, "dense_indel": 5002, "libs": 5003, "long": 5004, "spliced": 5005, "thresholds": 5014, "tiny": 5018, "mixed_len": 5027}


@pytest.mark.gpu
@pytest.mark.parametrize("kind", sorted(EXTREME))
def test_checked_build_extreme_scenarios_no_fault(checked_lib, oracle_lib, kind):
    import extreme
    seed = EXTREME[kind]
    got_kind, style, ref, arrs, regions, kw, clear = extreme.scenario(seed)
    assert got_kind == kind
    arrs = extreme.mutate_fields(seed, arrs)
    check_warn = not (kw.get("per_lib") and any(int(l) < 0 for l in arrs["lib"]))
    want, res = parity.compare_libs(checked_lib, oracle_lib, arrs, regions, ref=ref, clear_queue=clear, check_warn=check_warn, **kw)      # (a fault raises BrcError)
    for route in (dict(text_only=True), dict(device_text="chrS")):
        got, _ = parity.run_engine(checked_lib, arrs, regions, ref=ref, clear_queue=clear, **route, **kw)
        assert got == want, route
    extreme.api_routes(checked_lib, oracle_lib, seed, ref, arrs, regions, kw)


@pytest.mark.gpu
def test_checked_build_synthetic_configs_no_fault(checked_lib, oracle_lib):
    import synthgen
    for config, kw in (("wgs30x", dict(min_mapq=20, min_bq=13)), ("wgs30x_mixed", dict(min_mapq=20, min_bq=13)),
                       ("tumor200x", dict(per_lib=True, insertion_centric=True, lib_names=["lib0", "lib1", "lib2", "lib3"]))):
        n = 300_000 if config != "tumor200x" else 60_000
        ref, arrs = synthgen.generate(n, config, seed=77, n_chunks=4)
        parity.compare_libs(checked_lib, oracle_lib, arrs, [(0, n)], ref=ref, **kw)


@pytest.mark.gpu
def test_the_checker_fires_when_an_extent_is_too_short(tmp_path):
    """the same region twice in sub-processes of their own (the knob is read per pass; a faulted engine is not reused): as allocated —
    clean; with 4 kB taken off the end of the event-byte stream (buffer 4) — the windows of the last rows are outside: BRC_E_HIP with
    a fault record that names kernel, buffer and tile"""
    script = tmp_path / "run.py"
    script.write_text('''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tools")
import synthgen
from bam_readcount_amd import capi
lib = capi.Library(%r)
ref, arrs = synthgen.generate(100_000, "wgs30x", seed=5, n_chunks=2)
e = capi.Engine(lib, min_mapq=20, min_bq=13)
e.begin_region(0, 0, 100_000, ref); e.push_reads(arrs)
try:
    e.end_region()
    print("CLEAN")
except capi.BrcError as ex:
    print("FAULT", ex)
''' % (ROOT, ROOT, build_checked()))
    ok = subprocess.run([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert ok.returncode == 0 and b"CLEAN" in ok.stdout, ok.stderr.decode()[-2000:]
    bad = subprocess.run([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=dict(os.environ, BRC_CHECKED_SHRINK="4:4096"))
    out = bad.stdout.decode() + bad.stderr.decode()
    assert "FAULT" in out and "BRC_CHECKED" in out and "event bytes" in out and ("k_pileup2" in out or "k_annotate_groups" in out), out[-2000:]
