"""The device accumulate replaces fp32 divisions by a reciprocal multiply + two FMAs (brc_core.h div_rcp) and by
quotient tables; both must be bit-identical to `/`.  Exhaustive for read lengths <= 3000, sampled beyond."""
import os
import subprocess

from conftest import ROOT


def test_div_rcp_and_tables_are_bit_exact(tmp_path):
    exe = str(tmp_path / "exact_division")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", os.path.join(ROOT, "tests", "exact_division.cpp"), "-o", exe])
    out = subprocess.check_output([exe]).decode().split()
    assert int(out[0]) > 40_000_000 and int(out[1]) == 0, out
