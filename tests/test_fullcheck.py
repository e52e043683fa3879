"""tools/fullcheck.py (the whole-region validation bench.py runs on the GPU box) on the CPU: the device side played by the lane
simulator, two oracle processes.  The check must pass on equal results and must name the window and the part that differs
otherwise."""
import os
import sys

import numpy as np
import pytest

from bam_readcount_amd import capi
from conftest import ROOT
import parity

sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("config,text", [("wgs30x", True), ("tumor200x", True), ("tumor200x", False)])
def test_whole_region_check_against_the_oracle_pool(sim_lib, oracle_lib, config, text):
    import fullcheck
    import synthgen
    n = 60_000 if config == "wgs30x" else 12_000
    ref, arrs = synthgen.generate(n, config, seed=11, n_chunks=1)
    per_lib = config == "tumor200x"
    names = ["lib%d" % i for i in range(synthgen.CONFIGS[config]["n_libs"])] if per_lib else ()
    opts = dict(min_mapq=20, min_bq=13) if not per_lib else dict(min_mapq=0, min_bq=0, per_lib=True, insertion_centric=True)
    pool = fullcheck.OraclePool(2, ref, arrs, names, opts, capi)
    eng = capi.Engine(sim_lib, lib_names=names, **opts)
    eng.begin_region(0, 0, n, ref); eng.push_reads(arrs); eng.upload(); eng.compute()
    ev, _ = eng.counts()
    out = fullcheck.check_region(pool, eng, parity, fullcheck.windows_of(0, n, 7), want_events=ev, with_text=text)
    assert out["full_contig"] and out["events"] == ev and out["windows"] == 7 and (out["text_byte_exact"] is True if text else ("text_byte_exact" not in out and "not formatted" in out["whole_region_text"]))
    assert (out["text_bytes"] > 0) == text
    eng.close()


def test_whole_region_check_reports_a_difference(sim_lib, oracle_lib):
    import fullcheck
    import synthgen
    n = 30_000
    ref, arrs = synthgen.generate(n, "wgs30x", seed=12, n_chunks=1)
    pool = fullcheck.OraclePool(2, ref, arrs, (), dict(min_mapq=20, min_bq=13), capi)
    eng = capi.Engine(sim_lib, min_mapq=20, min_bq=14)            # (another base-quality threshold: depth and buckets differ somewhere)
    eng.begin_region(0, 0, n, ref); eng.push_reads(arrs); eng.upload(); eng.compute()
    with pytest.raises(AssertionError, match=r"window \[\d+, \d+\): (depth|istat|fstat|text)"):
        fullcheck.check_region(pool, eng, parity, fullcheck.windows_of(0, n, 4))
    eng.close()
