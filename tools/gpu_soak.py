#!/usr/bin/env python3
"""GPU soak of the engine-level differential fuzz (tests/test_ref_compiled.py: random_scenario) with seeds the suite does not
run: the product library through all three text routes against the reference-compiled library (text), and its planes /
indel buckets / warning counters against the oracle.  Prints one line per failure and a summary; exit status 1 on any."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("BRC_DEVICE_TEXT_MAX_SHARE", "100")
import numpy as np
from bam_readcount_amd import capi
import parity
from test_ref_compiled import random_scenario, REF_LIB

ap = argparse.ArgumentParser(); ap.add_argument("--first", type=int, default=1000); ap.add_argument("--count", type=int, default=300); ap.add_argument("--big", type=int, default=40)
a = ap.parse_args()
hip = capi.load_product(); oracle = capi.Library(os.path.join(ROOT, "oracle", "libbrc_oracle.so")); ref_lib = capi.Library(REF_LIB)
bad = 0; t0 = time.time(); n_ev = 0
for i in range(a.count + a.big):
    big = i >= a.count; seed = a.first + i
    ref, arrs, regions, kw, clear, style = random_scenario(seed, big=big)
    try:
        want, _ = parity.run_engine(ref_lib if not big else oracle, arrs, regions, ref=ref, clear_queue=clear, **kw)     # (the reference's std::map per position is slow on deep piles)
        for route in ({}, dict(text_only=True), dict(device_text="chrS")):
            got, res = parity.run_engine(hip, arrs, regions, ref=ref, clear_queue=clear, **route, **kw)
            assert got == want, "text differs (route %r)" % (route,)
        _, res = parity.compare_libs(hip, oracle, arrs, regions, ref=ref, clear_queue=clear, check_warn=not any(int(l) < 0 for l in arrs["lib"]) or not kw.get("per_lib"), **kw)
        n_ev += sum(r.n_events for r in res)
    except Exception as ex:                                          # noqa: BLE001
        bad += 1; print("FAIL seed %d big=%s style=%s kw=%r regions=%r clear=%s: %s" % (seed, big, style, kw, regions, clear, str(ex)[:300]), flush=True)
print("soak: %d scenarios (%d big), %d events, %d failures, %.1f s" % (a.count + a.big, a.big, n_ev, bad, time.time() - t0))
sys.exit(1 if bad else 0)
