#!/usr/bin/env python3
"""GPU soak of the engine-level differential fuzz (tests/test_ref_compiled.py: random_scenario) with seeds the suite does not
run: the product library through all three text routes against the reference-compiled library (text), and its planes /
indel buckets / warning counters against the oracle.  Prints one line per failure and a summary; exit status 1 on any."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
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
        # round 4: random windows of the first non-empty region through brc_fetch_window (against slices of the whole result, after
        # several back-to-back passes) and through brc_region_windows (every announced window prints what it prints without the hint)
        rr = np.random.default_rng(seed)
        big_regions = [r for r in regions if r[1] - r[0] >= 4]
        if big_regions:
            b0, e0 = big_regions[0]
            ends = capi.read_ends(arrs)
            sub = capi.select_reads(arrs, capi.fetch_overlapping(arrs, ends, b0 - 1, e0))
            eng = capi.Engine(hip, **kw)
            eng.begin_region(0, b0, e0, ref); eng.push_reads(sub); eng.upload(); eng.compute_n(int(rr.integers(1, 5)))
            whole = eng.fetch_result(); whole_text = eng.format_region("chrS"); eng.clear_indel_queue()
            cuts = sorted(set([b0, e0] + [int(x) for x in rr.integers(b0, e0 + 1, 4)]))
            joined = b""
            for wi, (wa, wb) in enumerate(zip(cuts[:-1], cuts[1:])):
                hip.lib.brc_set_option(eng.h, 6, 1 if wi else 0)          # BRC_OPT_CONTINUES_PREVIOUS: the lead position was the window before's last
                w = eng.fetch_window(wa, wb); joined += eng.format_region("chrS")
                for x, y in zip(parity.slice_result(w, wa - 1, wb), parity.slice_result(whole, wa - 1, wb)):
                    assert (x == y) if isinstance(x, list) else np.array_equal(x, y), "fetch_window [%d,%d) differs from the whole result" % (wa, wb)
            assert joined == whole_text, "windows formatted in order differ from the region's text"
            hip.lib.brc_set_option(eng.h, 6, 0); eng.close()
            wins = sorted((int(x), int(x) + int(rr.integers(1, 70))) for x in rr.integers(b0, max(e0 - 1, b0 + 1), 5)); wins = [(x, min(y, e0)) for x, y in wins if x < e0]
            e1 = capi.Engine(hip, text_only=True, **kw); e2 = capi.Engine(hip, text_only=True, **kw)
            e1.begin_region(0, b0, e0, ref); e1.push_reads(sub); e1.end_region()
            e2.begin_region(0, b0, e0, ref); e2.push_reads(sub); e2.region_windows(np.array([w[0] for w in wins], np.int32), np.array([w[1] for w in wins], np.int32)); e2.end_region()
            for wa, wb in wins:
                assert e1.format_window("chrS", wa, wb, 0) == e2.format_window("chrS", wa, wb, 0), "announced window [%d,%d) differs" % (wa, wb)
            e1.close(); e2.close()
    except Exception as ex:                                          # noqa: BLE001
        bad += 1; print("FAIL seed %d big=%s style=%s kw=%r regions=%r clear=%s: %s" % (seed, big, style, kw, regions, clear, str(ex)[:300]), flush=True)
print("soak: %d scenarios (%d big), %d events, %d failures, %.1f s" % (a.count + a.big, a.big, n_ev, bad, time.time() - t0))
sys.exit(1 if bad else 0)
