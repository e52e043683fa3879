"""A few scenarios of tools/fuzz/extreme.py — one of every kind: reads of kilobases, a pile thousands deep, many libraries, thresholds
beyond their ranges, reads of a few bases, table lengths around 255 / 512, dense indels, spliced alignments — with qualities and NM / SM
at the edges of their types, through the dense planes, both text routes and the other entry points (multi-batch push, brc_compute_n,
brc_fetch_window, brc_region_windows), on the simulator (CPU suite) and on the GPU.  The tool itself runs thousands of them; this keeps
it running in every suite.  (Its finds: brc_core.h choose_pack, k_pileup2 BRC_STAGE — DESIGN.md §5.)"""
import os
import sys

import pytest

import parity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "fuzz"))

SEEDS = {"deep": 5001, "dense_indel": 5002, "libs": 5003, "long": 5004, "spliced": 5005, "thresholds": 5014, "tiny": 5018, "mixed_len": 5027}


@pytest.mark.parametrize("kind", sorted(SEEDS))
def test_extreme_scenario(dev_lib, oracle_lib, kind):
    import extreme
    seed = SEEDS[kind]
    got_kind, style, ref, arrs, regions, kw, clear = extreme.scenario(seed)
    assert got_kind == kind                      # (the generator's stream: a changed draw order would silently test something else)
    arrs = extreme.mutate_fields(seed, arrs)
    check_warn = not (kw.get("per_lib") and any(int(l) < 0 for l in arrs["lib"]))
    want, res = parity.compare_libs(dev_lib, oracle_lib, arrs, regions, ref=ref, clear_queue=clear, check_warn=check_warn, **kw)
    assert sum(r.n_events for r in res) > 1000
    for route in (dict(text_only=True), dict(device_text="chrS")):
        got, _ = parity.run_engine(dev_lib, arrs, regions, ref=ref, clear_queue=clear, **route, **kw)
        assert got == want, route
    extreme.api_routes(dev_lib, oracle_lib, seed, ref, arrs, regions, kw)
