"""world_size-2 CPU test (gloo) of the N>1 path: interval sharding + ordered concatenation equals the single-process
result.  Uses the CPU lane simulator as the engine (test infrastructure), because there is no GPU here."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np
import torch.distributed as dist
from bam_readcount_amd import capi, shard
import synth
rank, world, port, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % port, rank=rank, world_size=world)
lib = capi.Library(os.path.join(sys.argv[1], "tests", "sim", "libbrc_sim.so"))
rng = np.random.default_rng(21)
ref = synth.make_ref(rng, 6000)
arrs = synth.make_batch(22, ref, 900, style="mixed", n_libs=2)
regions = [(100, 101), (100, 101), (900, 2500)] + shard.split_region(2500, 6000, 5) + [(50, 60)]
text, (ev, npos) = shard.run_sharded(lib, arrs, regions, 0, "chrT", ref, dist=dist, per_lib=True, lib_names=["libA", "libB"], min_bq=10)
if rank == 0:
    single, (ev1, np1) = shard.run_sharded(lib, arrs, regions, 0, "chrT", ref, dist=None, per_lib=True, lib_names=["libA", "libB"], min_bq=10)
    assert text == single, "sharded text differs"
    assert (ev, npos) == (ev1, np1), ((ev, npos), (ev1, np1))
    open(out, "w").write("ok %d %d %d" % (ev, npos, len(text)))
dist.destroy_process_group()
'''


def test_two_rank_interval_sharding_matches_single_process(tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "result.txt")
    script = str(tmp_path / "worker.py"); open(script, "w").write(WORKER)
    procs = [subprocess.Popen([sys.executable, script, ROOT, str(r), "2", str(port), out]) for r in range(2)]
    rcs = [p.wait(timeout=300) for p in procs]
    assert rcs == [0, 0]
    got = open(out).read().split()
    assert got[0] == "ok" and int(got[1]) > 0 and int(got[2]) > 0


SKEW_WORKER = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests")); sys.path.insert(0, os.path.join(sys.argv[1], "tools"))
import numpy as np
import torch.distributed as dist
from bam_readcount_amd import capi, shard
import synthgen
rank, world, port, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % port, rank=rank, world_size=world)
lib = capi.Library(os.path.join(sys.argv[1], "tests", "sim", "libbrc_sim.so"))
n = 96_000
ref, arrs = synthgen.generate(n, "wgs30x", seed=31, n_chunks=1)
# uneven depth: nine reads in ten of the first 60 % of the contig are dropped (3x there, 30x behind)
keep = (arrs["pos"] >= int(n * 0.6)) | (np.random.default_rng(1).random(len(arrs["pos"])) < 0.1)
arrs = capi.select_reads(arrs, np.nonzero(keep)[0])
regions = shard.split_region(0, n, 192)                      # the work list: abutting pieces in order (the command line cuts a long region the same way)
cum = shard.bin_events(arrs, n, bin_size=1024)
weights = [shard.interval_events(cum, a, b, bin_size=1024) for a, b in regions]
each = []
text, (ev, npos) = shard.run_sharded(lib, arrs, regions, 0, "chrS", ref, dist=dist, clear_queue=False, weights=weights, per_rank=each, min_mapq=20, min_bq=13)
by_len = [sum(shard.interval_events(cum, a, b, bin_size=1024) for a, b in part) for part in shard.partition(regions, world)]
if rank == 0:
    single, (ev1, np1) = shard.run_sharded(lib, arrs, regions, 0, "chrS", ref, dist=None, clear_queue=False, min_mapq=20, min_bq=13)
    assert text == single, "sharded text differs"
    assert (ev, npos) == (ev1, np1) and sum(each) == ev
    json.dump({"per_rank_events": each, "events": ev, "by_length": by_len}, open(out, "w"))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [4, 8])
def test_event_weighted_partition_balances_uneven_depth(tmp_path, world):
    """slices balanced by estimated event count.  A contig whose first 60 % is ten times shallower than the rest,
    cut into 192 pieces in order: with the window-table weights (shard.bin_events — what BAI chunk sizes give a caller) every
    rank's events stay within 10 % of the mean, the text is the single-process text; the by-length partition of the same list
    is far off (that is what round 3 shipped)."""
    import json
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "result.json")
    script = str(tmp_path / "worker.py"); open(script, "w").write(SKEW_WORKER)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, script, ROOT, str(r), str(world), str(port), out], env=env) for r in range(world)]
    rcs = [p.wait(timeout=600) for p in procs]
    assert rcs == [0] * world
    got = json.load(open(out))
    ev = np.array(got["per_rank_events"], float); mean = ev.sum() / world
    assert len(ev) == world and ev.sum() == got["events"] > 0
    assert np.abs(ev - mean).max() <= 0.10 * mean, got
    bl = np.array(got["by_length"], float)
    assert np.abs(bl - bl.mean()).max() > 0.5 * bl.mean()


def test_partition_is_contiguous_and_complete():
    from bam_readcount_amd import shard
    items = [(0, 10), (10, 1000), (5, 6), (2000, 2100), (7, 8)]
    for w in (1, 2, 3, 8):
        parts = shard.partition(items, w)
        assert len(parts) == w and [x for p in parts for x in p] == items
    assert shard.split_region(0, 10, 3) == [(0, 3), (3, 6), (6, 10)]
    # weights: contiguous, complete, and the heavy item alone on a rank
    parts = shard.partition(items, 3, weights=[1, 1, 100, 1, 1])
    assert [x for p in parts for x in p] == items and [(5, 6)] in parts
    import synthgen
    from bam_readcount_amd import capi
    ref, arrs = synthgen.generate(50_000, "wgs30x", seed=4, n_chunks=1)
    cum = shard.bin_events(arrs, 50_000)
    ends = capi.read_ends(arrs)
    assert cum[-1] == float((ends - arrs["pos"]).sum())
    exact = int((np.minimum(ends, 30_000) - np.minimum(arrs["pos"], 30_000)).clip(0).sum() - (np.minimum(ends, 10_000) - np.minimum(arrs["pos"], 10_000)).clip(0).sum())
    assert abs(shard.interval_events(cum, 10_000, 30_000) - exact) < 0.02 * exact
    cuts = shard.split_region_by_events(0, 50_000, 4, cum)
    assert cuts[0][0] == 0 and cuts[-1][1] == 50_000 and all(a[1] == b[0] for a, b in zip(cuts[:-1], cuts[1:]))
    w = [shard.interval_events(cum, a, b) for a, b in cuts]
    assert max(w) - min(w) < 0.1 * sum(w) / 4
