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


def test_partition_is_contiguous_and_complete():
    from bam_readcount_amd import shard
    items = [(0, 10), (10, 1000), (5, 6), (2000, 2100), (7, 8)]
    for w in (1, 2, 3, 8):
        parts = shard.partition(items, w)
        assert len(parts) == w and [x for p in parts for x in p] == items
    assert shard.split_region(0, 10, 3) == [(0, 3), (3, 6), (6, 10)]
