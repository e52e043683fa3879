"""The N > 1 branch on real RCCL, with the one GPU a test box has: `bench.py --force-dist` under torch.distributed.run
--nproc-per-node 1 runs init_process_group("nccl", device_id=...), the barriers, both all-reduces, the all-gathers (per-rank
clocks, per-rank verdicts), the MIN-reduced validation of every rank's OWN interval against the oracle, and the destroy — so that
an 8-GPU run is not the first time RCCL sees this code; shard.run_sharded / gather_text likewise at world 1 on cuda.
(The rank arithmetic at world 2 / 4 / 8 is covered on CPUs with gloo: tests/test_bench_multirank.py, tests/test_shard_gloo.py.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


SMALL = ["--steps", "3", "--warmup", "1", "--cpu-sample-mbp", "0", "--e2e-mbp", "0", "--abi-mbp", "0", "--other-configs", "0", "--e2e-configs", "0"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [["--mode", "weak", "--contig-mbp", "3"], ["--mode", "strong", "--contig-mbp", "0.5"], ["--mode", "sites", "--contig-mbp", "3", "--sites", "5000"]])
def test_bench_distributed_branch_runs_on_rccl_and_every_rank_validates(mode):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
                          os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist"] + mode + SMALL, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    assert run.returncode == 0, run.stderr.decode()[-3000:]
    line = json.loads([l for l in run.stdout.decode().splitlines() if l.startswith("{")][-1])
    v = line["validated"]
    assert v["distributed_backend"] == "nccl" and v["all_ranks_ok"] is True and v["rank0_error"] is None
    pr = line["per_rank"]
    assert len(pr) == 1 and pr[0]["validated_ok"] is True and pr[0]["validated_events"] > 0
    assert line["value"] > 0 and pr[0]["events"] == line["config"]["events_per_step"]


@pytest.mark.gpu
def test_bench_force_dist_without_a_launcher():
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--mode", "weak", "--contig-mbp", "2"] + SMALL,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert run.returncode == 0, run.stderr.decode()[-3000:]
    line = json.loads([l for l in run.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert line["validated"]["distributed_backend"] == "nccl" and line["validated"]["all_ranks_ok"] is True


WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests")); sys.path.insert(0, os.path.join(sys.argv[1], "tools"))
import numpy as np
import torch
import torch.distributed as dist
from bam_readcount_amd import capi, shard
import synthgen
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=0, world_size=1, device_id=torch.device("cuda", 0))
hip = capi.load_product()
oracle = capi.Library(os.path.join(sys.argv[1], "oracle", "libbrc_oracle.so"))
n = 300_000
ref, arrs = synthgen.generate(n, "tumor200x", seed=9, n_chunks=4)
names = ["lib0", "lib1", "lib2", "lib3"]
regions = [(100, 101), (100, 101)] + shard.split_region(2000, 60_000, 7) + [(50, 60)]
each = []
text, (ev, npos) = shard.run_sharded(hip, arrs, regions, 0, "chrS", ref, dist=dist, per_rank=each, per_lib=True, insertion_centric=True, lib_names=names)
want, (ev1, np1) = shard.run_sharded(oracle, arrs, regions, 0, "chrS", ref, dist=None, per_lib=True, insertion_centric=True, lib_names=names)
assert text == want, "sharded text on RCCL differs from the oracle's"
assert (ev, npos) == (ev1, np1) and each == [ev]
big = shard.gather_text(b"x" * 50_000_001, dist)            # a text larger than any staging granule, through the padded uint8 gather
assert len(big) == 50_000_001 and big[:3] == b"xxx"
dist.barrier(); dist.destroy_process_group()
open(sys.argv[3], "w").write("ok %d %d" % (ev, len(text)))
'''


@pytest.mark.gpu
def test_run_sharded_and_gather_text_on_rccl_world1(tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    script = str(tmp_path / "worker.py"); open(script, "w").write(WORKER)
    out = str(tmp_path / "ok.txt")
    run = subprocess.run([sys.executable, script, ROOT, str(free_port()), out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert run.returncode == 0, run.stderr.decode()[-3000:]
    got = open(out).read().split()
    assert got[0] == "ok" and int(got[1]) > 0 and int(got[2]) > 0


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_run_the_whole_n_rank_flow():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, two ranks), on the ONE GPU of a test box: BRC_BENCH_SHARE_GPU0=1
    puts both engines on GPU 0 and the process group on gloo (RCCL refuses two ranks on one device; it is covered at world 1 above).
    Everything else is the N > 1 run: max-over-ranks clock, summed counters, per-rank table, every rank validating its own interval, and —
    round 6 — rank 0's e2e_sharded legs: configs 4 and 5 through `bam-readcount --brc-ranks 2`, whole output byte-identical to one process."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BRC_BENCH_SHARE_GPU0="1")
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "weak", "--contig-mbp", "2", "--steps", "3", "--warmup", "1", "--cpu-sample-mbp", "0", "--e2e-mbp", "0",
                          "--abi-mbp", "0", "--other-configs", "0", "--e2e-configs", "1", "--e2e-sites-mbp", "1", "--e2e-tumor-mbp", "0.4"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=1500)
    assert run.returncode == 0, run.stderr.decode()[-3000:]
    line = json.loads([l for l in run.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and len(line["per_rank"]) == 2 and line["validated"]["all_ranks_ok"] is True
    assert line["config"]["events_per_step"] == sum(p["events"] for p in line["per_rank"]) and all(p["validated_ok"] for p in line["per_rank"])
    sh = line["e2e_sharded"]
    for leg in ("sites", "tumor"):
        assert sh[leg] is not None and sh[leg]["ranks"] == 2 and sh[leg]["whole_output_byte_identical_to_one_process"] and len(sh[leg]["per_rank"]) == 2, (leg, line.get("e2e_" + leg))
