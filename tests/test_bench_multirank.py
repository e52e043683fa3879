"""bench.py's rank arithmetic without GPUs: two ranks under torch.distributed.run with the gloo backend, the C-ABI served by
the CPU lane simulator (`--dry-run-lib`, test infrastructure: the line it prints says "dry_run" and carries no throughput).
In every mode the totals rank 0 reduces must equal the sum of what each rank owns when its share is run on its own
(`--as-rank r --as-world 2`, no launcher), and the partition must be the documented one: weak — every rank its own contig;
strong — the contig cut into world_size intervals; sites — the -l list cut into contiguous slices whose sizes add up.
What is left for the real multi-GPU run is RCCL itself."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

SIM = os.path.join(ROOT, "tests", "sim", "libbrc_sim.so")
COMMON = ["--steps", "1", "--warmup", "0", "--cpu-sample-mbp", "0", "--e2e-mbp", "0", "--other-configs", "0", "--dry-run-lib", SIM]
MODES = {"weak": ["--mode", "weak", "--contig-mbp", "0.06"],
         "strong": ["--mode", "strong", "--contig-mbp", "0.02"],
         "sites": ["--mode", "sites", "--contig-mbp", "0.06", "--sites", "301"]}


def line_of(out):
    return json.loads([l for l in out.decode().splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("mode,world", [("strong", 4), ("sites", 8)])
def test_more_ranks_describe_themselves(mode, world):
    """world 4 and 8 (the driver's scaling run goes to 8): the line carries every rank's own clock, share and memory, the
    shares add up to the totals, and the partition is the documented one."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, OMP_NUM_THREADS="1")
    small = {"strong": ["--mode", "strong", "--contig-mbp", "0.016"], "sites": ["--mode", "sites", "--contig-mbp", "0.03", "--sites", "203"]}[mode]
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + small + COMMON, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    assert run.returncode == 0, run.stderr.decode()[-2000:]
    total = line_of(run.stdout)
    pr = total["per_rank"]
    assert total["n_gpus"] == world and [x["rank"] for x in pr] == list(range(world))
    assert sum(x["events"] for x in pr) == total["events_per_step"] > 0 and sum(x["positions"] for x in pr) == total["positions_per_step"]
    if mode == "sites":
        assert total["positions_per_step"] == 203 and sorted(x["positions"] for x in pr) == sorted([203 // world + (1 if r < 203 % world else 0) for r in range(world)])
    else:
        for x in pr:
            assert 3_500 < x["positions"] <= 4_000           # 16 kb cut into 4 intervals


@pytest.mark.parametrize("mode", sorted(MODES))
def test_two_ranks_reduce_what_each_rank_owns(mode):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, OMP_NUM_THREADS="1")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2"] + MODES[mode] + COMMON, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    assert two.returncode == 0, two.stderr.decode()[-2000:]
    total = line_of(two.stdout)
    assert total["dry_run"].startswith("rank arithmetic only") and total["value"] is None and total["n_gpus"] == 2
    own = []
    for r in (0, 1):
        one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--as-rank", str(r), "--as-world", "2"] + MODES[mode] + COMMON,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
        assert one.returncode == 0, one.stderr.decode()[-2000:]
        own.append(line_of(one.stdout))
    assert total["events_per_step"] == own[0]["own_events"] + own[1]["own_events"] > 0
    assert total["positions_per_step"] == own[0]["own_positions"] + own[1]["own_positions"] > 0
    assert total["own_events"] == own[0]["own_events"]                       # rank 0 of the pair did rank 0's share
    if mode == "sites":
        assert total["positions_per_step"] == 301 and {own[0]["own_positions"], own[1]["own_positions"]} == {150, 151}
    if mode == "strong":
        # fixed total work: each rank an interval of half the contig (0.02 Mbp -> 10 kb each), about 200x deep
        for o in own:
            assert 9_000 < o["own_positions"] <= 10_000 and 150 * 9_000 < o["own_events"] < 250 * 10_000
    if mode == "weak":
        for o in own:
            assert 59_000 < o["own_positions"] <= 60_000
        assert own[0]["own_events"] != own[1]["own_events"]                  # different contigs (seed per rank)
