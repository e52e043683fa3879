#!/bin/bash
# round 5: the zero-copy feed (brc_push_reads_pinned) in the command line against the copying pushes: wall time, stage accounts, same text
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "adopted" 2>&1 | tail -2
python - 2>&1 <<'PY' | tee gpurun_out/r05_e2e_zero_copy.log
import os, sys, time, subprocess, tempfile, hashlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np, synthgen
CLI = os.path.abspath("bam_readcount_amd/csrc/bam-readcount")
d = tempfile.mkdtemp(prefix="brc_zc_")
n = int(float(os.environ.get("ZC_MBP", "30")) * 1e6)
ref, a = synthgen.generate(n, "wgs30x", seed=3)
synthgen.write_bam(os.path.join(d, "syn.bam"), "chrS", n, a)
synthgen.write_fasta(os.path.join(d, "syn.fa"), [("chrS", ref)])
ends = a["pos"].astype(np.int64) + 150
ev = int((np.minimum(ends, n) - a["pos"].astype(np.int64)).clip(min=0).sum())
def run(env, label):
    best = None; err = ""
    for _ in range(4):
        t0 = time.perf_counter()
        p = subprocess.run([CLI, "-w", "0", "-q", "20", "-b", "13", "-f", "syn.fa", "syn.bam", "chrS"], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, BRC_CLI_TIMING="1", BRC_ENGINE_TIMING="1", **env))
        t = time.perf_counter() - t0
        assert p.returncode == 0, p.stderr.decode()[-500:]
        if best is None or t < best: best, err = t, p.stderr.decode()
    print("%-28s %.3f s  (%.2f G events/s incl. the orderly exit of BRC_ENGINE_TIMING)" % (label, best, ev / best / 1e9)); print("    " + "\n    ".join(l for l in err.splitlines() if l.startswith(("timing:", "startup:", "engine timing"))))
    q = subprocess.run([CLI, "-w", "0", "-q", "20", "-b", "13", "-f", "syn.fa", "syn.bam", "chrS"], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, **env))
    tb = []
    for _ in range(3):
        t0 = time.perf_counter(); subprocess.run([CLI, "-w", "0", "-q", "20", "-b", "13", "-f", "syn.fa", "syn.bam", "chrS"], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, **env)); tb.append(time.perf_counter() - t0)
    print("    plain runs (fast exit): %s -> best %.3f s = %.2f G events/s" % (" ".join("%.3f" % x for x in tb), min(tb), ev / min(tb) / 1e9))
    return min(tb)
c = run({"BRC_ZERO_COPY": "0"}, "copying pushes")
z = run({}, "zero-copy pushes (default)")
print("zero-copy / copying: %.2fx" % (z / c))
outs = []
for env in ({"BRC_ZERO_COPY": "0"}, {}):
    p = subprocess.run([CLI, "-w", "3", "-q", "20", "-b", "13", "-f", "syn.fa", "syn.bam", "chrS:1-9000000"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
    outs.append((hashlib.md5(p.stdout).hexdigest(), hashlib.md5(p.stderr).hexdigest(), len(p.stdout)))
print("same stdout and stderr on chrS:1-9000000 (9 pieces):", outs[0] == outs[1], outs[0][2], "bytes")
import shutil; shutil.rmtree(d, ignore_errors=True)
PY
