#!/bin/bash
# round 5: does the one-process CLI scale with engines?  BRC_DEVICES=0,0 --brc-gpus 2 (two engines on the one GPU of the box) against one
# engine on the 30-Mbp BAM of the bench's e2e leg and on a site list; wall time, stage accounts.  
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/r05_cli_two_engines.log
python - > $OUT 2>&1 <<'PY'
import os, sys, time, subprocess, tempfile
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np, synthgen
CLI = os.path.abspath("bam_readcount_amd/csrc/bam-readcount")
d = tempfile.mkdtemp(prefix="brc_scal_")
n = 30_000_000
ref, a = synthgen.generate(n, "wgs30x", seed=3)
synthgen.write_bam(os.path.join(d, "syn.bam"), "chrS", n, a)
synthgen.write_fasta(os.path.join(d, "syn.fa"), [("chrS", ref)])
rng = np.random.default_rng(9)
sites = np.sort(rng.integers(1000, n - 1000, 20000))
open(os.path.join(d, "sites"), "w").write("".join("chrS\t%d\t%d\n" % (s, s) for s in sites))
def run(env, args, label):
    best = None; err = ""
    for _ in range(3):
        t0 = time.perf_counter()
        p = subprocess.run([CLI, "-w", "0", "-q", "20", "-b", "13", "-f", "syn.fa"] + args, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, BRC_CLI_TIMING="1", **env))
        t = time.perf_counter() - t0
        assert p.returncode == 0, p.stderr.decode()[-500:]
        if best is None or t < best: best, err = t, p.stderr.decode()
    print("%-44s %.3f s" % (label, best)); print("    " + "\n    ".join(l for l in err.splitlines() if l.startswith(("timing:", "sites:", "startup:"))))
    return best
one = run({}, ["syn.bam", "chrS"], "region chrS (30 Mbp), one engine")
two = run({"BRC_DEVICES": "0,0"}, ["--brc-gpus", "2", "syn.bam", "chrS"], "region chrS, BRC_DEVICES=0,0 --brc-gpus 2")
s1 = run({}, ["-l", "sites", "syn.bam"], "-l 20000 sites, one engine")
s2 = run({"BRC_DEVICES": "0,0"}, ["--brc-gpus", "2", "-l", "sites", "syn.bam"], "-l 20000 sites, BRC_DEVICES=0,0 --brc-gpus 2")
print("two engines / one engine: region %.2fx, sites %.2fx (wall time; > 1 = slower)" % (two / one, s2 / s1))
# same text?
for args in (["syn.bam", "chrS:1-3000000"], ["-l", "sites", "syn.bam"]):
    outs = []
    for env, extra in (({}, []), ({"BRC_DEVICES": "0,0"}, ["--brc-gpus", "2"])):
        p = subprocess.run([CLI, "-w", "0", "-q", "20", "-b", "13", "-f", "syn.fa"] + extra + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=dict(os.environ, **env))
        import hashlib; outs.append(hashlib.md5(p.stdout).hexdigest())
    print("same text with one and two engines:", args[-1], outs[0] == outs[1])
import shutil; shutil.rmtree(d, ignore_errors=True)
PY
cat $OUT
SITES_MBP=${SITES_MBP:-50} bash tools/gpu_r5_e2e.sh 2>&1 | tail -12 | cut -c1-1800
