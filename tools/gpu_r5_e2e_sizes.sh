#!/bin/bash
# round 5: config 3 end to end through the drop-in command line at two input sizes (the fixed start-up — process start, HIP initialisation,
# engine creation: ~0.28 s — is a third of a 30-Mbp run), and rocprofv3 kernel stats of the driver's command at HEAD
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
python - 2>&1 <<'PY' | tee gpurun_out/r05_e2e_sizes.log
import os, sys, time, subprocess, tempfile
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np, synthgen
CLI = os.path.abspath("bam_readcount_amd/csrc/bam-readcount")
for mbp in (30, 120):
    d = tempfile.mkdtemp(prefix="brc_e2e_")
    n = int(mbp * 1e6)
    t0 = time.perf_counter()
    ref, a = synthgen.generate(n, "wgs30x", seed=3)
    synthgen.write_bam(os.path.join(d, "syn.bam"), "chrS", n, a)
    synthgen.write_fasta(os.path.join(d, "syn.fa"), [("chrS", ref)])
    ends = a["pos"].astype(np.int64) + 150
    ev = int((np.minimum(ends, n) - a["pos"].astype(np.int64)).clip(min=0).sum())
    print("== %d Mbp at 30x: %d reads, %d events, BAM %.2f GB (generated in %.0f s)" % (mbp, len(a["pos"]), ev, os.path.getsize(os.path.join(d, "syn.bam")) / 1e9, time.perf_counter() - t0))
    del a, ref
    cmd = [CLI, "-w", "0", "-q", "20", "-b", "13", "-f", "syn.fa", "syn.bam", "chrS"]
    tb = []
    for _ in range(4):
        t0 = time.perf_counter(); p = subprocess.run(cmd, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE); tb.append(time.perf_counter() - t0)
        assert p.returncode == 0, p.stderr.decode()[-400:]
    print("   bam-readcount -w0 -q20 -b13 -f syn.fa syn.bam chrS > /dev/null: %s s -> best %.3f s = %.2f G events/s" % (" ".join("%.3f" % x for x in tb), min(tb), ev / min(tb) / 1e9))
    p = subprocess.run(cmd, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, BRC_CLI_TIMING="1", BRC_ENGINE_TIMING="1"))
    print("   " + "\n   ".join(l for l in p.stderr.decode().splitlines() if l.startswith(("timing:", "startup:", "engine timing"))))
    subprocess.run(["rm", "-rf", d])
PY
R=$PWD
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_wgs -o trace -- python $R/bench.py --steps 20 --warmup 5 --e2e-mbp 0 --abi-mbp 0 --cpu-ref-mbp 0 --cpu-sample-mbp 0 --other-configs 0 --e2e-configs 0 --full-check 0 ) > gpurun_out/rocprof_wgs.log 2>&1
f=$(find /tmp/prof_wgs -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r05_rocprofv3_kernel_stats_wgs30x_head.csv && head -8 "$f" | cut -c1-60,300-420
