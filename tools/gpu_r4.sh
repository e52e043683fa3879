#!/bin/bash
# round-4 GPU session pieces: tools/gpu_r4.sh <tag> [suite] [bench] [ab <libs...>]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; tag=$1; shift; out=gpurun_out/r04; mkdir -p $out
for what in "$@"; do
  case $what in
    suite) ( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $out/pytest_gpu_$tag.log 2>&1 ;;
    suitex) ( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 ) > $out/pytest_gpu_$tag.log 2>&1 ;;
    bench) ( time timeout 1500 python bench.py --steps 20 --warmup 5 ) > $out/bench_$tag.log 2> $out/bench_$tag.err ;;
    bench300) ( time timeout 1500 python bench.py ) > $out/bench300_$tag.log 2> $out/bench300_$tag.err ;;
    ab) ( time timeout 1200 python tools/gpu_ab_multi.py --libs $AB_LIBS --reps ${AB_REPS:-3} --shapes ${AB_SHAPES:-tumor,wgs} ) > $out/ab_$tag.log 2>&1 ;;
    sites) ( time timeout 900 python bench.py --mode sites --steps 60 --warmup 5 --e2e-mbp 0 --other-configs 0 ) > $out/sites_$tag.log 2> $out/sites_$tag.err ;;
    e2e) ( python - <<'PY'
import os, sys, time, subprocess, tempfile
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import numpy as np, synthgen
n = 30_000_000; d = tempfile.mkdtemp(prefix="brc_e2e_")
r2, a2 = synthgen.generate(n, "wgs30x", seed=3)
synthgen.write_bam(os.path.join(d, "syn.bam"), "chrS", n, a2)
rows = (n + 59) // 60; pad = np.full(rows * 60, 10, np.uint8); pad[:n] = r2
open(os.path.join(d, "syn.fa"), "wb").write(b">chrS\n" + np.concatenate([pad.reshape(rows, 60), np.full((rows, 1), 10, np.uint8)], axis=1).tobytes())
open(os.path.join(d, "syn.fa.fai"), "w").write("chrS\t%d\t6\t60\t61\n" % n)
cli = os.path.abspath("bam_readcount_amd/csrc/bam-readcount")
print("bam bytes", os.path.getsize(os.path.join(d, "syn.bam")))
for rep in range(3):
    t0 = time.perf_counter()
    p = subprocess.run([cli, "-w", "0", "-q", "20", "-b", "13", "-f", "syn.fa", "syn.bam", "chrS"], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, BRC_CLI_TIMING="1", BRC_ENGINE_TIMING="1"))
    print("rep", rep, round(time.perf_counter() - t0, 3), "s"); print(p.stderr.decode()[-1500:])
PY
      ) > $out/e2e_$tag.log 2>&1 ;;
    quick) ( time timeout 900 python bench.py --steps 20 --warmup 5 --other-configs 0 --e2e-mbp 0 ) > $out/quick_$tag.log 2> $out/quick_$tag.err ;;
  esac
done
tail -c 1500 $out/*_$tag.log
