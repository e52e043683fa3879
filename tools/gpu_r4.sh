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
    quick) ( time timeout 900 python bench.py --steps 20 --warmup 5 --other-configs 0 --e2e-mbp 0 ) > $out/quick_$tag.log 2> $out/quick_$tag.err ;;
  esac
done
tail -c 1500 $out/*_$tag.log
