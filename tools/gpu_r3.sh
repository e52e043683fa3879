#!/bin/bash
# round-3 GPU session: the whole -m gpu suite (no -x), same-process A/B of the builds under ab/ against the product build,
# a kernel trace of the default bench shape.  Everything lands under gpurun_out/r03/<tag>/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
tag="${1:-run}"; out="gpurun_out/r03/$tag"; mkdir -p "$out"; export BRC_CRASH_DIR="$PWD/$out"
if [ -z "${SKIP_TESTS:-}" ]; then ( time timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 ) > "$out/pytest_gpu.log" 2>&1; tail -5 "$out/pytest_gpu.log"; fi
if [ -z "${SKIP_AB:-}" ]; then
  timeout 600 python tools/gpu_ab_multi.py --libs bam_readcount_amd/csrc/libbrc_hip.so ${AB_LIBS:-$(ls ab/*.so 2>/dev/null)} --shapes "${SHAPES:-wgs,tumor}" --reps "${REPS:-2}" --steps "${STEPS:-6}" 2>&1 | grep -v amdgpu.ids | tee "$out/ab.log"
fi
if [ -n "${TRACE:-}" ]; then
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o trace -- python "$OLDPWD/bench.py" --steps 5 --warmup 1 --cpu-sample-mbp 0 --e2e-mbp 0 ${BENCH_ARGS:-} > "$OLDPWD/$out/trace_bench.log" 2>&1 )
  find /tmp/prof_$tag -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
  cut -c1-60,200- "$out/kernel_stats.csv" 2>/dev/null | head -30
fi
