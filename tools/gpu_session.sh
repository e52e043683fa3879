#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel trace.  Everything lands in gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== nproc $(nproc); $(rocminfo 2>/dev/null | grep -m1 gfx9 || true)" | tee gpurun_out/session.log
( time timeout 900 python -m pytest tests -q -m gpu ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/session.log
tail -5 gpurun_out/pytest_gpu.log | tee -a gpurun_out/session.log
( time timeout 300 python __graft_entry__.py smoke ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/session.log
tail -3 gpurun_out/smoke.log | tee -a gpurun_out/session.log
( time timeout 900 python bench.py --steps ${BENCH_STEPS:-10} --warmup 2 ${BENCH_ARGS:-} ) > gpurun_out/bench.log 2>&1; echo "bench rc=$?" | tee -a gpurun_out/session.log
tail -4 gpurun_out/bench.log | tee -a gpurun_out/session.log
if [ "${DO_PROF:-1}" = "1" ]; then
  rm -rf gpurun_out/prof
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o trace -- python "$OLDPWD/bench.py" --steps 5 --warmup 1 --cpu-sample-mbp 0 ${BENCH_ARGS:-} ) > gpurun_out/rocprof.log 2>&1; echo "rocprof rc=$?" | tee -a gpurun_out/session.log
  find gpurun_out/prof -name "*stats*" | head | tee -a gpurun_out/session.log
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f" | tee -a gpurun_out/session.log
  # keep the merge small: drop the big per-dispatch trace, keep stats
  find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
fi
