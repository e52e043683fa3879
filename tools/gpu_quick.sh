#!/bin/bash
# quick A/B: parity tests + kernel times for a list of BRC_PILEUP_VARIANT values
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -3 ) | tee gpurun_out/quick.log
for v in ${VARIANTS:-0}; do
  BRC_PILEUP_VARIANT=$v timeout 300 python bench.py --steps 5 --warmup 1 --cpu-sample-mbp 0 ${BENCH_ARGS:-} 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('variant $v', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'])" | tee -a gpurun_out/quick.log
done
