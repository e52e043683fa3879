#!/bin/bash
# quick A/B: parity tests + kernel times for a list of BRC_PILEUP_VARIANT values / BRC_ANNOTATE modes
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
if [ -z "${SKIP_TESTS:-}" ]; then ( timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 ) | tee gpurun_out/quick.log; else : > gpurun_out/quick.log; fi
for v in ${VARIANTS:-0}; do
  BRC_PILEUP_VARIANT=$v timeout 300 python bench.py --steps 5 --warmup 1 --cpu-sample-mbp 0 --e2e-mbp 0 ${BENCH_ARGS:-} 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('variant $v', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'])" | tee -a gpurun_out/quick.log
done
for m in ${ANNOTATE_MODES:-}; do
  BRC_ANNOTATE=$m timeout 300 python bench.py --steps 5 --warmup 1 --cpu-sample-mbp 0 --e2e-mbp 0 ${BENCH_ARGS:-} 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('annotate $m', d['value'], d['roofline']['kernel_ms'])" | tee -a gpurun_out/quick.log
done
for g in ${LDS_PADS:-}; do
  BRC_PILEUP_LDS_PAD=$g timeout 300 python bench.py --steps 5 --warmup 1 --cpu-sample-mbp 0 --e2e-mbp 0 ${BENCH_ARGS:-} 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('lds_pad $g', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms']['k_pileup'])" | tee -a gpurun_out/quick.log
done
