#!/bin/bash
# k_pileup time for a list of "ENV=val,ENV=val" settings (same box): profiling ablations / occupancy sweeps
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/quick2.log
IFS=';' read -ra SETS <<< "${SETTINGS:-none}"
for st in "${SETS[@]}"; do
  envs=$(echo "$st" | tr ',' ' '); [ "$st" = none ] && envs=""
  env $envs timeout 300 python bench.py --steps 5 --warmup 1 --cpu-sample-mbp 0 --e2e-mbp 0 ${BENCH_ARGS:-} 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernel_ms']; print('$st', 'step', d['ms_per_step'], 'pileup', k['k_pileup'], 'annotate', k['k_annotate'])" | tee -a gpurun_out/quick2.log
done
