#!/bin/bash
# A/B on ONE box: kernel times of the in-tree libbrc_hip.so vs other builds (LIBS="path1 path2"), interleaved REPS times
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/ab.log
for r in $(seq 1 ${REPS:-2}); do
  for lib in default ${LIBS:-}; do
    if [ "$lib" = default ]; then unset BRC_HIP_LIB; else export BRC_HIP_LIB="$PWD/$lib"; fi
    timeout 300 python bench.py --steps 5 --warmup 1 --cpu-sample-mbp 0 --e2e-mbp 0 ${BENCH_ARGS:-} 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernel_ms']; print('$lib', round(d['value']/1e9,1), 'pileup', k['k_pileup'], 'annotate', k['k_annotate'])" | tee -a gpurun_out/ab.log
  done
done
for g in ${WG_PER_CU:-}; do
  unset BRC_HIP_LIB
  BRC_PILEUP_WG_PER_CU=$g timeout 300 python bench.py --steps 5 --warmup 1 --cpu-sample-mbp 0 --e2e-mbp 0 ${BENCH_ARGS:-} 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernel_ms']; print('wg_per_cu $g', round(d['value']/1e9,1), 'pileup', k['k_pileup'])" | tee -a gpurun_out/ab.log
done
