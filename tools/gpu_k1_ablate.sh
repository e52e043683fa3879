#!/bin/bash
# K1 phase ablations (experiment build with -DBRC_EXP_KNOBS): kernel times of config 3 per BRC_ANN_VARIANT
#   0 whole · 1 no event-byte stores · 2 no piece stores · 3 no per-base pass · 4 no phase C · 5 phase A only
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out/r04
export BRC_HIP_LIB=ab/libbrc_hip_knobs.so
for v in 0 1 2 3 4 5; do
  BRC_ANN_VARIANT=$v timeout 300 python bench.py --steps 20 --warmup 3 --cpu-sample-mbp 0 --e2e-mbp 0 --abi-mbp 0 --other-configs 0 --full-check 0 ${BENCH_ARGS:-} 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ann_variant $v', d['ms_per_step'], d['roofline']['kernel_ms']['k_annotate'])"
done | tee gpurun_out/r04/k1_ablations.log
