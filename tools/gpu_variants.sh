#!/bin/bash
# timing-only ablations of K1 (BRC_ANN_VARIANT) / k_pileup2 (BRC_PILEUP_VARIANT): kernel times of the default bench shape.
# The product library does not contain these knobs: build an experiment library first (tools/build_variant.sh knobs) and
# name it with BRC_HIP_LIB=ab/libbrc_hip_knobs.so (bench.py's validation fails on a variant > 0, as it should: --cpu-sample-mbp 0).
: "${BRC_HIP_LIB:?build ab/libbrc_hip_knobs.so with tools/build_variant.sh knobs and export BRC_HIP_LIB}"
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out/r03
for v in ${ANN:-0 1 2 3 4 5}; do
  BRC_ANN_VARIANT=$v timeout 300 python bench.py --steps 6 --warmup 1 --cpu-sample-mbp 0 --e2e-mbp 0 ${BENCH_ARGS:-} 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ann_variant $v', d['ms_per_step'], d['roofline']['kernel_ms'])"
done | tee gpurun_out/r03/variants_${1:-x}.log
