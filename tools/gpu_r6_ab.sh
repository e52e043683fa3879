#!/bin/bash
# A/B of two builds of the library on the GPU box: LIBS="ab/libbrc_hip_x.so default" CFGS="wgs30x novaseq ..." (default = the tree's build).
# Each (library, config) is run ROUNDS times, interleaved; prints ms_per_step and the big kernels' durations.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; out=gpurun_out/r06/ab_${TAG:-run}; mkdir -p $out
for r in $(seq 1 ${ROUNDS:-2}); do
  for cfg in ${CFGS:-wgs30x wgs30x_mixed novaseq}; do
    for lib in ${LIBS:-default}; do
      A="--mode weak --config $cfg"; [ $cfg = tumor200x ] && A="--mode strong --contig-mbp 6.25"
      name=$(basename $lib .so)
      ( [ $lib != default ] && export BRC_HIP_LIB=$PWD/$lib; timeout 600 python bench.py --gpus 1 --steps 30 --warmup 5 --cpu-sample-mbp ${SAMPLE:-1} --cpu-ref-mbp 0 --e2e-mbp 0 --abi-mbp 0 --other-configs 0 --e2e-configs 0 --full-check 0 $A ) > $out/${cfg}_${name}_$r.log 2>&1
      grep '^{' $out/${cfg}_${name}_$r.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$cfg', '$name', $r, d['ms_per_step'], d['roofline']['kernel_ms'], d['per_rank'][0].get('hbm_bytes'))" || tail -3 $out/${cfg}_${name}_$r.log
    done
  done
done
