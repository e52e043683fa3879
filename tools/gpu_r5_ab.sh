#!/bin/bash
# round 5: same-process A/B of product builds (ab/libbrc_hip_<name>.so against the tree's library) + the GPU parity tests of the paths a change touches
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
LOG=gpurun_out/${AB_LOG:-r05_ab.log}
timeout 1200 python tools/gpu_ab_multi.py --libs ${AB_LIBS:-ab/libbrc_hip_head.so bam_readcount_amd/csrc/libbrc_hip.so} --shapes ${AB_SHAPES:-wgs,tumor,long} --reps ${AB_REPS:-3} 2>&1 | tee $LOG | grep -v "^$" | cut -c1-220
if [ -n "$AB_TESTS" ]; then timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$AB_TESTS" 2>&1 | tail -3 | tee -a $LOG; fi
