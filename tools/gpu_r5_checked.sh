#!/bin/bash
# round 5: the new GPU tests (checked build, RCCL at world 1, e2e configs) + the extreme fuzz on the bounds-checked library
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_checked_build.py tests/test_dist_gpu.py tests/test_e2e_configs.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r05_pytest_new_gpu.log
timeout ${FUZZ_TIMEOUT:-1500} python tools/fuzz/extreme.py --lib bam_readcount_amd/csrc/libbrc_hip_checked.so --first ${FUZZ_FIRST:-6000} --count ${FUZZ_COUNT:-600} > gpurun_out/r05_extreme_fuzz_checked.log 2>&1; echo "fuzz rc $?"; tail -5 gpurun_out/r05_extreme_fuzz_checked.log
