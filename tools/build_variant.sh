#!/bin/bash
# build ab/libbrc_hip_<name>.so from the working tree with extra hipcc flags (experiments: tools/gpu_ab_multi.py compares them in one process)
#   tools/build_variant.sh <name> [-DBRC_EXP=5 ...]
# (experiment builds only: -DBRC_EXP_KNOBS compiles in the run-time ablation knobs BRC_PILEUP_VARIANT / BRC_ANN_VARIANT /
# BRC_PILEUP_LDS_PAD / BRC_INDEL_OVERLAP, which the product library does not contain)
set -e
cd "$(dirname "$0")/.."; name=$1; shift
mkdir -p ab; C=bam_readcount_amd/csrc
make -s -C $C brc_host.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -DBRC_EXP_KNOBS "$@" -O3 -std=c++17 -fPIC -ffp-contract=off -c $C/brc_engine.hip -o ab/engine_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ab/engine_$name.o $C/brc_host.o -o ab/libbrc_hip_$name.so -pthread
echo "ab/libbrc_hip_$name.so"
