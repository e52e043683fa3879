#!/bin/bash
# A/B of alternative builds (LIBS) on both timed shapes, same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
BENCH_ARGS="--mode strong --contig-mbp 6.25" REPS=${REPS:-2} bash tools/gpu_ab.sh; cp gpurun_out/ab.log gpurun_out/ab_tumor.log
if [ -z "${SKIP_WGS:-}" ]; then BENCH_ARGS="" REPS=${REPS:-2} bash tools/gpu_ab.sh; cp gpurun_out/ab.log gpurun_out/ab_wgs.log; fi
