#!/bin/bash
# round-2 session 1: parity tests + smoke + the default bench line (cpu_baseline, e2e, validation), the CLI end to end,
# and the k_pileup2 ablations on the config-5 per-GPU shape
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
DO_PROF=0 BENCH_STEPS=300 bash tools/gpu_session.sh
MBP=10 bash tools/gpu_cli_bench.sh > gpurun_out/cli_bench.log 2>&1
BENCH_ARGS="--mode strong --contig-mbp 6.25" SETTINGS="none;BRC_PILEUP_VARIANT=1;BRC_PILEUP_VARIANT=4;BRC_PILEUP_VARIANT=5;BRC_PILEUP_VARIANT=6;BRC_PILEUP_VARIANT=7;BRC_PILEUP_VARIANT=10" bash tools/gpu_quick2.sh
