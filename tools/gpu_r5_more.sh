#!/bin/bash
# round 5: 400 more scenarios of the extreme fuzz on the bounds-checked library, and a DENSE site list end to end (100 000 lines on the same
# 400-Mbp genome: one line per 4 kb, several lines share an index window)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python tools/e2e_configs.py --leg sites --contigs 8 --contig-mbp 50 --sites 100000 --check-lines 1000 > gpurun_out/r05_e2e_sites_100k_dense.json 2> gpurun_out/r05_e2e_sites_100k_dense.err; echo "dense sites rc $?"; tail -2 gpurun_out/r05_e2e_sites_100k_dense.err; python -c "
import json; j=json.load(open('gpurun_out/r05_e2e_sites_100k_dense.json')); print(j['seconds'], j['sites_per_s'], j['validated']); print('\n'.join(j['stages'][:3]))"
timeout 1500 python tools/fuzz/extreme.py --lib bam_readcount_amd/csrc/libbrc_hip_checked.so --first 7000 --count 420 > gpurun_out/r05_extreme_fuzz_checked_2.log 2>&1; echo "fuzz rc $?"; tail -3 gpurun_out/r05_extreme_fuzz_checked_2.log
