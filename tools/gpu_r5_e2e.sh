#!/bin/bash
# round 5: BASELINE configs 4 and 5 through the drop-in CLI on multi-contig BAM + BAI (tools/e2e_configs.py), on the GPU box
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; free -g | head -2; df -h /tmp | tail -1
timeout 900 python -m pytest tests/test_e2e_configs.py -x -q -m gpu 2>&1 | tail -3
timeout 1500 python tools/e2e_configs.py --leg sites --contigs ${SITES_CONTIGS:-8} --contig-mbp ${SITES_MBP:-50} --check-lines 1000 > gpurun_out/r05_e2e_sites.json 2> gpurun_out/r05_e2e_sites.err; echo "sites rc $?"; tail -3 gpurun_out/r05_e2e_sites.err
timeout 1500 python tools/e2e_configs.py --leg tumor --contig-mbp 6.25 --check-mbp 1.0 > gpurun_out/r05_e2e_tumor.json 2> gpurun_out/r05_e2e_tumor.err; echo "tumor rc $?"; tail -3 gpurun_out/r05_e2e_tumor.err
cat gpurun_out/r05_e2e_sites.json gpurun_out/r05_e2e_tumor.json | cut -c1-3000
