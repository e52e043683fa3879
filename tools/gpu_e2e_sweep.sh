#!/bin/bash
# thread-budget sweep of the command line on a CPU-quota-limited box (cgroup cpu.max): formatter threads x fetch stripes x pieces ahead
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python - <<PY
import sys, numpy as np
sys.path.insert(0, "tools")
import synthgen
L = 30000000
ref, arrs = synthgen.generate(L, "wgs30x", seed=3)
synthgen.write_bam("/tmp/wgs30x.bam", "chrS", L, arrs)
rows = (L + 59) // 60
pad = np.full(rows * 60, 10, np.uint8); pad[:L] = ref
open("/tmp/wgs30x.fa", "wb").write(b">chrS\n" + np.concatenate([pad.reshape(rows, 60), np.full((rows, 1), 10, np.uint8)], axis=1).tobytes())
open("/tmp/wgs30x.fa.fai", "w").write("chrS\t%d\t6\t60\t61\n" % L)
PY
CLI=bam_readcount_amd/csrc/bam-readcount
for cfg in ${SWEEP:-"64 32 2" "16 8 2" "12 6 2" "8 8 2" "8 4 2" "16 8 1" "12 12 1" "8 8 1" "16 16 1" "24 8 2"}; do
  set -- $cfg
  t0=$(date +%s%N)
  BRC_FORMAT_THREADS=$1 BRC_FETCH_THREADS=$2 BRC_FETCH_AHEAD=$3 BRC_CLI_TIMING=1 $CLI -w 0 -q 20 -b 13 -f /tmp/wgs30x.fa /tmp/wgs30x.bam chrS > /dev/null 2> /tmp/err.txt
  t1=$(date +%s%N)
  echo "format=$1 fetch=$2 ahead=$3: $(( (t1 - t0) / 1000000 )) ms   $(grep '^timing' /tmp/err.txt)"
done
cat /sys/fs/cgroup/cpu.stat | grep -E 'nr_throttled|throttled_usec'
