#!/bin/bash
# end-to-end timing of the product command line on a synthetic BAM (30x, 150 bp) written on the box
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
MBP=${MBP:-2}
python - <<PY
import sys, time, numpy as np
sys.path.insert(0, "tools")
import synthgen, bamio
L = int($MBP * 1e6)
t = time.time(); ref, arrs = synthgen.generate(L, "wgs30x", seed=3); print("generate", round(time.time() - t, 1), "s", len(arrs["pos"]), "reads")
t = time.time()
bamio.write_bam("/tmp/syn.bam", [("chrS", L)], arrs, np.zeros(len(arrs["pos"]), int), block_bytes=60000)
print("write_bam", round(time.time() - t, 1), "s")
with open("/tmp/syn.fa", "wb") as f:
    f.write(b">chrS\n")
    r = np.asarray(ref, np.uint8); n = r.size; rows = (n + 59) // 60
    pad = np.full(rows * 60, 10, np.uint8); pad[:n] = r
    f.write(np.concatenate([pad.reshape(rows, 60), np.full((rows, 1), 10, np.uint8)], axis=1).tobytes())
open("/tmp/syn.fa.fai", "w").write("chrS\t%d\t6\t60\t61\n" % L)
PY
ls -la /tmp/syn.bam
for i in 1 2; do
  echo "cli region:"; time bam_readcount_amd/csrc/bam-readcount -w 1 -q 20 -b 13 -f /tmp/syn.fa /tmp/syn.bam chrS > /tmp/out.txt 2> /tmp/err.txt; tail -1 /tmp/err.txt; wc -lc /tmp/out.txt
done
python - <<PY
import random
random.seed(1)
L = int($MBP * 1e6)
open("/tmp/sites.txt", "w").write("".join("chrS\t%d\t%d\n" % (p, p) for p in sorted(random.sample(range(1, L), 20000))))
PY
echo "cli 20k sites (planner):"; time bam_readcount_amd/csrc/bam-readcount -w 1 -q 20 -b 13 -f /tmp/syn.fa -l /tmp/sites.txt /tmp/syn.bam > /tmp/out2.txt 2> /tmp/err2.txt; tail -1 /tmp/err2.txt; wc -l /tmp/out2.txt
head -2000 /tmp/sites.txt > /tmp/s2.txt; echo "cli 2k sites (line by line):"; time bam_readcount_amd/csrc/bam-readcount -w 1 -q 20 -b 13 --brc-plan 0 -f /tmp/syn.fa -l /tmp/s2.txt /tmp/syn.bam > /tmp/out3.txt 2>/dev/null; wc -l /tmp/out3.txt
echo "cli region -> /dev/null:"; time bam_readcount_amd/csrc/bam-readcount -w 1 -q 20 -b 13 -f /tmp/syn.fa /tmp/syn.bam chrS > /dev/null 2>/dev/null
echo "cli region, 1 formatter thread -> /dev/null:"; time BRC_FORMAT_THREADS=1 bam_readcount_amd/csrc/bam-readcount -w 1 -q 20 -b 13 -f /tmp/syn.fa /tmp/syn.bam chrS > /dev/null 2>/dev/null
echo "decode only (region with no reads kept: -q 255 still decodes):"; time bam_readcount_amd/csrc/bam-readcount -w 1 -q 20 -b 13 -f /tmp/syn.fa /tmp/syn.bam chrS:1-1000 > /dev/null 2>/dev/null
python - <<'PY'
import sys, time
sys.path.insert(0, "tools")
import bamio
t = time.time(); n = 0
import subprocess
PY
for ch in 250000 1000000 4000000; do echo "cli region --brc-chunk $ch -> /dev/null:"; time bam_readcount_amd/csrc/bam-readcount -w 1 -q 20 -b 13 --brc-chunk $ch -f /tmp/syn.fa /tmp/syn.bam chrS > /dev/null 2>/dev/null; done
echo "cli region default chunk, timing:"; time BRC_CLI_TIMING=1 bam_readcount_amd/csrc/bam-readcount -w 1 -q 20 -b 13 -f /tmp/syn.fa /tmp/syn.bam chrS 2>&1 > /dev/null | tail -1
for fc in 4096 16384; do echo "cli region BRC_FORMAT_CHUNK=$fc:"; time BRC_FORMAT_CHUNK=$fc BRC_CLI_TIMING=1 bam_readcount_amd/csrc/bam-readcount -w 1 -q 20 -b 13 -f /tmp/syn.fa /tmp/syn.bam chrS 2>&1 > /dev/null | tail -1; done
