#!/bin/bash
# end-to-end on the config-5 data model (200x, 4 libraries, -p -i): text routes compared
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python - <<PY
import sys, numpy as np
sys.path.insert(0, "tools")
import synthgen
L = int(${TMBP:-4} * 1e6)
ref, arrs = synthgen.generate(L, "tumor200x", seed=3)
synthgen.write_bam("/tmp/tumor.bam", "chrS", L, arrs, n_libs=4)
rows = (L + 59) // 60
pad = np.full(rows * 60, 10, np.uint8); pad[:L] = ref
open("/tmp/tumor.fa", "wb").write(b">chrS\n" + np.concatenate([pad.reshape(rows, 60), np.full((rows, 1), 10, np.uint8)], axis=1).tobytes())
open("/tmp/tumor.fa.fai", "w").write("chrS\t%d\t6\t60\t61\n" % L)
from bam_readcount_amd import capi
print("events", int((np.minimum(capi.read_ends(arrs), L) - arrs["pos"].astype(np.int64)).clip(min=0).sum()))
PY
CLI=bam_readcount_amd/csrc/bam-readcount
if [ -n "${MD5:-}" ]; then
for v in "X=1" "BRC_DEVICE_TEXT_MAX_SHARE=100" "BRC_DEVICE_TEXT=0"; do
  env $v timeout 120 $CLI -w 0 -p -i -f /tmp/tumor.fa /tmp/tumor.bam chrS 2>/dev/null | md5sum | cut -c1-8 > /tmp/md5.txt; echo "$v: md5 $(cat /tmp/md5.txt)"
done
fi
for v in "X=1" "BRC_DEVICE_TEXT_MAX_SHARE=100" "BRC_DEVICE_TEXT=0" "X=1" "BRC_DEVICE_TEXT_MAX_SHARE=100" "BRC_DEVICE_TEXT=0"; do sleep 1
  t0=$(date +%s%N); env $v BRC_CLI_TIMING=1 timeout 120 $CLI -w 0 -p -i -f /tmp/tumor.fa /tmp/tumor.bam chrS 2>/tmp/err.txt >/dev/null; t1=$(date +%s%N)
  echo "$v: $(( (t1 - t0) / 1000000 )) ms $(grep -E '^timing' /tmp/err.txt | tr '\n' ' ')"
done
