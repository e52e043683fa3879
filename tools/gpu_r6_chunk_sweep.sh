#!/bin/bash
# The piece size of long regions (cli.cpp: auto_chunk, BRC_CHUNK_BYTES = compressed bytes per piece): config 5 end to end
# (tools/e2e_configs.py --leg tumor; one process and two ranks on one GPU) and config 3 end to end (a 60-Mbp 30x BAM) at several sizes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; out=gpurun_out/r06; mkdir -p $out
for cb in ${CHUNK_BYTES:-13e6 26e6 52e6 208e6}; do
  BRC_CHUNK_BYTES=$cb timeout 900 python tools/e2e_configs.py --leg tumor --contig-mbp 6.25 --check-mbp 0.1 --reps 3 --ranks 2 --rank-devices 0,0 > $out/e2e_tumor_chunkbytes_$cb.log 2>&1
  python - $cb $out/e2e_tumor_chunkbytes_$cb.log <<'PY'
import sys, json
for l in open(sys.argv[2]):
    if l.startswith("{"):
        d = json.loads(l); print("config 5: bytes per piece", sys.argv[1], "one process", d["seconds"], "s, two ranks", d["sharded"]["seconds"], "s |", d["stages"][1], "|", d["stages"][2][:60])
PY
done
python - <<'PY'
import sys, time, numpy as np
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import synthgen
L = 60_000_000
ref, arrs = synthgen.generate(L, "wgs30x", seed=3)
synthgen.write_bam("/tmp/c3.bam", "chrS", L, arrs)
synthgen.write_fasta("/tmp/c3.fa", [("chrS", ref)])
PY
for cb in ${CHUNK_BYTES:-13e6 26e6 52e6 208e6}; do
  for rep in 1 2 3; do
    t0=$(date +%s%N); BRC_CHUNK_BYTES=$cb BRC_CLI_TIMING=1 bam_readcount_amd/csrc/bam-readcount -w 0 -q 20 -b 13 -f /tmp/c3.fa /tmp/c3.bam chrS > /dev/null 2> /tmp/err.txt; t1=$(date +%s%N)
    echo "config 3 (60 Mbp): bytes per piece $cb: $(( (t1 - t0) / 1000000 )) ms | $(grep '^timing' /tmp/err.txt)"
  done
done
