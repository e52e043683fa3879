#!/bin/bash
# end-to-end timing of the drop-in command line (BAM decode -> engine -> text -> /dev/null) on synthetic BAMs written on the
# box: engines per GPU (--brc-streams), phase timers of the engine (BRC_ENGINE_TIMING) and of the CLI (BRC_CLI_TIMING)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
MBP=${MBP:-30}
python - <<PY
import sys, time, numpy as np
sys.path.insert(0, "tools")
import synthgen
for cfg, mbp, nl in (("wgs30x", $MBP, 1), ("tumor200x", ${TMBP:-3}, 4)):
    L = int(mbp * 1e6)
    t = time.time(); ref, arrs = synthgen.generate(L, cfg, seed=3); t1 = time.time()
    synthgen.write_bam("/tmp/%s.bam" % cfg, "chrS", L, arrs, n_libs=nl); t2 = time.time()
    rows = (L + 59) // 60
    pad = np.full(rows * 60, 10, np.uint8); pad[:L] = ref
    open("/tmp/%s.fa" % cfg, "wb").write(b">chrS\n" + np.concatenate([pad.reshape(rows, 60), np.full((rows, 1), 10, np.uint8)], axis=1).tobytes())
    open("/tmp/%s.fa.fai" % cfg, "w").write("chrS\t%d\t6\t60\t61\n" % L)
    from bam_readcount_amd import capi
    ev = int((np.minimum(capi.read_ends(arrs), L) - arrs["pos"].astype(np.int64)).clip(min=0).sum())
    print(cfg, "generate %.1f s, write_bam %.1f s, %d reads, %d events" % (t1 - t, t2 - t1, len(arrs["pos"]), ev))
PY
ls -la /tmp/*.bam
CLI=bam_readcount_amd/csrc/bam-readcount
run() { # label, env..., -- args
  local label=$1; shift
  local t0=$(date +%s%N)
  env "$@" > /dev/null 2> /tmp/err.txt
  local t1=$(date +%s%N)
  echo "$label: $(( (t1 - t0) / 1000000 )) ms"; grep -E '^timing|^engine timing' /tmp/err.txt
}
for s in ${STREAMS:-1 2 3 4}; do
  run "wgs30x ${MBP}Mbp streams=$s" BRC_CLI_TIMING=1 BRC_ENGINE_TIMING=1 $CLI -w 0 -q 20 -b 13 --brc-streams $s -f /tmp/wgs30x.fa /tmp/wgs30x.bam chrS
done
run "wgs30x ${MBP}Mbp streams=3 (again)" $CLI -w 0 -q 20 -b 13 -f /tmp/wgs30x.fa /tmp/wgs30x.bam chrS
run "wgs30x first 10 Mbp streams=3" $CLI -w 0 -q 20 -b 13 -f /tmp/wgs30x.fa /tmp/wgs30x.bam chrS:1-10000000
run "wgs30x first 1 kbp" BRC_ENGINE_TIMING=1 $CLI -w 0 -q 20 -b 13 -f /tmp/wgs30x.fa /tmp/wgs30x.bam chrS:1-1000
for ft in 32 96 128; do run "wgs30x streams=3 format threads $ft" BRC_FORMAT_THREADS=$ft BRC_ENGINE_TIMING=1 $CLI -w 0 -q 20 -b 13 -f /tmp/wgs30x.fa /tmp/wgs30x.bam chrS; done
run "wgs30x chunk 500k streams=4" $CLI -w 0 -q 20 -b 13 --brc-streams 4 --brc-chunk 500000 -f /tmp/wgs30x.fa /tmp/wgs30x.bam chrS
run "wgs30x chunk 2M streams=3" $CLI -w 0 -q 20 -b 13 --brc-streams 3 --brc-chunk 2000000 -f /tmp/wgs30x.fa /tmp/wgs30x.bam chrS
run "wgs30x to a file, streams=3" bash -c "$CLI -w 0 -q 20 -b 13 -f /tmp/wgs30x.fa /tmp/wgs30x.bam chrS:1-10000000 > /tmp/out.txt"
ls -la /tmp/out.txt; rm -f /tmp/out.txt
for s in 1 3; do
  run "tumor200x -p -i streams=$s" BRC_CLI_TIMING=1 BRC_ENGINE_TIMING=1 $CLI -w 0 -p -i --brc-streams $s -f /tmp/tumor200x.fa /tmp/tumor200x.bam chrS
done
