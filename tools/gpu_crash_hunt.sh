#!/bin/bash
# stress of the multi-engine command line (several engines on one device) to catch the rare start-up crash with a backtrace
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; out=gpurun_out/r03/crash; mkdir -p $out; export BRC_CRASH_DIR="$PWD/$out"
python - <<'PY'
import sys, pathlib
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "tools")
import numpy as np, test_cli
class F:
    def mktemp(self, n):
        p = pathlib.Path("/tmp/synfix"); p.mkdir(exist_ok=True); return p
d = test_cli.synthetic_bam.__wrapped__(F())
rng = np.random.default_rng(11)
sites = [("chrA", int(p), int(p)) for p in rng.integers(1, 5000, 200)] + [("chrB", 10, 2500), ("chrA", 300, 2900), ("chrB", 5, 5)]
test_cli._sites_file(d, "sites_multi.txt", sites)
PY
CLI=$PWD/bam_readcount_amd/csrc/bam-readcount
cd /tmp/synfix
n=0; bad=0
for i in $(seq 1 ${ROUNDS:-30}); do
  for extra in "--brc-gpus 2" "--brc-gpus 3" "--brc-streams 3" "--brc-gpus 2 --brc-streams 2"; do
    BRC_DEVICES=0,0,0 $CLI -w 0 --brc-chunk 333 $extra -p -f syn.fa -l sites_multi.txt syn.bam > /dev/null 2> /tmp/err_$i.txt &
    BRC_DEVICES=0,0,0 $CLI -w 0 --brc-chunk 333 $extra -q 10 -b 5 -f syn.fa -l sites_multi.txt --brc-plan 16 syn.bam > /dev/null 2> /tmp/err2_$i.txt
    rc2=$?; wait $!; rc1=$?
    n=$((n+2)); [ $rc1 -ne 0 ] && bad=$((bad+1)) && cp /tmp/err_$i.txt $BRC_CRASH_DIR/stderr_${i}_a.txt; [ $rc2 -ne 0 ] && bad=$((bad+1)) && cp /tmp/err2_$i.txt $BRC_CRASH_DIR/stderr_${i}_b.txt
  done
done
echo "runs $n, failures $bad"; ls $BRC_CRASH_DIR | head
