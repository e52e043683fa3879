#!/bin/bash
# Ablations + PMC counters for k_pileup (profiling session; results in gpurun_out/)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 0 1 2 3; do
  BRC_PILEUP_VARIANT=$v timeout 300 python bench.py --steps 5 --warmup 1 --cpu-sample-mbp 0 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('variant $v', d['roofline']['kernel_ms'])" | tee -a gpurun_out/ablate.log
done
run_pmc() { # name, counters
  rm -rf gpurun_out/pmc_$1
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d "$OLDPWD/gpurun_out/pmc_$1" -o pmc -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --cpu-sample-mbp 0 ) > gpurun_out/pmc_$1.log 2>&1
  f=$(find gpurun_out/pmc_$1 -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' | tee -a gpurun_out/ablate.log
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "k_pileup" in k or "k_annotate" in k:
        print(k, {c: sum(v) / len(v) for c, v in d.items()})
PY
  find gpurun_out/pmc_$1 -name "*.csv" -size +5M -delete
}
run_pmc sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
run_pmc sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS"
run_pmc fetch "FETCH_SIZE"
run_pmc write "WRITE_SIZE"
run_pmc tcc "TCC_HIT_sum TCC_MISS_sum"
