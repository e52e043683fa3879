#!/bin/bash
# PMC counters (SQ instruction mix + waits) for the pileup/annotate kernels
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/pmc.log
run_pmc() {
  rm -rf gpurun_out/pmc_$1
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d "$OLDPWD/gpurun_out/pmc_$1" -o pmc -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --cpu-sample-mbp 0 --e2e-mbp 0 ${BENCH_ARGS:-} ) > gpurun_out/pmc_$1.log 2>&1
  f=$(find gpurun_out/pmc_$1 -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' | tee -a gpurun_out/pmc.log
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "k_pileup" in k or "k_annotate" in k:
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
  rm -rf gpurun_out/pmc_$1
}
for set in ${PMC_SETS:-sq1 sq2}; do
  case $set in
    sq1) run_pmc sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS";;
    sq2) run_pmc sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM";;
    sq3) run_pmc sq3 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_BRANCH SQ_WAVES_EQ_64 SQ_INSTS_VSKIPPED SQ_INST_LEVEL_LDS";;
    sqc) run_pmc sqc "SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_IFETCH SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES";;
    lvl) run_pmc lvl "SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SMEM SQ_BUSY_CYCLES";;
    fetch) run_pmc fetch "FETCH_SIZE";;
    write) run_pmc write "WRITE_SIZE";;
  esac
done
