#!/bin/bash
# Round-5 evidence for profiles/: for BASELINE config 3 (wgs30x, 50 Mbp) and the per-GPU shape of config 5 (tumor200x,
# 50 Mbp / 8 = 6.25 Mbp, 4 libraries, -p -i): bench line, rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE and SQ counters
# (each --pmc set in its own pass, with --kernel-trace only).  Output: gpurun_out/r05prof/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05prof; mkdir -p $O; export TMPDIR=/tmp
Q="--cpu-sample-mbp 0 --e2e-mbp 0 --abi-mbp 0 --e2e-configs 0"
for cfg in wgs30x tumor200x; do
  if [ $cfg = wgs30x ]; then A="--mode weak"; else A="--mode strong --contig-mbp 6.25"; fi
  timeout 600 python bench.py --steps 100 --warmup 5 $Q --other-configs 0 $A 2>/dev/null | grep '^{' > $O/bench_line_$cfg.json
  rm -rf /tmp/prof_$cfg
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg -o trace -- python "$OLDPWD/bench.py" --steps 20 --warmup 5 $Q --other-configs 0 $A ) > $O/rocprof_$cfg.log 2>&1
  f=$(find /tmp/prof_$cfg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprofv3_kernel_stats_$cfg.csv
  for set in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq1:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "sq2:SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_WR"; do
    name=${set%%:*}; ctrs=${set#*:}
    rm -rf /tmp/pmc_${cfg}_$name
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc_${cfg}_$name -o pmc -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 $Q $A ) > $O/pmc_${cfg}_$name.log 2>&1
    f=$(find /tmp/pmc_${cfg}_$name -name "*counter_collection.csv" | head -1)
    # raw per-dispatch rows of the two big kernels only (the full CSV is large)
    [ -n "$f" ] && ( head -1 "$f"; grep -E "k_pileup2|k_annotate_groups" "$f" ) > $O/pmc_${cfg}_${name}_raw.csv
  done
done
python - <<'PY'
import csv, glob, json, collections, os
O = "gpurun_out/r05prof"
out = {}
for cfg in ("wgs30x", "tumor200x"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(O, "pmc_%s_*_raw.csv" % cfg)):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[cfg] = {k: {c: round(sum(v) / len(v)) for c, v in d.items()} for k, d in acc.items()}
    p = out[cfg].get("k_pileup2", {})
    if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
        # FETCH_SIZE / WRITE_SIZE are in KiB... units per /opt/skills/guides/MI355X_MICROARCH.md; gfx950: x2 on FETCH for wide streaming reads
        out[cfg]["k_pileup_hbm_bytes_per_launch"] = int(p["FETCH_SIZE"] * 1024 * 2 + p["WRITE_SIZE"] * 1024)
json.dump(out, open(os.path.join(O, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
