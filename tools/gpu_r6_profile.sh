#!/bin/bash
# Round-6 evidence for profiles/: per workload (CFGS, default: BASELINE config 3 = wgs30x 50 Mbp; the per-GPU shape of config 5 = tumor200x
# 6.25 Mbp, 4 libraries, -p -i; the mixed-length and the NovaSeq-like models) the bench line, rocprofv3 kernel stats, and — each --pmc set
# in a pass of its own, with --kernel-trace only — FETCH_SIZE, WRITE_SIZE and the SQ counters of EVERY kernel of the step.  The summaries
# are stamped with the hash of the kernel object they ran on (capi.kernel_object_hash): bench.py reports the counters only for that object.
# Output: gpurun_out/r06prof/ (copy traffic.json / pmc_summary.json / the kernel stats to profiles/r06_*).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r06prof; mkdir -p $O; export TMPDIR=/tmp
Q="--cpu-sample-mbp 0 --e2e-mbp 0 --abi-mbp 0 --e2e-configs 0 --other-configs 0"
STEPS_PMC=2
for cfg in ${CFGS:-wgs30x tumor200x wgs30x_mixed novaseq}; do
  case $cfg in
    wgs30x) A="--mode weak";;
    tumor200x) A="--mode strong --contig-mbp 6.25";;
    *) A="--mode weak --config $cfg";;
  esac
  timeout 600 python bench.py --steps 100 --warmup 5 $Q $A 2>/dev/null | grep '^{' > $O/bench_line_$cfg.json
  rm -rf /tmp/prof_$cfg
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg -o trace -- python "$OLDPWD/bench.py" --steps 20 --warmup 5 $Q $A ) > $O/rocprof_$cfg.log 2>&1
  f=$(find /tmp/prof_$cfg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprofv3_kernel_stats_$cfg.csv
  for set in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq1:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "sq2:SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_WR"; do
    name=${set%%:*}; ctrs=${set#*:}
    [ -n "${PMC_SETS:-}" ] && ! echo " $PMC_SETS " | grep -q " $name " && continue
    rm -rf /tmp/pmc_${cfg}_$name
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc_${cfg}_$name -o pmc -- python "$OLDPWD/bench.py" --steps $STEPS_PMC --warmup 1 $Q $A ) > $O/pmc_${cfg}_$name.log 2>&1
    f=$(find /tmp/pmc_${cfg}_$name -name "*counter_collection.csv" | head -1)
    # per-dispatch rows of the engine's kernels (the full CSV also holds torch's)
    [ -n "$f" ] && ( head -1 "$f"; grep -E "brc::k_|k_pileup2|k_annotate|k_xev|k_indel|k_scan|k_tiles|k_reach|k_finalize|k_refcode|k_pick_wave|k_unavail|k_compact|k_count_piece" "$f" ) > $O/pmc_${cfg}_${name}_raw.csv
  done
done
python - <<'PY'
import csv, glob, json, collections, os, sys
sys.path.insert(0, ".")
from bam_readcount_amd import capi
O = "gpurun_out/r06prof"
kobj = capi.kernel_object_hash()
traffic, pmc = {}, {}
mbp = {"tumor200x": 6.25}
for cfg in ("wgs30x", "tumor200x", "wgs30x_mixed", "novaseq"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    launches = collections.Counter()
    for f in glob.glob(os.path.join(O, "pmc_%s_*_raw.csv" % cfg)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not acc:
        continue
    pmc[cfg] = {k: {c: round(sum(v) / len(v)) for c, v in d.items()} for k, d in acc.items()}
    pmc[cfg]["kernel_object_sha256_16"] = kobj
    t = {"kernel_object_sha256_16": kobj, "contig_mbp": mbp.get(cfg, 50.0),
         "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/gpu_r6_profile.sh), averaged per launch of every kernel of the step; bytes = FETCH_SIZE x 1024 x 2 "
                "(gfx950: wide streaming reads are tallied at half their size, /opt/skills/guides/MI355X_MICROARCH.md; an upper bound for kernels whose loads are not wide) + WRITE_SIZE x 1024; "
                "step_hbm_bytes = the sum over the kernels of one step (3 passes per run: 1 warm-up + 2 timed; kernels that run several times per step — the scans — counted as often)"}
    step = 0.0
    # launches per step of every kernel: dispatches seen / passes (warm-up + steps), from the FETCH pass
    f = os.path.join(O, "pmc_%s_fetch_raw.csv" % cfg)
    per_kernel_rows = collections.Counter()
    if os.path.exists(f):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "FETCH_SIZE":
                per_kernel_rows[r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]] += 1
    passes = max(per_kernel_rows.get("k_pileup2", 3), 1)
    for k, d in pmc[cfg].items():
        if not isinstance(d, dict) or "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
            continue
        b = int(d["FETCH_SIZE"] * 1024 * 2 + d["WRITE_SIZE"] * 1024)
        n_per_step = per_kernel_rows.get(k, passes) / passes
        t[k] = {"FETCH_SIZE_KiB": d["FETCH_SIZE"], "WRITE_SIZE_KiB": d["WRITE_SIZE"], "hbm_bytes_per_launch": b, "launches_per_step": round(n_per_step, 2)}
        step += b * n_per_step
    t["step_hbm_bytes"] = int(step)
    traffic[cfg] = t
json.dump(traffic, open(os.path.join(O, "traffic.json"), "w"), indent=1)
json.dump(pmc, open(os.path.join(O, "pmc_summary.json"), "w"), indent=1)
print(json.dumps({c: {k: (v if not isinstance(v, dict) else v.get("hbm_bytes_per_launch")) for k, v in t.items() if k != "how"} for c, t in traffic.items()}, indent=1)[:4000])
PY
