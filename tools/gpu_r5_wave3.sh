#!/bin/bash
# round 5: read-wise tile compaction with sizes from a pass over the pieces and the search over keyreach[] — parity, then the `ont` model validated whole
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_checked_build.py --maxfail 6 -q -m gpu -k "operator_every or compacted or wave_form or extreme_scenarios or announced or fetch_window" 2>&1 | tail -8
run() {   # label, env assignment or "", contig Mbp, steps, warmup, oracle sample Mbp
  local E="$2"; [ -z "$E" ] && E="X_=1"
  env $E BRC_HIP_LIB=$PWD/bam_readcount_amd/csrc/libbrc_hip_testknobs.so timeout 900 python bench.py --config ont --contig-mbp $3 --steps $4 --warmup $5 --e2e-mbp 0 --abi-mbp 0 --cpu-ref-mbp 0 --cpu-sample-mbp $6 --other-configs 0 --e2e-configs 0 > gpurun_out/r05_bench_line_ont_$1.json 2> gpurun_out/r05_bench_line_ont_$1.err; echo "== $1 rc $?"; tail -c 300 gpurun_out/r05_bench_line_ont_$1.err
  python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/r05_bench_line_ont_$1.json").read().strip().splitlines()[-1])
    print("ms_per_step", j["ms_per_step"], "value %.4g" % j["value"], "events", j["config"]["events_per_step"], j["roofline"]["kernel_ms"], j["config"]["piece_steps"], {k: j["validated"].get(k) for k in ("full_contig", "events", "planes_bit_exact", "text_byte_exact")}, "cpu", (j.get("cpu_baseline") or {}).get("value"))
except Exception as ex:
    print("no line:", ex)
PY
}
run 20mbp_final "" 20 5 1 1
R=$PWD
( cd /tmp && BRC_HIP_LIB=$R/bam_readcount_amd/csrc/libbrc_hip_testknobs.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ont -o trace -- python $R/bench.py --config ont --contig-mbp 20 --steps 3 --warmup 1 --e2e-mbp 0 --abi-mbp 0 --cpu-ref-mbp 0 --cpu-sample-mbp 0 --other-configs 0 --e2e-configs 0 --full-check 0 ) > gpurun_out/rocprof_ont.log 2>&1
f=$(find /tmp/prof_ont -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r05_rocprofv3_kernel_stats_ont.csv && head -22 "$f" | cut -c1-180
