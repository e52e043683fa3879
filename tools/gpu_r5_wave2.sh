#!/bin/bash
# round 5: read-wise tile compaction (k_compact_reads) + 4-position indel buckets for dense indels — parity, then the `ont` model validated whole
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_checked_build.py --maxfail 6 -q -m gpu -k "operator_every or compacted or deep_indel or wave_form or extreme_scenarios or fuzz or rare_device" 2>&1 | tail -8
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
run 20mbp_wave_form "" 20 5 1 1
run 20mbp_range_walk "BRC_COMPACT_TILES=2" 20 3 1 0
run 20mbp_buckets16 "BRC_IBUCKET_SHIFT=4" 20 3 1 0
run 50mbp_wave_form "" 50 3 1 0
