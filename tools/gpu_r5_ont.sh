#!/bin/bash
# round 5: tile compaction — parity with it forced on (every fuzz family, the extreme kinds, long dense-indel reads) and the `ont` model timed + validated whole
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_checked_build.py -x -q -m gpu -k "compacted or operator_every or extreme_scenarios" 2>&1 | tail -4
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
run 20mbp "" 20 5 1 1
run 1mbp_compacted "" 1 3 1 0
run 1mbp_not_compacted "BRC_COMPACT_TILES=0" 1 1 0 0
