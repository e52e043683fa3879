#!/bin/bash
# round 5: coalesced copy of the read-wise compaction — parity, the `ont` model validated whole, then the extreme fuzz on the bounds-checked
# library
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_checked_build.py --maxfail 6 -q -m gpu -k "operator_every or compacted or wave_form or extreme_scenarios or announced" 2>&1 | tail -6
E="X_=1"
env $E BRC_HIP_LIB=$PWD/bam_readcount_amd/csrc/libbrc_hip_testknobs.so timeout 900 python bench.py --config ont --contig-mbp 20 --steps 5 --warmup 1 --e2e-mbp 0 --abi-mbp 0 --cpu-ref-mbp 0 --cpu-sample-mbp 1 --other-configs 0 --e2e-configs 0 > gpurun_out/r05_bench_line_ont_20mbp_03.json 2> gpurun_out/r05_bench_line_ont_20mbp_03.err; echo "== ont rc $?"
python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/r05_bench_line_ont_20mbp_03.json").read().strip().splitlines()[-1])
    print("ms_per_step", j["ms_per_step"], "value %.4g" % j["value"], j["roofline"]["kernel_ms"], j["config"]["piece_steps"], {k: j["validated"].get(k) for k in ("full_contig", "events", "planes_bit_exact", "text_byte_exact")})
except Exception as ex:
    print("no line:", ex)
PY

