#!/bin/bash
# round 5: `bench.py --config ont_ul` — 30-70-kb reads with an operator every ~15 bases (2 000-4 500 M operators per read: the wave-form
# annotator's one-wave-per-workgroup instantiation), 20 Mbp at 30x, the whole region validated
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 800 python bench.py --config ont_ul --contig-mbp 20 --steps 3 --warmup 1 --e2e-mbp 0 --abi-mbp 0 --cpu-ref-mbp 0 --cpu-sample-mbp 1 --other-configs 0 --e2e-configs 0 > gpurun_out/r05_bench_line_ont_ul_20mbp.json 2> gpurun_out/r05_bench_line_ont_ul_20mbp.err; echo "rc $?"; tail -c 400 gpurun_out/r05_bench_line_ont_ul_20mbp.err
python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/r05_bench_line_ont_ul_20mbp.json").read().strip().splitlines()[-1])
    print("ms_per_step", j["ms_per_step"], "value %.4g" % j["value"], "events", j["config"]["events_per_step"], j["roofline"]["kernel_ms"], j["config"]["piece_steps"], {k: j["validated"].get(k) for k in ("full_contig", "events", "planes_bit_exact", "text_byte_exact")}, j["config"]["workload"][:120])
except Exception as ex:
    print("no line:", ex)
PY
