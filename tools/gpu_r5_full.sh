#!/bin/bash
# round 5: the whole GPU suite + the driver's bench command
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) 2>&1 | tee gpurun_out/r05_pytest_gpu_full.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err ) 2>&1 | tail -4
tail -c 600 gpurun_out/r05_bench_default.err; python - <<'PY'
import json
j = json.loads(open("gpurun_out/r05_bench_default.json").read().strip().splitlines()[-1])
print("value %.4g ms %.4f" % (j["value"], j["ms_per_step"])); print(j["roofline"]["kernel_ms"]); print("validated", {k: j["validated"][k] for k in list(j["validated"])[:6]})
for k in ("e2e", "e2e_sites", "e2e_tumor"):
    v = j.get(k) or {}
    print(k, {x: v.get(x) for x in ("seconds", "value", "sites_per_s", "error")}, (v.get("validated") or {}))
print({k: (v.get("ms_per_step"), v.get("error")) for k, v in (j.get("other_configs") or {}).items()})
PY
