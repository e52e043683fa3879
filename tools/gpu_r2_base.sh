#!/bin/bash
# Round-2 baseline: bench lines + rocprofv3 kernel stats for config 3 (wgs30x 50 Mbp) and the config-5 per-GPU shape
# (tumor200x, 50 Mbp / 8 GPUs = 6.25 Mbp, 4 libraries, -p -i).  Output: gpurun_out/base_*.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${TAG:-base}
for cfg in wgs30x tumor200x; do
  if [ $cfg = wgs30x ]; then A="--config wgs30x"; else A="--config tumor200x --contig-mbp 6.25"; fi
  timeout 600 python bench.py --steps 10 --warmup 2 --cpu-sample-mbp 0 $A 2>&1 | grep '^{' > gpurun_out/${T}_bench_$cfg.json
  python -c "import json;d=json.load(open('gpurun_out/${T}_bench_$cfg.json'));print('$cfg',d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['kernel_ms'])"
  rm -rf gpurun_out/${T}_prof_$cfg
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/${T}_prof_$cfg" -o trace -- python "$OLDPWD/bench.py" --steps 5 --warmup 1 --cpu-sample-mbp 0 $A ) > gpurun_out/${T}_rocprof_$cfg.log 2>&1
  find gpurun_out/${T}_prof_$cfg -name "*kernel_trace.csv" -delete; find gpurun_out/${T}_prof_$cfg -name "*.db" -delete
  f=$(find gpurun_out/${T}_prof_$cfg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
done
