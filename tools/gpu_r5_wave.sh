#!/bin/bash
# round 5: the wave-form annotator (k_annotate_wave) — parity (many-operator reads, every fuzz family, extreme kinds, checked build), then the
# `ont` model timed with and without it, and an A/B of the default configuration (does the extra launch cost / the shorter K1 tail pay?)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_checked_build.py --maxfail 6 -q -m gpu 2>&1 | tail -25
run() {   # label, env assignment or "", config, contig Mbp, steps, warmup
  local E="$2"; [ -z "$E" ] && E="X_=1"
  env $E BRC_HIP_LIB=$PWD/bam_readcount_amd/csrc/libbrc_hip_testknobs.so timeout 900 python bench.py --config $3 --contig-mbp $4 --steps $5 --warmup $6 --e2e-mbp 0 --abi-mbp 0 --cpu-ref-mbp 0 --cpu-sample-mbp 0 --other-configs 0 --e2e-configs 0 > gpurun_out/r05_wave_$1.json 2> gpurun_out/r05_wave_$1.err; echo "== $1 rc $?"; tail -c 300 gpurun_out/r05_wave_$1.err
  python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/r05_wave_$1.json").read().strip().splitlines()[-1])
    print("ms_per_step", j["ms_per_step"], "value %.4g" % j["value"], j["roofline"]["kernel_ms"], {k: j["validated"].get(k) for k in ("full_contig", "events", "planes_bit_exact", "text_byte_exact")})
except Exception as ex:
    print("no line:", ex)
PY
}
run ont20_wave "" ont 20 5 1
run ont20_serial "BRC_WAVE_FORM=0" ont 20 2 1
run ont1_wave "" ont 1 3 1
run wgs_wave "" wgs30x 50 20 5
run wgs_serial "BRC_WAVE_FORM=0" wgs30x 50 20 5
run wgs_wave2 "" wgs30x 50 20 5
run wgs_serial2 "BRC_WAVE_FORM=0" wgs30x 50 20 5
run tumor_wave "" tumor200x 6.25 10 3
run tumor_serial "BRC_WAVE_FORM=0" tumor200x 6.25 10 3
run long_wave "" long10k 25 10 3
run long_serial "BRC_WAVE_FORM=0" long10k 25 10 3
