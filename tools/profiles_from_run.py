#!/usr/bin/env python3
"""Copy the evidence of one tools/gpu_r4_profile.sh run (gpurun_out/r04prof/) into profiles/ under the round's names and derive
profiles/r04_traffic.json (what bench.py quotes as `roofline.traffic`) from the FETCH_SIZE / WRITE_SIZE passes.

    python tools/profiles_from_run.py [gpurun_out/r04prof] [r04]
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r04prof")
tag = sys.argv[2] if len(sys.argv) > 2 else "r04"
P = os.path.join(ROOT, "profiles")
HOW = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/gpu_rNN_profile.sh, at the shipped kernels of that round), "
       "per k_pileup2 launch; bytes = FETCH_SIZE x 1024 x 2 (gfx950: wide streaming reads are tallied at half their size, "
       "/opt/skills/guides/MI355X_MICROARCH.md; an upper bound here, the scalar record loads are not wide) + WRITE_SIZE x 1024")
summary = json.load(open(os.path.join(src, "pmc_summary.json")))
traffic = {}
for cfg, mbp in (("wgs30x", 50.0), ("tumor200x", 6.25)):
    for name in ("fetch", "write", "sq1", "sq2"):
        shutil.copy(os.path.join(src, "pmc_%s_%s_raw.csv" % (cfg, name)), os.path.join(P, "%s_pmc_%s_%s_raw.csv" % (tag, cfg, name)))
    shutil.copy(os.path.join(src, "rocprofv3_kernel_stats_%s.csv" % cfg), os.path.join(P, "%s_rocprofv3_kernel_stats_%s.csv" % (tag, cfg)))
    shutil.copy(os.path.join(src, "bench_line_%s.json" % cfg), os.path.join(P, "%s_bench_line_%s.json" % (tag, cfg)))
    kp, ka = summary[cfg]["k_pileup2"], summary[cfg]["k_annotate_groups"]
    traffic[cfg] = {"k_pileup_hbm_bytes_per_launch": int(kp["FETCH_SIZE"] * 1024 * 2 + kp["WRITE_SIZE"] * 1024),
                    "FETCH_SIZE_KiB": kp["FETCH_SIZE"], "WRITE_SIZE_KiB": kp["WRITE_SIZE"], "how": HOW, "contig_mbp": mbp,
                    "k_annotate_groups": {"FETCH_SIZE_KiB": ka["FETCH_SIZE"], "WRITE_SIZE_KiB": ka["WRITE_SIZE"],
                                          "hbm_bytes_per_launch": int(ka["FETCH_SIZE"] * 1024 * 2 + ka["WRITE_SIZE"] * 1024)}}
json.dump(summary, open(os.path.join(P, "%s_pmc_summary.json" % tag), "w"), indent=1)
json.dump(traffic, open(os.path.join(P, "%s_traffic.json" % tag), "w"), indent=1)
print(json.dumps(traffic, indent=1))
