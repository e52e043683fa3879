#!/bin/bash
# round 5: the drop-in command line on CRAM input against the same reads as BAM (15 Mbp at 30x, 3 M reads; the CRAM was written by
# tools/cramio.py in the build container, 5 minutes of Python, into gpurun_in/ — git-ignored, but it travels to the GPU box:
# `mkdir -p gpurun_in && cd gpurun_in && python ../tools/cram_e2e_gen.py 15 && rm syn.bam syn.bam.bai`; remove it afterwards, 300 MB are pushed with every call)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
python - 2>&1 <<'PY' | tee gpurun_out/r05_e2e_cram.log
import os, sys, time, subprocess, hashlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np, synthgen
CLI = os.path.abspath("bam_readcount_amd/csrc/bam-readcount")
d = os.path.abspath("gpurun_in")
n = 15_000_000
ref, a = synthgen.generate(n, "wgs30x", seed=3)
synthgen.write_bam(os.path.join(d, "syn.bam"), "chrS", n, a)
ends = a["pos"].astype(np.int64) + 150
ev = int((np.minimum(ends, n) - a["pos"].astype(np.int64)).clip(min=0).sum())
print("15 Mbp at 30x: %d reads, %d events; BAM %.0f MB, CRAM %.0f MB" % (len(a["pos"]), ev, os.path.getsize(os.path.join(d, "syn.bam")) / 1e6, os.path.getsize(os.path.join(d, "syn.cram")) / 1e6))
md5 = {}
for f, envs in (("syn.bam", [{}]), ("syn.cram", [{}, {"BRC_FETCH_STRIPE_MIN": "1000000000"}])):
    for env in envs:
        cmd = [CLI, "-w", "0", "-q", "20", "-b", "13", "-f", "syn.fa", f, "chrS"]
        tb = []
        for _ in range(3):
            t0 = time.perf_counter(); p = subprocess.run(cmd, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, **env)); tb.append(time.perf_counter() - t0)
            assert p.returncode == 0, p.stderr.decode()[-400:]
        label = f + (" (one reader, no stripes)" if env else "")
        print("%-38s %s s -> best %.3f s = %.2f G events/s" % (label, " ".join("%.3f" % x for x in tb), min(tb), ev / min(tb) / 1e9))
        p = subprocess.run(cmd, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, BRC_CLI_TIMING="1", **env))
        md5[label] = hashlib.md5(p.stdout).hexdigest()
        print("   " + "\n   ".join(l for l in p.stderr.decode().splitlines() if l.startswith(("timing:", "startup:"))))
print("same text from BAM and CRAM:", len(set(md5.values())) == 1, md5)
PY
