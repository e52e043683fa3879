#!/bin/bash
# debug helper: golden site list through the product CLI, default annotate kernel vs BRC_ANNOTATE=batch
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
python - <<'PY'
import os, subprocess, numpy as np, sys
sys.path.insert(0, "tests")
from conftest import GOLDEN, ROOT, load_fixture
d = "/tmp/dbg"; os.makedirs(d, exist_ok=True)
tb = load_fixture("test_bam.npz")
ref = tb["ref"]; n = ref.size; L = 60; rows = (n + L - 1) // L
pad = np.full(rows * L, ord("\n"), np.uint8); pad[:n] = ref
body = np.concatenate([pad.reshape(rows, L), np.full((rows, 1), 10, np.uint8)], axis=1).tobytes()
open(d + "/ref.fa", "wb").write(b">21\n" + body)
open(d + "/ref.fa.fai", "w").write("21\t%d\t4\t60\t61\n" % n)
exe = os.path.join(ROOT, "bam_readcount_amd", "csrc", "bam-readcount")
outs = {}
for mode in ("groups", "batch"):
    env = dict(os.environ); env["BRC_ANNOTATE"] = mode
    p = subprocess.run([exe, "-w", "1", "-f", d + "/ref.fa", os.path.join(GOLDEN, "test.bam"), "21:10402985-10402990"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    lines = p.stdout.decode(errors="replace").splitlines()
    for l in lines:
        if l.startswith("DBG"): print(mode, l)
    outs[mode] = [l for l in lines if not l.startswith("DBG")]
for a, b in zip(outs["groups"], outs["batch"]):
    if a != b:
        fa, fb = a.split("\t"), b.split("\t")
        for x, y in zip(fa, fb):
            if x != y: print("G", x); print("B", y)
        break
else:
    print("identical", len(outs["groups"]))
PY
