import os, sys, subprocess
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
code = r'''
import os, sys
sys.path.insert(0, "%s"); sys.path.insert(0, "%s/tools")
import numpy as np, synthgen
from bam_readcount_amd import capi
lib = capi.Library(os.path.abspath(sys.argv[1]))
cfg = sys.argv[2]
per_lib = cfg == "tumor200x"
names = ["lib%%d" %% i for i in range(synthgen.CONFIGS[cfg]["n_libs"])] if per_lib else ()
opts = dict(min_mapq=20, min_bq=13) if not per_lib else dict(min_mapq=0, min_bq=0, per_lib=True, insertion_centric=True)
ref, arrs = synthgen.generate(int(sys.argv[3]), cfg, seed=5)
eng = capi.Engine(lib, lib_names=names, **opts)
eng.begin_region(0, 0, len(ref), ref); eng.push_reads(arrs); r = eng.end_region(); print("ok", r.n_events)
''' % (ROOT, ROOT)
open("/tmp/w8one.py", "w").write(code)
for cfg in ("wgs30x", "tumor200x"):
    for n in ("20000", "300000"):
        for v in ("0", "3", "4", "5"):
            env = dict(os.environ, BRC_ANN_VARIANT=v)
            p = subprocess.run([sys.executable, "/tmp/w8one.py", sys.argv[1], cfg, n], env=env, capture_output=True, text=True, timeout=120)
            print(cfg, n, "variant", v, "rc", p.returncode, (p.stdout.strip().splitlines() or [""])[-1][:60], (p.stderr.strip().splitlines() or [""])[-1][:160], flush=True)
