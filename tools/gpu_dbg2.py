"""debug helper (GPU box): hip vs oracle on small cases, one subprocess each so a crash does not hide the others"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASE = r'''
import os, sys, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
from bam_readcount_amd import capi
import parity, synth
hip = capi.load_product(); oracle = capi.Library(os.path.join(%(root)r, "oracle", "libbrc_oracle.so"))
seed, style, n, opts = %(seed)d, %(style)r, %(n)d, %(opts)r
rng = np.random.default_rng(seed)
ref = synth.make_ref(rng, 3000)
arrs = synth.make_batch(seed + 100, ref, n, style=style)
ta, ra = parity.run_engine(hip, arrs, [(0, 3000)], ref=ref, **opts)
tb, rb = parity.run_engine(oracle, arrs, [(0, 3000)], ref=ref, **opts)
a, b = ra[0], rb[0]
for name in ("ncol", "depth", "istat", "fstat"):
    x, y = getattr(a, name), getattr(b, name)
    if name == "fstat": x, y = x.view(np.uint32), y.view(np.uint32)
    bad = np.argwhere(x != y)
    print(name, "mismatches", len(bad), "first", bad[:5].tolist(), [(int(x[tuple(i)]), int(y[tuple(i)])) for i in bad[:5]])
print("text equal", ta == tb, "n_events", a.n_events, b.n_events)
'''
cases = [(2, "simple", 300, {}, {"BRC_FLUSH_K": "3"}), (2, "simple", 300, {}, {"BRC_FORCE_DOM": "0"}), (2, "simple", 300, {}, {"BRC_PACK_LIM": "255"}), (2, "simple", 300, {}, {"BRC_NO_TABLE": "1"}), (5, "simple", 1000, {}, {}), (5, "simple", 2500, {}, {"BRC_FLUSH_K": "63"})]
for seed, style, n, opts, env in cases:
    src = CASE % dict(root=ROOT, seed=seed, style=style, n=n, opts=opts)
    p = subprocess.run([sys.executable, "-c", src], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120, env=dict(os.environ, **env))
    print("== case", seed, style, n, env, "rc", p.returncode)
    print(p.stdout.decode()[-1500:])
