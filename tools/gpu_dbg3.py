import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["BRC_FORCE_DOM"] = "0"
from bam_readcount_amd import capi
import parity, synth
hip = capi.load_product(); oracle = capi.Library(os.path.join(ROOT, "oracle", "libbrc_oracle.so"))
seed = 2
rng = np.random.default_rng(seed)
ref = synth.make_ref(rng, 3000)
arrs = synth.make_batch(seed + 100, ref, 300, style="simple")
ta, ra = parity.run_engine(hip, arrs, [(0, 3000)], ref=ref)
tb, rb = parity.run_engine(oracle, arrs, [(0, 3000)], ref=ref)
a, b = ra[0], rb[0]
bad = sorted(set(int(i[3]) for i in np.argwhere(a.istat != b.istat)))
print("bad positions", bad[:40], len(bad))
for k in bad[:6]:
    print("pos", k, "ref", chr(ref[a.pos0 + k]), "tile", k // 64, "lane", k % 64)
    print("  hip n per bucket", a.istat[0, :, 0, k].tolist(), "oracle", b.istat[0, :, 0, k].tolist())
    print("  hip smq", a.istat[0, :, 1, k].tolist(), "oracle", b.istat[0, :, 1, k].tolist())
    print("  hip sev", a.fstat[0, :, 0, k].tolist(), "oracle", b.fstat[0, :, 0, k].tolist())
