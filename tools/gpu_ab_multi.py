#!/usr/bin/env python3
"""A/B of several builds of libbrc_hip.so in ONE process on one box: the data of each timed shape is generated once, every
build runs it REPS times (interleaved), and every build's result on a small contig of the same data model is compared
with the first build's (planes bit for bit).  Prints one line per (shape, build): median kernel times.

    python tools/gpu_ab_multi.py --libs bam_readcount_amd/csrc/libbrc_hip.so ab/libbrc_hip_x.so ... [--shapes tumor,wgs]
"""
import argparse
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def result_digest(np, r):
    h = hashlib.sha1()
    for a in (r.ncol, r.depth, r.istat, r.fstat):
        h.update(np.ascontiguousarray(a).view(np.uint8).tobytes())
    for x in r.indels:
        h.update(repr((x["pos"], x["lib"], x["len"], x["allele"])).encode())
        h.update(np.ascontiguousarray(x["i"]).tobytes()); h.update(np.ascontiguousarray(x["f"]).tobytes())
    return h.hexdigest()[:12]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", required=True)
    ap.add_argument("--shapes", default="tumor,wgs")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--wgs-mbp", type=float, default=50.0)
    ap.add_argument("--tumor-mbp", type=float, default=6.25)
    ap.add_argument("--long-mbp", type=float, default=20.0)
    args = ap.parse_args()
    import numpy as np
    import torch
    import synthgen
    from bam_readcount_amd import capi
    assert torch.cuda.is_available() or os.environ.get("BRC_AB_NO_GPU")
    libs = [(os.path.basename(p).replace("libbrc_hip", "").replace(".so", "").strip("_") or "default", capi.Library(os.path.abspath(p))) for p in args.libs]
    for shape in args.shapes.split(","):
        config = {"tumor": "tumor200x", "mixed": "wgs30x_mixed", "long": "long10k"}.get(shape, "wgs30x")
        # "wgs_notable": config 3 with the quotient tables switched off (BRC_NO_TABLE, read at upload): every piece takes the
        # path of a read of another length — what that path costs per piece, measured directly
        if shape.endswith("_notable"):
            os.environ["BRC_NO_TABLE"] = "1"
        else:
            os.environ.pop("BRC_NO_TABLE", None)
        per_lib = config == "tumor200x"
        names = ["lib%d" % i for i in range(synthgen.CONFIGS[config]["n_libs"])] if per_lib else ()
        opts = dict(min_mapq=20, min_bq=13) if not per_lib else dict(min_mapq=0, min_bq=0, per_lib=True, insertion_centric=True)
        # ---- same results?
        sref, sarrs = synthgen.generate(300_000, config, seed=5)
        want = None
        for name, lib in libs:
            eng = capi.Engine(lib, lib_names=names, **opts)
            eng.begin_region(0, 0, len(sref), sref); eng.push_reads(sarrs)
            d = result_digest(np, eng.end_region()); eng.close()
            if want is None:
                want = d
            print("%s check %-10s %s %s" % (shape, name, d, "ok" if d == want else "DIFFERENT"), flush=True)
        # ---- timings
        n = int((args.tumor_mbp if per_lib else args.long_mbp if shape == "long" else args.wgs_mbp) * 1e6)
        ref, arrs = synthgen.generate(n, config, seed=1)
        times = {name: [] for name, _ in libs}
        for rep in range(args.reps):
            for name, lib in libs:
                t0 = time.time()
                eng = capi.Engine(lib, lib_names=names, **opts)
                eng.begin_region(0, 0, n, ref); eng.push_reads(arrs); eng.upload()
                eng.compute()
                kn = lib.kernel_names(); acc = np.zeros(len(kn))
                for _ in range(args.steps):
                    ms, _tot = eng.compute(); acc += np.array(ms)
                acc /= args.steps
                eng.close()
                ix = lambda k: kn.index(k) if k in kn else 0
                times[name].append((acc[ix("k_pileup")], acc[ix("k_annotate")], _tot if False else acc.sum()) + tuple(acc))
        for name, _ in libs:
            t = np.array(times[name])
            print("%s %-10s pileup %s  median %.4f  annotate %.4f  all %.4f" % (shape, name, " ".join("%.4f" % x for x in t[:, 0]), np.median(t[:, 0]), np.median(t[:, 1]), np.median(t[:, 2])), flush=True)
            print("%s %-10s slots %s" % (shape, name, " ".join("%s=%.3f" % (k.replace("k_", ""), v) for k, v in zip(libs[0][1].kernel_names(), np.median(t[:, 3:], axis=0)) if k)), flush=True)


if __name__ == "__main__":
    main()
