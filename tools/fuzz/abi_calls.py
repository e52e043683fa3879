#!/usr/bin/env python3
"""Random call sequences against the C-ABI (begin / push / upload / compute / fetch / end / format / window / warnings /
options in any order, regions of any size, text routes at random) on the sanitizer build of the simulator-backed engine:
every misuse must come back as an error code, never as a sanitizer report.

    LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:allocator_may_return_null=1 \
        python tools/fuzz/abi_calls.py /tmp/brc_asan/libbrc_sim.so <seed> <engines>"""
import sys, os, random, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np, synth
from bam_readcount_amd import capi
lib = capi.Library(sys.argv[1])
random.seed(int(sys.argv[2]))
rng = np.random.default_rng(1)
ref = synth.make_ref(rng, 1500)
batches = [synth.make_batch(s, ref, 120, style=st, n_libs=2) for s, st in ((1, "mixed"), (2, "wild"), (3, "indel"))]
big = synth.make_batch(9, ref, 40000, style="mixed", n_libs=2, read_len=(20, 40))
errs = {}
for it in range(int(sys.argv[3])):
    per_lib = random.random() < 0.5
    try:
        eng = capi.Engine(lib, per_lib=per_lib, lib_names=["a", "b"] if per_lib else (), text_only=random.random() < 0.5,
                          device_text="chrS" if random.random() < 0.3 else None, insertion_centric=random.random() < 0.5)
    except capi.BrcError as e:
        errs[str(e)[:40]] = errs.get(str(e)[:40], 0) + 1; continue
    for step in range(random.randint(3, 25)):
        op = random.choice(["begin", "push", "upload", "compute", "fetch", "end", "format", "clear", "counts", "warn", "opt", "chrom", "window",
                            "compute_n", "fetch_window", "windows", "bigpush"])
        try:
            if op == "begin":
                a = random.randint(0, 1200); eng.begin_region(0, a, a + random.choice([0, 1, 50, 700]), ref if random.random() < 0.9 else None)
            elif op == "push": eng.push_reads(random.choice(batches))
            elif op == "upload": eng.upload()
            elif op == "compute": eng.compute()
            elif op == "fetch": eng.fetch_result()
            elif op == "end": eng.end_region()
            elif op == "format": eng.format_region("chrS")
            elif op == "clear": eng.clear_indel_queue()
            elif op == "counts": eng.counts()
            elif op == "warn": capi._region_warnings(eng, "chrS") if hasattr(capi, "_region_warnings") else None
            elif op == "opt": lib.lib.brc_set_option(eng.h, random.randint(0, 9), random.choice([0, 1, 2, -1, 10**12]))
            elif op == "chrom": lib.lib.brc_set_chrom(eng.h, random.choice([b"chrS", b"", None]))
            elif op == "compute_n": eng.compute_n(random.choice([-1, 0, 1, 3, 40]))
            elif op == "fetch_window": a = random.randint(0, 1400); eng.fetch_window(a, a + random.choice([-3, 0, 1, 64, 65, 900]))
            elif op == "windows":
                k = random.randint(0, 6); wb = np.array([random.randint(-50, 1600) for _ in range(k)], np.int32)
                eng.region_windows(wb, wb + np.array([random.choice([0, 1, 2, 70, 400]) for _ in range(k)], np.int32))
            elif op == "bigpush": eng.push_reads(big)              # (above the staging pool's threshold)
            elif op == "window": capi._format_window(eng, "chrS", random.randint(0, 500), random.randint(0, 900), random.randint(-5, 5))
        except capi.BrcError as e:
            k = str(e)[:60]; errs[k] = errs.get(k, 0) + 1
        except Exception as e:
            k = "PY " + repr(e)[:60]; errs[k] = errs.get(k, 0) + 1
    eng.close()
print("done", it + 1)
for k, v in sorted(errs.items(), key=lambda kv: -kv[1])[:12]: print("  %5d  %s" % (v, k))
