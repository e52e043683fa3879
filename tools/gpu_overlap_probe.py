#!/usr/bin/env python3
"""How much do K1 (store / latency bound) and k_pileup2 (issue bound) gain from sharing the machine?  Two engines on ONE GPU,
each with half of the config-3 contig, computing concurrently (their kernels interleave on the CUs), against one engine with
the whole contig.  Timing only."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, synthgen
from bam_readcount_amd import capi
hip = capi.load_product()
opts = dict(min_mapq=20, min_bq=13)
def make(n, seed):
    ref, arrs = synthgen.generate(n, "wgs30x", seed=seed)
    e = capi.Engine(hip, **opts); e.begin_region(0, 0, n, ref); e.push_reads(arrs); e.upload(); e.compute(); return e
whole = make(50_000_000, 1)
halves = [make(25_000_000, 2), make(25_000_000, 3)]
def timeit(engs, steps=20, offset_ms=0.0):
    def run(e, i):
        if i: time.sleep(i * offset_ms * 1e-3)
        for _ in range(steps): e.compute()
    t0 = time.perf_counter(); th = [threading.Thread(target=run, args=(e, i)) for i, e in enumerate(engs)]
    [t.start() for t in th]; [t.join() for t in th]; return (time.perf_counter() - t0) / steps * 1e3
for rep in range(3):
    print("one engine, 50 Mbp: %.3f ms/step;  two engines x 25 Mbp concurrently: %.3f ms per pair;  one 25-Mbp engine alone: %.3f ms" % (timeit([whole]), timeit(halves), timeit(halves[:1])), flush=True)
for off in (0.5, 1.0, 1.5, 2.0):
    print("two engines x 25 Mbp, the second started %.1f ms later: %.3f ms per pair" % (off, timeit(halves, steps=40, offset_ms=off)), flush=True)
quarters = [make(12_500_000, 10 + i) for i in range(4)]
print("four engines x 12.5 Mbp concurrently: %.3f ms per set" % timeit(quarters))
