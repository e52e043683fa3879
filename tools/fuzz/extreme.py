#!/usr/bin/env python3
"""Differential fuzz at the EDGES of the parameter space (test infrastructure): the product's device functions (CPU lane simulator by
default, or the HIP library with --hip on a GPU box) against the C oracle on scenarios the suite's families do not reach — reads
of tens of kilobases, piles thousands deep at one position, a dozen libraries, base-quality and mapping-quality thresholds at
and beyond their ranges, tiny -d.  Planes, indel buckets, warning counters and text must be identical.

    python tools/fuzz/extreme.py [--first 0] [--count 200] [--hip | --lib libbrc_hip_checked.so] [--ref]

Found with it: the 16-bit packed sums overflowing on reads above 5461 bases (brc_core.h: choose_pack)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bam_readcount_amd import capi
import parity
import synth


def spliced_cigar(rng, L, style):
    """RNA-seq-like and structural-variant-like alignments: introns (N) of 50 bases to 100 kb, deletions and insertions of tens to
    hundreds of bases, soft clips of up to half the read, hard clips"""
    if L < 12:
        return [(0, L)]
    ops = []; rem = L
    if rng.random() < 0.2:
        ops.append((5, int(rng.integers(1, 200))))
    if rng.random() < 0.3:
        c = int(rng.integers(1, max(2, rem // 2))); ops.append((4, c)); rem -= c
    tail = 0
    if rng.random() < 0.3 and rem > 8:
        tail = int(rng.integers(1, max(2, rem // 2))); rem -= tail
    nseg = int(rng.integers(1, 5))
    while True:
        if nseg <= 1 or rem < 4:
            ops.append((0, rem)); rem = 0
            break
        m = int(rng.integers(1, rem - 1)); ops.append((0, m)); rem -= m
        r = rng.random()
        if r < 0.5:
            ops.append((3, int(np.exp(rng.uniform(np.log(50), np.log(100_000))))))
        elif r < 0.75:
            ops.append((2, int(rng.integers(20, 600))))
        elif rem > 3:
            k = int(rng.integers(1, min(120, rem - 1))); ops.append((1, k)); rem -= k
        nseg -= 1
    if ops[-1][0] != 0:
        ops.append((0, 1))
        for i, (o, l) in enumerate(ops[:-1]):
            if o == 0 and l > 1:
                ops[i] = (o, l - 1); break
        else:
            ops.pop()
    if tail:
        ops.append((4, tail))
    out = []
    for o, l in ops:
        if out and out[-1][0] == o:
            out[-1] = (o, out[-1][1] + l)
        else:
            out.append((o, l))
    assert sum(l for o, l in out if o in (0, 1, 4)) == L, out
    return out


def scenario(seed):
    rng = np.random.default_rng(seed)
    kinds = ["long", "deep", "libs", "thresholds", "tiny", "mixed_len", "dense_indel", "spliced", "spliced"]
    if seed >= 20_000:          # (seeds below keep the scenarios of the earlier logs) reads with hundreds to thousands of operators: wave-form annotator, tile compaction
        kinds += ["many_ops", "many_ops", "many_ops"]
    kind = str(rng.choice(kinds))
    if kind == "spliced":
        RL = int(rng.integers(150_000, 400_000))
        ref = synth.make_ref(rng, RL + 450_000, weird=float(rng.choice([0, 0.02])))
        n_libs = int(rng.choice([1, 2, 4])); read_len = (int(rng.integers(12, 100)), int(rng.choice([100, 150, 250, 600])))
        saved = synth.random_cigar; synth.random_cigar = spliced_cigar
        try:
            arrs = synth.make_batch(seed + 77, ref, int(rng.integers(200, 2_500)), style="indel", n_libs=n_libs, read_len=read_len, region=(0, RL),
                                    p_nolib=float(rng.choice([0, 0.05])), mismatch=float(rng.choice([0.0, 0.02, 0.3])))
        finally:
            synth.random_cigar = saved
        regions = []
        for _ in range(int(rng.integers(1, 4))):
            a = int(rng.integers(0, RL)); regions.append((a, min(a + int(rng.choice([1, 64, 5_000, 120_000, RL])), RL + 400_000)))
        kw = dict(min_mapq=int(rng.choice([0, 20])), min_bq=int(rng.choice([0, 13])), insertion_centric=bool(rng.random() < 0.4))
        if rng.random() < 0.5:
            kw.update(per_lib=True, lib_names=["lib%03d" % i for i in range(n_libs)])
        return kind, "spliced", ref, arrs, regions, kw, bool(rng.random() < 0.5)
    RL = int(rng.integers(2_000, 40_000)) if kind in ("long", "many_ops") else int(rng.integers(300, 4_000))
    n_libs = int(rng.choice([5, 12, 64, 254])) if kind == "libs" else int(rng.choice([1, 1, 2, 4]))
    style = str(rng.choice(["simple", "indel", "wild", "mixed", "clip"] if kind != "dense_indel" else ["wild", "indel"]))
    if kind == "many_ops":
        style = "many"; hi = int(rng.choice([400, 1_500, 6_000, 20_000])); read_len = (int(rng.integers(60, min(hi, 3_000))), hi); n_reads = int(rng.integers(20, 400 if hi < 6_000 else 80))
    elif kind == "long":
        hi = int(rng.choice([6_000, 12_000, 30_000])); read_len = (int(rng.integers(50, hi)), hi); n_reads = int(rng.integers(20, 200))
    elif kind == "deep":
        read_len = (int(rng.integers(20, 100)), int(rng.integers(100, 300))); n_reads = int(rng.integers(4_000, 15_000)); RL = int(rng.integers(150, 600))
    elif kind == "tiny":
        read_len = (1, int(rng.integers(1, 12))); n_reads = int(rng.integers(50, 3_000))
    elif kind == "mixed_len":
        read_len = (int(rng.integers(1, 60)), int(rng.choice([255, 256, 257, 511, 512, 513, 700]))); n_reads = int(rng.integers(300, 3_000))
    else:
        read_len = (20, int(rng.integers(40, 400))); n_reads = int(rng.integers(100, 2_500))
    ref = synth.make_ref(rng, RL + read_len[1] + 200, weird=float(rng.choice([0, 0, 0.02, 0.2])))
    arrs = synth.make_batch(seed + 77, ref, n_reads, style=style, n_libs=n_libs, p_nolib=float(rng.choice([0, 0, 0.05, 0.5])), read_len=read_len,
                            region=(0, RL), mismatch=float(rng.choice([0.0, 0.02, 0.02, 0.3, 0.9])), p_q2tail=float(rng.choice([0.0, 0.3, 0.9])),
                            p_nonm=float(rng.choice([0.0, 0.1, 1.0])), p_sm=float(rng.choice([0.0, 0.5, 1.0])), p_flagdrop=float(rng.choice([0.0, 0.02, 0.4])))
    if rng.random() < 0.4:
        arrs = synth.pile_indels(arrs, int(rng.integers(10, max(RL - 10, 11))), seed=seed, frac=float(rng.choice([0.2, 0.9])))
    regions = []
    for _ in range(int(rng.integers(1, 5))):
        a = int(rng.integers(0, RL)); b = a + int(rng.choice([0, 1, 2, 63, 64, 65, 200, RL]))
        regions.append((a, min(b, RL + 100)))
    kw = dict(min_mapq=int(rng.choice([0, 0, 1, 20, 60, 254, 255, 256])), min_bq=int(rng.choice([0, 0, 1, 2, 3, 13, 40, 62, 63, 64, 93, 255, 300])),
              insertion_centric=bool(rng.random() < 0.4))
    if rng.random() < 0.25:
        kw["max_cnt"] = int(rng.choice([1, 2, 5, 50, 1000]))
    if rng.random() < 0.5:
        kw.update(per_lib=True, lib_names=["lib%03d" % i for i in range(n_libs)])
    return kind, style, ref, arrs, regions, kw, bool(rng.random() < 0.5)


def api_routes(dev, oracle, seed, ref, arrs, regions, kw):
    """the first region of some size again through the other entry points: reads pushed in several batches, passes queued back to back
    (brc_compute_n), windows of the resident result (brc_fetch_window) against slices of the whole, announced windows
    (brc_region_windows) against the unhinted engine"""
    rr = np.random.default_rng(seed + 5)
    big = [r for r in regions if r[1] - r[0] >= 4]
    if not big:
        return
    b0, e0 = big[0]
    ends = capi.read_ends(arrs)
    idx = capi.fetch_overlapping(arrs, ends, b0 - 1, e0)
    sub = capi.select_reads(arrs, idx)
    want, _ = parity.run_engine(oracle, sub, [(b0, e0)], ref=ref, **kw)
    eng = capi.Engine(dev, **kw)
    eng.begin_region(0, b0, e0, ref)
    n = len(idx); cuts = sorted(set([0, n] + [int(x) for x in rr.integers(0, n + 1, int(rr.integers(0, 4)))]))
    for a, b in zip(cuts[:-1], cuts[1:]):
        eng.push_reads(capi.select_reads(sub, np.arange(a, b)))
    eng.upload(); eng.compute_n(int(rr.integers(1, 4)))
    whole = eng.fetch_result(); whole_text = eng.format_region("chrS"); eng.clear_indel_queue()
    assert whole_text == want, "reads pushed in %d batches: text differs from the oracle's" % (len(cuts) - 1)
    wc = sorted(set([b0, e0] + [int(x) for x in rr.integers(b0, e0 + 1, 4)]))
    joined = b""
    for wi, (wa, wb) in enumerate(zip(wc[:-1], wc[1:])):
        dev.lib.brc_set_option(eng.h, 6, 1 if wi else 0)          # BRC_OPT_CONTINUES_PREVIOUS
        w = eng.fetch_window(wa, wb); joined += eng.format_region("chrS")
        for x, y in zip(parity.slice_result(w, wa - 1, wb), parity.slice_result(whole, wa - 1, wb)):
            assert (x == y) if isinstance(x, list) else np.array_equal(x, y), "fetch_window [%d,%d) differs from the whole result" % (wa, wb)
    assert joined == whole_text, "windows formatted in order differ from the region's text"
    dev.lib.brc_set_option(eng.h, 6, 0); eng.close()
    wins = sorted((int(x), int(x) + int(rr.choice([1, 1, 2, 64, 70, 300]))) for x in rr.integers(b0, max(e0 - 1, b0 + 1), int(rr.integers(1, 9))))
    wins = [(x, min(y, e0)) for x, y in wins if x < e0]
    e1 = capi.Engine(dev, text_only=True, **kw); e2 = capi.Engine(dev, text_only=True, **kw)
    e1.begin_region(0, b0, e0, ref); e1.push_reads(sub); e1.end_region()
    e2.begin_region(0, b0, e0, ref); e2.push_reads(sub); e2.region_windows(np.array([w[0] for w in wins], np.int32), np.array([w[1] for w in wins], np.int32)); e2.end_region()
    for wa, wb in wins:
        assert e1.format_window("chrS", wa, wb, 0) == e2.format_window("chrS", wa, wb, 0), "announced window [%d,%d) differs" % (wa, wb)
    e1.close(); e2.close()


def mutate_fields(seed, arrs):
    """qualities at the edges of the event byte (0, 1, 2, 62, 63, 64, 93, 255; whole reads of 255 = a '*' quality string), NM / SM
    values at the edges of their types — in place, on a copy"""
    rng = np.random.default_rng(seed + 31)
    arrs = dict(arrs)
    if rng.random() < 0.6 and len(arrs["qual"]):
        q = arrs["qual"].copy(); p = float(rng.choice([0.001, 0.05, 0.5]))
        hit = rng.random(len(q)) < p
        q[hit] = rng.choice(np.array([0, 1, 2, 3, 61, 62, 63, 64, 93, 127, 128, 200, 254, 255], np.uint8), int(hit.sum()))
        if rng.random() < 0.3:
            for r in rng.integers(0, len(arrs["pos"]), max(1, len(arrs["pos"]) // 20)):
                o = int(arrs["qual_off"][r]); q[o:o + int(arrs["l_qseq"][r])] = 255
        arrs["qual"] = q
    n = len(arrs["pos"])
    if rng.random() < 0.3 and n:
        nm = arrs["nm"].copy(); idx = rng.integers(0, n, max(1, n // 5)); nm[idx] = rng.choice([0, 1, 255, 256, 65535, 70000, 2**31 - 1, -1, -5], len(idx)).astype(nm.dtype); arrs["nm"] = nm
        sm = arrs["sm"].copy(); idx = rng.integers(0, n, max(1, n // 5)); sm[idx] = rng.choice([0, 255, 256, 100000, 2**31 - 1, -1], len(idx)).astype(sm.dtype); arrs["sm"] = sm
    return arrs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=0); ap.add_argument("--count", type=int, default=200); ap.add_argument("--hip", action="store_true"); ap.add_argument("--verbose", action="store_true"); ap.add_argument("--sim", default=None, help="another build of the simulator library (a sanitizer build: run under LD_PRELOAD of libasan / libubsan)")
    ap.add_argument("--ref", action="store_true", help="also: the oracle's text against the reference's own sources compiled over the htslib shim (oracle/_ref; skips the deep piles, its std::map per position is slow there)")
    ap.add_argument("--lib", default=None, help="another build of the HIP library on the GPU box: bam_readcount_amd/csrc/libbrc_hip_checked.so, the bounds-checked instantiation "
                    "(make -C bam_readcount_amd/csrc checked) — an out-of-bounds device address fails the scenario with its fault record")
    a = ap.parse_args()
    dev = capi.Library(os.path.abspath(a.lib)) if a.lib else capi.load_product() if a.hip else capi.Library(a.sim or os.path.join(ROOT, "tests", "sim", "libbrc_sim.so"))
    oracle = capi.Library(os.path.join(ROOT, "oracle", "libbrc_oracle.so"))
    ref_lib = None
    if a.ref:
        from test_ref_compiled import REF_LIB
        ref_lib = capi.Library(REF_LIB)
    bad = 0; t0 = time.time(); ev = 0
    for seed in range(a.first, a.first + a.count):
        kind, style, ref, arrs, regions, kw, clear = scenario(seed)
        arrs = mutate_fields(seed, arrs)
        if a.verbose:
            print("seed %d kind %s style %s reads %d kw %r regions %r" % (seed, kind, style, len(arrs["pos"]), {k: v for k, v in kw.items() if k != "lib_names"}, regions), flush=True)
        try:
            check_warn = not (kw.get("per_lib") and any(int(l) < 0 for l in arrs["lib"]))
            _, res = parity.compare_libs(dev, oracle, arrs, regions, ref=ref, clear_queue=clear, check_warn=check_warn, **kw)
            ev += sum(r.n_events for r in res)
            for route in (dict(text_only=True), dict(device_text="chrS")):
                want, _ = parity.run_engine(oracle, arrs, regions, ref=ref, clear_queue=clear, **kw)
                got, _ = parity.run_engine(dev, arrs, regions, ref=ref, clear_queue=clear, **route, **kw)
                assert got == want, "text differs (route %r)" % (route,)
            api_routes(dev, oracle, seed, ref, arrs, regions, kw)
            if ref_lib is not None and kind != "deep":
                want, _ = parity.run_engine(ref_lib, arrs, regions, ref=ref, clear_queue=clear, **kw)
                got, _ = parity.run_engine(oracle, arrs, regions, ref=ref, clear_queue=clear, **kw)
                assert got == want, "the oracle's text differs from the reference-compiled library's"
        except Exception as ex:                                   # noqa: BLE001 — every failure is reported, the run goes on
            bad += 1
            print("FAIL seed %d kind %s style %s kw %r regions %r: %s: %s" % (seed, kind, style, {k: v for k, v in kw.items() if k != "lib_names"}, regions, type(ex).__name__, str(ex)[:300].replace("\n", " ")), flush=True)
    print("extreme fuzz: %d scenarios from seed %d, %d events, %d failures, %.1f s" % (a.count, a.first, ev, bad, time.time() - t0), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
