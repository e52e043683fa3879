"""Whole-region validation of a computed HIP region against the CPU oracle on all usable cores (bench / test infrastructure).

The region the benchmark times (BASELINE config 3: 50 Mbp, 1.5 G events; the config-5 shapes) is cut into abutting windows.
Every window is (a) computed by the C oracle as a stand-alone region — reads fetched the reference's way, [beg0 - 1, end),
bamreadcount.cpp:602 — in one of N worker processes, and (b) read back from the HIP engine's result of the TIMED region with
brc_fetch_window (no second computation).  Both sides reduce a window to the same digests: ncol, depth, the dense integer and
float planes bit for bit (the fp32 sums as their bit patterns), the sorted indel list, the printed text, the event count.
A window whose digests differ raises with its coordinates and the name of the differing part.

The workers are forked BEFORE the process touches the GPU (they inherit the generated reads copy-on-write; a process that has
initialised the HIP runtime must not fork) and sleep on a pipe until the timed region is over.
"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np
import xxhash

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CTX = {}       # what the forked workers inherit: ref, arrs, ends, names, opts, oracle path


def result_digests(parity, res, text, lo, hi):
    """digests of one window's result over the positions [lo, hi) (planes outside the result's own window are zeros)"""
    ncol, depth, istat, fstat, ind = parity.slice_result(res, lo, hi)
    d = {}
    for k, a in (("ncol", ncol), ("depth", depth), ("istat", istat), ("fstat", fstat)):
        d[k] = xxhash.xxh3_128(np.ascontiguousarray(a)).hexdigest()
    h = xxhash.xxh3_128()
    for x in ind:
        h.update(repr(x[:4]).encode()); h.update(x[4]); h.update(x[5])
    d["indel"] = h.hexdigest(); d["n_indel"] = len(ind)
    d["text"] = xxhash.xxh3_128(text).hexdigest(); d["text_bytes"] = int(len(text))
    d["events"] = int(res.n_events)
    d["lines"] = int(np.count_nonzero(np.asarray(text) == 10)) if len(text) else 0
    return d


def window_reads(capi, arrs, ends, pos64, max_span, a, b):
    """indices of the reads samfetch would hand over for the region [a - 1, b) (bamreadcount.cpp:602); pos64 is sorted"""
    lo = int(np.searchsorted(pos64, max(a - 1, 0) - max_span, side="left")); hi = int(np.searchsorted(pos64, b, side="left"))
    return lo + np.nonzero(ends[lo:hi] > max(a - 1, 0))[0]


def _worker(task_q, result_q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from bam_readcount_amd import capi
    import parity
    c = _CTX
    try:
        oracle = capi.Library(c["oracle"])
        oe = capi.Engine(oracle, lib_names=c["names"], **c["opts"])
        pos64 = c["pos64"]
        while True:
            t = task_q.get()
            if t is None:
                break
            a, b, with_text = t
            t0 = time.perf_counter()
            sub = capi.select_reads(c["arrs"], window_reads(capi, c["arrs"], c["ends"], pos64, c["max_span"], a, b))
            t1 = time.perf_counter()
            oe.begin_region(0, a, b, c["ref"]); oe.push_reads(sub); oe.upload(); oe.compute()
            t2 = time.perf_counter()
            res = oe.fetch_result(); text = oe.format_region_np("chrS") if with_text else np.zeros(0, np.uint8); oe.clear_indel_queue()
            t3 = time.perf_counter()
            d = result_digests(parity, res, text, max(a - 1, 0), b)
            d.update(window=(a, b), select_s=t1 - t0, pileup_s=t2 - t1, text_s=t3 - t2, digest_s=time.perf_counter() - t3)
            result_q.put(d)
        oe.close()
    except BaseException as ex:                      # noqa: BLE001 — reported to the parent, which raises
        result_q.put({"error": "%s: %s" % (type(ex).__name__, ex)})


class OraclePool:
    """N oracle processes, forked now, working later."""

    def __init__(self, nproc, ref, arrs, names, opts, capi):
        ends = capi.read_ends(arrs)
        pos64 = np.asarray(arrs["pos"]).astype(np.int64)
        _CTX.update(ref=ref, arrs=arrs, ends=ends, pos64=pos64, names=list(names), opts=dict(opts),
                    max_span=int((ends - pos64).max()) if len(pos64) else 0,
                    oracle=os.path.join(ROOT, "oracle", "libbrc_oracle.so"))
        self.ends, self.pos64, self.max_span = ends, pos64, _CTX["max_span"]
        ctx = mp.get_context("fork")
        self.task_q, self.result_q = ctx.Queue(), ctx.Queue()
        self.procs = [ctx.Process(target=_worker, args=(self.task_q, self.result_q), daemon=True) for _ in range(max(1, nproc))]
        for p in self.procs:
            p.start()
        self.n_tasks = 0

    def submit(self, windows):
        for w in windows:
            self.task_q.put(tuple(int(x) for x in w)); self.n_tasks += 1
        for _ in self.procs:
            self.task_q.put(None)

    def collect(self):
        out = {}
        import queue as _queue
        for _ in range(self.n_tasks):
            # a worker killed from outside (the OOM killer, a crash inside the oracle library) posts nothing: do not wait for it forever
            while True:
                try:
                    d = self.result_q.get(timeout=5.0); break
                except _queue.Empty:
                    dead = [(p.pid, p.exitcode) for p in self.procs if not p.is_alive() and p.exitcode not in (0, None)]
                    if dead:
                        self.close()
                        raise RuntimeError("oracle worker(s) died without a result (pid, exit code): %r" % (dead,))
                    if not any(p.is_alive() for p in self.procs) and self.result_q.empty():
                        self.close()
                        raise RuntimeError("every oracle worker has exited and %d window(s) are still missing" % (self.n_tasks - len(out)))
            if "error" in d:
                self.close()
                raise RuntimeError("oracle worker failed: " + d["error"])
            out[tuple(d["window"])] = d
        for p in self.procs:
            p.join(timeout=30)
        return out

    def close(self):
        for p in self.procs:
            if p.is_alive():
                p.terminate()


def windows_of(beg0, end, n_windows):
    cuts = np.unique(np.linspace(beg0, end, n_windows + 1).astype(np.int64))
    return [(int(a), int(b)) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]


def check_region(pool, eng, parity, windows, want_events=None, with_text=True):
    """eng: the HIP engine holding the computed (timed) region.  Returns the `validated` fields; raises on the first difference."""
    t0 = time.perf_counter()
    pool.submit([(a, b, int(with_text)) for a, b in windows])
    got = {}
    t_hip = 0.0
    for a, b in windows:                                 # (overlaps the oracle workers: the device side is a download and a digest)
        t1 = time.perf_counter()
        eng.clear_indel_queue()
        w = eng.fetch_window(a, b); text = eng.format_region_np("chrS") if with_text else np.zeros(0, np.uint8)
        got[(a, b)] = result_digests(parity, w, text, max(a - 1, 0), b)
        t_hip += time.perf_counter() - t1
    eng.clear_indel_queue()
    want = pool.collect()
    wall = time.perf_counter() - t0
    keys = ("ncol", "depth", "istat", "fstat", "indel", "n_indel", "text", "text_bytes", "events", "lines")
    for wdw in windows:
        for k in keys:
            if got[wdw][k] != want[wdw][k]:
                raise AssertionError("full-region validation: window [%d, %d): %s of the HIP engine and of the oracle differ (%r vs %r)" % (wdw[0], wdw[1], k, got[wdw][k], want[wdw][k]))
    events = sum(d["events"] for d in got.values())
    if want_events is not None and events != want_events:
        raise AssertionError("full-region validation: the windows hold %d events, the timed region %d" % (events, want_events))
    h = xxhash.xxh3_128()
    for wdw in windows:
        h.update(got[wdw]["text"].encode())
    pile = sum(d["pileup_s"] for d in want.values()); text_s = sum(d["text_s"] for d in want.values())
    return {"full_contig": True, "windows": len(windows), "events": int(events), "lines": int(sum(d["lines"] for d in got.values())),
            "text_bytes": int(sum(d["text_bytes"] for d in got.values())), "indel_buckets": int(sum(d["n_indel"] for d in got.values())),
            "planes_bit_exact": True, **({"text_byte_exact": True} if with_text else {"whole_region_text": "not formatted (planes, indel lists and event counts of every window are compared; the text of the prefix is: text_byte_exact)"}), "digest": "xxh3_128 per window of ncol / depth / dense istat / dense fstat bits / indel list / text",
            "text_digest_of_digests": h.hexdigest(), "seconds": round(wall, 2), "hip_side_seconds": round(t_hip, 2),
            "oracle_cpu_seconds": {"pileup": round(pile, 2), "text": round(text_s, 2), "select": round(sum(d["select_s"] for d in want.values()), 2),
                                   "digest": round(sum(d["digest_s"] for d in want.values()), 2)},
            "oracle_processes": len(pool.procs)}
