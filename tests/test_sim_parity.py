"""CPU-only checks of the DEVICE algorithm (bam_readcount_amd/csrc/brc_core.h run lane-by-lane by tests/sim)
and the host formatter against the oracle.  The parity tests proper live in test_gpu_parity.py, whose every body runs
twice — [sim] here on the CPU, [hip] on the GPU (conftest.py: dev_lib); this file keeps what only the simulator can do
(host formatter internals, staging refusals) and the shared FUZZ table."""
import os
import subprocess

import numpy as np
import pytest

from bam_readcount_amd import capi
from conftest import ROOT
import parity
import synth
from test_oracle_golden import CASES, golden, run_case


FUZZ = [
    dict(seed=1, style="simple", n=300, opts=dict()),
    dict(seed=2, style="indel", n=400, opts=dict()),
    dict(seed=3, style="wild", n=400, opts=dict()),
    dict(seed=4, style="wild", n=400, opts=dict(min_mapq=20, min_bq=13)),
    dict(seed=5, style="mixed", n=500, opts=dict(insertion_centric=True)),
    dict(seed=6, style="wild", n=400, opts=dict(per_lib=True), n_libs=3),
    dict(seed=7, style="wild", n=400, opts=dict(per_lib=True, insertion_centric=True, min_mapq=10, min_bq=5), n_libs=4, p_nolib=0.01),
    dict(seed=8, style="mixed", n=600, opts=dict(per_lib=True), n_libs=2, p_nolib=0.2),
    dict(seed=9, style="wild", n=300, opts=dict(min_bq=30), weird=0.1),
    dict(seed=10, style="indel", n=50, opts=dict()),
]

def test_sim_without_reference(sim_lib, oracle_lib):
    rng = np.random.default_rng(77)
    ref = synth.make_ref(rng, 1000)
    arrs = synth.make_batch(78, ref, 150, style="indel")
    parity.compare_libs(sim_lib, oracle_lib, arrs, [(0, 1000)], ref=None)


def test_fmt_f2_matches_printf(sim_lib):
    import ctypes as C
    lib = C.CDLL(sim_lib.path)
    f = getattr(lib, "_ZN3brc6fmt_f2EPcf")
    f.argtypes = [C.c_char_p, C.c_float]; f.restype = C.c_int
    rng = np.random.default_rng(3)
    vals = np.concatenate([rng.random(20000).astype(np.float32) * np.float32(10.0) ** rng.integers(-4, 6, 20000),
                           np.array([0.0, -0.0, 0.005, 0.015, 0.025, 0.125, 0.375, 2.675, 1e9, 237.765, -0.001, -1.005, 0.994999,
                                     np.inf, -np.inf, np.nan, 3.4e38, 1e-30, 52.805, 0.495, 0.505], np.float32),
                           (np.arange(0, 4000, dtype=np.float32) * np.float32(0.005))]).astype(np.float32)
    buf = C.create_string_buffer(128)
    for v in vals:
        n = f(buf, C.c_float(float(v)))
        assert buf.raw[:n].decode() == "%.2f" % float(np.float32(v)), float(v)


def test_threaded_formatter_chunk_boundaries(sim_lib, oracle_lib, monkeypatch):
    """brc_format_region formats chunks of positions in parallel and seeds each chunk's deletion queue from the
    position before it; tiny chunks put a boundary next to every deletion."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synthgen
    ref, arrs = synthgen.generate(40_000, "tumor200x", seed=11, n_chunks=8)
    names = ["libA", "libB", "libC", "libD"]
    want, _ = parity.run_engine(oracle_lib, arrs, [(0, 40_000), (5000, 5001), (77, 30000)], ref=ref, lib_names=names, per_lib=True, clear_queue=False)
    assert want.count(b"\t-") > 1000          # plenty of deletions in the text
    for chunk, threads in (("7", "8"), ("1", "3"), ("1000", "1")):
        monkeypatch.setenv("BRC_FORMAT_CHUNK", chunk); monkeypatch.setenv("BRC_FORMAT_THREADS", threads)
        got, _ = parity.run_engine(sim_lib, arrs, [(0, 40_000), (5000, 5001), (77, 30000)], ref=ref, lib_names=names, per_lib=True, clear_queue=False)
        assert got == want, (chunk, threads)




@pytest.mark.parametrize("case", [FUZZ[2], FUZZ[4], FUZZ[6], FUZZ[7]], ids=lambda c: "seed%d-%s" % (c["seed"], c["style"]))
def test_sim_text_only_engine_prints_the_same(sim_lib, oracle_lib, case):
    """BRC_OPT_TEXT_ONLY (what the drop-in command line sets): the formatter reads the compact result — two bucket slots per
    position plus the sparse third-allele table — instead of dense planes; the text must not change.  High depth so that
    third and fourth alleles occur at many positions."""
    rng = np.random.default_rng(case["seed"])
    ref = synth.make_ref(rng, 1200, weird=case.get("weird", 0.0))
    n_libs = case.get("n_libs", 1)
    arrs = synth.make_batch(case["seed"] + 300, ref, 1500, style=case["style"], n_libs=n_libs, p_nolib=case.get("p_nolib", 0.0))
    names = ["lib%c" % (65 + i) for i in range(n_libs)] if case["opts"].get("per_lib") else ()
    regions = [(0, 1200), (300, 301), (700, 900)]
    want, _ = parity.run_engine(oracle_lib, arrs, regions, ref=ref, lib_names=names, **case["opts"])
    got, res = parity.run_engine(sim_lib, arrs, regions, ref=ref, lib_names=names, text_only=True, **case["opts"])
    assert got == want
    assert not res[0].istat.any()                     # no dense planes were built
    os.environ["BRC_FORMAT_CHUNK"] = "64"; os.environ["BRC_FORMAT_THREADS"] = "5"
    try:
        got2, _ = parity.run_engine(sim_lib, arrs, regions, ref=ref, lib_names=names, text_only=True, **case["opts"])
    finally:
        del os.environ["BRC_FORMAT_CHUNK"]; del os.environ["BRC_FORMAT_THREADS"]
    assert got2 == want
    # BRC_OPT_DEVICE_TEXT: the lines are written by the device code (text_line) and the host only rewrites the lines with
    # indel buckets, queued deletions or a third base
    got3, res3 = parity.run_engine(sim_lib, arrs, regions, ref=ref, lib_names=names, device_text="chrS", **case["opts"])
    assert got3 == want
    assert not res3[0].ncol.any()                     # no planes at all came to the host


def test_sim_deep_indel_key_equals_oracle(sim_lib, oracle_lib):
    """Amplicon-like pile: ~500 reads share one deletion / insertion position — the keyed indel reduction sorts the events of a
    key by read index (heapsort above 48 events) before folding them in pileup-column order."""
    rng = np.random.default_rng(41)
    ref = synth.make_ref(rng, 600)
    arrs = synth.pile_indels(synth.make_batch(141, ref, 700, style="simple", region=(215, 262), read_len=(80, 100), n_libs=2), 270, seed=3)
    assert int((arrs["n_cigar"] == 3).sum()) > 400
    for kw in (dict(), dict(per_lib=True, insertion_centric=True, lib_names=["libA", "libB"])):
        text, res = parity.compare_libs(sim_lib, oracle_lib, arrs, [(0, 600), (271, 272)], ref=ref, **kw)
        assert max(int(d["i"][0]) for d in res[0].indels) > 100


def test_large_batches_are_staged_on_several_threads_like_on_one(sim_lib, oracle_lib):
    """brc_push_reads takes batches of 32 768 reads and more through a pool of threads (per-chunk sums, a scan, offsets): the
    staging — rows of the event-byte stream, indel slots, pieces, extent, histogram — and the first error in file order must be
    those of the one-thread pass (BRC_OPT_FORMAT_THREADS = 1 caps the pool at one thread: the serial pass)."""
    import sys
    from bam_readcount_amd import capi
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synthgen
    import parity
    ref, arrs = synthgen.generate(120_000, "tumor200x", seed=21, n_chunks=4)          # 160 000 reads, 10 % with an indel
    assert len(arrs["pos"]) >= 100_000
    names = ["lib%d" % i for i in range(4)]

    def run(threads, batch):
        eng = capi.Engine(sim_lib, per_lib=True, insertion_centric=True, lib_names=names)
        if threads:
            sim_lib.lib.brc_set_option(eng.h, 8, threads)
        try:
            eng.begin_region(0, 0, 120_000, ref); eng.push_reads(batch)
            r = eng.end_region(); return r, eng.format_region("chrS")
        finally:
            eng.close()
    r1, t1 = run(1, arrs); r8, t8 = run(0, arrs)
    parity.assert_results_equal(r8, r1, "pooled vs one-thread staging"); assert t8 == t1
    # two batches: the second continues the first one's running sums
    half = len(arrs["pos"]) // 2
    eng = capi.Engine(sim_lib, per_lib=True, insertion_centric=True, lib_names=names)
    eng.begin_region(0, 0, 120_000, ref); eng.push_reads(capi.select_reads(arrs, np.arange(half))); eng.push_reads(capi.select_reads(arrs, np.arange(half, len(arrs["pos"]))))
    parity.assert_results_equal(eng.end_region(), r1, "two pooled batches"); assert eng.format_region("chrS") == t1; eng.close()
    # the first bad record in file order decides the error, whichever chunk meets one first
    for where, field, what in ((90_001, "l_qseq", "CIGAR and sequence length disagree"), (40_000, "pos", "not coordinate-sorted"), (150_000, "lib", "library index out of range")):
        bad = {k: (v.copy() if v is not None else None) for k, v in arrs.items()}
        bad["l_qseq"][120_000] += 1                                             # a later bad record that must not win
        if field == "l_qseq": bad["l_qseq"][where] += 2
        elif field == "pos": bad["pos"][where] = bad["pos"][where - 1] - 5
        else: bad["lib"][where] = 9; bad["l_qseq"][120_000] -= 1
        for threads in (1, 0):
            eng = capi.Engine(sim_lib, per_lib=True, insertion_centric=True, lib_names=names)
            if threads: sim_lib.lib.brc_set_option(eng.h, 8, threads)
            eng.begin_region(0, 0, 120_000, ref)
            with pytest.raises(capi.BrcError) as ei:
                eng.push_reads(bad)
            assert what in str(ei.value), (where, threads, str(ei.value))
            eng.close()


def test_push_reads_refuses_inconsistent_records(sim_lib):
    """The staging code is the last stop before the kernels index a read's rows: a CIGAR that walks more or fewer query bases
    than the record has, offsets outside the arenas and unsorted reads are errors, not work."""
    from bam_readcount_amd import capi
    rng = np.random.default_rng(1)
    ref = synth.make_ref(rng, 600)
    good = synth.make_batch(3, ref, 20, style="indel")

    def refused(arrs, what):
        eng = capi.Engine(sim_lib)
        eng.begin_region(0, 0, 500, ref)
        with pytest.raises(capi.BrcError) as ei:
            eng.push_reads(arrs)
        assert what in str(ei.value), str(ei.value)
        # the refused batch abandons the region (its staging is half appended): a later batch is not taken on top of it ...
        with pytest.raises(capi.BrcError) as ei:
            eng.push_reads(good)
        assert "outside an open region" in str(ei.value)
        # ... and the next region of the same engine starts clean
        eng.begin_region(0, 0, 500, ref); eng.push_reads(good); again = eng.end_region()
        assert again.n_events == clean.n_events and np.array_equal(again.istat, clean.istat)
        eng.close()
    eng = capi.Engine(sim_lib)
    eng.begin_region(0, 0, 500, ref); eng.push_reads(good); clean = eng.end_region(); eng.close()
    bad = {k: v.copy() for k, v in good.items()}
    bad["l_qseq"][5] += 3
    refused(bad, "CIGAR and sequence length disagree")
    bad = {k: v.copy() for k, v in good.items()}
    bad["cigar"][int(bad["cigar_off"][7])] += 2 << 4
    refused(bad, "CIGAR and sequence length disagree")
    # an M operator of length zero behind the read's last base (the reference reads past the read's qualities there): the one empty operator
    # that stays refused — every other one is piled up by the iterator's own cursor (test_empty_m_operators_*)
    bad = {k: v.copy() for k, v in good.items()}
    i6 = int(np.flatnonzero((good["n_cigar"] == 1) & ((good["flag"] & 4) == 0) & (good["mapq"] >= 0))[0]); c0 = int(bad["cigar_off"][i6])
    tail = np.array([(1 << 4) | 2, 0, (1 << 4) | 2], np.uint32)                 # ... 1D 0M 1D behind the read's only match
    bad["cigar"] = np.concatenate([bad["cigar"][:c0 + 1], tail, bad["cigar"][c0 + 1:]]); bad["n_cigar"][i6] = 4
    bad["cigar_off"] = np.where(np.arange(len(bad["pos"])) > i6, bad["cigar_off"] + 3, bad["cigar_off"]).astype(np.uint64)
    refused(bad, "empty M/=/X CIGAR operator behind its last base")
    bad = {k: v.copy() for k, v in good.items()}
    bad["pos"][9] = bad["pos"][3] - 1 if bad["pos"][3] > 0 else 0; bad["pos"][10] = bad["pos"][9] - 1 if bad["pos"][9] > 0 else -1
    refused(bad, "coordinate-sorted")
    bad = {k: v.copy() for k, v in good.items()}
    bad["qual_off"][19] = np.uint64(len(bad["qual"]))
    refused(bad, "outside the batch arenas")
    bad = {k: v.copy() for k, v in good.items()}
    bad["l_qseq"][4] = 0; bad["flag"][4] = 16
    refused(bad, "without sequence")
    # ... but not when the record is unmapped: it never reaches a column, whatever CIGAR an aligner left on it
    ok = {k: v.copy() for k, v in good.items()}
    ok["cigar"][int(ok["cigar_off"][7])] += 2 << 4; ok["flag"][7] |= 4
    eng = capi.Engine(sim_lib)
    eng.begin_region(0, 0, 500, ref); eng.push_reads(ok); eng.end_region(); eng.close()


def test_library_names_must_come_in_the_reference_s_order(sim_lib):
    """brc.h: library names bytewise-sorted and distinct — the order of the reference's std::map (bamreadcount.cpp:273), in which
    it prints them (:360).  "lib10" sorts before "lib2": a caller numbering its libraries would print them in another order than
    the reference; brc_create refuses such a list instead (found when tools/fuzz/extreme.py compared 254 numbered libraries with
    the reference-compiled library)."""
    import pytest
    from bam_readcount_amd import capi
    capi.Engine(sim_lib, per_lib=True, lib_names=["lib10", "lib2"]).close()
    capi.Engine(sim_lib, per_lib=True, lib_names=["", "A", "a"]).close()
    for names in (["lib2", "lib10"], ["b", "a"], ["same", "same"]):
        with pytest.raises(capi.BrcError):
            capi.Engine(sim_lib, per_lib=True, lib_names=names)


@pytest.mark.parametrize("style,opts", [("indel", dict()), ("indel", dict(insertion_centric=True, min_mapq=10, min_bq=8)), ("mixed", dict(per_lib=True)), ("many", dict(insertion_centric=True))])
def test_empty_m_operators_are_piled_up_like_the_iterator_does(sim_lib, oracle_lib, style, opts):
    """M / = / X operators of length zero (round 6; refused until then): htslib's cursor steps onto such an operator for one column — the
    column is a match at the operator's query offset, a deletion behind it is seen one column late and never announced.  The device
    algorithm (walk_pieces_cursor / enumerate_indels_cursor, one lane per such read) against the oracle's stateful cursor: planes, indel
    lists, text, warnings."""
    rng = np.random.default_rng(5)
    ref = synth.make_ref(rng, 3000, weird=0.01)
    names = ["libA", "libB"] if opts.get("per_lib") else ()
    base = synth.make_batch(21, ref, 400, read_len=(40, 160), style=style, n_libs=max(len(names), 1))
    arrs = synth.inject_empty_mops(base, seed=3, frac=0.6)
    assert int(arrs["n_cigar"].sum()) > int(base["n_cigar"].sum()) + 100
    parity.compare_libs(sim_lib, oracle_lib, arrs, [(0, 3000), (700, 900), (1500, 1501)], ref=ref, lib_names=names, **opts)
    # ... and through the text routes of the engine (compact planes, device-side text)
    want, _ = parity.run_engine(oracle_lib, arrs, [(0, 3000)], ref=ref, lib_names=names, **opts)
    for route in (dict(text_only=True), dict(device_text="chrS")):
        got, _ = parity.run_engine(sim_lib, arrs, [(0, 3000)], ref=ref, lib_names=names, **dict(opts, **route))
        assert got == want, route


@pytest.mark.parametrize("n_reads", [1500, 9000])
def test_host_scan_for_escape_bases_is_k1_s_predicate(sim_lib, oracle_lib, n_reads):
    """The wide stream is sparse: only the reads brc_push_reads finds an escape base in (eight qualities / sixteen base codes
    per step, tails byte by byte) get a wide row, and K1 must mark exactly those reads' pieces PF_WIDE — the CPU twin compares the two
    for every read.  Reads of 1..70 bases with ONE special byte each: every quality next to the event byte's range (0, 1, 62, 63,
    64, 127, 128, 255), every base code 0..15, at any offset — word boundaries, the last base of reads of odd length.  9000 reads:
    the batch is staged by the pool (one thread below 8192 reads)."""
    rng = np.random.default_rng(97)
    ref = synth.make_ref(rng, 3000)
    arrs = synth.make_batch(197, ref, n_reads, style="simple", read_len=(1, 70), mismatch=0.02, p_iupac_read=0.0, p_q2tail=0.0)
    n_wide = synth.one_special_byte_per_read(arrs, rng)
    assert n_wide > n_reads // 8
    parity.compare_libs(sim_lib, oracle_lib, arrs, [(0, 3000)], ref=ref, min_bq=0)
    parity.compare_libs(sim_lib, oracle_lib, arrs, [(0, 3000)], ref=None, min_bq=20, insertion_centric=True)
    if n_reads < 8192:            # library-major rows (-p): wide reads of three libraries
        arrs3 = synth.make_batch(198, ref, n_reads, style="mixed", read_len=(1, 70), n_libs=3, p_iupac_read=0.0)
        synth.one_special_byte_per_read(arrs3, rng)
        parity.compare_libs(sim_lib, oracle_lib, arrs3, [(0, 3000), (1000, 1100)], ref=ref, min_bq=13, lib_names=["libA", "libB", "libC"], per_lib=True)
