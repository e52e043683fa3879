"""Parity tests proper: the HIP engine (through the C-ABI) against the reference goldens and the CPU oracle.
Integer planes bit-exact, fp32 planes bit-exact (order-preserving sums; the north_star tolerance is 1e-6 relative),
text byte-exact.

Every test body takes `dev_lib` (conftest.py), which is parametrised over two libraries exporting the same C-ABI:
  [hip]  the product library on the GPU                                     — marked gpu: the parity tests proper
  [sim]  the CPU lane simulator of the same device functions (tests/sim)     — runs in the `-m "not gpu"` suite
so every assertion of this file — literals included — is also executed on the CPU box before it reaches the GPU one."""
import os

import numpy as np
import pytest

from bam_readcount_amd import capi
import parity
import synth
from test_oracle_golden import CASES, golden, run_case
from test_sim_parity import FUZZ
from synth import MANY_OPS



@pytest.mark.parametrize("name,opts,bad_rg", CASES)
def test_hip_matches_reference_goldens(dev_lib, test_bam, name, opts, bad_rg):
    text, _ = run_case(dev_lib, test_bam, opts, bad_rg)
    assert text == golden(name)


def test_hip_regions_on_cmdline(dev_lib, test_bam):
    text, _ = run_case(dev_lib, test_bam, dict(per_lib=False, insertion_centric=False), False, site_mode=False)
    assert text == golden("expected_all_lib")


def test_hip_full_window_of_test_bam_equals_oracle(dev_lib, oracle_lib, test_bam):
    names = [str(s) for s in test_bam["lib_names"]]
    for per_lib in (False, True):
        for ic in (False, True):
            text, _ = parity.compare_libs(dev_lib, oracle_lib, test_bam, [(10402736, 10405248)], tid=20, chrom="21",
                                          ref=test_bam["ref"], lib_names=names if per_lib else (), per_lib=per_lib,
                                          insertion_centric=ic)
            # 21:10402737-10405248 -> 796 emitted lines, sum of depths 96243
            lines = text.decode().splitlines()
            assert len(lines) == 796
            assert sum(int(l.split("\t")[3]) for l in lines) == 96243


def test_hip_twolib_cram_fixture(dev_lib, oracle_lib, twolib):
    # BASELINE config 2(ii): twolib.sorted.cram -p -i, site list "rand1k 50 60" -> 11 lines, GPU vs oracle (no reference golden exists)
    names = [str(s) for s in twolib["lib_names"]]
    text, res = parity.compare_libs(dev_lib, oracle_lib, twolib, [(49, 60)], tid=0, chrom="rand1k", ref=twolib["ref"],
                                    lib_names=names, per_lib=True, insertion_centric=True, ref_len_check=True)
    lines = text.decode().splitlines()
    assert len(lines) == 11
    assert lines[0].startswith("rand1k\t50\tA\t1\treads1_lb\t{") and lines[0].endswith("\t}")
    assert "A:1:60.00:255.00:60.00:1:0:0.37:0.00:0.00:1:0.15:60.00:0.15" in lines[0]
    # NM-missing warnings (BasicStat.cpp:100): one per counted event of a read without NM — the 11 reported positions plus
    # the lead position beg-1, which pileup_func also processes (bamreadcount.cpp:269).  The fixture carries the NM that
    # htslib's CRAM decoder regenerates (tools/make_fixtures.py), so the expected count follows the fixture's tag bits.
    has_nm = bool(twolib["tags"][0] & 1)
    assert res[0].warn[1] == (0 if has_nm else 12)


@pytest.mark.parametrize("case", FUZZ, ids=lambda c: "seed%d-%s" % (c["seed"], c["style"]))
def test_hip_fuzz_equals_oracle(dev_lib, oracle_lib, case):
    rng = np.random.default_rng(case["seed"])
    ref = synth.make_ref(rng, 3000, weird=case.get("weird", 0.0))
    n_libs = case.get("n_libs", 1)
    arrs = synth.make_batch(case["seed"] + 100, ref, case["n"], style=case["style"], n_libs=n_libs, p_nolib=case.get("p_nolib", 0.0))
    names = ["lib%c" % (65 + i) for i in range(n_libs)] if case["opts"].get("per_lib") else ()
    regions = [(0, 3000), (100, 101), (700, 1500), (2990, 3200), (1500, 1500)]
    nolib = case.get("p_nolib", 0.0) > 0
    parity.compare_libs(dev_lib, oracle_lib, arrs, regions, ref=ref, lib_names=names, check_warn=not nolib, **case["opts"])


@pytest.mark.parametrize("mismatch,read_len,style", [(0.3, (1, 700), "mixed"), (0.95, (200, 600), "simple"), (0.6, (1, 40), "indel"),
                                                      (1.0, (64, 64), "simple"), (0.5, (500, 900), "wild"), (0.05, (3000, 6000), "indel")])
def test_hip_annotate_mismatch_runs_stress(dev_lib, oracle_lib, mismatch, read_len, style):
    """K1's group form: mismatch runs that cross 8-base groups, 64-group passes and whole groups, Q2 tails, reads shorter
    than a group, reads with more than two M operators (serial path) — Zm sums / Q2 / three-prime must stay exact."""
    rng = np.random.default_rng(int(mismatch * 100) + read_len[1])
    ref = synth.make_ref(rng, 8000, weird=0.01)
    long_reads = read_len[0] >= 3000          # long reads: few of them, all inside the reference
    arrs = synth.make_batch(77 + read_len[0], ref, 60 if long_reads else 400, read_len=read_len, style=style, mismatch=mismatch, p_q2tail=0.5,
                            region=(0, 1900) if long_reads else (0, 7000))
    parity.compare_libs(dev_lib, oracle_lib, arrs, [(0, 8000), (3000, 3100)], ref=ref)
    parity.compare_libs(dev_lib, oracle_lib, arrs, [(0, 8000)], ref=ref, min_mapq=10, min_bq=15, insertion_centric=True)


KNOBS = [{"BRC_NO_TABLE": "1"}, {"BRC_FLUSH_K": "3"}, {"BRC_PACK_LIM": "255", "BRC_FLUSH_K": "9"}, {"BRC_FORCE_DOM": "0"},
         {"BRC_FORCE_DOM": "3", "BRC_FLUSH_K": "1"}, {"BRC_FORCE_DOM": "5", "BRC_PACK_LIM": "255"},
         {"BRC_FORCE_DOM": "3", "BRC_XEV_CAP": "1"}]          # one entry per third-allele sub-list: grow and compute again


@pytest.mark.parametrize("env", KNOBS, ids=lambda e: "+".join("%s=%s" % kv for kv in e.items()))
def test_hip_rare_device_paths(knob_lib, oracle_lib, monkeypatch, env):
    """Paths of k_pileup2 that ordinary data takes for a few events only, forced for all of them by test knobs: event terms by
    exact reciprocal division (BRC_NO_TABLE), flushes of the packed integer registers every K pieces (BRC_FLUSH_K), PF_HUGE
    pieces whose integers are drained (BRC_PACK_LIM), third-allele queue + live planes at the final store (BRC_FORCE_DOM
    makes every lane treat one bucket as dominant).  Same bits as the oracle in every case."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(99)
    ref = synth.make_ref(rng, 3000, weird=0.01)
    arrs = synth.make_batch(199, ref, 700, style="mixed", n_libs=3, p_nolib=0.02)
    names = ["libA", "libB", "libC"]
    parity.compare_libs(knob_lib, oracle_lib, arrs, [(0, 3000), (1200, 1300)], ref=ref, lib_names=names, per_lib=True, check_warn=False)
    parity.compare_libs(knob_lib, oracle_lib, arrs, [(0, 3000)], ref=ref, min_mapq=5, min_bq=10, insertion_centric=True)
    deep = synth.make_batch(299, ref, 2500, style="mixed", region=(900, 1400), read_len=(100, 150))       # ~600x: many half-batches per tile
    parity.compare_libs(knob_lib, oracle_lib, deep, [(800, 1600)], ref=ref)


def test_product_ignores_ablation_environment(dev_lib, oracle_lib, monkeypatch):
    """An inherited BRC_PILEUP_VARIANT / BRC_ANN_VARIANT (the profiling ablations of experiment builds: kernels with parts
    switched off) must not change what the product computes: same bits as the oracle with every one of them set."""
    rng = np.random.default_rng(41)
    ref = synth.make_ref(rng, 4000, weird=0.01)
    arrs = synth.make_batch(241, ref, 900, style="mixed")
    for pv, av in (("4", "3"), ("1", "1"), ("6", "5"), ("11", "2")):
        monkeypatch.setenv("BRC_PILEUP_VARIANT", pv); monkeypatch.setenv("BRC_ANN_VARIANT", av)
        monkeypatch.setenv("BRC_IBUCKET_SHIFT", "40")          # (an unsupported bucket size is ignored, not used as a shift count)
        parity.compare_libs(dev_lib, oracle_lib, arrs, [(0, 4000), (1000, 1100)], ref=ref, min_mapq=5, min_bq=10)


@pytest.mark.parametrize("per_lib", [False, True])
def test_hip_fetch_window_equals_the_stand_alone_region(dev_lib, oracle_lib, per_lib):
    """brc_fetch_window: windows of a computed region come back as the stand-alone regions would (oracle: one region per window,
    reads fetched the reference's way), planes bit for bit and — formatted after a queue reset — byte for byte; windows formatted
    in order without resets print the whole region's text; a whole-region fetch stays valid beside them."""
    rng = np.random.default_rng(23)
    ref = synth.make_ref(rng, 6000, weird=0.01)
    arrs = synth.make_batch(523, ref, 2500, style="mixed", n_libs=3 if per_lib else 1, p_nolib=0.01 if per_lib else 0.0, mismatch=0.05)
    names = ["libA", "libB", "libC"] if per_lib else ()
    opts = dict(per_lib=per_lib, insertion_centric=per_lib, min_mapq=5, min_bq=10)
    eng = capi.Engine(dev_lib, lib_names=names, **opts)
    oe = capi.Engine(oracle_lib, lib_names=names, **opts)
    ends = capi.read_ends(arrs)
    eng.begin_region(0, 100, 5900, ref); eng.push_reads(capi.select_reads(arrs, capi.fetch_overlapping(arrs, ends, 99, 5900)))
    eng.upload(); eng.compute()
    whole = eng.fetch_result(); whole_text = eng.format_region("chrS"); eng.clear_indel_queue()
    cuts = [100, 101, 164, 1000, 1001, 2500, 4097, 5900]
    in_order = b""
    for wi, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        dev_lib.lib.brc_set_option(eng.h, 6, 1 if wi else 0)       # BRC_OPT_CONTINUES_PREVIOUS: a window's lead position was the last one of the window before
        w = eng.fetch_window(a, b)
        in_order += eng.format_region("chrS")
        assert (w.beg0, w.end, w.pos0) == (a, b, max(a - 1, whole.pos0)) and w.pos0 + w.n_pos == min(b, whole.pos0 + whole.n_pos)
        for x, y in zip(parity.slice_result(w, a - 1, b), parity.slice_result(whole, a - 1, b)):
            assert (x == y) if isinstance(x, list) else np.array_equal(x, y)
    assert in_order == whole_text
    dev_lib.lib.brc_set_option(eng.h, 6, 0)
    assert eng.fetch_window(5800, 5900).n_pos >= 0                     # (a window that may lie behind the reads' extent: empty, not an error)
    eng.clear_indel_queue()
    for a, b in zip(cuts[:-1], cuts[1:]):
        oe.begin_region(0, a, b, ref); oe.push_reads(capi.select_reads(arrs, capi.fetch_overlapping(arrs, ends, a - 1, b)))
        want = oe.end_region(); want_text = oe.format_region("chrS"); oe.clear_indel_queue()
        got = eng.fetch_window(a, b); got_text = eng.format_region("chrS"); eng.clear_indel_queue()
        for x, y in zip(parity.slice_result(got, a - 1, b), parity.slice_result(want, a - 1, b)):
            assert (x == y) if isinstance(x, list) else np.array_equal(x, y)
        assert got_text == want_text and got.n_events == want.n_events
        if per_lib:
            g_un = np.full(b - a + 1, 0xFFFFFFFF, np.uint32); w_un = g_un.copy()
            g_un[got.pos0 - (a - 1):got.pos0 - (a - 1) + got.n_pos] = got.unavail != 0xFFFFFFFF; w_un[want.pos0 - (a - 1):want.pos0 - (a - 1) + want.n_pos] = want.unavail != 0xFFFFFFFF
            assert np.array_equal(g_un, w_un)
    # argument handling
    with pytest.raises(capi.BrcError):
        eng.fetch_window(50, 200)
    with pytest.raises(capi.BrcError):
        eng.fetch_window(200, 6000)
    eng.close(); oe.close()


def test_hip_passes_back_to_back_leave_the_result_of_one(knob_lib, oracle_lib, monkeypatch):
    """brc_compute_n: passes queued without a host wait in between (more of them than the engine has event sets; with a tiny
    third-allele list: the grow-and-repeat path inside a batch) leave exactly the result of one brc_compute."""
    rng = np.random.default_rng(61)
    ref = synth.make_ref(rng, 4000, weird=0.01)
    arrs = synth.make_batch(261, ref, 1200, style="mixed", mismatch=0.2)
    want_t, want_r = parity.run_engine(oracle_lib, arrs, [(0, 4000)], ref=ref, min_bq=10)
    for env in ({}, {"BRC_XEV_CAP": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = capi.Engine(knob_lib, min_bq=10)
        eng.begin_region(0, 0, 4000, ref); eng.push_reads(arrs); eng.upload()
        ms, tot = eng.compute_n(70)
        assert len(ms) == 8 and tot >= 0
        got = eng.fetch_result()
        parity.assert_results_equal(got, want_r[0], "after 70 passes")
        assert eng.format_region("chrS") == want_t
        eng.compute_n(1); parity.assert_results_equal(eng.fetch_result(), want_r[0], "after one more")
        with pytest.raises(capi.BrcError):
            eng.compute_n(0)
        eng.close()


@pytest.mark.parametrize("min_bq", [0, 13, 63, 64, 200])
def test_hip_escape_bytes_and_the_wide_stream(dev_lib, oracle_lib, min_bq):
    """The event byte holds qualities 1..62 and buckets nameable relative to the reference base's; everything else is an escape
    byte + a word in the wide stream + PF_WIDE pieces: qualities 0 and 63..255, N / '=' bases off the reference's bucket, any
    mismatch over a reference base that is not A C G T (those events go to the third-allele list).  Base-quality thresholds
    below, at and above the byte's range."""
    rng = np.random.default_rng(71)
    ref = synth.make_ref(rng, 4000, weird=0.03)
    arrs = synth.make_batch(271, ref, 1500, style="mixed", mismatch=0.05, n_libs=2)
    q = arrs["qual"].copy(); r = np.random.default_rng(5)
    pick = r.random(q.size)
    q[pick < 0.02] = 0; q[(pick >= 0.02) & (pick < 0.04)] = 63; q[(pick >= 0.04) & (pick < 0.06)] = r.integers(64, 256, int(((pick >= 0.04) & (pick < 0.06)).sum())); q[(pick >= 0.06) & (pick < 0.07)] = 62; q[(pick >= 0.07) & (pick < 0.08)] = 1
    arrs["qual"] = q
    s4 = arrs["seq4"].copy(); pk = r.random(s4.size)
    s4[pk < 0.01] = (s4[pk < 0.01] & 0x0f) | 0xf0            # N in the high nibble
    s4[(pk >= 0.01) & (pk < 0.02)] &= 0xf0                  # '=' in the low nibble
    s4[(pk >= 0.02) & (pk < 0.025)] = 0x5a                  # IUPAC codes (bucket N)
    arrs["seq4"] = s4
    names = ["libA", "libB"]
    parity.compare_libs(dev_lib, oracle_lib, arrs, [(0, 4000), (2000, 2100)], ref=ref, min_bq=min_bq, lib_names=names, per_lib=True, insertion_centric=True)
    parity.compare_libs(dev_lib, oracle_lib, arrs, [(0, 4000)], ref=ref, min_bq=min_bq, min_mapq=3)
    parity.compare_libs(dev_lib, oracle_lib, arrs, [(0, 4000)], ref=None, min_bq=min_bq)


@pytest.mark.gpu
@pytest.mark.parametrize("n_reads", [1500, 9000])
def test_hip_sparse_wide_stream_rows_of_exactly_the_reads_with_an_escape(dev_lib, oracle_lib, n_reads):
    """The wide stream has rows only for the reads brc_push_reads found an escape base in (k_wide_rows writes them, K1 marks the
    pieces PF_WIDE from its own look at the bases): one special byte per read at any offset — tests/test_sim_parity.py's case on the
    device; a read the host missed would read words nobody wrote."""
    rng = np.random.default_rng(97)
    ref = synth.make_ref(rng, 3000)
    arrs = synth.make_batch(197, ref, n_reads, style="simple", read_len=(1, 70), mismatch=0.02, p_iupac_read=0.0, p_q2tail=0.0)
    assert synth.one_special_byte_per_read(arrs, rng) > n_reads // 8
    parity.compare_libs(dev_lib, oracle_lib, arrs, [(0, 3000)], ref=ref, min_bq=0)
    parity.compare_libs(dev_lib, oracle_lib, arrs, [(0, 3000)], ref=None, min_bq=20, insertion_centric=True)
    names = ["libA", "libB", "libC"]
    arrs3 = synth.make_batch(198, ref, n_reads, style="mixed", read_len=(1, 70), n_libs=3, p_iupac_read=0.0)
    synth.one_special_byte_per_read(arrs3, rng)
    parity.compare_libs(dev_lib, oracle_lib, arrs3, [(0, 3000), (1000, 1100)], ref=ref, min_bq=13, lib_names=names, per_lib=True)


def shuffled_arenas(arrs, seed):
    """The same reads with their QUAL / SEQ / CIGAR rows laid out in a random order inside the arenas (legal: brc.h asks for
    offsets inside the arenas, not for increasing ones), with gaps between the rows."""
    rng = np.random.default_rng(seed)
    n = len(arrs["pos"]); order = rng.permutation(n)
    out = dict(arrs)
    lq = np.asarray(arrs["l_qseq"]).astype(np.int64); nc = np.asarray(arrs["n_cigar"]).astype(np.int64)
    for arena, off_name, lens in (("qual", "qual_off", lq), ("seq4", "seq_off", (lq + 1) // 2), ("cigar", "cigar_off", nc)):
        src = np.asarray(arrs[arena]); offs = np.asarray(arrs[off_name]).astype(np.int64)
        gaps = rng.integers(0, 5, n)
        new_off = np.zeros(n, np.int64); at = int(rng.integers(0, 9))
        for r in order:
            new_off[r] = at; at += int(lens[r]) + int(gaps[r])
        dst = np.full(at + 8, 7 if arena != "cigar" else 0, src.dtype)
        for r in range(n):
            dst[new_off[r]:new_off[r] + lens[r]] = src[offs[r]:offs[r] + lens[r]]
        out[arena] = dst; out[off_name] = new_off.astype(np.uint64)
    return out


def test_hip_rows_in_any_arena_order(dev_lib, oracle_lib):
    """K1's per-base pass addresses a wave's QUAL / SEQ rows as wave-uniform base + 32-bit lane offset: the base must be the
    smallest offset of the wave's reads, not lane 0's (rows laid out in any order inside the batch arenas)."""
    rng = np.random.default_rng(17)
    ref = synth.make_ref(rng, 5000, weird=0.01)
    arrs = synth.make_batch(317, ref, 1500, style="mixed", mismatch=0.1, p_q2tail=0.3)
    sh = shuffled_arenas(arrs, 5)
    want_t, want_r = parity.run_engine(oracle_lib, arrs, [(0, 5000)], ref=ref)
    got_t, got_r = parity.run_engine(dev_lib, sh, [(0, 5000)], ref=ref)
    parity.assert_results_equal(got_r[0], want_r[0], "shuffled arenas")
    assert got_t == want_t


def test_hip_edge_cases(dev_lib, oracle_lib):
    rng = np.random.default_rng(5)
    ref = synth.make_ref(rng, 500)
    arrs = synth.make_batch(6, ref, 40, style="indel", region=(200, 300))
    # empty region (no reads overlap), region before / after all reads, single base, zero reads pushed, no reference
    text, res = parity.compare_libs(dev_lib, oracle_lib, arrs, [(0, 50), (480, 500), (250, 251), (0, 500)], ref=ref)
    assert res[0].n_pos == 0
    parity.compare_libs(dev_lib, oracle_lib, capi.select_reads(arrs, []), [(0, 100)], ref=ref)
    parity.compare_libs(dev_lib, oracle_lib, arrs, [(0, 500)], ref=None)
    a2 = synth.make_batch(12, ref, 300, style="simple", region=(100, 110), read_len=(50, 60))
    a2["pos"] = np.sort(np.where(np.arange(300) % 3 == 0, 100, a2["pos"])).astype(np.int32)
    for d in (1, 5, 40):
        parity.compare_libs(dev_lib, oracle_lib, a2, [(90, 200)], ref=ref, max_cnt=d)


def test_hip_is_deterministic_and_repeatable(dev_lib):
    rng = np.random.default_rng(99)
    ref = synth.make_ref(rng, 20000)
    arrs = synth.make_batch(100, ref, 6000, style="mixed", n_libs=3)
    eng = capi.Engine(dev_lib, per_lib=True, lib_names=["a", "b", "c"])
    eng.begin_region(0, 0, 20000, ref); eng.push_reads(arrs); eng.upload()
    eng.compute(); r1 = eng.fetch_result(); t1 = eng.format_region("x")
    eng.compute(); r2 = eng.fetch_result(); t2 = eng.format_region("x")
    parity.assert_results_equal(r1, r2, "repeat")
    assert t1 == t2
    eng.close()


def test_hip_wgs_sample_equals_oracle_and_full_size_properties(dev_lib, oracle_lib):
    """BASELINE config 3 data model at reduced contig length vs the oracle, then size-independent properties on a larger
    contig: (i) tiling invariance — the same contig computed as one region and as abutting sub-regions gives the
    same planes; (ii) conservation — sum of ncol over positions == sum over reads of in-window reference span."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import synthgen as gen
    ref, arrs = gen.generate(300_000, "wgs30x", seed=3, n_chunks=16)
    parity.compare_libs(dev_lib, oracle_lib, arrs, [(0, 300_000)], ref=ref, min_mapq=20, min_bq=13)
    # (the CPU lane simulator gets the same body on a contig an eighth the size)
    n = 8_000_000 if dev_lib.kind().startswith("hip") else 1_000_000
    ref, arrs = gen.generate(n, "wgs30x", seed=5, n_chunks=64)
    eng = capi.Engine(dev_lib, min_mapq=20, min_bq=13)
    eng.begin_region(0, 0, n, ref); eng.push_reads(arrs); whole = eng.end_region()
    ends = capi.read_ends(arrs)
    span = (np.minimum(ends, n) - np.maximum(arrs["pos"].astype(np.int64), 0)).clip(min=0).sum()
    assert int(whole.ncol.sum(dtype=np.uint64)) == int(span) == whole.n_events
    cuts = [0, n // 8 + 1, n // 8 + 2, n // 2 + 304, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        idx = capi.fetch_overlapping(arrs, ends, a - 1, b)
        eng.begin_region(0, a, b, ref); eng.push_reads(capi.select_reads(arrs, idx)); part = eng.end_region()
        p_lo = max(a, whole.pos0, part.pos0); p_hi = min(b, whole.pos0 + whole.n_pos, part.pos0 + part.n_pos)
        assert p_hi - p_lo > (b - a) - 400
        w = slice(p_lo - whole.pos0, p_hi - whole.pos0); q = slice(p_lo - part.pos0, p_hi - part.pos0)
        np.testing.assert_array_equal(part.istat[..., q], whole.istat[..., w])
        np.testing.assert_array_equal(part.fstat[..., q].view(np.uint32), whole.fstat[..., w].view(np.uint32))
        np.testing.assert_array_equal(part.depth[:, q], whole.depth[:, w])
    eng.close()


def test_hip_every_read_has_a_wide_row(dev_lib, oracle_lib):
    """HiFi-like qualities on the config-3 and config-5 data models: base qualities up to 93, so that nearly every read has a base the
    event byte cannot describe — every read gets a row in the sparse wide stream (k_wide_rows), every piece is PF_WIDE, and the lanes
    take quality and bucket from the words.  Planes, indel buckets and all three text routes against the oracle."""
    import synthgen
    hip = dev_lib.kind().startswith("hip")
    n = 400_000 if hip else 60_000
    for config, kw in (("wgs30x", dict(min_mapq=20, min_bq=13)), ("tumor200x", dict(per_lib=True, insertion_centric=True, lib_names=["libA", "libB", "libC", "libD"], min_bq=70))):
        m = n if config == "wgs30x" else n // 8
        ref, arrs = synthgen.generate(m, config, seed=17, n_chunks=4)
        q = arrs["qual"].astype(np.int32) + 52; q[::7] = 0; q[3::11] = 63; arrs["qual"] = np.minimum(q, 93).astype(np.uint8)
        regions = [(0, m), (m // 2, m // 2 + 1)]
        text, _ = parity.compare_libs(dev_lib, oracle_lib, arrs, regions, ref=ref, **kw)
        for route in (dict(text_only=True), dict(device_text="chrS")):
            got, _ = parity.run_engine(dev_lib, arrs, regions, ref=ref, **route, **kw)
            assert got == text


def test_hip_text_only_engine_prints_the_same(dev_lib, oracle_lib):
    """BRC_OPT_TEXT_ONLY (the command line's setting): the formatter reads the compact device result directly."""
    import synthgen
    ref, arrs = synthgen.generate(300_000, "tumor200x", seed=11, n_chunks=4)
    names = ["libA", "libB", "libC", "libD"]
    for kw in (dict(min_mapq=20, min_bq=13), dict(per_lib=True, insertion_centric=True, lib_names=names)):
        want, _ = parity.run_engine(oracle_lib, arrs, [(0, 40000), (150000, 150001)], ref=ref, **kw)
        got, res = parity.run_engine(dev_lib, arrs, [(0, 40000), (150000, 150001)], ref=ref, text_only=True, **kw)
        assert got == want and want.count(b"\n") > 39000
        assert not res[0].istat.any()
        got2, res2 = parity.run_engine(dev_lib, arrs, [(0, 40000), (150000, 150001)], ref=ref, device_text="chrS", **kw)   # BRC_OPT_DEVICE_TEXT
        assert got2 == want and not res2[0].ncol.any()


@pytest.mark.parametrize("case", FUZZ, ids=lambda c: "seed%d-%s" % (c["seed"], c["style"]))
def test_hip_device_text_equals_oracle_on_every_fuzz_family(dev_lib, oracle_lib, case):
    """k_text_len / k_text_write + the host's line patcher against the oracle's text (see the simulator twin)."""
    rng = np.random.default_rng(case["seed"])
    ref = synth.make_ref(rng, 3000, weird=case.get("weird", 0.0))
    n_libs = case.get("n_libs", 1)
    arrs = synth.make_batch(case["seed"] + 100, ref, case["n"], style=case["style"], n_libs=n_libs, p_nolib=case.get("p_nolib", 0.0))
    names = ["lib%c" % (65 + i) for i in range(n_libs)] if case["opts"].get("per_lib") else ()
    regions = [(0, 3000), (100, 101), (700, 1500), (1500, 1501), (1501, 2200), (2990, 3200), (1500, 1500), (5, 900)]
    for clear in (True, False):
        want, _ = parity.run_engine(oracle_lib, arrs, regions, ref=ref, lib_names=names, clear_queue=clear, **case["opts"])
        got, _ = parity.run_engine(dev_lib, arrs, regions, ref=ref, lib_names=names, clear_queue=clear, device_text="chrS", **case["opts"])
        assert got == want, clear


@pytest.mark.parametrize("shift", ["2", "4", "6", "0"])
def test_hip_deep_indel_key_equals_oracle(knob_lib, oracle_lib, monkeypatch, shift):
    """k_indel_reduce with hundreds (and, at 6000 reads, thousands) of events on one (position, library) key; every bucket size
    the engine chooses from (4 / 16 / 64 positions; "0": an unsupported value, ignored)."""
    monkeypatch.setenv("BRC_IBUCKET_SHIFT", shift)
    rng = np.random.default_rng(41)
    ref = synth.make_ref(rng, 600)
    for n in (700, 6000):
        arrs = synth.pile_indels(synth.make_batch(141, ref, n, style="simple", region=(215, 262), read_len=(80, 100), n_libs=2), 270, seed=3)
        for kw in (dict(), dict(per_lib=True, insertion_centric=True, lib_names=["libA", "libB"])):
            text, res = parity.compare_libs(knob_lib, oracle_lib, arrs, [(0, 600)], ref=ref, **kw)
            assert max(int(d["i"][0]) for d in res[0].indels) > n // 8


def test_hip_sequenceless_secondary_read(dev_lib, oracle_lib):
    """A secondary alignment stored without its sequence (SEQ '*', l_qseq 0) but with a CIGAR: in the columns, never counted."""
    rng = np.random.default_rng(1)
    ref = synth.make_ref(rng, 600)
    arrs = synth.add_sequenceless_secondary(synth.make_batch(5, ref, 60, style="simple", region=(100, 300)), 200)
    arrs = synth.add_sequenceless_secondary(arrs, 420, span=30)            # beyond every other read: positions that print only because of it
    for kw in (dict(), dict(min_mapq=10, min_bq=5), dict(per_lib=True, lib_names=["libA"])):
        text, _ = parity.compare_libs(dev_lib, oracle_lib, arrs, [(0, 600)], ref=ref, **kw)
        got, _ = parity.run_engine(dev_lib, arrs, [(0, 600)], ref=ref, device_text="chrS", **kw)
        assert got == text and b"\t421\t" in text


def test_hip_arena_offsets_beyond_2_gib(dev_lib, oracle_lib):
    """Offsets into the QUAL / SEQ arenas of a batch are 64-bit (include/brc.h): a region of more than 14 M 150-bp reads has
    rows beyond 2 GiB, one of more than 28 M beyond 4 GiB.  A few hundred reads whose rows sit around the 2^31 and 2^32 marks of
    (mostly empty) arenas: K1 assembles wave-uniform 64-bit bases from two 32-bit halves — a sign-extended low half read 4 GiB
    below the rows until round 3 (silently wrong, or a memory fault once the arenas pass 4 GiB)."""
    rng = np.random.default_rng(23)
    ref = synth.make_ref(rng, 4000)
    arrs = synth.make_batch(123, ref, 600, style="mixed", read_len=(100, 150))
    n = len(arrs["pos"])
    big = {k: v for k, v in arrs.items()}
    # row i of the new arenas: the first third below 2^31, the second third straddling and above it, the last third above 2^32
    marks = np.where(np.arange(n) < n // 3, (1 << 31) - 40_000, np.where(np.arange(n) < 2 * n // 3, (1 << 31) - 600, (1 << 32) - 300)).astype(np.int64)
    qlen = arrs["l_qseq"].astype(np.int64); slen = (qlen + 1) // 2
    # consecutive rows inside each third (arena order = read order, as a reader would lay them out)
    qoff = np.zeros(n, np.int64); soff = np.zeros(n, np.int64)
    for third in (slice(0, n // 3), slice(n // 3, 2 * n // 3), slice(2 * n // 3, n)):
        qoff[third] = marks[third] + np.concatenate([[0], np.cumsum(qlen[third])[:-1]])
        soff[third] = marks[third] + np.concatenate([[0], np.cumsum(slen[third])[:-1]])
    qual = np.zeros(int(qoff[-1] + qlen[-1]) + 64, np.uint8); seq4 = np.zeros(int(soff[-1] + slen[-1]) + 64, np.uint8)
    for i in range(n):
        o = int(arrs["qual_off"][i]); qual[qoff[i]:qoff[i] + qlen[i]] = arrs["qual"][o:o + qlen[i]]
        o = int(arrs["seq_off"][i]); seq4[soff[i]:soff[i] + slen[i]] = arrs["seq4"][o:o + slen[i]]
    big["qual"] = qual; big["seq4"] = seq4; big["qual_off"] = qoff.astype(np.uint64); big["seq_off"] = soff.astype(np.uint64)
    want, _ = parity.run_engine(oracle_lib, arrs, [(0, 4000)], ref=ref, min_mapq=5, min_bq=10)          # (the small arenas: the oracle's answer)
    got, res = parity.run_engine(dev_lib, big, [(0, 4000)], ref=ref, min_mapq=5, min_bq=10)
    assert got == want and want.count(b"\n") > 3000


@pytest.mark.parametrize("case", [c for c in FUZZ if c["style"] in ("indel", "mixed", "clip")][:3] or FUZZ[:3], ids=lambda c: "seed%d-%s" % (c["seed"], c["style"]))
def test_hip_announced_windows_equal_the_whole_region(dev_lib, case):
    """brc_region_windows (the site-list planner's hint): every announced window prints what it prints without the hint —
    deletion carry from its lead position included — and tiles no window touches come back empty."""
    rng = np.random.default_rng(case["seed"])
    ref = synth.make_ref(rng, 3000, weird=case.get("weird", 0.0))
    n_libs = case.get("n_libs", 1)
    arrs = synth.make_batch(case["seed"] + 100, ref, case["n"], style=case["style"], n_libs=n_libs, p_nolib=case.get("p_nolib", 0.0))
    names = ["lib%c" % (65 + i) for i in range(n_libs)] if case["opts"].get("per_lib") else ()
    # single sites, a window over a tile boundary, overlapping and duplicate windows, the region's first and last positions
    wins = [(0, 1), (5, 6), (63, 65), (64, 65), (127, 129), (700, 760), (730, 731), (730, 731), (1500, 1501), (2047, 2048), (2999, 3000), (2900, 3000)]
    eng = capi.Engine(dev_lib, lib_names=names, **case["opts"])

    def run(hint):
        eng.begin_region(0, 0, 3000, ref); eng.push_reads(arrs)
        if hint:
            eng.region_windows(np.array([w[0] for w in wins], np.int32), np.array([w[1] for w in wins], np.int32))
        res = eng.end_region()
        return [eng.format_window("chrS", b, e, 0) for b, e in wins], res.ncol.copy(), int(res.pos0), int(res.n_pos)

    want, ncol_all, pos0, n_pos = run(False)
    got, ncol_hint, pos0_h, n_pos_h = run(True)
    assert got == want and sum(len(t) for t in want) > 0 and (pos0_h, n_pos_h) == (pos0, n_pos)
    # what is piled up: per 64-position tile of the planes [pos0, pos0 + n_pos), from the first to the last position a window
    # [b - 1, e) asks for
    nt = (n_pos + 63) // 64
    lo = np.full(nt, 64, np.int64); hi = np.full(nt, -1, np.int64)
    for b, e in wins:
        k0, k1 = max(b - 1 - pos0, 0), min(e - pos0, n_pos)
        for t in range(k0 // 64, (k1 - 1) // 64 + 1 if k1 > k0 else 0):
            lo[t] = min(lo[t], max(k0 - 64 * t, 0)); hi[t] = max(hi[t], min(k1 - 1 - 64 * t, 63))
    lane = np.arange(nt * 64) % 64
    per_pos = ((lane >= np.repeat(lo, 64)) & (lane <= np.repeat(hi, 64)))[:n_pos]
    assert not ncol_hint[..., ~per_pos].any()                       # nothing piled up outside
    np.testing.assert_array_equal(ncol_hint[..., per_pos], ncol_all[..., per_pos])
    assert ncol_all[..., ~per_pos].any()
    # the hint does not outlive its region
    eng.begin_region(0, 0, 3000, ref); eng.push_reads(arrs)
    np.testing.assert_array_equal(eng.end_region().ncol, ncol_all)
    eng.close()


@pytest.mark.gpu
def test_hip_announced_windows_on_a_compacted_region(hip_lib, oracle_lib):
    """The site-list planner's windows over reads with an operator every few bases: k_narrow_tiles cuts every wanted tile's range to its
    wanted lanes (a tile's range may then START behind its neighbour's), k_compact_reads takes eight consecutive tiles per wave and
    carries a read's position from one tile to the next.  Every window prints what the whole region prints for it, and what the oracle
    prints for a region of its own over the same reads."""
    import synthgen
    ref, arrs = synthgen.generate_dense(60_000, "ont", seed=23, n_chunks=2)
    opts = dict(min_mapq=20, min_bq=13)
    rng = np.random.default_rng(5)
    starts = np.sort(rng.integers(600, 59_000, 70))
    wins = [(int(b), int(b) + int(rng.choice([1, 1, 1, 2, 30, 64, 200]))) for b in starts] + [(30_000, 30_001), (30_000, 30_001), (30_063, 30_066), (30_064, 30_065), (30_127, 30_129)]
    eng = capi.Engine(hip_lib, **opts)

    def run(hint):
        eng.begin_region(0, 0, 60_000, ref); eng.push_reads(arrs)
        if hint:
            eng.region_windows(np.array([w[0] for w in wins], np.int32), np.array([w[1] for w in wins], np.int32))
        res = eng.end_region()
        assert eng.piece_steps()[0] > 0                       # the region was compacted
        return [eng.format_window("chrS", b, e, 0) for b, e in wins], res.ncol.copy()

    want, ncol_all = run(False)
    got, ncol_hint = run(True)
    assert got == want and sum(len(t) > 0 for t in want) > len(want) * 3 // 4
    assert int(ncol_hint.sum()) < int(ncol_all.sum()) // 2
    ends = capi.read_ends(arrs)
    for b, e in wins[::9]:
        o = capi.Engine(oracle_lib, **opts)
        o.begin_region(0, b, e, ref); o.push_reads(capi.select_reads(arrs, capi.fetch_overlapping(arrs, ends, b - 1, e))); o.end_region()
        assert o.format_region("chrS") == want[wins.index((b, e))], (b, e)
        o.close()
    eng.close()


def test_hip_region_windows_argument_handling(dev_lib):
    """brc_region_windows: only inside an open region, windows must not end before they begin, windows outside the planes are
    clipped away, n = 0 withdraws the hint."""
    rng = np.random.default_rng(7)
    ref = synth.make_ref(rng, 2000)
    arrs = synth.make_batch(107, ref, 300, style="mixed")
    eng = capi.Engine(dev_lib, min_mapq=0, min_bq=0)
    with pytest.raises(capi.BrcError):
        eng.region_windows(np.array([5], np.int32), np.array([6], np.int32))            # no open region
    eng.begin_region(0, 0, 2000, ref); eng.push_reads(arrs)
    with pytest.raises(capi.BrcError):
        eng.region_windows(np.array([10], np.int32), np.array([9], np.int32))           # ends before it begins
    eng.region_windows(np.array([100], np.int32), np.array([101], np.int32))
    eng.region_windows(np.zeros(0, np.int32), np.zeros(0, np.int32))                    # withdrawn: everything is piled up
    whole = eng.end_region(); ncol_all = whole.ncol.copy(); text_all = eng.format_window("chrS", 100, 101, 0)
    assert ncol_all.any()
    eng.begin_region(0, 0, 2000, ref); eng.push_reads(arrs)
    eng.region_windows(np.array([-500, 100, 1_000_000], np.int32), np.array([-400, 101, 1_000_001], np.int32))   # two of them miss the planes
    res = eng.end_region()
    assert eng.format_window("chrS", 100, 101, 0) == text_all
    k = 100 - 1 - int(res.pos0)
    keep = np.zeros(int(res.n_pos), bool); keep[k:k + 2] = True                            # the window's lead position and the position itself
    assert not res.ncol[..., ~keep].any() and np.array_equal(res.ncol[..., keep], ncol_all[..., keep])
    eng.close()


@pytest.mark.parametrize("env", [{}, {"BRC_PACK_LIM": "90"}, {"BRC_PACK_LIM": "90", "BRC_FLUSH_K": "5"}, {"BRC_FORCE_DOM": "3", "BRC_PACK_LIM": "90"}],
                         ids=lambda e: "+".join("%s=%s" % kv for kv in e.items()) or "default")
def test_hip_table_piece_flag_combinations(knob_lib, oracle_lib, monkeypatch, env):
    """Reads of ONE length, so that nearly every piece is a table piece of k_pileup2 — and then every way of not being one behind
    the kernel's single flag test: soft-clipped reads (PF_TABQ: the event location divided out in the lane), -i's one-base pieces
    (PF_NB), pieces with huge integers (PF_HUGE, forced by a small packing limit: both a former table piece and a soft-clipped
    one), reads without a Q2 position, other lengths (rare record + exact divisions).  Same bits as the oracle."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(41)
    ref = synth.make_ref(rng, 4000, weird=0.01)
    same = synth.make_batch(411, ref, 900, read_len=(100, 100), style="mixed", n_libs=2, p_nolib=0.01, p_q2tail=0.4, mismatch=0.04)
    other = synth.make_batch(412, ref, 60, read_len=(40, 140), style="mixed", n_libs=2)
    both = {}                                                     # the two batches as one, in coordinate order
    for k in ("pos", "flag", "mapq", "lib", "l_qseq", "n_cigar", "nm", "sm", "tags", "cigar", "seq4", "qual"):
        both[k] = np.concatenate([np.asarray(same[k]), np.asarray(other[k])])
    for k, arena in (("cigar_off", "cigar"), ("seq_off", "seq4"), ("qual_off", "qual")):
        both[k] = np.concatenate([np.asarray(same[k]), np.asarray(other[k]) + np.uint64(len(same[arena]))])
    arrs = capi.select_reads(both, np.argsort(both["pos"], kind="stable"))
    names = ["libA", "libB"]
    parity.compare_libs(knob_lib, oracle_lib, arrs, [(0, 4000), (2000, 2001)], ref=ref, lib_names=names, per_lib=True, insertion_centric=True, check_warn=False)
    parity.compare_libs(knob_lib, oracle_lib, arrs, [(0, 4000)], ref=ref, min_mapq=10, min_bq=12)
    deep = synth.make_batch(413, ref, 2500, read_len=(100, 100), style="mixed", region=(1500, 1900), mismatch=0.03)        # ~600x of one length
    parity.compare_libs(knob_lib, oracle_lib, deep, [(1400, 2100)], ref=ref, insertion_centric=True)


@pytest.mark.parametrize("offset", [1_073_000_000, 2_147_000_000], ids=["above_2^30", "end_of_int32"])
def test_hip_coordinates_up_to_the_end_of_int32(dev_lib, oracle_lib, offset):
    """The same reads at the far end of the longest contig the formats allow (positions are int32 in BAM and in the reference's
    pileup_data_t, bamreadcount.cpp:52-66): every position-derived quantity of the device path — tile starts, the lanes of a tile
    past the region (a bias of 2^31 on their coordinate), slice offsets into the reference, indel keys — must behave at 2.147 G as
    it does near 0.  Mixed read lengths and indels: tables, in-lane division and the indel side path all run."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import synthgen
    n = 60_000
    small, arrs = synthgen.generate(n, "wgs30x_mixed", seed=11)
    L = min(offset + n, 2**31 - 1)
    ref = np.full(L, ord("N"), np.uint8); ref[offset:] = small[:L - offset]
    moved = dict(arrs); moved["pos"] = (np.asarray(arrs["pos"]).astype(np.int64) + offset).astype(np.int32)
    moved = capi.select_reads(moved, np.nonzero(capi.read_ends(moved).astype(np.int64) <= L)[0])
    near = capi.select_reads(arrs, np.nonzero(capi.read_ends(arrs).astype(np.int64) <= L - offset)[0])
    regions = [(offset + 100, L - 1), (offset + 5_000, offset + 5_001), (L - 300, L)]
    text_far, res_far = parity.compare_libs(dev_lib, oracle_lib, moved, regions, ref=ref, min_mapq=20, min_bq=13)
    # ... and the far result is the near one moved: same counts, same statistics, coordinates aside
    text_near, res_near = parity.compare_libs(dev_lib, oracle_lib, near, [(a - offset, b - offset) for a, b in regions], ref=small[:L - offset], min_mapq=20, min_bq=13)
    assert sum(r.n_events for r in res_far) == sum(r.n_events for r in res_near) > 1_000_000
    far_lines, near_lines = text_far.decode().splitlines(), text_near.decode().splitlines()
    assert len(far_lines) == len(near_lines)
    for a, b in zip(far_lines[::97], near_lines[::97]):
        fa, fb = a.split("\t"), b.split("\t")
        assert int(fa[1]) - offset == int(fb[1]) and fa[2:] == fb[2:]


@pytest.mark.parametrize("style,read_len,n", [("simple", (5400, 5600), 160), ("wild", (3000, 20000), 120), ("mixed", (100, 7000), 500)],
                         ids=["around_the_16_bit_limit", "20_kb_reads", "short_and_long_mixed"])
def test_hip_long_reads(dev_lib, oracle_lib, style, read_len, n):
    """Reads of thousands of bases (PacBio / ONT alignments): a lane's packed 16-bit sums (clipped length, mismatch-quality sum,
    single-ended mapping quality) are emptied between half-batches of twelve pieces only, so whatever does not fit twelve times
    must take the PF_HUGE path (brc_core.h: choose_pack).  Until the end of round 4 the limit followed K instead: with reads above
    5461 bases the sums of a half-batch overflowed into their neighbours — found by exactly this comparison."""
    rng = np.random.default_rng(3)
    ref = synth.make_ref(rng, 80_000)
    arrs = synth.make_batch(5, ref, n, read_len=read_len, style=style, region=(0, 60_000))
    for kw in (dict(min_mapq=0, min_bq=0), dict(min_mapq=20, min_bq=13, insertion_centric=True)):
        _, res = parity.compare_libs(dev_lib, oracle_lib, arrs, [(1000, 75_000)], ref=ref, **kw)
        assert sum(r.n_events for r in res) > 200_000


@pytest.mark.parametrize("seed", [60030, 20040])
def test_hip_spliced_and_structural_alignments(dev_lib, oracle_lib, seed):
    """Introns of up to 100 kb (N), deletions and insertions of hundreds of bases, soft clips of half a read: a piece then SPANS
    tiles it has no base in (ext >> len), and the window of its event bytes such a tile would stage lies 100 kb from the read's row —
    before the stream or past its end.  k_pileup2 staged it anyway (nobody reads the copy) and faulted on the address; the
    window now falls back to the row's first bytes (found by tools/fuzz/extreme.py on the GPU: the simulator has no staging)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz"))
    import extreme
    import zlib
    rng = np.random.default_rng(seed)
    while True:                      # (the first seed at or after `seed` whose scenario is of the spliced kind)
        kind, style, ref, arrs, regions, kw, clear = extreme.scenario(seed)
        if kind == "spliced":
            break
        seed += 1
    check_warn = not (kw.get("per_lib") and any(int(l) < 0 for l in arrs["lib"]))
    text, res = parity.compare_libs(dev_lib, oracle_lib, arrs, regions, ref=ref, clear_queue=clear, check_warn=check_warn, **kw)
    assert sum(r.n_events for r in res) > 0
    got, _ = parity.run_engine(dev_lib, arrs, regions, ref=ref, clear_queue=clear, device_text="chrS", **kw)
    assert zlib.crc32(got) == zlib.crc32(text)


def test_hip_device_text_hands_a_region_back_when_its_lines_do_not_fit(knob_lib, oracle_lib, monkeypatch):
    """The device writes a region's lines behind 32-bit offsets.  Whether they fit is estimated before the line kernels run
    (brc_host.cpp: 700 bytes per position and library) — an estimate that knows neither the lengths of the library names nor sums at
    the far end of int32 — and CHECKED after the length pass: the true 64-bit total against the 32-bit one.  A region that does
    not fit is formatted on the host instead; the text is the same.  (BRC_DEVICE_TEXT_LIMIT lowers the limit for the test: the
    real one needs 4 GiB of text.)"""
    ref, arrs = synth.make_ref(np.random.default_rng(5), 3000), None
    arrs = synth.make_batch(9, ref, 900, style="mixed", n_libs=2, region=(0, 2400))
    kw = dict(per_lib=True, lib_names=["L" * 500, "M" * 900], min_mapq=0, min_bq=0)
    want, _ = parity.run_engine(oracle_lib, arrs, [(10, 2300), (2300, 2350)], ref=ref, **kw)
    assert len(want) > 1_000_000
    for limit in ("100", "1500000", None):
        if limit is None:
            monkeypatch.delenv("BRC_DEVICE_TEXT_LIMIT", raising=False)
        else:
            monkeypatch.setenv("BRC_DEVICE_TEXT_LIMIT", limit)
        got, _ = parity.run_engine(knob_lib, arrs, [(10, 2300), (2300, 2350)], ref=ref, device_text="chrS", **kw)
        assert got == want, limit


@pytest.mark.parametrize("per_lib", [False, True])
def test_hip_adopted_arenas_equal_copied_ones(dev_lib, oracle_lib, per_lib):
    """brc_push_reads_pinned (include/brc.h, the zero-copy feed): SEQ / QUAL stay in the caller's page-locked memory, the engine uploads them
    from there segment by segment and reads them again for the allele text and the warning lines — planes, indel alleles, text and warnings
    of a region pushed that way in several batches equal the copied region's and the oracle's; the two kinds do not mix in one region."""
    rng = np.random.default_rng(77)
    ref = synth.make_ref(rng, 5000, weird=0.01)
    arrs = synth.make_batch(78, ref, 1500, style="indel", n_libs=3 if per_lib else 1)
    names = ["a", "b", "c"] if per_lib else ()
    opts = dict(per_lib=per_lib, insertion_centric=True, min_bq=5)
    want_text, want = parity.run_engine(oracle_lib, arrs, [(0, 5000)], ref=ref, lib_names=names, **opts)
    n = len(arrs["pos"]); cuts = [0, n // 4, n // 4, n // 2, n]             # (an empty batch among them)
    eng = capi.Engine(dev_lib, lib_names=names, **opts)
    eng.begin_region(0, 0, 5000, ref)
    for a, b in zip(cuts[:-1], cuts[1:]):
        eng.push_reads_pinned(capi.select_reads(arrs, np.arange(a, b)))
    got = eng.end_region()
    parity.assert_results_equal(got, want[0], "adopted arenas")
    assert eng.format_region("chrS") == want_text
    w_ad = capi._region_warnings(eng, "chrS") if hasattr(capi, "_region_warnings") else None
    # the same engine, next region, copied this time; then a region that tries both kinds
    eng.begin_region(0, 0, 5000, ref); eng.push_reads(arrs)
    got2 = eng.end_region()
    parity.assert_results_equal(got2, want[0], "copied arenas")
    if w_ad is not None:
        assert capi._region_warnings(eng, "chrS") == w_ad
    if dev_lib.kind().startswith("hip"):
        eng.begin_region(0, 0, 5000, ref); eng.push_reads(capi.select_reads(arrs, np.arange(0, 10)))
        with pytest.raises(capi.BrcError):
            eng.push_reads_pinned(capi.select_reads(arrs, np.arange(10, 20)))
    eng.close()


@pytest.mark.parametrize("case", FUZZ, ids=lambda c: "seed%d-%s" % (c["seed"], c["style"]))
def test_hip_compacted_tile_ranges_equal_oracle_on_every_fuzz_family(knob_lib, oracle_lib, monkeypatch, case):
    """k_compact_tiles (regions whose reads average more than a dozen pieces: the pieces of a tile's range that do not touch the tile are
    dropped before the pileup, in stream order) forced on for every fuzz family — planes, indel lists, text, device-side text and a window
    read back from the computed region equal the oracle's.  (BRC_COMPACT_TILES exists in the test-knobs library only; [sim]: the
    simulator has no compaction — it checks the body itself.)"""
    monkeypatch.setenv("BRC_COMPACT_TILES", "1")
    rng = np.random.default_rng(case["seed"])
    ref = synth.make_ref(rng, 3000, weird=case.get("weird", 0.0))
    n_libs = case.get("n_libs", 1)
    arrs = synth.make_batch(case["seed"] + 100, ref, case["n"], style=case["style"], n_libs=n_libs, p_nolib=case.get("p_nolib", 0.0))
    names = ["lib%c" % (65 + i) for i in range(n_libs)] if case["opts"].get("per_lib") else ()
    regions = [(0, 3000), (100, 101), (700, 1500), (2990, 3200)]
    nolib = case.get("p_nolib", 0.0) > 0
    want, _ = parity.compare_libs(knob_lib, oracle_lib, arrs, regions, ref=ref, lib_names=names, check_warn=not nolib, **case["opts"])
    got, _ = parity.run_engine(knob_lib, arrs, regions, ref=ref, lib_names=names, device_text="chrS", **case["opts"])
    assert got == want


@pytest.mark.parametrize("case", MANY_OPS, ids=lambda c: "seed%d" % c["seed"])
def test_hip_wave_form_annotator_on_reads_with_many_operators(dev_lib, knob_lib, oracle_lib, monkeypatch, case):
    """k_annotate_wave (one wave per read, for reads with more than two M operators): 100-1200-base reads with up to several hundred
    operators of every regular kind (tests/synth.py: many_cigar — one-base and hundred-base match runs, I / D / N, D then I, I then D,
    hard and soft clips at both ends, an insertion or a deletion in front of the first match), quality-2 tails, N bases, escapes; with
    -i, per library (with library-less reads), over IUPAC / lower-case reference characters and over NUL characters in the reference
    (the annotator's break, :151: the wave re-annotates such a read on one lane).  Planes, indel lists, text, device text and windows
    equal the oracle's; the test-knobs library with the wave form switched off (K1's serial walk) gives the same bytes.  ([sim]: the
    simulator has no wave form — the same inputs through the serial annotator and the shared host code.)"""
    ref, arrs, names, regions, nolib = synth.many_ops_inputs(case)
    if case.get("nul"):
        # (NUL characters INSIDE a reference text are no input a FASTA can produce — the oracle's C strings end there —: the wave form's
        # re-annotation of such reads is held against K1's serial walk instead)
        want, res = parity.run_engine(dev_lib, arrs, regions, ref=ref, lib_names=names, **case["opts"])
        monkeypatch.setenv("BRC_WAVE_FORM", "0")
        off, res_off = parity.run_engine(knob_lib, arrs, regions, ref=ref, lib_names=names, **case["opts"])
        for a, b in zip(res, res_off): parity.assert_results_equal(a, b, "NUL reference characters")
        assert off == want
        return
    want, _ = parity.compare_libs(dev_lib, oracle_lib, arrs, regions, ref=ref, lib_names=names, check_warn=not nolib, **case["opts"])
    got, _ = parity.run_engine(dev_lib, arrs, regions, ref=ref, lib_names=names, device_text="chrS", **case["opts"])
    assert got == want
    monkeypatch.setenv("BRC_WAVE_FORM", "0")
    off, _ = parity.run_engine(knob_lib, arrs, regions, ref=ref, lib_names=names, **case["opts"])
    assert off == want


def test_hip_wave_form_annotator_on_reads_with_eqx_operators(dev_lib, knob_lib, oracle_lib, monkeypatch):
    """= / X operators (pbmm2, minimap2 --eqx) beside M operators, in reads with many operators: fetch_func moves NEITHER of its cursors on
    = and X (bamreadcount.cpp:133-197) — the M operators behind them are compared at lagging offsets, pure = / X reads compare nothing —
    while the pileup iterator treats = and X like M.  Round 6: such reads take the wave form's EQX instantiations (k_annotate_wave<.., true>:
    the list keeps the annotator's own query offset beside the true one) instead of K1's one-lane walk.  Planes, indel lists, text and
    device text equal the oracle's; the wave form switched off (serial walk) gives the same bytes.  [sim]: the serial annotator."""
    for case in (dict(seed=3, opts=dict(min_mapq=10, min_bq=8)), dict(seed=4, opts=dict(insertion_centric=True, per_lib=True), n_libs=3), dict(seed=5, opts=dict(), keep_m=0.0)):
        ref, arrs, names, regions, nolib = synth.many_ops_inputs(dict(seed=case["seed"], n=300, opts=case["opts"], n_libs=case.get("n_libs", 1)))
        arrs = synth.eqx_cigars(arrs, seed=case["seed"], frac=0.8, keep_m=case.get("keep_m", 0.3))
        assert int(((arrs["cigar"] & 15) >= 7).sum()) > 1000
        want, _ = parity.compare_libs(dev_lib, oracle_lib, arrs, regions, ref=ref, lib_names=names, check_warn=not nolib, **case["opts"])
        got, _ = parity.run_engine(dev_lib, arrs, regions, ref=ref, lib_names=names, device_text="chrS", **case["opts"])
        assert got == want
        monkeypatch.setenv("BRC_WAVE_FORM", "0")
        off, _ = parity.run_engine(knob_lib, arrs, regions, ref=ref, lib_names=names, **case["opts"])
        monkeypatch.delenv("BRC_WAVE_FORM")
        assert off == want


def test_hip_wave_form_operator_count_limits(dev_lib, oracle_lib):
    """The wave form holds the M operators of a read in LDS: up to 1024 with four waves per workgroup, up to 5120 with one (reads of
    ~80 kb with a match run of ~15 bases between operators), up to 13 500 with one wave per CU (158 of the CU's 160 KB: ~210 kb); a read
    with more, and reads with P operators, keep K1's serial walk (= / X: the EQX instantiations, round 6) — side by side in one region, all
    equal to the oracle.
    [sim]: the serial walk."""
    ref, arrs = synth.operator_limit_reads()
    for opts in (dict(), dict(insertion_centric=True, min_bq=10))[:2 if dev_lib.kind().startswith("hip") else 1]:      # (the CPU twin: one pass)
        parity.compare_libs(dev_lib, oracle_lib, arrs, [(0, 40000), (900, 1100)], ref=ref, **opts)


def test_hip_reads_with_an_operator_every_few_bases(dev_lib, knob_lib, oracle_lib, monkeypatch):
    """ONT / CLR-like alignments (tools/synth_gen.c: synth_reads_dense — 3-10-kb reads, an insertion or a deletion every ~15 bases, 800
    operators per read): the PRODUCT library compacts their tile ranges by itself (more than a dozen pieces per read), reports how many
    piece-steps that saved, and equals the oracle — as does the same region with compaction forbidden, and through brc_compute_n /
    brc_fetch_window / announced windows."""
    import synthgen
    hip = dev_lib.kind().startswith("hip")
    n = 120_000 if hip else 60_000              # (the lane simulator walks every tile's whole piece range: a smaller region on CPUs)
    hi = 90_000 if hip else 32_000
    ref, arrs = synthgen.generate_dense(n, "ont", seed=11, n_chunks=2)
    assert float(arrs["n_cigar"].mean()) > 400
    opts = dict(min_mapq=20, min_bq=13)
    want_text, want = parity.run_engine(oracle_lib, arrs, [(500, hi)], ref=ref, **opts)
    eng = capi.Engine(dev_lib, **opts)
    eng.begin_region(0, 500, hi, ref); eng.push_reads(arrs); eng.upload()
    eng.compute_n(2)
    got = eng.fetch_result()
    parity.assert_results_equal(got, want[0], "dense operators, compacted")
    assert eng.format_region("chrS") == want_text
    if dev_lib.kind().startswith("hip"):
        ranged, walked = eng.piece_steps()
        assert 0 < walked < 0.2 * ranged, (ranged, walked)          # four fifths and more of the ranges' pieces do not touch their tile
    eng.clear_indel_queue()
    w = eng.fetch_window(30_000, 31_000)
    o2 = capi.Engine(oracle_lib, **opts)
    ends = capi.read_ends(arrs)
    o2.begin_region(0, 30_000, 31_000, ref); o2.push_reads(capi.select_reads(arrs, capi.fetch_overlapping(arrs, ends, 29_999, 31_000)))
    parity.assert_results_equal(w, o2.end_region(), "window of the compacted region"); o2.close()
    eng.close()
    # compaction forbidden (test-knobs library): the same bits, the slow way
    monkeypatch.setenv("BRC_COMPACT_TILES", "0")
    t2, r2 = parity.run_engine(knob_lib, arrs, [(500, 20_000)], ref=ref, **opts)
    t3, r3 = parity.run_engine(oracle_lib, arrs, [(500, 20_000)], ref=ref, **opts)
    parity.assert_results_equal(r2[0], r3[0], "dense operators, not compacted"); assert t2 == t3
    # compacted by the walk over whole ranges (k_compact_tiles: what several libraries use) instead of read-wise (k_compact_reads)
    monkeypatch.setenv("BRC_COMPACT_TILES", "2")
    t4, r4 = parity.run_engine(knob_lib, arrs, [(500, 20_000)], ref=ref, **opts)
    parity.assert_results_equal(r4[0], r3[0], "dense operators, compacted by the range walk"); assert t4 == t3
    # several libraries (library-major slots: the range walk) with library-less reads among them
    monkeypatch.delenv("BRC_COMPACT_TILES")
    lib2 = dict(arrs); rng = np.random.default_rng(5)
    lib2["lib"] = np.where(rng.random(len(arrs["pos"])) < 0.03, -1, rng.integers(0, 3, len(arrs["pos"]))).astype(np.int16)
    parity.compare_libs(dev_lib, oracle_lib, lib2, [(500, 20_000)], ref=ref, lib_names=["libA", "libB", "libC"], check_warn=False, per_lib=True, **opts)
    # one library with -p and library-less reads (their slots are the running ones: piece_off[] stays non-decreasing for the read-wise search)
    lib1 = dict(arrs); lib1["lib"] = np.where(rng.random(len(arrs["pos"])) < 0.1, -1, 0).astype(np.int16)
    parity.compare_libs(dev_lib, oracle_lib, lib1, [(500, 20_000)], ref=ref, lib_names=["libA"], check_warn=False, per_lib=True, **opts)
