"""Pins the C oracle (and through it everything else) against the REFERENCE'S OWN SOURCES compiled here.

oracle/ref_shim/ compiles /root/reference/src/exe/bam-readcount/bamreadcount.cpp (fetch_func, pileup_func, main) and
src/lib/bamrc/{BasicStat,IndelQueue,IndelQueueEntry}.cpp UNMODIFIED into oracle/_ref/ against a shim of the
samtools-1.10 / htslib-1.10 API (the only part that is not the reference's own code: htslib is a missing blob).

  * the reference's main() reproduces its own four golden files on the six integration commands;
  * oracle text == reference-compiled text, byte for byte, over the fuzz families (every CIGAR operator, flagged reads,
    -q/-b/-d/-p/-i, SM/NM present or missing, IUPAC / lower-case reference, queue kept or cleared);
  * the five Zm integers of fetch_func: SURVEY Appendix-B known answers;
  * the 13 raw BasicStat accumulators (IEEE bits) and operator<< text: oracle planes vs BasicStat::process_read;
  * the reference's unit KATs re-expressed: test/lib/bamrc/TestIndelQueue.cpp:21-88, TestIndelQueueEntry.cpp:24-34,
    TestAuxFields.cpp:9-48, TestReadWarnings.cpp:36-71.

The prebuilt oracle/_ref files travel to boxes without /root/reference; where neither exists the tests skip."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from bam_readcount_amd import capi
from test_cli import synthetic_bam  # noqa: F401  (fixture)
from conftest import GOLDEN, ROOT
import parity
import synth

REF_TREE = "/root/reference"
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
REF_LIB = os.path.join(REF_DIR, "libbamrc_ref.so")
REF_CLI = os.path.join(REF_DIR, "bam-readcount-ref")


@pytest.fixture(scope="session")
def ref_lib():
    if os.path.isdir(REF_TREE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "ref_shim")])
    if not os.path.exists(REF_LIB):
        pytest.skip("oracle/_ref not built and no reference checkout here")
    lib = capi.Library(REF_LIB)
    assert lib.kind() == "reference-compiled"
    L = lib.lib
    L.bamrc_ref_annotate.argtypes = [C.POINTER(capi.ReadBatch), C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    L.bamrc_ref_basicstat.argtypes = [C.POINTER(capi.ReadBatch), C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_void_p,
                                      C.c_void_p, C.POINTER(capi.Stat), C.c_char_p, C.c_size_t, C.c_void_p]
    L.bamrc_ref_queue_new.restype = C.c_void_p
    L.bamrc_ref_queue_free.argtypes = [C.c_void_p]
    L.bamrc_ref_queue_push.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_char_p]
    L.bamrc_ref_queue_size.argtypes = [C.c_void_p]; L.bamrc_ref_queue_size.restype = C.c_size_t
    L.bamrc_ref_queue_process.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t]
    L.bamrc_ref_zm_roundtrip.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p]
    L.bamrc_ref_readwarnings.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.POINTER(C.c_char_p), C.c_int, C.c_char_p, C.c_size_t]
    L.bamrc_ref_set_max_warnings.argtypes = [C.c_void_p, C.c_int64]
    L.bamrc_ref_warnings.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t)]
    return lib


# ---------------------------------------------------------------- the reference's main() on its own integration tests

def test_reference_main_reproduces_its_goldens(ref_lib, workdir):
    from test_cli import RUNS, run_cli
    for exp, bam, extra, how in RUNS:
        rc, out, err = run_cli(REF_CLI, workdir, bam, extra, how)
        assert rc == 0, err
        assert out == open(os.path.join(GOLDEN, exp), "rb").read(), (exp, bam, extra, how)
        assert "Minimum mapping quality is set to 0" in err


# ---------------------------------------------------------------- INTEGRATION.md section 2, compiled and run
# oracle/ref_shim/ref_bound.cpp: the reference's own main() with fetch_func / the plbuf calls bound to the brc C-ABI —
# option parsing, file opening, the site-list and region loops, messages and exit codes are the reference's; everything
# fetch_func, pileup_func, BasicStat and IndelQueue did happens behind include/brc.h (libbrc_sim.so here, libbrc_hip.so on the GPU).

BOUND_SIM = os.path.join(REF_DIR, "bam-readcount-bound-sim")
BOUND_HIP = os.path.join(REF_DIR, "bam-readcount-bound-hip")


def _bound_main_check(exe, workdir, synthetic_bam):
    from test_cli import RUNS, run_cli
    assert os.path.exists(exe), "oracle/_ref is built by oracle/ref_shim/Makefile where the reference checkout exists and travels prebuilt"
    # the reference's six integration commands: its goldens, and stdout + stderr + exit code of the reference-compiled main()
    for w in ("1", "3", "-1", "0"):
        for exp, bam, extra, how in RUNS:
            tail = ["-l", "site_list", bam] if how == "list" else [bam, "21:10402985-10402985", "21:10405200-10405200"]
            want = subprocess.run([REF_CLI, "-w", w] + extra + ["-f", "ref.fa"] + tail, cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            got = subprocess.run([exe, "-w", w] + extra + ["-f", "ref.fa"] + tail, cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert got.stdout == open(os.path.join(GOLDEN, exp), "rb").read(), (exp, bam, extra, how)
            assert (got.returncode, got.stdout) == (want.returncode, want.stdout), got.stderr[-500:]
            assert got.stderr == want.stderr, (w, bam, extra, how)
    # synthetic reads with every CIGAR operator, two libraries, regions in any order (deletions pending across them), a
    # site list with duplicate and overlapping lines, an unknown contig, no reference.  (Regions keep clear of the reads that
    # hang over a contig's end: past it the reference's annotator reads unowned memory, bamreadcount.cpp:149 — DESIGN.md.)
    d = synthetic_bam
    open(d / "sites_bound.txt", "w").write("chrA\t100\t160\nchrB\t5\t900\nchrA\t100\t160\nnochr\t1\t2\nchrA\t3390\t3400\nchrA\t130\t400\nbad line\n")
    for args in (["-f", "syn.fa", "syn.bam", "chrA:1-2000", "chrA:2001-3500", "chrB:1-1200", "chrA:100-100"],
                 ["-p", "-i", "-q", "10", "-b", "5", "-f", "syn.fa", "syn.bam", "chrB:1-1500", "chrA:1-3000"],
                 ["-w", "2", "-p", "-f", "syn.fa", "-l", "sites_bound.txt", "syn.bam"],
                 ["-w", "0", "-i", "-d", "30", "-f", "syn.fa", "-l", "sites_bound.txt", "syn.bam"],
                 ["-f", "syn.fa", "syn.bam", "chrA:1-50", "nochr:1-2", "chrB:1-5"]):
        want = subprocess.run([REF_CLI] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        got = subprocess.run([exe] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert (got.returncode, got.stdout) == (want.returncode, want.stdout), (args, got.stderr[-500:])
        assert got.stderr == want.stderr, args
        assert want.returncode == 1 or want.stdout.count(b"\n") > 50


def test_reference_main_bound_to_the_c_abi_cpu(ref_lib, workdir, synthetic_bam):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    _bound_main_check(BOUND_SIM, workdir, synthetic_bam)


@pytest.mark.gpu
def test_reference_main_bound_to_the_c_abi_gpu(ref_lib, workdir, synthetic_bam):
    """The same binding linked to libbrc_hip.so: the reference's main() driving the MI355X engine."""
    _bound_main_check(BOUND_HIP, workdir, synthetic_bam)


# ---------------------------------------------------------------- oracle text == reference-compiled text

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
    # flagged reads (SECONDARY / QCFAIL / DUP stay in the column, UNMAP does not), heavy
    dict(seed=11, style="mixed", n=400, opts=dict(), p_flagdrop=0.4),
    dict(seed=12, style="wild", n=400, opts=dict(per_lib=True, insertion_centric=True), n_libs=3, p_flagdrop=0.4, p_nolib=0.02),
    # max-count
    dict(seed=13, style="indel", n=500, opts=dict(max_cnt=1)),
    dict(seed=14, style="mixed", n=500, opts=dict(max_cnt=5)),
    dict(seed=15, style="wild", n=500, opts=dict(max_cnt=40, min_mapq=1)),
    # tags: nothing / everything
    dict(seed=16, style="indel", n=300, opts=dict(), p_nonm=1.0, p_sm=0.0),
    dict(seed=17, style="indel", n=300, opts=dict(min_bq=20), p_nonm=0.0, p_sm=1.0),
]


def fuzz_inputs(case, ref_len=3000, tail=400):
    """Reads start inside [0, ref_len); the reference is `tail` bases longer so that no read and no deletion allele
    reaches its end: past the end the reference reads unowned memory (bamreadcount.cpp:149,337 — undefined behaviour,
    where the oracle substitutes NUL / 'N')."""
    rng = np.random.default_rng(case["seed"])
    ref = synth.make_ref(rng, ref_len + tail, weird=case.get("weird", 0.0))
    n_libs = case.get("n_libs", 1)
    kw = {k: case[k] for k in ("p_flagdrop", "p_nonm", "p_sm") if k in case}
    arrs = synth.make_batch(case["seed"] + 100, ref, case["n"], style=case["style"], n_libs=n_libs, p_nolib=case.get("p_nolib", 0.0),
                            region=(0, ref_len), **kw)
    names = ["lib%c" % (65 + i) for i in range(n_libs)] if case["opts"].get("per_lib") else ()
    return ref, arrs, names


@pytest.mark.parametrize("case", FUZZ, ids=lambda c: "seed%d-%s" % (c["seed"], c["style"]))
def test_oracle_text_equals_reference_compiled(oracle_lib, ref_lib, case):
    ref, arrs, names = fuzz_inputs(case)
    regions = [(0, 3000), (100, 101), (700, 1500), (1490, 1700), (2990, 3200), (1500, 1500)]
    for clear_queue, site_mode in ((True, True), (False, False)):
        ta, _ = parity.run_engine(oracle_lib, arrs, regions, ref=ref, lib_names=names, clear_queue=clear_queue, ref_len_check=site_mode, **case["opts"])
        tb, _ = parity.run_engine(ref_lib, arrs, regions, ref=ref, lib_names=names, clear_queue=clear_queue, ref_len_check=site_mode, **case["opts"])
        assert len(ta) > 1000
        assert ta == tb, "oracle and reference-compiled text differ (clear_queue=%r)" % clear_queue


# (no-reference runs cannot be pinned: without -f the reference dereferences a NULL `ref` in fetch_func,
# bamreadcount.cpp:117,149 — it crashes on the first M operator.  The engines define that case: no mismatch qualities, no alleles, 'N'.)


def test_oracle_equals_reference_compiled_on_test_bam_full_window(oracle_lib, ref_lib, test_bam):
    names = [str(s) for s in test_bam["lib_names"]]
    for per_lib in (False, True):
        for ic in (False, True):
            kw = dict(tid=20, chrom="21", ref=test_bam["ref"], lib_names=names if per_lib else (), per_lib=per_lib, insertion_centric=ic)
            ta, _ = parity.run_engine(oracle_lib, test_bam, [(10402736, 10405248)], **kw)
            tb, _ = parity.run_engine(ref_lib, test_bam, [(10402736, 10405248)], **kw)
            assert ta == tb and ta.count(b"\n") == 796


# ---------------------------------------------------------------- fetch_func: the five Zm integers

APPENDIX_B = {0: (0, 250, 0, 248, 248), 1: (8, 250, 0, 234, 234), 2: (2, 250, 0, 17, 17), 4: (29, 250, 0, 83, 83),
              9: (9, 247, 0, 245, 245), 10: (38, 242, 8, 8, -1), 17: (12, 237, 13, 63, 63), 41: (6, 230, 0, 173, 173)}


def ref_annotate(ref_lib, arrs, ref, check=0):
    b, keep = capi.make_batch(arrs)
    zm = np.zeros(5 * len(arrs["pos"]), np.int32)
    r = np.ascontiguousarray(ref, np.uint8)
    assert ref_lib.lib.bamrc_ref_annotate(C.byref(b), r.ctypes.data, r.size, check, zm.ctypes.data) == 0
    return zm.reshape(-1, 5)


def test_reference_fetch_func_known_answers(ref_lib, test_bam):
    zm = ref_annotate(ref_lib, test_bam, test_bam["ref"])
    for rec, want in APPENDIX_B.items():
        assert tuple(int(x) for x in zm[rec]) == want, rec


# ---------------------------------------------------------------- BasicStat::process_read + operator<<, raw accumulators

def column_events(arrs, p, min_mapq=0, min_bq=0):
    """(read, qpos, bucket) of every counted event at position p for M-only reads, in column (= file) order."""
    canon = [0, 1, 2, 5, 3, 5, 5, 5, 4, 5, 5, 5, 5, 5, 5, 5]
    ev = []
    for i in range(len(arrs["pos"])):
        pos, L, fl = int(arrs["pos"][i]), int(arrs["l_qseq"][i]), int(arrs["flag"][i])
        if not (pos <= p < pos + L) or (fl & 4):
            continue
        q = p - pos
        if int(arrs["mapq"][i]) < min_mapq or int(arrs["qual"][int(arrs["qual_off"][i]) + q]) < min_bq or (fl & (256 | 512 | 1024)):
            continue
        b4 = (int(arrs["seq4"][int(arrs["seq_off"][i]) + (q >> 1)]) >> ((~q & 1) << 2)) & 15
        ev.append((i, q, canon[b4]))
    return ev


def ref_basicstat(ref_lib, arrs, ref, events, with_zm=1, is_indel=0):
    b, keep = capi.make_batch(arrs)
    r = np.ascontiguousarray(ref, np.uint8)
    er = np.array([e[0] for e in events], np.int32); eq = np.array([e[1] for e in events], np.int32)
    st = capi.Stat(); text = C.create_string_buffer(512); warn = np.zeros(3, np.uint64)
    assert ref_lib.lib.bamrc_ref_basicstat(C.byref(b), r.ctypes.data, r.size, with_zm, is_indel, len(events), er.ctypes.data, eq.ctypes.data,
                                           C.byref(st), text, 512, warn.ctypes.data) == 0
    return np.array(st.i[:], np.uint32), np.array(st.f[:], np.float32), text.value, [int(x) for x in warn]


@pytest.mark.parametrize("opts", [dict(), dict(min_mapq=20, min_bq=13)], ids=["q0b0", "q20b13"])
def test_oracle_raw_sums_equal_reference_basicstat(oracle_lib, ref_lib, opts):
    ref, arrs, _ = fuzz_inputs(dict(seed=31, style="simple", n=500, opts={}, p_flagdrop=0.1))
    text, res = parity.run_engine(oracle_lib, arrs, [(0, 3000)], ref=ref, **opts)
    r = res[0]
    lines = {int(l.split(b"\t")[1]): l for l in text.split(b"\n") if l}
    checked = 0
    for p in range(40, 2960, 37):
        ev = column_events(arrs, p, **opts)
        k = p - r.pos0
        for b in range(6):
            sub = [(i, q) for (i, q, bb) in ev if bb == b]
            ii, ff, txt, _ = ref_basicstat(ref_lib, arrs, ref, sub)
            np.testing.assert_array_equal(ii, r.istat[0, b, :, k], err_msg="pos %d bucket %d" % (p, b))
            np.testing.assert_array_equal(ff.view(np.uint32), r.fstat[0, b, :, k].view(np.uint32), err_msg="pos %d bucket %d floats" % (p, b))
            assert (b"\t" + b"=ACGTN"[b:b + 1] + b":" + txt) in lines[p + 1]                  # operator<< (BasicStat.cpp:110-159)
            checked += len(sub)
    assert checked > 400


def test_reference_basicstat_missing_tags_and_indel_flag(oracle_lib, ref_lib):
    ref, arrs, _ = fuzz_inputs(dict(seed=32, style="simple", n=60, opts={}, p_nonm=0.5, p_sm=0.3))
    ev = [(i, int(arrs["l_qseq"][i]) // 2) for i in range(60)]
    ii, ff, txt, warn = ref_basicstat(ref_lib, arrs, ref, ev, with_zm=0)
    assert warn[2] == 60 and ii[capi.I_NAMES.index("smmq")] == 0 and ii[capi.I_NAMES.index("sclip")] == 0     # Zm missing: BasicStat.cpp:73-75
    assert warn[1] == int(((arrs["tags"] & 1) == 0).sum())
    assert warn[0] == int((((arrs["tags"] & 2) == 0) & ((arrs["flag"] & 2) != 0)).sum())
    ii2, _, txt2, _ = ref_basicstat(ref_lib, arrs, ref, ev, is_indel=1)
    assert ii2[capi.I_NAMES.index("sbq")] == 0 and txt2.split(b":")[2] == b"0.00"                               # :101-103, :120-122
    _, _, txt0, _ = ref_basicstat(ref_lib, arrs, ref, [])
    assert txt0 == b"0:0.00:0.00:0.00:0:0:0.00:0.00:0.00:0:0.00:0.00:0.00"                                      # :142-155


# ---------------------------------------------------------------- the reference's unit tests, re-expressed

def test_kat_indel_queue(ref_lib):
    """test/lib/bamrc/TestIndelQueue.cpp:21-88 on the reference-compiled IndelQueue."""
    L = ref_lib.lib
    buf = C.create_string_buffer(1024)
    q = L.bamrc_ref_queue_new()                                   # push
    assert L.bamrc_ref_queue_size(q) == 0
    L.bamrc_ref_queue_push(q, 0, 0, 0, 0, b"")
    assert L.bamrc_ref_queue_size(q) == 1
    L.bamrc_ref_queue_free(q)
    q = L.bamrc_ref_queue_new()                                   # process_irrelevant
    L.bamrc_ref_queue_push(q, 1, 1, 0, 0, b""); L.bamrc_ref_queue_push(q, 1, 5, 0, 0, b"")
    L.bamrc_ref_queue_process(q, 1, 2, buf, 1024)
    assert L.bamrc_ref_queue_size(q) == 1
    L.bamrc_ref_queue_free(q)
    q = L.bamrc_ref_queue_new()                                   # process_new_chromosome
    L.bamrc_ref_queue_push(q, 1, 1, 0, 0, b""); L.bamrc_ref_queue_push(q, 1, 5, 0, 0, b"")
    L.bamrc_ref_queue_process(q, 10, 2, buf, 1024)
    assert L.bamrc_ref_queue_size(q) == 0
    L.bamrc_ref_queue_free(q)
    q = L.bamrc_ref_queue_new()                                   # process_relevant
    L.bamrc_ref_queue_push(q, 1, 1, 0, 0, b""); L.bamrc_ref_queue_push(q, 1, 5, 10, 0, b"")
    depth = L.bamrc_ref_queue_process(q, 1, 5, buf, 1024)
    assert L.bamrc_ref_queue_size(q) == 0 and depth == 10 and len(buf.value) != 0
    L.bamrc_ref_queue_free(q)
    # TestIndelQueueEntry.cpp:24-34: entry text = allele ':' stat
    q = L.bamrc_ref_queue_new()
    L.bamrc_ref_queue_push(q, 1, 20, 0, 1, b"-AA")
    L.bamrc_ref_queue_process(q, 1, 20, buf, 1024)
    assert buf.value == b"\t-AA:0:0.00:0.00:0.00:0:0:0.00:0.00:0.00:0:0.00:0.00:0.00"
    L.bamrc_ref_queue_free(q)


def deletion_reads():
    """two reads 40M3D40M at 60 and 70 over an ACGT reference: deletion of ref[100:103] / ref[110:113]"""
    rng = np.random.default_rng(3)
    ref = synth.make_ref(rng, 400)
    n = 2; L = 80
    cig = np.array([(40 << 4) | 0, (3 << 4) | 2, (40 << 4) | 0] * n, np.uint32)
    pos = np.array([60, 70], np.int32)
    seqs, quals = [], []
    for s in pos:
        bases = np.concatenate([ref[s:s + 40], ref[s + 43:s + 83]])
        codes = np.array([synth.CODE[chr(c)] for c in bases], np.uint8)
        seqs.append(((codes[0::2] << 4) | codes[1::2]).astype(np.uint8)); quals.append(np.full(L, 30, np.uint8))
    arrs = dict(pos=pos, flag=np.array([0, 16], np.uint16), mapq=np.array([60, 50], np.uint8), lib=np.zeros(n, np.int16),
                l_qseq=np.full(n, L, np.int32), n_cigar=np.full(n, 3, np.uint32), cigar_off=np.array([0, 3], np.uint64),
                seq_off=np.array([0, 40], np.uint64), qual_off=np.array([0, 80], np.uint64), nm=np.full(n, 3, np.int32), sm=np.zeros(n, np.int32),
                tags=np.full(n, 1, np.uint8), cigar=cig, seq4=np.concatenate(seqs), qual=np.concatenate(quals))
    return ref, arrs


def test_kat_indel_queue_through_the_engines(oracle_lib, ref_lib):
    """The same three behaviours through the C-ABI (IndelQueue.cpp:3-15 inside pileup_func :391-409): a deletion found at
    position 99 (0-based) is queued for position 100; whether it prints depends on what is processed next."""
    ref, arrs = deletion_reads()
    dele = b"\t-" + bytes(ref[100:103])

    def run(lib, steps):
        eng = capi.Engine(lib)
        out = []
        ends = capi.read_ends(arrs)
        for tid, beg0, end, clear in steps:
            idx = capi.fetch_overlapping(arrs, ends, beg0 - 1, end)       # the same reads exist on every contig
            eng.begin_region(tid, beg0, end, ref)
            eng.push_reads(capi.select_reads(arrs, idx))
            eng.end_region()
            out.append(eng.format_region("c%d" % tid))
            if clear:
                eng.clear_indel_queue()
        eng.close()
        return out

    def depth(t):
        return int(t.split(b"\t")[3])

    for lib in (oracle_lib, ref_lib):
        # At 0-based position 100 the column holds read 2's base (mapq_n = 1) and read 1's deletion (not counted).  Every
        # region recomputes its lead position 99, so it queues the deletion (0,100) itself: one print, depth 1 + 1.
        b = run(lib, [(0, 90, 100, True), (0, 100, 101, False)])
        assert dele not in b[0] and b[1].count(dele) == 1 and depth(b[1]) == 2
        # relevant (process_relevant): without the clear of :605 the entry queued by the first region is still there and
        # prints too — abutting command-line regions report the deletion twice and count it twice
        a = run(lib, [(0, 90, 100, False), (0, 100, 101, False)])
        assert dele not in a[0] and a[1].count(dele) == 2 and depth(a[1]) == 3
        # irrelevant (process_irrelevant): a region further right passes the stale entry by (entry.pos < pos): dropped silently
        c = run(lib, [(0, 90, 100, False), (0, 105, 106, False), (0, 100, 101, False)])
        assert dele not in c[1] and c[2].count(dele) == 1 and depth(c[2]) == 2
        # new chromosome (process_new_chromosome): the stale entry of contig 0 is dropped when contig 1 is processed
        d = run(lib, [(0, 90, 100, False), (1, 100, 101, False)])
        assert d[1].count(dele) == 1 and depth(d[1]) == 2
    for steps in ([(0, 90, 100, False), (0, 100, 101, False)], [(0, 90, 100, False), (1, 100, 101, False), (0, 95, 120, False)]):
        assert run(oracle_lib, steps) == run(ref_lib, steps)


def test_kat_aux_fields(ref_lib):
    """test/lib/bamrc/TestAuxFields.cpp:9-48"""
    for vals, text in (((1, 2, 3, 4, 5), b"1 2 3 4 5"), ((-1, -2, -3, -4, -5), b"-1 -2 -3 -4 -5")):
        a = np.array(vals, np.int32); out = np.zeros(5, np.int32); buf = C.create_string_buffer(128)
        ref_lib.lib.bamrc_ref_zm_roundtrip(a.ctypes.data, buf, 128, out.ctypes.data)
        assert buf.value == text and tuple(out) == vals


def run_readwarnings(ref_lib, max_per_type, rounds):
    types = np.array([0, 1, 2], np.int32)
    names = (C.c_char_p * 3)(b"x", b"y", b"z")
    buf = C.create_string_buffer(1 << 16)
    ref_lib.lib.bamrc_ref_readwarnings(max_per_type, rounds, types.ctypes.data, names, 3, buf, 1 << 16)
    return buf.value.decode().splitlines()


def test_kat_read_warnings(ref_lib):
    """test/lib/bamrc/TestReadWarnings.cpp:36-71"""
    lines = run_readwarnings(ref_lib, -1, 100)
    assert len(lines) == 300
    assert sum("SM tag" in l for l in lines) == 100 and sum("NM tag" in l for l in lines) == 100 and sum("generated tag" in l for l in lines) == 100
    lines = run_readwarnings(ref_lib, 5, 100)
    assert len(lines) == 15 + 3
    assert sum("SM tag" in l for l in lines) == 5 and sum("NM tag" in l for l in lines) == 5 and sum("generated tag" in l for l in lines) == 5
    assert lines[0] == "WARNING: In read x: Couldn't find single-end mapping quality. Check to see if the SM tag is in BAM."
    assert "The previous warning has been emitted 5 times and will be disabled." in lines


# ---------------------------------------------------------------- ReadWarnings through the engines (bamreadcount.cpp:178,192,239,108)

WARN_CASES = [c for c in FUZZ if c["seed"] in (3, 7, 8, 12, 16)]


def warnings_per_region(lib, arrs, regions, ref, names, max_w, opts, site_mode=True, window_of=None):
    """Per region: the WARNING text a fresh ReadWarnings(max_w) would print (each compute of the reference-compiled engine
    starts one, as main() does once per run).  `window_of`: compute that covering region once and ask for each region as a
    sub-window instead (the site-list planner's path, brc_window_warnings)."""
    eng = capi.Engine(lib, lib_names=names, ref_len_check=site_mode, **opts)
    ends = capi.read_ends(arrs)
    out = []
    try:
        if lib.kind() == "reference-compiled":
            lib.lib.bamrc_ref_set_max_warnings(eng.h, max_w)
        if window_of is not None:
            idx = capi.fetch_overlapping(arrs, ends, window_of[0] - 1, window_of[1])
            eng.begin_region(0, window_of[0], window_of[1], ref); eng.push_reads(capi.select_reads(arrs, idx)); eng.end_region()
            for (b, e) in regions:
                out.append(eng.warnings_text(eng.window_warnings(b, e, max_w), max_w, [0, 0, 0, 0]))
            return out
        for (b, e) in regions:
            idx = capi.fetch_overlapping(arrs, ends, b - 1, e)
            eng.begin_region(0, b, e, ref); eng.push_reads(capi.select_reads(arrs, idx)); eng.end_region()
            if lib.kind() == "reference-compiled":
                p = C.c_char_p(); n = C.c_size_t()
                lib.lib.bamrc_ref_warnings(eng.h, C.byref(p), C.byref(n))
                out.append(C.string_at(p, n.value))
            else:
                out.append(eng.warnings_text(eng.region_warnings("chrS", max_w), max_w, [0, 0, 0, 0]))
            eng.clear_indel_queue()
    finally:
        eng.close()
    return out


def warn_inputs(case):
    ref, arrs, names = fuzz_inputs(case)
    arrs = dict(arrs)
    arrs["qname"] = [b"read/%d:%d" % (case["seed"], i) for i in range(len(arrs["pos"]))]
    return ref, arrs, names


WARN_REGIONS = [(0, 3000), (100, 101), (700, 1500), (1490, 1700), (2990, 3200), (1500, 1500)]


@pytest.mark.parametrize("case", WARN_CASES, ids=lambda c: "seed%d-%s" % (c["seed"], c["style"]))
def test_warning_text_equals_reference_compiled(oracle_lib, sim_lib, ref_lib, case):
    """The per-read WARNING lines (missing SM / NM tag, library unavailable) and their -w cap: the product's host generator
    (csrc/brc_host.cpp warnings_impl, linked into tests/sim) and the oracle's recorder against the reference's own WARN
    stream, byte for byte, in the order the reference's column walk emits them."""
    ref, arrs, names = warn_inputs(case)
    some = 0
    for max_w in (1, 3, -1, 0):
        want = warnings_per_region(ref_lib, arrs, WARN_REGIONS, ref, names, max_w, case["opts"])
        got = warnings_per_region(sim_lib, arrs, WARN_REGIONS, ref, names, max_w, case["opts"])
        assert got == want, "product warnings differ at -w %d" % max_w
        if max_w in (1, 3):         # the oracle records at most BRC_ORACLE_WARN_CAP events per type
            assert warnings_per_region(oracle_lib, arrs, WARN_REGIONS, ref, names, max_w, case["opts"]) == want
        some += sum(len(t) for t in want)
        # a planned window of a covering batch prints what its own run would have printed
        if max_w != 0:
            narrow = [(100, 101), (1490, 1700), (2500, 2501), (1500, 1500)]
            w_ref = warnings_per_region(ref_lib, arrs, narrow, ref, names, max_w, case["opts"])
            w_win = warnings_per_region(sim_lib, arrs, narrow, ref, names, max_w, case["opts"], window_of=(90, 2700))
            assert w_win == w_ref, "window warnings differ at -w %d" % max_w
    assert some > 0


def test_cli_bounds_warning_equals_reference_main(ref_lib, tmp_path):
    """fetch_func's "WARNING: Request for position .. is > length of .." line (bamreadcount.cpp:144-148, site-list mode,
    straight to stderr, not capped by -w).  It can only fire for an M operator that starts beyond the loaded contig without
    any earlier M base landing on the terminating NUL (:150,175 stop the walk there): here a read whose deletion spans the
    end of a FASTA contig that is shorter than the BAM header says.  stdout, stderr (with the interleaving of WARN lines
    and bounds lines) and exit code against the reference's main()."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bamio
    from test_cli import SIM_CLI, _write_fasta
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    d = tmp_path
    rng = np.random.default_rng(5)
    ref = synth.make_ref(rng, 5000)
    arrs = synth.make_batch(31, ref, 1500, style="mixed", n_libs=2, p_nolib=0.05)
    cut = None
    for i in range(len(arrs["pos"]) - 1, -1, -1):               # the last read with M..D..M: cut the contig at its deletion
        x = int(arrs["pos"][i]); ops = arrs["cigar"][int(arrs["cigar_off"][i]):int(arrs["cigar_off"][i]) + int(arrs["n_cigar"][i])]
        seen_m = False
        for k, c in enumerate(ops):
            op, ln = int(c) & 15, int(c) >> 4
            if op == 2 and seen_m and any((int(c2) & 15) == 0 for c2 in ops[k + 1:]) and not (int(arrs["flag"][i]) & 4):
                cut = x; break
            seen_m |= op == 0
            if op in (0, 2, 3, 7, 8): x += ln
        if cut: break
    assert cut and cut > 4000
    rgs = [["rgA1", "rgB1"][int(l)] if l >= 0 else None for l in arrs["lib"]]
    bamio.write_bam(str(d / "syn.bam"), [("chrA", 5000)], arrs, np.zeros(len(arrs["pos"]), int), rg_of_read=rgs,
                    rg_lines=["@RG\tID:rgA1\tLB:libA\tSM:s", "@RG\tID:rgB1\tLB:libB\tSM:s"], block_bytes=6000)
    _write_fasta(d / "short.fa", [("chrA", ref[:cut])])
    sites = [("chrA", cut - 60, cut - 5), ("chrA", cut - 10, cut - 10), ("chrA", 100, 120), ("chrA", cut - 4, cut - 1)]
    open(d / "sites.txt", "w").write("".join("%s\t%d\t%d\n" % s for s in sites))
    for w in ("1", "-1"):
        for extra in ([], ["-p"], ["-i", "-p"]):
            want = subprocess.run([REF_CLI, "-w", w, "-f", "short.fa", "-l", "sites.txt"] + extra + ["syn.bam"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert want.returncode == 0 and want.stderr.count(b"WARNING: Request for position") > 10
            for more in ([], ["--brc-gpus", "2"], ["--brc-plan", "0"]):
                got = subprocess.run([SIM_CLI, "-w", w, "-f", "short.fa", "-l", "sites.txt"] + extra + more + ["syn.bam"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                assert (got.returncode, got.stdout) == (want.returncode, want.stdout)
                assert got.stderr == want.stderr, (w, extra, more)


def test_cli_stderr_equals_reference_main(ref_lib, workdir):
    """The six integration commands (and the multi-engine / line-by-line variants of the drop-in): stdout, stderr and exit
    code of the drop-in CLI == the reference's own main() compiled over the shim, for -w 1, 3, unlimited and 0."""
    from test_cli import RUNS, SIM_CLI
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    for w in ("1", "3", "-1", "0"):
        for _, bam, extra, how in RUNS:
            tail = ["-l", "site_list", bam] if how == "list" else [bam, "21:10402985-10402985", "21:10405200-10405200"]
            want = subprocess.run([REF_CLI, "-w", w] + extra + ["-f", "ref.fa"] + tail, cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            for more in ([], ["--brc-gpus", "2"], ["--brc-plan", "0"]):
                got = subprocess.run([SIM_CLI, "-w", w] + extra + ["-f", "ref.fa"] + more + tail, cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                assert (got.returncode, got.stdout) == (want.returncode, want.stdout)
                assert got.stderr == want.stderr, (w, bam, extra, how, more)


@pytest.mark.gpu
def test_cli_stderr_equals_reference_main_gpu(ref_lib, workdir):
    """The product binary on the GPU: stdout + stderr + exit code of the six integration commands == the reference's main()
    (oracle/_ref travels prebuilt), -w 1 and unlimited, one and three engines on the device."""
    from test_cli import RUNS, HIP_CLI
    env = dict(os.environ, BRC_DEVICES="0,0,0")
    for w in ("1", "-1"):
        for _, bam, extra, how in RUNS:
            tail = ["-l", "site_list", bam] if how == "list" else [bam, "21:10402985-10402985", "21:10405200-10405200"]
            want = subprocess.run([REF_CLI, "-w", w] + extra + ["-f", "ref.fa"] + tail, cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            for more, e in (([], None), (["--brc-plan", "0"], env)):
                got = subprocess.run([HIP_CLI, "-w", w] + extra + ["-f", "ref.fa"] + more + tail, cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
                assert (got.returncode, got.stdout) == (want.returncode, want.stdout), got.stderr[-400:]
                assert got.stderr == want.stderr, (w, bam, extra, how, more)


@pytest.mark.gpu
@pytest.mark.parametrize("case", WARN_CASES[:3], ids=lambda c: "seed%d-%s" % (c["seed"], c["style"]))
def test_hip_warning_text_equals_reference_compiled(hip_lib, ref_lib, case):
    ref, arrs, names = warn_inputs(case)
    for max_w in (1, -1):
        want = warnings_per_region(ref_lib, arrs, WARN_REGIONS, ref, names, max_w, case["opts"])
        assert warnings_per_region(hip_lib, arrs, WARN_REGIONS, ref, names, max_w, case["opts"]) == want


def test_pending_deletion_blocks_the_queue_like_the_reference(oracle_lib, sim_lib, ref_lib):
    """IndelQueue::process only ever looks at the FRONT of its queue (IndelQueue.cpp:3-15).  Regions given on the command line
    are not separated by a clear (:641-657), so a deletion left pending at the end of one region (due at a position the next
    region never reaches) blocks every deletion queued behind it: the second region prints none of its deletions, and when a
    third region finally reaches the pending position, one entry is printed, the stale ones behind it are dropped only at
    the NEXT position, and the second copy queued by the third region itself stays stuck.  All host routes of the product
    (dense planes, compact planes, device-side text) must reproduce that, depth column included."""
    rng = np.random.default_rng(7)
    ref = synth.make_ref(rng, 2000)
    arrs = synth.make_batch(11, ref, 900, style="simple", region=(600, 1500), read_len=(80, 100), n_libs=2)
    arrs = synth.pile_indels(arrs, 1499, seed=1, frac=0.5)        # deletions behind position 1499: queued for 1500, outside the first region
    arrs = synth.pile_indels(arrs, 799, seed=2, frac=0.5)
    regions = [(1400, 1500), (700, 900), (1490, 1510)]
    for kw in (dict(), dict(per_lib=True, lib_names=["libA", "libB"])):
        want, _ = parity.run_engine(ref_lib, arrs, regions, ref=ref, clear_queue=False, **kw)
        line801 = [l for l in want.split(b"\n") if l.startswith(b"chrS\t801\t")][0]
        assert b"\t-" not in line801                                  # the second region's deletions are stuck behind the pending one
        assert parity.run_engine(oracle_lib, arrs, regions, ref=ref, clear_queue=False, **kw)[0] == want
        for route in (dict(), dict(text_only=True), dict(device_text="chrS")):
            got, _ = parity.run_engine(sim_lib, arrs, regions, ref=ref, clear_queue=False, **route, **kw)
            assert got == want, (kw, route)
        # the pooled formatter cuts a region into chunks of positions: with something pending it must not
        os.environ["BRC_FORMAT_CHUNK"] = "16"; os.environ["BRC_FORMAT_THREADS"] = "4"
        try:
            for route in (dict(), dict(text_only=True)):
                got, _ = parity.run_engine(sim_lib, arrs, regions, ref=ref, clear_queue=False, **route, **kw)
                assert got == want, (kw, route, "chunked")
        finally:
            del os.environ["BRC_FORMAT_CHUNK"]; del os.environ["BRC_FORMAT_THREADS"]


@pytest.mark.gpu
def test_pending_deletion_blocks_the_queue_like_the_reference_gpu(hip_lib, ref_lib):
    """The same scenario through the product library on the GPU (all three text routes)."""
    rng = np.random.default_rng(7)
    ref = synth.make_ref(rng, 2000)
    arrs = synth.make_batch(11, ref, 900, style="simple", region=(600, 1500), read_len=(80, 100), n_libs=2)
    arrs = synth.pile_indels(synth.pile_indels(arrs, 1499, seed=1, frac=0.5), 799, seed=2, frac=0.5)
    regions = [(1400, 1500), (700, 900), (1490, 1510)]
    for kw in (dict(), dict(per_lib=True, lib_names=["libA", "libB"])):
        want, _ = parity.run_engine(ref_lib, arrs, regions, ref=ref, clear_queue=False, **kw)
        for route in (dict(), dict(text_only=True), dict(device_text="chrS")):
            got, _ = parity.run_engine(hip_lib, arrs, regions, ref=ref, clear_queue=False, **route, **kw)
            assert got == want, (kw, route)


def random_scenario(seed, big=False):
    """One scenario of the engine-level differential fuzz (also driven from tools/gpu_soak.py with other seeds; big: deeper
    piles and longer reads, so that flushes of the packed registers and many half-batches per tile happen without knobs)."""
    rng = np.random.default_rng(seed)
    RL = int(rng.integers(400, 2500))
    ref = synth.make_ref(rng, RL + (1400 if big else 600), weird=float(rng.choice([0, 0, 0.02])))
    n_libs = int(rng.choice([1, 1, 2, 3]))
    style = str(rng.choice(["simple", "indel", "wild", "mixed"]))
    n_reads = int(rng.integers(50, 700)) if not big else int(rng.integers(2000, 9000))
    max_len = int(rng.integers(40, 160)) if not big else int(rng.integers(100, 900))
    arrs = synth.make_batch(seed + 1000, ref, n_reads, style=style, n_libs=n_libs, p_nolib=float(rng.choice([0, 0, 0.05])),
                            read_len=(20, max_len), region=(0, RL))
    if rng.random() < 0.4:
        arrs = synth.pile_indels(arrs, int(rng.integers(50, RL - 50)), seed=seed, frac=0.5)
    regions = []
    for _ in range(int(rng.integers(1, 7))):
        a = int(rng.integers(0, RL)); b = a + int(rng.choice([0, 1, 2, 17, 200, RL]))
        regions.append((a, min(b, RL + 100)))
    if rng.random() < 0.3:
        regions.append((regions[0][1], regions[0][1] + 50))
    per_lib = bool(rng.random() < 0.4)
    kw = dict(min_mapq=int(rng.choice([0, 0, 10, 30])), min_bq=int(rng.choice([0, 0, 13, 25])), insertion_centric=bool(rng.random() < 0.4))
    if rng.random() < 0.15:
        kw["max_cnt"] = int(rng.choice([1, 3, 10]))
    if per_lib:
        kw.update(per_lib=True, lib_names=["lib%c" % (65 + i) for i in range(n_libs)])
    clear = bool(rng.random() < 0.5)
    # round 6, a random stream of its own (the scenarios of the seeds above stay what they were): = / X operators beside M, and M / = / X
    # operators of length zero (the iterator's cursor), on some scenarios
    rng2 = np.random.default_rng(seed + 555555)
    u = rng2.random()
    if u < 0.15:
        arrs = synth.eqx_cigars(arrs, seed=seed, frac=0.7, keep_m=float(rng2.choice([0.0, 0.3])))
    elif u < 0.27:
        arrs = synth.inject_empty_mops(arrs, seed=seed, frac=0.5)
    return ref, arrs, regions, kw, clear, style


@pytest.mark.parametrize("block", range(4))
def test_random_scenarios_equal_reference_compiled(oracle_lib, sim_lib, ref_lib, block):
    """Differential fuzz against the reference's own code: random data styles, libraries, option sets (-q -b -i -p -d), lists of
    regions in any order (empty, single-base, abutting, overlapping, whole contig), deletion queues cleared between them or
    not — the oracle and every text route of the product (dense planes, compact planes, device-side text) print what the
    reference-compiled library prints.  (The reference string is longer than any read reaches: past its end the reference
    reads out of bounds, see DESIGN.md.)"""
    for seed in range(block * 12, block * 12 + 12):
        ref, arrs, regions, kw, clear, style = random_scenario(seed)
        want, _ = parity.run_engine(ref_lib, arrs, regions, ref=ref, clear_queue=clear, **kw)
        for lib, route in ((oracle_lib, {}), (sim_lib, {}), (sim_lib, dict(text_only=True)), (sim_lib, dict(device_text="chrS"))):
            got, _ = parity.run_engine(lib, arrs, regions, ref=ref, clear_queue=clear, **route, **kw)
            assert got == want, (seed, style, kw, regions, clear, route)


def test_cli_coordinates_beyond_the_bai_limit_equal_reference_main(ref_lib, tmp_path):
    """Reads past 2^29 on a contig that needs a CSI index (depth 6): regions, a bare start and a site list print what the
    reference's main() prints."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bamio
    from test_cli import SIM_CLI
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    d = tmp_path
    OFF = (1 << 29) + 1000; CL = OFF + 5000
    rng = np.random.default_rng(1)
    small = synth.make_ref(rng, 3600)
    arrs = synth.make_batch(5, small, 400, style="mixed", n_libs=2, region=(0, 3000))
    arrs["pos"] = (arrs["pos"].astype(np.int64) + OFF).astype(arrs["pos"].dtype)
    rgs = [["rg0", "rg1"][int(l)] if l >= 0 else None for l in arrs["lib"]]
    bamio.write_bam(str(d / "x.bam"), [("chrBig", CL)], arrs, np.zeros(len(arrs["pos"]), int), rg_of_read=rgs,
                    rg_lines=["@RG\tID:rg0\tLB:libA\tSM:s", "@RG\tID:rg1\tLB:libB\tSM:s"], csi=(14, 6))
    with open(d / "r.fa", "wb") as f:                     # one line: N up to the reads, the reads' reference, N to the end
        f.write(b">chrBig\n"); left = OFF; chunk = b"N" * (1 << 22)
        while left > 0:
            f.write(chunk[:min(left, len(chunk))]); left -= min(left, len(chunk))
        f.write(bytes(small)); f.write(b"N" * (CL - OFF - len(small))); f.write(b"\n")
    open(d / "r.fa.fai", "w").write("chrBig\t%d\t8\t%d\t%d\n" % (CL, CL, CL + 1))
    open(d / "s.txt", "w").write("".join("chrBig\t%d\t%d\n" % (OFF + x, OFF + x + 2) for x in (5, 900, 901, 2500)))
    for args in (["x.bam", "chrBig:%d-%d" % (OFF + 100, OFF + 600)], ["-p", "x.bam", "chrBig:%d" % (OFF + 2000)], ["-l", "s.txt", "x.bam"]):
        a = subprocess.run([REF_CLI, "-w", "1", "-f", "r.fa"] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        b = subprocess.run([SIM_CLI, "-w", "1", "-f", "r.fa"] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert (a.returncode, a.stdout == b.stdout, a.stderr == b.stderr) == (b.returncode, True, True), args
        assert a.returncode == 0 and a.stdout.count(b"\n") >= 12


def _long_read_check(lib, ref_lib, seeds):
    """Reads of 3-20 kb with an insertion or deletion every dozen bases (a thousand and more CIGAR operators each, skips
    of up to 200 bases): one region over everything and a narrow one inside."""
    for seed in seeds:
        rng = np.random.default_rng(seed)
        RL = 30000
        ref = synth.make_ref(rng, RL + 45000)
        n_libs = int(rng.choice([1, 2]))
        arrs = synth.make_batch(seed + 9000, ref, int(rng.integers(10, 30)), style="dense", n_libs=n_libs, read_len=(3000, 20000), region=(0, RL))
        assert len(arrs["cigar"]) > 200 * len(arrs["pos"])
        kw = dict(min_bq=int(rng.choice([0, 10])), insertion_centric=bool(rng.random() < 0.5))
        if rng.random() < 0.5:
            kw.update(per_lib=True, lib_names=["lib%c" % (65 + i) for i in range(n_libs)])
        regions = [(0, RL + 25000), (12000, 12800)]
        want, _ = parity.run_engine(ref_lib, arrs, regions, ref=ref, clear_queue=False, **kw)
        for route in (dict(), dict(text_only=True), dict(device_text="chrS")):
            got, _ = parity.run_engine(lib, arrs, regions, ref=ref, clear_queue=False, **route, **kw)
            assert got == want, (seed, kw, route)


def test_long_dense_indel_reads_equal_reference_compiled(sim_lib, ref_lib):
    _long_read_check(sim_lib, ref_lib, range(1))


@pytest.mark.gpu
def test_long_dense_indel_reads_equal_reference_compiled_gpu(hip_lib, ref_lib):
    _long_read_check(hip_lib, ref_lib, range(10, 13))


def _extreme_case(seed):
    """Reads with values at the edges of their BAM fields: base qualities 0 / 93 / 255, MAPQ 0 / 255, NM / SM of any int32
    (negative too), proper pairs without SM, long reads (2 kb, many operators), thresholds that sit on those edges."""
    rng = np.random.default_rng(seed)
    RL = int(rng.integers(300, 1500))
    ref = synth.make_ref(rng, RL + 3000, weird=float(rng.choice([0, 0.02])))
    n_libs = int(rng.choice([1, 2]))
    long_reads = rng.random() < 0.3
    arrs = synth.make_batch(seed + 7000, ref, int(rng.integers(30, 300)), style=str(rng.choice(["simple", "indel", "wild", "mixed"])), n_libs=n_libs,
                            read_len=(200, 2000) if long_reads else (20, 150), region=(0, RL))
    n = len(arrs["pos"]); q = arrs["qual"]
    if rng.random() < 0.5:
        idx = rng.integers(0, len(q), max(1, len(q) // 10)); q[idx] = rng.choice([0, 1, 2, 93, 94, 127, 128, 200, 254, 255], len(idx)).astype(q.dtype)
    if rng.random() < 0.5:
        idx = rng.integers(0, n, max(1, n // 5)); arrs["mapq"][idx] = rng.choice([0, 1, 254, 255], len(idx)).astype(arrs["mapq"].dtype)
    if rng.random() < 0.5:
        idx = rng.integers(0, n, max(1, n // 5)); arrs["nm"][idx] = rng.choice([0, 1, 255, 256, 65535, 70000, 2**31 - 1, -1, -5], len(idx)).astype(arrs["nm"].dtype)
    if rng.random() < 0.5:
        idx = rng.integers(0, n, max(1, n // 5)); arrs["sm"][idx] = rng.choice([0, 255, 256, 100000, 2**31 - 1, -1], len(idx)).astype(arrs["sm"].dtype)
    if rng.random() < 0.3:
        arrs["flag"][rng.integers(0, n, max(1, n // 3))] |= 2
    kw = dict(min_mapq=int(rng.choice([0, 1, 255])), min_bq=int(rng.choice([0, 2, 94, 255])), insertion_centric=bool(rng.random() < 0.4))
    if rng.random() < 0.4:
        kw.update(per_lib=True, lib_names=["lib%c" % (65 + i) for i in range(n_libs)])
    a0 = int(rng.integers(0, RL))
    return ref, arrs, [(0, RL + 50), (a0, a0 + 300)], kw


def _extreme_check(lib, ref_lib, seeds, oracle_lib=None):
    for seed in seeds:
        ref, arrs, regions, kw = _extreme_case(seed)
        want, _ = parity.run_engine(ref_lib, arrs, regions, ref=ref, clear_queue=False, **kw)
        routes = [(lib, {}), (lib, dict(text_only=True)), (lib, dict(device_text="chrS"))] + ([(oracle_lib, {})] if oracle_lib is not None else [])
        for l, route in routes:
            got, _ = parity.run_engine(l, arrs, regions, ref=ref, clear_queue=False, **route, **kw)
            assert got == want, (seed, kw, route)


def test_extreme_field_values_equal_reference_compiled(oracle_lib, sim_lib, ref_lib):
    _extreme_check(sim_lib, ref_lib, range(24), oracle_lib)


@pytest.mark.gpu
def test_extreme_field_values_equal_reference_compiled_gpu(hip_lib, ref_lib):
    _extreme_check(hip_lib, ref_lib, range(100, 112))


def _random_cli_case(seed, d):
    """A random two-contig BAM (+ .bai, FASTA) and a random bam-readcount command line over it; returns (reference options,
    drop-in extras, positional arguments, environment of the drop-in)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bamio
    from test_cli import _write_fasta
    rng = np.random.default_rng(seed)
    L0 = int(rng.integers(800, 6000)); L1 = int(rng.integers(500, 3000))
    refs = [synth.make_ref(rng, L0 + 600, weird=float(rng.choice([0, 0.03]))), synth.make_ref(rng, L1 + 600)]      # (longer than any read reaches)
    nl = int(rng.choice([1, 2, 3]))
    parts = [synth.make_batch(seed * 2 + 1, refs[0], int(rng.integers(100, 1500)), style=str(rng.choice(["simple", "indel", "wild", "mixed"])), n_libs=nl, region=(0, L0), p_nolib=float(rng.choice([0, 0.03]))),
             synth.make_batch(seed * 2 + 2, refs[1], int(rng.integers(50, 600)), style=str(rng.choice(["simple", "indel", "mixed"])), n_libs=nl, region=(0, L1))]
    arrs = {}
    for k in ("pos", "flag", "mapq", "lib", "l_qseq", "n_cigar", "nm", "sm", "tags"):
        arrs[k] = np.concatenate([p[k] for p in parts])
    for arena, off in (("cigar", "cigar_off"), ("seq4", "seq_off"), ("qual", "qual_off")):
        arrs[arena] = np.concatenate([p[arena] for p in parts]); arrs[off] = np.concatenate([parts[0][off], parts[1][off] + np.uint64(parts[0][arena].size)])
    tids = np.concatenate([np.zeros(len(parts[0]["pos"]), int), np.ones(len(parts[1]["pos"]), int)])
    rg_ids = ["rg%d" % i for i in range(nl)]
    rgs = [rg_ids[int(l)] if l >= 0 else None for l in arrs["lib"]]
    bamio.write_bam(os.path.join(d, "x.bam"), [("chrA", L0 + 600), ("chrB", L1 + 600)], arrs, tids, rg_of_read=rgs,
                    rg_lines=["@RG\tID:%s\tLB:lib%c\tSM:s" % (r, 65 + i) for i, r in enumerate(rg_ids)], block_bytes=int(rng.choice([3000, 20000])),
                    int_types=[str(t) for t in rng.choice(list("cCsSiI"), len(arrs["pos"]))])
    _write_fasta(os.path.join(d, "r.fa"), [("chrA", refs[0]), ("chrB", refs[1])])
    opts = []
    if rng.random() < 0.5: opts += ["-q", str(int(rng.choice([1, 10, 30])))]
    if rng.random() < 0.5: opts += ["-b", str(int(rng.choice([5, 13, 25])))]
    if rng.random() < 0.4: opts += ["-p"]
    if rng.random() < 0.4: opts += ["-i"]
    if rng.random() < 0.15: opts += ["-d", str(int(rng.choice([2, 10])))]
    opts += ["-w", str(int(rng.choice([0, 1, 3, -1]))), "-f", "r.fa"]          # (without -f the reference dereferences a null pointer)

    def region():
        c = str(rng.choice(["chrA", "chrB"])); Lc = L0 if c == "chrA" else L1
        k = rng.random()
        if k < 0.2: return c
        a = int(rng.integers(1, Lc))
        return "%s:%d" % (c, a) if k < 0.35 else "%s:%d-%d" % (c, a, a + int(rng.choice([0, 1, 20, 300, Lc])))
    if rng.random() < 0.5:
        args = ["x.bam"] + [region() for _ in range(int(rng.integers(1, 5)))]
    else:
        lines = []
        for _ in range(int(rng.integers(1, 40))):
            c = "chrZ" if rng.random() < 0.05 else str(rng.choice(["chrA", "chrB"]))
            Lc = L0 if c == "chrA" else L1; a = int(rng.integers(1, Lc)); lines.append("%s\t%d\t%d\n" % (c, a, a + int(rng.choice([0, 0, 0, 1, 30, 400]))))
        if rng.random() < 0.5: lines.sort(key=lambda t: (t.split("\t")[0], int(t.split("\t")[1])))
        open(os.path.join(d, "s.txt"), "w").write("".join(lines)); args = ["-l", "s.txt", "x.bam"]
    extra = []
    if rng.random() < 0.5: extra += ["--brc-chunk", str(int(rng.choice([64, 333, 5000])))]
    if rng.random() < 0.3: extra += ["--brc-gpus", str(int(rng.choice([2, 3])))]
    if rng.random() < 0.2: extra += ["--brc-plan", str(int(rng.choice([0, 3])))]
    env = dict(os.environ, BRC_DEVICE_TEXT_MAX_SHARE=str(rng.choice(["100", "0.06"])))
    if rng.random() < 0.3:
        env["BRC_FETCH_STRIPE_MIN"] = "1"; env["BRC_FETCH_THREADS"] = str(int(rng.choice([2, 5])))
    # one process per GPU (--brc-ranks; a random stream of its own: the cases of the seeds above stay what they were)
    rng2 = np.random.default_rng(seed + 777777)
    if "--brc-gpus" not in extra and rng2.random() < 0.3:
        extra += ["--brc-ranks", str(int(rng2.choice([2, 3])))]; env["BRC_RANK_CUT"] = str(int(rng2.choice([97, 500, 65536])))
    return opts, extra, args, env


def _cli_fuzz(cli, ref_lib, tmp_path, seeds, one_device=False):
    for seed in seeds:
        d = tmp_path / ("s%d" % seed); d.mkdir()
        opts, extra, args, env = _random_cli_case(seed, str(d))
        if one_device and "--brc-gpus" in extra:          # the GPU box has one device: several engines on it
            i = extra.index("--brc-gpus"); env["BRC_DEVICES"] = ",".join(["0"] * int(extra[i + 1])); del extra[i:i + 2]
        if one_device and "--brc-ranks" in extra:         # ... or several processes, each with a context of its own on it
            env["BRC_DEVICES"] = ",".join(["0"] * int(extra[extra.index("--brc-ranks") + 1]))
        a = subprocess.run([REF_CLI] + opts + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        b = subprocess.run([cli] + opts + extra + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert (a.returncode, a.stdout == b.stdout, a.stderr == b.stderr) == (b.returncode, True, True), (seed, opts, extra, args)


def _weird_site_line(rng):
    c = str(rng.choice(["chrA", "chrB", "chrZ", "", "chrA ", "*"]))
    a = int(rng.integers(1, 900)); b = a + int(rng.choice([0, 1, 5, 50]))
    forms = ["%s\t%d\t%d" % (c, a, b), "%s %d %d" % (c, a, b), "%s\t%d\t%d\textra\tcols" % (c, a, b), "%s\t%d\t%d\r" % (c, a, b),
             "%s\t%d" % (c, a), "%s" % c, "", "\t\t", "%s\t%d\t%d" % (c, b + 3, a), "%s\t%d\t%d" % (c, a + 1, a - 1), "%s\t0\t%d" % (c, b), "%s\t-5\t%d" % (c, b),
             "%s\t%d\t-1" % (c, a), "%s\t%d.5\t%d" % (c, a, b), "%s\t1e2\t%d" % (c, b), "%s\tx\t%d" % (c, b), "%s\t%d\ty" % (c, a), "%s\t+%d\t%d" % (c, a, b),
             "%s\t%d\t99999999999" % (c, a), "%s\t99999999999\t%d" % (c, b), "%s\t%d\t%d" % (c, a, 2**31 - 1), "%s\t%d\t%d" % (c, a, 2**31),
             "  %s\t%d\t%d" % (c, a, b), "%s\t\t%d\t%d" % (c, a, b), "#%s\t%d\t%d" % (c, a, b), "%s\t0x10\t%d" % (c, b), "%s\t010\t%d" % (c, b),
             "%s\t%d\t%d abc" % (c, a, b), "%s\t%d\t%dabc" % (c, a, b), "%s\t%d,000\t%d" % (c, a, b), "%s\t%d-%d" % (c, a, b)]
    return forms[int(rng.integers(0, len(forms)))]


def _weird_region(rng):
    c = str(rng.choice(["chrA", "chrB", "chrZ", "", "chrA ", "*", "chra"]))
    a = int(rng.integers(1, 900)); b = a + int(rng.choice([0, 1, 5, 50]))
    forms = ["%s:%d-%d" % (c, a, b), "%s" % c, "%s:" % c, "%s:%d" % (c, a), "%s:%d-" % (c, a), "%s:-%d" % (c, b), "%s:%d-%d" % (c, b + 2, a),
             "%s:%d-%d" % (c, a, a), "%s:0-%d" % (c, b), "%s:1,0%02d-2,000" % (c, a % 100), "%s:abc" % c, "%s:%d-%d-%d" % (c, a, b, b + 5),
             "%s:1e2-%d" % (c, b), " %s:%d-%d" % (c, a, b), "%s:%d - %d" % (c, a, b), "%s:%d-%d " % (c, a, b), "%s:%d-99999999999" % (c, a),
             "%s:99999999999-%d" % (c, b), "%s:%d-2147483647" % (c, a), "%s:%d-2147483648" % (c, a), "%s:+%d-%d" % (c, a, b), "%s:%d.5-%d" % (c, a, b),
             "%s:%d-%dx" % (c, a, b), "%s:-" % c, "%s::%d-%d" % (c, a, b), ":%d-%d" % (a, b), "%s:%d--%d" % (c, a, b), "%s:0" % c, "%s:0-0" % c, "%s:1-1" % c,
             "%s: %d- %d" % (c, a, b), "%s:0.%dk-1k" % (c, a % 10), "%s:1K" % c]
    return forms[int(rng.integers(0, len(forms)))]


def test_cli_malformed_site_lines_and_regions_equal_reference_main(ref_lib, tmp_path):
    """The reference reads a site-list line with `stringstream >> name >> int >> int` (bamreadcount.cpp:574-577: any white
    space, signs, a number cut at the first other character, overflow = no line) and works on [beg - 2, end) (:268); a
    command-line region goes through samtools' bam_parse_region (:644; restated in oracle/ref_shim/shim_hts.cpp from
    htslib-1.10's hts_parse_reg / hts_parse_decimal: last colon, k/M/G, fractions, notes on stderr).  Malformed, reversed,
    overflowing and unknown-contig input: stdout, stderr and the exit code of the drop-in are the reference's."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    from test_cli import SIM_CLI
    d = tmp_path / "case"; d.mkdir()
    _random_cli_case(4242, str(d))
    rng = np.random.default_rng(11)
    for it in range(120):
        o = ["-w", "2", "-f", "r.fa"] + (["-p"] if rng.random() < 0.3 else [])
        if rng.random() < 0.2: o += ["-d", str(int(rng.choice([0, -3, 1])))]       # (:592,:651 hand any value to the iterator)
        if it % 2 == 0:
            txt = "\n".join(_weird_site_line(rng) for _ in range(int(rng.integers(1, 12)))) + ("" if rng.random() < 0.3 else "\n")
            open(d / "w.txt", "w", newline="").write(txt)
            args = ["-l", "w.txt", "x.bam"]
        else:
            txt = None; args = ["x.bam"] + [_weird_region(rng) for _ in range(int(rng.integers(1, 4)))]
        a = subprocess.run([REF_CLI] + o + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        b = subprocess.run([SIM_CLI] + o + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert (a.returncode, a.stdout == b.stdout, a.stderr == b.stderr) == (b.returncode, True, True), (it, txt, args)


def test_cli_read_group_header_quirks_equal_reference_main(ref_lib, tmp_path):
    """@RG lines as the reference sees them: find_library_names (bamreadcount.cpp:92-111) skips the first tag of every line
    and takes every later "LB" one (its "Expect library" lines); per read, samtools' bam_get_library (:280, restated in the
    shim from bam.c) wants an ID that is followed by a tab and takes the last LB of the first matching line.  LB written
    first, twice, empty or missing, duplicate IDs, reads naming an unknown group, a comment that mentions LB."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bamio
    from test_cli import SIM_CLI, _write_fasta
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    for seed in range(24):
        rng = np.random.default_rng(seed)
        d = tmp_path / ("h%d" % seed); d.mkdir()
        L = 1200
        ref = synth.make_ref(rng, L + 600)
        arrs = synth.make_batch(seed, ref, 250, style="mixed", n_libs=3, region=(0, L), p_nolib=0.05)
        ids = ["rg0", "rg1", "rg2"]

        def rgline(i):
            k = rng.random(); lb = str(rng.choice(["libA", "libB", "libC", "", "lib A", "LB"]))
            if k < 0.15: return "@RG\tLB:%s\tID:%s" % (lb, ids[i])
            if k < 0.30: return "@RG\tID:%s\tSM:s" % ids[i]
            if k < 0.40: return "@RG\tID:%s\tLB:%s\tLB:other" % (ids[i], lb)
            if k < 0.50: return "@RG\tID:%s\tPL:x\tLB:%s\tSM:s" % (ids[i], lb)
            if k < 0.55: return "@RG\tSM:s\tID:%s\tLB:%s" % (ids[i], lb)
            return "@RG\tID:%s\tLB:%s\tSM:s" % (ids[i], lb)
        lines = [rgline(i) for i in range(3)]
        if rng.random() < 0.2: lines.append("@RG\tID:rg0\tLB:dup")
        if rng.random() < 0.2: lines = lines[:2]
        if rng.random() < 0.1: lines = []
        if rng.random() < 0.3: lines.insert(0, "@CO\tsome comment LB:fake")
        rgs = [ids[int(l)] if l >= 0 else None for l in arrs["lib"]]
        bamio.write_bam(str(d / "x.bam"), [("chrA", L + 600)], arrs, np.zeros(len(arrs["pos"]), int), rg_of_read=rgs, rg_lines=lines)
        _write_fasta(d / "r.fa", [("chrA", ref)])
        for o in (["-p"], [], ["-p", "-i"]):
            args = ["-w", "2", "-f", "r.fa"] + o + ["x.bam", "chrA:100-400"]
            a = subprocess.run([REF_CLI] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            b = subprocess.run([SIM_CLI] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert (a.returncode, a.stdout == b.stdout, a.stderr == b.stderr) == (b.returncode, True, True), (seed, o, lines)


@pytest.mark.parametrize("block", range(3))
def test_cli_random_command_lines_equal_reference_main(ref_lib, tmp_path, block):
    """Differential fuzz of the whole command line against the reference's own main(): random BAMs (all CIGAR operators, reads
    without library / tags), options, regions in any order or -l lists (unsorted, unknown contigs, wide lines), warning caps —
    and, on the drop-in's side, random piece sizes, engine counts, planner settings, striped fetches and text routes.
    stdout, stderr and the exit code must be the reference's."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    from test_cli import SIM_CLI
    _cli_fuzz(SIM_CLI, ref_lib, tmp_path, range(block * 10, block * 10 + 10))


@pytest.mark.gpu
def test_cli_random_command_lines_equal_reference_main_gpu(ref_lib, tmp_path):
    from test_cli import HIP_CLI
    _cli_fuzz(HIP_CLI, ref_lib, tmp_path, range(100, 112), one_device=True)


def _empty_mop_case():
    rng = np.random.default_rng(5)
    ref = synth.make_ref(rng, 3000, weird=0.01)
    base = synth.make_batch(21, ref, 400, read_len=(40, 160), style="indel", n_libs=2)
    return ref, synth.inject_empty_mops(base, seed=3, frac=0.6)


def test_empty_m_operators_equal_the_reference_s_own_sources(oracle_lib, sim_lib, ref_lib):
    """CIGARs with M / = / X operators of length zero (round 6: piled up by the iterator's cursor instead of refused): the reference's own
    fetch_func / pileup_func over the shim's pileup iterator against the oracle and every text route of the device algorithm."""
    ref, arrs = _empty_mop_case()
    regions = [(0, 3000), (700, 900)]
    for kw in (dict(), dict(insertion_centric=True, min_mapq=10, min_bq=8), dict(per_lib=True, lib_names=["libA", "libB"], insertion_centric=True)):
        want, _ = parity.run_engine(ref_lib, arrs, regions, ref=ref, **kw)
        assert parity.run_engine(oracle_lib, arrs, regions, ref=ref, **kw)[0] == want, kw
        for route in (dict(), dict(text_only=True), dict(device_text="chrS")):
            assert parity.run_engine(sim_lib, arrs, regions, ref=ref, **route, **kw)[0] == want, (kw, route)


@pytest.mark.gpu
def test_empty_m_operators_equal_the_reference_s_own_sources_gpu(hip_lib, oracle_lib, ref_lib):
    ref, arrs = _empty_mop_case()
    regions = [(0, 3000), (700, 900)]
    for kw in (dict(), dict(insertion_centric=True, min_mapq=10, min_bq=8), dict(per_lib=True, lib_names=["libA", "libB"], insertion_centric=True)):
        want, _ = parity.run_engine(ref_lib, arrs, regions, ref=ref, **kw)
        for route in (dict(), dict(text_only=True), dict(device_text="chrS")):
            assert parity.run_engine(hip_lib, arrs, regions, ref=ref, **route, **kw)[0] == want, (kw, route)
    parity.compare_libs(hip_lib, oracle_lib, arrs, regions, ref=ref)
