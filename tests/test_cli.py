"""The reference's six integration tests (integration-test/bam-readcount_test.py:29-116) re-expressed for the drop-in
command line: same arguments, byte-exact stdout against the reference's own golden files.

CPU: the CLI linked against the lane simulator (tests/sim/bam-readcount-sim) exercises option parsing, BGZF/BAM/BAI/
FASTA input and the region plumbing end-to-end.  GPU (-m gpu): the product binary bam_readcount_amd/csrc/bam-readcount."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

SIM_CLI = os.path.join(ROOT, "tests", "sim", "bam-readcount-sim")
HIP_CLI = os.path.join(ROOT, "bam_readcount_amd", "csrc", "bam-readcount")

RUNS = [  # (expected file, bam, extra args, how the sites are given)
    ("expected_all_lib", "test.bam", [], "list"),
    ("expected_per_lib", "test.bam", ["-p"], "list"),
    ("expected_all_lib", "test.bam", [], "regions"),
    ("expected_all_lib", "test_bad_rg.bam", [], "list"),
    ("expected_insertion_centric_all_lib", "test.bam", ["-i"], "list"),
    ("expected_insertion_centric_per_lib", "test.bam", ["-i", "-p"], "list"),
]


def run_cli(exe, workdir, bam, extra, how):
    args = [exe, "-w", "1"] + extra + ["-f", "ref.fa"]
    if how == "list":
        args += ["-l", "site_list", bam]
    else:
        args += [bam, "21:10402985-10402985", "21:10405200-10405200"]
    p = subprocess.run(args, cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return p.returncode, p.stdout, p.stderr.decode()


def check(exe, workdir):
    for exp, bam, extra, how in RUNS:
        rc, out, err = run_cli(exe, workdir, bam, extra, how)
        assert rc == 0, err
        assert out == open(os.path.join(GOLDEN, exp), "rb").read(), (exp, bam, extra, how)
        assert "Minimum mapping quality is set to 0" in err
        if bam == "test.bam":
            assert "Expect library: Solexa-135852 in BAM" in err and "Expect library: Solexa-135853 in BAM" in err


def test_cli_reference_integration_tests_cpu(workdir):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    check(SIM_CLI, workdir)


def test_cli_option_spellings_and_errors_cpu(workdir):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    exp = open(os.path.join(GOLDEN, "expected_insertion_centric_per_lib"), "rb").read()
    for extra in (["-pi"], ["--per-library", "--insertion-centric"], ["--per-lib", "--insertion-c"], ["-i", "-p", "-q0", "-b", "0", "--max-count=10000000"]):
        rc, out, _ = run_cli(SIM_CLI, workdir, "test.bam", extra, "list")
        assert rc == 0 and out == exp, extra
    # -h / -v print to stdout and return 1 (bamreadcount.cpp:467-475); missing index / unknown contig messages
    p = subprocess.run([SIM_CLI, "-h"], stdout=subprocess.PIPE)
    assert p.returncode == 1 and b"Usage: bam-readcount [OPTIONS] bam_file|cram_file [region]" in p.stdout
    p = subprocess.run([SIM_CLI, "-v"], stdout=subprocess.PIPE)
    assert p.returncode == 1 and p.stdout.startswith(b"bam-readcount version: ")
    p = subprocess.run([SIM_CLI, "-f", "ref.fa", "test.bam", "nochr:1-10"], cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"Invalid region nochr:1-10" in p.stderr
    open(workdir / "sl2", "w").write("junk line\nchrZ 5 6\n21 10402985 10402985\n")
    p = subprocess.run([SIM_CLI, "-w1", "-f", "ref.fa", "-l", "sl2", "test.bam"], cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and b"chrZ not found in bam file. Region chrZ 5 6 skipped." in p.stderr
    assert p.stdout == open(os.path.join(GOLDEN, "expected_all_lib"), "rb").read().split(b"\n")[0] + b"\n"
    # a thresholded run and a multi-kilobase region, chunked vs unchunked: identical text
    a = subprocess.run([SIM_CLI, "-q", "20", "-b", "13", "-f", "ref.fa", "test.bam", "21:10402737-10405248"], cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    b = subprocess.run([SIM_CLI, "-q", "20", "-b", "13", "--brc-chunk", "500", "-f", "ref.fa", "test.bam", "21:10402737-10405248"], cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert a.returncode == 0 and b.returncode == 0 and a.stdout == b.stdout and a.stdout.count(b"\n") > 700


@pytest.mark.gpu
def test_cli_reference_integration_tests_gpu(workdir):
    if not os.path.exists(HIP_CLI):      # host-only link step against the (already built) engine library
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(HIP_CLI), "bam-readcount"])
    check(HIP_CLI, workdir)


def _write_fasta(path, name_seq):
    with open(path, "wb") as f, open(str(path) + ".fai", "w") as fai:
        off = 0
        for name, seq in name_seq:
            hdr = b">" + name.encode() + b"\n"; f.write(hdr); off += len(hdr)
            fai.write("%s\t%d\t%d\t60\t61\n" % (name, len(seq), off))
            for i in range(0, len(seq), 60):
                f.write(bytes(seq[i:i + 60]) + b"\n")
            off += len(seq) + (len(seq) + 59) // 60


@pytest.fixture(scope="session")
def synthetic_bam(tmp_path_factory):
    """Two contigs of synthetic reads with every CIGAR operator, two libraries (one RG without LB), written as a real
    multi-block BAM + BAI (tools/bamio.write_bam) and FASTA + .fai."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bamio
    import synth
    d = tmp_path_factory.mktemp("synbam")
    rng = np.random.default_rng(5)
    refs = [synth.make_ref(rng, 5000, weird=0.01), synth.make_ref(rng, 3000)]
    parts = [synth.make_batch(31, refs[0], 1500, style="mixed", n_libs=2), synth.make_batch(32, refs[1], 600, style="wild", n_libs=2)]
    from bam_readcount_amd import capi
    arrs = {}
    for k in ("pos", "flag", "mapq", "lib", "l_qseq", "n_cigar", "nm", "sm", "tags"):
        arrs[k] = np.concatenate([p[k] for p in parts])
    for arena, off, unit in (("cigar", "cigar_off", None), ("seq4", "seq_off", None), ("qual", "qual_off", None)):
        arrs[arena] = np.concatenate([p[arena] for p in parts])
        arrs[off] = np.concatenate([parts[0][off], parts[1][off] + np.uint64(parts[0][arena].size)])
    tids = np.concatenate([np.zeros(len(parts[0]["pos"]), int), np.ones(len(parts[1]["pos"]), int)])
    rgs = [["rgA1", "rgB1"][int(l)] if l >= 0 else None for l in arrs["lib"]]
    bamio.write_bam(str(d / "syn.bam"), [("chrA", 5000), ("chrB", 3000)], arrs, tids, rg_of_read=rgs,
                    rg_lines=["@RG\tID:rgA1\tLB:libA\tSM:s", "@RG\tID:rgB1\tLB:libB\tSM:s"], block_bytes=6000)
    _write_fasta(d / "syn.fa", [("chrA", refs[0]), ("chrB", refs[1])])
    # CRAM does not keep '=' / 'X' operators (they come back as 'M', adjacent runs merged) and the reference's annotator
    # treats them differently from 'M' (bamreadcount.cpp:133-200 only walks BAM_CMATCH), so the BAM twin of the CRAM is
    # written from the normalised CIGARs
    norm = dict(arrs); cig = []; ncs = []
    for i in range(len(arrs["pos"])):
        ops = []
        for c in arrs["cigar"][int(arrs["cigar_off"][i]):int(arrs["cigar_off"][i]) + int(arrs["n_cigar"][i])]:
            op, ln = int(c) & 15, int(c) >> 4
            if op in (7, 8): op = 0
            if ops and ops[-1][0] == op: ops[-1][1] += ln
            else: ops.append([op, ln])
        cig += [(ln << 4) | op for op, ln in ops]; ncs.append(len(ops))
    norm["cigar"] = np.array(cig, np.uint32); norm["n_cigar"] = np.array(ncs, np.uint32)
    norm["cigar_off"] = np.concatenate([[0], np.cumsum(ncs)[:-1]]).astype(np.uint64)
    # ... and with the NM tag htslib generates when it decodes a mapped CRAM record that was stored without one (samtools
    # drops NM / MD on the way into CRAM): substituted, inserted and deleted bases
    norm["tags"] = arrs["tags"].copy(); norm["nm"] = arrs["nm"].copy()
    for i in range(len(arrs["pos"])):
        if (int(arrs["tags"][i]) & 1) or (int(arrs["flag"][i]) & 4):
            continue
        norm["tags"][i] |= 1; norm["nm"][i] = _cram_nm(norm, i, refs[int(tids[i])])
    bamio.write_bam(str(d / "syn_m.bam"), [("chrA", 5000), ("chrB", 3000)], norm, tids, rg_of_read=rgs,
                    rg_lines=["@RG\tID:rgA1\tLB:libA\tSM:s", "@RG\tID:rgB1\tLB:libB\tSM:s"], block_bytes=6000)
    import cramio
    cramio.write_cram(str(d / "syn.cram"), [("chrA", 5000), ("chrB", 3000)], arrs, tids, refs, rg_of_read=rgs,
                      rg_lines=["@RG\tID:rgA1\tLB:libA\tSM:s", "@RG\tID:rgB1\tLB:libB\tSM:s"], per_container=280)
    # the same reads with every block compression method (raw, gzip, bzip2, lzma, rANS 4x8 order 0 and order 1; the core
    # bit stream compressed too) and the GAMMA / SUBEXP integer codecs
    cramio.write_cram(str(d / "syn_rans.cram"), [("chrA", 5000), ("chrB", 3000)], arrs, tids, refs, rg_of_read=rgs,
                      rg_lines=["@RG\tID:rgA1\tLB:libA\tSM:s", "@RG\tID:rgB1\tLB:libB\tSM:s"], per_container=310,
                      methods=(4, 5, 0, 1, 2, 3, 5), int_codecs=True)
    # embedded reference slices, and a reference-less file (RR = 0, every base a feature): neither needs the FASTA to rebuild
    # its reads — syn_alt.fa differs from the reference they were written against at every 50th base
    cramio.write_cram(str(d / "syn_emb.cram"), [("chrA", 5000), ("chrB", 3000)], arrs, tids, refs, rg_of_read=rgs,
                      rg_lines=["@RG\tID:rgA1\tLB:libA\tSM:s", "@RG\tID:rgB1\tLB:libB\tSM:s"], per_container=280, embed_ref=True, write_crai=True)
    cramio.write_cram(str(d / "syn_noref.cram"), [("chrA", 5000), ("chrB", 3000)], arrs, tids, refs, rg_of_read=rgs,
                      rg_lines=["@RG\tID:rgA1\tLB:libA\tSM:s", "@RG\tID:rgB1\tLB:libB\tSM:s"], per_container=280, no_ref=True)
    bamio.write_bam(str(d / "syn_m_nonm.bam"), [("chrA", 5000), ("chrB", 3000)], dict(norm, tags=arrs["tags"], nm=arrs["nm"]), tids, rg_of_read=rgs,
                    rg_lines=["@RG\tID:rgA1\tLB:libA\tSM:s", "@RG\tID:rgB1\tLB:libB\tSM:s"], block_bytes=6000)
    alt = [r.copy() for r in refs]
    for r in alt:
        r[::50] = np.frombuffer(bytes({65: 67, 67: 71, 71: 84, 84: 65}.get(int(b) & ~32, 65) for b in r[::50]), np.uint8)
    _write_fasta(d / "syn_alt.fa", [("chrA", alt[0]), ("chrB", alt[1])])
    return d


def _cram_nm(arrs, i, ref):
    """NM as htslib's CRAM decoder counts it (cram_decode.c cram_decode_seq): mismatching bases of the match operators
    against the upper-cased reference ('N' past its end), plus inserted and deleted bases."""
    NT16 = "=ACMGRSVTWYHKDBN"
    L = int(arrs["l_qseq"][i]); s4 = arrs["seq4"][int(arrs["seq_off"][i]):int(arrs["seq_off"][i]) + (L + 1) // 2]
    seq = [NT16[(int(s4[j >> 1]) >> (4 if j % 2 == 0 else 0)) & 15] for j in range(L)]
    rp = int(arrs["pos"][i]); sp = 0; nm = 0
    for c in arrs["cigar"][int(arrs["cigar_off"][i]):int(arrs["cigar_off"][i]) + int(arrs["n_cigar"][i])]:
        op, ln = int(c) & 15, int(c) >> 4
        if op in (0, 7, 8):
            for j in range(ln):
                rb = chr(ref[rp + j]).upper() if 0 <= rp + j < len(ref) else "N"
                nm += seq[sp + j] != rb
            rp += ln; sp += ln
        elif op == 1: nm += ln; sp += ln
        elif op == 4: sp += ln
        elif op == 2: nm += ln if rp + ln <= len(ref) else max(len(ref) - rp, 0); rp += ln
        elif op == 3: rp += ln
    return nm


def _sites_file(d, name, sites):
    open(d / name, "w").write("".join("%s\t%d\t%d\n" % s for s in sites))
    return name


def _planner_check(cli, synthetic_bam):
    """n3: the batched virtual-axis planner must print exactly what one engine pass per -l line prints — including
    duplicate lines, overlapping ranges, unsorted order, both contigs, per-library and insertion-centric modes."""
    d = synthetic_bam
    rng = np.random.default_rng(9)
    sites = [("chrA", int(p), int(p)) for p in rng.integers(1, 5000, 120)] + [("chrB", int(p), int(p) + int(w)) for p, w in zip(rng.integers(1, 2900, 40), rng.integers(0, 60, 40))]
    sites += [("chrA", 100, 100), ("chrA", 100, 100), ("chrA", 90, 130), ("chrB", 1, 1), ("chrA", 1, 3), ("chrA", 4990, 5000), ("chrA", 2000, 2700)]
    order = rng.permutation(len(sites)); sites = [sites[i] for i in order]
    sl = _sites_file(d, "sites.txt", sites)
    for extra in ([], ["-p"], ["-i", "-q", "10", "-b", "8"], ["-p", "-i"]):
        base = [cli, "-w", "0", "-f", "syn.fa", "-l", sl] + extra
        a = subprocess.run(base + ["--brc-plan", "0", "syn.bam"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        b = subprocess.run(base + ["--brc-plan", "64", "syn.bam"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        c = subprocess.run(base + ["syn.bam"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert a.returncode == 0 and b.returncode == 0 and c.returncode == 0, (a.stderr, b.stderr)
        assert a.stdout.count(b"\n") > 400
        assert a.stdout == b.stdout == c.stdout, extra
    return a.stdout


def test_site_list_planner_equals_line_by_line_cpu(synthetic_bam):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    _planner_check(SIM_CLI, synthetic_bam)


def test_site_list_planner_sub_batches_cpu(synthetic_bam, monkeypatch):
    """A batch whose windows (with the overhang of their reads) outgrow the chunk size is cut into several engine passes:
    BRC_PLAN_VMAX forces a cut every few windows; the text must not change."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    want = _planner_check(SIM_CLI, synthetic_bam)
    monkeypatch.setenv("BRC_PLAN_VMAX", "700")
    assert _planner_check(SIM_CLI, synthetic_bam) == want


def _multi_engine_check(cli, d):
    """--brc-gpus N: region pieces and site-list batches dealt out to N engines must print, in file order, exactly what one
    engine prints — including a deletion pending across two abutting command-line regions (the first piece of a region
    follows the previous region onto its engine) and pieces cut inside deletions."""
    rng = np.random.default_rng(11)
    sites = [("chrA", int(p), int(p)) for p in rng.integers(1, 5000, 200)] + [("chrB", 10, 2500), ("chrA", 300, 2900), ("chrB", 5, 5)]
    sl = _sites_file(d, "sites_multi.txt", sites)
    runs = [["-f", "syn.fa", "syn.bam", "chrA:1-2000", "chrA:2001-5000", "chrB", "chrA:100-100"],
            ["-p", "-i", "-f", "syn.fa", "syn.bam", "chrA", "chrB:1-1500", "chrB:1501-3000"],
            ["-q", "10", "-b", "5", "-f", "syn.fa", "-l", sl, "--brc-plan", "16", "syn.bam"],
            ["-p", "-f", "syn.fa", "-l", sl, "syn.bam"]]
    for args in runs:
        env1 = dict(os.environ); env1.pop("BRC_DEVICES", None)
        one = subprocess.run([cli, "-w", "0", "--brc-chunk", "333", "--brc-streams", "1"] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env1)
        assert one.returncode == 0 and one.stdout.count(b"\n") > 1000, one.stderr
        # engines = GPUs x streams
        for extra in (["--brc-gpus", "2"], ["--brc-gpus", "3"], ["--brc-streams", "3"], ["--brc-gpus", "2", "--brc-streams", "2"]):
            many = subprocess.run([cli, "-w", "0", "--brc-chunk", "333"] + extra + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert many.returncode == 0, many.stderr
            assert many.stdout == one.stdout, (args, extra)
        whole = subprocess.run([cli, "-w", "0"] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env1)
        assert whole.stdout == one.stdout, args                  # and the tiling itself changes nothing
        # the three text routes: production threshold (this indel-rich data goes to the host's compact-plane formatter),
        # device-side text forced (the suite's default), host formatter forced
        for k, v in (("BRC_DEVICE_TEXT_MAX_SHARE", "0.06"), ("BRC_DEVICE_TEXT_MAX_SHARE", "100"), ("BRC_DEVICE_TEXT", "0")):
            alt = subprocess.run([cli, "-w", "0", "--brc-chunk", "333"] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(env1, **{k: v}))
            assert alt.returncode == 0 and alt.stdout == one.stdout, (args, k, v)
    bad = subprocess.run([cli, "--brc-gpus", "2", "-f", "syn.fa", "syn.bam", "chrA:1-50", "nochr:1-2", "chrB:1-5"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert bad.returncode == 1 and b"Invalid region nochr:1-2" in bad.stderr and bad.stdout.startswith(b"chrA\t") and b"chrB" not in bad.stdout


def test_cli_multi_engine_equals_single_cpu(synthetic_bam):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    _multi_engine_check(SIM_CLI, synthetic_bam)


@pytest.mark.gpu
def test_cli_multi_engine_equals_single_gpu(synthetic_bam, monkeypatch):
    """The GPU box has one device: BRC_DEVICES=0,0,0 still creates one engine, stream and worker thread per entry."""
    monkeypatch.setenv("BRC_DEVICES", "0,0,0")
    _multi_engine_check(HIP_CLI, synthetic_bam)


@pytest.mark.gpu
def test_site_list_planner_equals_line_by_line_gpu(synthetic_bam):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    got = _planner_check(HIP_CLI, synthetic_bam)
    assert got == _planner_check(SIM_CLI, synthetic_bam)      # and the GPU binary prints what the CPU simulator prints


def _records_to_arrays(recs, rg2lib):
    """brc_read_batch arrays from the record dicts of tools/bamio.read_bam"""
    a = dict(pos=np.array([r["pos"] for r in recs], np.int32), flag=np.array([r["flag"] for r in recs], np.uint16),
             mapq=np.array([r["mapq"] for r in recs], np.uint8), l_qseq=np.array([r["l_seq"] for r in recs], np.int32),
             lib=np.array([rg2lib.get(r["aux"].get("RG", (None, None))[1], -1) for r in recs], np.int16),
             n_cigar=np.array([len(r["cigar"]) for r in recs], np.uint32),
             nm=np.array([r["aux"]["NM"][1] if "NM" in r["aux"] else 0 for r in recs], np.int32),
             sm=np.array([r["aux"]["SM"][1] if "SM" in r["aux"] else 0 for r in recs], np.int32),
             tags=np.array([(1 if "NM" in r["aux"] else 0) | (2 if "SM" in r["aux"] else 0) for r in recs], np.uint8))
    for arena, key, dt in (("cigar", "cigar", np.uint32), ("seq4", "seq4", np.uint8), ("qual", "qual", np.uint8)):
        lens = np.array([len(r[key]) for r in recs], np.int64)
        a[arena] = np.concatenate([r[key] for r in recs]).astype(dt) if recs else np.zeros(0, dt)
        a[{"cigar": "cigar_off", "seq4": "seq_off", "qual": "qual_off"}[arena]] = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64) if recs else np.zeros(0, np.uint64)
    return a


def test_cli_bam_reader_region_equals_oracle_cpu(synthetic_bam, oracle_lib):
    """The host BGZF/BAM/BAI reader + CLI on a synthetic multi-block, two-contig BAM against the oracle fed with the
    original arrays (checks index queries, aux parsing, RG->LB mapping with a library-less read group)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bamio
    from bam_readcount_amd import capi
    import parity
    d = synthetic_bam
    text, refs, recs = bamio.read_bam(str(d / "syn.bam"))
    got = subprocess.run([SIM_CLI, "-p", "-f", "syn.fa", "syn.bam", "chrA:1000-1800", "chrB", "chrA:1-40"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert got.returncode == 0, got.stderr
    assert b"Expect library: libA in BAM" in got.stderr and b"Expect library: libB in BAM" in got.stderr
    assert got.stdout.count(b"\n") > 3000 and got.stdout.startswith(b"chrA\t1000\t")
    assert b"\tlibA\t{" in got.stdout and b"chrB\t" in got.stdout
    # ... and byte for byte what the oracle prints for the records the test-side BAM decoder reads from the same file
    # (command-line regions: the deletion queue is not cleared between them; a bare contig name runs to INT_MAX)
    names = ["libA", "libB"]
    want = b""
    eng = capi.Engine(oracle_lib, per_lib=True, lib_names=names)
    for tid, chrom, beg0, end in ((0, "chrA", 999, 1800), (1, "chrB", 0, 3000 + 1000), (0, "chrA", 0, 40)):
        arrs = _records_to_arrays([r for r in recs if r["tid"] == tid], {"rgA1": 0, "rgB1": 1})
        t, _ = capi.run_regions(eng, arrs, [(beg0, end)], tid, chrom, np.frombuffer(open(d / "syn.fa", "rb").read().split(b">")[tid + 1].split(b"\n", 1)[1].replace(b"\n", b""), np.uint8), clear_queue=False)
        want += t
    eng.close()
    assert got.stdout == want


def _cram_check(cli, oracle_lib, twolib):
    """n4: the reference's CRAM smoke run (test-data/cram_site_test.sh:1 — it has no expected file) through the minimal
    CRAM 3.0 reader: the CLI's text must equal the oracle fed with the same 4 reads re-created from rand1k.fa."""
    import parity
    names = [str(s) for s in twolib["lib_names"]]
    for extra, kw in (([], dict()), (["-p", "-i"], dict(per_lib=True, insertion_centric=True, lib_names=names)),
                      (["-p"], dict(per_lib=True, lib_names=names))):
        want, _ = parity.run_engine(oracle_lib, twolib, [(49, 60)], tid=0, chrom="rand1k", ref=twolib["ref"], ref_len_check=True, **kw)
        p = subprocess.run([cli, "-w", "0"] + extra + ["-l", "twolib_site_list.txt", "-f", "rand1k.fa", "twolib.sorted.cram"],
                           cwd=GOLDEN, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr
        assert p.stdout == want and p.stdout.count(b"\n") == 11, extra
        assert b"Expect library: reads1_lb in BAM" in p.stderr and b"Expect library: reads2_lb in BAM" in p.stderr
    # region mode over all four reads (both libraries), and a query that starts inside the second read
    want, _ = parity.run_engine(oracle_lib, twolib, [(0, 1000), (100, 130)], tid=0, chrom="rand1k", ref=twolib["ref"], per_lib=True, lib_names=names, clear_queue=False)
    p = subprocess.run([cli, "-w", "0", "-p", "-f", "rand1k.fa", "twolib.sorted.cram", "rand1k", "rand1k:101-130"], cwd=GOLDEN, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and p.stdout == want and p.stdout.count(b"\n") == 240 + 30
    # the records were stored without NM: the decoder supplies it like htslib's does, so nothing warns about the tag
    p = subprocess.run([cli, "-l", "twolib_site_list.txt", "-f", "rand1k.fa", "twolib.sorted.cram"], cwd=GOLDEN, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and b"WARNING" not in p.stderr, p.stderr
    # mapped CRAM records cannot be rebuilt without the reference
    p = subprocess.run([cli, "twolib.sorted.cram", "rand1k:1-10"], cwd=GOLDEN, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"reference FASTA" in p.stderr


def test_cli_cram_input_equals_oracle_cpu(oracle_lib, twolib):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    _cram_check(SIM_CLI, oracle_lib, twolib)


@pytest.mark.gpu
def test_cli_cram_input_equals_oracle_gpu(oracle_lib, twolib):
    _cram_check(HIP_CLI, oracle_lib, twolib)


def test_cli_cram_reader_equals_bam_reader_cpu(synthetic_bam):
    """The same synthetic reads written as BAM and (tools/cramio.py) as a multi-container reference-based CRAM with every
    feature code, a multi-reference slice, Huffman/BETA core-stream series: the CLI must print identical text for both."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    d = synthetic_bam
    for extra in ([], ["-p", "-i"], ["-q", "15", "-b", "10"]):
        regs = ["chrA:1-5000", "chrB", "chrA:2400-2450", "chrB:1-20"]
        a = subprocess.run([SIM_CLI, "-w", "0", "-f", "syn.fa"] + extra + ["syn_m.bam"] + regs, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        b = subprocess.run([SIM_CLI, "-w", "0", "-f", "syn.fa"] + extra + ["syn.cram"] + regs, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert a.returncode == 0 and b.returncode == 0, (a.stderr, b.stderr)
        assert a.stdout.count(b"\n") > 7000
        assert a.stdout == b.stdout, extra
        assert a.stderr == b.stderr
    sl = _sites_file(d, "csites.txt", [("chrA", 1200, 1260), ("chrB", 5, 9), ("chrA", 1200, 1200), ("chrB", 2990, 3005)])
    a = subprocess.run([SIM_CLI, "-w", "0", "-f", "syn.fa", "-p", "-l", sl, "--brc-plan", "0", "syn_m.bam"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    b = subprocess.run([SIM_CLI, "-w", "0", "-f", "syn.fa", "-p", "-l", sl, "syn.cram"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert a.returncode == 0 and b.returncode == 0 and a.stdout == b.stdout and a.stdout.count(b"\n") > 60


def test_cli_builds_a_missing_fasta_index(synthetic_bam, tmp_path):
    """htslib's fai_load / samtools' samfaipath build <fasta>.fai when it is missing (bamreadcount.cpp:501-506): so does the
    drop-in — the same index (wrapped lines, a short last line, a header with a description), the same text."""
    import shutil
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    d = synthetic_bam
    for f in ("syn.bam", "syn.bam.bai", "syn.fa"):
        shutil.copy(d / f, tmp_path / f)
    fa = open(tmp_path / "syn.fa", "rb").read().replace(b">chrA\n", b">chrA some description\n")
    open(tmp_path / "syn.fa", "wb").write(fa)
    args = ["-w", "0", "-p", "-f", "syn.fa", "syn.bam", "chrA:4900-5000", "chrB:1-80"]
    want = subprocess.run([SIM_CLI] + ["-w", "0", "-p", "-f", str(d / "syn.fa"), "syn.bam", "chrA:4900-5000", "chrB:1-80"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    got = subprocess.run([SIM_CLI] + args, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert got.returncode == 0 and want.returncode == 0 and got.stdout == want.stdout and got.stdout.count(b"\n") > 150
    built = open(tmp_path / "syn.fa.fai").read().split("\n"); orig = open(d / "syn.fa.fai").read().split("\n")
    shift = len(b" some description")
    assert built[0] == "chrA\t5000\t%d\t60\t61" % (6 + shift) and [l.split("\t")[:2] + l.split("\t")[3:] for l in built] == [l.split("\t")[:2] + l.split("\t")[3:] for l in orig]
    assert int(built[1].split("\t")[2]) == int(orig[1].split("\t")[2]) + shift
    # uneven lines inside a sequence are refused, as by fai_build
    open(tmp_path / "bad.fa", "wb").write(b">x\nACGT\nAC\nACGT\n")
    bad = subprocess.run([SIM_CLI, "-f", "bad.fa", "syn.bam", "chrA:1-2"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert bad.returncode != 0 and not os.path.exists(tmp_path / "bad.fa.fai")


def test_cli_cram_index_cpu(synthetic_bam, tmp_path):
    """A .crai next to the CRAM replaces the walk over the container headers: same text with it; an index that leaves out
    the containers of one contig makes that contig's reads invisible (so it is the index that is used); an index that
    points into the middle of a container is refused (container header CRC32)."""
    import gzip, shutil
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    d = synthetic_bam
    for f in ("syn.cram", "syn.fa", "syn.fa.fai"):
        shutil.copy(d / f, tmp_path / f)
    regs = ["chrA:1-5000", "chrB", "chrA:2400-2450"]
    run = lambda: subprocess.run([SIM_CLI, "-w", "0", "-p", "-f", "syn.fa", "syn.cram"] + regs, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    want = run()
    assert want.returncode == 0 and want.stdout.count(b"\n") > 7000
    # the index of this very file: re-create the CRAM with the writer's .crai (same bytes, plus the index)
    # walk the container headers here to write the index by hand: (ref, start, span, offset) of every data container
    import struct
    raw = open(tmp_path / "syn.cram", "rb").read()
    def itf8(b, o):
        v = b[o]
        if v < 0x80: return v, o + 1
        if v < 0xc0: return ((v & 0x3f) << 8) | b[o + 1], o + 2
        if v < 0xe0: return ((v & 0x1f) << 16) | (b[o + 1] << 8) | b[o + 2], o + 3
        if v < 0xf0: return ((v & 0x0f) << 24) | (b[o + 1] << 16) | (b[o + 2] << 8) | b[o + 3], o + 4
        x = ((v & 0x0f) << 28) | (b[o + 1] << 20) | (b[o + 2] << 12) | (b[o + 3] << 4) | (b[o + 4] & 0x0f)
        return x - (1 << 32) if x & (1 << 31) else x, o + 5
    def ltf8_skip(b, o):
        v = b[o]; n = 0
        while n < 8 and (v << n) & 0x80: n += 1
        return o + 1 + n
    o = 26; first = True; conts = []
    while o < len(raw):
        at = o
        ln = struct.unpack_from("<i", raw, o)[0]; o += 4
        ref, o = itf8(raw, o); st, o = itf8(raw, o); sp, o = itf8(raw, o); nrec, o = itf8(raw, o)
        o = ltf8_skip(raw, o); o = ltf8_skip(raw, o)
        nb, o = itf8(raw, o); nl, o = itf8(raw, o)
        for _ in range(nl): _, o = itf8(raw, o)
        o += 4 + ln
        if not first and nrec > 0: conts.append((ref, st, sp, at))
        first = False
    assert len(conts) > 4
    def write_index(entries):
        txt = ""
        for ref, st, sp, at in entries:
            if ref == -2: txt += "0\t1\t5000\t%d\t0\t0\n1\t1\t3000\t%d\t0\t0\n" % (at, at)
            else: txt += "%d\t%d\t%d\t%d\t0\t0\n" % (ref, st, sp, at)
        gzip.open(tmp_path / "syn.cram.crai", "wb").write(txt.encode())
    write_index(conts)
    got = run()
    assert got.returncode == 0 and got.stdout == want.stdout
    write_index([c for c in conts if c[0] == 0])                                  # no chrB (and no multi-reference) container
    got = run()
    assert got.returncode == 0 and got.stdout != want.stdout and b"chrB\t2000\t" not in got.stdout and b"chrA\t2000\t" in got.stdout
    write_index([(r, s0, sp, at + 7) for r, s0, sp, at in conts])
    got = run()
    assert got.returncode != 0 and b"crai" in got.stderr


def test_cli_cram_with_damaged_contents_is_refused(synthetic_bam, tmp_path):
    """Blocks whose CRC32 is right but whose contents are not: read features that point outside their read, a truncated tag
    dictionary, code lengths no Huffman code has — an error message and a non-zero exit code, no output for the damaged part."""
    import sys, shutil
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import cramio
    import synth
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    d = synthetic_bam
    for f in ("syn.fa", "syn.fa.fai"):
        shutil.copy(d / f, tmp_path / f)
    ref = np.frombuffer(open(d / "syn.fa", "rb").read().split(b">")[1].split(b"\n", 1)[1].replace(b"\n", b""), np.uint8)
    arrs = synth.make_batch(31, ref, 300, style="mixed", n_libs=2)
    rgs = [["rgA1", "rgB1"][int(l)] if l >= 0 else None for l in arrs["lib"]]
    orig = cramio.block
    def damaged(which):
        def blk(method, ctype, cid, data):
            data = bytearray(data)
            if which == "features" and ctype == 4 and cid == cramio.IDS["FP"] and len(data) > 8: data[5] = 0x7f; data[6] = 0x7f
            if which == "dictionary" and ctype == 1 and len(data) > 12: data = data[:len(data) // 2]
            if which == "lengths" and ctype == 4 and cid == cramio.IDS["RL"] and len(data) > 4: data[0:5] = b"\xff\xff\xff\xff\x0f"
            return orig(method, ctype, cid, bytes(data))
        return blk
    try:
        for which in ("features", "dictionary", "lengths"):
            cramio.block = damaged(which)
            cramio.write_cram(str(tmp_path / "bad.cram"), [("chrA", 5000)], arrs, np.zeros(len(arrs["pos"]), int), [ref], rg_of_read=rgs,
                              rg_lines=["@RG\tID:rgA1\tLB:libA\tSM:s", "@RG\tID:rgB1\tLB:libB\tSM:s"], per_container=400)
            p = subprocess.run([SIM_CLI, "-w", "0", "-f", "syn.fa", "bad.cram", "chrA"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert p.returncode == 1 and b"CRAM" in p.stderr and p.stdout == b"", (which, p.returncode, p.stderr[-200:])
    finally:
        cramio.block = orig


def _cram_embedded_and_reference_less(cli, d):
    regs = ["chrA:1-5000", "chrB", "chrA:2400-2450"]
    # (the container that holds the end of chrA and the start of chrB is a multi-reference slice: those cannot embed a
    # reference and are rebuilt from the FASTA like any other — the embedded-reference comparison stays clear of it)
    eregs = ["chrA:1-4200", "chrB:1300-3000", "chrA:2400-2450"]
    for extra in ([], ["-p", "-i"]):
        # embedded reference: the reads come back as written even under another FASTA, NM generated from the embedded bases
        a = subprocess.run([cli, "-w", "3", "-f", "syn_alt.fa"] + extra + ["syn_m.bam"] + eregs, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        b = subprocess.run([cli, "-w", "3", "-f", "syn_alt.fa"] + extra + ["syn_emb.cram"] + eregs, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert a.returncode == 0 and b.returncode == 0, (a.stderr, b.stderr)
        assert a.stdout.count(b"\n") > 5000 and a.stdout == b.stdout and a.stderr == b.stderr, extra
        # ... which matters: rebuilt from the other FASTA the reads would differ
        c = subprocess.run([cli, "-w", "3", "-f", "syn_alt.fa"] + extra + ["syn.cram"] + eregs, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert c.returncode == 0 and c.stdout != a.stdout
        # no reference required: same reads, and no NM is made up for the records stored without one (htslib: no s->ref)
        a = subprocess.run([cli, "-w", "3", "-f", "syn_alt.fa"] + extra + ["syn_m_nonm.bam"] + regs, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        b = subprocess.run([cli, "-w", "3", "-f", "syn_alt.fa"] + extra + ["syn_noref.cram"] + regs, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert a.returncode == 0 and b.returncode == 0, (a.stderr, b.stderr)
        assert a.stdout == b.stdout and a.stderr == b.stderr and b"NM tag" in b.stderr, extra


def test_cli_cram_embedded_reference_and_reference_less_cpu(synthetic_bam):
    """CRAM slices that carry their own reference (slice header: embedded reference block id) and files written without one
    (preservation map RR = 0) decode without consulting the FASTA."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    _cram_embedded_and_reference_less(SIM_CLI, synthetic_bam)


def test_cli_cram_rans_bzip2_lzma_blocks_equal_bam_reader_cpu(synthetic_bam):
    """n4 "then rANS": the CRAM twin written with rANS 4x8 (order 0 and 1), bzip2, lzma, gzip and raw blocks, a compressed
    core block and GAMMA / SUBEXP series prints what the BAM prints; a flipped payload byte is caught by the block CRC."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    d = synthetic_bam
    raw = open(d / "syn_rans.cram", "rb").read()
    assert raw.count(b"BZh") > 3 and raw.count(b"\xfd7zXZ\x00") > 3          # bzip2 / xz streams are really in the file
    for extra in ([], ["-p", "-i"], ["-q", "15", "-b", "10"]):
        regs = ["chrA:1-5000", "chrB", "chrA:2400-2450"]
        a = subprocess.run([SIM_CLI, "-w", "0", "-f", "syn.fa"] + extra + ["syn_m.bam"] + regs, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        b = subprocess.run([SIM_CLI, "-w", "0", "-f", "syn.fa"] + extra + ["syn_rans.cram"] + regs, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert a.returncode == 0 and b.returncode == 0, (a.stderr, b.stderr)
        assert a.stdout.count(b"\n") > 7000 and a.stdout == b.stdout and a.stderr == b.stderr, extra
    bad = bytearray(raw); bad[len(bad) // 2] ^= 0x10
    open(d / "syn_bad.cram", "wb").write(bytes(bad))
    c = subprocess.run([SIM_CLI, "-w", "0", "-f", "syn.fa", "syn_bad.cram", "chrA", "chrB"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert c.returncode != 0 and b"CRC32" in c.stderr, c.stderr[-300:]


def test_cli_long_cigar_in_cg_tag_and_truncated_bam_cpu(tmp_path):
    """A record whose CIGAR sits in the CG:B,I tag behind the <l_seq>S<span>N placeholder (SAMv1 4.2.2) is read with its real
    operators; a BAM cut in the middle of a block is an error (exit code 1, message), not a silently shorter output."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bamio
    import synth
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    rng = np.random.default_rng(3)
    ref = synth.make_ref(rng, 4000)
    arrs = synth.make_batch(77, ref, 900, style="indel")
    tids = np.zeros(len(arrs["pos"]), int)
    multi = [i for i in range(len(arrs["pos"])) if int(arrs["n_cigar"][i]) >= 3][:40]
    assert len(multi) == 40
    bamio.write_bam(str(tmp_path / "plain.bam"), [("chrA", 4000)], arrs, tids, block_bytes=3000)
    bamio.write_bam(str(tmp_path / "cg.bam"), [("chrA", 4000)], arrs, tids, block_bytes=3000, long_cigar=set(multi))
    _write_fasta(tmp_path / "r.fa", [("chrA", ref)])
    a = subprocess.run([SIM_CLI, "-w", "0", "-f", "r.fa", "plain.bam", "chrA"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    b = subprocess.run([SIM_CLI, "-w", "0", "-f", "r.fa", "cg.bam", "chrA"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert a.returncode == 0 and b.returncode == 0 and a.stdout.count(b"\n") > 3500 and a.stdout == b.stdout
    raw = open(tmp_path / "plain.bam", "rb").read()
    open(tmp_path / "cut.bam", "wb").write(raw[:len(raw) * 2 // 3])
    os.link(tmp_path / "plain.bam.bai", tmp_path / "cut.bam.bai")
    c = subprocess.run([SIM_CLI, "-w", "0", "-f", "r.fa", "cut.bam", "chrA"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert c.returncode == 1 and b"read error" in c.stderr, (c.returncode, c.stderr[-200:])


def test_cli_csi_index_equals_bai_cpu(tmp_path):
    """The same BAM with a .bai and — in another directory, so that only one index can be found — with .csi indexes of two
    geometries (the default 14 / 5 and a finer, deeper one): identical output for whole contigs, narrow regions and a site list."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bamio
    import synth
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    rng = np.random.default_rng(5)
    refs = [synth.make_ref(rng, 40000), synth.make_ref(rng, 9000)]
    parts = [synth.make_batch(71, refs[0], 2500, style="indel"), synth.make_batch(72, refs[1], 700, style="mixed")]
    arrs = {}
    for k in ("pos", "flag", "mapq", "lib", "l_qseq", "n_cigar", "nm", "sm", "tags"):
        arrs[k] = np.concatenate([p[k] for p in parts])
    for arena, off in (("cigar", "cigar_off"), ("seq4", "seq_off"), ("qual", "qual_off")):
        arrs[arena] = np.concatenate([p[arena] for p in parts])
        arrs[off] = np.concatenate([parts[0][off], parts[1][off] + np.uint64(parts[0][arena].size)])
    tids = np.concatenate([np.zeros(len(parts[0]["pos"]), int), np.ones(len(parts[1]["pos"]), int)])
    outs = []
    site_lines = [("chrA", int(p), int(p) + 3) for p in rng.integers(1, 39000, 60)] + [("chrB", 17, 17), ("chrB", 8000, 9000)]
    for name, csi in (("bai", None), ("csi145", (14, 5)), ("csi106", (10, 6))):
        d = tmp_path / name; d.mkdir()
        bamio.write_bam(str(d / "x.bam"), [("chrA", 40000), ("chrB", 9000)], arrs, tids, block_bytes=4000, csi=csi)
        assert os.path.exists(d / ("x.bam.csi" if csi else "x.bam.bai")) and not os.path.exists(d / ("x.bam.bai" if csi else "x.bam.csi"))
        _write_fasta(d / "r.fa", [("chrA", refs[0]), ("chrB", refs[1])])
        sl = _sites_file(d, "s.txt", site_lines)
        got = []
        for args in (["x.bam", "chrA", "chrB"], ["x.bam", "chrA:16380-16390", "chrA:1024-1024", "chrB:8999-9000"], ["-l", sl, "x.bam"], ["-l", sl, "--brc-plan", "0", "x.bam"]):
            p = subprocess.run([SIM_CLI, "-w", "0", "-f", "r.fa"] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert p.returncode == 0, (name, args, p.stderr[-300:])
            got.append(p.stdout)
        outs.append(got)
    assert outs[0][0].count(b"\n") > 40000
    assert outs[1] == outs[0] and outs[2] == outs[0]


def test_rans_encoder_round_trip_sizes():
    """tools/cramio.py's rANS encoder against an independent pure-Python decoder written from the same format description
    (sizes 0..9 and larger, skewed and flat distributions, both orders) — guards the test-side writer itself."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import cramio
    rng = np.random.default_rng(11)
    for order in (0, 1):
        for n in list(range(0, 10)) + [63, 64, 257, 4099]:
            for kind in ("skew", "flat", "one"):
                if kind == "skew": data = bytes(rng.choice([33, 34, 35, 40, 41, 255, 0, 2], n, p=[.5, .2, .1, .05, .05, .04, .03, .03]).astype(np.uint8))
                elif kind == "flat": data = bytes(rng.integers(0, 256, n, dtype=np.uint8))
                else: data = bytes([7]) * n
                comp = cramio.rans_encode(data, order)
                assert cramio.rans_decode(comp) == data, (order, n, kind)


def test_cli_striped_cram_fetch_equals_single_reader_cpu(synthetic_bam):
    """CRAM pieces are decoded in stripes too (round 5): one CramReader per stripe, each decoding the slices that overlap its stripe and
    keeping the records that start in it, the contig's bases shared — forced onto small regions with odd thread counts and chunk sizes,
    over the multi-container CRAM with a multi-reference slice: the text of the one-reader route (and of the BAM)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    d = synthetic_bam
    for extra in ([], ["-p", "-i"]):
        base = [SIM_CLI, "-w", "0", "-f", "syn.fa"] + extra + ["syn.cram", "chrA", "chrB:100-2900", "chrA:4990-5000"]
        want = subprocess.run(base, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, BRC_FETCH_STRIPE_MIN="1000000000"))
        bam = subprocess.run([SIM_CLI, "-w", "0", "-f", "syn.fa"] + extra + ["syn_m.bam", "chrA", "chrB:100-2900", "chrA:4990-5000"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert want.returncode == 0 and want.stdout.count(b"\n") > 7000 and want.stdout == bam.stdout
        for threads, chunk in ((3, "700"), (7, "1000000"), (16, "64"), (2, "5000")):
            env = dict(os.environ, BRC_FETCH_STRIPE_MIN="1", BRC_FETCH_THREADS=str(threads))
            got = subprocess.run(base[:1] + ["--brc-chunk", chunk] + base[1:], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            assert got.returncode == 0, got.stderr
            assert got.stdout == want.stdout and got.stderr == want.stderr, (extra, threads, chunk)
    # the other CRAM files of the fixture — every block method and integer codec, embedded reference slices found through a .crai, a
    # reference-less file — in stripes against one reader
    for f in ("syn_rans.cram", "syn_emb.cram", "syn_noref.cram"):
        base = [SIM_CLI, "-w", "0", "-p", "-f", "syn.fa", f, "chrA", "chrB:100-2900"]
        want = subprocess.run(base, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, BRC_FETCH_STRIPE_MIN="1000000000"))
        got = subprocess.run(base[:1] + ["--brc-chunk", "900"] + base[1:], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, BRC_FETCH_STRIPE_MIN="1", BRC_FETCH_THREADS="5"))
        assert want.returncode == 0 and got.returncode == 0 and want.stdout.count(b"\n") > 6000, (f, want.stderr, got.stderr)
        assert got.stdout == want.stdout and got.stderr == want.stderr, f


def test_cli_striped_parallel_fetch_equals_single_handle_cpu(synthetic_bam, workdir):
    """Long chunks are decoded by several BAM handles in stripes of read start positions while the previous chunk is on
    the engine; forced here onto small regions (BRC_FETCH_STRIPE_MIN) with odd thread counts and chunk sizes."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    d = synthetic_bam
    base = [SIM_CLI, "-w", "0", "-p", "-f", "syn.fa", "syn.bam", "chrA", "chrB:100-2900", "chrA:4990-5000"]
    want = subprocess.run(base, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert want.returncode == 0 and want.stdout.count(b"\n") > 7000
    for threads, chunk in ((3, "700"), (7, "1000000"), (16, "64"), (2, "5000")):
        env = dict(os.environ, BRC_FETCH_STRIPE_MIN="1", BRC_FETCH_THREADS=str(threads))
        got = subprocess.run(base[:1] + ["--brc-chunk", chunk] + base[1:], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert got.returncode == 0, got.stderr
        assert got.stdout == want.stdout, (threads, chunk)
    env = dict(os.environ, BRC_FETCH_STRIPE_MIN="1", BRC_FETCH_THREADS="5")
    exp = open(os.path.join(GOLDEN, "expected_insertion_centric_per_lib"), "rb").read()
    rc, out, _ = run_cli(SIM_CLI, workdir, "test.bam", ["-i", "-p"], "list") if False else (0, None, None)
    p = subprocess.run([SIM_CLI, "-w", "1", "-i", "-p", "-f", "ref.fa", "-l", "site_list", "--brc-plan", "0", "test.bam"], cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert p.returncode == 0 and p.stdout == exp


def _ranks_check(cli, d, devices=None):
    """--brc-ranks N (one process per GPU): N processes, each with a contiguous event-weighted slice of the work list in file order, must
    print — stdout, stderr (the -w cap runs over the whole list) and exit code — exactly what one process prints: regions cut inside
    deletions, site lists with duplicate / overlapping / unsorted lines and lines on unknown contigs, per-library and insertion-centric
    modes; a bad region ends the run where one process would have stopped."""
    rng = np.random.default_rng(12)
    sites = [("chrA", int(p), int(p)) for p in rng.integers(1, 5000, 300)] + [("chrB", 10, 2500), ("chrA", 300, 2900), ("chrB", 5, 5), ("chrA", 100, 100), ("chrA", 100, 100), ("chrA", 90, 130)]
    order = rng.permutation(len(sites)); sites = [sites[i] for i in order]
    sites.insert(40, ("nochr", 5, 9)); sites.insert(200, ("chrZ", 1, 1))
    sl = _sites_file(d, "sites_ranks.txt", sites)
    runs = [["-w", "0", "-f", "syn.fa", "syn.bam", "chrA"],
            ["-w", "7", "-p", "-i", "-f", "syn.fa", "syn_m_nonm.bam", "chrA:200-4800"],
            ["-w", "0", "-q", "10", "-b", "5", "-f", "syn.fa", "-l", sl, "--brc-plan", "16", "syn.bam"],
            ["-w", "5", "-p", "-f", "syn.fa", "-l", sl, "syn_m_nonm.bam"],
            ["-w", "-1", "-f", "syn.fa", "-l", sl, "--brc-plan", "0", "syn_m_nonm.bam"],
            ["-w", "0", "-d", "30", "-f", "syn.fa", "-l", sl, "syn.bam"]]
    env1 = dict(os.environ); env1.pop("BRC_DEVICES", None); env1.pop("BRC_RANKS", None)
    for args in runs:
        one = subprocess.run([cli, "--brc-chunk", "333"] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env1)
        assert one.returncode == 0 and one.stdout.count(b"\n") > 500, one.stderr
        for n in (2, 3, 4):
            env = dict(env1, BRC_RANK_CUT="257")
            if devices is not None:
                env["BRC_DEVICES"] = ",".join([str(devices)] * n)
            many = subprocess.run([cli, "--brc-chunk", "333", "--brc-ranks", str(n)] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            assert many.returncode == 0, many.stderr
            assert many.stdout == one.stdout, (args, n)
            assert many.stderr == one.stderr, (args, n, many.stderr[-2000:], one.stderr[-2000:])
    # several command-line regions stay one process (a pending deletion can hold back every region behind it); BRC_RANKS asks like the option
    args = ["-w", "0", "-f", "syn.fa", "syn.bam", "chrA:1-2000", "chrA:2001-5000", "chrB"]
    one = subprocess.run([cli] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env1)
    env = dict(env1, BRC_RANKS="3")
    if devices is not None:
        env["BRC_DEVICES"] = ",".join([str(devices)] * 3)
    many = subprocess.run([cli] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert many.returncode == 0 and many.stdout == one.stdout and many.stderr == one.stderr
    # /dev/null as stdout: the ranks write there themselves; the stage account names every rank and its share of the estimated work
    with open(os.devnull, "wb") as dn:
        t = subprocess.run([cli, "--brc-ranks", "2", "-w", "0", "-f", "syn.fa", "syn.bam", "chrA"], cwd=d, stdout=dn, stderr=subprocess.PIPE, env=dict(env, BRC_CLI_TIMING="1", BRC_RANK_CUT="257"))
    assert t.returncode == 0 and b"rank 0 of 2" in t.stderr and b"rank 1 of 2" in t.stderr and b"ranks: 2 processes" in t.stderr, t.stderr
    # errors: an unknown contig in the region ends the run with the reference's message, once; a BAM that cannot be opened likewise
    bad = subprocess.run([cli, "--brc-ranks", "2", "-f", "syn.fa", "syn.bam", "nochr:1-2"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert bad.returncode == 1 and bad.stderr.count(b"Invalid region nochr:1-2") == 1 and bad.stdout == b""
    bad = subprocess.run([cli, "--brc-ranks", "3", "-f", "syn.fa", "missing.bam", "chrA"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    one = subprocess.run([cli, "-f", "syn.fa", "missing.bam", "chrA"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env1)
    assert bad.returncode == one.returncode == 1 and bad.stderr == one.stderr and bad.stdout == b""


def test_cli_ranks_equal_single_process_cpu(synthetic_bam):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    _ranks_check(SIM_CLI, synthetic_bam)


@pytest.mark.gpu
def test_cli_ranks_equal_single_process_gpu(synthetic_bam):
    """world 2, 3 and 4 on ONE GPU (BRC_DEVICES=0,0,..): every rank a process of its own with its own HIP context"""
    _ranks_check(HIP_CLI, synthetic_bam, devices=0)
