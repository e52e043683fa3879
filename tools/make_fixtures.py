#!/usr/bin/env python3
"""Generate the committed parity fixtures under tests/golden/ from the reference's test data.

Run in the build container (needs /root/reference); the GPU box only sees the committed outputs.

  tests/golden/test_bam.npz       decoded reads of test-data/test.bam as brc_read_batch arrays, the
                                  recovered pseudo-reference slice, library names
  tests/golden/expected_*         the reference's four golden outputs (integration-test/bam-readcount_test.py:29-116)
  tests/golden/site_list          the reference's site list
  tests/golden/twolib.npz         the 4 reads of test-data/twolib.sorted.cram re-created from rand1k.fa
                                  (the CRAM holds 4 perfect-match 60M reads) -- derived, unpinned

test-data/ref.fa is a missing blob in the reference checkout (.MISSING_LARGE_BLOBS).  It is recovered as the
per-position consensus of the M-op bases of test.bam over the covered window, 'N' elsewhere, with
21:10405200 forced to 'T' (column 3 of expected_all_lib line 2).  The script asserts that every read's NM tag
equals (mismatches against the pseudo-reference + inserted bases), i.e. the recovery is consistent with the aligner.
"""
import os
import shutil
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from bamio import read_bam, rg_to_lib  # noqa: E402

REF = "/root/reference/test-data"
OUT = os.path.join(HERE, "..", "tests", "golden")
NT16 = "=ACMGRSVTWYHKDBN"


def batch_arrays(recs, rg2lib, lib_names):
    n = len(recs)
    a = dict(
        pos=np.array([r["pos"] for r in recs], np.int32),
        flag=np.array([r["flag"] for r in recs], np.uint16),
        mapq=np.array([r["mapq"] for r in recs], np.uint8),
        l_qseq=np.array([r["l_seq"] for r in recs], np.int32),
        n_cigar=np.array([len(r["cigar"]) for r in recs], np.uint32),
    )
    lib = []
    for r in recs:
        rg = r["aux"].get("RG", (None, None))[1]
        lb = rg2lib.get(rg)
        lib.append(lib_names.index(lb) if lb in lib_names else -1)
    a["lib"] = np.array(lib, np.int16)
    a["cigar"] = np.concatenate([r["cigar"] for r in recs]).astype(np.uint32)
    a["seq4"] = np.concatenate([r["seq4"] for r in recs]).astype(np.uint8)
    a["qual"] = np.concatenate([r["qual"] for r in recs]).astype(np.uint8)
    a["cigar_off"] = np.concatenate([[0], np.cumsum(a["n_cigar"])[:-1]]).astype(np.uint64)
    a["seq_off"] = np.concatenate([[0], np.cumsum([(r["l_seq"] + 1) // 2 for r in recs])[:-1]]).astype(np.uint64)
    a["qual_off"] = np.concatenate([[0], np.cumsum([r["l_seq"] for r in recs])[:-1]]).astype(np.uint64)
    nm = np.zeros(n, np.int32); sm = np.zeros(n, np.int32); tags = np.zeros(n, np.uint8)
    for i, r in enumerate(recs):
        if "NM" in r["aux"]:
            nm[i] = r["aux"]["NM"][1]; tags[i] |= 1
        if "SM" in r["aux"]:
            sm[i] = r["aux"]["SM"][1]; tags[i] |= 2
    a["nm"], a["sm"], a["tags"] = nm, sm, tags
    a["qname"] = np.array([r["qname"] for r in recs])
    return a


def bases(r):
    s = r["seq4"]
    return [(s[i >> 1] >> (4 if (i & 1) == 0 else 0)) & 15 for i in range(r["l_seq"])]


def recover_reference(recs):
    lo = min(r["pos"] for r in recs)
    hi = lo
    cols = {}
    for r in recs:
        b = bases(r); rp = r["pos"]; qp = 0
        for c in r["cigar"]:
            op, ln = int(c) & 15, int(c) >> 4
            if op in (0, 7, 8):
                for j in range(ln):
                    cols.setdefault(rp + j, [0] * 16)[b[qp + j]] += 1
                rp += ln; qp += ln
            elif op in (2, 3):
                rp += ln
            elif op in (1, 4):
                qp += ln
        hi = max(hi, rp)
    ref = np.full(hi - lo, ord("N"), np.uint8)
    for p, cnt in cols.items():
        best = max(range(16), key=lambda k: cnt[k])
        ref[p - lo] = ord(NT16[best])
    ref[10405199 - lo] = ord("T")  # pinned by expected_all_lib line 2, column 3
    return lo, ref


def check_nm(recs, lo, ref):
    for r in recs:
        b = bases(r); rp = r["pos"]; qp = 0; nm = 0
        for c in r["cigar"]:
            op, ln = int(c) & 15, int(c) >> 4
            if op == 0:
                for j in range(ln):
                    rc = chr(ref[rp + j - lo])
                    if rc != "N" and NT16[b[qp + j]] != rc:
                        nm += 1
                rp += ln; qp += ln
            elif op == 2:
                nm += ln; rp += ln
            elif op == 1:
                nm += ln; qp += ln
            elif op == 4:
                qp += ln
        assert nm == r["aux"]["NM"][1], (r["qname"], nm, r["aux"]["NM"][1])


def main():
    os.makedirs(OUT, exist_ok=True)
    text, refs, recs = read_bam(os.path.join(REF, "test.bam"))
    assert len(recs) == 403
    rg2lib = rg_to_lib(text)
    lib_names = sorted(set(v for v in rg2lib.values() if v))
    a = batch_arrays(recs, rg2lib, lib_names)
    lo, ref = recover_reference(recs)
    check_nm(recs, lo, ref)
    # test_bad_rg.bam: identical reads, @RG lines lack LB  (bam-readcount_test.py:73-86)
    t2, _, r2 = read_bam(os.path.join(REF, "test_bad_rg.bam"))
    assert [x["qname"] for x in r2] == [x["qname"] for x in recs]
    assert all(v is None for v in rg_to_lib(t2).values())
    fai = open(os.path.join(REF, "ref.fa.fai")).read().split()
    np.savez_compressed(os.path.join(OUT, "test_bam.npz"), ref_start=np.int64(lo), ref_slice=ref,
                        ref_len=np.int64(int(fai[1])), contig=np.array(fai[0]), tid=np.int32(recs[0]["tid"]),
                        lib_names=np.array(lib_names), **a)
    # the four goldens, the site lists, and the reference's BAM inputs themselves (data, needed by the CLI tests)
    for f in ("expected_all_lib", "expected_per_lib", "expected_insertion_centric_all_lib",
              "expected_insertion_centric_per_lib", "site_list", "twolib_site_list.txt",
              "test.bam", "test.bam.bai", "test_bad_rg.bam", "test_bad_rg.bam.bai",
              "twolib.sorted.cram", "twolib.sorted.cram.crai", "rand1k.fa", "rand1k.fa.fai"):
        shutil.copyfile(os.path.join(REF, f), os.path.join(OUT, f))

    # twolib.sorted.cram: 4 reads 60M, flag 0, MAPQ 60, starts 0/60/120/180, perfect match, QUAL 0xFF, no NM/SM stored:
    # htslib's decoder regenerates NM (= 0 here) for a mapped record stored without it (cram_decode.c cram_decode_seq)
    fa = "".join(l.strip() for l in open(os.path.join(REF, "rand1k.fa")) if not l.startswith(">"))
    assert len(fa) == 1000
    code = {c: i for i, c in enumerate(NT16)}
    trecs = []
    for i, (name, lb) in enumerate([("read1-1", "reads1_lb"), ("read1-2", "reads1_lb"), ("read2-1", "reads2_lb"), ("read2-2", "reads2_lb")]):
        s = fa[i * 60:(i + 1) * 60].upper()
        nib = [code[c] for c in s]
        seq4 = np.array([(nib[j] << 4) | nib[j + 1] for j in range(0, 60, 2)], np.uint8)
        trecs.append(dict(tid=0, pos=i * 60, mapq=60, flag=0, l_seq=60, qname=name, cigar=np.array([60 << 4], np.uint32),
                          seq4=seq4, qual=np.full(60, 255, np.uint8), aux={"RG": ("Z", lb), "NM": ("I", 0)}))
    tl = ["reads1_lb", "reads2_lb"]
    ta = batch_arrays(trecs, {x: x for x in tl}, tl)
    np.savez_compressed(os.path.join(OUT, "twolib.npz"), ref_start=np.int64(0), ref_slice=np.frombuffer(fa.encode(), np.uint8),
                        ref_len=np.int64(1000), contig=np.array("rand1k"), tid=np.int32(0), lib_names=np.array(tl), **ta)
    print("fixtures written to", os.path.abspath(OUT), "ref window", lo, lo + len(ref))


if __name__ == "__main__":
    main()
