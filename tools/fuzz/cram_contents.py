#!/usr/bin/env python3
"""Damaged CRAM contents under the sanitizers, with VALID block CRC32s: mode "contents" changes the uncompressed bytes of
compression headers, slice headers, core and external blocks before they are compressed; mode "compressed" changes the
compressed payload (gzip, bzip2, lzma, rANS order 0/1) or the declared sizes and signs the result.  Expected: exit code 0
or 1, never a sanitizer report.    python tools/fuzz/cram_contents.py <bam-readcount-asan> <seed> <cases> <scratch dir> contents|compressed"""
import subprocess, random, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tools')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, cramio, synth
CLI, SEED, N, MODE = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[5]
os.makedirs(sys.argv[4], exist_ok=True); os.chdir(sys.argv[4])
random.seed(SEED)
rng = np.random.default_rng(4)
refs = [synth.make_ref(rng, 3600), synth.make_ref(rng, 2600)]
parts = [synth.make_batch(21, refs[0], 400, style="mixed", n_libs=2, region=(0, 3000)), synth.make_batch(22, refs[1], 200, style="wild", n_libs=2, region=(0, 2000))]
arrs = {}
for k in ("pos", "flag", "mapq", "lib", "l_qseq", "n_cigar", "nm", "sm", "tags"): arrs[k] = np.concatenate([p[k] for p in parts])
for arena, off in (("cigar", "cigar_off"), ("seq4", "seq_off"), ("qual", "qual_off")):
    arrs[arena] = np.concatenate([p[arena] for p in parts]); arrs[off] = np.concatenate([parts[0][off], parts[1][off] + np.uint64(parts[0][arena].size)])
tids = np.concatenate([np.zeros(400, int), np.ones(200, int)])
rgs = [["rg0", "rg1"][int(l)] if l >= 0 else None for l in arrs["lib"]]
with open("c.fa", "w") as f:
    f.write(">chrA\n" + bytes(refs[0]).decode() + "\n>chrB\n" + bytes(refs[1]).decode() + "\n")
open("c.fa.fai", "w").write("chrA\t3600\t6\t3600\t3601\nchrB\t2600\t3613\t2600\t2601\n")
orig_block = cramio.block
state = {"p": 0.0}
def bad_block(method, ctype, cid, data):
    data = bytearray(data)
    if ctype in (1, 2, 4, 5) and len(data) and random.random() < state["p"]:
        k = random.random()
        if k < 0.6:
            for _ in range(random.randint(1, 4)): data[random.randrange(len(data))] = random.randrange(256)
        elif k < 0.8: data = data[:random.randrange(len(data))]
        else: data += bytes(random.randrange(256) for _ in range(random.randint(1, 9)))
    return orig_block(method, ctype, cid, bytes(data))

import zlib, struct
def bad_block2(method, ctype, cid, data):
    # corrupt the COMPRESSED payload (and sometimes the declared sizes), then a valid CRC over the result
    from cramio import itf8, rans_encode
    comp = data
    if not data and method > 1: method = 0
    if method == 1:
        co = zlib.compressobj(6, zlib.DEFLATED, 31); comp = co.compress(data) + co.flush()
    elif method == 2:
        import bz2; comp = bz2.compress(data)
    elif method == 3:
        import lzma; comp = lzma.compress(data)
    elif method in (4, 5):
        comp = rans_encode(data, method - 4); method = 4
    comp = bytearray(comp); us = len(data)
    if method != 0 and len(comp) and random.random() < state["p"]:
        k = random.random()
        if k < 0.6:
            for _ in range(random.randint(1, 4)): comp[random.randrange(len(comp))] = random.randrange(256)
        elif k < 0.75: comp = comp[:random.randrange(len(comp))]
        elif k < 0.9: us = random.choice([0, 1, us + 100, us * 3 + 7, max(us - 5, 0), 2**27])
        else: comp += bytes(random.randrange(256) for _ in range(random.randint(1, 9)))
    b = bytes([method, ctype]) + itf8(cid) + itf8(len(comp)) + itf8(us) + bytes(comp)
    return b + struct.pack("<I", zlib.crc32(b))

cramio.block = bad_block2 if MODE == 'compressed' else bad_block
env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1")
issues = 0; rcs = {}
for it in range(N):
    state["p"] = random.choice([0.05, 0.15, 0.4])
    kw = random.choice([dict(methods=(4, 5, 0, 1, 2, 3, 5), int_codecs=True), dict(methods=(4, 5)), dict(methods=(1, 2, 3)), dict(methods=(5,))] if MODE == 'compressed' else [dict(), dict(methods=(4, 5, 0, 1, 2, 3, 5), int_codecs=True), dict(embed_ref=True), dict(no_ref=True)])
    try:
        cramio.write_cram("x.cram", [("chrA", 3600), ("chrB", 2600)], arrs, tids, refs, rg_of_read=rgs, rg_lines=["@RG\tID:rg0\tLB:libA\tSM:s", "@RG\tID:rg1\tLB:libB\tSM:s"], per_container=150, **kw)
    except Exception as e:
        continue
    try:
        p = subprocess.run([CLI, '-w', '1', '-f', 'c.fa', 'x.cram', 'chrA', 'chrB:1-2000'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
    except subprocess.TimeoutExpired:
        issues += 1; print("TIMEOUT", it, kw); continue
    rcs[p.returncode] = rcs.get(p.returncode, 0) + 1
    if b"AddressSanitizer" in p.stderr or b"runtime error" in p.stderr or p.returncode < 0 or p.returncode > 1:
        issues += 1; print("ISSUE", it, kw, p.returncode, p.stderr[-3500:].decode(errors='replace'))
        if issues > 2: break
print("done issues", issues, rcs)
