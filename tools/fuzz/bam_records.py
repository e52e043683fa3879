#!/usr/bin/env python3
"""Corrupted BAM RECORDS under the sanitizers: a BAM written with stored (uncompressed) BGZF blocks keeps its block layout —
and with it its index — when bytes inside records change, so the reader gets well-formed blocks with damaged fields
(block_size, l_seq, n_cigar, l_qname, pos, tid, any byte of the variable part).  Expected: exit code 0 or 1, never a
sanitizer report.    python tools/fuzz/bam_records.py <bam-readcount-asan> <seed> <cases> <scratch dir>"""
import subprocess, random, os, sys, struct, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tools')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, bamio, synth
CLI, SEED, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
os.makedirs(sys.argv[4], exist_ok=True); os.chdir(sys.argv[4])
random.seed(SEED)
def stored_block(payload):
    body = b"\x01" + struct.pack("<HH", len(payload), len(payload) ^ 0xffff) + payload
    bsize = 18 + len(body) + 8 - 1
    return b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize) + body + struct.pack("<II", zlib.crc32(payload) & 0xffffffff, len(payload))
bamio._bgzf_block = stored_block
rng = np.random.default_rng(3)
ref = synth.make_ref(rng, 3600)
arrs = synth.make_batch(9, ref, 500, style="mixed", n_libs=2, region=(0, 3000))
rgs = [["rg0", "rg1"][int(l)] if l >= 0 else None for l in arrs["lib"]]
bamio.write_bam("g.bam", [("chrA", 3600)], arrs, np.zeros(len(arrs["pos"]), int), rg_of_read=rgs, rg_lines=["@RG\tID:rg0\tLB:libA\tSM:s", "@RG\tID:rg1\tLB:libB\tSM:s"], block_bytes=8000)
open("g.fa", "w").write(">chrA\n" + bytes(ref).decode() + "\n"); open("g.fa.fai", "w").write("chrA\t3600\t6\t3600\t3601\n")
f = open("g.bam", "rb").read()
sizes = []; o = 0
while o < len(f):
    bsize = struct.unpack_from("<H", f, o + 16)[0] + 1
    sizes.append(struct.unpack_from("<I", f, o + bsize - 4)[0]); o += bsize
raw = bamio.bgzf_decompress("g.bam")
l_text = struct.unpack_from("<i", raw, 4)[0]; o = 8 + l_text; n_ref = struct.unpack_from("<i", raw, o)[0]; o += 4
for _ in range(n_ref):
    ln = struct.unpack_from("<i", raw, o)[0]; o += 4 + ln + 4
offs = []
while o + 4 <= len(raw):
    bs = struct.unpack_from("<i", raw, o)[0]; offs.append(o); o += 4 + bs
open("gs.txt", "w").write("chrA\t100\t105\nchrA\t2000\t2000\n")
env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1")
issues = 0; rcs = {}
try: os.remove("x.bam.bai")
except FileNotFoundError: pass
os.symlink("g.bam.bai", "x.bam.bai")
for it in range(N):
    b = bytearray(raw)
    for _ in range(random.randint(1, 3)):
        ro = offs[random.randrange(len(offs))]
        bs = struct.unpack_from("<i", raw, ro)[0]
        k = random.random()
        if k < 0.2: b[ro + 4 + random.choice([8, 12, 13, 16, 17, 18, 19])] = random.randrange(256)
        elif k < 0.4: struct.pack_into("<i", b, ro + 4 + 16, random.choice([-1, 0, 2**31 - 1, 70000, 5, 151]))
        elif k < 0.55: struct.pack_into("<i", b, ro + 4 + 4, random.choice([-1, 2**31 - 1, 0, 3599, 10**6]))
        elif k < 0.65: struct.pack_into("<i", b, ro + 4, random.choice([-1, 5, 1]))
        elif k < 0.75: struct.pack_into("<i", b, ro, random.choice([0, 31, 32, 33, bs + 7, bs - 3, 2**31 - 1, -5]))
        else: b[ro + 36 + random.randrange(max(bs - 32, 1))] = random.randrange(256)
    data = bytes(b); out = bytearray(); pos = 0
    for sz in sizes:
        out += stored_block(data[pos:pos + sz]); pos += sz
    open("x.bam", "wb").write(bytes(out))
    try:
        p = subprocess.run([CLI,'-w','1','-f','g.fa'] + random.choice([['x.bam', 'chrA:1-3600'], ['-p', 'x.bam', 'chrA:1-3600'], ['-l', 'gs.txt', 'x.bam']]), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
    except subprocess.TimeoutExpired:
        issues += 1; print("TIMEOUT", it); continue
    rcs[p.returncode] = rcs.get(p.returncode, 0) + 1
    if b"AddressSanitizer" in p.stderr or b"runtime error" in p.stderr or p.returncode < 0 or p.returncode > 1:
        issues += 1; print("ISSUE", it, p.returncode, p.stderr[-3000:].decode(errors='replace'))
        if issues > 2: break
print("done issues", issues, rcs)
