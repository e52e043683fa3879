"""Test-side CRAM 3.0 writer (no htslib here): turns brc_read_batch arrays into a reference-based CRAM so the minimal
CRAM reader of the drop-in CLI (bam_readcount_amd/csrc/io/cram.cpp) can be exercised on more than the reference's one
4-read fixture.  Written from the public CRAM 3.0 specification.  It deliberately uses every codec the reader supports:

  BF           HUFFMAN with real multi-symbol canonical codes (core bit stream)
  MQ           BETA (core bit stream)
  CF, TL       HUFFMAN single symbol, zero-length code
  RN, SC       BYTE_ARRAY_STOP            IN, BB      BYTE_ARRAY_LEN(EXTERNAL, EXTERNAL)
  tags         BYTE_ARRAY_LEN(HUFFMAN single symbol, EXTERNAL)
  the rest     EXTERNAL (ITF8 / bytes); the blocks cycle through `methods` (default: raw, gzip; also bzip2, lzma and
               rANS 4x8 order 0 / order 1 — the encoder below is the test-side counterpart of the reader's decoder)
  optional     RL as GAMMA, DL as SUBEXP (core bit stream) with int_codecs=True

Reads become features against the reference: X (substitution matrix), B (base+quality, for IUPAC read bases), I / i, D, N,
S, H, P.  A chunk of reads that spans two contigs is written as a multi-reference slice (ref id -2, RI series, absolute
AP); single-reference slices use delta AP.  '=' / 'X' CIGAR operators come back as 'M' (CRAM does not keep them)."""
import heapq
import struct
import zlib

NT16 = "=ACMGRSVTWYHKDBN"


def itf8(v):
    v &= 0xFFFFFFFF
    if v < 0x80: return bytes([v])
    if v < 0x4000: return bytes([0x80 | (v >> 8), v & 0xFF])
    if v < 0x200000: return bytes([0xC0 | (v >> 16), (v >> 8) & 0xFF, v & 0xFF])
    if v < 0x10000000: return bytes([0xE0 | (v >> 24), (v >> 16) & 0xFF, (v >> 8) & 0xFF, v & 0xFF])
    return bytes([0xF0 | (v >> 28), (v >> 20) & 0xFF, (v >> 12) & 0xFF, (v >> 4) & 0xFF, v & 0x0F])


def ltf8(v):
    assert 0 <= v < (1 << 28)
    return itf8(v)          # identical to ITF8 below 2^28


RANS_L = 1 << 23


def rans_normalise(cnt):
    """counts -> 12-bit frequencies summing to 4096, every present symbol >= 1"""
    tot = sum(cnt)
    F = [max(1, c * 4096 // tot) if c else 0 for c in cnt]
    while sum(F) != 4096:
        d = 4096 - sum(F)
        j = max(range(256), key=lambda k: F[k])
        F[j] += d if d > 0 or F[j] + d >= 1 else -(F[j] - 1)
    return F


def rans_table(F):
    """one frequency table: symbols ascending; a symbol whose predecessor is present is followed by the number of further
    consecutive symbols, which then come without their symbol byte"""
    out = bytearray(); rle = 0
    for j in range(256):
        if not F[j]: continue
        if rle: rle -= 1
        else:
            out.append(j)
            if j and F[j - 1]:
                r = j + 1
                while r < 256 and F[r]: r += 1
                rle = r - (j + 1); out.append(rle)
        out += bytes([F[j]]) if F[j] < 128 else bytes([128 | (F[j] >> 8), F[j] & 0xFF])
    out.append(0)
    return out


def rans_put(x, out, f, c):
    x_max = ((RANS_L >> 12) << 8) * f
    while x >= x_max:
        out.append(x & 0xFF); x >>= 8
    return ((x // f) << 12) + (x % f) + c


def rans_encode(data, order):
    """CRAM 3.0 rANS 4x8: order u8, compressed size u32, uncompressed size u32, tables, 4 states, bytes"""
    n = len(data); x = [RANS_L] * 4; rev = bytearray()
    if order == 0:
        cnt = [0] * 256
        for b in data: cnt[b] += 1
        F = rans_normalise(cnt); C = [0] * 256
        for j in range(1, 256): C[j] = C[j - 1] + F[j - 1]
        tab = rans_table(F)
        for i in range(n - 1, -1, -1):
            x[i & 3] = rans_put(x[i & 3], rev, F[data[i]], C[data[i]])
    else:
        q = n >> 2
        def ctx(i, k): return data[i - 1] if i > k * q else 0
        cnt = {}
        for k in range(4):
            for i in range(k * q, (k + 1) * q if k < 3 else n):
                cnt.setdefault(ctx(i, k), [0] * 256)[data[i]] += 1
        F = {c: rans_normalise(v) for c, v in cnt.items()}; C = {}
        for c, f in F.items():
            C[c] = [0] * 256
            for j in range(1, 256): C[c][j] = C[c][j - 1] + f[j - 1]
        present = [1 if c in F else 0 for c in range(256)]
        tab = bytearray(); rle = 0
        for i in range(256):
            if not present[i]: continue
            if rle: rle -= 1
            else:
                tab.append(i)
                if i and present[i - 1]:
                    r = i + 1
                    while r < 256 and present[r]: r += 1
                    rle = r - (i + 1); tab.append(rle)
            tab += rans_table(F[i])
        tab.append(0)
        for i in range(n - 1, 4 * q - 1, -1):
            c = ctx(i, 3); x[3] = rans_put(x[3], rev, F[c][data[i]], C[c][data[i]])
        for t in range(q - 1, -1, -1):
            for k in (3, 2, 1, 0):
                i = k * q + t; c = ctx(i, k); x[k] = rans_put(x[k], rev, F[c][data[i]], C[c][data[i]])
    body = bytes(tab) + struct.pack("<4I", *x) + bytes(reversed(rev))
    return bytes([order]) + struct.pack("<II", len(body), n) + body


def rans_decode(comp):
    """Plain decoder of rans_encode's output (test infrastructure: checks the encoder on its own)."""
    order = comp[0]; csz, n = struct.unpack_from("<II", comp, 1); assert csz + 9 == len(comp)
    p = [9]
    def u8():
        v = comp[p[0]]; p[0] += 1; return v
    def table():
        F = [0] * 256; C = [0] * 256; R = [0] * 4096; x = 0; rle = 0; j = u8()
        while True:
            f = u8()
            if f >= 128: f = ((f & 127) << 8) | u8()
            F[j] = f; C[j] = x
            for t in range(x, x + f): R[t] = j
            x += f
            if not rle and j + 1 == comp[p[0]]: j = u8(); rle = u8()
            elif rle: rle -= 1; j += 1
            else: j = u8()
            if j == 0: break
        return F, C, R
    out = bytearray(n)
    if n == 0: return bytes(out)
    def step(x, T):
        F, C, R = T; m = x & 4095; s = R[m]; x = F[s] * (x >> 12) + m - C[s]
        while x < RANS_L: x = (x << 8) | u8()
        return x, s
    if order == 0:
        T = table(); X = list(struct.unpack_from("<4I", comp, p[0])); p[0] += 16
        full = n & ~3
        for i in range(full): X[i & 3], out[i] = step(X[i & 3], T)
        for i in range(full, n): out[i] = T[2][X[i & 3] & 4095]
        return bytes(out)
    tabs = {}; rle = 0; i = u8()
    while True:
        tabs[i] = table()
        if not rle and i + 1 == comp[p[0]]: i = u8(); rle = u8()
        elif rle: rle -= 1; i += 1
        else: i = u8()
        if i == 0: break
    X = list(struct.unpack_from("<4I", comp, p[0])); p[0] += 16
    q = n >> 2; at = [0, q, 2 * q, 3 * q]; last = [0, 0, 0, 0]
    for t in range(q):
        for k in range(4):
            X[k], s = step(X[k], tabs[last[k]]); out[at[k]] = s; at[k] += 1; last[k] = s
    while at[3] < n:
        X[3], s = step(X[3], tabs[last[3]]); out[at[3]] = s; at[3] += 1; last[3] = s
    return bytes(out)


def block(method, ctype, cid, data):
    """method: 0 raw, 1 gzip, 2 bzip2, 3 lzma, 4 rANS order 0, 5 -> rANS order 1 (written as method 4)"""
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
    b = bytes([method, ctype]) + itf8(cid) + itf8(len(comp)) + itf8(len(data)) + comp
    return b + struct.pack("<I", zlib.crc32(b))


def enc(codec, params):
    return itf8(codec) + itf8(len(params)) + params


def e_ext(cid): return enc(1, itf8(cid))
def e_huff1(sym): return enc(3, itf8(1) + itf8(sym) + itf8(1) + itf8(0))
def e_stop(stop, cid): return enc(5, bytes([stop]) + itf8(cid))
def e_len(le, ve): return enc(4, le + ve)
def e_beta(off, bits): return enc(6, itf8(off) + itf8(bits))
def e_subexp(off, k): return enc(7, itf8(off) + itf8(k))
def e_gamma(off): return enc(9, itf8(off))


def huffman_lengths(freq):
    if len(freq) == 1: return {next(iter(freq)): 0}
    heap = [(f, i, [s]) for i, (s, f) in enumerate(sorted(freq.items()))]
    heapq.heapify(heap); lens = {s: 0 for s in freq}; k = len(heap)
    while len(heap) > 1:
        fa, _, sa = heapq.heappop(heap); fb, _, sb = heapq.heappop(heap)
        for s in sa + sb: lens[s] += 1
        heapq.heappush(heap, (fa + fb, k, sa + sb)); k += 1
    return lens


def canonical(lens):
    order = sorted(lens, key=lambda s: (lens[s], s))
    codes = {}; code = 0; prev = lens[order[0]]
    for s in order:
        code <<= (lens[s] - prev); prev = lens[s]; codes[s] = code; code += 1
    return order, codes


class Bits:
    def __init__(self): self.out = bytearray(); self.n = 0
    def put(self, v, k):
        for i in range(k - 1, -1, -1):
            if self.n % 8 == 0: self.out.append(0)
            if (v >> i) & 1: self.out[-1] |= 1 << (7 - self.n % 8)
            self.n += 1
    def gamma(self, v, off):
        v += off; assert v >= 1
        nb = v.bit_length() - 1
        self.put(0, nb); self.put(v, nb + 1)
    def subexp(self, v, off, k):
        v += off; assert v >= 0
        if v < (1 << k): self.put(0, 1); self.put(v, k); return
        b = v.bit_length() - 1                      # b >= k;  i = b - k + 1 ones, a zero, then the b low bits
        i = b - k + 1
        self.put((1 << i) - 1, i); self.put(0, 1); self.put(v - (1 << b), b)


SM_ORDER = {"A": "CGTN", "C": "AGTN", "G": "ACTN", "T": "ACGN", "N": "ACGT"}   # alternatives in ACGTN order minus self
# substitution codes: a permutation per reference base (deliberately not the identity)
SM_CODES = {"A": [1, 0, 3, 2], "C": [0, 2, 1, 3], "G": [3, 2, 1, 0], "T": [2, 3, 0, 1], "N": [0, 1, 2, 3]}


def sm_bytes():
    return bytes(sum(c << (6 - 2 * k) for k, c in enumerate(SM_CODES[r])) for r in "ACGTN")


def refclass(ch):
    ch = ch.upper()
    return ch if ch in "ACGT" else "N"


# data-series content ids
IDS = dict(RI=1, RL=2, AP=3, RG=4, RN=5, MF=6, NS=7, NP=8, TS=9, NF=10, FN=11, FC=12, FP=13, BS=14, IN=15, INL=16, SC=17, DL=18, RS=19,
           HC=20, PD=21, BA=22, QS=23, BB=24, BBL=25, QQ=26)


def write_cram(path, contigs, arrs, tids, refs, rg_of_read=None, rg_lines=(), qnames=None, per_container=300, methods=(0, 1), int_codecs=False,
               embed_ref=False, no_ref=False, write_crai=False):
    """contigs [(name, len)], arrs = brc_read_batch arrays, tids = contig per read, refs = list of uint8 reference arrays.
    embed_ref: every single-reference slice carries its stretch of the reference as an external block (slice header field
    "embedded reference bases block content id"); no_ref: RR = 0 in the preservation map and every aligned base stored as
    a 'b' feature (what samtools writes with no_ref=1)."""
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % c for c in contigs) + "".join(l + "\n" for l in rg_lines)
    rg_ids = [dict(f.split(":", 1) for f in l.split("\t")[1:])["ID"] for l in rg_lines]
    out = bytearray(b"CRAM" + bytes([3, 0]) + b"brc-test-cram".ljust(20, b"\0"))

    def container(ref, start, span, nrec, blocks, landmarks):
        body = b"".join(blocks)
        h = struct.pack("<i", len(body)) + itf8(ref) + itf8(start) + itf8(span) + itf8(nrec) + ltf8(0) + ltf8(0) + itf8(len(blocks)) + \
            itf8(len(landmarks)) + b"".join(itf8(x) for x in landmarks)
        return h + struct.pack("<I", zlib.crc32(h)) + body

    out += container(0, 0, 0, 0, [block(0, 0, 0, struct.pack("<i", len(text)) + text.encode())], [0])
    crai = []                                                                    # lines of the .crai index
    n = len(arrs["pos"])
    for c0 in range(0, n, per_container):
        idx = range(c0, min(n, c0 + per_container))
        ctids = sorted(set(int(tids[i]) for i in idx))
        multi = len(ctids) > 1
        ext = {k: bytearray() for k in IDS}
        tag_blocks = {}; td_lines = []; core = Bits()
        flags = {}
        for i in idx: flags[int(arrs["flag"][i])] = flags.get(int(arrs["flag"][i]), 0) + 1
        blens = huffman_lengths(flags); border, bcodes = canonical(blens)
        first_pos = int(arrs["pos"][c0]) + 1
        last_ap = first_pos; max_end = first_pos
        extent = {}                                                              # tid -> [first start (1-based), last end] of this container
        for i in idx:
            L = int(arrs["l_qseq"][i]); nc = int(arrs["n_cigar"][i]); pos = int(arrs["pos"][i]); tid = int(tids[i]); flag = int(arrs["flag"][i])
            cig = [int(x) for x in arrs["cigar"][int(arrs["cigar_off"][i]):int(arrs["cigar_off"][i]) + nc]]
            s4 = arrs["seq4"][int(arrs["seq_off"][i]):int(arrs["seq_off"][i]) + (L + 1) // 2]
            seq = "".join(NT16[(int(s4[j >> 1]) >> (4 if j % 2 == 0 else 0)) & 15] for j in range(L))
            qual = bytes(arrs["qual"][int(arrs["qual_off"][i]):int(arrs["qual_off"][i]) + L])
            core.put(bcodes[flag], blens[flag])                                  # BF
            if multi: ext["RI"] += itf8(tid)
            if int_codecs: core.gamma(L, 1)
            else: ext["RL"] += itf8(L)
            ext["AP"] += itf8(pos + 1) if multi else itf8(pos + 1 - last_ap)
            last_ap = pos + 1
            rg = rg_of_read[i] if rg_of_read is not None else None
            ext["RG"] += itf8(rg_ids.index(rg) if rg in rg_ids else -1)
            ext["RN"] += (qnames[i] if qnames is not None else "r%d" % i).encode() + b"\0"
            # tags
            tl = []
            if int(arrs["tags"][i]) & 1: tl.append((b"NMC" if 0 <= int(arrs["nm"][i]) < 256 else b"NMi", int(arrs["nm"][i])))
            if int(arrs["tags"][i]) & 2: tl.append((b"SMC" if 0 <= int(arrs["sm"][i]) < 256 else b"SMi", int(arrs["sm"][i])))
            line = b"".join(k for k, _ in tl)
            if line not in td_lines: td_lines.append(line)
            ext.setdefault("TL", bytearray()); ext["TL"] += itf8(td_lines.index(line))
            for k, v in tl:
                key = (k[0] << 16) | (k[1] << 8) | k[2]
                tag_blocks.setdefault(key, bytearray()); tag_blocks[key] += bytes([v]) if k[2:] == b"C" else struct.pack("<i", v)
            ex = extent.setdefault(tid, [pos + 1, pos + 1]); ex[0] = min(ex[0], pos + 1)
            if not flag & 4:
                ref = refs[tid]; feats = []; rp = pos; sp = 1
                for c in cig:
                    op, ln = c & 15, c >> 4
                    if op in (0, 7, 8) and no_ref:
                        feats.append((sp, "b", seq[sp - 1:sp - 1 + ln])); rp += ln; sp += ln
                    elif op in (0, 7, 8):
                        for j in range(ln):
                            rb = chr(ref[rp + j]).upper() if 0 <= rp + j < len(ref) else "N"
                            b = seq[sp - 1 + j]
                            if b == rb: continue
                            if b in "ACGTN" and refclass(rb) != b: feats.append((sp + j, "X", SM_CODES[refclass(rb)][SM_ORDER[refclass(rb)].index(b)]))
                            else: feats.append((sp + j, "B", (b, qual[sp - 1 + j])))
                        rp += ln; sp += ln
                    elif op == 1:
                        feats.append((sp, "i", seq[sp - 1]) if ln == 1 and (sp % 2) else (sp, "I", seq[sp - 1:sp - 1 + ln])); sp += ln
                    elif op == 4: feats.append((sp, "S", seq[sp - 1:sp - 1 + ln])); sp += ln
                    elif op == 2: feats.append((sp, "D", ln)); rp += ln
                    elif op == 3: feats.append((sp, "N", ln)); rp += ln
                    elif op == 5: feats.append((sp, "H", ln))
                    elif op == 6: feats.append((sp, "P", ln))
                max_end = max(max_end, rp); ex[1] = max(ex[1], rp)
                ext["FN"] += itf8(len(feats)); prev = 0
                for fp, fc, v in feats:
                    ext["FC"] += fc.encode(); ext["FP"] += itf8(fp - prev); prev = fp
                    if fc == "X": ext["BS"].append(v)
                    elif fc == "B": ext["BA"] += v[0].encode(); ext["QS"].append(v[1])
                    elif fc == "i": ext["BA"] += v.encode()
                    elif fc == "b": ext["BBL"] += itf8(len(v)); ext["BB"] += v.encode()
                    elif fc == "I": ext["INL"] += itf8(len(v)); ext["IN"] += v.encode()
                    elif fc == "S": ext["SC"] += v.encode() + b"\0"
                    elif fc == "D":
                        if int_codecs: core.subexp(v, 0, 2)
                        else: ext["DL"] += itf8(v)
                    elif fc == "N": ext["RS"] += itf8(v)
                    elif fc == "H": ext["HC"] += itf8(v)
                    elif fc == "P": ext["PD"] += itf8(v)
                core.put(int(arrs["mapq"][i]) + 3, 9)                            # MQ: BETA offset 3, 9 bits
                ext["QS"] += qual
            else:
                ext["BA"] += seq.encode(); ext["QS"] += qual
                max_end = max(max_end, pos + 1)
        # ---- compression header
        td = b"".join(l + b"\0" for l in td_lines)
        pres = [b"RN" + b"\1", b"AP" + (b"\0" if multi else b"\1"), b"RR" + (b"\0" if no_ref else b"\1"), b"SM" + sm_bytes(), b"TD" + itf8(len(td)) + td]
        pm = itf8(len(pres)) + b"".join(pres)
        dse = {"BF": enc(3, itf8(len(border)) + b"".join(itf8(s) for s in border) + itf8(len(border)) + b"".join(itf8(blens[s]) for s in border)),
               "CF": e_huff1(1), "MQ": e_beta(3, 9), "RN": e_stop(0, IDS["RN"]), "SC": e_stop(0, IDS["SC"]),
               "IN": e_len(e_ext(IDS["INL"]), e_ext(IDS["IN"])), "BB": e_len(e_ext(IDS["BBL"]), e_ext(IDS["BB"])),
               "QQ": e_len(e_ext(IDS["BBL"]), e_ext(IDS["QQ"]))}
        tl_ids = 27
        dse["TL"] = e_huff1(0) if len(td_lines) == 1 else e_ext(tl_ids)
        for k in ("RI", "RL", "AP", "RG", "MF", "NS", "NP", "TS", "NF", "FN", "FC", "FP", "BS", "DL", "RS", "HC", "PD", "BA", "QS"):
            dse[k] = e_ext(IDS[k])
        if int_codecs: dse["RL"] = e_gamma(1); dse["DL"] = e_subexp(0, 2)
        dm = itf8(len(dse)) + b"".join(k.encode() + v for k, v in dse.items())
        te = {key: e_len(e_huff1(1 if (key & 0xFF) == ord("C") else 4), e_ext(key)) for key in tag_blocks}
        tm = itf8(len(te)) + b"".join(itf8(k) + v for k, v in te.items())
        ch = block(0, 1, 0, itf8(len(pm)) + pm + itf8(len(dm)) + dm + itf8(len(tm)) + tm)
        # ---- slice
        eblocks = []
        for j, (k, cid) in enumerate(sorted(IDS.items(), key=lambda kv: kv[1])):
            if ext[k]: eblocks.append(block(methods[(j + c0 // per_container) % len(methods)], 4, cid, bytes(ext[k])))
        if len(td_lines) > 1: eblocks.append(block(1, 4, tl_ids, bytes(ext["TL"])))
        for key, data in tag_blocks.items(): eblocks.append(block(methods[-1], 4, key, bytes(data)))
        sref = -2 if multi else ctids[0]
        sstart = 0 if multi else first_pos; sspan = 0 if multi else max_end - first_pos + 1
        ids = [IDS[k] for k in sorted(IDS, key=lambda kk: IDS[kk]) if ext[k]] + ([tl_ids] if len(td_lines) > 1 else []) + list(tag_blocks)
        emb_id = -1
        if embed_ref and not multi:
            emb_id = 99
            r0 = refs[ctids[0]]
            stretch = bytes(r0[sstart - 1:sstart - 1 + sspan]) + b"N" * max(0, sstart - 1 + sspan - len(r0))
            eblocks.append(block(methods[0], 4, emb_id, stretch)); ids.append(emb_id)
        sh = itf8(sref) + itf8(sstart) + itf8(sspan) + itf8(len(idx)) + ltf8(c0) + itf8(1 + len(eblocks)) + itf8(len(ids)) + \
            b"".join(itf8(x) for x in ids) + itf8(emb_id) + bytes(16)
        blocks = [ch, block(0, 2, 0, sh), block(methods[-1] if len(methods) > 2 else 0, 5, 0, bytes(core.out))] + eblocks
        cbytes = container(sref, sstart, sspan, len(idx), blocks, [len(ch)])
        for t in sorted(extent):
            crai.append("%d\t%d\t%d\t%d\t%d\t%d\n" % (t, extent[t][0], extent[t][1] - extent[t][0] + 1, len(out), len(ch), len(cbytes) - len(ch)))
        out += cbytes
    out += bytes.fromhex("0f000000ffffffff0fe0454f460000000001000 5bdd94f0001000606010001000100ee63014b".replace(" ", ""))
    open(path, "wb").write(bytes(out))
    if write_crai:
        import gzip
        gzip.open(path + ".crai", "wb").write("".join(crai).encode())
