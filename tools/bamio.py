"""Minimal pure-Python BGZF/BAM decoder used ONLY by fixture-generation scripts (tools/) and tests.

Wire format follows the public SAM/BAM specification (SAMv1 section 4).  This is not product code:
the product-side reader is C++ (bam_readcount_amd/csrc/host).  No reference code is involved.
"""
import struct
import zlib
import numpy as np


def bgzf_decompress(path):
    """Concatenate all BGZF members of `path` (each is a gzip member with a BC extra field)."""
    raw = open(path, "rb").read()
    out = []
    off = 0
    while off < len(raw):
        # gzip header: ID1 ID2 CM FLG MTIME(4) XFL OS XLEN(2)
        id1, id2, cm, flg = struct.unpack_from("<BBBB", raw, off)
        assert id1 == 31 and id2 == 139 and cm == 8 and (flg & 4), "not a BGZF member"
        xlen = struct.unpack_from("<H", raw, off + 10)[0]
        xoff = off + 12
        bsize = None
        xend = xoff + xlen
        while xoff < xend:
            si1, si2, slen = struct.unpack_from("<BBH", raw, xoff)
            if si1 == 66 and si2 == 67:
                bsize = struct.unpack_from("<H", raw, xoff + 4)[0]
            xoff += 4 + slen
        assert bsize is not None
        cdata = raw[xend: off + bsize + 1 - 8]
        out.append(zlib.decompress(cdata, -15))
        off += bsize + 1
    return b"".join(out)


def parse_aux(buf):
    """Return dict tag -> (type, value) for the aux area of one BAM record."""
    tags = {}
    o = 0
    n = len(buf)
    sizes = {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4, "A": 1}
    fmts = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f"}
    while o + 3 <= n:
        tag = buf[o:o + 2].decode()
        ty = chr(buf[o + 2])
        o += 3
        if ty in fmts:
            val = struct.unpack_from(fmts[ty], buf, o)[0]
            o += sizes[ty]
        elif ty == "A":
            val = chr(buf[o]); o += 1
        elif ty in "ZH":
            e = buf.index(b"\0", o)
            val = buf[o:e].decode(); o = e + 1
        elif ty == "B":
            sub = chr(buf[o]); cnt = struct.unpack_from("<I", buf, o + 1)[0]
            o += 5 + cnt * sizes[sub]; val = None
        else:
            raise ValueError("bad aux type " + ty)
        tags[tag] = (ty, val)
    return tags


def read_bam(path):
    """Decode a whole BAM file. Returns (header_text, [(name, length)], [record dict])."""
    data = bgzf_decompress(path)
    assert data[:4] == b"BAM\1"
    l_text = struct.unpack_from("<i", data, 4)[0]
    text = data[8:8 + l_text].split(b"\0")[0].decode()
    o = 8 + l_text
    n_ref = struct.unpack_from("<i", data, o)[0]; o += 4
    refs = []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", data, o)[0]; o += 4
        name = data[o:o + l_name - 1].decode(); o += l_name
        l_ref = struct.unpack_from("<i", data, o)[0]; o += 4
        refs.append((name, l_ref))
    recs = []
    while o < len(data):
        bs = struct.unpack_from("<i", data, o)[0]; o += 4
        (tid, pos, l_rn, mapq, _bin, n_cig, flag, l_seq, mtid, mpos, tlen) = struct.unpack_from("<iiBBHHHiiii", data, o)
        p = o + 32
        qname = data[p:p + l_rn - 1].decode(); p += l_rn
        cigar = np.frombuffer(data, dtype="<u4", count=n_cig, offset=p).copy(); p += 4 * n_cig
        seq4 = np.frombuffer(data, dtype=np.uint8, count=(l_seq + 1) // 2, offset=p).copy(); p += (l_seq + 1) // 2
        qual = np.frombuffer(data, dtype=np.uint8, count=l_seq, offset=p).copy(); p += l_seq
        aux = parse_aux(data[p:o + bs])
        recs.append(dict(tid=tid, pos=pos, mapq=mapq, flag=flag, l_seq=l_seq, qname=qname,
                         cigar=cigar, seq4=seq4, qual=qual, aux=aux))
        o += bs
    return text, refs, recs


def rg_to_lib(header_text):
    """@RG ID -> LB map (None when the RG line has no LB)."""
    m = {}
    for line in header_text.split("\n"):
        if line.startswith("@RG"):
            f = dict(x.split(":", 1) for x in line.split("\t")[1:] if ":" in x)
            m[f.get("ID")] = f.get("LB")
    return m


# ---------------------------------------------------------------- writer (tests only): BAM + BAI from brc_read_batch arrays

def _reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14: return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17: return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20: return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23: return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26: return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def _bgzf_block(payload):
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    c = co.compress(payload) + co.flush()
    bsize = len(c) + 25
    hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize)
    return hdr + c + struct.pack("<II", zlib.crc32(payload) & 0xffffffff, len(payload))


def _int_tag(tag, v, ty):
    """An integer aux field in one of BAM's six integer types (the smallest that holds v when ty is None)."""
    fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}
    lim = {"c": (-128, 127), "C": (0, 255), "s": (-32768, 32767), "S": (0, 65535), "i": (-2**31, 2**31 - 1), "I": (0, 2**32 - 1)}
    if ty is None or not (lim[ty][0] <= v <= lim[ty][1]):
        ty = "i"
    return tag + ty.encode() + struct.pack(fmt[ty], v)


def write_bam(path, contigs, arrs, tids, rg_of_read=None, rg_lines=(), qnames=None, block_bytes=20000, long_cigar=(), csi=None, int_types=None):
    """contigs: [(name, length)]; arrs: brc_read_batch arrays (coordinate-sorted per contig); tids: contig index per
    read (non-decreasing).  Writes path and path + '.bai' — or, with csi=(min_shift, depth), path + '.csi' (CSIv1: BGZF-
    compressed, bins of that geometry with a left offset each, no linear index).  Aux: NM:i / SM:i when the tags bits say so, RG:Z."""
    csi_bins = [dict() for _ in contigs]; csi_loff = [dict() for _ in contigs]
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % c for c in contigs) + "".join(l + "\n" for l in rg_lines)
    head = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(contigs))
    for name, ln in contigs:
        head += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", ln)
    out = bytearray(); cur = bytearray(head)
    coff = 0
    bins = [dict() for _ in contigs]; linear = [dict() for _ in contigs]
    n = len(arrs["pos"])

    def flush():
        nonlocal coff, cur
        if cur:
            blk = _bgzf_block(bytes(cur)); out.extend(blk); coff += len(blk); cur = bytearray()

    flush()
    for i in range(n):
        if len(cur) > block_bytes:
            flush()
        v0 = (coff << 16) | len(cur)
        L = int(arrs["l_qseq"][i]); nc = int(arrs["n_cigar"][i]); pos = int(arrs["pos"][i]); tid = int(tids[i])
        cig = arrs["cigar"][int(arrs["cigar_off"][i]):int(arrs["cigar_off"][i]) + nc]
        rl = sum(int(c) >> 4 for c in cig if (int(c) & 15) in (0, 2, 3, 7, 8))
        flag = int(arrs["flag"][i])
        end = pos + (rl if (nc and not flag & 4) else 1)
        qn = (qnames[i] if qnames is not None else "r%d" % i).encode() + b"\0"
        aux = b""
        # (int_types: per read the type letter of its NM / SM fields — c C s S i I — for readers that must take all six)
        if int(arrs["tags"][i]) & 1: aux += _int_tag(b"NM", int(arrs["nm"][i]), int_types[i] if int_types is not None else "i")
        if int(arrs["tags"][i]) & 2: aux += _int_tag(b"SM", int(arrs["sm"][i]), int_types[i] if int_types is not None else "i")
        if rg_of_read is not None and rg_of_read[i] is not None: aux += b"RGZ" + rg_of_read[i].encode() + b"\0"
        if i in long_cigar:      # SAMv1 4.2.2: the real operators travel in CG:B,I behind the placeholder <l_seq>S<span>N
            aux += b"CGBI" + struct.pack("<I", nc) + b"".join(struct.pack("<I", int(c)) for c in cig)
            cig = [(L << 4) | 4, (rl << 4) | 3]; nc = 2
        seq = bytes(arrs["seq4"][int(arrs["seq_off"][i]):int(arrs["seq_off"][i]) + (L + 1) // 2])
        qual = bytes(arrs["qual"][int(arrs["qual_off"][i]):int(arrs["qual_off"][i]) + L])
        b = _reg2bin(pos, end)
        body = struct.pack("<iiBBHHHiiii", tid, pos, len(qn), int(arrs["mapq"][i]), b & 0xffff, nc, flag, L, -1, -1, 0) + qn + \
            b"".join(struct.pack("<I", int(c)) for c in cig) + seq + qual + aux
        cur.extend(struct.pack("<i", len(body)) + body)
        v1 = (coff << 16) | len(cur)
        ch = bins[tid].setdefault(b, [])
        if ch and ch[-1][1] == v0: ch[-1][1] = v1
        else: ch.append([v0, v1])
        for w in range(pos >> 14, ((end - 1) >> 14) + 1):
            linear[tid].setdefault(w, v0)
        if csi:
            ms, dp = csi
            # the record's bin: the finest level at which it fits one bin
            l, sh, t = dp, ms, ((1 << (3 * dp)) - 1) // 7
            while l > 0 and (pos >> sh) != ((end - 1) >> sh):
                l -= 1; sh += 3; t -= 1 << (3 * l)
            cb = t + (pos >> sh)
            chc = csi_bins[tid].setdefault(cb, [])
            if chc and chc[-1][1] == v0: chc[-1][1] = v1
            else: chc.append([v0, v1])
            # left offsets: every bin (at every level) the record overlaps
            t2, sh2 = 0, ms + 3 * dp
            for l2 in range(dp + 1):
                for bb in range(t2 + (pos >> sh2), t2 + ((end - 1) >> sh2) + 1):
                    if bb not in csi_loff[tid] or v0 < csi_loff[tid][bb]: csi_loff[tid][bb] = v0
                t2 += 1 << (3 * l2); sh2 -= 3
    flush()
    out.extend(_bgzf_block(b""))
    open(path, "wb").write(bytes(out))
    if csi:
        ms, dp = csi
        raw = bytearray(b"CSI\1" + struct.pack("<iii", ms, dp, 0) + struct.pack("<i", len(contigs)))
        for t in range(len(contigs)):
            raw += struct.pack("<i", len(csi_bins[t]))
            for b, chunks in sorted(csi_bins[t].items()):
                raw += struct.pack("<IQi", b, csi_loff[t].get(b, chunks[0][0]), len(chunks))
                for a, e in chunks: raw += struct.pack("<QQ", a, e)
        comp = b"".join(_bgzf_block(bytes(raw[o:o + 60000])) for o in range(0, len(raw), 60000)) + _bgzf_block(b"")
        open(path + ".csi", "wb").write(comp)
        return
    bai = bytearray(b"BAI\1" + struct.pack("<i", len(contigs)))
    for t in range(len(contigs)):
        bai += struct.pack("<i", len(bins[t]))
        for b, chunks in sorted(bins[t].items()):
            bai += struct.pack("<Ii", b, len(chunks))
            for a, e in chunks: bai += struct.pack("<QQ", a, e)
        nl = (max(linear[t]) + 1) if linear[t] else 0
        bai += struct.pack("<i", nl)
        last = 0
        for w in range(nl):
            last = linear[t].get(w, last)
            bai += struct.pack("<Q", last)
    open(path + ".bai", "wb").write(bytes(bai))
