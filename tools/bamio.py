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
