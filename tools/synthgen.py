"""ctypes wrapper of tools/synth_gen.c (bench / full-size test infrastructure)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libsynth.so")


class Params(C.Structure):
    _fields_ = [("contig_len", C.c_int64), ("n_reads", C.c_int64), ("read_len", C.c_int32), ("n_libs", C.c_int32),
                ("seed", C.c_uint64), ("p_sub", C.c_double), ("p_clip", C.c_double), ("p_ins", C.c_double), ("p_del", C.c_double),
                ("indel_max", C.c_int32), ("n_chunks", C.c_int32), ("p_trim", C.c_double), ("p_long", C.c_double), ("trim_min", C.c_int32), ("long_len", C.c_int32),
                ("qual_bins", C.c_int32), ("p_dup", C.c_double), ("p_sec", C.c_double), ("p_supp", C.c_double), ("p_mapq0", C.c_double), ("eqx", C.c_int32)]


def build():
    src = os.path.join(HERE, "synth_gen.c")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-std=c99", "-fopenmp", "-fPIC", "-shared", src, "-o", LIB, "-lm"])
    build_bamwrite()
    return LIB


BAMLIB = os.path.join(HERE, "libbamwrite.so")


def build_bamwrite():
    src = os.path.join(HERE, "bam_write.c")
    if not os.path.exists(BAMLIB) or os.path.getmtime(BAMLIB) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-fopenmp", "-fPIC", "-shared", src, "-o", BAMLIB, "-lz"])
    return BAMLIB


_ORDER = [("pos", np.int32), ("flag", np.uint16), ("mapq", np.uint8), ("lib", np.int16), ("l_qseq", np.int32), ("n_cigar", np.uint32),
          ("cigar_off", np.uint64), ("seq_off", np.uint64), ("qual_off", np.uint64), ("nm", np.int32), ("sm", np.int32), ("tags", np.uint8),
          ("cigar", np.uint32), ("seq4", np.uint8), ("qual", np.uint8)]


def header_text(contigs, n_libs=1, rgs_per_lib=1):
    """@HD / @SQ / @RG lines: read group rg<l * rgs_per_lib + j> belongs to library lib<l>."""
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (nm, ln) for nm, ln in contigs)
    if n_libs > 1:
        text += "".join("@RG\tID:rg%d\tLB:lib%d\tSM:s\n" % (k, k // rgs_per_lib) for k in range(n_libs * rgs_per_lib))
    return text


class BamWriter:
    """Multi-contig BAM + BAI written contig by contig (tools/bam_write.c): add(tid, arrays) in @SQ order, then close()."""

    def __init__(self, path, contigs, n_libs=1, rgs_per_lib=1, block_bytes=60000, level=1):
        self.L = C.CDLL(build_bamwrite())
        self.L.brc_bamw_open.restype = C.c_void_p
        names = (C.c_char_p * len(contigs))(*[nm.encode() for nm, _ in contigs])
        lens = (C.c_int32 * len(contigs))(*[int(ln) for _, ln in contigs])
        self.n_libs, self.rgs = n_libs, rgs_per_lib
        self.h = self.L.brc_bamw_open(path.encode(), header_text(contigs, n_libs, rgs_per_lib).encode(), C.c_int32(len(contigs)), names, lens,
                                      C.c_int32(block_bytes), C.c_int32(level))
        if not self.h:
            raise RuntimeError("cannot create %s" % path)

    def add(self, tid, arrs):
        keep = [np.ascontiguousarray(arrs[k], dt) for k, dt in _ORDER]
        rc = self.L.brc_bamw_add(C.c_void_p(self.h), C.c_int32(tid), C.c_int64(len(arrs["pos"])), C.c_int32(self.n_libs), C.c_int32(self.rgs),
                                 *[a.ctypes.data_as(C.c_void_p) for a in keep])
        if rc != 0:
            raise RuntimeError("brc_bamw_add failed: %d" % rc)

    def close(self):
        if self.h:
            rc = self.L.brc_bamw_close(C.c_void_p(self.h)); self.h = None
            if rc != 0:
                raise RuntimeError("brc_bamw_close failed: %d" % rc)


def write_bam(path, contig, contig_len, arrs, n_libs=1, block_bytes=60000, level=1, rgs_per_lib=1):
    """Fast single-contig BAM + BAI of a brc_read_batch (tools/bam_write.c); libraries become @RG rg<k> with LB lib<k // rgs_per_lib>."""
    w = BamWriter(path, [(contig, contig_len)], n_libs, rgs_per_lib, block_bytes, level)
    w.add(0, arrs)
    w.close()


def write_fasta(path, contigs_with_ref, width=60):
    """FASTA + .fai of [(name, uint8 array)]"""
    fai = []
    with open(path, "wb") as f:
        off = 0
        for name, ref in contigs_with_ref:
            head = b">" + name.encode() + b"\n"
            f.write(head); off += len(head)
            n = len(ref); rows = (n + width - 1) // width
            pad = np.full(rows * width, 10, np.uint8); pad[:n] = ref
            body = np.concatenate([pad.reshape(rows, width), np.full((rows, 1), 10, np.uint8)], axis=1).reshape(-1)
            # (the last line is short: drop its padding newlines but one)
            last = n - (rows - 1) * width
            body = body[:(rows - 1) * (width + 1) + last + 1].copy(); body[-1] = 10
            f.write(body.tobytes())
            fai.append("%s\t%d\t%d\t%d\t%d\n" % (name, n, off, width, width + 1))
            off += len(body)
    open(path + ".fai", "w").write("".join(fai))


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


# the BASELINE.json configurations
CONFIGS = {
    # config 3: 30x, 150 bp, one contig
    "wgs30x": dict(depth=30.0, read_len=150, n_libs=1, p_sub=0.005, p_clip=0.05, p_ins=0.01, p_del=0.01, indel_max=3),
    # config 5: 200x tumour, 4 libraries, 10 % of reads carry one I or D of length U[1,10]
    "tumor200x": dict(depth=200.0, read_len=150, n_libs=4, p_sub=0.005, p_clip=0.05, p_ins=0.05, p_del=0.05, indel_max=10),
    # config 3's model with mixed read lengths (adapter-trimmed reads, two run types in one file): 30 % of the reads trimmed to
    # U[100, 149] bases, 10 % are 250 bases long, the rest 150; the number of reads keeps the depth at 30x
    # long reads (PacBio-HiFi-like lengths; the generator's indel model stays per read): every piece is above the 16-bit packing limit
    # and takes the PF_HUGE path — a functional check and a throughput point, not one of BASELINE's configurations
    "long10k": dict(depth=30.0, read_len=10000, n_libs=1, p_sub=0.005, p_clip=0.05, p_ins=0.3, p_del=0.3, indel_max=3),
    "wgs30x_mixed": dict(depth=30.0, read_len=150, n_libs=1, p_sub=0.005, p_clip=0.05, p_ins=0.01, p_del=0.01, indel_max=3,
                         p_trim=0.3, p_long=0.1, trim_min=100, long_len=250),
    # NovaSeq-like short reads (round 6): 151 bases, the four RTA3 quality bins {2, 12, 23, 37}, a quarter of the reads adapter-trimmed to
    # U[35, 150] bases, 10 % soft-clipped, 8 % duplicates / 1 % secondary / 1 % supplementary records, 5 % MAPQ-0 multimappers: the shape of
    # real Illumina data that the one-modal-length fast path of k_pileup2 meets (uniform 150-base reads are its best case)
    "novaseq": dict(depth=30.0, read_len=151, n_libs=1, p_sub=0.005, p_clip=0.10, p_ins=0.01, p_del=0.01, indel_max=3,
                    p_trim=0.25, p_long=0.0, trim_min=35, long_len=0, qual_bins=1, p_dup=0.08, p_sec=0.01, p_supp=0.01, p_mapq0=0.05),
}


def _hifi_qualities(a):
    """BRC_SYNTH_QUAL_SHIFT=n (measurements only): every base quality raised by n, capped at 93 — PacBio-HiFi-like values, above the 62 an
    event byte holds: every read then has a row in the wide stream and every lane takes the escape path."""
    sh = int(os.environ.get("BRC_SYNTH_QUAL_SHIFT", "0") or 0)
    if sh:
        a["qual"] = np.minimum(a["qual"].astype(np.int32) + sh, 93).astype(np.uint8)
    return a


def generate(contig_len, config="wgs30x", seed=1, n_chunks=64):
    """Returns (ref uint8[contig_len], arrays dict in brc_read_batch layout)."""
    cfg = CONFIGS[config]
    L = cfg["read_len"]
    p_trim, p_long = cfg.get("p_trim", 0.0), cfg.get("p_long", 0.0)
    trim_min, long_len = cfg.get("trim_min", 0), cfg.get("long_len", 0)
    mean_len = L * (1.0 - p_trim - p_long) + p_trim * (trim_min + L - 1) / 2.0 + p_long * long_len
    n = int(round(contig_len * cfg["depth"] / mean_len))
    L = max(L, long_len if p_long > 0 else 0)           # row stride of the arenas
    ref = np.empty(contig_len, np.uint8)
    lib().synth_ref(ref.ctypes.data_as(C.c_void_p), C.c_int64(contig_len), C.c_uint64(seed))
    a = dict(pos=np.empty(n, np.int32), flag=np.empty(n, np.uint16), mapq=np.empty(n, np.uint8), lib=np.empty(n, np.int16),
             l_qseq=np.empty(n, np.int32), n_cigar=np.empty(n, np.uint32), cigar_off=np.empty(n, np.uint64),
             seq_off=np.empty(n, np.uint64), qual_off=np.empty(n, np.uint64), nm=np.empty(n, np.int32), sm=np.empty(n, np.int32),
             tags=np.empty(n, np.uint8), cigar=np.empty(3 * n, np.uint32), seq4=np.empty(n * ((L + 1) // 2), np.uint8),
             qual=np.empty(n * L, np.uint8))
    p = Params(contig_len, n, cfg["read_len"], cfg["n_libs"], seed + 1, cfg["p_sub"], cfg["p_clip"], cfg["p_ins"], cfg["p_del"], cfg["indel_max"], n_chunks,
               p_trim, p_long, trim_min, long_len, cfg.get("qual_bins", 0), cfg.get("p_dup", 0.0), cfg.get("p_sec", 0.0), cfg.get("p_supp", 0.0), cfg.get("p_mapq0", 0.0), 0)
    order = ["pos", "flag", "mapq", "lib", "l_qseq", "n_cigar", "cigar_off", "seq_off", "qual_off", "nm", "sm", "tags", "cigar", "seq4", "qual"]
    rc = lib().synth_reads(C.byref(p), ref.ctypes.data_as(C.c_void_p), *[a[k].ctypes.data_as(C.c_void_p) for k in order])
    if rc != 0:
        raise RuntimeError("synth_reads failed: %d" % rc)
    return ref, _hifi_qualities(a)


# reads with an operator every ~15 bases (ONT / CLR-like alignments): 3-10 kb, 30x — the regime the tile compaction exists for
DENSE = {"ont": dict(depth=30.0, len_min=3000, len_max=10000, op_gap=15.0, indel_max=3, p_sub=0.01, n_libs=1),
         # ultra-long reads: 30-100 kb, 2 000-6 600 M operators each (the wave-form annotator's one-wave-per-workgroup and one-wave-per-CU instantiations)
         "ont_ul": dict(depth=30.0, len_min=30000, len_max=100000, op_gap=15.0, indel_max=3, p_sub=0.01, n_libs=1),
         # HiFi reads aligned with --eqx (pbmm2, minimap2 --eqx; round 6): 10-20 kb, an insertion or a deletion every ~150 bases, a substitution
         # every ~300, the match runs written as = and X — no M operator at all: fetch_func compares nothing (bamreadcount.cpp:133-197)
         "hifi_eqx": dict(depth=30.0, len_min=10000, len_max=20000, op_gap=150.0, indel_max=3, p_sub=0.0033, n_libs=1, eqx=1)}


def generate_dense(contig_len, config="ont", seed=1, n_chunks=64):
    """(ref, arrays) like generate(), from tools/synth_gen.c: synth_reads_dense; CIGARs packed (cigar_off = running sum of n_cigar)."""
    cfg = DENSE[config]
    lmin, lmax = cfg["len_min"], cfg["len_max"]
    n = int(round(contig_len * cfg["depth"] / ((lmin + lmax) / 2.0)))
    stride = int(2.5 * lmax / cfg["op_gap"]) + 16 + (int(4.0 * cfg["p_sub"] * lmax) + 64 if cfg.get("eqx") else 0)
    ref = np.empty(contig_len, np.uint8)
    lib().synth_ref(ref.ctypes.data_as(C.c_void_p), C.c_int64(contig_len), C.c_uint64(seed))
    a = dict(pos=np.empty(n, np.int32), flag=np.empty(n, np.uint16), mapq=np.empty(n, np.uint8), lib=np.empty(n, np.int16),
             l_qseq=np.empty(n, np.int32), n_cigar=np.empty(n, np.uint32), cigar_off=np.empty(n, np.uint64),
             seq_off=np.empty(n, np.uint64), qual_off=np.empty(n, np.uint64), nm=np.empty(n, np.int32), sm=np.empty(n, np.int32),
             tags=np.empty(n, np.uint8), cigar=np.zeros(n * stride, np.uint32), seq4=np.empty(n * ((lmax + 1) // 2), np.uint8),
             qual=np.empty(n * lmax, np.uint8))
    p = Params(contig_len, n, lmax, cfg["n_libs"], seed + 1, cfg["p_sub"], 0.0, 0.0, 0.0, cfg["indel_max"], n_chunks, 0.0, 0.0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, cfg.get("eqx", 0))
    order = ["pos", "flag", "mapq", "lib", "l_qseq", "n_cigar", "cigar_off", "seq_off", "qual_off", "nm", "sm", "tags", "cigar", "seq4", "qual"]
    L = lib(); L.synth_reads_dense.argtypes = None
    rc = L.synth_reads_dense(C.byref(p), C.c_int32(lmin), C.c_int32(lmax), C.c_double(cfg["op_gap"]), C.c_int32(stride), ref.ctypes.data_as(C.c_void_p),
                             *[a[k].ctypes.data_as(C.c_void_p) for k in order])
    if rc != 0:
        raise RuntimeError("synth_reads_dense failed: %d" % rc)
    # pack the CIGARs: row i keeps its first n_cigar[i] operators
    nc = a["n_cigar"].astype(np.int64)
    off = np.concatenate([[0], np.cumsum(nc)])
    idx = np.repeat(np.arange(n, dtype=np.int64) * stride, nc) + (np.arange(int(off[-1]), dtype=np.int64) - np.repeat(off[:-1], nc))
    a["cigar"] = a["cigar"][idx].copy(); a["cigar_off"] = off[:-1].astype(np.uint64)
    return ref, _hifi_qualities(a)


def algorithmic_bytes(arrs, n_positions, n_libs_printed, n_indel_buckets=0, ref_positions=None):
    """B_in + B_ref + B_out (compulsory HBM traffic of the path, implementation independent)."""
    L = arrs["l_qseq"].astype(np.int64)
    b_in = int((32 + 4 * arrs["n_cigar"].astype(np.int64) + (L + 1) // 2 + L).sum())
    b_ref = int(ref_positions if ref_positions is not None else n_positions)
    b_out = 312 * int(n_positions) * int(n_libs_printed) + 52 * int(n_indel_buckets)
    return b_in, b_ref, b_out
