// brc_core.h — per-lane algorithms of the MI355X pileup/readcount pipeline.
//
// Everything here is written once as __host__ __device__ code: the HIP kernels in brc_engine.hip are thin
// wave/grid wrappers around these functions, and tests/sim/ runs the very same functions lane-by-lane on the
// CPU so the device algorithm can be debugged without a GPU.  (The simulator is test infrastructure; the
// product library contains only the HIP kernels.)
//
// Work decomposition:
//   K1  annotate_read     one lane per read      : fetch_func's five "Zm" integers + per-read constants
//                                                   (reference bamreadcount.cpp:114-256), then the read's PIECES
//                                                   (walk_pieces: one per M/=/X segment)
//   K1' enumerate_indels  one lane per read      : the (position, qpos, length) of every indel event the
//                                                   pileup would see for this read (htslib resolve_cigar2 peek)
//   KB  k_pileup2         one lane per position  : wave-uniform walk over the pieces covering the lane's 64-position
//                                                   tile, in stream order = pileup column order; BasicStat::process_read
//                                                   per event into two register-resident buckets with packed integers
//                                                   (BasicStat.cpp:28-107, bamreadcount.cpp:276-348)
//   KI  reduce_indel_bucket one lane per (tile,lib): the bucket's indel events sorted by (position, library, read), every
//                                                   key's run folded in column order into per-allele BasicStats
//                                                   (bamreadcount.cpp:315-342)
#ifndef BRC_CORE_H
#define BRC_CORE_H

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define BRC_HD __host__ __device__ __forceinline__
#else
#define BRC_HD static inline __attribute__((always_inline))
#endif
#if defined(__HIPCC__)
#define BRC_HDM __host__ __device__ __forceinline__        // (member functions)
#else
#define BRC_HDM inline __attribute__((always_inline))
#endif

namespace brc {

// BAM constants (SAMv1 4.2)
enum { CMATCH = 0, CINS, CDEL, CREF_SKIP, CSOFT_CLIP, CHARD_CLIP, CPAD, CEQUAL, CDIFF };
enum { FPROPER_PAIR = 2, FUNMAP = 4, FREVERSE = 16, FSECONDARY = 256, FQCFAIL = 512, FDUP = 1024 };
// engine-private flag bit set on the host for reads bam_plp_push would drop because of -d/max-count
enum { FHOSTDROP = 0x8000 };
// Reads that never enter a pileup column: htslib 1.10's bam_plp_push skips UNMAPPED reads only ("any additional filtering
// must be done in iter->func") + the host's max-count drops.
#define BRC_PUSH_MASK (brc::FUNMAP | brc::FHOSTDROP)
// Reads that sit in the column (they create their library's entry, can abandon a -p position and make a position print,
// bamreadcount.cpp:276-286) but that pileup_func never counts (:295-310).
#define BRC_NOCOUNT_MASK (brc::FSECONDARY | brc::FQCFAIL | brc::FDUP)

enum { NBUCKET = 6, NI = 9, NF = 4 };
// integer plane order == brc.h BRC_I_*; float plane order == BRC_F_*
enum { I_N = 0, I_SMQ, I_SSE, I_PLUS, I_MINUS, I_NQ2, I_SMMQ, I_SCLIP, I_SBQ };
enum { F_SEV = 0, F_SQ2, F_SNM, F_S3P };
// register accumulators per bucket: the 8 stored integers (read_count = plus + minus) + 4 floats
enum { A_SMQ = 0, A_SSE, A_PLUS, A_MINUS, A_NQ2, A_SMMQ, A_SCLIP, A_SBQ, NACC_I };

static const uint32_t NONE32 = 0xFFFFFFFFu;
static const int TILE = 64;   // positions per wave = lanes per wavefront on gfx950
// brc_region_windows: a tile's entry in the wanted table — TILE_UNWANTED, or first lane | last lane << 8 of what the windows ask for
enum { TILE_UNWANTED = 0xffff };
BRC_HD bool tile_wants(uint32_t w, uint32_t lane) { return w != (uint32_t)TILE_UNWANTED && lane >= (w & 0xffu) && lane <= (w >> 8); }

struct DevCfg {
    int32_t min_mapq, min_bq, per_lib, insertion_centric, Lp, ref_len_check, has_ref;
    int32_t beg0, end;      // reporting window [beg0,end)
    int32_t pos0;           // reference position of plane index 0
    int64_t P;              // plane length (positions)
    int64_t PS;             // plane stride in elements: P rounded up to 64 so every wave's plane store is one aligned 256-B segment
    int64_t ref_lo, ref_hi; // the device reference slice holds contig positions [ref_lo, ref_hi)
    int64_t ref_len;        // contig length (positions >= ref_len read as NUL)
    int64_t n_reads;
    int32_t table_len;      // L0: modal read length of the region (host); reads with l_qseq == clipped == L0 take their terms from tables
    // (the layout of the packed sums, choose_pack, lives in the two slots the experiment builds' ablation knobs use: the structure — the first
    // kernel argument of both big kernels — keeps its size and the offsets of everything behind it; a structure 8 bytes longer cost
    // k_pileup2 1.5 % on every shape, tools/experiments/README.md)
#ifdef BRC_EXP_KNOBS
    int32_t variant;        // experiment builds (-DBRC_EXP_KNOBS): BRC_PILEUP_VARIANT (see brc_engine.hip)
    int32_t pack_shift;
#else
    int32_t pack_shift;     // bits of the narrow packed fields: 16 (reads up to 5461 bases) or 12 (longer reads: 20 bits for the wide fields) — choose_pack
#endif
    int32_t ibucket_shift;  // log2 of the positions per indel bucket (indel_bucket_shift)
#ifdef BRC_EXP_KNOBS
    int32_t ann_variant;    // experiment builds: BRC_ANN_VARIANT (ablations of K1: wrong results, timing only)
    uint32_t pack_lim_lo;
#else
    uint32_t pack_lim_lo;   // largest per-read value of a NARROW packed field (mapping quality, single-ended mapping quality): (2^pack_shift - 1) / max(K, HALF), never below 255
#endif
    int32_t force_dom;      // test knob (BRC_FORCE_DOM): -1, or the bucket every lane treats as dominant (stresses the alternate / third-allele paths)
    int64_t n_pieces;       // pieces of all libraries (KB v2)
    int32_t flush_k;        // K: pieces a lane may accumulate in its packed integer registers between two flushes (1..127)
    uint32_t pack_lim;      // largest per-read value of a WIDE packed field (mismatch-quality sum, clipped length): (2^(32 - pack_shift) - 1) / max(K, HALF); PF_HUGE above it
    int32_t max_lqseq;      // longest read of the batch (k_pileup2: a staged window further from its row than that belongs to a piece that only SPANS the tile)
#ifdef BRC_CHECKED
    void* chk;              // (bounds-checked fuzzing build only) the engine's ChkState in device memory
#endif
};

// ---------------------------------------------------------------- the bounds-checked instantiation (-DBRC_CHECKED)
//
// csrc/Makefile builds libbrc_hip_checked.so from the same sources with -DBRC_CHECKED (a fuzzing build: never shipped, never
// loaded by the product's callers).  In it every data-dependent device address of K1 and of k_pileup2 — the staged event-byte
// windows, the scalar record loads two pieces ahead, rare records, wide-stream words, QUAL / SEQ / reference-code / CIGAR
// windows, the event-byte and wide rows K1 writes, piece / key slots, plane, queue and list stores — is compared with the
// extent its buffer was ALLOCATED FOR (the bytes the host asked for, not the rounded-up capacity).  A violation is counted,
// the first one is recorded {kernel, site, buffer, address, bytes, unit = tile or read, piece}, and the access is redirected to
// the buffer's first bytes; the host reads the record back after every pass and fails the call with it.  (No s_trap: a trapped
// queue takes the record with it, and on a shared pool the box.)  Everywhere else BRC_CK is the pointer itself.
enum ChkBuf { CB_CIGAR = 0, CB_SEQ, CB_QUAL, CB_REFCODE, CB_EB, CB_BQW, CB_PIECES, CB_RARE, CB_KEYREACH, CB_READS, CB_EVRAW, CB_CNT, CB_WANTED, CB_RNG,
              CB_UNAVAIL, CB_TILELIST, CB_NCOL, CB_DEPTH, CB_SLOTID, CB_SI, CB_SF, CB_XEV, CB_XEVN, CB_TILECTR, CB_LDS_ROWS, CB_LDS_QUEUE,
              CB_KP_PIECES, CB_KP_RARE, CB_KP_RNG /* what k_pileup2 reads: K1's stream and ranges, or the compacted ones */, CB_N };
enum ChkKernel { CK_ANNOTATE = 1, CK_PILEUP = 2 };
struct ChkExt { uint64_t lo, hi; };
struct ChkState { ChkExt ext[CB_N]; uint32_t count, kernel, site, buf; uint64_t addr, bytes; int64_t unit, piece; };
#if defined(BRC_CHECKED) && defined(__HIPCC__)
__host__ __device__ __forceinline__ uint64_t chk_range(void* state, uint32_t kernel, uint32_t site, uint32_t buf, uint64_t addr, uint64_t bytes, int64_t unit, int64_t piece) {
#if defined(__HIP_DEVICE_COMPILE__)
    ChkState* st = (ChkState*)state;
    const uint64_t lo = st->ext[buf].lo, hi = st->ext[buf].hi;
    if (addr >= lo && addr + bytes <= hi && addr + bytes >= addr) return addr;
    if (atomicAdd(&st->count, 1u) == 0u) { st->kernel = kernel; st->site = site; st->buf = buf; st->addr = addr; st->bytes = bytes; st->unit = unit; st->piece = piece; }
    return lo;
#else
    (void)state; (void)kernel; (void)site; (void)buf; (void)bytes; (void)unit; (void)piece;
    return addr;
#endif
}
// pointer form: p itself when [p, p + bytes) lies inside buffer BUF, the buffer's start otherwise (recorded)
// (BRC_CHK_MASK_A / _P: compile-time masks of the sites of K1 / k_pileup2 that are checked — all of them unless a debugging build narrows them)
#ifndef BRC_CHK_MASK_A
#define BRC_CHK_MASK_A 0xffffffffffffffffull
#endif
#ifndef BRC_CHK_MASK_P
#define BRC_CHK_MASK_P 0xffffffffffffffffull
#endif
#define BRC_CHK_ON(K, SITE) (((((K) == brc::CK_ANNOTATE) ? (BRC_CHK_MASK_A) : (BRC_CHK_MASK_P)) >> (SITE)) & 1ull)
#define BRC_CK(c, K, SITE, BUF, p, bytes, unit, piece) (BRC_CHK_ON(K, SITE) ? (decltype((p) + 0))brc::chk_range((c).chk, (K), (SITE), (BUF), (uint64_t)(p), (uint64_t)(bytes), (int64_t)(unit), (int64_t)(piece)) : (decltype((p) + 0))(p))
// index form (LDS arrays, counters): idx itself when idx < n, 0 otherwise (recorded with the index as the address)
#define BRC_CKI(c, K, SITE, BUF, idx, n, unit, piece) ((!BRC_CHK_ON(K, SITE) || (uint64_t)(idx) < (uint64_t)(n)) ? (idx) : (brc::chk_fail_index((c).chk, (K), (SITE), (BUF), (uint64_t)(idx), (uint64_t)(n), (int64_t)(unit), (int64_t)(piece)), (decltype((idx) + 0))0))
__host__ __device__ __forceinline__ void chk_fail_index(void* state, uint32_t kernel, uint32_t site, uint32_t buf, uint64_t idx, uint64_t n, int64_t unit, int64_t piece) {
#if defined(__HIP_DEVICE_COMPILE__)
    ChkState* st = (ChkState*)state;
    if (atomicAdd(&st->count, 1u) == 0u) { st->kernel = kernel; st->site = site; st->buf = buf; st->addr = idx; st->bytes = n; st->unit = unit; st->piece = piece; }
#else
    (void)state; (void)kernel; (void)site; (void)buf; (void)idx; (void)n; (void)unit; (void)piece;
#endif
}
#else
#define BRC_CK(c, K, SITE, BUF, p, bytes, unit, piece) (p)
#define BRC_CKI(c, K, SITE, BUF, idx, n, unit, piece) (idx)
#endif

// Region inputs exactly as brc_read_batch lays them out (uploaded as-is; offsets rebased per region).
struct DevIn {
    const int32_t* pos; const uint16_t* flag; const uint8_t* mapq; const int16_t* lib; const int32_t* l_qseq;
    const uint32_t* n_cigar; const uint64_t* cig_off; const uint64_t* seq_off; const uint64_t* qual_off;
    const int32_t* nm; const int32_t* sm; const uint8_t* tags;
    const uint32_t* cigar; const uint8_t* seq4; const uint8_t* qual; const char* ref;
    // device-produced by K1: ONE BYTE per base, the "event byte" (see eb_make below): quality << 2 | base index (A C G T = 0..3)
    // — `byte >= min_bq << 2` is the base-quality test of bamreadcount.cpp:288 without unpacking, `byte & 3` against the lane's
    // reference base says "the reference base's bucket".  Read i's row starts at element bq_row[i] (rows are padded to multiples
    // of 16 elements so every row is 16-byte aligned: KB copies row windows into LDS 16 bytes at a time).  The few bases a byte
    // cannot describe (quality 0 or above 62, an N or '=' base) carry an ESCAPE byte that still answers the quality test, and
    // their full word quality << 8 | bucket("=ACGTN") is in the WIDE stream `bqw` (written for those 8-base groups only); the
    // pieces of such a read are marked PF_WIDE.  The wide stream is SPARSE: a row (as long as the read's byte row) exists only
    // for the reads the host found an escape base in when they were pushed (Staged.wide: eb_make's predicate, which asks
    // nothing of the reference or of -b).  The buffer `bqw` starts with a table of one u32 per 16 elements of the byte stream:
    // the entry of the chunk a read's byte row starts with holds where its wide row starts, in units of 16 elements from
    // `bqw` itself (wide_base below: one pointer serves the table and the rows).  Entries of reads without a wide row are
    // never read.
    const uint8_t* eb;
    const uint16_t* bqw;
    const uint64_t* bq_row;  // [n_reads] host-computed prefix sums of roundup16(l_qseq)
    const uint32_t* iev_off; // [n_reads] host-computed: first slot of the read in the raw indel-event list (one slot per I/D/P operator)
    const struct RcpPair* rcp;   // [n_reads], device-produced by K1
};

// per-read float constants written by K1 next to each DRead (one 16-byte scalar load in KB): correctly rounded
// reciprocals 1/(float)l_qseq and 1/((float)clipped_length/2), and the two denominators themselves
struct RcpPair { float rcpL, rcpC, Lf, center; };

// first element of the wide row of the read whose event-byte row starts at element `row` (DevIn.bqw)
BRC_HD uint64_t wide_base(const uint16_t* bqw, uint64_t row) { return (uint64_t)reinterpret_cast<const uint32_t*>(bqw)[row >> 4] << 4; }

// Packed per-read record written by K1 and read with ONE scalar load (s_load_dwordx16) by KB: 64 bytes.
enum { M_REV = 1, M_Q2OK = 2, M_SMW = 4, M_NMW = 8, M_SIMPLE = 16,
       M_CLIPM = 32,    // CIGAR is one M with soft clips around it ([S] M [S]): qpos = p - pos + left, no CIGAR walk in KB
       M_STAGED = 64,   // KB can stage this read's bq window in LDS: simple CIGAR, or deleted + inserted + clipped bases <= 24
       M_FAST = 128 };  // SIMPLE && STAGED && library known && l_qseq == clipped_length == DevCfg.table_len: KB's branch-free
                        // probe applies and the event terms come from the quotient tables (finish_misc)
static const uint32_t M_NOCOUNT = 0x80000000u;   // SECONDARY / QCFAIL / DUP: in the column, never counted (BRC_NOCOUNT_MASK)
// misc layout: bits 0-7 flags | 8-15 mapq | 16-23 library index + 1 (0 = unavailable) | 24-30 total D/N bases (staged reads) | 31 M_NOCOUNT
enum { STAGE_SLACK = 24 };
struct alignas(64) DRead {
    int32_t pos, end;          // [pos,end) on the reference; end == pos when the read never enters a column
    uint32_t cig_off, n_cigar;
    uint64_t bq_off;           // index of the read's first base in bq[] (multiple of 8: 16-byte aligned row)
    uint32_t misc;             // see the misc layout above
    int32_t l_qseq;
    int32_t q2, tp, left, clipped;   // Zm: q2_pos, three_prime_index, left_clip, clipped_length
    uint32_t zm_sum, sse_add;        // Zm sum_of_mismatch_qualities; per-event addend of sum_single_ended_map_qualities
    float snm_add;                   // per-event addend of sum_number_of_mismatches: NM / (float)clipped_length
    int32_t clipped_dup;             // == clipped: the accumulate stage reads {zm_sum, sse_add, snm_add, clipped} as ONE 16-byte word
};

// table: l_qseq == clipped_length == DevCfg.table_len;  clipm: the CIGAR is [S] M [S] and left_clip is the leading clip
BRC_HD uint32_t finish_misc(uint32_t misc, bool table, bool clipm) {
    const uint32_t need = M_SIMPLE | M_STAGED;
    if (clipm && (misc & M_STAGED)) misc |= M_CLIPM;
    if (table && (misc & need) == need && ((misc >> 16) & 0xffu) != 0u) misc |= M_FAST;
    return misc;
}
// shape of a CIGAR for M_CLIPM, fed operator by operator
struct CigShape { int n_m = 0, n_other = 0, lead_s = 0; };
BRC_HD void shape_add(CigShape& s, uint32_t op, int len) {
    if (op == 0u /* CMATCH */) ++s.n_m;
    else if (op == 4u /* CSOFT_CLIP */) { if (!s.n_m) s.lead_s += len; }
    else ++s.n_other;
}
BRC_HD bool shape_clipm(const CigShape& s, uint32_t nc, int left_clip) { return nc >= 2u && s.n_m == 1 && s.n_other == 0 && s.lead_s == left_clip; }

// One indel event, produced by the per-read enumeration, consumed by the per-key reduction (16 bytes).
struct IndelEv { uint32_t read; int32_t qpos; int32_t len; uint32_t key_lo; };
// One reduced indel bucket.
struct IndelOut { int32_t pos, lib, len; uint32_t rep_read; int32_t rep_qpos; uint32_t i[NI]; float f[NF]; };

// Device result of KB, compact: a position keeps the BasicStats of TWO buckets per library — slot 0 = the bucket of its
// reference base ("dominant"), slot 1 = the first other base seen ("alternate") — instead of all six (four of the six
// are all-zero at practically every position: 116 instead of 320 bytes per position and library leave the GPU).  Events
// of a third, fourth, ... base (sequencing errors: well under 1 % of the positions) are appended to a list as raw
// addends and folded in by the host when it expands the slots to the ABI's dense planes (expand_slots, brc_host.cpp),
// in list order = pileup-column order, so the fp32 sums stay bit-exact.
enum { XEV_CTR_STRIDE = 16 };   // words between the cursors of two sub-lists (64 bytes)
struct XEv {                // one third-allele event (48 bytes)
    uint32_t k;             // plane index of the position
    uint32_t lib_b;         // library << 8 | bucket
    uint32_t mapq, sse, zm, clip;
    uint32_t qf;            // base quality | reverse << 8 | q2ok << 9
    float fq2, fs3p, fsnm;
    double sev;             // the event-location term, added through double (BasicStat.cpp:70)
};
struct Planes {
    uint32_t* ncol;     // [Lp][PS]
    uint32_t* depth;    // [Lp][PS]
    uint32_t* slotid;   // [Lp][PS]  dominant bucket | alternate bucket << 8 (NB_NONE: slot 1 unused)
    uint32_t* si;       // [Lp][2][9][PS]
    float* sf;          // [Lp][2][4][PS]
    uint32_t* unavail;  // [PS]
    XEv* xev;           // third-allele events: xev_shards sub-lists of xev_cap entries each, one atomic cursor per sub-list
    uint32_t* xev_n;    // cursors, XEV_CTR_STRIDE words apart (a single cursor serialises every wave of the launch on one
                        // L2 atomic unit: ~12 ns per append); entries past xev_cap are dropped — the host sees a cursor
                        // above xev_cap, grows the lists and computes again
    uint32_t xev_cap;
    uint32_t xev_shards;   // power of two (1 in the CPU simulator)
};

// Third-allele events folded ON THE DEVICE (round 6; until then the host added them up): one record per (position, library, bucket) that
// has such events, its 13 sums accumulated in list order = pileup-column order (every event of a position sits in ONE sub-list, in append
// order, and the compaction keeps that order) — BasicStat::process_read's work (BasicStat.cpp:28-107) for the few events that found both of
// their position's slots taken.  The records of one (64-position tile, library) lie together, sorted by (position, bucket): a lane that
// writes a position's line, and the host that expands the slots to the ABI's dense planes, find a bucket's sums in ONE place — the slot that
// names the bucket, or this table.
struct alignas(16) XAgg { uint32_t k; uint32_t lib_b; uint32_t i[NI]; float f[NF]; uint32_t pad; };     // 64 bytes
static_assert(sizeof(XAgg) == 64, "XAgg is copied to the host as it is");
// fold of ONE (tile, library) bucket: idx[0..n) = the bucket's events as indices into the compacted list `list`, in any order;
// sorted here by (position, bucket, index) — insertion sort: a bucket holds a handful —, every run folded in index order into
// out[0..).  Returns the number of records.
BRC_HD int fold_xev_bucket(const XEv* list, uint32_t* idx, int n, XAgg* out) {
    auto key = [&](uint32_t i) -> uint64_t { const XEv& e = list[i]; return ((uint64_t)(e.k & 63u) << 40) | ((uint64_t)(e.lib_b & 0xffu) << 32) | (uint64_t)i; };      // (one tile: the position's lane orders it)
    for (int a = 1; a < n; ++a) { const uint32_t t = idx[a]; const uint64_t kt = key(t); int b = a - 1; while (b >= 0 && key(idx[b]) > kt) { idx[b + 1] = idx[b]; --b; } idx[b + 1] = t; }
    int no = 0;
    for (int a = 0; a < n; ++a) {
        const XEv& e = list[idx[a]];
        if (no == 0 || out[no - 1].k != e.k || out[no - 1].lib_b != e.lib_b) {
            XAgg& o = out[no++]; o.k = e.k; o.lib_b = e.lib_b; o.pad = 0u;
            for (int f = 0; f < NI; ++f) o.i[f] = 0u;
            for (int f = 0; f < NF; ++f) o.f[f] = 0.0f;
        }
        XAgg& o = out[no - 1];
        const uint32_t rev = (e.qf >> 8) & 1u;
        o.i[I_N] += 1u; o.i[I_SMQ] += e.mapq; o.i[I_SSE] += e.sse; o.i[I_PLUS] += 1u - rev; o.i[I_MINUS] += rev;
        o.i[I_NQ2] += (e.qf >> 9) & 1u; o.i[I_SMMQ] += e.zm; o.i[I_SCLIP] += e.clip; o.i[I_SBQ] += e.qf & 0xffu;
        o.f[F_SQ2] += e.fq2; o.f[F_S3P] += e.fs3p; o.f[F_SNM] += e.fsnm;
        o.f[F_SEV] = (float)((double)o.f[F_SEV] + e.sev);                    // through double, like BasicStat.cpp:69-70
    }
    return no;
}

// ---------------------------------------------------------------- small tables as packed constants

// htslib seq_nt16_table (IUPAC char -> 4-bit code; '=' 0; '0'..'3' 1,2,4,8; everything else 15), bamreadcount.cpp:149
BRC_HD uint32_t nt16_of_char(uint32_t c) {
    uint32_t l = (c | 0x20u) - 'a';
    if (l < 26u) {
        // a b c d e f g h i j k l m n o p | q r s t u v w x y z
        const uint64_t lo = 0xFFF3FCFFB4FFD2E1ull;  // nibbles a..p (a in the lowest nibble)
        const uint64_t hi = 0x000000FAF97F865Full;  // nibbles q..z
        return (uint32_t)(((l < 16u) ? (lo >> (l * 4)) : (hi >> ((l - 16u) * 4))) & 15u);
    }
    if (c == '=') return 0;
    if (c - '0' < 4u) return 1u << (c - '0');
    return 15;
}
// bam_nt16_canonical_table (bamreadcount.cpp:36-39): 4-bit base code -> bucket index in "=ACGTN"
BRC_HD uint32_t canon_bucket(uint32_t b4) { return (uint32_t)((0x5555555455535210ull >> (b4 * 4)) & 15u); }

BRC_HD uint32_t seqi(const uint8_t* s, int64_t i) { return (s[i >> 1] >> ((~i & 1) << 2)) & 0xfu; }
BRC_HD bool is_refop(uint32_t op) { return op == CMATCH || op == CDEL || op == CREF_SKIP || op == CEQUAL || op == CDIFF; }
BRC_HD bool is_mop(uint32_t op) { return op == CMATCH || op == CEQUAL || op == CDIFF; }
BRC_HD int iabs(int x) { return x < 0 ? -x : x; }

BRC_HD uint32_t ref_at(const DevCfg& c, const char* ref, int64_t p) {
    return (p >= 0 && p < c.ref_len && p >= c.ref_lo && p < c.ref_hi) ? (uint32_t)(uint8_t)ref[p - c.ref_lo] : 0u;
}

// ---------------------------------------------------------------- the event byte
//
//   byte = q << 2 | i         1 <= q <= 62: the base quality;  i = 0..3: the base is A C G T (bucket i + 1)
//   byte = 63 << 2 / 0        ESCAPE (quality 0 or >= 63, or an N / '=' base): 63 << 2 when the base passes the base-quality
//                             filter (q >= -b), 0 when it does not — so `byte >= piece_thr` is the filter for every byte; the
//                             truth is in DevIn.bqw
// Nothing in a byte depends on the reference: K1 writes it for every base alike (the indel side path compares inserted bases
// through the same bytes, same_allele).  A lane compares `byte & 3` with the index of its reference base (dom_index; 4 and 5
// for a reference base that is not A C G T: no byte matches).  Events of an N / '=' base are rare and never enter a lane's two
// slots: they go to the third-allele list whatever the slots hold ("exotic").
// The accumulators add event VALUES q << 2 | i with the full quality (up to 255): every event of a slot has the slot's index,
// so the slot's base-quality sum is (sum - count * i) >> 2 — no unpacking per event.
enum { EB_ESC = 63 };
BRC_HD bool bucket_acgt(uint32_t b) { return b - 1u < 4u; }
BRC_HD uint32_t eb_index(uint32_t b) { return bucket_acgt(b) ? b - 1u : 0u; }              // the index an event VALUE of bucket b carries
BRC_HD uint32_t dom_index(uint32_t dom_b) { return bucket_acgt(dom_b) ? dom_b - 1u : (dom_b == 5u ? 4u : 5u); }   // what a lane compares `byte & 3` with
BRC_HD uint32_t dom_bucket_of_index(uint32_t di) { return di < 4u ? di + 1u : (di == 4u ? 5u : 0u); }
BRC_HD bool eb_is_escape(uint32_t byte) { const uint32_t q6 = byte >> 2; return q6 == 0u || q6 == (uint32_t)EB_ESC; }
BRC_HD uint32_t eb_min_bq(int32_t min_bq) { return (uint32_t)(min_bq < 0 ? 0 : (min_bq > EB_ESC ? EB_ESC : min_bq)); }
// the byte of a base; esc: it is an escape (the caller stores the wide word of its 8-base group too)
BRC_HD uint32_t eb_make(int32_t min_bq, uint32_t q, uint32_t b, bool& esc) {
    esc = q == 0u || q >= (uint32_t)EB_ESC || !bucket_acgt(b);
    if (!esc) return (q << 2) | (b - 1u);
    return (int32_t)q >= min_bq ? ((uint32_t)EB_ESC << 2) : 0u;
}

// ---------------------------------------------------------------- K1: per-read annotation (fetch_func)

// Restates bamreadcount.cpp:114-256 for read i and packs everything KB needs into a DRead.
// `wide`: the read has a base its event byte cannot describe (its pieces are marked PF_WIDE)
BRC_HD DRead annotate_read(const DevCfg& c, const DevIn& in, int64_t i, uint8_t* eb_out, uint16_t* bqw_out, bool& wide) {
    DRead r;
    const int32_t pos = in.pos[i];
    const uint32_t flag = in.flag[i];
    const int32_t L = in.l_qseq[i];
    const uint32_t nc = in.n_cigar[i];
    const uint32_t* cig = BRC_CK(c, CK_ANNOTATE, 20, CB_CIGAR, in.cigar + in.cig_off[i], 4ull * nc, i, -1);
    const uint8_t* seq = BRC_CK(c, CK_ANNOTATE, 21, CB_SEQ, in.seq4 + in.seq_off[i], (uint64_t)((L + 1) / 2), i, -1);
    const uint8_t* qual = BRC_CK(c, CK_ANNOTATE, 22, CB_QUAL, in.qual + in.qual_off[i], (uint64_t)(L > 0 ? L : 0), i, -1);
    const uint32_t mapq = in.mapq[i];
    const uint32_t tags = in.tags[i];
    // (the rows this read writes: its event bytes, and — where an escape byte needs them — the wide words of whole 8-base groups)
    (void)BRC_CK(c, CK_ANNOTATE, 23, CB_EB, eb_out + in.bq_row[i], (uint64_t)(L > 0 ? L : 0), i, -1);

    uint32_t sum = 0;
    int left_clip = 0, clipped = L, right_clip = L;
    int last_mm_pos = -1, last_mm_qual = 0;
    int read_position = 0;
    int64_t reference_position = pos;
    int32_t rlen = 0;           // reference length: bam_cigar2rlen
    int64_t tot_d = 0, tot_is = 0;   // deleted/skipped reference bases, inserted + soft-clipped query bases
    CigShape shape;
    bool stop = false;          // ':151/:175' out-of-reference break: ends ALL CIGAR processing of the annotator
    for (uint32_t k = 0; k < nc; ++k) {
        const uint32_t op = cig[k] & 0xfu;
        const int len = (int)(cig[k] >> 4);
        if (is_refop(op)) rlen += len;
        shape_add(shape, op, len);
        if (op == CDEL || op == CREF_SKIP) tot_d += len;
        if (op == CINS || op == CSOFT_CLIP) tot_is += len;
        if (stop) continue;     // rlen (the pileup's view of the read) still needs the remaining ops
        if (op == CMATCH) {
            int j = 0;
            // a record stored without its sequence (SEQ '*', l_qseq == 0: secondary alignments of minimap2 / bwa) still has a
            // CIGAR; it is never counted (brc_push_reads accepts it only with a no-count flag), so nothing below is ever
            // used — and its base / quality bytes do not exist: walking them would read the neighbouring records' bytes or
            // past the end of the arenas.  Only the cursors move.
            if (L == 0) j = len;
            for (; j < len; ++j) {
                const int cur = read_position + j;
                const int64_t refpos = reference_position + j;
                if (c.ref_len_check && c.ref_len && refpos > c.ref_len) continue;          // :144-148
                const uint32_t rc = ref_at(c, in.ref, refpos);
                if (rc == 0) break;                                                          // :151
                const uint32_t refb = nt16_of_char(rc);
                const uint32_t rb = seqi(seq, cur);
                if (rb != refb && refb != 15u && rb != 0u) {                                 // :152
                    const int q = qual[cur];
                    if (last_mm_pos != -1) {
                        if (last_mm_pos + 1 != cur) { sum += (uint32_t)last_mm_qual; last_mm_qual = q; }
                        else if (last_mm_qual < q) last_mm_qual = q;
                    } else last_mm_qual = q;
                    last_mm_pos = cur;
                }
            }
            if (j < len) { stop = true; continue; }                                          // :175
            reference_position += len; read_position += len;
        } else if (op == CDEL || op == CREF_SKIP) {
            reference_position += len;
        } else if (op == CINS) {
            read_position += len;
        } else if (op == CSOFT_CLIP) {
            read_position += len; clipped -= len;
            if (k == 0) left_clip += len; else right_clip -= len;
        }
    }
    sum += (uint32_t)last_mm_qual;                                                           // :199

    int tp, q2 = -1, kq, inc;                                                                // :201-238
    const bool rev = (flag & FREVERSE) != 0;
    if (rev) { kq = tp = 0; inc = 1; if (tp < left_clip) tp = left_clip; }
    else { kq = tp = L - 1; inc = -1; if (tp > right_clip) tp = right_clip; }
    while (kq >= 0 && kq < L) {
        if (qual[kq] != 2) { q2 = kq - 1; break; }
        kq += inc;
    }
    if (rev) { if (tp < q2) tp = q2; }
    else { if (tp > q2 && q2 != -1) tp = q2; }

    // bam_plp_push drop rules + bam_endpos; a lone non-M operator never resolves (htslib resolve_cigar2 k == -1)
    bool dropped = (flag & BRC_PUSH_MASK) != 0;
    if (nc == 0) dropped = true;   // bam_endpos = pos + 1, but resolve_cigar2 has no operator to stand on
    if (nc == 1 && !is_mop(cig[0] & 0xfu)) dropped = true;
    r.pos = pos;
    r.end = dropped ? pos : pos + rlen;
    r.cig_off = (uint32_t)in.cig_off[i];
    r.n_cigar = nc;
    r.bq_off = in.bq_row[i];
    // per-base stream for KB: the event bytes; an escape byte brings the whole 8-base group's words into the wide stream
    wide = false;
    {
        const uint64_t row = in.bq_row[i];
        for (int j = 0; j < L; ++j) {
            bool esc;
            eb_out[row + (uint64_t)j] = (uint8_t)eb_make(c.min_bq, qual[j], canon_bucket(seqi(seq, j)), esc);
            if (esc) {
                wide = true;
                if (bqw_out) {                  // (the device's kernels pass none: k_wide_rows writes the words of every read the host found an escape in)
                    const int g0 = j & ~7;
                    uint16_t* const wrow = bqw_out + wide_base(bqw_out, row);
                    for (int t = g0; t < g0 + 8 && t < L; ++t) wrow[t] = (uint16_t)((qual[t] << 8) | canon_bucket(seqi(seq, t)));
                }
            }
        }
    }
    const int lib = c.per_lib ? (int)in.lib[i] : 0;
    uint32_t misc = (mapq << 8) | ((uint32_t)((lib + 1) & 0xff) << 16);
    if (rev) misc |= M_REV;
    if (flag & BRC_NOCOUNT_MASK) misc |= M_NOCOUNT;
    if (q2 > -1) misc |= M_Q2OK;
    if (nc == 1 && (cig[0] & 0xfu) == CMATCH) misc |= M_SIMPLE;
    if (tot_d + tot_is <= STAGE_SLACK) misc |= M_STAGED | ((uint32_t)tot_d << 24);
    uint32_t sse;
    if (flag & FPROPER_PAIR) {                                                               // BasicStat.cpp:78-91
        if (tags & 2u) sse = (uint32_t)in.sm[i]; else { sse = 0; misc |= M_SMW; }
    } else sse = mapq;
    float snm = 0.0f;
    if (tags & 1u) snm = (float)in.nm[i] / (float)clipped;                                   // BasicStat.cpp:94-97
    else misc |= M_NMW;
    const bool table = c.table_len > 0 && L == c.table_len && clipped == L;
    r.misc = finish_misc(misc, table, shape_clipm(shape, nc, left_clip)); r.l_qseq = L; r.q2 = q2; r.tp = tp; r.left = left_clip; r.clipped = clipped;
    r.zm_sum = sum; r.sse_add = sse; r.snm_add = snm; r.clipped_dup = clipped;
    return r;
}

// ---------------------------------------------------------------- BasicStat::process_read for one event

// BasicStat.cpp:28-107 with the Zm string round trip removed, split in two so that the divisions are evaluated once
// per event (event_terms) and only the adds sit under the bucket dispatch (acc_apply).
// All float arithmetic is fp32 round-to-nearest exactly where the reference's is, and the event-location sum goes
// through double exactly like `float += 1.0 - float_expr` (BasicStat.cpp:69-70).  Build with -ffp-contract=off.
struct EvTerms { float q2, s3p; double sev; };   // q2: 0.0f when the read has no Q2 position; see acc_apply

BRC_HD EvTerms event_terms(const DRead& r, int qpos) {
    EvTerms t;
    const float Lf = (float)r.l_qseq;
    t.q2 = (r.misc & M_Q2OK) ? (float)iabs(qpos - r.q2) / Lf : 0.0f;   // BasicStat.cpp:60-62
    t.s3p = (float)iabs(qpos - r.tp) / Lf;                              // :66
    const float center = (float)r.clipped * 0.5f;                       // :69  (float)clipped_length/2.0, exact
    float d = (float)(qpos - r.left) - center;
    d = d < 0.0f ? -d : d;
    t.sev = 1.0 - (double)(d / center);                                 // :70  double expression
    return t;
}

// ai = NACC_I integers, af = NF floats of ONE bucket.
BRC_HD void acc_apply(uint32_t* ai, float* af, const DRead& r, const EvTerms& t, uint32_t q, bool is_indel) {
    const uint32_t m = r.misc;
    ai[A_SMQ] += (m >> 8) & 0xffu;
    // branch-free on purpose: an if/else over two different accumulators becomes a select-of-pointers in LLVM, which
    // blocks scalar replacement of the accumulator struct and sends all 78 accumulators to scratch memory
    const uint32_t rev = m & M_REV;
    ai[A_MINUS] += rev; ai[A_PLUS] += 1u - rev;
    ai[A_SMMQ] += r.zm_sum;
    // adding +0.0f is the identity on these sums (they are never -0.0), so the two uniform conditions are folded into
    // the addends once per event (event_addends) instead of once per bucket arm
    af[F_SQ2] += t.q2; ai[A_NQ2] += (m & M_Q2OK) ? 1u : 0u;
    af[F_S3P] += t.s3p;
    ai[A_SCLIP] += (uint32_t)r.clipped;
    af[F_SEV] = (float)((double)af[F_SEV] + t.sev);
    ai[A_SSE] += r.sse_add;
    af[F_SNM] += r.snm_add;                                             // 0.0f when NM is missing (annotate_read)
    if (!is_indel) ai[A_SBQ] += q;
}

BRC_HD void acc_event(uint32_t* ai, float* af, const DRead& r, int qpos, uint32_t q, bool is_indel) {
    const EvTerms t = event_terms(r, qpos);
    acc_apply(ai, af, r, t, q, is_indel);
}

// a / b with one multiply and two FMAs, given y = RN(1/b): q = RN(a*y); r = a - q*b (exact in one FMA);
// result = RN(q + r*y).  With a correctly rounded reciprocal this IS the correctly rounded quotient (Markstein);
// tests/test_exact_division.py checks it bit-for-bit against `/` over every (numerator, denominator) the path can form.
BRC_HD float div_rcp(float a, float b, float y) {
    const float q = a * y;
    const float r = fmaf(-q, b, a);
    return fmaf(r, y, q);
}

// |a - b| for two non-negative ints
BRC_HD uint32_t absdiff_u(uint32_t a, uint32_t b) { return (a > b ? a : b) - (a < b ? a : b); }
#define BRC_ABSDIFF(a, b) ((int)absdiff_u((uint32_t)(a), (uint32_t)(b)))
// (|lane value - wave-uniform value| of unsigned operands: one v_sad_u32 on the device — LLVM does not form it from max - min)
BRC_HD uint32_t absdiff_vs(uint32_t a, uint32_t b_uniform) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t d; asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "s"(b_uniform)); return d;
#else
    return absdiff_u(a, b_uniform);
#endif
}

// Quotient tables for reads with l_qseq == clipped_length == L0 (DevCfg.table_len): every division of the event terms is
// then n / L0 with an integer 0 <= n <= L0:  |qpos-q2| / L,  |qpos-tp| / L  and  |(qpos-left) - cl/2| / (cl/2) = |2(qpos-left) - cl| / cl.
//   q[n] = (float)n / (float)L0   (fp32 division, correctly rounded: identical to what the reference computes)
//   e[n] = 1.0 - (double)q[n]     (the double-precision event-location term, BasicStat.cpp:70)
// Built once per workgroup in LDS (k_pileup) / once per region in the simulator.
enum { TABLE_MAX = 512 };
struct TermTab { const float* q; const double* e; };

// ---------------------------------------------------------------- KB helpers

enum { NB_NONE = 7 };      // "no alternate bucket yet"
#if defined(__clang__)
#define BRC_UNROLL _Pragma("unroll")
#else
#define BRC_UNROLL _Pragma("GCC unroll 16")
#endif

// dominant bucket of position p: the bucket of its reference base ('A' when there is no reference); BRC_FORCE_DOM (test
// knob) names one bucket for every position
BRC_HD uint32_t dominant_bucket(const DevCfg& c, const DevIn& in, int64_t p) {
    if (c.force_dom >= 0) return (uint32_t)c.force_dom;
    if (!c.has_ref) return 1u;
    return canon_bucket(nt16_of_char(ref_at(c, in.ref, p)));
}

// ================================================================ KB v2: pieces
//
// The pileup kernel does not walk reads, it walks PIECES: one piece = one M/=/X segment of one read, i.e. a run of
// events qpos = p - a over the reference interval [rs, rs + len).  Every CIGAR is cut into pieces once, in K1 (the host
// counts them at push time so that each read knows its slots), which leaves exactly one kind of work item in the hot
// loop — htslib's resolve_cigar2 is not evaluated per (read, position) any more:
//   * deletions / reference skips produce no events; they only keep the read in the column, so the piece in front of
//     them carries an extended column length `ext` (pieces of one read tile [pos, end) without gaps);
//   * with -i the last base before an insertion is counted in the depth but not in a base bucket (bamreadcount.cpp:312
//     vs :343): it becomes a one-base piece flagged PF_NB;
//   * a read that can never be counted (MAPQ below -q, SECONDARY/QCFAIL/DUP) is ONE event-less piece over [pos, end).
// Pieces of a read are adjacent and reads keep their file order, so every bucket still sees its events in pileup-column
// order.  In per-library mode (-p) the pieces are laid out library-major (all pieces of library 0 in file order, then
// library 1, ...): a (tile, library) wave walks only its own library's stream.
enum PieceFlag { PF_TABLE = 1,    // event terms come from the quotient tables (l_qseq == clipped == table_len, left_clip == 0, q2 in {tp, none})
       PF_Q2OK = 2, PF_NB = 4,
       PF_HUGE = 8,     // a per-read integer does not fit its packed field: w2/w3 carry only the mapping quality, the rest is added by drain_int()
       PF_WIDE = 16,    // the read has escape bytes in its row (never with PF_TABLE): lanes that meet one take quality and bucket from DevIn.bqw; an N / '=' base goes to the third-allele list
       PF_DIV = 32,     // a read of another length (<= 255 bases; q2 in {tp, none}; no PF_TABLE / PF_TABQ / PF_HUGE): the event terms are divided out in the
                        // lane from the record itself — its tp field holds three_prime_index | l_qseq << 8 | left_clip << 16 — without the rare record
       PF_REV = 64,
       PF_TABQ = 128,
       PF_SMW = 256, PF_NMW = 512 };   // (ReadConst.flags only: the record's flag byte has no room for them, its ww field carries them)  // without PF_HUGE: no PF_TABLE only because the read is soft-clipped (l_qseq == table_len, left_clip < 512, q2 in {tp, none}, no PF_HUGE): the two
                         // distances still come from the quotient table, the event location is divided out in the lane — from clipped_length (w3)
                         // and left_clip (the record's tp field), without the piece's rare record.  With PF_HUGE: a table piece (all terms from the tables)
// an event byte w passes the base-quality test (:288) iff w >= piece_thr(c)  (-b above 62: only escape bytes of passing bases reach it)
BRC_HD uint32_t piece_thr(const DevCfg& c) { return eb_min_bq(c.min_bq) << 2; }

// One piece record, 48 bytes.  Dwords 0-9 are what the read loop of k_pileup2 needs of every piece (scalar loads x8 + x2, two
// pieces ahead); bytes 32-47 = {ww, a, bq_off} are the one 16-byte word the staging lanes load.
struct alignas(16) Piece {
    int32_t rs;            // reference position of the first base
    int32_t len;           // events: positions [rs, rs + len); every piece with len > 0 belongs to a read that counts
    int32_t ext;           // column: positions [rs, rs + ext), ext >= len
    uint32_t tp_flags;     // bits 24-31: PF_*; bits 0-23: with PF_TABLE / PF_TABQ bits 0-14 = 16 * three_prime_index - 8 * table_len (signed: the byte distance
                           // between the two table addresses of a probe, so that the second is one scalar add away from the first) and bits 15-23 =
                           // left_clip; else three_prime_index
    uint32_t w1, w2, w3;   // packed integer addends: three 10-bit counters 1 | rev << 10 | q2ok << 20;  mapq | sse << 16;  zm_sum | clipped << 16 — or, in a region with long reads (DevCfg.pack_shift 12), mapq | zm_sum << 12;  sse | clipped << 12
    float snm;             // NM / (float)clipped_length, 0 when NM is missing
    uint32_t ww;           // per-lane (not per-bucket) warning counters: SM-missing | NM-missing << 16 (process_read warnings, BasicStat.cpp:85,100)
    int32_t a;             // rs - query offset: the lane on position p sees query base p - a
    uint64_t bq_off;       // first element of the read's row in the event-byte stream
};
// What only the rare paths need, 32 bytes: the exact-division constants of pieces whose event terms are divided out (no
// PF_TABLE) and the raw integers of pieces whose packed addends left them out (PF_HUGE), and of any piece that leaves a
// third-allele event.  K1 WRITES it only for pieces without PF_TABLE or with PF_HUGE; for all others (93 % at 30x) the
// record is derived from the piece itself (piece_rare_of): l_qseq == clipped_length == table_len, no left clip, q2 in {tp, none}.
struct alignas(32) PieceRare {
    float rcpL, Lf, rcpC, center;   // correctly rounded 1 / (float)l_qseq, (float)l_qseq, 1 / center, center = (float)clipped_length / 2
    int32_t left, q2;
    uint32_t zm_raw, sse_raw;
};
BRC_HD uint32_t piece_flags(const Piece& h) { return h.tp_flags >> 24; }
BRC_HD int32_t piece_tp_field(uint32_t tp_flags) { return (int32_t)(tp_flags << 17) >> 17; }     // (bits 0-14, sign-extended)
BRC_HD int piece_left_field(uint32_t tp_flags) { return (int)((tp_flags >> 15) & 0x1ffu); }
BRC_HD int piece_tp_of(uint32_t tp_flags, int table_len) {
    const uint32_t fl = tp_flags >> 24;
    return (fl & (PF_TABLE | PF_TABQ)) ? (piece_tp_field(tp_flags) + 8 * table_len) >> 4 : (fl & PF_DIV) ? (int)(tp_flags & 0xffu) : (int)(tp_flags & 0xffffffu);
}
// n / m for small non-negative integers held in floats (n < 2^12, 0 < m < 2^10), correctly rounded.  On the device: the hardware's
// approximate reciprocal (v_rcp_f32, 1 ulp), one multiply, two FMAs — q0 = n*y, e = n - q0*m (exact in one FMA), q = RN(q0 + e*y).
// The error y carries into e*y is some 2^-46 of the quotient, and a quotient of such integers is either a float (then e*y lands on
// it) or at least 2^-35 of itself away from every midpoint between two floats (|n/m - M| >= 1/(m 2^s) for a 25-bit M = Mint/2^s):
// the last rounding cannot go the other way.  The engine checks the whole domain against the compiler's IEEE division when it is
// created (k_divcheck in brc_engine.hip) and refuses to come up otherwise; tests/exact_division.cpp walks the same sequence over the
// domain with RN(1/m) and every reciprocal up to 2 ulp away from it.
// On the host (simulator): `/`.
BRC_HD float div_small(float n, float m) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float y = __builtin_amdgcn_rcpf(m);
    const float q = n * y;
    const float e = __builtin_fmaf(-q, m, n);
    return __builtin_fmaf(e, y, q);
#else
    return n / m;
#endif
}
enum { DIV_SMALL_N = 4096, DIV_SMALL_M = 1024 };    // the domain k_divcheck covers: numerators below / denominators below
// the event-location term of a PF_TABQ piece at query position qpos: |(qpos - left) - cl/2| / (cl/2) is the correctly rounded quotient of the
// rational |2 (qpos - left) - cl| / cl — the same float whichever pair of exactly represented operands is divided (BasicStat.cpp:69-70)
BRC_HD double tabq_sev(int qpos, int left, uint32_t clipped) {
    return 1.0 - (double)div_small((float)absdiff_vs(2u * (uint32_t)qpos, 2u * (uint32_t)left + clipped), (float)clipped);   // (left < 512, clipped <= table_len <= 512: make_piece)
}
BRC_HD int piece_tp(const DevCfg& c, const Piece& h) { return piece_tp_of(h.tp_flags, c.table_len); }
BRC_HD bool piece_has_rare(uint32_t fl) { return (fl & PF_TABLE) == 0u || (fl & PF_HUGE) != 0u; }
// (sh: DevCfg.pack_shift — k_pileup2 hands over its template constant; -1: take it from the structure)
BRC_HD PieceRare piece_rare_of(const DevCfg& c, const Piece& h, int sh = -1) {       // for pieces with PF_TABLE and without PF_HUGE
    if (sh < 0) sh = c.pack_shift;
    PieceRare r;
    r.Lf = (float)c.table_len; r.center = (float)c.table_len * 0.5f; r.rcpL = 1.0f / r.Lf; r.rcpC = 1.0f / r.center;
    r.left = 0; r.q2 = (piece_flags(h) & PF_Q2OK) ? piece_tp(c, h) : -1;
    if (sh == 16) { r.zm_raw = h.w3 & 0xffffu; r.sse_raw = h.w2 >> 16; }
    else { r.zm_raw = h.w2 >> sh; r.sse_raw = h.w3 & ((1u << sh) - 1u); }
    return r;
}
BRC_HD uint32_t piece_mapq(int sh, const Piece& h) { return h.w2 & ((1u << sh) - 1u); }    // (w2 = mapq | zm << s, or mapq alone: PF_HUGE)
BRC_HD uint32_t piece_clipped(const PieceRare& r) { return (uint32_t)(r.center * 2.0f); }     // exact: clipped_length < 2^22

// what K1 knows about a read once it is annotated
struct ReadConst {
    int32_t pos, l_qseq, clipped, left, tp, q2;
    uint32_t flags;        // PF_REV | PF_Q2OK | PF_SMW | PF_NMW | PF_WIDE
    uint32_t mapq, zm, sse;
    float snm;
    uint64_t bq_off; uint32_t read;
    bool counts;           // MAPQ >= -q and no SECONDARY/QCFAIL/DUP flag
};

BRC_HD ReadConst read_const(const DevCfg& c, const DRead& rd, uint32_t read_index, bool wide) {
    ReadConst rc;
    rc.pos = rd.pos; rc.l_qseq = rd.l_qseq; rc.clipped = rd.clipped; rc.left = rd.left; rc.tp = rd.tp; rc.q2 = rd.q2;
    rc.flags = ((rd.misc & M_REV) ? (uint32_t)PF_REV : 0u) | ((rd.misc & M_Q2OK) ? (uint32_t)PF_Q2OK : 0u) | ((rd.misc & M_SMW) ? (uint32_t)PF_SMW : 0u) | ((rd.misc & M_NMW) ? (uint32_t)PF_NMW : 0u) | (wide ? (uint32_t)PF_WIDE : 0u);
    rc.mapq = (rd.misc >> 8) & 0xffu; rc.zm = rd.zm_sum; rc.sse = rd.sse_add; rc.snm = rd.snm_add; rc.bq_off = rd.bq_off; rc.read = read_index;
    rc.counts = (int)rc.mapq >= c.min_mapq && !(rd.misc & M_NOCOUNT);
    return rc;
}

// does the base in front of CIGAR operator k+1.. see an insertion (resolve_cigar2's peek: I, or P ... I)?
// (the walks over a read's operators take them through an accessor cig(k): a pointer on the host, in the simulator and in the serial
// device paths; K1's phase C hands over the first operators in registers — every cig[k] from memory is a dependent round trip for
// the whole wave there)
struct CigPtr { const uint32_t* p; BRC_HDM uint32_t operator()(uint32_t k) const { return p[k]; } };
template <class C>
BRC_HD bool peek_insertion_at(const C& cig, uint32_t nc, uint32_t k) {
    if (k + 1 >= nc) return false;
    const uint32_t c1 = cig(k + 1);
    const uint32_t op2 = c1 & 0xfu;
    if (op2 == CINS) return (c1 >> 4) > 0;
    if (op2 == CPAD && k + 2 < nc) {
        int l3 = 0;
        for (uint32_t kk = k + 2; kk < nc; ++kk) {
            const uint32_t ck = cig(kk);
            const uint32_t o = ck & 0xfu;
            if (o == CINS) l3 += (int)(ck >> 4);
            else if (o == CDEL || o == CMATCH || o == CREF_SKIP || o == CEQUAL || o == CDIFF) break;
        }
        return l3 > 0;
    }
    return false;
}
BRC_HD bool peek_insertion(const uint32_t* cig, uint32_t nc, uint32_t k) { CigPtr a; a.p = cig; return peek_insertion_at(a, nc, k); }

// THE piece decomposition (host: counting at push time; K1: emission; the two must agree, so both call this).
// f(rs, len, ext, qoff, nb).  `entered`: the read enters pileup columns at all (not dropped at push, has a usable CIGAR).
template <class C, class F>
BRC_HD void walk_pieces_at(bool insertion_centric, bool entered, bool counts, int32_t pos, const C& cig, uint32_t nc, F f) {
    if (!entered) return;
    int32_t x = pos; int y = 0;
    if (!counts) {
        int32_t rlen = 0;
        for (uint32_t k = 0; k < nc; ++k) { const uint32_t ck = cig(k); if (is_refop(ck & 0xfu)) rlen += (int32_t)(ck >> 4); }
        if (rlen > 0) f(pos, 0, rlen, 0, false);
        return;
    }
    bool have = false; int32_t crs = 0, clen = 0; int cq = 0; bool cnb = false;
    for (uint32_t k = 0; k < nc; ++k) {
        const uint32_t ck = cig(k); const uint32_t op = ck & 0xfu; const int32_t len = (int32_t)(ck >> 4);
        if (is_mop(op)) {
            if (len > 0) {
                if (have) f(crs, clen, x - crs, cq, cnb);
                else if (x > pos) f(pos, 0, x - pos, 0, false);              // leading deletion / skip: column only
                have = true;
                if (insertion_centric && peek_insertion_at(cig, nc, k)) {
                    if (len > 1) f(x, len - 1, len - 1, y, false);
                    crs = x + len - 1; clen = 1; cq = y + len - 1; cnb = true;
                } else { crs = x; clen = len; cq = y; cnb = false; }
            }
            x += len; y += len;
        } else if (op == CDEL || op == CREF_SKIP) x += len;
        else if (op == CINS || op == CSOFT_CLIP) y += len;
    }
    if (have) f(crs, clen, x - crs, cq, cnb);
    else if (x > pos) f(pos, 0, x - pos, 0, false);
}

// ---------------------------------------------------------------- CIGARs with an M / = / X operator of length zero
//
// htslib-1.10's resolve_cigar2 keeps a cursor (operator k, its reference start x, the query offset y) per buffered read and moves it by ONE
// reference-consuming operator whenever the column has passed the current one — without asking whether the column lies inside the operator it
// lands on.  Every aligner's CIGAR has operators of positive length and the question never arises; on an operator of length zero the cursor
// stands for one column: the column is reported as a match at the operator's query offset, whatever follows is seen one column late, and a
// deletion behind it is never announced (its "last base of the operator before" test looks at the empty operator).  The formulas of
// walk_pieces_at / enumerate_indels_at — which compute every segment from the operators' lengths — cannot say that; such reads (round 6: until
// then refused at push) take the cursor itself, column by column: exact, one lane per read, and rare enough not to matter.
// (Restated from the published algorithm, htslib sam.c; the oracle's resolve_cigar2 and the shim's are the two other statements of it.)
struct CigCursor { int32_t k, x; int y; };
BRC_HD bool has_empty_mop(const uint32_t* cig, uint32_t nc) { for (uint32_t k = 0; k < nc; ++k) if (is_mop(cig[k] & 0xfu) && (cig[k] >> 4) == 0u) return true; return false; }
// the pileup entry of a read at column `col` (called for every column from the read's start on, ascending); false: not in the column
BRC_HD bool cursor_resolve(const uint32_t* cig, uint32_t nc, int32_t rpos, int32_t col, CigCursor& s, int& qpos, bool& is_del, int& indel) {
    const int32_t n = (int32_t)nc;
    if (s.k == -1) {
        qpos = 0;
        if (n == 1) { if (is_mop(cig[0] & 0xfu)) { s.k = 0; s.x = rpos; s.y = 0; } }
        else {
            int32_t k = 0; s.x = rpos; s.y = 0;
            for (; k < n; ++k) { const uint32_t op = cig[k] & 0xfu; if (is_refop(op)) break; else if (op == CINS || op == CSOFT_CLIP) s.y += (int)(cig[k] >> 4); }
            s.k = k;
        }
        if (s.k < 0 || s.k >= n) return false;
    } else {
        const int32_t l = (int32_t)(cig[s.k] >> 4);
        if (col - s.x >= l) {
            if (s.k + 1 >= n) return false;
            const bool cur_m = is_mop(cig[s.k] & 0xfu);
            if (is_refop(cig[s.k + 1] & 0xfu)) { if (cur_m) s.y += l; s.x += l; ++s.k; }
            else {
                if (cur_m) s.y += l;
                s.x += l;
                int32_t k = s.k + 1;
                for (; k < n; ++k) { const uint32_t op = cig[k] & 0xfu; if (is_refop(op)) break; else if (op == CINS || op == CSOFT_CLIP) s.y += (int)(cig[k] >> 4); }
                s.k = k;
            }
            if (s.k >= n) return false;
        }
    }
    const uint32_t op = cig[s.k] & 0xfu; const int32_t l = (int32_t)(cig[s.k] >> 4);
    is_del = false; indel = 0;
    if (s.x + l - 1 == col && s.k + 1 < n) {
        const uint32_t op2 = cig[s.k + 1] & 0xfu; const int l2 = (int)(cig[s.k + 1] >> 4);
        if (op2 == CDEL) indel = -l2;
        else if (op2 == CINS) indel = l2;
        else if (op2 == CPAD && s.k + 2 < n) {
            int l3 = 0;
            for (int32_t kk = s.k + 2; kk < n; ++kk) { const uint32_t o = cig[kk] & 0xfu; if (o == CINS) l3 += (int)(cig[kk] >> 4); else if (is_refop(o)) break; }
            if (l3 > 0) indel = l3;
        }
    }
    if (is_mop(op)) qpos = s.y + (col - s.x);
    else if (op == CDEL || op == CREF_SKIP) { is_del = true; qpos = s.y; }
    return true;
}
// walk_pieces_at for such a read: the segments the cursor's columns form
template <class F>
BRC_HD void walk_pieces_cursor(bool insertion_centric, bool entered, bool counts, int32_t pos, const uint32_t* cig, uint32_t nc, F f) {
    if (!entered) return;
    int32_t rlen = 0;
    for (uint32_t k = 0; k < nc; ++k) if (is_refop(cig[k] & 0xfu)) rlen += (int32_t)(cig[k] >> 4);
    if (!counts) { if (rlen > 0) f(pos, 0, rlen, 0, false); return; }
    CigCursor s; s.k = -1; s.x = pos; s.y = 0;
    bool have = false; int32_t crs = 0, clen = 0; int cq = 0; bool cnb = false;
    int32_t col = pos, last_in = pos;           // last_in: one past the last column the read was seen in
    for (; col < pos + rlen; ++col) {
        int qpos = 0, indel = 0; bool is_del = false;
        if (!cursor_resolve(cig, nc, pos, col, s, qpos, is_del, indel)) continue;      // (the cursor has run out of operators: nothing behind this column either)
        last_in = col + 1;
        if (is_del) continue;
        const bool nb = insertion_centric && indel > 0;                                  // :343: counted in the depth, in no base bucket
        if (have && !cnb && !nb && col == crs + clen && qpos == cq + clen) { ++clen; continue; }
        if (have) f(crs, clen, col - crs, cq, cnb);
        else if (col > pos) f(pos, 0, col - pos, 0, false);                              // leading deletion / skip: column only
        have = true; crs = col; clen = 1; cq = qpos; cnb = nb;
    }
    if (have) f(crs, clen, last_in - crs, cq, cnb);
    else if (last_in > pos) f(pos, 0, last_in - pos, 0, false);
}

template <class F>
BRC_HD void walk_pieces(bool insertion_centric, bool entered, bool counts, int32_t pos, const uint32_t* cig, uint32_t nc, F f) {
    if (has_empty_mop(cig, nc)) { walk_pieces_cursor(insertion_centric, entered, counts, pos, cig, nc, f); return; }
    CigPtr a; a.p = cig; walk_pieces_at(insertion_centric, entered, counts, pos, a, nc, f);
}

// does a read enter pileup columns?  (bam_plp_push drop rules + a CIGAR resolve_cigar2 can stand on)
BRC_HD bool read_enters(uint32_t flag, const uint32_t* cig, uint32_t nc) {
    if (flag & BRC_PUSH_MASK) return false;
    if (nc == 0) return false;
    if (nc == 1 && !is_mop(cig[0] & 0xfu)) return false;
    return true;
}

// (sh = DevCfg.pack_shift, handed over separately: K1 is instantiated per layout, so the shifts of its common instantiation are
// immediates — as a run-time value they cost it the registers that keep its piece walk free of spills)
BRC_HD void make_piece(const DevCfg& c, const ReadConst& r, int32_t rs, int32_t len, int32_t ext, int qoff, bool nb, Piece& h, PieceRare& rare, const int sh) {
    h.rs = rs; h.a = rs - qoff; h.len = len; h.ext = ext;
    uint32_t fl = r.flags;
    const bool q2ok = (fl & PF_Q2OK) != 0;
    if (c.table_len > 0 && r.l_qseq == c.table_len && r.clipped == c.table_len && r.left == 0 && r.tp >= 0 && r.tp <= c.table_len &&
        (!q2ok || r.q2 == r.tp)) fl |= PF_TABLE;
    // a wide read's pieces sit behind the same test; one that would have been a table piece keeps "every term from the table" as
    // PF_TABQ (no clip: the in-lane event location of the PF_TABQ path is the table's value)
    if ((fl & PF_WIDE) && (fl & PF_TABLE)) fl = (fl & ~(uint32_t)PF_TABLE) | PF_TABQ;
    // k_pileup2 finds every unusual piece behind ONE test, "no PF_TABLE": -i's one-base pieces, and pieces with huge integers —
    // those keep PF_TABQ as the mark of "all three terms from the tables" when they were table pieces
    if (nb) fl = (fl | PF_NB) & ~(uint32_t)PF_TABLE;
    const bool huge = r.zm > c.pack_lim || r.sse > (sh == 16 ? c.pack_lim : c.pack_lim_lo) || (uint32_t)r.clipped > c.pack_lim;   // (16 + 16 bits: one limit)
    if (huge) fl = (fl & PF_TABLE) ? ((fl & ~(uint32_t)PF_TABLE) | PF_HUGE | PF_TABQ) : (fl | PF_HUGE);
    if (!(fl & (PF_TABLE | PF_TABQ)) && !nb && !huge && c.table_len > 0 && r.l_qseq == c.table_len && r.clipped > 0 && r.left >= 0 && r.left < 512 &&
        r.tp >= 0 && r.tp <= c.table_len && (!q2ok || r.q2 == r.tp)) fl |= PF_TABQ;
    if (!(fl & (PF_TABLE | PF_TABQ | PF_NB)) && !huge && r.l_qseq > 0 && r.l_qseq <= 255 && r.clipped > 0 && r.left >= 0 && r.left <= 255 && r.tp >= 0 && r.tp <= 255 &&
        (!q2ok || r.q2 == r.tp)) fl |= PF_DIV;
    h.tp_flags = ((fl & (PF_TABLE | PF_TABQ)) ? (((uint32_t)(16 * r.tp - 8 * c.table_len) & 0x7fffu) | ((uint32_t)r.left << 15)) :
                  (fl & PF_DIV) ? ((uint32_t)r.tp | ((uint32_t)r.l_qseq << 8) | ((uint32_t)r.left << 16)) : ((uint32_t)r.tp & 0xffffffu)) | ((fl & 0xffu) << 24);      // l_qseq < 2^22 is checked at push
    h.w1 = 1u | ((fl & PF_REV) ? (1u << 10) : 0u) | (q2ok ? (1u << 20) : 0u);
    if (sh == 16) { h.w2 = r.mapq | (huge ? 0u : (r.sse << 16)); h.w3 = huge ? 0u : (r.zm | ((uint32_t)r.clipped << 16)); }             // 16 + 16: mapq | sse, zm | clipped
    else { h.w2 = r.mapq | (huge ? 0u : (r.zm << sh)); h.w3 = huge ? 0u : (r.sse | ((uint32_t)r.clipped << sh)); }                     // 12 + 20: mapq | zm, sse | clipped
    h.snm = r.snm;
    h.ww = ((fl & PF_SMW) ? 1u : 0u) | ((fl & PF_NMW) ? (1u << 16) : 0u);
    h.bq_off = r.bq_off;
    // (the rare record is stored only when piece_has_rare(flags): the two divisions are skipped with it)
    if (piece_has_rare(fl)) {
        rare.Lf = (float)r.l_qseq; rare.center = (float)r.clipped * 0.5f; rare.rcpL = 1.0f / rare.Lf; rare.rcpC = 1.0f / rare.center;
        rare.left = r.left; rare.q2 = r.q2; rare.zm_raw = r.zm; rare.sse_raw = r.sse;
    }
}

// event terms of a piece at query position qpos by exact division (any read) ...
BRC_HD EvTerms piece_terms_div(uint32_t fl, int tp, const PieceRare& r, int qpos) {
    EvTerms t;
    t.q2 = (fl & PF_Q2OK) ? div_rcp((float)BRC_ABSDIFF(qpos, r.q2), r.Lf, r.rcpL) : 0.0f;
    t.s3p = div_rcp((float)BRC_ABSDIFF(qpos, tp), r.Lf, r.rcpL);
    float d = (float)(qpos - r.left) - r.center;
    d = d < 0.0f ? -d : d;
    t.sev = 1.0 - (double)div_rcp(d, r.center, r.rcpC);
    return t;
}
// ... a PF_DIV piece's from its own record (tp | l_qseq << 8 | left_clip << 16 in the tp field, clipped_length in w3): the
// reference's expressions, BasicStat.cpp:60-70, with both sides of the second quotient doubled — |(qpos - left) - clipped/2| /
// (clipped/2) = |2 (qpos - left) - clipped| / clipped as real numbers, so the correctly rounded quotients are the same float —
// which makes every operand a small integer (q2 == tp or no q2; l_qseq, left, tp <= 255 and 0 < clipped <= l_qseq: make_piece)
BRC_HD EvTerms piece_terms_inlane(uint32_t fl, uint32_t tp_flags, uint32_t cl /* clipped_length: w3 >> pack_shift */, uint32_t qpos) {
    EvTerms t;
    const uint32_t tp = tp_flags & 0xffu, left = (tp_flags >> 16) & 0xffu;
    t.s3p = div_small((float)absdiff_vs(qpos, tp), (float)((tp_flags >> 8) & 0xffu));
    t.q2 = (fl & PF_Q2OK) ? t.s3p : 0.0f;
    t.sev = 1.0 - (double)div_small((float)absdiff_vs(2u * qpos, 2u * left + cl), (float)cl);     // |2 (qpos - left) - cl|
    return t;
}
// ... and from the quotient tables (PF_TABLE): one float look-up serves both distances (q2 == tp or no q2)
BRC_HD EvTerms piece_terms_tab(const Piece& h, const TermTab& tt, int table_len, int qpos) {
    EvTerms t;
    t.s3p = tt.q[absdiff_u((uint32_t)qpos, (uint32_t)piece_tp_of(h.tp_flags, table_len))];
    t.q2 = (piece_flags(h) & PF_Q2OK) ? t.s3p : 0.0f;
    t.sev = tt.e[absdiff_u(2u * (uint32_t)qpos, (uint32_t)table_len)];
    return t;
}

// One bucket of a lane between two flushes: three packed integer registers (w1: three 10-bit counters; w2, w3: a narrow and a wide
// sum each, whose widths and the K they bound choose_pack picks per region), the sum of the EVENT VALUES (quality << 2 | code: every event of a slot
// has the slot's code, so the base-quality sum is (sum - count * code) >> 2 — no unpacking per event) and the four
// order-sensitive float sums.
struct PackAcc { uint32_t w1, w2, w3, sw; float f[NF]; };
BRC_HD void pack_init(PackAcc& a) { a.w1 = a.w2 = a.w3 = a.sw = 0; for (int f = 0; f < NF; ++f) a.f[f] = 0.0f; }
BRC_HD void pack_event(PackAcc& a, const Piece& h, const EvTerms& t, uint32_t word) {
    a.w1 += h.w1; a.w2 += h.w2; a.w3 += h.w3; a.sw += word;
    a.f[F_SQ2] += t.q2; a.f[F_S3P] += t.s3p;
    a.f[F_SEV] = (float)((double)a.f[F_SEV] + t.sev);
    a.f[F_SNM] += h.snm;
}
// the nine integer plane values held by a PackAcc whose events all carry index b (I_* order)
BRC_HD void pack_unpack(const PackAcc& a, uint32_t b, uint32_t sh /* DevCfg.pack_shift */, uint32_t* v) {
    const uint32_t n = a.w1 & 0x3ffu, minus = (a.w1 >> 10) & 0x3ffu, lo = (1u << sh) - 1u;
    v[I_N] = n; v[I_SMQ] = a.w2 & lo; v[I_PLUS] = n - minus; v[I_MINUS] = minus;
    v[I_NQ2] = a.w1 >> 20; v[I_SCLIP] = a.w3 >> sh; v[I_SBQ] = (a.sw - n * b) >> 2;
    if (sh == 16) { v[I_SSE] = a.w2 >> 16; v[I_SMMQ] = a.w3 & 0xffffu; } else { v[I_SMMQ] = a.w2 >> sh; v[I_SSE] = a.w3 & lo; }
}

// The packed sums of a lane, their widths and their limits.  w2 and w3 each hold a NARROW and a WIDE sum: mapping quality | mismatch-
// quality sum, single-ended mapping quality | clipped length, the narrow one in the low `shift` bits.  K = pieces between two
// flushes; a flush can only happen BETWEEN half-batches (HALF pieces), so up to max(K, HALF) pieces are summed into a field before it is
// emptied, and a per-read value above (field maximum) / max(K, HALF) sends its piece down the PF_HUGE path (make_piece).
//   reads up to 5461 bases (65535 / HALF):  16 + 16 bits, K = 65535 / longest read, at most 127 — every field holds K reads' worth;
//   longer reads:                           12 + 20 bits, K = (2^20 - 1) / longest read, at most 16 (255 x 16 < 2^12: a mapping quality
//                                           always fits); clipped lengths and mismatch-quality sums up to 87 381 stay packed.
// Both big kernels are instantiated per layout (template parameter SH), the host launches the instantiation of the region's
// DevCfg.pack_shift: the short-read instantiations keep their shifts as immediates and are, instruction for instruction, the kernels that
// were measured before the second layout existed (tools/isa_diff.py).
// (Until round 4's end the limit was 65535 / K whatever HALF: twelve 20-kb reads in one half-batch overflowed the 16-bit sums — no test
// had a read longer than 900 bases; tests/test_gpu_parity.py::test_hip_long_reads.  The 12 + 20 layout took long reads off the
// PF_HUGE path: 10-kb reads ran k_pileup2 at 4.9x the time per event of 150-base reads.)
// (the overrides are test knobs: a small K exercises the flushes, a small limit the PF_HUGE path)
#ifndef BRC_HALF
#define BRC_HALF 12
#endif
enum { HALF = BRC_HALF };
static_assert(255 * HALF < (1 << 12), "a half-batch of mapping qualities of 255 must fit the 12-bit narrow field (choose_pack: lim_lo >= 255)");
BRC_HD void choose_pack(int32_t max_lqseq, int32_t k_override, int32_t lim_override, int32_t& K, uint32_t& lim, uint32_t& lim_lo, int32_t& shift) {
    const int32_t m = max_lqseq < 255 ? 255 : max_lqseq;
    if ((int64_t)m * HALF <= 65535) { shift = 16; K = 65535 / m; if (K > 127) K = 127; }
    else { shift = 12; K = (int32_t)(((1u << 20) - 1u) / (uint32_t)m); if (K > 16) K = 16; }
    if (K < 1) K = 1;
    if (k_override > 0 && k_override < K) K = k_override;
    const uint32_t keff = (uint32_t)(K > (int32_t)HALF ? K : (int32_t)HALF);
    lim = ((shift == 16 ? 0xffffu : 0xfffffu)) / keff;
    lim_lo = ((1u << shift) - 1u) / keff;                                                       // >= 255 by construction (keff <= 127 resp. 16)
    if (lim_override >= 255 && (uint32_t)lim_override < lim) lim = (uint32_t)lim_override;    // >= 255: a mapping quality always fits
    if (lim_override >= 255 && (uint32_t)lim_override < lim_lo) lim_lo = (uint32_t)lim_override;
}

// The event-byte stream on the device is padded: EB_PAD_FRONT bytes before its first row, EB_PAD_BACK + (the batch's longest read) behind
// its last.  A tile stages, for every piece of its range, the 80-byte window [ws, ws + 80) of the piece's row that its 64 positions can
// touch, ws = floor16(p0 - a) — up to 79 bytes before the row, up to 79 + 15 past it.  A piece that only SPANS the tile (an intron, a long
// deletion) or does not touch it has its window anywhere — 100 kb off the row — and nobody reads the copy: a window further from the row's
// start than the longest read of the batch takes the row's first bytes instead; what is left — a SHORT read's window up to a long read's
// length past its own row, when both are in the batch — is what the back pad is sized for.
// (k_pileup2's BRC_STAGE states the same two lines in place; the simulator checks with this function that every window the device
// would copy lies inside the padded stream — the fault on spliced alignments of round 4 was invisible to it before.)
enum { EB_PAD_FRONT = 128, EB_PAD_BACK = 512, EB_WINDOW = 80 };
// what the comments above promise, checked where the constants are defined: a window starts at most 95 bytes before a row (the
// test below lets ws in [-96, max_lqseq + 16]; floor16 keeps it >= -80 for a piece that touches the tile) — inside the front pad;
// a window that survives the test ends at most max_lqseq + 16 + 80 past the row's START, and the stream is allocated with
// EB_PAD_BACK + max_lqseq behind its last row: 96 <= EB_PAD_BACK.  (brc_upload repeats the run-time half on its own numbers.)
static_assert(EB_PAD_FRONT >= 96 + 16, "a staged window may start 96 bytes before the stream's first row");
static_assert(EB_PAD_BACK >= 16 + EB_WINDOW + 16, "a staged window of the last row ends inside the back pad");
static_assert(EB_WINDOW == 80 && TILE + 15 < EB_WINDOW, "64 tile positions + 15 bytes of alignment fit a window");
BRC_HD int32_t stage_window_start(int32_t p0, int32_t a, int32_t max_lqseq) {
    int32_t ws = (p0 - a) & ~15;
    if ((uint32_t)(ws + 96) > (uint32_t)max_lqseq + 96u + 16u) ws = 0;
    return ws;
}

// Per-lane state of KB v2.
// (HALF, defined above choose_pack:) pieces per staging half-batch (12 rows x 5 chunks of 16 event bytes = 60 lanes of one direct-to-LDS instruction; a multiple of 3,
                            // the rotation period of the piece-record registers); queue drains and flushes happen between half-batches
struct LaneAcc2 {
    PackAcc dom, alt;
    uint32_t dom_b, alt_b, ncol, depth, ww;   // ww: SM-missing | NM-missing << 16 warnings of this position
};
BRC_HD void lane2_init(LaneAcc2& a, uint32_t dom_b) {
    pack_init(a.dom); pack_init(a.alt); a.dom_b = dom_b; a.alt_b = NB_NONE; a.ncol = a.depth = a.ww = 0;
}
BRC_HD uint32_t* slot_i(const DevCfg& c, const Planes& pl, int lib, uint32_t slot, int64_t k) { return pl.si + (((int64_t)lib * 2 + slot) * NI) * c.PS + k; }
BRC_HD float* slot_f(const DevCfg& c, const Planes& pl, int lib, uint32_t slot, int64_t k) { return pl.sf + (((int64_t)lib * 2 + slot) * NF) * c.PS + k; }

// The functions below are the RARE paths of KB (a flush every K pieces, a drained event per few tiles).  They are written
// as rolled loops over the plane index with the value picked by a select chain: slow, but a handful of live registers —
// inlined into the read loop they must not raise its register peak.
#if defined(__clang__)
#define BRC_NOUNROLL _Pragma("unroll 1")
#else
#define BRC_NOUNROLL _Pragma("GCC unroll 1")
#endif
BRC_HD uint32_t pack_field(const PackAcc& a, uint32_t b, uint32_t sh, int f) {
    const uint32_t n = a.w1 & 0x3ffu, minus = (a.w1 >> 10) & 0x3ffu, lo = (1u << sh) - 1u;
    uint32_t v = n;                                            // I_N
    const uint32_t w2hi = a.w2 >> sh, w3lo = a.w3 & lo;
    v = f == I_SMQ ? (a.w2 & lo) : v; v = f == I_SSE ? (sh == 16 ? w2hi : w3lo) : v; v = f == I_PLUS ? n - minus : v; v = f == I_MINUS ? minus : v;
    v = f == I_NQ2 ? (a.w1 >> 20) : v; v = f == I_SMMQ ? (sh == 16 ? w3lo : w2hi) : v; v = f == I_SCLIP ? (a.w3 >> sh) : v; v = f == I_SBQ ? ((a.sw - n * b) >> 2) : v;
    return v;
}
// packed registers -> the integer planes of one slot (adds when the tile has flushed before), registers reset; b = the slot's index (eb_index)
BRC_HD void flush_slot(const DevCfg& c, const Planes& pl, int lib, int64_t k, PackAcc& a, uint32_t slot, uint32_t b, bool live, int sh = -1) {
    if (sh < 0) sh = c.pack_shift;
    uint32_t* ip = BRC_CK(c, CK_PILEUP, 40, CB_SI, slot_i(c, pl, lib, slot, k), ((uint64_t)(NI - 1) * (uint64_t)c.PS + 1u) * 4u, k, -1);
    BRC_NOUNROLL
    for (int f = 0; f < NI; ++f) { const uint32_t v = pack_field(a, b, (uint32_t)sh, f); ip[(int64_t)f * c.PS] = v + (live ? ip[(int64_t)f * c.PS] : 0u); }
    a.w1 = a.w2 = a.w3 = a.sw = 0;
}
// `live`: the tile has flushed before (wave-uniform)
BRC_HD void lane2_flush(const DevCfg& c, const Planes& pl, int lib, int64_t k, LaneAcc2& a, bool live, int sh = -1) {
    flush_slot(c, pl, lib, k, a.dom, 0u, eb_index(a.dom_b), live, sh);
    flush_slot(c, pl, lib, k, a.alt, 1u, eb_index(a.alt_b), live, sh);                 // (no alternate yet: nothing to unpack)
}
// one event of a third (fourth, ...) base at this position: its raw addends, for the list
BRC_HD XEv make_xev(const DevCfg& c, int lib, int64_t k, const Piece& h, const PieceRare& rare, int qpos, uint32_t q, uint32_t b, int sh = -1) {
    if (sh < 0) sh = c.pack_shift;
    XEv e;
    const uint32_t fl = piece_flags(h);
    const EvTerms t = piece_terms_div(fl, piece_tp(c, h), rare, qpos);
    e.k = (uint32_t)k; e.lib_b = ((uint32_t)lib << 8) | b;
    e.mapq = piece_mapq(sh, h); e.sse = rare.sse_raw; e.zm = rare.zm_raw; e.clip = piece_clipped(rare);
    e.qf = q | ((fl & PF_REV) ? 0x100u : 0u) | ((fl & PF_Q2OK) ? 0x200u : 0u);
    e.fq2 = t.q2; e.fs3p = t.s3p; e.fsnm = h.snm; e.sev = t.sev;
    return e;
}
// the integers of a PF_HUGE piece that its packed addends left out, for a lane whose event went to `slot` (the slot planes
// are live: the caller flushed)
BRC_HD void drain_int(const DevCfg& c, const Planes& pl, int lib, int64_t k, const PieceRare& rare, uint32_t slot) {
    uint32_t* ip = BRC_CK(c, CK_PILEUP, 41, CB_SI, slot_i(c, pl, lib, slot, k), ((uint64_t)(NI - 1) * (uint64_t)c.PS + 1u) * 4u, k, -1);
    ip[(int64_t)I_SSE * c.PS] += rare.sse_raw; ip[(int64_t)I_SMMQ * c.PS] += rare.zm_raw; ip[(int64_t)I_SCLIP * c.PS] += piece_clipped(rare);
}
// end of the tile (reference statement; the kernel stores the same values with coalesced selects)
BRC_HD void lane2_store(const DevCfg& c, const Planes& pl, int lib, int64_t k, LaneAcc2& a, bool dead, bool live) {
    const int64_t P = c.PS;
    pl.ncol[(int64_t)lib * P + k] = dead ? 0u : a.ncol;
    pl.depth[(int64_t)lib * P + k] = dead ? 0u : a.depth;
    pl.slotid[(int64_t)lib * P + k] = dead ? (1u | ((uint32_t)NB_NONE << 8)) : (a.dom_b | (a.alt_b << 8));
    for (uint32_t sl = 0; sl < 2u; ++sl) {
        PackAcc& r = sl ? a.alt : a.dom;
        float* fp = slot_f(c, pl, lib, sl, k);
        for (int f = 0; f < NF; ++f) fp[(int64_t)f * P] = dead ? 0.0f : r.f[f];
        flush_slot(c, pl, lib, k, r, sl, eb_index(sl ? a.alt_b : a.dom_b), live);
        if (dead) { uint32_t* ip = slot_i(c, pl, lib, sl, k); for (int f = 0; f < NI; ++f) ip[(int64_t)f * P] = 0u; }
    }
}

// tile -> piece range inside one library's stream [seg_lo, seg_hi): lo = first piece whose running-max reach exceeds the
// tile's first position, hi = first piece whose read starts after its last one (key = start of the parent read)
BRC_HD void tile_range2(const DevCfg& c, const int32_t* prefmax_reach, const int32_t* key, int64_t seg_lo, int64_t seg_hi, int64_t t, uint32_t& lo, uint32_t& hi) {
    const int64_t p0 = (int64_t)c.pos0 + t * TILE, p1 = p0 + TILE - 1;
    int64_t a = seg_lo, b = seg_hi;
    while (a < b) { const int64_t m = (a + b) >> 1; if ((int64_t)prefmax_reach[m] > p0) b = m; else a = m + 1; }
    lo = (uint32_t)a;
    b = seg_hi;
    while (a < b) { const int64_t m = (a + b) >> 1; if ((int64_t)key[m] > p1) b = m; else a = m + 1; }
    hi = (uint32_t)a;
}

// ---------------------------------------------------------------- K1': per-read indel event enumeration

// Calls emit(p, qpos, len) for every event of read `rd` that pileup_func would bucket as an indel allele
// (bamreadcount.cpp:288-342): last base of an M/=/X operator followed by I, D or P..I, inside the processing
// window [beg0-1,end), passing the MAPQ / base-quality filters.
// qual_row = the read's QUAL bytes (in.qual + qual_off[read])
template <class C, class F>
BRC_HD void enumerate_indels_at(const DevCfg& c, const C& cig, const DRead& rd, const uint8_t* qual_row, F emit) {
    if (rd.end <= rd.pos || (rd.misc & M_SIMPLE) || !c.has_ref) return;
    if (((rd.misc >> 16) & 0xffu) == 0) return;                         // library unavailable: position is abandoned
    if ((int)((rd.misc >> 8) & 0xffu) < c.min_mapq || (rd.misc & M_NOCOUNT)) return;
    const uint32_t nc = rd.n_cigar;
    int32_t x = rd.pos; int y = 0;
    for (uint32_t k = 0; k < nc; ++k) {
        const uint32_t ck = cig(k); const uint32_t op = ck & 0xfu; const int len = (int)(ck >> 4);
        if (is_refop(op)) {
            if (is_mop(op) && len > 0 && k + 1 < nc) {
                const uint32_t c1 = cig(k + 1); const uint32_t op2 = c1 & 0xfu; const int l2 = (int)(c1 >> 4);
                int indel = 0;
                if (op2 == CDEL) indel = -l2;
                else if (op2 == CINS) indel = l2;
                else if (op2 == CPAD && k + 2 < nc) {
                    int l3 = 0;
                    for (uint32_t kk = k + 2; kk < nc; ++kk) {
                        const uint32_t ck2 = cig(kk); const uint32_t o = ck2 & 0xfu;
                        if (o == CINS) l3 += (int)(ck2 >> 4);
                        else if (o == CDEL || o == CMATCH || o == CREF_SKIP || o == CEQUAL || o == CDIFF) break;
                    }
                    if (l3 > 0) indel = l3;
                }
                if (indel != 0) {
                    const int32_t p = x + len - 1; const int qpos = y + len - 1;
                    if (p >= c.beg0 - 1 && p < c.end && p >= c.pos0 && (int64_t)p < (int64_t)c.pos0 + c.P) {
                        const uint32_t q = qual_row[qpos];
                        if ((int)q >= c.min_bq) emit(p, qpos, indel);
                    }
                }
            }
            x += len;
            if (is_mop(op)) y += len;
        } else if (op == CINS || op == CSOFT_CLIP) y += len;
    }
}

// ... for a read with an empty M / = / X operator: the indel announcements of the cursor's own columns (see cursor_resolve)
template <class F>
BRC_HD void enumerate_indels_cursor(const DevCfg& c, const uint32_t* cig, const DRead& rd, const uint8_t* qual_row, F emit) {
    if (rd.end <= rd.pos || (rd.misc & M_SIMPLE) || !c.has_ref) return;
    if (((rd.misc >> 16) & 0xffu) == 0) return;
    if ((int)((rd.misc >> 8) & 0xffu) < c.min_mapq || (rd.misc & M_NOCOUNT)) return;
    CigCursor s; s.k = -1; s.x = rd.pos; s.y = 0;
    for (int32_t col = rd.pos; col < rd.end; ++col) {
        int qpos = 0, indel = 0; bool is_del = false;
        if (!cursor_resolve(cig, rd.n_cigar, rd.pos, col, s, qpos, is_del, indel)) continue;
        if (is_del || indel == 0) continue;
        if (col >= c.beg0 - 1 && col < c.end && col >= c.pos0 && (int64_t)col < (int64_t)c.pos0 + c.P && qpos < rd.l_qseq && (int)qual_row[qpos] >= c.min_bq) emit(col, qpos, indel);
    }
}
template <class F>
BRC_HD void enumerate_indels(const DevCfg& c, const DevIn& in, const DRead& rd, const uint8_t* qual_row, F emit) {
    if (has_empty_mop(in.cigar + rd.cig_off, rd.n_cigar)) { enumerate_indels_cursor(c, in.cigar + rd.cig_off, rd, qual_row, emit); return; }
    CigPtr a; a.p = in.cigar + rd.cig_off; enumerate_indels_at(c, a, rd, qual_row, emit);
}

// bucket of the base at element e of the event-byte stream (an escape byte: from the wide stream)
BRC_HD uint32_t base_bucket(const DevIn& in, uint64_t row, uint64_t q) { const uint32_t w = in.eb[row + q]; return eb_is_escape(w) ? (uint32_t)(in.bqw[wide_base(in.bqw, row) + q] & 0xffu) : (w & 3u) + 1u; }
// Same allele?  Deletions: same length (the allele text is the reference, identical for equal length);
// insertions: same canonical ("=ACGTN") inserted bases (bamreadcount.cpp:324-338).
BRC_HD bool same_allele(const DevIn& in, const DRead* reads, const IndelEv& a, const IndelEv& b) {
    if (a.len != b.len) return false;
    if (a.len < 0) return true;
    const DRead& ra = reads[a.read]; const DRead& rb = reads[b.read];
    for (int j = 0; j < a.len; ++j) {
        const int qa = a.qpos + 1 + j, qb = b.qpos + 1 + j;
        const uint32_t ca = qa < ra.l_qseq ? base_bucket(in, ra.bq_off, (uint64_t)qa) : 5u;
        const uint32_t cb = qb < rb.l_qseq ? base_bucket(in, rb.bq_off, (uint64_t)qb) : 5u;
        if (ca != cb) return false;
    }
    return true;
}

// Order of the indel events inside a bucket: by key (position, library), then by read index (= pileup column order; a read
// has at most one event per key, so there are no ties).
BRC_HD bool iev_before(const IndelEv& a, const IndelEv& b) { return a.key_lo < b.key_lo || (a.key_lo == b.key_lo && a.read < b.read); }
BRC_HD void sort_indel_events(IndelEv* ev, int n) {
    if (n <= 48) {
        for (int i = 1; i < n; ++i) {                                   // insertion sort: n is tiny as a rule
            const IndelEv t = ev[i]; int j = i - 1;
            while (j >= 0 && iev_before(t, ev[j])) { ev[j + 1] = ev[j]; --j; }
            ev[j + 1] = t;
        }
    } else {
        // deep targeted data can put thousands of reads on one indel: heapsort (in place, n log n moves)
        auto sift = [&](int root, int end) {
            const IndelEv t = ev[root];
            for (;;) {
                int ch = 2 * root + 1;
                if (ch >= end) break;
                if (ch + 1 < end && iev_before(ev[ch], ev[ch + 1])) ++ch;
                if (!iev_before(t, ev[ch])) break;
                ev[root] = ev[ch]; root = ch;
            }
            ev[root] = t;
        };
        for (int i = n / 2 - 1; i >= 0; --i) sift(i, n);
        for (int e2 = n - 1; e2 > 0; --e2) { const IndelEv t = ev[0]; ev[0] = ev[e2]; ev[e2] = t; sift(0, e2); }
    }
}

// KI: ordered reduction of the n events of one (position, library) key, already in read order (= pileup column order): folded
// into out[0..na) (one IndelOut per distinct allele, first-seen order).  Returns na.  w_sm/w_nm: warning counts.
BRC_HD int fold_indel_key(const DevCfg& c, const DevIn& in, const DRead* reads, const IndelEv* ev, int n, int32_t pos, int lib,
                          IndelOut* out, uint32_t& w_sm, uint32_t& w_nm) {
    (void)c;
    int na = 0;
    for (int i = 0; i < n; ++i) {
        const IndelEv e = ev[i];
        int g = 0;
        for (; g < na; ++g) {
            IndelEv rep; rep.read = out[g].rep_read; rep.qpos = out[g].rep_qpos; rep.len = out[g].len; rep.key_lo = 0;
            if (same_allele(in, reads, rep, e)) break;
        }
        if (g == na) {
            IndelOut o; o.pos = pos; o.lib = lib; o.len = e.len; o.rep_read = e.read; o.rep_qpos = e.qpos;
            for (int f = 0; f < NI; ++f) o.i[f] = 0;
            for (int f = 0; f < NF; ++f) o.f[f] = 0.0f;
            out[na++] = o;
        }
        const DRead& rd = reads[e.read];
        uint32_t ai[NACC_I]; float af[NF];
        ai[A_SMQ] = out[g].i[I_SMQ]; ai[A_SSE] = out[g].i[I_SSE]; ai[A_PLUS] = out[g].i[I_PLUS]; ai[A_MINUS] = out[g].i[I_MINUS];
        ai[A_NQ2] = out[g].i[I_NQ2]; ai[A_SMMQ] = out[g].i[I_SMMQ]; ai[A_SCLIP] = out[g].i[I_SCLIP]; ai[A_SBQ] = 0;
        for (int f = 0; f < NF; ++f) af[f] = out[g].f[f];
        acc_event(ai, af, rd, e.qpos, 0, true);
        if (rd.misc & M_SMW) w_sm++;
        if (rd.misc & M_NMW) w_nm++;
        out[g].i[I_N] = ai[A_PLUS] + ai[A_MINUS]; out[g].i[I_SMQ] = ai[A_SMQ]; out[g].i[I_SSE] = ai[A_SSE];
        out[g].i[I_PLUS] = ai[A_PLUS]; out[g].i[I_MINUS] = ai[A_MINUS]; out[g].i[I_NQ2] = ai[A_NQ2];
        out[g].i[I_SMMQ] = ai[A_SMMQ]; out[g].i[I_SCLIP] = ai[A_SCLIP]; out[g].i[I_SBQ] = 0;
        for (int f = 0; f < NF; ++f) out[g].f[f] = af[f];
    }
    return na;
}

// The indel side path works on BUCKETS of events: bucket = (16 or 64 consecutive positions, library), the events of a region's reads
// scattered into them in any order (k_indel_scatter).  One lane reduces one bucket: sort by (key, read), fold every key's
// run into out[run start ..] (one slot per event: a key's alleles take the first na slots of its run, the others get
// len = 0), skipping the keys of positions abandoned for a library-less read (bamreadcount.cpp:281-284).
BRC_HD void reduce_indel_bucket(const DevCfg& c, const DevIn& in, const DRead* reads, IndelEv* ev, int n, const uint32_t* unavail,
                                IndelOut* out, uint32_t& w_sm, uint32_t& w_nm) {
    sort_indel_events(ev, n);
    for (int i = 0; i < n;) {
        int j = i + 1;
        while (j < n && ev[j].key_lo == ev[i].key_lo) ++j;
        const uint32_t k = c.Lp == 1 ? ev[i].key_lo : ev[i].key_lo / (uint32_t)c.Lp; const int lib = (int)(ev[i].key_lo - k * (uint32_t)c.Lp);
        int na = 0;
        if (!(c.per_lib && unavail[k] != NONE32))
            na = fold_indel_key(c, in, reads, ev + i, j - i, (int32_t)(c.pos0 + k), lib, out + i, w_sm, w_nm);
        for (int q = i + na; q < j; ++q) out[q].len = 0;
        i = j;
    }
}
// bucket of an event: (tile of the plane index, library); from a key (= plane index * Lp + library) with 32-bit divisions
// positions per indel bucket = 1 << DevCfg.ibucket_shift, chosen per region (indel_bucket_shift): 64 where events are sparse
// (30x: one bucket in four holds an event — fewer buckets to scan), 16 where they are dense (200x with 10 % indel reads: many
// short per-lane sorts instead of a few long ones)
// — and 4 where there is an indel operator for every other position and more (long reads with an operator every ~15 bases at 30x: two events
// per position; 16-position buckets gave every lane 32 events to sort and fold, 13.9 ms of a 54-ms step)
BRC_HD int32_t indel_bucket_shift(uint64_t n_indel_ops, int64_t P, int Lp) {
    if (P <= 0) return 6;
    const uint64_t cells = (uint64_t)P * (uint64_t)Lp;
    return n_indel_ops * 2ull >= cells ? 2 : n_indel_ops * 64ull >= cells ? 4 : 6;
}
BRC_HD int64_t indel_buckets(const DevCfg& c) { return ((c.P + ((int64_t)1 << c.ibucket_shift) - 1) >> c.ibucket_shift) * c.Lp; }
BRC_HD uint32_t indel_bucket_of(const DevCfg& c, uint32_t k, uint32_t lib) { return (k >> c.ibucket_shift) * (uint32_t)c.Lp + lib; }
BRC_HD uint32_t indel_bucket(const DevCfg& c, uint32_t key) {
    if (c.Lp == 1) return key >> c.ibucket_shift;
    const uint32_t k = key / (uint32_t)c.Lp;
    return indel_bucket_of(c, k, key - k * (uint32_t)c.Lp);
}

// ================================================================ device-side text (the "on device" option)
//
// The line pileup_func prints for one position (bamreadcount.cpp:351-416 with operator<<(BasicStat), BasicStat.cpp:110-159),
// written straight from the device's own results: chrom, 1-based position, reference character, depth, and per library present in
// the column its six base buckets (the two slots + the folded third-allele table), the insertion alleles of the position in the
// order std::map<std::string, BasicStat> iterates them (:389-401) and the deletion alleles the position BEFORE it queued for this
// one (:391-396 with IndelQueue::process, IndelQueue.cpp:3-15), whose read counts join the depth column (:415).  Round 6: until then
// a lane left indel buckets, queued deletions and third bases to the host, which rewrote those lines one by one — every line of deep,
// indel-rich data (BASELINE config 5).
// What a lane cannot know is what an EARLIER region left in the host's deletion queues (the reference does not clear them between
// command-line regions, :641-657: a deletion left pending can be printed a second time, or hold back everything queued behind it):
// the lines assume queues that hold nothing but what the position before queued — true inside a region that started from empty
// queues or continues the piece before it (BRC_OPT_CONTINUES_PREVIOUS) —, and the host, which knows its queues, rewrites the indel
// entries of a region that started otherwise (brc_host.cpp: format_device_text).
// Two passes over the same code: lengths (w == nullptr), an exclusive scan, then the bytes.
struct TextCtx {
    const char* chrom; int32_t chrom_len;
    const char* lib_names; const int32_t* lib_off;      // library l's name: lib_names[lib_off[l] .. lib_off[l + 1])
};
// the two side tables of a computed region as a lane finds them
struct TextAux {
    const XAgg* xagg; const uint32_t* xagg_end; const uint32_t* xagg_cnt;      // records of (tile, library) bucket b = tile * Lp + library: [xagg_end[b] - xagg_cnt[b], xagg_end[b]); nullptr: none
    const IndelOut* iout; const uint32_t* ib_end; const uint32_t* ib_cnt;      // reduced indel buckets: slots [ib_end[b] - ib_cnt[b], ib_end[b]) of bucket b = indel_bucket_of(c, k, library); unused slots have len == 0
    const DRead* reads;                                                       // (an insertion allele's bases: the event bytes of its first read)
};
struct TextSink { char* w; uint32_t n; };
BRC_HD void ts_put(TextSink& s, char ch) { if (s.w) s.w[s.n] = ch; ++s.n; }
BRC_HD void ts_bytes(TextSink& s, const char* p, int len) { if (s.w) for (int i = 0; i < len; ++i) s.w[s.n + i] = p[i]; s.n += (uint32_t)len; }
BRC_HD void ts_u64(TextSink& s, uint64_t v) {
    int nd = 1; for (uint64_t t = v; t >= 10; t /= 10) ++nd;
    if (s.w) for (int i = nd - 1; i >= 0; --i) { s.w[s.n + i] = (char)('0' + (int)(v % 10)); v /= 10; }
    s.n += (uint32_t)nd;
}
BRC_HD void ts_u32(TextSink& s, uint32_t v) {
    int nd = 1; for (uint32_t t = v; t >= 10; t /= 10) ++nd;
    if (s.w) for (int i = nd - 1; i >= 0; --i) { s.w[s.n + i] = (char)('0' + (int)(v % 10)); v /= 10; }
    s.n += (uint32_t)nd;
}
// "%.2f" of the exact binary value (see fmt_f2 in brc_host.cpp): v * 100 is exact in double, rint() rounds it half-even
BRC_HD void ts_f2(TextSink& s, float v) {
    const double h = (double)v * 100.0;
    const uint64_t u = (uint64_t)__builtin_rint(h < 0 ? -h : h);
    if (__builtin_signbit(v)) ts_put(s, '-');
    ts_u64(s, u / 100);
    const int fr = (int)(u % 100);
    ts_put(s, '.'); ts_put(s, (char)('0' + fr / 10)); ts_put(s, (char)('0' + fr % 10));
}
// operator<<(ostream&, BasicStat) for a bucket with at least one read (BasicStat.cpp:110-140); is_indel: the base-quality field reads 0.00 (:123)
BRC_HD void ts_stat(TextSink& s, const uint32_t* si, const float* sf, bool is_indel = false) {
    const float c = (float)si[I_N];
    ts_u32(s, si[I_N]); ts_put(s, ':');
    ts_f2(s, (float)si[I_SMQ] / c); ts_put(s, ':');
    if (is_indel) { ts_put(s, '0'); ts_put(s, '.'); ts_put(s, '0'); ts_put(s, '0'); } else ts_f2(s, (float)si[I_SBQ] / c);
    ts_put(s, ':');
    ts_f2(s, (float)si[I_SSE] / c); ts_put(s, ':');
    ts_u32(s, si[I_PLUS]); ts_put(s, ':');
    ts_u32(s, si[I_MINUS]); ts_put(s, ':');
    ts_f2(s, sf[F_SEV] / c); ts_put(s, ':');
    ts_f2(s, sf[F_SNM] / c); ts_put(s, ':');
    ts_f2(s, (float)si[I_SMMQ] / c); ts_put(s, ':');
    ts_u32(s, si[I_NQ2]); ts_put(s, ':');
    if (si[I_NQ2] > 0) ts_f2(s, sf[F_SQ2] / (float)si[I_NQ2]); else { ts_put(s, '0'); ts_put(s, '.'); ts_put(s, '0'); ts_put(s, '0'); }
    ts_put(s, ':');
    ts_f2(s, (float)si[I_SCLIP] / c); ts_put(s, ':');
    ts_f2(s, sf[F_S3P] / c);
}
// character j of an indel bucket's allele text behind its sign (bamreadcount.cpp:324-338): an inserted base as "=ACGTN"[canonical code]
// ('N' past the read's end), a deleted one as the reference's raw character ('N' where there is none)
BRC_HD char allele_char(const DevCfg& c, const DevIn& in, const TextAux& ax, const IndelOut& o, int j) {
    if (o.len > 0) {
        const DRead& rd = ax.reads[o.rep_read];
        const int q = o.rep_qpos + 1 + j;
        const char bases[] = "=ACGTN";
        return q < rd.l_qseq ? bases[base_bucket(in, rd.bq_off, (uint64_t)q)] : 'N';
    }
    const uint32_t rc = ref_at(c, in.ref, (int64_t)o.pos + 1 + j);
    return rc ? (char)rc : 'N';
}
// a < b as std::string compares "+ACG" / "-TT" (the order of std::map<std::string, BasicStat>, :389): '+' before '-', then bytewise, a
// prefix before the longer text.  (Deletions at one position are prefixes of one another: the shorter first.)
BRC_HD bool allele_before(const DevCfg& c, const DevIn& in, const TextAux& ax, const IndelOut& a, const IndelOut& b) {
    if ((a.len > 0) != (b.len > 0)) return a.len > 0;
    const int la = iabs(a.len), lb = iabs(b.len);
    if (a.len < 0) return la < lb;
    const int n = la < lb ? la : lb;
    for (int j = 0; j < n; ++j) { const char ca = allele_char(c, in, ax, a, j), cb = allele_char(c, in, ax, b, j); if (ca != cb) return (unsigned char)ca < (unsigned char)cb; }
    return la < lb;
}
// the slots of the indel bucket that holds key (plane index k, library l)
BRC_HD void indel_slots(const DevCfg& c, const TextAux& ax, int64_t k, int l, uint32_t& s0, uint32_t& s1) {
    s0 = s1 = 0u;
    if (!ax.iout) return;
    const uint32_t b = indel_bucket_of(c, (uint32_t)k, (uint32_t)l);
    s1 = ax.ib_end[b]; s0 = s1 - ax.ib_cnt[b];
}
// The alleles of key (k, l) with want_ins ? len > 0 : len < 0, in allele order: one "\t<allele>:<stat>" each.  Returns the sum of their read counts.
// (No array of them: the next one to print is found by a pass over the bucket's few slots — the smallest allele behind the one printed last.)
BRC_HD uint32_t ts_indels(TextSink& s, const DevCfg& c, const DevIn& in, const TextAux& ax, int64_t k, int l, bool want_ins, bool print) {
    uint32_t s0, s1; indel_slots(c, ax, k, l, s0, s1);
    const int32_t pos = (int32_t)(c.pos0 + k);
    uint32_t nsum = 0; int64_t last = -1;
    for (;;) {
        int64_t best = -1;
        for (uint32_t i = s0; i < s1; ++i) {
            const IndelOut& o = ax.iout[i];
            if (o.len == 0 || o.pos != pos || (o.len > 0) != want_ins) continue;
            if (last >= 0 && ((int64_t)i == last || !allele_before(c, in, ax, ax.iout[last], o))) continue;       // not behind the last one printed
            if (best < 0 || allele_before(c, in, ax, o, ax.iout[best])) best = (int64_t)i;
        }
        if (best < 0) break;
        const IndelOut& o = ax.iout[best];
        nsum += o.i[I_N];
        if (print) {
            ts_put(s, '\t'); ts_put(s, o.len > 0 ? '+' : '-');
            const int n = iabs(o.len);
            for (int j = 0; j < n; ++j) ts_put(s, allele_char(c, in, ax, o, j));
            ts_put(s, ':');
            ts_stat(s, o.i, o.f, true);
        }
        last = best;
    }
    return nsum;
}
// The line of plane index k, or nothing (returns 0) when no read covers the position / the position was abandoned (:281-284).
// Lines are produced for every index, the lead position included (the host needs its shape; it does not print it).
BRC_HD uint32_t text_line(const DevCfg& c, const DevIn& in, const Planes& pl, const TextCtx& t, const TextAux& ax, int64_t k, char* w) {
    if (c.per_lib && pl.unavail[k] != NONE32) return 0u;
    uint32_t tot = 0, depth = 0;
    for (int l = 0; l < c.Lp; ++l) { tot += pl.ncol[(int64_t)l * c.PS + k]; depth += pl.depth[(int64_t)l * c.PS + k]; }
    if (tot == 0) return 0u;
    TextSink s; s.w = w; s.n = 0;
    const int64_t p = (int64_t)c.pos0 + k;
    // the deletions position k - 1 queued are due here, for the libraries present in this column (IndelQueue.cpp:3-15; an abandoned
    // position queued nothing: its keys were never reduced)
    const bool prev = ax.iout != nullptr && k > 0;
    if (prev) for (int l = 0; l < c.Lp; ++l) if (pl.ncol[(int64_t)l * c.PS + k] != 0) depth += ts_indels(s, c, in, ax, k - 1, l, false, false);
    ts_bytes(s, t.chrom, t.chrom_len); ts_put(s, '\t');
    ts_u32(s, (uint32_t)(p + 1)); ts_put(s, '\t');
    { const uint32_t rc = c.has_ref ? ref_at(c, in.ref, p) : 0u; ts_put(s, rc ? (char)rc : 'N'); }                 // :353
    ts_put(s, '\t');
    ts_u32(s, depth);
    const char zero[] = "0:0.00:0.00:0.00:0:0:0.00:0.00:0.00:0:0.00:0.00:0.00";
    const char bases[] = "=ACGTN";
    const int64_t tile = k >> 6;
    for (int l = 0; l < c.Lp; ++l) {
        if (pl.ncol[(int64_t)l * c.PS + k] == 0) continue;                                                             // :286,360
        if (c.per_lib) { ts_put(s, '\t'); ts_bytes(s, t.lib_names + t.lib_off[l], t.lib_off[l + 1] - t.lib_off[l]); ts_put(s, '\t'); ts_put(s, '{'); }
        const uint32_t sid = pl.slotid[(int64_t)l * c.PS + k];
        const uint32_t b0 = sid & 0xffu, b1 = (sid >> 8) & 0xffu;
        // third-allele records of this (tile, library): a bucket's sums sit in ONE place — the slot that names it, or here (also for a
        // bucket a slot names: the events of an N / '=' base never enter the slots)
        uint32_t x0 = 0u, x1 = 0u;
        if (ax.xagg) { const int64_t xb = tile * c.Lp + l; x1 = ax.xagg_end[xb]; x0 = x1 - ax.xagg_cnt[xb]; }
        for (uint32_t b = 0; b < (uint32_t)NBUCKET; ++b) {
            ts_put(s, '\t'); ts_put(s, bases[b]); ts_put(s, ':');
            const int sl = b == b0 ? 0 : (b == b1 ? 1 : -1);
            uint32_t si[NI]; float sf[NF];
            si[I_N] = 0;
            if (sl >= 0) {
                const uint32_t* ip = pl.si + (((int64_t)l * 2 + sl) * NI) * c.PS + k;
                si[I_N] = ip[(int64_t)I_N * c.PS];
                if (si[I_N]) {
                    const float* fp = pl.sf + (((int64_t)l * 2 + sl) * NF) * c.PS + k;
                    for (int f = 0; f < NI; ++f) si[f] = ip[(int64_t)f * c.PS];
                    for (int f = 0; f < NF; ++f) sf[f] = fp[(int64_t)f * c.PS];
                }
            }
            if (!si[I_N]) for (uint32_t x = x0; x < x1; ++x) {
                const XAgg& a = ax.xagg[x];
                if (a.k == (uint32_t)k && (a.lib_b & 0xffu) == b) { for (int f = 0; f < NI; ++f) si[f] = a.i[f]; for (int f = 0; f < NF; ++f) sf[f] = a.f[f]; break; }
            }
            if (si[I_N]) ts_stat(s, si, sf); else ts_bytes(s, zero, (int)sizeof(zero) - 1);
        }
        if (ax.iout) (void)ts_indels(s, c, in, ax, k, l, true, true);                                                  // insertions: printed now (:399)
        if (prev) (void)ts_indels(s, c, in, ax, k - 1, l, false, true);                                                // deletions queued by the position before (:391-396, IndelQueue.cpp:9-12)
        if (c.per_lib) { ts_put(s, '\t'); ts_put(s, '}'); }
    }
    ts_put(s, '\n');
    return s.n;
}

}  // namespace brc
#endif
