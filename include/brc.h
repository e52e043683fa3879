/*
 * brc.h — C-ABI drop-in boundary of the MI355X pileup/readcount engine.
 *
 * The reference (genome/bam-readcount) has no plugin API: its per-position accumulation sits behind two
 * htslib callbacks and one C++ method.  Each entry point below names the reference seam it replaces
 * (paths relative to the reference tree):
 *
 *   brc_create / brc_destroy   <- pileup_data_t setup + WARN.reset      src/exe/bam-readcount/bamreadcount.cpp:430-432,499,508-511
 *   brc_begin_region           <- d.beg/d.end + load_reference + bam_plbuf_init + bam_plp_set_maxcnt   bamreadcount.cpp:588-592, 644-651
 *   brc_push_reads             <- fetch_func (per-read "Zm" annotation) + bam_plbuf_push               bamreadcount.cpp:114-261
 *   brc_end_region             <- bam_plbuf_push(0,buf) flush + every pileup_func callback             bamreadcount.cpp:265-419, 603-605, 655-656
 *                                 (BasicStat::process_read, src/lib/bamrc/BasicStat.cpp:28-107)
 *   brc_format_region          <- record assembly, IndelQueue::process, operator<<(BasicStat)          bamreadcount.cpp:351-416,
 *                                 src/lib/bamrc/IndelQueue.cpp:3-15, BasicStat.cpp:110-159
 *
 * Conventions: extern "C", opaque handle, plain pointers + sizes, 0 = ok / negative = error
 * (brc_strerror).  No exceptions cross the boundary.  One engine per GPU / rank; one producer thread
 * per engine.  All caller arrays are borrowed only for the duration of the call unless stated.
 *
 * Threads: calls on one engine are not re-entrant, with ONE exception that lets a caller pipeline regions: the host-side
 * assembly of region k (brc_format_region / brc_format_region_parts / brc_format_window, brc_clear_indel_queue) may run on
 * a second thread while the producer thread already runs brc_begin_region, brc_push_reads, brc_upload and brc_compute of
 * region k + 1 — those touch the staging and device buffers only.  brc_fetch_result / brc_end_region of region k + 1
 * overwrite the host result and must wait until region k has been formatted.
 */
#ifndef BRC_H
#define BRC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BRC_ABI_VERSION 1

/* error codes */
#define BRC_OK            0
#define BRC_E_ARG        -1   /* bad argument / call order */
#define BRC_E_NODEVICE   -2   /* no HIP device / kernels unavailable: the engine never falls back to CPU */
#define BRC_E_HIP        -3   /* HIP runtime error (see brc_last_error) */
#define BRC_E_NOMEM      -4
#define BRC_E_LIMIT      -5   /* region / batch exceeds an engine limit: 2^32 - 16 reads, CIGAR operators or read segments per region; reads of
                                 2^22 bases and more; (positions x libraries) and the indel operators of a region below 2^32 when a reference
                                 is given (brc_last_error names the one that was hit) — split the region */

/* number of base buckets per (position, library): "=ACGTN"  (bamreadcount.cpp:34-39) */
#define BRC_NBUCKET 6
/* integer and float accumulator planes per bucket (BasicStat.hpp:12-24) */
#define BRC_NI 9
#define BRC_NF 4

/* integer plane order */
enum {
    BRC_I_N = 0,      /* read_count */
    BRC_I_SMQ,        /* sum_map_qualities */
    BRC_I_SSE,        /* sum_single_ended_map_qualities */
    BRC_I_PLUS,       /* num_plus_strand */
    BRC_I_MINUS,      /* num_minus_strand */
    BRC_I_NQ2,        /* num_q2_reads */
    BRC_I_SMMQ,       /* sum_of_mismatch_qualities */
    BRC_I_SCLIP,      /* sum_of_clipped_lengths */
    BRC_I_SBQ         /* sum_base_qualities */
};
/* float plane order (fp32 running sums in pileup-column order) */
enum {
    BRC_F_SEV = 0,    /* sum_event_location */
    BRC_F_SQ2,        /* sum_q2_distance */
    BRC_F_SNM,        /* sum_number_of_mismatches */
    BRC_F_S3P         /* sum_3p_distance */
};

/* One accumulated bucket, 52 bytes: the 13 BasicStat accumulators (BasicStat.hpp:12-24). */
typedef struct brc_stat {
    uint32_t i[BRC_NI];
    float    f[BRC_NF];
} brc_stat;

/* tag presence bits in brc_read_batch.tags */
#define BRC_TAG_NM 1u
#define BRC_TAG_SM 2u

typedef struct brc_config {
    int32_t abi_version;        /* BRC_ABI_VERSION */
    int32_t min_mapq;           /* -q  (bamreadcount.cpp:438) */
    int32_t min_bq;             /* -b  (:439) */
    int32_t max_cnt;            /* -d  (:440); 0 => default 10000000 */
    int32_t per_lib;            /* -p  (:444) */
    int32_t insertion_centric;  /* -i  (:446) */
    int32_t n_libs;             /* number of library names (per_lib only) */
    const char* const* lib_names; /* bytewise-sorted, distinct library names (std::map order, :273,360); copied; brc_create returns BRC_E_ARG for any other order */
    int32_t device;             /* HIP device ordinal */
    int32_t ref_len_check;      /* 1 in site-list mode: fetch_data_t.ref_len != 0 (:594-600,144-148) */
} brc_config;

/*
 * A batch of decoded alignment records, structure-of-arrays, coordinate-sorted (file order).
 * Field meanings are BAM's (SAMv1 4.2): pos 0-based leftmost; cigar = len<<4|op; seq4 = two bases
 * per byte, high nibble first, each read starting on a byte boundary; qual = phred bytes.
 * Reads with flag UNMAP/SECONDARY/QCFAIL/DUP may be present; the engine drops them exactly where
 * bam_plp_push would (after annotation).
 */
typedef struct brc_read_batch {
    int64_t n_reads;
    const int32_t*  pos;        /* [n] */
    const uint16_t* flag;       /* [n] */
    const uint8_t*  mapq;       /* [n] */
    const int16_t*  lib;        /* [n] index into brc_config.lib_names, -1 = library unavailable; may be NULL when !per_lib */
    const int32_t*  l_qseq;     /* [n] */
    const uint32_t* n_cigar;    /* [n] */
    const uint64_t* cigar_off;  /* [n] index of the read's first op in cigar[] */
    const uint64_t* seq_off;    /* [n] byte offset of the read's first base pair in seq4[] */
    const uint64_t* qual_off;   /* [n] byte offset of the read's first quality in qual[] */
    const int32_t*  nm;         /* [n] NM:i value (valid when tags & BRC_TAG_NM) */
    const int32_t*  sm;         /* [n] SM:i value (valid when tags & BRC_TAG_SM) */
    const uint8_t*  tags;       /* [n] BRC_TAG_* presence bits */
    const uint32_t* cigar;      /* [n_cigar_total] */
    const uint8_t*  seq4;       /* [seq_bytes] */
    const uint8_t*  qual;       /* [qual_bytes] */
    uint64_t n_cigar_total, seq_bytes, qual_bytes;
    const char* const* qname;   /* [n] optional (warning text only); may be NULL */
} brc_read_batch;

/* One indel bucket (LibraryCounts::indel_stats entry, bamreadcount.cpp:47,315-342). */
typedef struct brc_indel {
    int32_t  pos;        /* 0-based reference position of the base BEFORE the indel (the pileup position) */
    int32_t  lib;        /* library index (0 in all-lib mode) */
    int32_t  len;        /* >0 insertion length, <0 deletion length */
    uint32_t rep_read;   /* region-wide index (push order) of the first read carrying this allele */
    int32_t  rep_qpos;   /* qpos of that read at pos; inserted bases are rep_qpos+1 .. rep_qpos+len */
    uint32_t allele_off; /* offset into brc_result.alleles of the allele text ("+ACG" / "-TT"), not NUL-terminated */
    uint32_t allele_len; /* bytes, including the leading sign */
    brc_stat stat;
} brc_indel;

/* warning counters, same order as ReadWarnings::WarningType (ReadWarnings.hpp:13-19) */
enum { BRC_W_SM_MISSING = 0, BRC_W_NM_MISSING, BRC_W_ZM_MISSING, BRC_W_LIB_UNAVAILABLE, BRC_N_WARN };

/*
 * Result of one region, position-major planes ("SoA"): element plane*stride + k is position pos0 + k ("[..][P]" below
 * means P valid elements per plane, planes `stride` elements apart).  The plane window [pos0, pos0+P) is the processing
 * window [max(beg0-1,0), end) intersected with the extent of the pushed reads that pass the flag mask.
 * Position index 0 is the lead position beg0-1 (processed only so that deletions starting there
 * can be reported at beg0, bamreadcount.cpp:269 vs :414); when beg0 == 0 there is no lead position
 * and pos0 == 0.  All arrays are engine-owned host memory, valid until the next begin_region/destroy.
 */
typedef struct brc_result {
    int32_t tid, beg0, end;     /* reporting window [beg0,end) */
    int32_t pos0;               /* reference position of index 0 */
    int64_t n_pos;              /* P */
    int64_t stride;             /* elements between consecutive planes (>= P; the HIP engine pads P to a multiple of 64) */
    int32_t n_lib;              /* Lp: 1 in all-lib mode, n_libs in per-lib mode */
    const uint32_t* ncol;       /* [Lp][P] pileup column entries of that library (pre-filter, incl. deletions/ref-skips): lib_counts[] creation, :286 */
    const uint32_t* depth;      /* [Lp][P] mapq_n contribution (:312) */
    const uint32_t* istat;      /* [Lp][BRC_NBUCKET][BRC_NI][P] */
    const float*    fstat;      /* [Lp][BRC_NBUCKET][BRC_NF][P] */
    const uint32_t* unavail;    /* [P] per-lib mode: region-wide index of the first library-unavailable read in the column, 0xFFFFFFFF if none (:281-284); NULL in all-lib mode */
    const char* refbase;        /* [P] raw reference character printed in column 3 ('N' when no reference / past its end, :353) */
    int64_t n_indel;
    const brc_indel* indel;     /* sorted by (pos, lib, allele text bytewise) = std::map iteration order (:389-401) */
    const char* alleles;        /* allele text arena */
    uint64_t alleles_len;
    uint64_t n_events;          /* pileup base-events inside [beg0,end) (SURVEY 8d unit of work) */
    uint64_t warn[BRC_N_WARN];  /* process_read-level warning counts (BasicStat.cpp:74,85,100) + positions abandoned (:282) */
} brc_result;

/* per-kernel timing of the last brc_compute (HIP events on the engine's stream) */
#define BRC_NKERNEL 8
typedef struct brc_timing {
    float ms[BRC_NKERNEL];          /* see brc_kernel_name() */
    float total_ms;                 /* first launch -> last completion */
} brc_timing;

typedef struct brc_engine brc_engine;

const char* brc_strerror(int code);
const char* brc_last_error(const brc_engine*);
const char* brc_kernel_name(int k);     /* name of timing slot k, NULL if unused */
const char* brc_engine_kind(void);      /* "hip-gfx950" for the product library, "oracle-c" for oracle/ */

/* brc_create: BRC_E_NODEVICE without a HIP device (there is no CPU fallback).  The HIP engine's first launch is a self-check of the
 * one place where it leans on an approximate instruction — the small-integer quotients of reads of another length than the region's
 * modal one (hardware reciprocal + multiply + two FMAs, brc_core.h: div_small) are compared with IEEE division over their whole
 * domain, bit for bit; a device that disagrees anywhere is refused with BRC_E_HIP instead of being trusted. */
int  brc_create(const brc_config* cfg, brc_engine** out);
void brc_destroy(brc_engine*);

/* Engine options (set after brc_create, before the first region).
 *   BRC_OPT_TEXT_ONLY  1: the caller consumes results only through brc_format_region / brc_format_window (the drop-in command
 *                      line: the reference itself only prints, bamreadcount.cpp:351-416).  brc_result.istat / fstat are then
 *                      NULL — the dense 312-byte-per-position planes are never built on the host, the formatter reads the
 *                      engine's compact result (two bucket slots per position + the third-allele list) — and such a result
 *                      must be formatted before the next region of this engine.  Everything else of brc_result is filled. */
#define BRC_OPT_TEXT_ONLY 1
/*   BRC_OPT_EXPECT_READS / BRC_OPT_EXPECT_BASES  sizing hints for the next regions (reads and quality bytes a region is
 *                      expected to push): the pinned staging is allocated once instead of grown batch by batch */
#define BRC_OPT_EXPECT_READS 2
#define BRC_OPT_EXPECT_BASES 3
/*   BRC_OPT_DEVICE_TEXT  1 (with BRC_OPT_TEXT_ONLY and a column-1 name set by brc_set_chrom): the lines of the following regions
 *                      are written on the GPU and downloaded as text — round 6: the WHOLE lines, indel buckets (in the order of the
 *                      reference's std::map, :389-401), the deletions the position before queued (IndelQueue.cpp:3-15) and buckets of a
 *                      third base included.  brc_fetch_result downloads no planes at all (brc_result.ncol / depth / istat / fstat /
 *                      refbase are NULL) and assembles no indel list (n_indel = 0: the lines carry the entries).  What
 *                      brc_format_region[_parts] still does is keep the deletion queues across regions: a region that starts from
 *                      empty queues, or continues the piece before it (BRC_OPT_CONTINUES_PREVIOUS = 1, set before the region is
 *                      formatted), is printed as it arrived; a region that finds something else pending — a deletion an earlier
 *                      command-line region left behind, :641-657 — has the deletion entries of its lines rewritten by
 *                      IndelQueue::process's own rules.  Such a result must be FETCHED after the region before it has been formatted
 *                      (the threads rule above) and formatted, and the returned text consumed, before the next brc_fetch_result of this
 *                      engine; brc_format_window needs planes and refuses it.  Regions whose text could exceed 4 GiB fall back to the
 *                      host formatter by themselves. */
#define BRC_OPT_DEVICE_TEXT 4
/*   BRC_OPT_EXPECT_TEXT  bytes of text a coming region will print: the two pinned text buffers are allocated now (this one
 *                      option may be set from another thread while the first region is being staged — pinning hundreds of
 *                      megabytes takes as long as decoding the first reads) */
#define BRC_OPT_EXPECT_TEXT 5
/*   BRC_OPT_CONTINUES_PREVIOUS  1: the next region continues the previous region of this engine (same contig, its beg0 == the
 *                      previous end) — a caller cutting a long region into abutting pieces.  Its lead position beg0-1 was the
 *                      previous piece's last position and is not processed again: no deletions are queued twice, the queues
 *                      (with whatever an earlier region left pending, which can block them — IndelQueue::process looks at the
 *                      front only) carry over untouched, and brc_region_warnings leaves that position's events out.  Without
 *                      it a caller has to call brc_clear_indel_queue before every piece but the first, which
 *                      is exact only when nothing older is pending.
 *                      2: the piece before this one ran on ANOTHER engine: the queues start empty here and the lead position
 *                      queues its deletions as usual, only brc_region_warnings leaves its events out. */
#define BRC_OPT_CONTINUES_PREVIOUS 6
/*   BRC_OPT_MAX_COUNT  the pileup's max-count exactly as given (-d, :440/:592/:651 bam_plp_set_maxcnt): brc_config.max_cnt
 *   treats values <= 0 as "the default"; the reference hands them to the iterator, whose drop rule `count > maxcnt` then holds
 *   for every read that starts where the previous one did.  Set before the next region's reads are pushed. */
#define BRC_OPT_MAX_COUNT 7
/*   BRC_OPT_FORMAT_THREADS  threads of this engine's host formatter pool (0: its share of the CPUs the process may use, the
 *                      default).  A caller that runs several engines in one process divides the CPUs among them with this
 *                      option — not through the environment: setenv() in a process whose other threads are inside the HIP
 *                      runtime (which reads the environment while it starts) is a data race. */
#define BRC_OPT_FORMAT_THREADS 8
int  brc_set_option(brc_engine*, int option, int64_t value);
/* Target name printed in column 1 of the following regions' lines (BRC_OPT_DEVICE_TEXT: the text is written at
 * brc_fetch_result time, before brc_format_region names the contig); copied. */
int  brc_set_chrom(brc_engine*, const char* chrom);

/* Open the reporting window [beg0,end) on contig tid.  ref = raw FASTA characters of the whole contig
 * (ref[i] = base at 0-based i), borrowed until brc_end_region/brc_fetch_result returns.  ref may be NULL
 * (no -f): then indel alleles are not collected and the reference base prints as 'N' (:315,353). */
int  brc_begin_region(brc_engine*, int32_t tid, int32_t beg0, int32_t end, const char* ref, int64_t ref_len);

/* Append coordinate-sorted reads (may be called repeatedly; batches must be in file order).
 * The engine copies what it needs into pinned staging before returning.  BRC_E_ARG for records the kernels could not index
 * safely: offsets outside the batch arenas, reads out of order, a mapped record's CIGAR whose query-consuming operators (M I S = X) do
 * not add up to l_qseq (an UNMAPPED record with such a CIGAR is taken without it: it never reaches a column), a record without sequence (l_qseq == 0) that is neither unmapped nor secondary / QC-fail / duplicate, a
 * library index outside the configured ones, a mapped record whose CIGAR has an EMPTY M / = / X operator behind the read's last base (the
 * reference reads past the read's qualities there).  Every other M / = / X operator of length zero is accepted (round 6): htslib's
 * resolve_cigar2 steps ONTO such an operator for one column — reported as a match at the operator's query offset, whatever follows seen one
 * column late — and such reads are piled up by that cursor itself, column by column (brc_core.h: cursor_resolve); no aligner writes them. A refused batch ABANDONS the region (its staging is half appended): further
 * brc_push_reads / brc_upload calls fail until the next brc_begin_region. */
int  brc_push_reads(brc_engine*, const brc_read_batch*);

/* Zero-copy feed of the two big arenas (round 5).  brc_push_reads copies every array of a batch into the engine's page-locked staging; two
 * of them — seq4 and qual, 225 of the ~290 bytes of a 150-base read — are only ever uploaded and, rarely, read back on the host (the
 * inserted bases of an indel allele, a warning line's quality test).  A caller that DECODES INTO page-locked memory itself (the BAM / CRAM
 * reader of the command line: the north_star's "decoded read records are batched into pinned host buffers and hipMemcpyAsync'd") hands
 * those two arenas over instead: brc_upload copies them to the device from where they lie.
 *   brc_host_alloc / brc_host_free   page-locked host memory any engine of this process can upload from (hipHostMalloc, portable; usable
 *                      before an engine exists — the first call starts the HIP runtime, which engine creation would do anyway).  NULL when it
 *                      cannot be had.
 *   brc_push_reads_pinned   brc_push_reads, except that batch->seq4 and batch->qual MUST point into brc_host_alloc memory and stay the
 *                      caller's: unchanged and valid until the next brc_begin_region (or brc_destroy) of this engine — the engine reads
 *                      them at brc_upload, brc_fetch_result (allele text) and brc_region_warnings.  Everything else of the batch is copied
 *                      as usual.  A region takes either kind of push, not both (BRC_E_ARG otherwise); results are identical. */
void* brc_host_alloc(size_t bytes);
void  brc_host_free(void* p);
int   brc_push_reads_pinned(brc_engine*, const brc_read_batch*);

/* Split form of brc_end_region, used by the benchmark to keep inputs/outputs resident in HBM:
 *   brc_upload   : staging -> HBM (async on the engine stream, then waits)
 *   brc_compute  : launch the whole device pipeline on the engine stream and wait; repeatable
 *   brc_fetch_result : HBM -> host planes + host-side allele ordering                                  */
int  brc_upload(brc_engine*);
int  brc_compute(brc_engine*, brc_timing* timing /* may be NULL */);
int  brc_fetch_result(brc_engine*, brc_result* out);
/* n passes of brc_compute over the uploaded region, queued back to back on the engine's stream with ONE wait at the end: between
 * two passes the device never waits for the host (a caller that keeps a region resident and runs it repeatedly — the benchmark's
 * timed steps; the state afterwards is that of one brc_compute).  timing (may be NULL): per-kernel times averaged over the
 * passes, from HIP events of every pass; total_ms = the passes' own spans (first launch to last completion), averaged. */
int  brc_compute_n(brc_engine*, int32_t n, brc_timing* timing);

/*
 * A window of the last computed region, as the stand-alone region [beg0, end) would have returned it — without computing again and
 * without the whole region's dense planes on the host: a caller that wants the statistics of a region far larger than 312 bytes per
 * position and library of host memory allow (a 50-Mbp contig with four libraries is 62 GB dense, 23 GB in the engine's compact form
 * on the device) walks it window by window; only the window's compact planes cross PCIe.  [beg0, end) must lie inside the computed
 * region; the planes cover [max(beg0 - 1, 0), end) clipped to the region's planes (index 0 is the window's lead position, see
 * brc_result).  `out` is a complete brc_result with a stride of its own (engine-owned arrays, valid until the next brc_fetch_window
 * / brc_begin_region of this engine; a result of brc_fetch_result stays valid beside it); brc_format_region accepts it — windows
 * formatted one after the other, in order, with BRC_OPT_CONTINUES_PREVIOUS = 1 for every window but the first (a window's lead
 * position was the last position of the window before: it is not processed again), print the region's text; and after
 * brc_clear_indel_queue (option 0) a window prints exactly what the reference prints for a -l line [beg0 + 1, end]
 * (bamreadcount.cpp:574-607).  A window behind the extent of the region's reads comes back without positions.  n_events counts the window's columns inside
 * [beg0, end); warn[] is a whole-region quantity and comes back zero.  Callable any number of times between brc_compute and the
 * next brc_begin_region, before or after brc_fetch_result (not with BRC_OPT_DEVICE_TEXT regions whose planes the text replaced: it
 * reads the device planes, which stay in place).
 */
int  brc_fetch_window(brc_engine*, int32_t beg0, int32_t end, brc_result* out);

/* Forget deletions queued for pos+1 (d.indel_queue_map.clear(), bamreadcount.cpp:605: after every -l line,
 * NOT between command-line regions).  The queue lives in the host-side assembler (brc_format_region). */
int  brc_clear_indel_queue(brc_engine*);

/* upload + compute + fetch_result */
int  brc_end_region(brc_engine*, brc_result* out);

/* Number of pileup base-events / emitted positions of the last compute without downloading planes. */
int  brc_region_counts(brc_engine*, uint64_t* n_events, uint64_t* n_positions);

/* Piece-steps of the last compute, for regions whose tile ranges were compacted (reads averaging more than a dozen CIGAR segments — ONT /
 * CLR-like alignments: see DESIGN.md): `ranged` = the pieces the tiles' contiguous ranges hold (what the pileup kernel would have walked),
 * `walked` = the pieces that touch their tile (what it walked).  Both 0 when the region was not compacted. */
int  brc_region_piece_steps(brc_engine*, uint64_t* ranged, uint64_t* walked);

/*
 * Host-side record assembly for the last fetched region: produces the reference's exact stdout text
 * ("chr\tpos\tref\tdepth\t..." lines, one per emitted position) in an engine-owned buffer that stays
 * valid until the next call on this engine.  chrom = target name; library names come from the config.
 */
int  brc_format_region(brc_engine*, const brc_result*, const char* chrom,
                       const char** text, size_t* text_len);

/* The same text as brc_format_region, handed over as n_parts consecutive pieces — the buffers the formatter's threads wrote,
 * not concatenated: saves a pass over hundreds of megabytes per region when the caller only writes the text out (the
 * reference streams its lines to stdout one by one, bamreadcount.cpp:414-416).  Engine-owned, valid until the next
 * format call on this engine; advances the deletion queues exactly like brc_format_region (call one of the two). */
int  brc_format_region_parts(brc_engine*, const brc_result*, const char* chrom,
                             const char* const** parts, const size_t** part_lens, size_t* n_parts);

/*
 * Site-list planner support: format only the sub-window [vbeg0, vend) of a fetched region (plus its lead position
 * vbeg0-1 for deletions), starting from EMPTY deletion queues (the reference clears them after every -l line,
 * bamreadcount.cpp:605) and printing coordinate pos + 1 - delta.  Lets a caller lay many independent -l windows side
 * by side on one virtual coordinate axis (reads and reference translated by delta), run the device pipeline once, and
 * emit each line's text in file order.  Does not touch the engine's persistent queues.
 */
int  brc_format_window(brc_engine*, const brc_result*, const char* chrom, int32_t vbeg0, int32_t vend, int32_t delta,
                       const char** text, size_t* text_len);
/*
 * ... and the hint that makes such a shared region cheap: the windows the caller is going to format.  The reference piles up
 * nothing outside a -l line — every line is its own fetch + pileup (bamreadcount.cpp:574-607), and pileup_func returns at
 * once for positions outside [beg - 1, end) (:269) — while a window laid on a shared axis brings the whole extent of its
 * reads with it (the annotator needs the reference under every base of a read, :139-174).  With the hint the engine piles
 * up only the positions a window [vbeg0[i] - 1, vend[i]) asks for (per 64-position tile: from the first to the last such
 * position, with the reads that reach them); everything else comes back EMPTY (no column, no depth, all-zero dense planes, no
 * indel buckets: its statistics are not computed at all).  Only brc_format_window / brc_window_warnings of the announced
 * windows are meaningful on such a region; brc_region_counts, brc_result.n_events and brc_result.warn[] then cover those
 * positions only.  Call between brc_begin_region and brc_end_region; n = 0 withdraws the hint, the next
 * brc_begin_region forgets it.
 */
int  brc_region_windows(brc_engine*, const int32_t* vbeg0, const int32_t* vend, int64_t n);

/*
 * The stderr side of the path: what ReadWarnings::warn (src/lib/bamrc/ReadWarnings.hpp:39-50) would be called with while the
 * last computed region was piled up — LIBRARY_UNAVAILABLE from pileup_func (bamreadcount.cpp:281-284), SM / NM tag missing
 * from BasicStat::process_read (BasicStat.cpp:85,100) — and the uncapped "Request for position ..." lines of fetch_func
 * (bamreadcount.cpp:144-148), in the order the reference emits them (read by read, position by position, column order).
 * One event per line, tagged so that the caller applies the -w cap ACROSS regions (the reference's counters are global):
 *     "S\t<read name>\n"  SM_TAG_MISSING      "N\t<read name>\n"  NM_TAG_MISSING      "L\t<read name>\n"  LIBRARY_UNAVAILABLE
 *     "B\t<the complete line fetch_func prints>\n"
 * At most `cap` events of each capped type are listed (cap < 0: all of them — one line per pileup event of an untagged
 * read, exactly what the reference floods stderr with).  Needs brc_read_batch.qname; reads pushed without names are
 * listed as "?".  brc_warnings_text() turns such a stream into the reference's text given the running counters.
 */
int  brc_region_warnings(brc_engine*, const char* chrom /* target name, printed by the "B" lines */, int64_t cap,
                         const char** events, size_t* events_len);
/* The same for the sub-window [vbeg0, vend) of the region (plus its lead position), for callers that laid several windows side
 * by side (see brc_format_window); no "B" lines: such callers keep windows that reach past a contig out of shared regions. */
int  brc_window_warnings(brc_engine*, int32_t vbeg0, int32_t vend, int64_t cap, const char** events, size_t* events_len);
/* counts[BRC_N_WARN]: events seen so far per type (updated); max_per_type: -w.  Appends to an engine-owned buffer that is
 * reset by the call; returns the text ReadWarnings would have written for these events. */
int  brc_warnings_text(brc_engine*, const char* events, size_t events_len, int64_t max_per_type, int64_t* counts,
                       const char** text, size_t* text_len);

#ifdef __cplusplus
}
#endif
#endif /* BRC_H */
