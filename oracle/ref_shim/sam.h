/*
 * sam.h — SHIM of the samtools-1.10 legacy API + htslib-1.10 pileup types, just wide enough to compile the
 * reference's own sources unmodified (src/exe/bam-readcount/bamreadcount.cpp, src/lib/bamrc/*.cpp) into
 * oracle/_ref/.  TEST INFRASTRUCTURE ONLY (see oracle/ref_shim/README.md).
 *
 * samtools/htslib are not in the reference checkout (vendor/samtools-1.10.tar.bz2 is a missing blob) and not on
 * this system, so the declarations below are written from the public API those sources use; the implementations
 * (shim_hts.cpp) restate the published behaviour of htslib 1.10 (sam.c: bam_plp_push / bam_plp_next /
 * resolve_cigar2, bam_aux_*; samtools bam_plbuf.c, sam.c legacy wrappers) on top of this repository's own readers.
 */
#ifndef BRC_REF_SHIM_SAM_H
#define BRC_REF_SHIM_SAM_H

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int64_t hts_pos_t;

/* BAM CIGAR operators and flags (SAMv1) */
#define BAM_CMATCH 0
#define BAM_CINS 1
#define BAM_CDEL 2
#define BAM_CREF_SKIP 3
#define BAM_CSOFT_CLIP 4
#define BAM_CHARD_CLIP 5
#define BAM_CPAD 6
#define BAM_CEQUAL 7
#define BAM_CDIFF 8
#define BAM_FPAIRED 1
#define BAM_FPROPER_PAIR 2
#define BAM_FUNMAP 4
#define BAM_FMUNMAP 8
#define BAM_FREVERSE 16
#define BAM_FMREVERSE 32
#define BAM_FREAD1 64
#define BAM_FREAD2 128
#define BAM_FSECONDARY 256
#define BAM_FQCFAIL 512
#define BAM_FDUP 1024
#define BAM_FSUPPLEMENTARY 2048

typedef struct {
    hts_pos_t pos;
    int32_t tid;
    uint16_t bin;
    uint8_t qual;
    uint8_t l_extranul;
    uint16_t flag;
    uint16_t l_qname;
    uint32_t n_cigar;
    int32_t l_qseq;
    int32_t mtid;
    hts_pos_t mpos;
    hts_pos_t isize;
} bam1_core_t;

typedef struct {
    bam1_core_t core;
    uint64_t id;
    uint8_t* data;
    int l_data;
    uint32_t m_data;
} bam1_t;

/* legacy accessor macros (samtools bam.h) */
#define bam1_qname(b) ((char*)((b)->data))
#define bam1_cigar(b) ((uint32_t*)((b)->data + (b)->core.l_qname))
#define bam1_seq(b) ((b)->data + ((b)->core.n_cigar << 2) + (b)->core.l_qname)
#define bam1_qual(b) ((b)->data + ((b)->core.n_cigar << 2) + (b)->core.l_qname + (((b)->core.l_qseq + 1) >> 1))
#define bam1_aux(b) ((b)->data + ((b)->core.n_cigar << 2) + (b)->core.l_qname + (b)->core.l_qseq + (((b)->core.l_qseq + 1) >> 1))
#define bam1_seqi(s, i) ((s)[(i) >> 1] >> ((~(i) & 1) << 2) & 0xf)

extern const unsigned char seq_nt16_table[256];
#define bam_nt16_table seq_nt16_table

uint8_t* bam_aux_get(const bam1_t* b, const char tag[2]);
int64_t bam_aux2i(const uint8_t* s);
int bam_aux_append(bam1_t* b, const char tag[2], char type, int len, const uint8_t* data);

typedef union { void* p; int64_t i; double f; } bam_pileup_cd;
typedef struct {
    bam1_t* b;
    int32_t qpos;
    int indel, level;
    uint32_t is_del : 1, is_head : 1, is_tail : 1, is_refskip : 1, /* reserved */ : 1, aux : 27;
    bam_pileup_cd cd;
} bam_pileup1_t;

struct __bam_plp_t;
typedef struct __bam_plp_t* bam_plp_t;
void bam_plp_set_maxcnt(bam_plp_t iter, int maxcnt);

typedef int (*bam_pileup_f)(uint32_t tid, uint32_t pos, int n, const bam_pileup1_t* pl, void* data);
typedef struct {
    bam_plp_t iter;
    bam_pileup_f func;
    void* data;
} bam_plbuf_t;
bam_plbuf_t* bam_plbuf_init(bam_pileup_f func, void* data);
void bam_plbuf_destroy(bam_plbuf_t* buf);
int bam_plbuf_push(const bam1_t* b, bam_plbuf_t* buf);

/* header + file handles */
typedef struct {
    int32_t n_targets, ignore_sam_err;
    size_t l_text;
    uint32_t* target_len;
    char** target_name;
    char* text;
    void* sdict;
    void* shim; /* brcio::BamHeader* of the shim */
} bam_hdr_t;
typedef bam_hdr_t bam_header_t;

typedef struct htsFile htsFile;
typedef htsFile samFile;
typedef struct hts_idx_t hts_idx_t;
typedef struct {
    samFile* file;
    struct { void* bam; } x;
    bam_hdr_t* header;
    unsigned short is_write : 1;
} samfile_t;

#define HTS_IDX_SAVE_REMOTE 1

samfile_t* samopen(const char* fn, const char* mode, const void* aux);
void samclose(samfile_t* fp);
typedef int (*bam_fetch_f)(const bam1_t* b, void* data);
int samfetch(samfile_t* fp, const hts_idx_t* idx, int tid, int beg, int end, void* data, bam_fetch_f func);
int sampileup(samfile_t* fp, int mask, bam_pileup_f func, void* data);
hts_idx_t* sam_index_load3(htsFile* fp, const char* fn, const char* fnidx, int flags);
void hts_idx_destroy(hts_idx_t* idx);
int hts_set_fai_filename(htsFile* fp, const char* fn_aux);
char* samfaipath(const char* fn_ref);
int bam_parse_region(bam_header_t* header, const char* str, int* ref_id, int* begin, int* end);
const char* bam_get_library(bam_header_t* header, const bam1_t* b);

#ifdef __cplusplus
}
#endif
#endif
