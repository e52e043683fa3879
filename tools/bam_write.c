/* bam_write.c — fast BAM + BAI writer for the end-to-end benchmarks (bench / test infrastructure, not product code).
 *
 * tools/bamio.py's writer costs ~8 us per read in Python; this one lays coordinate-sorted brc_read_batch arrays out in
 * BGZF blocks, deflates the blocks in parallel (OpenMP, zlib) and writes the .bai (bins + 16-kb linear index, SAMv1
 * section 5).  A file is written contig by contig (brc_bamw_open / brc_bamw_add / brc_bamw_close: one call per contig, in
 * @SQ order, so that a multi-contig "genome" never sits in memory as a whole); brc_write_bam is the single-contig form.
 * Aux fields: NM:i / SM:i per the tags bits, RG:Z:rg<k> when n_libs > 1 — with `rgs_per_lib` read groups per library the
 * reads of library l alternate over rg<l * rgs_per_lib + j> (the header text the caller passes maps them back to LB).
 * Same record layout and index conventions as tools/bamio.py (a record never spans two blocks). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

typedef struct { uint64_t* v; int n, cap; } Chunks;
enum { NBIN = 37450 };

static void put32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static void put16(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }

static int reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

/* payload -> one BGZF block in out (room for 65536 + 64 bytes); returns its size */
static size_t bgzf_block(const uint8_t* payload, size_t n, uint8_t* out, int level) {
    z_stream zs; memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    zs.next_in = (Bytef*)payload; zs.avail_in = (uInt)n; zs.next_out = out + 18; zs.avail_out = 65536 + 32;
    deflate(&zs, Z_FINISH);
    const size_t c = zs.total_out; deflateEnd(&zs);
    static const uint8_t hdr[12] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0};
    memcpy(out, hdr, 12); out[12] = 66; out[13] = 67; put16(out + 14, 2); put16(out + 16, (uint32_t)(c + 25));
    put32(out + 18 + c, (uint32_t)crc32(crc32(0L, Z_NULL, 0), payload, (uInt)n)); put32(out + 22 + c, (uint32_t)n);
    return c + 26;
}

typedef struct {
    FILE* f; char* path; int level, block_bytes, n_ref, next_tid;
    uint64_t at;                 /* compressed bytes written so far */
    int64_t name_base;           /* running read number: names stay unique over the contigs */
    int32_t* lens;
    /* per reference: what the .bai needs */
    struct RefIdx { Chunks* bins; uint64_t* lin; uint8_t* lin_set; int64_t nlin, max_lin; } * idx;
} BamW;

void* brc_bamw_open(const char* path, const char* header_text, int32_t n_ref, const char* const* names, const int32_t* lens,
                    int32_t block_bytes, int32_t level) {
    BamW* w = (BamW*)calloc(1, sizeof(BamW));
    w->f = fopen(path, "wb");
    if (!w->f) { free(w); return NULL; }
    w->path = strdup(path); w->level = level; w->block_bytes = block_bytes > 60000 ? 60000 : block_bytes; w->n_ref = n_ref;
    w->lens = (int32_t*)malloc((size_t)n_ref * 4); memcpy(w->lens, lens, (size_t)n_ref * 4);
    w->idx = (struct RefIdx*)calloc((size_t)n_ref, sizeof *w->idx);
    /* header payload, cut into blocks of at most 60000 bytes */
    const size_t lt = strlen(header_text);
    size_t hlen = 4 + 4 + lt + 4;
    for (int i = 0; i < n_ref; ++i) hlen += 4 + strlen(names[i]) + 1 + 4;
    uint8_t* head = (uint8_t*)malloc(hlen); size_t o = 0;
    memcpy(head, "BAM\1", 4); put32(head + 4, (uint32_t)lt); memcpy(head + 8, header_text, lt); o = 8 + lt;
    put32(head + o, (uint32_t)n_ref); o += 4;
    for (int i = 0; i < n_ref; ++i) {
        const size_t ln = strlen(names[i]) + 1;
        put32(head + o, (uint32_t)ln); memcpy(head + o + 4, names[i], ln); put32(head + o + 4 + ln, (uint32_t)lens[i]); o += 8 + ln;
    }
    uint8_t* cb = (uint8_t*)malloc(65536 + 1024);
    for (size_t a = 0; a < hlen; a += 60000) {
        const size_t n = hlen - a < 60000 ? hlen - a : 60000;
        const size_t c = bgzf_block(head + a, n, cb, level);
        fwrite(cb, 1, c, w->f); w->at += c;
    }
    free(cb); free(head);
    return w;
}

/* one contig's coordinate-sorted reads; contigs in ascending tid order (a tid may be skipped: a contig without reads) */
int brc_bamw_add(void* hw, int32_t tid, int64_t n, int32_t n_libs, int32_t rgs_per_lib,
                 const int32_t* pos, const uint16_t* flag, const uint8_t* mapq, const int16_t* lib, const int32_t* l_qseq,
                 const uint32_t* n_cigar, const uint64_t* cigar_off, const uint64_t* seq_off, const uint64_t* qual_off,
                 const int32_t* nm, const int32_t* sm, const uint8_t* tags, const uint32_t* cigar, const uint8_t* seq4, const uint8_t* qual) {
    BamW* w = (BamW*)hw;
    if (tid < w->next_tid || tid >= w->n_ref) return -3;
    w->next_tid = tid + 1;
    if (rgs_per_lib < 1) rgs_per_lib = 1;
    const int block_bytes = w->block_bytes, level = w->level;
    const int64_t nb0 = w->name_base;
    /* ---- pass 1: record sizes, block assignment */
    uint32_t* rsize = (uint32_t*)malloc((size_t)(n + 1) * 4);
    int64_t* blk_first = (int64_t*)malloc((size_t)(n + 2) * 8);    /* first record of every data block */
    int64_t nblk = 0; size_t fill = (size_t)block_bytes + 1;
    for (int64_t i = 0; i < n; ++i) {
        char nmb[32]; const int qn = snprintf(nmb, sizeof nmb, "r%lld", (long long)(nb0 + i)) + 1;
        size_t aux = 0;
        if (tags[i] & 1) aux += 7;
        if (tags[i] & 2) aux += 7;
        if (n_libs > 1 && lib[i] >= 0) aux += 3 + (size_t)snprintf(nmb, sizeof nmb, "rg%d", (int)lib[i] * rgs_per_lib + (int)(i % rgs_per_lib)) + 1;
        rsize[i] = (uint32_t)(36 + qn + 4u * n_cigar[i] + (uint32_t)((l_qseq[i] + 1) / 2) + (uint32_t)l_qseq[i] + aux);
        if (fill > (size_t)block_bytes) { blk_first[nblk++] = i; fill = 0; }
        fill += rsize[i];
    }
    blk_first[nblk] = n;
    /* ---- pass 2: build + deflate the blocks in parallel */
    uint8_t** cdata = (uint8_t**)calloc((size_t)nblk + 1, sizeof(uint8_t*));
    size_t* csize = (size_t*)calloc((size_t)nblk + 1, sizeof(size_t));
    int64_t b;
#pragma omp parallel for schedule(dynamic, 8)
    for (b = 0; b < nblk; ++b) {
        uint8_t* pay = (uint8_t*)malloc(65536 + 1024); size_t o = 0;
        for (int64_t i = blk_first[b]; i < blk_first[b + 1]; ++i) {
            uint8_t* p = pay + o;
            char nmb[32]; const int qn = snprintf(nmb, sizeof nmb, "r%lld", (long long)(nb0 + i)) + 1;
            const uint32_t nc = n_cigar[i]; const int32_t L = l_qseq[i];
            const uint32_t* cg = cigar + cigar_off[i];
            int64_t rl = 0;
            for (uint32_t k = 0; k < nc; ++k) { const uint32_t op = cg[k] & 15u; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += cg[k] >> 4; }
            const int64_t end = pos[i] + ((nc && !(flag[i] & 4)) ? rl : 1);
            put32(p, rsize[i] - 4); put32(p + 4, (uint32_t)tid); put32(p + 8, (uint32_t)pos[i]); p[12] = (uint8_t)qn; p[13] = mapq[i];
            put16(p + 14, (uint32_t)reg2bin(pos[i], end)); put16(p + 16, nc); put16(p + 18, flag[i]); put32(p + 20, (uint32_t)L);
            put32(p + 24, 0xffffffffu); put32(p + 28, 0xffffffffu); put32(p + 32, 0);
            uint8_t* q = p + 36; memcpy(q, nmb, (size_t)qn); q += qn;
            for (uint32_t k = 0; k < nc; ++k) { put32(q, cg[k]); q += 4; }
            memcpy(q, seq4 + seq_off[i], (size_t)((L + 1) / 2)); q += (L + 1) / 2;
            memcpy(q, qual + qual_off[i], (size_t)L); q += L;
            if (tags[i] & 1) { memcpy(q, "NMi", 3); put32(q + 3, (uint32_t)nm[i]); q += 7; }
            if (tags[i] & 2) { memcpy(q, "SMi", 3); put32(q + 3, (uint32_t)sm[i]); q += 7; }
            if (n_libs > 1 && lib[i] >= 0) { memcpy(q, "RGZ", 3); q += 3; q += snprintf((char*)q, 16, "rg%d", (int)lib[i] * rgs_per_lib + (int)(i % rgs_per_lib)) + 1; }
            o += rsize[i];
        }
        cdata[b] = (uint8_t*)malloc(65536 + 1024);
        csize[b] = bgzf_block(pay, o, cdata[b], level);
        free(pay);
    }
    /* ---- file */
    uint64_t* coff = (uint64_t*)malloc((size_t)(nblk + 1) * 8);
    for (int64_t k = 0; k < nblk; ++k) { coff[k] = w->at; fwrite(cdata[k], 1, csize[k], w->f); w->at += csize[k]; free(cdata[k]); }
    /* ---- index of this reference */
    struct RefIdx* ix = &w->idx[tid];
    ix->bins = (Chunks*)calloc(NBIN, sizeof(Chunks));
    ix->nlin = ((int64_t)w->lens[tid] >> 14) + 2;
    ix->lin = (uint64_t*)calloc((size_t)ix->nlin, 8); ix->lin_set = (uint8_t*)calloc((size_t)ix->nlin, 1);
    for (int64_t bb = 0; bb < nblk; ++bb) {
        size_t o = 0;
        for (int64_t i = blk_first[bb]; i < blk_first[bb + 1]; ++i) {
            const uint64_t v0 = (coff[bb] << 16) | o, v1 = (coff[bb] << 16) | (o + rsize[i]);
            o += rsize[i];
            const uint32_t nc = n_cigar[i]; const uint32_t* cg = cigar + cigar_off[i];
            int64_t rl = 0;
            for (uint32_t k = 0; k < nc; ++k) { const uint32_t op = cg[k] & 15u; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += cg[k] >> 4; }
            const int64_t end = pos[i] + ((nc && !(flag[i] & 4)) ? rl : 1);
            Chunks* c = &ix->bins[reg2bin(pos[i], end)];
            if (c->n && c->v[2 * c->n - 1] == v0) c->v[2 * c->n - 1] = v1;
            else { if (c->n == c->cap) { c->cap = c->cap ? 2 * c->cap : 4; c->v = (uint64_t*)realloc(c->v, (size_t)c->cap * 16); } c->v[2 * c->n] = v0; c->v[2 * c->n + 1] = v1; c->n++; }
            for (int64_t win = pos[i] >> 14; win <= (end - 1) >> 14 && win < ix->nlin; ++win) { if (!ix->lin_set[win]) { ix->lin_set[win] = 1; ix->lin[win] = v0; } if (win + 1 > ix->max_lin) ix->max_lin = win + 1; }
        }
    }
    w->name_base += n;
    free(coff); free(cdata); free(csize); free(rsize); free(blk_first);
    return 0;
}

int brc_bamw_close(void* hw) {
    BamW* w = (BamW*)hw;
    { uint8_t eofb[64]; const size_t e = bgzf_block((const uint8_t*)"", 0, eofb, w->level); fwrite(eofb, 1, e, w->f); }
    fclose(w->f);
    char* ipath = (char*)malloc(strlen(w->path) + 8); strcpy(ipath, w->path); strcat(ipath, ".bai");
    FILE* f = fopen(ipath, "wb");
    free(ipath);
    if (!f) return -1;
    uint8_t w8[16];
    fwrite("BAI\1", 1, 4, f); put32(w8, (uint32_t)w->n_ref); fwrite(w8, 1, 4, f);
    for (int r = 0; r < w->n_ref; ++r) {
        struct RefIdx* ix = &w->idx[r];
        int nb = 0;
        if (ix->bins) for (int k = 0; k < NBIN; ++k) if (ix->bins[k].n) ++nb;
        put32(w8, (uint32_t)nb); fwrite(w8, 1, 4, f);
        if (ix->bins) for (int k = 0; k < NBIN; ++k) if (ix->bins[k].n) {
            put32(w8, (uint32_t)k); put32(w8 + 4, (uint32_t)ix->bins[k].n); fwrite(w8, 1, 8, f);
            fwrite(ix->bins[k].v, 8, (size_t)ix->bins[k].n * 2, f);      /* little-endian host */
            free(ix->bins[k].v);
        }
        put32(w8, (uint32_t)ix->max_lin); fwrite(w8, 1, 4, f);
        uint64_t last = 0;
        for (int64_t win = 0; win < ix->max_lin; ++win) { if (ix->lin_set[win]) last = ix->lin[win]; fwrite(&last, 8, 1, f); }
        free(ix->bins); free(ix->lin); free(ix->lin_set);
    }
    fclose(f);
    free(w->idx); free(w->lens); free(w->path); free(w);
    return 0;
}

int brc_write_bam(const char* path, const char* header_text, const char* contig, int32_t contig_len, int64_t n, int32_t n_libs,
                  const int32_t* pos, const uint16_t* flag, const uint8_t* mapq, const int16_t* lib, const int32_t* l_qseq,
                  const uint32_t* n_cigar, const uint64_t* cigar_off, const uint64_t* seq_off, const uint64_t* qual_off,
                  const int32_t* nm, const int32_t* sm, const uint8_t* tags, const uint32_t* cigar, const uint8_t* seq4, const uint8_t* qual,
                  int32_t block_bytes, int32_t level) {
    void* w = brc_bamw_open(path, header_text, 1, &contig, &contig_len, block_bytes, level);
    if (!w) return -1;
    const int rc = brc_bamw_add(w, 0, n, n_libs, 1, pos, flag, mapq, lib, l_qseq, n_cigar, cigar_off, seq_off, qual_off, nm, sm, tags, cigar, seq4, qual);
    const int rc2 = brc_bamw_close(w);
    return rc ? rc : rc2;
}
