/*
 * shim_hts.cpp — implementation of the samtools-1.10 / htslib-1.10 API slice declared in sam.h, header.h and
 * htslib/faidx.h of this directory.  TEST INFRASTRUCTURE ONLY: it exists so that the reference's own sources
 * (bamreadcount.cpp, BasicStat.cpp, IndelQueue*.cpp) can be compiled unmodified into oracle/_ref/ and used to pin the
 * C oracle (oracle/brc_oracle.c).  htslib itself is absent here (vendor/samtools-1.10.tar.bz2 is a missing blob of the
 * reference checkout), so:
 *   - bam_aux_get / bam_aux2i / bam_aux_append follow the BAM aux wire format (SAMv1 4.2.4);
 *   - the pileup iterator restates htslib 1.10 sam.c (bam_plp_push, bam_plp_next, resolve_cigar2, bam_endpos) —
 *     control flow as published (lazy node removal, max_pos gating, max-count rule on mp->cnt), while the per-entry
 *     CIGAR resolution is written STATELESSLY (a pure function of (cigar, pos)) as a second formulation next to
 *     the stateful cursor of oracle/brc_oracle.c;
 *   - samopen / samfetch / sam_index_load3 / fai_* for BAM and FASTA input sit on a SECOND, independent decoder written here
 *     (IndepBam / IndepFasta below: whole-file BGZF inflate with zlib, sequential record scan, no index arithmetic) — so the
 *     differential tests of the drop-in command line against the reference's main() also cover the product's own reader
 *     stack (bam_readcount_amd/csrc/io/bamio.*: BGZF blocks, BAI / CSI bins and linear index, striped fetches, CG-tag CIGARs,
 *     .fai arithmetic) instead of sharing its bugs.  CRAM input still goes through the product's CRAM decoder (cram.o) —
 *     the one reader both sides share; BRC_SHIM_PRODUCT_READER=1 routes BAM / FASTA through the product's readers as well
 *     (A/B of the two decoders).
 */
#include <limits.h>
#include <stdio.h>

#include <map>
#include <string>
#include <vector>

#include "../../bam_readcount_amd/csrc/io/bamio.h"
#include "header.h"
#include "htslib/faidx.h"
#include "sam.h"

/* ---------------------------------------------------------------- tables */

/* IUPAC character -> 4-bit base code ('=' 0, A 1, C 2, G 4, T 8, N and anything else 15, '0'..'3' -> 1,2,4,8) */
extern "C" const unsigned char seq_nt16_table[256] = {
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15,
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15,
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15,
     1,  2,  4,  8, 15, 15, 15, 15, 15, 15, 15, 15, 15,  0, 15, 15,
    15,  1, 14,  2, 13, 15, 15,  4, 11, 15, 15, 12, 15,  3, 15, 15,
    15, 15,  5,  6,  8, 15,  7,  9, 15, 10, 15, 15, 15, 15, 15, 15,
    15,  1, 14,  2, 13, 15, 15,  4, 11, 15, 15, 12, 15,  3, 15, 15,
    15, 15,  5,  6,  8, 15,  7,  9, 15, 10, 15, 15, 15, 15, 15, 15,
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15,
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15,
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15,
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15,
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15,
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15,
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15,
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15};

/* ---------------------------------------------------------------- aux fields */

static int aux_type_size(int t) {
    switch (t) { case 'A': case 'c': case 'C': return 1; case 's': case 'S': return 2; case 'i': case 'I': case 'f': return 4; case 'd': return 8; default: return 0; }
}
static const uint8_t* aux_skip(const uint8_t* s, const uint8_t* end) {
    if (s >= end) return end;
    const int t = *s++;
    if (t == 'Z' || t == 'H') { while (s < end && *s) ++s; return s < end ? s + 1 : end; }
    if (t == 'B') {
        if (s + 5 > end) return end;
        const int sz = aux_type_size(*s); uint32_t n; memcpy(&n, s + 1, 4);
        s += 5 + (size_t)sz * n;
        return s > end ? end : s;
    }
    const int sz = aux_type_size(t);
    if (!sz) return end;
    return s + sz > end ? end : s + sz;
}
extern "C" uint8_t* bam_aux_get(const bam1_t* b, const char tag[2]) {
    const uint8_t* s = bam1_aux(b);
    const uint8_t* end = b->data + b->l_data;
    while (s + 3 <= end) {
        if (s[0] == (uint8_t)tag[0] && s[1] == (uint8_t)tag[1]) return (uint8_t*)s + 2;   // points at the type byte
        s = aux_skip(s + 2, end);
    }
    return 0;
}
extern "C" int64_t bam_aux2i(const uint8_t* s) {
    const int t = *s++;
    switch (t) {
        case 'c': return (int8_t)*s;
        case 'C': return *s;
        case 's': { int16_t v; memcpy(&v, s, 2); return v; }
        case 'S': { uint16_t v; memcpy(&v, s, 2); return v; }
        case 'i': { int32_t v; memcpy(&v, s, 4); return v; }
        case 'I': { uint32_t v; memcpy(&v, s, 4); return v; }
        default: return 0;
    }
}
extern "C" int bam_aux_append(bam1_t* b, const char tag[2], char type, int len, const uint8_t* data) {
    const uint32_t need = (uint32_t)b->l_data + 3u + (uint32_t)len;
    if (need > b->m_data) {
        uint32_t m = b->m_data ? b->m_data : 64; while (m < need) m <<= 1;
        b->data = (uint8_t*)realloc(b->data, m); b->m_data = m;
    }
    b->data[b->l_data] = (uint8_t)tag[0]; b->data[b->l_data + 1] = (uint8_t)tag[1]; b->data[b->l_data + 2] = (uint8_t)type;
    memcpy(b->data + b->l_data + 3, data, (size_t)len);
    b->l_data += 3 + len;
    return 0;
}

/* ---------------------------------------------------------------- pileup iterator */

static inline bool refop(int op) { return op == BAM_CMATCH || op == BAM_CDEL || op == BAM_CREF_SKIP || op == BAM_CEQUAL || op == BAM_CDIFF; }
static inline bool mop(int op) { return op == BAM_CMATCH || op == BAM_CEQUAL || op == BAM_CDIFF; }

static hts_pos_t shim_endpos(const bam1_t* b) {   // bam_endpos: rlen 0 (or unmapped) counts as 1
    hts_pos_t l = 0;
    if (!(b->core.flag & BAM_FUNMAP) && b->core.n_cigar > 0) {
        const uint32_t* c = bam1_cigar(b);
        for (uint32_t k = 0; k < b->core.n_cigar; ++k) if (refop(c[k] & 0xf)) l += c[k] >> 4;
    }
    return b->core.pos + (l ? l : 1);
}

/* resolve_cigar2 (htslib 1.10 sam.c, restated from its published algorithm): every buffered read keeps a cursor — the reference-consuming
 * operator it stands on (k), that operator's reference start (x) and the query offset there (y) — which moves on by ONE such operator when
 * the column has passed the current one.  For CIGARs whose operators all have a positive length this is the same as asking which operator
 * holds the column (what this file did until round 6); on an M / = / X operator of length zero the cursor stands for one column, reports it
 * as a match at the operator's query offset, and sees what follows one column late.  Returns false when the read has no entry in the column. */
struct PlpCursor { int k = -1; hts_pos_t x = 0; int y = 0; };
static bool resolve_cursor(const bam1_t* b, hts_pos_t pos, PlpCursor* s, bam_pileup1_t* p) {
    const uint32_t* c = bam1_cigar(b); const int n = (int)b->core.n_cigar;
    if (s->k == -1) {                                   /* first column of the read */
        p->qpos = 0;
        if (n == 1) { if (mop(c[0] & 0xf)) { s->k = 0; s->x = b->core.pos; s->y = 0; } }
        else {
            int k = 0; s->x = b->core.pos; s->y = 0;
            for (; k < n; ++k) { const int op = c[k] & 0xf; if (refop(op)) break; if (op == BAM_CINS || op == BAM_CSOFT_CLIP) s->y += c[k] >> 4; }
            s->k = k;
        }
        if (s->k < 0 || s->k >= n) return false;        /* htslib asserts here; nothing to stand on */
    } else {
        const int l = c[s->k] >> 4;
        if (pos - s->x >= l) {                          /* the column has left the operator: on to the next reference-consuming one */
            if (s->k + 1 >= n) return false;
            if (mop(c[s->k] & 0xf)) s->y += l;
            s->x += l;
            int k = s->k + 1;
            for (; k < n; ++k) { const int op = c[k] & 0xf; if (refop(op)) break; if (op == BAM_CINS || op == BAM_CSOFT_CLIP) s->y += c[k] >> 4; }
            s->k = k;
            if (s->k >= n) return false;
        }
    }
    const int op = c[s->k] & 0xf, l = c[s->k] >> 4;
    p->is_del = p->is_refskip = 0; p->indel = 0;
    if (s->x + l - 1 == pos && s->k + 1 < n) {          /* the operator's last reference base looks ahead */
        const int op2 = c[s->k + 1] & 0xf, l2 = c[s->k + 1] >> 4;
        if (op2 == BAM_CDEL) p->indel = -l2;
        else if (op2 == BAM_CINS) p->indel = l2;
        else if (op2 == BAM_CPAD && s->k + 2 < n) {
            int l3 = 0;
            for (int j = s->k + 2; j < n; ++j) { const int o = c[j] & 0xf; if (o == BAM_CINS) l3 += c[j] >> 4; else if (refop(o)) break; }
            if (l3 > 0) p->indel = l3;
        }
    }
    if (mop(op)) p->qpos = s->y + (int)(pos - s->x);
    else { p->is_del = 1; p->qpos = s->y; p->is_refskip = (op == BAM_CREF_SKIP); }
    return true;
}

struct PlpNode { bam1_t b; hts_pos_t beg, end; PlpCursor cur; };
struct __bam_plp_t {
    std::vector<PlpNode*> list;          // head..tail (without htslib's spare tail node)
    std::vector<bam_pileup1_t> plp;
    int tid = 0, max_tid = -1; hts_pos_t pos = 0, max_pos = -1;
    int is_eof = 0, maxcnt = 8000;
};
static void node_free(PlpNode* n) { free(n->b.data); delete n; }

extern "C" void bam_plp_set_maxcnt(bam_plp_t iter, int maxcnt) { iter->maxcnt = maxcnt; }

static int plp_push(bam_plp_t it, const bam1_t* b) {
    if (b) {
        if (b->core.tid < 0) return 0;
        if (b->core.flag & BAM_FUNMAP) return 0;   // htslib 1.10: "Skip only unmapped reads here, any additional filtering must be done in iter->func"
        // mp->cnt counts live nodes plus the spare tail node: cnt > maxcnt  <=>  live >= maxcnt
        if (it->tid == b->core.tid && it->pos == b->core.pos && (int)it->list.size() + 1 > it->maxcnt) return 0;
        PlpNode* n = new PlpNode();
        n->b = *b;
        n->b.data = (uint8_t*)malloc((size_t)(b->l_data > 0 ? b->l_data : 1));
        memcpy(n->b.data, b->data, (size_t)b->l_data);
        n->b.m_data = (uint32_t)b->l_data;
        n->beg = b->core.pos; n->end = shim_endpos(b);
        it->max_tid = b->core.tid; it->max_pos = n->beg;
        it->list.push_back(n);
    } else it->is_eof = 1;
    return 0;
}

static const bam_pileup1_t* plp_next(bam_plp_t it, int* tid, int* pos, int* n_out) {
    *n_out = 0;
    if (it->is_eof && it->list.empty()) return 0;
    while (it->is_eof || it->max_tid > it->tid || (it->max_tid == it->tid && it->max_pos > it->pos)) {
        int n_plp = 0; size_t w = 0;
        it->plp.clear();
        for (size_t i = 0; i < it->list.size(); ++i) {
            PlpNode* p = it->list[i];
            if (p->b.core.tid < it->tid || (p->b.core.tid == it->tid && p->end <= it->pos)) { node_free(p); continue; }
            if (p->b.core.tid == it->tid && p->beg <= it->pos) {
                bam_pileup1_t e; memset(&e, 0, sizeof e);
                e.b = &p->b;
                if (resolve_cursor(&p->b, it->pos, &p->cur, &e)) {
                    e.is_head = (it->pos == p->beg); e.is_tail = (it->pos == p->end - 1);
                    it->plp.push_back(e); ++n_plp;
                }
            }
            it->list[w++] = p;
        }
        it->list.resize(w);
        *n_out = n_plp; *tid = it->tid; *pos = (int)it->pos;
        if (!it->list.empty()) {
            PlpNode* h = it->list[0];
            if (it->tid < h->b.core.tid) { it->tid = h->b.core.tid; it->pos = h->beg; }
            else if (it->pos < h->beg) it->pos = h->beg;
            else ++it->pos;
        } else ++it->pos;
        if (n_plp) return it->plp.data();
        if (it->is_eof && it->list.empty()) break;
    }
    *n_out = 0;
    return 0;
}

extern "C" bam_plbuf_t* bam_plbuf_init(bam_pileup_f func, void* data) {
    bam_plbuf_t* buf = (bam_plbuf_t*)calloc(1, sizeof(bam_plbuf_t));
    buf->iter = new __bam_plp_t();
    buf->func = func; buf->data = data;
    return buf;
}
extern "C" void bam_plbuf_destroy(bam_plbuf_t* buf) {
    if (!buf) return;
    for (size_t i = 0; i < buf->iter->list.size(); ++i) node_free(buf->iter->list[i]);
    delete buf->iter;
    free(buf);
}
extern "C" int bam_plbuf_push(const bam1_t* b, bam_plbuf_t* buf) {
    int ret, n_plp, tid, pos;
    const bam_pileup1_t* plp;
    ret = plp_push(buf->iter, b);
    if (ret < 0) return ret;
    while ((plp = plp_next(buf->iter, &tid, &pos, &n_plp)) != 0) buf->func((uint32_t)tid, (uint32_t)pos, n_plp, plp, buf->data);
    return 0;
}

/* ---------------------------------------------------------------- the independent BAM / FASTA decoder */
#include <zlib.h>

static bool product_reader() { static const bool v = getenv("BRC_SHIM_PRODUCT_READER") && atoi(getenv("BRC_SHIM_PRODUCT_READER")) != 0; return v; }

struct IndepRec { int32_t tid, pos, l_seq, mtid, mpos, tlen, end; uint16_t flag, bin; uint32_t n_cigar; uint8_t mapq; uint32_t l_qname; std::vector<uint8_t> data; };

struct IndepBam {
    std::string text; std::vector<std::string> names; std::vector<int32_t> lengths;
    std::vector<IndepRec> recs;
    static uint32_t u32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
    static uint16_t u16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
    // SAMv1 4.1: a BGZF file is a series of gzip members with a 'BC' extra subfield holding the member's size - 1
    static bool inflate_all(const std::vector<uint8_t>& in, std::vector<uint8_t>* out) {
        size_t at = 0;
        while (at < in.size()) {
            if (in.size() - at < 18 || in[at] != 31 || in[at + 1] != 139 || in[at + 2] != 8 || !(in[at + 3] & 4)) return false;
            const uint32_t xlen = u16(&in[at + 10]); size_t x = at + 12; const size_t xend = x + xlen; int64_t bsize = -1;
            if (xend > in.size()) return false;
            while (x + 4 <= xend) { const uint32_t slen = u16(&in[x + 2]); if (in[x] == 'B' && in[x + 1] == 'C' && slen == 2 && x + 6 <= xend) bsize = u16(&in[x + 4]); x += 4 + slen; }
            if (bsize < 0 || at + (size_t)bsize + 1 > in.size() || (size_t)bsize + 1 < 12 + xlen + 8) return false;
            const size_t cbeg = xend, cend = at + (size_t)bsize + 1 - 8;
            const uint32_t isize = u32(&in[cend + 4]);
            const size_t o = out->size(); out->resize(o + isize);
            if (isize) {
                z_stream zs; memset(&zs, 0, sizeof zs);
                if (inflateInit2(&zs, -15) != Z_OK) return false;
                zs.next_in = const_cast<Bytef*>(&in[cbeg]); zs.avail_in = (uInt)(cend - cbeg); zs.next_out = &(*out)[o]; zs.avail_out = isize;
                const int r = inflate(&zs, Z_FINISH); inflateEnd(&zs);
                if (r != Z_STREAM_END || zs.avail_out != 0) return false;
                if (crc32(crc32(0L, Z_NULL, 0), &(*out)[o], isize) != u32(&in[cend])) return false;
            }
            at += (size_t)bsize + 1;
        }
        return true;
    }
    // bam_endpos: pos + reference length of the CIGAR; pos + 1 for unmapped records and records without one
    static int32_t endpos(const IndepRec& r) {
        if ((r.flag & 4) || r.n_cigar == 0) return r.pos + 1;
        int64_t l = 0; const uint8_t* c = r.data.data() + r.l_qname;
        for (uint32_t k = 0; k < r.n_cigar; ++k) { const uint32_t v = u32(c + 4 * k), op = v & 15; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) l += v >> 4; }
        return l ? (int32_t)(r.pos + l) : r.pos + 1;
    }
    // SAMv1 4.2.2: a CIGAR of more than 65535 operators is stored in the CG:B,I tag behind a placeholder <l_seq>S<ref>N;
    // htslib 1.10 (bam_tag2cigar) moves it back into place and drops the tag
    static void splice_cg(IndepRec& r) {
        if (r.n_cigar != 2 || r.tid < 0 || r.pos < 0) return;
        const uint8_t* c = r.data.data() + r.l_qname;
        const uint32_t c0 = u32(c);
        if ((c0 & 15) != 4 || (int32_t)(c0 >> 4) != r.l_seq) return;
        const size_t aux0 = r.l_qname + 4u * r.n_cigar + ((size_t)r.l_seq + 1) / 2 + (size_t)r.l_seq;
        size_t a = aux0;
        while (a + 3 <= r.data.size()) {
            const uint8_t* t = &r.data[a]; const char ty = (char)t[2]; size_t len;
            if (ty == 'A' || ty == 'c' || ty == 'C') len = 1; else if (ty == 's' || ty == 'S') len = 2; else if (ty == 'i' || ty == 'I' || ty == 'f') len = 4;
            else if (ty == 'Z' || ty == 'H') { len = 0; while (a + 3 + len < r.data.size() && t[3 + len]) ++len; ++len; }
            else if (ty == 'B') { if (a + 8 > r.data.size()) return; const char st = (char)t[3]; const uint32_t n = u32(t + 4); const size_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4; len = 5 + (size_t)n * es; }
            else return;
            if (a + 3 + len > r.data.size()) return;
            if (t[0] == 'C' && t[1] == 'G' && ty == 'B' && (char)t[3] == 'I') {
                const uint32_t n = u32(t + 4);
                if (n == 0) return;
                std::vector<uint8_t> d(r.data.begin(), r.data.begin() + r.l_qname);
                d.insert(d.end(), t + 8, t + 8 + 4u * (size_t)n);                                                  // the real CIGAR
                d.insert(d.end(), r.data.begin() + r.l_qname + 4u * r.n_cigar, r.data.begin() + a);                // seq, qual, aux before CG
                d.insert(d.end(), r.data.begin() + a + 3 + len, r.data.end());                                     // aux behind CG
                r.data.swap(d); r.n_cigar = n;
                return;
            }
            a += 3 + len;
        }
    }
    bool open(const char* fn) {
        FILE* f = fopen(fn, "rb");
        if (!f) return false;
        std::vector<uint8_t> raw; uint8_t buf[1 << 16]; size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) raw.insert(raw.end(), buf, buf + n);
        fclose(f);
        std::vector<uint8_t> u;
        if (!inflate_all(raw, &u)) return false;
        if (u.size() < 12 || memcmp(u.data(), "BAM\1", 4) != 0) return false;
        size_t at = 4; const uint32_t l_text = u32(&u[at]); at += 4;
        if (at + l_text + 4 > u.size()) return false;
        text.assign((const char*)&u[at], l_text); { const size_t z = text.find('\0'); if (z != std::string::npos) text.resize(z); }
        at += l_text;
        const uint32_t n_ref = u32(&u[at]); at += 4;
        for (uint32_t i = 0; i < n_ref; ++i) {
            if (at + 4 > u.size()) return false;
            const uint32_t ln = u32(&u[at]); at += 4;
            if (at + ln + 4 > u.size() || ln == 0) return false;
            names.push_back(std::string((const char*)&u[at], ln - 1)); at += ln;
            lengths.push_back((int32_t)u32(&u[at])); at += 4;
        }
        while (at + 4 <= u.size()) {
            const uint32_t bs = u32(&u[at]); at += 4;
            if (bs < 32 || at + bs > u.size()) return false;                 // a truncated record is an error, like the product's reader says
            const uint8_t* p = &u[at]; IndepRec r;
            r.tid = (int32_t)u32(p); r.pos = (int32_t)u32(p + 4); r.l_qname = p[8]; r.mapq = p[9]; r.bin = u16(p + 10); r.n_cigar = u16(p + 12); r.flag = u16(p + 14);
            r.l_seq = (int32_t)u32(p + 16); r.mtid = (int32_t)u32(p + 20); r.mpos = (int32_t)u32(p + 24); r.tlen = (int32_t)u32(p + 28);
            if (r.l_seq < 0 || 32 + (uint64_t)r.l_qname + 4ull * r.n_cigar + ((uint64_t)r.l_seq + 1) / 2 + (uint64_t)r.l_seq > bs) return false;
            r.data.assign(p + 32, p + bs);
            splice_cg(r);
            r.end = endpos(r);
            recs.push_back(r); at += bs;
        }
        return at == u.size();
    }
    // the records an indexed fetch of [beg, end) on tid returns: file order, pos < end, endpos > beg
    template <class F> void fetch(int tid, int beg, int end, F f) const {
        for (size_t i = 0; i < recs.size(); ++i) { const IndepRec& r = recs[i]; if (r.tid == tid && r.pos < end && r.end > beg) f(r); }
    }
};

// FASTA through its .fai (name, length, offset, line bases, line bytes)
struct IndepFasta {
    std::string path; std::map<std::string, std::vector<long long> > ent;
    bool open(const char* fn) {
        path = fn;
        FILE* f = fopen((path + ".fai").c_str(), "r");
        if (!f) return false;
        char name[4096]; long long a, b, c, d;
        while (fscanf(f, "%4095s %lld %lld %lld %lld", name, &a, &b, &c, &d) == 5) { std::vector<long long> v; v.push_back(a); v.push_back(b); v.push_back(c); v.push_back(d); if (!ent.count(name)) ent[name] = v; }
        fclose(f);
        return true;
    }
    bool fetch(const char* name, std::string* seq) const {
        std::map<std::string, std::vector<long long> >::const_iterator it = ent.find(name);
        if (it == ent.end()) return false;
        const long long len = it->second[0], off = it->second[1], lb = it->second[2], lw = it->second[3];
        if (lb <= 0 || lw < lb) return false;
        FILE* f = fopen(path.c_str(), "rb");
        if (!f) return false;
        seq->clear(); seq->reserve((size_t)len);
        if (fseek(f, (long)off, SEEK_SET) != 0) { fclose(f); return false; }
        int ch;
        while ((long long)seq->size() < len && (ch = fgetc(f)) != EOF) if (ch > ' ') seq->push_back((char)ch);     // (faidx keeps every graphic character)
        fclose(f);
        return (long long)seq->size() == len;
    }
};

/* ---------------------------------------------------------------- header, files, index */

struct htsFile {
    std::string path, fai_path;
    bool is_cram = false, opened = false;
    brcio::BamReader bam;            // (BRC_SHIM_PRODUCT_READER=1 only)
    IndepBam ibam;
    brcio::CramReader* cram = 0;
    brcio::Fasta fasta;
};
struct hts_idx_t { brcio::BamIndex idx; bool none = false; };
struct faidx_shim_t { brcio::Fasta fa; IndepFasta ifa; };

static bam_hdr_t* make_header(const brcio::BamHeader& h) {
    bam_hdr_t* o = (bam_hdr_t*)calloc(1, sizeof(bam_hdr_t));
    o->n_targets = (int32_t)h.names.size();
    o->target_len = (uint32_t*)calloc(h.names.size() + 1, 4);
    o->target_name = (char**)calloc(h.names.size() + 1, sizeof(char*));
    for (size_t i = 0; i < h.names.size(); ++i) { o->target_len[i] = (uint32_t)h.lengths[i]; o->target_name[i] = strdup(h.names[i].c_str()); }
    o->l_text = h.text.size(); o->text = strdup(h.text.c_str());
    o->shim = new brcio::BamHeader(h);
    return o;
}
static void free_header(bam_hdr_t* o) {
    if (!o) return;
    for (int i = 0; i < o->n_targets; ++i) free(o->target_name[i]);
    free(o->target_name); free(o->target_len); free(o->text);
    delete (brcio::BamHeader*)o->shim;
    free(o);
}

static bool ensure_open(samfile_t* fp) {
    htsFile* f = fp->file;
    if (f->opened) return true;
    if (f->is_cram) {
        if (f->fai_path.empty()) { fprintf(stderr, "[shim] CRAM input needs a reference\n"); return false; }
        std::string fa = f->fai_path; if (fa.size() > 4 && fa.compare(fa.size() - 4, 4, ".fai") == 0) fa.resize(fa.size() - 4);
        if (!f->fasta.open(fa)) return false;
        f->cram = new brcio::CramReader();
        if (!f->cram->open(f->path, &f->fasta)) return false;
    }
    f->opened = true;
    return true;
}

extern "C" samfile_t* samopen(const char* fn, const char* mode, const void*) {
    if (!fn || !mode || mode[0] != 'r') return 0;
    samfile_t* fp = (samfile_t*)calloc(1, sizeof(samfile_t));
    fp->file = new htsFile();
    fp->file->path = fn;
    fp->file->is_cram = brcio::CramReader::is_cram(fn);
    if (!fp->file->is_cram && !product_reader()) {
        if (!fp->file->ibam.open(fn)) { delete fp->file; free(fp); return 0; }
        fp->file->opened = true;
        brcio::BamHeader h; h.text = fp->file->ibam.text; h.names = fp->file->ibam.names; h.lengths = fp->file->ibam.lengths;    // (a plain record of what was read)
        for (size_t i = 0; i < h.names.size(); ++i) if (!h.name2tid.count(h.names[i])) h.name2tid[h.names[i]] = (int)i;
        fp->header = make_header(h);
    } else if (!fp->file->is_cram) {
        if (!fp->file->bam.open(fn)) { delete fp->file; free(fp); return 0; }
        fp->file->opened = true;
        fp->header = make_header(fp->file->bam.header());
    } else {
        // the header of a CRAM needs no reference: open a header-only view with a throw-away reader
        brcio::CramReader probe;
        if (!probe.open(fn, 0)) { delete fp->file; free(fp); return 0; }
        fp->header = make_header(probe.header());
    }
    return fp;
}
extern "C" void samclose(samfile_t* fp) {
    if (!fp) return;
    free_header(fp->header);
    delete fp->file->cram;
    delete fp->file;
    free(fp);
}
extern "C" int hts_set_fai_filename(htsFile* fp, const char* fn_aux) { fp->fai_path = fn_aux ? fn_aux : ""; return 0; }
extern "C" char* samfaipath(const char* fn_ref) {
    if (!fn_ref) return 0;
    std::string p = std::string(fn_ref) + ".fai";
    FILE* f = fopen(p.c_str(), "rb");
    if (!f) {   // samtools' samfaipath builds the index (fai_build) when it is missing and the FASTA is readable
        brcio::Fasta fa;
        if (!fa.open(fn_ref)) { fprintf(stderr, "[samfaipath] fail to build FASTA index.\n"); return 0; }
    } else fclose(f);
    return strdup(p.c_str());
}
extern "C" hts_idx_t* sam_index_load3(htsFile* fp, const char* fn, const char*, int) {
    hts_idx_t* x = new hts_idx_t();
    if (fp->is_cram) { x->none = true; return x; }      // the CRAM reader scans its own container index
    if (!product_reader()) {
        // the independent reader scans the records: the index only has to EXIST (main() reports its absence, :583,:637)
        const std::string b = fn; x->none = true;
        const char* tries[] = {".bai", ".csi"};
        for (int k = 0; k < 2; ++k) { FILE* f = fopen((b + tries[k]).c_str(), "rb"); if (f) { fclose(f); return x; } }
        if (b.size() > 4 && b.compare(b.size() - 4, 4, ".bam") == 0) { FILE* f = fopen((b.substr(0, b.size() - 4) + ".bai").c_str(), "rb"); if (f) { fclose(f); return x; } }
        delete x; return 0;
    }
    if (!x->idx.load(fn)) { delete x; return 0; }
    return x;
}
extern "C" void hts_idx_destroy(hts_idx_t* idx) { delete idx; }

static void to_bam1(const brcio::BamRecord& r, bam1_t* b) {
    memset(&b->core, 0, sizeof b->core);
    b->core.pos = r.pos; b->core.tid = r.tid; b->core.bin = r.bin; b->core.qual = r.mapq; b->core.flag = r.flag;
    b->core.l_qname = (uint16_t)r.l_qname; b->core.n_cigar = r.n_cigar; b->core.l_qseq = r.l_seq;
    b->core.mtid = r.mtid; b->core.mpos = r.mpos; b->core.isize = r.tlen;
    if (b->m_data < r.data.size() + 64) { b->m_data = (uint32_t)r.data.size() + 64; b->data = (uint8_t*)realloc(b->data, b->m_data); }
    memcpy(b->data, r.data.data(), r.data.size());
    b->l_data = (int)r.data.size();
}

extern "C" int samfetch(samfile_t* fp, const hts_idx_t* idx, int tid, int beg, int end, void* data, bam_fetch_f func) {
    if (!ensure_open(fp)) return -1;
    bam1_t b; memset(&b, 0, sizeof b);
    auto cb = [&](const brcio::BamRecord& r) { to_bam1(r, &b); func(&b, data); };
    bool ok = true;
    if (fp->file->is_cram) ok = fp->file->cram->fetch(tid, beg, end, cb);
    else if (product_reader()) ok = fp->file->bam.fetch(idx->idx, tid, beg, end, cb);
    else fp->file->ibam.fetch(tid, beg < 0 ? 0 : beg, end, [&](const IndepRec& r) {
        memset(&b.core, 0, sizeof b.core);
        b.core.pos = r.pos; b.core.tid = r.tid; b.core.bin = r.bin; b.core.qual = r.mapq; b.core.flag = r.flag;
        b.core.l_qname = (uint16_t)r.l_qname; b.core.n_cigar = r.n_cigar; b.core.l_qseq = r.l_seq; b.core.mtid = r.mtid; b.core.mpos = r.mpos; b.core.isize = r.tlen;
        if (b.m_data < r.data.size() + 64) { b.m_data = (uint32_t)r.data.size() + 64; b.data = (uint8_t*)realloc(b.data, b.m_data); }
        memcpy(b.data, r.data.data(), r.data.size()); b.l_data = (int)r.data.size();
        func(&b, data);
    });
    free(b.data);
    return ok ? 0 : -1;
}
extern "C" int sampileup(samfile_t*, int, bam_pileup_f, void*) {
    fprintf(stderr, "[shim] whole-file mode (sampileup) is not provided: the reference itself marks it broken (bamreadcount.cpp:624)\n");
    return -1;
}

/* htslib-1.10 hts_parse_decimal (hts.c), with HTS_PARSE_THOUSANDS_SEP: white space, sign, digits and commas, fraction,
 * exponent or k/M/G suffix; hts_log warnings (stderr at the default log level) for a discarded fraction and, when the caller
 * does not take the end pointer, for characters left over. */
static long long shim_parse_decimal(const char* str, char** strend) {
    long long n = 0; int decimals = 0, e = 0, lost = 0; char sign = '+', esign = '+';
    while (isspace((unsigned char)*str)) str++;
    const char* s = str;
    if (*s == '+' || *s == '-') sign = *s++;
    while (*s) {
        if (isdigit((unsigned char)*s)) { const int d = *s++ - '0'; n = n > (LLONG_MAX - d) / 10 ? LLONG_MAX : 10 * n + d; }
        else if (*s == ',') s++;
        else break;
    }
    if (*s == '.') { s++; while (isdigit((unsigned char)*s)) { const int d = *s++ - '0'; decimals++; n = n > (LLONG_MAX - d) / 10 ? LLONG_MAX : 10 * n + d; } }
    switch (*s) {
        case 'e': case 'E':
            s++; if (*s == '+' || *s == '-') esign = *s++;
            while (isdigit((unsigned char)*s)) e = 10 * e + (*s++ - '0');
            if (esign == '-') e = -e;
            break;
        case 'k': case 'K': e += 3; s++; break;
        case 'm': case 'M': e += 6; s++; break;
        case 'g': case 'G': e += 9; s++; break;
    }
    e -= decimals;
    while (e > 0) n *= 10, e--;
    while (e < 0) lost += (int)(n % 10), n /= 10, e++;
    if (lost > 0) fprintf(stderr, "[W::hts_parse_decimal] Discarding fractional part of %.*s\n", (int)(s - str), str);
    if (strend) *strend = (char*)s;
    else if (*s) fprintf(stderr, "[W::hts_parse_decimal] Ignoring unknown characters after %.*s[%s]\n", (int)(s - str), str, s);
    return sign == '+' ? n : -n;
}

/* samtools-1.10 legacy bam_parse_region (bam.c) over htslib-1.10 hts_parse_reg / hts_parse_reg64: the name ends at the last
 * colon; "chr" covers [0, INT_MAX), "chr:beg" runs to the end; an interval that does not parse, is empty or lies past
 * INT_MAX makes the whole string the name. */
extern "C" int bam_parse_region(bam_header_t* header, const char* str, int* ref_id, int* begin, int* end) {
    const brcio::BamHeader* h = (const brcio::BamHeader*)header->shim;
    const char* name_lim = 0;
    long long b = 0, e = 0;
    const char* colon = strrchr(str, ':');
    if (!colon) { b = 0; e = LLONG_MAX; name_lim = str + strlen(str); }
    else {
        char* hyphen = 0;
        b = shim_parse_decimal(colon + 1, &hyphen) - 1;
        if (b < 0) b = 0;
        name_lim = colon;
        if (*hyphen == '\0') e = LLONG_MAX;
        else if (*hyphen == '-') e = shim_parse_decimal(hyphen + 1, 0);
        else name_lim = 0;
        if (name_lim && b >= e) name_lim = 0;
    }
    if (b > INT_MAX) { fprintf(stderr, "[E::hts_parse_reg] Position %lld too large\n", b); name_lim = 0; }
    else if (e > INT_MAX) {
        if (e == LLONG_MAX) e = INT_MAX;
        else { fprintf(stderr, "[E::hts_parse_reg] Position %lld too large\n", e); name_lim = 0; }
    }
    *begin = (int)b; *end = (int)e;
    std::map<std::string, int>::const_iterator it;
    if (name_lim) it = h->name2tid.find(std::string(str, name_lim));
    else { it = h->name2tid.find(str); *begin = 0; *end = INT_MAX; }
    *ref_id = it == h->name2tid.end() ? -1 : it->second;
    if (*ref_id == -1) return -1;
    return *begin <= *end ? 0 : -1;
}

/* samtools-1.10 legacy bam_get_library (bam.c), restated: scan the header TEXT for the first @RG line that has an ID and an
 * LB and whose ID is the read's RG:Z — the last "ID:" / "LB:" after a tab count, the ID must be followed by a tab, the LB is
 * copied (at most 1023 bytes) into a static buffer */
extern "C" const char* bam_get_library(bam_header_t* header, const bam1_t* b) {
    const char* rg = (const char*)bam_aux_get(b, "RG");
    const char* cp = header->text;
    if (!rg || !cp) return 0;
    rg++;
    while (*cp) {
        const char *ID = 0, *LB = 0;
        char last = '\t';
        if (strncmp(cp, "@RG", 3) != 0) {
            while (*cp && *cp != '\n') cp++;
            if (*cp) cp++;
            continue;
        }
        cp += 4;
        while (*cp && *cp != '\n') {
            if (last == '\t') {
                if (strncmp(cp, "LB:", 3) == 0) LB = cp + 3;
                else if (strncmp(cp, "ID:", 3) == 0) ID = cp + 3;
            }
            last = *cp++;
        }
        if (!ID || !LB) continue;
        if (strncmp(rg, ID, strlen(rg)) != 0 || ID[strlen(rg)] != '\t') continue;
        static char LB_text[1024];
        for (cp = LB; *cp && *cp != '\t' && *cp != '\n'; cp++) {}
        const size_t n = (size_t)(cp - LB) < 1023 ? (size_t)(cp - LB) : 1023;
        strncpy(LB_text, LB, n); LB_text[n] = 0;
        return LB_text;
    }
    return 0;
}

/* sam_hdr_parse, reduced to what find_library_names walks: every @RG line as a linked list of its tags in file order */
extern "C" sam_hdr_t* sam_hdr_parse(size_t l_text, const char* text) {
    sam_hdr_t* hd = (sam_hdr_t*)calloc(1, sizeof(sam_hdr_t));
    hd->hrecs = (sam_hrecs_t*)calloc(1, sizeof(sam_hrecs_t));
    std::vector<sam_hrec_rg_t> rgs;
    std::string all(text ? text : "", text ? l_text : 0);
    size_t p = 0;
    while (p < all.size()) {
        size_t e = all.find('\n', p); if (e == std::string::npos) e = all.size();
        std::string line = all.substr(p, e - p); p = e + 1;
        if (!line.empty() && line[line.size() - 1] == '\r') line.resize(line.size() - 1);
        if (line.compare(0, 3, "@RG") != 0) continue;
        sam_hrec_type_t* ty = (sam_hrec_type_t*)calloc(1, sizeof(sam_hrec_type_t));
        sam_hrec_tag_t* last = 0;
        size_t q = 3;
        while (q < line.size()) {
            if (line[q] == '\t') { ++q; continue; }
            size_t t = line.find('\t', q); if (t == std::string::npos) t = line.size();
            sam_hrec_tag_t* tag = (sam_hrec_tag_t*)calloc(1, sizeof(sam_hrec_tag_t));
            char* s = strdup(line.substr(q, t - q).c_str());
            tag->str = s; tag->len = (int)(t - q);
            if (last) last->next = tag; else ty->tag = tag;
            last = tag; q = t;
        }
        if (!ty->tag) { free(ty); continue; }
        sam_hrec_rg_t rg; memset(&rg, 0, sizeof rg); rg.ty = ty; rg.id = (int)rgs.size();
        rgs.push_back(rg);
    }
    hd->hrecs->nrg = (int)rgs.size();
    hd->hrecs->rg = (sam_hrec_rg_t*)calloc(rgs.size() + 1, sizeof(sam_hrec_rg_t));
    for (size_t i = 0; i < rgs.size(); ++i) hd->hrecs->rg[i] = rgs[i];
    return hd;   // the reference never frees it (bamreadcount.cpp:94)
}

/* ---------------------------------------------------------------- faidx */
extern "C" faidx_t* fai_load(const char* fn) {
    faidx_shim_t* f = new faidx_shim_t();
    if (product_reader() ? !f->fa.open(fn) : !f->ifa.open(fn)) { delete f; return 0; }
    return f;
}
extern "C" void fai_destroy(faidx_t* fai) { delete fai; }
extern "C" char* fai_fetch(const faidx_t* fai, const char* reg, int* len) {
    std::string seq;
    if (product_reader() ? !const_cast<faidx_shim_t*>(fai)->fa.fetch(reg, &seq) : !fai->ifa.fetch(reg, &seq)) { *len = -2; return 0; }
    char* s = (char*)malloc(seq.size() + 1);
    memcpy(s, seq.data(), seq.size()); s[seq.size()] = 0;
    *len = (int)seq.size();
    return s;
}
