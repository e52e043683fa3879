// cram.cpp — minimal CRAM 3.0 reader, written from the public CRAM 3.0 specification; no htslib code.
//
// Scope: what reference-based CRAMs written by `samtools view -C` with gzip (or raw) blocks need — ITF8/LTF8, container
// / block / slice structure, the compression header (preservation map incl. substitution matrix and tag dictionary,
// data-series and tag encodings), the codecs EXTERNAL, HUFFMAN (canonical codes from the core bit stream, incl. the
// zero-length single-symbol form), BYTE_ARRAY_LEN, BYTE_ARRAY_STOP and BETA, and read reconstruction from the reference
// plus the standard feature codes (X I i D N S H P B b Q q).  Block compression: raw, gzip, rANS 4x8 order 0 and 1 (the
// codec samtools uses for qualities and most byte series; decoder written from the specification's description of the
// frequency tables and the four interleaved states), bzip2 and lzma (through the system's libbz2 / liblzma, looked up at
// run time: the image has the shared objects but no headers); every block's CRC32 is checked.  Integer codecs GAMMA and
// SUBEXP are decoded too; GOLOMB / GOLOMB_RICE (no known writer) are reported as unsupported.  Records come out as
// BAM-layout BamRecords (bamio.h) so the CLI's batcher is unchanged — mapped ones with the NM tag htslib's decoder generates
// when none is stored.  The reference of a slice comes from the FASTA, from the slice's embedded reference block, or from
// nowhere (RR = 0).  Region queries use the .crai when there is one, else one walk over the container headers (ref id,
// start, span; CRC32 checked).  Everything a file says about lengths and positions is bounded before it is used: a block
// with a valid CRC32 and damaged contents ends in an error, not in memory it does not own (tools/fuzz/cram_contents.py).
#include <dlfcn.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <map>

#include "bamio.h"

namespace brcio {

namespace {

struct Cur { const uint8_t* p; const uint8_t* e; bool bad = false;
    uint8_t u8() { if (p >= e) { bad = true; return 0; } return *p++; }
    int32_t itf8() {
        const uint8_t v = u8();
        if (v < 0x80) return v;
        if (v < 0xc0) return ((v & 0x3f) << 8) | u8();
        if (v < 0xe0) { const int a = u8(), b = u8(); return ((v & 0x1f) << 16) | (a << 8) | b; }
        if (v < 0xf0) { const int a = u8(), b = u8(), c = u8(); return ((v & 0x0f) << 24) | (a << 16) | (b << 8) | c; }
        const int a = u8(), b = u8(), c = u8(), d = u8();
        return (int32_t)(((uint32_t)(v & 0x0f) << 28) | ((uint32_t)a << 20) | ((uint32_t)b << 12) | ((uint32_t)c << 4) | (uint32_t)(d & 0x0f));
    }
    int64_t ltf8() {
        const uint8_t v = u8();
        int n = 0; for (uint8_t m = v; m & 0x80; m = (uint8_t)(m << 1)) ++n;
        if (n == 0) return v;
        int64_t val = n < 8 ? (v & (0xff >> (n + 1))) : 0;
        for (int k = 0; k < n; ++k) val = (val << 8) | u8();
        return val;
    }
    int32_t i32() { uint32_t v = 0; for (int k = 0; k < 4; ++k) v |= (uint32_t)u8() << (8 * k); return (int32_t)v; }
};

struct Block { int method = 0, type = 0, id = 0; std::vector<uint8_t> data; size_t pos = 0; };

struct Enc {            // one data-series / tag encoding
    int codec = 0;      // 0 NULL 1 EXTERNAL 3 HUFFMAN 4 BYTE_ARRAY_LEN 5 BYTE_ARRAY_STOP 6 BETA
    int ext_id = -1;    // EXTERNAL / BYTE_ARRAY_STOP
    uint8_t stop = 0;
    std::vector<int32_t> sym, len;   // HUFFMAN
    std::vector<uint32_t> code;      // canonical codes matching sym/len (sorted)
    int beta_off = 0, beta_bits = 0;     // BETA; GAMMA: offset; SUBEXP: offset, k
    std::vector<Enc> sub;            // BYTE_ARRAY_LEN: [0] lengths, [1] values
    Block* blk = nullptr;            // EXTERNAL / BYTE_ARRAY_STOP: the slice's block with this content id (bound once per slice: bind_blocks)
};

struct BitReader { const uint8_t* p = nullptr; size_t n = 0, bit = 0;
    int get() { if ((bit >> 3) >= n) return 0; const int b = (p[bit >> 3] >> (7 - (bit & 7))) & 1; ++bit; return b; }
    uint32_t bits(int k) { uint32_t v = 0; for (int i = 0; i < k && i < 32; ++i) v = (v << 1) | (uint32_t)get(); return v; }
};

bool parse_enc(Cur& c, Enc* e, std::string* err) {
    e->codec = c.itf8();
    const int32_t plen = c.itf8();
    if (c.bad || plen < 0 || plen > c.e - c.p) { *err = "truncated CRAM encoding"; return false; }
    Cur q; q.p = c.p; q.e = c.p + plen; c.p += plen;
    switch (e->codec) {
        case 0: return true;
        case 1: e->ext_id = q.itf8(); return !q.bad;
        case 3: {
            const int na = q.itf8(); for (int i = 0; i < na && !q.bad; ++i) e->sym.push_back(q.itf8());
            const int nl = q.itf8(); for (int i = 0; i < nl && !q.bad; ++i) e->len.push_back(q.itf8());
            if (q.bad || na != nl || na < 0) { *err = "bad HUFFMAN encoding"; return false; }
            for (int32_t l : e->len) if (l < 0 || l > 32) { *err = "bad HUFFMAN code length"; return false; }
            // canonical code assignment: sort by (length, symbol)
            std::vector<int> ord((size_t)na); for (int i = 0; i < na; ++i) ord[(size_t)i] = i;
            std::sort(ord.begin(), ord.end(), [&](int a, int b) { return e->len[(size_t)a] != e->len[(size_t)b] ? e->len[(size_t)a] < e->len[(size_t)b] : e->sym[(size_t)a] < e->sym[(size_t)b]; });
            std::vector<int32_t> s2, l2; for (int i : ord) { s2.push_back(e->sym[(size_t)i]); l2.push_back(e->len[(size_t)i]); }
            e->sym = s2; e->len = l2; e->code.assign((size_t)na, 0);
            uint32_t code = 0; int prev = na ? e->len[0] : 0;
            for (int i = 0; i < na; ++i) { const int sh = e->len[(size_t)i] - prev; code = sh >= 32 ? 0u : code << sh; prev = e->len[(size_t)i]; e->code[(size_t)i] = code++; }
            return true;
        }
        case 4: { e->sub.resize(2); return parse_enc(q, &e->sub[0], err) && parse_enc(q, &e->sub[1], err); }
        case 5: e->stop = q.u8(); e->ext_id = q.itf8(); return !q.bad;
        case 6: e->beta_off = q.itf8(); e->beta_bits = q.itf8(); if (q.bad || e->beta_bits < 0 || e->beta_bits > 32) { *err = "bad BETA encoding"; return false; } return true;
        case 7: e->beta_off = q.itf8(); e->beta_bits = q.itf8(); if (q.bad || e->beta_bits < 0 || e->beta_bits > 30) { *err = "bad SUBEXP encoding"; return false; } return true;     // SUBEXP: offset, k
        case 9: e->beta_off = q.itf8(); return !q.bad;                            // GAMMA: offset
        default: *err = "CRAM codec " + std::to_string(e->codec) + " not supported by the minimal reader"; return false;
    }
}

// ---- rANS 4x8 (block method 4).  Stream: order u8, compressed size u32, uncompressed size u32, frequency table(s),
// four little-endian 32-bit states, renormalisation bytes.  Frequencies are 12-bit (sum 4096 per table); a state x yields
// the symbol whose cumulative range holds x & 4095 and continues as F * (x >> 12) + (x & 4095) - C, refilled bytewise
// while below 2^23.
struct RansTab { uint16_t F[256], C[256]; uint8_t R[4096]; bool used = false; };

// one frequency table: symbols ascending, runs of consecutive symbols announced by (symbol, run length)
bool rans_read_table(Cur& c, RansTab* t, bool order1) {
    memset(t->F, 0, sizeof t->F); memset(t->C, 0, sizeof t->C); memset(t->R, 0, sizeof t->R); t->used = true;
    int rle = 0, x = 0; int j = c.u8();
    do {
        int f = c.u8();
        if (f >= 128) f = ((f & 127) << 8) | c.u8();
        if (f == 0 && order1) f = 4096;
        if (c.bad || x + f > 4096) return false;
        t->F[j] = (uint16_t)f; t->C[j] = (uint16_t)x;
        memset(t->R + x, j, (size_t)f); x += f;
        if (!rle && c.p < c.e && j + 1 == *c.p) { j = c.u8(); rle = c.u8(); }
        else if (rle) { --rle; ++j; if (j > 255) return false; }
        else j = c.u8();
    } while (j && !c.bad);
    return !c.bad;
}

inline bool rans_step(uint32_t* x, const RansTab& t, Cur& c, uint8_t* sym) {
    const uint32_t m = *x & 4095u; const uint8_t s = t.R[m]; *sym = s;
    if (!t.F[s]) return false;                          // a slot no symbol owns: corrupt stream
    *x = t.F[s] * (*x >> 12) + m - t.C[s];
    while (*x < (1u << 23)) { if (c.p >= c.e) { *x <<= 8; c.bad = true; break; } *x = (*x << 8) | *c.p++; }
    return true;
}

bool rans_decode(const uint8_t* in, size_t n, size_t expect, std::vector<uint8_t>* out, std::string* err) {
    Cur c; c.p = in; c.e = in + n;
    const int order = c.u8(); const uint32_t csz = (uint32_t)c.i32(), usz = (uint32_t)c.i32();
    if (c.bad || order > 1 || (size_t)csz + 9 != n || usz != expect) { *err = "bad CRAM rANS block header"; return false; }
    out->assign(usz, 0);
    if (!usz) return true;
    uint32_t R[4];
    if (order == 0) {
        RansTab t;
        if (!rans_read_table(c, &t, false)) { *err = "bad CRAM rANS frequency table"; return false; }
        for (int k = 0; k < 4; ++k) R[k] = (uint32_t)c.i32();
        if (c.bad) { *err = "truncated CRAM rANS block"; return false; }
        // symbol i comes from state i & 3; the last (usz & 3) symbols are read without advancing their states
        const size_t full = usz & ~(size_t)3;
        for (size_t i = 0; i < full; ++i) if (!rans_step(&R[i & 3], t, c, &(*out)[i])) { *err = "corrupt CRAM rANS stream"; return false; }
        for (size_t i = full; i < usz; ++i) (*out)[i] = t.R[R[i & 3] & 4095u];
        return true;
    }
    std::vector<RansTab> tabs(256);
    {   // the contexts come as an outer table of the same (symbol, run) form
        int rle = 0; int i = c.u8();
        do {
            if (!rans_read_table(c, &tabs[(size_t)i], true)) { *err = "bad CRAM rANS frequency table"; return false; }
            if (!rle && c.p < c.e && i + 1 == *c.p) { i = c.u8(); rle = c.u8(); }
            else if (rle) { --rle; ++i; if (i > 255) { *err = "bad CRAM rANS frequency table"; return false; } }
            else i = c.u8();
        } while (i && !c.bad);
    }
    for (int k = 0; k < 4; ++k) R[k] = (uint32_t)c.i32();
    if (c.bad) { *err = "truncated CRAM rANS block"; return false; }
    // four quarter streams, each with its own state and its own previous symbol (0 at the start); the fourth also
    // carries what is left after 4 * (usz / 4)
    const size_t q = usz >> 2; size_t at[4] = {0, q, 2 * q, 3 * q}; uint8_t last[4] = {0, 0, 0, 0};
    for (size_t t = 0; t < q; ++t)
        for (int k = 0; k < 4; ++k) {
            const RansTab& tb = tabs[last[k]];
            if (!tb.used || !rans_step(&R[k], tb, c, &(*out)[at[k]])) { *err = "corrupt CRAM rANS stream"; return false; }
            last[k] = (*out)[at[k]++];
        }
    for (; at[3] < usz; ++at[3]) {
        const RansTab& tb = tabs[last[3]];
        if (!tb.used || !rans_step(&R[3], tb, c, &(*out)[at[3]])) { *err = "corrupt CRAM rANS stream"; return false; }
        last[3] = (*out)[at[3]];
    }
    return true;
}

// ---- bzip2 / lzma blocks through the system libraries (no headers in this image: prototypes restated from their
// public API, resolved with dlopen on first use)
bool bz2_decode(const uint8_t* in, size_t n, std::vector<uint8_t>* out, std::string* err) {
    typedef int (*Fn)(char*, unsigned*, char*, unsigned, int, int);
    // (looked up once, by whichever thread comes first — several readers decode stripes of one piece side by side: a function-local
    // static's initialisation is the lock)
    static const Fn fn = []() -> Fn { void* h = dlopen("libbz2.so.1.0", RTLD_NOW); if (!h) h = dlopen("libbz2.so.1", RTLD_NOW); if (!h) h = dlopen("libbz2.so", RTLD_NOW);
                                      return h ? (Fn)dlsym(h, "BZ2_bzBuffToBuffDecompress") : (Fn)nullptr; }();
    if (!fn) { *err = "CRAM bzip2 block: libbz2 is not available on this system"; return false; }
    unsigned len = (unsigned)out->size();
    if (fn((char*)out->data(), &len, (char*)in, (unsigned)n, 0, 0) != 0 || len != out->size()) { *err = "CRAM bzip2 block decompression failed"; return false; }
    return true;
}
bool lzma_decode(const uint8_t* in, size_t n, std::vector<uint8_t>* out, std::string* err) {
    typedef int (*Fn)(uint64_t*, uint32_t, const void*, const uint8_t*, size_t*, size_t, uint8_t*, size_t*, size_t);
    static const Fn fn = []() -> Fn { void* h = dlopen("liblzma.so.5", RTLD_NOW); if (!h) h = dlopen("liblzma.so", RTLD_NOW);
                                      return h ? (Fn)dlsym(h, "lzma_stream_buffer_decode") : (Fn)nullptr; }();
    if (!fn) { *err = "CRAM lzma block: liblzma is not available on this system"; return false; }
    uint64_t memlimit = UINT64_MAX; size_t ip = 0, op = 0;
    if (fn(&memlimit, 0, nullptr, in, &ip, n, out->data(), &op, out->size()) != 0 || op != out->size()) { *err = "CRAM lzma block decompression failed"; return false; }
    return true;
}

}  // namespace

struct CramReader::Impl {
    FILE* f = nullptr;
    BamHeader hdr;
    std::vector<std::string> rg_ids;            // @RG IDs in header order (RG data series indexes this)
    Fasta* fa = nullptr;
    std::string err;
    int ref_tid = -1; std::string ref;
    int shared_tid = -1; const std::string* shared_ref = nullptr;   // CramReader::share_reference
    off_t data_start = 0;
    // per-slice decode state
    std::map<int, Block> ext; Block core; BitReader br;
    std::map<std::string, Enc> ds; std::map<int32_t, Enc> tagenc;
    bool rn_preserved = true, ap_delta = true, ref_required = true; uint8_t sm[5] = {0, 0, 0, 0, 0};
    // the reference of the slice being decoded: the contig loaded from the FASTA, or the slice's embedded reference block
    // (its first base is the slice's alignment start), or none (RR = 0: every base is stored as a feature)
    const uint8_t* emb_ref = nullptr; int64_t emb_start0 = 0, emb_len = 0;
    std::vector<std::vector<int32_t> > td;      // tag dictionary: per line the tag keys (tag0<<16|tag1<<8|type)

    // containers of the file, found by one walk over their headers
    // containers of the file: from the .crai when there is one (header_at known, the header itself read on first use), else
    // from one walk over the container headers
    struct Cont { int32_t ref, start, span, nrec, len; off_t body_at; off_t header_at; bool loaded; };
    std::vector<Cont> table; bool table_built = false;
    std::string path;

    // step over a block without decompressing it
    bool skip_block(Cur& c) {
        (void)c.u8(); (void)c.u8(); (void)c.itf8();
        const int32_t cs = c.itf8(); (void)c.itf8();
        if (c.bad || cs < 0 || c.p + cs + 4 > c.e) { err = "truncated CRAM block"; return false; }
        c.p += cs + 4;
        return true;
    }
    bool read_block(Cur& c, Block* b) {
        const uint8_t* const b0 = c.p;
        b->method = c.u8(); b->type = c.u8(); b->id = c.itf8();
        const int32_t cs = c.itf8(), us = c.itf8();
        if (c.bad || cs < 0 || us < 0 || c.p + cs + 4 > c.e) { err = "truncated CRAM block"; return false; }
        {   // CRC32 over the block header and its (compressed) data
            uint32_t want = 0; for (int k = 0; k < 4; ++k) want |= (uint32_t)c.p[cs + k] << (8 * k);
            if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), b0, (uInt)(c.p + cs - b0)) != want) { err = "CRAM block CRC32 mismatch"; return false; }
        }
        if (b->method == 0) b->data.assign(c.p, c.p + cs);
        else if (b->method == 1) {
            b->data.assign((size_t)us, 0);
            z_stream zs; memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, 15 + 32) != Z_OK) { err = "zlib init failed"; return false; }
            zs.next_in = (Bytef*)c.p; zs.avail_in = (uInt)cs; zs.next_out = b->data.data(); zs.avail_out = (uInt)us;
            const int rc = inflate(&zs, Z_FINISH); inflateEnd(&zs);
            if (rc != Z_STREAM_END) { err = "CRAM gzip block inflate failed"; return false; }
        } else if (b->method == 2) { b->data.assign((size_t)us, 0); if (!bz2_decode(c.p, (size_t)cs, &b->data, &err)) return false; }
        else if (b->method == 3) { b->data.assign((size_t)us, 0); if (!lzma_decode(c.p, (size_t)cs, &b->data, &err)) return false; }
        else if (b->method == 4) { if (!rans_decode(c.p, (size_t)cs, (size_t)us, &b->data, &err)) return false; }
        else { err = "CRAM block compression method " + std::to_string(b->method) + " not supported by the minimal reader"; return false; }
        b->pos = 0; c.p += cs + 4;
        return true;
    }

    // ---- decoders
    bool dec_int(const Enc& e, int32_t* out) {
        switch (e.codec) {
            case 1: { Block* const b = e.blk; if (!b) { err = "missing CRAM external block"; return false; }
                      Cur c; c.p = b->data.data() + b->pos; c.e = b->data.data() + b->data.size();
                      *out = c.itf8(); b->pos = (size_t)(c.p - b->data.data()); return !c.bad; }
            case 3: { if (e.sym.size() == 1 && e.len[0] == 0) { *out = e.sym[0]; return true; }
                      uint32_t code = 0; int l = 0; size_t i = 0;
                      while (i < e.sym.size()) { while (l < e.len[i]) { if ((br.bit >> 3) >= br.n) { err = "CRAM core block exhausted"; return false; } code = (code << 1) | (uint32_t)br.get(); ++l; }
                          for (; i < e.sym.size() && e.len[i] == l; ++i) if (e.code[i] == code) { *out = e.sym[i]; return true; } }
                      err = "bad CRAM Huffman code"; return false; }
            case 6: *out = (int32_t)br.bits(e.beta_bits) - e.beta_off; return true;
            case 7: { int i = 0; while (br.get() == 1 && i < 32) ++i;                     // SUBEXP: i ones, a zero, then i ? i + k - 1 : k bits
                      const int tail = i ? i + e.beta_bits - 1 : e.beta_bits;
                      if (tail > 31) { err = "bad CRAM SUBEXP value"; return false; }
                      uint32_t v = br.bits(tail); if (i) v += 1u << tail;
                      *out = (int32_t)v - e.beta_off; return true; }
            case 9: { int nz = 0; while (br.get() == 0 && nz < 31) { ++nz; if ((br.bit >> 3) >= br.n) { err = "CRAM core block exhausted"; return false; } }   // GAMMA: nz zeros, a one, nz bits
                      *out = (int32_t)((1u << nz) | br.bits(nz)) - e.beta_off; return true; }
            default: err = "CRAM integer codec " + std::to_string(e.codec) + " not supported"; return false;
        }
    }
    bool dec_byte(const Enc& e, uint8_t* out) {
        if (e.codec == 1) { Block* const b = e.blk; if (!b || b->pos >= b->data.size()) { err = "CRAM external block exhausted"; return false; }
            *out = b->data[b->pos++]; return true; }
        int32_t v; if (!dec_int(e, &v)) return false; *out = (uint8_t)v; return true;
    }
    bool dec_bytes(const Enc& e, std::vector<uint8_t>* out) {
        out->clear();
        if (e.codec == 5) { if (!e.blk) { err = "missing CRAM external block"; return false; }
            Block& b = *e.blk; const uint8_t* const p0 = b.data.data() + b.pos; const size_t left = b.data.size() - b.pos;
            const uint8_t* const q = (const uint8_t*)memchr(p0, e.stop, left); const size_t n = q ? (size_t)(q - p0) : left;
            out->assign(p0, p0 + n); b.pos += n + (q ? 1u : 0u); return true; }
        if (e.codec == 4) { int32_t n; if (!dec_int(e.sub[0], &n)) return false; if (n < 0) { err = "corrupt CRAM byte array length"; return false; }
            const Enc& v = e.sub[1];
            if (v.codec == 1 && v.blk && v.blk->data.size() - v.blk->pos >= (size_t)n) { out->assign(v.blk->data.data() + v.blk->pos, v.blk->data.data() + v.blk->pos + n); v.blk->pos += (size_t)n; return true; }
            for (int i = 0; i < n; ++i) { uint8_t b1; if (!dec_byte(v, &b1)) return false; out->push_back(b1); } return true; }
        err = "CRAM byte-array codec " + std::to_string(e.codec) + " not supported"; return false;
    }
    // The per-value path: a data series is found through a table indexed by its two-letter key (rebuilt per compression header), an
    // encoding's external block through a pointer bound once per slice — a map lookup with a string key per value made every read cost
    // 10 us (150 qualities alone are 150 values)
    std::vector<const Enc*> fast = std::vector<const Enc*>(65536, nullptr);
    const Enc* series(const char* k) const { return fast[((size_t)(uint8_t)k[0] << 8) | (size_t)(uint8_t)k[1]]; }
    void bind_enc(Enc& e) {
        e.blk = nullptr;
        if (e.codec == 1 || e.codec == 5) { auto it = ext.find(e.ext_id); if (it != ext.end()) e.blk = &it->second; }
        for (Enc& s : e.sub) bind_enc(s);
    }
    void bind_blocks() { for (auto& kv : ds) bind_enc(kv.second); for (auto& kv : tagenc) bind_enc(kv.second); }
    void index_series() { std::fill(fast.begin(), fast.end(), (const Enc*)nullptr); for (auto& kv : ds) if (kv.first.size() == 2) fast[((size_t)(uint8_t)kv.first[0] << 8) | (size_t)(uint8_t)kv.first[1]] = &kv.second; }
    bool geti(const char* k, int32_t* v) { const Enc* e = series(k); if (!e) { err = std::string("CRAM data series ") + k + " missing"; return false; } return dec_int(*e, v); }
    bool getb(const char* k, uint8_t* v) { const Enc* e = series(k); if (!e) { err = std::string("CRAM data series ") + k + " missing"; return false; } return dec_byte(*e, v); }
    // n values of a byte series at once (the quality array of a record): straight out of the block when the series is EXTERNAL
    bool getn(const char* k, uint8_t* dst, int n) {
        if (n <= 0) return true;                                   // (an empty read never looks its series up: a header without BA / QS decodes it)
        const Enc* e = series(k); if (!e) { err = std::string("CRAM data series ") + k + " missing"; return false; }
        if (e->codec == 1 && e->blk && n >= 0 && e->blk->data.size() - e->blk->pos >= (size_t)n) { memcpy(dst, e->blk->data.data() + e->blk->pos, (size_t)n); e->blk->pos += (size_t)n; return true; }
        for (int i = 0; i < n; ++i) if (!dec_byte(*e, dst + i)) return false;
        return true;
    }
    bool geta(const char* k, std::vector<uint8_t>* v) { const Enc* e = series(k); if (!e) { err = std::string("CRAM data series ") + k + " missing"; return false; } return dec_bytes(*e, v); }

    bool parse_comp_header(const Block& b) {
        ds.clear(); tagenc.clear(); td.clear(); rn_preserved = true; ap_delta = true; ref_required = true;
        Cur c; c.p = b.data.data(); c.e = c.p + b.data.size();
        auto sub = [&](Cur* m) {                                  // a size-prefixed map inside the header block
            const int32_t sz = c.itf8();
            if (c.bad || sz < 0 || sz > c.e - c.p) { err = "truncated CRAM compression header"; return false; }
            m->p = c.p; m->e = c.p + sz; c.p += sz; return true;
        };
        { Cur m; if (!sub(&m)) return false;
          const int n = m.itf8();
          for (int i = 0; i < n && !m.bad; ++i) {
              const char k0 = (char)m.u8(), k1 = (char)m.u8();
              if (k0 == 'R' && k1 == 'N') rn_preserved = m.u8() != 0;
              else if (k0 == 'A' && k1 == 'P') ap_delta = m.u8() != 0;
              else if (k0 == 'R' && k1 == 'R') ref_required = m.u8() != 0;
              else if (k0 == 'S' && k1 == 'M') for (int j = 0; j < 5; ++j) sm[j] = m.u8();
              else if (k0 == 'T' && k1 == 'D') { const int32_t l = m.itf8();
                  if (m.bad || l < 0 || l > m.e - m.p) { err = "truncated CRAM tag dictionary"; return false; }
                  const uint8_t* t = m.p; m.p += l;
                  std::vector<int32_t> line; for (int32_t o = 0; o < l;) { if (t[o] == 0) { td.push_back(line); line.clear(); ++o; } else { if (o + 3 > l) { err = "truncated CRAM tag dictionary"; return false; } line.push_back((t[o] << 16) | (t[o + 1] << 8) | t[o + 2]); o += 3; } } }
              else { err = "unknown CRAM preservation key"; return false; } }
          if (m.bad) { err = "truncated CRAM preservation map"; return false; } }
        { Cur m; if (!sub(&m)) return false;
          const int n = m.itf8();
          for (int i = 0; i < n; ++i) { std::string k; k.push_back((char)m.u8()); k.push_back((char)m.u8()); if (m.bad) { err = "truncated CRAM data series map"; return false; } Enc e; if (!parse_enc(m, &e, &err)) return false; ds[k] = e; } }
        { Cur m; if (!sub(&m)) return false;
          const int n = m.itf8();
          for (int i = 0; i < n; ++i) { const int32_t key = m.itf8(); if (m.bad) { err = "truncated CRAM tag encoding map"; return false; } Enc e; if (!parse_enc(m, &e, &err)) return false; tagenc[key] = e; } }
        index_series();
        return !c.bad;
    }

    static int base_index(char b) { switch (b) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; } }
    char substitute(char refb, int code) const {
        static const char B[5] = {'A', 'C', 'G', 'T', 'N'};
        const int r = base_index(refb); int k = 0;
        for (int a = 0; a < 5; ++a) { if (a == r) continue; if (((sm[r] >> (6 - 2 * k)) & 3) == code) return B[a]; ++k; }
        return 'N';
    }

    static void push_cigar(std::vector<uint32_t>& cg, uint32_t op, uint32_t len) {
        if (!len) return;
        if (!cg.empty() && (cg.back() & 15u) == op) cg.back() += len << 4; else cg.push_back((len << 4) | op);
    }

    // <cram>.crai (gzip text, one line per slice and reference: seq id, alignment start, span, container offset, slice offset,
    // slice size): one table entry per container, spanning its lines; several references in it = a multi-reference container
    bool load_crai() {
        gzFile z = gzopen((path + ".crai").c_str(), "rb");
        if (!z) return false;
        std::string text; char buf[65536]; int n;
        while ((n = gzread(z, buf, sizeof buf)) > 0) text.append(buf, (size_t)n);
        gzclose(z);
        std::map<long long, Cont> by_off;
        size_t p = 0; bool any = false;
        while (p < text.size()) {
            size_t e = text.find('\n', p); if (e == std::string::npos) e = text.size();
            long long sid, st, sp, coff, soff, ssz;
            if (sscanf(text.substr(p, e - p).c_str(), "%lld %lld %lld %lld %lld %lld", &sid, &st, &sp, &coff, &soff, &ssz) == 6) {
                any = true;
                auto it = by_off.find(coff);
                if (it == by_off.end()) { Cont c; c.ref = (int32_t)sid; c.start = (int32_t)st; c.span = (int32_t)sp; c.nrec = 0; c.len = 0; c.body_at = 0; c.header_at = (off_t)coff; c.loaded = false; by_off[coff] = c; }
                else {
                    Cont& c = it->second;
                    if (c.ref != (int32_t)sid) c.ref = -2;
                    else { const long long a = std::min<long long>(c.start, st), b = std::max<long long>((long long)c.start + c.span, st + sp); c.start = (int32_t)a; c.span = (int32_t)(b - a); }
                }
            }
            p = e + 1;
        }
        if (!any) return false;
        for (auto& kv : by_off) if (kv.second.ref != -1) table.push_back(kv.second);     // (unmapped-only containers are never asked for)
        return true;
    }

    // decode the records of one slice; cb(record) for those overlapping [beg,end) on tid
    template <class F>
    bool decode_slice(int slice_ref, int slice_start, int nrec, int tid, int64_t beg, int64_t end, F& cb) {
        br.p = core.data.data(); br.n = core.data.size(); br.bit = 0;
        int32_t last_ap = slice_start;
        // (buffers of the record loop: cleared, not reallocated, per record)
        std::vector<uint8_t> name, tmp, aux, qual; std::string seq; std::vector<uint32_t> cg; BamRecord rec;
        static const struct Nt16 { uint8_t t[256]; Nt16() { memset(t, 15, sizeof t); const char* codes = "=ACMGRSVTWYHKDBN"; for (int i = 0; i < 16; ++i) { t[(uint8_t)codes[i]] = (uint8_t)i; t[(uint8_t)tolower(codes[i])] = (uint8_t)i; } } } nt16;
        for (int r = 0; r < nrec; ++r) {
            int32_t bf, cf, ri = slice_ref, rl, ap, rg;
            if (!geti("BF", &bf) || !geti("CF", &cf)) return false;
            if (slice_ref == -2 && !geti("RI", &ri)) return false;
            if (!geti("RL", &rl) || !geti("AP", &ap) || !geti("RG", &rg)) return false;
            if (rl < 0 || rl > (1 << 28) || (ri < 0 && !(bf & 4)) || ri >= (int32_t)hdr.names.size()) { err = "corrupt CRAM record"; return false; }
            if (ap_delta) { ap += last_ap; last_ap = ap; }
            name.clear(); tmp.clear();
            if (rn_preserved && !geta("RN", &name)) return false;
            if (cf & 2) { int32_t mf, ns, np, ts; if (!geti("MF", &mf)) return false; if (!rn_preserved && !geta("RN", &name)) return false;
                          if (!geti("NS", &ns) || !geti("NP", &np) || !geti("TS", &ts)) return false; }
            else if (cf & 4) { int32_t nf; if (!geti("NF", &nf)) return false; }
            int32_t tl; if (!geti("TL", &tl)) return false;
            aux.clear();
            bool has_nm = false; uint32_t nm = 0;
            if (tl >= 0 && (size_t)tl < td.size()) for (int32_t key : td[(size_t)tl]) {
                if ((key >> 8) == (('N' << 8) | 'M')) has_nm = true;
                auto it = tagenc.find(key); if (it == tagenc.end()) { err = "CRAM tag encoding missing"; return false; }
                if (!dec_bytes(it->second, &tmp)) return false;
                aux.push_back((uint8_t)(key >> 16)); aux.push_back((uint8_t)(key >> 8)); aux.push_back((uint8_t)key);
                aux.insert(aux.end(), tmp.begin(), tmp.end());
            }
            if (rg >= 0 && (size_t)rg < rg_ids.size()) { aux.push_back('R'); aux.push_back('G'); aux.push_back('Z'); aux.insert(aux.end(), rg_ids[(size_t)rg].begin(), rg_ids[(size_t)rg].end()); aux.push_back(0); }
            seq.assign((size_t)std::max(rl, 0), 'N'); qual.assign((size_t)std::max(rl, 0), 0xff); cg.clear();
            int32_t mq = 0;
            if (!(bf & 4)) {
                const bool have_ref = emb_ref != nullptr || ref_required;              // htslib: s->ref
                const bool use_shared = shared_ref && ri == shared_tid;
                if (!emb_ref && ref_required && !use_shared && ri != ref_tid) { if (!fa || !fa->fetch(hdr.names[(size_t)ri], &ref)) { err = "CRAM needs the reference FASTA (-f) to reconstruct reads"; return false; } ref_tid = ri; }
                const std::string& R = use_shared ? *shared_ref : ref;
                // raw reference character at 0-based x ('N' outside what is known)
                auto ref_raw = [&](int64_t x) -> char {
                    if (emb_ref) { const int64_t k = x - emb_start0; return (k >= 0 && k < emb_len) ? (char)emb_ref[k] : 'N'; }
                    if (!ref_required) return 'N';
                    return (x >= 0 && x < (int64_t)R.size()) ? R[(size_t)x] : 'N';
                };
                const int64_t ref_end = emb_ref ? emb_start0 + emb_len : (ref_required ? (int64_t)R.size() : 0);
                int32_t fn; if (!geti("FN", &fn)) return false;
                int64_t refp = (int64_t)ap - 1; int sp = 1, prev = 0;
                auto ref_at = [&](int64_t x) { return (char)toupper((unsigned char)ref_raw(x)); };
                auto copy_ref = [&](int len) { for (int i = 0; i < len; ++i) seq[(size_t)(sp - 1 + i)] = ref_at(refp + i); };
                // every feature must stay inside the read (positions are 1-based; a clip or pad may sit at rl + 1)
                auto bad_record = [&]() { err = "corrupt CRAM record (a read feature outside its read)"; return false; };
                auto room = [&](int64_t at1, int64_t n) { return at1 >= 1 && n >= 0 && at1 - 1 + n <= (int64_t)rl; };
                for (int fi = 0; fi < fn; ++fi) {
                    uint8_t fc; int32_t fp; if (!getb("FC", &fc) || !geti("FP", &fp)) return false;
                    const int64_t pos64 = (int64_t)prev + fp;
                    if (pos64 < 1 || pos64 > (int64_t)rl + 1) return bad_record();
                    const int pos = (int)pos64; prev = pos;
                    if (pos > sp) { const int l = pos - sp; copy_ref(l); push_cigar(cg, 0, (uint32_t)l); refp += l; sp = pos; }
                    uint8_t b1; int32_t iv = 0;
                    switch (fc) {
                        case 'X': case 'i': case 'B': if (!room(sp, 1)) return bad_record(); break;
                        case 'Q': if (!room(pos, 1)) return bad_record(); break;
                        default: break;
                    }
                    switch (fc) {
                        case 'X': if (!getb("BS", &b1)) return false; seq[(size_t)(sp - 1)] = substitute(ref_raw(refp), b1); push_cigar(cg, 0, 1); ++refp; ++sp; ++nm; break;
                        case 'I': if (!geta("IN", &tmp)) return false; if (!room(sp, (int64_t)tmp.size())) return bad_record(); for (size_t i = 0; i < tmp.size(); ++i) seq[(size_t)(sp - 1) + i] = (char)tmp[i]; push_cigar(cg, 1, (uint32_t)tmp.size()); sp += (int)tmp.size(); nm += (uint32_t)tmp.size(); break;
                        case 'i': if (!getb("BA", &b1)) return false; seq[(size_t)(sp - 1)] = (char)b1; push_cigar(cg, 1, 1); ++sp; ++nm; break;
                        case 'S': if (!geta("SC", &tmp)) return false; if (!room(sp, (int64_t)tmp.size())) return bad_record(); for (size_t i = 0; i < tmp.size(); ++i) seq[(size_t)(sp - 1) + i] = (char)tmp[i]; push_cigar(cg, 4, (uint32_t)tmp.size()); sp += (int)tmp.size(); break;
                        case 'D': if (!geti("DL", &iv)) return false; if (iv < 0 || iv >= (1 << 28)) return bad_record(); push_cigar(cg, 2, (uint32_t)iv);
                                  nm += refp + iv <= ref_end ? (uint32_t)iv : (uint32_t)std::max<int64_t>(ref_end - refp, 0);
                                  refp += iv; break;
                        case 'N': if (!geti("RS", &iv)) return false; if (iv < 0 || iv >= (1 << 28)) return bad_record(); push_cigar(cg, 3, (uint32_t)iv); refp += iv; break;
                        case 'H': if (!geti("HC", &iv)) return false; if (iv < 0 || iv >= (1 << 28)) return bad_record(); push_cigar(cg, 5, (uint32_t)iv); break;
                        case 'P': if (!geti("PD", &iv)) return false; if (iv < 0 || iv >= (1 << 28)) return bad_record(); push_cigar(cg, 6, (uint32_t)iv); break;
                        case 'B': { uint8_t q; if (!getb("BA", &b1) || !getb("QS", &q)) return false; seq[(size_t)(sp - 1)] = (char)b1; qual[(size_t)(sp - 1)] = q; push_cigar(cg, 0, 1);
                                    if (ref_at(refp) != (char)b1) ++nm;
                                    ++refp; ++sp; break; }
                        case 'b': if (!geta("BB", &tmp)) return false; if (!room(sp, (int64_t)tmp.size())) return bad_record();
                                  for (size_t i = 0; i < tmp.size(); ++i) { seq[(size_t)(sp - 1) + i] = (char)tmp[i]; if (ref_at(refp + (int64_t)i) != (char)tmp[i]) ++nm; }
                                  push_cigar(cg, 0, (uint32_t)tmp.size()); refp += (int64_t)tmp.size(); sp += (int)tmp.size(); break;
                        case 'Q': { uint8_t q; if (!getb("QS", &q)) return false; qual[(size_t)(pos - 1)] = q; break; }
                        case 'q': if (!geta("QQ", &tmp)) return false; if (!room(pos, (int64_t)tmp.size())) return bad_record(); for (size_t i = 0; i < tmp.size(); ++i) qual[(size_t)(pos - 1) + i] = tmp[i]; break;
                        default: err = "unknown CRAM feature code"; return false;
                    }
                }
                if (sp <= rl) { const int l = rl - sp + 1; copy_ref(l); push_cigar(cg, 0, (uint32_t)l); }
                if (!geti("MQ", &mq)) return false;
                if ((cf & 1) && !getn("QS", qual.data(), rl)) return false;
                // htslib regenerates NM (and MD, which this path never reads) for a mapped record that was stored without
                // it — samtools drops both tags when it writes CRAM (cram_decode.c cram_decode_seq, decode_md = 1 by default):
                // substitutions, inserted and deleted bases, and literal bases that differ from the reference
                if (!has_nm && !(cf & 8) && ri >= 0 && have_ref) { aux.push_back('N'); aux.push_back('M'); aux.push_back('I'); for (int k = 0; k < 4; ++k) aux.push_back((uint8_t)(nm >> (8 * k))); }
            } else {
                if (!getn("BA", reinterpret_cast<uint8_t*>(&seq[0]), rl)) return false;
                if ((cf & 1) && !getn("QS", qual.data(), rl)) return false;
            }
            // ---- BAM-layout record
            rec.tid = ri; rec.pos = ap - 1; rec.mapq = (uint8_t)mq; rec.flag = (uint16_t)bf; rec.l_seq = rl; rec.n_cigar = (uint32_t)cg.size();
            if (name.empty()) { const std::string gen = "cram" + std::to_string(r); name.assign(gen.begin(), gen.end()); }
            rec.l_qname = (uint32_t)name.size() + 1;
            rec.data.clear();
            rec.data.reserve(name.size() + 1 + 4 * cg.size() + (size_t)(rl + 1) / 2 + qual.size() + aux.size());
            rec.data.assign(name.begin(), name.end()); rec.data.push_back(0);
            for (uint32_t v : cg) for (int k = 0; k < 4; ++k) rec.data.push_back((uint8_t)(v >> (8 * k)));
            for (int i = 0; i < rl; i += 2) {
                const uint8_t hi = nt16.t[(uint8_t)seq[(size_t)i]], lo = i + 1 < rl ? nt16.t[(uint8_t)seq[(size_t)i + 1]] : 0;
                rec.data.push_back((uint8_t)((hi << 4) | lo));
            }
            rec.data.insert(rec.data.end(), qual.begin(), qual.end());
            rec.data.insert(rec.data.end(), aux.begin(), aux.end());
            if (rec.tid == tid && rec.pos < end && rec.endpos() > beg) cb(rec);
        }
        return true;
    }
};

CramReader::CramReader() : d_(new Impl) {}
CramReader::~CramReader() { if (d_->f) fclose(d_->f); delete d_; }
const BamHeader& CramReader::header() const { return d_->hdr; }
const std::string& CramReader::error() const { return d_->err; }
void CramReader::share_reference(int tid, const std::string* bases) { d_->shared_tid = bases ? tid : -1; d_->shared_ref = bases; }

bool CramReader::is_cram(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb"); if (!f) return false;
    char m[4] = {0, 0, 0, 0}; const size_t n = fread(m, 1, 4, f); fclose(f);
    return n == 4 && memcmp(m, "CRAM", 4) == 0;
}

// *bad (when given): set when bytes were there but are not a container header (its CRC32 does not match), left alone at the
// end of the file
static bool read_container_header(FILE* f, int32_t* length, int32_t* ref, int32_t* start, int32_t* span, int32_t* nrec, int32_t* nblocks, std::vector<int32_t>* land, bool* bad = nullptr) {
    uint8_t buf[1024];
    const off_t at = ftello(f);
    const size_t got = fread(buf, 1, sizeof buf, f);
    if (got < 8) return false;
    Cur c; c.p = buf; c.e = buf + got;
    *length = c.i32(); *ref = c.itf8(); *start = c.itf8(); *span = c.itf8(); *nrec = c.itf8();
    (void)c.ltf8(); (void)c.ltf8(); *nblocks = c.itf8();
    const int nl = c.itf8(); land->clear(); for (int i = 0; i < nl && !c.bad; ++i) land->push_back(c.itf8());
    if (c.bad || c.p + 4 > c.e) { if (bad) *bad = true; return false; }
    uint32_t want = 0; for (int k = 0; k < 4; ++k) want |= (uint32_t)c.p[k] << (8 * k);
    if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), buf, (uInt)(c.p - buf)) != want) { if (bad) *bad = true; return false; }
    c.p += 4;
    fseeko(f, at + (off_t)(c.p - buf), SEEK_SET);
    return true;
}

bool CramReader::open(const std::string& path, Fasta* fasta) {
    Impl& d = *d_;
    d.fa = fasta;
    d.f = fopen(path.c_str(), "rb");
    if (!d.f) { d.err = "cannot open " + path; return false; }
    d.path = path;
    uint8_t def[26];
    if (fread(def, 1, 26, d.f) != 26 || memcmp(def, "CRAM", 4) != 0) { d.err = "not a CRAM file"; return false; }
    if (def[4] != 3) { d.err = "only CRAM 3.x is supported by the minimal reader"; return false; }
    int32_t len, ref, start, span, nrec, nblocks; std::vector<int32_t> land;
    if (!read_container_header(d.f, &len, &ref, &start, &span, &nrec, &nblocks, &land)) { d.err = "bad CRAM header container"; return false; }
    std::vector<uint8_t> body((size_t)len);
    if (fread(body.data(), 1, body.size(), d.f) != body.size()) { d.err = "truncated CRAM header container"; return false; }
    d.data_start = ftello(d.f);
    Cur c; c.p = body.data(); c.e = c.p + body.size();
    Block b; if (!d.read_block(c, &b)) return false;
    if (b.data.size() < 4) { d.err = "bad CRAM SAM header block"; return false; }
    uint32_t tl = 0; for (int k = 0; k < 4; ++k) tl |= (uint32_t)b.data[(size_t)k] << (8 * k);
    d.hdr.text.assign((const char*)b.data.data() + 4, std::min<size_t>(tl, b.data.size() - 4));
    // @SQ / @RG
    size_t p = 0;
    while (p < d.hdr.text.size()) {
        size_t e = d.hdr.text.find('\n', p); if (e == std::string::npos) e = d.hdr.text.size();
        const std::string line = d.hdr.text.substr(p, e - p);
        auto field = [&](const char* key) { const std::string k = std::string("\t") + key + ":"; const size_t q = line.find(k); if (q == std::string::npos) return std::string(); size_t t = line.find('\t', q + 1); return line.substr(q + k.size(), (t == std::string::npos ? line.size() : t) - q - k.size()); };
        if (line.compare(0, 3, "@SQ") == 0) { const std::string n = field("SN"); d.hdr.name2tid[n] = (int)d.hdr.names.size(); d.hdr.names.push_back(n); d.hdr.lengths.push_back(atoi(field("LN").c_str())); }
        if (line.compare(0, 3, "@RG") == 0) d.rg_ids.push_back(field("ID"));
        p = e + 1;
    }
    d.hdr.parse_read_groups();
    return true;
}

bool CramReader::fetch_impl(int tid, int64_t beg, int64_t end, void (*thunk)(void*, const BamRecord&), void* ctx) {
    Impl& d = *d_;
    if (beg < 0) beg = 0;
    // The container headers are walked ONCE (reference id, start, span, where the body lies): later queries — a site list is
    // one query per line — go straight to the containers that can overlap.  (A .crai would give the same table without the walk.)
    if (!d.table_built) {
        if (!d.load_crai()) {
            fseeko(d.f, d.data_start, SEEK_SET);
            for (;;) {
                int32_t len, ref, start, span, nrec, nblocks; std::vector<int32_t> land;
                const off_t at = ftello(d.f);
                bool bad = false;
                if (!read_container_header(d.f, &len, &ref, &start, &span, &nrec, &nblocks, &land, &bad)) {
                    if (bad) { d.err = "CRAM container header CRC32 mismatch"; return false; }
                    break;
                }
                Impl::Cont ct; ct.ref = ref; ct.start = start; ct.span = span; ct.nrec = nrec; ct.len = len; ct.body_at = ftello(d.f); ct.header_at = at; ct.loaded = true;
                if (!(ref == -1 && nrec == 0) && nrec > 0) d.table.push_back(ct);          // (not the EOF marker / empty containers)
                if (fseeko(d.f, ct.body_at + len, SEEK_SET) != 0) break;
            }
        }
        d.table_built = true;
    }
    std::vector<uint8_t> body;
    for (Impl::Cont& ct : d.table) {
        const bool may = ct.ref == -2 || (ct.ref == tid && (int64_t)ct.start - 1 < end && (int64_t)ct.start - 1 + ct.span > beg);
        if (!may) continue;
        if (!ct.loaded) {                                                         // an index entry: now read the container's header
            int32_t ref, start, span, nblocks; std::vector<int32_t> land;
            if (fseeko(d.f, ct.header_at, SEEK_SET) != 0 || !read_container_header(d.f, &ct.len, &ref, &start, &span, &ct.nrec, &nblocks, &land)) { d.err = "the .crai index points at something that is not a CRAM container"; return false; }
            ct.body_at = ftello(d.f); ct.loaded = true;
        }
        body.resize((size_t)ct.len);
        if (fseeko(d.f, ct.body_at, SEEK_SET) != 0 || fread(body.data(), 1, body.size(), d.f) != body.size()) { d.err = "truncated CRAM container"; return false; }
        Cur c; c.p = body.data(); c.e = c.p + body.size();
        Block ch; if (!d.read_block(c, &ch) || ch.type != 1 || !d.parse_comp_header(ch)) { if (d.err.empty()) d.err = "bad CRAM compression header"; return false; }
        while (c.p < c.e) {
            Block sh; if (!d.read_block(c, &sh)) return false;
            if (sh.type != 2) { d.err = "expected a CRAM slice header"; return false; }
            Cur s; s.p = sh.data.data(); s.e = s.p + sh.data.size();
            const int32_t sref = s.itf8(), sstart = s.itf8(); const int32_t sspan = s.itf8(); const int32_t snrec = s.itf8(); (void)s.ltf8(); const int32_t snb = s.itf8();
            d.ext.clear();
            // a single-reference slice that cannot overlap is skipped without inflating its blocks
            const bool slice_may = sref == -2 || sref < 0 || (sref == tid && (int64_t)sstart - 1 < end && (int64_t)sstart - 1 + sspan > beg);
            if (!slice_may) { for (int i = 0; i < snb; ++i) if (!d.skip_block(c)) return false; continue; }
            for (int i = 0; i < snb; ++i) { Block b; if (!d.read_block(c, &b)) return false; if (b.type == 5) d.core = std::move(b); else d.ext[b.id] = std::move(b); }
            d.bind_blocks();
            // (slice header, continued: the block content ids, then the id of an embedded reference block or -1)
            { const int32_t ncid = s.itf8(); for (int32_t i = 0; i < ncid; ++i) (void)s.itf8(); }
            const int32_t emb = s.bad ? -1 : s.itf8();
            d.emb_ref = nullptr; d.emb_len = 0;
            if (!s.bad && emb >= 0) {
                auto it = d.ext.find(emb);
                if (it == d.ext.end()) { d.err = "CRAM slice names an embedded reference block it does not have"; return false; }
                d.emb_ref = it->second.data.data(); d.emb_len = (int64_t)it->second.data.size(); d.emb_start0 = (int64_t)sstart - 1;
            }
            auto cb = [&](const BamRecord& r) { thunk(ctx, r); };
            if (!d.decode_slice(sref, sstart, snrec, tid, beg, end, cb)) return false;
        }
    }
    return true;
}

}  // namespace brcio
