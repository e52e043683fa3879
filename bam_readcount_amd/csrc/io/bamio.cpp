// bamio.cpp — see bamio.h.  Wire formats per the public SAMv1 specification; no htslib code involved.
#include "bamio.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <set>

namespace brcio {

static inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

// ---------------------------------------------------------------- BGZF

// Raw-deflate decoding of the blocks: libdeflate when the system has it (2-3x zlib on 64-KB blocks; the image ships the
// shared object without headers, so its four entry points are looked up at run time), zlib otherwise.
namespace {
struct LibDeflate {
    void* (*alloc)() = nullptr; void (*release)(void*) = nullptr;
    int (*inflate)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    uint32_t (*crc)(uint32_t, const void*, size_t) = nullptr;
    bool ok = false;
    LibDeflate() {
        if (getenv("BRC_NO_LIBDEFLATE")) return;
        void* h = dlopen("libdeflate.so.0", RTLD_NOW); if (!h) h = dlopen("libdeflate.so", RTLD_NOW);
        if (!h) return;
        alloc = (void* (*)())dlsym(h, "libdeflate_alloc_decompressor"); release = (void (*)(void*))dlsym(h, "libdeflate_free_decompressor");
        inflate = (int (*)(void*, const void*, size_t, void*, size_t, size_t*))dlsym(h, "libdeflate_deflate_decompress");
        crc = (uint32_t (*)(uint32_t, const void*, size_t))dlsym(h, "libdeflate_crc32");
        ok = alloc && release && inflate && crc;
    }
};
const LibDeflate& libdeflate() { static const LibDeflate L; return L; }
}  // namespace

bool Bgzf::open(const std::string& path) {
    close();
    f_ = fopen(path.c_str(), "rb");
    if (!f_) { err_ = "cannot open " + path; return false; }
    cbuf_.resize(1 << 16); ubuf_.resize(1 << 16);
    block_coff_ = next_coff_ = 0; pos_ = len_ = 0; eof_ = false;
    return true;
}
void Bgzf::close() { if (f_) fclose(f_); f_ = nullptr; if (ld_) { libdeflate().release(ld_); ld_ = nullptr; } }

// One BGZF member: gzip header with the BC extra subfield (total block size - 1), raw deflate payload, CRC32, ISIZE.
bool Bgzf::load_block(uint64_t coff) {
    if (fseeko(f_, (off_t)coff, SEEK_SET) != 0) { err_ = "seek failed"; return false; }
    uint8_t h[18];
    const size_t got = fread(h, 1, 18, f_);
    if (got == 0) { eof_ = true; len_ = pos_ = 0; block_coff_ = coff; next_coff_ = coff; return false; }
    if (got < 18 || h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { err_ = "not a BGZF block"; return false; }
    const uint16_t xlen = rd16(h + 10);
    // locate the BC subfield (normally first)
    std::vector<uint8_t> extra(xlen);
    memcpy(extra.data(), h + 12, std::min<size_t>(6, xlen));
    if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, f_) != (size_t)(xlen - 6)) { err_ = "truncated BGZF header"; return false; }
    int bsize = -1;
    for (size_t o = 0; o + 4 <= extra.size();) {
        const uint16_t slen = rd16(extra.data() + o + 2);
        if (extra[o] == 66 && extra[o + 1] == 67 && slen == 2) bsize = rd16(extra.data() + o + 4);
        o += 4 + slen;
    }
    if (bsize < 0) { err_ = "BGZF block without BC subfield"; return false; }
    const size_t clen = (size_t)bsize + 1 - 12 - xlen - 8;
    if (clen > cbuf_.size()) cbuf_.resize(clen);
    uint8_t tail[8];
    if (fread(cbuf_.data(), 1, clen, f_) != clen || fread(tail, 1, 8, f_) != 8) { err_ = "truncated BGZF block"; return false; }
    const uint32_t isize = rd32(tail + 4);
    if (isize > ubuf_.size()) ubuf_.resize(isize);
    const LibDeflate& L = libdeflate();
    uint32_t crc;
    if (L.ok) {
        if (!ld_) ld_ = L.alloc();
        size_t got_out = 0;
        if (!ld_ || L.inflate(ld_, cbuf_.data(), clen, ubuf_.data(), ubuf_.size(), &got_out) != 0 || got_out != isize) { err_ = "inflate failed"; return false; }
        crc = L.crc(0, ubuf_.data(), isize);
    } else {
        z_stream zs; memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) { err_ = "zlib init failed"; return false; }
        zs.next_in = cbuf_.data(); zs.avail_in = (uInt)clen; zs.next_out = ubuf_.data(); zs.avail_out = (uInt)ubuf_.size();
        const int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        if (rc != Z_STREAM_END || zs.total_out != isize) { err_ = "inflate failed"; return false; }
        crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), ubuf_.data(), (uInt)isize);
    }
    if (crc != rd32(tail)) { err_ = "BGZF block CRC32 mismatch"; return false; }
    block_coff_ = coff; next_coff_ = coff + (uint64_t)bsize + 1; len_ = isize; pos_ = 0;
    return true;
}

bool Bgzf::seek(uint64_t voffset) {
    const uint64_t coff = voffset >> 16; const size_t uoff = (size_t)(voffset & 0xffff);
    eof_ = false;
    if (!(len_ > 0 && coff == block_coff_)) { if (!load_block(coff)) return eof_ ? (uoff == 0) : false; }
    if (uoff > len_) { err_ = "virtual offset beyond block"; return false; }
    pos_ = uoff;
    return true;
}

bool Bgzf::read(void* dst, size_t n) {
    uint8_t* d = (uint8_t*)dst;
    while (n > 0) {
        if (pos_ == len_) {
            // blocks with ISIZE 0 (EOF marker) are skipped
            do { if (!load_block(next_coff_)) return false; } while (len_ == 0);
        }
        const size_t k = std::min(n, len_ - pos_);
        memcpy(d, ubuf_.data() + pos_, k);
        pos_ += k; d += k; n -= k;
    }
    return true;
}

size_t Bgzf::read_some(void* dst, size_t n) {
    uint8_t* d = (uint8_t*)dst; size_t got = 0;
    while (got < n) {
        if (pos_ == len_) { bool ok = true; do { ok = load_block(next_coff_); } while (ok && len_ == 0); if (!ok) break; }
        const size_t k = std::min(n - got, len_ - pos_);
        memcpy(d + got, ubuf_.data() + pos_, k);
        pos_ += k; got += k;
    }
    return got;
}

// ---------------------------------------------------------------- BAM records

int32_t BamRecord::endpos() const {
    if (!(flag & 4) && n_cigar > 0) {
        int32_t l = 0;
        const uint8_t* c = data.data() + l_qname;
        for (unsigned k = 0; k < n_cigar; ++k) {
            const uint32_t v = rd32(c + 4 * k); const uint32_t op = v & 15;
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) l += (int32_t)(v >> 4);
        }
        return pos + l;
    }
    return pos + 1;
}

static const uint8_t* aux_find(const uint8_t* p, const uint8_t* e, const char tag[2]) {
    while (p + 3 <= e) {
        const bool hit = p[0] == (uint8_t)tag[0] && p[1] == (uint8_t)tag[1];
        const uint8_t ty = p[2];
        const uint8_t* v = p + 3;
        size_t sz;
        switch (ty) {
            case 'A': case 'c': case 'C': sz = 1; break;
            case 's': case 'S': sz = 2; break;
            case 'i': case 'I': case 'f': sz = 4; break;
            case 'd': sz = 8; break;
            case 'Z': case 'H': { const uint8_t* q = v; while (q < e && *q) ++q; sz = (size_t)(q - v) + 1; break; }
            case 'B': {
                if (v + 5 > e) return nullptr;
                const uint8_t st = v[0]; const uint32_t n = rd32(v + 1);
                const size_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
                sz = 5 + (size_t)n * es; break;
            }
            default: return nullptr;
        }
        if (hit) return p + 2;   // points at the type byte
        p = v + sz;
    }
    return nullptr;
}

bool BamRecord::aux_int(const char tag[2], int32_t* out) const {
    const uint8_t* t = aux_find(aux(), data.data() + data.size(), tag);
    if (!t) return false;
    const uint8_t* v = t + 1;
    switch (*t) {   // bam_aux2i
        case 'c': *out = (int8_t)v[0]; return true;
        case 'C': *out = v[0]; return true;
        case 's': *out = (int16_t)rd16(v); return true;
        case 'S': *out = rd16(v); return true;
        case 'i': case 'I': *out = (int32_t)rd32(v); return true;
        default: *out = 0; return true;   // bam_aux2i returns 0 for non-integer types; the tag is still "present"
    }
}

const char* BamRecord::aux_str(const char tag[2]) const {
    const uint8_t* t = aux_find(aux(), data.data() + data.size(), tag);
    if (!t || (*t != 'Z' && *t != 'H')) return nullptr;
    return (const char*)(t + 1);
}

std::vector<std::string> BamHeader::libraries() const {
    std::set<std::string> s;
    for (const auto& kv : rg2lb) s.insert(kv.second);
    return std::vector<std::string>(s.begin(), s.end());
}

// The two ways the reference looks at @RG lines:
//  * per read, samtools-1.10's legacy bam_get_library (bamreadcount.cpp:280): the first @RG line of the header text with an
//    ID and an LB whose ID equals the read's RG:Z — the scan takes the LAST "ID:" / "LB:" that follow a tab, and only
//    accepts an ID that is itself followed by a tab (an ID at the end of its line never matches); LB is cut at 1023 bytes;
//  * once, find_library_names (:92-111) for the "Expect library" lines: every tag of every @RG line EXCEPT THE FIRST
//    (`while (tag->next) { tag = tag->next; ...`) that starts with "LB", so an LB written first is not announced and a
//    second LB on a line is.
void BamHeader::parse_read_groups() {
    rg2lb.clear(); expected.clear();
    std::set<std::string> exp;
    size_t p = 0;
    while (p < text.size()) {
        size_t e = text.find('\n', p); if (e == std::string::npos) e = text.size();
        if (text.compare(p, 3, "@RG") == 0) {
            size_t id = std::string::npos, lb = std::string::npos; char last = '\t';
            for (size_t q = p + 4; q < e; ++q) {
                if (last == '\t') { if (text.compare(q, 3, "LB:") == 0) lb = q + 3; else if (text.compare(q, 3, "ID:") == 0) id = q + 3; }
                last = text[q];
            }
            if (id != std::string::npos && lb != std::string::npos) {
                const size_t ie = text.find('\t', id);
                if (ie != std::string::npos && ie < e) {
                    size_t le = lb; while (le < e && text[le] != '\t') ++le;
                    const std::string key = text.substr(id, ie - id);
                    if (!rg2lb.count(key)) rg2lb[key] = text.substr(lb, std::min<size_t>(le - lb, 1023));
                }
            }
            size_t le = e; if (le > p && text[le - 1] == '\r') --le;
            bool first = true;
            for (size_t q = p + 3; q < le;) {
                if (text[q] == '\t') { ++q; continue; }
                size_t t = text.find('\t', q); if (t == std::string::npos || t > le) t = le;
                if (!first && t - q >= 2 && text.compare(q, 2, "LB") == 0) exp.insert(t - q > 3 ? text.substr(q + 3, t - q - 3) : std::string());
                first = false; q = t;
            }
        }
        p = e + 1;
    }
    expected.assign(exp.begin(), exp.end());
}

bool BamReader::open(const std::string& path) {
    if (!bg_.open(path)) { err_ = bg_.error(); return false; }
    uint8_t m[8];
    if (!bg_.read(m, 8) || memcmp(m, "BAM\1", 4) != 0) { err_ = "not a BAM file"; return false; }
    const uint32_t l_text = rd32(m + 4);
    hdr_.text.resize(l_text);
    if (l_text && !bg_.read(&hdr_.text[0], l_text)) { err_ = "truncated BAM header"; return false; }
    hdr_.text.resize(strlen(hdr_.text.c_str()));
    uint8_t b4[4];
    if (!bg_.read(b4, 4)) { err_ = "truncated BAM header"; return false; }
    const uint32_t n_ref = rd32(b4);
    for (uint32_t i = 0; i < n_ref; ++i) {
        if (!bg_.read(b4, 4)) return false;
        const uint32_t ln = rd32(b4);
        std::string name(ln, '\0');
        if (!bg_.read(&name[0], ln) || !bg_.read(b4, 4)) return false;
        name.resize(strlen(name.c_str()));
        hdr_.name2tid[name] = (int)hdr_.names.size();
        hdr_.names.push_back(name); hdr_.lengths.push_back((int32_t)rd32(b4));
    }
    hdr_.parse_read_groups();
    return true;
}

bool BamReader::next(BamRecord* r) {
    uint8_t h[36];
    if (!bg_.read(h, 4)) { if (!bg_.eof_clean()) err_ = bg_.error().empty() ? "truncated BAM file" : bg_.error(); return false; }
    const uint32_t bs = rd32(h);
    if (bs < 32 || !bg_.read(h + 4, 32)) { err_ = "truncated BAM record"; return false; }
    r->tid = (int32_t)rd32(h + 4); r->pos = (int32_t)rd32(h + 8);
    r->l_qname = h[12]; r->mapq = h[13]; r->bin = rd16(h + 14);
    r->n_cigar = rd16(h + 16); r->flag = rd16(h + 18); r->l_seq = (int32_t)rd32(h + 20);
    r->mtid = (int32_t)rd32(h + 24); r->mpos = (int32_t)rd32(h + 28); r->tlen = (int32_t)rd32(h + 32);
    r->data.resize(bs - 32);
    if (bs > 32 && !bg_.read(r->data.data(), bs - 32)) { err_ = "truncated BAM record"; return false; }
    if (r->l_seq < 0 || (uint64_t)r->l_qname + 4ull * r->n_cigar + ((uint64_t)r->l_seq + 1) / 2 + (uint64_t)r->l_seq > r->data.size()) { err_ = "corrupt BAM record"; return false; }
    // Long CIGARs (SAMv1 4.2.2): more than 65535 operators are stored in the CG:B,I tag behind the placeholder <l_seq>S<span>N;
    // put the real operators in place (htslib's bam_tag2cigar does the same when it reads the record)
    if (r->n_cigar == 2) {
        const uint32_t c0 = rd32(r->data.data() + r->l_qname), c1 = rd32(r->data.data() + r->l_qname + 4);
        if ((c0 & 15u) == 4u && (int32_t)(c0 >> 4) == r->l_seq && (c1 & 15u) == 3u) {
            const uint8_t* t = aux_find(r->aux(), r->data.data() + r->data.size(), "CG");
            if (t && t[0] == 'B' && (t[1] == 'I' || t[1] == 'i')) {
                const uint32_t n = rd32(t + 2);
                const uint8_t* ops = t + 6;
                if (n > 0 && ops + 4ull * n <= r->data.data() + r->data.size()) {
                    std::vector<uint8_t> d2;
                    d2.reserve(r->data.size() + 4ull * n);
                    d2.insert(d2.end(), r->data.begin(), r->data.begin() + r->l_qname);
                    d2.insert(d2.end(), ops, ops + 4ull * n);
                    d2.insert(d2.end(), r->seq(), t - 2);                  // seq, qual and the aux fields before CG
                    d2.insert(d2.end(), ops + 4ull * n, (const uint8_t*)(r->data.data() + r->data.size()));
                    r->data.swap(d2); r->n_cigar = n;
                }
            }
        }
    }
    return true;
}

// ---------------------------------------------------------------- BAI

bool BamIndex::load(const std::string& bam_path) {
    std::vector<std::string> cand;
    cand.push_back(bam_path + ".bai");
    if (bam_path.size() > 4) cand.push_back(bam_path.substr(0, bam_path.size() - 4) + ".bai");
    FILE* f = nullptr;
    for (const std::string& c : cand) if ((f = fopen(c.c_str(), "rb"))) break;
    if (!f) {
        for (const std::string& c : {bam_path + ".csi", bam_path.size() > 4 ? bam_path.substr(0, bam_path.size() - 4) + ".csi" : std::string()})
            if (!c.empty()) { FILE* g = fopen(c.c_str(), "rb"); if (g) { fclose(g); return load_csi(c); } }
        err_ = "index not found"; return false;
    }
    std::vector<uint8_t> d;
    uint8_t buf[1 << 16]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
    fclose(f);
    if (d.size() < 8 || memcmp(d.data(), "BAI\1", 4) != 0) { err_ = "not a BAI file"; return false; }
    size_t o = 4;
    const uint32_t n_ref = rd32(d.data() + o); o += 4;
    refs_.assign(n_ref, Ref());
    for (uint32_t i = 0; i < n_ref; ++i) {
        if (o + 4 > d.size()) { err_ = "truncated BAI"; return false; }
        const uint32_t n_bin = rd32(d.data() + o); o += 4;
        for (uint32_t b = 0; b < n_bin; ++b) {
            if (o + 8 > d.size()) { err_ = "truncated BAI"; return false; }
            const uint32_t bin = rd32(d.data() + o); const uint32_t n_chunk = rd32(d.data() + o + 4); o += 8;
            if (o + 16ull * n_chunk > d.size()) { err_ = "truncated BAI"; return false; }
            std::vector<Chunk>& v = refs_[i].bins[bin];
            for (uint32_t c = 0; c < n_chunk; ++c) { Chunk ch; ch.beg = rd64(d.data() + o); ch.end = rd64(d.data() + o + 8); o += 16; v.push_back(ch); }
        }
        if (o + 4 > d.size()) { err_ = "truncated BAI"; return false; }
        const uint32_t n_intv = rd32(d.data() + o); o += 4;
        if (o + 8ull * n_intv > d.size()) { err_ = "truncated BAI"; return false; }
        for (uint32_t k = 0; k < n_intv; ++k) { refs_[i].linear.push_back(rd64(d.data() + o)); o += 8; }
    }
    return true;
}

// CSI (coordinate-sorted index, CSIv1): BGZF-compressed; like BAI with a configurable finest bin width (2^min_shift) and
// number of levels, a left offset per bin instead of the 16-kb linear index
bool BamIndex::load_csi(const std::string& path) {
    Bgzf z;
    if (!z.open(path)) { err_ = z.error(); return false; }
    std::vector<uint8_t> d; uint8_t buf[1 << 16]; size_t n;
    while ((n = z.read_some(buf, sizeof buf)) > 0) d.insert(d.end(), buf, buf + n);
    if (!z.error().empty() && !z.eof_clean()) { err_ = "cannot read the CSI index: " + z.error(); return false; }
    if (d.size() < 16 || memcmp(d.data(), "CSI\1", 4) != 0) { err_ = "not a CSI file"; return false; }
    size_t o = 4;
    min_shift_ = (int)rd32(d.data() + o); depth_ = (int)rd32(d.data() + o + 4); const uint32_t l_aux = rd32(d.data() + o + 8); o += 12;
    if (min_shift_ < 1 || min_shift_ > 30 || depth_ < 1 || depth_ > 9 || o + l_aux + 4 > d.size()) { err_ = "bad CSI header"; return false; }
    o += l_aux;
    const uint32_t n_ref = rd32(d.data() + o); o += 4;
    refs_.assign(n_ref, Ref());
    for (uint32_t i = 0; i < n_ref; ++i) {
        if (o + 4 > d.size()) { err_ = "truncated CSI"; return false; }
        const uint32_t n_bin = rd32(d.data() + o); o += 4;
        for (uint32_t b = 0; b < n_bin; ++b) {
            if (o + 16 > d.size()) { err_ = "truncated CSI"; return false; }
            const uint32_t bin = rd32(d.data() + o); const uint64_t loff = rd64(d.data() + o + 4); const uint32_t n_chunk = rd32(d.data() + o + 12); o += 16;
            if (o + 16ull * n_chunk > d.size()) { err_ = "truncated CSI"; return false; }
            refs_[i].loffset[bin] = loff;
            std::vector<Chunk>& v = refs_[i].bins[bin];
            for (uint32_t c = 0; c < n_chunk; ++c) { Chunk ch; ch.beg = rd64(d.data() + o); ch.end = rd64(d.data() + o + 8); o += 16; v.push_back(ch); }
        }
    }
    csi_ = true;
    return true;
}

double BamIndex::span_bytes(int tid, int64_t beg, int64_t end) const {
    if (tid < 0 || (size_t)tid >= refs_.size()) return -1.0;
    const Ref& r = refs_[(size_t)tid];
    if (!r.woff_built) {
        // compressed file offset of the first record overlapping every window (virtual offset >> 16); windows without one (zero in
        // older BAI files, no bin in a CSI) take the next window's; one more edge behind the last window, a window's worth further
        std::vector<uint64_t> lin;
        if (!csi_) lin = r.linear;
        else {
            const uint32_t leaf0 = (uint32_t)(((1ull << (3 * depth_)) - 1) / 7);
            for (const auto& kv : r.loffset) if (kv.first >= leaf0 && kv.first - leaf0 < (1u << 24)) { const size_t w = (size_t)(kv.first - leaf0); if (lin.size() <= w) lin.resize(w + 1, 0); lin[w] = kv.second; }
        }
        std::vector<double>& o = r.woff; o.clear();
        if (!lin.empty()) {
            o.resize(lin.size() + 1);
            uint64_t next = 0; bool have = false;
            for (size_t i = lin.size(); i-- > 0;) { if (lin[i]) { next = lin[i]; have = true; } o[i] = have ? (double)(next >> 16) : -1.0; }
            double last = 0; for (size_t i = 0; i < lin.size(); ++i) if (o[i] >= 0) last = o[i];
            for (size_t i = 0; i < lin.size(); ++i) if (o[i] < 0) o[i] = last;                   // (trailing windows without records)
            for (size_t i = 1; i < lin.size(); ++i) if (o[i] < o[i - 1]) o[i] = o[i - 1];           // monotone (an index written out of order)
            const double step = lin.size() > 1 ? (o[lin.size() - 1] - o[0]) / (double)(lin.size() - 1) : 65536.0;
            o[lin.size()] = o[lin.size() - 1] + (step > 0 ? step : 0);
        }
        r.woff_built = true;
    }
    const std::vector<double>& o = r.woff;
    if (o.size() < 2) return -1.0;
    const int shift = csi_ ? min_shift_ : 14;
    auto at = [&](int64_t x) {
        if (x < 0) x = 0;
        const int64_t nw = (int64_t)o.size() - 1;
        const int64_t w = x >> shift;
        if (w >= nw) return o[(size_t)nw];
        const double f = (double)(x - (w << shift)) / (double)(1ll << shift);
        return o[(size_t)w] + (o[(size_t)w + 1] - o[(size_t)w]) * f;
    };
    const double v = at(end) - at(beg);
    return v > 0 ? v : 0.0;
}

std::vector<Chunk> BamIndex::query(int tid, int64_t beg, int64_t end) const {
    std::vector<Chunk> out;
    if (tid < 0 || (size_t)tid >= refs_.size() || end <= beg) return out;
    const int64_t maxpos = 1ll << (min_shift_ + 3 * depth_);
    if (end > maxpos) end = maxpos;
    if (beg >= end) return out;
    const Ref& r = refs_[(size_t)tid];
    // reg2bins (SAMv1 5.3, generalised to min_shift / depth): level l starts at bin ((1 << 3l) - 1) / 7
    std::vector<uint32_t> bins;
    const int64_t e = end - 1;
    {
        int64_t t = 0; int s = min_shift_ + 3 * depth_;
        for (int l = 0; l <= depth_; ++l) {
            for (int64_t k = t + (beg >> s); k <= t + (e >> s); ++k) bins.push_back((uint32_t)k);
            t += 1ll << (3 * l); s -= 3;
        }
    }
    uint64_t min_off = 0;
    if (!csi_) {
        const size_t li = (size_t)(beg >> 14);
        if (!r.linear.empty()) min_off = li < r.linear.size() ? r.linear[li] : r.linear.back();
    } else {
        // the left offset of the finest existing bin that contains beg (walk up from the last level)
        int64_t bin = ((1ll << (3 * depth_)) - 1) / 7 + (beg >> min_shift_);
        for (int l = depth_; l >= 0; --l) {
            auto it = r.loffset.find((uint32_t)bin);
            if (it != r.loffset.end()) { min_off = it->second; break; }
            bin = (bin - 1) >> 3;                                    // parent
        }
    }
    for (uint32_t b : bins) {
        auto it = r.bins.find(b);
        if (it == r.bins.end()) continue;
        for (const Chunk& c : it->second) if (c.end > min_off) out.push_back(c);
    }
    std::sort(out.begin(), out.end(), [](const Chunk& a, const Chunk& b) { return a.beg < b.beg; });
    std::vector<Chunk> merged;
    for (const Chunk& c : out) {
        if (!merged.empty() && c.beg <= merged.back().end) { if (c.end > merged.back().end) merged.back().end = c.end; }
        else merged.push_back(c);
    }
    return merged;
}

// ---------------------------------------------------------------- FASTA

// The index htslib's fai_load builds when <fasta>.fai is missing (faidx.c fai_build_core): name = the header line up to the
// first white space, length, offset of the first base, bases and bytes per line; every line of a sequence but its last must
// have the same length.  Written next to the FASTA like htslib does; when that is not possible the index is only kept in
// memory (htslib would give up the reference).
bool Fasta::build_index() {
    FILE* g = fopen(path_.c_str(), "rb");
    if (!g) { err_ = "cannot open " + path_; return false; }
    std::vector<std::pair<std::string, Ent> > order;
    char* line = nullptr; size_t cap = 0; ssize_t n; int64_t off = 0;
    bool have = false, last_short = false; std::string name; Ent cur{0, 0, 0, 0};
    auto close_seq = [&]() { if (have) order.emplace_back(name, cur); };
    while ((n = getline(&line, &cap, g)) >= 0) {
        if (n > 0 && line[0] == '>') {
            close_seq();
            size_t k = 1; while (k < (size_t)n && !isspace((unsigned char)line[k])) ++k;
            name.assign(line + 1, k - 1); cur = Ent{0, off + n, 0, 0}; have = true; last_short = false;
        } else if (have) {
            int64_t bases = 0; for (ssize_t i = 0; i < n; ++i) if (isgraph((unsigned char)line[i])) ++bases;
            if (bases || n > 1) {
                if (last_short && bases) { err_ = "different line length in sequence '" + name + "' of " + path_; free(line); fclose(g); return false; }
                if (cur.linebases == 0) { cur.linebases = bases; cur.linewidth = n; }
                else if (bases != cur.linebases || n != cur.linewidth) {
                    if (bases > cur.linebases) { err_ = "different line length in sequence '" + name + "' of " + path_; free(line); fclose(g); return false; }
                    last_short = true;
                }
                cur.len += bases;
            }
        }
        off += n;
    }
    close_seq();
    free(line); fclose(g);
    if (order.empty()) { err_ = "no sequence in " + path_; return false; }
    for (auto& kv : order) { if (kv.second.linebases == 0) { kv.second.linebases = 1; kv.second.linewidth = 1; } if (!idx_.count(kv.first)) idx_[kv.first] = kv.second; }
    if (FILE* w = fopen((path_ + ".fai").c_str(), "w")) {
        for (const auto& kv : order) fprintf(w, "%s\t%lld\t%lld\t%lld\t%lld\n", kv.first.c_str(), (long long)kv.second.len, (long long)kv.second.off, (long long)kv.second.linebases, (long long)kv.second.linewidth);
        fclose(w);
    }
    return true;
}

bool Fasta::open(const std::string& path) {
    path_ = path;
    FILE* f = fopen((path + ".fai").c_str(), "r");
    if (!f) return build_index();                                    // (fai_load / samfaipath build it, bamreadcount.cpp:501-506)
    char name[4096]; long long len, off, lb, lw;                     // (a contig written on one line: its width can pass 2^31)
    char line[8192];
    while (fgets(line, sizeof line, f)) {
        if (sscanf(line, "%4095s %lld %lld %lld %lld", name, &len, &off, &lb, &lw) == 5 && lb > 0 && lw >= lb) { Ent e; e.len = len; e.off = off; e.linebases = lb; e.linewidth = lw; idx_[name] = e; }
    }
    fclose(f);
    FILE* g = fopen(path.c_str(), "rb");
    if (!g) { err_ = "cannot open " + path; return false; }
    fclose(g);
    return true;
}

bool Fasta::fetch(const std::string& name, std::string* seq) {
    auto it = idx_.find(name);
    if (it == idx_.end()) { err_ = "sequence " + name + " not in FASTA index"; return false; }
    const Ent& e = it->second;
    FILE* f = fopen(path_.c_str(), "rb");
    if (!f) { err_ = "cannot open " + path_; return false; }
    // an index entry must fit the file it describes (a damaged or foreign .fai must not turn into a huge allocation)
    fseeko(f, 0, SEEK_END); const int64_t fsize = (int64_t)ftello(f);
    if (e.len < 0 || e.off < 0 || e.linebases <= 0 || e.linewidth < e.linebases || e.off > fsize || e.len > fsize - e.off) {
        fclose(f); err_ = "the FASTA index entry of " + name + " does not fit " + path_; return false;
    }
    seq->clear(); seq->reserve((size_t)e.len);
    if (fseeko(f, (off_t)e.off, SEEK_SET) != 0) { fclose(f); err_ = "seek failed"; return false; }
    const int64_t nlines = (e.len + e.linebases - 1) / e.linebases;
    const size_t raw = (size_t)std::min<__int128>((__int128)nlines * e.linewidth, (__int128)(fsize - e.off));
    std::vector<char> buf(raw + 1);
    const size_t got = fread(buf.data(), 1, raw, f);
    fclose(f);
    // the index promises uniform lines (linebases of linewidth bytes): copy line by line
    seq->resize((size_t)e.len);
    int64_t done = 0;
    for (int64_t l = 0; l < nlines && done < e.len; ++l) {
        const size_t o = (size_t)(l * e.linewidth);
        if (o >= got) break;
        const size_t n = (size_t)std::min<int64_t>(std::min<int64_t>(e.linebases, e.len - done), (int64_t)(got - o));
        memcpy(&(*seq)[(size_t)done], buf.data() + o, n);
        done += (int64_t)n;
    }
    seq->resize((size_t)done);
    return true;
}

Fasta::~Fasta() { if (map_) munmap((void*)map_, (size_t)map_len_); }

int64_t Fasta::read_range(const std::string& name, int64_t beg, int64_t n, char* dst, int64_t* contig_len) {
    auto it = idx_.find(name);
    if (it == idx_.end()) { err_ = "sequence " + name + " not in FASTA index"; return -1; }
    const Ent& e = it->second;
    if (!map_tried_) {
        map_tried_ = true;
        const int fd = ::open(path_.c_str(), O_RDONLY);
        if (fd >= 0) {
            struct stat st;
            if (fstat(fd, &st) == 0 && st.st_size > 0) {
                void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m != MAP_FAILED) { map_ = (const char*)m; map_len_ = (int64_t)st.st_size; }
            }
            ::close(fd);
        }
    }
    if (!map_) { err_ = "cannot map " + path_; return -1; }
    // the same entry checks as fetch(); the length fetch() would return: the index's, cut where the file ends
    if (e.len < 0 || e.off < 0 || e.linebases <= 0 || e.linewidth < e.linebases || e.off > map_len_ || e.len > map_len_ - e.off) { err_ = "the FASTA index entry of " + name + " does not fit " + path_; return -1; }
    int64_t have = e.len;
    {   // bases actually present: whole lines of linebases, the last line as far as the file goes
        const int64_t avail = map_len_ - e.off;
        const int64_t full = avail / e.linewidth, rest = std::min<int64_t>(avail - full * e.linewidth, e.linebases);
        have = std::min<int64_t>(e.len, full * e.linebases + rest);
    }
    if (contig_len) *contig_len = have;
    if (beg < 0) beg = 0;
    if (beg >= have || n <= 0) return 0;
    if (n > have - beg) n = have - beg;
    int64_t done = 0;
    while (done < n) {
        const int64_t x = beg + done, line = x / e.linebases, col = x % e.linebases;
        const int64_t k = std::min<int64_t>(e.linebases - col, n - done);
        memcpy(dst + done, map_ + e.off + line * e.linewidth + col, (size_t)k);
        done += k;
    }
    return done;
}

}  // namespace brcio
