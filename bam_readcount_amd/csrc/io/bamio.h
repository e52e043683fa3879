// bamio.h — host-side input layer of the drop-in CLI: BGZF, BAM records, BAI region queries, FASTA + .fai.
//
// The reference delegates all of this to samtools-1.10/htslib-1.10 (bamreadcount.cpp:16-19: samopen, samfetch,
// sam_index_load3, fai_load/fai_fetch, bam_get_library).  htslib is not available in this environment, so this is a
// from-scratch reader of the public wire formats (SAMv1 spec sections 4.1 BGZF, 4.2 BAM, 5.2 BAI, CSIv1; faidx format).  It
// only implements what the readcount path needs: sequential inflate of BGZF blocks, seeking by virtual offset, the
// overlap query of an indexed region, whole-contig FASTA fetch, and the @SQ/@RG header fields.
#ifndef BRC_BAMIO_H
#define BRC_BAMIO_H

#include <stdint.h>
#include <stdio.h>

#include <map>
#include <string>
#include <vector>

namespace brcio {

// ---------------------------------------------------------------- BGZF
class Bgzf {
  public:
    bool open(const std::string& path);
    void close();
    ~Bgzf() { close(); }
    // virtual offset = compressed block offset << 16 | offset inside the uncompressed block
    bool seek(uint64_t voffset);
    uint64_t tell() const { return (block_coff_ << 16) | (uint64_t)pos_; }
    // read exactly n bytes (crossing blocks); returns false at EOF / error
    bool read(void* dst, size_t n);
    size_t read_some(void* dst, size_t n);         // up to n bytes; fewer only at EOF / error
    bool eof_clean() const { return eof_; }
    const std::string& error() const { return err_; }

  private:
    bool load_block(uint64_t coff);
    FILE* f_ = nullptr;
    std::vector<uint8_t> cbuf_, ubuf_;
    uint64_t block_coff_ = 0, next_coff_ = 0;
    size_t pos_ = 0, len_ = 0;
    bool eof_ = false;
    void* ld_ = nullptr;             // libdeflate decompressor of this handle (when the library is present)
    std::string err_;
};

// ---------------------------------------------------------------- BAM
struct BamRecord {
    int32_t tid = -1, pos = -1, l_seq = 0, mtid = -1, mpos = -1, tlen = 0;
    uint16_t flag = 0, bin = 0;
    uint32_t n_cigar = 0;            // (32 bits: a CIGAR of more than 65535 operators arrives in the CG tag, SAMv1 4.2.2)
    uint8_t mapq = 0;
    std::vector<uint8_t> data;       // the variable part: qname\0, cigar, seq, qual, aux
    uint32_t l_qname = 0;
    const char* qname() const { return (const char*)data.data(); }
    const uint32_t* cigar() const { return (const uint32_t*)(data.data() + l_qname); }
    const uint8_t* seq() const { return data.data() + l_qname + 4u * n_cigar; }
    const uint8_t* qual() const { return seq() + ((int64_t)l_seq + 1) / 2; }
    const uint8_t* aux() const { return qual() + l_seq; }
    size_t aux_len() const { return data.size() - (size_t)(aux() - data.data()); }
    int32_t endpos() const;          // bam_endpos
    // integer aux tag (types c C s S i I); returns false when absent or not an integer
    bool aux_int(const char tag[2], int32_t* out) const;
    // Z aux tag
    const char* aux_str(const char tag[2]) const;
};

struct BamHeader {
    std::string text;
    std::vector<std::string> names;
    std::vector<int32_t> lengths;
    std::map<std::string, int> name2tid;
    std::map<std::string, std::string> rg2lb;      // read group -> library as bam_get_library resolves it (bamreadcount.cpp:280)
    std::vector<std::string> expected;             // what find_library_names collects (:92-111), sorted: the "Expect library" lines
    std::vector<std::string> libraries() const;    // sorted unique libraries a read can resolve to
    void parse_read_groups();                      // fills rg2lb and expected from text
};

struct Chunk { uint64_t beg, end; };

class BamIndex {
  public:
    bool load(const std::string& bam_path);        // <bam>.bai, <bam minus .bam>.bai, then <bam>.csi (SAMv1 5.2 / CSIv1)
    // chunks (virtual offset ranges) that may hold records overlapping [beg,end) on tid, merged and sorted
    std::vector<Chunk> query(int tid, int64_t beg, int64_t end) const;
    // Estimated compressed bytes of the records that START in [beg, end) of tid: differences of the file offsets the index keeps per
    // window (BAI: the 16-kb linear index; CSI: the left offsets of its finest bins), linear inside a window.  What a caller that has
    // nothing but the index can balance work by ("balanced by estimated event count — BAI linear-index / chunk sizes").
    // < 0: the index holds no such offsets for this contig.
    double span_bytes(int tid, int64_t beg, int64_t end) const;
    const std::string& error() const { return err_; }

  private:
    struct Ref { std::map<uint32_t, std::vector<Chunk> > bins; std::vector<uint64_t> linear; std::map<uint32_t, uint64_t> loffset;
                 mutable std::vector<double> woff; mutable bool woff_built = false; };      // span_bytes: compressed offset at every window edge (built on first use)
    bool load_csi(const std::string& path);
    std::vector<Ref> refs_;
    int min_shift_ = 14, depth_ = 5;               // BAI's fixed binning; a CSI file carries its own
    bool csi_ = false;
    std::string err_;
};

class BamReader {
  public:
    bool open(const std::string& path);
    const BamHeader& header() const { return hdr_; }
    bool next(BamRecord* r);                        // sequential; false at EOF
    // samfetch: every record with tid == tid, pos < end, endpos > beg (beg clamped at 0), in file order
    template <class F>
    bool fetch(const BamIndex& idx, int tid, int64_t beg, int64_t end, F cb) {
        if (beg < 0) beg = 0;
        if (end <= beg) return true;
        err_.clear();
        BamRecord r;
        for (const Chunk& c : idx.query(tid, beg, end)) {
            if (!bg_.seek(c.beg)) return false;
            while (bg_.tell() < c.end) {
                if (!next(&r)) {
                    // the end of the file inside a chunk is not an error (the index may point past the last record); a
                    // truncated record, a failed inflate or a bad block is — the caller must not print a partial result
                    if (bg_.eof_clean() && err_.empty()) break;
                    if (err_.empty()) err_ = bg_.error().empty() ? "read error" : bg_.error();
                    return false;
                }
                if (r.tid != tid || r.pos >= end) { if (r.tid > tid || (r.tid == tid && r.pos >= end)) return true; continue; }
                if (r.endpos() > beg) cb(r);
            }
        }
        return true;
    }
    const std::string& error() const { return err_; }

  private:
    Bgzf bg_;
    BamHeader hdr_;
    std::string err_;
};

// ---------------------------------------------------------------- FASTA + .fai
class Fasta {
  public:
    bool open(const std::string& path);            // reads <path>.fai, or builds (and writes) it like fai_load does
    // whole contig, raw characters (case preserved), like fai_fetch(fai, name, &len)
    bool fetch(const std::string& name, std::string* seq);
    // Bases [beg, beg + n) of a contig as raw characters, WITHOUT loading the contig (the file is mapped once): the site-list
    // planner needs a few hundred bases around each -l line, and a whole-genome list visits every contig — fai_fetch of 24
    // chromosomes for 100 000 one-base windows was a sixth of the run.  Returns the bases copied (fewer where the contig — or the
    // file behind a longer index entry — ends), -1 for an unknown contig or an unreadable file; *contig_len = the length a whole
    // fetch() would return.
    int64_t read_range(const std::string& name, int64_t beg, int64_t n, char* dst, int64_t* contig_len);
    ~Fasta();
    const std::string& error() const { return err_; }

  private:
    struct Ent { int64_t len, off, linebases, linewidth; };
    bool build_index();
    std::map<std::string, Ent> idx_;
    std::string path_, err_;
    const char* map_ = nullptr; int64_t map_len_ = 0; bool map_tried_ = false;      // read_range: the file, mapped read-only
};

// ---------------------------------------------------------------- CRAM 3.0 (minimal, cram.cpp)
// The reference reads CRAM through htslib's samopen (bamreadcount.cpp:411, test-data/twolib.sorted.cram); this reader
// covers CRAM 3.0 — external, embedded or no reference; raw / gzip / rANS 4x8 / bzip2 / lzma blocks.  Records come out in BAM layout, mapped
// ones with the NM tag htslib's decoder generates when the file does not store it.
class CramReader {
  public:
    CramReader();
    ~CramReader();
    CramReader(const CramReader&) = delete;
    CramReader& operator=(const CramReader&) = delete;
    static bool is_cram(const std::string& path);
    bool open(const std::string& path, Fasta* fasta);   // fasta: needed to reconstruct mapped reads
    const BamHeader& header() const;
    template <class F>
    bool fetch(int tid, int64_t beg, int64_t end, F cb) {
        if (beg < 0) beg = 0;
        if (end <= beg) return true;
        return fetch_impl(tid, beg, end, [](void* c, const BamRecord& r) { (*(F*)c)(r); }, &cb);
    }
    const std::string& error() const;
    // The bases of contig `tid`, already loaded by the caller and alive until the next call of this function: used instead of a copy of
    // the reader's own (several readers decoding stripes of one region side by side would each load the whole contig).  nullptr: none.
    void share_reference(int tid, const std::string* bases);

  private:
    struct Impl;
    bool fetch_impl(int tid, int64_t beg, int64_t end, void (*thunk)(void*, const BamRecord&), void* ctx);
    Impl* d_;
};

}  // namespace brcio
#endif
