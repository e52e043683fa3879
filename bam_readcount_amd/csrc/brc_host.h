// brc_host.h — host side of the engine above the device pipeline (pure C++, no HIP).
//
//   Staged        region inputs accumulated by brc_push_reads in (pinned) host memory, offsets rebased
//   Backend       what the C-ABI glue needs from a device pipeline; the product implements it with HIP
//                 (brc_engine.hip).  tests/sim/ implements it with the CPU lane simulator — never shipped.
//   assemble / format   brc_fetch_result's allele ordering and brc_format_region's record assembly
#ifndef BRC_HOST_H
#define BRC_HOST_H

#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/brc.h"
#include "brc_core.h"

namespace brc {

struct HostAlloc {
    void* (*alloc)(size_t bytes);
    void (*release)(void* p);
};

// Grow-only host array living in backend-provided (pinned) memory.
template <class T>
struct HBuf {
    T* p = nullptr;
    size_t n = 0, cap = 0;
    const HostAlloc* A = nullptr;
    bool reserve(size_t want) {
        if (want <= cap) return true;
        // (pinned allocations are expensive: double, and never start small)
        size_t c = cap ? 2 * cap : (size_t)(1u << 16);
        if (c < want) c = want + want / 8;
        T* q = (T*)A->alloc(c * sizeof(T));
        if (!q) return false;
        if (n) memcpy(q, p, n * sizeof(T));
        if (p) A->release(p);
        p = q; cap = c;
        return true;
    }
    bool append(const T* src, size_t k) {
        if (!reserve(n + k + 16)) return false;
        if (k) memcpy(p + n, src, k * sizeof(T));
        n += k;
        return true;
    }
    void clear() { n = 0; }
    void destroy() { if (p) A->release(p); p = nullptr; n = cap = 0; }
};

struct Staged {
    HBuf<int32_t> pos; HBuf<uint16_t> flag; HBuf<uint8_t> mapq; HBuf<int16_t> lib; HBuf<int32_t> l_qseq;
    HBuf<uint32_t> n_cigar; HBuf<uint64_t> cig_off, seq_off, qual_off; HBuf<int32_t> nm, sm; HBuf<uint8_t> tags;
    HBuf<uint32_t> cigar; HBuf<uint8_t> seq4, qual;
    // brc_push_reads_pinned: arenas the caller keeps in page-locked memory of its own — logical bytes [off, off + n) of seq4 / qual live at
    // p (nothing is copied into the HBufs above, which stay empty for such a region); seq_total / qual_total are the arenas' logical
    // lengths in either case (what the next batch's offsets are rebased by, what the device buffers are sized for)
    struct Seg { const uint8_t* p; uint64_t off, n; };
    std::vector<Seg> seq_seg, qual_seg;
    uint64_t seq_total = 0, qual_total = 0;
    bool adopted() const { return !seq_seg.empty() || !qual_seg.empty(); }
    static const uint8_t* seg_at(const std::vector<Seg>& v, uint64_t off) {
        size_t lo = 0, hi = v.size();                              // last segment whose off <= the wanted offset
        while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (v[mid].off <= off) lo = mid; else hi = mid; }
        return v[lo].p + (off - v[lo].off);
    }
    const uint8_t* seq_at(uint64_t off) const { return seq_seg.empty() ? seq4.p + off : seg_at(seq_seg, off); }
    const uint8_t* qual_at(uint64_t off) const { return qual_seg.empty() ? qual.p + off : seg_at(qual_seg, off); }
    HBuf<uint64_t> bq_row;              // per read: first element of its 16-byte-aligned row in the device's event-byte stream
    uint64_t bq_elems = 0;              // total elements of the event-byte stream (sum of roundup16(l_qseq))
    // the sparse wide stream (brc_core.h: DevIn.bqw): wide[i] != 0 when read i has a base its event byte cannot describe
    // (read_has_escape at push: eb_make's predicate on the bytes the device will see) — only those reads get a wide row.
    // wide_layout: one pair per such read, in file order — (the read, first16 + its wide row's start / 16: the table in front of the
    // rows takes first16 units of 16 elements); returns the elements of all wide rows.
    HBuf<uint8_t> wide;
    struct WidePair { uint32_t read, w16; };
    uint64_t wide_layout(WidePair* pairs /* n_wide() of them; nullptr: only the sum */, uint32_t first16) const;
    size_t n_wide() const { size_t k = 0; for (int64_t i = 0; i < n; ++i) k += wide.p[i] != 0; return k; }
    // KB v2: pieces (walk_pieces in brc_core.h).  piece_cnt is filled at push time; piece_off (library-major slot of a
    // read's first piece) and lib_base (first slot of every library's stream, Lp + 1 entries) at upload
    HBuf<char> qnames; HBuf<uint64_t> qname_off;   // read names when the caller gave them (warning text only); qname_off[i] = ~0 without
    HBuf<uint32_t> piece_cnt, piece_off;
    HBuf<uint32_t> iev_off;             // per read: its first slot in the raw indel-event list (one slot per I / D / P operator; n_indel_ops in all)
    std::vector<int64_t> lib_base;
    int64_t n_pieces = 0;
    int32_t max_lqseq = 0;
    uint32_t max_ncigar = 0;        // most CIGAR operators of a staged read
    bool has_eqx = false;           // a read with an = or X operator was staged (the wave-form annotator's EQX instantiations)
    bool has_empty_m = false;       // a mapped read with an M / = / X operator of length zero was staged (brc_core.h: cursor_resolve)
    int64_t max_span = 0;               // longest reference span of a pushed read
    void layout_pieces(int Lp, bool per_lib);
    uint32_t len_hist[TABLE_MAX + 1] = {0};   // histogram of l_qseq <= TABLE_MAX (modal length -> DevCfg.table_len)
    int32_t modal_len() const { uint32_t best = 0; int32_t arg = 0; for (int l = 1; l <= TABLE_MAX; ++l) if (len_hist[l] > best) { best = len_hist[l]; arg = l; } return arg; }
    int64_t n = 0;
    int64_t min_pos = 0, max_end = 0;   // extent of reads that enter the pileup
    uint64_t n_indel_ops = 0;           // I / D / P operators of all reads = slots of the raw indel-event list (an upper bound on the events)
    std::vector<int32_t> win_beg, win_end;   // brc_region_windows: the only windows [beg - 1, end) of the region anybody will format (empty: all of it)
    // per 64-position tile of the planes [pos0, pos0 + P): TILE_UNWANTED, or the first and last lane any window [beg - 1, end) asks
    // for (lo | hi << 8) — the tile is piled up for those lanes only (empty vector: no hint, everything is wanted)
    std::vector<uint16_t> wanted_tiles(int32_t pos0, int64_t P) const;
    void init(const HostAlloc* A);
    void clear();
    void destroy();
};

struct Geometry {
    int32_t tid = 0, beg0 = 0, end = 0;
    int32_t pos0 = 0; int64_t P = 0; int Lp = 1;
    int64_t PS = 0;                     // plane stride chosen by the backend at upload (>= P)
    const char* ref = nullptr; int64_t ref_len = 0;
    int64_t ref_lo = 0, ref_hi = 0;     // slice of the contig the device needs
};

// Host view of the device results after fetch: the compact slot planes + the folded third-allele records (struct Planes, XAgg, brc_core.h);
// brc_fetch_result expands them to the ABI's dense planes (expand_slots).
struct HostPlanes {
    uint32_t *ncol = nullptr, *depth = nullptr, *slotid = nullptr, *si = nullptr, *unavail = nullptr;
    float* sf = nullptr;
    const XAgg* xagg = nullptr; uint64_t n_xagg = 0;       // third-allele records, folded on the device: one per (position, library, bucket), grouped by (64-position tile, library)
    const IndelOut* indel = nullptr; int64_t n_indel = 0;
    uint64_t n_events = 0, n_positions = 0;
    uint64_t warn[BRC_N_WARN] = {0, 0, 0, 0};
};

// The region's text as the device wrote it (BRC_OPT_DEVICE_TEXT): line of plane index k = text[off[k] .. off[k + 1]), empty
// for positions that print nothing; n + 1 offsets.
struct HostText { const char* text = nullptr; const uint32_t* off = nullptr; uint64_t total = 0; int64_t n = 0;
                  const uint32_t* last_processed = nullptr; };      // [Lp] the last plane index every library was processed at (NONE32: never)

enum { BRC_TEXT_TOO_LONG = 1000 };      // Backend::text_begin only (not an ABI code)

class Backend {
  public:
    virtual ~Backend() {}
    virtual const HostAlloc* host_alloc() = 0;
    // the backend uploads adopted arenas (Staged::seq_seg / qual_seg) from where they lie; false: brc_push_reads_pinned copies like
    // brc_push_reads (the CPU lane simulator, whose "device" pointers ARE the staging arrays)
    virtual bool adopts_arenas() const { return false; }
    virtual int upload(const brc_config& cfg, const Staged& s, Geometry& g) = 0;         // staging -> device; sets g.PS
    virtual int compute(brc_timing* t) = 0;                                              // whole pipeline, waits
    // n passes queued back to back with one wait (default: n waits); t = per-kernel times averaged over the passes
    virtual int compute_n(int n, brc_timing* t) {
        brc_timing acc; memset(&acc, 0, sizeof acc);
        for (int i = 0; i < n; ++i) { brc_timing one; memset(&one, 0, sizeof one); const int rc = compute(&one); if (rc) return rc; for (int k = 0; k < BRC_NKERNEL; ++k) acc.ms[k] += one.ms[k]; acc.total_ms += one.total_ms; }
        if (t && n > 0) { for (int k = 0; k < BRC_NKERNEL; ++k) t->ms[k] = acc.ms[k] / (float)n; t->total_ms = acc.total_ms / (float)n; }
        return BRC_OK;
    }
    virtual int fetch(HostPlanes* out, bool planes) = 0;                                 // device -> host: counters, third-allele and indel lists, and (planes) the slot planes
    // the slot planes of plane indices [k0, k0 + n) only (host planes `*stride` elements apart), and the region's two lists whole
    virtual int fetch_window(int64_t k0, int64_t n, HostPlanes* out, int64_t* stride) = 0;
    // device-side text of the computed region: text_begin launches the line kernels and starts the download (the device
    // buffers of the region stay untouched until it is done: same stream), text_wait waits for it
    // (two host buffers: `slot` names the one this region's text goes to — the text of the region before may still be
    // in the writer's hands while the next region's download is already running)
    // (BRC_TEXT_TOO_LONG: the region's lines would pass the 4 GiB its 32-bit offsets can address — nothing was started, the caller
    // formats this region on the host)
    virtual int text_begin(const std::string& chrom, const std::vector<std::string>& libs, int* slot) = 0;
    virtual int text_wait(int slot, HostText* out) = 0;
    virtual int reserve_text(size_t) { return BRC_OK; }
    virtual void list_sizes(uint64_t* n_xev, uint64_t* n_indel_slots) { *n_xev = 0; *n_indel_slots = 0; }   // of the last compute                                  // room for the text of coming regions (optional)
    virtual int counts(uint64_t* n_events, uint64_t* n_positions) = 0;
    // piece-steps of the last compute: what the tile ranges hold / what the pileup kernel walked (0, 0: not counted — no compaction ran)
    virtual void piece_steps(uint64_t* ranged, uint64_t* walked) { *ranged = 0; *walked = 0; }
    virtual const char* last_error() const = 0;
};

// Provided by exactly one translation unit per library: brc_engine.hip (product) or tests/sim/brc_sim.cpp.
Backend* make_backend(const brc_config& cfg, int* err);
void* backend_host_alloc(size_t bytes);          // brc_host_alloc / brc_host_free
void backend_host_free(void* p);
const char* backend_kind();
const char* backend_kernel_name(int k);

// slot planes + event list -> dense planes istat[Lp][6][9][PS], fstat[Lp][6][4][PS] (every element written); multi-threaded
void expand_slots(const HostPlanes& hp, int Lp, int64_t P, int64_t PS, uint32_t* istat, float* fstat);

// CPUs the process may use at once (affinity, cgroup quota): sizes the host-side thread pools
unsigned effective_cpus();

// Test knobs (tests/test_gpu_parity.py: small K -> flushes, a small packing limit -> PF_HUGE, a forced dominant bucket -> third
// alleles, tiny third-allele lists -> grow and recompute, no quotient tables, either indel bucket size, a low device-text limit).
// They change which device path runs, never a result — and they exist only in libbrc_hip_testknobs.so (brc_knobs.cpp compiled
// with -DBRC_TEST_KNOBS, which maps an index to an environment variable): in the product test_knob() is the constant nullptr and
// the library contains neither the names nor a getenv for them (tests/test_abi.py) — an inherited BRC_NO_TABLE=1 cannot turn the
// shipped kernel into its slow path without a word.
enum TestKnob { TK_NO_TABLE = 0, TK_FLUSH_K, TK_PACK_LIM, TK_FORCE_DOM, TK_IBUCKET_SHIFT, TK_XEV_CAP, TK_DEVICE_TEXT_LIMIT, TK_FORMAT_THREADS, TK_FORMAT_CHUNK, TK_COMPACT, TK_WAVE_FORM, TK_N };
const char* test_knob(int which);

// exact "%.2f" of a float (== iostream fixed/setprecision(2), BasicStat.cpp:116); returns bytes written
int fmt_f2(char* out, float v);
int fmt_u32(char* out, uint32_t v);

}  // namespace brc
#endif
