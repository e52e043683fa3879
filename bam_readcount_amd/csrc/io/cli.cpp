// cli.cpp — `bam-readcount`-compatible command line over the C-ABI engine (include/brc.h).
//
// Mirrors main() of the reference (src/exe/bam-readcount/bamreadcount.cpp:421-670): same options and spellings
// (boost::program_options unix style: -q20, -q 20, --min-mapping-quality=20, --min-mapping-quality 20, sticky short
// switches -pi, unambiguous long prefixes), same positional arguments (BAM then regions), same stdout records, the same
// informational stderr lines, same -l site-list and region semantics including the "fetch one base early" rule (:602).
// What differs is below the callbacks: reads are batched and handed to the MI355X engine instead of bam_plbuf.
//
// Not reproduced (documented in DESIGN.md): whole-file mode without region or site list (the reference marks it FIXME
// and produces degraded output) is refused.
//
// Engine-side options (not in the reference): --brc-chunk (tiling of long regions), --brc-plan (site-list planner),
// --brc-gpus N / BRC_DEVICES=0,1,.. and --brc-streams K (K engines per GPU, each with a worker thread; work items — region
// pieces, site-list batches — are dealt out in file order and their text is written in file order).  Every engine
// pipelines consecutive pieces: piece k is formatted and written while piece k+1 is staged and on the GPU and piece k+2
// is being decoded.
//
// --brc-ranks N (one rank per GPU): the command line becomes N PROCESSES, started before any of them has touched
// the HIP runtime; rank r binds to GPU devices[r] and to its share of the CPUs, takes a contiguous, event-weighted slice of the
// work list IN FILE ORDER (the rule of shard.partition; weights from the index's file offsets, BamIndex::span_bytes) and prints
// it; the coordinating process writes the ranks' text and stderr in rank order — see "ranks" below.
#include <errno.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <functional>
#include <future>
#include <mutex>
#include <string>
#include <vector>

#include "../../../include/brc.h"
#include "bamio.h"

using namespace brcio;

struct Options {
    bool help = false, version = false, per_lib = false, insertion_centric = false, distribution = false;
    unsigned seen = 0;                 // options given so far
    int min_mapq = 0, min_bq = 0, max_cnt = 10000000;
    long long max_warnings = -1;
    std::string site_list, fasta, bam;
    std::vector<std::string> regions;
    long long chunk_bp = 1000000;   // engine-side tiling of long regions (not a reference option): --brc-chunk
    bool chunk_given = false;       // ... given on the command line (else: sized by the data under a region, run_region: auto_chunk)
    long long plan_sites = 4096;    // site-list planner: -l lines batched per engine pass (0 = one pass per line): --brc-plan
    long long gpus = 1;             // GPUs: --brc-gpus
    long long streams = 0;          // engines per GPU (0: one; every engine already overlaps decode | GPU | format of consecutive pieces): --brc-streams
    long long ranks = 0;            // processes, one per GPU: --brc-ranks (BRC_RANKS)
    std::string rank_of;            // "r:N": this process IS rank r of N (what the coordinating process starts its children with): --brc-rank-of
    std::string tmpdir;             // where the ranks behind the first keep their text until its turn comes: --brc-tmpdir (TMPDIR, /tmp)
};

static const char* kUsage =
    "Usage: bam-readcount [OPTIONS] bam_file|cram_file [region]\n"
    "Generate metrics for bam_file at single nucleotide positions.\n"
    "Example: bam-readcount -f ref.fa some.bam|some.cram\n\n"
    "Available options:\n"
    "  -h [ --help ]                         produce this message\n"
    "  -v [ --version ]                      output the version number\n"
    "  -q [ --min-mapping-quality ] arg (=0) minimum mapping quality of reads used \n"
    "                                        for counting.\n"
    "  -b [ --min-base-quality ] arg (=0)    minimum base quality at a position to \n"
    "                                        use the read for counting.\n"
    "  -d [ --max-count ] arg (=10000000)    max depth to avoid excessive memory \n"
    "                                        usage.\n"
    "  -l [ --site-list ] arg                file containing a list of regions to \n"
    "                                        report readcounts within.\n"
    "  -f [ --reference-fasta ] arg          reference sequence in the fasta format.\n"
    "  -D [ --print-individual-mapq ] arg    report the mapping qualities as a comma \n"
    "                                        separated list.\n"
    "  -p [ --per-library ]                  report results by library.\n"
    "  -w [ --max-warnings ] arg             maximum number of warnings of each type \n"
    "                                        to emit. -1 gives an unlimited number.\n"
    "  -i [ --insertion-centric ]            generate indel centric readcounts. Reads \n"
    "                                        containing insertions will not be \n"
    "                                        included in per-base counts\n\n";

struct OptSpec { char s; const char* l; bool takes_value; };
static const OptSpec kSpecs[] = {
    {'h', "help", false}, {'v', "version", false}, {'q', "min-mapping-quality", true}, {'b', "min-base-quality", true},
    {'d', "max-count", true}, {'l', "site-list", true}, {'f', "reference-fasta", true}, {'D', "print-individual-mapq", true},
    {'p', "per-library", false}, {'w', "max-warnings", true}, {'i', "insertion-centric", false}, {0, "brc-chunk", true}, {1, "brc-plan", true}, {2, "brc-gpus", true}, {3, "brc-streams", true},
    {4, "brc-ranks", true}, {5, "brc-rank-of", true}, {6, "brc-tmpdir", true},
};

static bool apply(Options& o, const OptSpec& sp, const std::string& v, std::string* err) {
    auto to_ll = [&](long long* out) {
        char* e = nullptr; errno = 0; const long long x = strtoll(v.c_str(), &e, 10);
        if (errno || e == v.c_str() || *e) { *err = "the argument ('" + v + "') for option '--" + sp.l + "' is invalid"; return false; }
        *out = x; return true;
    };
    // (boost::program_options: none of the reference's options is composing — a second occurrence is an error, :463)
    const unsigned bit = 1u << (unsigned)(&sp - kSpecs);
    if (o.seen & bit) { *err = std::string("option '--") + sp.l + "' cannot be specified more than once"; return false; }
    o.seen |= bit;
    long long x = 0;
    switch (sp.s) {
        case 'h': o.help = true; return true;
        case 'v': o.version = true; return true;
        case 'p': o.per_lib = true; return true;
        case 'i': o.insertion_centric = true; return true;
        case 'q': if (!to_ll(&x)) return false; o.min_mapq = (int)x; return true;
        case 'b': if (!to_ll(&x)) return false; o.min_bq = (int)x; return true;
        case 'd': if (!to_ll(&x)) return false; o.max_cnt = (int)x; return true;
        case 'w': if (!to_ll(&x)) return false; o.max_warnings = x; return true;
        case 'l': o.site_list = v; return true;
        case 'f': o.fasta = v; return true;
        case 'D': o.distribution = (v == "1" || v == "true" || v == "yes" || v == "on"); return true;
        case 1: if (!to_ll(&x)) return false; o.plan_sites = x; return true;
        case 2: if (!to_ll(&x)) return false; o.gpus = x; return true;
        case 3: if (!to_ll(&x)) return false; o.streams = x; return true;
        case 4: if (!to_ll(&x)) return false; o.ranks = x; return true;
        case 5: o.rank_of = v; return true;
        case 6: o.tmpdir = v; return true;
        default: if (!to_ll(&x)) return false; o.chunk_bp = x; o.chunk_given = true; return true;
    }
}

static bool parse_args(int argc, char** argv, Options& o, std::string* err) {
    std::vector<std::string> pos;
    bool only_pos = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (only_pos || a.size() < 2 || a[0] != '-') { pos.push_back(a); continue; }
        if (a == "--") { only_pos = true; continue; }
        if (a[1] == '-') {                                   // long option, optional =value, unambiguous prefix
            const size_t eq = a.find('=');
            const std::string name = a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
            const OptSpec* hit = nullptr; int nhit = 0;
            for (const OptSpec& sp : kSpecs) {
                if (name == sp.l) { hit = &sp; nhit = 1; break; }
                if (strncmp(sp.l, name.c_str(), name.size()) == 0) { hit = &sp; ++nhit; }
            }
            if (nhit == 0) { *err = "unrecognised option '" + a + "'"; return false; }
            if (nhit > 1) { *err = "option '" + a + "' is ambiguous"; return false; }
            std::string v;
            if (hit->takes_value) {
                if (eq != std::string::npos) v = a.substr(eq + 1);
                else if (i + 1 < argc) v = argv[++i];
                else { *err = std::string("the required argument for option '--") + hit->l + "' is missing"; return false; }
            } else if (eq != std::string::npos) { *err = "option '--" + name + "' does not take any arguments"; return false; }
            if (!apply(o, *hit, v, err)) return false;
            continue;
        }
        for (size_t k = 1; k < a.size(); ++k) {              // short options, sticky
            const OptSpec* hit = nullptr;
            for (const OptSpec& sp : kSpecs) if (sp.s >= 'A' && sp.s == a[k]) hit = &sp;      // (codes below 'A' name the long-only --brc-* options)
            if (!hit) { *err = std::string("unrecognised option '-") + a[k] + "'"; return false; }
            if (hit->takes_value) {
                std::string v = a.substr(k + 1);
                if (v.empty()) { if (i + 1 < argc) v = argv[++i]; else { *err = std::string("the required argument for option '--") + hit->l + "' is missing"; return false; } }
                if (!apply(o, *hit, v, err)) return false;
                break;
            }
            if (!apply(o, *hit, "", err)) return false;
        }
    }
    if (!pos.empty()) { o.bam = pos[0]; o.regions.assign(pos.begin() + 1, pos.end()); }
    return true;
}

// The two big arenas of a batch (SEQ, QUAL: 225 of the ~290 bytes of a 150-base read) can live in page-locked memory
// (brc_host_alloc): the engine then uploads them from where the decoder wrote them (brc_push_reads_pinned) instead of copying them
// into its own staging first — the copy was the longest stage of the engine thread.  A stateful allocator: `pinned` chooses.
#include <new>
template <class T> struct ArenaAlloc {
    typedef T value_type; typedef std::true_type propagate_on_container_copy_assignment; typedef std::true_type propagate_on_container_move_assignment; typedef std::true_type propagate_on_container_swap;
    bool pinned = false;
    ArenaAlloc() {}
    explicit ArenaAlloc(bool p) : pinned(p) {}
    template <class U> ArenaAlloc(const ArenaAlloc<U>& o) : pinned(o.pinned) {}
    // (a 16-byte header in front of every block says where it came from: when no more page-locked memory can be had a block comes from the
    // heap instead of failing the decode thread — hipMemcpyAsync takes pageable memory too, only slower)
    T* allocate(size_t n) {
        char* p = pinned ? (char*)brc_host_alloc(n * sizeof(T) + 16) : nullptr; uint64_t tag = 1;
        if (!p) { p = (char*)malloc(n * sizeof(T) + 16); tag = 0; }
        if (!p) throw std::bad_alloc();
        memcpy(p, &tag, sizeof tag);
        return (T*)(p + 16);
    }
    void deallocate(T* q, size_t) { char* p = (char*)q - 16; uint64_t tag; memcpy(&tag, p, sizeof tag); if (tag) brc_host_free(p); else free(p); }
    template <class U> bool operator==(const ArenaAlloc<U>& o) const { return pinned == o.pinned; }
    template <class U> bool operator!=(const ArenaAlloc<U>& o) const { return pinned != o.pinned; }
};
typedef std::vector<uint8_t, ArenaAlloc<uint8_t> > ArenaVec;

// SoA batch in the brc_read_batch layout
struct Batcher {
    std::vector<int32_t> pos, l_qseq, nm, sm; std::vector<uint16_t> flag; std::vector<uint8_t> mapq, tags;
    std::vector<int16_t> lib; std::vector<uint32_t> n_cigar, cigar; std::vector<uint64_t> cig_off, seq_off, qual_off;
    ArenaVec seq4, qual;
    bool pinned = false; size_t seen_seq = 0, seen_qual = 0;       // arenas in page-locked memory; the largest fills seen so far (what a pinned arena is sized by)
    // from now on this batch's arenas are page-locked, with room for what it held before and a quarter more (an allocation of
    // page-locked memory costs milliseconds: once per buffer, not once per growth step)
    void use_pinned() {
        if (pinned) return;
        try {
            ArenaVec s2{ArenaAlloc<uint8_t>(true)}, q2{ArenaAlloc<uint8_t>(true)};
            s2.reserve(seen_seq + seen_seq / 4 + 4096); q2.reserve(seen_qual + seen_qual / 4 + 4096);
            seq4.swap(s2); qual.swap(q2); pinned = true;
        } catch (...) { pinned = false; }                            // (no page-locked memory to be had: the copying path stays)
    }
    std::vector<char> names; std::vector<size_t> name_off; mutable std::vector<const char*> name_ptr;   // read names (warning text)
    bool keep_names = true;           // -w 0: no warning text will ever be printed
    void clear() { seen_seq = std::max(seen_seq, seq4.size()); seen_qual = std::max(seen_qual, qual.size()); names.clear(); name_off.clear(); pos.clear(); l_qseq.clear(); nm.clear(); sm.clear(); flag.clear(); mapq.clear(); tags.clear(); lib.clear(); n_cigar.clear(); cigar.clear(); cig_off.clear(); seq_off.clear(); qual_off.clear(); seq4.clear(); qual.clear(); }
    void add(const BamRecord& r, int lib_index) {
        pos.push_back(r.pos); flag.push_back(r.flag); mapq.push_back(r.mapq); l_qseq.push_back(r.l_seq); n_cigar.push_back(r.n_cigar);
        lib.push_back((int16_t)lib_index);
        cig_off.push_back(cigar.size()); seq_off.push_back(seq4.size()); qual_off.push_back(qual.size());
        const uint32_t* c = r.cigar(); cigar.insert(cigar.end(), c, c + r.n_cigar);
        seq4.insert(seq4.end(), r.seq(), r.seq() + (r.l_seq + 1) / 2);
        qual.insert(qual.end(), r.qual(), r.qual() + r.l_seq);
        uint8_t t = 0; int32_t vnm = 0, vsm = 0;
        if (r.aux_int("NM", &vnm)) t |= BRC_TAG_NM;      // bam_aux_get + bam_aux2i (BasicStat.cpp:94-96)
        if (r.aux_int("SM", &vsm)) t |= BRC_TAG_SM;      // (BasicStat.cpp:79-81)
        nm.push_back(vnm); sm.push_back(vsm); tags.push_back(t);
        if (keep_names) { name_off.push_back(names.size()); const char* q = r.qname(); const size_t ql = strnlen(q, r.l_qname); names.insert(names.end(), q, q + ql); names.push_back(0); }   // (a record whose name lacks its NUL stays inside the record)
    }
    brc_read_batch view() const {
        brc_read_batch v; memset(&v, 0, sizeof v);
        v.n_reads = (int64_t)pos.size(); v.pos = pos.data(); v.flag = flag.data(); v.mapq = mapq.data(); v.lib = lib.data();
        v.l_qseq = l_qseq.data(); v.n_cigar = n_cigar.data(); v.cigar_off = cig_off.data(); v.seq_off = seq_off.data(); v.qual_off = qual_off.data();
        v.nm = nm.data(); v.sm = sm.data(); v.tags = tags.data(); v.cigar = cigar.data(); v.seq4 = seq4.data(); v.qual = qual.data();
        v.n_cigar_total = cigar.size(); v.seq_bytes = seq4.size(); v.qual_bytes = qual.size();
        name_ptr.resize(name_off.size());
        for (size_t i = 0; i < name_off.size(); ++i) name_ptr[i] = names.data() + name_off[i];
        v.qname = name_ptr.empty() ? nullptr : name_ptr.data();
        return v;
    }
};

#include <atomic>
#include <chrono>
#include <memory>
#include <thread>
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// CPUs this process may really use at once: hardware threads capped by a cgroup CPU quota (a container with 16 CPUs of
// quota on a 256-thread host is throttled for most of every period if its pools are sized by the hardware)
static unsigned g_engines = 1;            // engines of this process (they share the CPUs)
static unsigned g_ranks = 1;              // --brc-ranks: processes side by side on this node; each sizes its pools by its share of the CPUs (set before the first effective_cpus())
static int g_rank = -1;                   // this process's rank (-1: not one of a group)
static std::atomic<unsigned> g_format_threads{0};   // formatter threads per engine when several engines share the process (0: the engine's default)
static unsigned effective_cpus() {
    // (worker threads call this: a function-local static is initialised once, thread-safely)
    static const unsigned cached = []() -> unsigned {
        unsigned n = std::thread::hardware_concurrency(); if (n == 0) n = 1;
        double quota = 0;
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64]; long long per = 0;
            if (fscanf(f, "%63s %lld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) quota = atof(q) / (double)per;
            fclose(f);
        } else {
            long long q = -1, per = 0;
            if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &q) != 1) q = -1; fclose(g); }
            if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &per) != 1) per = 0; fclose(g); }
            if (q > 0 && per > 0) quota = (double)q / (double)per;
        }
        if (quota > 0 && quota < (double)n) n = (unsigned)(quota + 0.999);
        if (g_ranks > 1) n = (n + g_ranks - 1) / g_ranks;              // one rank's share
        return n < 1 ? 1u : n;
    }();
    return cached;
}

// ReadWarnings::warn (src/lib/bamrc/ReadWarnings.hpp:39-50) over a tagged event stream of the engine (brc_region_warnings)
static void print_warn_events(const char* ev, size_t n, long long max, int64_t* counts, FILE* fp) {
    static const char* const kMsg[BRC_N_WARN] = {
        "Couldn't find single-end mapping quality. Check to see if the SM tag is in BAM.",
        "Couldn't find number of mismatches. Check to see if the NM tag is in BAM.",
        "Couldn't find the generated tag.",
        "Library unavailable. Check to make sure the LB tag is present in the @RG entries of the header."};
    for (size_t i = 0; i < n;) {
        size_t j = i; while (j < n && ev[j] != '\n') ++j;
        const char* tp = strchr("SNZL", ev[i]);
        if (ev[i] == 'B') { if (j > i + 2) fwrite(ev + i + 2, 1, j - (i + 2), fp); fputc('\n', fp); }
        else if (tp && *tp && j >= i + 2) {
            const int t = (int)(tp - "SNZL");
            ++counts[t];
            if (!(max >= 0 && counts[t] > max)) {
                fputs("WARNING: In read ", fp); fwrite(ev + i + 2, 1, j - (i + 2), fp); fputs(": ", fp); fputs(kMsg[t], fp); fputc('\n', fp);
                if (max >= 0 && counts[t] == max) fprintf(fp, "The previous warning has been emitted %lld times and will be disabled.\n", (long long)counts[t]);
            }
        }
        i = j + 1;
    }
}

// Many pieces of text -> stdout with few system calls: a region's text arrives as thousands of pieces (the device's text
// between the lines the host rewrote); one write() each would cost more than producing them.
#include <sys/uio.h>
static void write_parts(const char* const* parts, const size_t* lens, size_t n) {
    fflush(stdout);
    const int fd = fileno(stdout);
    std::vector<struct iovec> iov; iov.reserve(1024);
    size_t i = 0;
    while (i < n) {
        iov.clear();
        for (; i < n && iov.size() < 1024; ++i) if (lens[i]) { struct iovec v; v.iov_base = (void*)parts[i]; v.iov_len = lens[i]; iov.push_back(v); }
        size_t k = 0;
        while (k < iov.size()) {
            const ssize_t w = writev(fd, iov.data() + k, (int)(iov.size() - k));
            if (w < 0) { if (errno == EINTR || errno == EAGAIN) continue; return; }     // (a closed pipe: like the reference, carry on silently)
            size_t left = (size_t)w;
            while (k < iov.size() && left >= iov[k].iov_len) { left -= iov[k].iov_len; ++k; }
            if (k < iov.size() && left) { iov[k].iov_base = (char*)iov[k].iov_base + left; iov[k].iov_len -= left; }
        }
    }
}

// Reads of one chunk [a-1, b) (the reference's own fetch rule, :602), decoded by a pool of BAM handles: see fetch_chunk
struct Fetched { std::vector<Batcher> parts; bool ok = true; std::string err; unsigned uses = 0; bool allow_pinned = false;
                 std::vector<std::unique_ptr<BamReader> > pool;      // the BAM handles of this buffer's stripes (fetches of different buffers run side by side)
                 std::vector<std::unique_ptr<CramReader> > cram_pool; };   // ... or its CRAM readers

struct Ctx {
    double t_fetch = 0, t_engine = 0, t_format = 0, t_write = 0;   // BRC_CLI_TIMING=1 prints them on stderr
    double t_site_fetch = 0, t_site_fetch_threads = 0, t_site_layout = 0, t_site_engine = 0, t_site_format = 0; uint64_t n_site_lines = 0, n_site_clusters = 0, n_site_reads = 0, n_site_batches = 0;   // the site-list planner's own account
    Options opt; BamReader bam; BamIndex idx; Fasta fa; bool have_fa = false;
    CramReader cram; bool is_cram = false;          // CRAM 3.0 input (cram.cpp); region queries go through the .crai or one walk over the container headers
    const BamHeader& header() const { return is_cram ? cram.header() : bam.header(); }
    brc_engine* eng = nullptr;
    std::vector<std::string> libs;
    int ref_tid = -1; std::string ref;      // currently loaded contig (load_reference, :83-90)
    Batcher batch;
    uint64_t warn[BRC_N_WARN] = {0, 0, 0, 0};
    // where a work item's text goes: straight to stdout / stderr (one engine), or into the item's buffers (several engines:
    // the main thread writes them in file order)
    std::string* out_buf = nullptr; std::string* err_buf = nullptr;
    std::string* wev_buf = nullptr;       // tagged warning events of the item (several engines) — else they are printed at once
    int64_t wcount[BRC_N_WARN] = {0, 0, 0, 0};   // ReadWarnings' counters (the main context's are the global ones)
    // a rank behind the first: its warning events travel to the coordinating process as they are, one per stderr line behind a
    // 0x01 byte — ReadWarnings' counters are global (-w caps every type across the whole run), so the process that sees the
    // ranks' streams in file order applies them
    bool tag_warnings = false;
    void warn_events(const char* ev, size_t n) {
        if (!n) return;
        if (wev_buf) { wev_buf->append(ev, n); return; }
        if (tag_warnings) {
            for (size_t i = 0; i < n;) { size_t j = i; while (j < n && ev[j] != '\n') ++j; fputc(1, stderr); fwrite(ev + i, 1, j - i, stderr); fputc('\n', stderr); i = j + 1; }
            return;
        }
        print_warn_events(ev, n, opt.max_warnings, wcount, stderr);
    }
    void emit(const char* t, size_t n) { if (!n) return; if (out_buf) out_buf->append(t, n); else fwrite(t, 1, n, stdout); }
    // several engines, region pieces: the text stays in the engine's buffer (no copy of hundreds of megabytes per piece);
    // the main thread writes it from there, and this engine's next format call waits for that (pre_format)
    std::function<bool()> need_engine;      // the first context's engine is created in the background: wait for it (false: it failed)
    std::function<void()> pre_format;
    std::function<void()> after_take;       // run once the piece has taken its reads (the worker starts the next fetch then)
    const char* const** zc_parts = nullptr; const size_t** zc_lens = nullptr; size_t* zc_n = nullptr;
    void emit_region(const char* const* parts, const size_t* lens, size_t n) {
        if (zc_parts) { *zc_parts = parts; *zc_lens = lens; *zc_n = n; return; }
        if (out_buf) { for (size_t i = 0; i < n; ++i) emit(parts[i], lens[i]); return; }
        write_parts(parts, lens, n);
    }
    // several engines: the reads of this engine's next piece, fetched while the current piece is on the GPU / being formatted
    struct Prefetch { bool valid = false; int tid = 0; int64_t a = 0, b = 0; } pf;
    std::unique_ptr<Fetched> pf_buf;
    // run_region's fetch buffers.  They outlive the call: the engine may still hold pointers into the page-locked arenas of the last piece
    // it adopted (brc_push_reads_pinned: valid until the next brc_begin_region or brc_destroy, include/brc.h), and the next region finds
    // its handles open and its arenas pinned
    std::vector<Fetched> bufs;
    void complain(const std::string& m) { if (err_buf) err_buf->append(m); else fputs(m.c_str(), stderr); }
};

static int lib_index(const Ctx& c, const BamRecord& r) {      // bam_get_library: RG tag -> @RG ID -> LB
    const char* rg = r.aux_str("RG");
    if (!rg) return -1;
    auto it = c.header().rg2lb.find(rg);
    if (it == c.header().rg2lb.end()) return -1;
    for (size_t i = 0; i < c.libs.size(); ++i) if (c.libs[i] == it->second) return (int)i;
    return -1;
}

// Reads of one chunk [a-1, b) (the reference's own fetch rule, :602), decoded by a pool of BAM handles: the chunk is cut
// by read START position into K stripes; stripe i keeps the records whose pos lies in its stripe (stripe 0 also the reads
// that start before the chunk), so the stripes concatenated are exactly the single-handle fetch, in file order.

static long long stripe_min_bp() { static const long long v = getenv("BRC_FETCH_STRIPE_MIN") ? atoll(getenv("BRC_FETCH_STRIPE_MIN")) : 65536; return v; }   // (tests force small chunks into stripes)
static void fetch_chunk(Ctx& c, int tid, int64_t a, int64_t b, Fetched& out) {
    out.ok = true; out.err.clear();
    const int64_t q0 = a - 1 < 0 ? 0 : a - 1;
    unsigned K = 1;
    const long long stripe_min = stripe_min_bp();
    if (b - q0 >= stripe_min) {
        K = effective_cpus() * 3 / 4 / g_engines; if (K < 2) K = 2; if (K > 32) K = 32;
        if (const char* t = getenv("BRC_FETCH_THREADS")) { const int v = atoi(t); if (v > 0) K = (unsigned)v; }
        if ((int64_t)K > b - q0) K = (unsigned)(b - q0);          // every stripe at least one position wide (stripe 0 must contain q0)
    }
    if (out.parts.size() < K) out.parts.resize(K);
    for (Batcher& p : out.parts) { p.clear(); p.keep_names = c.opt.max_warnings != 0; }
    // a buffer that has been through one piece knows its sizes: from its second piece on its arenas are page-locked and the engine
    // uploads them in place (run_region: brc_push_reads_pinned)
    static const bool zero_copy = !(getenv("BRC_ZERO_COPY") && atoi(getenv("BRC_ZERO_COPY")) == 0);
    if (zero_copy && out.allow_pinned && out.uses++ > 0) for (Batcher& p : out.parts) if (p.seen_qual) p.use_pinned();
    if (c.is_cram) {
        // (also a piece too short for stripes, K == 1: a reader of this BUFFER, never the context's one — two pieces are fetched side by side)
        // stripes of a CRAM piece: a reader of its own per stripe (a stripe decodes the slices that overlap it — a slice that straddles a
        // stripe boundary twice — and keeps the records that START in it); the contig's bases, which run_region holds, are shared
        while (out.cram_pool.size() < K) {
            out.cram_pool.emplace_back(new CramReader());
            if (!out.cram_pool.back()->open(c.opt.bam, c.have_fa ? &c.fa : nullptr)) { out.ok = false; out.err = out.cram_pool.back()->error(); return; }
        }
        const bool share = c.have_fa && c.ref_tid == tid && !c.ref.empty();
        std::atomic<int> failed(0); std::mutex em;
        auto work = [&](unsigned i) {
            const int64_t s0 = q0 + (b - q0) * (int64_t)i / (int64_t)K, s1 = q0 + (b - q0) * (int64_t)(i + 1) / (int64_t)K;
            CramReader& rd = *out.cram_pool[i];
            rd.share_reference(tid, share ? &c.ref : nullptr);
            if (!rd.fetch(tid, i == 0 ? a - 1 : s0, s1, [&](const BamRecord& r) {
                    if (r.pos >= s1 || (i > 0 && r.pos < s0)) return;
                    out.parts[i].add(r, c.opt.per_lib ? lib_index(c, r) : 0);
                })) { failed = 1; std::lock_guard<std::mutex> g(em); if (out.err.empty()) out.err = rd.error(); }
        };
        std::vector<std::thread> th;
        for (unsigned i = 1; i < K; ++i) th.emplace_back(work, i);
        work(0);
        for (std::thread& t : th) t.join();
        if (failed) out.ok = false;
        return;
    }
    while (out.pool.size() < K) { out.pool.emplace_back(new BamReader()); if (!out.pool.back()->open(c.opt.bam)) { out.ok = false; out.err = "cannot reopen " + c.opt.bam; return; } }
    std::atomic<int> failed(0);
    auto work = [&](unsigned i) {
        const int64_t s0 = q0 + (b - q0) * (int64_t)i / (int64_t)K, s1 = q0 + (b - q0) * (int64_t)(i + 1) / (int64_t)K;
        if (!out.pool[i]->fetch(c.idx, tid, i == 0 ? a - 1 : s0, s1, [&](const BamRecord& r) {
                if (r.pos >= s1 || (i > 0 && r.pos < s0)) return;
                out.parts[i].add(r, c.opt.per_lib ? lib_index(c, r) : 0);
            })) { failed = 1; if (out.err.empty()) out.err = out.pool[i]->error(); }
    };
    std::vector<std::thread> th;
    for (unsigned i = 1; i < K; ++i) th.emplace_back(work, i);
    work(0);
    for (std::thread& t : th) t.join();
    if (failed) { out.ok = false; out.err = "read error while fetching a region stripe" + (out.err.empty() ? std::string() : ": " + out.err); }
}

// one reporting window [beg0,end) on tid: the body of the site-list / region loops (:588-605, :649-656)
// keep_queue: the first piece keeps the deletions the previous command-line region left pending (like the reference, :641-657)
// inner_piece: (several engines) this window is a piece of a longer region whose previous piece ran elsewhere
// The piece a long region is cut into, when the command line does not say (--brc-chunk): by the DATA under it, not by positions — about
// 26 MB of compressed records (200 000 reads of 150 bases), weighed by the index's file offsets (BamIndex::span_bytes): 1 Mbp of a 30x
// genome, 125 kb of a 200x tumour.  A piece sizes everything that is page-locked (decode arenas, staging, two text buffers — 1.4 GB per
// Mbp with four libraries) and the latency of the three-stage pipeline's first piece: BASELINE config 5 end to end went from 1.7 s at the
// fixed 1 Mbp to 0.93 s at 125 kb on the same box (profiles/r06_e2e_tumor_chunk_sweep.log), a third of it process exit (the kernel unpins
// what was pinned).  Multiples of 64 kb; never above the 1 Mbp that suits shallow data.
static int64_t auto_chunk(const Ctx& c, int tid, int64_t beg0, int64_t end) {
    const int64_t max_chunk = (int64_t)c.opt.chunk_bp;
    if (c.opt.chunk_given || c.is_cram || end - beg0 < 2 * 65536 || c.opt.max_cnt < 1000000) return max_chunk;
    static const double target = getenv("BRC_CHUNK_BYTES") && atof(getenv("BRC_CHUNK_BYTES")) > 0 ? atof(getenv("BRC_CHUNK_BYTES")) : 26.0e6;
    const double bytes = c.idx.span_bytes(tid, beg0, end);
    if (bytes <= 0) return max_chunk;
    const double per_bp = bytes / (double)(end - beg0);
    int64_t ch = (int64_t)(target / per_bp);
    ch = ((ch + 65535) / 65536) * 65536;
    if (ch < 65536) ch = 65536;
    return ch < max_chunk ? ch : max_chunk;
}

static int run_region(Ctx& c, int tid, int64_t beg0, int64_t end, bool site_mode, bool keep_queue = true, bool inner_piece = false) {
    const BamHeader& h = c.header();
    if (c.have_fa && tid != c.ref_tid) {
        if (!c.fa.fetch(h.names[(size_t)tid], &c.ref)) { c.ref.clear(); fprintf(stderr, "bam-readcount: %s: no reference bases for %s\n", c.fa.error().c_str(), h.names[(size_t)tid].c_str()); }   // (the reference dereferences a null pointer here)
        c.ref_tid = tid;
    }
    const char* ref = c.have_fa && !c.ref.empty() ? c.ref.data() : nullptr;
    if (end > (int64_t)h.lengths[(size_t)tid] + 1000) end = (int64_t)h.lengths[(size_t)tid] + 1000;   // nothing aligns past the contig
    if (end < beg0) end = beg0;
    // long regions are cut into abutting pieces; each piece fetches one base early exactly like the reference (:602).
    // The next `ahead` pieces are fetched and decoded in the background (each by its own pool of striped BAM handles)
    // while the current one is on the GPU / being formatted: decoding is the slowest of the three stages.
    const int64_t chunk = auto_chunk(c, tid, beg0, end);
    const int64_t npieces = std::max<int64_t>(1, (end - beg0 + chunk - 1) / chunk);
    static const int ahead_env = getenv("BRC_FETCH_AHEAD") ? atoi(getenv("BRC_FETCH_AHEAD")) : 0;
    const int ahead = ahead_env > 0 ? ahead_env : 2;
    std::vector<Fetched>& bufs = c.bufs; if (bufs.size() < (size_t)ahead + 1) bufs.resize((size_t)ahead + 1);
    std::vector<std::thread> fth((size_t)ahead + 1);
    for (Fetched& f : bufs) { if (npieces > (int64_t)ahead + 1) f.allow_pinned = true; }        // (only a run of pieces reuses its buffers)
    auto start_fetch = [&](int64_t j) {
        if (j >= npieces) return;
        const size_t slot = (size_t)(j % (ahead + 1));
        const int64_t fa = beg0 + j * chunk, fb = std::min<int64_t>(fa + chunk, end);
        fth[slot] = std::thread([&c, tid, fa, fb, &bufs, slot]() { fetch_chunk(c, tid, fa, fb, bufs[slot]); });
    };
    struct JoinAll { std::vector<std::thread>& t; ~JoinAll() { for (std::thread& x : t) if (x.joinable()) x.join(); } } join_all{fth};
    int64_t a = beg0;
    double t0 = now_s();
    if (c.pf.valid && c.pf.tid == tid && c.pf.a == a && c.pf.b == std::min<int64_t>(a + chunk, end)) std::swap(bufs[0], *c.pf_buf);   // fetched ahead by the worker
    else start_fetch(0);
    c.pf.valid = false;
    if (c.after_take) { if (fth[0].joinable()) fth[0].join(); c.after_take(); c.after_take = nullptr; }
    for (int64_t j = 1; j < ahead; ++j) start_fetch(j);
    c.t_fetch += now_s() - t0;
    // Two stages, one piece apart: while a helper thread formats and writes piece k (host planes of the last download),
    // this thread stages, uploads and computes piece k + 1 (staging and device buffers only); the download of k + 1 —
    // which overwrites the host planes — waits for the helper.  include/brc.h states this concurrency rule.
    brc_result res[2]; int slot = 0;
    std::thread fmt; int fmt_rc = 0; double fmt_s = 0;
    auto join_fmt = [&]() { if (fmt.joinable()) { fmt.join(); c.t_format += fmt_s; fmt_s = 0; } return fmt_rc; };
    int rc = 0;
    for (int64_t j = 0; j < npieces; ++j) {
        const int64_t b = std::min<int64_t>(a + chunk, end);
        const size_t cur = (size_t)(j % (ahead + 1));
        { const double w0 = now_s(); if (fth[cur].joinable()) fth[cur].join(); c.t_fetch += now_s() - w0; }   // only what the background fetch did not hide
        Fetched& F = bufs[cur];
        if (!F.ok) { join_fmt(); c.complain("bam-readcount: read error: " + F.err + "\n"); return 1; }
        double t1 = now_s();
        if (c.need_engine && !c.need_engine()) { join_fmt(); return 1; }
        rc = brc_begin_region(c.eng, tid, (int32_t)a, (int32_t)b, ref, (int64_t)c.ref.size());
        // (behind brc_begin_region: the buffer piece j - 1 has left may hold arenas the engine adopted — brc_push_reads_pinned —
        // which stay the engine's to read until this call)
        start_fetch(j + ahead);
        // the lines of a region piece are written on the GPU and come back as text (BRC_DEVICE_TEXT=0: host formatter)
        static const bool dev_text = !(getenv("BRC_DEVICE_TEXT") && atoi(getenv("BRC_DEVICE_TEXT")) == 0);
        brc_set_option(c.eng, BRC_OPT_DEVICE_TEXT, dev_text ? 1 : 0);
        if (dev_text) brc_set_chrom(c.eng, h.names[(size_t)tid].c_str());
        {   // the stripes of a piece arrive as separate batches: tell the engine their total so it sizes its staging once
            size_t nr = 0, nq = 0; for (const Batcher& part : F.parts) { nr += part.pos.size(); nq += part.qual.size(); }
            brc_set_option(c.eng, BRC_OPT_EXPECT_READS, (int64_t)(nr + nr / 8)); brc_set_option(c.eng, BRC_OPT_EXPECT_BASES, (int64_t)(nq + nq / 8));
        }
        bool all_pinned = true;
        for (const Batcher& part : F.parts) if (!part.pos.empty() && !part.pinned) all_pinned = false;
        for (const Batcher& part : F.parts) {
            if (rc || part.pos.empty()) continue;
            const brc_read_batch v = part.view();
            rc = all_pinned ? brc_push_reads_pinned(c.eng, &v) : brc_push_reads(c.eng, &v);
        }
        if (!rc) rc = brc_upload(c.eng);
        if (!rc) rc = brc_compute(c.eng, nullptr);
        double t2 = now_s(); c.t_engine += t2 - t1;
        const int prev_rc = join_fmt();                       // piece k is out: its host planes may be overwritten
        double t3 = now_s(); c.t_write += t3 - t2;           // (time this thread waited for the formatter / writer)
        if (!rc && prev_rc) rc = prev_rc;
        brc_result& R = res[slot]; slot ^= 1;
        if (!rc && c.pre_format) c.pre_format();             // several engines: the writer is done with this engine's previous text
        if (!rc) rc = brc_fetch_result(c.eng, &R);
        c.t_engine += now_s() - t3;
        if (!rc) {
            // an internal piece boundary is not a region boundary: the deletions left pending by the previous piece start at
            // its last position, which is this piece's lead position and queues them again — drop the leftovers (the FIRST
            // piece keeps whatever the previous command-line region left, like the reference, :641-657)
            const bool clear_first = a == beg0 && !keep_queue;
            // a piece after the first continues the piece before it on this engine: its lead position is not processed again
            // and the deletion queues carry over.  (Set between the download of this piece and the download of the next:
            // the formatter thread and the warnings of THIS piece are its only readers.)
            brc_set_option(c.eng, BRC_OPT_CONTINUES_PREVIOUS, a > beg0 ? 1 : (inner_piece ? 2 : 0));
            const char* chrom = h.names[(size_t)tid].c_str();
            fmt_rc = 0;
            fmt = std::thread([&c, &R, chrom, clear_first, &fmt_rc, &fmt_s]() {
                const double f0 = now_s();
                if (clear_first) brc_clear_indel_queue(c.eng);
                const char* const* tparts = nullptr; const size_t* tlens = nullptr; size_t tn = 0;
                fmt_rc = brc_format_region_parts(c.eng, &R, chrom, &tparts, &tlens, &tn);
                if (!fmt_rc) c.emit_region(tparts, tlens, tn);
                fmt_s = now_s() - f0;
            });
            // (the warnings read the staged reads of this piece: before the next brc_begin_region)
            if (c.opt.max_warnings != 0) {
                const char* ev = ""; size_t evn = 0;
                if (brc_region_warnings(c.eng, chrom, c.opt.max_warnings, &ev, &evn) == 0) c.warn_events(ev, evn);
            }
            for (int w = 0; w < BRC_N_WARN; ++w) c.warn[w] += R.warn[w];
        }
        if (rc) { join_fmt(); c.complain(std::string("bam-readcount: engine error: ") + brc_strerror(rc) + " (" + brc_last_error(c.eng) + ")\n"); return 1; }
        a = b;
    }
    if ((rc = join_fmt())) { c.complain(std::string("bam-readcount: engine error: ") + brc_strerror(rc) + " (" + brc_last_error(c.eng) + ")\n"); return 1; }
    if (site_mode) brc_clear_indel_queue(c.eng);                                      // :605
    return 0;
}

// ---------------------------------------------------------------- site-list planner
// The reference runs every -l line as an independent fetch + pileup (bamreadcount.cpp:574-607).  Here a batch of lines is
// laid out side by side on one virtual coordinate axis: window i keeps its own reads (fetched exactly like the
// reference would, [beg0-1, end)), translated by delta_i, and its own slice of the reference; the engine runs ONE region
// over the whole axis, and each line's text is cut out of the shared planes with brc_format_window (fresh deletion
// queues, coordinates translated back).  Duplicate and overlapping lines stay exact because every window carries its own
// copy of its reads.  The indexed fetches of a batch run on a pool of threads, each with its own BAM handle.
#include <atomic>
#include <thread>

struct Site { int tid; int64_t beg0, end; };

// the indexed fetches of one batch of lines: every line's own reads (exactly what samfetch would hand over for [beg0 - 1, end))
// and the extent they span.  Reads only what never changes while the run lasts (options, index, header, libraries), so the
// fetch of the NEXT batch runs on threads of its own while this batch is laid out, computed and printed.
struct SiteFetch { std::vector<Batcher> parts; std::vector<int64_t> lo, hi; bool ok = true; size_t n_clusters = 0; double seconds = 0; };

static void fetch_site_batch(const Ctx& c, const std::vector<Site>& sites, SiteFetch& F) {
    const double ts0 = now_s();
    const size_t n = sites.size();
    std::vector<Batcher>& parts = F.parts; parts.clear(); parts.resize(n);
    for (Batcher& b : parts) b.keep_names = c.opt.max_warnings != 0;
    std::vector<int64_t>& lo = F.lo; std::vector<int64_t>& hi = F.hi; lo.assign(n, 0); hi.assign(n, 0);
    std::atomic<size_t> next(0); std::atomic<int> failed(0);
    // Clusters of consecutive lines that one indexed fetch serves at no extra cost: an index points at 16-kb windows (the
    // linear index; a CSI's finest bins), so a fetch for line k decodes its window from the window's first record up to the
    // line.  A following line in the SAME window (ascending, same contig) would decode the same records again — with one site
    // per kb, sixteen times — so it joins the cluster and every record is handed to the lines it overlaps with exactly
    // samfetch's test.  A line in a LATER window starts a cluster of its own: one fetch over both would also decode
    // everything between them (at the 31-kb spacing of a whole-genome SNV list that was three times the bytes).
    std::vector<std::pair<size_t, size_t> > clusters;
    for (size_t i0 = 0; i0 < n;) {
        size_t i1 = i0 + 1;
        int64_t reach = sites[i0].end;                     // the cluster's fetch decodes up to here anyway
        while (i1 < n && i1 - i0 < 256 && sites[i1].tid == sites[i0].tid && sites[i1].beg0 >= sites[i1 - 1].beg0 && sites[i1].beg0 - sites[i0].beg0 <= 65536 &&
               (((sites[i1].beg0 > 0 ? sites[i1].beg0 - 1 : 0) >> 14) <= (reach >> 14) || sites[i1].beg0 - reach <= 2048)) { reach = std::max(reach, sites[i1].end); ++i1; }
        clusters.push_back(std::make_pair(i0, i1));
        i0 = i1;
    }
    auto work = [&]() {
        BamReader rd;
        if (!rd.open(c.opt.bam)) { failed = 1; return; }
        for (;;) {
            const size_t ci = next.fetch_add(1);
            if (ci >= clusters.size()) break;
            const size_t i0 = clusters[ci].first, i1 = clusters[ci].second;
            int64_t cend = 0;
            for (size_t i = i0; i < i1; ++i) { lo[i] = sites[i].beg0 > 0 ? sites[i].beg0 - 1 : 0; hi[i] = sites[i].end; cend = std::max(cend, sites[i].end); }
            size_t jlo = i0;                       // lines before it end at or before every coming record's start (lines are < 1 kb wide)
            if (!rd.fetch(c.idx, sites[i0].tid, sites[i0].beg0 - 1, cend, [&](const BamRecord& rec) {
                    const int64_t rend = rec.endpos();
                    while (jlo < i1 && sites[jlo].beg0 + 1000 <= rec.pos) ++jlo;
                    for (size_t j = jlo; j < i1 && sites[j].beg0 - 1 < rend; ++j) {
                        const Site& st = sites[j];
                        const int64_t qb = st.beg0 - 1 < 0 ? 0 : st.beg0 - 1;                  // samfetch: beg clamped at 0, pos < end && endpos > beg
                        if (st.end <= qb || rec.pos >= st.end || rend <= qb) continue;
                        parts[j].add(rec, c.opt.per_lib ? lib_index(c, rec) : 0);
                        if (rec.pos < lo[j]) lo[j] = rec.pos;
                        if (rend > hi[j]) hi[j] = rend;
                    }
                })) failed = 1;
            for (size_t i = i0; i < i1; ++i) hi[i] += 64;      // room for deletion alleles read from the reference past the last read
        }
    };
    unsigned nthr = effective_cpus(); if (nthr > 64) nthr = 64;
    if (const char* t = getenv("BRC_FETCH_THREADS")) { const int v = atoi(t); if (v > 0) nthr = (unsigned)v; }
    std::vector<std::thread> th;
    for (unsigned k = 1; k < nthr && (size_t)k < clusters.size(); ++k) th.emplace_back(work);
    work();
    for (std::thread& t : th) t.join();
    F.ok = !failed; F.n_clusters = clusters.size(); F.seconds = now_s() - ts0;
}

// pre: the batch's reads when they were fetched ahead (one engine: the main loop overlaps the next batch's fetch with this batch)
static int run_site_batch(Ctx& c, const std::vector<Site>& sites, SiteFetch* pre = nullptr) {
    const size_t n = sites.size();
    const BamHeader& h = c.header();
    SiteFetch own;
    if (!pre) { const double w0 = now_s(); fetch_site_batch(c, sites, own); pre = &own; c.t_site_fetch += now_s() - w0; }
    if (!pre->ok) { c.complain("bam-readcount: read error while fetching sites\n"); return 1; }
    std::vector<Batcher>& parts = pre->parts; const std::vector<int64_t>& lo = pre->lo; const std::vector<int64_t>& hi = pre->hi;
    c.t_site_fetch_threads += pre->seconds; c.n_site_lines += n; c.n_site_clusters += pre->n_clusters;
    // Virtual layout, in sub-batches: the engine keeps planes for every virtual position, and a window's extent is known
    // only now (it includes the overhang of its reads — with long reads far more than the line asked for), so the batch is
    // cut wherever the axis would outgrow the chunk size.  A window wider than that runs alone.
    static const long long vmax_env = getenv("BRC_PLAN_VMAX") ? atoll(getenv("BRC_PLAN_VMAX")) : 0;   // (tests force tiny sub-batches)
    const int64_t vmax = vmax_env > 0 ? vmax_env : std::max<int64_t>(4 * c.opt.chunk_bp, 1 << 20);
    std::vector<int64_t> delta(n);
    for (size_t i0 = 0; i0 < n;) {
        const double tl0 = now_s();
        int64_t V = 1; size_t i1 = i0;
        while (i1 < n && (i1 == i0 || V + (hi[i1] - lo[i1]) + 1 <= vmax)) { delta[i1] = V - lo[i1]; V += (hi[i1] - lo[i1]) + 1; ++i1; }
        if (V >= (int64_t)INT_MAX - 64) { c.complain("bam-readcount: a site-list window is too wide for the planner; rerun with --brc-plan 0\n"); return 1; }
        std::string vref;
        if (c.have_fa) vref.assign((size_t)V + 1, '\0');
        size_t nr = 0, nq = 0;
        for (size_t i = i0; i < i1; ++i) {
            const Site& st = sites[i];
            if (c.have_fa) {
                // The window's slice of the reference, read straight out of the mapped FASTA (Fasta::read_range): a whole-genome list
                // visits every contig, and loading each one whole (fai_fetch, as the reference does per -l line's contig) was a sixth of
                // the run.  Past the contig: the annotator stops at the terminating NUL (x == len, :151) but merely skips positions
                // x > len (site-list mode, :144-148); 'N' reproduces the skip (it never counts as a mismatch, :152).
                const int64_t x0 = std::max<int64_t>(lo[i], 0), x1 = hi[i];
                int64_t clen = 0;
                int64_t got = x1 > x0 ? c.fa.read_range(h.names[(size_t)st.tid], x0, x1 - x0, &vref[(size_t)(x0 + delta[i])], &clen) : 0;
                if (got < 0) {         // (no mapping / unknown contig: the whole contig, as before)
                    if (st.tid != c.ref_tid) {
                        if (!c.fa.fetch(h.names[(size_t)st.tid], &c.ref)) { c.ref.clear(); fprintf(stderr, "bam-readcount: %s: no reference bases for %s\n", c.fa.error().c_str(), h.names[(size_t)st.tid].c_str()); }
                        c.ref_tid = st.tid;
                    }
                    clen = (int64_t)c.ref.size();
                    const int64_t xin = std::min(x1, clen);
                    if (xin > x0) memcpy(&vref[(size_t)(x0 + delta[i])], c.ref.data() + x0, (size_t)(xin - x0));
                }
                for (int64_t x = std::max(x0, clen); x < x1; ++x) vref[(size_t)(x + delta[i])] = x == clen ? '\0' : 'N';
            }
            // the line's reads move to its window of the virtual axis where they are (no second copy of the batch)
            Batcher& b = parts[i];
            for (size_t k = 0; k < b.pos.size(); ++k) b.pos[k] = (int32_t)(b.pos[k] + delta[i]);
            nr += b.pos.size(); nq += b.qual.size();
        }
        const double te0 = now_s(); c.t_site_layout += te0 - tl0; c.n_site_reads += nr; ++c.n_site_batches;
        if (c.need_engine && !c.need_engine()) return 1;
        brc_set_option(c.eng, BRC_OPT_DEVICE_TEXT, 0);           // windows are cut out of shared planes on the host
        int rc = brc_begin_region(c.eng, 0, 1, (int32_t)V, c.have_fa ? vref.data() : nullptr, V);
        brc_set_option(c.eng, BRC_OPT_EXPECT_READS, (int64_t)(nr + 16)); brc_set_option(c.eng, BRC_OPT_EXPECT_BASES, (int64_t)(nq + 16));
        for (size_t i = i0; i < i1 && !rc; ++i) {                // window by window, in axis order (= coordinate order)
            if (parts[i].pos.empty()) continue;
            const brc_read_batch v = parts[i].view();
            rc = brc_push_reads(c.eng, &v);
        }
        if (!rc) {         // only the lines' own positions are ever formatted: the engine need not pile up the rest of their reads' extent
            std::vector<int32_t> wb, we;
            for (size_t i = i0; i < i1; ++i) { wb.push_back((int32_t)(sites[i].beg0 + delta[i])); we.push_back((int32_t)(sites[i].end + delta[i])); }
            rc = brc_region_windows(c.eng, wb.data(), we.data(), (int64_t)wb.size());
        }
        brc_result res;
        if (!rc) rc = brc_end_region(c.eng, &res);
        if (rc) { c.complain(std::string("bam-readcount: engine error: ") + brc_strerror(rc) + " (" + brc_last_error(c.eng) + ")\n"); return 1; }
        const double tf0 = now_s(); c.t_site_engine += tf0 - te0;
        for (size_t i = i0; i < i1; ++i) {
            const Site& st = sites[i];
            const char* text = ""; size_t len = 0;
            rc = brc_format_window(c.eng, &res, h.names[(size_t)st.tid].c_str(), (int32_t)(st.beg0 + delta[i]), (int32_t)(st.end + delta[i]), (int32_t)delta[i], &text, &len);
            if (rc) { c.complain(std::string("bam-readcount: engine error: ") + brc_strerror(rc) + "\n"); return 1; }
            c.emit(text, len);
            if (c.opt.max_warnings != 0) { const char* ev = ""; size_t evn = 0; if (brc_window_warnings(c.eng, (int32_t)(st.beg0 + delta[i]), (int32_t)(st.end + delta[i]), c.opt.max_warnings, &ev, &evn) == 0) c.warn_events(ev, evn); }
        }
        for (int w = 0; w < BRC_N_WARN; ++w) c.warn[w] += res.warn[w];
        c.t_site_format += now_s() - tf0;
        i0 = i1;
    }
    return 0;
}

// A command-line region as samtools-1.10's legacy bam_parse_region reads it (bamreadcount.cpp:644; bam.c: hts_parse_reg, and
// when that fails the whole string as a contig name): the name ends at the LAST colon; numbers go through
// hts_parse_decimal (leading white space, sign, thousands commas, fraction, exponent, k/M/G suffix; stderr notes for a
// discarded fraction and for trailing characters); "chr:beg" runs to the end, beg <= 0 means 1, an interval that is empty
// or past INT_MAX is no region.  Returns tid, 0-based beg, end.
static long long parse_decimal_like_htslib(const char* str, const char** strend, std::string* log) {
    auto push = [](long long v, char ch) { const int d = ch - '0'; return v > (LLONG_MAX - d) / 10 ? LLONG_MAX : 10 * v + d; };
    long long n = 0; int decimals = 0, e = 0, lost = 0; char sign = '+', esign = '+';
    while (*str == ' ' || (*str >= '\t' && *str <= '\r')) ++str;
    const char* q = str;
    if (*q == '+' || *q == '-') sign = *q++;
    while (*q) { if (*q >= '0' && *q <= '9') n = push(n, *q++); else if (*q == ',') ++q; else break; }
    if (*q == '.') { ++q; while (*q >= '0' && *q <= '9') { ++decimals; n = push(n, *q++); } }
    switch (*q) {
        case 'e': case 'E':
            ++q; if (*q == '+' || *q == '-') esign = *q++;
            while (*q >= '0' && *q <= '9') e = (int)push(e, *q++);
            if (esign == '-') e = -e;
            break;
        case 'k': case 'K': e += 3; ++q; break;
        case 'm': case 'M': e += 6; ++q; break;
        case 'g': case 'G': e += 9; ++q; break;
    }
    e -= decimals;
    while (e > 0) { n *= 10; --e; }
    while (e < 0) { lost += (int)(n % 10); n /= 10; ++e; }
    if (lost > 0) *log += "[W::hts_parse_decimal] Discarding fractional part of " + std::string(str, q) + "\n";
    if (strend) *strend = q;
    else if (*q) *log += "[W::hts_parse_decimal] Ignoring unknown characters after " + std::string(str, q) + "[" + q + "]\n";
    return sign == '+' ? n : -n;
}

static bool parse_region(const BamHeader& h, const std::string& s, int* tid, int64_t* beg, int64_t* end, std::string* log) {
    // (log: what htslib writes to stderr while it parses — the caller prints it when the reference would reach this region)
    const size_t colon = s.rfind(':');
    bool ok = true;                                                 // parsable as a region
    long long b = 0, e = 0;
    if (colon == std::string::npos) { b = 0; e = LLONG_MAX; }
    else {
        const char* hyphen = nullptr;
        b = parse_decimal_like_htslib(s.c_str() + colon + 1, &hyphen, log) - 1;
        if (b < 0) b = 0;
        if (*hyphen == '\0') e = LLONG_MAX;
        else if (*hyphen == '-') e = parse_decimal_like_htslib(hyphen + 1, nullptr, log);
        else ok = false;
        if (ok && b >= e) ok = false;
    }
    // hts_parse_reg narrows hts_parse_reg64's positions to int whatever that returned
    if (b > INT_MAX) { *log += "[E::hts_parse_reg] Position " + std::to_string(b) + " too large\n"; ok = false; }
    else if (e > INT_MAX) { if (e == LLONG_MAX) e = INT_MAX; else { *log += "[E::hts_parse_reg] Position " + std::to_string(e) + " too large\n"; ok = false; } }
    size_t name_lim = colon == std::string::npos ? s.size() : colon;
    if (!ok) { name_lim = s.size(); b = 0; e = INT_MAX; }            // ... but possibly a contig named "foo:a"
    auto it = h.name2tid.find(s.substr(0, name_lim));
    if (it == h.name2tid.end()) return false;
    *tid = it->second; *beg = b; *end = e;
    return true;
}

// One line of the site list as `std::stringstream ss(line); ss >> ref_name >> beg >> end` reads it (bamreadcount.cpp:574-577):
// fields separated by any white space, the numbers as the stream's num_get takes them — optional sign, decimal digits, stop
// at the first other character; no digit, or a value outside int, fails the extraction and the line is skipped.
static bool parse_site_line(const char* p, size_t n, std::string* name, int* beg, int* end) {
    const char* e = p + n;
    if (n && e[-1] == '\n') --e;                                   // (getline drops the delimiter; a '\r' stays and is white space)
    auto space = [](char ch) { return ch == ' ' || (ch >= '\t' && ch <= '\r'); };
    auto skip = [&]() { while (p < e && space(*p)) ++p; };
    skip();
    const char* b = p;
    while (p < e && !space(*p)) ++p;
    if (p == b) return false;
    name->assign(b, p);
    auto integer = [&](int* out) {
        skip();
        bool neg = false;
        if (p < e && (*p == '+' || *p == '-')) { neg = *p == '-'; ++p; }
        const char* d = p; long long v = 0; bool over = false;
        while (p < e && *p >= '0' && *p <= '9') { if (v < (1ll << 40)) v = v * 10 + (*p - '0'); else over = true; ++p; }
        if (p == d) return false;
        if (neg) v = -v;
        if (over || v > 2147483647ll || v < -2147483648ll) return false;
        *out = (int)v; return true;
    };
    return integer(beg) && integer(end);
}

// ---------------------------------------------------------------- work items and engines
// The command line is turned into work items in file order: a site-list batch (narrow -l lines, planner), or a piece of a
// region / wide -l line.  One engine: the items run one after the other on it.  Several engines (--brc-gpus N, one per
// GPU, each with a worker thread and its own file handles): item i goes to engine i mod N — except the first piece of a
// command-line region, which follows the last piece of the previous region onto its engine, because it must see the
// deletions that piece left pending (the reference does not clear its queue between command-line regions, :641-657) —
// and the main thread writes the items' text in file order.
#include <condition_variable>
#include <mutex>

struct Work {
    int kind = 0;                     // 0: region piece, 1: site-list batch, 2: an error that ends the run, 3: a message (stderr) in file order
    int tid = 0; int64_t beg0 = 0, end = 0; bool site_mode = false, keep_queue = false, inner_piece = false;
    std::vector<Site> sites;
    int engine = 0;
    std::string out, err, wev; int rc = 0; bool done = false;
    const char* const* parts = nullptr; const size_t* lens = nullptr; size_t n_parts = 0;      // a region piece's text, still in its engine's buffers
};

static int run_item(Ctx& c, Work& w) {
    if (w.kind == 1) return run_site_batch(c, w.sites);
    return run_region(c, w.tid, w.beg0, w.end, w.site_mode, w.keep_queue, w.inner_piece);
}

static bool open_inputs(Ctx& c, bool quiet) {
    const Options& o = c.opt;
    if (!o.fasta.empty()) {
        if (!c.fa.open(o.fasta)) { if (!quiet) fprintf(stderr, "Fail to open reference file %s\n", o.fasta.c_str()); return false; }
        c.have_fa = true;
    }
    c.is_cram = CramReader::is_cram(o.bam);
    if (c.is_cram ? !c.cram.open(o.bam, c.have_fa ? &c.fa : nullptr) : !c.bam.open(o.bam)) {                                               // :513-516
        if (!quiet) { fprintf(stderr, "Fail to open BAM file %s\n", o.bam.c_str()); if (c.is_cram) fprintf(stderr, "bam-readcount: %s\n", c.cram.error().c_str()); }
        return false;
    }
    c.libs = c.header().libraries();
    return true;
}

static int make_engine(Ctx& c, int device) {
    const Options& o = c.opt;
    std::vector<const char*> names; for (const std::string& l : c.libs) names.push_back(l.c_str());
    brc_config cfg; memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = BRC_ABI_VERSION; cfg.min_mapq = o.min_mapq; cfg.min_bq = o.min_bq; cfg.max_cnt = o.max_cnt;
    cfg.per_lib = o.per_lib; cfg.insertion_centric = o.insertion_centric; cfg.n_libs = (int32_t)names.size();
    cfg.lib_names = names.empty() ? nullptr : names.data(); cfg.device = device;
    cfg.ref_len_check = (!o.site_list.empty() && c.have_fa) ? 1 : 0;                                                                        // :594-600
    const int rc = brc_create(&cfg, &c.eng);
    if (rc == 0) brc_set_option(c.eng, BRC_OPT_TEXT_ONLY, 1);      // the command line only prints: no dense planes on the host
    if (rc == 0 && g_format_threads.load()) brc_set_option(c.eng, BRC_OPT_FORMAT_THREADS, (int64_t)g_format_threads.load());
    if (rc == 0 && o.max_cnt <= 0) brc_set_option(c.eng, BRC_OPT_MAX_COUNT, o.max_cnt);   // -d 0 / -d -3 reach the iterator as they are (:592,:651)
    return rc;
}

// A crash must not be silent: frames of the faulting thread (module + offset: `addr2line -e <module> <offset>` resolves
// them) go to stderr before the default action takes the process down.
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
static void crash_handler(int sig) {
    void* frames[64];
    const int n = backtrace(frames, 64);
    static const char msg[] = "bam-readcount: fatal signal, backtrace of the faulting thread:\n";
    if (write(2, msg, sizeof msg - 1) < 0) {}
    backtrace_symbols_fd(frames, n, 2);
    if (const char* dir = getenv("BRC_CRASH_DIR")) {       // (test harnesses: the same frames into a file of their own)
        char path[512]; snprintf(path, sizeof path, "%s/brc_crash_%d.txt", dir, (int)getpid());
        const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd >= 0) { backtrace_symbols_fd(frames, n, fd); close(fd); }
    }
    signal(sig, SIG_DFL); raise(sig);
}

// ---------------------------------------------------------------- ranks: one process per GPU
// `bam-readcount --brc-ranks N ...` (or BRC_RANKS=N): this process becomes the COORDINATOR.  It never touches the HIP runtime: it
// starts N copies of itself (`--brc-rank-of r:N`), each a fresh process that binds to one GPU (BRC_DEVICES=a,b,.. names them, default
// 0..N-1; the child sees only its own through HIP_VISIBLE_DEVICES) and to its share of the CPUs, builds the same work list from the
// same files, cuts it — in FILE ORDER — into N contiguous slices of near-equal estimated work (the index's file offsets per 16-kb
// window: BamIndex::span_bytes) and runs its own slice exactly like a single process would.  No data-path exchange: a line depends on
// nothing but the reads over its position (and the position before it, which every piece fetches itself, bamreadcount.cpp:602);
// overlapping and repeated -l lines are printed as often as the list names them (:574-608), the deletion queue is cleared per line
// (:605).  Rank 0 writes to this process's stdout and stderr directly; the ranks behind it write into unlinked temporary files
// (--brc-tmpdir, TMPDIR, /tmp) which the coordinator copies to stdout in rank order, following rank r's file while it is still being
// written once the ranks before it are done (stdout /dev/null: they write there themselves).  stderr of those ranks arrives line by
// line, warning events tagged so that ReadWarnings' -w counters run over the whole run in file order.  A rank that fails ends the run
// where the reference would have stopped: the text of the ranks before it is out, the ranks behind it are stopped and their text dropped.
// Several command-line regions in one run stay one process: a deletion left pending by one region can hold back the deletions of every
// region behind it (the reference does not clear its queue there, :641-657), which only a process that has seen them all can know.
#include <sys/stat.h>
#include <sys/sysmacros.h>
#include <sys/sendfile.h>
#include <sys/wait.h>
#include <sched.h>
#include <poll.h>

static bool parse_rank_of(const std::string& v, int* r, int* n) {
    const size_t c = v.find(':');
    if (c == std::string::npos) return false;
    char* e = nullptr; const long a = strtol(v.c_str(), &e, 10); if (e != v.c_str() + c) return false;
    const long b = strtol(v.c_str() + c + 1, &e, 10); if (*e || e == v.c_str() + c + 1) return false;
    if (b < 1 || b > 4096 || a < 0 || a >= b) return false;
    *r = (int)a; *n = (int)b; return true;
}

static bool write_all(int fd, const char* p, size_t n) {
    while (n) { const ssize_t w = write(fd, p, n); if (w < 0) { if (errno == EINTR || errno == EAGAIN) continue; return false; } p += w; n -= (size_t)w; }
    return true;
}

static int coordinate(int argc, char** argv, const Options& o, int N) {
    const double t0 = now_s();
    // GPUs of the ranks
    std::vector<int> devs;
    if (const char* dv = getenv("BRC_DEVICES")) { for (const char* q = dv; *q;) { devs.push_back(atoi(q)); while (*q && *q != ',') ++q; if (*q) ++q; } }
    if (devs.empty()) for (int g = 0; g < N; ++g) devs.push_back(g);
    std::vector<std::string> visible;               // the caller's own HIP_VISIBLE_DEVICES: rank r's GPU is the devs[r]-th entry of it
    if (const char* hv = getenv("HIP_VISIBLE_DEVICES")) { std::string cur; for (const char* q = hv;; ++q) { if (*q == ',' || !*q) { visible.push_back(cur); cur.clear(); if (!*q) break; } else cur.push_back(*q); } }
    const bool narrow = !(getenv("BRC_RANK_NARROW") && atoi(getenv("BRC_RANK_NARROW")) == 0);
    struct stat st; const bool to_null = fstat(1, &st) == 0 && S_ISCHR(st.st_mode) && major(st.st_rdev) == 1 && minor(st.st_rdev) == 3;
    std::string tdir = !o.tmpdir.empty() ? o.tmpdir : (getenv("TMPDIR") && *getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp");
    auto tmpfd = [&]() -> int {
        std::string path = tdir + "/brc_rank_XXXXXX";
        std::vector<char> b(path.begin(), path.end()); b.push_back(0);
        const int fd = mkstemp(b.data());
        if (fd >= 0) unlink(b.data());
        return fd;
    };
    struct Child { pid_t pid = -1; int out = -1, err = -1, status = -1; bool reaped = false; int wstatus = 0; };
    std::vector<Child> ch((size_t)N);
    fflush(stdout); fflush(stderr);
    for (int r = 0; r < N; ++r) {
        Child& c = ch[(size_t)r];
        int sp[2];
        if (pipe(sp) != 0) { fprintf(stderr, "bam-readcount: cannot create a pipe for rank %d: %s\n", r, strerror(errno)); return 1; }
        if (r > 0) {
            if (!to_null) { c.out = tmpfd(); if (c.out < 0) { fprintf(stderr, "bam-readcount: cannot create a temporary file in %s for the text of rank %d: %s (--brc-tmpdir)\n", tdir.c_str(), r, strerror(errno)); return 1; } }
            c.err = tmpfd(); if (c.err < 0) { fprintf(stderr, "bam-readcount: cannot create a temporary file in %s: %s (--brc-tmpdir)\n", tdir.c_str(), strerror(errno)); return 1; }
        }
        const pid_t pid = fork();
        if (pid < 0) { fprintf(stderr, "bam-readcount: cannot start rank %d: %s\n", r, strerror(errno)); return 1; }
        if (pid == 0) {
            // the child: its text and stderr where the coordinator will look for them, its GPU, then a fresh image of this program
            close(sp[0]);
            if (r > 0) {
                if (c.out >= 0) dup2(c.out, 1);
                dup2(c.err, 2);
            }
            for (int k = 0; k < r; ++k) { if (ch[(size_t)k].status >= 0) close(ch[(size_t)k].status); }
            const int dev = devs[(size_t)r % devs.size()];
            char b[64];
            if (narrow) {
                if (!visible.empty()) { if ((size_t)dev < visible.size()) setenv("HIP_VISIBLE_DEVICES", visible[(size_t)dev].c_str(), 1); }
                else { snprintf(b, sizeof b, "%d", dev); setenv("HIP_VISIBLE_DEVICES", b, 1); }
                setenv("BRC_DEVICE", "0", 1);
            } else { snprintf(b, sizeof b, "%d", dev); setenv("BRC_DEVICE", b, 1); }
            unsetenv("BRC_DEVICES"); unsetenv("BRC_RANKS");
            snprintf(b, sizeof b, "%d", sp[1]); setenv("BRC_RANK_STATUS_FD", b, 1);
            std::vector<char*> av;
            av.push_back(argv[0]);
            std::string ro = "--brc-rank-of=" + std::to_string(r) + ":" + std::to_string(N);
            av.push_back(&ro[0]);
            for (int i = 1; i < argc; ++i) av.push_back(argv[i]);
            av.push_back(nullptr);
            execv("/proc/self/exe", av.data());
            execvp(argv[0], av.data());
            fprintf(stderr, "bam-readcount: cannot start rank %d: %s\n", r, strerror(errno));
            _exit(127);
        }
        close(sp[1]);
        c.pid = pid; c.status = sp[0];
    }
    auto reap = [&](Child& c, bool block) {
        if (c.reaped) return true;
        const pid_t w = waitpid(c.pid, &c.wstatus, block ? 0 : WNOHANG);
        if (w == c.pid) c.reaped = true;
        return c.reaped;
    };
    int ret = 0;
    int64_t gcount[BRC_N_WARN] = {0, 0, 0, 0};
    std::vector<double> secs((size_t)N, 0.0); std::vector<double> done_at((size_t)N, 0.0);
    std::vector<char> buf((size_t)1 << 20);
    const bool timing = getenv("BRC_CLI_TIMING") != nullptr;
    // A rank is DONE when it has written its last word to the status pipe (it flushes its text and its stderr first) — or when the pipe
    // closes without one (it died).  The process itself may take a good while longer to disappear: a rank that printed gigabytes holds as
    // many gigabytes of page-locked buffers, which the kernel unpins at exit (0.5-0.7 s for BASELINE config 5's share) — nothing this run has
    // to wait for: the coordinator goes on to the next rank's text, and leaves when the last rank is done.
    std::vector<std::string> status_line((size_t)N); std::vector<bool> status_eof((size_t)N, false);
    auto poll_status = [&](int r, int timeout_ms) -> bool {         // true: rank r has said its last word (or will never)
        Child& c = ch[(size_t)r];
        if (status_eof[(size_t)r]) return true;
        for (;;) {
            struct pollfd pf; pf.fd = c.status; pf.events = POLLIN; pf.revents = 0;
            const int pr = poll(&pf, 1, timeout_ms);
            if (pr < 0 && errno == EINTR) continue;
            if (pr <= 0) return false;
            char b[256]; const ssize_t g = read(c.status, b, sizeof b);
            if (g < 0 && (errno == EINTR || errno == EAGAIN)) continue;
            if (g <= 0) { status_eof[(size_t)r] = true; return true; }
            status_line[(size_t)r].append(b, (size_t)g);
            if (status_line[(size_t)r].find('\n') != std::string::npos) { status_eof[(size_t)r] = true; return true; }
            timeout_ms = 0;
        }
    };
    for (int r = 0; r < N && !ret; ++r) {
        Child& c = ch[(size_t)r];
        if (r > 0 && c.out >= 0) {
            // follow the rank's text: copy what is there, look whether the rank is done, copy the rest
            off_t off = 0; bool use_sendfile = true;
            for (;;) {
                const bool finished = poll_status(r, 0);
                struct stat fs; if (fstat(c.out, &fs) != 0) break;
                bool moved = false;
                while (off < fs.st_size) {
                    ssize_t w = -1;
                    if (use_sendfile) { w = sendfile(1, c.out, &off, (size_t)std::min<off_t>(fs.st_size - off, (off_t)1 << 30)); if (w < 0 && (errno == EINVAL || errno == ENOSYS)) use_sendfile = false; else if (w < 0 && (errno == EINTR || errno == EAGAIN)) continue; else if (w < 0) { off = fs.st_size; break; } }
                    if (!use_sendfile) {
                        const ssize_t g = pread(c.out, buf.data(), (size_t)std::min<off_t>(fs.st_size - off, (off_t)buf.size()), off);
                        if (g <= 0) { if (g < 0 && errno == EINTR) continue; off = fs.st_size; break; }
                        if (!write_all(1, buf.data(), (size_t)g)) { off = fs.st_size; break; }     // (a closed pipe: like the reference, carry on silently)
                        off += g;
                    }
                    moved = true;
                }
                if (finished) { struct stat f2; if (fstat(c.out, &f2) == 0 && f2.st_size > off) continue; break; }
                if (!moved) poll_status(r, 1);
            }
        }
        while (!poll_status(r, 1000)) {}
        done_at[(size_t)r] = now_s() - t0;
        // what the rank said when it ended: exit code, ReadWarnings' counters (rank 0 printed its own warnings), seconds
        int rc = 0;
        {
            const std::string& sline = status_line[(size_t)r];
            long long w[BRC_N_WARN] = {0, 0, 0, 0}; int src = 1; double s = 0;
            if (sscanf(sline.c_str(), "%d %lld %lld %lld %lld %lf", &src, &w[0], &w[1], &w[2], &w[3], &s) == 6) { secs[(size_t)r] = s; if (r == 0) for (int k = 0; k < BRC_N_WARN; ++k) gcount[k] = w[k]; rc = src; }
            else { reap(c, true); rc = WIFEXITED(c.wstatus) && WEXITSTATUS(c.wstatus) ? WEXITSTATUS(c.wstatus) : 1; }           // (a rank that ended without a word did not end well: wait for it, say how it went)
        }
        if (r > 0) {
            // its stderr, in order: plain lines as they are, tagged warning events through the run's counters
            fflush(stdout);
            std::string all; ssize_t g; off_t eo = 0;
            while ((g = pread(c.err, buf.data(), buf.size(), eo)) > 0 || (g < 0 && errno == EINTR)) if (g > 0) { all.append(buf.data(), (size_t)g); eo += g; }
            for (size_t i = 0; i < all.size();) {
                size_t j = i; while (j < all.size() && all[j] != '\n') ++j;
                if (all[i] == 1) { std::string ev(all, i + 1, j - (i + 1)); ev.push_back('\n'); print_warn_events(ev.data(), ev.size(), o.max_warnings, gcount, stderr); }
                else { fwrite(all.data() + i, 1, j - i, stderr); if (j < all.size()) fputc('\n', stderr); }
                i = j + 1;
            }
        }
        if (rc) ret = rc == 127 ? 1 : rc;
        if (c.reaped && WIFSIGNALED(c.wstatus)) { ret = 1; fprintf(stderr, "bam-readcount: rank %d ended on signal %d\n", r, WTERMSIG(c.wstatus)); }
    }
    // a failed run: the ranks behind the failure are stopped, their text is dropped.  (Ranks that are done are not waited for: see above.)
    if (ret) { for (int r = 0; r < N; ++r) if (!ch[(size_t)r].reaped && !status_eof[(size_t)r]) kill(ch[(size_t)r].pid, SIGTERM); }
    for (Child& c : ch) reap(c, false);
    if (timing) {
        fprintf(stderr, "ranks: %d processes", N);
        for (int r = 0; r < N; ++r) fprintf(stderr, "%s rank %d: %.3f s (its text was out after %.3f s)", r ? ";" : ":", r, secs[(size_t)r], done_at[(size_t)r]);
        fprintf(stderr, "; all told %.3f s\n", now_s() - t0);
    }
    fflush(stdout); fflush(stderr);
    return ret;
}

// the slice of `atoms` that rank r of n owns: contiguous, order-preserving, balanced by weight — shard.partition's rule (the atom that
// crosses the r-th boundary goes to whichever side leaves the smaller excess)
static void rank_slice(const std::vector<double>& w, int r, int n, size_t* lo, size_t* hi) {
    const size_t m = w.size();
    std::vector<double> cum(m); double t = 0; for (size_t i = 0; i < m; ++i) { t += w[i] > 1e-9 ? w[i] : 1e-9; cum[i] = t; }
    size_t start = 0, stop = 0;
    for (int k = 0; k <= r; ++k) {
        start = stop;
        if (k + 1 < n) {
            const double target = t * (double)(k + 1) / (double)n;
            stop = (size_t)(std::upper_bound(cum.begin(), cum.end(), target) - cum.begin());
            if (stop < m && stop >= start && (cum[stop] - target) < (target - (stop > 0 ? cum[stop - 1] : 0.0))) ++stop;
        } else stop = m;
        if (stop < start) stop = start;
        if (stop > m) stop = m;
    }
    *lo = start; *hi = stop;
}

int main(int argc, char** argv) {
    { struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_handler = crash_handler; sigemptyset(&sa.sa_mask); sa.sa_flags = SA_RESETHAND;
      sigaction(SIGSEGV, &sa, nullptr); sigaction(SIGBUS, &sa, nullptr); sigaction(SIGABRT, &sa, nullptr); sigaction(SIGFPE, &sa, nullptr); }
    Ctx c; std::string err;
    if (!parse_args(argc, argv, c.opt, &err)) { fprintf(stderr, "bam-readcount: %s\n", err.c_str()); return 1; }
    const Options& o = c.opt;
    // ---- ranks: this process is one of a group, or starts one
    int my_rank = -1, n_ranks = 1;
    if (!o.rank_of.empty()) {
        if (!parse_rank_of(o.rank_of, &my_rank, &n_ranks)) { fprintf(stderr, "bam-readcount: the argument ('%s') for option '--brc-rank-of' is invalid\n", o.rank_of.c_str()); return 1; }
        g_ranks = (unsigned)n_ranks; g_rank = my_rank;
        // this rank's share of the CPUs, as a contiguous slice of the ones the group may use (BRC_RANK_AFFINITY=0: left to the scheduler)
        if (n_ranks > 1 && !(getenv("BRC_RANK_AFFINITY") && atoi(getenv("BRC_RANK_AFFINITY")) == 0)) {
            cpu_set_t all; CPU_ZERO(&all);
            if (sched_getaffinity(0, sizeof all, &all) == 0) {
                std::vector<int> cpus; for (int k = 0; k < CPU_SETSIZE; ++k) if (CPU_ISSET(k, &all)) cpus.push_back(k);
                if ((int)cpus.size() >= n_ranks) {
                    cpu_set_t mine; CPU_ZERO(&mine);
                    const size_t a = cpus.size() * (size_t)my_rank / (size_t)n_ranks, b = cpus.size() * (size_t)(my_rank + 1) / (size_t)n_ranks;
                    for (size_t k = a; k < b; ++k) CPU_SET(cpus[k], &mine);
                    (void)sched_setaffinity(0, sizeof mine, &mine);
                }
            }
        }
    }
    const bool lead = my_rank <= 0;                 // the process whose informational lines reach the caller (a single process, or rank 0)
    const int status_fd = getenv("BRC_RANK_STATUS_FD") ? atoi(getenv("BRC_RANK_STATUS_FD")) : -1;
    const double t_start = now_s();
    auto leave = [&](int rc) {                      // a rank tells the coordinator how it ended (exit code, ReadWarnings' counters, seconds)
        if (my_rank >= 0 && status_fd >= 0) {
            char b[256]; const int n = snprintf(b, sizeof b, "%d %lld %lld %lld %lld %.6f\n", rc, (long long)c.wcount[0], (long long)c.wcount[1], (long long)c.wcount[2], (long long)c.wcount[3], now_s() - t_start);
            fflush(stdout); fflush(stderr);
            write_all(status_fd, b, (size_t)n);
        }
        return rc;
    };
    if (o.version) { if (lead) printf("bam-readcount version: 1.0.1-mi355x (engine %s, abi %d)\n", brc_engine_kind(), BRC_ABI_VERSION); return leave(1); }   // :467-470
    if (o.help || o.bam.empty()) { if (lead) { fputs(kUsage, stdout); fputs("\n", stdout); } return leave(1); }                                                   // :472-475
    {
        const long long want = o.ranks > 0 ? o.ranks : (getenv("BRC_RANKS") ? atoll(getenv("BRC_RANKS")) : 0);
        // (several command-line regions: one process — see "ranks" above; -D ends the run before any work)
        if (my_rank < 0 && want > 1 && !o.distribution && (!o.site_list.empty() || o.regions.size() == 1)) return coordinate(argc, argv, o, (int)std::min<long long>(want, 4096));
    }
    c.tag_warnings = my_rank > 0;
    if (lead) fprintf(stderr, "Minimum mapping quality is set to %d\n", o.min_mapq);                                                                 // :477
    if (!open_inputs(c, !lead)) return leave(1);
    const double t_inputs = now_s();
    if (lead) for (const std::string& l : c.header().expected) fprintf(stderr, "Expect library: %s in BAM\n", l.c_str());                                         // :526-529
    if (o.distribution) { if (lead) fprintf(stderr, "Not currently supporting distributions\n"); return leave(1); }                                          // :367 (the reference throws)
    // -d below any real depth changes which reads bam_plp_push keeps, and that depends on everything buffered before:
    // no internal pieces then (the planner is off for the same reason)
    if (o.max_cnt < 1000000) c.opt.chunk_bp = (long long)INT_MAX;
    // devices: BRC_DEVICES=0,2,3 or --brc-gpus N (devices 0..N-1); BRC_DEVICE=k for a single engine
    std::vector<int> devices;
    if (my_rank < 0) if (const char* dv = getenv("BRC_DEVICES")) { for (const char* q = dv; *q;) { devices.push_back(atoi(q)); while (*q && *q != ',') ++q; if (*q) ++q; } }
    if (my_rank >= 0) devices.push_back(getenv("BRC_DEVICE") ? atoi(getenv("BRC_DEVICE")) : 0);       // one rank, one GPU
    if (devices.empty()) for (long long g = 0; g < std::max<long long>(o.gpus, 1); ++g) devices.push_back((o.gpus <= 1 && getenv("BRC_DEVICE")) ? atoi(getenv("BRC_DEVICE")) : (int)g);
    if (my_rank < 0) {   // K engines per GPU, GPU-major round robin (engine i -> GPU i mod #GPUs)
        long long K = o.streams > 0 ? o.streams : (getenv("BRC_STREAMS") ? atoll(getenv("BRC_STREAMS")) : 1);
        if (K < 1) K = 1;
        if (o.max_cnt < 1000000) K = 1;                  // (no internal pieces then)
        const std::vector<int> one = devices;
        for (long long k = 1; k < K; ++k) devices.insert(devices.end(), one.begin(), one.end());
    }
    // Deletions left pending by one command-line region reach into the next (the reference does not clear its queue there,
    // :641-657) and can block its deletions from the first position to the last: such runs keep every piece on one engine,
    // in order.  (One region — a chromosome — and site lists spread over the engines.)
    if (o.site_list.empty() && o.regions.size() > 1) devices.resize(1);
    size_t N = devices.size();
    // The engine (HIP runtime start, streams) is created on a thread of its own while this one reads the index and the site
    // list and — inside the first work item — the reference and the first reads; whoever needs the engine waits for it.
    double t_engine0 = 0;
    std::promise<int> eng_promise; std::shared_future<int> eng_ready = eng_promise.get_future().share();
    std::thread eng_thread([&]() { const int r = make_engine(c, devices[0]); t_engine0 = now_s(); eng_promise.set_value(r); });
    bool eng_reported = false;
    auto wait_engine = [&]() -> bool {
        const int r = eng_ready.get();
        if (r && !eng_reported) { eng_reported = true; fprintf(stderr, "bam-readcount: cannot create the MI355X engine: %s\n", brc_strerror(r)); }
        return r == 0;
    };
    c.need_engine = wait_engine;
    struct JoinEng { std::thread& t; ~JoinEng() { if (t.joinable()) t.join(); } } join_eng{eng_thread};

    // ---- the work items, in file order
    // (a rank of a group first lists ATOMS — one -l line, or one 64-kb piece of a region / wide line, each with a weight —, takes its
    // slice of them and joins what lies side by side in it back into batches and regions: see below)
    const bool atoms = my_rank >= 0 && n_ranks > 1;
    // (64 kb: four windows of the index; BRC_RANK_CUT: tests cut their kilobase contigs finer)
    const int64_t rank_cut = o.max_cnt < 1000000 ? (int64_t)INT_MAX : (getenv("BRC_RANK_CUT") && atoll(getenv("BRC_RANK_CUT")) > 0 ? (int64_t)atoll(getenv("BRC_RANK_CUT")) : (int64_t)65536);
    std::vector<Work> items;
    int ret = 0;
    auto add_region = [&](int tid, int64_t beg0, int64_t end, bool site_mode) {
        // one engine: the region as a whole (run_region cuts it and fetches the next piece in the background); several:
        // its pieces become items of their own so that they spread over the GPUs
        const BamHeader& h = c.header();
        if (end > (int64_t)h.lengths[(size_t)tid] + 1000) end = (int64_t)h.lengths[(size_t)tid] + 1000;
        if (end < beg0) end = beg0;
        const int64_t step = atoms ? rank_cut : (N > 1 ? (int64_t)c.opt.chunk_bp : (int64_t)INT_MAX);
        bool first = true;
        int64_t a = beg0;
        do {
            // (atoms end on multiples of the cut: the index's windows, what the weights are known by)
            const int64_t b = std::min<int64_t>(atoms && step < (int64_t)INT_MAX ? (a / step + 1) * step : a + step, end);
            Work w; w.kind = 0; w.tid = tid; w.beg0 = a; w.end = b; w.site_mode = site_mode && b >= end;
            w.keep_queue = first && !site_mode;       // a -l line starts from an empty queue (:605 cleared it after the previous line)
            w.inner_piece = !first;
            items.push_back(std::move(w));
            first = false; a = b;
        } while (a < end);
    };
    if (!o.site_list.empty()) {
        FILE* fp = fopen(o.site_list.c_str(), "r");
        if (!fp) { if (lead) fprintf(stderr, "Failed to open region list file: %s\n", o.site_list.c_str()); return leave(1); }            // :535-538
        if (!c.is_cram && !c.idx.load(o.bam)) { if (lead) fprintf(stderr, "BAM indexing file is not available.\n"); return leave(1); }                  // :548-551
        // the planner needs -d to be out of play (its drop rule depends on what else is buffered) and narrow lines
        const bool plan = o.plan_sites > 0 && o.max_cnt >= 1000000 && !c.is_cram;
        Work pend; pend.kind = 1; int64_t pend_bp = 0;
        auto flush = [&]() { if (!pend.sites.empty()) { items.push_back(std::move(pend)); pend = Work(); pend.kind = 1; pend_bp = 0; } };
        char* line = nullptr; size_t line_cap = 0; ssize_t line_len;
        while ((line_len = getline(&line, &line_cap, fp)) >= 0) {                   // getline + ss >> ref_name >> beg >> end (:574-577)
            std::string name; int beg, end;
            if (!parse_site_line(line, (size_t)line_len, &name, &beg, &end)) continue;
            auto it = c.header().name2tid.find(name);
            if (it == c.header().name2tid.end()) {                                   // :580-582 — printed when the reference's loop reaches the line,
                flush();                                                             // i.e. behind the warnings of the lines before it: a work item of its own
                Work w; w.kind = 3; w.err = name + " not found in bam file. Region " + name + " " + std::to_string(beg) + " " + std::to_string(end) + " skipped.\n";
                items.push_back(std::move(w)); continue;
            }
            // pileup_func works on [beg - 2, end) in 0-based terms (:268 takes the position before the first printed one for
            // its deletions): a line whose end lies before that touches nothing
            if ((int64_t)end < (int64_t)beg - 1) continue;
            if (beg < 1) beg = 1;
            // (windows near the end of a contig run on their own: fetch_func's "Request for position" lines carry real coordinates)
            if (plan && (int64_t)end - beg < 1000 && (int64_t)end + 100000 < (int64_t)c.header().lengths[(size_t)it->second]) {
                Site st; st.tid = it->second; st.beg0 = (int64_t)beg - 1; st.end = end < beg - 1 ? beg - 1 : end;
                const int64_t clen = (int64_t)c.header().lengths[(size_t)st.tid];
                if (st.end > clen + 1000) st.end = std::max<int64_t>(clen + 1000, st.beg0);
                pend.sites.push_back(st); pend_bp += (st.end - st.beg0) + 600;
                if (atoms || (long long)pend.sites.size() >= o.plan_sites || pend_bp > 4 * (int64_t)c.opt.chunk_bp) flush();
                continue;
            }
            flush();
            add_region(it->second, (int64_t)beg - 1, end, true);
        }
        flush();
        free(line);
        fclose(fp);
    } else if (!o.regions.empty()) {
        if (!c.is_cram && !c.idx.load(o.bam)) { if (lead) fprintf(stderr, "BAM indexing file is not available.\n"); return leave(1); }                  // :637-640
        for (const std::string& r : o.regions) {
            int tid; int64_t beg, end; std::string log;
            if (!parse_region(c.header(), r, &tid, &beg, &end, &log)) {              // :645-648: the regions before it have been printed
                Work w; w.kind = 2; w.err = log + "Invalid region " + r + "\n"; w.rc = 1; items.push_back(std::move(w)); break;
            }
            if (!log.empty()) { Work w; w.kind = 3; w.err = log; items.push_back(std::move(w)); }
            add_region(tid, beg, end, false);
        }
    } else {
        if (lead) fprintf(stderr, "bam-readcount: give a region or a site list (-l); the reference's whole-file mode skips its per-read "
                                  "pre-processing (bamreadcount.cpp:624 FIXME) and is not reproduced\n");
        return leave(1);
    }

    if (atoms) {
        // ---- this rank's slice of the atoms, in file order, by estimated work: the compressed bytes the index places under an atom —
        // for a -l line, what its indexed fetch decodes: from the start of its 16-kb window to the line (bamio.h: span_bytes)
        std::vector<double> w(items.size(), 0.0); bool known = !c.is_cram;
        for (size_t i = 0; i < items.size() && known; ++i) {
            const Work& a = items[i];
            if (a.kind == 0) { w[i] = c.idx.span_bytes(a.tid, std::max<int64_t>(a.beg0 - 1, 0), std::max<int64_t>(a.end, a.beg0)); if (w[i] < 0) known = false; w[i] += 64.0; }
            else if (a.kind == 1) { const Site& s = a.sites[0]; w[i] = c.idx.span_bytes(s.tid, (std::max<int64_t>(s.beg0 - 1, 0) >> 14) << 14, std::max<int64_t>(s.end, s.beg0)); if (w[i] < 0) known = false; w[i] += 2048.0; }
        }
        if (!known) for (size_t i = 0; i < items.size(); ++i) { const Work& a = items[i]; w[i] = a.kind == 0 ? (double)(a.end - a.beg0) + 1.0 : (a.kind == 1 ? 1000.0 : 0.0); }   // (no offsets to weigh by: positions)
        size_t lo = 0, hi = 0;
        rank_slice(w, my_rank, n_ranks, &lo, &hi);
        // an error item ends the run where the reference's loop would have stopped: the rank that owns it reports it, the ranks behind
        // it have nothing to do (the coordinator drops whatever they print)
        std::vector<Work> mine;
        for (size_t i = lo; i < hi; ++i) {
            Work& a = items[i];
            if (!mine.empty()) {
                Work& p = mine.back();
                if (a.kind == 1 && p.kind == 1 && (long long)p.sites.size() < std::max<long long>(o.plan_sites, 1)) {
                    int64_t bp = 0; for (const Site& s : p.sites) bp += (s.end - s.beg0) + 600;
                    if (bp <= 4 * (int64_t)c.opt.chunk_bp) { p.sites.push_back(a.sites[0]); continue; }
                }
                if (a.kind == 0 && p.kind == 0 && a.inner_piece && a.tid == p.tid && a.beg0 == p.end && !p.site_mode) { p.end = a.end; p.site_mode = a.site_mode; continue; }
            }
            mine.push_back(std::move(a));
        }
        if (getenv("BRC_CLI_TIMING")) {
            double wm = 0, wt = 0; for (size_t i = 0; i < w.size(); ++i) { wt += w[i]; if (i >= lo && i < hi) wm += w[i]; }
            fprintf(stderr, "rank %d of %d: atoms [%zu, %zu) of %zu, %.4f of the estimated work (%s), %zu work items\n", my_rank, n_ranks, lo, hi, items.size(), wt > 0 ? wm / wt : 0.0, known ? "index offsets" : "positions", mine.size());
        }
        items.swap(mine);
        N = 1;
        // a rank's part of a region is a fraction of it: pieces small enough that the rank's decode | GPU | format pipeline still has a
        // few of them to overlap (a part of one piece would run its three stages one after the other)
        if (o.max_cnt >= 1000000) {
            int64_t widest = 0; for (const Work& w2 : items) if (w2.kind == 0) widest = std::max<int64_t>(widest, w2.end - w2.beg0);
            if (widest > 0 && widest < 4 * (int64_t)c.opt.chunk_bp) c.opt.chunk_bp = std::min<long long>(c.opt.chunk_bp, std::max<long long>(131072, ((widest / 4 + 65535) / 65536) * 65536));
        }
    }

    if (items.size() < N) N = std::max<size_t>(items.size(), 1);      // engines without work are never created
    if (N > 1) {    // the engines share this process's CPUs: each gets its part of the decode and formatter pools
        g_engines = (unsigned)N;
        c.opt.chunk_given = true;        // (the items ARE the pieces here: one brc_format_region per item, whose text stays in its engine's buffer)
        // (told to every engine with BRC_OPT_FORMAT_THREADS in make_engine — never through setenv(): the engine-creation
        // threads are inside the HIP runtime by now, which reads the environment while it starts, and getenv() racing a
        // setenv() that reallocates `environ` was a rare SIGSEGV before the first line of output)
        g_format_threads = std::max(2u, effective_cpus() / (unsigned)N);
        // engine 0 was created before the number of engines was known: it is told its share here, explicitly (the other
        // engines get theirs in make_engine) — not as a side effect of whoever waits for it next
        if (eng_ready.get() == 0) brc_set_option(c.eng, BRC_OPT_FORMAT_THREADS, (int64_t)g_format_threads.load());
    } else if (g_ranks > 1) {
        // a rank among others on this node: the engine's own pools (staging, formatter) get this rank's share of the CPUs too
        if (eng_ready.get() == 0) brc_set_option(c.eng, BRC_OPT_FORMAT_THREADS, (int64_t)std::max(2u, effective_cpus()));
    }
    // pin the text buffers of a long region's pieces while the first reads are being decoded
    std::thread pin_ahead;
    {
        int64_t widest = 0; for (const Work& w : items) if (w.kind == 0) widest = std::max<int64_t>(widest, std::min<int64_t>(w.end - w.beg0, auto_chunk(c, w.tid, w.beg0, w.end)));
        const bool dev_text = !(getenv("BRC_DEVICE_TEXT") && atoi(getenv("BRC_DEVICE_TEXT")) == 0);
        if (dev_text && widest >= 100000 && !(getenv("BRC_PIN_AHEAD") && atoi(getenv("BRC_PIN_AHEAD")) == 0)) {
            const int64_t bytes = widest * (int64_t)(o.per_lib ? std::max<size_t>(c.libs.size(), 1) : 1) * 400;
            pin_ahead = std::thread([&c, eng_ready, bytes]() { if (eng_ready.get() == 0) brc_set_option(c.eng, BRC_OPT_EXPECT_TEXT, bytes); });
        }
    }
    struct JoinPin { std::thread& t; ~JoinPin() { if (t.joinable()) t.join(); } } join_pin{pin_ahead};
    const bool clean_exit = getenv("BRC_CLEAN_EXIT") != nullptr || getenv("BRC_ENGINE_TIMING") != nullptr;
    if (N == 1) {
        // site-list batches: the reads of batch k + 1 are fetched and decoded (a pool of BAM handles) while batch k is laid out,
        // computed and printed
        std::unique_ptr<SiteFetch> ahead_f; std::thread ahead_t; size_t ahead_i = (size_t)-1;
        struct JoinAhead { std::thread& t; ~JoinAhead() { if (t.joinable()) t.join(); } } join_ahead{ahead_t};
        static const bool site_ahead = !(getenv("BRC_SITE_AHEAD") && atoi(getenv("BRC_SITE_AHEAD")) == 0);
        for (size_t i = 0; i < items.size(); ++i) {
            Work& w = items[i];
            if (w.kind == 2) { fflush(stdout); fputs(w.err.c_str(), stderr); ret = 1; break; }
            if (w.kind == 3) { fflush(stdout); fputs(w.err.c_str(), stderr); continue; }
            if (w.kind == 1 && site_ahead) {
                std::unique_ptr<SiteFetch> cur;
                const double w0 = now_s();
                if (ahead_i == i) { ahead_t.join(); cur = std::move(ahead_f); }
                else { cur.reset(new SiteFetch()); fetch_site_batch(c, w.sites, *cur); }
                c.t_site_fetch += now_s() - w0;
                if (i + 1 < items.size() && items[i + 1].kind == 1) {
                    ahead_f.reset(new SiteFetch()); ahead_i = i + 1;
                    SiteFetch* fp = ahead_f.get(); const std::vector<Site>* sp = &items[i + 1].sites; const Ctx* cp = &c;
                    ahead_t = std::thread([cp, sp, fp]() { fetch_site_batch(*cp, *sp, *fp); });
                }
                if ((ret = run_site_batch(c, w.sites, cur.get()))) break;
                continue;
            }
            if ((ret = run_item(c, w))) break;
        }
    } else {
        // engine of every item: round robin, the first piece of a command-line region after the region before it
        int last_engine = -1; size_t rr = 0;
        for (Work& w : items) {
            if (w.kind == 0 && w.keep_queue && last_engine >= 0) w.engine = last_engine;
            else w.engine = (int)(rr++ % N);
            if (w.kind == 0) last_engine = w.engine;
        }
        std::vector<std::unique_ptr<Ctx> > ctxs(N);
        std::mutex mu; std::condition_variable cv;
        size_t printed = 0; bool abort_all = false;
        int64_t gcount[BRC_N_WARN] = {0, 0, 0, 0};
        auto worker = [&](size_t g) {
            Ctx* wc = &c;
            if (g > 0) {                                                             // own handles, own engine
                ctxs[g].reset(new Ctx()); wc = ctxs[g].get(); wc->opt = c.opt;
                bool ok = open_inputs(*wc, true) && (wc->is_cram || wc->idx.load(c.opt.bam)) && make_engine(*wc, devices[g]) == 0;
                if (!ok) {
                    std::lock_guard<std::mutex> lk(mu);
                    for (Work& w : items) if ((size_t)w.engine == g && !w.done) { w.rc = 1; w.err = "bam-readcount: cannot set up the engine of GPU " + std::to_string(devices[g]) + "\n"; w.done = true; }
                    cv.notify_all(); return;
                }
            }
            size_t held = (size_t)-1;                                               // item whose text sits in this engine's buffer
            wc->pre_format = [&]() {
                if (held == (size_t)-1) return;
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&]() { return abort_all || printed > held; });
                held = (size_t)-1;
            };
            for (size_t i = 0; i < items.size(); ++i) {
                Work& w = items[i];
                if ((size_t)w.engine != g) continue;
                {   // bound the text held in memory: stay within 2 N items of the one being written
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&]() { return abort_all || i < printed + 2 * N; });
                    if (abort_all) return;
                }
                int r = w.rc;
                if (w.kind != 2 && w.kind != 3) {
                    wc->out_buf = &w.out; wc->err_buf = &w.err; wc->wev_buf = &w.wev;
                    // (a piece is formatted by exactly one brc_format_region call: its text can stay where it is)
                    const bool zc = w.kind == 0 && w.end - w.beg0 <= (int64_t)c.opt.chunk_bp;
                    wc->zc_parts = zc ? &w.parts : nullptr; wc->zc_lens = zc ? &w.lens : nullptr; wc->zc_n = zc ? &w.n_parts : nullptr;
                    if (!zc) wc->pre_format();
                    // the reads of this engine's next piece are fetched meanwhile (own handles: fetch_chunk only touches the
                    // context's reader pool, which a single-piece item does not use after its own fetch)
                    std::thread ahead;
                    size_t j = i + 1;
                    while (j < items.size() && (size_t)items[j].engine != g) ++j;
                    if (zc && j < items.size() && items[j].kind == 0 && !wc->is_cram) {
                        if (!wc->pf_buf) wc->pf_buf.reset(new Fetched());
                        const Work& nx = items[j];
                        const int64_t nb = std::min<int64_t>(nx.beg0 + (int64_t)c.opt.chunk_bp, nx.end);
                        if (!(wc->pf.valid && wc->pf.tid == nx.tid && wc->pf.a == nx.beg0 && wc->pf.b == nb)) {
                            Ctx* pc = wc; const int ntid = nx.tid; const int64_t na = nx.beg0;
                            // (started only after this item took its own prefetched reads: run_region swaps them out first thing)
                            wc->after_take = [pc, ntid, na, nb, &ahead]() { ahead = std::thread([pc, ntid, na, nb]() { fetch_chunk(*pc, ntid, na, nb, *pc->pf_buf); pc->pf.tid = ntid; pc->pf.a = na; pc->pf.b = nb; }); };
                        }
                    }
                    r = run_item(*wc, w);
                    wc->after_take = nullptr;
                    if (ahead.joinable()) { ahead.join(); wc->pf.valid = true; }
                    if (zc && w.n_parts) held = i;
                }
                std::lock_guard<std::mutex> lk(mu);
                w.rc = r; w.done = true; cv.notify_all();
            }
        };
        std::vector<std::thread> th;
        for (size_t g = 0; g < N; ++g) th.emplace_back(worker, g);
        for (size_t i = 0; i < items.size(); ++i) {
            Work& w = items[i];
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return w.done; }); }
            if (!w.out.empty()) fwrite(w.out.data(), 1, w.out.size(), stdout);
            if (w.n_parts) write_parts(w.parts, w.lens, w.n_parts);
            if (!w.wev.empty()) print_warn_events(w.wev.data(), w.wev.size(), c.opt.max_warnings, gcount, stderr);   // the global -w counters, in file order
            if (!w.err.empty()) fputs(w.err.c_str(), stderr);
            std::string().swap(w.out);
            { std::lock_guard<std::mutex> lk(mu); printed = i + 1; if (w.rc) { abort_all = true; ret = 1; } cv.notify_all(); }
            if (ret) break;
        }
        { std::lock_guard<std::mutex> lk(mu); abort_all = abort_all || ret != 0; printed = items.size(); cv.notify_all(); }
        for (std::thread& t : th) t.join();
        // (the process is about to end: see below for why the engines are only destroyed on request)
        for (size_t g = 1; g < N; ++g) if (ctxs[g]) { for (int w = 0; w < BRC_N_WARN; ++w) c.warn[w] += ctxs[g]->warn[w]; if (ctxs[g]->eng && clean_exit) brc_destroy(ctxs[g]->eng); }
        if (!clean_exit) for (auto& p : ctxs) (void)p.release();
    }
    const std::string who = my_rank >= 0 ? "rank " + std::to_string(my_rank) + ": " : std::string();
    if (getenv("BRC_CLI_TIMING")) fprintf(stderr, "%sstartup: open inputs %.3f s, create engine %.3f s\n", who.c_str(), t_inputs - t_start, t_engine0 - t_inputs);
    if (getenv("BRC_CLI_TIMING")) fprintf(stderr, "%stiming: fetch+decode %.3f s, engine (push, upload, kernels, download) %.3f s, format %.3f s, write %.3f s\n", who.c_str(), c.t_fetch, c.t_engine, c.t_format, c.t_write);
    if (getenv("BRC_CLI_TIMING") && c.n_site_lines)
        fprintf(stderr, "%ssites: %llu lines in %llu clusters, %llu engine regions of %llu reads all told: waiting for indexed fetch + decode %.3f s (the fetches themselves, next batch behind the current one: %.3f s), layout on the virtual axis %.3f s, engine (push, upload, kernels, download) %.3f s, cutting the lines out + writing %.3f s\n",
                who.c_str(), (unsigned long long)c.n_site_lines, (unsigned long long)c.n_site_clusters, (unsigned long long)c.n_site_batches, (unsigned long long)c.n_site_reads, c.t_site_fetch, c.t_site_fetch_threads, c.t_site_layout, c.t_site_engine, c.t_site_format);
    // Everything has been written.  Unpinning and freeing gigabytes of staging and the HIP runtime's own teardown only delay
    // the exit of a process that is done: leave them to the operating system (BRC_CLEAN_EXIT=1 keeps the orderly path).
    fflush(stdout); fflush(stderr);
    leave(ret);
    // (a rank: whoever reads the run's stdout / stderr through a pipe sees its end when the last holder lets go — now, not when the kernel has
    // unpinned this process's gigabytes)
    if (my_rank >= 0 && !clean_exit) { close(1); close(2); }
    if (!clean_exit) _exit(ret);
    if (wait_engine()) brc_destroy(c.eng);
    return ret;
}
