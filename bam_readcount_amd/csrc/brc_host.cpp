// brc_host.cpp — C-ABI glue (include/brc.h) and the host-side halves of the path:
//   * brc_push_reads : staging with rebased offsets, bam_plp_push's max-count drop rule, region extent
//   * brc_fetch_result : allele text + std::map ordering of indel buckets
//   * brc_format_region : pileup_func's record assembly (bamreadcount.cpp:351-416), IndelQueue::process
//                         (IndelQueue.cpp:3-15) and operator<<(BasicStat) (BasicStat.cpp:110-159) — from dense planes, from
//                         the compact result (BRC_OPT_TEXT_ONLY), or as the finishing pass over device-written text
//                         (BRC_OPT_DEVICE_TEXT: format_device_text)
//   * brc_region_warnings : the stderr side (ReadWarnings)
// No accumulation happens here: every number printed comes out of the device planes.
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include "brc_host.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <deque>
#include <thread>
#include <functional>
#include <queue>
#include <condition_variable>
#include <functional>

namespace brc {

// CPUs this process may really use at once: the affinity mask, capped by a cgroup CPU quota (cpu.max / cfs_quota_us).
// Pools sized by hardware_concurrency() on a 256-thread host inside a 16-CPU container burn the quota in a fraction of
// every scheduling period and are throttled for the rest of it.
static unsigned probe_cpus() {
    unsigned n = std::thread::hardware_concurrency(); if (n == 0) n = 1;
    double quota = 0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                        // cgroup v2: "<quota|max> <period>"
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
    if (n < 1) n = 1;
    return n;
}
unsigned effective_cpus() { static const unsigned cached = probe_cpus(); return cached; }   // (initialised once, thread-safe)

static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// run fn(i) for i in [0, n) on up to nthr threads (dynamic hand-out); the caller's thread works too
template <class F>
static void parallel_for(int64_t n, unsigned nthr, F fn) {
    if (n <= 0) return;
    if (nthr > (unsigned)n) nthr = (unsigned)n;
    if (nthr <= 1) { for (int64_t i = 0; i < n; ++i) fn(i); return; }
    std::atomic<int64_t> next(0);
    auto work = [&]() { for (;;) { const int64_t i = next.fetch_add(1); if (i >= n) break; fn(i); } };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nthr; ++t) th.emplace_back(work);
    work();
    for (std::thread& t : th) t.join();
}

// A small persistent pool for the engine's staging stage (brc_push_reads): a 1-Mbp piece brings 200 000 reads — a few
// milliseconds of per-read work — so threads spawned per call would cost more than they bring.  run(n, fn): fn(i) for
// i in [0, n), handed out dynamically; the caller works too; returns when all are done.
class Pool {
    std::vector<std::thread> th;
    std::mutex mu; std::condition_variable cv_go, cv_done;
    std::function<void(int64_t)> job; int64_t job_n = 0; std::atomic<int64_t> next{0};
    uint64_t gen = 0; unsigned busy = 0; bool stop = false;
    void worker() {
        uint64_t seen = 0;
        for (;;) {
            { std::unique_lock<std::mutex> lk(mu); cv_go.wait(lk, [&] { return stop || gen != seen; }); if (stop) return; seen = gen; }
            for (;;) { const int64_t i = next.fetch_add(1); if (i >= job_n) break; job(i); }
            { std::lock_guard<std::mutex> lk(mu); if (--busy == 0) cv_done.notify_one(); }
        }
    }
  public:
    // (thread creation can fail — std::system_error under a thread limit: the pool then runs with the workers it got, none in the
    // worst case; an exception leaving the constructor with started threads in `th` would end in std::terminate)
    explicit Pool(unsigned n) { try { th.reserve(n); for (unsigned t = 0; t < n; ++t) th.emplace_back([this] { worker(); }); } catch (...) {} }
    ~Pool() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv_go.notify_all(); for (std::thread& t : th) t.join(); }
    unsigned size() const { return (unsigned)th.size() + 1; }
    template <class F> void run(int64_t n, F fn) {
        if (n <= 0) return;
        if (th.empty() || n == 1) { for (int64_t i = 0; i < n; ++i) fn(i); return; }
        { std::lock_guard<std::mutex> lk(mu); job = fn; job_n = n; next = 0; busy = (unsigned)th.size(); ++gen; }
        cv_go.notify_all();
        for (;;) { const int64_t i = next.fetch_add(1); if (i >= n) break; fn(i); }
        std::unique_lock<std::mutex> lk(mu); cv_done.wait(lk, [&] { return busy == 0; });
    }
};

void Staged::init(const HostAlloc* A) {
    pos.A = A; flag.A = A; mapq.A = A; lib.A = A; l_qseq.A = A; n_cigar.A = A; cig_off.A = A; seq_off.A = A; qual_off.A = A;
    nm.A = A; sm.A = A; tags.A = A; cigar.A = A; seq4.A = A; qual.A = A; bq_row.A = A; wide.A = A; piece_cnt.A = A; piece_off.A = A; iev_off.A = A; qnames.A = A; qname_off.A = A;
}
void Staged::clear() {
    pos.clear(); flag.clear(); mapq.clear(); lib.clear(); l_qseq.clear(); n_cigar.clear(); cig_off.clear(); seq_off.clear();
    qual_off.clear(); nm.clear(); sm.clear(); tags.clear(); cigar.clear(); seq4.clear(); qual.clear(); bq_row.clear(); wide.clear();
    bq_elems = 0; memset(len_hist, 0, sizeof len_hist); n = 0; min_pos = 0; max_end = 0; n_indel_ops = 0;
    seq_seg.clear(); qual_seg.clear(); seq_total = 0; qual_total = 0;
    piece_cnt.clear(); piece_off.clear(); iev_off.clear(); lib_base.clear(); n_pieces = 0; max_lqseq = 0; max_ncigar = 0; has_empty_m = false; has_eqx = false; max_span = 0; qnames.clear(); qname_off.clear();
    win_beg.clear(); win_end.clear();
}
std::vector<uint16_t> Staged::wanted_tiles(int32_t pos0, int64_t P) const {
    std::vector<uint16_t> w;
    if (win_beg.empty() || P <= 0) return w;
    const int64_t nt = (P + TILE - 1) / TILE;
    w.assign((size_t)nt, (uint16_t)TILE_UNWANTED);
    for (size_t i = 0; i < win_beg.size(); ++i) {
        int64_t k0 = (int64_t)win_beg[i] - 1 - pos0, k1 = (int64_t)win_end[i] - pos0;      // plane indices of [beg - 1, end), as brc_format_window cuts them
        if (k0 < 0) k0 = 0;
        if (k1 > P) k1 = P;
        for (int64_t t = k0 / TILE; k1 > k0 && t <= (k1 - 1) / TILE; ++t) {
            const uint32_t a = (uint32_t)std::max<int64_t>(k0 - t * TILE, 0), b = (uint32_t)std::min<int64_t>(k1 - 1 - t * TILE, TILE - 1);
            uint16_t& x = w[(size_t)t];
            if (x == (uint16_t)TILE_UNWANTED) x = (uint16_t)(a | (b << 8));
            else x = (uint16_t)(std::min<uint32_t>(x & 0xffu, a) | (std::max<uint32_t>(x >> 8, b) << 8));
        }
    }
    return w;
}
void Staged::destroy() {
    pos.destroy(); flag.destroy(); mapq.destroy(); lib.destroy(); l_qseq.destroy(); n_cigar.destroy(); cig_off.destroy();
    seq_off.destroy(); qual_off.destroy(); nm.destroy(); sm.destroy(); tags.destroy(); cigar.destroy(); seq4.destroy(); qual.destroy(); bq_row.destroy(); wide.destroy();
    piece_cnt.destroy(); piece_off.destroy(); iev_off.destroy(); qnames.destroy(); qname_off.destroy();
}
// library-major slots: all pieces of library 0 in file order, then library 1, ... (one stream without -p)
void Staged::layout_pieces(int Lp, bool per_lib) {
    lib_base.assign((size_t)Lp + 1, 0);
    for (int64_t i = 0; i < n; ++i) if (piece_cnt.p[i]) lib_base[(size_t)(per_lib ? lib.p[i] : 0) + 1] += piece_cnt.p[i];
    for (int l = 0; l < Lp; ++l) lib_base[(size_t)l + 1] += lib_base[(size_t)l];
    n_pieces = lib_base[(size_t)Lp];
    std::vector<int64_t> cur(lib_base.begin(), lib_base.end() - 1);
    piece_off.n = (size_t)n;
    for (int64_t i = 0; i < n; ++i) {
        const int l = per_lib ? (int)lib.p[i] : 0;
        // (a read without pieces takes the running slot of its library — of library 0 when it has none —: with one library piece_off[]
        // is then the non-decreasing prefix sum the read-wise tile compaction searches, k_compact_reads)
        if (!piece_cnt.p[i]) { piece_off.p[i] = (uint32_t)cur[(size_t)((l >= 0 && l < Lp) ? l : 0)]; continue; }
        piece_off.p[i] = (uint32_t)cur[(size_t)l]; cur[(size_t)l] += piece_cnt.p[i];
    }
    // event-byte rows, library-major as well: the pieces a (tile, library) wave stages one after the other then lie one
    // after the other in the stream (reads without a library, which have no pieces, go last).  Without -p this is the
    // file order brc_push_reads already assigned.
    if (per_lib && Lp > 1) {
        std::vector<uint64_t> rows((size_t)Lp + 2, 0);
        auto slot = [&](int64_t i) { const int l = (int)lib.p[i]; return (size_t)((l >= 0 && l < Lp) ? l : Lp); };
        for (int64_t i = 0; i < n; ++i) rows[slot(i) + 1] += ((uint64_t)l_qseq.p[i] + 15u) & ~(uint64_t)15u;
        for (int l = 0; l <= Lp; ++l) rows[(size_t)l + 1] += rows[(size_t)l];
        for (int64_t i = 0; i < n; ++i) { const size_t k = slot(i); bq_row.p[i] = rows[k]; rows[k] += ((uint64_t)l_qseq.p[i] + 15u) & ~(uint64_t)15u; }
    }
}
uint64_t Staged::wide_layout(WidePair* pairs, uint32_t first16) const {
    uint64_t w = 0; size_t k = 0;
    for (int64_t i = 0; i < n; ++i) if (wide.p[i]) {
        if (pairs) { WidePair x; x.read = (uint32_t)i; x.w16 = first16 + (uint32_t)(w >> 4); pairs[k++] = x; }
        w += ((uint64_t)l_qseq.p[i] + 15u) & ~(uint64_t)15u;
    }
    return w;
}
// eb_make's escape predicate (brc_core.h) over a read's bytes as they were pushed: a quality of 0 or above 62, a base code
// that is not one of A C G T.  Eight qualities / sixteen base codes at a time.
static bool read_has_escape_words(const uint8_t* qual, const uint8_t* seq4, int32_t L) {
    if (L <= 0) return false;
    const uint64_t K01 = 0x0101010101010101ull, K80 = 0x8080808080808080ull, K7F = 0x7f7f7f7f7f7f7f7full;
    int32_t j = 0;
    for (; j + 8 <= L; j += 8) {
        uint64_t x; memcpy(&x, qual + j, 8);
        const uint64_t zero = (x - K01) & ~x & K80;                                  // some byte is 0
        const uint64_t big = (((x & K7F) + (uint64_t)(128 - EB_ESC) * K01) | x) & K80;   // some byte is >= 63
        if (zero | big) return true;
    }
    for (; j < L; ++j) if ((uint8_t)(qual[j] - 1u) >= (uint8_t)(EB_ESC - 1)) return true;
    const int32_t nb = L / 2;                                                       // whole bytes: two base codes each
    const uint64_t K11 = 0x1111111111111111ull, K88 = 0x8888888888888888ull;
    int32_t t = 0;
    for (; t + 8 <= nb; t += 8) {
        uint64_t x; memcpy(&x, seq4 + t, 8);
        if ((x - K11) & ~x & K88) return true;                                      // some code is 0 ('=')
        if (x & (x - K11)) return true;                                             // (no borrows now) some code has two bits and more
    }
    auto bad = [](uint32_t n) { return n == 0u || (n & (n - 1u)) != 0u; };
    for (; t < nb; ++t) if (bad(seq4[t] >> 4) || bad(seq4[t] & 15u)) return true;
    if ((L & 1) && bad(seq4[nb] >> 4)) return true;
    return false;
}
#if defined(__x86_64__)
// the same, sixteen qualities / thirty-two base codes at a time (SSSE3: every x86-64 of the last fifteen years; checked once).  A row's
// last block is loaded again from its end (the test is an OR over the bytes: scanning some twice changes nothing).  Ten million
// 150-base reads: 0.18 s of the staging pool with the word loop above, a quarter of it with this one.
__attribute__((target("ssse3"))) static bool read_has_escape_ssse3(const uint8_t* qual, const uint8_t* seq4, int32_t L) {
    if (L < 32) return read_has_escape_words(qual, seq4, L);
    const __m128i one = _mm_set1_epi8(1), lim = _mm_set1_epi8((char)(EB_ESC - 1)), zero = _mm_setzero_si128(), m0f = _mm_set1_epi8(0x0f);
    const __m128i lut = _mm_setr_epi8(0, 1, 1, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0);             // 1: the code of A, C, G or T
    __m128i bad = zero;
#define BRC_Q16(p) { const __m128i t_ = _mm_sub_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(p)), one); bad = _mm_or_si128(bad, _mm_cmpeq_epi8(_mm_max_epu8(t_, lim), t_)); }   /* (q - 1) mod 256 >= 62: q == 0 or q >= 63 */
#define BRC_S16(p) { const __m128i x_ = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p)); \
                     const __m128i ok_ = _mm_and_si128(_mm_shuffle_epi8(lut, _mm_and_si128(x_, m0f)), _mm_shuffle_epi8(lut, _mm_and_si128(_mm_srli_epi16(x_, 4), m0f))); \
                     bad = _mm_or_si128(bad, _mm_cmpeq_epi8(ok_, zero)); }
    int32_t j = 0;
    for (; j + 16 <= L; j += 16) BRC_Q16(qual + j)
    if (j < L) BRC_Q16(qual + L - 16)
    if (_mm_movemask_epi8(bad)) return true;
    const int32_t nb = L / 2;                                                                        // whole bytes: two base codes each (nb >= 16)
    int32_t t = 0;
    for (; t + 16 <= nb; t += 16) BRC_S16(seq4 + t)
    if (t < nb) BRC_S16(seq4 + nb - 16)
    if (_mm_movemask_epi8(bad)) return true;
#undef BRC_Q16
#undef BRC_S16
    if (L & 1) { const uint32_t n = seq4[nb] >> 4; if (n == 0u || (n & (n - 1u)) != 0u) return true; }
    return false;
}
static bool read_has_escape(const uint8_t* qual, const uint8_t* seq4, int32_t L) {
    static const bool ssse3 = __builtin_cpu_supports("ssse3") != 0;
    return ssse3 ? read_has_escape_ssse3(qual, seq4, L) : read_has_escape_words(qual, seq4, L);
}
#else
static bool read_has_escape(const uint8_t* qual, const uint8_t* seq4, int32_t L) { return read_has_escape_words(qual, seq4, L); }
#endif

// two decimal digits at a time
static const char kD2[201] =
    "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869"
    "707172737475767778798081828384858687888990919293949596979899";

static inline int fmt_u64(char* out, uint64_t v) {
    char t[24]; int n = 0;
    while (v >= 100) { const unsigned d = (unsigned)(v % 100); v /= 100; t[n++] = kD2[2 * d + 1]; t[n++] = kD2[2 * d]; }
    if (v >= 10) { t[n++] = kD2[2 * v + 1]; t[n++] = kD2[2 * v]; } else t[n++] = (char)('0' + v);
    for (int i = 0; i < n; ++i) out[i] = t[n - 1 - i];
    return n;
}
int fmt_u32(char* out, uint32_t v) {
    if (v < 10) { out[0] = (char)('0' + v); return 1; }
    if (v < 100) { out[0] = kD2[2 * v]; out[1] = kD2[2 * v + 1]; return 2; }
    return fmt_u64(out, v);
}

// "%.2f" of the exact binary value, round-half-even: v*100 is exact in double (24+7 significant bits), so rounding it
// to an integer in the current (to-nearest-even) mode gives the correctly rounded count of hundredths glibc's printf
// would print; adding and subtracting 2^52 does that rounding without a libm call.
int fmt_f2(char* out, float v) {
    if (!(fabsf(v) < 4.0e13f)) return snprintf(out, 64, "%.2f", (double)v);   // inf / nan / huge: libc path
    const double s = fabs((double)v * 100.0);
    volatile double big = s + 4503599627370496.0;       // (volatile: the sum must be rounded to double before the subtraction)
    const uint64_t u = (uint64_t)(big - 4503599627370496.0);
    int n = 0;
    if (signbit(v)) out[n++] = '-';
    const uint64_t ip = u / 100; const unsigned fr = (unsigned)(u % 100);
    n += ip < 100 ? fmt_u32(out + n, (uint32_t)ip) : fmt_u64(out + n, ip);
    out[n++] = '.'; out[n++] = kD2[2 * fr]; out[n++] = kD2[2 * fr + 1];
    return n;
}

void expand_slots(const HostPlanes& hp, int Lp, int64_t P, int64_t PS, uint32_t* istat, float* fstat) {
    const int64_t CH = 1 << 16;
    const int64_t nch = (PS + CH - 1) / CH;
    std::atomic<int64_t> next(0);
    auto work = [&]() {
        for (;;) {
            const int64_t ci = next.fetch_add(1);
            if (ci >= nch) break;
            const int64_t k0 = ci * CH, k1 = std::min(PS, k0 + CH);
            for (int l = 0; l < Lp; ++l) {
                for (int b = 0; b < NBUCKET; ++b) {
                    for (int f = 0; f < NI; ++f) memset(istat + (((int64_t)l * NBUCKET + b) * NI + f) * PS + k0, 0, (size_t)(k1 - k0) * 4);
                    for (int f = 0; f < NF; ++f) memset(fstat + (((int64_t)l * NBUCKET + b) * NF + f) * PS + k0, 0, (size_t)(k1 - k0) * 4);
                }
                const uint32_t* sid = hp.slotid + (int64_t)l * PS;
                for (int sl = 0; sl < 2; ++sl) {
                    for (int f = 0; f < NI; ++f) {
                        const uint32_t* src = hp.si + (((int64_t)l * 2 + sl) * NI + f) * PS;
                        for (int64_t k = k0; k < std::min(P, k1); ++k) {
                            const uint32_t b = (sid[k] >> (8 * sl)) & 0xffu;
                            if (b < (uint32_t)NBUCKET && src[k]) istat[(((int64_t)l * NBUCKET + b) * NI + f) * PS + k] = src[k];
                        }
                    }
                    for (int f = 0; f < NF; ++f) {
                        const float* src = hp.sf + (((int64_t)l * 2 + sl) * NF + f) * PS;
                        for (int64_t k = k0; k < std::min(P, k1); ++k) {
                            const uint32_t b = (sid[k] >> (8 * sl)) & 0xffu;
                            if (b < (uint32_t)NBUCKET) fstat[(((int64_t)l * NBUCKET + b) * NF + f) * PS + k] = src[k];
                        }
                    }
                }
            }
        }
    };
    unsigned nt = effective_cpus(); if (nt > 32) nt = 32; if (nt < 1) nt = 1;
    if (nch < (int64_t)nt) nt = (unsigned)nch;
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    // third-allele buckets: their sums were folded on the device (brc_core.h: fold_xev_bucket); a bucket's sums sit in ONE place — the
    // slot that names it, or this table (also for a bucket a slot names: the events of an N / '=' base never enter the slots)
    for (uint64_t i = 0; i < hp.n_xagg; ++i) {
        const XAgg& a = hp.xagg[i];
        const int64_t l = a.lib_b >> 8; const uint32_t b = a.lib_b & 0xffu; const int64_t k = a.k;
        if (l >= Lp || b >= (uint32_t)NBUCKET || k >= P) continue;
        uint32_t* ip = istat + ((l * NBUCKET + b) * NI) * PS + k; float* fp = fstat + ((l * NBUCKET + b) * NF) * PS + k;
        for (int f = 0; f < NI; ++f) ip[(int64_t)f * PS] = a.i[f];
        for (int f = 0; f < NF; ++f) fp[(int64_t)f * PS] = a.f[f];
    }
}

static const char kZeroStat[] = "0:0.00:0.00:0.00:0:0:0.00:0.00:0.00:0:0.00:0.00:0.00";
enum { ZERO_LEN = sizeof(kZeroStat) - 1, STAT_MAX = 13 * 48 };     // (a float field can be as long as "%.2f" of FLT_MAX)

// operator<<(ostream&, BasicStat) (BasicStat.cpp:110-159); writes at w (room for STAT_MAX bytes), returns the new end
static inline char* fmt_stat(char* w, const uint32_t* si, const float* sf, bool is_indel) {
    const uint32_t n = si[I_N];
    if (n == 0) { memcpy(w, kZeroStat, ZERO_LEN); return w + ZERO_LEN; }
    const float c = (float)n;
    w += fmt_u32(w, n); *w++ = ':';
    w += fmt_f2(w, (float)si[I_SMQ] / c); *w++ = ':';
    if (is_indel) { memcpy(w, "0.00", 4); w += 4; } else w += fmt_f2(w, (float)si[I_SBQ] / c);
    *w++ = ':';
    w += fmt_f2(w, (float)si[I_SSE] / c); *w++ = ':';
    w += fmt_u32(w, si[I_PLUS]); *w++ = ':';
    w += fmt_u32(w, si[I_MINUS]); *w++ = ':';
    w += fmt_f2(w, sf[F_SEV] / c); *w++ = ':';
    w += fmt_f2(w, sf[F_SNM] / c); *w++ = ':';
    w += fmt_f2(w, (float)si[I_SMMQ] / c); *w++ = ':';
    w += fmt_u32(w, si[I_NQ2]); *w++ = ':';
    if (si[I_NQ2] > 0) w += fmt_f2(w, sf[F_SQ2] / (float)si[I_NQ2]); else { memcpy(w, "0.00", 4); w += 4; }
    *w++ = ':';
    w += fmt_f2(w, (float)si[I_SCLIP] / c); *w++ = ':';
    w += fmt_f2(w, sf[F_S3P] / c);
    return w;
}

// grow-only text buffer the formatter writes into through raw pointers
struct TextBuf {
    char* p = nullptr; size_t n = 0, cap = 0;
    TextBuf() {}
    TextBuf(const TextBuf&) = delete;
    TextBuf& operator=(const TextBuf&) = delete;
    TextBuf(TextBuf&& o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
    ~TextBuf() { free(p); }
    void clear() { n = 0; }
    // pointer to the end of the text with at least k bytes of room behind it; NULL when memory runs out
    char* room(size_t k) {
        if (n + k > cap) {
            size_t c = cap + cap / 2 + (1u << 16);
            if (c < n + k) c = n + k + (n + k) / 4;
            char* q = (char*)realloc(p, c);
            if (!q) return nullptr;
            p = q; cap = c;
        }
        return p + n;
    }
};

}  // namespace brc

using namespace brc;

struct QEnt { uint32_t tid, pos; brc_stat st; std::string allele; };
struct XKey { uint64_t key; brc_stat st; };      // a third-allele bucket for the host formatter's merge: key = k << 16 | library << 8 | bucket

struct brc_engine {
    brc_config cfg;
    std::vector<std::string> libs;
    Backend* be = nullptr;
    Staged st;
    Geometry g;
    int state = 0;   // 0 idle, 1 region open, 2 uploaded, 3 computed, 4 fetched
    unsigned format_threads = 0;   // BRC_OPT_FORMAT_THREADS (0: effective_cpus())
    Pool* pool = nullptr;          // the staging stage's threads (created by the first large batch)
    std::string err;
    // bam_plp_push max-count emulation
    int64_t accepted = 0, n_ext = 0; int32_t last_acc_pos = 0; int32_t last_pos = 0;
    bool heap_built = false;
    std::priority_queue<int32_t, std::vector<int32_t>, std::greater<int32_t> > live_ends;
    // fetched result: compact planes from the backend, expanded to the ABI's dense planes
    HostPlanes hp;
    uint32_t* dense_i = nullptr; float* dense_f = nullptr; size_t dense_cap = 0;
    std::vector<brc_indel> indels;
    std::string alleles;
    std::vector<char> refbase;
    // brc_fetch_window: the dense result of one window of the computed region (buffers of its own: a fetched whole-region result stays valid)
    uint32_t* win_i = nullptr; float* win_f = nullptr; size_t win_cap = 0;
    std::vector<brc_indel> win_indels; std::string win_alleles; std::vector<char> win_refbase; std::vector<XAgg> win_xagg; std::vector<IndelOut> win_iout;
    // BRC_OPT_TEXT_ONLY: the caller only formats (brc_format_region / brc_format_window): no dense planes are built, the
    // formatter reads the compact slot planes; third-allele events are aggregated into a sparse (position, library,
    // bucket)-sorted table instead
    bool text_only = false;
    bool continues = false, warn_skip_lead = false; // BRC_OPT_CONTINUES_PREVIOUS (1: both; 2: the warnings only)
    bool device_text = false; std::string chrom;   // BRC_OPT_DEVICE_TEXT / brc_set_chrom
    bool text_result = false;                       // the last fetched result is device text (no planes on the host)
    bool text_computed = false; int text_slot_computed = 0, text_slot = 0;   // device text started by the last brc_compute / fetched
    // a text result and the host's deletion queues (prepare_text_queues / format_device_text): per library the deletion buckets at the
    // library's last position that has any, as the queue entries they would be; and — only when the queues were not empty when the
    // result was fetched — every deletion bucket of the region, sorted by (position, library, length)
    struct DelEnt { int32_t pos, lib, len; brc_stat st; std::string allele; };
    std::vector<std::vector<QEnt> > tail_dels; std::vector<int64_t> tail_pos;
    std::vector<DelEnt> dels; bool have_dels = false;
    TextBuf pbuf;                                   // the lines the host rewrote
    struct Patch { int64_t k; size_t off, len; };
    std::vector<Patch> patches;
    std::vector<XKey> xagg;
    size_t hint_reads = 0, hint_bases = 0;          // BRC_OPT_EXPECT_*: staging is sized once instead of grown batch by batch
    // formatter state: the text of the last call (one contiguous buffer, capacity kept across calls), the per-chunk
    // buffers the threads format into (kept too: a fresh 300-MB buffer per piece costs more in page faults than the text)
    char* tbuf = nullptr; size_t tcap = 0, tlen = 0;
    std::vector<TextBuf> fparts;
    std::vector<const char*> part_ptr; std::vector<size_t> part_len;
    TextBuf wbuf;                                   // brc_format_window's text
    std::string wev, wtext;
    std::vector<std::deque<QEnt> > queue;
    // host-side phase timers (BRC_ENGINE_TIMING=1: printed by brc_destroy)
    double t_push = 0, t_upload = 0, t_compute = 0, t_d2h = 0, t_post = 0, t_format = 0; int64_t n_regions = 0;
    uint64_t n_xev_total = 0, n_indel_total = 0; double t_textwait = 0;
};

static int fail(brc_engine* e, int code, const char* msg) { e->err = msg; return code; }

// HBuf::append for the two arenas that carry most of a batch's bytes (QUAL, SEQ): large copies into the pinned staging are
// cut into slices for the staging stage's threads (one thread moves ~10 GB/s; a 1-Mbp piece at 30x brings 45 MB)
template <class T>
static bool append_big(Pool* pool, HBuf<T>& h, const T* src, size_t k) {
    if (!pool || k < ((size_t)4 << 20)) return h.append(src, k);
    if (!h.reserve(h.n + k + 16)) return false;
    T* dst = h.p + h.n;
    const size_t SL = (size_t)1 << 20; const int64_t ns = (int64_t)((k + SL - 1) / SL);
    pool->run(ns, [&](int64_t i) { const size_t a = (size_t)i * SL, bb = std::min(k, a + SL); memcpy(dst + a, src + a, (bb - a) * sizeof(T)); });
    h.n += k;
    return true;
}

// allele text + std::map<std::string,BasicStat> iteration order (bamreadcount.cpp:323-342, 389-401): the device's reduced indel
// buckets -> the ABI's list, sorted by (position, library, allele text bytewise)
static void assemble_indels(const brc_engine* e, const IndelOut* list, int64_t n, std::vector<brc_indel>& out, std::string& alleles) {
    const Geometry& g = e->g; const Staged& s = e->st;
    std::vector<std::string> txt((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const IndelOut& o = list[i];
        std::string& a = txt[(size_t)i];
        if (o.len > 0) {
            a.push_back('+');
            const uint8_t* seq = s.seq_at(s.seq_off.p[o.rep_read]);
            const int32_t L = s.l_qseq.p[o.rep_read];
            for (int j = 0; j < o.len; ++j) { const int q = o.rep_qpos + 1 + j; a.push_back(q < L ? "=ACGTN"[canon_bucket(seqi(seq, q))] : 'N'); }
        } else {
            a.push_back('-');
            for (int j = 0; j < -o.len; ++j) { const int64_t p = (int64_t)o.pos + 1 + j; a.push_back((g.ref && p < g.ref_len && g.ref[p]) ? g.ref[p] : 'N'); }
        }
    }
    std::vector<uint32_t> ord((size_t)n);
    for (size_t i = 0; i < ord.size(); ++i) ord[i] = (uint32_t)i;
    std::sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) {
        const IndelOut& a = list[x]; const IndelOut& b = list[y];
        if (a.pos != b.pos) return a.pos < b.pos;
        if (a.lib != b.lib) return a.lib < b.lib;
        return txt[x] < txt[y];
    });
    out.resize(ord.size()); alleles.clear();
    for (size_t i = 0; i < ord.size(); ++i) {
        const IndelOut& o = list[ord[i]]; brc_indel& d = out[i];
        d.pos = o.pos; d.lib = o.lib; d.len = o.len; d.rep_read = o.rep_read; d.rep_qpos = o.rep_qpos;
        d.allele_off = (uint32_t)alleles.size(); d.allele_len = (uint32_t)txt[ord[i]].size();
        alleles += txt[ord[i]];
        for (int f = 0; f < BRC_NI; ++f) d.stat.i[f] = o.i[f];
        for (int f = 0; f < BRC_NF; ++f) d.stat.f[f] = o.f[f];
    }
}

// A text result (BRC_OPT_DEVICE_TEXT): the device wrote the indel entries of every line itself, assuming deletion queues that hold nothing but
// what the position before queued.  What is left for the host is the queues' own life ACROSS regions (the reference never clears them
// between command-line regions, bamreadcount.cpp:641-657): which deletions this region leaves pending — those queued at the last position a
// library was processed at (IndelQueue.cpp:3-15: an entry lives until the library's next processed position) — and, when the region did
// not start from clean queues, the true entries of its lines (format_device_text rewrites them).  Runs inside brc_fetch_result: the
// reference bases (deletion alleles are the reference's characters, :331-338) are the caller's only until that call returns, and the
// region before has been formatted by then (include/brc.h, threads), so the queues are what this region will find.
static std::string deletion_allele(const Geometry& g, const IndelOut& o) {
    std::string a; a.push_back('-');
    for (int j = 0; j < -o.len; ++j) { const int64_t p = (int64_t)o.pos + 1 + j; a.push_back((g.ref && p < g.ref_len && g.ref[p]) ? g.ref[p] : 'N'); }
    return a;
}
static void prepare_text_queues(brc_engine* e) {
    const Geometry& g = e->g; const HostPlanes& hp = e->hp; const int Lp = g.Lp;
    e->tail_dels.assign((size_t)Lp, std::vector<QEnt>()); e->tail_pos.assign((size_t)Lp, INT64_MIN);
    for (int64_t i = 0; i < hp.n_indel; ++i) { const IndelOut& o = hp.indel[i]; if (o.len < 0 && o.lib >= 0 && o.lib < Lp && (int64_t)o.pos > e->tail_pos[(size_t)o.lib]) e->tail_pos[(size_t)o.lib] = o.pos; }
    auto stat_of = [](const IndelOut& o) { brc_stat st; for (int f = 0; f < BRC_NI; ++f) st.i[f] = o.i[f]; for (int f = 0; f < BRC_NF; ++f) st.f[f] = o.f[f]; return st; };
    for (int64_t i = 0; i < hp.n_indel; ++i) {
        const IndelOut& o = hp.indel[i];
        if (o.len >= 0 || o.lib < 0 || o.lib >= Lp || (int64_t)o.pos != e->tail_pos[(size_t)o.lib]) continue;
        QEnt q; q.tid = (uint32_t)g.tid; q.pos = (uint32_t)o.pos + 1u; q.st = stat_of(o); q.allele = deletion_allele(g, o);
        e->tail_dels[(size_t)o.lib].push_back(q);
    }
    // (deletions at one position are prefixes of one another: allele order = by length, what std::map<std::string, ...> gives, :389)
    for (std::vector<QEnt>& v : e->tail_dels) std::sort(v.begin(), v.end(), [](const QEnt& a, const QEnt& b) { return a.allele.size() < b.allele.size(); });
    bool busy = false; for (const std::deque<QEnt>& q : e->queue) if (!q.empty()) busy = true;
    e->dels.clear(); e->have_dels = busy;
    if (busy) {
        for (int64_t i = 0; i < hp.n_indel; ++i) {
            const IndelOut& o = hp.indel[i];
            if (o.len >= 0) continue;
            brc_engine::DelEnt d; d.pos = o.pos; d.lib = o.lib; d.len = o.len; d.st = stat_of(o); d.allele = deletion_allele(g, o);
            e->dels.push_back(std::move(d));
        }
        std::sort(e->dels.begin(), e->dels.end(), [](const brc_engine::DelEnt& a, const brc_engine::DelEnt& b) { return a.pos != b.pos ? a.pos < b.pos : (a.lib != b.lib ? a.lib < b.lib : a.len > b.len); });
    }
}

extern "C" {

const char* brc_strerror(int code) {
    switch (code) {
        case BRC_OK: return "ok";
        case BRC_E_ARG: return "bad argument or call order";
        case BRC_E_NODEVICE: return "no HIP device / kernels unavailable (the engine has no CPU fallback)";
        case BRC_E_HIP: return "HIP runtime error";
        case BRC_E_NOMEM: return "out of memory";
        case BRC_E_LIMIT: return "engine limit exceeded";
        default: return "unknown error";
    }
}
const char* brc_last_error(const brc_engine* e) { return e ? e->err.c_str() : ""; }
const char* brc_kernel_name(int k) { return (k >= 0 && k < BRC_NKERNEL) ? backend_kernel_name(k) : NULL; }
const char* brc_engine_kind(void) { return backend_kind(); }

int brc_create(const brc_config* cfg, brc_engine** out) {
    if (!cfg || !out || cfg->abi_version != BRC_ABI_VERSION) return BRC_E_ARG;
    if (cfg->per_lib && (cfg->n_libs < 0 || (cfg->n_libs > 0 && !cfg->lib_names) || cfg->n_libs > 254)) return BRC_E_ARG;   // library index + 1 travels in 8 bits of the device read record
    // brc.h asks for bytewise-sorted, distinct names: the reference keeps its libraries in a std::map keyed by name (:273) and prints
    // them in that order (:360); names in another order would print in another order — refused instead of printed differently
    if (cfg->per_lib) for (int i = 0; i < cfg->n_libs; ++i) {
        if (!cfg->lib_names[i]) return BRC_E_ARG;
        if (i && strcmp(cfg->lib_names[i - 1], cfg->lib_names[i]) >= 0) return BRC_E_ARG;
    }
    brc_engine* e = new (std::nothrow) brc_engine();
    if (!e) return BRC_E_NOMEM;
    e->cfg = *cfg;
    if (e->cfg.max_cnt <= 0) e->cfg.max_cnt = 10000000;
    if (cfg->per_lib) for (int i = 0; i < cfg->n_libs; ++i) e->libs.push_back(cfg->lib_names[i]);
    e->cfg.lib_names = NULL;
    e->g.Lp = cfg->per_lib ? (cfg->n_libs > 0 ? cfg->n_libs : 1) : 1;
    int err = BRC_OK;
    e->be = make_backend(e->cfg, &err);
    if (!e->be) { delete e; return err ? err : BRC_E_NODEVICE; }
    e->st.init(e->be->host_alloc());
    e->queue.resize((size_t)e->g.Lp);
    *out = e;
    return BRC_OK;
}

void brc_destroy(brc_engine* e) {
    if (!e) return;
    if (getenv("BRC_ENGINE_TIMING"))
        fprintf(stderr, "engine timing (%lld regions): push %.3f s, upload %.3f s, compute %.3f s, download %.3f s, assemble %.3f s, format %.3f s (of which waiting for the device text %.3f s); third-allele events %llu, indel buckets %llu\n",
                (long long)e->n_regions, e->t_push, e->t_upload, e->t_compute, e->t_d2h, e->t_post, e->t_format, e->t_textwait, (unsigned long long)e->n_xev_total, (unsigned long long)e->n_indel_total);
    e->st.destroy();
    delete e->be;
    free(e->dense_i); free(e->dense_f); free(e->tbuf); free(e->win_i); free(e->win_f);
    delete e->pool;
    delete e;
}

int brc_set_chrom(brc_engine* e, const char* chrom) {
    if (!e || !chrom) return BRC_E_ARG;
    e->chrom = chrom;
    return BRC_OK;
}

int brc_set_option(brc_engine* e, int option, int64_t value) {
    if (!e) return BRC_E_ARG;
    switch (option) {
        case BRC_OPT_TEXT_ONLY: e->text_only = value != 0; return BRC_OK;
        // (hints size the staging buffers ahead of the pushes: a wild value must not become a wild allocation)
        case BRC_OPT_EXPECT_READS: e->hint_reads = value > 0 ? (size_t)std::min<int64_t>(value, (int64_t)1 << 31) : 0; return BRC_OK;
        case BRC_OPT_EXPECT_BASES: e->hint_bases = value > 0 ? (size_t)std::min<int64_t>(value, (int64_t)1 << 36) : 0; return BRC_OK;
        case BRC_OPT_DEVICE_TEXT: e->device_text = value != 0; return BRC_OK;
        case BRC_OPT_CONTINUES_PREVIOUS: e->continues = value == 1; e->warn_skip_lead = value == 1 || value == 2; return BRC_OK;
        case BRC_OPT_FORMAT_THREADS: e->format_threads = value > 0 ? (unsigned)std::min<int64_t>(value, 1024) : 0u; return BRC_OK;
        case BRC_OPT_MAX_COUNT: if (value < INT32_MIN || value > INT32_MAX) return fail(e, BRC_E_ARG, "max count out of range"); e->cfg.max_cnt = (int32_t)value; return BRC_OK;
        case BRC_OPT_EXPECT_TEXT: { if (value <= 0) return BRC_OK; const int rc = e->be->reserve_text((size_t)value); return rc ? fail(e, rc, e->be->last_error()) : BRC_OK; }
        default: return fail(e, BRC_E_ARG, "unknown engine option");
    }
}

int brc_begin_region(brc_engine* e, int32_t tid, int32_t beg0, int32_t end, const char* ref, int64_t ref_len) {
    if (!e) return BRC_E_ARG;
    if (beg0 < 0 || end < beg0 || (ref && ref_len < 0)) return fail(e, BRC_E_ARG, "bad region");
    e->st.clear();
    e->g.tid = tid; e->g.beg0 = beg0; e->g.end = end; e->g.ref = ref; e->g.ref_len = ref ? ref_len : 0;
    e->g.P = 0; e->g.pos0 = 0;
    e->accepted = 0; e->n_ext = 0; e->last_acc_pos = 0; e->last_pos = INT32_MIN; e->heap_built = false;
    while (!e->live_ends.empty()) e->live_ends.pop();
    e->state = 1;
    return BRC_OK;
}

// reference length of a CIGAR; -1 when it does not fit 31 bits (a damaged record: positions are int32 everywhere, as in the reference)
static inline int32_t cigar_rlen(const uint32_t* cig, uint32_t nc, uint64_t* n_indel_ops) {
    int64_t l = 0;
    for (uint32_t k = 0; k < nc; ++k) {
        const uint32_t op = cig[k] & 0xfu;
        if (is_refop(op)) l += (int64_t)(cig[k] >> 4);
        if (op == CINS || op == CDEL || op == CPAD) ++*n_indel_ops;
    }
    return l > INT32_MAX ? -1 : (int32_t)l;
}

static int push_reads_staged(brc_engine* e, const brc_read_batch* b, bool* touched, bool pinned);

// A refused batch leaves the staging arrays half appended (per-read arrays and arenas grow before a record is found bad),
// so the region cannot take further batches: it is abandoned — every later push / upload on it fails with "outside an open
// region" until the caller opens the next one with brc_begin_region (which resets the staging).
void* brc_host_alloc(size_t bytes) { return backend_host_alloc(bytes); }
void brc_host_free(void* p) { if (p) backend_host_free(p); }
// A mapped read with an M / = / X operator of length zero: htslib's resolve_cigar2 steps ONTO such an operator without asking whether the
// position lies inside it (it reports the column as a match at the operator's query offset and the deletion behind it one column late).
// Round 6: such reads are piled up by the cursor itself (brc_core.h: cursor_resolve — walk_pieces / enumerate_indels switch to it).  What
// stays refused is the one case in which the reference reads memory behind the read: an empty operator reported at the query offset
// l_qseq (every base of the read already consumed before it).
static const char* const kEmptyM = "a mapped read has an empty M/=/X CIGAR operator behind its last base (the reference would read past the read's qualities)";
static int push_reads_any(brc_engine* e, const brc_read_batch* b, bool pinned);
int brc_push_reads(brc_engine* e, const brc_read_batch* b) { return push_reads_any(e, b, false); }
int brc_push_reads_pinned(brc_engine* e, const brc_read_batch* b) { return push_reads_any(e, b, true); }
static int push_reads_any(brc_engine* e, const brc_read_batch* b, bool pinned) {
    if (!e || !b) return BRC_E_ARG;
    bool touched = false;
    int rc;
    // no exception crosses the C boundary: the parallel staging path allocates (per-chunk tables, the job's std::function), and a C
    // caller would see std::terminate
    try { rc = push_reads_staged(e, b, &touched, pinned); }
    catch (const std::bad_alloc&) { touched = true; rc = fail(e, BRC_E_NOMEM, "host allocation failed while staging reads"); }
    catch (const std::exception& ex) { touched = true; rc = fail(e, BRC_E_NOMEM, ex.what()); }
    catch (...) { touched = true; rc = fail(e, BRC_E_NOMEM, "unexpected failure while staging reads"); }
    if (rc != BRC_OK && touched) e->state = 0;
    return rc;
}

static int push_reads_staged(brc_engine* e, const brc_read_batch* b, bool* touched, bool pinned) {
    if (e->state != 1) return fail(e, BRC_E_ARG, "brc_push_reads outside an open region");
    if (b->n_reads < 0) return fail(e, BRC_E_ARG, "negative n_reads");
    if (e->cfg.per_lib && !b->lib && b->n_reads) return fail(e, BRC_E_ARG, "per-library mode needs brc_read_batch.lib");
    Staged& s = e->st;
    const double t_in = now_s();
    struct Tm { brc_engine* e; double t0; ~Tm() { e->t_push += now_s() - t0; } } tm_{e, t_in};
    const size_t n = (size_t)b->n_reads, n0 = (size_t)s.n;
    if ((uint64_t)s.n + n >= 0xFFFFFFF0ull || s.cigar.n + b->n_cigar_total >= 0xFFFFFFF0ull)
        return fail(e, BRC_E_LIMIT, "more than 2^32 reads or CIGAR operators in one region: split the region");
    // adopted arenas (brc_push_reads_pinned on a backend that uploads them in place): the region's first push decides for the region
    const bool adopt = pinned && e->be->adopts_arenas();
    if (s.n > 0 && adopt != s.adopted() && (s.seq_total || s.qual_total)) return fail(e, BRC_E_ARG, "a region takes brc_push_reads or brc_push_reads_pinned, not both");
    const uint64_t cb = s.cigar.n, sb = s.seq_total, qb = s.qual_total;
    if (e->hint_reads > n0 + n || e->hint_bases > qb + b->qual_bytes) {
        const size_t hr = std::max(e->hint_reads, n0 + n) + 16, hb = std::max<size_t>(e->hint_bases, qb + b->qual_bytes) + 16;
        bool okh = s.pos.reserve(hr) && s.flag.reserve(hr) && s.mapq.reserve(hr) && s.l_qseq.reserve(hr) && s.n_cigar.reserve(hr) && s.cig_off.reserve(hr) &&
                   s.seq_off.reserve(hr) && s.qual_off.reserve(hr) && s.bq_row.reserve(hr) && s.wide.reserve(hr) && s.piece_cnt.reserve(hr) && s.piece_off.reserve(hr) && s.iev_off.reserve(hr) && s.lib.reserve(hr) &&
                   s.nm.reserve(hr) && s.sm.reserve(hr) && s.tags.reserve(hr) && s.qname_off.reserve(hr) && s.cigar.reserve(hr + hr / 4) &&
                   (adopt || (s.qual.reserve(hb) && s.seq4.reserve(hb / 2 + hr)));
        if (!okh) return fail(e, BRC_E_NOMEM, "host staging allocation failed");
    }
    // the staging stage's threads (a share of the CPUs the process may use; BRC_OPT_FORMAT_THREADS caps it like the formatter's)
    unsigned pthr = effective_cpus(); if (pthr > 8) pthr = 8;
    if (e->format_threads && e->format_threads < pthr) pthr = e->format_threads;
    const bool par = n >= 8192 && pthr > 1;      // (a stripe of a 1-Mbp piece is 17 000 reads: with the arenas adopted, the per-read pass is what is left of a push)
    if (par && !e->pool) e->pool = new (std::nothrow) Pool(pthr - 1);
    Pool* const pool = par ? e->pool : nullptr;
    *touched = true;
    bool ok = s.pos.append(b->pos, n) && s.flag.append(b->flag, n) && s.mapq.append(b->mapq, n) && s.l_qseq.append(b->l_qseq, n) &&
              s.n_cigar.append(b->n_cigar, n) && s.cig_off.append(b->cigar_off, n) && s.seq_off.append(b->seq_off, n) &&
              s.qual_off.append(b->qual_off, n) && s.cigar.append(b->cigar, b->n_cigar_total) &&
              (adopt || (append_big(pool, s.seq4, b->seq4, b->seq_bytes) && append_big(pool, s.qual, b->qual, b->qual_bytes))) &&
              s.bq_row.reserve(n0 + n + 16) && s.wide.reserve(n0 + n + 16) && s.piece_cnt.reserve(n0 + n + 16) && s.piece_off.reserve(n0 + n + 16) && s.iev_off.reserve(n0 + n + 16) && s.lib.reserve(n0 + n + 16) && s.nm.reserve(n0 + n + 16) && s.sm.reserve(n0 + n + 16) && s.tags.reserve(n0 + n + 16);
    if (!ok) return fail(e, BRC_E_NOMEM, "host staging allocation failed");
    { uint32_t mx = s.max_ncigar; for (size_t i = 0; i < n; ++i) mx = b->n_cigar[i] > mx ? b->n_cigar[i] : mx; s.max_ncigar = mx; }   // (the engine's wave-form annotator is for reads with five operators and more)
    if (adopt) {
        if (b->seq_bytes) { Staged::Seg g; g.p = b->seq4; g.off = sb; g.n = b->seq_bytes; s.seq_seg.push_back(g); }
        if (b->qual_bytes) { Staged::Seg g; g.p = b->qual; g.off = qb; g.n = b->qual_bytes; s.qual_seg.push_back(g); }
    }
    s.seq_total = sb + b->seq_bytes; s.qual_total = qb + b->qual_bytes;
    if (!s.qname_off.reserve(n0 + n + 16)) return fail(e, BRC_E_NOMEM, "host staging allocation failed");
    for (size_t i = 0; i < n; ++i) {
        uint64_t off = ~0ull;
        if (b->qname && b->qname[i]) { off = s.qnames.n; if (!s.qnames.append(b->qname[i], strlen(b->qname[i]) + 1)) return fail(e, BRC_E_NOMEM, "host staging allocation failed"); }
        s.qname_off.p[n0 + i] = off;
    }
    s.qname_off.n = n0 + n;
    for (size_t i = 0; i < n; ++i) {
        s.lib.p[n0 + i] = (e->cfg.per_lib && b->lib) ? b->lib[i] : 0;
        s.nm.p[n0 + i] = b->nm ? b->nm[i] : 0;
        s.sm.p[n0 + i] = b->sm ? b->sm[i] : 0;
        s.tags.p[n0 + i] = b->tags ? b->tags[i] : 0;
    }
    s.lib.n = s.nm.n = s.sm.n = s.tags.n = s.bq_row.n = s.wide.n = s.piece_cnt.n = s.iev_off.n = n0 + n;
    const int32_t maxcnt = e->cfg.max_cnt;
    // ---- large batches: the per-read pass on several threads.  Everything a read contributes to a running quantity — its row
    // in the event-byte stream, its slots in the raw indel list, its pieces, the region's extent, the length histogram — is a
    // prefix sum or a reduction: every chunk of reads computes its own, the chunks' totals are scanned, a second pass turns
    // the per-read lengths into offsets.  The max-count rule (bam_plp_push drops reads that start where the previous one did
    // once more than -d are buffered) needs its serial state only when the count can be reached at all; the first bad record
    // in file order decides the error, as in the serial pass.  (One thread staged 6 M reads of a 30-Mbp piece list in 0.15-0.35 s:
    // the longest stage of the command line's engine thread.)
    if (pool && e->accepted + (int64_t)n < (int64_t)maxcnt && !e->heap_built) {
        struct Chunk {
            uint64_t bq = 0, idp = 0, np = 0; int32_t max_lq = 0; int64_t max_span = 0; int64_t min_pos = INT64_MAX, max_end = INT64_MIN, n_ext = 0, acc = 0;
            int32_t last_acc_pos = 0; int err = 0; const char* msg = nullptr; size_t err_at = 0; uint32_t hist[TABLE_MAX + 1]; bool empty_m = false, eqx = false;
        };
        const size_t CH = (n + (size_t)pool->size() * 4 - 1) / ((size_t)pool->size() * 4);
        const size_t nch = (n + CH - 1) / CH;
        std::vector<Chunk> ch(nch);
        pool->run((int64_t)nch, [&](int64_t ci) {
            Chunk& C = ch[(size_t)ci]; memset(C.hist, 0, sizeof C.hist);
            const size_t i0 = (size_t)ci * CH, i1 = std::min(n, i0 + CH);
            auto bad = [&](size_t i, int code, const char* m) { if (!C.err) { C.err = code; C.msg = m; C.err_at = i; } };
            for (size_t i = i0; i < i1 && !C.err; ++i) {
                const size_t r = n0 + i;
                uint32_t nc = s.n_cigar.p[r];
                if (b->cigar_off[i] + nc > b->n_cigar_total || b->qual_off[i] + (uint64_t)(s.l_qseq.p[r] > 0 ? s.l_qseq.p[r] : 0) > b->qual_bytes ||
                    b->seq_off[i] + (uint64_t)((s.l_qseq.p[r] + 1) / 2) > b->seq_bytes || s.l_qseq.p[r] < 0) { bad(i, BRC_E_ARG, "read offsets outside the batch arenas"); break; }
                if (e->cfg.per_lib && s.lib.p[r] >= e->g.Lp) { bad(i, BRC_E_ARG, "library index out of range"); break; }
                s.cig_off.p[r] += cb; s.seq_off.p[r] += sb; s.qual_off.p[r] += qb;
                const uint64_t rowlen = ((uint64_t)s.l_qseq.p[r] + 15u) & ~(uint64_t)15u;
                s.bq_row.p[r] = rowlen; C.bq += rowlen;                                  // (length now, offset in the second pass)
                if (s.l_qseq.p[r] <= TABLE_MAX) C.hist[s.l_qseq.p[r]]++;
                if (s.l_qseq.p[r] >= (1 << 22)) { bad(i, BRC_E_LIMIT, "reads of 4 Mbases and more are not supported"); break; }
                if (s.l_qseq.p[r] > C.max_lq) C.max_lq = s.l_qseq.p[r];
                const int32_t pos = s.pos.p[r];
                const int32_t prev = i == 0 ? e->last_pos : s.pos.p[r - 1];
                if (pos < prev) { bad(i, BRC_E_ARG, "reads are not coordinate-sorted"); break; }
                const uint16_t fl = (uint16_t)(s.flag.p[r] & 0x7fffu);
                if (s.l_qseq.p[r] > 0 && nc > 0) {
                    int64_t ql = 0; bool empty_m = false, eqx = false;
                    for (uint32_t k = 0; k < nc; ++k) { const uint32_t cg = s.cigar.p[s.cig_off.p[r] + k], op = cg & 0xfu; if (op == CMATCH || op == CINS || op == CSOFT_CLIP || op == CEQUAL || op == CDIFF) ql += cg >> 4; if (is_mop(op) && (cg >> 4) == 0u) empty_m = true; if (op == CEQUAL || op == CDIFF) eqx = true; }
                    if (ql != s.l_qseq.p[r]) {
                        if (!(fl & FUNMAP)) { bad(i, BRC_E_ARG, "a read's CIGAR and sequence length disagree"); break; }
                        nc = 0; s.n_cigar.p[r] = 0;
                    } else if (empty_m && !(fl & FUNMAP)) C.empty_m = true;
                    if (eqx) C.eqx = true;
                }
                uint64_t idp = 0;
                const int32_t rlen = cigar_rlen(s.cigar.p + s.cig_off.p[r], nc, &idp);
                if (rlen < 0 || (int64_t)s.pos.p[r] + rlen > (int64_t)INT32_MAX) {
                    if (!(fl & FUNMAP)) { bad(i, BRC_E_ARG, "a read ends beyond the last 32-bit position"); break; }
                    nc = 0; s.n_cigar.p[r] = 0; idp = 0;
                }
                s.iev_off.p[r] = (uint32_t)idp; C.idp += idp;                            // (count now, offset in the second pass)
                if (s.l_qseq.p[r] == 0 && nc > 0 && !(fl & (FUNMAP | BRC_NOCOUNT_MASK))) { bad(i, BRC_E_ARG, "a read without sequence would be counted"); break; }
                const int32_t end = (!(fl & FUNMAP) && nc > 0) ? pos + rlen : pos + 1;
                if (rlen > C.max_span) C.max_span = rlen;
                const bool accept = !(fl & FUNMAP) && pos >= 0;
                if (accept) { if (pos < C.min_pos) C.min_pos = pos; if (end > C.max_end) C.max_end = end; C.n_ext++; C.acc++; C.last_acc_pos = pos; }
                s.flag.p[r] = fl;
                const uint32_t* cg = s.cigar.p + s.cig_off.p[r];
                const bool entered = read_enters(fl, cg, nc) && pos >= 0 && !(e->cfg.per_lib && s.lib.p[r] < 0);
                const bool counts = (int)s.mapq.p[r] >= e->cfg.min_mapq && !(fl & BRC_NOCOUNT_MASK);
                uint32_t np = 0; bool past = false;
                const int32_t lq = s.l_qseq.p[r];
                walk_pieces(e->cfg.insertion_centric != 0, entered, counts, pos, cg, nc, [&](int32_t, int32_t len, int32_t, int qoff, bool) { ++np; if (len > 0 && qoff + len > lq) past = true; });
                if (past) { bad(i, BRC_E_ARG, kEmptyM); break; }
                s.piece_cnt.p[r] = np; C.np += np;
                s.wide.p[r] = read_has_escape(b->qual + b->qual_off[i], b->seq4 + b->seq_off[i], lq) ? 1 : 0;
            }
        });
        for (size_t ci = 0; ci < nch; ++ci) if (ch[ci].err) return fail(e, ch[ci].err, ch[ci].msg);      // (chunks are in file order: the first bad record's message)
        // scan of the chunks' totals, then offsets
        std::vector<uint64_t> bq0(nch), idp0(nch);
        for (size_t ci = 0; ci < nch; ++ci) {
            const Chunk& C = ch[ci];
            bq0[ci] = s.bq_elems; idp0[ci] = s.n_indel_ops;
            s.bq_elems += C.bq; s.n_indel_ops += C.idp;
            if ((uint64_t)(s.n_pieces += (int64_t)C.np) >= 0xFFFFFFF0ull) return fail(e, BRC_E_LIMIT, "more than 2^32 read segments in one region: split the region");
            for (int l = 0; l <= TABLE_MAX; ++l) s.len_hist[l] += C.hist[l];
            if (C.max_lq > s.max_lqseq) s.max_lqseq = C.max_lq;
            if (C.max_span > s.max_span) s.max_span = C.max_span;
            if (C.n_ext) {
                if (e->n_ext == 0) { s.min_pos = (int64_t)C.min_pos; s.max_end = (int64_t)C.max_end; }
                else { if (C.min_pos < s.min_pos) s.min_pos = C.min_pos; if (C.max_end > s.max_end) s.max_end = C.max_end; }
                e->n_ext += C.n_ext;
            }
            if (C.acc) { e->accepted += C.acc; e->last_acc_pos = C.last_acc_pos; }
            if (C.empty_m) s.has_empty_m = true;
            if (C.eqx) s.has_eqx = true;
        }
        pool->run((int64_t)nch, [&](int64_t ci) {
            uint64_t bq = bq0[(size_t)ci], idp = idp0[(size_t)ci];
            const size_t i0 = (size_t)ci * CH, i1 = std::min(n, i0 + CH);
            for (size_t i = i0; i < i1; ++i) {
                const size_t r = n0 + i;
                const uint64_t len = s.bq_row.p[r]; s.bq_row.p[r] = bq; bq += len;
                const uint32_t k = s.iev_off.p[r]; s.iev_off.p[r] = (uint32_t)idp; idp += k;
            }
        });
        e->last_pos = s.pos.p[n0 + n - 1];
        s.n += (int64_t)n;
        return BRC_OK;
    }
    for (size_t i = 0; i < n; ++i) {
        const size_t r = n0 + i;
        uint32_t nc = s.n_cigar.p[r];
        if (b->cigar_off[i] + nc > b->n_cigar_total || b->qual_off[i] + (uint64_t)(s.l_qseq.p[r] > 0 ? s.l_qseq.p[r] : 0) > b->qual_bytes ||
            b->seq_off[i] + (uint64_t)((s.l_qseq.p[r] + 1) / 2) > b->seq_bytes || s.l_qseq.p[r] < 0)
            return fail(e, BRC_E_ARG, "read offsets outside the batch arenas");
        if (e->cfg.per_lib && s.lib.p[r] >= e->g.Lp) return fail(e, BRC_E_ARG, "library index out of range");
        s.cig_off.p[r] += cb; s.seq_off.p[r] += sb; s.qual_off.p[r] += qb;
        s.bq_row.p[r] = s.bq_elems; s.bq_elems += ((uint64_t)s.l_qseq.p[r] + 15u) & ~(uint64_t)15u;
        if (s.l_qseq.p[r] <= TABLE_MAX) s.len_hist[s.l_qseq.p[r]]++;
        if (s.l_qseq.p[r] >= (1 << 22)) return fail(e, BRC_E_LIMIT, "reads of 4 Mbases and more are not supported");
        if (s.l_qseq.p[r] > s.max_lqseq) s.max_lqseq = s.l_qseq.p[r];
        const int32_t pos = s.pos.p[r];
        if (pos < e->last_pos) return fail(e, BRC_E_ARG, "reads are not coordinate-sorted");
        e->last_pos = pos;
        uint16_t fl = (uint16_t)(s.flag.p[r] & 0x7fffu);
        // A record whose CIGAR walks more (or fewer) query bases than it has would send the annotator outside the read's
        // quality / base rows (htslib indexes the record's memory just the same: undefined there).  Mapped: refused.
        // Unmapped (some aligners leave the mate's CIGAR on such records; they never reach a column): the CIGAR is dropped.
        if (s.l_qseq.p[r] > 0 && nc > 0) {
            int64_t ql = 0; bool empty_m = false, eqx = false;
            for (uint32_t k = 0; k < nc; ++k) { const uint32_t cg = s.cigar.p[s.cig_off.p[r] + k], op = cg & 0xfu; if (op == CMATCH || op == CINS || op == CSOFT_CLIP || op == CEQUAL || op == CDIFF) ql += cg >> 4; if (is_mop(op) && (cg >> 4) == 0u) empty_m = true; if (op == CEQUAL || op == CDIFF) eqx = true; }
            if (ql != s.l_qseq.p[r]) {
                if (!(fl & FUNMAP)) return fail(e, BRC_E_ARG, "a read's CIGAR and sequence length disagree");
                nc = 0; s.n_cigar.p[r] = 0;
            } else if (empty_m && !(fl & FUNMAP)) s.has_empty_m = true;
            if (eqx) s.has_eqx = true;
        }
        uint64_t idp = 0;                                     // I / D / P operators of the CIGAR the device will see
        const int32_t rlen = cigar_rlen(s.cigar.p + s.cig_off.p[r], nc, &idp);
        if (rlen < 0 || (int64_t)s.pos.p[r] + rlen > (int64_t)INT32_MAX) {
            if (!(fl & FUNMAP)) return fail(e, BRC_E_ARG, "a read ends beyond the last 32-bit position");
            nc = 0; s.n_cigar.p[r] = 0; idp = 0;
        }
        // the read's slots in the raw indel-event list (K1 writes every one of them: an event or an empty mark)
        s.iev_off.p[r] = (uint32_t)s.n_indel_ops; s.n_indel_ops += idp;
        if (s.l_qseq.p[r] == 0 && nc > 0 && !(fl & (FUNMAP | BRC_NOCOUNT_MASK))) {
            // (SEQ "*" on a record that pileup_func would count: the reference takes its bases from whatever follows the
            // empty sequence in the record)
            return fail(e, BRC_E_ARG, "a read without sequence would be counted");
        }
        const int32_t end = (!(fl & FUNMAP) && nc > 0) ? pos + rlen : pos + 1;           // bam_endpos
        if (rlen > s.max_span) s.max_span = rlen;
        bool accept = !(fl & FUNMAP) && pos >= 0;      // bam_plp_push (htslib 1.10) skips unmapped reads only
        if (accept) {   // region extent: every read bam_plp_push takes (max-count drops included)
            if (e->n_ext == 0) { s.min_pos = pos; s.max_end = end; }
            else { if (pos < s.min_pos) s.min_pos = pos; if (end > s.max_end) s.max_end = end; }
            e->n_ext++;
        }
        if (accept && e->accepted >= maxcnt) {
            // bam_plp_push: drop when iter->pos == b->core.pos && mempool count > maxcnt.  iter->pos equals the
            // start of the last accepted read; live nodes are accepted reads with end >= pos (lazy removal).
            if (!e->heap_built) {
                uint64_t dummy = 0;
                for (size_t j = 0; j < r; ++j) {
                    const uint16_t f2 = s.flag.p[j];
                    if (f2 & BRC_PUSH_MASK) continue;
                    const int32_t rl = cigar_rlen(s.cigar.p + s.cig_off.p[j], s.n_cigar.p[j], &dummy);
                    e->live_ends.push(s.n_cigar.p[j] > 0 ? s.pos.p[j] + rl : s.pos.p[j] + 1);
                }
                e->heap_built = true;
            }
            while (!e->live_ends.empty() && e->live_ends.top() < pos) e->live_ends.pop();
            if (pos == e->last_acc_pos && (int64_t)e->live_ends.size() + 1 > (int64_t)maxcnt) { accept = false; fl |= FHOSTDROP; }
        }
        s.flag.p[r] = fl;
        if (accept) {
            e->accepted++; e->last_acc_pos = pos;
            if (e->heap_built) e->live_ends.push(end);
        }
        {   // pieces of this read (KB v2): none for a read outside the columns or without a library (-p: it abandons positions instead)
            const uint32_t* cg = s.cigar.p + s.cig_off.p[r];
            const bool entered = read_enters(fl, cg, nc) && pos >= 0 && !(e->cfg.per_lib && s.lib.p[r] < 0);
            const bool counts = (int)s.mapq.p[r] >= e->cfg.min_mapq && !(fl & BRC_NOCOUNT_MASK);
            uint32_t np = 0; bool past = false;
            const int32_t lq = s.l_qseq.p[r];
            walk_pieces(e->cfg.insertion_centric != 0, entered, counts, pos, cg, nc, [&](int32_t, int32_t len, int32_t, int qoff, bool) { ++np; if (len > 0 && qoff + len > lq) past = true; });
            if (past) return fail(e, BRC_E_ARG, kEmptyM);
            s.piece_cnt.p[r] = np;
            if ((uint64_t)(s.n_pieces += np) >= 0xFFFFFFF0ull) return fail(e, BRC_E_LIMIT, "more than 2^32 read segments in one region: split the region");
            s.wide.p[r] = read_has_escape(b->qual + b->qual_off[i], b->seq4 + b->seq_off[i], lq) ? 1 : 0;
        }
    }
    s.n += (int64_t)n;
    return BRC_OK;
}

int brc_upload(brc_engine* e) {
    if (!e) return BRC_E_ARG;
    if (e->state != 1) return fail(e, BRC_E_ARG, "brc_upload needs an open region");
    const double t_in = now_s();
    Geometry& g = e->g; const Staged& s = e->st;
    int64_t lo = g.beg0 > 0 ? g.beg0 - 1 : 0, hi = g.end;
    if (e->n_ext == 0) { hi = lo; }
    else { if (s.min_pos > lo) lo = s.min_pos; if (s.max_end < hi) hi = s.max_end; if (hi < lo) hi = lo; }
    g.pos0 = (int32_t)lo; g.P = hi - lo;
    g.ref_lo = g.ref_hi = 0;
    if (g.ref && e->n_ext) {
        g.ref_lo = std::min<int64_t>(std::max<int64_t>(s.min_pos, 0), g.ref_len);
        g.ref_hi = std::min<int64_t>(std::max<int64_t>(s.max_end, g.ref_lo), g.ref_len);
    }
    e->st.layout_pieces(g.Lp, e->cfg.per_lib != 0);
    int rc = e->be->upload(e->cfg, s, g);
    if (rc) return fail(e, rc, e->be->last_error());
    e->state = 2; e->t_upload += now_s() - t_in; e->n_regions++;
    return BRC_OK;
}

static int compute_passes(brc_engine* e, int32_t n, brc_timing* t);
int brc_compute(brc_engine* e, brc_timing* t) { return compute_passes(e, 0, t); }
int brc_compute_n(brc_engine* e, int32_t n, brc_timing* t) {
    if (n < 1) return e ? fail(e, BRC_E_ARG, "brc_compute_n: at least one pass") : BRC_E_ARG;
    return compute_passes(e, n, t);
}
static int compute_passes(brc_engine* e, int32_t n, brc_timing* t) {
    if (!e) return BRC_E_ARG;
    if (e->state < 2) return fail(e, BRC_E_ARG, "brc_compute before brc_upload");
    const double t_in = now_s();
    int rc = n > 0 ? e->be->compute_n(n, t) : e->be->compute(t);
    if (rc) return fail(e, rc, e->be->last_error());
    // device-side text: the line kernels and the download start as soon as the region is computed (lines above 4 GiB per
    // region would overflow the 32-bit offsets: such regions are formatted on the host)
    // (indel entries are part of the lines now: their share of the estimate comes from the list sizes of this pass)
    uint64_t nx_ = 0, ni_ = 0; e->be->list_sizes(&nx_, &ni_);
    e->text_computed = e->text_only && e->device_text && !e->chrom.empty() &&
                       (double)e->g.P * (double)e->g.Lp * 700.0 + 64.0 * (double)e->g.P + 2.0 * (double)ni_ * (700.0 + (double)e->st.max_span) < 4.0e9;
    if (e->text_computed) {
        rc = e->be->text_begin(e->chrom, e->libs, &e->text_slot_computed);
        if (rc == BRC_TEXT_TOO_LONG) e->text_computed = false;          // (the estimate above was too low: long library names, sums at the far end of int32)
        else if (rc) return fail(e, rc, e->be->last_error());
    }
    e->state = 3; e->t_compute += now_s() - t_in;
    return BRC_OK;
}

int brc_fetch_result(brc_engine* e, brc_result* out) {
    if (!e || !out) return BRC_E_ARG;
    if (e->state < 3) return fail(e, BRC_E_ARG, "brc_fetch_result before brc_compute");
    const double t_in = now_s();
    // device-side text: no planes come to the host (lines above 4 GiB per region would overflow the 32-bit offsets)
    e->text_result = e->text_computed; e->text_slot = e->text_slot_computed;
    int rc = e->be->fetch(&e->hp, !e->text_result);
    if (rc) return fail(e, rc, e->be->last_error());
    const double t_dl = now_s(); e->t_d2h += t_dl - t_in;
    e->n_xev_total += e->hp.n_xagg; e->n_indel_total += (uint64_t)e->hp.n_indel;
    const Geometry& g = e->g; const HostPlanes& hp = e->hp;
    // column 3: raw reference character (bamreadcount.cpp:353)
    e->refbase.resize((size_t)g.P + 1);
    if (!e->text_result) {
        const int64_t have = g.ref ? std::max<int64_t>(0, std::min<int64_t>(g.P, g.ref_len - g.pos0)) : 0;
        if (have) memcpy(e->refbase.data(), g.ref + g.pos0, (size_t)have);
        for (int64_t k = 0; k < have; ++k) if (!e->refbase[(size_t)k]) e->refbase[(size_t)k] = 'N';
        if (g.P > have) memset(e->refbase.data() + have, 'N', (size_t)(g.P - have));
    }
    // a text result leaves the indel buckets to the device's lines; the host keeps what its deletion queues need (prepare_text_queues)
    if (e->text_result) { e->indels.clear(); e->alleles.clear(); prepare_text_queues(e); }
    else assemble_indels(e, hp.indel, hp.n_indel, e->indels, e->alleles);
    memset(out, 0, sizeof *out);
    out->tid = g.tid; out->beg0 = g.beg0; out->end = g.end; out->pos0 = g.pos0; out->n_pos = g.P; out->stride = g.PS; out->n_lib = g.Lp;
    if (e->text_result) {
    } else if (e->text_only) {
        // third-allele buckets -> sorted by (position, library, bucket) for the formatter's merge (the device groups them by tile and library)
        e->xagg.clear(); e->xagg.reserve((size_t)hp.n_xagg);
        for (uint64_t i = 0; i < hp.n_xagg; ++i) {
            const XAgg& x = hp.xagg[i];
            if ((int64_t)(x.lib_b >> 8) >= g.Lp || (x.lib_b & 0xffu) >= (uint32_t)NBUCKET || (int64_t)x.k >= g.P) continue;
            XKey a; a.key = ((uint64_t)x.k << 16) | (uint64_t)(x.lib_b & 0xffffu);
            for (int f = 0; f < BRC_NI; ++f) a.st.i[f] = x.i[f];
            for (int f = 0; f < BRC_NF; ++f) a.st.f[f] = x.f[f];
            e->xagg.push_back(a);
        }
        std::sort(e->xagg.begin(), e->xagg.end(), [](const XKey& a, const XKey& b) { return a.key < b.key; });
    } else {
        const size_t need = (size_t)g.Lp * NBUCKET * (size_t)g.PS + 16;
        if (need > e->dense_cap) {
            free(e->dense_i); free(e->dense_f);
            e->dense_i = (uint32_t*)malloc(need * NI * 4); e->dense_f = (float*)malloc(need * NF * 4); e->dense_cap = need;
            if (!e->dense_i || !e->dense_f) { e->dense_cap = 0; return fail(e, BRC_E_NOMEM, "host allocation of the dense planes failed"); }
        }
        expand_slots(hp, g.Lp, g.P, g.PS, e->dense_i, e->dense_f);
    }
    out->ncol = hp.ncol; out->depth = hp.depth;
    out->istat = e->text_only ? NULL : e->dense_i; out->fstat = e->text_only ? NULL : e->dense_f;
    out->unavail = e->cfg.per_lib ? hp.unavail : NULL;
    out->refbase = e->refbase.data();
    if (e->text_result) {
        out->ncol = out->depth = out->unavail = NULL; out->refbase = NULL;
    }
    out->n_indel = (int64_t)e->indels.size(); out->indel = e->indels.data();
    out->alleles = e->alleles.data(); out->alleles_len = e->alleles.size();
    out->n_events = hp.n_events;
    for (int w = 0; w < BRC_N_WARN; ++w) out->warn[w] = hp.warn[w];
    e->state = 4; e->t_post += now_s() - t_dl;
    return BRC_OK;
}

int brc_fetch_window(brc_engine* e, int32_t beg0, int32_t end, brc_result* out) {
    if (!e || !out) return BRC_E_ARG;
    if (e->state < 3) return fail(e, BRC_E_ARG, "brc_fetch_window before brc_compute");
    const Geometry& g = e->g;
    if (beg0 < g.beg0 || end > g.end || end < beg0) return fail(e, BRC_E_ARG, "brc_fetch_window: the window must lie inside the computed region");
    // plane indices of [beg0 - 1, end) clipped to the planes (the lead position only when the region processed it)
    int64_t k0 = (int64_t)(beg0 > 0 ? beg0 - 1 : 0) - g.pos0, k1 = (int64_t)end - g.pos0;
    if (k0 < 0) k0 = 0;
    if (k0 > g.P) k0 = g.P;                                  // (a window behind the reads' extent: no planes, an empty result)
    if (k1 > g.P) k1 = g.P;
    if (k1 < k0) k1 = k0;
    const int64_t n = k1 - k0;
    HostPlanes hw; int64_t WS = 0;
    int rc = e->be->fetch_window(k0, n, &hw, &WS);
    if (rc) return fail(e, rc, e->be->last_error());
    const int32_t wpos0 = (int32_t)(g.pos0 + k0);
    // the two lists cover the whole region: keep the window's entries (third-allele events re-based to the window's planes)
    e->win_xagg.clear(); e->win_iout.clear();
    for (uint64_t i = 0; i < hw.n_xagg; ++i) { const XAgg& x = hw.xagg[i]; if ((int64_t)x.k >= k0 && (int64_t)x.k < k1) { XAgg y = x; y.k = (uint32_t)((int64_t)x.k - k0); e->win_xagg.push_back(y); } }
    for (int64_t i = 0; i < hw.n_indel; ++i) { const IndelOut& o = hw.indel[i]; if (o.pos >= wpos0 && (int64_t)o.pos < (int64_t)wpos0 + n) e->win_iout.push_back(o); }
    hw.xagg = e->win_xagg.data(); hw.n_xagg = e->win_xagg.size();
    assemble_indels(e, e->win_iout.data(), (int64_t)e->win_iout.size(), e->win_indels, e->win_alleles);
    const size_t need = (size_t)g.Lp * NBUCKET * (size_t)WS + 16;
    if (need > e->win_cap) {
        free(e->win_i); free(e->win_f);
        e->win_i = (uint32_t*)malloc(need * NI * 4); e->win_f = (float*)malloc(need * NF * 4); e->win_cap = need;
        if (!e->win_i || !e->win_f) { e->win_cap = 0; return fail(e, BRC_E_NOMEM, "host allocation of the dense planes failed"); }
    }
    expand_slots(hw, g.Lp, n, WS, e->win_i, e->win_f);
    e->win_refbase.resize((size_t)n + 1);
    {
        const int64_t have = g.ref ? std::max<int64_t>(0, std::min<int64_t>(n, g.ref_len - wpos0)) : 0;
        if (have) memcpy(e->win_refbase.data(), g.ref + wpos0, (size_t)have);
        for (int64_t k = 0; k < have; ++k) if (!e->win_refbase[(size_t)k]) e->win_refbase[(size_t)k] = 'N';
        if (n > have) memset(e->win_refbase.data() + have, 'N', (size_t)(n - have));
    }
    memset(out, 0, sizeof *out);
    out->tid = g.tid; out->beg0 = beg0; out->end = end; out->pos0 = wpos0; out->n_pos = n; out->stride = WS; out->n_lib = g.Lp;
    out->ncol = hw.ncol; out->depth = hw.depth; out->istat = e->win_i; out->fstat = e->win_f;
    out->unavail = e->cfg.per_lib ? hw.unavail : NULL;
    out->refbase = e->win_refbase.data();
    out->n_indel = (int64_t)e->win_indels.size(); out->indel = e->win_indels.data();
    out->alleles = e->win_alleles.data(); out->alleles_len = e->win_alleles.size();
    // events of the window: its pileup columns inside [beg0, end), abandoned positions left out as the device counts them
    uint64_t ev = 0;
    for (int l = 0; l < g.Lp; ++l) for (int64_t k = 0; k < n; ++k) {
        if ((int64_t)wpos0 + k < (int64_t)beg0) continue;
        if (e->cfg.per_lib && hw.unavail && hw.unavail[k] != 0xFFFFFFFFu) continue;
        ev += hw.ncol[(int64_t)l * WS + k];
    }
    out->n_events = ev;
    return BRC_OK;
}

int brc_end_region(brc_engine* e, brc_result* out) {
    int rc = brc_upload(e);
    if (rc) return rc;
    rc = brc_compute(e, NULL);
    if (rc) return rc;
    return brc_fetch_result(e, out);
}

int brc_region_counts(brc_engine* e, uint64_t* n_events, uint64_t* n_positions) {
    if (!e) return BRC_E_ARG;
    if (e->state < 3) return fail(e, BRC_E_ARG, "brc_region_counts before brc_compute");
    int rc = e->be->counts(n_events, n_positions);
    if (rc) return fail(e, rc, e->be->last_error());
    return BRC_OK;
}

int brc_region_piece_steps(brc_engine* e, uint64_t* ranged, uint64_t* walked) {
    if (!e || !ranged || !walked) return BRC_E_ARG;
    if (e->state < 3) return fail(e, BRC_E_ARG, "brc_region_piece_steps before brc_compute");
    e->be->piece_steps(ranged, walked);
    return BRC_OK;
}

int brc_clear_indel_queue(brc_engine* e) {
    if (!e) return BRC_E_ARG;
    for (size_t l = 0; l < e->queue.size(); ++l) e->queue[l].clear();
    return BRC_OK;
}

// Record assembly for plane indices [k0,k1) into `out`, with deletion queues `queue` (one FIFO per library).
// Lines are printed for positions inside [wbeg0, wend) with coordinate pos + 1 - delta.  Returns false when memory ran out.
static bool format_range(const brc_engine* e, const brc_result* r, const char* chrom, int64_t k0, int64_t k1,
                         std::vector<std::deque<QEnt> >& queue, TextBuf& out, int32_t wbeg0, int32_t wend, int32_t delta) {
    const int Lp = r->n_lib; const int64_t S = r->stride;
    const bool per_lib = e->cfg.per_lib != 0;
    const size_t chrom_len = strlen(chrom);
    uint32_t si[BRC_NI]; float sf[BRC_NF];
    // room one line needs without its indel entries: chrom, coordinate, reference base, depth, and per library its name,
    // braces and six buckets
    size_t max_name = 0; for (const std::string& n : e->libs) max_name = std::max(max_name, n.size());
    const size_t line0 = chrom_len + 64 + (size_t)Lp * (max_name + 16 + (size_t)BRC_NBUCKET * (STAT_MAX + 4));
    // the six "\tX:" + all-zero buckets, precomposed
    char zero[BRC_NBUCKET][ZERO_LEN + 3];
    for (int b = 0; b < BRC_NBUCKET; ++b) { zero[b][0] = '\t'; zero[b][1] = "=ACGTN"[b]; zero[b][2] = ':'; memcpy(zero[b] + 3, kZeroStat, ZERO_LEN); }
    // cursor into the (pos, lib, allele)-sorted indel list: first entry with pos >= pos0 + k0
    int64_t ii;
    {
        const int32_t p0 = r->pos0 + (int32_t)k0;
        int64_t lo = 0, hi = r->n_indel;
        while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (r->indel[m].pos < p0) lo = m + 1; else hi = m; }
        ii = lo;
    }
    // text-only engines: the buckets come from the two slots of a position (+ the sparse third-allele table)
    const bool compact = r->istat == NULL;
    const HostPlanes& hp = e->hp;
    size_t xi = 0;
    if (compact) xi = (size_t)(std::lower_bound(e->xagg.begin(), e->xagg.end(), (uint64_t)k0 << 16, [](const XKey& a, uint64_t key) { return a.key < key; }) - e->xagg.begin());
    const uint32_t tid = (uint32_t)r->tid;
    for (int64_t k = k0; k < k1; ++k) {
        const int32_t pos = r->pos0 + (int32_t)k;
        while (ii < r->n_indel && r->indel[ii].pos < pos) ++ii;
        if (compact) while (xi < e->xagg.size() && (int64_t)(e->xagg[xi].key >> 16) < k) ++xi;
        if (per_lib && r->unavail && r->unavail[k] != 0xFFFFFFFFu) continue;            // :281-284: position abandoned
        uint32_t tot = 0, depth = 0;
        for (int l = 0; l < Lp; ++l) { tot += r->ncol[(int64_t)l * S + k]; depth += r->depth[(int64_t)l * S + k]; }
        if (tot == 0) continue;                                                           // no reads: no pileup callback
        // deletions queued for this position add their read counts to the depth column (IndelQueue.cpp:11, :415); only
        // the libraries present in the column look at their queue (:360-411)
        for (int l = 0; l < Lp; ++l) {
            const std::deque<QEnt>& q = queue[(size_t)l];
            if (q.empty() || r->ncol[(int64_t)l * S + k] == 0) continue;
            // exactly IndelQueue::process: stale entries are dropped from the FRONT only, then the run of entries due here
            // is taken — an entry behind one that is due later (or behind a stale one that follows a due one) stays
            bool taking = false;
            for (const QEnt& x : q) {
                const bool stale = x.tid != tid || x.pos < (uint32_t)pos;
                if (!taking && stale) continue;
                if (x.tid != tid || x.pos != (uint32_t)pos) break;
                taking = true; depth += x.st.i[I_N];
            }
        }
        const size_t line_start = out.n;
        char* w = out.room(line0);
        if (!w) return false;
        memcpy(w, chrom, chrom_len); w += chrom_len; *w++ = '\t';                        // :414-416
        w += fmt_u32(w, (uint32_t)(pos + 1 - delta)); *w++ = '\t';
        *w++ = r->refbase[k]; *w++ = '\t';
        w += fmt_u32(w, depth);
        for (int l = 0; l < Lp; ++l) {
            if (r->ncol[(int64_t)l * S + k] == 0) continue;                               // lib_counts has no entry (:286,360)
            if (per_lib) { const std::string& nm = e->libs[(size_t)l]; *w++ = '\t'; memcpy(w, nm.data(), nm.size()); w += nm.size(); *w++ = '\t'; *w++ = '{'; }
            if (compact) {
                const uint32_t sid = hp.slotid[(int64_t)l * S + k];
                const uint32_t b0 = sid & 0xffu, b1 = (sid >> 8) & 0xffu;
                for (uint32_t b = 0; b < (uint32_t)BRC_NBUCKET; ++b) {
                    // a bucket's events sit in ONE place: the slot that names it, or the third-allele table — also for a bucket a slot
                    // names: the events of an N / '=' base never enter the slots, not even where N is the reference's own bucket
                    const int sl = b == b0 ? 0 : (b == b1 ? 1 : -1);
                    const uint32_t* ip = sl >= 0 ? hp.si + (((int64_t)l * 2 + sl) * NI) * S + k : nullptr;
                    if (ip && ip[(int64_t)I_N * S] != 0) {
                        const float* fp = hp.sf + (((int64_t)l * 2 + sl) * NF) * S + k;
                        for (int f = 0; f < BRC_NI; ++f) si[f] = ip[(int64_t)f * S];
                        for (int f = 0; f < BRC_NF; ++f) sf[f] = fp[(int64_t)f * S];
                    } else {
                        si[I_N] = 0;
                        for (size_t x = xi; x < e->xagg.size() && (int64_t)(e->xagg[x].key >> 16) == k; ++x)
                            if ((e->xagg[x].key & 0xffffu) == (((uint64_t)l << 8) | (uint64_t)b)) { memcpy(si, e->xagg[x].st.i, sizeof si); memcpy(sf, e->xagg[x].st.f, sizeof sf); break; }
                        if (si[I_N] == 0) { memcpy(w, zero[b], ZERO_LEN + 3); w += ZERO_LEN + 3; continue; }
                    }
                    memcpy(w, zero[b], 3); w = fmt_stat(w + 3, si, sf, false);
                }
            } else {
                for (int b = 0; b < BRC_NBUCKET; ++b) {
                    // (the planes are position-major: look at the count plane first, the other 12 only for occupied buckets)
                    si[I_N] = r->istat[(((int64_t)l * BRC_NBUCKET + b) * BRC_NI + I_N) * S + k];
                    if (si[I_N] == 0) { memcpy(w, zero[b], ZERO_LEN + 3); w += ZERO_LEN + 3; continue; }
                    for (int f = 0; f < BRC_NI; ++f) si[f] = r->istat[(((int64_t)l * BRC_NBUCKET + b) * BRC_NI + f) * S + k];
                    for (int f = 0; f < BRC_NF; ++f) sf[f] = r->fstat[(((int64_t)l * BRC_NBUCKET + b) * BRC_NF + f) * S + k];
                    memcpy(w, zero[b], 3); w = fmt_stat(w + 3, si, sf, false);
                }
            }
            // an indel entry needs room of its own (the allele text is as long as the indel): re-anchor the write pointer
            auto entry = [&](const char* allele, size_t alen, const uint32_t* ei, const float* ef) -> bool {
                out.n = (size_t)(w - out.p);
                w = out.room(alen + 8 + STAT_MAX + line0);
                if (!w) return false;
                *w++ = '\t'; memcpy(w, allele, alen); w += alen; *w++ = ':';
                w = fmt_stat(w, ei, ef, true);
                return true;
            };
            while (ii < r->n_indel && r->indel[ii].pos == pos && r->indel[ii].lib < l) ++ii;
            for (; ii < r->n_indel && r->indel[ii].pos == pos && r->indel[ii].lib == l; ++ii) {
                const brc_indel& d = r->indel[ii];
                if (d.len < 0) {                                                          // :391-396
                    QEnt q; q.tid = tid; q.pos = (uint32_t)pos + 1; q.st = d.stat;
                    q.allele.assign(r->alleles + d.allele_off, d.allele_len);
                    queue[(size_t)l].push_back(q);
                } else if (!entry(r->alleles + d.allele_off, d.allele_len, d.stat.i, d.stat.f)) return false;   // :399
            }
            // IndelQueue::process (IndelQueue.cpp:3-15)
            std::deque<QEnt>& q = queue[(size_t)l];
            while (!q.empty() && ((q.front().tid == tid && q.front().pos < (uint32_t)pos) || q.front().tid != tid)) q.pop_front();
            while (!q.empty() && q.front().tid == tid && q.front().pos == (uint32_t)pos) {
                if (!entry(q.front().allele.data(), q.front().allele.size(), q.front().st.i, q.front().st.f)) return false;
                q.pop_front();
            }
            if (per_lib) { *w++ = '\t'; *w++ = '}'; }
        }
        *w++ = '\n';
        out.n = (pos >= wbeg0 && pos < wend) ? (size_t)(w - out.p) : line_start;          // the lead position only feeds the queues
    }
    return true;
}

// Chunks of positions are formatted by a pool of threads.  A queued deletion lives for exactly one position (pushed at
// p for p+1, emitted or dropped there), so a chunk starting at k0 > 0 reproduces the queue state it would inherit by
// replaying position k0-1 into a scratch buffer; chunk 0 continues the engine's persistent queues and the last chunk's
// final queues become the engine's (regions given on the command line are not separated by a clear, :641-657).
// BRC_OPT_DEVICE_TEXT: the lines came from the GPU (Backend::text_begin).  What a lane cannot know is finished here, for the
// few lines it concerns — the same record assembly as format_range, applied to the device's line instead of the planes:
//   * indel buckets of the position (:389-401): insertions are appended to their library's block, deletions are queued for pos+1;
//   * IndelQueue::process (IndelQueue.cpp:3-15) for the libraries present in the line: queued deletions due at this position
//     are appended and their read counts added to the depth column;
//   * buckets of a third base: the device printed them as empty, their sums are in the third-allele table.
// Every line of the region is a candidate while a queue is not empty (process() runs at every printed position), so the
// deque semantics — including entries stuck behind a later-due front entry — are the reference's.
// Output: e->part_ptr / e->part_len, pieces of the downloaded text interleaved with the rewritten lines.
static int format_device_text(brc_engine* e, const brc_result* r) {
    HostText ht;
    const double t_w = now_s();
    int rc = e->be->text_wait(e->text_slot, &ht);
    if (rc) return fail(e, rc, e->be->last_error());
    e->t_textwait += now_s() - t_w;
    const int Lp = r->n_lib; const int64_t P = r->n_pos;
    const bool per_lib = e->cfg.per_lib != 0;
    const uint32_t tid = (uint32_t)r->tid;
    e->pbuf.clear(); e->patches.clear();
    std::vector<std::deque<QEnt> >& queue = e->queue;
    const bool lead = P > 0 && r->pos0 < r->beg0;                   // plane index 0 is the lead position: never printed
    // ---- did the region start from the queues its lines assume?  Empty ones — or, for a region that continues the piece before it, nothing
    // but what that piece's last position (this region's lead position) queued: the device queued the same again
    bool busy = false, expected = true;
    for (const std::deque<QEnt>& q : queue) for (const QEnt& x : q) { busy = true; if (!(e->continues && lead && x.tid == tid && x.pos == (uint32_t)r->beg0)) expected = false; }
    if (!busy || expected) {
        e->part_ptr.clear(); e->part_len.clear();
        const uint64_t cur = lead ? ht.off[1] : 0;
        if (P > 0 && ht.total > cur) { e->part_ptr.push_back(ht.text + cur); e->part_len.push_back((size_t)(ht.total - cur)); }
        // what the region leaves pending: per library the deletions of its last processed position (a library that was never processed
        // keeps what it had)
        for (int l = 0; l < Lp; ++l) {
            const uint32_t lp = ht.last_processed ? ht.last_processed[l] : NONE32;
            if (lp == NONE32) continue;
            std::deque<QEnt>& q = queue[(size_t)l]; q.clear();
            if ((size_t)l < e->tail_pos.size() && e->tail_pos[(size_t)l] == (int64_t)r->pos0 + (int64_t)lp) for (const QEnt& x : e->tail_dels[(size_t)l]) q.push_back(x);
        }
        return BRC_OK;
    }
    // ---- queues that hold something else (a deletion an earlier command-line region left pending, :641-657): the deletion entries of the
    // lines are rewritten with IndelQueue::process's own rules (IndelQueue.cpp:3-15) — entries stuck behind a later-due front entry included.
    // Every line is a candidate while a queue is not empty (process() runs at every printed position).
    if (!e->have_dels) return fail(e, BRC_E_ARG, "a text result must be formatted after the region before it (the deletion queues changed behind its back)");
    const std::vector<brc_engine::DelEnt>& dels = e->dels;
    size_t di = 0;
    auto queues_busy = [&]() { for (const std::deque<QEnt>& q : queue) if (!q.empty()) return true; return false; };
    char nb[STAT_MAX + 64];
    int64_t k = -1;
    for (;;) {
        while (di < dels.size() && (int64_t)dels[di].pos - r->pos0 <= k) ++di;           // behind the last line looked at
        int64_t kc = INT64_MAX;
        if (di < dels.size()) kc = std::min<int64_t>(kc, (int64_t)dels[di].pos - r->pos0);
        if (queues_busy()) { int64_t j = k + 1; while (j < P && ht.off[j + 1] == ht.off[j]) ++j; if (j < P) kc = std::min(kc, j); }
        if (kc == INT64_MAX || kc >= P) break;
        if (kc <= k) kc = k + 1;                                   // (defensive: cursors always move forward)
        k = kc;
        const int32_t pos = r->pos0 + (int32_t)k;
        while (di < dels.size() && dels[di].pos < pos) ++di;
        if (ht.off[k + 1] == ht.off[k] || (k == 0 && e->continues && lead)) {
            // no line: no pileup callback here, nothing is queued or processed — nor at the lead position of a region that
            // continues the previous one (it was that region's last position)
            while (di < dels.size() && dels[di].pos == pos) ++di;
            continue;
        }
        const char* L0 = ht.text + ht.off[k]; const char* const L1 = ht.text + ht.off[k + 1] - 1;    // [L0, L1): the line without its newline
        // prefix: chrom \t pos \t ref \t depth
        const char* p = L0; int tabs = 0;
        while (p < L1 && tabs < 3) { if (*p == '\t') ++tabs; ++p; }
        const char* const dep0 = p; uint32_t depth = 0;
        while (p < L1 && *p != '\t') { depth = depth * 10u + (uint32_t)(*p - '0'); ++p; }
        // the blocks, rebuilt behind room for the prefix and a depth of up to ten digits (the depth is known last)
        const size_t pre = (size_t)(dep0 - L0);
        const size_t base = e->pbuf.n;
        size_t cap_need = pre + 16 + (size_t)(L1 - L0) + 64;
        char* w0 = e->pbuf.room(cap_need);
        if (!w0) return fail(e, BRC_E_NOMEM, "host allocation of the text buffers failed");
        size_t wn = pre + 10;                                      // bytes used behind w0 (body starts here)
        bool changed = false; uint32_t extra = 0, dev_extra = 0;
        auto put = [&](const char* src, size_t len) -> bool {
            if (wn + len + 64 > cap_need) {                        // queued entries can make a line longer than the device's
                cap_need = wn + len + 4096;
                if (!e->pbuf.room(cap_need)) return false;
                w0 = e->pbuf.p + base;
            }
            memcpy(w0 + wn, src, len); wn += len;
            return true;
        };
        // the tokens of one library's block: six buckets and the insertion entries as they are; the device's deletion entries are dropped
        // (their read counts leave the depth)
        auto block_tokens = [&](const char* end) -> bool {
            int tok = 0;
            while (p < end) {
                const char* t0 = p;
                const char* t1 = (const char*)memchr(p + 1, '\t', (size_t)(end - p - 1));
                p = t1 ? t1 : end;
                if (tok >= BRC_NBUCKET && t0[1] == '-') {
                    const char* c = (const char*)memchr(t0, ':', (size_t)(p - t0)); uint32_t n = 0;
                    if (c) for (++c; c < p && *c >= '0' && *c <= '9'; ++c) n = n * 10u + (uint32_t)(*c - '0');
                    dev_extra += n; changed = true;
                } else if (!put(t0, (size_t)(p - t0))) return false;
                ++tok;
            }
            return true;
        };
        auto lib_tail = [&](int l) -> bool {
            auto entry = [&](const char* allele, size_t alen, const uint32_t* ei, const float* ef) -> bool {
                char* w = fmt_stat(nb, ei, ef, true);
                changed = true;
                return put("\t", 1) && put(allele, alen) && put(":", 1) && put(nb, (size_t)(w - nb));
            };
            while (di < dels.size() && dels[di].pos == pos && dels[di].lib < l) ++di;
            for (; di < dels.size() && dels[di].pos == pos && dels[di].lib == l; ++di) {          // :391-396
                QEnt q; q.tid = tid; q.pos = (uint32_t)pos + 1; q.st = dels[di].st; q.allele = dels[di].allele;
                queue[(size_t)l].push_back(q);
            }
            std::deque<QEnt>& q = queue[(size_t)l];                                       // IndelQueue::process
            while (!q.empty() && ((q.front().tid == tid && q.front().pos < (uint32_t)pos) || q.front().tid != tid)) q.pop_front();
            while (!q.empty() && q.front().tid == tid && q.front().pos == (uint32_t)pos) {
                if (!entry(q.front().allele.data(), q.front().allele.size(), q.front().st.i, q.front().st.f)) return false;
                extra += q.front().st.i[I_N];
                q.pop_front();
            }
            return true;
        };
        bool ok = true;
        if (!per_lib) ok = block_tokens(L1) && lib_tail(0);
        else {
            while (ok && p < L1) {                                 // "\tname\t{" six buckets, indel entries, "\t}"
                const char* t0 = p; ++p; const char* n0 = p; while (p < L1 && *p != '\t') ++p;
                int l = -1; for (int x = 0; x < Lp; ++x) if (e->libs[(size_t)x].size() == (size_t)(p - n0) && memcmp(e->libs[(size_t)x].data(), n0, (size_t)(p - n0)) == 0) { l = x; break; }
                p += 2;                                            // "\t{"
                if (l < 0 || p > L1) return fail(e, BRC_E_ARG, "device text: unknown library block");
                // the block's end: the "\t}" in front of the next block's "\tname\t{" or of the line's end (a closing brace is a token of its own)
                const char* be = p;
                for (;;) { const char* t = (const char*)memchr(be + 1, '\t', (size_t)(L1 - be - 1)); if (be + 2 == (t ? t : L1) && be[1] == '}') break; if (!t) return fail(e, BRC_E_ARG, "device text: a library block without its end"); be = t; }
                ok = put(t0, (size_t)(p - t0)) && block_tokens(be) && lib_tail(l) && put(be, 2);
                p = be + 2;
            }
        }
        if (!ok) return fail(e, BRC_E_NOMEM, "host allocation of the text buffers failed");
        while (di < dels.size() && dels[di].pos == pos) ++di;      // (deletion entries of libraries without a block cannot exist)
        if (!changed) continue;                                    // (nothing was committed: pbuf.n is unchanged)
        w0[wn++] = '\n';
        char dg[16]; const int nd = fmt_u32(dg, depth - dev_extra + extra);
        char* line = w0 + 10 - nd;                                 // prefix + depth right in front of the body
        memmove(line, L0, pre); memcpy(line + pre, dg, (size_t)nd);
        brc_engine::Patch pt; pt.k = k; pt.off = base + (size_t)(10 - nd); pt.len = wn - (size_t)(10 - nd);
        e->pbuf.n = base + wn;
        e->patches.push_back(pt);
    }
    // parts: device text between the rewritten lines; the lead position (index 0 when pos0 < beg0) is never printed
    e->part_ptr.clear(); e->part_len.clear();
    uint64_t cur = lead ? ht.off[1] : 0;
    for (const brc_engine::Patch& pt : e->patches) {
        if (pt.k == 0 && lead) continue;
        if (ht.off[pt.k] > cur) { e->part_ptr.push_back(ht.text + cur); e->part_len.push_back((size_t)(ht.total - cur) < (size_t)(ht.off[pt.k] - cur) ? (size_t)(ht.total - cur) : (size_t)(ht.off[pt.k] - cur)); }
        e->part_ptr.push_back(e->pbuf.p + pt.off); e->part_len.push_back(pt.len);
        cur = ht.off[pt.k + 1];
    }
    if (P > 0 && ht.total > cur) { e->part_ptr.push_back(ht.text + cur); e->part_len.push_back((size_t)(ht.total - cur)); }
    return BRC_OK;
}

static int format_chunks(brc_engine* e, const brc_result* r, const char* chrom, int64_t* n_chunks, unsigned* threads) {
    const int Lp = r->n_lib; const int64_t P = r->n_pos;
    if ((size_t)Lp != e->queue.size()) return fail(e, BRC_E_ARG, "result does not belong to this engine");
    if (r->ncol == NULL) return fail(e, BRC_E_ARG, "this result carries device text only");
    if (r->istat == NULL && (!e->text_only || r->ncol != e->hp.ncol)) return fail(e, BRC_E_ARG, "a text-only result can only be formatted before the next download");
    unsigned nthr = effective_cpus(); if (nthr > 64) nthr = 64;
    if (e->format_threads) nthr = e->format_threads;
    if (const char* t = test_knob(TK_FORMAT_THREADS)) { const int v = atoi(t); if (v > 0) nthr = (unsigned)v; }   // (test knob: wins over the option)
    // about four chunks per thread, 2048 .. 65536 positions each (a 1-Mbp piece in 64-Ki chunks keeps only 15 threads busy)
    int64_t CH = P / (4 * (int64_t)nthr);
    if (CH < 2048) CH = 2048;
    if (CH > (1 << 16)) CH = 1 << 16;
    if (const char* t = test_knob(TK_FORMAT_CHUNK)) { const long long v = atoll(t); if (v > 0) CH = v; }   // test knob
    // Chunks after the first start from the deletions their previous position queued: right as long as nothing older sits in
    // the queues.  An entry a previous region left pending for a position still ahead blocks everything queued behind it
    // (IndelQueue::process looks at the front only) — then the region is assembled in one piece, in order.
    for (const std::deque<QEnt>& q : e->queue) if (!q.empty()) { CH = std::max<int64_t>(P, 1); break; }
    const int64_t nch = std::max<int64_t>((P + CH - 1) / CH, 1);
    while (e->fparts.size() < (size_t)nch) e->fparts.emplace_back();
    std::vector<TextBuf>& parts = e->fparts;
    std::vector<std::vector<std::deque<QEnt> > > qs((size_t)nch);
    std::atomic<int> nomem(0);
    // a region that continues the previous one does not process its lead position again (BRC_OPT_CONTINUES_PREVIOUS)
    const int64_t kfirst = (e->continues && P > 0 && r->pos0 < r->beg0) ? 1 : 0;
    parallel_for(nch, nthr, [&](int64_t c) {
        const int64_t k0 = std::max<int64_t>(c * CH, c == 0 ? kfirst : 0), k1 = std::min<int64_t>(P, c * CH + CH);
        bool ok = true;
        try {
            if (c == 0) qs[0] = e->queue;
            else { qs[(size_t)c].assign((size_t)Lp, std::deque<QEnt>()); TextBuf scratch; ok = format_range(e, r, chrom, k0 - 1, k0, qs[(size_t)c], scratch, r->beg0, r->end, 0); }
            TextBuf& part = parts[(size_t)c];
            part.clear();
            ok = ok && format_range(e, r, chrom, k0, k1, qs[(size_t)c], part, r->beg0, r->end, 0);
        } catch (...) { ok = false; }
        if (!ok) nomem = 1;
    });
    if (nomem) return fail(e, BRC_E_NOMEM, "host allocation of the text buffers failed");
    e->queue = qs[(size_t)nch - 1];
    *n_chunks = nch; *threads = nthr;
    return BRC_OK;
}

int brc_format_region(brc_engine* e, const brc_result* r, const char* chrom, const char** text, size_t* text_len) {
    if (!e || !r || !chrom || !text) return BRC_E_ARG;
    const double t_in = now_s();
    if (r->ncol == NULL && e->text_result) {                       // device text: concatenate the pieces
        const int rc0 = format_device_text(e, r);
        if (rc0) return rc0;
        size_t total = 0; for (size_t v : e->part_len) total += v;
        if (total + 1 > e->tcap) {
            free(e->tbuf); e->tcap = total + total / 4 + 4096; e->tbuf = (char*)malloc(e->tcap);
            if (!e->tbuf) { e->tcap = 0; return fail(e, BRC_E_NOMEM, "host allocation of the text buffer failed"); }
        }
        size_t at = 0; for (size_t i = 0; i < e->part_len.size(); ++i) { memcpy(e->tbuf + at, e->part_ptr[i], e->part_len[i]); at += e->part_len[i]; }
        e->tbuf[total] = 0; e->tlen = total; *text = e->tbuf; if (text_len) *text_len = total;
        e->t_format += now_s() - t_in;
        return BRC_OK;
    }
    int64_t nch = 0; unsigned nthr = 1;
    const int rc = format_chunks(e, r, chrom, &nch, &nthr);
    if (rc) return rc;
    std::vector<TextBuf>& parts = e->fparts;
    // one contiguous text: every chunk is copied to its offset by the pool (the buffer keeps its capacity across calls)
    std::vector<size_t> off((size_t)nch + 1, 0);
    for (int64_t c = 0; c < nch; ++c) off[(size_t)c + 1] = off[(size_t)c] + parts[(size_t)c].n;
    const size_t total = off[(size_t)nch];
    if (total + 1 > e->tcap) {
        free(e->tbuf); e->tcap = total + total / 4 + 4096; e->tbuf = (char*)malloc(e->tcap);
        if (!e->tbuf) { e->tcap = 0; return fail(e, BRC_E_NOMEM, "host allocation of the text buffer failed"); }
    }
    parallel_for(nch, nthr, [&](int64_t c) { if (parts[(size_t)c].n) memcpy(e->tbuf + off[(size_t)c], parts[(size_t)c].p, parts[(size_t)c].n); });
    e->tbuf[total] = 0; e->tlen = total;
    *text = e->tbuf;
    if (text_len) *text_len = total;
    e->t_format += now_s() - t_in;
    return BRC_OK;
}

int brc_format_region_parts(brc_engine* e, const brc_result* r, const char* chrom, const char* const** parts, const size_t** part_lens, size_t* n_parts) {
    if (!e || !r || !chrom || !parts || !part_lens || !n_parts) return BRC_E_ARG;
    const double t_in = now_s();
    if (r->ncol == NULL && e->text_result) {
        const int rc0 = format_device_text(e, r);
        if (rc0) return rc0;
        *parts = e->part_ptr.data(); *part_lens = e->part_len.data(); *n_parts = e->part_ptr.size();
        e->t_format += now_s() - t_in;
        return BRC_OK;
    }
    int64_t nch = 0; unsigned nthr = 1;
    const int rc = format_chunks(e, r, chrom, &nch, &nthr);
    if (rc) return rc;
    e->part_ptr.resize((size_t)nch); e->part_len.resize((size_t)nch);
    for (int64_t c = 0; c < nch; ++c) { e->part_ptr[(size_t)c] = e->fparts[(size_t)c].p ? e->fparts[(size_t)c].p : ""; e->part_len[(size_t)c] = e->fparts[(size_t)c].n; }
    *parts = e->part_ptr.data(); *part_lens = e->part_len.data(); *n_parts = (size_t)nch;
    e->t_format += now_s() - t_in;
    return BRC_OK;
}

int brc_region_windows(brc_engine* e, const int32_t* vbeg0, const int32_t* vend, int64_t n) {
    if (!e || n < 0 || (n > 0 && (!vbeg0 || !vend))) return BRC_E_ARG;
    if (e->state != 1) return fail(e, BRC_E_ARG, "brc_region_windows: between brc_begin_region and brc_end_region");
    for (int64_t i = 0; i < n; ++i) if (vend[i] < vbeg0[i]) return fail(e, BRC_E_ARG, "brc_region_windows: a window ends before it begins");
    try { e->st.win_beg.assign(vbeg0, vbeg0 + n); e->st.win_end.assign(vend, vend + n); } catch (...) { e->st.win_beg.clear(); e->st.win_end.clear(); return fail(e, BRC_E_NOMEM, "host allocation failed"); }
    return BRC_OK;
}

int brc_format_window(brc_engine* e, const brc_result* r, const char* chrom, int32_t vbeg0, int32_t vend, int32_t delta,
                      const char** text, size_t* text_len) {
    if (!e || !r || !chrom || !text) return BRC_E_ARG;
    if (vend < vbeg0) return fail(e, BRC_E_ARG, "brc_format_window: the window ends before it begins");
    if ((size_t)r->n_lib != e->queue.size()) return fail(e, BRC_E_ARG, "result does not belong to this engine");
    if (r->ncol == NULL) return fail(e, BRC_E_ARG, "brc_format_window needs planes: switch BRC_OPT_DEVICE_TEXT off for regions cut into windows");
    if (r->istat == NULL && (!e->text_only || r->ncol != e->hp.ncol)) return fail(e, BRC_E_ARG, "a text-only result can only be formatted before the next download");
    TextBuf& out = e->wbuf; out.clear();
    // plane indices of [vbeg0 - 1, vend) clipped to the planes; the lead position only feeds the deletion queue (:269 vs :414)
    int64_t k0 = (int64_t)vbeg0 - 1 - r->pos0, k1 = (int64_t)vend - r->pos0;
    if (k0 < 0) k0 = 0;
    if (k1 > r->n_pos) k1 = r->n_pos;
    if (k1 > k0) {
        bool ok = true;
        try { std::vector<std::deque<QEnt> > q((size_t)r->n_lib); ok = format_range(e, r, chrom, k0, k1, q, out, vbeg0, vend, delta); } catch (...) { ok = false; }
        if (!ok) return fail(e, BRC_E_NOMEM, "host allocation of the text buffer failed");
    }
    char* z = out.room(1);
    if (!z) return fail(e, BRC_E_NOMEM, "host allocation of the text buffer failed");
    *z = 0;
    *text = out.p;
    if (text_len) *text_len = out.n;
    return BRC_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- the stderr side (ReadWarnings)

namespace {
struct WEv { int qpos, indel; bool in_col, is_del; };
// htslib's resolve_cigar2 as a pure function of (read, position): what the pileup entry of read r at p looks like
WEv resolve_at(const uint32_t* cig, uint32_t nc, int32_t pos, int32_t p) {
    WEv e; e.qpos = 0; e.indel = 0; e.in_col = false; e.is_del = false;
    if (has_empty_mop(cig, nc)) {        // (an empty M / = / X operator: the iterator's own cursor, column by column from the read's start — brc_core.h)
        CigCursor s; s.k = -1; s.x = pos; s.y = 0;
        for (int32_t col = pos; col <= p; ++col) { int q = 0, ind = 0; bool del = false; const bool in = cursor_resolve(cig, nc, pos, col, s, q, del, ind); if (col == p) { e.in_col = in; e.is_del = del; e.qpos = q; e.indel = ind; } }
        return e;
    }
    int32_t x = pos; int y = 0;
    for (uint32_t k = 0; k < nc; ++k) {
        const uint32_t op = cig[k] & 0xfu; const int len = (int)(cig[k] >> 4);
        if (is_refop(op)) {
            if (p >= x && p < x + len) {
                const bool m = is_mop(op);
                e.in_col = true; e.is_del = !m; e.qpos = m ? y + (p - x) : y;
                if (p == x + len - 1 && k + 1 < nc) {
                    const uint32_t op2 = cig[k + 1] & 0xfu; const int l2 = (int)(cig[k + 1] >> 4);
                    if (op2 == CDEL) e.indel = -l2;
                    else if (op2 == CINS) e.indel = l2;
                    else if (peek_insertion(cig, nc, k)) { int l3 = 0; for (uint32_t kk = k + 2; kk < nc; ++kk) { const uint32_t o = cig[kk] & 0xfu; if (o == CINS) l3 += (int)(cig[kk] >> 4); else if (is_refop(o)) break; } e.indel = l3; }
                }
                return e;
            }
            x += len;
            if (is_mop(op)) y += len;
        } else if (op == CINS || op == CSOFT_CLIP) y += len;
    }
    return e;
}
}  // namespace

extern "C" {

// The events ReadWarnings::warn / fetch_func's fprintf would see, in the reference's order (see include/brc.h).  Only reads
// that can warn are walked: library-less reads (-p), reads without NM, proper pairs without SM, and — with a site list — reads
// hanging over the end of the reference.  They are swept position by position exactly like the pileup would present them
// (file order inside a column; a library-less read ends its position, :281-284); a read's "Request for position" lines
// come when fetch_func sees it, i.e. after the positions left of the previous pushed read.
static int warnings_impl(brc_engine* e, const char* chrom, int64_t wbeg0, int64_t wend, bool with_bounds, int64_t cap, const char** events, size_t* events_len) {
    if (!e || !chrom || !events) return BRC_E_ARG;
    if (e->state < 1) return fail(e, BRC_E_ARG, "brc_region_warnings needs a region");
    const Staged& s = e->st; const Geometry& g = e->g; const brc_config& cfg = e->cfg;
    std::string& out = e->wev; out.clear();
    const int64_t n = s.n;
    auto name_of = [&](int64_t i) -> const char* { return s.qname_off.p[i] == ~0ull ? "?" : s.qnames.p + s.qname_off.p[i]; };
    struct Rel { int64_t i; int32_t pos, end; bool lib_less, want_s, want_n, warns, bounds; int32_t barrier_q; };
    std::vector<Rel> rel;
    int32_t prev_enter_pos = INT32_MIN;
    int64_t i_first = 0;
    if (!with_bounds) i_first = std::lower_bound(s.pos.p, s.pos.p + n, (int32_t)std::max<int64_t>(wbeg0 - 1 - s.max_span, INT32_MIN)) - s.pos.p;   // window mode: reads sorted by pos
    for (int64_t i = i_first; i < n; ++i) {
        const uint32_t fl = s.flag.p[i]; const uint32_t nc = s.n_cigar.p[i]; const uint32_t* cg = s.cigar.p + s.cig_off.p[i];
        const int32_t pos = s.pos.p[i];
        const bool enters = read_enters(fl, cg, nc) && pos >= 0;
        Rel r; r.i = i; r.pos = pos; r.end = pos; r.barrier_q = prev_enter_pos;
        r.lib_less = cfg.per_lib && s.lib.p[i] < 0;
        r.want_s = (fl & FPROPER_PAIR) && !(s.tags.p[i] & BRC_TAG_SM);
        r.want_n = !(s.tags.p[i] & BRC_TAG_NM);
        r.warns = enters && (r.lib_less || r.want_s || r.want_n);
        if (!with_bounds && (pos >= wend)) break;                    // window mode: nothing right of the window matters
        r.bounds = false;
        int32_t rlen = 0, x = pos;
        for (uint32_t k = 0; k < nc; ++k) {
            const uint32_t op = cg[k] & 0xfu; const int32_t len = (int32_t)(cg[k] >> 4);
            if (is_refop(op)) rlen += len;
            if (op == CMATCH) { if (with_bounds && cfg.ref_len_check && g.ref && g.ref_len && (int64_t)x + len - 1 > g.ref_len) r.bounds = true; x += len; }
            else if (op == CDEL || op == CREF_SKIP) x += len;
        }
        r.end = pos + rlen;
        if ((r.warns && r.end > wbeg0 - 1 && r.pos < wend) || r.bounds) rel.push_back(r);
        if (enters) prev_enter_pos = pos;
    }
    int64_t listed[BRC_N_WARN] = {0, 0, 0, 0};
    auto emit = [&](int type, int64_t i) {
        if (cap >= 0 && listed[type] >= cap) return;
        listed[type]++;
        out.push_back("SNZL"[type]); out.push_back('\t'); out.append(name_of(i)); out.push_back('\n');
    };
    auto capped = [&]() { return cap >= 0 && listed[BRC_W_SM_MISSING] >= cap && listed[BRC_W_NM_MISSING] >= cap && (!cfg.per_lib || listed[BRC_W_LIB_UNAVAILABLE] >= cap); };
    // fetch_func's own lines of read r (:139-148): every M base whose reference position lies beyond the contig
    auto bounds_lines = [&](const Rel& r) {
        const uint32_t nc = s.n_cigar.p[r.i]; const uint32_t* cg = s.cigar.p + s.cig_off.p[r.i];
        int64_t x = r.pos; char t[320];
        for (uint32_t k = 0; k < nc; ++k) {
            const uint32_t op = cg[k] & 0xfu; const int32_t len = (int32_t)(cg[k] >> 4);
            if (op == CMATCH) {
                bool stopped = false;
                for (int32_t j = 0; j < len; ++j) {
                    const int64_t refpos = x + j;
                    if (refpos > g.ref_len) { const int m = snprintf(t, sizeof t, "B\tWARNING: Request for position %d in sequence %s is > length of %d!\n", (int)refpos, chrom, (int)g.ref_len); out.append(t, (size_t)m); continue; }
                    if (refpos == g.ref_len || refpos < 0 || g.ref[refpos] == 0) { stopped = true; break; }       // :151 the terminating NUL ends the walk
                }
                if (stopped) return;
                x += len;
            } else if (op == CDEL || op == CREF_SKIP) x += len;
        }
    };
    // sweep
    std::vector<size_t> active;      // indices into rel, file order
    size_t next_add = 0;             // next warning read to add to the active list
    int64_t cur = INT64_MIN;         // next position to present
    auto sweep_to = [&](int64_t limit) {    // present every position < limit
        for (;;) {
            if (capped()) return;
            // next position >= cur covered by an active read or by a read not yet added
            size_t w = 0; int64_t nextp = INT64_MAX;
            for (size_t a : active) { if (rel[a].end > cur) { active[w++] = a; nextp = std::min<int64_t>(nextp, std::max<int64_t>(cur, rel[a].pos)); } }
            active.resize(w);
            size_t na = next_add; while (na < rel.size() && !rel[na].warns) ++na;
            if (na < rel.size()) nextp = std::min<int64_t>(nextp, std::max<int64_t>(cur, rel[na].pos));
            if (nextp == INT64_MAX || nextp >= limit) { cur = std::max(cur, std::min<int64_t>(limit, nextp == INT64_MAX ? limit : nextp)); return; }
            const int64_t p = nextp;
            while (next_add < rel.size() && (!rel[next_add].warns || rel[next_add].pos <= p)) { if (rel[next_add].warns) active.push_back(next_add); ++next_add; }
            if (p >= wbeg0 - 1 && p < wend) {                                                               // :269
                for (size_t a : active) {
                    const Rel& r = rel[a];
                    if (!(r.pos <= p && p < r.end)) continue;
                    if (r.lib_less) { emit(BRC_W_LIB_UNAVAILABLE, r.i); break; }                            // :281-284
                    const WEv ev = resolve_at(s.cigar.p + s.cig_off.p[r.i], s.n_cigar.p[r.i], r.pos, (int32_t)p);
                    if (!ev.in_col || ev.is_del) continue;
                    if ((int)s.mapq.p[r.i] < cfg.min_mapq || ev.qpos >= s.l_qseq.p[r.i] || (int)s.qual_at(s.qual_off.p[r.i] + (uint64_t)ev.qpos)[0] < cfg.min_bq) continue;   // :288
                    if (s.flag.p[r.i] & BRC_NOCOUNT_MASK) continue;                                         // :295-310
                    const int calls = ((ev.indel != 0 && g.ref) ? 1 : 0) + ((ev.indel < 1 || !cfg.insertion_centric) ? 1 : 0);   // :315-346
                    for (int c2 = 0; c2 < calls; ++c2) {                                                    // BasicStat::process_read
                        if (r.want_s) emit(BRC_W_SM_MISSING, r.i);
                        if (r.want_n) emit(BRC_W_NM_MISSING, r.i);
                    }
                }
            }
            cur = p + 1;
        }
    };
    for (const Rel& r : rel) {
        if (!r.bounds) continue;
        if (r.barrier_q != INT32_MIN) sweep_to(r.barrier_q);          // positions left of the previous pushed read came out before fetch_func saw this one
        bounds_lines(r);
    }
    cur = std::max<int64_t>(cur, wbeg0 - 1);
    sweep_to(wend);
    *events = out.c_str();
    if (events_len) *events_len = out.size();
    return BRC_OK;
}

int brc_region_warnings(brc_engine* e, const char* chrom, int64_t cap, const char** events, size_t* events_len) {
    if (!e) return BRC_E_ARG;
    // (BRC_OPT_CONTINUES_PREVIOUS: the lead position was the last position of the region before this one, which has
    // reported that position's events already)
    return warnings_impl(e, chrom, (int64_t)e->g.beg0 + (e->warn_skip_lead ? 1 : 0), e->g.end, true, cap, events, events_len);
}
int brc_window_warnings(brc_engine* e, int32_t vbeg0, int32_t vend, int64_t cap, const char** events, size_t* events_len) {
    return warnings_impl(e, "", vbeg0, vend, false, cap, events, events_len);
}

// ReadWarnings::warn (ReadWarnings.hpp:39-50) applied to a tagged event stream
int brc_warnings_text(brc_engine* e, const char* ev, size_t n, int64_t max, int64_t* counts, const char** text, size_t* text_len) {
    if (!e || !text || !counts || (!ev && n)) return BRC_E_ARG;
    static const char* const kMsg[BRC_N_WARN] = {
        "Couldn't find single-end mapping quality. Check to see if the SM tag is in BAM.",
        "Couldn't find number of mismatches. Check to see if the NM tag is in BAM.",
        "Couldn't find the generated tag.",
        "Library unavailable. Check to make sure the LB tag is present in the @RG entries of the header."};
    std::string& out = e->wtext; out.clear();
    size_t i = 0;
    while (i < n) {
        size_t j = i; while (j < n && ev[j] != '\n') ++j;
        const char* tp = strchr("SNZL", ev[i]);
        if (ev[i] == 'B') { if (j > i + 2) out.append(ev + i + 2, j - (i + 2)); out.push_back('\n'); }
        else if (tp && *tp && j >= i + 2) {
            const int t = (int)(tp - "SNZL");
            ++counts[t];
            if (!(max >= 0 && counts[t] > max)) {
                out.append("WARNING: In read "); out.append(ev + i + 2, j - (i + 2)); out.append(": "); out.append(kMsg[t]); out.push_back('\n');
                if (max >= 0 && counts[t] == max) { out.append("The previous warning has been emitted "); out.append(std::to_string((long long)counts[t])); out.append(" times and will be disabled.\n"); }
            }
        }
        i = j + 1;
    }
    *text = out.c_str();
    if (text_len) *text_len = out.size();
    return BRC_OK;
}

}  // extern "C"
