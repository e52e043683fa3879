// brc_engine.hip — the HIP/CDNA4 (gfx950) device pipeline behind the C-ABI of include/brc.h.
//
// Kernels (per region, all on one engine-owned stream; see DESIGN.md for layouts and rooflines):
//   k_annotate      one lane per read: fetch_func's Zm integers + per-read constants -> 80-B DRead records;
//                   also counts each read's indel events per (position, library) key
//   k_scan_*        3-phase scans: inclusive running max of read ends (tile lower bounds), exclusive sum of
//                   indel-event counts (per-key offsets)
//   k_tiles         one lane per 64-position tile: [lo,hi) read range by binary search
//   k_pileup        THE hot kernel: one wave per (tile, library), lane == reference position, wave-uniform walk
//                   over the tile's reads in file order (read records in SGPRs, coalesced byte loads of QUAL/SEQ
//                   along the lanes), 6 buckets x 12 accumulators in VGPRs, order-preserving fp32 sums, coalesced
//                   256-B plane stores.  Integer/byte work, HBM-bound: no MFMA by design.
//   k_count_pos     emitted-position count
//   k_indel_fill / k_indel_reduce   indel side path (<1 % of events): keyed fill, ordered per-key reduction
//
// There is no CPU fallback here: without a HIP device make_backend() fails with BRC_E_NODEVICE.
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "brc_host.h"

namespace brc {

struct Counters {
    unsigned long long n_events, n_positions, w_sm, w_nm, w_lib;
    unsigned int n_indel_slots, pad;
};

// ---------------------------------------------------------------- wave helpers (wave64)

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------- K1

__global__ __launch_bounds__(256) void k_annotate(DevCfg c, DevIn in, DRead* __restrict__ reads, int32_t* __restrict__ ends,
                                                  uint16_t* __restrict__ bq, uint32_t* __restrict__ indel_cnt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.n_reads) return;
    const DRead r = annotate_read(c, in, i, bq);
    reads[i] = r;
    ends[i] = r.end;
    if (indel_cnt) {
        const int lib = (int)(r.misc >> 16) - 1;
        enumerate_indels(c, in, r, [&](int32_t p, int, int) {
            atomicAdd(&indel_cnt[(int64_t)(p - c.pos0) * c.Lp + lib], 1u);
        });
    }
}

// ---------------------------------------------------------------- scans (3-phase: block aggregates, scan of aggregates, apply)

enum { SCAN_T = 256, SCAN_ITEMS = 16, SCAN_CHUNK = SCAN_T * SCAN_ITEMS };

struct OpMaxI32 { typedef int32_t T; static __device__ __forceinline__ T id() { return INT32_MIN; } static __device__ __forceinline__ T op(T a, T b) { return a > b ? a : b; } };
struct OpSumU32 { typedef uint32_t T; static __device__ __forceinline__ T id() { return 0u; } static __device__ __forceinline__ T op(T a, T b) { return a + b; } };

template <class Op>
__device__ __forceinline__ typename Op::T block_reduce(typename Op::T v, typename Op::T* sh) {
    typedef typename Op::T T;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = Op::op(v, (T)__shfl_xor(v, o, 64));
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    T r = Op::id();
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) r = Op::op(r, sh[k]);
    __syncthreads();
    return r;
}

template <class Op>
__global__ __launch_bounds__(SCAN_T) void k_scan_reduce(const typename Op::T* __restrict__ in, int64_t n, typename Op::T* __restrict__ agg) {
    typedef typename Op::T T;
    __shared__ T sh[SCAN_T / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK;
    T v = Op::id();
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        const int64_t idx = base + (int64_t)j * SCAN_T + threadIdx.x;
        if (idx < n) v = Op::op(v, in[idx]);
    }
    v = block_reduce<Op>(v, sh);
    if (threadIdx.x == 0) agg[blockIdx.x] = v;
}

// single block: exclusive scan of the block aggregates, in place
template <class Op>
__global__ __launch_bounds__(1024) void k_scan_aggregates(typename Op::T* __restrict__ agg, int64_t nb) {
    typedef typename Op::T T;
    __shared__ T sh[1024];
    __shared__ T carry;
    if (threadIdx.x == 0) carry = Op::id();
    __syncthreads();
    for (int64_t base = 0; base < nb; base += 1024) {
        const int64_t idx = base + threadIdx.x;
        const T v = idx < nb ? agg[idx] : Op::id();
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {          // Hillis-Steele inclusive scan
            T t = sh[threadIdx.x];
            if ((int)threadIdx.x >= o) t = Op::op(sh[threadIdx.x - o], t);
            __syncthreads();
            sh[threadIdx.x] = t;
            __syncthreads();
        }
        const T incl = sh[threadIdx.x];
        const T excl = threadIdx.x ? sh[threadIdx.x - 1] : Op::id();
        const T c0 = carry;
        __syncthreads();
        if (idx < nb) agg[idx] = Op::op(c0, excl);
        if (threadIdx.x == 1023) carry = Op::op(c0, incl);
        __syncthreads();
    }
}

// each thread owns SCAN_ITEMS consecutive elements (blocked arrangement) so the scan is a serial pass per thread
// plus one block-level scan of the thread totals
template <class Op, bool INCLUSIVE>
__global__ __launch_bounds__(SCAN_T) void k_scan_apply(const typename Op::T* __restrict__ in, typename Op::T* __restrict__ out, int64_t n,
                                                       const typename Op::T* __restrict__ agg) {
    typedef typename Op::T T;
    __shared__ T sh[SCAN_T];
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
    T v[SCAN_ITEMS];
    T tot = Op::id();
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) { v[j] = (base + j < n) ? in[base + j] : Op::id(); tot = Op::op(tot, v[j]); }
    sh[threadIdx.x] = tot;
    __syncthreads();
    for (int o = 1; o < SCAN_T; o <<= 1) {
        T t = sh[threadIdx.x];
        if ((int)threadIdx.x >= o) t = Op::op(sh[threadIdx.x - o], t);
        __syncthreads();
        sh[threadIdx.x] = t;
        __syncthreads();
    }
    T run = Op::op(agg[blockIdx.x], threadIdx.x ? sh[threadIdx.x - 1] : Op::id());
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        if (base + j < n) {
            if (INCLUSIVE) { run = Op::op(run, v[j]); out[base + j] = run; }
            else { out[base + j] = run; run = Op::op(run, v[j]); }
        }
    }
}

// ---------------------------------------------------------------- tiles

__global__ __launch_bounds__(256) void k_tiles(DevCfg c, const int32_t* __restrict__ prefmax, const DRead* __restrict__ reads,
                                               int64_t ntiles, uint2* __restrict__ rng) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    uint32_t lo, hi;
    tile_range(c, prefmax, reads, t, lo, hi);
    rng[t] = make_uint2(lo, hi);
}

// ---------------------------------------------------------------- KB: pileup + BasicStat accumulation (the hot kernel)

enum { PILEUP_WAVES = 4 };   // 256 threads: 4 consecutive tiles (256 positions) per workgroup

__global__ __launch_bounds__(PILEUP_WAVES * 64) void k_pileup(DevCfg c, DevIn in, const DRead* __restrict__ reads,
                                                              const uint2* __restrict__ rng, int64_t ntiles, Planes pl,
                                                              uint4* __restrict__ tile_ctr) {
    // XCD-aware mapping: workgroup b runs on XCD b % 8 (observed dispatch order); give every XCD one contiguous
    // run of tiles so neighbouring tiles, which share most of their reads, hit the same 4-MiB L2.
    const uint32_t nb = gridDim.x;            // multiple of 8
    const uint32_t per = nb >> 3;
    const uint32_t wg = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)wg * PILEUP_WAVES + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const int lib = blockIdx.y;
    const uint2 r2 = rng[tile];
    const uint32_t lo = __builtin_amdgcn_readfirstlane(r2.x), hi = __builtin_amdgcn_readfirstlane(r2.y);
    const int64_t k = tile * TILE + lane;
    const bool valid = k < c.P;
    const int32_t p = (int32_t)(c.pos0 + k);

    LaneAcc a;
    lane_init(a);
    // Software pipeline over the tile's reads (file order): while read r is accumulated, the event word of read r+1
    // is in flight (vector load) and the record of read r+2 is in flight (scalar load).
    if (lo < hi) {
        const uint32_t libsel = (uint32_t)lib + 1u;
        DRead r0 = reads[lo];
        DRead r1 = reads[lo + 1 < hi ? lo + 1 : lo];
        Probe p0 = lane_probe(c, in, r0, lo, libsel, p, valid, a);
        uint32_t v0 = p0.want ? (uint32_t)in.bq[r0.bq_off + (uint64_t)p0.qpos] : 0u;
        for (uint32_t r = lo; r < hi; ++r) {
            const DRead r2 = reads[r + 2 < hi ? r + 2 : hi - 1];                    // stage A: record r+2 (uniform address)
            Probe p1; p1.qpos = 0; p1.indel = 0; p1.want = false;
            uint32_t v1 = 0u;
            if (r + 1 < hi) {                                                        // stage B: probe + event load of r+1
                p1 = lane_probe(c, in, r1, r + 1, libsel, p, valid, a);
                v1 = p1.want ? (uint32_t)in.bq[r1.bq_off + (uint64_t)p1.qpos] : 0u;
            }
            lane_accumulate(c, r0, p0, v0, a);                                       // stage C: accumulate r
            r0 = r1; r1 = r2; p0 = p1; v0 = v1;
        }
    }
    if (valid) lane_store(c, pl, lib, k, a);

    const bool dead = c.per_lib && a.unavail != NONE32;
    const bool live = valid && !dead;
    unsigned long long ev = (live && p >= c.beg0) ? a.ncol : 0u;
    unsigned long long wsm = live ? a.w_sm : 0u, wnm = live ? a.w_nm : 0u, wl = (valid && dead && lib == 0) ? 1u : 0u;
    ev = wave_sum_u64(ev); wsm = wave_sum_u64(wsm); wnm = wave_sum_u64(wnm); wl = wave_sum_u64(wl);
    // per-(tile, library) partials; k_finalize sums them (a single-address atomic per wave costs ~12 ns x 780 k waves)
    if (lane == 0) tile_ctr[(int64_t)lib * ntiles + tile] = make_uint4((uint32_t)ev, (uint32_t)wsm, (uint32_t)wnm, (uint32_t)wl);
}

template <int NV>
__device__ __forceinline__ void block_sum_u64(unsigned long long (&v)[NV], unsigned long long* sh /* [4*NV] */) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum_u64(v[i]);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) for (int i = 0; i < NV; ++i) sh[w * NV + i] = v[i];
    __syncthreads();
    if (threadIdx.x == 0) for (int i = 0; i < NV; ++i) { unsigned long long t = 0; for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += sh[k * NV + i]; v[i] = t; }
}

// emitted-position count + sum of the per-tile partials; grid-stride, one set of atomics per work-group
__global__ __launch_bounds__(256) void k_finalize(DevCfg c, const uint32_t* __restrict__ ncol, const uint4* __restrict__ tile_ctr,
                                                  int64_t n_tile_ctr, Counters* __restrict__ ctr) {
    __shared__ unsigned long long sh[4 * 5];
    unsigned long long v[5] = {0, 0, 0, 0, 0};   // positions, events, w_sm, w_nm, w_lib
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < c.P; k += stride) {
        if (c.pos0 + k < c.beg0) continue;
        uint32_t tot = 0;
        for (int l = 0; l < c.Lp; ++l) tot += ncol[(int64_t)l * c.PS + k];
        v[0] += tot ? 1u : 0u;
    }
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_tile_ctr; t += stride) {
        const uint4 x = tile_ctr[t];
        v[1] += x.x; v[2] += x.y; v[3] += x.z; v[4] += x.w;
    }
    block_sum_u64<5>(v, sh);
    if (threadIdx.x == 0) {
        if (v[0]) atomicAdd(&ctr->n_positions, v[0]);
        if (v[1]) atomicAdd(&ctr->n_events, v[1]);
        if (v[2]) atomicAdd(&ctr->w_sm, v[2]);
        if (v[3]) atomicAdd(&ctr->w_nm, v[3]);
        if (v[4]) atomicAdd(&ctr->w_lib, v[4]);
    }
}

// ---------------------------------------------------------------- indel side path

__global__ __launch_bounds__(256) void k_indel_fill(DevCfg c, DevIn in, const DRead* __restrict__ reads, uint32_t* __restrict__ cursor,
                                                    IndelEv* __restrict__ ev) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.n_reads) return;
    const DRead r = reads[i];
    const int lib = (int)(r.misc >> 16) - 1;
    enumerate_indels(c, in, r, [&](int32_t p, int qpos, int len) {
        const uint32_t slot = atomicAdd(&cursor[(int64_t)(p - c.pos0) * c.Lp + lib], 1u);
        IndelEv e; e.read = (uint32_t)i; e.qpos = qpos; e.len = len; e.key_lo = 0;
        ev[slot] = e;
    });
}

// cursor[key] has been advanced by k_indel_fill to the END of the key's events.  A key's reduced alleles are written to
// out[start .. start+na) where [start, start+n) is the key's own event range (no slot atomics); unused slots get len = 0.
__global__ __launch_bounds__(256) void k_indel_reduce(DevCfg c, DevIn in, const DRead* __restrict__ reads, const uint32_t* __restrict__ cnt,
                                                      const uint32_t* __restrict__ cursor, IndelEv* __restrict__ ev,
                                                      const uint32_t* __restrict__ unavail, IndelOut* __restrict__ out,
                                                      Counters* __restrict__ ctr) {
    __shared__ unsigned long long sh[4 * 2];
    unsigned long long w[2] = {0, 0};
    const int64_t nkeys = c.P * c.Lp;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t key = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; key < nkeys; key += stride) {
        const int n = (int)cnt[key];
        if (key == nkeys - 1) ctr->n_indel_slots = cursor[key];        // total number of event slots
        if (n == 0) continue;
        const int64_t k = key / c.Lp; const int lib = (int)(key % c.Lp);
        const uint32_t start = cursor[key] - (uint32_t)n;
        int na = 0;
        if (!(c.per_lib && unavail[k] != NONE32)) {                     // else: position abandoned (bamreadcount.cpp:281-284)
            uint32_t wsm = 0, wnm = 0;
            na = reduce_indel_key(c, in, reads, ev + start, n, (int32_t)(c.pos0 + k), lib, out + start, wsm, wnm);
            w[0] += wsm; w[1] += wnm;
        }
        for (int j = na; j < n; ++j) out[start + j].len = 0;
    }
    block_sum_u64<2>(w, sh);
    if (threadIdx.x == 0) {
        if (w[0]) atomicAdd(&ctr->w_sm, w[0]);
        if (w[1]) atomicAdd(&ctr->w_nm, w[1]);
    }
}

// ---------------------------------------------------------------- host side of the backend

static void* pinned_alloc(size_t n) {
    void* p = nullptr;
    if (hipHostMalloc(&p, n ? n : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
static void pinned_release(void* p) { (void)hipHostFree(p); }
static const HostAlloc kPinned = {pinned_alloc, pinned_release};

struct DBuf {
    void* p = nullptr; size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { e = hipMalloc(&p, bytes); want = bytes; }
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

enum { T_ANNOTATE = 0, T_SCAN_ENDS, T_TILES, T_PILEUP, T_COUNT, T_INDEL_SCAN, T_INDEL_FILL, T_INDEL_REDUCE, T_N };
static const char* kKernelNames[BRC_NKERNEL] = {"k_annotate", "k_scan_ends", "k_tiles", "k_pileup", "k_finalize",
                                                "k_scan_indel", "k_indel_fill", "k_indel_reduce"};

#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { return hip_fail(_e, #x); } } while (0)

class HipBackend : public Backend {
    std::string err;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t evt[T_N + 1];
    bool have_events = false;
    DevCfg c; DevIn in;
    int64_t ntiles = 0; uint64_t n_indel_cap = 0;
    // device buffers
    DBuf d_pos, d_flag, d_mapq, d_lib, d_lq, d_nc, d_co, d_so, d_qo, d_nm, d_sm, d_tags, d_cigar, d_seq, d_qual, d_ref;
    DBuf d_bq, d_reads, d_ends, d_prefmax, d_agg, d_rng, d_ncol, d_depth, d_istat, d_fstat, d_unavail, d_cnt, d_cursor, d_ev, d_iout, d_ctr, d_tilectr;
    // host result buffers (pinned)
    HBuf<uint32_t> h_ncol, h_depth, h_istat, h_unavail; HBuf<float> h_fstat; HBuf<IndelOut> h_iout;
    std::vector<IndelOut> iout_compact;
    Counters h_ctr;
    bool computed = false;

    int hip_fail(hipError_t e, const char* what) {
        char b[512];
        snprintf(b, sizeof b, "%s: %s", what, hipGetErrorString(e));
        err = b;
        return e == hipErrorOutOfMemory ? BRC_E_NOMEM : BRC_E_HIP;
    }

  public:
    int init(int dev) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { err = "no HIP device visible"; return BRC_E_NODEVICE; }
        if (dev < 0 || dev >= n) { err = "device ordinal out of range"; return BRC_E_NODEVICE; }
        device = dev;
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        for (int i = 0; i <= T_N; ++i) HIPCHK(hipEventCreate(&evt[i]));
        have_events = true;
        h_ncol.A = h_depth.A = h_istat.A = h_unavail.A = &kPinned; h_fstat.A = &kPinned; h_iout.A = &kPinned;
        return BRC_OK;
    }
    ~HipBackend() override {
        (void)hipSetDevice(device);
        DBuf* all[] = {&d_pos, &d_flag, &d_mapq, &d_lib, &d_lq, &d_nc, &d_co, &d_so, &d_qo, &d_nm, &d_sm, &d_tags, &d_cigar, &d_seq, &d_qual,
                       &d_ref, &d_bq, &d_reads, &d_ends, &d_prefmax, &d_agg, &d_rng, &d_ncol, &d_depth, &d_istat, &d_fstat, &d_unavail, &d_cnt,
                       &d_cursor, &d_ev, &d_iout, &d_ctr, &d_tilectr};
        for (DBuf* b : all) b->release();
        h_ncol.destroy(); h_depth.destroy(); h_istat.destroy(); h_unavail.destroy(); h_fstat.destroy(); h_iout.destroy();
        if (have_events) for (int i = 0; i <= T_N; ++i) (void)hipEventDestroy(evt[i]);
        if (stream) (void)hipStreamDestroy(stream);
    }
    const HostAlloc* host_alloc() override { return &kPinned; }
    const char* last_error() const override { return err.c_str(); }

    template <class T>
    int up(DBuf& d, const HBuf<T>& h, size_t n) {
        HIPCHK(d.ensure((n + 16) * sizeof(T)));
        if (n) HIPCHK(hipMemcpyAsync(d.p, h.p, n * sizeof(T), hipMemcpyHostToDevice, stream));
        return BRC_OK;
    }

    int upload(const brc_config& cfg, const Staged& s, Geometry& g) override {
        HIPCHK(hipSetDevice(device));
        computed = false;
        memset(&c, 0, sizeof c);
        c.min_mapq = cfg.min_mapq; c.min_bq = cfg.min_bq; c.per_lib = cfg.per_lib; c.insertion_centric = cfg.insertion_centric;
        c.Lp = g.Lp; c.ref_len_check = cfg.ref_len_check; c.has_ref = g.ref != nullptr;
        g.PS = (g.P + 63) & ~(int64_t)63;
        c.beg0 = g.beg0; c.end = g.end; c.pos0 = g.pos0; c.P = g.P; c.PS = g.PS; c.ref_lo = g.ref_lo; c.ref_hi = g.ref_hi; c.ref_len = g.ref_len;
        c.n_reads = s.n;
        const size_t n = (size_t)s.n;
        int rc;
        if ((rc = up(d_pos, s.pos, n)) || (rc = up(d_flag, s.flag, n)) || (rc = up(d_mapq, s.mapq, n)) || (rc = up(d_lib, s.lib, n)) ||
            (rc = up(d_lq, s.l_qseq, n)) || (rc = up(d_nc, s.n_cigar, n)) || (rc = up(d_co, s.cig_off, n)) || (rc = up(d_so, s.seq_off, n)) ||
            (rc = up(d_qo, s.qual_off, n)) || (rc = up(d_nm, s.nm, n)) || (rc = up(d_sm, s.sm, n)) || (rc = up(d_tags, s.tags, n)) ||
            (rc = up(d_cigar, s.cigar, s.cigar.n)) || (rc = up(d_seq, s.seq4, s.seq4.n)) || (rc = up(d_qual, s.qual, s.qual.n)))
            return rc;
        const size_t rl = (size_t)(g.ref_hi - g.ref_lo);
        HIPCHK(d_ref.ensure(rl + 16));
        if (rl) HIPCHK(hipMemcpyAsync(d_ref.p, g.ref + g.ref_lo, rl, hipMemcpyHostToDevice, stream));
        in.pos = (const int32_t*)d_pos.p; in.flag = (const uint16_t*)d_flag.p; in.mapq = (const uint8_t*)d_mapq.p; in.lib = (const int16_t*)d_lib.p;
        in.l_qseq = (const int32_t*)d_lq.p; in.n_cigar = (const uint32_t*)d_nc.p; in.cig_off = (const uint64_t*)d_co.p;
        in.seq_off = (const uint64_t*)d_so.p; in.qual_off = (const uint64_t*)d_qo.p; in.nm = (const int32_t*)d_nm.p; in.sm = (const int32_t*)d_sm.p;
        in.tags = (const uint8_t*)d_tags.p; in.cigar = (const uint32_t*)d_cigar.p; in.seq4 = (const uint8_t*)d_seq.p; in.qual = (const uint8_t*)d_qual.p;
        in.ref = (const char*)d_ref.p;
        HIPCHK(d_bq.ensure((s.qual.n + 16) * sizeof(uint16_t)));
        in.bq = (const uint16_t*)d_bq.p;
        // outputs / scratch
        const size_t P = (size_t)c.PS, Lp = (size_t)c.Lp;   // allocation sizes use the padded stride
        ntiles = (c.P + TILE - 1) / TILE;
        n_indel_cap = c.has_ref ? s.n_indel_ops : 0;
        const size_t nagg = std::max<size_t>((std::max<size_t>(n, P * Lp) + SCAN_CHUNK - 1) / SCAN_CHUNK, 1);
        HIPCHK(d_reads.ensure((n + 1) * sizeof(DRead))); HIPCHK(d_ends.ensure((n + 1) * 4)); HIPCHK(d_prefmax.ensure((n + 1) * 4));
        HIPCHK(d_agg.ensure(nagg * 4 + 16)); HIPCHK(d_rng.ensure(((size_t)ntiles + 1) * sizeof(uint2)));
        HIPCHK(d_ncol.ensure(Lp * P * 4 + 16)); HIPCHK(d_depth.ensure(Lp * P * 4 + 16)); HIPCHK(d_unavail.ensure(P * 4 + 16));
        HIPCHK(d_istat.ensure(Lp * NBUCKET * NI * P * 4 + 16)); HIPCHK(d_fstat.ensure(Lp * NBUCKET * NF * P * 4 + 16));
        HIPCHK(d_ctr.ensure(sizeof(Counters))); HIPCHK(d_tilectr.ensure(((size_t)ntiles * Lp + 1) * sizeof(uint4)));
        if (n_indel_cap) {
            HIPCHK(d_cnt.ensure(Lp * P * 4 + 16)); HIPCHK(d_cursor.ensure(Lp * P * 4 + 16));
            HIPCHK(d_ev.ensure((n_indel_cap + 1) * sizeof(IndelEv))); HIPCHK(d_iout.ensure((n_indel_cap + 1) * sizeof(IndelOut)));
        }
        HIPCHK(hipStreamSynchronize(stream));
        return BRC_OK;
    }

    template <class Op, bool INCL>
    int scan(const typename Op::T* src, typename Op::T* dst, int64_t n) {
        if (n <= 0) return BRC_OK;
        const int64_t nb = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
        typename Op::T* agg = (typename Op::T*)d_agg.p;
        hipLaunchKernelGGL((k_scan_reduce<Op>), dim3((unsigned)nb), dim3(SCAN_T), 0, stream, src, n, agg);
        hipLaunchKernelGGL((k_scan_aggregates<Op>), dim3(1), dim3(1024), 0, stream, agg, nb);
        hipLaunchKernelGGL((k_scan_apply<Op, INCL>), dim3((unsigned)nb), dim3(SCAN_T), 0, stream, src, dst, n, (const typename Op::T*)agg);
        HIPCHK(hipGetLastError());
        return BRC_OK;
    }

    int compute(brc_timing* t) override {
        HIPCHK(hipSetDevice(device));
        const int64_t n = c.n_reads, P = c.P; const int Lp = c.Lp;
        int rc;
        Counters* ctr = (Counters*)d_ctr.p;
        HIPCHK(hipMemsetAsync(ctr, 0, sizeof(Counters), stream));
        const bool indels = n_indel_cap > 0 && P > 0 && n > 0;
        if (indels) HIPCHK(hipMemsetAsync(d_cnt.p, 0, (size_t)(Lp * P) * 4, stream));
        Planes pl = {(uint32_t*)d_ncol.p, (uint32_t*)d_depth.p, (uint32_t*)d_istat.p, (float*)d_fstat.p, (uint32_t*)d_unavail.p};
        const DRead* reads = (const DRead*)d_reads.p;
        HIPCHK(hipEventRecord(evt[T_ANNOTATE], stream));
        if (n > 0)
            hipLaunchKernelGGL(k_annotate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, c, in, (DRead*)d_reads.p, (int32_t*)d_ends.p,
                               (uint16_t*)d_bq.p, indels ? (uint32_t*)d_cnt.p : (uint32_t*)nullptr);
        HIPCHK(hipEventRecord(evt[T_SCAN_ENDS], stream));
        if ((rc = scan<OpMaxI32, true>((const int32_t*)d_ends.p, (int32_t*)d_prefmax.p, n))) return rc;
        HIPCHK(hipEventRecord(evt[T_TILES], stream));
        if (ntiles > 0)
            hipLaunchKernelGGL(k_tiles, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, stream, c, (const int32_t*)d_prefmax.p, reads, ntiles,
                               (uint2*)d_rng.p);
        HIPCHK(hipEventRecord(evt[T_PILEUP], stream));
        if (ntiles > 0) {
            unsigned nwg = (unsigned)((ntiles + PILEUP_WAVES - 1) / PILEUP_WAVES);
            nwg = (nwg + 7u) & ~7u;
            hipLaunchKernelGGL(k_pileup, dim3(nwg, (unsigned)Lp), dim3(PILEUP_WAVES * 64), 0, stream, c, in, reads, (const uint2*)d_rng.p, ntiles, pl, (uint4*)d_tilectr.p);
        }
        HIPCHK(hipEventRecord(evt[T_COUNT], stream));
        if (P > 0) {
            const unsigned nb = (unsigned)std::min<int64_t>((P + 255) / 256, 2048);
            hipLaunchKernelGGL(k_finalize, dim3(nb), dim3(256), 0, stream, c, (const uint32_t*)d_ncol.p, (const uint4*)d_tilectr.p, (int64_t)ntiles * Lp, ctr);
        }
        HIPCHK(hipEventRecord(evt[T_INDEL_SCAN], stream));
        if (indels && (rc = scan<OpSumU32, false>((const uint32_t*)d_cnt.p, (uint32_t*)d_cursor.p, (int64_t)Lp * P))) return rc;
        HIPCHK(hipEventRecord(evt[T_INDEL_FILL], stream));
        if (indels)
            hipLaunchKernelGGL(k_indel_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, c, in, reads, (uint32_t*)d_cursor.p, (IndelEv*)d_ev.p);
        HIPCHK(hipEventRecord(evt[T_INDEL_REDUCE], stream));
        if (indels)
            hipLaunchKernelGGL(k_indel_reduce, dim3((unsigned)std::min<int64_t>(((int64_t)Lp * P + 255) / 256, 4096)), dim3(256), 0, stream, c, in, reads,
                               (const uint32_t*)d_cnt.p, (const uint32_t*)d_cursor.p, (IndelEv*)d_ev.p, (const uint32_t*)d_unavail.p,
                               (IndelOut*)d_iout.p, ctr);
        HIPCHK(hipEventRecord(evt[T_N], stream));
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&h_ctr, ctr, sizeof(Counters), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if (t) {
            memset(t, 0, sizeof *t);
            for (int i = 0; i < T_N; ++i) HIPCHK(hipEventElapsedTime(&t->ms[i], evt[i], evt[i + 1]));
            HIPCHK(hipEventElapsedTime(&t->total_ms, evt[0], evt[T_N]));
        }
        computed = true;
        return BRC_OK;
    }

    int counts(uint64_t* e, uint64_t* p) override {
        if (!computed) { err = "not computed"; return BRC_E_ARG; }
        if (e) *e = h_ctr.n_events;
        if (p) *p = h_ctr.n_positions;
        return BRC_OK;
    }

    int fetch(HostPlanes* out) override {
        HIPCHK(hipSetDevice(device));
        if (!computed) { err = "not computed"; return BRC_E_ARG; }
        const size_t P = (size_t)c.PS, Lp = (size_t)c.Lp;   // planes are copied with their padded stride
        if (!h_ncol.reserve(Lp * P + 4) || !h_depth.reserve(Lp * P + 4) || !h_unavail.reserve(P + 4) ||
            !h_istat.reserve(Lp * NBUCKET * NI * P + 4) || !h_fstat.reserve(Lp * NBUCKET * NF * P + 4)) { err = "pinned host allocation failed"; return BRC_E_NOMEM; }
        if (P) {
            HIPCHK(hipMemcpyAsync(h_ncol.p, d_ncol.p, Lp * P * 4, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpyAsync(h_depth.p, d_depth.p, Lp * P * 4, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpyAsync(h_istat.p, d_istat.p, Lp * NBUCKET * NI * P * 4, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpyAsync(h_fstat.p, d_fstat.p, Lp * NBUCKET * NF * P * 4, hipMemcpyDeviceToHost, stream));
            if (c.per_lib) HIPCHK(hipMemcpyAsync(h_unavail.p, d_unavail.p, P * 4, hipMemcpyDeviceToHost, stream));
        }
        const size_t ns = h_ctr.n_indel_slots;
        if (ns) {
            if (!h_iout.reserve(ns + 4)) { err = "pinned host allocation failed"; return BRC_E_NOMEM; }
            HIPCHK(hipMemcpyAsync(h_iout.p, d_iout.p, ns * sizeof(IndelOut), hipMemcpyDeviceToHost, stream));
        }
        HIPCHK(hipStreamSynchronize(stream));
        iout_compact.clear();
        for (size_t i = 0; i < ns; ++i) if (h_iout.p[i].len != 0) iout_compact.push_back(h_iout.p[i]);
        out->ncol = h_ncol.p; out->depth = h_depth.p; out->istat = h_istat.p; out->fstat = h_fstat.p; out->unavail = h_unavail.p;
        out->indel = iout_compact.data(); out->n_indel = (int64_t)iout_compact.size();
        out->n_events = h_ctr.n_events; out->n_positions = h_ctr.n_positions;
        out->warn[BRC_W_SM_MISSING] = h_ctr.w_sm; out->warn[BRC_W_NM_MISSING] = h_ctr.w_nm; out->warn[BRC_W_ZM_MISSING] = 0;
        out->warn[BRC_W_LIB_UNAVAILABLE] = h_ctr.w_lib;
        return BRC_OK;
    }
};

Backend* make_backend(const brc_config& cfg, int* errc) {
    HipBackend* b = new (std::nothrow) HipBackend();
    if (!b) { *errc = BRC_E_NOMEM; return nullptr; }
    const int rc = b->init(cfg.device);
    if (rc) {
        fprintf(stderr, "brc: cannot create the HIP engine: %s\n", b->last_error());
        *errc = rc; delete b; return nullptr;
    }
    *errc = BRC_OK;
    return b;
}
const char* backend_kind() { return "hip-gfx950"; }
const char* backend_kernel_name(int k) { return (k >= 0 && k < T_N) ? kKernelNames[k] : nullptr; }

}  // namespace brc
