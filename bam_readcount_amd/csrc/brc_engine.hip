// brc_engine.hip — the HIP/CDNA4 (gfx950) device pipeline behind the C-ABI of include/brc.h.
//
// Kernels (per region; the main pipeline on one engine-owned stream, the indel side path on a second one, the download of
// device-written text on a third; see DESIGN.md for layouts and rooflines):
//   k_refcode       reference characters -> 4-bit codes
//   k_annotate_groups  K1: fetch_func's Zm integers per read (8 bases per lane, byte-parallel), the event-BYTE stream
//                   (quality << 2 | base index per base; brc_core.h: eb_make — plus the sparse wide stream for escapes) and the read's PIECES (walk_pieces): 48-B records (+ a 32-B rare record for the few that need one) in
//                   library-major slots; also writes each read's indel events to its slots of the raw event list and counts
//                   them per (16 positions, library) bucket
//   k_unavail       -p only: first library-less read of every column (those positions are abandoned, :281-284)
//   k_reach_blockmax / k_scan_aggregates / k_tiles_all   [lo,hi) piece range of every (64-position tile, library): running
//                   maximum of the piece reaches (library << 32 | reach: one scan for all libraries), then one coalesced
//                   pass over the pieces — piece r owns the tiles whose lo is r and those whose hi is r + 1
//   k_scan_*        3-phase exclusive sum of the indel-event counts (per-bucket offsets), of the text line lengths
//   k_pileup2       THE hot kernel: one wave per (tile, library), lane == reference position, wave-uniform walk over the
//                   tile's pieces in column order: piece records by scalar loads, event-byte windows staged into LDS by
//                   direct-to-LDS loads, three packed integer accumulators + 4 order-preserving fp32 sums per bucket,
//                   coalesced 256-B plane stores.  Integer/byte work, HBM-bound: no MFMA by design.
//   k_xev_compact   the 1024 third-allele sub-lists (one atomic cursor each) -> one list
//   k_finalize      emitted-position count + per-tile partial counters
//   k_indel_scatter / k_indel_reduce   indel side path (<1 % of events), sparse: raw events -> (16 positions, library) buckets,
//                   one lane per bucket sorts by (position, library, read) and folds every key in column order
//   k_text_len / k_text_write   BRC_OPT_DEVICE_TEXT: the lines pileup_func prints, written from the compact result
//                   (brc_core.h: text_line): byte lengths -> exclusive scan -> bytes
//
// There is no CPU fallback here: without a HIP device make_backend() fails with BRC_E_NODEVICE.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <mutex>
#include <vector>

#include "brc_host.h"

namespace brc {

struct Counters {
    unsigned long long n_events, n_positions, w_sm, w_nm, w_lib;
    unsigned int n_indel_slots, n_xev;   // n_xev: third-allele events in the compacted list
    unsigned int xev_max, n_wave_reads;  // fullest sub-list's cursor (above its capacity: grow and compute again); reads K1 left to k_annotate_wave
    unsigned int n_wave_big, n_wave_huge;   // ... those of them with more than AW_MCAP / AW_MCAP_BIG M operators (the one-wave-per-workgroup instantiations)
    unsigned int n_literal, n_wave_eqx;      // reads with an empty M / = / X operator: piled up by the iterator's own cursor (k_annotate_cursor); wave-form reads with = / X operators ...
    unsigned int n_wave_eqx_big, pad_;       // ... those of them with more than AW_MCAP_EQX match operators
};

// Profiling ablations that switch parts of the kernels off (wrong results, timing only) exist only in experiment builds
// (tools/build_variant.sh passes -DBRC_EXP_KNOBS; such a library then reads BRC_PILEUP_VARIANT / BRC_ANN_VARIANT).  In the product
// both tests are the constant `false`: the shipped library contains neither the code nor the names of the knobs
// (tests/test_abi.py::test_product_ignores_ablation_environment).
#ifdef BRC_EXP_KNOBS
#define BRC_PVAR(n) (c.variant == (n))
#define BRC_AVAR(n) (c.ann_variant == (n))
#else
#define BRC_PVAR(n) false
#define BRC_AVAR(n) false
#endif

// ---------------------------------------------------------------- wave helpers (wave64)

// BRC_CKS: BRC_CK for an address that feeds a scalar load (inline assembly, "s" constraint).  The checked pointer comes out of a
// per-lane comparison, so the compiler keeps it in vector registers: it is made wave-uniform again (it is: every lane checked
// the same address).  Outside the checked build: the pointer itself.
#ifdef BRC_CHECKED
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const uint64_t v = (uint64_t)p;
    return (const char*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v));
}
#define BRC_CKS(c, K, SITE, BUF, p, bytes, unit, piece) uniform_ptr(BRC_CK(c, K, SITE, BUF, p, bytes, unit, piece))
#else
#define BRC_CKS(c, K, SITE, BUF, p, bytes, unit, piece) (p)
#endif

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// sum over the 64 lanes of a 32-bit value, result in every lane's return value (wave-uniform): six adds on the DPP
// network (quad swaps, mirrors inside a row of 16, row broadcasts) instead of six ds_bpermute round trips
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);     // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);    // row_half_mirror
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true);    // row_mirror: every lane holds its row's sum
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// minimum over the 64 lanes of a 64-bit value, wave-uniform (scalar registers)
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t u = ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o, 64) << 32) | (uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)v, o, 64);
        v = u < v ? u : v;
    }
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

// the value of the lane below / above (lane 0 / lane 63 get 0): one DPP move on the gfx9 wave-shift network
__device__ __forceinline__ int lane_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false); }   // wave_shr:1
__device__ __forceinline__ int lane_shl1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, false); }   // wave_shl:1

// ---------------------------------------------------------------- K1

// Reference characters -> 4-bit base codes, once per region (the annotator compares codes, bamreadcount.cpp:149-152).
// A NUL character keeps bit 7 set: the annotator stops at it (:151), which only the serial path reproduces.
// REFCODE_PAD bytes of padding (code 15) on both sides: an 8-byte window may start before / end after the slice.
enum { REFCODE_PAD = 16 };
// (launched once when an engine is created: the code object is loaded then, not inside the first region)
// First launch of an engine (loads the code object) and the proof obligation of div_small (brc_core.h): every quotient of the
// domain against the compiler's correctly rounded division, bit for bit, on THIS device.
__global__ void k_divcheck(uint32_t* __restrict__ bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = i / DIV_SMALL_M, m = i % DIV_SMALL_M;
    if (n >= DIV_SMALL_N || m == 0u) return;
    const float q = div_small((float)n, (float)m), want = (float)n / (float)m;
    if (__float_as_uint(q) != __float_as_uint(want)) atomicAdd(bad, 1u);
}
__global__ __launch_bounds__(256) void k_refcode(const char* __restrict__ ref, uint8_t* __restrict__ code, int64_t n) {
    // 16 codes per thread; `code` (padded buffer) is 16-byte aligned and REFCODE_PAD == 16, so chunk i of the output
    // holds the codes of ref[16 i - 16 .. 16 i)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t j0 = i * 16 - REFCODE_PAD;
    if (j0 >= n + REFCODE_PAD) return;
    uint32_t w[4];
    uint8_t src[16];
    if (j0 >= 0 && j0 + 16 <= n) __builtin_memcpy(src, ref + j0, 16);
    else for (int k = 0; k < 16; ++k) { const int64_t j = j0 + k; src[k] = (j >= 0 && j < n) ? (uint8_t)ref[j] : (uint8_t)'N'; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t j = j0 + 4 * q + k;
            const uint32_t ch = src[4 * q + k];
            const uint32_t cd = (j >= 0 && j < n) ? (ch ? nt16_of_char(ch) : 0x8fu) : 0x0fu;
            v |= cd << (8 * k);
        }
        w[q] = v;
    }
    *reinterpret_cast<uint4*>(code + i * 16) = make_uint4(w[0], w[1], w[2], w[3]);
}

// K1, group form (the one normally launched): one wave owns 64 consecutive reads; a lane of the per-base pass owns one
// GROUP of 8 consecutive bases of one read, groups of consecutive reads packed back to back over the lanes, so the
// QUAL / SEQ / reference-code loads (8 / 4 / 8 bytes per lane) and the 16-byte bq stores are contiguous across the wave
// and every per-base step works on 4 bases per instruction (bytes of a dword):
//   phase A (lane = read)  metadata + CIGAR walk: reference length, clips, and the (at most two) M operators as query
//                          ranges with their reference offsets; reads needing the per-base pass are compacted ("dense")
//                          and their parameters go to LDS; reads with more than two M operators, reads overhanging the
//                          reference (the annotator's break/continue quirks) and runs without a reference are left to
//                          the serial annotate_read() in phase C;
//   phase B (lane = group) 64 groups per pass.  lane -> read through one LDS word per pass (reads starting inside the pass
//                          set the bit of their first lane; mbcnt); base codes vs reference codes and the "=ACGTN" buckets
//                          are byte-parallel; mismatch qualities: runs of read-adjacent mismatches contribute their
//                          maximum (:152-172,199) — a lone mismatch inside its group is added directly, anything else (two
//                          mismatches in a group, a run crossing a group or pass boundary) goes through a lane-serial
//                          walk + neighbour links (DPP lane shifts); sums are accumulated per read in LDS;
//   phase C (lane = read)  the quality != 2 scan from the read's 3' end (:201-238; 8 bases per load, as a rule one load),
//                          three-prime / Q2 logic, DRead + float constants, the pieces, the indel events.
enum { AW_MCAP = 1024,               // M operators of a read the wave form (k_annotate_wave, below) holds in LDS: four waves per workgroup ...
       AW_MCAP_BIG = 5120,           // ... and one wave per workgroup (60 KB: reads of ~80 kb with a match run of 15 bases between two operators)
       AW_MCAP_HUGE = 13500,         // ... and one wave per CU (158 of gfx950's 160 KB of LDS: reads of ~210 kb)
       // reads with = / X operators (pbmm2, minimap2 --eqx): the list keeps a fourth word per entry — the ANNOTATOR's query offset, see k_annotate_wave
       AW_MCAP_EQX = 768, AW_MCAP_EQX_BIG = 3840 };
struct AnnPar { uint4 a, b, c; };    // a = {L, qrel, srel, brow.lo}  b = {S, m1lo, m1hi, d1}  c = {m2lo, m2hi, d2, brow.hi}

__device__ __forceinline__ uint32_t nzb7(uint32_t x) { return x + 0x7f7f7f7fu; }   // bytes <= 0x7f: bit 7 of a byte <=> byte != 0

// K1 at six waves per SIMD (80 VGPRs): left alone the allocator takes 88 — five waves — for three values that live from phase A
// to phase C across the whole per-base pass; at 80 they are spilled ONCE before the pass loop and reloaded after it (six scratch
// instructions per wave, none inside a loop: tests/test_abi.py::test_kernel_register_budget).  0 = the allocator's own choice.
#ifndef BRC_ANN_WAVES_PER_EU
#define BRC_ANN_WAVES_PER_EU 6
#endif
// (The per-library instantiation keeps the allocator's five waves: at six it spills inside the operator walks of phase C, and its
// time did not move with the occupancy — profiles/r04_ab_*.)
#if BRC_ANN_WAVES_PER_EU > 0
#define BRC_ANN_OCC __attribute__((amdgpu_waves_per_eu(one_stream ? BRC_ANN_WAVES_PER_EU : 5, one_stream ? BRC_ANN_WAVES_PER_EU : 5)))
#else
#define BRC_ANN_OCC
#endif
template <bool one_stream, int SH>     // one_stream: the event-byte rows and the pieces of consecutive reads are consecutive in memory (no per-library layout); SH: DevCfg.pack_shift
__global__ __launch_bounds__(256) BRC_ANN_OCC void k_annotate_groups(DevCfg c, DevIn in, DRead* __restrict__ reads, const uint32_t* __restrict__ piece_off,
                                                         Piece* __restrict__ pieces, PieceRare* __restrict__ rare, int2* __restrict__ keyreach,
                                                         uint8_t* __restrict__ eb, IndelEv* __restrict__ ev_raw, uint32_t* __restrict__ bucket_cnt,
                                                         const uint32_t* __restrict__ cigar_ro, const uint8_t* __restrict__ qual_ro,
                                                         const uint8_t* __restrict__ seq_ro, const uint8_t* __restrict__ refcode,
                                                         const uint16_t* __restrict__ wanted /* brc_region_windows: the wanted lanes of every tile, or null */) {
    c.pack_shift = SH;
    struct WaveLds { AnnPar par[64]; uint32_t sum[64]; uint32_t redo[64]; uint32_t wide[64]; uint32_t G[64]; unsigned long long mark; };
    __shared__ WaveLds lds_all[4];
    const int lane = threadIdx.x & 63;
    const uint32_t wv = __builtin_amdgcn_readfirstlane((uint32_t)threadIdx.x >> 6);
    WaveLds& W = lds_all[wv];
    const int64_t rb = ((int64_t)blockIdx.x * 4 + wv) * 64;
    if (rb >= c.n_reads) return;
    const int nrd = (int)((c.n_reads - rb) < 64 ? (c.n_reads - rb) : 64);
    const int64_t my = rb + (lane < nrd ? lane : nrd - 1);
    const bool have = lane < nrd;
    // ---- phase A
    const int32_t pos = in.pos[my];
    const uint32_t flag = in.flag[my];
    const int32_t L = in.l_qseq[my];
    const uint32_t nc = in.n_cigar[my];
    const uint64_t qoff = in.qual_off[my], soff = in.seq_off[my], brow = in.bq_row[my];
    const uint32_t coff = (uint32_t)in.cig_off[my];
    int32_t rlen = 0; int clipped = L, left_clip = 0, right_clip = L; int64_t tot_d = 0, tot_is = 0;
    uint32_t cig0 = 0, n_idp = 0;                     // n_idp: I / D / P operators = the read's slots in the raw indel-event list
    CigShape shape;
    int n_m = 0; int32_t m1lo = 0, m1hi = 0, m2lo = 0, m2hi = 0; int64_t d1 = 0, d2 = 0;
    {
        int rs = 0; int64_t x = pos;                     // the annotator's read / reference cursors (bamreadcount.cpp:133-198)
#ifdef BRC_CHECKED
        const uint32_t* const cig_row = BRC_CK(c, CK_ANNOTATE, 1, CB_CIGAR, cigar_ro + coff, 4ull * nc, my, -1);
#define BRC_CIG_AT(k) cig_row[k]
#else
#define BRC_CIG_AT(k) cigar_ro[coff + (k)]
#endif
        for (uint32_t k = 0; k < nc; ++k) {
            const uint32_t cg = BRC_CIG_AT(k);
            if (k == 0) cig0 = cg;
            const uint32_t op = cg & 0xfu; const int len = (int)(cg >> 4);
            shape_add(shape, op, len);
            if (is_refop(op)) rlen += len;
            if (op == CINS || op == CDEL || op == CPAD) ++n_idp;
            if (op == CDEL || op == CREF_SKIP) tot_d += len;
            if (op == CINS || op == CSOFT_CLIP) tot_is += len;
            if (op == CSOFT_CLIP) { clipped -= len; if (k == 0) left_clip += len; else right_clip -= len; }
            if (op == CMATCH) {
                if (n_m == 0) { m1lo = rs; m1hi = rs + len; d1 = x - rs - c.ref_lo; }
                else if (n_m == 1) { m2lo = rs; m2hi = rs + len; d2 = x - rs - c.ref_lo; }
                ++n_m; rs += len; x += len;
            } else if (op == CDEL || op == CREF_SKIP) x += len;
            else if (op == CINS || op == CSOFT_CLIP) rs += len;
        }
#undef BRC_CIG_AT
    }
    const bool simple = nc == 1 && (cig0 & 0xfu) == CMATCH;
    bool dropped = (flag & BRC_PUSH_MASK) != 0;
    if (nc == 0) dropped = true;
    if (nc == 1 && !is_mop(cig0 & 0xfu)) dropped = true;
    // query ranges are clamped to the read (malformed CIGARs), reference offsets must fit the 32-bit slice index
    if (m1hi > L) m1hi = L; if (m2hi > L) m2hi = L; if (m1lo > m1hi) m1lo = m1hi; if (m2lo > m2hi) m2lo = m2hi;
    // The per-base pass addresses QUAL / SEQ as a wave-uniform 64-bit base + a 32-bit lane offset.  The base is the SMALLEST
    // offset of the wave's reads (a batch may lay its rows out in any order — brc.h only asks for offsets inside the arenas —
    // so lane 0's row need not be the first); a read whose row starts 4 GiB or more above it takes the serial path, which
    // addresses with 64 bits.  (readfirstlane returns int: without the uint32_t casts a low half with bit 31 set sign-extends into
    // the high half — arenas beyond 2 GB then read 4 GB below their rows; tests/test_gpu_parity.py::test_hip_arena_offsets_beyond_2_gib)
    const uint64_t qbase = wave_min_u64(qoff), sbase = wave_min_u64(soff);
    const bool fallback = !c.has_ref || pos < 0 || (int64_t)pos + rlen > c.ref_len || n_m > 2 ||
                          d1 < -(int64_t)L || d2 < -(int64_t)L || d1 > 0x7fff0000ll || d2 > 0x7fff0000ll ||
                          qoff - qbase > 0xff000000ull || soff - sbase > 0xff000000ull;
    const bool work_me = have && !dropped && !fallback && L > 0;
    const unsigned long long work = __ballot(work_me);
    const int nd = __builtin_popcountll(work);                               // dense reads of this wave
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(work >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)work, 0u));
    uint32_t T = 0;
    if (BRC_AVAR(5)) return;
    if (nd) {
        if (work_me) {
            AnnPar p;
            // (the rows of a wave's reads are NOT near each other in per-library mode — every library has its own run of rows —
            // so the row start travels as a full 64-bit element index: a.w low, c.w high)
            p.a = make_uint4((uint32_t)L, (uint32_t)(qoff - qbase), (uint32_t)(soff - sbase), (uint32_t)brow);
            p.b = make_uint4(0u, (uint32_t)m1lo, (uint32_t)m1hi, (uint32_t)(int32_t)d1);
            p.c = make_uint4((uint32_t)m2lo, (uint32_t)m2hi, (uint32_t)(int32_t)d2, (uint32_t)(brow >> 32));
            W.par[rank] = p;
            W.G[rank] = ((uint32_t)L + 7u) >> 3;
        }
        W.sum[lane] = 0u; W.redo[lane] = 0u; W.wide[lane] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");               // lanes read each other's LDS records below
        // exclusive prefix sum of the group counts over the dense reads
        const uint32_t g_me = lane < nd ? W.G[lane] : 0u;
        uint32_t incl = g_me;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
        T = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t S_me = incl - g_me;                                   // first group of dense read `lane`
        if (lane < nd) W.par[lane].b.x = S_me;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");

        // ---- phase B
        int jbefore = -1;                          // last dense read that starts before the pass
        bool carry_open = false; uint32_t carry_t = 0; int carry_jr = -1;    // mismatch run open at the end of the previous pass
        const int64_t ref_n = c.ref_hi - c.ref_lo;
        const uint8_t* const qwave = qual_ro + qbase; const uint8_t* const swave = seq_ro + sbase; const uint8_t* const refpad = refcode - REFCODE_PAD;
        // Software pipeline, one pass deep: the loads of pass p + 1 (lane -> read mapping through LDS, then QUAL / SEQ /
        // reference codes from HBM) are issued before pass p is worked on, so a wave's memory round trips overlap its own
        // arithmetic instead of standing between two passes.
        struct Fetch { int jr; int32_t b; uint2 Q; uint32_t S; uint2 R1; };
        auto fetch = [&](uint32_t base) -> Fetch {
            Fetch F;
            // lane -> dense read: reads starting inside the pass set the bit of their first lane in one LDS word (the LDS
            // operations of a wave execute in issue order: clear, ORs, read)
            if (lane == 0) __hip_atomic_store(&W.mark, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            { const uint32_t d = S_me - base; if (lane < nd && g_me && d < 64u) __hip_atomic_fetch_or(&W.mark, 1ull << d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            unsigned long long M = __hip_atomic_load(&W.mark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            M = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)M) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(M >> 32)) << 32);
            const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(M >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)M, 0u));
            F.jr = jbefore + below + (int)((M >> lane) & 1ull);
            jbefore += __builtin_popcountll(M);
            const uint32_t G = base + (uint32_t)lane;
            const uint4 Pa = W.par[F.jr].a, Pb = W.par[F.jr].b;
            F.b = G < T ? (int32_t)((G - Pb.x) << 3) : 0;               // idle lanes of the last pass stay inside their read
            // (wave-uniform base + 32-bit lane offset: the scalar-base addressing form, no 64-bit vector arithmetic.  Non-temporal
            // LOADS of QUAL / SEQ — read exactly once — were measured: K1 +6 %)
            __builtin_memcpy(&F.Q, BRC_CK(c, CK_ANNOTATE, 2, CB_QUAL, qwave + (uint32_t)(Pa.y + (uint32_t)F.b), 8, rb + F.jr, F.b), 8);
            __builtin_memcpy(&F.S, BRC_CK(c, CK_ANNOTATE, 3, CB_SEQ, swave + (uint32_t)(Pa.z + ((uint32_t)F.b >> 1)), 4, rb + F.jr, F.b), 4);
            // reference codes under the first M operator (a window outside the slice: any in-bounds window, its bytes are masked)
            int64_t r1off = (int64_t)F.b + (int32_t)Pb.w;
            if (r1off < -8 || r1off > ref_n) r1off = 0;
            __builtin_memcpy(&F.R1, BRC_CK(c, CK_ANNOTATE, 4, CB_REFCODE, refpad + (uint32_t)((int32_t)r1off + REFCODE_PAD), 8, rb + F.jr, F.b), 8);
            return F;
        };
        if (BRC_AVAR(3)) T = 0;
        Fetch Fn = fetch(0u);
        for (uint32_t base = 0; base < T; base += 64u) {
            const Fetch F = Fn;
            if (base + 64u < T) Fn = fetch(base + 64u);
            const int jr = F.jr;
            const uint32_t G = base + (uint32_t)lane;
            const bool act = G < T;
            const AnnPar P = W.par[jr];

            const int32_t b = F.b;
            const int32_t Lr = (int32_t)P.a.x;
            const int nv = act ? (Lr - b < 8 ? Lr - b : 8) : 0;              // valid bases of the group (1..8)
            const uint2 Q = F.Q; const uint32_t S = F.S; const uint2 R1 = F.R1;
            // flags of the bases inside the first / second M operator, as bit 7 of each byte
            const unsigned long long vflags = (nv >= 8 ? ~0ull : ((1ull << (8 * nv)) - 1ull)) & 0x8080808080808080ull;
            unsigned long long f1 = vflags, f2 = 0ull;
            const bool plain = (int32_t)P.b.y == 0 && (int32_t)P.b.z == Lr;       // one M operator over the whole read
            if (__ballot(act && !plain)) {
                const int lo1 = (int32_t)P.b.y - b, hi1 = (int32_t)P.b.z - b;     // first M operator, group-relative
                const int a1 = lo1 < 0 ? 0 : (lo1 > 8 ? 8 : lo1), e1 = hi1 < 0 ? 0 : (hi1 > 8 ? 8 : hi1);
                const unsigned long long below_e1 = e1 >= 8 ? ~0ull : ((1ull << (8 * e1)) - 1ull), below_a1 = a1 >= 8 ? ~0ull : ((1ull << (8 * a1)) - 1ull);
                f1 = vflags & below_e1 & ~below_a1;
                const int lo2 = (int32_t)P.c.x - b, hi2 = (int32_t)P.c.y - b;
                const int a2 = lo2 < 0 ? 0 : (lo2 > 8 ? 8 : lo2), e2 = hi2 < 0 ? 0 : (hi2 > 8 ? 8 : hi2);
                const unsigned long long below_e2 = e2 >= 8 ? ~0ull : ((1ull << (8 * e2)) - 1ull), below_a2 = a2 >= 8 ? ~0ull : ((1ull << (8 * a2)) - 1ull);
                f2 = vflags & below_e2 & ~below_a2;
            }
            // ---- base codes of the 8 bases in read order: N.x = bases 0..3, N.y = bases 4..7 (one per byte)
            const uint32_t Ev = (S >> 4) & 0x0f0f0f0fu, Od = S & 0x0f0f0f0fu;
            uint2 N;
            N.x = __builtin_amdgcn_perm(Od, Ev, 0x05010400u);
            N.y = __builtin_amdgcn_perm(Od, Ev, 0x07030602u);
            // ---- mismatch flags (:152): base != reference code, reference code != 15, base != 0 — on M-operator bases
            uint2 mm; uint32_t nul;
            {
                const uint32_t rx = R1.x & 0x0f0f0f0fu, ry = R1.y & 0x0f0f0f0fu;
                mm.x = nzb7(N.x ^ rx) & nzb7(rx ^ 0x0f0f0f0fu) & nzb7(N.x) & (uint32_t)f1;
                mm.y = nzb7(N.y ^ ry) & nzb7(ry ^ 0x0f0f0f0fu) & nzb7(N.y) & (uint32_t)(f1 >> 32);
                nul = (R1.x & (uint32_t)f1) | (R1.y & (uint32_t)(f1 >> 32));
            }
            if (__ballot(f2 != 0ull)) {                                       // bases of a second M operator (after an indel)
                int64_t r2off = (int64_t)b + (int32_t)P.c.z;
                if (!f2 || r2off < -8 || r2off > ref_n) r2off = 0;
                uint2 R2; __builtin_memcpy(&R2, BRC_CK(c, CK_ANNOTATE, 5, CB_REFCODE, refpad + (uint32_t)((int32_t)r2off + REFCODE_PAD), 8, rb + jr, b), 8);
                const uint32_t rx = R2.x & 0x0f0f0f0fu, ry = R2.y & 0x0f0f0f0fu;
                mm.x |= nzb7(N.x ^ rx) & nzb7(rx ^ 0x0f0f0f0fu) & nzb7(N.x) & (uint32_t)f2;
                mm.y |= nzb7(N.y ^ ry) & nzb7(ry ^ 0x0f0f0f0fu) & nzb7(N.y) & (uint32_t)(f2 >> 32);
                nul |= (R2.x & (uint32_t)f2) | (R2.y & (uint32_t)(f2 >> 32));
            }
            // ---- the event bytes for KB (brc_core.h: eb_make), 8 per lane: quality << 2 | base index (A C G T = 0..3); the row is
            // padded to 16 elements.  Base codes -> index or "not A C G T" (bit 7) through two byte tables (codes 0..7 and 8..15).
            if (act && !BRC_AVAR(1)) {
                auto index_of = [](uint32_t codes) -> uint32_t {
                    const uint32_t sx = codes & 0x07070707u;
                    const uint32_t lx = __builtin_amdgcn_perm(0x80808002u, 0x80010080u, sx), hx = __builtin_amdgcn_perm(0x80808080u, 0x80808003u, sx);
                    const uint32_t gx = (codes + 0x78787878u) & 0x80808080u;                                       // code >= 8
                    const uint32_t mx = (gx - (gx >> 7)) | gx;                                                     // 0xff per such byte
                    return (hx & mx) | (lx & ~mx);
                };
                auto ev_bytes = [](uint32_t Qd, uint32_t Id, uint32_t valid7, uint32_t& esc7) -> uint32_t {
                    const uint32_t qlo = Qd & 0x7f7f7f7fu, qhi = Qd & 0x80808080u;
                    const uint32_t q0 = ~(nzb7(qlo) | qhi), q63 = (qlo + 0x41414141u) | qhi;                       // (bit 7 of each byte)
                    esc7 = (Id | q0 | q63) & valid7;
                    return ((Qd & 0x3f3f3f3fu) << 2) | (Id & 0x03030303u);
                };
                uint32_t ex, ey;
                uint2 E; E.x = ev_bytes(Q.x, index_of(N.x), (uint32_t)vflags, ex); E.y = ev_bytes(Q.y, index_of(N.y), (uint32_t)(vflags >> 32), ey);
                if (ex | ey) {
                    // escapes (a quality of 0 or above 62, an N or '=' base): the byte only answers the base-quality filter; the
                    // read's pieces are marked PF_WIDE, and its full words quality << 8 | bucket are in the wide stream (k_wide_rows
                    // wrote them: the host found the read when it was pushed)
                    // escape byte = 63 << 2 where the base passes -b, 0 where it does not: bytewise q >= min_bq (both sides split into
                    // their low seven bits and bit 7)
                    const uint32_t mq = (uint32_t)(c.min_bq < 0 ? 0 : (c.min_bq > 255 ? 255 : c.min_bq)) * 0x01010101u;
                    const uint32_t none = c.min_bq > 255 ? 0u : 0xffffffffu;                                   // (-b above 255: nothing passes)
                    auto ge7 = [](uint32_t x, uint32_t y) -> uint32_t {                                          // bit 7 of each byte: x >= y
                        const uint32_t t = ((x | 0x80808080u) - (y & 0x7f7f7f7fu));                              // bit 7: low seven bits of x >= those of y
                        return ((x & ~y) | (~(x ^ y) & t)) & 0x80808080u;
                    };
                    const uint32_t px = ((ge7(Q.x, mq) & none) >> 7) * 0xffu, py = ((ge7(Q.y, mq) & none) >> 7) * 0xffu;
                    const uint32_t mx2 = (ex >> 7) * 0xffu, my3 = (ey >> 7) * 0xffu;
                    E.x = (E.x & ~mx2) | (mx2 & px & 0xfcfcfcfcu); E.y = (E.y & ~my3) | (my3 & py & 0xfcfcfcfcu);
                    atomicOr(&W.wide[jr], 1u);
                }
                // (rows are 16-byte aligned; written once here, read by k_pileup2 a kernel later.  With ONE stream of rows — no
                // per-library layout — the wave's stores run on through memory and the non-temporal hint pays; with four
                // library-major streams it costs 2-8 %: measured, profiles/r03_ab_09)
                typedef uint32_t u32x2s __attribute__((ext_vector_type(2)));
                u32x2s* dst = reinterpret_cast<u32x2s*>(BRC_CK(c, CK_ANNOTATE, 7, CB_EB, eb + ((((uint64_t)P.c.w) << 32) | (uint64_t)P.a.w) + (uint32_t)b, 8, rb + jr, b));
                const u32x2s val = {E.x, E.y};
                if (one_stream) __builtin_nontemporal_store(val, dst); else *dst = val;
            }
            if (nul & 0x80808080u) atomicOr(&W.redo[jr], 1u);                           // a NUL reference character under an M base
            // ---- mismatch qualities
            const bool has_mm = (mm.x | mm.y) != 0u;
            const int nbits = __builtin_popcount(mm.x) + __builtin_popcount(mm.y);
            const bool bit0 = (mm.x & 0x80u) != 0u, bit7 = (mm.y & 0x80000000u) != 0u;
            const int pjr = lane_shr1(jr); const int pb7 = lane_shr1(bit7 ? 1 : 0);
            const bool link = bit0 && (lane ? (pjr == jr && pb7 != 0) : (carry_open && carry_jr == jr));
            // a run left open by the previous pass that does not continue into lane 0 ended there
            if (carry_open && lane == 0 && !link) atomicAdd(&W.sum[carry_jr], carry_t);
            const unsigned long long any_mm = __ballot(has_mm);
            uint32_t t_out = 0u;                                              // maximum of the run still open at the lane's last base
            if (any_mm) {
                // a mismatch on a group's last base is part of a longer run only if the NEXT lane links to it (lane 63: the next
                // pass decides): anything else stands alone (30 % of the passes took the lane-serial walk for that bit alone)
                const int nlink = lane_shl1(link ? 1 : 0);
                const unsigned long long hard = __ballot(has_mm && (nbits > 1 || link || (bit7 && (lane == 63 || nlink != 0))));
                if (!hard) {
                    // every mismatch stands alone inside its group: its quality is the run maximum
                    if (has_mm) {
                        const int k = mm.x ? (__builtin_ctz(mm.x) >> 3) : 4 + (__builtin_ctz(mm.y) >> 3);
                        const uint32_t qv = ((k < 4 ? Q.x : Q.y) >> ((k & 3) << 3)) & 0xffu;
                        atomicAdd(&W.sum[jr], qv);
                    }
                } else {
                    // lane-serial walk over the 8 bases (:152-172); runs entering from / leaving to a neighbour lane are linked
                    const unsigned long long m64 = (unsigned long long)mm.x | ((unsigned long long)mm.y << 32);
                    const unsigned long long q64 = (unsigned long long)Q.x | ((unsigned long long)Q.y << 32);
                    const bool full = nbits == 8;
                    // own tail-run maximum, ignoring what enters from the left (exact unless the whole lane is one run)
                    uint32_t t_own = 0u; { bool open = false; uint32_t cur = 0u;
                        _Pragma("unroll") for (int k = 0; k < 8; ++k) { const bool m = (m64 >> (8 * k + 7)) & 1ull; const uint32_t qv = (uint32_t)(q64 >> (8 * k)) & 0xffu;
                            if (m) { cur = open ? (cur > qv ? cur : qv) : qv; open = true; } else open = false; }
                        t_own = open ? cur : 0u; }
                    // propagate through lanes that are one run from end to end
                    uint32_t t = t_own;
                    for (;;) {
                        uint32_t pt = (uint32_t)lane_shr1((int)t);
                        if (lane == 0) pt = carry_t;
                        const uint32_t nt = (full && link) ? (t_own > pt ? t_own : pt) : t_own;
                        const bool ch = nt != t; t = nt;
                        if (!__ballot(ch)) break;
                    }
                    uint32_t pt = (uint32_t)lane_shr1((int)t);
                    if (lane == 0) pt = carry_t;
                    uint32_t add = 0u; { bool open = link; uint32_t cur = link ? pt : 0u;
                        _Pragma("unroll") for (int k = 0; k < 8; ++k) { const bool m = (m64 >> (8 * k + 7)) & 1ull; const uint32_t qv = (uint32_t)(q64 >> (8 * k)) & 0xffu;
                            if (m) { cur = open ? (cur > qv ? cur : qv) : qv; open = true; } else { if (open) add += cur; open = false; } }
                        if (open) { if (lane < 63 && !nlink) add += cur; else t_out = cur; } }   // lane 63: decided by the next pass
                    if (has_mm || link) { if (add) atomicAdd(&W.sum[jr], add); }
                    if (lane < 63) t_out = 0u;
                }
            }
            // carries into the next pass (from lane 63)
            carry_open = __builtin_amdgcn_readlane((bit7 && act) ? 1 : 0, 63) != 0;
            carry_t = (uint32_t)__builtin_amdgcn_readlane((int)t_out, 63);
            carry_jr = __builtin_amdgcn_readlane(jr, 63);
        }
        if (carry_open && lane == 0) atomicAdd(&W.sum[carry_jr], carry_t);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    // ---- phase C
    if (!have || BRC_AVAR(4)) return;
    // everything phase C reads from memory whose address is known by now, requested together (each load left to the place of its
    // use is one more dependent round trip for the wave: lane = read here, nothing else hides it)
    const uint32_t mapq_l = in.mapq[my], tags_l = in.tags[my];

    const int lib_l = c.per_lib ? (int)in.lib[my] : 0;

    const uint32_t cig1_l = nc > 1u ? cigar_ro[coff + 1u] : 0u, cig2_l = nc > 2u ? cigar_ro[coff + 2u] : 0u;
    const bool q2_pre = work_me && L >= 8;             // the 8 qualities at the read's 3' end: as a rule all the Q2 scan needs
    unsigned long long q2_w0 = 0ull;
    if (q2_pre) __builtin_memcpy(&q2_w0, BRC_CK(c, CK_ANNOTATE, 8, CB_QUAL, qual_ro + qoff + ((flag & FREVERSE) ? 0 : L - 8), 8, my, -1), 8);
    DRead r;
    bool serial = fallback;
    bool wide = work_me && W.wide[rank] != 0u;        // an escape byte in the read's row
    uint32_t my_sum = 0; int my_hi = -1, my_lo = -1;
    if (work_me && W.redo[rank]) serial = true;
    if (work_me && !serial) {
        my_sum = W.sum[rank];
        // first / last base with quality != 2, whichever the strand needs (:201-238): a scan from the read's 3' end, 8 bases
        // per load (as a rule the first load decides)
        const uint8_t* q = BRC_CK(c, CK_ANNOTATE, 9, CB_QUAL, qual_ro + qoff, (uint64_t)L, my, -1);       // (the scan below stays inside [0, L))
        if (flag & FREVERSE) {
            for (int k = 0; k < L && my_lo < 0; k += 8) {
                unsigned long long w; const int nv = L - k < 8 ? L - k : 8;
                if (k == 0 && q2_pre) w = q2_w0; else if (nv == 8) __builtin_memcpy(&w, q + k, 8); else { w = 0x0202020202020202ull; for (int j = 0; j < nv; ++j) w = (w & ~(0xffull << (8 * j))) | ((unsigned long long)q[k + j] << (8 * j)); }
                const unsigned long long x = w ^ 0x0202020202020202ull;                    // zero bytes: quality 2
                if (x) my_lo = k + (__builtin_ctzll(x) >> 3);
            }
        } else {
            for (int k = L; k > 0 && my_hi < 0; k -= 8) {
                unsigned long long w; const int lo8 = k - 8;
                if (k == L && q2_pre) w = q2_w0; else if (lo8 >= 0) __builtin_memcpy(&w, q + lo8, 8); else { w = 0x0202020202020202ull; for (int j = 0; j < k; ++j) w = (w & ~(0xffull << (8 * (j - lo8)))) | ((unsigned long long)q[j] << (8 * (j - lo8))); }
                const unsigned long long x = w ^ 0x0202020202020202ull;
                if (x) my_hi = lo8 + ((63 - __builtin_clzll(x)) >> 3);
            }
        }
    }
    if (serial) {
        r = annotate_read(c, in, my, eb, nullptr, wide);
    } else {
        const bool rev = (flag & FREVERSE) != 0;
        int tp, q2;
        if (rev) { tp = 0; if (tp < left_clip) tp = left_clip; q2 = my_lo >= 0 ? my_lo - 1 : -1; if (tp < q2) tp = q2; }
        else { tp = L - 1; if (tp > right_clip) tp = right_clip; q2 = my_hi >= 0 ? my_hi - 1 : -1; if (tp > q2 && q2 != -1) tp = q2; }
        const uint32_t mapq = mapq_l; const uint32_t tags = tags_l;
        r.pos = pos; r.end = dropped ? pos : pos + rlen;
        r.cig_off = coff; r.n_cigar = nc; r.bq_off = brow;
        const int lib = lib_l;
        uint32_t misc = (mapq << 8) | ((uint32_t)((lib + 1) & 0xff) << 16);
        if (rev) misc |= M_REV;
        if (flag & BRC_NOCOUNT_MASK) misc |= M_NOCOUNT;
        if (q2 > -1) misc |= M_Q2OK;
        if (simple) misc |= M_SIMPLE;
        if (tot_d + tot_is <= STAGE_SLACK) misc |= M_STAGED | ((uint32_t)tot_d << 24);
        uint32_t sse;
        if (flag & FPROPER_PAIR) { if (tags & 2u) sse = (uint32_t)in.sm[my]; else { sse = 0; misc |= M_SMW; } } else sse = mapq;
        float snm = 0.0f;
        if (tags & 1u) snm = (float)in.nm[my] / (float)clipped; else misc |= M_NMW;
        r.misc = finish_misc(misc, c.table_len > 0 && L == c.table_len && clipped == L, shape_clipm(shape, nc, left_clip)); r.l_qseq = L; r.q2 = q2; r.tp = tp; r.left = left_clip; r.clipped = clipped;
        r.zm_sum = my_sum; r.sse_add = sse; r.snm_add = snm; r.clipped_dup = clipped;
    }
    if (nc >= 2u) *BRC_CK(c, CK_ANNOTATE, 10, CB_READS, reads + my, sizeof(DRead), my, -1) = r;          // only the indel side path reads these records, and only for reads with an indel operator
    // The read's first three operators in registers for the two walks below (loaded at the start of the phase; the
    // first operator is still there from phase A): each cig[k] from memory inside a walk is a dependent round trip for the whole wave.
    struct CigRegs {
        uint32_t c0, c1, c2; const uint32_t* p;
        __device__ __forceinline__ uint32_t operator()(uint32_t k) const { return k == 0u ? c0 : k == 1u ? c1 : k == 2u ? c2 : p[k]; }
    } cigr;
    cigr.p = BRC_CK(c, CK_ANNOTATE, 11, CB_CIGAR, cigar_ro + coff, 4ull * nc, my, -1); cigr.c0 = cig0;
    cigr.c1 = cig1_l; cigr.c2 = cig2_l;
    {   // the read's pieces (the host counted them with the same walk_pieces: piece_off[] are their slots)
        const bool nolib = c.per_lib && lib_l < 0;
        const bool enters = r.end > r.pos && pos >= 0;
        const ReadConst rc = read_const(c, r, (uint32_t)my, wide);
        uint32_t slot = piece_off[my];
        walk_pieces_at(c.insertion_centric != 0, enters && !nolib, rc.counts, pos, cigr, nc, [&](int32_t rs, int32_t len, int32_t ext, int qoff, bool nb) {
            Piece h; PieceRare rr;
            make_piece(c, rc, rs, len, ext, qoff, nb, h, rr, SH);
            if (!BRC_AVAR(2)) {
                typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
                Piece* const pdst = BRC_CK(c, CK_ANNOTATE, 12, CB_PIECES, pieces + slot, sizeof(Piece), my, slot);
                const u32x4s* hs = reinterpret_cast<const u32x4s*>(&h); u32x4s* hd = reinterpret_cast<u32x4s*>(pdst);
                if (one_stream) { __builtin_nontemporal_store(hs[0], hd); __builtin_nontemporal_store(hs[1], hd + 1); __builtin_nontemporal_store(hs[2], hd + 2); }
                else *pdst = h;
                if (piece_has_rare(piece_flags(h))) *BRC_CK(c, CK_ANNOTATE, 13, CB_RARE, rare + slot, sizeof(PieceRare), my, slot) = rr;
                *BRC_CK(c, CK_ANNOTATE, 14, CB_KEYREACH, keyreach + slot, sizeof(int2), my, slot) = make_int2(pos, rs + ext);
            }
            ++slot;
        });
    }
    if (ev_raw && n_idp) {
        // indel events of this read (bamreadcount.cpp:315-342), written to the read's own slots of the raw list (the host
        // counted one slot per I / D / P operator: no cursor, no atomics on the list) and counted per (16 or 64 positions, library) bucket;
        // slots the read does not use are marked empty
        const int lib = (int)((r.misc >> 16) & 0xffu) - 1;
        IndelEv* slot = BRC_CK(c, CK_ANNOTATE, 15, CB_EVRAW, ev_raw + in.iev_off[my], sizeof(IndelEv) * (uint64_t)n_idp, my, -1); uint32_t used = 0;
        enumerate_indels_at(c, cigr, r, qual_ro + qoff, [&](int32_t p, int qpos, int len) {
            IndelEv e; e.read = (uint32_t)my; e.qpos = qpos; e.len = len; e.key_lo = (uint32_t)((int64_t)(p - c.pos0) * c.Lp + lib);   // (keys fit 32 bits: checked at upload)
            if (wanted && !tile_wants(*BRC_CK(c, CK_ANNOTATE, 16, CB_WANTED, wanted + ((uint32_t)(p - c.pos0) >> 6), 2, my, p), (uint32_t)(p - c.pos0) & 63u)) return;       // what no announced window touches comes back empty: no indel alleles either
            if (used < n_idp) { slot[used++] = e; atomicAdd(BRC_CK(c, CK_ANNOTATE, 17, CB_CNT, bucket_cnt + indel_bucket_of(c, (uint32_t)(p - c.pos0), (uint32_t)lib), 4, my, p), 1u); }
        });
        for (; used < n_idp; ++used) slot[used].key_lo = NONE32;
    }
}

// K1w, wave form: ONE WAVE PER READ, for the reads K1 would hand to the serial annotate_read() only because their CIGAR has more
// than two M operators (long reads with an indel every few bases — ONT / CLR / HiFi — and the few short reads with several
// indels).  There a lane of K1 walks one read base by base and operator by operator, every step a dependent round trip for its
// whole wave: 35-46 ms for ANY region of 3-10-kb reads with ~800 operators each, whatever its size.  Here the lanes of a wave
// share one read:
//   pass 1 (lane = operator)   running read / reference cursors by wave prefix sums; the M operators go to an LDS list
//                              {query start, reference start, length | "an insertion follows"}; clips and totals; the read's
//                              indel events (enumerate_indels_at, restated per operator) into its slots of the raw list;
//   pass 2 (lane = base)       operator of the base by binary search in the LDS list (64-entry window that moves with the
//                              pass), reference code, mismatch flag (:152), event byte + wide words (eb_make), the
//                              quality != 2 scan (:201-238), mismatch-run maxima (:152-172,199) by a segmented max-scan over the lanes
//                              with a carry between passes;
//   pass 3 (lane = M operator) the pieces (walk_pieces_at, restated per M operator: the next operator's reference start is
//                              the neighbour's list entry) through make_piece.
// K1 picks the reads (phase A knows everything the choice needs) and appends them to `wave_list`; this kernel is a fixed grid
// whose waves take list entries round robin — no host round trip for the count.  A read with a NUL reference character under
// an M base (the annotator's break, :151) is re-annotated by annotate_read() on one lane; pieces and indel events do not
// depend on it.  Eligible: mapped-and-pushed reads with bases, inside the reference, 3..AW_MCAP_HUGE M operators (three instantiations:
// up to AW_MCAP with four waves per workgroup, up to AW_MCAP_BIG one wave per workgroup, above it one wave per CU), no P / = / X
// operator and no empty M operator (those keep the serial path, as do reads with more M operators than the list holds).
__device__ __forceinline__ uint32_t mbcnt64(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
// Which reads take the wave form: lane = read.  K1 itself is not touched by the choice — a changed K1 is another register allocation, and
// its 80-register budget holds by a hair (tests/test_abi.py) —: it is launched with a COPY of the operator counts in which the chosen
// reads have none, and for a read without operators K1 writes nothing at all (no record, no pieces, no indel slots, no event bytes).
// (wave_list: the reads of the four-wave instantiation from the front, those of the one-wave instantiation from the back, list_cap - 1 downwards;
// those of the one-wave-per-CU instantiation in a second list behind it, from list_cap + 16 on)
// The sparse wide stream (DevIn.bqw) of the reads the host found an escape base in (Staged::wide_layout: the read, where its wide row
// starts in units of 16 elements — ascending in the list): the table entry of the chunk a read's byte row starts with, then the words
// quality << 8 | bucket of all its bases.  K1 — whatever its form — only marks such a read's pieces PF_WIDE.  A wave takes 64
// consecutive reads of the list and walks THEIR ROWS' 16-element chunks, a lane per chunk (short reads — every read of a run with
// qualities above 62 — fill the lanes like long ones; a wave per read took 3.5 ms for ten million 150-base reads): the chunk's owner by
// a search over the lanes' row starts, 16 qualities and 8 bytes of base codes in, two 16-byte stores out.
__global__ __launch_bounds__(256) void k_wide_rows(DevCfg c, DevIn in, const uint2* __restrict__ pairs, uint32_t n, uint16_t* __restrict__ bqw) {      // gridDim.y: waves that share a group's chunks (long reads)
    struct Par { uint32_t w16, L; uint64_t qoff, soff; };
    __shared__ Par par_all[4][64];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t first = (blockIdx.x * 4u + wv) * 64u;
    if (first >= n) return;
    Par* const par = par_all[wv];
    const uint32_t nr = n - first < 64u ? n - first : 64u;
    uint32_t w16_me = 0xffffffffu, cnt_me = 0u;
    if (lane < nr) {
        const uint2 p = pairs[first + lane];
        const int64_t i = (int64_t)p.x;
        const int32_t L = in.l_qseq[i];
        if (blockIdx.y == 0) *BRC_CK(c, CK_ANNOTATE, 51, CB_BQW, reinterpret_cast<uint32_t*>(bqw) + (in.bq_row[i] >> 4), 4, i, -1) = p.y;
        Par q; q.w16 = p.y; q.L = (uint32_t)L; q.qoff = in.qual_off[i]; q.soff = in.seq_off[i];
        par[lane] = q;
        w16_me = p.y; cnt_me = ((uint32_t)L + 15u) >> 4;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    const uint32_t c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)w16_me);
    const uint32_t c1 = (uint32_t)__builtin_amdgcn_readlane((int)(w16_me + cnt_me), (int)(nr - 1u));      // (rows follow each other in list order)
    for (uint32_t cb = c0 + 64u * blockIdx.y; cb < c1; cb += 64u * gridDim.y) {
        const uint32_t ch = cb + lane;
        uint32_t idx = 0u;                                   // the last lane whose row starts at or before the chunk (every lane takes part in the shuffles: a lane that has left reads as 0)
#pragma unroll
        for (uint32_t st = 32u; st; st >>= 1) { const uint32_t cand = idx + st; const uint32_t v = (uint32_t)__shfl((int)w16_me, (int)(cand & 63u), 64); if (cand < 64u && v <= ch) idx = cand; }
        if (ch >= c1) continue;
        const Par q = par[idx];
        const uint32_t j0 = (ch - q.w16) << 4;
        if (j0 >= q.L) continue;
        uint4 Q; uint2 S;
        __builtin_memcpy(&Q, BRC_CK(c, CK_ANNOTATE, 52, CB_QUAL, in.qual + q.qoff + j0, 16, (int64_t)pairs[first + idx].x, (int32_t)j0), 16);       // (the arenas are padded by 16 bytes: a row's last chunk may read past its read)
        __builtin_memcpy(&S, BRC_CK(c, CK_ANNOTATE, 53, CB_SEQ, in.seq4 + q.soff + (j0 >> 1), 8, (int64_t)pairs[first + idx].x, (int32_t)j0), 8);
        const uint32_t qd[4] = {Q.x, Q.y, Q.z, Q.w}; const uint32_t sd[2] = {S.x, S.y};
        uint32_t out[8];
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
            const uint32_t sb = (sd[k >> 3] >> (((k >> 1) & 3) << 3)) & 0xffu;           // the byte with bases k (high nibble) and k + 1
            const uint32_t q0 = (qd[k >> 2] >> ((k & 3) << 3)) & 0xffu, q1 = (qd[(k + 1) >> 2] >> (((k + 1) & 3) << 3)) & 0xffu;
            out[k >> 1] = ((q0 << 8) | canon_bucket(sb >> 4)) | (((q1 << 8) | canon_bucket(sb & 15u)) << 16);
        }
        uint4* const dst = reinterpret_cast<uint4*>(BRC_CK(c, CK_ANNOTATE, 54, CB_BQW, bqw + ((uint64_t)ch << 4), 32, (int64_t)pairs[first + idx].x, (int32_t)j0));
        dst[0] = make_uint4(out[0], out[1], out[2], out[3]); dst[1] = make_uint4(out[4], out[5], out[6], out[7]);
    }
}

__global__ __launch_bounds__(256) void k_pick_wave(DevCfg c, DevIn in, uint32_t* __restrict__ n_cigar_k1, uint32_t* __restrict__ wave_list, uint32_t list_cap,
                                                   unsigned int* __restrict__ wave_n, unsigned int* __restrict__ wave_n_big, unsigned int* __restrict__ wave_n_huge,
                                                   int wave_on, int cursor_on, unsigned int* __restrict__ cursor_n,
                                                   unsigned int* __restrict__ wave_n_eqx, unsigned int* __restrict__ wave_n_eqx_big) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool pick = false, big = false, huge = false, lit = false, eqx = false, eqx_big = false; uint32_t nc_me = 0u;
    if (i < c.n_reads) {
        const uint32_t nc = in.n_cigar[i]; nc_me = nc;
        const int32_t pos = in.pos[i], L = in.l_qseq[i];
        // a mapped read with an M / = / X operator of length zero (the host saw one in this region): third list, k_annotate_cursor
        if (cursor_on && nc > 0u && !(in.flag[i] & FUNMAP)) lit = has_empty_mop(BRC_CK(c, CK_ANNOTATE, 45, CB_CIGAR, in.cigar + in.cig_off[i], 4ull * nc, i, -1), nc);
        if (!lit && wave_on && nc >= 3u && L > 0 && pos >= 0 && c.has_ref && !(in.flag[i] & BRC_PUSH_MASK)) {       // (three match operators: five operators with M alone, three with = / X)
            const uint32_t* cig = BRC_CK(c, CK_ANNOTATE, 44, CB_CIGAR, in.cigar + in.cig_off[i], 4ull * nc, i, -1);
            int64_t rlen = 0; uint32_t n_m = 0, n_ml = 0; bool irregular = false, has_eqx = false;
            for (uint32_t k = 0; k < nc; ++k) {
                const uint32_t cg = cig[k], op = cg & 0xfu, len = cg >> 4;
                if (op == (uint32_t)CPAD || op > (uint32_t)CDIFF || (is_mop(op) && len == 0u)) irregular = true;  // P, unknown codes, an empty match operator: the serial path (or the cursor) restates those
                if (op == CMATCH) ++n_m;
                if (op == CEQUAL || op == CDIFF) has_eqx = true;
                if (is_mop(op)) ++n_ml;
                if (is_refop(op)) rlen += len;
            }
            const bool fits = !irregular && (int64_t)pos + rlen <= c.ref_len && rlen < 0x7fffffffll;
            if (has_eqx) {          // = / X beside anything: the instantiation that keeps the annotator's own cursors (k_annotate_wave<.., EQX>)
                eqx = fits && n_ml > 2u && n_ml <= (uint32_t)AW_MCAP_EQX_BIG;
                eqx_big = eqx && n_ml > (uint32_t)AW_MCAP_EQX;
            } else {
                pick = fits && n_m > 2u && n_m <= (uint32_t)AW_MCAP_HUGE;
                huge = pick && n_m > (uint32_t)AW_MCAP_BIG;
                big = pick && !huge && n_m > (uint32_t)AW_MCAP;
            }
        }
    }
    const int lane = threadIdx.x & 63;
    const unsigned long long pm = __ballot(pick && !big && !huge), pb = __ballot(big), ph = __ballot(huge);
    if (ph) {
        uint32_t base = 0u;
        if (lane == __builtin_ctzll(ph)) base = atomicAdd(wave_n_huge, (unsigned int)__builtin_popcountll(ph));
        base = (uint32_t)__shfl((int)base, __builtin_ctzll(ph), 64);
        if (huge) wave_list[list_cap + 16u + base + mbcnt64(ph)] = (uint32_t)i;
    }
    if (pm) {
        uint32_t base = 0u;
        if (lane == __builtin_ctzll(pm)) base = atomicAdd(wave_n, (unsigned int)__builtin_popcountll(pm));
        base = (uint32_t)__shfl((int)base, __builtin_ctzll(pm), 64);
        if (pick && !big && !huge) wave_list[base + mbcnt64(pm)] = (uint32_t)i;
    }
    if (pb) {
        uint32_t base = 0u;
        if (lane == __builtin_ctzll(pb)) base = atomicAdd(wave_n_big, (unsigned int)__builtin_popcountll(pb));
        base = (uint32_t)__shfl((int)base, __builtin_ctzll(pb), 64);
        if (big) wave_list[list_cap - 1u - (base + mbcnt64(pb))] = (uint32_t)i;
    }
    const unsigned long long pl = __ballot(lit);
    if (pl) {
        uint32_t base = 0u;
        if (lane == __builtin_ctzll(pl)) base = atomicAdd(cursor_n, (unsigned int)__builtin_popcountll(pl));
        base = (uint32_t)__shfl((int)base, __builtin_ctzll(pl), 64);
        if (lit) wave_list[2u * list_cap + 32u + base + mbcnt64(pl)] = (uint32_t)i;
    }
    const unsigned long long pe = __ballot(eqx && !eqx_big), peb = __ballot(eqx_big);
    if (pe) {
        uint32_t base = 0u;
        if (lane == __builtin_ctzll(pe)) base = atomicAdd(wave_n_eqx, (unsigned int)__builtin_popcountll(pe));
        base = (uint32_t)__shfl((int)base, __builtin_ctzll(pe), 64);
        if (eqx && !eqx_big) wave_list[3u * list_cap + 48u + base + mbcnt64(pe)] = (uint32_t)i;
    }
    if (peb) {
        uint32_t base = 0u;
        if (lane == __builtin_ctzll(peb)) base = atomicAdd(wave_n_eqx_big, (unsigned int)__builtin_popcountll(peb));
        base = (uint32_t)__shfl((int)base, __builtin_ctzll(peb), 64);
        if (eqx_big) wave_list[4u * list_cap + 47u - (base + mbcnt64(peb))] = (uint32_t)i;
    }
    if (i < c.n_reads) n_cigar_k1[i] = (pick || lit || eqx) ? 0u : nc_me;
}
// Reads with an empty M / = / X operator (brc_core.h: cursor_resolve): one lane per read — fetch_func's annotation by annotate_read() (the
// annotator walks the CIGAR operator by operator: an empty operator is an empty loop, bamreadcount.cpp:133-197), the read's segments and indel
// events by the iterator's own cursor, column by column (walk_pieces / enumerate_indels switch to it).  Exact and slow; no aligner writes such
// records, the reference piles them up all the same.
template <int SH>
__global__ __launch_bounds__(64) void k_annotate_cursor(DevCfg c, DevIn in, const uint32_t* __restrict__ list, const unsigned int* __restrict__ list_n,
                                                        DRead* __restrict__ reads, const uint32_t* __restrict__ piece_off, Piece* __restrict__ pieces, PieceRare* __restrict__ rare,
                                                        int2* __restrict__ keyreach, uint8_t* __restrict__ eb, IndelEv* __restrict__ ev_raw,
                                                        uint32_t* __restrict__ bucket_cnt, const uint16_t* __restrict__ wanted) {
    c.pack_shift = SH;
    const uint32_t n = *list_n;
    for (uint32_t li = blockIdx.x * 64u + threadIdx.x; li < n; li += gridDim.x * 64u) {
        const int64_t my = (int64_t)list[li];
        bool wide = false;
        const DRead r = annotate_read(c, in, my, eb, nullptr, wide);
        *BRC_CK(c, CK_ANNOTATE, 46, CB_READS, reads + my, sizeof(DRead), my, -1) = r;
        const uint32_t nc = in.n_cigar[my]; const uint32_t* cig = in.cigar + in.cig_off[my];
        const int lib = c.per_lib ? (int)in.lib[my] : 0;
        const bool nolib = c.per_lib && lib < 0;
        const bool enters = r.end > r.pos && r.pos >= 0;
        const ReadConst rc = read_const(c, r, (uint32_t)my, wide);
        uint32_t slot = piece_off[my];
        walk_pieces(c.insertion_centric != 0, enters && !nolib, rc.counts, r.pos, cig, nc, [&](int32_t rs, int32_t len, int32_t ext, int qoff, bool nb) {
            Piece h; PieceRare rr;
            make_piece(c, rc, rs, len, ext, qoff, nb, h, rr, SH);
            *BRC_CK(c, CK_ANNOTATE, 47, CB_PIECES, pieces + slot, sizeof(Piece), my, slot) = h;
            if (piece_has_rare(piece_flags(h))) *BRC_CK(c, CK_ANNOTATE, 48, CB_RARE, rare + slot, sizeof(PieceRare), my, slot) = rr;
            *BRC_CK(c, CK_ANNOTATE, 49, CB_KEYREACH, keyreach + slot, sizeof(int2), my, slot) = make_int2(r.pos, rs + ext);
            ++slot;
        });
        if (ev_raw) {
            uint32_t n_idp = 0;
            for (uint32_t k = 0; k < nc; ++k) { const uint32_t op = cig[k] & 0xfu; if (op == CINS || op == CDEL || op == CPAD) ++n_idp; }
            if (n_idp) {
                IndelEv* es = BRC_CK(c, CK_ANNOTATE, 50, CB_EVRAW, ev_raw + in.iev_off[my], sizeof(IndelEv) * (uint64_t)n_idp, my, -1); uint32_t used = 0;
                enumerate_indels(c, in, r, in.qual + in.qual_off[my], [&](int32_t p, int qpos, int len) {
                    IndelEv e; e.read = (uint32_t)my; e.qpos = qpos; e.len = len; e.key_lo = (uint32_t)((int64_t)(p - c.pos0) * c.Lp + lib);
                    if (wanted && !tile_wants(wanted[(uint32_t)(p - c.pos0) >> 6], (uint32_t)(p - c.pos0) & 63u)) return;
                    if (used < n_idp) { es[used++] = e; atomicAdd(bucket_cnt + indel_bucket_of(c, (uint32_t)(p - c.pos0), (uint32_t)lib), 1u); }
                });
                for (; used < n_idp; ++used) es[used].key_lo = NONE32;
            }
        }
    }
}
__device__ __forceinline__ int32_t wave_incl_sum(int32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int32_t o = __shfl_up(v, d, 64); if (lane >= d) v += o; }
    return v;
}
// EQX: reads with = / X operators.  fetch_func's operator loop has no branch for them (bamreadcount.cpp:133-197): NEITHER of its cursors moves,
// so the M operators behind them are compared at the annotator's own query / reference offsets — the true ones less the = / X bases before —,
// while the pileup iterator treats = and X like M.  The list then holds every match operator (M, =, X) with its TRUE offsets (the segments of
// pass 3), the annotator's query offset beside it (its reference offset is the true one less the same difference), and a bit that says "M":
// pass 2 finds a base's operator by the annotator's offsets and only an M operator makes it a compared base.
template <int SH, int MCAP, int WAVES, bool EQX = false>      // MCAP: match operators the LDS list holds; WAVES per workgroup (AW_MCAP x 4, or AW_MCAP_BIG x 1); list_step: +1 / -1 (the big reads are listed from the back)
__global__ __launch_bounds__(WAVES * 64) void k_annotate_wave(DevCfg c, DevIn in, const uint32_t* __restrict__ wave_list, int list_step, const unsigned int* __restrict__ wave_n,
                                                       DRead* __restrict__ reads, const uint32_t* __restrict__ piece_off,
                                                       Piece* __restrict__ pieces, PieceRare* __restrict__ rare, int2* __restrict__ keyreach,
                                                       uint8_t* __restrict__ eb, IndelEv* __restrict__ ev_raw, uint32_t* __restrict__ bucket_cnt,
                                                       const uint8_t* __restrict__ refcode, const uint16_t* __restrict__ wanted) {
    c.pack_shift = SH;
    struct WaveLds { int32_t y[MCAP]; int32_t x[MCAP]; uint32_t lw[MCAP]; int32_t ya[EQX ? MCAP : 1]; DRead r; uint32_t wide; };
    constexpr uint32_t LW_LEN = EQX ? 0x0fffffffu : 0x7fffffffu, LW_M = 1u << 30;      // (a CIGAR length has 28 bits)
    __shared__ WaveLds lds_all[WAVES];
    const int lane = threadIdx.x & 63;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    WaveLds& W = lds_all[wv];
    const uint32_t n_list = *wave_n;
    const uint32_t nwaves = gridDim.x * (uint32_t)WAVES;
    const int64_t ref_n = c.ref_hi - c.ref_lo;
    for (uint32_t li = blockIdx.x * (uint32_t)WAVES + wv; li < n_list; li += nwaves) {
        const int64_t my = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)wave_list[(int64_t)list_step * (int64_t)li]);
        const int32_t pos = __builtin_amdgcn_readfirstlane(in.pos[my]);
        const uint32_t flag = (uint32_t)__builtin_amdgcn_readfirstlane((int)in.flag[my]);
        const int32_t L = __builtin_amdgcn_readfirstlane(in.l_qseq[my]);
        const uint32_t nc = (uint32_t)__builtin_amdgcn_readfirstlane((int)in.n_cigar[my]);
        const uint32_t mapq = (uint32_t)__builtin_amdgcn_readfirstlane((int)in.mapq[my]);
        const uint32_t tags = (uint32_t)__builtin_amdgcn_readfirstlane((int)in.tags[my]);
        const int lib = c.per_lib ? __builtin_amdgcn_readfirstlane((int)in.lib[my]) : 0;
        const uint64_t qoff = in.qual_off[my], soff = in.seq_off[my], brow = in.bq_row[my];
        const uint32_t coff = (uint32_t)in.cig_off[my];
        const uint32_t* const cig = BRC_CK(c, CK_ANNOTATE, 30, CB_CIGAR, in.cigar + coff, 4ull * nc, my, -1);
        const uint8_t* const qual = BRC_CK(c, CK_ANNOTATE, 31, CB_QUAL, in.qual + qoff, (uint64_t)L, my, -1);
        const uint8_t* const seq = BRC_CK(c, CK_ANNOTATE, 32, CB_SEQ, in.seq4 + soff, (uint64_t)((L + 1) / 2), my, -1);
        uint8_t* const eb_row = BRC_CK(c, CK_ANNOTATE, 33, CB_EB, eb + brow, (uint64_t)L, my, -1);
        const bool nocount = (flag & BRC_NOCOUNT_MASK) != 0u;
        const bool ev_on = ev_raw && !(c.per_lib && lib < 0) && (int)mapq >= c.min_mapq && !nocount;
        IndelEv* const ev_slots = ev_raw ? ev_raw + in.iev_off[my] : nullptr;

        // ---- pass 1: lane = operator
        int32_t xc = pos, yc = 0, ec = 0; uint32_t mc = 0u, used = 0u, n_idp = 0u;         // ec: = / X bases so far (the annotator's cursors lag by them)
        uint32_t s_part = 0u, d_part = 0u, i_part = 0u; int32_t left_clip = 0;
        for (uint32_t kb = 0; kb < nc; kb += 64u) {
            const uint32_t k = kb + (uint32_t)lane; const bool on = k < nc;
            const uint32_t cg = on ? cig[k] : 0u, c1 = k + 1u < nc ? cig[k + 1u] : 0u;
            const uint32_t op = on ? (cg & 0xfu) : (uint32_t)CHARD_CLIP; const int32_t len = (int32_t)(cg >> 4);
            const bool isE = EQX && (op == CEQUAL || op == CDIFF);
            const bool isM = op == CMATCH || isE, isI = op == CINS, isD = op == CDEL, isN = op == CREF_SKIP, isS = op == CSOFT_CLIP;      // (isM: a match operator of the iterator's)
            const int32_t ql = (isM || isI || isS) ? len : 0, rl = (isM || isD || isN) ? len : 0;
            const int32_t qi = wave_incl_sum(ql, lane), ri = wave_incl_sum(rl, lane);
            const int32_t y = yc + qi - ql, x = xc + ri - rl;
            int32_t e_before = 0;
            if (EQX) { const int32_t el = isE ? len : 0; const int32_t ei = wave_incl_sum(el, lane); e_before = ec + ei - el; ec += __builtin_amdgcn_readlane(ei, 63); }
            if (isS) { s_part += (uint32_t)len; if (k == 0u) left_clip = len; }
            if (isD || isN) d_part += (uint32_t)len;
            if (isI || isS) i_part += (uint32_t)len;
            const unsigned long long mb = __ballot(isM);
            if (isM) {
                const uint32_t mr = mc + mbcnt64(mb);                                   // (k_pick_wave lists only reads with at most MCAP M operators)
                if (mr < (uint32_t)MCAP) {
                    W.y[mr] = y; W.x[mr] = x; W.lw[mr] = (uint32_t)len | (((c1 & 0xfu) == CINS && (c1 >> 4) > 0u) ? 0x80000000u : 0u) | ((EQX && !isE) ? LW_M : 0u);
                    if (EQX) W.ya[mr] = y - e_before;
                }
            }
            mc += (uint32_t)__builtin_popcountll(mb);
            n_idp += (uint32_t)__builtin_popcountll(__ballot(isI || isD));
            if (ev_on) {
                // (bamreadcount.cpp:288-342, as enumerate_indels_at: the last base of an M operator followed by D or I, inside the
                // processing window, passing -b)
                int32_t indel = 0;
                if (isM && k + 1u < nc) { const uint32_t op2 = c1 & 0xfu; const int32_t l2 = (int32_t)(c1 >> 4); if (op2 == CDEL) indel = -l2; else if (op2 == CINS) indel = l2; }
                const int32_t p = x + len - 1; const int32_t qp = y + len - 1;
                bool emit = false;
                if (indel != 0 && p >= c.beg0 - 1 && p < c.end && p >= c.pos0 && (int64_t)p < (int64_t)c.pos0 + c.P) {
                    emit = (int)qual[qp] >= c.min_bq;
                    if (emit && wanted && !tile_wants(*BRC_CK(c, CK_ANNOTATE, 35, CB_WANTED, wanted + ((uint32_t)(p - c.pos0) >> 6), 2, my, p), (uint32_t)(p - c.pos0) & 63u)) emit = false;
                }
                const unsigned long long em = __ballot(emit);
                if (emit) {
                    IndelEv e; e.read = (uint32_t)my; e.qpos = qp; e.len = indel; e.key_lo = (uint32_t)((int64_t)(p - c.pos0) * c.Lp + lib);
                    *BRC_CK(c, CK_ANNOTATE, 36, CB_EVRAW, ev_slots + used + mbcnt64(em), sizeof(IndelEv), my, p) = e;
                    atomicAdd(BRC_CK(c, CK_ANNOTATE, 37, CB_CNT, bucket_cnt + indel_bucket_of(c, (uint32_t)(p - c.pos0), (uint32_t)lib), 4, my, p), 1u);
                }
                used += (uint32_t)__builtin_popcountll(em);
            }
            yc += __builtin_amdgcn_readlane(qi, 63); xc += __builtin_amdgcn_readlane(ri, 63);
        }
        if (ev_raw) for (uint32_t u = used + (uint32_t)lane; u < n_idp; u += 64u) BRC_CK(c, CK_ANNOTATE, 38, CB_EVRAW, ev_slots + u, sizeof(IndelEv), my, -1)->key_lo = NONE32;
        const int32_t rlen = xc - pos;
        const uint32_t s_tot = wave_sum_u32(s_part), tot_d = wave_sum_u32(d_part), tot_is = wave_sum_u32(i_part);
        left_clip = __builtin_amdgcn_readfirstlane(left_clip);
        const int32_t clipped = L - (int32_t)s_tot;
        const int32_t right_clip = L - ((int32_t)s_tot - left_clip);
        const uint32_t nm_ops = mc < (uint32_t)MCAP ? mc : (uint32_t)MCAP;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");                       // the list is read by other lanes below

        // ---- pass 2: lane = base
        uint32_t sum_part = 0u; bool carry_open = false; uint32_t carry_t = 0u;
        int32_t my_lo = -1, my_hi = -1; bool wide = false, redo = false;
        uint32_t mlo = 0u;                                                            // M operators starting at or before the pass's first base, less one window
        for (int32_t jb = 0; jb < L; jb += 64) {
            const int32_t j = jb + lane; const bool valid = j < L;
            const int32_t jc = valid ? j : L - 1;
            const uint32_t q = qual[jc];
            const uint32_t nib = (seq[jc >> 1] >> ((~jc & 1) << 2)) & 0xfu;
            // number of list entries with query start <= j (entries are sorted; at most 64 start inside a pass)
            // (EQX: an = / X entry takes no query base of the annotator's — any number of entries may share an offset —: the whole list is searched)
            uint32_t cnt = EQX ? 0u : mlo;
#pragma unroll
            for (uint32_t st = EQX ? 4096u : 64u; st > 0u; st >>= 1) { const uint32_t t = cnt + st; if (t <= nm_ops && (EQX ? W.ya[t - 1u] : W.y[t - 1u]) <= jc) cnt = t; }
            mlo = (uint32_t)__builtin_amdgcn_readlane((int)cnt, 0);                   // (every later base has at least lane 0's count)
            bool in_m = false; uint32_t rcode = 0x0fu;
            if (cnt > 0u) {
                const uint32_t m = cnt - 1u; const int32_t y0 = EQX ? W.ya[m] : W.y[m];
                const uint32_t lwm = W.lw[m];
                const int32_t ln = (!EQX || (lwm & LW_M)) ? (int32_t)(lwm & LW_LEN) : 0;       // (an = / X operator compares nothing)
                if (valid && jc - y0 < ln) {
                    in_m = true;
                    const int64_t ri = (int64_t)W.x[m] - (EQX ? (int64_t)(W.y[m] - y0) : 0) - c.ref_lo + (jc - y0);
                    if (ri < 0 || ri >= ref_n) rcode = 0x8fu;                         // outside the uploaded slice: ref_at() gives 0 there — the serial path decides
                    else rcode = *BRC_CK(c, CK_ANNOTATE, 39, CB_REFCODE, refcode + ri, 1, my, jc);
                }
            }
            if (__ballot(in_m && (rcode & 0x80u))) redo = true;
            const uint32_t refb = rcode & 0xfu;
            const bool mm = in_m && nib != refb && refb != 15u && nib != 0u;          // :152
            // event byte
            bool esc; const uint32_t bucket = canon_bucket(nib);
            const uint32_t byte = eb_make(c.min_bq, q, bucket, esc);
            if (valid) eb_row[j] = (uint8_t)byte;
            if (__ballot(esc && valid)) wide = true;                                  // (its words are in the wide stream: k_wide_rows)
            // first / last base with quality != 2
            const unsigned long long nz = __ballot(valid && q != 2u);
            if (nz) { if (my_lo < 0) my_lo = jb + __builtin_ctzll(nz); my_hi = jb + 63 - __builtin_clzll(nz); }
            // mismatch qualities: every run of read-adjacent mismatches adds its maximum
            const unsigned long long M = __ballot(mm);
            if (carry_open && !(M & 1ull)) { if (lane == 0) sum_part += carry_t; carry_open = false; }
            if (M) {
                const unsigned long long zb = ~M & ((2ull << lane) - 1ull);           // lanes <= mine without a mismatch
                const int start = zb ? 64 - __builtin_clzll(zb) : 0;                  // first lane of my run (if I am in one)
                uint32_t v = mm ? q : 0u;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)v, d, 64); if (lane - d >= start) v = v > o ? v : o; }
                if (mm && start == 0 && carry_open) v = v > carry_t ? v : carry_t;
                const bool last = mm && lane < 63 && !((M >> (lane + 1)) & 1ull);
                if (last) sum_part += v;
                carry_open = (M >> 63) != 0ull;
                carry_t = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
            }
        }
        if (carry_open && lane == 0) sum_part += carry_t;
        const uint32_t zm_sum = wave_sum_u32(sum_part);

        // ---- the read's record (as K1's phase C) — or, for a read with a NUL reference character under an M base, annotate_read()
        if (lane == 0) {
            DRead r0; bool w0 = wide;
            if (redo) r0 = annotate_read(c, in, my, eb, nullptr, w0);
            else {
                const bool rev = (flag & FREVERSE) != 0;
                int tp, q2;
                if (rev) { tp = 0; if (tp < left_clip) tp = left_clip; q2 = my_lo >= 0 ? my_lo - 1 : -1; if (tp < q2) tp = q2; }
                else { tp = L - 1; if (tp > right_clip) tp = right_clip; q2 = my_hi >= 0 ? my_hi - 1 : -1; if (tp > q2 && q2 != -1) tp = q2; }
                r0.pos = pos; r0.end = pos + rlen; r0.cig_off = coff; r0.n_cigar = nc; r0.bq_off = brow;
                uint32_t misc = (mapq << 8) | ((uint32_t)((lib + 1) & 0xff) << 16);
                if (rev) misc |= M_REV;
                if (nocount) misc |= M_NOCOUNT;
                if (q2 > -1) misc |= M_Q2OK;
                if ((uint64_t)tot_d + (uint64_t)tot_is <= (uint64_t)STAGE_SLACK) misc |= M_STAGED | (tot_d << 24);
                uint32_t sse;
                if (flag & FPROPER_PAIR) { if (tags & 2u) sse = (uint32_t)in.sm[my]; else { sse = 0; misc |= M_SMW; } } else sse = mapq;
                float snm = 0.0f;
                if (tags & 1u) snm = (float)in.nm[my] / (float)clipped; else misc |= M_NMW;
                r0.misc = finish_misc(misc, c.table_len > 0 && L == c.table_len && clipped == L, false); r0.l_qseq = L; r0.q2 = q2; r0.tp = tp; r0.left = left_clip; r0.clipped = clipped;
                r0.zm_sum = zm_sum; r0.sse_add = sse; r0.snm_add = snm; r0.clipped_dup = clipped;
            }
            W.r = r0; W.wide = w0 ? 1u : 0u;
            *BRC_CK(c, CK_ANNOTATE, 40, CB_READS, reads + my, sizeof(DRead), my, -1) = r0;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        const DRead r = W.r;
        const ReadConst rc = read_const(c, r, (uint32_t)my, W.wide != 0u);

        // ---- pass 3: lane = M operator: the pieces (walk_pieces_at; the host counted them with it: piece_off[] are their slots)
        const bool entered = r.end > r.pos && !(c.per_lib && lib < 0);
        const uint32_t slot0 = piece_off[my];
        auto put_piece = [&](uint32_t slot, int32_t rs, int32_t len, int32_t ext, int qo, bool nb) {
            Piece h; PieceRare rr;
            make_piece(c, rc, rs, len, ext, qo, nb, h, rr, SH);
            *BRC_CK(c, CK_ANNOTATE, 41, CB_PIECES, pieces + slot, sizeof(Piece), my, slot) = h;
            if (piece_has_rare(piece_flags(h))) *BRC_CK(c, CK_ANNOTATE, 42, CB_RARE, rare + slot, sizeof(PieceRare), my, slot) = rr;
            *BRC_CK(c, CK_ANNOTATE, 43, CB_KEYREACH, keyreach + slot, sizeof(int2), my, slot) = make_int2(pos, rs + ext);
        };
        if (entered && !rc.counts) { if (lane == 0 && rlen > 0) put_piece(slot0, pos, 0, rlen, 0, false); }
        else if (entered) {
            const int32_t x_first = W.x[0];
            uint32_t ord = 0u;
            if (x_first > pos) { if (lane == 0) put_piece(slot0, pos, 0, x_first - pos, 0, false); ord = 1u; }     // leading deletion / skip: column only
            const bool ic = c.insertion_centric != 0;
            for (uint32_t mb0 = 0; mb0 < nm_ops; mb0 += 64u) {
                const uint32_t m = mb0 + (uint32_t)lane; const bool on = m < nm_ops;
                const uint32_t mi = on ? m : 0u;
                const int32_t x = W.x[mi], y = W.y[mi]; const uint32_t lw = W.lw[mi];
                const int32_t len = (int32_t)(lw & LW_LEN);
                const int32_t xnext = mi + 1u < nm_ops ? W.x[mi + 1u] : pos + rlen;
                const bool split = ic && (lw >> 31) != 0u;
                const int32_t cnt = on ? ((split && len > 1) ? 2 : 1) : 0;
                const int32_t ci = wave_incl_sum(cnt, lane);
                uint32_t slot = slot0 + ord + (uint32_t)(ci - cnt);
                if (on) {
                    if (!split) put_piece(slot, x, len, xnext - x, y, false);
                    else {
                        if (len > 1) { put_piece(slot, x, len - 1, len - 1, y, false); ++slot; }
                        put_piece(slot, x + len - 1, 1, xnext - (x + len - 1), y + len - 1, true);
                    }
                }
                ord += (uint32_t)__builtin_amdgcn_readlane(ci, 63);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");                       // the next read reuses the list
    }
}

// ---------------------------------------------------------------- scans (3-phase: block aggregates, scan of aggregates, apply)

enum { SCAN_T = 256, SCAN_ITEMS = 16, SCAN_CHUNK = SCAN_T * SCAN_ITEMS };

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
    static_assert(sizeof(T) == 4 && SCAN_ITEMS % 4 == 0, "16-byte accesses");
    T v[SCAN_ITEMS];
    T tot = Op::id();
    const bool whole = base + SCAN_ITEMS <= n;       // the thread's 16 elements (64 contiguous, 64-byte aligned bytes) are all inside
    if (whole) {
#pragma unroll
        for (int q = 0; q < SCAN_ITEMS / 4; ++q) {
            const uint4 x = reinterpret_cast<const uint4*>(in + base)[q];
            v[4 * q] = (T)x.x; v[4 * q + 1] = (T)x.y; v[4 * q + 2] = (T)x.z; v[4 * q + 3] = (T)x.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; ++j) v[j] = (base + j < n) ? in[base + j] : Op::id();
    }
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) tot = Op::op(tot, v[j]);
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
    T w[SCAN_ITEMS];
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        if (INCLUSIVE) { run = Op::op(run, v[j]); w[j] = run; }
        else { w[j] = run; run = Op::op(run, v[j]); }
    }
    if (whole) {
#pragma unroll
        for (int q = 0; q < SCAN_ITEMS / 4; ++q)
            reinterpret_cast<uint4*>(out + base)[q] = make_uint4((uint32_t)w[4 * q], (uint32_t)w[4 * q + 1], (uint32_t)w[4 * q + 2], (uint32_t)w[4 * q + 3]);
    } else {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; ++j) if (base + j < n) out[base + j] = w[j];
    }
}

// ---------------------------------------------------------------- tiles

// Piece range [lo, hi) of every 64-position tile (tile_range2 in brc_core.h states it as two binary searches):
//   lo(t) = first piece m with prefmax[m] > p0(t)          (prefmax = running maximum of the pieces' reaches, non-decreasing)
//   hi(t) = first piece m with key[m] > p1(t) = p0(t) + 63 (key = start of the piece's read; reads are sorted by pos)
// Inverted here so that the work is one coalesced pass over the pieces instead of two dependent-load searches per tile:
// piece r owns the tiles whose lo is r (prefmax[r-1] <= p0 < prefmax[r]) and the tiles whose hi is r + 1
// (key[r] <= p1 < key[r+1]) — usually none or one of each; the last piece also covers the tiles past the data.
__device__ __forceinline__ int64_t tiles_ceil_div64(int64_t x) { return x <= 0 ? 0 : (x + (TILE - 1)) / TILE; }

// All libraries in three launches (reduce, scan of the block aggregates, k_tiles_all), without materialising the running maxima.  The piece streams are library-major, so the library of a piece
// never decreases along the array: the running maximum of (library << 32 | reach) IS the running maximum of the reaches
// inside the current library (anything from an earlier library is smaller).  keyreach[m] = {start of the piece's read, reach}.
struct OpMaxU64 { typedef unsigned long long T; static __device__ __forceinline__ T id() { return 0ull; } static __device__ __forceinline__ T op(T a, T b) { return a > b ? a : b; } };
enum { TR_T = 256, TR_ITEMS = 4, TR_CHUNK = TR_T * TR_ITEMS };    // a thread owns 4 consecutive pieces (two 16-byte loads), a block 1024
__device__ __forceinline__ int lib_of_slot(const int64_t* __restrict__ lib_base, int Lp, int64_t r, int l) { while (l + 1 < Lp && r >= lib_base[l + 1]) ++l; return l; }
__device__ __forceinline__ unsigned long long shfl_up_u64(unsigned long long v, int d) {
    return ((unsigned long long)(uint32_t)__shfl_up((int)(uint32_t)(v >> 32), d, 64) << 32) | (uint32_t)__shfl_up((int)(uint32_t)v, d, 64);
}
// the thread's 4 {key, reach} pairs (+ the next piece's key), zero past the end
__device__ __forceinline__ void load_keyreach4(const int2* __restrict__ keyreach, int64_t b, int64_t n, int2 (&kr)[TR_ITEMS + 1]) {
    if (b + TR_ITEMS <= n) {
        const uint4 x = *reinterpret_cast<const uint4*>(keyreach + b), y = *reinterpret_cast<const uint4*>(keyreach + b + 2);
        kr[0] = make_int2((int)x.x, (int)x.y); kr[1] = make_int2((int)x.z, (int)x.w); kr[2] = make_int2((int)y.x, (int)y.y); kr[3] = make_int2((int)y.z, (int)y.w);
    } else {
#pragma unroll
        for (int j = 0; j < TR_ITEMS; ++j) kr[j] = b + j < n ? keyreach[b + j] : make_int2(0, 0);
    }
    kr[TR_ITEMS] = b + TR_ITEMS < n ? keyreach[b + TR_ITEMS] : make_int2(0, 0);
}

__global__ __launch_bounds__(TR_T) void k_reach_blockmax(const int2* __restrict__ keyreach, int64_t n, const int64_t* __restrict__ lib_base, int Lp,
                                                         unsigned long long* __restrict__ agg) {
    __shared__ unsigned long long sh[TR_T / 64];
    const int64_t b = (int64_t)blockIdx.x * TR_CHUNK + (int64_t)threadIdx.x * TR_ITEMS;
    unsigned long long v = 0ull;
    if (b < n) {
        int2 kr[TR_ITEMS + 1]; load_keyreach4(keyreach, b, n, kr);
        int l = lib_of_slot(lib_base, Lp, b, 0);
#pragma unroll
        for (int j = 0; j < TR_ITEMS; ++j) if (b + j < n) {
            l = lib_of_slot(lib_base, Lp, b + j, l);
            const unsigned long long x = ((unsigned long long)(uint32_t)l << 32) | (uint32_t)kr[j].y;                 // (reaches are positions: non-negative)
            v = x > v ? x : v;
        }
    }
    v = block_reduce<OpMaxU64>(v, sh);
    if (threadIdx.x == 0) agg[blockIdx.x] = v;
}

__global__ __launch_bounds__(TR_T) void k_tiles_all(DevCfg c, const int2* __restrict__ keyreach, int64_t n, const int64_t* __restrict__ lib_base, int Lp,
                                                    const unsigned long long* __restrict__ agg, int64_t ntiles, uint2* __restrict__ rng) {
    __shared__ unsigned long long sh[TR_T / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t b = (int64_t)blockIdx.x * TR_CHUNK + (int64_t)threadIdx.x * TR_ITEMS;
    int2 kr[TR_ITEMS + 1]; int lib[TR_ITEMS];
    load_keyreach4(keyreach, b, n, kr);
    unsigned long long tot = 0ull;
    {
        int l = b < n ? lib_of_slot(lib_base, Lp, b, 0) : 0;
#pragma unroll
        for (int j = 0; j < TR_ITEMS; ++j) {
            if (b + j < n) {
                l = lib_of_slot(lib_base, Lp, b + j, l);
                const unsigned long long x = ((unsigned long long)(uint32_t)l << 32) | (uint32_t)kr[j].y;
                tot = x > tot ? x : tot;
            }
            lib[j] = l;
        }
    }
    // exclusive running maximum in front of this thread's first piece: the blocks before (agg, already scanned), the waves
    // of this block before, the lanes of this wave before
    unsigned long long incl = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long u = shfl_up_u64(incl, d); if (lane >= d && u > incl) incl = u; }
    if (lane == 63) sh[wv] = incl;
    __syncthreads();
    unsigned long long run = agg[blockIdx.x];
    for (int w = 0; w < wv; ++w) { const unsigned long long u = sh[w]; run = u > run ? u : run; }
    { const unsigned long long u = shfl_up_u64(incl, 1); if (lane && u > run) run = u; }
    uint32_t* out = reinterpret_cast<uint32_t*>(rng);
#pragma unroll
    for (int j = 0; j < TR_ITEMS; ++j) {
        const int64_t r = b + j;
        if (r >= n) break;
        const int l = lib[j];
        const int64_t s0 = lib_base[l], s1 = lib_base[l + 1];
        uint32_t* row = out + 2 * (int64_t)l * ntiles;
        const unsigned long long x = ((unsigned long long)(uint32_t)l << 32) | (uint32_t)kr[j].y;
        // lo: tiles t with prefmax[r-1] <= p0(t) < prefmax[r]   (no earlier piece in this library: from tile 0)
        {
            const bool have_prev = (int)(run >> 32) == l && r > s0;
            const int64_t prev = have_prev ? (int64_t)(int32_t)(uint32_t)run : INT32_MIN;
            const unsigned long long cur64 = x > run ? x : run;
            const int64_t cur = (int64_t)(int32_t)(uint32_t)cur64;
            int64_t t0 = have_prev ? tiles_ceil_div64(prev - c.pos0) : 0;
            int64_t t1 = tiles_ceil_div64(cur - c.pos0);
            if (t1 > ntiles) t1 = ntiles;
            for (int64_t t = t0; t < t1; ++t) row[2 * t] = (uint32_t)r;
            if (r == s1 - 1) for (int64_t t = t1 > t0 ? t1 : t0; t < ntiles; ++t) row[2 * t] = (uint32_t)s1;
            run = cur64;
        }
        // hi: tiles t with key[r] <= p1(t) < key[r+1]      (p1(t) >= x  <=>  t >= ceil((x - pos0 - 63) / 64))
        {
            int64_t t0 = tiles_ceil_div64((int64_t)kr[j].x - c.pos0 - (TILE - 1));
            int64_t t1 = r + 1 < s1 ? tiles_ceil_div64((int64_t)kr[j + 1].x - c.pos0 - (TILE - 1)) : ntiles;
            if (t1 > ntiles) t1 = ntiles;
            if (r == s0) for (int64_t t = 0; t < t0 && t < ntiles; ++t) row[2 * t + 1] = (uint32_t)s0;
            for (int64_t t = t0; t < t1; ++t) row[2 * t + 1] = (uint32_t)(r + 1);
        }
    }
}

// brc_region_windows: only what an announced window asks for is piled up.  The host hands over the LIST of announced tiles
// (k_pileup2<true> is launched over that list: a site list touches a quarter of the tiles its reads span) and, per tile, the
// first and last lane a window asks for; everything else of the planes is emptied once, when the region is uploaded — no
// kernel of the pass writes there.  This kernel trims every announced tile's piece range to the pieces that can reach its
// wanted lanes: a line's window is one or two positions wide, the tile around it holds a third more pieces than cover those
// (keyreach = {start of the piece's read, end of its column}).
__global__ __launch_bounds__(256) void k_narrow_tiles(const uint16_t* __restrict__ wanted, const uint32_t* __restrict__ tile_list, int64_t n_listed, int64_t ntiles, int Lp, int32_t pos0,
                                                      uint2* __restrict__ rng, const int2* __restrict__ keyreach) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;           // (library, listed tile)
    if (i >= n_listed * Lp) return;
    const int64_t tile = tile_list[i % n_listed], lib = i / n_listed;
    const uint32_t w = wanted[tile];
    uint2 r = rng[lib * ntiles + tile];
    const int64_t p0w = (int64_t)pos0 + tile * TILE + (w & 0xffu), p1w = (int64_t)pos0 + tile * TILE + (w >> 8);
    while (r.x < r.y && (int64_t)keyreach[r.x].y <= p0w) ++r.x;
    while (r.y > r.x && (int64_t)keyreach[r.y - 1].x > p1w) --r.y;
    rng[lib * ntiles + tile] = r;
}

// Tile compaction for reads with MANY operators (round 5).  A tile's pieces are a contiguous range of the stream in read order, and ALL
// pieces of every read that overlaps the tile lie in it: a 10-kb read with an operator every 15 bases has 700 pieces, five of which
// touch a given tile — k_pileup2 would spend 98 % of its piece-steps (40 vector instructions each) on pieces that are not there.  For
// regions whose reads average more than a dozen pieces the range of every (tile, library) is first COMPACTED: one wave walks it 64
// pieces at a time (a 16-byte load and two compares per piece instead of a pipeline step), keeps those whose column extent
// [rs, rs + ext) meets the tile, in stream order (ballot + prefix count: the order the fp32 sums need), and copies their records —
// hot and rare — into a tile-major stream of their own.  k_pileup2 then runs unchanged over that stream and its ranges.
// COUNT: the live pieces per (tile, library) (+ the two totals of brc_region_piece_steps); else: the copy, to cmp_off[] (exclusive scan).
template <bool COUNT>
__global__ __launch_bounds__(256) void k_compact_tiles(DevCfg c, const uint4* __restrict__ pieces4, const PieceRare* __restrict__ rare, const uint2* __restrict__ rng,
                                                        int64_t ntiles, uint32_t* __restrict__ cnt, const uint32_t* __restrict__ cmp_off, uint4* __restrict__ out4,
                                                        PieceRare* __restrict__ out_rare, uint2* __restrict__ out_rng, unsigned long long* __restrict__ totals) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const int lib = blockIdx.y;
    const int64_t slot = (int64_t)lib * ntiles + tile;
    const uint2 r = rng[slot];
    const int64_t p0 = (int64_t)c.pos0 + tile * TILE, p1 = p0 + TILE;
    uint32_t run = COUNT ? 0u : cmp_off[slot];
    const uint32_t first = run;
    for (uint32_t base = r.x; base < r.y; base += 64u) {
        const uint32_t m = base + (uint32_t)lane;
        bool live = false; uint4 h0 = make_uint4(0u, 0u, 0u, 0u);
        if (m < r.y) { h0 = pieces4[(size_t)m * 3u]; live = (int64_t)(int32_t)h0.x < p1 && (int64_t)(int32_t)h0.x + (int64_t)(int32_t)h0.z > p0; }      // {rs, len, ext, tp|flags}
        const unsigned long long mask = __ballot(live);
        if (!COUNT && live) {
            const uint32_t at = run + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
            out4[(size_t)at * 3u] = h0; out4[(size_t)at * 3u + 1u] = pieces4[(size_t)m * 3u + 1u]; out4[(size_t)at * 3u + 2u] = pieces4[(size_t)m * 3u + 2u];
            if (piece_has_rare(h0.w >> 24)) out_rare[at] = rare[m];
        }
        run += (uint32_t)__builtin_popcountll(mask);
    }
    if (lane == 0) {
        if (COUNT) { cnt[slot] = run; atomicAdd(&totals[0], (unsigned long long)(r.y - r.x)); atomicAdd(&totals[1], (unsigned long long)run); }
        else out_rng[slot] = make_uint2(first, run);
    }
}

// The same compaction READ-WISE (one library: the slots of consecutive reads are consecutive, piece_off[] is their prefix sum): the
// pieces of one read are sorted by their reference start and their column extents follow each other without gaps or overlaps (piece q
// starts where piece q - 1 of its read ends, a read's first piece at the read's position), so the pieces of a read that touch a tile are
// ONE run of its slots — found by a binary search over keyreach[] = {read position, end of the piece's extent} (8 bytes per piece: the
// last levels of a search share a cache line; the 48-byte records are read only to be copied), a handful of steps forward to the run's
// end.  A lane takes one read of the tile's range: ~40 reads x ~14 probes per tile instead of the ~15 000 records a walk over the whole
// range reads when every read has 400 pieces.  The tiles' sizes come from a pass over the PIECES (k_count_piece_tiles: every piece adds
// one to each tile its extent meets — an upper bound where announced windows narrowed a tile's range), so the search runs once.
// (With several libraries the slots are library-major and piece_off[] is not monotone: k_compact_tiles serves those.)
__global__ __launch_bounds__(256) void k_count_piece_tiles(DevCfg c, const uint4* __restrict__ pieces4, int64_t n_pieces, uint32_t* __restrict__ cnt) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_pieces) return;
    const uint4 h = pieces4[(size_t)q * 3u];                                                 // {rs, len, ext, tp|flags}
    int64_t k0 = (int64_t)(int32_t)h.x - c.pos0, k1 = k0 + (int64_t)(int32_t)h.z - 1;
    if (k1 < 0 || k0 >= c.P || k1 < k0) return;
    if (k0 < 0) k0 = 0; if (k1 > c.P - 1) k1 = c.P - 1;
    for (int64_t t = k0 >> 6; t <= (k1 >> 6); ++t) atomicAdd(&cnt[t], 1u);
}
enum { CR_GROUP = 8 };       // consecutive tiles a wave of k_compact_reads takes: a read's binary search is paid once per group, a tile's run is a few steps behind the last one's
__global__ __launch_bounds__(256) void k_compact_reads(DevCfg c, const uint4* __restrict__ pieces4, const PieceRare* __restrict__ rare, const uint2* __restrict__ rng,
                                                        const uint32_t* __restrict__ piece_off, const int2* __restrict__ keyreach, int64_t ntiles, const uint32_t* __restrict__ cmp_off,
                                                        uint4* __restrict__ out4, PieceRare* __restrict__ out_rare, uint2* __restrict__ out_rng, unsigned long long* __restrict__ totals) {
    const int lane = threadIdx.x & 63;
    const int64_t tile0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * CR_GROUP;
    if (tile0 >= ntiles) return;
    const int ng = (int)(ntiles - tile0 < CR_GROUP ? ntiles - tile0 : CR_GROUP);
    __shared__ uint32_t src_all[4][64];
    __shared__ uint32_t run_all[4][CR_GROUP];
    uint32_t* const src = src_all[threadIdx.x >> 6];
    uint32_t* const runs = run_all[threadIdx.x >> 6];
    // the group's tiles: their ranges of the stream (k_tiles_all / k_narrow_tiles) and their blocks of the compacted stream
    uint2 r_me = make_uint2(0u, 0u);
    if (lane < ng) { r_me = rng[tile0 + lane]; runs[lane] = cmp_off[tile0 + lane]; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    const bool some = lane < ng && r_me.x < r_me.y;
    uint32_t gx = some ? r_me.x : 0xffffffffu, gy = some ? r_me.y : 0u;
#pragma unroll
    for (int d = 1; d < CR_GROUP; d <<= 1) { const uint32_t ox = (uint32_t)__shfl_xor((int)gx, d, 64), oy = (uint32_t)__shfl_xor((int)gy, d, 64); gx = ox < gx ? ox : gx; gy = oy > gy ? oy : gy; }
    gx = (uint32_t)__builtin_amdgcn_readfirstlane((int)gx); gy = (uint32_t)__builtin_amdgcn_readfirstlane((int)gy);
    bool over = false;       // over: more live pieces than a tile's block holds (never: the host fails the call if it happens)
    if (gx < gy) {
        const uint32_t n = (uint32_t)c.n_reads;
        // reads whose slots start at or before gx (64-ary search: piece_off[] is non-decreasing); the last of them holds slot gx
        uint32_t lo_i = 0u, len = n;
        while (len > 0u) {
            const uint32_t stride = (len + 63u) >> 6;
            uint32_t idx = lo_i + ((uint32_t)lane + 1u) * stride - 1u; const uint32_t last = lo_i + len - 1u;
            if (idx > last) idx = last;
            const uint32_t k = (uint32_t)__builtin_popcountll(__ballot(piece_off[idx] <= gx));
            const uint32_t nlo = lo_i + k * stride;
            if (k == 64u || nlo > last) { lo_i = last + 1u; break; }
            const uint32_t nlen = (stride < last + 1u - nlo ? stride : last + 1u - nlo);
            lo_i = nlo; len = nlen - 1u;                                  // (the last element of block k is above gx)
        }
        const uint32_t r_first = lo_i > 0u ? lo_i - 1u : 0u;
        for (uint32_t rb = r_first; rb < n; rb += 64u) {
            const uint32_t rd = rb + (uint32_t)lane;
            uint32_t a0 = 0u, b0 = 0u;
            if (rd < n) { a0 = piece_off[rd]; b0 = rd + 1u < n ? piece_off[rd + 1u] : (uint32_t)c.n_pieces; }
            if (!__ballot(rd < n && a0 < gy)) break;
            uint32_t cur = 0u; bool searched = false;          // cur: every slot of the read before it ends at or before the current tile's first position
            for (int g = 0; g < ng; ++g) {
                const int64_t tile = tile0 + g;
                const uint2 r = make_uint2((uint32_t)__builtin_amdgcn_readlane((int)r_me.x, g), (uint32_t)__builtin_amdgcn_readlane((int)r_me.y, g));
                if (r.x >= r.y) continue;
                // (the region's last tile ends with the region: k_count_piece_tiles does not count a piece that starts behind the last position)
                const int64_t p0 = (int64_t)c.pos0 + tile * TILE, p1 = p0 + TILE < (int64_t)c.pos0 + c.P ? p0 + TILE : (int64_t)c.pos0 + c.P;
                const bool inside = rd < n && a0 < r.y;
                if (!__ballot(inside)) continue;
                const uint32_t a = a0 < r.x ? r.x : a0, b = b0 > r.y ? r.y : b0;
                uint32_t m = a, nlive = 0u;
                if (inside && a < b) {
                    if (!searched) {
                        // first slot of [a, b) whose extent ends behind p0
                        uint32_t lo = a, hi = b;
                        while (lo < hi) {
                            const uint32_t mid = lo + ((hi - lo) >> 1);
                            if ((int64_t)keyreach[mid].y > p0) hi = mid; else lo = mid + 1u;
                        }
                        cur = lo; searched = true;
                    } else {
                        if (cur < a) cur = a;
                        while (cur < b && (int64_t)keyreach[cur].y <= p0) ++cur;
                    }
                    m = cur < b ? cur : b;
                    if (m < b) {
                        // its start: the read's position for the read's first slot, the end of the slot before it otherwise
                        int64_t rs = m == a0 ? (int64_t)keyreach[m].x : (int64_t)keyreach[m - 1u].y;
                        for (uint32_t q = m; q < b && rs < p1; ++q) { ++nlive; rs = (int64_t)keyreach[q].y; }
                    }
                }
                uint32_t incl = nlive;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
                // copy: consecutive lanes take consecutive OUTPUT slots (64 records = 3 KB of contiguous stores per round; the sources are
                // runs of a read's consecutive slots) — the owners publish the source slot of every output of the round through LDS
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63), excl = incl - nlive;
                if (!total) continue;
                const uint32_t limit = cmp_off[tile + 1];                 // (the tile's block, sized by k_count_piece_tiles from the same extents)
                uint32_t run = runs[g];
                for (uint32_t j0 = 0u; j0 < total; j0 += 64u) {
                    for (uint32_t i = 0u; i < nlive; ++i) { const uint32_t o = excl + i - j0; if (o < 64u) src[o] = m + i; }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    const uint32_t at = run + j0 + (uint32_t)lane;
                    if (j0 + (uint32_t)lane < total && at < limit) {
                        const uint32_t q = src[lane];
                        const uint4 h0 = pieces4[(size_t)q * 3u];
                        out4[(size_t)at * 3u] = h0; out4[(size_t)at * 3u + 1u] = pieces4[(size_t)q * 3u + 1u]; out4[(size_t)at * 3u + 2u] = pieces4[(size_t)q * 3u + 2u];
                        if (piece_has_rare(h0.w >> 24)) out_rare[at] = rare[q];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                }
                run += total;
                if (run > limit) { over = true; run = limit; }
                if (lane == 0) runs[g] = run;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (lane < ng) {
        const uint32_t first = cmp_off[tile0 + lane];
        out_rng[tile0 + lane] = make_uint2(first, runs[lane]);
        atomicAdd(&totals[0], (unsigned long long)(r_me.x < r_me.y ? r_me.y - r_me.x : 0u)); atomicAdd(&totals[1], (unsigned long long)(runs[lane] - first));
    }
    if (lane == 0 && over) atomicAdd(&totals[2], 1ull);
}

// ---------------------------------------------------------------- KB: pileup + BasicStat accumulation (the hot kernel)

// |a - b| of two unsigned values in one instruction (LLVM does not form v_sad_u32 from max - min)
__device__ __forceinline__ uint32_t sad_u32(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "s"(b));
    return d;
}
// x += 1 in the lanes of a wave mask held in scalar registers: one add-with-carry instead of select + add
__device__ __forceinline__ void count_if(uint32_t& x, uint64_t mask) {
    asm("v_addc_co_u32 %0, vcc, 0, %0, %1" : "+v"(x) : "s"(mask) : "vcc");
}

__device__ __forceinline__ void fadd_v(float& acc, float x) { acc += x; }
__device__ __forceinline__ void fadd_s(float& acc, float x) { acc += x; }
// acc = (float)((double)acc + x)   (sum_event_location, BasicStat.cpp:70), in place: left to the compiler the sum is computed
// into a temporary and copied back inside the exec region.  (The same treatment of the three plain float adds costs five
// registers — their operands must then sit in registers of their own — and with them a wave per SIMD.)
__device__ __forceinline__ void fadd_through_double(float& acc, double x) {
    double t;
    asm("v_cvt_f64_f32_e32 %1, %0\n\tv_add_f64 %1, %1, %2\n\tv_cvt_f32_f64_e32 %0, %1" : "+v"(acc), "=&v"(t) : "v"(x));
}

enum { PILEUP_WAVES = 4 };   // 256 threads: 4 consecutive tiles (256 positions) per workgroup
enum { WIN_U4 = 5 };         // event-byte window of a staged piece: 5 x 16 B = 80 elements >= 64 tile positions + 15 of alignment
enum { ROW_BYTES = WIN_U4 * 16 };
enum { QCAP = 3 * HALF + 2 };   // deferred entries of one half-batch: at most a third-allele, an unnameable-bucket and a huge-integer entry per piece
static_assert(HALF * WIN_U4 <= 64, "one direct-to-LDS instruction stages a half-batch");
static_assert(HALF % 3 == 0, "the piece-record registers rotate with period 3");

typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// dwords 0-9 of a Piece in scalar registers: f = {rs, len, ext, tp_flags, w1, w2, w3, snm}, g = {ww, a}
struct PRec { u32x8 f; u32x2 g; };

// One wave = one (64-position tile, library); lane == position.  The wave walks the tile's pieces [lo, hi) of its library
// in stream (= pileup column) order, in half-batches of HALF = 12 (brc_core.h: BRC_HALF):
//  * piece records: the 40 bytes every piece needs arrive by two scalar loads, two pieces ahead, rotating through three
//    scalar register sets — everything wave-uniform (positions, lengths, packed addends) lives in SGPRs.  The loads are
//    inline assembly so that they are issued exactly there (the scheduler sinks compiler-visible loads to their first use);
//    the matching s_waitcnt names the registers, which orders every use behind it;
//  * event bytes: the 80-byte window of each piece's row that this tile can touch is copied by ONE direct-to-LDS
//    instruction per half-batch of 12 pieces (global_load_lds_dwordx4: lane = row * 5 + chunk, no VGPR round trip) into a two-half
//    ring; the copy of half-batch h + 2 is issued when h is done, its addresses come from a 16-byte load of the pieces' {ww, a, bq_off} words
//    issued one half-batch earlier;
//  * one pipeline step = probe of piece j + 1 (coverage ballots, event byte and table look-ups: LDS reads only) and
//    accumulate of piece j (quality / bucket ballots, one exec region with the 10 adds of the dominant bucket, a usually
//    skipped one for everything else); lane conditions are 64-bit masks in scalar registers;
//  * per bucket a lane holds three PACKED integer registers (three 10-bit counters; mapq | sse; zm | clipped), the sum of
//    its event bytes (= 4 x base-quality sum + count x base index) and the four fp32 sums; the integers are flushed to the
//    planes every K pieces (K = 127 for short reads) and at the end of the tile;
//  * third alleles (a lane keeps its reference base and the first other base in registers) and PF_HUGE integers are
//    queued and drained into the planes between half-batches, in piece order.
#ifndef BRC_EXP
#define BRC_EXP 0               // compile-time experiments of tools/experiments/README.md (0 = the product)
#endif

#ifndef BRC_WAVES_PER_EU
#define BRC_WAVES_PER_EU 7      // 72 VGPRs: 16 values spill into the rare paths (measured: 6 waves 3.92 ms, 7 waves 3.78 ms, 8 waves 5.6 ms — spills reach the loop)
#endif
template <bool WINDOWS, int SH>      // SH: DevCfg.pack_shift as a compile-time constant (the layout of the packed sums, choose_pack: the instantiation of
                             // short-read regions keeps its shifts as immediates — the code that was measured); WINDOWS: brc_region_windows is in force: tiles whose range k_mask_tiles marked lo > hi are skipped (an instantiation of its own: the
                             // common one stays the code that was measured — one more branch at its head moved its register allocation and cost 2.4 %)
__global__ __launch_bounds__(PILEUP_WAVES * 64) __attribute__((amdgpu_waves_per_eu(BRC_WAVES_PER_EU, BRC_WAVES_PER_EU))) void k_pileup2(DevCfg c, DevIn in, const uint4* __restrict__ pieces4, const PieceRare* __restrict__ rare,
                                                               const uint2* __restrict__ rng, int64_t ntiles, Planes pl, uint4* __restrict__ tile_ctr,
                                                               const uint8_t* __restrict__ eb_ro, const uint16_t* __restrict__ bqw_ro, const uint32_t* __restrict__ unavail_ro,
                                                               const uint8_t* __restrict__ refcode, const uint16_t* __restrict__ wanted_ro,
                                                               const uint32_t* __restrict__ tile_list, int64_t n_listed) {
    // (the host launches the instantiation of DevCfg.pack_shift; the structure itself is left alone — assigning the constant to its field
    // moved the kernel's argument loads and, with them, its register allocation: the SH = 16 instantiation must stay, instruction for
    // instruction, the kernel that was measured — the few helpers that need the shift get it as an argument)
    // XCD-aware mapping: workgroup b runs on XCD b % 8 (observed dispatch order); give every XCD one contiguous
    // run of tiles so neighbouring tiles, which share most of their pieces, hit the same 4-MiB L2.
    const uint32_t nbk = gridDim.x;           // multiple of 8
    const uint32_t per = nbk >> 3;
    const uint32_t wg = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    //   q[n] = (float)n / (float)L0 for 0 <= n <= L0 (see piece_terms_tab);   e[n] = 1.0 - (double)q[n]
    //   The q entries stand 16 bytes apart and the e entries 8: both look-ups of a probe, q[|qpos - tp|] and e[|2 qpos - L0|],
    //   then have byte offsets |16 lane + const| — ONE lane register (16 lane + bias) and one v_sad_u32 each.
    struct QEnt { uint32_t piece, kind, mlo, mhi; };
    struct Lds {
        alignas(16) float q16[(TABLE_MAX + 2) * 4];
        double e[TABLE_MAX + 2];
        alignas(16) uint4 rows[PILEUP_WAVES][2 * HALF][WIN_U4];
        QEnt queue[PILEUP_WAVES][QCAP];
    };
    __shared__ Lds lds;
    {
        const float l0 = (float)c.table_len;
        for (int n = threadIdx.x; n <= c.table_len; n += PILEUP_WAVES * 64) {
            const float qv = (float)n / l0;
            lds.q16[4 * n] = qv; lds.e[n] = 1.0 - (double)qv;
        }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const uint32_t wv = __builtin_amdgcn_readfirstlane((uint32_t)threadIdx.x >> 6);   // wave of the workgroup (scalar)
    // (brc_region_windows: the launch covers the list of announced tiles, not the region)
    const int64_t slot = (int64_t)wg * PILEUP_WAVES + wv;
    if (slot >= (WINDOWS ? n_listed : ntiles)) return;
    const int64_t tile = WINDOWS ? (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)*BRC_CK(c, CK_PILEUP, 1, CB_TILELIST, tile_list + slot, 4, slot, -1)) : slot;
    const int lib = blockIdx.y;
    const uint2 r2 = *BRC_CK(c, CK_PILEUP, 2, CB_KP_RNG, rng + ((int64_t)lib * ntiles + tile), sizeof(uint2), tile, -1);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(r2.x), hi = __builtin_amdgcn_readfirstlane(r2.y);
    const int64_t k = tile * TILE + lane;
    const bool inreg = k < c.P;
    const int64_t kk = inreg ? k : 0;
    // a position abandoned for a library-less read (:281-284) accumulates nothing: it behaves like a lane outside the region
    const bool dead = c.per_lib && inreg && *BRC_CK(c, CK_PILEUP, 3, CB_UNAVAIL, unavail_ro + kk, 4, tile, -1) != NONE32;
    // (brc_region_windows: the lanes no window asks for behave like lanes outside the region — k_narrow_tiles has trimmed the tile's
    // piece range to what reaches the others)
    const bool valid = inreg && !dead && (!WINDOWS || tile_wants(*BRC_CK(c, CK_PILEUP, 4, CB_WANTED, wanted_ro + tile, 2, tile, -1), (uint32_t)lane));
    const int32_t p = (int32_t)(c.pos0 + k);
    const int32_t p0 = (int32_t)(c.pos0 + tile * TILE);                     // first position of the tile (scalar)

    LaneAcc2 a;
    {
        // dominant bucket = the bucket of the position's reference base (dominant_bucket, brc_core.h), from the 4-bit codes
        // k_refcode wrote for K1 (padded with code 15 on both sides; a NUL character has code 15 in its low nibble too)
        uint32_t dom = 1u;
        if (c.has_ref && valid) {
            const int64_t ri = (int64_t)p - c.ref_lo;
            dom = (ri >= -(int64_t)REFCODE_PAD && ri < c.ref_hi - c.ref_lo + (int64_t)REFCODE_PAD) ? canon_bucket(*BRC_CK(c, CK_PILEUP, 5, CB_REFCODE, refcode + ri, 1, tile, -1) & 15u) : 5u;
        }
        lane2_init(a, c.force_dom >= 0 ? (uint32_t)c.force_dom : dom);
    }
    // what `event byte & 3` is compared with (4, 5: no byte matches).  The lane keeps THIS across the read loop; the bucket
    // itself is recomputed from it where a rare path or the tile's end needs it (one register, not two, live in the loop)
    const uint32_t dom_i = dom_index(a.dom_b);
#define BRC_DOM_B() dom_bucket_of_index(dom_i)
    bool flushed = false;                                                   // (scalar) the slot planes of this tile hold partial integer sums
    unsigned long long wsm_tot = 0, wnm_tot = 0;                            // (scalar) warnings moved out of the lanes at flushes

    if (lo < hi && !BRC_PVAR(4)) {            // (variant: profiling ablations, BRC_PILEUP_VARIANT — 4: no piece loop, 1: no plane stores, 5: one half-batch only)
        // lanes past the region's last position stand far left of every piece: no coverage test is ever true for them
        // (d = 2^31 + lane + p0 - rs >= 2^31 - (rs - p0) >= ext for every piece, because rs + ext <= 2^31 - 1 and p0 >= 0)
        const uint32_t lanev = valid ? (uint32_t)lane : (0x80000000u | (uint32_t)lane);
        // 16 lane + bias: the lane side of both table addresses (the bias keeps the scalar side, bias - 16 (s_c - tp) and
        // bias - (16 s_c - 8 L0), positive: |s_c| < 2^22 + 64)
        enum : uint32_t { LBIAS = 1u << 28 };
        const uint32_t lane16b = ((uint32_t)lane << 4) + (uint32_t)LBIAS;
        // (the rare paths below recompute the lane index and the plane index instead of keeping them alive across the read loop:
        // every register that stays live there pushes a value of the alternate-bucket path into scratch)
#define BRC_LANE() ((int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)))
#define BRC_KK() (valid ? tile * TILE + BRC_LANE() : (int64_t)0)
        const uint32_t L0 = (uint32_t)c.table_len;
        const uint32_t cb = (uint32_t)LBIAS + (L0 << 3);
        const uint32_t thr0 = piece_thr(c);
        constexpr uint32_t pack_sh = (uint32_t)SH;                                // (the clipped length of a piece: its w3 above the narrow field)
        char* const rows_base = reinterpret_cast<char*>(&lds.rows[wv][0][0]);
        QEnt* const queue = lds.queue[wv];
        // sub-list of this workgroup's third-allele events: workgroups in flight on an XCD are consecutive, so their
        // cursors are different words in different lines
        const uint32_t xshard = (wg + (uint32_t)lib * 257u) & (pl.xev_shards - 1u);
        uint32_t qn = 0;                                                       // queued entries (scalar)
        int32_t since_flush = 0;
        // ---- staging: lane -> (row, chunk) of a half-batch
        const uint32_t srow = (uint32_t)lane / (uint32_t)WIN_U4, schunk = (uint32_t)lane % (uint32_t)WIN_U4;
        const bool slane = lane < HALF * WIN_U4;
        const uint32_t win_span = (uint32_t)c.max_lqseq + 96u + 16u;          // legitimate window starts: -95 .. max_lqseq + 15 (rows are padded to 16)
        // {bq_off, a} of piece b0 + srow (clamped): what the window copy of that row needs
        // ... and, by the first lane of every row, one dword of its hot record: nothing uses the value — the load pulls the
        // record's cache line into L2 a half-batch before the scalar loads of the read loop ask for it (their own look-ahead
        // of two pieces covers an L2 hit, not an HBM miss)
#define BRC_LD_TAB(T, b0) { const uint32_t mi = (b0) + srow < hi ? (b0) + srow : hi - 1u; T = *BRC_CK(c, CK_PILEUP, 6, CB_KP_PIECES, pieces4 + ((size_t)mi * 3u + 2u), 16, tile, mi);   /* {ww, a, bq_off} */ \
                            asm volatile("" :: "v"(pf)); if (schunk == 0u) pf = BRC_CK(c, CK_PILEUP, 7, CB_KP_PIECES, pieces4 + (size_t)mi * 3u, 16, tile, mi)->x; }
        // window copy of the half-batch starting at piece b0 into the ring half at byte offset hoff: element window
        // [ws, ws + 80) of the row, ws = floor16(p0 - a) (may start before the row: the event-byte stream is padded)
#define BRC_STAGE(T, b0, hoff)                                                                                           \
        {                                                                                                                 \
            if (slane && (b0) + srow < hi) {                                                                              \
                const int64_t boff = (int64_t)(((uint64_t)T.w << 32) | T.z);                                              \
                int32_t ws = (p0 - (int32_t)T.y) & ~15;                                                                   \
                /* a piece that only SPANS the tile (an intron, a long deletion: ext >> len) or does not touch it at all has its  \
                   window anywhere — 100 kb from its row, before the stream or past its end; no lane will read the copy: take    \
                   the row's first bytes instead (round 4: a memory fault on spliced alignments, tools/fuzz/extreme.py) */      \
                if ((uint32_t)(ws + 96) > win_span) ws = 0;                                                               \
                const uint8_t* src = BRC_CK(c, CK_PILEUP, 8, CB_EB, eb_ro + (boff + ws) + 16u * schunk, 16, tile, (b0) + srow);   \
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,                      \
                    (void __attribute__((address_space(3)))*)(rows_base + (hoff)), 16, 0, 0);                             \
            }                                                                                                             \
        }
        // Scalar loads of the record at rp (issued HERE, two pieces ahead), and the wait that makes them usable.  The three
        // record sets live in FIXED scalar registers, named in the constraints of both statements: the load writes exactly the
        // registers the wait hands over, nothing is copied in between.  What the compiler does with them between the two
        // statements is not left to belief: tools/check_isa.py walks the control-flow graph of the built kernel from every
        // such load to the first s_waitcnt lgkmcnt(0) on every path and fails the build if any instruction on the way reads
        // or writes one of the registers in flight (bam_readcount_amd/csrc/Makefile runs it; tests/test_abi.py repeats it
        // for builds at 6, 7 and 8 waves per SIMD).
#define BRC_F_R0 "{s[56:63]}"
#define BRC_G_R0 "{s[64:65]}"
#define BRC_F_R1 "{s[68:75]}"
#define BRC_G_R1 "{s[66:67]}"
#define BRC_F_R2 "{s[76:83]}"
#define BRC_G_R2 "{s[84:85]}"
#ifdef BRC_CHECKED
        // (checked build: the address is compared first, and the load waits for itself — the registers are valid when the statement
        // ends, so this build does not depend on what the compiler does between a load and its wait and needs no ISA check)
#define BRC_LD_REC(R, rp) { const char* rpc = BRC_CKS(c, CK_PILEUP, 9, CB_KP_PIECES, (rp), 40, tile, ((rp) - reinterpret_cast<const char*>(pieces4)) / 48); \
                            asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx2 %1, %2, 0x20\n\ts_waitcnt lgkmcnt(0)" : "=" BRC_F_##R (R.f), "=" BRC_G_##R (R.g) : "s"(rpc)); }
#else
#define BRC_LD_REC(R, rp) asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx2 %1, %2, 0x20" : "=" BRC_F_##R (R.f), "=" BRC_G_##R (R.g) : "s"(rp));
#endif
#define BRC_WAIT_REC(R) asm volatile("s_waitcnt lgkmcnt(0)" : "+" BRC_F_##R (R.f), "+" BRC_G_##R (R.g));
        // the division constants of piece m (its rare record) by scalar loads, on demand: only the few pieces that neither look
        // their terms up nor divide them out from their own record (PF_DIV) — reads longer than 255 bases, a Q2 position of its own.
        // (The wait below also waits for the record loads just issued: a synchronous round trip.  It used to be the price of EVERY
        // read of another length than the region's modal one.)
#define BRC_LD_DIV(H, R, m)                                                                                             \
        {                                                                                                                 \
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));                                                   \
            u32x4 dA; u32x2 dB; const char* dp = BRC_CKS(c, CK_PILEUP, 10, CB_KP_RARE, reinterpret_cast<const char*>(rare) + (size_t)(m) * 32u, 24, tile, (m)); \
            asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dwordx2 %1, %2, 0x10\n\ts_waitcnt lgkmcnt(0)" : "=&s"(dA), "=&s"(dB) : "s"(dp)); \
            H.rcpL = __uint_as_float(dA[0]); H.Lf = __uint_as_float(dA[1]); H.rcpC = __uint_as_float(dA[2]);              \
            H.center = __uint_as_float(dA[3]); H.left = (int32_t)dB[0]; H.q2 = (int32_t)dB[1]; H.zm_raw = 0u; H.sse_raw = 0u; \
        }
        struct Stage { uint32_t w; float t; double sev; uint64_t m_in, m_cov; int32_t s_c; };
        // PROBE of the piece in R, staged in the ring row at byte offset roff: coverage ballots and three LDS reads whose
        // results nothing in this stage touches (a copy or a select here would stall the wave on its own reads).  The table
        // look-ups are issued for every piece; a piece without PF_TABLE ignores them and divides in its accumulate stage.
#define BRC_PROBE(R, roff, S)                                                                                           \
        {                                                                                                                 \
            const int32_t s_d = p0 - (int32_t)R.f[0]; S.s_c = p0 - (int32_t)R.g[1];                                       \
            const uint32_t d = lanev + (uint32_t)s_d;                                                                     \
            S.m_cov = __builtin_amdgcn_ballot_w64(d < R.f[2]);   /* counted in the accumulate stage: a probe may run past the tile's last piece */ \
            S.m_in = __builtin_amdgcn_ballot_w64(d < R.f[1]);                                                             \
            const uint32_t off = (uint32_t)(roff) + ((uint32_t)S.s_c & 15u);                                              \
            S.w = (uint32_t)*reinterpret_cast<const uint8_t*>(rows_base + BRC_CKI(c, CK_PILEUP, 11, CB_LDS_ROWS, (uint32_t)lane + off, 2u * HALF * ROW_BYTES, tile, -1)); \
            /* (scalar) the lane-independent side of both table addresses: the second is the record's signed distance away from the first */ \
            const uint32_t eoff = cb - ((uint32_t)S.s_c << 4), qoff16 = eoff + (uint32_t)piece_tp_field(R.f[3]);          \
            if (BRC_EXP == 2) { S.t = 0.5f; S.sev = 0.25; } else {     /* (2: timing only, no table look-ups) */                 \
            S.t = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(lds.q16) + sad_u32(lane16b, qoff16));       \
            S.sev = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(lds.e) + sad_u32(lane16b, eoff)); }      \
        }
        // the 10 adds of the dominant bucket, in the lanes of m_dom
#if BRC_EXP == 5
        // (experiment: ONE assembly statement, accumulators updated in place under an exec mask set and restored inside it)
#define BRC_DOM_REGION(R, S, m_dom, tq2, ts3p, tsev)                                                                    \
                {                                                                                                         \
                    double dtmp; uint64_t sv;                                                                             \
                    asm("s_and_saveexec_b64 %[sv], %[m]\n\t"                                                              \
                        "v_add_u32_e32 %[w1], %[a1], %[w1]\n\tv_add_u32_e32 %[w2], %[a2], %[w2]\n\tv_add_u32_e32 %[w3], %[a3], %[w3]\n\t" \
                        "v_add_u32_e32 %[sw], %[w], %[sw]\n\t"                                                            \
                        "v_add_f32_e32 %[q2], %[q2], %[tq]\n\tv_add_f32_e32 %[s3], %[s3], %[t3]\n\t"                       \
                        "v_cvt_f64_f32_e32 %[tmp], %[sev]\n\tv_add_f64 %[tmp], %[tmp], %[ts]\n\tv_cvt_f32_f64_e32 %[sev], %[tmp]\n\t" \
                        "v_add_f32_e32 %[snm], %[an], %[snm]\n\tv_add_u32_e32 %[ww], %[aw], %[ww]\n\t"                      \
                        "s_mov_b64 exec, %[sv]"                                                                           \
                        : [w1] "+v"(a.dom.w1), [w2] "+v"(a.dom.w2), [w3] "+v"(a.dom.w3), [sw] "+v"(a.dom.sw),           \
                          [q2] "+v"(a.dom.f[F_SQ2]), [s3] "+v"(a.dom.f[F_S3P]), [sev] "+v"(a.dom.f[F_SEV]), [snm] "+v"(a.dom.f[F_SNM]), \
                          [ww] "+v"(a.ww), [tmp] "=&v"(dtmp), [sv] "=&s"(sv)                                             \
                        : [m] "s"(m_dom), [a1] "s"(R.f[4]), [a2] "s"(R.f[5]), [a3] "s"(R.f[6]), [an] "s"(R.f[7]), [aw] "s"(R.g[0]), \
                          [w] "v"(S.w), [tq] "v"(tq2), [t3] "v"(ts3p), [ts] "v"(tsev) : "scc");                          \
                }
#else
#define BRC_DOM_REGION(R, S, m_dom, tq2, ts3p, tsev)                                                                    \
                if (__builtin_expect(__builtin_amdgcn_inverse_ballot_w64(m_dom), 1)) {                                    \
                    a.dom.w1 += R.f[4]; a.dom.w2 += R.f[5]; a.dom.w3 += R.f[6]; a.dom.sw += S.w;                          \
                    fadd_v(a.dom.f[F_SQ2], tq2); fadd_v(a.dom.f[F_S3P], ts3p);                                            \
                    BRC_SEV_ADD(a.dom.f[F_SEV], tsev);                                                                    \
                    fadd_s(a.dom.f[F_SNM], __uint_as_float(R.f[7])); a.ww += R.g[0];                                      \
                }                                                                                                         \
                /* (no instruction: naming the four sums here keeps each in ONE register across the join — left alone, the allocator   \
                   computes them into temporaries inside the region and copies them back, or packs two of the adds into a v_pk_add_f32   \
                   with a move in front and a 64-bit move behind: the region is 11 vector instructions with the four named, 12-13 without) */ \
                asm volatile("" : "+v"(a.dom.f[F_SEV]), "+v"(a.dom.f[F_SQ2]), "+v"(a.dom.f[F_S3P]), "+v"(a.dom.f[F_SNM]));
#endif
#if BRC_EXP == 1      // (timing only, wrong sums: the event-location sum without its three double-precision instructions)
#define BRC_SEV_ADD(acc, x) acc += (float)__double2hiint(x)
#else
#define BRC_SEV_ADD(acc, x) fadd_through_double(acc, x)
#endif
        // ACC of the piece in R (S = its probe results), piece index m
        // (the entry's kind is materialised by an instruction of its own: left to the allocator, the two constants live in a
        // register pair across the whole read loop and are spilled to scratch around every push)
#define BRC_QPUSH(mm, kd, mask) { if (BRC_LANE() == 0) { uint32_t kq; asm volatile("v_mov_b32 %0, %1" : "=v"(kq) : "n"(kd)); QEnt e; e.piece = (mm); e.kind = kq; e.mlo = (uint32_t)(mask); e.mhi = (uint32_t)((mask) >> 32); queue[BRC_CKI(c, CK_PILEUP, 12, CB_LDS_QUEUE, qn, (uint32_t)QCAP, tile, (mm))] = e; } ++qn; }
#define BRC_ACC(R, S, m)                                                                                                \
        {                                                                                                                 \
            const uint32_t fl = BRC_EXP == 6 ? (uint32_t)(PF_TABLE | PF_Q2OK) : BRC_EXP == 7 ? ((R.f[3] >> 24) | (uint32_t)PF_TABLE) :   \
                                BRC_EXP == 8 ? ((R.f[3] >> 24) & ~(uint32_t)(PF_NB | PF_HUGE)) : R.f[3] >> 24;   /* (6, 7, 8: timing only, flag tests folded away) */ \
            count_if(a.ncol, S.m_cov);                                                         /* lib_counts[library] (:286) */ \
            const uint64_t m_p = S.m_in & __builtin_amdgcn_ballot_w64(S.w >= thr0);           /* :288 (escape bytes answer it too) */ \
            count_if(a.depth, m_p);                                                            /* mapq_n (:312) */         \
            if (BRC_EXP != 3) {                                                                /* (3: timing only, probe and counters alone) */ \
                uint64_t m_b = m_p;                            /* lanes whose event goes to a base bucket */               \
                /* the terms live in the stage's own registers: a piece without PF_TABLE overwrites them (no copies on the   \
                   common path); q2 == tp, or no Q2 position (every reverse read without a Q2 run): then +0.0f, the identity  \
                   on these sums — the flag bit spread over a scalar register masks the look-up */                        \
                float tq2 = __uint_as_float(__float_as_uint(S.t) & (uint32_t)((int32_t)(R.f[3] << 6) >> 31));           \
                if (__builtin_expect((fl & PF_TABLE) == 0u, 0) && m_p != 0ull) {   /* every unusual piece (make_piece) with an event in this tile */ \
                    if (fl & PF_NB) m_b = 0ull;            /* :343 with -i: counted in the depth, in no bucket */            \
                    else {                                                                                                \
                        if (fl & PF_WIDE) {                                                                               \
                            /* a wide read: lanes that meet an escape byte take quality and bucket from the wide stream; an N / '='   \
                               base goes to the third-allele list whatever the lane's slots hold */                       \
                            const uint64_t m_esc = m_b & __builtin_amdgcn_ballot_w64(eb_is_escape(S.w));                  \
                            if (m_esc) {                                                                                  \
                                u32x2 bo; const char* bp = BRC_CKS(c, CK_PILEUP, 13, CB_KP_PIECES, reinterpret_cast<const char*>(pieces4) + (size_t)(m) * 48u, 48, tile, (m)); \
                                asm volatile("s_load_dwordx2 %0, %1, 0x28\n\ts_waitcnt lgkmcnt(0)" : "=&s"(bo) : "s"(bp)); \
                                /* (the read's wide row: where it starts, in units of 16 elements, is in the table at the head of the stream — \
                                   wide_base; its entry is 4 x (bq_off / 16) bytes in.  Both loads are the lanes' own: the row's address in \
                                   vector registers costs this rare path nothing, in scalar ones it cost the whole kernel 40 spills) */ \
                                const uint32_t toff = (bo[0] >> 2) | (bo[1] << 30);                                       \
                                bool exo = false;                                                                         \
                                if (__builtin_amdgcn_inverse_ballot_w64(m_esc)) {                                         \
                                    const uint32_t w16v = *BRC_CK(c, CK_PILEUP, 42, CB_BQW, reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(bqw_ro) + toff), 4, tile, (m)); \
                                    const uint32_t w16 = *BRC_CK(c, CK_PILEUP, 14, CB_BQW, bqw_ro + (((uint64_t)w16v << 4) + ((uint32_t)BRC_LANE() + (uint32_t)S.s_c)), 2, tile, (m)); \
                                    exo = !bucket_acgt(w16 & 0xffu);                                                      \
                                    if (exo) a.ww += R.g[0]; else S.w = ((w16 >> 8) << 2) | ((w16 & 0xffu) - 1u);         \
                                }                                                                                         \
                                const uint64_t m_exo = __builtin_amdgcn_ballot_w64(exo);                                  \
                                if (m_exo) { BRC_QPUSH((m), 0u, m_exo) m_b &= ~m_exo; }                                   \
                            }                                                                                             \
                        }                                                                                                 \
                        if ((fl & (PF_TABQ | PF_HUGE)) == PF_TABQ) {   /* soft-clipped: only the event location differs, and it needs no rare record */ \
                            S.sev = tabq_sev((int)((uint32_t)BRC_LANE() + (uint32_t)S.s_c), piece_left_field(R.f[3]), R.f[6] >> pack_sh); \
                        } else if (fl & PF_DIV) {      /* another read length: the terms divided out in the lane, from the record itself */ \
                            const EvTerms t = piece_terms_inlane(fl, R.f[3], R.f[6] >> pack_sh, (uint32_t)BRC_LANE() + (uint32_t)S.s_c); \
                            S.t = t.s3p; tq2 = t.q2; S.sev = t.sev;                                                       \
                        } else if (!(fl & PF_TABQ)) {                                                                     \
                            PieceRare H; BRC_LD_DIV(H, R, m)                                                              \
                            const EvTerms t = piece_terms_div(fl, (int)(R.f[3] & 0xffffffu), H, (int)((uint32_t)BRC_LANE() + (uint32_t)S.s_c));  \
                            S.t = t.s3p; tq2 = t.q2; S.sev = t.sev;                                                       \
                        }                                                                                                 \
                        if (fl & PF_HUGE) {                                                                               \
                            /* the integers the packed addends left out are added at the next boundary, to the lanes whose event goes   \
                               to one of the two slots: every lane of m_b but those that will overflow to the third-allele list —   \
                               known before the adds, a lane's alternate bucket changes only by its own event */              \
                            const uint32_t bh = (S.w & 3u) + 1u;                                                          \
                            const bool ovf_l = bh != BRC_DOM_B() && a.alt_b != NB_NONE && a.alt_b != bh;                   \
                            const uint64_t m_int = m_b & ~__builtin_amdgcn_ballot_w64(ovf_l);                             \
                            if (m_int) BRC_QPUSH((m), 1u, m_int)                                                          \
                        }                                                                                                 \
                    }                                                                                                     \
                }                                                                                                         \
                const float ts3p = S.t; const double tsev = S.sev;                                                        \
                const uint64_t m_dom = m_b & __builtin_amdgcn_ballot_w64((S.w & 3u) == dom_i);  /* the base of the reference */ \
                BRC_DOM_REGION(R, S, m_dom, tq2, ts3p, tsev)                                                              \
                const uint64_t m_rest = m_b & ~m_dom;                                                                     \
                if (__builtin_expect(m_rest != 0ull, 0)) {                                                                \
                    bool ovf = false;                                                                                     \
                    if (__builtin_amdgcn_inverse_ballot_w64(m_rest)) {                                                    \
                        const uint32_t b = (S.w & 3u) + 1u;                                                               \
                        const bool take_alt = a.alt_b == NB_NONE || a.alt_b == b;                                         \
                        if (take_alt) {                                                                                   \
                            a.alt_b = b;                                                                                  \
                            a.alt.w1 += R.f[4]; a.alt.w2 += R.f[5]; a.alt.w3 += R.f[6]; a.alt.sw += S.w;                  \
                            fadd_v(a.alt.f[F_SQ2], tq2); fadd_v(a.alt.f[F_S3P], ts3p);                                    \
                            fadd_through_double(a.alt.f[F_SEV], tsev);                                                    \
                            fadd_s(a.alt.f[F_SNM], __uint_as_float(R.f[7]));                                              \
                        }                                                                                                 \
                        a.ww += R.g[0];                        /* alternate and third alleles alike */                    \
                        ovf = !take_alt;                                                                                  \
                    }                                                                                                     \
                    const uint64_t m_ovf = __builtin_amdgcn_ballot_w64(ovf);                                              \
                    if (m_ovf) BRC_QPUSH((m), 0u, m_ovf)                                                                  \
                }                                                                                                         \
            }                                                                                                             \
        }
        // one pipeline step inside a half-batch: J = step (compile time); piece base + J is accumulated, base + J + 1 probed,
        // the record of base + J + 2 requested.  RC / RN / RL = the three record sets in their roles of this step.  A probe
        // past the tile's last piece reads a stale row and a record of the slack behind the stream; its result is never
        // accumulated.
#define BRC_STEP(J, RC, RN, RL, SC, SN, G)                                                                              \
        if ((J) == 0 || (J) < nb) {                                                                                       \
            /* the last step probes the first row of the OTHER ring half: its copy (and the record load behind it)       \
               was issued at the previous half-batch boundary */                                                         \
            if ((J) == HALF - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                          \
            /* record RN was requested a whole step ago, the LDS results of the previous probe likewise */               \
            BRC_WAIT_REC(RN)                                                                                              \
            BRC_LD_REC(RL, recp) recp += recstep;                                                                         \
            {                                                                                                             \
                BRC_PROBE(RN, ((J) + 1 < HALF ? hoff + ((J) + 1) * ROW_BYTES : (hoff ^ HOFF_X)), SN)                      \
            }                                                                                                             \
            BRC_ACC(RC, SC, base + (J))                                                                                   \
        }
        // packed integers -> slot planes; the 16-bit warning counters of the lanes move to the wave's totals
#define BRC_FLUSH()                                                                                                     \
        {                                                                                                                 \
            a.dom_b = BRC_DOM_B();                                                                                        \
            if (valid) lane2_flush(c, pl, lib, BRC_KK(), a, flushed, SH);                                                     \
            wsm_tot += wave_sum_u32(valid ? (a.ww & 0xffffu) : 0u); wnm_tot += wave_sum_u32(valid ? (a.ww >> 16) : 0u); a.ww = 0u; \
            flushed = true; since_flush = 0;                                                                              \
        }
        // between half-batches: drain the queue (the event bytes of this half-batch are still staged in ring half hoff),
        // flush when the packed fields could overflow during the next half-batch, then reuse the ring half just processed
        // (every LDS read of it has returned after the lgkmcnt wait): copy half-batch base + 2 HALF into it and request
        // the addresses of the one after
#define BRC_BOUNDARY(nbv)                                                                                               \
        {                                                                                                                 \
            since_flush += (int32_t)(nbv);                                                                                \
            if (__builtin_expect(qn != 0u, 0)) {                                                                          \
                for (uint32_t e = 0; e < qn; ++e) {                                                                       \
                    (void)BRC_CKI(c, CK_PILEUP, 21, CB_LDS_QUEUE, e, (uint32_t)QCAP, tile, -1);                           \
                    const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane(queue[e].piece), kind = (uint32_t)__builtin_amdgcn_readfirstlane(queue[e].kind); \
                    /* (readfirstlane returns int: without the casts a set bit 31 of the low half would sign-extend) */  \
                    const uint64_t mask = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(queue[e].mhi) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(queue[e].mlo); \
                    Piece H; PieceRare RR;                                                                                \
                    {   /* the piece into scalar registers; its rare record likewise when K1 stored one, derived otherwise */ \
                        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));                                       \
                        u32x8 hf; u32x4 hg;                                                                               \
                        const char* hp = BRC_CKS(c, CK_PILEUP, 15, CB_KP_PIECES, reinterpret_cast<const char*>(pieces4) + (size_t)m * 48u, 48, tile, m); \
                        asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x20\n\ts_waitcnt lgkmcnt(0)" : "=&s"(hf), "=&s"(hg) : "s"(hp)); \
                        H.rs = (int32_t)hf[0]; H.len = (int32_t)hf[1]; H.ext = (int32_t)hf[2]; H.tp_flags = hf[3];       \
                        H.w1 = hf[4]; H.w2 = hf[5]; H.w3 = hf[6]; H.snm = __uint_as_float(hf[7]); H.ww = hg[0]; H.a = (int32_t)hg[1]; \
                        H.bq_off = ((uint64_t)hg[3] << 32) | hg[2];                                                       \
                        if (piece_has_rare(H.tp_flags >> 24)) {                                                           \
                            u32x8 rf; const char* rp = BRC_CKS(c, CK_PILEUP, 16, CB_KP_RARE, reinterpret_cast<const char*>(rare) + (size_t)m * 32u, 32, tile, m); \
                            asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(rf) : "s"(rp));    \
                            RR.rcpL = __uint_as_float(rf[0]); RR.Lf = __uint_as_float(rf[1]); RR.rcpC = __uint_as_float(rf[2]); RR.center = __uint_as_float(rf[3]); \
                            RR.left = (int32_t)rf[4]; RR.q2 = (int32_t)rf[5]; RR.zm_raw = rf[6]; RR.sse_raw = rf[7];      \
                        } else RR = piece_rare_of(c, H, SH);                                                                  \
                    }                                                                                                     \
                    if (kind == 1u && !flushed) {              /* huge integers go straight to the slot planes: make them live */ \
                        BRC_FLUSH()                                                                                       \
                    }                                                                                                     \
                    const int lr = BRC_LANE(); const int64_t kr = BRC_KK();                                               \
                    const bool mine = ((mask >> lr) & 1ull) != 0ull;                                                      \
                    const int32_t s_c = p0 - H.a;                                                                         \
                    const uint32_t off = hoff + (m - base) * (uint32_t)ROW_BYTES + ((uint32_t)s_c & 15u);                 \
                    const uint32_t w = (uint32_t)*reinterpret_cast<const uint8_t*>(rows_base + BRC_CKI(c, CK_PILEUP, 17, CB_LDS_ROWS, off + (uint32_t)lr, 2u * HALF * ROW_BYTES, tile, m)); \
                    /* the lane's quality and bucket: from its event byte, or — an escape byte of a wide read — from the wide stream */ \
                    uint32_t eq = w >> 2, ebk = (w & 3u) + 1u;                                                            \
                    if (((H.tp_flags >> 24) & PF_WIDE) && mine && eb_is_escape(w)) {                                      \
                        const uint32_t w16 = *BRC_CK(c, CK_PILEUP, 18, CB_BQW, bqw_ro + ((int64_t)wide_base(bqw_ro, H.bq_off) + (int64_t)(lr + s_c)), 2, tile, m); eq = w16 >> 8; ebk = w16 & 0xffu; \
                    }                                                                                                     \
                    if (kind == 0u) {                          /* third alleles: raw addends to the list, in piece order */ \
                        uint32_t at0 = 0;                                                                                 \
                        if (lr == 0 && !BRC_PVAR(11)) at0 = atomicAdd(BRC_CK(c, CK_PILEUP, 19, CB_XEVN, pl.xev_n + (size_t)xshard * XEV_CTR_STRIDE, 4, tile, m), (uint32_t)__builtin_popcountll(mask)); /* (11: profiling, no list cursor) */ \
                        at0 = (uint32_t)__builtin_amdgcn_readfirstlane(at0);                                              \
                        const uint64_t below = mask & ((1ull << lr) - 1ull);                                              \
                        const uint32_t at = at0 + (uint32_t)__builtin_popcountll(below);                                  \
                        if (mine && at < pl.xev_cap) *BRC_CK(c, CK_PILEUP, 20, CB_XEV, pl.xev + ((size_t)xshard * pl.xev_cap + at), sizeof(XEv), tile, m) = make_xev(c, lib, kr, H, RR, lr + s_c, eq, ebk, SH); \
                    } else if (mine) drain_int(c, pl, lib, kr, RR, ebk == BRC_DOM_B() ? 0u : 1u);                         \
                }                                                                                                         \
                qn = 0;                                                                                                   \
            }                                                                                                             \
            if (__builtin_expect(since_flush + HALF > c.flush_k, 0)) {                                                    \
                BRC_FLUSH()                                                                                               \
            }                                                                                                             \
            if (base + 2u * (uint32_t)HALF < hi) {                                                                        \
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                               \
                BRC_STAGE(T, base + 2u * (uint32_t)HALF, hoff)                                                            \
                BRC_LD_TAB(T, base + 3u * (uint32_t)HALF)                                                                 \
            }                                                                                                             \
        }
        enum { HOFF_X = HALF * ROW_BYTES };                                    // byte offset of the second ring half
        uint4 T; uint32_t pf = 0;
        BRC_LD_TAB(T, lo)
        if (!BRC_PVAR(7)) BRC_STAGE(T, lo, 0u)
        BRC_LD_TAB(T, lo + (uint32_t)HALF)
        if (!BRC_PVAR(7)) BRC_STAGE(T, lo + (uint32_t)HALF, (uint32_t)HOFF_X)
        BRC_LD_TAB(T, lo + 2u * (uint32_t)HALF)
        PRec R0, R1, R2;
        const char* recp = reinterpret_cast<const char*>(pieces4) + (size_t)lo * 48u;   // (scalar) next record to request
        const uint32_t recstep = BRC_PVAR(10) ? 0u : 48u;                   // (profiling: 10 = every scalar load hits the same line)
        BRC_LD_REC(R0, recp) recp += 48;
        BRC_LD_REC(R1, recp) recp += 48;
        asm volatile("" : "=" BRC_F_R2 (R2.f), "=" BRC_G_R2 (R2.g));             // (defined before its first wait: whatever the registers hold; not a copy of a set in flight)
        Stage S0, S1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // first two half-batches staged (tile prologue)
        BRC_WAIT_REC(R0)
        BRC_PROBE(R0, 0u, S0)
        S1 = S0;
        uint32_t hoff = 0;                                                     // byte offset of the current ring half
        uint32_t base = (BRC_PVAR(6) || BRC_PVAR(7)) ? hi : lo;
        for (; base < hi; base += (uint32_t)HALF, hoff ^= (uint32_t)HOFF_X) {
            const uint32_t nb = (hi - base) < (uint32_t)HALF ? (hi - base) : (uint32_t)HALF;
            if (BRC_PVAR(5) && base != lo) continue;
            BRC_STEP(0, R0, R1, R2, S0, S1, true)
            BRC_STEP(1, R1, R2, R0, S1, S0, true)
            BRC_STEP(2, R2, R0, R1, S0, S1, true)
            BRC_STEP(3, R0, R1, R2, S1, S0, true)
            BRC_STEP(4, R1, R2, R0, S0, S1, true)
            BRC_STEP(5, R2, R0, R1, S1, S0, true)
#if BRC_HALF == 12
            BRC_STEP(6, R0, R1, R2, S0, S1, true)
            BRC_STEP(7, R1, R2, R0, S1, S0, true)
            BRC_STEP(8, R2, R0, R1, S0, S1, true)
            BRC_STEP(9, R0, R1, R2, S1, S0, true)
            BRC_STEP(10, R1, R2, R0, S0, S1, true)
            BRC_STEP(11, R2, R0, R1, S1, S0, true)
#endif
            BRC_BOUNDARY(nb)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+" BRC_F_R0 (R0.f), "+" BRC_G_R0 (R0.g), "+" BRC_F_R1 (R1.f), "+" BRC_G_R1 (R1.g), "+" BRC_F_R2 (R2.f), "+" BRC_G_R2 (R2.g));   // no scalar load may outlive its registers
        asm volatile("" :: "v"(pf));
#undef BRC_BOUNDARY
#undef BRC_KK
#undef BRC_LANE
#undef BRC_FLUSH
#undef BRC_STEP
#undef BRC_ACC
#undef BRC_QPUSH
#undef BRC_DOM_REGION
#undef BRC_SEV_ADD
#undef BRC_PROBE
#undef BRC_WAIT_REC
#undef BRC_F_R0
#undef BRC_G_R0
#undef BRC_F_R1
#undef BRC_G_R1
#undef BRC_F_R2
#undef BRC_G_R2
#undef BRC_LD_DIV
#undef BRC_LD_REC
#undef BRC_STAGE
#undef BRC_LD_TAB
    }
    // ---- end of the tile: registers -> the two slots of every position (coalesced: lane == position).  Every plane address
    // is a wave-uniform base in a scalar register pair + the lane's byte offset, one VGPR for all 29 stores; the bases advance
    // by scalar adds.  (Inline assembly: left to the optimiser, the lane offset is folded into ONE 64-bit vector address and
    // the other 28 are derived from it with a 64-bit vector add each.)
    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // (the lane index again: see above)
    if (inreg && !BRC_PVAR(1)) {
        const int64_t P = c.PS;
        const int64_t tb = tile * TILE;                                    // (scalar) plane index of lane 0
        const uint32_t loff = (uint32_t)lane_e << 2;
#ifdef BRC_CHECKED
        // (checked build: the plane bases come out of v_readfirstlane — uniform_ptr — and a VMEM instruction that reads an SGPR a VALU
        // instruction wrote needs five wait states in between; the compiler's hazard recogniser cannot see into the assembly statement)
#define BRC_ST(base, val) asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2 nt" :: "v"(loff), "v"(val), "s"(base) : "memory")
#else
#define BRC_ST(base, val) asm volatile("global_store_dword %0, %1, %2 nt" :: "v"(loff), "v"(val), "s"(base) : "memory")   /* (written once per step, read by nobody on the device: streaming) */
#endif
        a.dom_b = BRC_DOM_B();
        const uint32_t sid = a.dom_b | (a.alt_b << 8);
        // (checked build: the wave's 256-byte segment of every plane — stores are coalesced, lane == position, so the segment of
        // the tile is what each of the 29 store instructions may touch)
        { const uint32_t* q = (const uint32_t*)BRC_CKS(c, CK_PILEUP, 22, CB_NCOL, reinterpret_cast<const char*>(pl.ncol + (int64_t)lib * P + tb), 256, tile, -1); BRC_ST(q, a.ncol); }       // (dead lanes accumulated nothing: zeros)
        { const uint32_t* q = (const uint32_t*)BRC_CKS(c, CK_PILEUP, 23, CB_DEPTH, reinterpret_cast<const char*>(pl.depth + (int64_t)lib * P + tb), 256, tile, -1); BRC_ST(q, a.depth); }
        { const uint32_t* q = (const uint32_t*)BRC_CKS(c, CK_PILEUP, 24, CB_SLOTID, reinterpret_cast<const char*>(pl.slotid + (int64_t)lib * P + tb), 256, tile, -1); BRC_ST(q, sid); }
        uint32_t dv[NI], av[NI];
        pack_unpack(a.dom, eb_index(a.dom_b), (uint32_t)SH, dv); pack_unpack(a.alt, eb_index(a.alt_b), (uint32_t)SH, av);     // (the slots' base indices; no alternate yet: all zeros)
        uint32_t* i0 = (uint32_t*)BRC_CKS(c, CK_PILEUP, 25, CB_SI, reinterpret_cast<const char*>(slot_i(c, pl, lib, 0u, tb)), ((uint64_t)(NI - 1) * (uint64_t)P + 64u) * 4u, tile, -1);
        uint32_t* i1 = (uint32_t*)BRC_CKS(c, CK_PILEUP, 26, CB_SI, reinterpret_cast<const char*>(slot_i(c, pl, lib, 1u, tb)), ((uint64_t)(NI - 1) * (uint64_t)P + 64u) * 4u, tile, -1);
        if (__builtin_expect(flushed, 0)) {
            // (wave-uniform) earlier flushes of this tile left partial integer sums in the planes; dead lanes never flush
#pragma unroll
            for (int f = 0; f < NI; ++f) {
                const uint32_t x0 = i0[(int64_t)f * P + lane_e], x1 = i1[(int64_t)f * P + lane_e];
                dv[f] += dead ? 0u : x0; av[f] += dead ? 0u : x1;
            }
        }
        {
            const uint32_t* q0 = i0; const uint32_t* q1 = i1;
#pragma unroll
            for (int f = 0; f < NI; ++f) { BRC_ST(q0, dv[f]); BRC_ST(q1, av[f]); q0 += P; q1 += P; }
            const float* g0 = (const float*)BRC_CKS(c, CK_PILEUP, 27, CB_SF, reinterpret_cast<const char*>(slot_f(c, pl, lib, 0u, tb)), ((uint64_t)(NF - 1) * (uint64_t)P + 64u) * 4u, tile, -1);
            const float* g1 = (const float*)BRC_CKS(c, CK_PILEUP, 28, CB_SF, reinterpret_cast<const char*>(slot_f(c, pl, lib, 1u, tb)), ((uint64_t)(NF - 1) * (uint64_t)P + 64u) * 4u, tile, -1);
#pragma unroll
            for (int f = 0; f < NF; ++f) { BRC_ST(g0, a.dom.f[f]); BRC_ST(g1, a.alt.f[f]); g0 += P; g1 += P; }
        }
#undef BRC_ST
    }
#undef BRC_DOM_B
    // per-(tile, library) partials; k_finalize sums them (a single-address atomic per wave costs ~12 ns x 780 k waves):
    // events (columns of the reporting window), the lanes' warning counters, abandoned positions, and — when there is one
    // library — the emitted positions (with several, a position prints if ANY library's column is non-empty: k_finalize
    // looks at the ncol planes then)
    const bool rep = valid && (int32_t)(c.pos0 + tile * TILE + lane_e) >= c.beg0;
    const uint32_t ev_lo = wave_sum_u32(rep ? (a.ncol & 0xffffu) : 0u);
    const uint64_t m_big = __builtin_amdgcn_ballot_w64(rep && (a.ncol >> 16) != 0u);
    unsigned long long ev = ev_lo;
    if (__builtin_expect(m_big != 0ull, 0)) ev += (unsigned long long)wave_sum_u32(rep ? (a.ncol >> 16) : 0u) << 16;
    const unsigned long long wsm = wave_sum_u32(valid ? (a.ww & 0xffffu) : 0u) + wsm_tot, wnm = wave_sum_u32(valid ? (a.ww >> 16) : 0u) + wnm_tot;
    const uint32_t wl = lib == 0 ? (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(dead)) : 0u;
    const uint32_t npos = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(rep && a.ncol != 0u));
    // (ev of one tile: 64 lanes x a column count; the 32-bit slot holds it up to 67 M reads deep — deeper, it saturates the
    // warning slots first: all four are summed in 64 bits by k_finalize, a tile's share travels in 32)
    if (lane_e == 0) *BRC_CK(c, CK_PILEUP, 29, CB_TILECTR, tile_ctr + ((int64_t)lib * ntiles + tile), sizeof(uint4), tile, -1) = make_uint4((uint32_t)ev, (uint32_t)wsm, (uint32_t)wnm, wl | (npos << 8));
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

// sum of the per-(tile, library) partials of k_pileup2 (events, warnings, abandoned positions, and with one library the
// emitted positions); with several libraries the emitted positions are counted here from the ncol planes (a position
// prints if any library's column is non-empty).  Grid-stride, per-block partials.
__global__ __launch_bounds__(256) void k_finalize(DevCfg c, const uint32_t* __restrict__ ncol, const uint4* __restrict__ tile_ctr,
                                                  int64_t n_tile_ctr, unsigned long long* __restrict__ part) {
    __shared__ unsigned long long sh[4 * 5];
    unsigned long long v[5] = {0, 0, 0, 0, 0};   // positions, events, w_sm, w_nm, w_lib
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // four positions per thread and load (the plane stride PS is a multiple of 64, the planes are 16-byte aligned)
    if (c.Lp > 1)
    for (int64_t k4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; k4 < c.P; k4 += stride * 4) {
        uint4 tot = make_uint4(0u, 0u, 0u, 0u);
        for (int l = 0; l < c.Lp; ++l) { const uint4 x = *reinterpret_cast<const uint4*>(ncol + (int64_t)l * c.PS + k4); tot.x |= x.x; tot.y |= x.y; tot.z |= x.z; tot.w |= x.w; }
        const int64_t pk = c.pos0 + k4;
        v[0] += (tot.x && pk >= c.beg0 ? 1u : 0u) + (tot.y && pk + 1 >= c.beg0 && k4 + 1 < c.P ? 1u : 0u) +
                (tot.z && pk + 2 >= c.beg0 && k4 + 2 < c.P ? 1u : 0u) + (tot.w && pk + 3 >= c.beg0 && k4 + 3 < c.P ? 1u : 0u);
    }
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_tile_ctr; t += stride) {
        const uint4 x = tile_ctr[t];
        v[1] += x.x; v[2] += x.y; v[3] += x.z; v[4] += x.w & 0xffu;
        if (c.Lp == 1) v[0] += x.w >> 8;
    }
    block_sum_u64<5>(v, sh);
    // per-block partials, summed by k_finalize_sum: thousands of atomics on one cache line cost ~12 ns each
    if (threadIdx.x == 0) for (int i = 0; i < 5; ++i) part[(int64_t)blockIdx.x * 5 + i] = v[i];
}

__global__ __launch_bounds__(256) void k_finalize_sum(const unsigned long long* __restrict__ part, int nblocks, Counters* __restrict__ ctr) {
    __shared__ unsigned long long sh[4 * 5];
    unsigned long long v[5] = {0, 0, 0, 0, 0};
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x)
        for (int i = 0; i < 5; ++i) v[i] += part[(int64_t)b * 5 + i];
    block_sum_u64<5>(v, sh);
    // (w_sm / w_nm: k_indel_reduce adds to them from another stream)
    if (threadIdx.x == 0) { ctr->n_positions += v[0]; ctr->n_events += v[1]; atomicAdd(&ctr->w_sm, v[2]); atomicAdd(&ctr->w_nm, v[3]); ctr->w_lib += v[4]; }
}

// ---------------------------------------------------------------- indel side path

// Third-allele sub-lists -> one list (sub-list order; the events of one position all sit in one sub-list, in append
// order, which is all their fold needs): block s copies its entries behind those of the sub-lists before it, and counts them per
// (64-position tile, library) bucket for the fold below.
__global__ __launch_bounds__(256) void k_xev_compact(const XEv* __restrict__ lists, const uint32_t* __restrict__ cursors, uint32_t cap, uint32_t shards,
                                                    XEv* __restrict__ out, Counters* __restrict__ ctr, uint32_t* __restrict__ bucket_cnt, int Lp) {
    __shared__ unsigned long long sh[4];
    const uint32_t s = blockIdx.x;
    unsigned long long before = 0; uint32_t mx = 0;
    for (uint32_t t = threadIdx.x; t < shards; t += 256) {
        const uint32_t n = cursors[(size_t)t * XEV_CTR_STRIDE];
        if (t < s) before += n < cap ? n : cap;
        mx = n > mx ? n : mx;
    }
    before = wave_sum_u64(before);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t v = (uint32_t)__shfl_xor((int)mx, o, 64); mx = v > mx ? v : mx; }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = before;
    __syncthreads();
    const unsigned long long off = sh[0] + sh[1] + sh[2] + sh[3];
    const uint32_t mine = cursors[(size_t)s * XEV_CTR_STRIDE], n = mine < cap ? mine : cap;
    const uint4* src = reinterpret_cast<const uint4*>(lists + (size_t)s * cap); uint4* dst = reinterpret_cast<uint4*>(out + off);
    for (uint32_t i = threadIdx.x; i < n * 3u; i += 256) {                 // 48-byte entries as 3 x 16 bytes
        const uint4 v = src[i]; dst[i] = v;
        if (i % 3u == 0u) atomicAdd(bucket_cnt + ((size_t)(v.x >> 6) * (size_t)Lp + (size_t)(v.y >> 8)), 1u);      // (first 16 bytes of an entry: k, library << 8 | bucket)
    }
    if (s == shards - 1 && threadIdx.x == 0) ctr->n_xev = (unsigned int)(off + n);
    if (s == 0 && (threadIdx.x & 63) == 0) atomicMax(&ctr->xev_max, mx);
}
// ... the compacted events' indices into their buckets (cursor[] = the exclusive scan of the counts, advanced to the buckets' ends) ...
__global__ __launch_bounds__(256) void k_xev_scatter(const XEv* __restrict__ list, const Counters* __restrict__ ctr, uint32_t* __restrict__ cursor, uint32_t* __restrict__ idx, int Lp) {
    const uint32_t n = ctr->n_xev;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const XEv& e = list[i];
        idx[atomicAdd(cursor + ((size_t)(e.k >> 6) * (size_t)Lp + (size_t)(e.lib_b >> 8)), 1u)] = i;
    }
}
// ... and the fold (brc_core.h: fold_xev_bucket), one lane per bucket that holds events: BasicStat::process_read's sums (BasicStat.cpp:28-107)
// of the events that found both slots of their position taken, in column order.  A bucket's records take the first slots of its run in
// out[]; end[] / cnt[] are rewritten to name them ([end - cnt, end)), the run's other slots are marked unused (k = NONE32).
__global__ __launch_bounds__(256) void k_xev_fold(const XEv* __restrict__ list, uint32_t* __restrict__ cnt, uint32_t* __restrict__ end, int64_t n_buckets, uint32_t* __restrict__ idx, XAgg* __restrict__ out) {
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < n_buckets; b += (int64_t)gridDim.x * 256) {
        const uint32_t n = cnt[b];
        if (!n) continue;
        const uint32_t start = end[b] - n;
        const int nd = fold_xev_bucket(list, idx + start, (int)n, out + start);
        for (uint32_t j = (uint32_t)nd; j < n; ++j) out[start + j].k = NONE32;
        cnt[b] = (uint32_t)nd; end[b] = start + (uint32_t)nd;
    }
}
// The last plane index at which every library was processed (a non-empty, not abandoned column: the positions pileup_func's loop body ran
// for that library, :360): what the host needs to know which deletions a region leaves queued behind it (brc_host.cpp: format_device_text).
// One block per library, from the back; NONE32: never.
__global__ __launch_bounds__(256) void k_last_processed(DevCfg c, const uint32_t* __restrict__ ncol, const uint32_t* __restrict__ unavail, uint32_t* __restrict__ last) {
    __shared__ uint32_t best;
    const int l = blockIdx.x;
    if (threadIdx.x == 0) best = 0u;                                          // (plane index + 1; 0: none yet)
    __syncthreads();
    for (int64_t hi = c.P; hi > 0; hi -= 256) {
        const int64_t k = hi - 1 - (int64_t)threadIdx.x;
        if (k >= 0 && ncol[(int64_t)l * c.PS + k] != 0u && !(c.per_lib && unavail[k] != NONE32)) atomicMax(&best, (uint32_t)k + 1u);
        __syncthreads();
        const uint32_t seen = best;
        __syncthreads();
        if (seen) break;
    }
    if (threadIdx.x == 0) last[l] = best ? best - 1u : NONE32;
}

// Device-side text (brc_core.h: text_line): byte length of every position's line, then — after an exclusive scan — the bytes.
// One lane per position; a lane's stores walk its own line, neighbouring lanes write neighbouring lines.
__global__ __launch_bounds__(256) void k_text_len(DevCfg c, DevIn in, Planes pl, TextCtx t, TextAux ax, uint32_t* __restrict__ len) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > c.P) return;
    len[k] = k < c.P ? text_line(c, in, pl, t, ax, k, nullptr) : 0u;        // (entry P: the scan turns it into the total)
}
// the true total of the line lengths, in 64 bits (the offsets are 32-bit: a region whose text would pass 4 GiB must not be written)
__global__ __launch_bounds__(256) void k_text_total(const uint32_t* __restrict__ len, int64_t n, unsigned long long* __restrict__ total) {
    unsigned long long v = 0ull;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) v += len[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (unsigned long long)__shfl_xor((long long)v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(total, v);
}
__global__ __launch_bounds__(256) void k_text_write(DevCfg c, DevIn in, Planes pl, TextCtx t, TextAux ax, const uint32_t* __restrict__ off, char* __restrict__ text) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= c.P || off[k + 1] == off[k]) return;
    (void)text_line(c, in, pl, t, ax, k, text + off[k]);
}

// raw indel events (K1: one slot per I / D / P operator, unused ones marked NONE32) -> their (16 or 64 positions, library) buckets;
// cursor[] holds the buckets' start offsets (exclusive scan of K1's counts) and ends up at their ends
__global__ __launch_bounds__(256) void k_indel_scatter(DevCfg c, const IndelEv* __restrict__ raw, int64_t n_raw, uint32_t* __restrict__ cursor,
                                                       IndelEv* __restrict__ ev) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_raw) return;
    const IndelEv e = raw[i];
    if (e.key_lo == NONE32) return;
    ev[atomicAdd(&cursor[indel_bucket(c, e.key_lo)], 1u)] = e;
}

// -p: reads without a library abandon every position they cover (bamreadcount.cpp:281-284).  unavail[k] = index of the
// first such read in the column (NONE32 when there is none); those reads have no pieces.
__global__ __launch_bounds__(256) void k_unavail(DevCfg c, DevIn in, uint32_t* __restrict__ unavail) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.n_reads || in.lib[i] >= 0) return;
    const uint32_t nc = in.n_cigar[i]; const uint32_t* cg = in.cigar + in.cig_off[i];
    if (!read_enters(in.flag[i], cg, nc) || in.pos[i] < 0) return;
    int64_t rlen = 0;
    for (uint32_t k = 0; k < nc; ++k) if (is_refop(cg[k] & 0xfu)) rlen += (int64_t)(cg[k] >> 4);
    int64_t k0 = (int64_t)in.pos[i] - c.pos0, k1 = k0 + rlen;
    if (k0 < 0) k0 = 0;
    if (k1 > c.P) k1 = c.P;
    for (int64_t k = k0; k < k1; ++k) atomicMin(&unavail[k], (uint32_t)i);
}

// cursor[b] has been advanced by k_indel_scatter to the END of bucket b's events, which sit in consecutive slots
// [end - cnt[b], end).  One lane per bucket (buckets with events are a few in a hundred): reduce_indel_bucket sorts the
// bucket and folds every key; a key's alleles are written to the first slots of its run in out[], unused slots get len = 0.
__global__ __launch_bounds__(256) void k_indel_reduce(DevCfg c, DevIn in, const DRead* __restrict__ reads, const uint32_t* __restrict__ cnt,
                                                      const uint32_t* __restrict__ cursor, int64_t n_buckets, IndelEv* __restrict__ ev,
                                                      const uint32_t* __restrict__ unavail, IndelOut* __restrict__ out,
                                                      Counters* __restrict__ ctr) {
    __shared__ unsigned long long sh[4 * 2];
    unsigned long long w[2] = {0, 0};
    if (blockIdx.x == 0 && threadIdx.x == 0) ctr->n_indel_slots = n_buckets > 0 ? cursor[n_buckets - 1] : 0u;   // total number of event slots
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_buckets; b += stride) {
        const uint32_t n = cnt[b];
        if (!n) continue;
        const uint32_t start = cursor[b] - n;
        uint32_t wsm = 0, wnm = 0;
        reduce_indel_bucket(c, in, reads, ev + start, (int)n, unavail, out + start, wsm, wnm);
        w[0] += wsm; w[1] += wnm;
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

static std::atomic<uint64_t> g_dev_allocs{0}, g_dev_alloc_ns{0};     // BRC_ENGINE_TIMING: (re)allocations of device buffers and the time they took (several engines allocate from their own threads)
struct DBuf {
    void* p = nullptr; size_t cap = 0;
    size_t req = 0;      // the bytes the last ensure() asked for: the extent the kernels may touch (the checked build compares addresses with it, not with the rounded-up capacity)
    hipError_t ensure(size_t bytes) {
        req = bytes;
        if (bytes <= cap) return hipSuccess;
        struct Tm { std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                    ~Tm() { g_dev_alloc_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); ++g_dev_allocs; } } tm_;
        // a FIRST allocation takes an eighth more than asked (the 18 GB of a config-5 contig must not be padded by half); a buffer that has to
        // grow again belongs to a run of regions of fluctuating size (the command line's 1-Mbp pieces): half more, so that the run settles
        // after a few pieces instead of paying hipFree (a device-wide synchronisation) + hipMalloc for 40-odd buffers on every new maximum
        const bool regrow = p != nullptr;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + (regrow ? bytes / 2 : bytes / 8) + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { e = hipMalloc(&p, bytes); want = bytes; }
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

enum { T_ANNOTATE = 0, T_SCAN_ENDS, T_TILES, T_PILEUP, T_COUNT, T_INDEL_SCAN, T_INDEL_FILL, T_INDEL_REDUCE, T_N };
static const char* kKernelNames[BRC_NKERNEL] = {"k_annotate", "k_scan_ends", "k_tiles", "k_pileup", "k_xev_fold+finalize",
                                                "k_scan_indel", "k_indel_fill", "k_indel_reduce"};

#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { return hip_fail(_e, #x); } } while (0)

class HipBackend : public Backend {
    std::string err;
    int device = 0;
    hipStream_t stream = nullptr;
    // one set of timing events per pass in flight: brc_compute uses set 0, brc_compute_n a ring of them (passes queued back to
    // back record into sets of their own, read after the one wait at the end of a batch)
    enum { EV_RING = 32 };
    struct EvSet { hipEvent_t evt[T_N + 1]; hipEvent_t ev_indel[4]; };
    std::vector<EvSet> evsets;
    hipEvent_t* evt = nullptr;           // the current pass's sets (point into evsets)
    bool have_events = false;
    DevCfg c; DevIn in;
    int64_t ntiles = 0; uint64_t n_indel_cap = 0;
    std::vector<int64_t> lib_base;      // first piece of every library's stream (Lp + 1 entries)
    // device buffers
    DBuf d_pos, d_flag, d_mapq, d_lib, d_lq, d_nc, d_co, d_so, d_qo, d_nm, d_sm, d_tags, d_cigar, d_seq, d_qual, d_ref, d_refcode;
    std::vector<uint16_t> h_wanted; bool has_wanted = false; DBuf d_wanted; std::vector<uint32_t> h_tilelist; DBuf d_tilelist;      // brc_region_windows (kept alive for the asynchronous copy)
    HBuf<Staged::WidePair> h_wpairs; unsigned wide_slices = 1;      // (page-locked: the list is every read of a run with HiFi qualities)
    DBuf d_wpairs;
    DBuf d_bq, d_bqw, d_bqrow, d_pieceoff, d_pieces, d_rare, d_keyreach, d_libbase, d_reads, d_agg, d_rng, d_ncol, d_depth, d_slotid, d_si, d_sf, d_xev, d_xevc, d_xevn, d_unavail, d_cnt, d_cursor, d_ev, d_evraw, d_ievoff, d_iout, d_ctr, d_tilectr, d_part, d_wavelist, d_nc_k1;
    DBuf d_tlen, d_toff, d_text, d_tctx, d_total64, d_lastproc;
    DBuf d_xcnt, d_xend, d_xidx, d_xagg;   // the third-allele fold: events per (tile, library) bucket, the buckets' ends, the events' indices by bucket, the folded records
    // device-side text, downloaded (pinned) on its own stream into one of two host buffers
    HBuf<char> h_text[2]; HBuf<uint32_t> h_toff[2]; HBuf<uint32_t> h_total;
    hipStream_t stream2 = nullptr; hipEvent_t ev_text[2] = {nullptr, nullptr}, ev_lines = nullptr;
    hipStream_t stream3 = nullptr; hipEvent_t* ev_indel = nullptr; DBuf d_agg2;   // the indel side path's stream
    bool text_started[2] = {false, false}; uint64_t text_total[2] = {0, 0}; int64_t text_n[2] = {0, 0}; int text_slot = 0;
    Planes pl_last;                      // the planes of the last compute
    // tile compaction (k_compact_tiles): on for regions whose reads average more than COMPACT_PIECES_PER_READ pieces
    enum { COMPACT_PIECES_PER_READ = 12 };
    bool compact_on = false; uint64_t compact_total = 0; bool compact_sized = false;
    bool wave_on = false;                      // this region's reads with more than two M operators go to k_annotate_wave
    bool wave_big = false;                     // ... and some may have more than AW_MCAP M operators (a read with more than AW_MCAP operators exists)
    enum { WAVE_FORM_BLOCKS = 768,             // its fixed grid: 3 blocks of 4 waves per CU (48 KB of LDS each)
           WAVE_FORM_BLOCKS_BIG = 512,         // the one-wave instantiation: 2 blocks per CU (60 KB of LDS each)
           WAVE_FORM_BLOCKS_HUGE = 256 };      // one block per CU (158 KB of LDS)
    bool wave_huge = false;                    // ... or more than AW_MCAP_BIG
    bool wave_eqx = false; uint32_t s_max_ncigar = 0;   // reads with = / X operators exist: the EQX instantiations are launched too
    bool cursor_on = false;                    // a read with an empty M / = / X operator was staged: k_pick_wave lists such reads for k_annotate_cursor
    unsigned long long h_steps[3] = {0, 0, 0};   // piece-steps of the last pass: what the tile ranges hold / what k_pileup2 walked (brc_region_piece_steps)
    DBuf d_ccnt, d_coff, d_cpieces, d_crare, d_crng, d_ctot;
    // host result buffers (pinned)
    HBuf<uint32_t> h_ncol, h_depth, h_slotid, h_si, h_unavail; HBuf<float> h_sf; HBuf<IndelOut> h_iout; HBuf<XAgg> h_xagg; HBuf<uint32_t> h_lastproc[2];
    std::vector<XAgg> xagg_compact;
    size_t xev_cap = 0;                  // entries per sub-list
    enum { XEV_SHARDS = 1024 };
    std::vector<IndelOut> iout_compact;
    Counters h_ctr;
    bool computed = false;
#ifdef BRC_CHECKED
    // the bounds-checked build: extents of every buffer K1 and k_pileup2 address (what the host asked for), uploaded in front of
    // every pass; the fault record is read back behind it (finish_passes)
    DBuf d_chk; ChkState h_chk;
    void chk_fill() {
        memset(&h_chk, 0, sizeof h_chk);
        auto set = [&](int b, const DBuf& d) { h_chk.ext[b].lo = (uint64_t)d.p; h_chk.ext[b].hi = (uint64_t)d.p + d.req; };
        set(CB_CIGAR, d_cigar); set(CB_SEQ, d_seq); set(CB_QUAL, d_qual); set(CB_REFCODE, d_refcode); set(CB_EB, d_bq); set(CB_BQW, d_bqw);
        set(CB_PIECES, d_pieces); set(CB_RARE, d_rare); set(CB_KEYREACH, d_keyreach); set(CB_READS, d_reads); set(CB_EVRAW, d_evraw); set(CB_CNT, d_cnt);
        set(CB_WANTED, d_wanted); set(CB_RNG, d_rng); set(CB_UNAVAIL, d_unavail); set(CB_TILELIST, d_tilelist); set(CB_NCOL, d_ncol); set(CB_DEPTH, d_depth);
        set(CB_SLOTID, d_slotid); set(CB_SI, d_si); set(CB_SF, d_sf); set(CB_XEV, d_xev); set(CB_XEVN, d_xevn); set(CB_TILECTR, d_tilectr);
        // what k_pileup2 reads its records and ranges from: K1's stream, or the compacted one (k_compact_tiles)
        const bool cz = compact_on && compact_sized;
        set(CB_KP_PIECES, cz ? d_cpieces : d_pieces); set(CB_KP_RARE, cz ? d_crare : d_rare); set(CB_KP_RNG, cz ? d_crng : d_rng);
        // (self-test of the checker, tests/test_checked_build.py: BRC_CHECKED_SHRINK=<buffer index>:<bytes> takes bytes off a buffer's end)
        if (const char* sh = getenv("BRC_CHECKED_SHRINK")) { const int b = atoi(sh); const char* cpos = strchr(sh, ':'); if (b >= 0 && b < CB_N && cpos) { const uint64_t by = strtoull(cpos + 1, nullptr, 10); ChkExt& x = h_chk.ext[b]; x.hi = x.hi - x.lo > by ? x.hi - by : x.lo; } }
    }
    int chk_report() {
        if (hipMemcpy(&h_chk, d_chk.p, sizeof h_chk, hipMemcpyDeviceToHost) != hipSuccess) { err = "checked build: cannot read the fault record"; return BRC_E_HIP; }
        if (!h_chk.count) return BRC_OK;
        static const char* const kBuf[CB_N] = {"cigar", "seq4", "qual", "refcode", "event bytes", "wide words", "pieces", "rare records", "keyreach", "reads", "raw indel events", "indel bucket counts",
                                               "wanted lanes", "tile ranges", "unavail", "tile list", "ncol", "depth", "slotid", "integer slot planes", "float slot planes", "third-allele lists",
                                               "third-allele cursors", "tile counters", "LDS rows", "LDS queue", "pieces (pileup)", "rare records (pileup)", "tile ranges (pileup)"};
        char b[512];
        snprintf(b, sizeof b, "BRC_CHECKED: %u out-of-bounds device access(es); first: kernel %s, site %u, buffer '%s' [0x%llx, 0x%llx), address 0x%llx + %llu bytes (offset %lld), %s %lld, piece %lld",
                 h_chk.count, h_chk.kernel == CK_ANNOTATE ? "k_annotate_groups" : "k_pileup2", h_chk.site, h_chk.buf < CB_N ? kBuf[h_chk.buf] : "?",
                 (unsigned long long)h_chk.ext[h_chk.buf < CB_N ? h_chk.buf : 0].lo, (unsigned long long)h_chk.ext[h_chk.buf < CB_N ? h_chk.buf : 0].hi, (unsigned long long)h_chk.addr,
                 (unsigned long long)h_chk.bytes, (long long)(h_chk.addr - h_chk.ext[h_chk.buf < CB_N ? h_chk.buf : 0].lo), h_chk.kernel == CK_ANNOTATE ? "read" : "tile", (long long)h_chk.unit, (long long)h_chk.piece);
        err = b; fprintf(stderr, "%s\n", b);
        return BRC_E_HIP;
    }
#endif

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
        { const int rc0 = ensure_evsets(1); if (rc0) return rc0; }
        HIPCHK(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&stream3, hipStreamNonBlocking));
        {
            uint32_t* d_bad = nullptr; uint32_t bad = 0;
            HIPCHK(hipMalloc(&d_bad, sizeof(uint32_t)));
            HIPCHK(hipMemsetAsync(d_bad, 0, sizeof(uint32_t), stream));
            hipLaunchKernelGGL(k_divcheck, dim3((DIV_SMALL_N * DIV_SMALL_M + 255) / 256), dim3(256), 0, stream, d_bad);
            HIPCHK(hipMemcpyAsync(&bad, d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            (void)hipFree(d_bad);
            if (bad) { err = "this device's reciprocal does not give correctly rounded small-integer quotients (div_small self-check failed)"; return BRC_E_HIP; }
        }
        for (int i = 0; i < 2; ++i) { HIPCHK(hipEventCreateWithFlags(&ev_text[i], hipEventDisableTiming)); h_text[i].A = &kPinned; h_toff[i].A = &kPinned; }
        HIPCHK(hipEventCreateWithFlags(&ev_lines, hipEventDisableTiming));
        h_total.A = &kPinned;
        h_ncol.A = h_depth.A = h_slotid.A = h_si.A = h_unavail.A = &kPinned; h_sf.A = &kPinned; h_iout.A = &kPinned; h_xagg.A = &kPinned; h_wpairs.A = &kPinned; h_lastproc[0].A = h_lastproc[1].A = &kPinned;
        return BRC_OK;
    }
    ~HipBackend() override {
        (void)hipSetDevice(device);
        if (getenv("BRC_ENGINE_TIMING")) fprintf(stderr, "device buffers: %llu (re)allocations, %.3f s\n", (unsigned long long)g_dev_allocs.load(), (double)g_dev_alloc_ns.load() * 1e-9);
        DBuf* all[] = {&d_pos, &d_flag, &d_mapq, &d_lib, &d_lq, &d_nc, &d_co, &d_so, &d_qo, &d_nm, &d_sm, &d_tags, &d_cigar, &d_seq, &d_qual,
                       &d_ref, &d_refcode, &d_bq, &d_bqw, &d_bqrow, &d_pieceoff, &d_pieces, &d_rare, &d_keyreach, &d_libbase, &d_reads, &d_agg, &d_rng, &d_ncol, &d_depth, &d_slotid, &d_si, &d_sf, &d_xev, &d_xevc, &d_xevn, &d_unavail, &d_cnt,
                       &d_cursor, &d_ev, &d_evraw, &d_ievoff, &d_iout, &d_ctr, &d_tilectr, &d_part, &d_tlen, &d_toff, &d_text, &d_tctx, &d_total64, &d_lastproc, &d_xcnt, &d_xend, &d_xidx, &d_xagg, &d_wanted, &d_tilelist, &d_wavelist, &d_nc_k1, &d_wpairs};
        for (DBuf* b : all) b->release();
        d_ccnt.release(); d_coff.release(); d_cpieces.release(); d_crare.release(); d_crng.release(); d_ctot.release();
#ifdef BRC_CHECKED
        d_chk.release();
#endif
        for (int i = 0; i < 2; ++i) { h_text[i].destroy(); h_toff[i].destroy(); if (ev_text[i]) (void)hipEventDestroy(ev_text[i]); }
        h_total.destroy();
        if (ev_lines) (void)hipEventDestroy(ev_lines);
        if (stream2) (void)hipStreamDestroy(stream2);
        if (stream3) (void)hipStreamDestroy(stream3);
        d_agg2.release();
        h_ncol.destroy(); h_depth.destroy(); h_slotid.destroy(); h_si.destroy(); h_unavail.destroy(); h_sf.destroy(); h_iout.destroy(); h_xagg.destroy(); h_wpairs.destroy(); h_lastproc[0].destroy(); h_lastproc[1].destroy();
        if (w_init) { w_ncol.destroy(); w_depth.destroy(); w_slotid.destroy(); w_si.destroy(); w_unavail.destroy(); w_sf.destroy(); }
        for (EvSet& es : evsets) { for (int i = 0; i <= T_N; ++i) if (es.evt[i]) (void)hipEventDestroy(es.evt[i]); for (int i = 0; i < 4; ++i) if (es.ev_indel[i]) (void)hipEventDestroy(es.ev_indel[i]); }
        if (stream) (void)hipStreamDestroy(stream);
    }
    int ensure_evsets(size_t n) {
        while (evsets.size() < n) {
            EvSet es; memset(&es, 0, sizeof es);
            evsets.push_back(es);
            for (int i = 0; i <= T_N; ++i) HIPCHK(hipEventCreate(&evsets.back().evt[i]));
            for (int i = 0; i < 4; ++i) HIPCHK(hipEventCreate(&evsets.back().ev_indel[i]));
        }
        evt = evsets[0].evt; ev_indel = evsets[0].ev_indel; have_events = true;
        return BRC_OK;
    }
    const HostAlloc* host_alloc() override { return &kPinned; }
    const char* last_error() const override { return err.c_str(); }

    template <class T>
    int up(DBuf& d, const HBuf<T>& h, size_t n) {
        HIPCHK(d.ensure((n + 16) * sizeof(T)));
        if (n) HIPCHK(hipMemcpyAsync(d.p, h.p, n * sizeof(T), hipMemcpyHostToDevice, stream));
        return BRC_OK;
    }

    // SEQ / QUAL: from the staging copy, or — brc_push_reads_pinned — segment by segment from the caller's page-locked buffers
    int up_arena(DBuf& d, const HBuf<uint8_t>& h, const std::vector<Staged::Seg>& segs, uint64_t total) {
        if (segs.empty()) return up(d, h, h.n);
        HIPCHK(d.ensure((size_t)total + 16));
        for (const Staged::Seg& g : segs) if (g.n) HIPCHK(hipMemcpyAsync((uint8_t*)d.p + g.off, g.p, (size_t)g.n, hipMemcpyHostToDevice, stream));
        return BRC_OK;
    }
    bool adopts_arenas() const override { return true; }

    int upload(const brc_config& cfg, const Staged& s, Geometry& g) override {
        HIPCHK(hipSetDevice(device));
        computed = false;
        memset(&c, 0, sizeof c);
        c.min_mapq = cfg.min_mapq; c.min_bq = cfg.min_bq; c.per_lib = cfg.per_lib; c.insertion_centric = cfg.insertion_centric;
        c.Lp = g.Lp; c.ref_len_check = cfg.ref_len_check; c.has_ref = g.ref != nullptr;
        g.PS = (g.P + 63) & ~(int64_t)63;
        c.beg0 = g.beg0; c.end = g.end; c.pos0 = g.pos0; c.P = g.P; c.PS = g.PS; c.ref_lo = g.ref_lo; c.ref_hi = g.ref_hi; c.ref_len = g.ref_len;
        c.n_reads = s.n; c.table_len = test_knob(TK_NO_TABLE) ? 0 : s.modal_len();
        c.n_pieces = s.n_pieces; lib_base = s.lib_base; c.max_lqseq = s.max_lqseq;
        c.ibucket_shift = indel_bucket_shift(s.n_indel_ops, c.P, c.Lp);
        if (const char* ib = test_knob(TK_IBUCKET_SHIFT)) { const int v = atoi(ib); if (v == 2 || v == 4 || v == 6) c.ibucket_shift = v; }   // (test knob: the supported sizes; anything else is ignored)
        // test knobs (brc_host.h: TestKnob — the constant nullptr in the product): small K -> flushes, small limit -> PF_HUGE, forced dominant bucket -> third alleles
        choose_pack(s.max_lqseq, test_knob(TK_FLUSH_K) ? atoi(test_knob(TK_FLUSH_K)) : 0, test_knob(TK_PACK_LIM) ? atoi(test_knob(TK_PACK_LIM)) : 0, c.flush_k, c.pack_lim, c.pack_lim_lo, c.pack_shift);
        c.force_dom = test_knob(TK_FORCE_DOM) ? atoi(test_knob(TK_FORCE_DOM)) : -1;
#ifdef BRC_CHECKED
        HIPCHK(d_chk.ensure(sizeof(ChkState))); c.chk = d_chk.p;
#endif
#ifdef BRC_EXP_KNOBS
        c.variant = getenv("BRC_PILEUP_VARIANT") ? atoi(getenv("BRC_PILEUP_VARIANT")) : 0;
        c.ann_variant = getenv("BRC_ANN_VARIANT") ? atoi(getenv("BRC_ANN_VARIANT")) : 0;
#endif
        // reads with many operators: compact every tile's piece range before the pileup (k_compact_tiles); TK_COMPACT: 1 forces it, 0 forbids
        compact_on = s.n > 0 && s.n_pieces > (int64_t)COMPACT_PIECES_PER_READ * s.n;
        if (const char* ck = test_knob(TK_COMPACT)) compact_on = atoi(ck) != 0 && s.n_pieces > 0;
        compact_sized = false; compact_total = 0; h_steps[0] = h_steps[1] = h_steps[2] = 0;
        const size_t n = (size_t)s.n;
        int rc;
        if ((rc = up(d_pos, s.pos, n)) || (rc = up(d_flag, s.flag, n)) || (rc = up(d_mapq, s.mapq, n)) || (rc = up(d_lib, s.lib, n)) ||
            (rc = up(d_lq, s.l_qseq, n)) || (rc = up(d_nc, s.n_cigar, n)) || (rc = up(d_co, s.cig_off, n)) || (rc = up(d_so, s.seq_off, n)) ||
            (rc = up(d_qo, s.qual_off, n)) || (rc = up(d_nm, s.nm, n)) || (rc = up(d_sm, s.sm, n)) || (rc = up(d_tags, s.tags, n)) ||
            (rc = up(d_cigar, s.cigar, s.cigar.n)) || (rc = up_arena(d_seq, s.seq4, s.seq_seg, s.seq_total)) || (rc = up_arena(d_qual, s.qual, s.qual_seg, s.qual_total)))
            return rc;
        const size_t rl = (size_t)(g.ref_hi - g.ref_lo);
        HIPCHK(d_ref.ensure(rl + 16));
        HIPCHK(d_refcode.ensure(rl + 2 * REFCODE_PAD + 16));
        if (rl) HIPCHK(hipMemcpyAsync(d_ref.p, g.ref + g.ref_lo, rl, hipMemcpyHostToDevice, stream));
        in.pos = (const int32_t*)d_pos.p; in.flag = (const uint16_t*)d_flag.p; in.mapq = (const uint8_t*)d_mapq.p; in.lib = (const int16_t*)d_lib.p;
        in.l_qseq = (const int32_t*)d_lq.p; in.n_cigar = (const uint32_t*)d_nc.p; in.cig_off = (const uint64_t*)d_co.p;
        in.seq_off = (const uint64_t*)d_so.p; in.qual_off = (const uint64_t*)d_qo.p; in.nm = (const int32_t*)d_nm.p; in.sm = (const int32_t*)d_sm.p;
        in.tags = (const uint8_t*)d_tags.p; in.cigar = (const uint32_t*)d_cigar.p; in.seq4 = (const uint8_t*)d_seq.p; in.qual = (const uint8_t*)d_qual.p;
        in.ref = (const char*)d_ref.p;
        // event-byte stream, padded on both sides: a staged window starts up to 79 elements before / ends after a row
        // (the wide stream — full words of the few 8-base groups with an escape byte — has rows for the reads the host found such
        // a base in: DevIn.bqw)
        enum { BQ_PAD = EB_PAD_FRONT };
        HIPCHK(d_bq.ensure(s.bq_elems + BQ_PAD + EB_PAD_BACK + (size_t)std::max<int32_t>(s.max_lqseq, 0)));     // (brc_core.h: stage_window_start — a staged window ends at most that far past the stream)
        {   // the sparse wide stream: [table: one u32 per 16 elements of the byte stream][rows of the wide reads] (brc_core.h: DevIn.bqw)
            if (s.bq_elems >> 34) { err = "region too large: 2^34 bases and more"; return BRC_E_LIMIT; }      // (k_pileup2 reaches a table entry through a 32-bit byte offset)
            const size_t tab_bytes = ((((size_t)(s.bq_elems >> 4) + 2) * 4) + 255) & ~(size_t)255;
            h_wpairs.clear();
            { const size_t nw = s.n_wide(); if (!h_wpairs.reserve(nw + 1)) { err = "host allocation failed (wide-stream list)"; return BRC_E_NOMEM; } h_wpairs.n = nw; }
            const uint64_t wq_elems = s.wide_layout(h_wpairs.p, (uint32_t)(tab_bytes / 32));
            {   // k_wide_rows: a wave per 64 listed reads and slice of their rows' chunks — 512 chunks (eight rounds) per wave and more
                const uint64_t groups = (h_wpairs.n + 63) / 64, per_group = groups ? (wq_elems >> 4) / groups : 0;
                wide_slices = (unsigned)std::min<uint64_t>(std::max<uint64_t>(per_group / 512, 1), 64);
            }
            if ((tab_bytes / 32 + (wq_elems >> 4)) >> 32) { err = "region too large: 2^36 bases of reads with escape bases"; return BRC_E_LIMIT; }
            HIPCHK(d_bqw.ensure(tab_bytes + ((size_t)wq_elems + 16) * sizeof(uint16_t)));
            in.bqw = (const uint16_t*)d_bqw.p;
#ifdef BRC_CHECKED
            HIPCHK(hipMemsetAsync(d_bqw.p, 0xff, tab_bytes, stream));        // (an entry nobody set points far outside the stream)
#endif
            if (h_wpairs.n) {
                HIPCHK(d_wpairs.ensure(h_wpairs.n * sizeof(Staged::WidePair)));
                HIPCHK(hipMemcpyAsync(d_wpairs.p, h_wpairs.p, h_wpairs.n * sizeof(Staged::WidePair), hipMemcpyHostToDevice, stream));
            }
        }
        in.eb = (const uint8_t*)d_bq.p + BQ_PAD;
        if ((rc = up(d_bqrow, s.bq_row, n)) || (rc = up(d_pieceoff, s.piece_off, n))) return rc;
        in.bq_row = (const uint64_t*)d_bqrow.p;
        in.rcp = nullptr;
        const size_t np = (size_t)c.n_pieces;
        HIPCHK(d_pieces.ensure((np + 4) * sizeof(Piece))); HIPCHK(d_rare.ensure((np + 2) * sizeof(PieceRare)));      // (the read loop requests records up to two past the last)
        HIPCHK(d_keyreach.ensure((np + 16) * sizeof(int2)));
        // reads with more than two M operators are annotated a wave per read (k_annotate_wave); TK_WAVE_FORM=0 keeps them on K1's serial path
        wave_eqx = n > 0 && c.has_ref && s.has_eqx && s.max_ncigar >= 3;            // (= / X operators: three match operators take three operators)
        wave_on = n > 0 && c.has_ref && (s.max_ncigar >= 5 || wave_eqx);            // (three M operators take at least five operators)
        s_max_ncigar = s.max_ncigar;
        if (const char* wk = test_knob(TK_WAVE_FORM)) { wave_on = wave_on && atoi(wk) != 0; wave_eqx = wave_eqx && wave_on; }
        // (k_pick_wave sorts the reads by their count of M operators, which the host does not keep: a read can have that many only with at
        // least as many operators — adjacent M operators are legal —, so the operator count decides which instantiations are launched; one
        // whose list stays empty costs a launch)
        wave_big = wave_on && s.max_ncigar > (uint32_t)AW_MCAP;
        wave_huge = wave_on && s.max_ncigar > (uint32_t)AW_MCAP_BIG;
        cursor_on = n > 0 && s.has_empty_m;
        if (wave_on || cursor_on) { HIPCHK(d_wavelist.ensure((4 * (size_t)n + 96) * sizeof(uint32_t))); HIPCHK(d_nc_k1.ensure(((size_t)n + 16) * sizeof(uint32_t))); }
        HIPCHK(d_libbase.ensure((lib_base.size() + 1) * sizeof(int64_t)));
        if (!lib_base.empty()) HIPCHK(hipMemcpyAsync(d_libbase.p, lib_base.data(), lib_base.size() * sizeof(int64_t), hipMemcpyHostToDevice, stream));
        // outputs / scratch
        const size_t P = (size_t)c.PS, Lp = (size_t)c.Lp;   // allocation sizes use the padded stride
        ntiles = (c.P + TILE - 1) / TILE;
        n_indel_cap = c.has_ref ? s.n_indel_ops : 0;
        h_wanted = s.wanted_tiles(c.pos0, c.P); has_wanted = !h_wanted.empty();
        h_tilelist.clear();
        if (has_wanted) {
            HIPCHK(d_wanted.ensure(h_wanted.size() * 2 + 16)); HIPCHK(hipMemcpyAsync(d_wanted.p, h_wanted.data(), h_wanted.size() * 2, hipMemcpyHostToDevice, stream));
            for (size_t t = 0; t < h_wanted.size(); ++t) if (h_wanted[t] != (uint16_t)TILE_UNWANTED) h_tilelist.push_back((uint32_t)t);
            HIPCHK(d_tilelist.ensure(h_tilelist.size() * 4 + 16));
            if (!h_tilelist.empty()) HIPCHK(hipMemcpyAsync(d_tilelist.p, h_tilelist.data(), h_tilelist.size() * 4, hipMemcpyHostToDevice, stream));
        }
        if (n_indel_cap && ((uint64_t)c.P * (uint64_t)c.Lp >= 0xffffffffull || n_indel_cap >= 0xfffffff0ull)) { err = "region too large: (positions x libraries) and the indel operators must stay below 2^32"; return BRC_E_ARG; }
        const size_t nagg = std::max<size_t>(std::max<size_t>((std::max<size_t>(np, P * Lp) + SCAN_CHUNK - 1) / SCAN_CHUNK, (np + TR_CHUNK - 1) / TR_CHUNK), 1);
        HIPCHK(d_reads.ensure((n + 1) * sizeof(DRead)));
        HIPCHK(d_agg.ensure(nagg * 8 + 16)); HIPCHK(d_agg2.ensure(nagg * 4 + 16)); HIPCHK(d_rng.ensure(((size_t)ntiles * Lp + 1) * sizeof(uint2)));
        HIPCHK(d_ncol.ensure(Lp * P * 4 + 16)); HIPCHK(d_depth.ensure(Lp * P * 4 + 16)); HIPCHK(d_unavail.ensure(P * 4 + 16));
        HIPCHK(d_slotid.ensure(Lp * P * 4 + 16)); HIPCHK(d_si.ensure(Lp * 2 * NI * P * 4 + 16)); HIPCHK(d_sf.ensure(Lp * 2 * NF * P * 4 + 16));
        // third-allele lists: XEV_SHARDS sub-lists; about one piece in 25 leaves an event at 30-50x, capacity for twice that,
        // spread evenly (the grow-and-recompute path covers the rest)
        {
            const char* xc = test_knob(TK_XEV_CAP);           // (test knob: tiny lists exercise the grow-and-recompute path)
            const size_t want = xc ? (size_t)std::max(atoi(xc), 1) : std::max<size_t>(256, np / 8 / XEV_SHARDS);
            if (want > xev_cap) xev_cap = want;
        }
        HIPCHK(d_xev.ensure(((size_t)XEV_SHARDS * xev_cap + 1) * sizeof(XEv))); HIPCHK(d_xevc.ensure(((size_t)XEV_SHARDS * xev_cap + 1) * sizeof(XEv)));
        HIPCHK(d_xevn.ensure((size_t)XEV_SHARDS * XEV_CTR_STRIDE * 4));
        HIPCHK(d_xidx.ensure(((size_t)XEV_SHARDS * xev_cap + 1) * 4)); HIPCHK(d_xagg.ensure(((size_t)XEV_SHARDS * xev_cap + 1) * sizeof(XAgg)));
        HIPCHK(d_xcnt.ensure(((size_t)ntiles * Lp + 2) * 4)); HIPCHK(d_xend.ensure(((size_t)ntiles * Lp + 2) * 4));
        HIPCHK(d_part.ensure(4096 * 5 * sizeof(unsigned long long)));
        HIPCHK(d_ctr.ensure(sizeof(Counters))); HIPCHK(d_tilectr.ensure(((size_t)ntiles * Lp + 1) * sizeof(uint4)));
        in.iev_off = nullptr;
        if (n_indel_cap) {
            // indel side path: raw events (one slot per I / D / P operator, at host-computed per-read offsets), their counts
            // per (16 or 64 positions, library) bucket, the bucketed events and the reduced alleles
            const size_t nbk = (size_t)indel_buckets(c);
            if ((rc = up(d_ievoff, s.iev_off, n))) return rc;
            in.iev_off = (const uint32_t*)d_ievoff.p;
            HIPCHK(d_cnt.ensure(nbk * 4 + 16)); HIPCHK(d_cursor.ensure(nbk * 4 + 16)); HIPCHK(d_evraw.ensure((n_indel_cap + 1) * sizeof(IndelEv)));
            HIPCHK(d_ev.ensure((n_indel_cap + 1) * sizeof(IndelEv))); HIPCHK(d_iout.ensure((n_indel_cap + 1) * sizeof(IndelOut)));
        }
        if (has_wanted && P) {
            // brc_region_windows: what no window asks for comes back empty — written here, ONCE per uploaded region: the passes
            // over it only touch the announced tiles (their columns, their counters)
            HIPCHK(hipMemsetAsync(d_ncol.p, 0, Lp * P * 4, stream)); HIPCHK(hipMemsetAsync(d_depth.p, 0, Lp * P * 4, stream));
            HIPCHK(hipMemsetD32Async((hipDeviceptr_t)d_slotid.p, (int)((uint32_t)NB_NONE | ((uint32_t)NB_NONE << 8)), Lp * P, stream));
            HIPCHK(hipMemsetAsync(d_tilectr.p, 0, (size_t)ntiles * Lp * sizeof(uint4), stream));
        }
        HIPCHK(hipStreamSynchronize(stream));
        return BRC_OK;
    }

    template <class Op, bool INCL>
    int scan(const typename Op::T* src, typename Op::T* dst, int64_t n) { return scan_on<Op, INCL>(src, dst, n, stream, d_agg); }
    // (the indel side path scans on its own stream with its own block-aggregate scratch)
    template <class Op, bool INCL>
    int scan_on(const typename Op::T* src, typename Op::T* dst, int64_t n, hipStream_t st, DBuf& scratch) {
        if (n <= 0) return BRC_OK;
        const int64_t nb = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
        typename Op::T* agg = (typename Op::T*)scratch.p;
        hipLaunchKernelGGL((k_scan_reduce<Op>), dim3((unsigned)nb), dim3(SCAN_T), 0, st, src, n, agg);
        hipLaunchKernelGGL((k_scan_aggregates<Op>), dim3(1), dim3(1024), 0, st, agg, nb);
        hipLaunchKernelGGL((k_scan_apply<Op, INCL>), dim3((unsigned)nb), dim3(SCAN_T), 0, st, src, dst, n, (const typename Op::T*)agg);
        HIPCHK(hipGetLastError());
        return BRC_OK;
    }

    // One pass of the whole pipeline, queued on the engine's stream; its timing events go to event set `set`.
    int enqueue_pass(size_t set) {
        evt = evsets[set].evt; ev_indel = evsets[set].ev_indel;
        const int64_t n = c.n_reads, P = c.P; const int Lp = c.Lp;
        int rc;
        Counters* ctr = (Counters*)d_ctr.p;
        lists_host = false;
        HIPCHK(hipMemsetAsync(ctr, 0, sizeof(Counters), stream));
        HIPCHK(hipMemsetAsync(d_xevn.p, 0, (size_t)XEV_SHARDS * XEV_CTR_STRIDE * 4, stream));
        const bool indels = n_indel_cap > 0 && P > 0 && n > 0;
        const int64_t n_buckets = indel_buckets(c);     // indel buckets: (16 or 64 positions, library)
        if (indels) HIPCHK(hipMemsetAsync(d_cnt.p, 0, (size_t)n_buckets * 4, stream));
#ifdef BRC_CHECKED
        if (set == 0) { chk_fill(); HIPCHK(hipMemcpyAsync(d_chk.p, &h_chk, sizeof h_chk, hipMemcpyHostToDevice, stream)); HIPCHK(hipStreamSynchronize(stream)); }   // (faults of a ring of passes accumulate in one record)
#endif
        Planes pl = {(uint32_t*)d_ncol.p, (uint32_t*)d_depth.p, (uint32_t*)d_slotid.p, (uint32_t*)d_si.p, (float*)d_sf.p, (uint32_t*)d_unavail.p,
                     (XEv*)d_xev.p, (uint32_t*)d_xevn.p, (uint32_t)xev_cap, (uint32_t)XEV_SHARDS};
        pl_last = pl;
        const DRead* reads = (const DRead*)d_reads.p;
        HIPCHK(hipEventRecord(evt[T_ANNOTATE], stream));
        if (n > 0) {
            const int64_t rl = c.ref_hi - c.ref_lo;
            if (c.has_ref)
                hipLaunchKernelGGL(k_refcode, dim3((unsigned)(((rl + 2 * REFCODE_PAD + 15) / 16 + 255) / 256)), dim3(256), 0, stream, in.ref, (uint8_t*)d_refcode.p, rl);
            if (h_wpairs.n)        // the wide rows of the reads the host found an escape base in
                hipLaunchKernelGGL(k_wide_rows, dim3((unsigned)((h_wpairs.n + 255) / 256), wide_slices), dim3(256), 0, stream, c, in, (const uint2*)d_wpairs.p, (uint32_t)h_wpairs.n, (uint16_t*)in.bqw);
            DevIn in_k1 = in;
            if (wave_on || cursor_on) {   // reads with more than two M operators: listed for k_annotate_wave, without operators in K1's copy of the counts (reads with an empty M / = / X operator: for k_annotate_cursor)
                hipLaunchKernelGGL(k_pick_wave, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, c, in, (uint32_t*)d_nc_k1.p, (uint32_t*)d_wavelist.p, (uint32_t)n, &ctr->n_wave_reads, &ctr->n_wave_big, &ctr->n_wave_huge,
                                   wave_on ? 1 : 0, cursor_on ? 1 : 0, &ctr->n_literal, &ctr->n_wave_eqx, &ctr->n_wave_eqx_big);
                in_k1.n_cigar = (const uint32_t*)d_nc_k1.p;
            }
            {   // K1: one instantiation per (row layout, width of the narrow packed fields — choose_pack)
#define BRC_LAUNCH_K1(OS, SH) hipLaunchKernelGGL((k_annotate_groups<OS, SH>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, c, in_k1, (DRead*)d_reads.p, (const uint32_t*)d_pieceoff.p, \
                                   (Piece*)d_pieces.p, (PieceRare*)d_rare.p, (int2*)d_keyreach.p,                                                                               \
                                   (uint8_t*)in.eb, indels ? (IndelEv*)d_evraw.p : (IndelEv*)nullptr, (uint32_t*)d_cnt.p,                                    \
                                   in.cigar, in.qual, in.seq4, (const uint8_t*)d_refcode.p + REFCODE_PAD, has_wanted ? (const uint16_t*)d_wanted.p : (const uint16_t*)nullptr)
                if (Lp == 1) { if (c.pack_shift == 16) BRC_LAUNCH_K1(true, 16); else BRC_LAUNCH_K1(true, 12); }
                else { if (c.pack_shift == 16) BRC_LAUNCH_K1(false, 16); else BRC_LAUNCH_K1(false, 12); }
#undef BRC_LAUNCH_K1
                if (wave_on) {
                    // the reads K1 left to the wave form (their number stays on the device: a fixed grid takes them round robin)
                    const unsigned nb = (unsigned)std::min<int64_t>((n + 3) / 4, (int64_t)WAVE_FORM_BLOCKS);
#define BRC_LAUNCH_K1W(SH, MCAP, WAVES, GRID, LIST, STEP, COUNT) hipLaunchKernelGGL((k_annotate_wave<SH, MCAP, WAVES>), dim3(GRID), dim3(WAVES * 64), 0, stream, c, in, LIST, STEP, (const unsigned int*)(COUNT),     \
                                   (DRead*)d_reads.p, (const uint32_t*)d_pieceoff.p, (Piece*)d_pieces.p, (PieceRare*)d_rare.p, (int2*)d_keyreach.p,                                   \
                                   (uint8_t*)in.eb, indels ? (IndelEv*)d_evraw.p : (IndelEv*)nullptr, (uint32_t*)d_cnt.p,                                          \
                                   (const uint8_t*)d_refcode.p + REFCODE_PAD, has_wanted ? (const uint16_t*)d_wanted.p : (const uint16_t*)nullptr)
                    const uint32_t* const wl = (const uint32_t*)d_wavelist.p;
                    if (c.pack_shift == 16) BRC_LAUNCH_K1W(16, AW_MCAP, 4, nb, wl, 1, &ctr->n_wave_reads); else BRC_LAUNCH_K1W(12, AW_MCAP, 4, nb, wl, 1, &ctr->n_wave_reads);
                    if (wave_big) {     // reads with more than AW_MCAP M operators (more than 2 * AW_MCAP operators): one wave per workgroup, listed from the back
                        const unsigned nbb = (unsigned)std::min<int64_t>(n, (int64_t)WAVE_FORM_BLOCKS_BIG);
                        if (c.pack_shift == 16) BRC_LAUNCH_K1W(16, AW_MCAP_BIG, 1, nbb, wl + (n - 1), -1, &ctr->n_wave_big); else BRC_LAUNCH_K1W(12, AW_MCAP_BIG, 1, nbb, wl + (n - 1), -1, &ctr->n_wave_big);
                    }
                    if (wave_eqx) {     // reads with = / X operators: the instantiations that keep the annotator's own cursors; fourth list, the longer ones from its back
#define BRC_LAUNCH_K1E(SH, MCAP, WAVES, GRID, LIST, STEP, COUNT) hipLaunchKernelGGL((k_annotate_wave<SH, MCAP, WAVES, true>), dim3(GRID), dim3(WAVES * 64), 0, stream, c, in, LIST, STEP, (const unsigned int*)(COUNT),     \
                                   (DRead*)d_reads.p, (const uint32_t*)d_pieceoff.p, (Piece*)d_pieces.p, (PieceRare*)d_rare.p, (int2*)d_keyreach.p,                                   \
                                   (uint8_t*)in.eb, indels ? (IndelEv*)d_evraw.p : (IndelEv*)nullptr, (uint32_t*)d_cnt.p,                                          \
                                   (const uint8_t*)d_refcode.p + REFCODE_PAD, has_wanted ? (const uint16_t*)d_wanted.p : (const uint16_t*)nullptr)
                        if (c.pack_shift == 16) BRC_LAUNCH_K1E(16, AW_MCAP_EQX, 4, nb, wl + (3 * n + 48), 1, &ctr->n_wave_eqx); else BRC_LAUNCH_K1E(12, AW_MCAP_EQX, 4, nb, wl + (3 * n + 48), 1, &ctr->n_wave_eqx);
                        if (s_max_ncigar > (uint32_t)AW_MCAP_EQX) {
                            const unsigned nbe = (unsigned)std::min<int64_t>(n, (int64_t)WAVE_FORM_BLOCKS_BIG);
                            if (c.pack_shift == 16) BRC_LAUNCH_K1E(16, AW_MCAP_EQX_BIG, 1, nbe, wl + (4 * n + 47), -1, &ctr->n_wave_eqx_big); else BRC_LAUNCH_K1E(12, AW_MCAP_EQX_BIG, 1, nbe, wl + (4 * n + 47), -1, &ctr->n_wave_eqx_big);
                        }
#undef BRC_LAUNCH_K1E
                    }
                    if (wave_huge) {    // more than AW_MCAP_BIG M operators: one wave per CU (its list fills the CU's LDS), second list
                        const unsigned nbh = (unsigned)std::min<int64_t>(n, (int64_t)WAVE_FORM_BLOCKS_HUGE);
                        if (c.pack_shift == 16) BRC_LAUNCH_K1W(16, AW_MCAP_HUGE, 1, nbh, wl + (n + 16), 1, &ctr->n_wave_huge); else BRC_LAUNCH_K1W(12, AW_MCAP_HUGE, 1, nbh, wl + (n + 16), 1, &ctr->n_wave_huge);
                    }
#undef BRC_LAUNCH_K1W
                }
                if (cursor_on) {
                    const unsigned nbc = (unsigned)std::min<int64_t>((n + 63) / 64, 1024);
#define BRC_LAUNCH_K1C(SH) hipLaunchKernelGGL((k_annotate_cursor<SH>), dim3(nbc), dim3(64), 0, stream, c, in, (const uint32_t*)d_wavelist.p + (2 * (size_t)n + 32), (const unsigned int*)&ctr->n_literal,    \
                                   (DRead*)d_reads.p, (const uint32_t*)d_pieceoff.p, (Piece*)d_pieces.p, (PieceRare*)d_rare.p, (int2*)d_keyreach.p, (uint8_t*)in.eb,                     \
                                   indels ? (IndelEv*)d_evraw.p : (IndelEv*)nullptr, (uint32_t*)d_cnt.p, has_wanted ? (const uint16_t*)d_wanted.p : (const uint16_t*)nullptr)
                    if (c.pack_shift == 16) BRC_LAUNCH_K1C(16); else BRC_LAUNCH_K1C(12);
#undef BRC_LAUNCH_K1C
                }
            }
            if (c.per_lib) {
                HIPCHK(hipMemsetAsync(d_unavail.p, 0xff, (size_t)c.PS * 4, stream));
                hipLaunchKernelGGL(k_unavail, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, c, in, (uint32_t*)d_unavail.p);
            }
        } else if (c.per_lib && P > 0) HIPCHK(hipMemsetAsync(d_unavail.p, 0xff, (size_t)c.PS * 4, stream));
        HIPCHK(hipEventRecord(evt[T_SCAN_ENDS], stream));
        // The indel side path (<1 % of the events: keyed count -> scan -> fill -> ordered reduce) depends on K1 only.  It can run
        // on a stream of its own under the pileup kernel (BRC_INDEL_OVERLAP=1) — measured: the step does not get shorter, the
        // two only share the machine (k_pileup2 3.75 -> 3.95 ms, step 6.2 ms either way) — so by default it follows the
        // pileup on the main stream.
#ifdef BRC_EXP_KNOBS
        static const bool indel_overlap = getenv("BRC_INDEL_OVERLAP") && atoi(getenv("BRC_INDEL_OVERLAP")) != 0;
#else
        const bool indel_overlap = false;
#endif
        auto launch_indel = [&](hipStream_t si, DBuf& scratch) -> int {
            HIPCHK(hipEventRecord(ev_indel[0], si));
            int r2;
            if ((r2 = scan_on<OpSumU32, false>((const uint32_t*)d_cnt.p, (uint32_t*)d_cursor.p, n_buckets, si, scratch))) return r2;
            HIPCHK(hipEventRecord(ev_indel[1], si));
            hipLaunchKernelGGL(k_indel_scatter, dim3((unsigned)((n_indel_cap + 255) / 256)), dim3(256), 0, si, c, (const IndelEv*)d_evraw.p, (int64_t)n_indel_cap,
                               (uint32_t*)d_cursor.p, (IndelEv*)d_ev.p);
            HIPCHK(hipEventRecord(ev_indel[2], si));
            hipLaunchKernelGGL(k_indel_reduce, dim3((unsigned)std::min<int64_t>((n_buckets + 255) / 256, 4096)), dim3(256), 0, si, c, in, reads,
                               (const uint32_t*)d_cnt.p, (const uint32_t*)d_cursor.p, n_buckets, (IndelEv*)d_ev.p, (const uint32_t*)d_unavail.p,
                               (IndelOut*)d_iout.p, ctr);
            HIPCHK(hipEventRecord(ev_indel[3], si));
            return BRC_OK;
        };
        if (indels && indel_overlap) {
            HIPCHK(hipStreamWaitEvent(stream3, evt[T_SCAN_ENDS], 0));
            if ((rc = launch_indel(stream3, d_agg2))) return rc;
        }
        // the piece range of every (tile, library): running maximum of the piece reaches (block aggregates, their scan), then
        // one pass over the pieces of all libraries
        const int64_t np_all = c.n_pieces;
        const int64_t trb = (np_all + TR_CHUNK - 1) / TR_CHUNK;
        if (np_all > 0 && ntiles > 0) {
            hipLaunchKernelGGL(k_reach_blockmax, dim3((unsigned)trb), dim3(TR_T), 0, stream, (const int2*)d_keyreach.p, np_all, (const int64_t*)d_libbase.p, Lp, (unsigned long long*)d_agg.p);
            hipLaunchKernelGGL((k_scan_aggregates<OpMaxU64>), dim3(1), dim3(1024), 0, stream, (unsigned long long*)d_agg.p, trb);
        }
        HIPCHK(hipEventRecord(evt[T_TILES], stream));
        for (int l = 0; l < Lp && ntiles > 0; ++l)        // (a library without pieces: empty ranges)
            if (lib_base[(size_t)l + 1] == lib_base[(size_t)l]) HIPCHK(hipMemsetAsync((uint2*)d_rng.p + (int64_t)l * ntiles, 0, (size_t)ntiles * sizeof(uint2), stream));
        if (np_all > 0 && ntiles > 0)
            hipLaunchKernelGGL(k_tiles_all, dim3((unsigned)trb), dim3(TR_T), 0, stream, c, (const int2*)d_keyreach.p, np_all, (const int64_t*)d_libbase.p, Lp,
                               (const unsigned long long*)d_agg.p, ntiles, (uint2*)d_rng.p);
        const int64_t n_listed = (int64_t)h_tilelist.size();
        if (has_wanted && n_listed > 0)
            hipLaunchKernelGGL(k_narrow_tiles, dim3((unsigned)((n_listed * Lp + 255) / 256)), dim3(256), 0, stream, (const uint16_t*)d_wanted.p, (const uint32_t*)d_tilelist.p, n_listed, ntiles, Lp, c.pos0,
                               (uint2*)d_rng.p, (const int2*)d_keyreach.p);
        const uint4* kp_pieces = (const uint4*)d_pieces.p; const PieceRare* kp_rare = (const PieceRare*)d_rare.p; const uint2* kp_rng = (const uint2*)d_rng.p;
        if (compact_on && ntiles > 0 && np_all > 0) {
            // count the live pieces of every (tile, library), scan, (first pass of the region: size the compacted stream — one wait), copy
            const size_t nslot = (size_t)ntiles * (size_t)Lp;
            HIPCHK(d_ccnt.ensure((nslot + 2) * 4)); HIPCHK(d_coff.ensure((nslot + 2) * 4)); HIPCHK(d_crng.ensure((nslot + 1) * sizeof(uint2))); HIPCHK(d_ctot.ensure(24));
            HIPCHK(hipMemsetAsync(d_ctot.p, 0, 24, stream));
            HIPCHK(hipMemsetAsync((uint32_t*)d_ccnt.p + nslot, 0, 4, stream));
            const dim3 cg((unsigned)((ntiles + 3) / 4), (unsigned)Lp);
            // (one library: read-wise, a binary search per read instead of a walk over the whole range; TK_COMPACT=2 keeps the walk)
            const bool by_read = Lp == 1 && !(test_knob(TK_COMPACT) && atoi(test_knob(TK_COMPACT)) == 2);
            if (by_read) {
                HIPCHK(hipMemsetAsync(d_ccnt.p, 0, nslot * 4, stream));
                hipLaunchKernelGGL(k_count_piece_tiles, dim3((unsigned)((np_all + 255) / 256)), dim3(256), 0, stream, c, (const uint4*)d_pieces.p, np_all, (uint32_t*)d_ccnt.p);
            }
            else
            hipLaunchKernelGGL((k_compact_tiles<true>), cg, dim3(256), 0, stream, c, (const uint4*)d_pieces.p, (const PieceRare*)d_rare.p, (const uint2*)d_rng.p, ntiles,
                               (uint32_t*)d_ccnt.p, (const uint32_t*)nullptr, (uint4*)nullptr, (PieceRare*)nullptr, (uint2*)nullptr, (unsigned long long*)d_ctot.p);
            if ((rc = scan<OpSumU32, false>((const uint32_t*)d_ccnt.p, (uint32_t*)d_coff.p, (int64_t)nslot + 1))) return rc;
            const bool first_pass = !compact_sized;
            if (!compact_sized) {
                uint32_t tot = 0;
                HIPCHK(hipMemcpyAsync(&tot, (const uint32_t*)d_coff.p + nslot, 4, hipMemcpyDeviceToHost, stream));
                HIPCHK(hipMemcpyAsync(h_steps, d_ctot.p, 16, hipMemcpyDeviceToHost, stream));
                HIPCHK(hipStreamSynchronize(stream));
                compact_total = tot; compact_sized = true;
                HIPCHK(d_cpieces.ensure(((size_t)compact_total + 4) * sizeof(Piece))); HIPCHK(d_crare.ensure(((size_t)compact_total + 2) * sizeof(PieceRare)));
            }
            if (by_read)
                hipLaunchKernelGGL(k_compact_reads, dim3((unsigned)((ntiles + 4 * CR_GROUP - 1) / (4 * CR_GROUP))), dim3(256), 0, stream, c, (const uint4*)d_pieces.p, (const PieceRare*)d_rare.p, (const uint2*)d_rng.p, (const uint32_t*)d_pieceoff.p, (const int2*)d_keyreach.p, ntiles,
                                   (const uint32_t*)d_coff.p, (uint4*)d_cpieces.p, (PieceRare*)d_crare.p, (uint2*)d_crng.p, (unsigned long long*)d_ctot.p);
            else
            hipLaunchKernelGGL((k_compact_tiles<false>), cg, dim3(256), 0, stream, c, (const uint4*)d_pieces.p, (const PieceRare*)d_rare.p, (const uint2*)d_rng.p, ntiles,
                               (uint32_t*)d_ccnt.p, (const uint32_t*)d_coff.p, (uint4*)d_cpieces.p, (PieceRare*)d_crare.p, (uint2*)d_crng.p, (unsigned long long*)d_ctot.p);
            if (by_read && first_pass) {   // (the read-wise copy adds up the piece-steps itself: brc_region_piece_steps reads them after the region's first pass)
                HIPCHK(hipMemcpyAsync(h_steps, d_ctot.p, 24, hipMemcpyDeviceToHost, stream));
                HIPCHK(hipStreamSynchronize(stream));
                if (h_steps[2]) { err = "tile compaction: a tile holds more live read segments than the pass over the segments counted for it"; return BRC_E_HIP; }
            }
            kp_pieces = (const uint4*)d_cpieces.p; kp_rare = (const PieceRare*)d_crare.p; kp_rng = (const uint2*)d_crng.p;
#ifdef BRC_CHECKED
            {   // (the compacted stream exists now: its extents replace K1's for the pileup's sites; the fault count so far is kept)
                HIPCHK(hipStreamSynchronize(stream));
                ChkState seen; HIPCHK(hipMemcpy(&seen, d_chk.p, sizeof seen, hipMemcpyDeviceToHost));
                chk_fill(); ChkState upd = h_chk; upd.count = seen.count; upd.kernel = seen.kernel; upd.site = seen.site; upd.buf = seen.buf; upd.addr = seen.addr; upd.bytes = seen.bytes; upd.unit = seen.unit; upd.piece = seen.piece;
                HIPCHK(hipMemcpy(d_chk.p, &upd, sizeof upd, hipMemcpyHostToDevice));
            }
#endif
        }
        HIPCHK(hipEventRecord(evt[T_PILEUP], stream));
        if (ntiles > 0) {
            unsigned nwg = (unsigned)(((has_wanted ? n_listed : ntiles) + PILEUP_WAVES - 1) / PILEUP_WAVES);
            nwg = (nwg + 7u) & ~7u;
            // profiling knob: unused dynamic LDS lowers the number of resident waves (occupancy sweeps)
#ifdef BRC_EXP_KNOBS
            static const unsigned dyn_lds = getenv("BRC_PILEUP_LDS_PAD") ? (unsigned)atoi(getenv("BRC_PILEUP_LDS_PAD")) : 0u;
#else
            const unsigned dyn_lds = 0u;
#endif
#define BRC_LAUNCH_KP(W, SH) hipLaunchKernelGGL((k_pileup2<W, SH>), dim3(nwg, (unsigned)Lp), dim3(PILEUP_WAVES * 64), dyn_lds, stream, c, in, kp_pieces, kp_rare, \
                               kp_rng, ntiles, pl, (uint4*)d_tilectr.p, in.eb, in.bqw, (const uint32_t*)d_unavail.p,                                                     \
                               (const uint8_t*)d_refcode.p + REFCODE_PAD, (const uint16_t*)d_wanted.p, (const uint32_t*)d_tilelist.p, n_listed)
            if (has_wanted) { if (n_listed > 0) { if (c.pack_shift == 16) BRC_LAUNCH_KP(true, 16); else BRC_LAUNCH_KP(true, 12); } }
            else { if (c.pack_shift == 16) BRC_LAUNCH_KP(false, 16); else BRC_LAUNCH_KP(false, 12); }
#undef BRC_LAUNCH_KP
        }
        HIPCHK(hipEventRecord(evt[T_COUNT], stream));      // (the k_pileup slot is k_pileup2 alone: what rocprofv3's average for the kernel must agree with)
        if (ntiles > 0) {
            // third-allele events: one list, then their fold per (position, library, bucket) in column order — the rest of
            // BasicStat::process_read's accumulation (the host did this until round 6); timed with the counters below
            const int64_t nxb = ntiles * Lp;
            HIPCHK(hipMemsetAsync(d_xcnt.p, 0, (size_t)nxb * 4, stream));
            hipLaunchKernelGGL(k_xev_compact, dim3((unsigned)XEV_SHARDS), dim3(256), 0, stream, (const XEv*)d_xev.p, (const uint32_t*)d_xevn.p, (uint32_t)xev_cap,
                               (uint32_t)XEV_SHARDS, (XEv*)d_xevc.p, ctr, (uint32_t*)d_xcnt.p, Lp);
            if ((rc = scan<OpSumU32, false>((const uint32_t*)d_xcnt.p, (uint32_t*)d_xend.p, nxb))) return rc;
            hipLaunchKernelGGL(k_xev_scatter, dim3(256), dim3(256), 0, stream, (const XEv*)d_xevc.p, (const Counters*)ctr, (uint32_t*)d_xend.p, (uint32_t*)d_xidx.p, Lp);
            hipLaunchKernelGGL(k_xev_fold, dim3((unsigned)std::min<int64_t>((nxb + 255) / 256, 2048)), dim3(256), 0, stream, (const XEv*)d_xevc.p, (uint32_t*)d_xcnt.p, (uint32_t*)d_xend.p, nxb,
                               (uint32_t*)d_xidx.p, (XAgg*)d_xagg.p);
        }
        if (P > 0) {
            const unsigned nb = (unsigned)std::min<int64_t>(((Lp > 1 ? P / 4 : ntiles) + 255) / 256 + 1, Lp > 1 ? 4096 : 256);
            hipLaunchKernelGGL(k_finalize, dim3(nb), dim3(256), 0, stream, c, (const uint32_t*)d_ncol.p, (const uint4*)d_tilectr.p, (int64_t)ntiles * Lp,
                               (unsigned long long*)d_part.p);
            hipLaunchKernelGGL(k_finalize_sum, dim3(1), dim3(256), 0, stream, (const unsigned long long*)d_part.p, (int)nb, ctr);
        }
        HIPCHK(hipEventRecord(evt[T_INDEL_SCAN], stream));
        // (the timing slots of the indel path's three stages are filled from its own events)
        if (indels && !indel_overlap) { if ((rc = launch_indel(stream, d_agg))) return rc; }
        if (indels && indel_overlap) HIPCHK(hipStreamWaitEvent(stream, ev_indel[3], 0));
        HIPCHK(hipEventRecord(evt[T_N], stream));
        HIPCHK(hipGetLastError());
        return BRC_OK;
    }
    // the counters of the last queued pass -> host, wait; 1 = a third-allele sub-list was too short (lists grown: compute again)
    int finish_passes(bool* again) {
        HIPCHK(hipMemcpyAsync(&h_ctr, d_ctr.p, sizeof(Counters), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        *again = false;
#ifdef BRC_CHECKED
        { const int rcc = chk_report(); if (rcc) return rcc; }
#endif
        if ((size_t)h_ctr.xev_max > xev_cap) {
            xev_cap = (size_t)h_ctr.xev_max * 2;
            HIPCHK(d_xev.ensure(((size_t)XEV_SHARDS * xev_cap + 1) * sizeof(XEv))); HIPCHK(d_xevc.ensure(((size_t)XEV_SHARDS * xev_cap + 1) * sizeof(XEv)));
            HIPCHK(d_xidx.ensure(((size_t)XEV_SHARDS * xev_cap + 1) * 4)); HIPCHK(d_xagg.ensure(((size_t)XEV_SHARDS * xev_cap + 1) * sizeof(XAgg)));
            *again = true;
        }
        return BRC_OK;
    }
    // adds the per-kernel times of the pass recorded in event set `set` to t (slots and total)
    int add_timing(size_t set, brc_timing* t) {
        const bool indels = n_indel_cap > 0 && c.P > 0 && c.n_reads > 0;
        const EvSet& es = evsets[set]; float ms = 0;
        for (int i = 0; i < T_INDEL_SCAN; ++i) { HIPCHK(hipEventElapsedTime(&ms, es.evt[i], es.evt[i + 1])); t->ms[i] += ms; }
        if (indels) for (int i = 0; i < 3; ++i) { HIPCHK(hipEventElapsedTime(&ms, es.ev_indel[i], es.ev_indel[i + 1])); t->ms[T_INDEL_SCAN + i] += ms; }
        HIPCHK(hipEventElapsedTime(&ms, es.evt[0], es.evt[T_N])); t->total_ms += ms;
        return BRC_OK;
    }
    int compute(brc_timing* t) override {
        HIPCHK(hipSetDevice(device));
        for (;;) {
            int rc = enqueue_pass(0); if (rc) return rc;
            bool again = false;
            if ((rc = finish_passes(&again))) return rc;
            if (!again) break;                           // (a third-allele sub-list was too short: the lists were grown, compute again)
        }
        if (t) { memset(t, 0, sizeof *t); const int rc = add_timing(0, t); if (rc) return rc; }
        computed = true;
        return BRC_OK;
    }
    // n passes over the uploaded region queued back to back, ONE wait per ring of event sets: between two passes the device
    // never waits for the host (submission of pass k + 1 runs under the kernels of pass k).  t: per-kernel times averaged over
    // the passes; total_ms: the passes' own first-launch-to-last-completion spans, averaged.
    int compute_n(int n, brc_timing* t) override {
        HIPCHK(hipSetDevice(device));
        if (n <= 0) return BRC_OK;
        int rc = ensure_evsets((size_t)std::min<int>(n, EV_RING)); if (rc) return rc;
        brc_timing acc; memset(&acc, 0, sizeof acc);
        int done = 0;
        while (done < n) {
            const int k = std::min<int>(n - done, EV_RING);
            for (int i = 0; i < k; ++i) if ((rc = enqueue_pass((size_t)i))) return rc;
            bool again = false;
            if ((rc = finish_passes(&again))) return rc;
            if (again) continue;                         // (lists grown: this ring's passes are repeated)
            if (t) for (int i = 0; i < k; ++i) if ((rc = add_timing((size_t)i, &acc))) return rc;
            done += k;
        }
        if (t) { for (int i = 0; i < BRC_NKERNEL; ++i) t->ms[i] = acc.ms[i] / (float)n; t->total_ms = acc.total_ms / (float)n; }
        computed = true;
        return BRC_OK;
    }

    int text_begin(const std::string& chrom, const std::vector<std::string>& libs, int* slot_out) override {
        HIPCHK(hipSetDevice(device));
        if (!computed) { err = "not computed"; return BRC_E_ARG; }
        const int64_t P = c.P;
        const int slot = (text_slot ^= 1); *slot_out = slot;
        text_started[slot] = true; text_total[slot] = 0; text_n[slot] = P;
        { const int rc0 = enqueue_lists(); if (rc0) return rc0; }
        if (!h_toff[slot].reserve((size_t)P + 4) || !h_total.reserve(8) || !h_lastproc[slot].reserve((size_t)c.Lp + 4)) { err = "pinned host allocation failed"; return BRC_E_NOMEM; }
        for (int l = 0; l < c.Lp; ++l) h_lastproc[slot].p[l] = NONE32;
        if (P == 0) { h_toff[slot].p[0] = 0; HIPCHK(hipEventRecord(ev_text[slot], stream)); return BRC_OK; }
        // column 1 and the library names, behind the offsets of the names
        std::vector<int32_t> loff((size_t)c.Lp + 1, 0); std::string bytes = chrom;
        if (c.per_lib) for (int l = 0; l < c.Lp; ++l) { loff[(size_t)l] = (int32_t)bytes.size(); if ((size_t)l < libs.size()) bytes += libs[(size_t)l]; loff[(size_t)l + 1] = (int32_t)bytes.size(); }
        const size_t ob = loff.size() * 4;
        std::vector<char> ctx(ob + bytes.size() + 1);
        memcpy(ctx.data(), loff.data(), ob); memcpy(ctx.data() + ob, bytes.data(), bytes.size());
        HIPCHK(d_tctx.ensure(ctx.size() + 16)); HIPCHK(d_tlen.ensure(((size_t)P + 8) * 4)); HIPCHK(d_toff.ensure(((size_t)P + 8) * 4));
        // the download before this one read d_toff / d_text on the copy stream: it is long done, but say so to the device
        HIPCHK(hipStreamWaitEvent(stream, ev_text[slot ^ 1], 0));
        HIPCHK(hipMemcpyAsync(d_tctx.p, ctx.data(), ctx.size(), hipMemcpyHostToDevice, stream));
        TextCtx t; t.lib_off = (const int32_t*)d_tctx.p; t.chrom = (const char*)d_tctx.p + ob; t.chrom_len = (int32_t)chrom.size(); t.lib_names = t.chrom;
        const unsigned nb = (unsigned)((P + 1 + 255) / 256);
        const bool indels = n_indel_cap > 0 && c.P > 0 && c.n_reads > 0;
        TextAux ax; memset(&ax, 0, sizeof ax);
        ax.xagg = (const XAgg*)d_xagg.p; ax.xagg_end = (const uint32_t*)d_xend.p; ax.xagg_cnt = (const uint32_t*)d_xcnt.p;
        if (indels) { ax.iout = (const IndelOut*)d_iout.p; ax.ib_end = (const uint32_t*)d_cursor.p; ax.ib_cnt = (const uint32_t*)d_cnt.p; ax.reads = (const DRead*)d_reads.p; }
        hipLaunchKernelGGL(k_text_len, dim3(nb), dim3(256), 0, stream, c, in, pl_last, t, ax, (uint32_t*)d_tlen.p);
        int rc;
        if ((rc = scan<OpSumU32, false>((const uint32_t*)d_tlen.p, (uint32_t*)d_toff.p, P + 1))) return rc;
        // the 32-bit total, and the true one: the caller's estimate of the text size (brc_host.cpp) does not know the lengths of the
        // library names or of sums at the far end of int32 — a region whose lines pass 4 GiB is handed back for the host formatter
        HIPCHK(d_total64.ensure(16)); HIPCHK(hipMemsetAsync(d_total64.p, 0, 8, stream));
        hipLaunchKernelGGL(k_text_total, dim3((unsigned)std::min<int64_t>((P + 255) / 256, 1024)), dim3(256), 0, stream, (const uint32_t*)d_tlen.p, P, (unsigned long long*)d_total64.p);
        HIPCHK(hipMemcpyAsync(h_total.p, (const uint32_t*)d_toff.p + P, 4, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemcpyAsync(h_total.p + 2, d_total64.p, 8, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));                                  // (also covers the local `ctx`)
        const uint64_t total = h_total.p[0];
        { uint64_t t64; memcpy(&t64, h_total.p + 2, 8);
          // (TK_DEVICE_TEXT_LIMIT: test knob, a lower limit — the route changes, the text does not)
          const uint64_t limit = test_knob(TK_DEVICE_TEXT_LIMIT) ? strtoull(test_knob(TK_DEVICE_TEXT_LIMIT), nullptr, 10) : ~0ull;
          if (t64 != total || t64 > limit) { text_started[slot] = false; text_slot ^= 1; return BRC_TEXT_TOO_LONG; } }
        text_total[slot] = total;
        std::lock_guard<std::mutex> lk(text_mu);
        HIPCHK(d_text.ensure((size_t)total + 64));
        if (!h_text[slot].reserve((size_t)total + 64)) { err = "pinned host allocation failed"; return BRC_E_NOMEM; }
        hipLaunchKernelGGL(k_text_write, dim3(nb), dim3(256), 0, stream, c, in, pl_last, t, ax, (const uint32_t*)d_toff.p, (char*)d_text.p);
        // the last position every library was processed at: the host's account of what the region leaves in the deletion queues
        HIPCHK(d_lastproc.ensure((size_t)c.Lp * 4 + 16));
        hipLaunchKernelGGL(k_last_processed, dim3((unsigned)c.Lp), dim3(256), 0, stream, c, (const uint32_t*)d_ncol.p, (const uint32_t*)d_unavail.p, (uint32_t*)d_lastproc.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(ev_lines, stream));
        // the copies run on their own stream: uploads and kernels of the next region do not queue behind 300 MB of text
        HIPCHK(hipStreamWaitEvent(stream2, ev_lines, 0));
        HIPCHK(hipMemcpyAsync(h_toff[slot].p, d_toff.p, ((size_t)P + 1) * 4, hipMemcpyDeviceToHost, stream2));
        HIPCHK(hipMemcpyAsync(h_lastproc[slot].p, d_lastproc.p, (size_t)c.Lp * 4, hipMemcpyDeviceToHost, stream2));
        if (total) HIPCHK(hipMemcpyAsync(h_text[slot].p, d_text.p, (size_t)total, hipMemcpyDeviceToHost, stream2));
        HIPCHK(hipEventRecord(ev_text[slot], stream2));
        // ... but the next region's kernels overwrite d_toff / d_text / the planes only after the copies (d_text is rewritten
        // by the next text_begin, which waits above; d_tlen / d_toff likewise)
        return BRC_OK;
    }
    int text_wait(int slot, HostText* out) override {
        HIPCHK(hipSetDevice(device));
        slot &= 1;
        if (!text_started[slot]) { err = "no device text was started"; return BRC_E_ARG; }
        HIPCHK(hipEventSynchronize(ev_text[slot]));
        out->text = h_text[slot].p; out->off = h_toff[slot].p; out->total = text_total[slot]; out->n = text_n[slot]; out->last_processed = h_lastproc[slot].p;
        return BRC_OK;
    }

    void piece_steps(uint64_t* ranged, uint64_t* walked) override { *ranged = compact_on ? h_steps[0] : 0; *walked = compact_on ? h_steps[1] : 0; }
    int counts(uint64_t* e, uint64_t* p) override {
        if (!computed) { err = "not computed"; return BRC_E_ARG; }
        if (e) *e = h_ctr.n_events;
        if (p) *p = h_ctr.n_positions;
        return BRC_OK;
    }

    // brc_fetch_window: the compact planes of plane indices [k0, k0 + n) -> pinned window buffers (strided copies: a plane
    // row of n elements out of every PS), the region's two lists whole (downloaded once per computed region)
    HBuf<uint32_t> w_ncol, w_depth, w_slotid, w_si, w_unavail; HBuf<float> w_sf; bool w_init = false;
    bool lists_host = false;             // h_xev / h_iout / iout_compact hold the lists of the last compute
    int fetch_window(int64_t k0, int64_t n, HostPlanes* out, int64_t* stride) override {
        HIPCHK(hipSetDevice(device));
        if (!computed) { err = "not computed"; return BRC_E_ARG; }
        if (k0 < 0 || n < 0 || k0 + n > c.P) { err = "window outside the planes"; return BRC_E_ARG; }
        if (!w_init) { w_ncol.A = w_depth.A = w_slotid.A = w_si.A = w_unavail.A = &kPinned; w_sf.A = &kPinned; w_init = true; }
        const size_t WS = (size_t)((n + 63) & ~(int64_t)63), Lp = (size_t)c.Lp, PS = (size_t)c.PS;
        if (!w_ncol.reserve(Lp * WS + 4) || !w_depth.reserve(Lp * WS + 4) || !w_slotid.reserve(Lp * WS + 4) || !w_unavail.reserve(WS + 4) ||
            !w_si.reserve(Lp * 2 * NI * WS + 4) || !w_sf.reserve(Lp * 2 * NF * WS + 4)) { err = "pinned host allocation failed"; return BRC_E_NOMEM; }
        if (n) {
            const size_t wb = (size_t)n * 4;
            HIPCHK(hipMemcpy2DAsync(w_ncol.p, WS * 4, (const uint32_t*)d_ncol.p + k0, PS * 4, wb, Lp, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpy2DAsync(w_depth.p, WS * 4, (const uint32_t*)d_depth.p + k0, PS * 4, wb, Lp, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpy2DAsync(w_slotid.p, WS * 4, (const uint32_t*)d_slotid.p + k0, PS * 4, wb, Lp, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpy2DAsync(w_si.p, WS * 4, (const uint32_t*)d_si.p + k0, PS * 4, wb, Lp * 2 * NI, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpy2DAsync(w_sf.p, WS * 4, (const float*)d_sf.p + k0, PS * 4, wb, Lp * 2 * NF, hipMemcpyDeviceToHost, stream));
            if (c.per_lib) HIPCHK(hipMemcpyAsync(w_unavail.p, (const uint32_t*)d_unavail.p + k0, wb, hipMemcpyDeviceToHost, stream));
        }
        if (!lists_host) {
            const int rc = enqueue_lists(); if (rc) return rc;
            lists_enqueued = false;
            HIPCHK(hipStreamSynchronize(stream));
            iout_compact.clear();
            for (size_t i = 0; i < (size_t)h_ctr.n_indel_slots; ++i) if (h_iout.p[i].len != 0) iout_compact.push_back(h_iout.p[i]);
            xagg_compact.clear();
            for (size_t i = 0; i < (size_t)h_ctr.n_xev; ++i) if (h_xagg.p[i].k != NONE32) xagg_compact.push_back(h_xagg.p[i]);
            lists_host = true;
        } else HIPCHK(hipStreamSynchronize(stream));
        *out = HostPlanes();
        out->ncol = w_ncol.p; out->depth = w_depth.p; out->slotid = w_slotid.p; out->si = w_si.p; out->sf = w_sf.p; out->unavail = w_unavail.p;
        out->xagg = xagg_compact.data(); out->n_xagg = xagg_compact.size();
        out->indel = iout_compact.data(); out->n_indel = (int64_t)iout_compact.size();
        out->n_events = h_ctr.n_events; out->n_positions = h_ctr.n_positions;
        *stride = (int64_t)WS;
        return BRC_OK;
    }
    // the two small lists (third-allele events, indel buckets) -> pinned host memory.  With device-side text they are
    // requested BEFORE the text: the DMA engine serves copies in order, and the caller needs the lists first.
    bool lists_enqueued = false;
    int enqueue_lists() {
        const size_t nx = h_ctr.n_xev, ns = h_ctr.n_indel_slots;
        if (!h_xagg.reserve(nx + 4) || !h_iout.reserve(ns + 4)) { err = "pinned host allocation failed"; return BRC_E_NOMEM; }
        if (nx) HIPCHK(hipMemcpyAsync(h_xagg.p, d_xagg.p, nx * sizeof(XAgg), hipMemcpyDeviceToHost, stream));        // (a record per event at the most; unused slots are marked)
        if (ns) HIPCHK(hipMemcpyAsync(h_iout.p, d_iout.p, ns * sizeof(IndelOut), hipMemcpyDeviceToHost, stream));
        lists_enqueued = true;
        return BRC_OK;
    }
    // pinned / device room for the text of regions of about `bytes` (called early, from any thread, before the first region)
    void list_sizes(uint64_t* n_xev, uint64_t* n_indel_slots) override { *n_xev = h_ctr.n_xev; *n_indel_slots = h_ctr.n_indel_slots; }
    std::mutex text_mu;                  // reserve_text may run on another thread while the first region is staged
    int reserve_text(size_t bytes) override {
        std::lock_guard<std::mutex> lk(text_mu);
        HIPCHK(hipSetDevice(device));
        for (int i = 0; i < 2; ++i) if (!h_text[i].reserve(bytes)) { err = "pinned host allocation failed"; return BRC_E_NOMEM; }
        HIPCHK(d_text.ensure(bytes));
        return BRC_OK;
    }
    int fetch(HostPlanes* out, bool planes) override {
        HIPCHK(hipSetDevice(device));
        if (!computed) { err = "not computed"; return BRC_E_ARG; }
        const size_t P = planes ? (size_t)c.PS : 0, Lp = (size_t)c.Lp;   // planes are copied with their padded stride
        const size_t nx = h_ctr.n_xev;
        if (!h_ncol.reserve(Lp * P + 4) || !h_depth.reserve(Lp * P + 4) || !h_slotid.reserve(Lp * P + 4) || !h_unavail.reserve(P + 4) ||
            !h_si.reserve(Lp * 2 * NI * P + 4) || !h_sf.reserve(Lp * 2 * NF * P + 4)) { err = "pinned host allocation failed"; return BRC_E_NOMEM; }
        if (P) {
            HIPCHK(hipMemcpyAsync(h_ncol.p, d_ncol.p, Lp * P * 4, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpyAsync(h_depth.p, d_depth.p, Lp * P * 4, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpyAsync(h_slotid.p, d_slotid.p, Lp * P * 4, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpyAsync(h_si.p, d_si.p, Lp * 2 * NI * P * 4, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpyAsync(h_sf.p, d_sf.p, Lp * 2 * NF * P * 4, hipMemcpyDeviceToHost, stream));
            if (c.per_lib) HIPCHK(hipMemcpyAsync(h_unavail.p, d_unavail.p, P * 4, hipMemcpyDeviceToHost, stream));
        }
        const size_t ns = h_ctr.n_indel_slots;
        if (!lists_enqueued) { const int rc = enqueue_lists(); if (rc) return rc; }
        lists_enqueued = false;
        HIPCHK(hipStreamSynchronize(stream));
        iout_compact.clear();
        for (size_t i = 0; i < ns; ++i) if (h_iout.p[i].len != 0) iout_compact.push_back(h_iout.p[i]);
        xagg_compact.clear();
        for (size_t i = 0; i < nx; ++i) if (h_xagg.p[i].k != NONE32) xagg_compact.push_back(h_xagg.p[i]);
        lists_host = true;
        out->ncol = h_ncol.p; out->depth = h_depth.p; out->slotid = h_slotid.p; out->si = h_si.p; out->sf = h_sf.p; out->unavail = h_unavail.p;
        out->xagg = xagg_compact.data(); out->n_xagg = xagg_compact.size();
        out->indel = iout_compact.data(); out->n_indel = (int64_t)iout_compact.size();
        out->n_events = h_ctr.n_events; out->n_positions = h_ctr.n_positions;
        out->warn[BRC_W_SM_MISSING] = h_ctr.w_sm; out->warn[BRC_W_NM_MISSING] = h_ctr.w_nm; out->warn[BRC_W_ZM_MISSING] = 0;
        out->warn[BRC_W_LIB_UNAVAILABLE] = h_ctr.w_lib;
        return BRC_OK;
    }
};

// brc_host_alloc: page-locked memory every device context of the process can copy from (the engines of one process may sit on several GPUs)
void* backend_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) return nullptr;
    return p;
}
void backend_host_free(void* p) { (void)hipHostFree(p); }

Backend* make_backend(const brc_config& cfg, int* errc) {
    // Engines are created one at a time, process-wide: the first calls into the HIP runtime (device initialisation, code-object
    // load at the first launch) need not meet each other on several threads.  The later, per-engine work (streams, uploads,
    // launches) runs concurrently.  (Callers: the HIP runtime reads the environment while it starts — never setenv() in a
    // process that is creating an engine; see BRC_OPT_FORMAT_THREADS.)
    static std::mutex create_mu;
    std::lock_guard<std::mutex> create_lock(create_mu);
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
#ifdef BRC_CHECKED
const char* backend_kind() { return "hip-gfx950-checked"; }
#else
const char* backend_kind() { return "hip-gfx950"; }
#endif
const char* backend_kernel_name(int k) { return (k >= 0 && k < T_N) ? kKernelNames[k] : nullptr; }

}  // namespace brc
