// brc_sim.cpp — CPU lane-by-lane simulator of the device pipeline.  TEST INFRASTRUCTURE ONLY.
//
// Runs the __host__ __device__ functions of bam_readcount_amd/csrc/brc_core.h — the same code the HIP kernels
// wrap — one lane at a time, in the kernels' launch structure (K1 annotate, prefix-max, tile ranges, KB pileup
// tiles of 64 lanes, indel count/scan/fill/reduce).  It lets `-m "not gpu"` tests check the device algorithm
// against the oracle without a GPU.  It is never linked into the product library and the product never loads it.
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../bam_readcount_amd/csrc/brc_host.h"

namespace brc {

static void* sim_alloc(size_t n) { return malloc(n ? n : 1); }
static void sim_release(void* p) { free(p); }
static const HostAlloc kAlloc = {sim_alloc, sim_release};

class SimBackend : public Backend {
    DevCfg c; DevIn in; std::string err;
    std::vector<DRead> reads; std::vector<int32_t> prefmax; std::vector<uint16_t> bq; size_t bq_n = 0; std::vector<RcpPair> rcp; std::vector<float> tq; std::vector<double> te;
    std::vector<uint32_t> ncol, depth, istat, unavail; std::vector<float> fstat;
    std::vector<IndelOut> iout;
    uint64_t n_events = 0, n_positions = 0, warn[BRC_N_WARN] = {0, 0, 0, 0};

  public:
    const HostAlloc* host_alloc() override { return &kAlloc; }
    const char* last_error() const override { return err.c_str(); }
    int upload(const brc_config& cfg, const Staged& s, Geometry& g) override {
        g.PS = g.P + 3;   // deliberately odd stride: exercises the stride handling of the host side
        memset(&c, 0, sizeof c);
        c.min_mapq = cfg.min_mapq; c.min_bq = cfg.min_bq; c.per_lib = cfg.per_lib; c.insertion_centric = cfg.insertion_centric;
        c.Lp = g.Lp; c.ref_len_check = cfg.ref_len_check; c.has_ref = g.ref != nullptr;
        c.beg0 = g.beg0; c.end = g.end; c.pos0 = g.pos0; c.P = g.P; c.PS = g.PS; c.ref_lo = g.ref_lo; c.ref_hi = g.ref_hi; c.ref_len = g.ref_len;
        c.n_reads = s.n; c.table_len = s.modal_len();
        tq.assign((size_t)TABLE_MAX + 2, 0.0f); te.assign((size_t)TABLE_MAX + 2, 0.0);
        for (int k = 0; k <= c.table_len; ++k) { tq[(size_t)k] = (float)k / (float)c.table_len; te[(size_t)k] = 1.0 - (double)tq[(size_t)k]; }
        in.pos = s.pos.p; in.flag = s.flag.p; in.mapq = s.mapq.p; in.lib = s.lib.p; in.l_qseq = s.l_qseq.p; in.n_cigar = s.n_cigar.p;
        in.cig_off = s.cig_off.p; in.seq_off = s.seq_off.p; in.qual_off = s.qual_off.p; in.nm = s.nm.p; in.sm = s.sm.p; in.tags = s.tags.p;
        in.cigar = s.cigar.p; in.seq4 = s.seq4.p; in.qual = s.qual.p; in.ref = g.ref ? g.ref + g.ref_lo : nullptr;
        bq_n = s.bq_elems; in.bq = nullptr; in.bq_row = s.bq_row.p;
        return BRC_OK;
    }
    int compute(brc_timing* t) override {
        if (t) memset(t, 0, sizeof *t);
        const int64_t n = c.n_reads, P = c.P, PS = c.PS; const int Lp = c.Lp;
        reads.resize((size_t)n); prefmax.resize((size_t)n); bq.assign(bq_n + 1, 0); in.bq = bq.data(); rcp.resize((size_t)n + 1); in.rcp = rcp.data();
        for (int64_t i = 0; i < n; ++i) reads[(size_t)i] = annotate_read(c, in, i, bq.data(), rcp.data());   // K1
        int32_t m = INT32_MIN;
        for (int64_t i = 0; i < n; ++i) { if (reads[(size_t)i].end > m) m = reads[(size_t)i].end; prefmax[(size_t)i] = m; }
        ncol.assign((size_t)(Lp * PS), 0); depth.assign((size_t)(Lp * PS), 0); unavail.assign((size_t)PS, NONE32);
        istat.assign((size_t)(Lp * NBUCKET * NI * PS), 0); fstat.assign((size_t)(Lp * NBUCKET * NF * PS), 0.0f);
        Planes pl = {ncol.data(), depth.data(), istat.data(), fstat.data(), unavail.data()};
        n_events = n_positions = 0; memset(warn, 0, sizeof warn);
        const int64_t ntiles = (P + TILE - 1) / TILE;
        for (int l = 0; l < Lp; ++l) for (int64_t tl = 0; tl < ntiles; ++tl) {                         // KB
            uint32_t lo, hi; tile_range(c, prefmax.data(), reads.data(), tl, lo, hi);
            for (int lane = 0; lane < TILE; ++lane) {
                const int64_t k = tl * TILE + lane; const bool valid = k < P;
                LaneAcc a; lane_init(a); a.dom_b = dominant_bucket(c, in, c.pos0 + k);
                if (getenv("BRC_SIM_DOM")) a.dom_b = (uint32_t)atoi(getenv("BRC_SIM_DOM"));   // stress the alternate/overflow paths
                LaneOut o; o.pl = pl; o.lib = l; o.k = valid ? k : 0;
                TermTab tt; tt.q = tq.data(); tt.e = te.data();
                for (uint32_t r = lo; r < hi; ++r) lane_visit_read(c, in, reads[r], r, (uint32_t)l + 1, (int32_t)(c.pos0 + k), valid, tt, o, a);
                if (!valid) continue;
                lane_store(c, o, a);
                const bool dead = c.per_lib && a.unavail != NONE32;
                if (!dead) { warn[BRC_W_SM_MISSING] += a.w_sm; warn[BRC_W_NM_MISSING] += a.w_nm; if (c.pos0 + k >= c.beg0) n_events += a.ncol; }
                if (dead && l == 0) warn[BRC_W_LIB_UNAVAILABLE]++;
            }
        }
        for (int64_t k = 0; k < P; ++k) {
            if (c.pos0 + k < c.beg0) continue;
            uint32_t tot = 0; for (int l = 0; l < Lp; ++l) tot += ncol[(size_t)(l * PS + k)];
            if (tot) n_positions++;
        }
        // indel events: count -> scan -> fill -> reduce
        std::vector<uint32_t> cnt((size_t)(P * Lp) + 1, 0), off((size_t)(P * Lp) + 1, 0);
        for (int64_t i = 0; i < n; ++i) {
            const DRead& rd = reads[(size_t)i]; const int lib = (int)((rd.misc >> 16) & 0xffu) - 1;
            enumerate_indels(c, in, rd, in.qual + in.qual_off[i], [&](int32_t p, int, int) { cnt[(size_t)((int64_t)(p - c.pos0) * Lp + lib)]++; });
        }
        uint32_t run = 0;
        for (size_t k = 0; k < cnt.size(); ++k) { off[k] = run; run += cnt[k]; }
        std::vector<IndelEv> ev(run + 1); std::vector<uint32_t> cur(off);
        for (int64_t i = n - 1; i >= 0; --i) {   // reversed on purpose: the reduction must not depend on fill order
            const DRead& rd = reads[(size_t)i]; const int lib = (int)((rd.misc >> 16) & 0xffu) - 1;
            enumerate_indels(c, in, rd, in.qual + in.qual_off[i], [&](int32_t p, int qpos, int len) {
                IndelEv e; e.read = (uint32_t)i; e.qpos = qpos; e.len = len; e.key_lo = 0;
                ev[cur[(size_t)((int64_t)(p - c.pos0) * Lp + lib)]++] = e;
            });
        }
        iout.clear();
        std::vector<IndelOut> tmp;
        for (int64_t key = 0; key < P * Lp; ++key) {
            const int nk = (int)cnt[(size_t)key]; if (!nk) continue;
            const int64_t k = key / Lp; const int lib = (int)(key % Lp);
            if (c.per_lib && unavail[(size_t)k] != NONE32) continue;                                  // position abandoned
            tmp.resize((size_t)nk);
            uint32_t wsm = 0, wnm = 0;
            const int na = reduce_indel_key(c, in, reads.data(), ev.data() + off[(size_t)key], nk, (int32_t)(c.pos0 + k), lib, tmp.data(), wsm, wnm);
            warn[BRC_W_SM_MISSING] += wsm; warn[BRC_W_NM_MISSING] += wnm;
            for (int a = 0; a < na; ++a) iout.push_back(tmp[(size_t)a]);
        }
        return BRC_OK;
    }
    int fetch(HostPlanes* out) override {
        out->ncol = ncol.data(); out->depth = depth.data(); out->istat = istat.data(); out->fstat = fstat.data(); out->unavail = unavail.data();
        out->indel = iout.data(); out->n_indel = (int64_t)iout.size(); out->n_events = n_events; out->n_positions = n_positions;
        memcpy(out->warn, warn, sizeof warn);
        return BRC_OK;
    }
    int counts(uint64_t* e, uint64_t* p) override { if (e) *e = n_events; if (p) *p = n_positions; return BRC_OK; }
};

Backend* make_backend(const brc_config&, int* err) { *err = BRC_OK; return new SimBackend(); }
const char* backend_kind() { return "sim-cpu"; }
const char* backend_kernel_name(int) { return nullptr; }

}  // namespace brc
