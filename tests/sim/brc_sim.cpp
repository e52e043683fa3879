// brc_sim.cpp — CPU simulator of the device pipeline.  TEST INFRASTRUCTURE ONLY.
//
// Runs the __host__ __device__ functions of bam_readcount_amd/csrc/brc_core.h — the same code the HIP kernels wrap — in
// the kernels' launch structure: K1 (annotate + piece emission), per-library running max + tile ranges, KB (64-lane
// tiles walking pieces in half-batches of HALF, packed integer registers with flushes every K pieces, queued third-allele
// / huge-integer events drained between half-batches), indel count / scan / fill / reduce.  It lets `-m "not gpu"` tests
// check the device ALGORITHM against the oracle without a GPU.  Never linked into the product, never loaded by it.
#include <stdlib.h>
#include <string.h>

#include <string>
#include <map>
#include <vector>

#include "../../bam_readcount_amd/csrc/brc_host.h"

namespace brc {

static void* sim_alloc(size_t n) { return malloc(n ? n : 1); }
static void sim_release(void* p) { free(p); }
static const HostAlloc kAlloc = {sim_alloc, sim_release};

class SimBackend : public Backend {
    DevCfg c; DevIn in; std::string err;
    const Staged* st = nullptr;
    std::vector<DRead> reads; std::vector<uint8_t> eb; std::vector<uint32_t> wbuf; size_t bq_n = 0;      // wbuf: the sparse wide stream, table then rows (DevIn.bqw)
     std::vector<float> tq; std::vector<double> te;
    std::vector<Piece> hot; std::vector<PieceRare> rare; std::vector<int32_t> key, reach, prefmax;
    // the rare record of piece m: stored by K1 only when piece_has_rare(flags), derived from the piece otherwise
    PieceRare rare_of(uint32_t m) const { return piece_has_rare(piece_flags(hot[m])) ? rare[m] : piece_rare_of(c, hot[m]); }
    std::vector<uint32_t> ncol, depth, slotid, si, unavail; std::vector<float> sf; std::vector<XEv> xev; uint32_t xev_n = 0;
    std::vector<IndelOut> iout;
    // the side tables a text lane reads (brc_core.h: TextAux), in the kernels' layout: folded third-allele records per (tile, library) bucket,
    // reduced indel buckets in their slots
    std::vector<XAgg> xagg, xagg_list; std::vector<uint32_t> xagg_end, xagg_cnt; std::vector<IndelOut> ib_slots; std::vector<uint32_t> ib_end, ib_cnt; bool have_indel_tables = false;
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
        c.n_reads = s.n; c.table_len = getenv("BRC_NO_TABLE") ? 0 : s.modal_len(); c.n_pieces = s.n_pieces; c.max_lqseq = s.max_lqseq;
        c.force_dom = getenv("BRC_FORCE_DOM") ? atoi(getenv("BRC_FORCE_DOM")) : -1;
        c.ibucket_shift = indel_bucket_shift(s.n_indel_ops, c.P, c.Lp);
        if (const char* ib = getenv("BRC_IBUCKET_SHIFT")) { const int v = atoi(ib); if (v == 2 || v == 4 || v == 6) c.ibucket_shift = v; }   // (as the HIP backend: the supported sizes)
        choose_pack(s.max_lqseq, getenv("BRC_FLUSH_K") ? atoi(getenv("BRC_FLUSH_K")) : 0, getenv("BRC_PACK_LIM") ? atoi(getenv("BRC_PACK_LIM")) : 0, c.flush_k, c.pack_lim, c.pack_lim_lo, c.pack_shift);
        tq.assign((size_t)TABLE_MAX + 2, 0.0f); te.assign((size_t)TABLE_MAX + 2, 0.0);
        for (int k = 0; k <= c.table_len; ++k) { tq[(size_t)k] = (float)k / (float)c.table_len; te[(size_t)k] = 1.0 - (double)tq[(size_t)k]; }
        in.pos = s.pos.p; in.flag = s.flag.p; in.mapq = s.mapq.p; in.lib = s.lib.p; in.l_qseq = s.l_qseq.p; in.n_cigar = s.n_cigar.p;
        in.cig_off = s.cig_off.p; in.seq_off = s.seq_off.p; in.qual_off = s.qual_off.p; in.nm = s.nm.p; in.sm = s.sm.p; in.tags = s.tags.p;
        in.cigar = s.cigar.p; in.seq4 = s.seq4.p; in.qual = s.qual.p; in.ref = g.ref ? g.ref + g.ref_lo : nullptr;
        bq_n = s.bq_elems; in.eb = nullptr; in.bqw = nullptr; in.bq_row = s.bq_row.p;
        st = &s;
        return BRC_OK;
    }

    bool stage_fault = false;            // a staged window outside the padded event-byte stream: the device would fault (checked at the end of compute)
    bool stats_on = false; uint64_t stat_steps = 0, stat_dead = 0;      // BRC_SIM_PIECE_STATS: piece-steps, and those whose piece does not touch the tile
    uint32_t tile_want = 0u | (63u << 8);     // brc_region_windows: the lanes of the current tile that a window asks for
    // one (tile, library) wave of KB
    void pileup_tile(const Planes& pl, int lib, int64_t tl, uint32_t lo, uint32_t hi) {
        LaneAcc2 a[TILE]; bool valid[TILE], inreg[TILE]; int32_t p[TILE]; int64_t kk[TILE];
        for (int l = 0; l < TILE; ++l) {
            kk[l] = tl * TILE + l; inreg[l] = kk[l] < c.P; p[l] = (int32_t)(c.pos0 + kk[l]);
            // a position abandoned for a library-less read (:281-284) accumulates nothing: it behaves like a lane outside the region
            valid[l] = inreg[l] && !(c.per_lib && unavail[(size_t)kk[l]] != NONE32) && tile_wants(tile_want, (uint32_t)l);
            const uint32_t dom = valid[l] ? dominant_bucket(c, in, p[l]) : 1u;
            lane2_init(a[l], dom);
        }
        TermTab tt; tt.q = tq.data(); tt.e = te.data();
        const uint32_t thr = piece_thr(c);
        struct QEnt { uint32_t piece; int kind; bool lane[TILE]; };    // kind 0: third-allele events, 1: integers of a huge piece
        std::vector<QEnt> queue;
        int since_flush = 0; bool flushed = false;
        for (uint32_t base = lo; base < hi; base += HALF) {
            const uint32_t nb = hi - base < (uint32_t)HALF ? hi - base : (uint32_t)HALF;
            for (uint32_t m = base; m < base + nb; ++m) {
                const Piece& h = hot[m];
                const uint32_t fl = piece_flags(h);
                QEnt full, ints; full.piece = ints.piece = m; full.kind = 0; ints.kind = 1; bool any_full = false, any_int = false;
                {   // the window of this piece's event bytes the device would stage for this tile must lie inside the padded stream
                    const int32_t p0w = (int32_t)(c.pos0 + tl * TILE);
                    const int64_t w0 = (int64_t)h.bq_off + stage_window_start(p0w, h.a, c.max_lqseq);
                    if (w0 < -(int64_t)EB_PAD_FRONT || w0 + EB_WINDOW > (int64_t)bq_n + EB_PAD_BACK + c.max_lqseq) stage_fault = true;     // (the HIP backend's allocation)
                }
                if (stats_on) { const int64_t t0 = c.pos0 + tl * TILE; ++stat_steps; if (!((int64_t)h.rs < t0 + TILE && (int64_t)h.rs + h.ext > t0)) ++stat_dead; }
                for (int l = 0; l < TILE; ++l) {
                    full.lane[l] = ints.lane[l] = false;
                    if (!valid[l]) continue;
                    const uint32_t d = (uint32_t)(p[l] - h.rs);
                    if (d < (uint32_t)h.ext) a[l].ncol++;                                             // lib_counts[library] (:286)
                    if (!(d < (uint32_t)h.len)) continue;
                    const int qpos = p[l] - h.a;
                    uint32_t w = eb[h.bq_off + (uint64_t)qpos];
                    if (w < thr) continue;                                                            // :288
                    a[l].depth++;                                                                     // mapq_n (:312)
                    if (fl & PF_NB) continue;                                                         // :343 with -i
                    // a wide read's escape bytes: quality and bucket from the wide stream; an N / '=' base goes to the third-allele list
                    // whatever the slots hold
                    if ((fl & PF_WIDE) && eb_is_escape(w)) {
                        const uint32_t w16 = in.bqw[wide_base(in.bqw, h.bq_off) + (uint64_t)qpos];
                        if (!bucket_acgt(w16 & 0xffu)) { a[l].ww += h.ww; full.lane[l] = true; any_full = true; continue; }
                        w = ((w16 >> 8) << 2) | ((w16 & 0xffu) - 1u);
                    }
                    EvTerms t = (fl & (PF_TABLE | PF_TABQ)) ? piece_terms_tab(h, tt, c.table_len, qpos) : (fl & PF_DIV) ? piece_terms_inlane(fl, h.tp_flags, h.w3 >> c.pack_shift, qpos) : piece_terms_div(fl, piece_tp(c, h), rare[m], qpos);
                    if ((fl & PF_TABQ) && !(fl & PF_HUGE)) t.sev = tabq_sev(qpos, piece_left_field(h.tp_flags), h.w3 >> c.pack_shift);       // (as k_pileup2 does: no rare record)
                    const uint32_t b = (w & 3u) + 1u;
                    a[l].ww += h.ww;
                    if (b == a[l].dom_b) { pack_event(a[l].dom, h, t, w); if (fl & PF_HUGE) { ints.lane[l] = true; any_int = true; } }
                    else if (a[l].alt_b == NB_NONE || a[l].alt_b == b) { a[l].alt_b = b; pack_event(a[l].alt, h, t, w); if (fl & PF_HUGE) { ints.lane[l] = true; any_int = true; } }
                    else { full.lane[l] = true; any_full = true; }
                }
                if (any_full) queue.push_back(full);
                if (any_int) queue.push_back(ints);
            }
            since_flush += (int)nb;
            // between half-batches: drain the queue (the event bytes of this half-batch are still staged), flush when the
            // packed fields could overflow during the next half-batch
            for (const QEnt& e : queue) {
                const Piece& h = hot[e.piece]; const PieceRare rr = rare_of(e.piece);
                if (e.kind == 1 && !flushed) {                                                        // huge integers go straight to the slot planes: make them live
                    for (int l = 0; l < TILE; ++l) if (valid[l]) lane2_flush(c, pl, lib, kk[l], a[l], false);
                    flushed = true; since_flush = 0;
                }
                for (int l = 0; l < TILE; ++l) {
                    if (!e.lane[l]) continue;
                    const int qpos = p[l] - h.a;
                    const uint32_t w = eb[h.bq_off + (uint64_t)qpos];
                    uint32_t q = w >> 2, b = (w & 3u) + 1u;
                    if ((piece_flags(h) & PF_WIDE) && eb_is_escape(w)) { const uint32_t w16 = in.bqw[wide_base(in.bqw, h.bq_off) + (uint64_t)qpos]; q = w16 >> 8; b = w16 & 0xffu; }
                    if (e.kind == 0) { const XEv x = make_xev(c, lib, kk[l], h, rr, qpos, q, b); const uint32_t at = (*pl.xev_n)++; if (at < pl.xev_cap) pl.xev[at] = x; }
                    else drain_int(c, pl, lib, kk[l], rr, b == a[l].dom_b ? 0u : 1u);
                }
            }
            queue.clear();
            if (since_flush + HALF > c.flush_k) {
                for (int l = 0; l < TILE; ++l) if (valid[l]) lane2_flush(c, pl, lib, kk[l], a[l], flushed);
                flushed = true; since_flush = 0;
            }
        }
        for (int l = 0; l < TILE; ++l) {
            if (!inreg[l]) continue;
            const bool dead = !valid[l];
            lane2_store(c, pl, lib, kk[l], a[l], dead, flushed && !dead);
            if (!dead) { warn[BRC_W_SM_MISSING] += a[l].ww & 0xffffu; warn[BRC_W_NM_MISSING] += a[l].ww >> 16; if (p[l] >= c.beg0) n_events += a[l].ncol; }
            if (dead && lib == 0) warn[BRC_W_LIB_UNAVAILABLE]++;
        }
    }

    int compute(brc_timing* t) override {
        if (t) memset(t, 0, sizeof *t);
        const int64_t n = c.n_reads, P = c.P, PS = c.PS; const int Lp = c.Lp;
        const int64_t np = c.n_pieces;
        reads.resize((size_t)n); eb.assign(bq_n + 1, 0); in.eb = eb.data();
        {   // table entries nobody set point far outside the rows; wide words nobody wrote must not be read
            const size_t tab = (((bq_n >> 4) + 2) + 7) & ~(size_t)7;                  // u32 entries: a multiple of 16 elements
            std::vector<Staged::WidePair> pairs(st->n_wide()); const uint64_t wq = st->wide_layout(pairs.data(), (uint32_t)(tab / 8));
            wbuf.assign(tab + (size_t)(wq / 2) + 8, 0xdeaddeadu);
            for (size_t k = 0; k < tab; ++k) wbuf[k] = 0xffffffffu;
            for (const Staged::WidePair& x : pairs) wbuf[(size_t)(st->bq_row.p[x.read] >> 4)] = x.w16;      // (a wide read has bases: no other read's row starts in its first chunk)
            in.bqw = reinterpret_cast<const uint16_t*>(wbuf.data());
        }
        uint16_t* const bqw_w = const_cast<uint16_t*>(in.bqw);
        hot.assign((size_t)np + 1, Piece()); { PieceRare poison; memset(&poison, 0xff, sizeof poison); rare.assign((size_t)np + 1, poison); }   // (a rare record nobody wrote must not be read)
        key.assign((size_t)np + 1, 0); reach.assign((size_t)np + 1, 0); prefmax.assign((size_t)np + 1, 0);
        unavail.assign((size_t)PS, NONE32);
        for (int64_t i = 0; i < n; ++i) {                                                             // K1
            bool wide = false;
            const DRead rd = reads[(size_t)i] = annotate_read(c, in, i, eb.data(), bqw_w, wide);
            if (wide != (st->wide.p[i] != 0)) { err = "the host's scan for escape bases and K1 differ"; return BRC_E_ARG; }
            const uint32_t* cg = in.cigar + in.cig_off[i];
            const bool nolib = c.per_lib && in.lib[i] < 0;
            const bool enters = read_enters(in.flag[i], cg, in.n_cigar[i]) && in.pos[i] >= 0;
            if (enters && nolib)                                                                       // k_unavail: first library-less read of every column (:281-284)
                for (int64_t q = rd.pos; q < rd.end; ++q) { const int64_t k = q - c.pos0; if (k >= 0 && k < P && unavail[(size_t)k] > (uint32_t)i) unavail[(size_t)k] = (uint32_t)i; }
            const ReadConst rc = read_const(c, rd, (uint32_t)i, wide);
            uint32_t slot = st->piece_off.p[i], cnt = 0;
            walk_pieces(c.insertion_centric != 0, enters && !nolib, rc.counts, rd.pos, cg, in.n_cigar[i], [&](int32_t rs, int32_t len, int32_t ext, int qoff, bool nb) {
                PieceRare rr; make_piece(c, rc, rs, len, ext, qoff, nb, hot[slot], rr, c.pack_shift);
                if (piece_has_rare(piece_flags(hot[slot]))) rare[slot] = rr;
                key[slot] = rd.pos; reach[slot] = rs + ext; ++slot; ++cnt;
            });
            if (cnt != st->piece_cnt.p[i]) { err = "piece count of the host and of K1 differ"; return BRC_E_ARG; }
        }
        stats_on = getenv("BRC_SIM_PIECE_STATS") != nullptr; stat_steps = stat_dead = 0;
        if (getenv("BRC_SIM_PIECE_STATS")) {      // (diagnostics: how many pieces of each flag combination a data set produces)
            std::map<uint32_t, uint64_t> hist;
            for (int64_t m = 0; m < np; ++m) hist[piece_flags(hot[(size_t)m])]++;
            for (const auto& kv : hist) fprintf(stderr, "piece flags 0x%03x: %llu (%.2f %%)\n", kv.first, (unsigned long long)kv.second, 100.0 * (double)kv.second / (double)np);
        }
        ncol.assign((size_t)(Lp * PS), 0); depth.assign((size_t)(Lp * PS), 0); slotid.assign((size_t)(Lp * PS), 0xdeadbeefu);
        si.assign((size_t)(Lp * 2 * NI * PS), 0xdeadbeefu); sf.assign((size_t)(Lp * 2 * NF * PS), -1.0f);   // KB must write every plane element
        const char* xc = getenv("BRC_XEV_CAP");                                                            // (test knob: a tiny list exercises the grow-and-recompute path)
        if (xev.empty()) xev.resize(xc ? (size_t)atoi(xc) : 1024);
      again:
        xev_n = 0;
        Planes pl = {ncol.data(), depth.data(), slotid.data(), si.data(), sf.data(), unavail.data(), xev.data(), &xev_n, (uint32_t)xev.size(), 1u};
        n_events = n_positions = 0; memset(warn, 0, sizeof warn);
        const int64_t ntiles = (P + TILE - 1) / TILE;
        const std::vector<uint16_t> wanted = st->wanted_tiles(c.pos0, c.P);
        for (int l = 0; l < Lp; ++l) {
            const int64_t s0 = st->lib_base[(size_t)l], s1 = st->lib_base[(size_t)l + 1];
            int32_t m = INT32_MIN;
            for (int64_t i = s0; i < s1; ++i) { if (reach[(size_t)i] > m) m = reach[(size_t)i]; prefmax[(size_t)i] = m; }
            for (int64_t tl = 0; tl < ntiles; ++tl) {                                                 // KB
                uint32_t lo, hi; tile_range2(c, prefmax.data(), key.data(), s0, s1, tl, lo, hi);
                if (!wanted.empty() && wanted[(size_t)tl] == (uint16_t)TILE_UNWANTED) {
                    // brc_region_windows: exactly what k_mask_tiles leaves of a tile nobody announced — no column, no depth, no slot;
                    // its statistics planes are never written (and must never be read: they keep their poison here)
                    for (int ln = 0; ln < TILE; ++ln) { const int64_t k = tl * TILE + ln; if (k >= P) break;
                        ncol[(size_t)(l * PS + k)] = 0; depth[(size_t)(l * PS + k)] = 0; slotid[(size_t)(l * PS + k)] = (uint32_t)NB_NONE | ((uint32_t)NB_NONE << 8); }
                    continue;
                }
                if (!wanted.empty()) {
                    // an announced tile is piled up for the lanes its windows ask for: pieces that cannot reach them are trimmed off
                    // both ends of its range (k_mask_tiles), the other lanes behave like lanes outside the region
                    const uint32_t w = wanted[(size_t)tl];
                    const int64_t p0w = (int64_t)c.pos0 + tl * TILE + (w & 0xffu), p1w = (int64_t)c.pos0 + tl * TILE + (w >> 8);
                    while (lo < hi && (int64_t)reach[lo] <= p0w) ++lo;
                    while (hi > lo && (int64_t)key[hi - 1] > p1w) --hi;
                    tile_want = w;
                } else tile_want = 0u | (63u << 8);
                pileup_tile(pl, l, tl, lo, hi);
            }
        }
        if (xev_n > xev.size()) { xev.resize((size_t)xev_n * 2); goto again; }                                 // the list was too short: grow, compute again
        pl_last = pl; have_pl = true;
        {   // k_xev_compact's count / scan / k_xev_scatter / k_xev_fold: the events of every (tile, library) bucket folded in list order
            const int64_t nxb = ntiles * Lp;
            std::vector<uint32_t> cnt((size_t)nxb + 1, 0), off((size_t)nxb + 1, 0);
            for (uint32_t i = 0; i < xev_n; ++i) cnt[(size_t)((int64_t)(xev[i].k >> 6) * Lp + (xev[i].lib_b >> 8))]++;
            uint32_t run = 0; for (int64_t b = 0; b < nxb; ++b) { off[(size_t)b] = run; run += cnt[(size_t)b]; }
            std::vector<uint32_t> idx(run + 1), cur(off);
            for (int64_t i = (int64_t)xev_n - 1; i >= 0; --i) idx[cur[(size_t)((int64_t)(xev[(size_t)i].k >> 6) * Lp + (xev[(size_t)i].lib_b >> 8))]++] = (uint32_t)i;    // reversed on purpose: the fold must not depend on scatter order
            xagg.assign(run + 1, XAgg()); xagg_end.assign((size_t)nxb + 1, 0); xagg_cnt.assign((size_t)nxb + 1, 0); xagg_list.clear();
            for (int64_t b = 0; b < nxb; ++b) {
                const uint32_t nk = cnt[(size_t)b]; if (!nk) { xagg_end[(size_t)b] = off[(size_t)b]; continue; }
                const int nd = fold_xev_bucket(xev.data(), idx.data() + off[(size_t)b], (int)nk, xagg.data() + off[(size_t)b]);
                xagg_cnt[(size_t)b] = (uint32_t)nd; xagg_end[(size_t)b] = off[(size_t)b] + (uint32_t)nd;
                for (int j = 0; j < nd; ++j) xagg_list.push_back(xagg[off[(size_t)b] + (size_t)j]);
            }
        }
        for (int64_t k = 0; k < P; ++k) {
            if (c.pos0 + k < c.beg0) continue;
            uint32_t tot = 0; for (int l = 0; l < Lp; ++l) tot += ncol[(size_t)(l * PS + k)];
            if (tot) n_positions++;
        }
        // indel side path, in the kernels' structure: K1 writes every read's events to its own slots of the raw list (one slot
        // per I / D / P operator, unused ones marked empty) and counts them per (tile, library) bucket; scan; scatter; one
        // reduce_indel_bucket per bucket
        iout.clear(); have_indel_tables = false;
        if (c.has_ref && P > 0 && n > 0) {
            const int64_t nbk = indel_buckets(c);
            std::vector<IndelEv> raw((size_t)st->n_indel_ops + 1);
            std::vector<uint32_t> cnt((size_t)nbk + 1, 0), off((size_t)nbk + 1, 0);
            for (int64_t i = 0; i < n; ++i) {
                const DRead& rd = reads[(size_t)i]; const int lib = (int)((rd.misc >> 16) & 0xffu) - 1;
                uint32_t n_idp = 0;
                for (uint32_t k = 0; k < in.n_cigar[i]; ++k) { const uint32_t op = in.cigar[in.cig_off[i] + k] & 0xfu; if (op == CINS || op == CDEL || op == CPAD) ++n_idp; }
                const uint64_t next = i + 1 < n ? st->iev_off.p[i + 1] : st->n_indel_ops;
                if (next - st->iev_off.p[i] != n_idp) { err = "raw indel slots of the host and of K1 differ"; return BRC_E_ARG; }
                IndelEv* slot = raw.data() + st->iev_off.p[i]; uint32_t used = 0;
                enumerate_indels(c, in, rd, in.qual + in.qual_off[i], [&](int32_t p, int qpos, int len) {
                    IndelEv e; e.read = (uint32_t)i; e.qpos = qpos; e.len = len; e.key_lo = (uint32_t)((int64_t)(p - c.pos0) * Lp + lib);
                    if (!wanted.empty() && !tile_wants(wanted[(size_t)((uint32_t)(p - c.pos0) >> 6)], (uint32_t)(p - c.pos0) & 63u)) return;     // (as K1: no indel alleles outside the announced windows)
                    if (used < n_idp) { slot[used++] = e; cnt[indel_bucket_of(c, (uint32_t)(p - c.pos0), (uint32_t)lib)]++; }
                });
                for (; used < n_idp; ++used) slot[used].key_lo = NONE32;
            }
            uint32_t run = 0;
            for (int64_t b = 0; b < nbk; ++b) { off[(size_t)b] = run; run += cnt[(size_t)b]; }
            std::vector<IndelEv> ev(run + 1); std::vector<uint32_t> cur(off);
            for (int64_t j = (int64_t)st->n_indel_ops - 1; j >= 0; --j) {   // reversed on purpose: the reduction must not depend on scatter order
                const IndelEv& e = raw[(size_t)j];
                if (e.key_lo != NONE32) ev[cur[indel_bucket(c, e.key_lo)]++] = e;
            }
            std::vector<IndelOut> tmp(run + 1);
            for (int64_t b = 0; b < nbk; ++b) {
                const uint32_t nk = cnt[(size_t)b]; if (!nk) continue;
                uint32_t wsm = 0, wnm = 0;
                reduce_indel_bucket(c, in, reads.data(), ev.data() + off[(size_t)b], (int)nk, unavail.data(), tmp.data() + off[(size_t)b], wsm, wnm);
                warn[BRC_W_SM_MISSING] += wsm; warn[BRC_W_NM_MISSING] += wnm;
            }
            for (uint32_t j = 0; j < run; ++j) if (tmp[j].len != 0) iout.push_back(tmp[j]);
            ib_slots = tmp; ib_cnt = cnt; ib_end.assign((size_t)nbk + 1, 0); for (int64_t b = 0; b < nbk; ++b) ib_end[(size_t)b] = off[(size_t)b] + cnt[(size_t)b];
            have_indel_tables = true;
        }
        if (stage_fault) { stage_fault = false; err = "a tile would stage a window of event bytes outside the padded stream (k_pileup2: BRC_STAGE)"; return BRC_E_HIP; }
        if (stats_on) fprintf(stderr, "piece-steps %llu, of them %llu (%.2f %%) of a piece that does not touch the tile; %.1f events per step\n", (unsigned long long)stat_steps,
                              (unsigned long long)stat_dead, 100.0 * (double)stat_dead / (double)(stat_steps ? stat_steps : 1), (double)n_events / (double)(stat_steps ? stat_steps : 1));
        return BRC_OK;
    }
    // "download": like the HIP backend, the host view is a copy — upload / compute of the next region may run while the
    // previous result is still being formatted (include/brc.h, threads)
    std::vector<uint32_t> h_ncol, h_depth, h_slotid, h_si, h_unavail; std::vector<float> h_sf; std::vector<XAgg> h_xagg; std::vector<IndelOut> h_iout;
    std::vector<char> t_text2[2]; std::vector<uint32_t> t_off2[2], t_last[2]; int t_slot = 0; Planes pl_last; bool have_pl = false;
    int text_begin(const std::string& chrom, const std::vector<std::string>& libs, int* slot) override {
        if (!have_pl) return BRC_E_ARG;
        t_slot ^= 1; *slot = t_slot;
        std::vector<char>& t_text = t_text2[t_slot]; std::vector<uint32_t>& t_off = t_off2[t_slot];
        std::vector<int32_t> loff((size_t)c.Lp + 1, 0); std::string names;
        if (c.per_lib) for (int l = 0; l < c.Lp; ++l) { loff[(size_t)l] = (int32_t)names.size(); if ((size_t)l < libs.size()) names += libs[(size_t)l]; loff[(size_t)l + 1] = (int32_t)names.size(); }
        TextCtx t; t.chrom = chrom.data(); t.chrom_len = (int32_t)chrom.size(); t.lib_names = names.data(); t.lib_off = loff.data();
        TextAux ax; memset(&ax, 0, sizeof ax);
        ax.xagg = xagg.data(); ax.xagg_end = xagg_end.data(); ax.xagg_cnt = xagg_cnt.data();
        if (have_indel_tables) { ax.iout = ib_slots.data(); ax.ib_end = ib_end.data(); ax.ib_cnt = ib_cnt.data(); ax.reads = reads.data(); }
        t_last[t_slot].assign((size_t)c.Lp, NONE32);
        for (int l = 0; l < c.Lp; ++l) for (int64_t k = c.P - 1; k >= 0; --k) if (ncol[(size_t)(l * c.PS + k)] != 0 && !(c.per_lib && unavail[(size_t)k] != NONE32)) { t_last[t_slot][(size_t)l] = (uint32_t)k; break; }     // k_last_processed
        t_off.assign((size_t)c.P + 1, 0);
        uint64_t total64 = 0;
        for (int64_t k = 0; k < c.P; ++k) { const uint32_t n = text_line(c, in, pl_last, t, ax, k, nullptr); t_off[(size_t)k + 1] = t_off[(size_t)k] + n; total64 += n; }
        const uint64_t limit = getenv("BRC_DEVICE_TEXT_LIMIT") ? strtoull(getenv("BRC_DEVICE_TEXT_LIMIT"), nullptr, 10) : ~0ull;     // (test knob, as in the HIP backend)
        if (total64 != (uint64_t)t_off[(size_t)c.P] || total64 > limit) { t_slot ^= 1; return BRC_TEXT_TOO_LONG; }     // (as the HIP backend: 32-bit offsets cannot address it)
        t_text.assign((size_t)t_off[(size_t)c.P] + 1, 0);
        for (int64_t k = 0; k < c.P; ++k) if (t_off[(size_t)k + 1] > t_off[(size_t)k]) (void)text_line(c, in, pl_last, t, ax, k, t_text.data() + t_off[(size_t)k]);
        return BRC_OK;
    }
    void list_sizes(uint64_t* nx, uint64_t* ni) override { *nx = xev_n; *ni = iout.size(); }
    int text_wait(int slot, HostText* out) override {
        const std::vector<char>& t_text = t_text2[slot & 1]; const std::vector<uint32_t>& t_off = t_off2[slot & 1];
        out->text = t_text.data(); out->off = t_off.data(); out->total = t_off.empty() ? 0 : t_off.back(); out->n = (int64_t)t_off.size() - 1; out->last_processed = t_last[slot & 1].data(); return BRC_OK;
    }
    int fetch(HostPlanes* out, bool) override {
        h_ncol = ncol; h_depth = depth; h_slotid = slotid; h_si = si; h_unavail = unavail; h_sf = sf; h_xagg = xagg_list; h_iout = iout;
        out->ncol = h_ncol.data(); out->depth = h_depth.data(); out->slotid = h_slotid.data(); out->si = h_si.data(); out->sf = h_sf.data(); out->unavail = h_unavail.data();
        out->xagg = h_xagg.data(); out->n_xagg = h_xagg.size();
        out->indel = h_iout.data(); out->n_indel = (int64_t)h_iout.size(); out->n_events = n_events; out->n_positions = n_positions;
        memcpy(out->warn, warn, sizeof warn);
        return BRC_OK;
    }
    std::vector<uint32_t> w_ncol, w_depth, w_slotid, w_si, w_unavail; std::vector<float> w_sf;
    int fetch_window(int64_t k0, int64_t n, HostPlanes* out, int64_t* stride) override {
        if (!have_pl || k0 < 0 || n < 0 || k0 + n > c.P) return BRC_E_ARG;
        const int64_t WS = n + 5, PS = c.PS; const int Lp = c.Lp;      // (an odd stride of its own)
        auto cut = [&](const auto& src, auto& dst, int64_t planes) {
            dst.assign((size_t)(planes * WS), 0);
            for (int64_t pl = 0; pl < planes; ++pl) for (int64_t k = 0; k < n; ++k) dst[(size_t)(pl * WS + k)] = src[(size_t)(pl * PS + k0 + k)];
        };
        cut(ncol, w_ncol, Lp); cut(depth, w_depth, Lp); cut(slotid, w_slotid, Lp); cut(si, w_si, (int64_t)Lp * 2 * NI); cut(sf, w_sf, (int64_t)Lp * 2 * NF); cut(unavail, w_unavail, 1);
        h_xagg = xagg_list; h_iout = iout;
        *out = HostPlanes();
        out->ncol = w_ncol.data(); out->depth = w_depth.data(); out->slotid = w_slotid.data(); out->si = w_si.data(); out->sf = w_sf.data(); out->unavail = w_unavail.data();
        out->xagg = h_xagg.data(); out->n_xagg = h_xagg.size(); out->indel = h_iout.data(); out->n_indel = (int64_t)h_iout.size();
        out->n_events = n_events; out->n_positions = n_positions;
        *stride = WS;
        return BRC_OK;
    }
    int counts(uint64_t* e, uint64_t* p) override { if (e) *e = n_events; if (p) *p = n_positions; return BRC_OK; }
};

void* backend_host_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }      // (no device: plain memory; pushes copy, Backend::adopts_arenas)
void backend_host_free(void* p) { free(p); }

Backend* make_backend(const brc_config&, int* err) { *err = BRC_OK; return new SimBackend(); }
const char* backend_kind() { return "sim-cpu"; }
const char* backend_kernel_name(int) { return nullptr; }

}  // namespace brc
