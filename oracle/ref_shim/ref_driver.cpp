/*
 * ref_driver.cpp — drives the REFERENCE'S OWN callbacks, compiled unmodified, behind the brc C-ABI.
 *
 * TEST INFRASTRUCTURE ONLY (oracle/_ref/libbamrc_ref.so; never linked or loaded by the product).  This translation
 * unit #includes /root/reference/src/exe/bam-readcount/bamreadcount.cpp as it lies (main renamed), so fetch_func
 * (bamreadcount.cpp:114-261) and pileup_func (:265-419) are the reference's code; BasicStat.cpp, IndelQueue.cpp and
 * IndelQueueEntry.cpp are compiled from src/lib/bamrc/ next to it (Makefile).  Only the samtools/htslib layer under
 * them is a shim (sam.h / shim_hts.cpp in this directory).  The region plumbing of main() (:588-605, :644-656) is
 * repeated in run_region() below; the reference's main() itself is also linked into oracle/_ref/bam-readcount-ref.
 *
 * Exports the subset of include/brc.h that makes sense for a text-only engine (planes are not observable in the
 * reference: brc_fetch_result reports n_pos = 0) plus bamrc_ref_* entry points for unit-level pinning.
 */
#define main brc_reference_main
#include "src/exe/bam-readcount/bamreadcount.cpp"
#undef main

#include <sstream>

#include "../../bam_readcount_amd/csrc/io/bamio.h"
#include "brc.h"

namespace {

struct RefRead { bam1_t b; };

void build_bam1(bam1_t* b, int tid, const brc_read_batch* bt, int64_t i, int64_t global_index, bool per_lib) {
    memset(b, 0, sizeof *b);
    char namebuf[32];
    const char* qn = (bt->qname && bt->qname[i]) ? bt->qname[i] : (snprintf(namebuf, sizeof namebuf, "r%lld", (long long)global_index), namebuf);
    const size_t lq = strlen(qn) + 1;
    const uint32_t nc = bt->n_cigar[i]; const int32_t L = bt->l_qseq[i];
    std::vector<uint8_t> d;
    d.insert(d.end(), (const uint8_t*)qn, (const uint8_t*)qn + lq);
    const uint8_t* cg = (const uint8_t*)(bt->cigar + bt->cigar_off[i]);
    d.insert(d.end(), cg, cg + 4u * nc);
    const uint8_t* sq = bt->seq4 + bt->seq_off[i];
    d.insert(d.end(), sq, sq + (L + 1) / 2);
    const uint8_t* ql = bt->qual + bt->qual_off[i];
    d.insert(d.end(), ql, ql + L);
    const uint8_t tags = bt->tags ? bt->tags[i] : 0;
    if (tags & BRC_TAG_NM) { int32_t v = bt->nm[i]; d.push_back('N'); d.push_back('M'); d.push_back('i'); d.insert(d.end(), (uint8_t*)&v, (uint8_t*)&v + 4); }
    if (tags & BRC_TAG_SM) { int32_t v = bt->sm[i]; d.push_back('S'); d.push_back('M'); d.push_back('i'); d.insert(d.end(), (uint8_t*)&v, (uint8_t*)&v + 4); }
    if (per_lib && bt->lib && bt->lib[i] >= 0) {
        char rg[32]; const int n = snprintf(rg, sizeof rg, "rg%d", (int)bt->lib[i]);
        d.push_back('R'); d.push_back('G'); d.push_back('Z'); d.insert(d.end(), (uint8_t*)rg, (uint8_t*)rg + n + 1);
    }
    b->core.pos = bt->pos[i]; b->core.tid = tid; b->core.qual = bt->mapq[i]; b->core.flag = bt->flag[i];
    b->core.l_qname = (uint16_t)lq; b->core.n_cigar = nc; b->core.l_qseq = L; b->core.mtid = -1; b->core.mpos = -1;
    b->m_data = (uint32_t)d.size() + 64; b->data = (uint8_t*)malloc(b->m_data);
    memcpy(b->data, d.data(), d.size()); b->l_data = (int)d.size();
}

int noop_pileup(uint32_t, uint32_t, int, const bam_pileup1_t*, void*) { return 0; }

}  // namespace

struct brc_engine {
    brc_config cfg;
    std::vector<std::string> lib_names;
    int64_t max_warnings = -1;
    pileup_data_t d;
    samfile_t in;
    bam_hdr_t hdr;
    brcio::BamHeader shim_hdr;
    std::string hdr_text;          // header text for bam_get_library (one @RG line per library)
    std::vector<char*> target_names;
    // region
    bool in_region = false; int32_t tid = 0, beg0 = 0, end = 0; const char* ref = 0; int64_t ref_len = 0;
    std::vector<bam1_t> reads;
    std::string text, warn_text, fmt_text, err;
    uint64_t n_lines = 0;
    brc_engine() : d() {}
};

static void free_reads(brc_engine* e) {
    for (size_t i = 0; i < e->reads.size(); ++i) free(e->reads[i].data);
    e->reads.clear();
}

extern "C" {

const char* brc_strerror(int code) { return code == 0 ? "ok" : "error"; }
const char* brc_last_error(const brc_engine* e) { return e ? e->err.c_str() : ""; }
const char* brc_kernel_name(int) { return 0; }
const char* brc_engine_kind(void) { return "reference-compiled"; }

int brc_create(const brc_config* cfg, brc_engine** out) {
    if (!cfg || !out || cfg->abi_version != BRC_ABI_VERSION) return BRC_E_ARG;
    brc_engine* e = new brc_engine();
    e->cfg = *cfg;
    for (int l = 0; cfg->per_lib && l < cfg->n_libs; ++l) e->lib_names.push_back(cfg->lib_names[l]);
    // pileup_data_t as main() sets it up (bamreadcount.cpp:430-432,508-511)
    e->d.tid = -1; e->d.min_mapq = cfg->min_mapq; e->d.min_bq = cfg->min_bq;
    e->d.max_cnt = cfg->max_cnt > 0 ? cfg->max_cnt : 10000000;
    e->d.beg = 0; e->d.end = 0x7fffffff; e->d.distribution = 0;
    e->d.per_lib = cfg->per_lib != 0; e->d.insertion_centric = cfg->insertion_centric != 0;
    e->d.indel_queue_map = indel_queue_map_t();
    memset(&e->in, 0, sizeof e->in); memset(&e->hdr, 0, sizeof e->hdr);
    // bam_get_library scans the header text: one @RG line per library of the batch interface (read i of library l carries RG:Z:rg<l>)
    for (size_t l = 0; l < e->lib_names.size(); ++l) e->hdr_text += "@RG\tID:rg" + std::to_string(l) + "\tLB:" + e->lib_names[l] + "\n";
    e->hdr.text = &e->hdr_text[0]; e->hdr.l_text = e->hdr_text.size();
    e->hdr.shim = &e->shim_hdr;
    e->in.header = &e->hdr;
    e->d.in = &e->in;
    *out = e;
    return BRC_OK;
}
void brc_destroy(brc_engine* e) {
    if (!e) return;
    free_reads(e);
    free(e->d.ref);
    for (size_t i = 0; i < e->target_names.size(); ++i) free(e->target_names[i]);
    delete e;
}
int brc_begin_region(brc_engine* e, int32_t tid, int32_t beg0, int32_t end, const char* ref, int64_t ref_len) {
    if (!e || beg0 < 0 || end < beg0 || tid < 0) return BRC_E_ARG;
    free_reads(e);
    e->tid = tid; e->beg0 = beg0; e->end = end; e->ref = ref; e->ref_len = ref ? ref_len : 0; e->in_region = true;
    while ((int)e->target_names.size() <= tid) e->target_names.push_back(strdup("\x01"));   // chrom placeholder, see brc_format_region
    e->hdr.n_targets = (int32_t)e->target_names.size(); e->hdr.target_name = e->target_names.data();
    return BRC_OK;
}
int brc_push_reads(brc_engine* e, const brc_read_batch* b) {
    if (!e || !e->in_region || !b) return BRC_E_ARG;
    for (int64_t i = 0; i < b->n_reads; ++i) {
        bam1_t r; build_bam1(&r, e->tid, b, i, (int64_t)e->reads.size(), e->cfg.per_lib != 0);
        e->reads.push_back(r);
    }
    return BRC_OK;
}
int brc_upload(brc_engine* e) { return e ? BRC_OK : BRC_E_ARG; }

int brc_compute(brc_engine* e, brc_timing* t) {
    if (!e || !e->in_region) return BRC_E_ARG;
    if (t) memset(t, 0, sizeof *t);
    pileup_data_t& d = e->d;
    // load_reference (:83-90) by hand: the contig text as fai_fetch would return it (NUL-terminated)
    free(d.ref); d.ref = 0; d.len = 0; d.fai = 0; d.tid = e->tid;
    if (e->ref) { d.ref = (char*)malloc((size_t)e->ref_len + 1); memcpy(d.ref, e->ref, (size_t)e->ref_len); d.ref[e->ref_len] = 0; d.len = (int)e->ref_len; }
    std::ostringstream out, warn;
    std::streambuf* old = std::cout.rdbuf(out.rdbuf());
    WARN.reset(new ReadWarnings(warn, e->max_warnings));                       // :499
    // the per-region sequence of main(): :588-605 (site list) / :644-656 (command-line region)
    d.beg = e->beg0; d.end = e->end;
    fetch_data_t* f = (fetch_data_t*)calloc(1, sizeof(pileup_data_t));
    bam_plbuf_t* buf = bam_plbuf_init(pileup_func, &d);
    bam_plp_set_maxcnt(buf->iter, d.max_cnt);
    f->pileup_buffer = buf;
    if (e->cfg.ref_len_check && e->ref) { f->ref_len = d.len; f->seq_name = e->hdr.target_name[e->tid]; } else { f->ref_len = 0; f->seq_name = 0; }
    f->ref_pointer = &d.ref;
    for (size_t i = 0; i < e->reads.size(); ++i) fetch_func(&e->reads[i], f);  // what samfetch does with each overlapping record
    bam_plbuf_push(0, buf);
    bam_plbuf_destroy(buf);
    free(f);
    std::cout.rdbuf(old);
    e->text = out.str(); e->warn_text = warn.str();
    e->n_lines = 0; for (size_t i = 0; i < e->text.size(); ++i) e->n_lines += e->text[i] == '\n';
    return BRC_OK;
}
int brc_fetch_result(brc_engine* e, brc_result* out) {
    if (!e || !out) return BRC_E_ARG;
    memset(out, 0, sizeof *out);
    out->tid = e->tid; out->beg0 = e->beg0; out->end = e->end; out->n_lib = e->cfg.per_lib ? e->cfg.n_libs : 1;
    return BRC_OK;
}
int brc_end_region(brc_engine* e, brc_result* out) { int rc = brc_compute(e, 0); return rc ? rc : brc_fetch_result(e, out); }
int brc_clear_indel_queue(brc_engine* e) { if (!e) return BRC_E_ARG; e->d.indel_queue_map.clear(); return BRC_OK; }   // :605
int brc_region_counts(brc_engine* e, uint64_t* n_events, uint64_t* n_positions) {
    if (!e) return BRC_E_ARG;
    if (n_events) *n_events = 0;
    if (n_positions) *n_positions = e->n_lines;
    return BRC_OK;
}
int brc_format_region(brc_engine* e, const brc_result*, const char* chrom, const char** text, size_t* text_len) {
    if (!e || !chrom || !text) return BRC_E_ARG;
    e->fmt_text.clear();
    for (size_t i = 0; i < e->text.size(); ++i) { if (e->text[i] == '\x01') e->fmt_text += chrom; else e->fmt_text += e->text[i]; }
    *text = e->fmt_text.c_str(); if (text_len) *text_len = e->fmt_text.size();
    return BRC_OK;
}
int brc_format_window(brc_engine*, const brc_result*, const char*, int32_t, int32_t, int32_t, const char**, size_t*) { return BRC_E_ARG; }

/* ---- extras (not part of include/brc.h) */

/* -w / --max-warnings of the next computes (:445,499) */
int bamrc_ref_set_max_warnings(brc_engine* e, int64_t n) { if (!e) return BRC_E_ARG; e->max_warnings = n; return BRC_OK; }
/* what WARN wrote to its stream during the last compute */
int bamrc_ref_warnings(brc_engine* e, const char** text, size_t* len) {
    if (!e || !text) return BRC_E_ARG;
    *text = e->warn_text.c_str(); if (len) *len = e->warn_text.size();
    return BRC_OK;
}

/* fetch_func on every read of `bt` (no pileup): zm_out[5 i .. 5 i + 5) = the five integers of the Zm tag it appended */
int bamrc_ref_annotate(const brc_read_batch* bt, const char* ref, int64_t ref_len, int ref_len_check, int32_t* zm_out) {
    pileup_data_t d = pileup_data_t();
    char* refz = 0;
    if (ref) { refz = (char*)malloc((size_t)ref_len + 1); memcpy(refz, ref, (size_t)ref_len); refz[ref_len] = 0; }
    fetch_data_t* f = (fetch_data_t*)calloc(1, sizeof(pileup_data_t));
    bam_plbuf_t* buf = bam_plbuf_init(noop_pileup, &d);
    f->pileup_buffer = buf; f->ref_pointer = &refz;
    if (ref_len_check && ref) { f->ref_len = (int)ref_len; f->seq_name = "ref"; }
    for (int64_t i = 0; i < bt->n_reads; ++i) {
        bam1_t b; build_bam1(&b, 0, bt, i, i, false);
        fetch_func(&b, f);
        uint8_t* z = bam_aux_get(&b, "Zm");
        if (!z) { free(b.data); bam_plbuf_destroy(buf); free(f); free(refz); return BRC_E_ARG; }
        const aux_zm_t zm = aux_zm_t::from_string((const char*)(z + 1));
        zm_out[5 * i] = zm.sum_of_mismatch_qualities; zm_out[5 * i + 1] = zm.clipped_length; zm_out[5 * i + 2] = zm.left_clip;
        zm_out[5 * i + 3] = zm.three_prime_index; zm_out[5 * i + 4] = zm.q2_pos;
        free(b.data);
    }
    bam_plbuf_push(0, buf); bam_plbuf_destroy(buf); free(f); free(refz);
    return BRC_OK;
}

/* BasicStat::process_read (BasicStat.cpp:28-107) over a list of (read index, qpos) events in order, then operator<<
 * (:110-159).  Reads are annotated by fetch_func first, unless with_zm == 0 (Zm-missing branch).  Returns the 13 raw
 * accumulators and the printed text (up to text_cap - 1 bytes). */
int bamrc_ref_basicstat(const brc_read_batch* bt, const char* ref, int64_t ref_len, int with_zm, int is_indel,
                        int64_t n_events, const int32_t* ev_read, const int32_t* ev_qpos,
                        brc_stat* stat_out, char* text_out, size_t text_cap, uint64_t* warn_counts /* [3]: SM, NM, Zm */) {
    pileup_data_t d = pileup_data_t();
    char* refz = 0;
    if (ref) { refz = (char*)malloc((size_t)ref_len + 1); memcpy(refz, ref, (size_t)ref_len); refz[ref_len] = 0; }
    fetch_data_t* f = (fetch_data_t*)calloc(1, sizeof(pileup_data_t));
    bam_plbuf_t* buf = bam_plbuf_init(noop_pileup, &d);
    f->pileup_buffer = buf; f->ref_pointer = &refz;
    std::vector<bam1_t> reads((size_t)bt->n_reads);
    for (int64_t i = 0; i < bt->n_reads; ++i) { build_bam1(&reads[i], 0, bt, i, i, false); if (with_zm) fetch_func(&reads[i], f); }
    std::ostringstream warn;
    WARN.reset(new ReadWarnings(warn, -1));
    BasicStat st(is_indel != 0);
    for (int64_t k = 0; k < n_events; ++k) {
        bam_pileup1_t pl; memset(&pl, 0, sizeof pl);
        pl.b = &reads[ev_read[k]]; pl.qpos = ev_qpos[k];
        st.process_read(&pl);
    }
    stat_out->i[BRC_I_N] = st.read_count; stat_out->i[BRC_I_SMQ] = st.sum_map_qualities; stat_out->i[BRC_I_SSE] = st.sum_single_ended_map_qualities;
    stat_out->i[BRC_I_PLUS] = st.num_plus_strand; stat_out->i[BRC_I_MINUS] = st.num_minus_strand; stat_out->i[BRC_I_NQ2] = st.num_q2_reads;
    stat_out->i[BRC_I_SMMQ] = st.sum_of_mismatch_qualities; stat_out->i[BRC_I_SCLIP] = st.sum_of_clipped_lengths; stat_out->i[BRC_I_SBQ] = st.sum_base_qualities;
    stat_out->f[BRC_F_SEV] = st.sum_event_location; stat_out->f[BRC_F_SQ2] = st.sum_q2_distance;
    stat_out->f[BRC_F_SNM] = st.sum_number_of_mismatches; stat_out->f[BRC_F_S3P] = st.sum_3p_distance;
    std::ostringstream os; os << st;
    if (text_out && text_cap) { const std::string s = os.str(); const size_t n = s.size() < text_cap - 1 ? s.size() : text_cap - 1; memcpy(text_out, s.data(), n); text_out[n] = 0; }
    if (warn_counts) {
        warn_counts[0] = warn_counts[1] = warn_counts[2] = 0;
        std::istringstream is(warn.str()); std::string line;
        while (std::getline(is, line)) {
            if (line.find("SM tag") != std::string::npos) warn_counts[0]++;
            else if (line.find("NM tag") != std::string::npos) warn_counts[1]++;
            else if (line.find("generated tag") != std::string::npos) warn_counts[2]++;
        }
    }
    for (size_t i = 0; i < reads.size(); ++i) free(reads[i].data);
    bam_plbuf_push(0, buf); bam_plbuf_destroy(buf); free(f); free(refz);
    return BRC_OK;
}

/* IndelQueue / IndelQueueEntry (IndelQueue.cpp:3-15, IndelQueueEntry.cpp:3-8): a queue handle for the unit KATs */
void* bamrc_ref_queue_new(void) { return new IndelQueue(); }
void bamrc_ref_queue_free(void* q) { delete (IndelQueue*)q; }
void bamrc_ref_queue_push(void* q, uint32_t tid, uint32_t pos, uint32_t read_count, int is_indel, const char* allele) {
    BasicStat st(is_indel != 0); st.read_count = read_count;
    ((IndelQueue*)q)->push(IndelQueueEntry(tid, pos, st, allele ? allele : ""));
}
size_t bamrc_ref_queue_size(void* q) { return ((IndelQueue*)q)->queue.size(); }
int bamrc_ref_queue_process(void* q, uint32_t tid, uint32_t pos, char* text_out, size_t text_cap) {
    std::ostringstream os;
    const int depth = ((IndelQueue*)q)->process(tid, pos, os);
    if (text_out && text_cap) { const std::string s = os.str(); const size_t n = s.size() < text_cap - 1 ? s.size() : text_cap - 1; memcpy(text_out, s.data(), n); text_out[n] = 0; }
    return depth;
}
/* aux_zm_t round trip (auxfields.hpp:13-34) */
int bamrc_ref_zm_roundtrip(const int32_t in[5], char* text_out, size_t text_cap, int32_t out[5]) {
    aux_zm_t zm; zm.sum_of_mismatch_qualities = in[0]; zm.clipped_length = in[1]; zm.left_clip = in[2]; zm.three_prime_index = in[3]; zm.q2_pos = in[4];
    const std::string s = zm.to_string();
    if (text_out && text_cap) { const size_t n = s.size() < text_cap - 1 ? s.size() : text_cap - 1; memcpy(text_out, s.data(), n); text_out[n] = 0; }
    const aux_zm_t z2 = aux_zm_t::from_string(s.c_str());
    out[0] = z2.sum_of_mismatch_qualities; out[1] = z2.clipped_length; out[2] = z2.left_clip; out[3] = z2.three_prime_index; out[4] = z2.q2_pos;
    return BRC_OK;
}
/* ReadWarnings (ReadWarnings.hpp:21-50): n warnings of each listed type with a cap; returns the stream text */
int bamrc_ref_readwarnings(int64_t max_per_type, int64_t n_rounds, const int32_t* types, const char* const* names, int n_types,
                           char* text_out, size_t text_cap) {
    std::ostringstream ss;
    ReadWarnings w(ss, max_per_type);
    for (int64_t r = 0; r < n_rounds; ++r) for (int k = 0; k < n_types; ++k) w.warn((ReadWarnings::WarningType)types[k], names[k]);
    const std::string s = ss.str();
    if (text_out && text_cap) { const size_t n = s.size() < text_cap - 1 ? s.size() : text_cap - 1; memcpy(text_out, s.data(), n); text_out[n] = 0; }
    return (int)s.size();
}

}  // extern "C"
