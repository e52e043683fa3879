/*
 * ref_bound.cpp — INTEGRATION.md section 2, compiled: the REFERENCE'S OWN main() (bamreadcount.cpp:421-670, read in place,
 * not a line of it changed) with its hot path bound to the brc C-ABI of include/brc.h.
 *
 * TEST INFRASTRUCTURE (oracle/_ref/bam-readcount-bound-hip, -sim; never linked or loaded by the product).  What a maintainer
 * of bam-readcount would edit by hand is done here by redirecting the six call sites of main() before its source is
 * included:
 *
 *   bam_plbuf_init(pileup_func, &d)   :591,:650   ->  brc_create (first time) + brc_begin_region(tid, d.beg, d.end, d.ref, d.len)
 *   bam_plp_set_maxcnt(buf->iter, n)  :592,:651   ->  -d, handed to the engine unchanged (BRC_OPT_MAX_COUNT for n <= 0)
 *   samfetch(..., f, fetch_func)      :602,:654   ->  the same indexed fetch, the records batched into a brc_read_batch
 *                                                     (Batcher below: INTEGRATION.md's struct) instead of annotated one by one
 *   bam_plbuf_push(0, buf)            :603,:655   ->  brc_push_reads + brc_end_region + brc_format_region -> stdout,
 *                                                     brc_region_warnings -> the reference's own ReadWarnings object
 *   bam_plbuf_destroy(buf)            :604,:656   ->  nothing left to free
 *   d.indel_queue_map.clear()         :605        ->  + brc_clear_indel_queue
 *
 * fetch_func (:114-261), pileup_func (:265-419), BasicStat and IndelQueue are still compiled — and never called: everything
 * they did now happens behind the C-ABI (on the GPU when linked to libbrc_hip.so).  Option parsing, file opening, the
 * site-list loop, the region loop, messages and exit codes are the reference's.  tests/test_ref_compiled.py diffs stdout,
 * stderr and exit code of this program against the reference goldens and against the reference-compiled main().
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
// every header the reference's file includes, BEFORE the redirections below (their include guards then make the
// reference's own #include lines no-ops, so the macros only ever touch the text of bamreadcount.cpp itself)
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <cmath>
#include <cstddef>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#define private public                  // ReadWarnings::_max_count_per_type (-w lives in a local variable of main())
#include "bamrc/ReadWarnings.hpp"
#undef private
#include "bamrc/auxfields.hpp"
#include "version.h"
#include <boost/program_options.hpp>
#include "bamrc/BasicStat.hpp"
#include "bamrc/IndelQueueEntry.hpp"
#include "bamrc/IndelQueue.hpp"
#include "sam.h"
#include "header.h"
#include "htslib/faidx.h"
#include "htslib/khash.h"

#include "brc.h"

namespace bound {
// the shim's indexed fetch, captured before `samfetch` is redirected
static int (*const real_samfetch)(samfile_t*, const hts_idx_t*, int, int, int, void*, bam_fetch_f) = samfetch;
bam_plbuf_t* plbuf_init(void* pileup_data);
void set_maxcnt(int n);
int fetch(samfile_t* in, const hts_idx_t* idx, int tid, int beg, int end, void* fetch_data);
int push(const bam1_t* b, bam_plbuf_t* buf);
void destroy(bam_plbuf_t* buf);
void queue_cleared();
}  // namespace bound

#define bam_plbuf_init(func, data) bound::plbuf_init(data)
#define bam_plp_set_maxcnt(iter, n) bound::set_maxcnt(n)
#define samfetch(in, idx, tid, beg, end, data, func) bound::fetch(in, idx, tid, beg, end, data)
#define bam_plbuf_push(b, buf) bound::push(b, buf)
#define bam_plbuf_destroy(buf) bound::destroy(buf)
#define clear() clear(), bound::queue_cleared()
#define main brc_bound_main
#include "src/exe/bam-readcount/bamreadcount.cpp"
#undef main
#undef clear
#undef bam_plbuf_destroy
#undef bam_plbuf_push
#undef samfetch
#undef bam_plp_set_maxcnt
#undef bam_plbuf_init

namespace bound {

// decoded-read SoA batch in the BAM field layout (brc_read_batch): INTEGRATION.md section 2
struct Batcher {
    std::vector<int32_t> pos, l_qseq, nm, sm; std::vector<uint16_t> flag; std::vector<uint8_t> mapq, tags;
    std::vector<int16_t> lib; std::vector<uint32_t> n_cigar, cigar; std::vector<uint64_t> cig_off, seq_off, qual_off;
    std::vector<uint8_t> seq4, qual; std::vector<std::string> names; std::vector<const char*> name_ptr;
    void add(const bam1_t* b, int lib_index) {                       // called where fetch_func used to be (:114)
        pos.push_back(b->core.pos); flag.push_back((uint16_t)b->core.flag); mapq.push_back((uint8_t)b->core.qual);
        l_qseq.push_back(b->core.l_qseq); n_cigar.push_back(b->core.n_cigar); lib.push_back((int16_t)lib_index);
        cig_off.push_back(cigar.size()); seq_off.push_back(seq4.size()); qual_off.push_back(qual.size());
        cigar.insert(cigar.end(), bam1_cigar(b), bam1_cigar(b) + b->core.n_cigar);
        seq4.insert(seq4.end(), bam1_seq(b), bam1_seq(b) + (b->core.l_qseq + 1) / 2);
        qual.insert(qual.end(), bam1_qual(b), bam1_qual(b) + b->core.l_qseq);
        uint8_t t = 0; int32_t vnm = 0, vsm = 0;                      // the two tags process_read looks up (BasicStat.cpp:79,94)
        if (uint8_t* x = bam_aux_get(b, "NM")) { vnm = (int32_t)bam_aux2i(x); t |= BRC_TAG_NM; }
        if (uint8_t* x = bam_aux_get(b, "SM")) { vsm = (int32_t)bam_aux2i(x); t |= BRC_TAG_SM; }
        nm.push_back(vnm); sm.push_back(vsm); tags.push_back(t);
        names.push_back(bam1_qname(b));                               // read names: the warning text only (ReadWarnings.hpp:39-50)
    }
    brc_read_batch view() {
        name_ptr.clear(); for (size_t i = 0; i < names.size(); ++i) name_ptr.push_back(names[i].c_str());
        brc_read_batch v; memset(&v, 0, sizeof v); v.n_reads = (int64_t)pos.size();
        v.pos = pos.data(); v.flag = flag.data(); v.mapq = mapq.data(); v.lib = lib.data(); v.l_qseq = l_qseq.data();
        v.n_cigar = n_cigar.data(); v.cigar_off = cig_off.data(); v.seq_off = seq_off.data(); v.qual_off = qual_off.data();
        v.nm = nm.data(); v.sm = sm.data(); v.tags = tags.data(); v.cigar = cigar.data(); v.seq4 = seq4.data(); v.qual = qual.data();
        v.n_cigar_total = cigar.size(); v.seq_bytes = seq4.size(); v.qual_bytes = qual.size(); v.qname = name_ptr.empty() ? 0 : name_ptr.data();
        return v;
    }
};

struct State {
    brc_engine* eng; pileup_data_t* d; Batcher batch; std::vector<std::string> libs; int max_cnt; bool failed;
    bam_plbuf_t dummy;
    State() : eng(0), d(0), max_cnt(0), failed(false) { memset(&dummy, 0, sizeof dummy); }
};
static State S;

static void die(const char* what, int rc) {
    fprintf(stderr, "bam-readcount (bound): %s: %s (%s)\n", what, brc_strerror(rc), S.eng ? brc_last_error(S.eng) : "");
    fflush(stdout); _exit(1);
}

bam_plbuf_t* plbuf_init(void* pileup_data) {                          // where bam_plbuf_init stood (:591,:650)
    S.d = (pileup_data_t*)pileup_data;
    return &S.dummy;
}
static void create_engine(const fetch_data_t* f) {
    pileup_data_t* d = S.d;
    S.libs.assign(d->lib_names.begin(), d->lib_names.end());          // std::set: the bytewise order the reference prints libraries in (:273,360)
    std::vector<const char*> names; for (size_t i = 0; i < S.libs.size(); ++i) names.push_back(S.libs[i].c_str());
    brc_config cfg; memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = BRC_ABI_VERSION; cfg.min_mapq = d->min_mapq; cfg.min_bq = d->min_bq; cfg.max_cnt = d->max_cnt;
    cfg.per_lib = d->per_lib ? 1 : 0; cfg.insertion_centric = d->insertion_centric ? 1 : 0;
    cfg.n_libs = d->per_lib ? (int32_t)names.size() : 0; cfg.lib_names = (d->per_lib && !names.empty()) ? names.data() : 0;
    // site-list mode with a reference: fetch_func checks positions against the contig length (f->ref_len != 0, :594-600);
    // regions on the command line never set it (:652-653)
    cfg.ref_len_check = f->ref_len != 0 ? 1 : 0;
    const int rc = brc_create(&cfg, &S.eng);
    if (rc) { fprintf(stderr, "bam-readcount (bound): cannot create the engine: %s\n", brc_strerror(rc)); fflush(stdout); _exit(1); }
    if (S.max_cnt <= 0) brc_set_option(S.eng, BRC_OPT_MAX_COUNT, S.max_cnt);     // -d 0 / -d -3 reach the iterator as they are (:592,:651)
}
void set_maxcnt(int n) { S.max_cnt = n; }

static int batch_cb(const bam1_t* b, void* data) {                    // same signature samfetch expects (:114)
    State* s = (State*)data;
    int lib = 0;
    if (s->d->per_lib) {                                              // RG -> LB -> index in the sorted list, -1 if none (:280-284)
        const char* l = bam_get_library(s->d->in->header, b);
        lib = -1;
        if (l) for (size_t i = 0; i < s->libs.size(); ++i) if (s->libs[i] == l) { lib = (int)i; break; }
    }
    s->batch.add(b, lib);
    return 0;
}

int fetch(samfile_t* in, const hts_idx_t* idx, int tid, int beg, int end, void* fetch_data) {   // where samfetch stood (:602,:654)
    pileup_data_t* d = S.d;
    if (!S.eng) create_engine((const fetch_data_t*)fetch_data);
    // (the engine wants the region before the reads: d.beg / d.end / d.ref / d.len are all set by now)
    const int rc = brc_begin_region(S.eng, tid, d->beg, d->end, d->ref, d->ref ? d->len : 0);
    if (rc) die("brc_begin_region", rc);
    S.batch = Batcher();
    return real_samfetch(in, idx, tid, beg, end, &S, batch_cb);
}

int push(const bam1_t* b, bam_plbuf_t*) {                             // where bam_plbuf_push(0, buf) stood (:603,:655)
    if (b) { fprintf(stderr, "bam-readcount (bound): fetch_func was called\n"); abort(); }
    pileup_data_t* d = S.d;
    brc_read_batch v = S.batch.view();
    int rc = brc_push_reads(S.eng, &v);
    if (rc) die("brc_push_reads", rc);
    brc_result res; const char* text; size_t len;
    if ((rc = brc_end_region(S.eng, &res))) die("brc_end_region", rc);               // upload, device pipeline, download
    const char* chrom = d->in->header->target_name[d->tid];
    if ((rc = brc_format_region(S.eng, &res, chrom, &text, &len))) die("brc_format_region", rc);
    // the reference flushes every line (endl, :416): stdout before this region's warnings
    fwrite(text, 1, len, stdout); fflush(stdout);
    // the warnings of this region, through the reference's own ReadWarnings object (global -w counters, its text)
    const int64_t cap = WARN.get() ? WARN->_max_count_per_type : -1;
    const char* ev; size_t evlen;
    if ((rc = brc_region_warnings(S.eng, chrom, cap, &ev, &evlen))) die("brc_region_warnings", rc);
    for (size_t i = 0; i < evlen;) {
        const char* nl = (const char*)memchr(ev + i, '\n', evlen - i); const size_t e = nl ? (size_t)(nl - ev) : evlen;
        const char tag = ev[i]; const std::string body(ev + i + 2, e - (i + 2));
        if (tag == 'S') WARN->warn(ReadWarnings::SM_TAG_MISSING, body.c_str());
        else if (tag == 'N') WARN->warn(ReadWarnings::NM_TAG_MISSING, body.c_str());
        else if (tag == 'L') WARN->warn(ReadWarnings::LIBRARY_UNAVAILABLE, body.c_str());
        else if (tag == 'B') { fputs(body.c_str(), stderr); fputc('\n', stderr); }
        i = e + 1;
    }
    return 0;
}
void destroy(bam_plbuf_t*) {}
void queue_cleared() { if (S.eng) brc_clear_indel_queue(S.eng); }       // d.indel_queue_map.clear() (:605)

}  // namespace bound

int main(int argc, char* argv[]) {
    const int rc = brc_bound_main(argc, argv);
    fflush(stdout);
    if (bound::S.eng) brc_destroy(bound::S.eng);
    return rc;
}
