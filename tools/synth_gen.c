/*
 * synth_gen.c — deterministic synthetic WGS read generator for bench.py and the full-size parity tests
 * (bench/test infrastructure; not part of the product library).
 *
 * Data model: SURVEY.md section 8(d).  Reference bases iid uniform ACGT; N reads of fixed length with starts
 * stratified-uniform over the contig (the contig is cut into equal chunks, each chunk draws its share of starts
 * uniformly and sorts them, so the whole batch is coordinate-sorted and chunks can be generated in parallel);
 * strand 1/2; 98 % proper-pair flags {99,147,83,163}, 2 % {65,129,121,73}; MAPQ 90 % 60 / 10 % U[0,59];
 * QUAL clamp(round(N(33,6)),3,40) with a 3' run of Q2 (geometric, p = 0.1) on 5 % of reads; substitutions
 * p_sub per base; CIGAR: p_clip one soft clip U[1,20] at one end, p_ins one I, p_del one D (length U[1,indel_max]);
 * NM = mismatches + indel bases on every read; SM (= MAPQ) on 50 % of reads; library uniform over n_libs.
 *
 * PRNG: xoshiro256** seeded per chunk through splitmix64(seed, chunk) -> results do not depend on thread count.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int64_t contig_len;
    int64_t n_reads;
    int32_t read_len;
    int32_t n_libs;
    uint64_t seed;
    double p_sub, p_clip, p_ins, p_del;
    int32_t indel_max;
    int32_t n_chunks;      /* generation chunks (fixed by the caller so that output is thread-count independent) */
    /* mixed read lengths (adapter-trimmed / two run types in one file): a read is trimmed to U[trim_min, read_len - 1] with
     * probability p_trim, is long_len bases long with probability p_long, read_len otherwise.  Both 0: every read read_len
     * (and no extra draw: the fixed-length configurations generate the bytes they always did). */
    double p_trim, p_long;
    int32_t trim_min, long_len;
    /* NovaSeq-like records (round 6, the "novaseq" model): qual_bins != 0 -> qualities from the four RTA3 bins {2, 12, 23, 37} (2 %, 6 %, 12 %,
     * 80 %; the 3' Q2 run stays); p_dup / p_sec / p_supp: duplicate (0x400), secondary (0x100) and supplementary (0x800) records; p_mapq0:
     * multimappers with MAPQ 0.  All 0: the records the other models always had (and no extra draw). */
    int32_t qual_bins;
    double p_dup, p_sec, p_supp, p_mapq0;
    /* synth_reads_dense: eqx != 0 -> the M operators are written as runs of = and X (pbmm2 / minimap2 --eqx: X where the read's base differs
     * from the reference's) */
    int32_t eqx;
} synth_params;

typedef struct { uint64_t s[4]; } rng_t;
static inline uint64_t splitmix64(uint64_t* x) { uint64_t z = (*x += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t rng_next(rng_t* r) {
    const uint64_t res = rotl(r->s[1] * 5, 7) * 9, t = r->s[1] << 17;
    r->s[2] ^= r->s[0]; r->s[3] ^= r->s[1]; r->s[1] ^= r->s[2]; r->s[0] ^= r->s[3]; r->s[2] ^= t; r->s[3] = rotl(r->s[3], 45);
    return res;
}
static void rng_seed(rng_t* r, uint64_t seed, uint64_t stream) { uint64_t x = seed * 0x100000001b3ull + stream; for (int i = 0; i < 4; ++i) r->s[i] = splitmix64(&x); }
static inline double rng_u01(rng_t* r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }
static inline uint32_t rng_below(rng_t* r, uint32_t n) { return (uint32_t)(((rng_next(r) >> 32) * (uint64_t)n) >> 32); }

static const char BASES[4] = {'A', 'C', 'G', 'T'};
static inline uint8_t code_of(uint8_t c) { return c == 'A' ? 1 : c == 'C' ? 2 : c == 'G' ? 4 : 8; }

int synth_ref(uint8_t* ref, int64_t len, uint64_t seed) {
    const int64_t CH = 1 << 20;
    const int64_t nch = (len + CH - 1) / CH;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t c = 0; c < nch; ++c) {
        rng_t r; rng_seed(&r, seed, (uint64_t)c);
        const int64_t b = c * CH, e = b + CH < len ? b + CH : len;
        int64_t i = b;
        while (i < e) { uint64_t x = rng_next(&r); for (int k = 0; k < 32 && i < e; ++k, ++i, x >>= 2) ref[i] = (uint8_t)BASES[x & 3]; }
    }
    return 0;
}

static int cmp_i32(const void* a, const void* b) { int32_t x = *(const int32_t*)a, y = *(const int32_t*)b; return (x > y) - (x < y); }

/* 4096-quantile table of clamp(round(N(33,6)),3,40) */
static uint8_t QTAB[4096];
static void build_qtab(void) {
    /* inverse normal CDF by bisection on erf; done once */
    for (int i = 0; i < 4096; ++i) {
        const double u = (i + 0.5) / 4096.0;
        double lo = -8, hi = 8;
        for (int it = 0; it < 60; ++it) { const double m = 0.5 * (lo + hi); if (0.5 * (1.0 + erf(m / sqrt(2.0))) < u) lo = m; else hi = m; }
        double q = floor(33.0 + 6.0 * 0.5 * (lo + hi) + 0.5);
        if (q < 3) q = 3;
        if (q > 40) q = 40;
        QTAB[i] = (uint8_t)q;
    }
}

/* Arrays are caller-allocated: per-read arrays [n_reads]; cigar [3*n_reads] (stride 3 per read);
 * seq4 [n_reads * ceil(Lmax/2)]; qual [n_reads * Lmax], Lmax = max(read_len, long_len when p_long > 0): rows at a fixed stride. */
int synth_reads(const synth_params* P, const uint8_t* ref, int32_t* pos, uint16_t* flag, uint8_t* mapq, int16_t* lib, int32_t* l_qseq,
                uint32_t* n_cigar, uint64_t* cig_off, uint64_t* seq_off, uint64_t* qual_off, int32_t* nm, int32_t* sm, uint8_t* tags,
                uint32_t* cigar, uint8_t* seq4, uint8_t* qual) {
    const int mixed = (P->p_trim > 0 || P->p_long > 0);
    const int Lmax = (P->p_long > 0 && P->long_len > P->read_len) ? P->long_len : P->read_len, SB = (Lmax + 1) / 2;
    const int nch = P->n_chunks > 0 ? P->n_chunks : 64;
    if (P->read_len < 30 || Lmax > 100000 || P->contig_len < 4 * Lmax) return -1;
    if (mixed && (P->trim_min < 30 || P->trim_min >= P->read_len || (P->p_long > 0 && P->long_len < 30))) return -1;
    build_qtab();
    const double lg1mp = P->p_sub > 0 ? log(1.0 - P->p_sub) : 0.0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int c = 0; c < nch; ++c) {
        rng_t r; rng_seed(&r, P->seed, (uint64_t)c + 1000003u);
        const int64_t r0 = P->n_reads * c / nch, r1 = P->n_reads * (c + 1) / nch;
        const int64_t span = P->contig_len - Lmax - 2 * P->indel_max - 1;
        const int64_t p0 = span * c / nch, p1 = span * (c + 1) / nch;
        for (int64_t i = r0; i < r1; ++i) pos[i] = (int32_t)(p0 + (int64_t)(rng_u01(&r) * (double)(p1 - p0)));
        qsort(pos + r0, (size_t)(r1 - r0), sizeof(int32_t), cmp_i32);
        uint8_t* codes = (uint8_t*)malloc((size_t)Lmax + 64);     /* (reads of up to 100 kb: not on the stack) */
        for (int64_t i = r0; i < r1; ++i) {
            const int rev = (int)(rng_next(&r) >> 63);
            const double uf = rng_u01(&r);
            uint16_t f;
            if (uf < 0.98) { const int first = (int)(rng_next(&r) >> 63); f = rev ? (first ? 83 : 147) : (first ? 99 : 163); }
            else { static const uint16_t odd[4] = {65, 129, 121, 73}; f = odd[rng_below(&r, 4)]; if (rev) f |= 16; else f &= (uint16_t)~16u; }
            const int nova = P->qual_bins != 0 || P->p_dup > 0 || P->p_sec > 0 || P->p_supp > 0 || P->p_mapq0 > 0;
            if (nova) { const double ux = rng_u01(&r); if (ux < P->p_dup) f |= 0x400; else if (ux < P->p_dup + P->p_sec) f |= 0x100; else if (ux < P->p_dup + P->p_sec + P->p_supp) f |= 0x800; }
            flag[i] = f;
            mapq[i] = (rng_u01(&r) < 0.9) ? 60 : (uint8_t)rng_below(&r, 60);
            if (nova && rng_u01(&r) < P->p_mapq0) mapq[i] = 0;
            lib[i] = (int16_t)(P->n_libs > 1 ? rng_below(&r, (uint32_t)P->n_libs) : 0);
            int L = P->read_len;
            if (mixed) { const double ul = rng_u01(&r); if (ul < P->p_trim) L = P->trim_min + (int)rng_below(&r, (uint32_t)(P->read_len - P->trim_min)); else if (ul < P->p_trim + P->p_long) L = P->long_len; }
            l_qseq[i] = L;
            cig_off[i] = (uint64_t)i * 3; seq_off[i] = (uint64_t)i * SB; qual_off[i] = (uint64_t)i * Lmax;
            uint32_t* cg = cigar + i * 3;
            /* CIGAR */
            const double uc = rng_u01(&r);
            int nmv = 0, ncg;
            int ins_at = -1, ins_len = 0, del_at = -1, del_len = 0, clipL = 0, clipR = 0;
            if (uc < P->p_clip) { const int s = 1 + (int)rng_below(&r, 20); if (rng_next(&r) >> 63) clipL = s; else clipR = s; }
            else if (uc < P->p_clip + P->p_ins) { ins_len = 1 + (int)rng_below(&r, (uint32_t)P->indel_max); ins_at = 10 + (int)rng_below(&r, (uint32_t)(L - 20 - ins_len)); }
            else if (uc < P->p_clip + P->p_ins + P->p_del) { del_len = 1 + (int)rng_below(&r, (uint32_t)P->indel_max); del_at = 10 + (int)rng_below(&r, (uint32_t)(L - 20)); }
            if (clipL) { cg[0] = ((uint32_t)clipL << 4) | 4; cg[1] = ((uint32_t)(L - clipL) << 4) | 0; ncg = 2; }
            else if (clipR) { cg[0] = ((uint32_t)(L - clipR) << 4) | 0; cg[1] = ((uint32_t)clipR << 4) | 4; ncg = 2; }
            else if (ins_len) { cg[0] = ((uint32_t)ins_at << 4) | 0; cg[1] = ((uint32_t)ins_len << 4) | 1; cg[2] = ((uint32_t)(L - ins_at - ins_len) << 4) | 0; ncg = 3; nmv += ins_len; }
            else if (del_len) { cg[0] = ((uint32_t)del_at << 4) | 0; cg[1] = ((uint32_t)del_len << 4) | 2; cg[2] = ((uint32_t)(L - del_at) << 4) | 0; ncg = 3; nmv += del_len; }
            else { cg[0] = ((uint32_t)L << 4) | 0; ncg = 1; }
            for (int k = ncg; k < 3; ++k) cg[k] = 0;
            n_cigar[i] = (uint32_t)ncg;
            /* bases: walk the alignment */
            int64_t rp = pos[i];
            int qp = 0;
            for (int k = 0; k < ncg; ++k) {
                const int op = cg[k] & 15, len = (int)(cg[k] >> 4);
                if (op == 0) { for (int j = 0; j < len; ++j) codes[qp + j] = code_of(ref[rp + j]); qp += len; rp += len; }
                else if (op == 1 || op == 4) { for (int j = 0; j < len; ++j) codes[qp + j] = (uint8_t)(1u << rng_below(&r, 4)); qp += len; }
                else if (op == 2) rp += len;
            }
            /* substitutions at geometric gaps (aligned bases only change NM when they differ from the reference) */
            if (P->p_sub > 0) {
                int q = (int)floor(log(1.0 - rng_u01(&r)) / lg1mp);
                while (q < L) {
                    const uint8_t old = codes[q];
                    uint8_t nw = (uint8_t)(1u << rng_below(&r, 4));
                    if (nw == old) nw = (uint8_t)(old == 8 ? 1 : old << 1);
                    codes[q] = nw;
                    const int aligned = (q >= clipL && q < L - clipR && !(ins_len && q >= ins_at && q < ins_at + ins_len));
                    if (aligned) nmv++;
                    q += 1 + (int)floor(log(1.0 - rng_u01(&r)) / lg1mp);
                }
            }
            uint8_t* s4 = seq4 + (uint64_t)i * SB;
            for (int j = 0; j + 1 < L; j += 2) s4[j >> 1] = (uint8_t)((codes[j] << 4) | codes[j + 1]);
            if (L & 1) s4[L >> 1] = (uint8_t)(codes[L - 1] << 4);
            /* qualities */
            uint8_t* qq = qual + (uint64_t)i * Lmax;
            if (P->qual_bins) {
                for (int j = 0; j < L; j += 8) {
                    uint64_t x = rng_next(&r);
                    for (int k = 0; k < 8 && j + k < L; ++k, x >>= 8) { const unsigned u = (unsigned)(x & 255); qq[j + k] = u < 5 ? 2 : u < 20 ? 12 : u < 51 ? 23 : 37; }
                }
            } else
            for (int j = 0; j < L; j += 5) {
                uint64_t x = rng_next(&r);
                for (int k = 0; k < 5 && j + k < L; ++k, x >>= 12) qq[j + k] = QTAB[x & 4095];
            }
            if (rng_u01(&r) < 0.05) {
                int run = 1 + (int)floor(log(1.0 - rng_u01(&r)) / log(0.9));
                if (run > L) run = L;
                if (rev) for (int j = 0; j < run; ++j) qq[j] = 2; else for (int j = 0; j < run; ++j) qq[L - 1 - j] = 2;
            }
            nm[i] = nmv;
            uint8_t t = 1;
            if (rng_next(&r) >> 63) { t |= 2; sm[i] = mapq[i]; } else sm[i] = 0;
            tags[i] = t;
        }
        free(codes);
    }
    return 0;
}

/*
 * Long reads with an operator every few bases (ONT / PacBio-CLR-like alignments): read lengths U[len_min, len_max], the CIGAR alternates
 * M segments of 1 + Geometric(mean op_gap) bases with an insertion or a deletion (50 / 50) of U[1, indel_max] bases until the read is
 * used up; 5 % of the reads start or end with a soft clip U[1, 200].  Everything else as in synth_reads (flags, MAPQ, qualities, SM,
 * substitutions at p_sub, NM = mismatches + indel bases).  Arrays are caller-allocated with fixed strides: cigar [n_reads * cig_stride],
 * seq4 [n_reads * ceil(len_max / 2)], qual [n_reads * len_max]; cig_off / seq_off / qual_off are set to the strided rows and n_cigar to
 * the operators used (the Python wrapper packs the CIGARs).  A read whose operators would pass cig_stride - 2 ends with one M.
 */
int synth_reads_dense(const synth_params* P, int32_t len_min, int32_t len_max, double op_gap, int32_t cig_stride, const uint8_t* ref,
                      int32_t* pos, uint16_t* flag, uint8_t* mapq, int16_t* lib, int32_t* l_qseq, uint32_t* n_cigar, uint64_t* cig_off, uint64_t* seq_off,
                      uint64_t* qual_off, int32_t* nm, int32_t* sm, uint8_t* tags, uint32_t* cigar, uint8_t* seq4, uint8_t* qual) {
    const int Lmax = len_max, SB = (Lmax + 1) / 2;
    const int nch = P->n_chunks > 0 ? P->n_chunks : 64;
    if (len_min < 64 || len_max < len_min || len_max > 100000 || cig_stride < 8 || op_gap < 2.0 || P->contig_len < 6 * (int64_t)Lmax) return -1;
    build_qtab();
    const double lg1mp = P->p_sub > 0 ? log(1.0 - P->p_sub) : 0.0, lgap = log(1.0 - 1.0 / op_gap);
#pragma omp parallel for schedule(dynamic, 1)
    for (int c = 0; c < nch; ++c) {
        rng_t r; rng_seed(&r, P->seed, (uint64_t)c + 2000003u);
        const int64_t r0 = P->n_reads * c / nch, r1 = P->n_reads * (c + 1) / nch;
        const int64_t span = P->contig_len - 2 * (int64_t)Lmax - 1;            /* (deletions make a read's reference span longer than its length) */
        const int64_t p0 = span * c / nch, p1 = span * (c + 1) / nch;
        for (int64_t i = r0; i < r1; ++i) pos[i] = (int32_t)(p0 + (int64_t)(rng_u01(&r) * (double)(p1 - p0)));
        qsort(pos + r0, (size_t)(r1 - r0), sizeof(int32_t), cmp_i32);
        uint8_t* codes = (uint8_t*)malloc((size_t)Lmax + 64);
        uint8_t* aligned = (uint8_t*)malloc((size_t)Lmax + 64);
        for (int64_t i = r0; i < r1; ++i) {
            const int rev = (int)(rng_next(&r) >> 63);
            flag[i] = rev ? 16 : 0;
            mapq[i] = (rng_u01(&r) < 0.9) ? 60 : (uint8_t)rng_below(&r, 60);
            lib[i] = (int16_t)(P->n_libs > 1 ? rng_below(&r, (uint32_t)P->n_libs) : 0);
            const int L = len_min + (int)rng_below(&r, (uint32_t)(len_max - len_min + 1));
            l_qseq[i] = L;
            cig_off[i] = (uint64_t)i * (uint64_t)cig_stride; seq_off[i] = (uint64_t)i * SB; qual_off[i] = (uint64_t)i * Lmax;
            uint32_t* cg = cigar + (uint64_t)i * (uint64_t)cig_stride;
            int ncg = 0, used = 0, nmv = 0;
            int64_t rp = pos[i];
            int clipL = 0, clipR = 0;
            if (rng_u01(&r) < 0.05) { const int s = 1 + (int)rng_below(&r, 200); if (rng_next(&r) >> 63) clipL = s; else clipR = s; }
            if (clipL) { cg[ncg++] = ((uint32_t)clipL << 4) | 4; for (int j = 0; j < clipL; ++j) { codes[used + j] = (uint8_t)(1u << rng_below(&r, 4)); aligned[used + j] = 0; } used += clipL; }
            const int body_end = L - clipR;
            while (used < body_end) {
                int m = 1 + (int)floor(log(1.0 - rng_u01(&r)) / lgap);
                if (m > body_end - used || ncg >= cig_stride - 3) m = body_end - used;
                if (rp + m >= P->contig_len) m = body_end - used;                  /* (cannot happen with the span above; keeps rp inside the contig in any case) */
                cg[ncg++] = ((uint32_t)m << 4) | 0;
                for (int j = 0; j < m; ++j) { codes[used + j] = code_of(ref[rp + j]); aligned[used + j] = 1; }
                used += m; rp += m;
                if (used >= body_end) break;
                const int k = 1 + (int)rng_below(&r, (uint32_t)P->indel_max);
                if ((rng_next(&r) >> 63) && body_end - used > k) {                      /* insertion (never the read's last operator before a clip) */
                    cg[ncg++] = ((uint32_t)k << 4) | 1;
                    for (int j = 0; j < k; ++j) { codes[used + j] = (uint8_t)(1u << rng_below(&r, 4)); aligned[used + j] = 0; }
                    used += k; nmv += k;
                } else { cg[ncg++] = ((uint32_t)k << 4) | 2; rp += k; nmv += k; }
            }
            if ((cg[ncg - 1] & 15u) != 0u) { /* ends on an indel: give it a last aligned base by shortening nothing — turn the operator into M of the next base */
                if ((cg[ncg - 1] & 15u) == 2u) { rp -= (int64_t)(cg[ncg - 1] >> 4); nmv -= (int)(cg[ncg - 1] >> 4); --ncg; }   /* a trailing deletion is dropped */
            }
            if (clipR) { cg[ncg++] = ((uint32_t)clipR << 4) | 4; for (int j = 0; j < clipR; ++j) { codes[used + j] = (uint8_t)(1u << rng_below(&r, 4)); aligned[used + j] = 0; } used += clipR; }
            for (int k = ncg; k < cig_stride && k < ncg + 2; ++k) cg[k] = 0;
            n_cigar[i] = (uint32_t)ncg;
            if (P->p_sub > 0) {
                int q = (int)floor(log(1.0 - rng_u01(&r)) / lg1mp);
                while (q < L) {
                    const uint8_t old = codes[q];
                    uint8_t nw = (uint8_t)(1u << rng_below(&r, 4));
                    if (nw == old) nw = (uint8_t)(old == 8 ? 1 : old << 1);
                    codes[q] = nw;
                    if (aligned[q]) nmv++;
                    q += 1 + (int)floor(log(1.0 - rng_u01(&r)) / lg1mp);
                }
            }
            if (P->eqx) {
                /* every M operator becomes runs of = (the base equals the reference's) and X (it differs); the row holds cig_stride operators */
                uint32_t* tmp = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)cig_stride);
                int no = 0, q = 0; int64_t rq = pos[i]; int full = 0;
                for (int k = 0; k < ncg && !full; ++k) {
                    const int op = (int)(cg[k] & 15u), len = (int)(cg[k] >> 4);
                    if (op != 0) { if (no >= cig_stride - 1) { full = 1; break; } tmp[no++] = cg[k]; if (op == 1 || op == 4) q += len; else if (op == 2) rq += len; continue; }
                    int j = 0;
                    while (j < len) {
                        const int eq = codes[q + j] == code_of(ref[rq + j]);
                        int e = j + 1;
                        while (e < len && (codes[q + e] == code_of(ref[rq + e])) == eq) ++e;
                        if (no >= cig_stride - 1) { full = 1; break; }
                        tmp[no++] = ((uint32_t)(e - j) << 4) | (eq ? 7u : 8u);
                        j = e;
                    }
                    q += len; rq += len;
                }
                if (!full) { for (int k = 0; k < no; ++k) cg[k] = tmp[k]; ncg = no; for (int k = ncg; k < cig_stride && k < ncg + 2; ++k) cg[k] = 0; n_cigar[i] = (uint32_t)ncg; }     /* (a row too short for the runs keeps its M operators) */
                free(tmp);
            }
            uint8_t* s4 = seq4 + (uint64_t)i * SB;
            for (int j = 0; j + 1 < L; j += 2) s4[j >> 1] = (uint8_t)((codes[j] << 4) | codes[j + 1]);
            if (L & 1) s4[L >> 1] = (uint8_t)(codes[L - 1] << 4);
            uint8_t* qq = qual + (uint64_t)i * Lmax;
            for (int j = 0; j < L; j += 5) {
                uint64_t x = rng_next(&r);
                for (int k = 0; k < 5 && j + k < L; ++k, x >>= 12) qq[j + k] = QTAB[x & 4095];
            }
            nm[i] = nmv;
            uint8_t t = 1;
            if (rng_next(&r) >> 63) { t |= 2; sm[i] = mapq[i]; } else sm[i] = 0;
            tags[i] = t;
        }
        free(codes); free(aligned);
    }
    return 0;
}
