// Exhaustive check (test infrastructure) that the reciprocal-based division and the quotient tables used by the device
// accumulate are bit-identical to fp32 `/` for every (numerator, denominator) the path can form; see brc_core.h div_rcp.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../bam_readcount_amd/csrc/brc_core.h"
static unsigned bits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
int main() {
    long bad = 0, tot = 0;
    for (int L = 1; L <= 3000; ++L) {                       // |qpos - x| / l_qseq
        const float Lf = (float)L, y = 1.0f / Lf;
        for (int n = 0; n <= 2 * L + 2; ++n, ++tot)
            if (bits((float)n / Lf) != bits(brc::div_rcp((float)n, Lf, y))) ++bad;
    }
    for (int cl = 1; cl <= 3000; ++cl) {                    // |k - cl/2| / (cl/2), half-integers
        const float c = (float)cl * 0.5f, y = 1.0f / c;
        for (int k = -cl - 2; k <= 2 * cl + 2; ++k, ++tot) {
            float d = (float)k - c; d = d < 0 ? -d : d;
            if (bits(d / c) != bits(brc::div_rcp(d, c, y))) ++bad;
            // table identity: |k - cl/2| / (cl/2) == |2k - cl| / cl  (what event_terms_tab looks up)
            if (k >= 0 && k < cl) { const unsigned m = brc::absdiff_u(2u * (unsigned)k, (unsigned)cl); if (bits(d / c) != bits((float)m / (float)cl)) ++bad; }
        }
    }
    for (int cl = 1; cl <= 512; ++cl) {                     // the event-location term of a soft-clipped read as k_pileup2 divides it out (PF_TABQ)
        const float c = (float)cl * 0.5f;
        const int lefts[4] = {0, 1, 37, 511};
        for (int li = 0; li < 4; ++li)
            for (int k = 0; k < cl; ++k, ++tot) {
                float d = (float)k - c; d = d < 0 ? -d : d;
                const double want = 1.0 - (double)(d / c);                 // event_terms (BasicStat.cpp:69-70)
                const double got = brc::tabq_sev(lefts[li] + k, lefts[li], (uint32_t)cl);
                if (memcmp(&want, &got, 8) != 0) ++bad;
            }
    }
    // div_small (brc_core.h): what the device computes — q0 = n*y, e = fma(-q0, m, n), q = fma(e, y, q0) — with the reciprocal the hardware
    // hands over.  v_rcp_f32 promises 1 ulp; the argument in brc_core.h holds for any y that close, so the whole domain is walked with
    // RN(1/m) and its neighbours up to 2 ulp either side (the engine checks the real instruction over the same domain when it is created).
    for (int m = 1; m < brc::DIV_SMALL_M; ++m) {
        const float mf = (float)m, y0 = 1.0f / mf;
        for (int du = -2; du <= 2; ++du) {
            unsigned yb = bits(y0) + (unsigned)du; float y; memcpy(&y, &yb, 4);
            for (int n = 0; n < brc::DIV_SMALL_N; ++n, ++tot) {
                const float nf = (float)n, q0 = nf * y, e = fmaf(-q0, mf, nf), q = fmaf(e, y, q0);
                if (bits(q) != bits(nf / mf)) ++bad;
            }
        }
    }
    // a read of another length (PF_DIV): the integer form against the reference's own expressions (BasicStat.cpp:60-70)
    for (int L = 1; L <= 255; L += (L < 40 ? 1 : 7))
        for (int cl = 1; cl <= L; cl += (cl < 20 ? 1 : 5))
            for (int left = 0; left + cl <= L; left += 3)
                for (int tp = 0; tp < L; tp += 11)
                    for (int qpos = 0; qpos < L; ++qpos, ++tot) {
                        const brc::EvTerms t = brc::piece_terms_inlane(brc::PF_Q2OK, (uint32_t)tp | ((uint32_t)L << 8) | ((uint32_t)left << 16), (uint32_t)cl, (uint32_t)qpos);
                        const float center = (float)cl * 0.5f; float d = (float)(qpos - left) - center; d = d < 0 ? -d : d;
                        const float s3p = (float)(qpos > tp ? qpos - tp : tp - qpos) / (float)L; const double sev = 1.0 - (double)(d / center);
                        if (bits(t.s3p) != bits(s3p) || bits(t.q2) != bits(s3p) || memcmp(&sev, &t.sev, 8) != 0) ++bad;
                    }
    srand(7);
    for (long it = 0; it < 20000000; ++it, ++tot) {         // long reads, sampled
        const int L = 1 + rand() % 3000000, n = rand() % (L + 1);
        const float Lf = (float)L;
        if (bits((float)n / Lf) != bits(brc::div_rcp((float)n, Lf, 1.0f / Lf))) ++bad;
    }
    printf("%ld %ld\n", tot, bad);
    return bad != 0;
}
