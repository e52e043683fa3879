// Exhaustive check (test infrastructure) that the reciprocal-based division and the quotient tables used by the device
// accumulate are bit-identical to fp32 `/` for every (numerator, denominator) the path can form; see brc_core.h div_rcp.
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
    srand(7);
    for (long it = 0; it < 20000000; ++it, ++tot) {         // long reads, sampled
        const int L = 1 + rand() % 3000000, n = rand() % (L + 1);
        const float Lf = (float)L;
        if (bits((float)n / Lf) != bits(brc::div_rcp((float)n, Lf, 1.0f / Lf))) ++bad;
    }
    printf("%ld %ld\n", tot, bad);
    return bad != 0;
}
