// brc_knobs.cpp — the one translation unit that differs between the product library and the test-knobs build (brc_host.h: TestKnob).
#include <stdlib.h>

#include "brc_host.h"

namespace brc {
#ifdef BRC_TEST_KNOBS
const char* test_knob(int which) {
    static const char* const kNames[TK_N] = {"BRC_NO_TABLE", "BRC_FLUSH_K", "BRC_PACK_LIM", "BRC_FORCE_DOM", "BRC_IBUCKET_SHIFT", "BRC_XEV_CAP", "BRC_DEVICE_TEXT_LIMIT", "BRC_FORMAT_THREADS", "BRC_FORMAT_CHUNK", "BRC_COMPACT_TILES", "BRC_WAVE_FORM"};
    return (which >= 0 && which < TK_N) ? getenv(kNames[which]) : nullptr;
}
#else
const char* test_knob(int) { return nullptr; }
#endif
}  // namespace brc
