// abi.hip -- version / status entry points of the C ABI (include/kbnet_hip.h), the debug-knob table
// (environment read once at load) and small per-device caches shared by the launch code.
#include <stdlib.h>
#include <string.h>

#include "kbn_common.h"

namespace kbn {

KnobValue g_knobs[KNOB_COUNT];

static const char* const kKnobNames[KNOB_COUNT] = {
    "KBN_DEBUG", "KBN_FORCE_MW", "KBN_FORCE_TWB", "KBN_EPI_LDS", "KBN_NO_WINO",
    "KBN_NO_UP2X9", "KBN_NO_UP2X3", "KBN_WINO_RT",
    "KBN_NO_KB_PAIR", "KBN_NO_KB_DEPTH_FUSION", "KBN_PAIR_CAND", "KBN_S2D_DEBUG", "KBN_AUTOTUNE",
    "KBN_NO_HEAD_FUSION", "KBN_NO_SPLIT", "KBN_NO_OVERLAP", "KBN_NO_PAIR", "KBN_NO_PAIR_MID", "KBN_NO_PAIR_ENC",
    "KBN_NO_PAIR_TAIL", "KBN_NO_DEPTH_FRONT_FUSION", "KBN_FP16_ONE_TERM", "KBN_DEPTH_FRONT_FUSION", "KBN_NO_FRONT_NEXT"};

void tune_reload_env();   // tune.hip

static void load_knobs() {
    for (int k = 0; k < KNOB_COUNT; ++k) {
        const char* v = getenv(kKnobNames[k]);
        g_knobs[k].set = (v && *v) ? 1 : 0;
        int val = 0;
        if (v && *v) {
            char* end = nullptr;
            const long parsed = strtol(v, &end, 10);
            // "true" / "yes" / "on": set, not a number -> 1 (the pre-kbn_knob host switches treated any non-empty value but "0" as set)
            val = (end == v) ? 1 : (int)parsed;
        }
        g_knobs[k].value = val;
    }
}

namespace {
struct KnobInit {
    KnobInit() { load_knobs(); }
} g_knob_init;   // runs when the shared library is loaded
}  // namespace

int device_cu_count() {
    static std::atomic<int> cache[256];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    std::atomic<int>& slot = cache[dev & 255];
    int v = slot.load(std::memory_order_relaxed);
    if (v > 0) return v;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) return -1;
    slot.store(v, std::memory_order_relaxed);
    return v;
}

}  // namespace kbn

extern "C" {

int kbn_version(void) { return KBN_ABI_VERSION; }

const char* kbn_status_string(int status) {
    switch (status) {
        case KBN_OK: return "ok";
        case KBN_ERR_INVALID_ARGUMENT: return "invalid argument";
        case KBN_ERR_UNSUPPORTED: return "configuration outside the kernel limits";
        case KBN_ERR_WORKSPACE: return "buffer too small";
        case KBN_ERR_LAUNCH: return "HIP launch error";
        default: return "unknown status";
    }
}

int kbn_knob(const char* name) {
    if (!name) return 0;
    for (int k = 0; k < kbn::KNOB_COUNT; ++k)
        if (!strcmp(name, kbn::kKnobNames[k])) return kbn::g_knobs[k].value;
    return 0;
}

void kbn_reload_env(void) {
    kbn::load_knobs();
    kbn::tune_reload_env();
}

}  // extern "C"
