// tune.hip -- first-use tuning of launch geometry (tile shape of the direct convs, region shape of the
// Winograd kernel, tile shape of the up-convs).
//
// The analytic cost models are within 10-15 % of the best geometry on some layers (workgroup-round
// quantisation, HBM-bound stores; profiles/r01/v6_tile_sweep.txt).  The first launch of a problem
// shape therefore times every candidate on the caller's stream (HIP events; the output is simply
// rewritten with identical values -- geometry never changes an accumulation order) and the winner is
// cached for the process.  No tuning while the stream is being captured into a graph (the model's
// choice is used and nothing is cached) or with KBN_AUTOTUNE=0.  The cache holds integers only.
// KBN_TUNE_CACHE=<file> makes it persistent: entries are read at first use and appended as they are
// found (one line of 11 integers each), so that e.g. a profiled run replays the choices of an earlier
// run without any timing launches of its own.
#include <stdio.h>
#include <stdlib.h>

#include <map>
#include <mutex>

#include "conv_common.h"

namespace kbn {

static std::map<TuneKey, int> g_cache;
static std::mutex g_mutex;
static bool g_loaded = false;

static void load_file_locked() {   // g_mutex held
    if (g_loaded) return;
    g_loaded = true;
    const char* path = getenv("KBN_TUNE_CACHE");
    if (!path || !*path) return;
    FILE* f = fopen(path, "r");
    if (!f) return;
    TuneKey k;
    int cand;
    while (fscanf(f, "%d %d %d %d %d %d %d %d %d %d %d", &k[0], &k[1], &k[2], &k[3], &k[4], &k[5], &k[6], &k[7], &k[8],
                  &k[9], &cand) == 11)
        g_cache[k] = cand;
    fclose(f);
}

static void append_file_locked(const TuneKey& k, int cand) {
    const char* path = getenv("KBN_TUNE_CACHE");
    if (!path || !*path) return;
    FILE* f = fopen(path, "a");
    if (!f) return;
    fprintf(f, "%d %d %d %d %d %d %d %d %d %d %d\n", k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7], k[8], k[9], cand);
    fclose(f);
}

bool tune_enabled() {
    static const bool on = !(getenv("KBN_AUTOTUNE") && atoi(getenv("KBN_AUTOTUNE")) == 0);
    return on;
}

bool tune_lookup(const TuneKey& key, int* cand) {
    std::lock_guard<std::mutex> g(g_mutex);
    load_file_locked();
    auto it = g_cache.find(key);
    if (it == g_cache.end()) return false;
    *cand = it->second;
    return true;
}

int tune_pick(const TuneKey& key, int ncand, int model, const std::function<int(int)>& launch, hipStream_t stream) {
    if (!tune_enabled()) return model;
    int cand = model;
    if (tune_lookup(key, &cand)) return cand;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return model;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess) return model;
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return model; }
    // Two passes over the candidates; each sample is the average of 3 back-to-back launches (a single
    // launch's time flatters geometries whose first wave of workgroups finds everything in cache), the
    // score the better of the two samples.  The model's choice is kept unless something is >= 3 % faster.
    float score[64];
    bool valid[64];
    if (ncand > 64) ncand = 64;
    for (int c = 0; c < ncand; ++c) {
        score[c] = 1e30f;
        valid[c] = launch(c) == KBN_OK;   // warm (kernel attributes, caches); also filters invalid candidates
    }
    for (int pass = 0; pass < 2; ++pass) {
        for (int c = 0; c < ncand; ++c) {
            if (!valid[c]) continue;
            (void)hipEventRecord(e0, stream);
            int rc = KBN_OK;
            for (int rep = 0; rep < 3 && rc == KBN_OK; ++rep) rc = launch(c);
            (void)hipEventRecord(e1, stream);
            float t = 1e30f;
            if (rc != KBN_OK || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&t, e0, e1) != hipSuccess) t = 1e30f;
            score[c] = t < score[c] ? t : score[c];
        }
    }
    int best = model;
    float best_ms = (model >= 0 && model < ncand && valid[model]) ? score[model] * 0.97f : 1e30f;
    for (int c = 0; c < ncand; ++c)
        if (valid[c] && c != model && score[c] < best_ms) { best_ms = score[c]; best = c; }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (best_ms < 1e29f) {
        std::lock_guard<std::mutex> g(g_mutex);
        g_cache[key] = best;
        append_file_locked(key, best);
    }
    return best;
}

}  // namespace kbn
