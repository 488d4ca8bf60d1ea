// tune.hip -- OPT-IN first-use tuning of launch geometry (tile shape of the direct convs, region shape of
// the Winograd kernel, tile shape of the up-convs).
//
// By default every ABI call launches the geometry its analytic cost model picks (or a choice cached by an
// earlier tuning pass / a preloaded KBN_TUNE_CACHE file) and returns: nothing is timed and nothing
// synchronises.  The cost models are within 10-15 % of the best geometry on some layers (workgroup-round
// quantisation, HBM-bound stores; profiles/r01/v6_tile_sweep.txt), so a caller that wants the last few
// percent switches tuning on around a warm-up pass:
//
//     kbn_set_autotune(1);  <one forward of every problem shape>;  kbn_set_autotune(0);
//
// While it is on, the first launch of a problem shape times every candidate on the caller's stream (HIP
// events; the output is simply rewritten with identical values -- geometry never changes an accumulation
// order) and the winner is cached per device for the process.  Never while the stream is being captured.
// The cache holds integers only.  KBN_TUNE_CACHE=<file> (read when the library is loaded / kbn_reload_env)
// preloads choices that apply to every device and receives the entries later tuning passes find, so that
// e.g. a profiled run replays the choices of an earlier run without any timing launches of its own.  The file
// starts with a line naming the candidate-table version it was written for (kTuneTables: bump it whenever a
// kernel family's candidate list changes); a file with another version -- or none -- is ignored, and every
// lookup is range-checked against the caller's candidate count, so a stale file can never select a geometry
// that does not exist.
#include <stdio.h>
#include <stdlib.h>
#include <sys/file.h>

#include <map>
#include <mutex>
#include <string>
#include <utility>

#include "conv_common.h"

namespace kbn {

typedef std::pair<int, TuneKey> DevKey;          // device ordinal (-1 = any: entries from the cache file)
static std::map<DevKey, int> g_cache;
static std::mutex g_mutex;
static std::string g_path;
static std::atomic<int> g_autotune{0};
constexpr int kTuneTables = 3;   // candidate tables of round 3 (KBN_ABI_VERSION is checked beside it)
static const char kHeaderFmt[] = "kbn-tune-cache abi %d tables %d\n";

static void load_file_locked() {   // g_mutex held
    if (g_path.empty()) return;
    FILE* f = fopen(g_path.c_str(), "r");
    if (!f) return;
    int abi = -1, tables = -1;
    if (fscanf(f, "kbn-tune-cache abi %d tables %d", &abi, &tables) != 2 || abi != KBN_ABI_VERSION || tables != kTuneTables) {
        fclose(f);   // another build's file (or not a cache file): its candidate indices mean nothing here
        fprintf(stderr, "[kbnet] KBN_TUNE_CACHE %s was not written by this build: ignored\n", g_path.c_str());
        g_path.clear();
        return;
    }
    TuneKey k;
    int cand;
    while (fscanf(f, "%d %d %d %d %d %d %d %d %d %d %d", &k[0], &k[1], &k[2], &k[3], &k[4], &k[5], &k[6], &k[7], &k[8],
                  &k[9], &cand) == 11)
        g_cache[DevKey(-1, k)] = cand;
    fclose(f);
}

static void append_file_locked(const TuneKey& k, int cand) {
    if (g_path.empty()) return;
    FILE* f = fopen(g_path.c_str(), "a");
    if (!f) return;
    // several ranks of one job may share the file (bench.py --gpus N: one process per GPU): the "new file -> header"
    // test and the entry go out under an exclusive lock, as ONE write each (the stream is flushed before the unlock)
    const bool locked = flock(fileno(f), LOCK_EX) == 0;
    fseek(f, 0, SEEK_END);
    if (ftell(f) == 0) fprintf(f, kHeaderFmt, KBN_ABI_VERSION, kTuneTables);   // new file
    fprintf(f, "%d %d %d %d %d %d %d %d %d %d %d\n", k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7], k[8], k[9], cand);
    fflush(f);
    if (locked) (void)flock(fileno(f), LOCK_UN);
    fclose(f);
}

void tune_reload_env() {   // library load and kbn_reload_env(): the only places the environment is read
    std::lock_guard<std::mutex> g(g_mutex);
    const char* path = getenv("KBN_TUNE_CACHE");
    const std::string p = (path && *path) ? path : "";
    if (p != g_path) {
        g_path = p;
        load_file_locked();
    }
    const char* a = getenv("KBN_AUTOTUNE");
    if (a && *a) g_autotune.store(atoi(a) != 0, std::memory_order_relaxed);
}

namespace {
struct TuneInit {
    TuneInit() { tune_reload_env(); }
} g_tune_init;
}  // namespace

bool tune_enabled() { return g_autotune.load(std::memory_order_relaxed) != 0; }

static int current_device() {
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess ? dev : 0;
}

bool tune_lookup(const TuneKey& key, int* cand, int ncand) {
    const int dev = current_device();
    std::lock_guard<std::mutex> g(g_mutex);
    auto it = g_cache.find(DevKey(dev, key));
    if (it == g_cache.end()) it = g_cache.find(DevKey(-1, key));
    if (it == g_cache.end() || it->second < 0 || it->second >= ncand) return false;   // out of range: the cost model serves
    *cand = it->second;
    return true;
}

int tune_pick(const TuneKey& key, int ncand, int model, const std::function<int(int)>& launch, hipStream_t stream) {
    int cand = model;
    if (tune_lookup(key, &cand, ncand)) return cand;
    if (!tune_enabled()) return model;
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
        const int dev = current_device();
        std::lock_guard<std::mutex> g(g_mutex);
        g_cache[DevKey(dev, key)] = best;
        append_file_locked(key, best);
    }
    return best;
}

}  // namespace kbn

extern "C" {

int kbn_set_autotune(int enabled) {
    return kbn::g_autotune.exchange(enabled ? 1 : 0, std::memory_order_relaxed);
}

int kbn_get_autotune(void) { return kbn::g_autotune.load(std::memory_order_relaxed); }

}  // extern "C"
