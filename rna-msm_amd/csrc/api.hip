// Error plumbing, version and device probes of the C ABI.
#include "common.h"
#include <string.h>
#include <mutex>
#include <vector>

namespace rnamsm {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// ---------------------------------------------------------------------------------- kernel timing
namespace {
struct Record {
    int category;
    double flops, bytes, mfma_s, hbm_s, valu_s;
    hipEvent_t start, stop;
};
struct Totals {
    long long launches = 0;
    double ms = 0, flops = 0, bytes = 0;
    double bound_ms = 0, mfma_ms = 0, hbm_ms = 0;      // sums over launches of max(mfma, hbm, valu), mfma, hbm time at the peaks
    double valu_ms = 0;                                // ... and of the vector-ALU issue time (softmax-bound attention kernels)
};
std::mutex g_tmu;
bool g_ton = false;
std::vector<Record> g_pending;
std::vector<hipEvent_t> g_free;
Totals g_tot[TC_COUNT];
const char* const g_tnames[TC_COUNT] = {"gemm_f32", "row_logits", "softmax_rows", "row_apply",
                                        "col_attn", "layernorm", "embed_ln", "pack_outputs"};
hipEvent_t take_event() {
    if (!g_free.empty()) {
        hipEvent_t e = g_free.back();
        g_free.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

bool timing_enabled() { return g_ton; }
void timing_begin(int category, double flops, double bytes, double mfma_s, double hbm_s, double valu_s, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_tmu);
    Record r{category, flops, bytes, mfma_s, hbm_s, valu_s, take_event(), take_event()};
    (void)hipEventRecord(r.start, stream);
    g_pending.push_back(r);
}
void timing_end(hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_tmu);
    if (!g_pending.empty()) (void)hipEventRecord(g_pending.back().stop, stream);
}
}  // namespace rnamsm

extern "C" int rnamsm_timing_enable(int on) {
    std::lock_guard<std::mutex> lk(rnamsm::g_tmu);
    rnamsm::g_ton = on != 0;
    return RNAMSM_OK;
}
extern "C" int rnamsm_timing_collect(void) {
    using namespace rnamsm;
    std::lock_guard<std::mutex> lk(g_tmu);
    for (Record& r : g_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(r.stop) == hipSuccess && hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
            Totals& t = g_tot[r.category];
            t.launches += 1;
            t.ms += ms;
            t.flops += r.flops;
            t.bytes += r.bytes;
            const double two = r.mfma_s > r.hbm_s ? r.mfma_s : r.hbm_s;
            t.bound_ms += 1e3 * (two > r.valu_s ? two : r.valu_s);
            t.mfma_ms += 1e3 * r.mfma_s;
            t.hbm_ms += 1e3 * r.hbm_s;
            t.valu_ms += 1e3 * r.valu_s;
        }
        g_free.push_back(r.start);
        g_free.push_back(r.stop);
    }
    g_pending.clear();
    return TC_COUNT;
}
extern "C" int rnamsm_timing_get(int category, const char** name, long long* launches, double* ms, double* flops,
                                 double* bytes) {
    using namespace rnamsm;
    if (category < 0 || category >= TC_COUNT) return fail(RNAMSM_ERR_INVALID, "timing_get: bad category %d", category);
    std::lock_guard<std::mutex> lk(g_tmu);
    if (name) *name = g_tnames[category];
    if (launches) *launches = g_tot[category].launches;
    if (ms) *ms = g_tot[category].ms;
    if (flops) *flops = g_tot[category].flops;
    if (bytes) *bytes = g_tot[category].bytes;
    return RNAMSM_OK;
}
extern "C" int rnamsm_timing_get_bound(int category, double* bound_ms, double* mfma_ms, double* hbm_ms) {
    using namespace rnamsm;
    if (category < 0 || category >= TC_COUNT) return fail(RNAMSM_ERR_INVALID, "timing_get_bound: bad category %d", category);
    std::lock_guard<std::mutex> lk(g_tmu);
    if (bound_ms) *bound_ms = g_tot[category].bound_ms;
    if (mfma_ms) *mfma_ms = g_tot[category].mfma_ms;
    if (hbm_ms) *hbm_ms = g_tot[category].hbm_ms;
    return RNAMSM_OK;
}
extern "C" int rnamsm_timing_get_valu_bound(int category, double* valu_ms) {
    using namespace rnamsm;
    if (category < 0 || category >= TC_COUNT) return fail(RNAMSM_ERR_INVALID, "timing_get_valu_bound: bad category %d", category);
    std::lock_guard<std::mutex> lk(g_tmu);
    if (valu_ms) *valu_ms = g_tot[category].valu_ms;
    return RNAMSM_OK;
}
extern "C" void rnamsm_timing_reset(void) {
    using namespace rnamsm;
    std::lock_guard<std::mutex> lk(g_tmu);
    for (Totals& t : g_tot) t = Totals();
}

namespace rnamsm {
Tuning& tuning() {
    static Tuning t;
    return t;
}
std::shared_mutex& tuning_lock() {
    static std::shared_mutex m;
    return m;
}
int& gemm16_big_rows_override() {
    static thread_local int v = 0;
    return v;
}
}  // namespace rnamsm
extern "C" int rnamsm_set_param(const char* name, int value) {
    std::unique_lock<std::shared_mutex> exclusive(rnamsm::tuning_lock(), std::try_to_lock);
    if (!exclusive.owns_lock())
        return rnamsm::fail(RNAMSM_ERR_INVALID, "set_param: a forward driver is enqueuing on another thread; knobs are process-global "
                                                "and may only change between forwards");
    if (name && !strcmp(name, "gemm16_dma")) {
        rnamsm::tuning().gemm16_dma = value;
        return RNAMSM_OK;
    }
    if (name && !strcmp(name, "gemm_group")) {
        if (value < 0 || value > 64) return rnamsm::fail(RNAMSM_ERR_INVALID, "set_param: gemm_group must be in [0, 64]");
        rnamsm::tuning().gemm_group = value;
        return RNAMSM_OK;
    }
    if (name && !strcmp(name, "gemm_tile")) {
        if (value < 0 || value > 4) return rnamsm::fail(RNAMSM_ERR_INVALID, "set_param: gemm_tile must be in [0, 4]");
        rnamsm::tuning().gemm_tile = value;
        return RNAMSM_OK;
    }
    if (name && !strcmp(name, "row_narrow")) {
        rnamsm::tuning().row_narrow = value != 0;
        return RNAMSM_OK;
    }
    if (name && !strcmp(name, "attn16")) {
        rnamsm::tuning().attn16 = value;
        return RNAMSM_OK;
    }
    if (name && !strcmp(name, "greedy_fused")) {
        rnamsm::tuning().greedy_fused = value < 0 ? 0 : (value > 2 ? 2 : value);
        return RNAMSM_OK;
    }
    if (name && !strcmp(name, "ln_fold")) {
        rnamsm::tuning().ln_fold = value < 0 ? 0 : (value > 3 ? 3 : value);
        return RNAMSM_OK;
    }
    if (name && !strcmp(name, "gemm_splitk")) {
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8)
            return rnamsm::fail(RNAMSM_ERR_INVALID, "set_param: gemm_splitk must be 0, 1, 2, 4 or 8");
        rnamsm::tuning().gemm_splitk = value;
        return RNAMSM_OK;
    }
    if (name && !strcmp(name, "gemm16_mfma16")) {
        rnamsm::tuning().gemm16_mfma16 = value < 0 ? 0 : (value > 2 ? 2 : value);
        return RNAMSM_OK;
    }
    if (name && !strcmp(name, "row16_max_rows")) {
        if (value < 0 || value > 1024) return rnamsm::fail(RNAMSM_ERR_INVALID, "set_param: row16_max_rows must be in [0, 1024]");
        rnamsm::tuning().row16_max_rows = value;
        return RNAMSM_OK;
    }
    if (name && !strcmp(name, "col_small")) {
        rnamsm::tuning().col_small = value != 0;
        return RNAMSM_OK;
    }
    if (name && !strcmp(name, "col_dma")) {
        rnamsm::tuning().col_dma = value < 0 ? -1 : (value != 0);
        return RNAMSM_OK;
    }
    if (name && !strcmp(name, "col_fast")) {
        rnamsm::tuning().col_fast = value != 0;
        return RNAMSM_OK;
    }
    if (name && !strcmp(name, "gemm_splitk_short")) {
        rnamsm::tuning().gemm_splitk_short = value <= 0 ? 0 : (value >= 4 ? 4 : 2);
        return RNAMSM_OK;
    }
    return rnamsm::fail(RNAMSM_ERR_INVALID, "set_param: unknown parameter %s", name ? name : "(null)");
}
extern "C" int rnamsm_get_param(const char* name) {
    if (name && !strcmp(name, "gemm16_dma")) return rnamsm::tuning().gemm16_dma;
    if (name && !strcmp(name, "attn16")) return rnamsm::tuning().attn16;
    if (name && !strcmp(name, "ln_fold")) return rnamsm::tuning().ln_fold;
    if (name && !strcmp(name, "greedy_fused")) return rnamsm::tuning().greedy_fused;
    if (name && !strcmp(name, "col_small")) return rnamsm::tuning().col_small;
    if (name && !strcmp(name, "col_dma")) return rnamsm::tuning().col_dma;
    if (name && !strcmp(name, "col_fast")) return rnamsm::tuning().col_fast;
    if (name && !strcmp(name, "gemm_splitk_short")) return rnamsm::tuning().gemm_splitk_short;
    if (name && !strcmp(name, "row16_max_rows")) return rnamsm::tuning().row16_max_rows;
    if (name && !strcmp(name, "gemm16_mfma16")) return rnamsm::tuning().gemm16_mfma16;
    if (name && !strcmp(name, "gemm_splitk")) return rnamsm::tuning().gemm_splitk;
    if (name && !strcmp(name, "ln_fold_min_tokens")) return (int)rnamsm::LN_FOLD_MIN_TOKENS;      // read-only (common.h)
    if (name && !strcmp(name, "row_narrow")) return rnamsm::tuning().row_narrow;
    if (name && !strcmp(name, "gemm_tile")) return rnamsm::tuning().gemm_tile;
    if (name && !strcmp(name, "gemm_group")) return rnamsm::tuning().gemm_group;
    return -1;
}

extern "C" int rnamsm_version(void) { return RNAMSM_VERSION; }
extern "C" const char* rnamsm_last_error(void) { return rnamsm::g_err; }
extern "C" int rnamsm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}
