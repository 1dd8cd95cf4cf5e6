// Error plumbing and library-level queries of libs2s_mi355x.
#include <algorithm>
#include <cstdlib>
#include "ops.h"

#include <stdarg.h>

#include <map>
#include <mutex>
#include <utility>
#include <vector>

namespace s2s {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

// ---- in-library kernel timing (HIP events on the launching stream) -----------------------
struct ProfCat {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> spans;
    double flops = 0, bytes = 0;
};
static ProfCat g_prof[S2S_PROF_NCAT];
static unsigned g_prof_mask = 0;
static std::vector<hipEvent_t> g_event_pool;
static std::mutex g_prof_mu;

static hipEvent_t take_event() {
    if (!g_event_pool.empty()) {
        hipEvent_t e = g_event_pool.back();
        g_event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

std::mutex& capture_mutex() {
    static std::mutex mu;
    return mu;
}

static bool g_graphs = getenv("S2S_NO_GRAPHS") == nullptr;
bool prof_active() { return g_prof_mask != 0; }
bool graphs_enabled() { return g_graphs; }

static thread_local hipEvent_t t_cur_start = nullptr, t_cur_stop = nullptr;
static thread_local bool t_cur_on = false, t_cur_used = false;

bool prof_current_events(hipEvent_t* start, hipEvent_t* stop) {
    if (!t_cur_on || t_cur_used) return false;      // one timed launch per scope
    t_cur_used = true;
    *start = t_cur_start;
    *stop = t_cur_stop;
    return true;
}

ProfScope::ProfScope(hipStream_t st, int cat, double flops, double bytes) : st_(st), cat_(cat), on_(false) {
    if (!(g_prof_mask & (1u << cat))) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    on_ = true;
    start_ = take_event();
    stop_ = take_event();
    g_prof[cat].flops += flops;
    g_prof[cat].bytes += bytes;
    t_cur_start = (hipEvent_t)start_;
    t_cur_stop = (hipEvent_t)stop_;
    t_cur_on = true;
    t_cur_used = false;
}
ProfScope::~ProfScope() {
    if (!on_) return;
    const bool used = t_cur_used;
    t_cur_on = false;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (used) {
        g_prof[cat_].spans.emplace_back((hipEvent_t)start_, (hipEvent_t)stop_);
    } else {   // nothing was launched through launch_k inside the scope: drop the pair
        g_event_pool.push_back((hipEvent_t)start_);
        g_event_pool.push_back((hipEvent_t)stop_);
    }
}

}  // namespace s2s

extern "C" int32_t s2s_prof_enable(uint32_t category_mask) {
    std::lock_guard<std::mutex> lk(s2s::g_prof_mu);
    s2s::g_prof_mask = category_mask;
    return S2S_OK;
}

extern "C" int32_t s2s_graphs_enable(int32_t on) {
    s2s::g_graphs = on != 0;
    return S2S_OK;
}

// Device time between a start and a stop event with nothing in between (median of 64 pairs on an idle device).  Round 1
// bracketed kernels with stream events and subtracted this; the spans now come from the kernels' own timestamps
// (hipExtLaunchKernelGGL) and nothing is subtracted - kept as a diagnostic (s2s_prof_event_overhead_us measures on demand).
static double g_event_overhead_ms = -1.0;
static int measure_event_overhead() {
    if (g_event_overhead_ms >= 0.0) return S2S_OK;
    constexpr int N = 64;
    hipStream_t st = nullptr;
    S2S_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(2 * N);
    for (auto& e : ev) S2S_HIP_CHECK(hipEventCreate(&e));
    for (int i = 0; i < 2 * N; ++i) S2S_HIP_CHECK(hipEventRecord(ev[i], st));
    S2S_HIP_CHECK(hipStreamSynchronize(st));
    std::vector<float> t(N);
    for (int i = 0; i < N; ++i) S2S_HIP_CHECK(hipEventElapsedTime(&t[i], ev[2 * i], ev[2 * i + 1]));
    std::sort(t.begin(), t.end());
    g_event_overhead_ms = t[N / 2];
    for (auto& e : ev) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(st);
    return S2S_OK;
}

extern "C" double s2s_prof_event_overhead_us(void) {
    (void)measure_event_overhead();
    return g_event_overhead_ms < 0 ? 0.0 : g_event_overhead_ms * 1e3;
}

extern "C" int32_t s2s_prof_collect(int32_t category, double* total_ms, double* flops, double* bytes, int64_t* launches) {
    S2S_REQUIRE(category >= 0 && category < S2S_PROF_NCAT, "prof_collect: bad category %d", category);
    std::lock_guard<std::mutex> lk(s2s::g_prof_mu);
    s2s::ProfCat& c = s2s::g_prof[category];
    double ms = 0;
    for (auto& sp : c.spans) {
        S2S_HIP_CHECK(hipEventSynchronize(sp.second));
        float t = 0;
        S2S_HIP_CHECK(hipEventElapsedTime(&t, sp.first, sp.second));
        ms += (double)t;       // kernel begin -> end (hipExtLaunchKernelGGL events): nothing to subtract
        s2s::g_event_pool.push_back(sp.first);
        s2s::g_event_pool.push_back(sp.second);
    }
    if (total_ms) *total_ms = ms;
    if (flops) *flops = c.flops;
    if (bytes) *bytes = c.bytes;
    if (launches) *launches = (int64_t)c.spans.size();
    c.spans.clear();
    c.flops = c.bytes = 0;
    return S2S_OK;
}

// ---- graph-replay timing of arbitrary op sequences (microbenchmarks: per-kernel cost INSIDE a captured graph) -------------
static hipStream_t g_bench_stream = nullptr;
static bool g_bench_capturing = false;

extern "C" int32_t s2s_graph_bench_begin(void** out_stream) {
    S2S_REQUIRE(out_stream && !g_bench_capturing, "graph_bench_begin: bad state");
    if (!g_bench_stream) S2S_HIP_CHECK(hipStreamCreateWithFlags(&g_bench_stream, hipStreamNonBlocking));
    {   // the K-slice workspace of the decode projections cannot be allocated inside the capture
        const int rc = s2s::dec2_ensure_default_ws();
        if (rc) return rc;
    }
    S2S_HIP_CHECK(hipStreamBeginCapture(g_bench_stream, hipStreamCaptureModeThreadLocal));
    g_bench_capturing = true;
    *out_stream = (void*)g_bench_stream;
    return S2S_OK;
}

extern "C" int32_t s2s_graph_bench_end(int32_t warm, int32_t reps, double* us_per_replay) {
    S2S_REQUIRE(g_bench_capturing && reps >= 1 && us_per_replay, "graph_bench_end: bad state");
    g_bench_capturing = false;
    hipGraph_t graph = nullptr;
    S2S_HIP_CHECK(hipStreamEndCapture(g_bench_stream, &graph));
    hipGraphExec_t exec = nullptr;
    const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    S2S_HIP_CHECK(ie);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    S2S_HIP_CHECK(hipEventCreate(&e0));
    S2S_HIP_CHECK(hipEventCreate(&e1));
    for (int i = 0; i < warm; ++i) S2S_HIP_CHECK(hipGraphLaunch(exec, g_bench_stream));
    S2S_HIP_CHECK(hipEventRecord(e0, g_bench_stream));
    for (int i = 0; i < reps; ++i) S2S_HIP_CHECK(hipGraphLaunch(exec, g_bench_stream));
    S2S_HIP_CHECK(hipEventRecord(e1, g_bench_stream));
    S2S_HIP_CHECK(hipStreamSynchronize(g_bench_stream));
    float ms = 0.f;
    S2S_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *us_per_replay = (double)ms * 1e3 / reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipGraphExecDestroy(exec);
    return S2S_OK;
}

extern "C" const char* s2s_last_error(void) { return s2s::g_last_error.c_str(); }
extern "C" int32_t s2s_abi_version(void) { return S2S_ABI_VERSION; }
extern "C" int32_t s2s_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// ---- S2S_GUARD: poisoned gaps + unmapped tail pages (development; see common.h) ---------------------------------------------------
namespace s2s {
namespace {
struct GuardGap {
    char* p;
    size_t len;
    std::string owner;
    int index;
};
struct GuardMap {       // one mode-2 allocation: reserved range [va, va + reserved), mapped [va, va + mapped), user pointer inside
    char* va;
    size_t reserved, mapped;
    hipMemGenericAllocationHandle_t handle;
};
std::mutex g_guard_mu;
std::vector<GuardGap> g_guard_gaps;
std::map<void*, GuardMap> g_guard_maps;
constexpr unsigned char GUARD_POISON = 0xA5;
}  // namespace

int guard_mode() {
    static const int mode = [] {
        const char* e = getenv("S2S_GUARD");
        return e ? atoi(e) : 0;
    }();
    return mode;
}

size_t guard_gap() { return guard_mode() > 0 ? (size_t)64 << 10 : 0; }

int guard_alloc(char** out, size_t bytes, const char* owner) {
    *out = nullptr;
    if (guard_mode() < 2) {
        S2S_HIP_CHECK(hipMalloc((void**)out, bytes));
        if (guard_mode() == 1) fprintf(stderr, "[s2s guard] %-10s %p .. %p (%zu bytes)\n", owner, (void*)*out, (void*)(*out + bytes), bytes);
        return S2S_OK;
    }
    int dev = 0;
    S2S_HIP_CHECK(hipGetDevice(&dev));
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    S2S_HIP_CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    if (gran < 4096) gran = 4096;
    const size_t need = (bytes + 255) & ~(size_t)255;
    const size_t mapped = (need + gran - 1) / gran * gran;
    GuardMap g{};
    g.mapped = mapped;
    g.reserved = mapped + gran;                         // one granule behind the mapping stays reserved and UNMAPPED
    void* va = nullptr;
    S2S_HIP_CHECK(hipMemAddressReserve(&va, g.reserved, gran, nullptr, 0));
    g.va = (char*)va;
    S2S_HIP_CHECK(hipMemCreate(&g.handle, mapped, &prop, 0));
    S2S_HIP_CHECK(hipMemMap(va, mapped, 0, g.handle, 0));
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    S2S_HIP_CHECK(hipMemSetAccess(va, mapped, &acc, 1));
    *out = g.va + (mapped - need);                      // right-aligned: the allocation ENDS where the mapping ends
    fprintf(stderr, "[s2s guard] %-10s %p .. %p (%zu bytes), unmapped from %p\n", owner, (void*)*out, (void*)(*out + need), bytes, (void*)(g.va + mapped));
    std::lock_guard<std::mutex> lk(g_guard_mu);
    g_guard_maps[*out] = g;
    return S2S_OK;
}

void guard_free(void* p) {
    if (!p) return;
    if (!guard_mode()) {         // the product path: plain hipFree, nothing else
        (void)hipFree(p);
        return;
    }
    {
        std::lock_guard<std::mutex> lk(g_guard_mu);
        // gaps that live inside this allocation go with it (an arena's gaps, or the single gap behind a stand-alone buffer)
        auto it = g_guard_maps.find(p);
        const char* lo = (const char*)p;
        const char* hi = it != g_guard_maps.end() ? it->second.va + it->second.mapped : nullptr;
        if (!hi) {
            size_t sz = 0;
            hipDeviceptr_t base = nullptr;
            if (hipMemGetAddressRange(&base, &sz, (hipDeviceptr_t)p) == hipSuccess) hi = (const char*)base + sz;
            else (void)hipGetLastError();
        }
        if (hi) g_guard_gaps.erase(std::remove_if(g_guard_gaps.begin(), g_guard_gaps.end(), [&](const GuardGap& g) { return g.p >= lo && g.p < hi; }), g_guard_gaps.end());
        if (it != g_guard_maps.end()) {
            const GuardMap g = it->second;
            g_guard_maps.erase(it);
            (void)hipMemUnmap(g.va, g.mapped);
            (void)hipMemRelease(g.handle);
            (void)hipMemAddressFree(g.va, g.reserved);
            return;
        }
    }
    (void)hipFree(p);
}

int guard_add_gap(char* p, size_t len, const char* owner, int index) {
    if (!guard_mode() || !len) return S2S_OK;
    S2S_HIP_CHECK(hipMemset(p, GUARD_POISON, len));
    std::lock_guard<std::mutex> lk(g_guard_mu);
    g_guard_gaps.push_back(GuardGap{p, len, owner, index});
    return S2S_OK;
}
}  // namespace s2s

extern "C" int32_t s2s_guard_mode(void) { return s2s::guard_mode(); }

// Checks every poisoned gap (synchronous copies: call it between launches, never while a stream is being captured); writes one line per
// corrupted gap into `report` and re-poisons it.  Returns the number of corrupted gaps, or < 0 on a HIP error.
extern "C" int32_t s2s_guard_check(char* report, int32_t cap) {
    using namespace s2s;
    if (report && cap > 0) report[0] = 0;
    if (!guard_mode()) return 0;
    std::lock_guard<std::mutex> no_capture(capture_mutex());
    std::lock_guard<std::mutex> lk(g_guard_mu);
    S2S_HIP_CHECK(hipDeviceSynchronize());
    std::vector<unsigned char> host;
    int bad = 0;
    size_t used = 0;
    for (const GuardGap& g : g_guard_gaps) {
        host.resize(g.len);
        S2S_HIP_CHECK(hipMemcpy(host.data(), g.p, g.len, hipMemcpyDeviceToHost));
        size_t first = g.len, n_bad = 0, last = 0;
        for (size_t i = 0; i < g.len; ++i)
            if (host[i] != GUARD_POISON) {
                if (first == g.len) first = i;
                last = i;
                ++n_bad;
            }
        if (!n_bad) continue;
        ++bad;
        if (report && used + 1 < (size_t)cap) {
            const int w = snprintf(report + used, (size_t)cap - used, "%s buffer %d: %zu byte(s) written behind its end, offsets +%zu .. +%zu (gap %p, %zu bytes)\n",
                                   g.owner.c_str(), g.index, n_bad, first, last, (void*)g.p, g.len);
            if (w > 0) used = std::min((size_t)cap - 1, used + (size_t)w);
        }
        S2S_HIP_CHECK(hipMemset(g.p, GUARD_POISON, g.len));
    }
    return bad;
}
