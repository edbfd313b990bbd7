// Error reporting and the opt-in event profiler of libdruggen_hip.so.
#include "common.h"
#include "pair.h"
#include "traversal.h"
#include "row_gemm_k384.h"
#include "row_gemm_n384.h"
#include "wgrad_stream.h"

#include <cstring>
#include <mutex>
#include <vector>

namespace dg {

char* error_buffer() {
    static thread_local char buf[512] = "ok";
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what) {
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(static_cast<int>(err), "%s: %s", what, hipGetErrorString(err));
    return 0;
}

int opt_in_dynamic_lds(std::atomic<unsigned long long>* done, const void* kernel, int bytes) {
    int dev = 0;
    hipError_t err = hipGetDevice(&dev);
    if (err != hipSuccess) return fail(static_cast<int>(err), "hipGetDevice: %s", hipGetErrorString(err));
    const unsigned long long bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;
    if (bit && (done->load(std::memory_order_acquire) & bit)) return 0;
    err = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (err != hipSuccess)
        return fail(static_cast<int>(err), "hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d) on device %d: %s", bytes,
                    dev, hipGetErrorString(err));
    if (bit) done->fetch_or(bit, std::memory_order_release);
    return 0;
}

// ---- profiler ---------------------------------------------------------------
namespace {
struct Span {
    int id;
    hipEvent_t start, stop;
};
std::mutex g_mu;
unsigned g_mask = 0;   // bit k set: launches of kernel id k are bracketed by events
std::vector<Span> g_spans;
std::vector<hipEvent_t> g_free;

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

ProfScope::ProfScope(int kernel_id, hipStream_t s) : id(kernel_id), stream(s), slot(nullptr) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (!((g_mask >> kernel_id) & 1u)) return;
    Span sp{kernel_id, take_event(), take_event()};
    (void)hipEventRecord(sp.start, stream);
    g_spans.push_back(sp);
    slot = reinterpret_cast<void*>(g_spans.size());  // index + 1
}

ProfScope::~ProfScope() {
    if (!slot) return;
    std::lock_guard<std::mutex> lock(g_mu);
    size_t idx = reinterpret_cast<size_t>(slot) - 1;
    if (idx < g_spans.size()) (void)hipEventRecord(g_spans[idx].stop, stream);
}

static std::atomic<int64_t> g_edge_rows{DG_EDGE_ROWS};
int64_t edge_rows() { return g_edge_rows.load(std::memory_order_relaxed); }

static thread_local int g_pair_depth = 0;      // dg_launch_pair_begin / _end nest: launches wait while the depth is > 0
bool pair_mode() { return g_pair_depth > 0; }

static thread_local int g_last_dir = 1;      // (the first edge-level launch ascends)
int take_direction(int64_t R) {
    if (R < edge_rows()) return 0;
    g_last_dir = !g_last_dir;
    return g_last_dir;
}
void note_forward(int64_t R) {
    if (R >= edge_rows()) g_last_dir = 0;
}

}  // namespace dg

extern "C" {

int dg_launch_pair_begin(void) {
    ++dg::g_pair_depth;
    return 0;
}

int dg_launch_pair_end(dg_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (dg::g_pair_depth > 0) --dg::g_pair_depth;
    if (dg::g_pair_depth > 0) return 0;      // an enclosing region is still open: its _end launches what waits
    int st = dg::flush_row_gemm_n384(stream);
    const int st2 = dg::flush_row_gemm_k384(stream);
    const int st3 = dg::flush_wgrad_stream(stream);
    st = st ? st : (st2 ? st2 : st3);
    return st ? st : dg::check_launch("dg_launch_pair_end");
}

int dg_version(void) { return DG_VERSION; }

int dg_set_edge_rows(int64_t rows) {
    dg::g_edge_rows.store(rows > 0 ? rows : static_cast<int64_t>(DG_EDGE_ROWS), std::memory_order_relaxed);
    return 0;
}

int64_t dg_edge_rows(void) { return dg::edge_rows(); }

const char* dg_last_error_string(void) { return dg::error_buffer(); }

int dg_prof_enable(int mask) {
    std::lock_guard<std::mutex> lock(dg::g_mu);
    dg::g_mask = static_cast<unsigned>(mask);
    return 0;
}

int dg_prof_reset(void) {
    std::lock_guard<std::mutex> lock(dg::g_mu);
    for (auto& sp : dg::g_spans) {
        (void)hipEventSynchronize(sp.stop);
        dg::g_free.push_back(sp.start);
        dg::g_free.push_back(sp.stop);
    }
    dg::g_spans.clear();
    return 0;
}

int dg_prof_read(int kernel_id, int64_t* launches, double* total_ms) {
    if (!launches || !total_ms) return dg::fail(DG_E_ARG, "dg_prof_read: null output");
    std::lock_guard<std::mutex> lock(dg::g_mu);
    int64_t n = 0;
    double ms = 0.0;
    for (auto& sp : dg::g_spans) {
        if (sp.id != kernel_id) continue;
        hipError_t err = hipEventSynchronize(sp.stop);
        if (err != hipSuccess) return dg::fail(static_cast<int>(err), "dg_prof_read: %s", hipGetErrorString(err));
        float t = 0.f;
        err = hipEventElapsedTime(&t, sp.start, sp.stop);
        if (err != hipSuccess) return dg::fail(static_cast<int>(err), "dg_prof_read: %s", hipGetErrorString(err));
        ms += t;
        ++n;
    }
    *launches = n;
    *total_ms = ms;
    return 0;
}

}  // extern "C"
