// Error reporting and version of libaig_hip.so.
#include <mutex>
#include <vector>

#include "common.h"

namespace aig {

static thread_local char g_err[512] = "";
thread_local const int* g_row_limit = nullptr;

char* err_buf() { return g_err; }

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

namespace {
__global__ void k_fill_words(unsigned* __restrict__ p, unsigned word, size_t words) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) p[i] = word;
}
}  // namespace

// (kernel, device) pairs whose dynamic-LDS limit has been raised, and to how much
namespace {
struct LdsGrant { const void* kernel; int device; size_t bytes; };
std::mutex g_lds_mutex;
std::vector<LdsGrant> g_lds_grants;
}  // namespace

hipError_t ensure_dynamic_lds(const void* kernel, size_t bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(g_lds_mutex);
    for (LdsGrant& g : g_lds_grants)
        if (g.kernel == kernel && g.device == dev) {
            if (g.bytes >= bytes) return hipSuccess;
            e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e == hipSuccess) g.bytes = bytes;
            return e;
        }
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) g_lds_grants.push_back({kernel, dev, bytes});
    return e;
}

hipError_t fill_words(void* p, unsigned word, size_t bytes, hipStream_t s) {
    const size_t words = bytes / 4;
    if (words == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fill_words, dim3(grid_for((int64_t)words, 256 * 8, 2048)), dim3(256), 0, s,
                       static_cast<unsigned*>(p), word, words);
    return hipGetLastError();
}

// ---- per-kernel timing (benchmark aid)
namespace {
struct TimedLaunch { hipEvent_t start, stop; };
std::mutex g_kt_mutex;
volatile unsigned g_kt_mask = 0u;   // bit k: kernel k is being timed
std::vector<TimedLaunch> g_kt[AIG_KT_COUNT];
const char* const g_kt_names[AIG_KT_COUNT] = {"k_gat_aggregate", "k_gat_bwd_dst", "k_gat_bwd_src", "k_pool_walk",
                                              "k_linear128", "k_wgrad128", "k_gat1_fwd", "k_gat1_bwd"};

std::vector<hipEvent_t> g_kt_free;   // events of forgotten records, handed out again

void kt_clear() {
    for (auto& v : g_kt) {
        for (TimedLaunch& t : v) { g_kt_free.push_back(t.start); g_kt_free.push_back(t.stop); }
        v.clear();
    }
}

bool kt_event(hipEvent_t* e) {   // under g_kt_mutex
    if (!g_kt_free.empty()) { *e = g_kt_free.back(); g_kt_free.pop_back(); return true; }
    return hipEventCreate(e) == hipSuccess;
}
}  // namespace

bool kernel_timing_events(int kernel, hipStream_t s, hipEvent_t* start, hipEvent_t* stop) {
    if (kernel < 0 || kernel >= AIG_KT_COUNT || !(g_kt_mask >> kernel & 1u)) return false;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return false;
    TimedLaunch t{};
    std::lock_guard<std::mutex> lock(g_kt_mutex);
    if (!kt_event(&t.start)) return false;
    if (!kt_event(&t.stop)) { g_kt_free.push_back(t.start); return false; }
    g_kt[kernel].push_back(t);
    *start = t.start;
    *stop = t.stop;
    return true;
}

}  // namespace aig

extern "C" int aig_kernel_timing(unsigned kernel_mask) {
    std::lock_guard<std::mutex> lock(aig::g_kt_mutex);
    if (kernel_mask && !aig::g_kt_mask) aig::kt_clear();
    aig::g_kt_mask = kernel_mask;
    return AIG_OK;
}

extern "C" const char* aig_kernel_timing_name(int kernel) {
    return kernel >= 0 && kernel < AIG_KT_COUNT ? aig::g_kt_names[kernel] : nullptr;
}

extern "C" int aig_kernel_timing_read(int kernel, double* total_ms, int64_t* launches) {
    AIG_REQUIRE(kernel >= 0 && kernel < AIG_KT_COUNT && total_ms && launches, "aig_kernel_timing_read: bad argument");
    std::lock_guard<std::mutex> lock(aig::g_kt_mutex);
    double ms = 0.0;
    for (aig::TimedLaunch& t : aig::g_kt[kernel]) {
        AIG_HIP_CHECK(hipEventSynchronize(t.stop));
        float one = 0.f;
        AIG_HIP_CHECK(hipEventElapsedTime(&one, t.start, t.stop));
        ms += one;
    }
    *total_ms = ms;
    *launches = (int64_t)aig::g_kt[kernel].size();
    return AIG_OK;
}

extern "C" int aig_abi_version(void) { return 5; }

extern "C" const char* aig_last_error(void) { return aig::err_buf(); }
