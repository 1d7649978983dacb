// Error reporting and version of libaig_hip.so.
#include <mutex>
#include <vector>

#include "common.h"

namespace aig {

static thread_local char g_err[512] = "";

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

}  // namespace aig

extern "C" int aig_abi_version(void) { return 2; }

extern "C" const char* aig_last_error(void) { return aig::err_buf(); }
