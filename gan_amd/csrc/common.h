// Shared helpers of libaig_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>

#include "../../include/aig_abi.h"

namespace aig {

constexpr int WAVE = 64;

char* err_buf();  // thread-local, 512 bytes
int fail(int code, const char* fmt, ...);

#define AIG_HIP_CHECK(expr)                                                                   \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return ::aig::fail(AIG_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                               __FILE__, __LINE__);                                           \
    } while (0)

#define AIG_LAUNCH_CHECK() AIG_HIP_CHECK(hipGetLastError())

#define AIG_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) return ::aig::fail(AIG_EINVAL, __VA_ARGS__); \
    } while (0)

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// carve aligned pieces out of a caller-provided workspace
struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* p) : base(static_cast<char*>(p)) {}
    template <typename T>
    T* take(size_t count) {
        T* p = reinterpret_cast<T*>(base + off);
        off += align_up(count * sizeof(T));
        return p;
    }
};

inline int grid_for(int64_t items, int per_block, int cap = 1 << 20) {
    int64_t g = (items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return static_cast<int>(g);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

}  // namespace aig
