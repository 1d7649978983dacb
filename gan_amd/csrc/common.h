// Shared helpers of libaig_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

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

// Fills `bytes` (a multiple of 4, 4-byte aligned) with a 32-bit word by a kernel.  The library never calls
// hipMemsetAsync: recorded in a HIP graph, a memset node of this ROCm release writes garbage into the head or tail of
// its range from the second replay on (tools/graph_memset_probe2.py), and every entry point here may be recorded.
hipError_t fill_words(void* p, unsigned word, size_t bytes, hipStream_t s);

// Raises `kernel`'s dynamic-LDS limit to at least `bytes` on the CURRENT device, once per kernel and device
// (a kernel attribute belongs to the device's copy of the code object); safe to call from several threads.
hipError_t ensure_dynamic_lds(const void* kernel, size_t bytes);

// Per-kernel GPU time for the benchmark (aig_kernel_timing in the header): while a kernel is selected, its
// launch carries a start and a stop event of its own (the dispatch itself stamps them at the kernel's first and
// last wave: no marker packets around it, so the figure is the kernel's duration and nothing else) - not while
// its stream is being recorded into a graph.  Off, the macro is one load and a branch in front of the launch.
bool kernel_timing_events(int kernel, hipStream_t s, hipEvent_t* start, hipEvent_t* stop);

#define AIG_LAUNCH_TIMED(id, kernel, grid, block, lds, stream, ...)                                           \
    do {                                                                                                      \
        hipEvent_t kt_a_ = nullptr, kt_b_ = nullptr;                                                          \
        if (::aig::kernel_timing_events(id, stream, &kt_a_, &kt_b_))                                          \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, kt_a_, kt_b_, 0, __VA_ARGS__);            \
        else                                                                                                  \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                \
    } while (0)

// A row count that only the device knows.  An entry point whose caller sizes the launches by a capacity (rows past the
// real count hold neutral values) sets this pointer for the calls it makes; the launchers of aig_linear128 and
// aig_wgrad pass it on and their kernels work on min(host count, *limit) rows.  Null: the host count stands.
extern thread_local const int* g_row_limit;
struct RowLimitScope {
    const int* prev;
    explicit RowLimitScope(const int* p) : prev(g_row_limit) { g_row_limit = p; }
    ~RowLimitScope() { g_row_limit = prev; }
};
__device__ __forceinline__ int limited_rows(int n, const int* __restrict__ limit) { return limit ? min(n, limit[0]) : n; }

// Sum over the 64 lanes of a wave, in every lane, in the order of a butterfly that pairs lanes 32, 16, 8, 4, 2 and 1
// apart - what `for (off = 32; off; off >>= 1) v += __shfl_xor(v, off)` computes, bit for bit (a + b == b + a, and
// after the step for distance d the lanes d apart hold equal values, so a rotation by d/2 inside a row of 16 fetches
// the same value the exchange would) - out of vector instructions: the two half-swaps of gfx950 and four row
// rotations.  A shuffle goes through the LDS crossbar and the compiler waits for each one: six in a row cost ~0.3 us,
// which is what the single-workgroup kernels (critic head, readout finish) mostly consisted of.
__device__ __forceinline__ float wave_sum_butterfly(float v) {
    {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false));   // row_ror:8
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, false));   // row_ror:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xF, 0xF, false));   // row_ror:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xF, 0xF, false));   // row_ror:1
    return v;
}

// y = act(x W^T + bias), x [M,128] f32, on the bf16 matrix cores at f32 accuracy (dense6.hip).  W: [128][ldw] rows =
// output columns, or (wt) stored [in][out]; act 0 none, 1 ReLU, 2 ELU.
int launch_linear128_x6(hipStream_t s, int M, const float* X, const float* W, int ldw, int wt, const float* bias, int act,
                        float* out);
int launch_linear128_x6_terms(hipStream_t s, int M, const float* X, const float* W, int ldw, int wt, float* out,
                              const float* att_src, const float* att_dst, const float* lin_edge_w, const float* att_edge,
                              float* a_s, float* a_d, float* ce);
// dW = g1^T x1 + g2^T x2 over n rows ([128,F] in PyTorch's [out,in] layout; F in {1, 2, 4, 128}); ws holds
// 2 * aig_wgrad_ws_bytes(n, F) bytes (dense.hip)
int wgrad_sum2(int64_t n, int64_t F, const float* g1, const float* x1, const float* g2, const float* x2, float* dW, void* ws,
               size_t ws_bytes, hipStream_t s);
// partial sums of g^T x (g, x [n,128]) on the bf16 matrix cores at f32 accuracy: partial[blocks][128][128], units of
// 32 rows dealt round-robin over the blocks (wgrad6.hip)
int launch_wgrad128_x6(hipStream_t s, int n, int blocks, const float* g, const float* x, float* partial);

// The closed-form gradient penalty over tiles of 32 rows (gp_tile.hip): forward, input gradient (partial[blocks] =
// shares of |g_x|^2, zero for blocks without rows) and everything behind the norm (partial: gp_tiles_partial_bytes()).
// AIG_GP_TILES=0 in the environment keeps the chain of one kernel per pass (gp.hip).
bool gp_tiles_enabled();
size_t gp_tiles_partial_bytes();
int launch_gp_tiles_fwd(hipStream_t s, int n, const int* row_limit, const float* xr, const float* xf, const float* alpha,
                        const float* W1, const float* b1, const float* W2, const float* b2, float* xhat, float* h1, float* h2,
                        void* clear, size_t clear_bytes);
int launch_gp_tiles_input_grad(hipStream_t s, int n, const int* row_limit, int L, const int* meta, const float* g_r,
                               const int* node_level, const float* w, const int* argmax, const float* W1, const float* W2,
                               const float* h1, const float* h2, float* g_z2, float* g_h1, float* g_z1, float* g_x,
                               float* partial, int* blocks);
int launch_gp_tiles_param_grad(hipStream_t s, int n, const int* row_limit, int L, const int* meta, const float* coef,
                               const float* g_r, const int* node_level, const float* w, const int* argmax, const float* W1,
                               const float* W2, const float* xhat, const float* h1, const float* h2, const float* g_z2,
                               const float* g_h1, const float* g_z1, const float* g_x, float* dW1, float* db1, float* dW2,
                               float* db2, float* ubar_gr, float* partial, float out_scale);

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Direct global -> LDS load of 16 bytes per lane that the compiler's wait-count bookkeeping does not
// see (after the builtin it waits for ALL memory traffic before the next LDS read, which would put the
// whole load latency in front of the products it is meant to hide behind).  The kernel waits itself:
// counted wait + barrier at the top of a trip.  lds_byte_addr: wave-uniform LDS address of the 1 KB
// the 64 lanes fill.
__device__ __forceinline__ void load_to_lds16_uncounted(const float* src, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                 :
                 : "s"(lds_byte_addr), "v"(src)
                 : "memory");
}
__device__ __forceinline__ unsigned lds_address(const float* p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const float*)p;
}

}  // namespace aig
