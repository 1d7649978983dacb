// Times rocprim::radix_sort_pairs with the default configuration (merge sort below 1M items) against the same call
// forced onto the onesweep path, for the sizes and key widths the constraint kernels sort.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

using forced = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 4096>;

template <typename K, typename Cfg>
int run(size_t n, int bits, const char* label) {
    std::vector<K> h(n);
    std::mt19937_64 rng(1);
    const unsigned long long mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1);
    for (auto& v : h) v = (K)(rng() & mask);
    K *k0, *k1; int *v0, *v1;
    CK(hipMalloc(&k0, n * sizeof(K))); CK(hipMalloc(&k1, n * sizeof(K)));
    CK(hipMalloc(&v0, n * 4)); CK(hipMalloc(&v1, n * 4));
    CK(hipMemcpy(k0, h.data(), n * sizeof(K), hipMemcpyHostToDevice));
    CK(hipMemset(v0, 0, n * 4));
    size_t tb = 0;
    CK((rocprim::radix_sort_pairs<Cfg>(nullptr, tb, k0, k1, v0, v1, n, 0, bits, 0)));
    void* temp; CK(hipMalloc(&temp, tb));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) CK((rocprim::radix_sort_pairs<Cfg>(temp, tb, k0, k1, v0, v1, n, 0, bits, 0)));
    CK(hipEventRecord(a, 0));
    const int R = 50;
    for (int i = 0; i < R; ++i) CK((rocprim::radix_sort_pairs<Cfg>(temp, tb, k0, k1, v0, v1, n, 0, bits, 0)));
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-8s n=%8zu key=%zuB bits=%2d  %8.1f us/sort  temp=%zu\n", label, n, sizeof(K), bits, ms * 1e3 / R, tb);
    hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(temp);
    return 0;
}

int main() {
    for (size_t n : {20000ul, 60000ul, 200000ul, 400000ul, 2000000ul}) {
        for (int bits : {17, 21}) {
            if (run<unsigned, rocprim::default_config>(n, bits, "default")) return 1;
            if (run<unsigned, forced>(n, bits, "onesweep")) return 1;
        }
        if (run<unsigned long long, rocprim::default_config>(n, 34, "default")) return 1;
        if (run<unsigned long long, forced>(n, 34, "onesweep")) return 1;
    }
    return 0;
}
