// Tuning aid: wave_sum_butterfly (common.h) against the shuffle loop it replaces, bit for bit, on random data.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Igan_amd/csrc -Iinclude tools/wave_sum_check.hip -o /tmp/wsc && /tmp/wsc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "common.h"
__global__ void k(const float* in, float* a, float* b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = in[i], w = v;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    a[i] = v;
    b[i] = aig::wave_sum_butterfly(w);
}
int main() {
    const int n = 1 << 16;
    std::vector<float> h(n);
    srand(7);
    for (auto& x : h) x = (float)rand() / RAND_MAX * 2.f - 1.f + (rand() % 7 == 0 ? 1e4f : 0.f);
    float *in, *a, *b;
    hipMalloc(&in, n * 4); hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
    hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(in, a, b);
    std::vector<unsigned> ha(n), hb(n);
    hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += ha[i] != hb[i];
    printf("wave_sum_butterfly: %d of %d lanes differ from the shuffle loop\n", bad, n);
    return bad != 0;
}
