// Probe: in which order do v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32 add the products of their k
// slices to the accumulator, and is every step a fused multiply-add?  The kernels run one instruction on
// random operands with a non-zero C; the host replays candidate orders with fmaf and compares bit patterns.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// A[i][k] in lane (i + 32k) / (i + 16k); B[k][j] in lane (j + 32k) / (j + 16k)
__global__ void k32(const float* A, const float* B, const float* C, float* D) {  // A[32][2], B[2][32], C/D[32][32]
    const int l = threadIdx.x, i = l & 31, k = l >> 5;
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = C[((r & 3) + 8 * (r >> 2) + 4 * k) * 32 + i];
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * 2 + k], B[k * 32 + i], c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * k) * 32 + i] = c[r];
}
__global__ void k16(const float* A, const float* B, const float* C, float* D) {  // A[16][4], B[4][16], C/D[16][16]
    const int l = threadIdx.x, i = l & 15, k = l >> 4;
    f32x4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[(4 * k + r) * 16 + i];
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * 4 + k], B[k * 16 + i], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * k + r) * 16 + i] = c[r];
}

static float frand() { return (float)((rand() % 20001) - 10000) / 977.0f; }

int main() {
    srand(7);
    float hA[64], hB[64], hC[1024], hD[1024];
    float *A, *B, *C, *D;
    hipMalloc(&A, 256); hipMalloc(&B, 256); hipMalloc(&C, 4096); hipMalloc(&D, 4096);
    for (int shape = 0; shape < 2; ++shape) {
        const int n = shape ? 16 : 32, K = shape ? 4 : 2;
        int ok_fwd = 0, ok_rev = 0, ok_unfused = 0, trials = 0;
        for (int t = 0; t < 50; ++t) {
            for (int e = 0; e < 64; ++e) { hA[e] = frand(); hB[e] = frand(); }
            for (int e = 0; e < n * n; ++e) hC[e] = frand();
            hipMemcpy(A, hA, 256, hipMemcpyHostToDevice); hipMemcpy(B, hB, 256, hipMemcpyHostToDevice);
            hipMemcpy(C, hC, n * n * 4, hipMemcpyHostToDevice);
            if (shape) hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, A, B, C, D);
            else hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, A, B, C, D);
            hipMemcpy(hD, D, n * n * 4, hipMemcpyDeviceToHost);
            bool fwd = true, rev = true, unf = true;
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    // D[i][j]: row index = A's row, column = B's column
                    float f = hC[i * n + j], r = hC[i * n + j], u = hC[i * n + j];
                    for (int k = 0; k < K; ++k) f = fmaf(hA[i * K + k], hB[k * n + j], f);
                    for (int k = K - 1; k >= 0; --k) r = fmaf(hA[i * K + k], hB[k * n + j], r);
                    for (int k = 0; k < K; ++k) { volatile float p = hA[i * K + k] * hB[k * n + j]; u = u + p; }
                    const float d = hD[i * n + j];
                    fwd &= !memcmp(&d, &f, 4); rev &= !memcmp(&d, &r, 4); unf &= !memcmp(&d, &u, 4);
                }
            ok_fwd += fwd; ok_rev += rev; ok_unfused += unf; ++trials;
        }
        printf("%s: of %d random instructions, every element equals the fmaf chain k ascending in %d, "
               "k descending in %d, unfused multiply then add in %d\n",
               shape ? "v_mfma_f32_16x16x4_f32" : "v_mfma_f32_32x32x2_f32", trials, ok_fwd, ok_rev, ok_unfused);
    }
    return 0;
}
