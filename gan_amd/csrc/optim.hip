// One critic iteration's parameter update in one launch: the three terms' gradients summed into the parameters'
// gradient tensors and the Adam update of train.py:221,266 (torch.optim.Adam, betas (0.5, 0.9)) applied to them.
// The critic has 59,137 parameters in 22 tensors: zeroing their gradients, three multi-tensor adds, the step-count
// bump and the fused optimizer kernel are six launches that together take ~110 us of an otherwise idle GPU per
// iteration; the numbers here are the same sums in the same order and the same update.
#include <climits>

#include "common.h"

namespace aig {
namespace {

constexpr int ADAM_THREADS = 256;

// `count` consecutive elements of one parameter tensor (at most ADAM_THREADS)
struct AdamChunk {
    float* param;
    float* grad;          // receives g0 + g1 + g2 (absent terms are skipped)
    const float* g0;
    const float* g1;
    const float* g2;
    float* exp_avg;
    float* exp_avg_sq;
    long long count;
};
static_assert(sizeof(AdamChunk) == 64, "the host side writes this layout");

__global__ __launch_bounds__(ADAM_THREADS) void k_adam_sum3(const AdamChunk* __restrict__ chunks, float* const* __restrict__ steps,
                                                            int nsteps, float step_value, double lr, double beta1, double beta2,
                                                            double eps, float bias1, float bias2_sqrt, int mode) {
    const AdamChunk c = chunks[blockIdx.x];
    const int i = threadIdx.x;
    const bool sum = mode & 1, update = mode & 2;
    if (update && blockIdx.x == 0 && i < nsteps) *steps[i] = step_value;     // (nobody reads the counters in this launch)
    if (i >= c.count) return;
    float g;
    if (sum) {
        g = 0.f;
        if (c.g0) g += c.g0[i];
        if (c.g1) g += c.g1[i];
        if (c.g2) g += c.g2[i];
        c.grad[i] = g;
    } else {
        g = c.grad[i];      // (summed by an earlier launch, averaged over the ranks since)
    }
    if (!update) return;
    // first and second moment, bias-corrected step: the products with the double-precision hyper-parameters are
    // taken in double and rounded to float where the moments are stored, as the optimizer's own kernel does
    const float m = (float)(beta1 * (double)c.exp_avg[i] + (1.0 - beta1) * (double)g);
    const float v = (float)(beta2 * (double)c.exp_avg_sq[i] + (1.0 - beta2) * (double)g * (double)g);
    const float step_size = (float)(lr / (double)bias1);
    const float denom = (float)((double)(sqrtf(v) / bias2_sqrt) + eps);
    c.param[i] -= step_size * m / denom;
    c.exp_avg[i] = m;
    c.exp_avg_sq[i] = v;
}

}  // namespace
}  // namespace aig

using namespace aig;

extern "C" int aig_adam_chunk_elems(void) { return ADAM_THREADS; }

extern "C" int aig_adam_sum3(int64_t nchunks, const void* chunks, int64_t nsteps, const void* step_ptrs, float step_value,
                             double lr, double beta1, double beta2, double eps, float bias_correction1,
                             float bias_correction2_sqrt, int mode, void* stream) {
    AIG_REQUIRE(nchunks > 0 && nchunks < INT_MAX && chunks, "aig_adam_sum3: no chunks");
    AIG_REQUIRE(nsteps >= 0 && nsteps <= ADAM_THREADS && (nsteps == 0 || step_ptrs), "aig_adam_sum3: at most %d step counters",
                ADAM_THREADS);
    AIG_REQUIRE(mode >= 1 && mode <= 3, "aig_adam_sum3: mode is 1 (sum), 2 (update) or 3 (both)");
    AIG_REQUIRE(!(mode & 2) || (bias_correction1 > 0.f && bias_correction2_sqrt > 0.f),
                "aig_adam_sum3: bias corrections must be positive");
    hipLaunchKernelGGL(k_adam_sum3, dim3((unsigned)nchunks), dim3(ADAM_THREADS), 0, static_cast<hipStream_t>(stream),
                       static_cast<const AdamChunk*>(chunks), static_cast<float* const*>(step_ptrs), (int)nsteps, step_value,
                       lr, beta1, beta2, eps, bias_correction1, bias_correction2_sqrt, mode);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

// ------------------------------------------------------------------------------------------ the step's logged scalars
namespace aig {
namespace {
// out8 = {D_real, D_fake, gap, gp, grad_norm, loss_D, loss_G, pen} of train.py:263-272,289-296 from the device
// scalars the recorded terms leave behind: gap = D_real - D_fake, loss_D = (D_fake - D_real) + gp_lambda gp,
// pen = pen_scale * mean_g pen9[g][8], loss_G = -D_raw + lambda_cons pen.  One thread: a dozen scalar launches otherwise.
__global__ void k_step_metrics(int B, const float* __restrict__ d_real, const float* __restrict__ d_fake,
                               const float* __restrict__ gp, const float* __restrict__ grad_norm,
                               const float* __restrict__ d_raw, const float* __restrict__ pen9, float gp_lambda,
                               float lambda_cons, float pen_scale, float* __restrict__ out8) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float dr = d_real[0], df = d_fake[0], g = gp[0];
    float s = 0.f;
    for (int i = 0; i < B; ++i) s += pen9[9 * i + 8];
    float pen = s / (float)B;
    if (pen_scale != 1.f) pen = pen * pen_scale;
    out8[0] = dr;
    out8[1] = df;
    out8[2] = dr - df;
    out8[3] = g;
    out8[4] = grad_norm[0];
    // (products and sums rounded separately, as the tensor expressions they replace: no contraction into fma)
    out8[5] = __fadd_rn(df - dr, __fmul_rn(gp_lambda, g));
    out8[6] = __fadd_rn(-d_raw[0], __fmul_rn(lambda_cons, pen));
    out8[7] = pen;
}
}  // namespace
}  // namespace aig

extern "C" int aig_step_metrics(int64_t B, const float* d_real, const float* d_fake, const float* gp, const float* grad_norm,
                                const float* d_raw, const float* pen9, float gp_lambda, float lambda_cons, float pen_scale,
                                float* out8, void* stream) {
    AIG_REQUIRE(B >= 1 && B < INT_MAX, "aig_step_metrics: needs at least one graph");
    AIG_REQUIRE(d_real && d_fake && gp && grad_norm && d_raw && pen9 && out8, "aig_step_metrics: null pointer");
    hipLaunchKernelGGL(k_step_metrics, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), (int)B, d_real, d_fake, gp,
                       grad_norm, d_raw, pen9, gp_lambda, lambda_cons, pen_scale, out8);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}
