// TEST INFRASTRUCTURE ONLY - CPU checker for the tie-exact top-k select.
//
// Restates the CPU path of torch.topk(largest=True, sorted=True) that the reference's
// generator calls at models/generator.py:167.  The algorithm lives in a third-party
// dependency, not in the reference tree: ATen (torch 2.10.0) native/TopKImpl.h:31-95 -
// (value, index) pairs; std::partial_sort of the first k when k*64 <= n, otherwise
// std::nth_element(k-1) followed by std::sort of the first k-1; comparator
// "NaN first, then larger value"; the first k pairs are the answer.  The selection order
// among equal values is whatever libstdc++'s heap-select / introselect leave, so this
// file calls those very routines.  Parity status: PINNED by tests/golden/topk_kat.npz
// (outputs of torch.topk itself for 2,560 crafted rows, n = 1..512).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>

namespace {
using elem_t = std::pair<float, int64_t>;
inline bool nan_first_greater(const elem_t& x, const elem_t& y) {
    return (std::isnan(x.first) && !std::isnan(y.first)) || (x.first > y.first);
}
}  // namespace

extern "C" int orc_topk(const float* values, int64_t n, int64_t k, int64_t* out_idx) {
    if (k <= 0) return 0;
    if (k > n) return -1;
    std::vector<elem_t> queue(static_cast<size_t>(n));
    for (int64_t j = 0; j < n; ++j) queue[j] = {values[j], j};
    if (k * 64 <= n) {
        std::partial_sort(queue.begin(), queue.begin() + k, queue.end(), nan_first_greater);
    } else {
        std::nth_element(queue.begin(), queue.begin() + k - 1, queue.end(), nan_first_greater);
        std::sort(queue.begin(), queue.begin() + k - 1, nan_first_greater);
    }
    for (int64_t j = 0; j < k; ++j) out_idx[j] = queue[j].second;
    return 0;
}

// rows of a ragged batch: row t holds cnt[t] values at values + t*ld; k[t] in {0,1,2};
// out_idx[t*2 + j] receives the j-th pick (or -1)
extern "C" int orc_topk_rows(const float* values, int64_t rows, int64_t ld, const int64_t* cnt,
                             const int64_t* k, int64_t* out_idx) {
    for (int64_t t = 0; t < rows; ++t) {
        out_idx[2 * t] = out_idx[2 * t + 1] = -1;
        int64_t kk = std::min<int64_t>(k[t], cnt[t]);
        if (kk <= 0) continue;
        if (orc_topk(values + t * ld, cnt[t], kk, out_idx + 2 * t)) return -1;
    }
    return 0;
}
