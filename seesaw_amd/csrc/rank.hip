// rank.hip -- the sort-based rank-loss functions of the reference as counting kernels (gfx950 / MI355X).
//
// Replaces (seesaw/rank_loss.py:109-187, seesaw/pairwise_rank_loss.py:24-43):
//   quick_pairwise_gradient_zero_margin  two stable lexicographic sorts -> net position change of every item
//   _CheapPairwiseRankingLoss            |gradient| / total_pairs forward, gradient / total_pairs backward
//   compute_inversions                   per item: opposite-label items ranked on the wrong side of it
// The reference obtains an item's rank by SORTING (torch.sort, stable, twice per lexicographic key).  A rank is
// also a COUNT -- pos(x) = #{y : key(y) < key(x)} -- and counting needs no sort network, no scratch and no
// tie-breaking rules beyond the key itself: one thread per item walks all n items once (LDS tiles) and counts
// under both keys at the same time.  n is the labelled set of a feedback round (tens to a few thousand, at most
// 10^4 with pseudo-labels): O(n^2) compares = 10^8 at the very top, microseconds of VALU work, and the result is
// exact integers -- identical to the reference's for any input, ties included:
//   first sort   key1 = (target asc, score asc, original index asc)          [lexicographic_sort(target, scores)]
//   second sort  key2 = (score asc, target DESC, position under key1 asc)    [lexicographic_sort(scores, -target)]
//   gradient(x)  = 2 (pos2(x) - pos1(x));  max_reversals(x) = n - #{y : target_y == target_x}
// Latency-bound; nothing here touches HBM beyond 8 n bytes.
#include "ssw_common.h"

namespace ssw {
namespace {

constexpr int RK_THREADS = 256;

__global__ __launch_bounds__(RK_THREADS) void k_rank_quick(const float *__restrict__ target,
                                                           const float *__restrict__ scores, int n,
                                                           float *__restrict__ out_grad, float *__restrict__ out_maxrev,
                                                           unsigned long long *__restrict__ out_total) {
    __shared__ float st[RK_THREADS], ss[RK_THREADS];
    const int x = blockIdx.x * RK_THREADS + threadIdx.x;
    const bool act = x < n;
    const float tx = act ? target[x] : 0.f, sx = act ? scores[x] : 0.f;
    int pos1 = 0, pos2 = 0, same = 0;
    for (int base = 0; base < n; base += RK_THREADS) {
        const int y0 = base + threadIdx.x;
        st[threadIdx.x] = y0 < n ? target[y0] : 0.f;
        ss[threadIdx.x] = y0 < n ? scores[y0] : 0.f;
        __syncthreads();
        const int lim = min(RK_THREADS, n - base);
        for (int j = 0; j < lim; ++j) {
            const int y = base + j;
            const float ty = st[j], sy = ss[j];
            // key1(y) < key1(x): (target, score, index)
            const bool k1 = ty < tx || (ty == tx && (sy < sx || (sy == sx && y < x)));
            // key2(y) < key2(x): (score, -target, pos1); with score and target equal, pos1 orders by index
            const bool k2 = sy < sx || (sy == sx && (ty > tx || (ty == tx && y < x)));
            pos1 += k1;
            pos2 += k2;
            same += (ty == tx);
        }
        __syncthreads();
    }
    if (act) {
        out_grad[x] = 2.f * (float)(pos2 - pos1);
        if (out_maxrev) out_maxrev[x] = (float)(n - same);
        if (out_total) atomicAdd(out_total, (unsigned long long)(n - same));  // total_pairs = n^2 - sum of class sizes squared
    }
}

// compute_inversions (pairwise_rank_loss.py:24-43): in descending score order (stable: ties by index), a positive
// counts the negatives ranked before it, a negative the positives ranked after it
__global__ __launch_bounds__(RK_THREADS) void k_rank_inversions(const unsigned char *__restrict__ labs,
                                                                const float *__restrict__ scores, int n,
                                                                long long *__restrict__ out) {
    __shared__ float ss[RK_THREADS];
    __shared__ unsigned char sl[RK_THREADS];
    const int x = blockIdx.x * RK_THREADS + threadIdx.x;
    const bool act = x < n;
    const float sx = act ? scores[x] : 0.f;
    const bool lx = act ? labs[x] != 0 : false;
    long long cnt = 0;
    for (int base = 0; base < n; base += RK_THREADS) {
        const int y0 = base + threadIdx.x;
        ss[threadIdx.x] = y0 < n ? scores[y0] : 0.f;
        sl[threadIdx.x] = y0 < n ? labs[y0] : 0;
        __syncthreads();
        const int lim = min(RK_THREADS, n - base);
        for (int j = 0; j < lim; ++j) {
            const int y = base + j;
            const bool before = ss[j] > sx || (ss[j] == sx && y < x);  // y ranked before x (descending, stable)
            const bool after = ss[j] < sx || (ss[j] == sx && y > x);
            const bool ly = sl[j] != 0;
            cnt += lx ? (!ly && before) : (ly && after);
        }
        __syncthreads();
    }
    if (act) out[x] = cnt;
}

}  // namespace
}  // namespace ssw

namespace ssw {
// device-resident form for the feedback engine: target / scores / outputs are device pointers
ssw_status launch_rank_quick(const float *target_dev, const float *scores_dev, int n, float *grad_dev,
                             float *maxrev_dev_or_null, unsigned long long *total_dev_or_null, hipStream_t stream) {
    if (n <= 0) return SSW_OK;
    hipLaunchKernelGGL(k_rank_quick, dim3((unsigned)((n + RK_THREADS - 1) / RK_THREADS)), dim3(RK_THREADS), 0, stream,
                       target_dev, scores_dev, n, grad_dev, maxrev_dev_or_null, total_dev_or_null);  // NULL outputs are skipped
    SSW_HIP_TRY(hipGetLastError());
    return SSW_OK;
}
}  // namespace ssw

using namespace ssw;

extern "C" {

ssw_status ssw_rank_quick_gradient(int32_t device, const float *target_host, const float *scores_host, int32_t n,
                                   float *out_grad, float *out_max_reversals, int64_t *out_total_pairs) {
    SSW_REQUIRE(n >= 0 && n <= SSW_RANK_MAX_ITEMS, "rank_quick_gradient: n = %d outside [0, %d]", n, SSW_RANK_MAX_ITEMS);
    if (out_total_pairs) *out_total_pairs = 0;
    if (n == 0) return SSW_OK;
    SSW_REQUIRE(target_host && scores_host && out_grad, "NULL argument");
    DeviceGuard guard(device);
    float *buf = nullptr;  // target | scores | grad | maxrev
    unsigned long long *total = nullptr;
    SSW_HIP_TRY(hipMalloc((void **)&buf, (size_t)4 * n * sizeof(float) + sizeof(unsigned long long) + 16));
    total = reinterpret_cast<unsigned long long *>(buf + (((size_t)4 * n + 3) / 4) * 4);
    auto run = [&]() -> ssw_status {
        SSW_HIP_TRY(hipMemcpy(buf, target_host, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
        SSW_HIP_TRY(hipMemcpy(buf + n, scores_host, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
        SSW_HIP_TRY(hipMemset(total, 0, sizeof(unsigned long long)));
        hipLaunchKernelGGL(k_rank_quick, dim3((unsigned)((n + RK_THREADS - 1) / RK_THREADS)), dim3(RK_THREADS), 0, 0, buf,
                           buf + n, (int)n, buf + 2 * n, buf + 3 * n, total);
        SSW_HIP_TRY(hipGetLastError());
        SSW_HIP_TRY(hipMemcpy(out_grad, buf + 2 * n, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
        if (out_max_reversals)
            SSW_HIP_TRY(hipMemcpy(out_max_reversals, buf + 3 * n, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
        unsigned long long t = 0;
        SSW_HIP_TRY(hipMemcpy(&t, total, sizeof(t), hipMemcpyDeviceToHost));
        if (out_total_pairs) *out_total_pairs = (int64_t)t;
        return SSW_OK;
    };
    const ssw_status st = run();
    (void)hipFree(buf);
    return st;
}

ssw_status ssw_rank_inversions(int32_t device, const uint8_t *labels_host, const float *scores_host, int32_t n,
                               int64_t *out_inversions) {
    SSW_REQUIRE(n >= 0 && n <= SSW_RANK_MAX_ITEMS, "rank_inversions: n = %d outside [0, %d]", n, SSW_RANK_MAX_ITEMS);
    if (n == 0) return SSW_OK;
    SSW_REQUIRE(labels_host && scores_host && out_inversions, "NULL argument");
    DeviceGuard guard(device);
    unsigned char *buf = nullptr;
    const size_t off_s = (((size_t)n + 15) / 16) * 16, off_o = off_s + (((size_t)n * 4 + 15) / 16) * 16;
    SSW_HIP_TRY(hipMalloc((void **)&buf, off_o + (size_t)n * 8));
    auto run = [&]() -> ssw_status {
        SSW_HIP_TRY(hipMemcpy(buf, labels_host, (size_t)n, hipMemcpyHostToDevice));
        SSW_HIP_TRY(hipMemcpy(buf + off_s, scores_host, (size_t)n * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_rank_inversions, dim3((unsigned)((n + RK_THREADS - 1) / RK_THREADS)), dim3(RK_THREADS), 0, 0,
                           buf, reinterpret_cast<const float *>(buf + off_s), (int)n,
                           reinterpret_cast<long long *>(buf + off_o));
        SSW_HIP_TRY(hipGetLastError());
        SSW_HIP_TRY(hipMemcpy(out_inversions, buf + off_o, (size_t)n * 8, hipMemcpyDeviceToHost));
        return SSW_OK;
    };
    const ssw_status st = run();
    (void)hipFree(buf);
    return st;
}

}  // extern "C"
