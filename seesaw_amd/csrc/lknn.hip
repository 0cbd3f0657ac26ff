// lknn.hip -- L-KNN active search: the two-step look-ahead score of every node in one pass (gfx950 / MI355X).
//
// Replaces _top_sum as _opt_expected_utility_helper_lknn2 calls it
// (seesaw/research/active_search/efficient_nonmyopic_search.py:94-205; model: seesaw/loops/LKNN_model.py:76-281):
//   for every node i:  E_y(i) = sum of the K largest scores among all OTHER unlabelled nodes, had i been labelled y
//   (labelling i changes only the scores of i's D neighbours: (num_j + y) / (den_j + 1));
//   value(i) = s_i (1 + E_1(i)) + (1 - s_i) E_0(i);   next node = nanargmax(value).
// The reference materialises, for every node, the K + D globally best scores plus its D neighbours' new scores --
// an N x (K + 2D) f64 matrix, twice -- and argsorts every row (seconds at 10^5 nodes, O(N (K+2D)) memory).
//
// Here nothing is materialised.  The K + D globally best (id, score) pairs are shared by all rows and already
// sorted; a row only (a) strikes out of that list itself and those of its neighbours that are in it (each node's
// rank in the list is a 4-byte gather), and (b) merges what is left with its D neighbours' new scores, sorted in
// registers -- a two-pointer merge that stops after K picks.  One thread per row, O(K + D log D) work, no LDS
// beyond the shared list, D random 20-byte gathers per row: bound by those gathers (lines out of L2 / Infinity
// Cache), nowhere near HBM.
//
// Bit-exactness: the K picked values come out in descending order, exactly the row numpy sums, and the sum is
// formed in numpy's pairwise order for a contiguous row of K <= 128 doubles (8 running sums over the elements
// 0..8 floor(K/8), combined ((0+1)+(2+3))+((4+5)+(6+7)), then the tail one by one; plain left-to-right below
// 8 elements).  Divisions and the final combination are single IEEE operations, as numpy's elementwise ops.
#include <vector>

#include "ssw_common.h"

// Compiled with -ffp-contract=off (csrc/Makefile): results of this file are compared bit for bit with numpy /
// scipy / torch, so every product and sum must round on its own (hipcc would contract a * b + c into an fma).

namespace ssw {
namespace {

constexpr int LK_THREADS = 256;
constexpr int LK_MAX_D = 32;      // neighbours per node held in registers
constexpr int LK_MAX_LIST = 192;  // K + D entries of the shared list (3 x 64-bit strike-out mask)

// numpy's pairwise summation of a contiguous row of n <= 128 doubles (np.add.reduce over the last axis), fed in order:
// n < 8: res = 0.; res += a[i].  Otherwise r[j] = a[j] (j < 8); r[j] += a[i + j] over i = 8, 16, ... < n - n % 8;
// res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)); then res += a[i] for the n % 8 tail elements.
struct RowSum {
    double r[8];
    double res;
    int n, i, body;
    bool combined;
    __device__ void init(int count) {
        n = count;
        i = 0;
        body = count < 8 ? 0 : count - (count % 8);
        res = 0.0;
        combined = false;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = 0.0;
    }
    __device__ void add(double v) {
        if (n < 8) {
            res = __dadd_rn(res, v);  // res = 0.; res += a[i]
        } else if (i < 8) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (q == i) r[q] = v;
        } else if (i < body) {
            const int j = i & 7;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (q == j) r[q] = __dadd_rn(r[q], v);
        } else {
            if (!combined) {
                res = __dadd_rn(__dadd_rn(__dadd_rn(r[0], r[1]), __dadd_rn(r[2], r[3])),
                                __dadd_rn(__dadd_rn(r[4], r[5]), __dadd_rn(r[6], r[7])));
                combined = true;
            }
            res = __dadd_rn(res, v);
        }
        ++i;
    }
    __device__ double result() {
        if (n >= 8 && !combined)
            res = __dadd_rn(__dadd_rn(__dadd_rn(r[0], r[1]), __dadd_rn(r[2], r[3])),
                            __dadd_rn(__dadd_rn(r[4], r[5]), __dadd_rn(r[6], r[7])));
        return res;
    }
};

__global__ void k_lknn_mark(int32_t *__restrict__ rank_in_list, const int32_t *__restrict__ list_ids, int L) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < L) rank_in_list[list_ids[j]] = j;
}

__global__ void k_lknn_unmark(int32_t *__restrict__ rank_in_list, const int32_t *__restrict__ list_ids, int L) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < L) rank_in_list[list_ids[j]] = -1;
}

// numer: numerators + gamma (-inf for labelled nodes); denom: denominators + 1
__global__ __launch_bounds__(LK_THREADS) void k_lknn_top_sum(int64_t n, int D, int K, int L,
                                                             const int32_t *__restrict__ nbr /* [n, D] ascending */,
                                                             const double *__restrict__ numer,
                                                             const double *__restrict__ denom,
                                                             const int32_t *__restrict__ list_ids /* [L] score desc */,
                                                             const int32_t *__restrict__ rank_in_list,
                                                             double *__restrict__ out) {
    __shared__ double lscore[LK_MAX_LIST];
    for (int j = threadIdx.x; j < L; j += LK_THREADS) {
        const int id = list_ids[j];
        lscore[j] = numer[id] / denom[id];
    }
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * LK_THREADS + threadIdx.x;
    if (i >= n) return;
    unsigned long long gone[3] = {0ull, 0ull, 0ull};  // list entries struck out for this row
    auto strike = [&](int rk) {
        if (rk >= 0) gone[rk >> 6] |= 1ull << (rk & 63);
    };
    strike(rank_in_list[i]);
    double s1[LK_MAX_D], s0[LK_MAX_D];
    for (int d = 0; d < LK_MAX_D; ++d) {
        if (d < D) {
            const int j = nbr[i * D + d];
            strike(rank_in_list[j]);
            const double nd = __dadd_rn(denom[j], 1.0);
            const double nj = numer[j];
            const bool self = j == (int)i;
            s0[d] = self ? -INFINITY : __ddiv_rn(nj, nd);
            s1[d] = self ? -INFINITY : __ddiv_rn(__dadd_rn(nj, 1.0), nd);
        } else {
            s0[d] = s1[d] = -INFINITY;
        }
    }
    // the neighbours' new scores in descending order (insertion sort in registers; D <= 32)
    auto sort_desc = [&](double *a) {
        for (int p = 1; p < LK_MAX_D; ++p) {
            if (p >= D) break;
            const double v = a[p];
            int q = p - 1;
            while (q >= 0 && a[q] < v) {
                a[q + 1] = a[q];
                --q;
            }
            a[q + 1] = v;
        }
    };
    sort_desc(s0);
    sort_desc(s1);
    auto top_sum = [&](const double *nb) -> double {
        RowSum acc;
        acc.init(K);
        int a = 0, b = 0;
        for (int picked = 0; picked < K; ++picked) {
            while (a < L && ((gone[a >> 6] >> (a & 63)) & 1ull)) ++a;
            const double va = a < L ? lscore[a] : -INFINITY;
            const double vb = b < D ? nb[b] : -INFINITY;
            if (va >= vb) {
                acc.add(va);
                ++a;
            } else {
                acc.add(vb);
                ++b;
            }
        }
        return acc.result();
    };
    const double e1 = top_sum(s1), e0 = top_sum(s0);
    const double s = __ddiv_rn(numer[i], denom[i]);
    // scores * (1 + expected1) + (1 - scores) * expected0, one rounding per operation as numpy evaluates it
    out[i] = __dadd_rn(__dmul_rn(s, __dadd_rn(1.0, e1)), __dmul_rn(__dadd_rn(1.0, -s), e0));
}

// np.nanargmax: the first index of the largest non-NaN value
__global__ __launch_bounds__(LK_THREADS) void k_lknn_argmax(const double *__restrict__ v, int64_t n,
                                                            double *__restrict__ best_v, long long *__restrict__ best_i) {
    __shared__ double sv[LK_THREADS];
    __shared__ long long si[LK_THREADS];
    double bv = -INFINITY;
    long long bi = -1;
    for (int64_t i = (int64_t)blockIdx.x * LK_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * LK_THREADS) {
        const double x = v[i];
        if (x != x) continue;
        if (bi < 0 || x > bv) {
            bv = x;
            bi = i;
        }
    }
    sv[threadIdx.x] = bv;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = LK_THREADS / 2; s >= 1; s >>= 1) {
        if (threadIdx.x < s) {
            const double ov = sv[threadIdx.x + s];
            const long long oi = si[threadIdx.x + s];
            const bool take = oi >= 0 && (si[threadIdx.x] < 0 || ov > sv[threadIdx.x] ||
                                          (ov == sv[threadIdx.x] && oi < si[threadIdx.x]));
            if (take) {
                sv[threadIdx.x] = ov;
                si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        best_v[blockIdx.x] = sv[0];
        best_i[blockIdx.x] = si[0];
    }
}

}  // namespace
}  // namespace ssw

using namespace ssw;

struct ssw_lknn {
    int device = 0;
    int64_t n = 0;
    int D = 0;
    int32_t *nbr = nullptr;           // [n, D] ascending per row
    double *numer = nullptr, *denom = nullptr, *value = nullptr;  // [n]
    int32_t *rank_in_list = nullptr;  // [n], -1 outside the shared list
    int32_t *list_ids = nullptr;      // [LK_MAX_LIST]
    double *blk_v = nullptr;
    long long *blk_i = nullptr;
    int argmax_blocks = 0;
    hipStream_t stream = nullptr;
};

extern "C" {

ssw_status ssw_lknn_destroy(ssw_lknn *h) {
    if (!h) return SSW_OK;
    DeviceGuard guard(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (void *p : {(void *)h->nbr, (void *)h->numer, (void *)h->denom, (void *)h->value, (void *)h->rank_in_list,
                    (void *)h->list_ids, (void *)h->blk_v, (void *)h->blk_i})
        (void)hipFree(p);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return SSW_OK;
}

ssw_status ssw_lknn_create(int32_t device, int64_t n, int32_t D, const int32_t *neighbors_sorted_host, ssw_lknn **out) {
    SSW_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    SSW_REQUIRE(n > 0 && D >= 1 && D <= LK_MAX_D && neighbors_sorted_host, "lknn: n=%lld, D=%d (1..%d)", (long long)n, D, LK_MAX_D);
    SSW_REQUIRE(n < (int64_t)0x7fffffff, "lknn: node ids are 32-bit");
    for (int64_t i = 0; i < n; ++i)
        for (int d = 0; d < D; ++d) {
            const int32_t j = neighbors_sorted_host[i * D + d];
            SSW_REQUIRE(j >= 0 && j < n, "lknn: neighbour %d of node %lld out of range", j, (long long)i);
            SSW_REQUIRE(d == 0 || neighbors_sorted_host[i * D + d - 1] <= j, "lknn: neighbours of node %lld are not sorted", (long long)i);
        }
    DeviceGuard guard(device);
    ssw_lknn *h = new (std::nothrow) ssw_lknn();
    if (!h) return SSW_ERR_NOMEM;
    h->device = device;
    h->n = n;
    h->D = D;
    h->argmax_blocks = (int)std::min<int64_t>(1024, (n + LK_THREADS - 1) / LK_THREADS);
    bool ok = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc((void **)&h->nbr, (size_t)n * D * 4) == hipSuccess &&
              hipMalloc((void **)&h->numer, (size_t)n * 8) == hipSuccess &&
              hipMalloc((void **)&h->denom, (size_t)n * 8) == hipSuccess &&
              hipMalloc((void **)&h->value, (size_t)n * 8) == hipSuccess &&
              hipMalloc((void **)&h->rank_in_list, (size_t)n * 4) == hipSuccess &&
              hipMalloc((void **)&h->list_ids, LK_MAX_LIST * 4) == hipSuccess &&
              hipMalloc((void **)&h->blk_v, (size_t)h->argmax_blocks * 8) == hipSuccess &&
              hipMalloc((void **)&h->blk_i, (size_t)h->argmax_blocks * 8) == hipSuccess &&
              hipMemcpy(h->nbr, neighbors_sorted_host, (size_t)n * D * 4, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemset(h->rank_in_list, 0xff, (size_t)n * 4) == hipSuccess;
    if (!ok) {
        ssw_lknn_destroy(h);
        set_error("lknn: allocation / upload failed");
        return SSW_ERR_NOMEM;
    }
    *out = h;
    return SSW_OK;
}

ssw_status ssw_lknn_top_sum(ssw_lknn *h, const double *numer_host, const double *denom_host, const int32_t *top_ids_desc,
                            int32_t K, double *out_values_or_null, int64_t *out_best_idx, double *out_best_value) {
    SSW_REQUIRE(h && numer_host && denom_host && top_ids_desc, "NULL argument");
    const int L = K + h->D;
    SSW_REQUIRE(K >= 1 && K <= 128, "lknn: K=%d outside [1, 128] (the row sum follows numpy's order for one 128-element block)", K);
    SSW_REQUIRE(L <= LK_MAX_LIST && L <= h->n, "lknn: K + D = %d exceeds %d or the node count", L, LK_MAX_LIST);
    for (int j = 0; j < L; ++j)
        SSW_REQUIRE(top_ids_desc[j] >= 0 && top_ids_desc[j] < h->n, "lknn: list id %d out of range", top_ids_desc[j]);
    DeviceGuard guard(h->device);
    hipStream_t s = h->stream;
    const int64_t n = h->n;
    SSW_HIP_TRY(hipMemcpyAsync(h->numer, numer_host, (size_t)n * 8, hipMemcpyHostToDevice, s));
    SSW_HIP_TRY(hipMemcpyAsync(h->denom, denom_host, (size_t)n * 8, hipMemcpyHostToDevice, s));
    SSW_HIP_TRY(hipMemcpyAsync(h->list_ids, top_ids_desc, (size_t)L * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_lknn_mark, dim3(1), dim3(LK_THREADS), 0, s, h->rank_in_list, h->list_ids, L);
    hipLaunchKernelGGL(k_lknn_top_sum, dim3((unsigned)((n + LK_THREADS - 1) / LK_THREADS)), dim3(LK_THREADS), 0, s, n, h->D,
                       (int)K, L, h->nbr, h->numer, h->denom, h->list_ids, h->rank_in_list, h->value);
    hipLaunchKernelGGL(k_lknn_unmark, dim3(1), dim3(LK_THREADS), 0, s, h->rank_in_list, h->list_ids, L);
    hipLaunchKernelGGL(k_lknn_argmax, dim3(h->argmax_blocks), dim3(LK_THREADS), 0, s, h->value, n, h->blk_v, h->blk_i);
    SSW_HIP_TRY(hipGetLastError());
    std::vector<double> bv((size_t)h->argmax_blocks);
    std::vector<long long> bi((size_t)h->argmax_blocks);
    SSW_HIP_TRY(hipMemcpyAsync(bv.data(), h->blk_v, bv.size() * 8, hipMemcpyDeviceToHost, s));
    SSW_HIP_TRY(hipMemcpyAsync(bi.data(), h->blk_i, bi.size() * 8, hipMemcpyDeviceToHost, s));
    if (out_values_or_null) SSW_HIP_TRY(hipMemcpyAsync(out_values_or_null, h->value, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    SSW_HIP_TRY(hipStreamSynchronize(s));
    long long best = -1;
    double bval = 0.0;
    for (size_t b = 0; b < bv.size(); ++b) {
        if (bi[b] < 0) continue;
        if (best < 0 || bv[b] > bval || (bv[b] == bval && bi[b] < best)) {
            best = bi[b];
            bval = bv[b];
        }
    }
    if (out_best_idx) *out_best_idx = best;
    if (out_best_value) *out_best_value = bval;
    return SSW_OK;
}

}  // extern "C"
