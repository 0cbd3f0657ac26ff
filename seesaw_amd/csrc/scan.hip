// scan.hip -- brute-force cosine scan: scores[i] = <X[i,:], q>  (gfx950 / MI355X)
//
// Replaces `scores = vectors @ vector.reshape(-1)` of the reference
// (seesaw/indices/multiscale/multiscale_index.py:171, :285, :345;
//  seesaw/indices/coarse/coarse_index.py:38, :73).
//
// Roofline: HBM-bound.  Algorithmic traffic = dim*4 bytes per row read once
// (2048 B at dim=512) + 4 B per row of score written.  No reuse -> no LDS staging:
// rows go straight from HBM to VGPRs with 16-byte loads, a wave reading one whole
// row (2 KiB at dim=512) with `dim/256` fully coalesced 1-KiB wave-instructions.
//
// Work decomposition (64-wide wavefronts):
//   * one wave owns a BATCH of 64 consecutive rows (128 KiB contiguous at dim=512) and
//     walks it in GROUPS of U rows (U = 2 at dim=512); the loads of group g+1 are
//     issued before group g is reduced (register double buffer).  Loads are non-temporal
//     (`global_load_dwordx4 ... nt`): the index is read once per query and must not churn
//     L2 / Infinity Cache.  One 4-wave workgroup per CU turned out to be the fastest
//     residency (see the schedule note below).
//   * lane l accumulates, for one row, two fmaf chains over the elements it loaded
//     (float4 v[c] = X[row, 256*c + 4*l .. +3], c < dim/256):
//         a0 = fma(v[c].x, q.x, a0); a1 = fma(v[c].y, q.y, a1);
//         a0 = fma(v[c].z, q.z, a0); a1 = fma(v[c].w, q.w, a1);     (c ascending)
//     both starting from +0.0f; the lane partial is p = a0 + a1.  (The two chains
//     are what v_pk_fma_f32 computes on a register pair.)
//   * the 64 lane partials of a row are summed by the xor-butterfly
//         v <- v + v[lane ^ off],   off = 1, 2, 4, 8, 16, 32
//     (f32 add is commutative, so every lane ends with the same bits).  For the
//     first log2(U) offsets the U rows of a group are reduced TOGETHER by a
//     transpose-reduce (each exchange halves the number of live registers and leaves
//     lane l with row l % U), which is value-identical to the butterfly; the
//     remaining offsets are plain butterfly steps.  U + 5 - log2(U) exchanges per U
//     rows instead of 6 per row.
//   * lane l = U*g + j then owns the finished score of row j of group g; after the
//     64/U groups of a batch lane l holds the score of row l and the wave stores its
//     64 scores with one coalesced 256-B write.
//   oracle/ssw_oracle.c::ssw_oracle_scores_kernel_order restates exactly this order,
//   which makes the scores BIT-EXACT against the CPU oracle, not merely close.
//
// Grid: persistent, (#CUs x resident blocks per CU) blocks of 256 threads; waves
// stride over the batches.
#include <cstdlib>

#include "ssw_common.h"

namespace ssw {

namespace {

template <int C>
struct RowFrag {
    float4 v[C];
};

// NT: non-temporal loads (`global_load_dwordx4 ... nt`): the index is streamed once per query
// and is far larger than L2 / Infinity Cache, so it should not displace anything.
template <int C, bool NT = false>
__device__ __forceinline__ RowFrag<C> load_row(const float4 *__restrict__ X4, int row, int lane) {
    RowFrag<C> r;
    const float4 *p = X4 + (int64_t)row * (C * 64) + lane;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        if (NT) {
            typedef float f32x4_t __attribute__((ext_vector_type(4)));
            const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(p + c * 64));
            r.v[c] = make_float4(v.x, v.y, v.z, v.w);
        } else {
            r.v[c] = p[c * 64];
        }
    }
    return r;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// the two fmaf chains (a0 over .x/.z, a1 over .y/.w) as packed math on register pairs
template <int C>
__device__ __forceinline__ float dot_frag(const RowFrag<C> &x, const RowFrag<C> &q) {
    f32x2 a = {0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < C; ++c) {
        a = __builtin_elementwise_fma(f32x2{x.v[c].x, x.v[c].y}, f32x2{q.v[c].x, q.v[c].y}, a);
        a = __builtin_elementwise_fma(f32x2{x.v[c].z, x.v[c].w}, f32x2{q.v[c].z, q.v[c].w}, a);
    }
    return a.x + a.y;
}

// U lane-partial registers (acc[j] = row j of the group) -> every lane l holds the
// complete sum of row (l % U): transpose-reduce over offsets 1..U/2, then butterfly
// over offsets U..32.  Canonical offset order 1,2,4,8,16,32 for every U.
template <int U>
__device__ __forceinline__ float group_reduce(float (&acc)[U], int lane) {
#pragma unroll
    for (int off = 1; off < U; off <<= 1) {
        // register distance of the pair merged at this offset == off
        const bool upper = (lane & off) != 0;
#pragma unroll
        for (int i = 0; i < U; i += 2 * off) {
            const float keep = upper ? acc[i + off] : acc[i];
            const float send = upper ? acc[i] : acc[i + off];
            acc[i] = keep + __shfl_xor(send, off, 64);
        }
    }
    float v = acc[0];
#pragma unroll
    for (int off = U; off < 64; off <<= 1) v = v + __shfl_xor(v, off, 64);
    return v;
}

template <int C, int U>
struct Group {
    RowFrag<C> r[U];
};

template <int C, int U, bool NT>
__device__ __forceinline__ void load_group(Group<C, U> &g, const float4 *__restrict__ X4,
                                           int first_row, int last, int lane) {
#pragma unroll
    for (int u = 0; u < U; ++u) g.r[u] = load_row<C, NT>(X4, min(first_row + u, last), lane);
}

template <int C, int U, bool NT>
__global__ __launch_bounds__(256) void scan_scores_kernel(const float *__restrict__ X,
                                                         const float *__restrict__ q,
                                                         float *__restrict__ scores, int n) {
    constexpr int GPB = 64 / U;  // groups per batch
    const int lane = threadIdx.x & 63;
    // wave index through readfirstlane: keeps every row/address computation on the SALU
    // (rows are 32-bit: n < 2^31 - 2^16 is checked by the launcher)
    const int gwave = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = gridDim.x * 4;
    const int nbatches = (n + 63) >> 6;
    const int last = n - 1;
    const float4 *X4 = reinterpret_cast<const float4 *>(X);
    if (gwave >= nbatches) return;

    RowFrag<C> qf;
#pragma unroll
    for (int c = 0; c < C; ++c) qf.v[c] = reinterpret_cast<const float4 *>(q)[c * 64 + lane];

    const int my_group = lane / U;
    Group<C, U> cur, nxt;
    load_group<C, U, NT>(cur, X4, gwave << 6, last, lane);
    for (int b = gwave; b < nbatches; b += nwaves) {
        const int row0 = b << 6;
        const int nb = b + nwaves;
        const int next0 = (nb < nbatches ? nb : b) << 6;  // no next batch: re-touch own rows
        float out = 0.0f;
#pragma unroll 2
        for (int g = 0; g < GPB; ++g) {
            // request group g+1 (or the first group of this wave's next batch) ...
            const int nrow = (g + 1 < GPB) ? row0 + (g + 1) * U : next0;
            load_group<C, U, NT>(nxt, X4, nrow, last, lane);
            // ... then finish group g
            float acc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u] = dot_frag<C>(cur.r[u], qf);
            const float v = group_reduce<U>(acc, lane);
            out = (my_group == g) ? v : out;
            cur = nxt;
        }
        if (row0 + lane < n) scores[row0 + lane] = out;
    }
}

// Small index (fewer rows than the streaming kernel needs to fill the chip: an LVIS-subset index has 14 417): the
// streaming kernel gives each wave 64 rows, fetched two at a time -- 32 dependent round trips to HBM, and only
// rows / 256 workgroups.  A CU streams ~10 B per clock from HBM (MICROARCH guide), so 29 MB on 57 or 113 CUs is 10+ us
// however the loads are scheduled; the bytes have to be spread over every CU.  Here a wave takes U rows per step (all in
// flight at once), four waves a workgroup: 32 rows per workgroup, 451 workgroups for the LVIS subset.  The query comes
// through LDS (one wave's read per workgroup, from L2).  Same dot_frag / group_reduce order: identical bits.  Plain
// loads: an index this size stays in the Infinity Cache between rounds.
template <int C, int U>
__global__ __launch_bounds__(256) void scan_small_kernel(const float *__restrict__ X, const float *__restrict__ q,
                                                         float *__restrict__ scores, int n, int steps) {
    __shared__ float4 ql[C * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int last = n - 1;
    const float4 *X4 = reinterpret_cast<const float4 *>(X);
    const int row_base = (blockIdx.x * 4 + wave) * U * steps;
    Group<C, U> cur, nxt;
    load_group<C, U, false>(cur, X4, row_base, last, lane);
    for (int i = threadIdx.x; i < C * 64; i += 256) ql[i] = reinterpret_cast<const float4 *>(q)[i];
    __syncthreads();
    RowFrag<C> qf;
#pragma unroll
    for (int c = 0; c < C; ++c) qf.v[c] = ql[c * 64 + lane];
    for (int g = 0; g < steps; ++g) {
        const int row0 = row_base + g * U;
        if (row0 >= n) break;
        if (g + 1 < steps) load_group<C, U, false>(nxt, X4, row0 + U, last, lane);
        float acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = dot_frag<C>(cur.r[u], qf);
        const float v = group_reduce<U>(acc, lane);
        if (lane < U && row0 + lane < n) scores[row0 + lane] = v;
        cur = nxt;
    }
}

// scores of an explicit list of rows, same summation order as the full scan (one wave per
// row, plain butterfly) -- stage-2 rescoring against a second vector
// (multiscale_index.py:347-349).
template <int C>
__global__ __launch_bounds__(256) void score_rows_kernel(const float *__restrict__ X,
                                                        const float *__restrict__ q,
                                                        const int64_t *__restrict__ rows, int64_t n,
                                                        float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n) return;
    const float4 *X4 = reinterpret_cast<const float4 *>(X);
    RowFrag<C> qf;
#pragma unroll
    for (int c = 0; c < C; ++c) qf.v[c] = reinterpret_cast<const float4 *>(q)[c * 64 + lane];
    const RowFrag<C> x = load_row<C, false>(X4, (int)rows[w], lane);
    float v = dot_frag<C>(x, qf);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v = v + __shfl_xor(v, off, 64);
    if (lane == 0) out[w] = v;
}

// k-NN graph build, last stage (knn.hip): one wave per row re-scores its <= M candidate columns in
// the scan's summation order (so the scores are the bits a brute-force scan of that row returns),
// orders them by (score desc, row id asc) with a 64-lane bitonic network and certifies the row:
// every column outside the list has fp16-path score <= b, hence exact score <= b + E.
template <int C>
__global__ __launch_bounds__(256) void knn_rescore_kernel(const float *__restrict__ X, const int32_t *__restrict__ perm,
                                                         int r0, int rows, const uint64_t *__restrict__ buf, int cap,
                                                         const unsigned *__restrict__ cnt,
                                                         const unsigned char *__restrict__ overflow, int M,
                                                         const float *__restrict__ norms, float inv_scale2,
                                                         float maxnorm, int k1, int32_t *__restrict__ out_dst,
                                                         float *__restrict__ out_score,
                                                         unsigned char *__restrict__ out_cert) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= rows) return;
    const float4 *X4 = reinterpret_cast<const float4 *>(X);
    const int orig_i = perm[r0 + w];
    const int c = min((int)cnt[w], M);
    const uint64_t key = lane < c ? buf[(int64_t)w * cap + lane] : 0ull;
    const int orig_c = lane < c ? perm[0xFFFFFFFFu - (uint32_t)key] : -1;
    const float approx = ord_to_f32((uint32_t)(key >> 32));  // fp16-path score (scaled)
    const RowFrag<C> xi = load_row<C, false>(X4, orig_i, lane);
    float mine = -INFINITY;
    for (int t = 0; t < c; ++t) {
        const int j = __shfl(orig_c, t, 64);
        const RowFrag<C> xj = load_row<C, false>(X4, j, lane);
        float v = dot_frag<C>(xj, xi);
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) v = v + __shfl_xor(v, off, 64);
        if (lane == t) mine = v;
    }
    uint32_t hi = lane < c ? f32_to_ord(mine) : 0u, lo = lane < c ? 0xFFFFFFFFu - (uint32_t)orig_c : 0u;
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride >= 1; stride >>= 1) {
            const uint32_t ohi = __shfl_xor(hi, stride, 64), olo = __shfl_xor(lo, stride, 64);
            const bool desc = (lane & size) == 0, lower = (lane & stride) == 0;
            const bool other_gt = ohi > hi || (ohi == hi && olo > lo);
            if ((lower == desc) == other_gt) {  // keep the larger key on the descending side's lower lane
                hi = ohi;
                lo = olo;
            }
        }
    }
    const bool have = (hi | lo) != 0u;
    const float score = have ? ord_to_f32(hi) : -INFINITY;
    if (lane < k1) {
        out_dst[(int64_t)orig_i * k1 + lane] = have ? (int32_t)(0xFFFFFFFFu - lo) : -1;
        out_score[(int64_t)orig_i * k1 + lane] = score;
    }
    const float s_k1 = __shfl(score, k1 - 1, 64);
    const float b = __shfl(approx, M - 1, 64) * inv_scale2;  // meaningful when c == M
    if (lane == 0) {
        const float E = 1.2e-3f * norms[orig_i] * maxnorm + 2.5e-5f * maxnorm * maxnorm;
        const bool cert = !overflow[w] && (c < M || (s_k1 - E > b));
        out_cert[orig_i] = cert ? 1 : 0;
    }
}

// Schedule (measured on MI355X, interleaved A/B in one process, tools/sweep_scan.py):
// non-temporal loads are worth +4...7 %, and with them FEWER resident waves stream faster --
// at 100 M rows u2+nt with one block per CU reads 6.70 TB/s, u4 (default policy, 8 waves/SIMD)
// 6.17 TB/s; 1 M rows: 6.66 vs 5.65 TB/s.  All variants produce identical bits.
constexpr int64_t SCAN_SMALL_ROWS = 65536;  // below: the streaming kernel has under one 4-wave workgroup per CU
SSW_TUNABLE bool g_scan_small = true;       // tuning hook (ssw_tune_scan variant -2: streaming kernel at every size)
SSW_TUNABLE int g_scan_variant = -1;        // tuning hook (ssw_tune_scan): -1 = default (u2 + nt)
SSW_TUNABLE int g_scan_blocks_per_cu = -1;  // -1 = default (1 per CU for dim 512), 0 = as many as fit

template <int C, int U, bool NT>
ssw_status launch_scan_t(const float *X, const float *q, float *scores, int64_t n, int device,
                         hipStream_t stream) {
    static int max_blocks_per_cu[16] = {0};
    int dev_slot = device & 15;
    if (max_blocks_per_cu[dev_slot] == 0) {
        int nb = 0;
        SSW_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, scan_scores_kernel<C, U, NT>,
                                                                 256, 0));
        if (nb < 1) nb = 1;
        if (nb > 8) nb = 8;
        max_blocks_per_cu[dev_slot] = nb;
    }
    int blocks_per_cu[16];
    blocks_per_cu[dev_slot] = max_blocks_per_cu[dev_slot];
    const int cap = g_scan_blocks_per_cu < 0 ? (C == 2 ? 1 : 2) : g_scan_blocks_per_cu;
    if (cap >= 1 && cap < blocks_per_cu[dev_slot]) blocks_per_cu[dev_slot] = cap;
    if (n >= (int64_t)0x7fff0000) {
        set_error("scan: n=%lld rows exceeds the 2^31 row limit of one index shard", (long long)n);
        return SSW_ERR_UNSUPPORTED;
    }
    if (n < SCAN_SMALL_ROWS && g_scan_small) {
        constexpr int SU = C <= 2 ? 8 : 4;
        const int64_t per_step = 4 * SU;  // rows a workgroup takes per step
        const int steps = (int)((n + 4096 * per_step - 1) / (4096 * per_step));  // 1 below 131 072 rows
        const int64_t sgrid = (n + per_step * steps - 1) / (per_step * steps);
        hipLaunchKernelGGL((scan_small_kernel<C, SU>), dim3((unsigned)sgrid), dim3(256), 0, stream, X, q, scores, (int)n,
                           steps);
        SSW_HIP_TRY(hipGetLastError());
        return SSW_OK;
    }
    const int64_t nbatches = (n + 63) >> 6;
    int64_t grid = (int64_t)num_cus(device) * blocks_per_cu[dev_slot];
    const int64_t need = (nbatches + 3) / 4;
    if (grid > need) grid = need;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((scan_scores_kernel<C, U, NT>), dim3((unsigned)grid), dim3(256), 0, stream, X,
                       q, scores, (int)n);
    SSW_HIP_TRY(hipGetLastError());
    return SSW_OK;
}

}  // namespace

ssw_status launch_scan(const float *X, const float *q_dev, float *scores, int64_t n, int32_t dim,
                       int device, hipStream_t stream) {
    if (n <= 0) return SSW_OK;
    switch (dim) {
        case 256: return launch_scan_t<1, 8, true>(X, q_dev, scores, n, device, stream);
        case 512: {
            // variants differ in schedule only (rows per group, load policy); numerics are identical
            switch (g_scan_variant) {
                case 0: return launch_scan_t<2, 4, false>(X, q_dev, scores, n, device, stream);
                case 2: return launch_scan_t<2, 8, false>(X, q_dev, scores, n, device, stream);
                case 3: return launch_scan_t<2, 8, true>(X, q_dev, scores, n, device, stream);
                case 4: return launch_scan_t<2, 2, true>(X, q_dev, scores, n, device, stream);
                case 1: return launch_scan_t<2, 4, true>(X, q_dev, scores, n, device, stream);
                default: return launch_scan_t<2, 2, true>(X, q_dev, scores, n, device, stream);
            }
        }
        case 768: return launch_scan_t<3, 2, true>(X, q_dev, scores, n, device, stream);
        case 1024: return launch_scan_t<4, 2, true>(X, q_dev, scores, n, device, stream);
        default:
            set_error("scan: dim=%d unsupported (need a multiple of 256, <= 1024)", dim);
            return SSW_ERR_UNSUPPORTED;
    }
}

ssw_status launch_score_rows(const float *X, const float *q_dev, const int64_t *rows_dev, int64_t n,
                             int32_t dim, float *out, hipStream_t stream) {
    if (n <= 0) return SSW_OK;
    const dim3 grid((unsigned)((n + 3) / 4)), block(256);
    switch (dim) {
        case 256: hipLaunchKernelGGL(score_rows_kernel<1>, grid, block, 0, stream, X, q_dev, rows_dev, n, out); break;
        case 512: hipLaunchKernelGGL(score_rows_kernel<2>, grid, block, 0, stream, X, q_dev, rows_dev, n, out); break;
        case 768: hipLaunchKernelGGL(score_rows_kernel<3>, grid, block, 0, stream, X, q_dev, rows_dev, n, out); break;
        case 1024: hipLaunchKernelGGL(score_rows_kernel<4>, grid, block, 0, stream, X, q_dev, rows_dev, n, out); break;
        default:
            set_error("score_rows: dim=%d unsupported", dim);
            return SSW_ERR_UNSUPPORTED;
    }
    SSW_HIP_TRY(hipGetLastError());
    return SSW_OK;
}

ssw_status launch_knn_rescore(const float *X, int32_t dim, const int32_t *perm, int r0, int rows, const uint64_t *buf,
                              int cap, const unsigned *cnt, const unsigned char *overflow, int M, const float *norms,
                              float scale, float maxnorm, int k1, int32_t *out_dst, float *out_score,
                              unsigned char *out_cert, hipStream_t stream) {
    if (rows <= 0) return SSW_OK;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    const float inv_scale2 = 1.0f / (scale * scale);
#define SSW_KNN_RESCORE(C)                                                                                          \
    hipLaunchKernelGGL(knn_rescore_kernel<C>, grid, block, 0, stream, X, perm, r0, rows, buf, cap, cnt, overflow, M, \
                       norms, inv_scale2, maxnorm, k1, out_dst, out_score, out_cert)
    switch (dim) {
        case 256: SSW_KNN_RESCORE(1); break;
        case 512: SSW_KNN_RESCORE(2); break;
        case 768: SSW_KNN_RESCORE(3); break;
        case 1024: SSW_KNN_RESCORE(4); break;
        default:
            set_error("knn_rescore: dim=%d unsupported", dim);
            return SSW_ERR_UNSUPPORTED;
    }
#undef SSW_KNN_RESCORE
    SSW_HIP_TRY(hipGetLastError());
    return SSW_OK;
}

#ifdef SSW_DEBUG_HOOKS
void tune_scan(int variant, int blocks_per_cu) {
    g_scan_small = variant == -1;  // an explicit variant (or -2) means the streaming kernel at every size
    if (variant < -1) variant = -1;
    g_scan_variant = variant;
    g_scan_blocks_per_cu = blocks_per_cu;
}
#endif

}  // namespace ssw
