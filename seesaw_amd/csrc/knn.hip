// knn.hip -- exact k-NN graph construction over the resident index (gfx950 / MI355X)
//
// Replaces compute_exact_knn of the reference (seesaw/knn_graph.py:170-191):
//     all_pairs = 1 - X @ X.T ;  nn = argsort(all_pairs, axis=-1)[:, :k+1]
// which is O(N^2 d) and only feasible for small N there (the production graphs come from the
// approximate pynndescent, knn_graph.py:194-243).  Here the graph is EXACT at any N that fits HBM:
//
//   1. candidates   S~ = Xh Xh^T with Xh = fp16(2^e X) on the matrix cores (v_mfma_f32_16x16x32_f16,
//                   the LDS-DMA tile pipeline of gemm_bf16.hip: 128 x 128 tiles, 8 x 8 super-tiles per
//                   XCD so both operand panels stay in that XCD's L2).  Nothing is stored: the epilogue
//                   compares the accumulators with a per-row threshold and appends the few survivors
//                   (score, column) to a per-row buffer.
//   2. thresholds   columns are visited in a random order (seeded permutation) in geometrically growing
//                   levels (512, x <= 16, ..., N); after each level a per-row LDS bitonic sort keeps the best M = 32 entries and
//                   raises the row's threshold to the M-th score, so a level appends ~ M x ratio
//                   entries per row whatever the data looks like (exchangeability of the permutation).
//   3. exact scores the <= M candidates of every row are re-scored in f32 in the scan kernel's fixed
//                   summation order (scan.hip), sorted by (score desc, row id asc), and the best k+1
//                   are returned -- the same bits and order as a brute-force scan of that row.
//   (symmetry)    S~ is symmetric: when the buffers of all rows fit in HBM at once, the last level -- 97 % of
//                   the work -- computes only the tiles J >= I and an off-diagonal tile feeds both its rows
//                   and its columns; the earlier levels (thresholds for every row) stay rectangular.
//   4. certificate  a column outside the candidate list has fp16-path score <= b (the M-th kept), and
//                   |fp16-path - exact| <= E_i (derived below), so the row is PROVEN exact when
//                   exact(k+1-th) - E_i > b.  Rows that fail (or whose buffer overflowed) are flagged;
//                   the caller reruns them through the ordinary exact scan (ssw_index_topk).
//
// Error bound E_i: fp16 rounding is <= 2^-11 relative per operand (scaled components lie in
// [2^-14, 256), values below 2^-14 may be flushed), so the product sum differs from the exact one by
// at most (2^-10 + 2^-22) sum|x_d y_d| <= 2^-10 |x||y|; the f32 accumulation of 512 terms adds
// <= 512 * 2^-23 |x||y| on either side (truncating adds assumed) and the flushed tails <= 2^-21 maxabs sqrt(D)
// (|x_i| + max|x|) <= 1.1e-5 (|x_i| + max|x|) max|x| for D <= 1024... E_i is set to
// 1.2e-3 |x_i| max|x| + 2.5e-5 max|x|^2, which covers all of it with margin.
//
// Roofline: MFMA-bound (2 N^2 d flop, fp16 dense peak 2.5 PFLOP/s); operand traffic per 128^2 tile is
// 256 KB out of L2, i.e. the same L2 -> LDS ceiling as the tower GEMM.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "ssw_common.h"

namespace ssw {
namespace {

typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int KNN_M_MAX = 64;   // candidates kept per row between levels: 32 (k <= 15) or 64 (k <= 31)
constexpr int KNN_CAP = 1024;   // row buffer capacity: entries appended within one level (~ M x ratio <= 512) + M
constexpr int KT = 128;         // tile edge
constexpr int KBK = 64;         // k-step
constexpr int K_OPER = 16384;   // one operand image: 128 rows x 128 B
constexpr int K_STAGE = 2 * K_OPER;

// ---------------------------------------------------------------------------------------
// preparation
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_knn_stats(const float *__restrict__ X, int64_t n, int D,
                                                   float *__restrict__ norms, unsigned *__restrict__ maxima) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), waves = (int64_t)gridDim.x * 4;
    float mx_all = 0.f, nr_all = 0.f;
    for (int64_t r = wave; r < n; r += waves) {  // grid-stride: one pair of atomics per wave, not per row
        const float *x = X + r * D;
        float ss = 0.f, mx = 0.f;
        for (int c = lane; c < D; c += 64) {
            const float v = x[c];
            ss = fmaf(v, v, ss);
            mx = fmaxf(mx, fabsf(v));
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            ss += __shfl_xor(ss, off, 64);
            mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        }
        const float nr = sqrtf(ss) * 1.0001f;  // upper bound of the norm (f32 sum of squares: ~3e-5 relative)
        if (lane == 0) norms[r] = nr;
        mx_all = fmaxf(mx_all, mx);
        nr_all = fmaxf(nr_all, nr);
    }
    if (lane == 0) {
        atomicMax(&maxima[0], __float_as_uint(mx_all));  // non-negative floats order like their bits
        atomicMax(&maxima[1], __float_as_uint(nr_all));
    }
}

// Xh[p, :] = fp16(scale * X[perm[p], :])
__global__ void k_knn_convert(const float *__restrict__ X, const int32_t *__restrict__ perm, int64_t n, int D,
                              float scale, f16 *__restrict__ Xh) {
    const int64_t chunks = n * (D / 8);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / (D / 8);
        const int c = (int)(i % (D / 8)) * 8;
        const float4 *src = reinterpret_cast<const float4 *>(X + (int64_t)perm[p] * D + c);
        const float4 a = src[0], b = src[1];
        f16x8 o;
        o[0] = (f16)(a.x * scale); o[1] = (f16)(a.y * scale); o[2] = (f16)(a.z * scale); o[3] = (f16)(a.w * scale);
        o[4] = (f16)(b.x * scale); o[5] = (f16)(b.y * scale); o[6] = (f16)(b.z * scale); o[7] = (f16)(b.w * scale);
        *reinterpret_cast<f16x8 *>(Xh + p * D + c) = o;
    }
}

// rows_padded - rows trailing entries get thr = +inf (nothing is ever appended to them)
__global__ void k_knn_reset(float *__restrict__ thr, unsigned *__restrict__ cnt, unsigned char *__restrict__ overflow,
                            int rows, int rows_padded) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows_padded) {
        thr[i] = i < rows ? -INFINITY : INFINITY;
        cnt[i] = 0;
        overflow[i] = 0;
    }
}

// ---------------------------------------------------------------------------------------
// candidate pass: S~ tile on the matrix cores, thresholded append
// ---------------------------------------------------------------------------------------

// rows [r0, r1) of Xh against columns [c0, c1) of Xh; thr / cnt / buf are indexed by row - r0.
// SYM (r0 = c0 = 0, r1 = c1 = n): S~ is symmetric, so only tiles J >= I of the square are computed and an
// off-diagonal tile serves both its rows (entries (i, j)) and its columns (entries (j, i)); tiles whose
// columns all lie below tile index `mirror_from` were covered by the earlier (non-symmetric) levels and are
// skipped, and the mirrored entries of tiles with I < mirror_from likewise.
template <bool SYM>
__global__ __launch_bounds__(512) void k_knn_gemm_filter(const f16 *__restrict__ Xh, int D, int r0, int r1, int c0,
                                                         int c1, const float *__restrict__ thr,
                                                         unsigned *__restrict__ cnt, uint64_t *__restrict__ buf,
                                                         int i_tiles, int j_tiles, int sj_count, int n_super,
                                                         int mirror_from) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // 8 x 8 super-tiles dealt round-robin to the XCDs (block ids b and b + 8 share an XCD): the 8
    // workgroups of a super-tile (one per row-tile, each walking the 8 column tiles in the same order) run
    // on one XCD and share the operand panels through its L2
    // (the grid is two-dimensional only because one dimension is limited to 2^32 work-items)
    const int64_t bid = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = (int)(bid & 7);
    const int64_t local = bid >> 3;
    if ((local >> 3) * 8 + xcd >= n_super) return;
    const int g = (int)(local >> 3) * 8 + xcd;
    const int within = (int)(local & 7);  // row-tile of the super-tile; the workgroup walks its 8 column tiles
    int SI, SJ;
    if (SYM) {  // g enumerates the super-tile pairs SI <= SJ row by row: row SI starts at SI*ns - SI(SI-1)/2
        const int ns = sj_count;
        const double b = 2.0 * ns + 1.0;
        SI = (int)((b - sqrt(b * b - 8.0 * (double)g)) * 0.5);
        SI = max(0, min(SI, ns - 1));
        while (SI > 0 && SI * ns - SI * (SI - 1) / 2 > g) --SI;
        while (SI + 1 < ns && (SI + 1) * ns - (SI + 1) * SI / 2 <= g) ++SI;
        SJ = SI + (g - (SI * ns - SI * (SI - 1) / 2));
    } else {
        SI = g / sj_count;
        SJ = g % sj_count;
    }
    const int I = SI * 8 + within;
    if (I >= i_tiles) return;
    // column tiles of this workgroup: [j_lo, j_hi)
    int j_lo = SJ * 8, j_hi = min(SJ * 8 + 8, j_tiles);
    if (SYM) j_lo = max(j_lo, max(I, mirror_from));
    if (j_lo >= j_hi) return;
    const int m0 = r0 + I * KT;
    const int wm = wave >> 2, wn = wave & 3;  // 8 waves: 2 x 4, each 64 rows x 32 columns of the 128 x 128 tile

    const f16 *a_src[2];
    int b_chunk[2], b_row[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {  // 16 pieces (8 rows x 128 B) per operand image, 2 per wave
        const int row = (wave * 2 + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        a_src[i] = Xh + (int64_t)min(m0 + row, r1 - 1) * D + chunk * 8;
        b_row[i] = row;
        b_chunk[i] = chunk * 8;
    }
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char *)smem);
    const unsigned a_dst = lds0 + wave * 2048, b_dst = lds0 + K_OPER + wave * 2048;
    // stage (column tile J, k-step kt) -> ring slot
#define KNN_ISSUE(J, kt, slot)                                                                         \
    {                                                                                                  \
        const int k0 = (kt) * KBK, nb = c0 + (J) * KT;                                                 \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                  \
            glds16(a_src[i] + k0, a_dst + (slot) * K_STAGE + i * 1024);                                \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                  \
            glds16(Xh + (int64_t)min(nb + b_row[i], c1 - 1) * D + b_chunk[i] + k0,                     \
                   b_dst + (slot) * K_STAGE + i * 1024);                                               \
    }
    const int fr = lane & 15, fq = lane >> 4;
    const int frag0 = fr * 128 + ((fq ^ (fr >> 1)) << 4);
    const int a_frag = wm * 8192 + frag0, b_frag = K_OPER + wn * 4096 + frag0;
    // thresholds of this lane's four rows (accumulator row = m0 + wm*64 + i*16 + fr): the same for every
    // column tile of the walk
    float trow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = m0 + wm * 64 + i * 16 + fr;
        trow[i] = row < r1 ? thr[row - r0] : INFINITY;
    }
    asm volatile("" ::"s"(cnt), "s"(buf), "s"(c1));  // all scalar loads done before the loop (gemm_bf16.hip)
    const int nk = D / KBK;
    KNN_ISSUE(j_lo, 0, 0)
    int slot = 0;
    for (int J = j_lo; J < j_hi; ++J) {
        f32x4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // the ring never drains between tiles: the first stage of the next column tile is requested
            // while this tile's last k-step multiplies and its epilogue runs
            if (kt + 1 < nk) {
                KNN_ISSUE(J, kt + 1, slot ^ 1)
            } else if (J + 1 < j_hi) {
                KNN_ISSUE(J + 1, 0, slot ^ 1)
            }
            const unsigned char *sb = smem + slot * K_STAGE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 a[4], b[2];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    a[i] = *reinterpret_cast<const f16x8 *>(sb + ((a_frag + i * 2048) ^ (ks * 64)));
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    b[j] = *reinterpret_cast<const f16x8 *>(sb + ((b_frag + j * 2048) ^ (ks * 64)));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
            }
            asm volatile("" ::: "memory");
            slot ^= 1;
        }
        const int n0 = c0 + J * KT;
        const bool mirror = SYM && I != J && I >= mirror_from;
        // acc[i][j][r] = S~[row m0 + wm*64 + i*16 + fr][column n0 + wn*32 + j*16 + fq*4 + r]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float tr = trow[i];
            const int lrow = m0 + wm * 64 + i * 16 + fr - r0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 v = acc[i][j];
                if (fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])) > tr) {
                    const int col = n0 + wn * 32 + j * 16 + fq * 4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (v[r] > tr && col + r < c1) {
                            const unsigned pos = atomicAdd(&cnt[lrow], 1u);
                            if (pos < (unsigned)KNN_CAP)
                                buf[(int64_t)lrow * KNN_CAP + pos] =
                                    ((uint64_t)f32_to_ord(v[r]) << 32) | (uint64_t)(0xFFFFFFFFu - (unsigned)(col + r));
                        }
                    }
                }
            }
        }
        if (mirror) {  // entry (j, i) of the symmetric matrix: this tile's columns as rows, its rows as columns
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wn * 32 + j * 16 + fq * 4;  // thr is padded to a tile multiple with +inf
                const f32x4 tc = *reinterpret_cast<const f32x4 *>(thr + col);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 v = acc[i][j];
                    const int row = m0 + wm * 64 + i * 16 + fr;
                    if (row < r1 && (v[0] > tc[0] || v[1] > tc[1] || v[2] > tc[2] || v[3] > tc[3])) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (v[r] > tc[r]) {
                                const unsigned pos = atomicAdd(&cnt[col + r], 1u);
                                if (pos < (unsigned)KNN_CAP)
                                    buf[(int64_t)(col + r) * KNN_CAP + pos] =
                                        ((uint64_t)f32_to_ord(v[r]) << 32) | (uint64_t)(0xFFFFFFFFu - (unsigned)row);
                            }
                        }
                    }
                }
            }
        }
    }
#undef KNN_ISSUE
}

// ---------------------------------------------------------------------------------------
// the symmetric last level on 256 x 256 tiles (round 2)
// ---------------------------------------------------------------------------------------
// The loop of gemm_bf16.hip's gemm_256 (its comment has the schedule and the hazards): 8 waves of 128 x 64, two LDS
// stages of four 128-row halves {A0, A1, W0, W1}, a K-tile in four phases of 16 MFMAs, halves restaged as soon as their
// last fragment read is behind a barrier.  A wave that owns 128 x 64 reads 0.375 fragments per MFMA against 0.75 in
// the 128 x 128 kernel above, whose K-loop is bound by LDS read bandwidth.  Here the K-tile sequence runs on across the
// workgroup's column tiles (step s = tile * nk + kt): the first K-tiles of the next column tile are requested while
// the current tile's last phases multiply and its accumulators are filtered, so the ring never drains.
// Tiles, super-tiles (8 x 8 tiles per XCD group), the upper-triangle enumeration and `mirror_from` mean what they mean
// above with an edge of 256; c0 (the columns done by earlier levels) must be a multiple of 256.
constexpr int KT2 = 256;
constexpr int K2_HALF = 16384;           // 128 rows x 128 B
constexpr int K2_STAGE = 4 * K2_HALF;    // A0 A1 W0 W1

__global__ __launch_bounds__(512) void k_knn_gemm_filter256(const f16 *__restrict__ Xh, int D, int n,
                                                            const float *__restrict__ thr, unsigned *__restrict__ cnt,
                                                            uint64_t *__restrict__ buf, int i_tiles, int sj_count,
                                                            int n_super, int mirror_from) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int64_t bid = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = (int)(bid & 7);
    const int64_t local = bid >> 3;
    if ((local >> 3) * 8 + xcd >= n_super) return;
    const int g = (int)(local >> 3) * 8 + xcd;
    const int within = (int)(local & 7);
    int SI, SJ;
    {  // g enumerates the super-tile pairs SI <= SJ row by row: row SI starts at SI*ns - SI(SI-1)/2
        const int ns = sj_count;
        const double b = 2.0 * ns + 1.0;
        SI = (int)((b - sqrt(b * b - 8.0 * (double)g)) * 0.5);
        SI = max(0, min(SI, ns - 1));
        while (SI > 0 && SI * ns - SI * (SI - 1) / 2 > g) --SI;
        while (SI + 1 < ns && (SI + 1) * ns - (SI + 1) * SI / 2 <= g) ++SI;
        SJ = SI + (g - (SI * ns - SI * (SI - 1) / 2));
    }
    const int I = SI * 8 + within;
    if (I >= i_tiles) return;
    const int j_lo = max(SJ * 8, max(I, mirror_from)), j_hi = min(SJ * 8 + 8, i_tiles);
    if (j_lo >= j_hi) return;
    const int m0 = I * KT2;

    // staging: a half = 16 pieces of 8 rows x 128 B; wave w issues pieces 2w and 2w + 1 of every half
    const f16 *a_src[2], *a_src1[2];
    int p_row[2], p_chunk[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = (wave * 2 + p) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);  // the same for row and row + 128
        p_row[p] = row;
        p_chunk[p] = chunk * 8;
        a_src[p] = Xh + (int64_t)min(m0 + row, n - 1) * D + chunk * 8;
        a_src1[p] = Xh + (int64_t)min(m0 + 128 + row, n - 1) * D + chunk * 8;
    }
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char *)smem);
    const unsigned piece_dst = lds0 + wave * 2048;
#define K2_ISSUE_A0(k0, st) { glds16(a_src[0] + (k0), piece_dst + (st) * K2_STAGE); \
                              glds16(a_src[1] + (k0), piece_dst + (st) * K2_STAGE + 1024); }
#define K2_ISSUE_A1(k0, st) { glds16(a_src1[0] + (k0), piece_dst + (st) * K2_STAGE + K2_HALF); \
                              glds16(a_src1[1] + (k0), piece_dst + (st) * K2_STAGE + K2_HALF + 1024); }
#define K2_ISSUE_W(J, half, k0, st)                                                                              \
    {                                                                                                            \
        const int nb = (J) * KT2 + (half) * 128;                                                                 \
        _Pragma("unroll") for (int p = 0; p < 2; ++p)                                                            \
            glds16(Xh + (int64_t)min(nb + p_row[p], n - 1) * D + p_chunk[p] + (k0),                              \
                   piece_dst + (st) * K2_STAGE + (2 + (half)) * K2_HALF + p * 1024);                             \
    }
    const int fr = lane & 15, fq = lane >> 4;
    const int frag0 = fr * 128 + ((fq ^ (fr >> 1)) << 4);
    const int a_frag = wr * K2_HALF + frag0;                                       // + i * 2048, i = 0..7
    const int w_frag = (2 + (wc >> 1)) * K2_HALF + (wc & 1) * 4 * 2048 + frag0;    // + j * 2048, j = 0..3
#define K2_READ_A(st, half, dst)                                                                         \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                        \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                 \
            dst[i][ks] = *reinterpret_cast<const f16x8 *>(smem + (st) * K2_STAGE + ((a_frag + ((half) * 4 + i) * 2048) ^ (ks * 64)));
#define K2_READ_W(st, half, dst)                                                                         \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                        \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                 \
            dst[j][ks] = *reinterpret_cast<const f16x8 *>(smem + (st) * K2_STAGE + ((w_frag + ((half) * 2 + j) * 2048) ^ (ks * 64)));
#define K2_MFMA(ahalf, a, whalf, b)                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                                       \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                     \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                    \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                \
                acc[(ahalf) * 4 + i][(whalf) * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(          \
                    b[j][ks], a[i][ks], acc[(ahalf) * 4 + i][(whalf) * 2 + j], 0, 0, 0);                  \
    __builtin_amdgcn_s_setprio(0);
#define K2_LGKM0_BARRIER()                                                                               \
    __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) */                                                 \
    __builtin_amdgcn_s_barrier();                                                                        \
    asm volatile("" ::: "memory");

    // thresholds of this lane's eight rows (accumulator row = m0 + wr*128 + i*16 + fr)
    float trow[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = m0 + wr * 128 + i * 16 + fr;
        trow[i] = row < n ? thr[row] : INFINITY;
    }
    asm volatile("" ::"s"(cnt), "s"(buf), "s"(thr));  // scalar loads done before the loop (gemm_bf16.hip)
    const int nk = D / KBK;
    const int S = (j_hi - j_lo) * nk;  // K-tile steps of this workgroup
    // cursors of the steps s + 1 and s + 2 (column tile, k offset in elements)
    int J1 = j_lo, k1 = KBK, J2 = j_lo, k2 = 2 * KBK;
    if (nk == 1) { J1 = j_lo + 1; k1 = 0; }
    if (nk <= 2) { J2 = j_lo + (nk == 1 ? 2 : 1); k2 = 0; }
    // prologue: step 0 complete, then W0 W1 A0 of step 1
    K2_ISSUE_A0(0, 0) K2_ISSUE_A1(0, 0) K2_ISSUE_W(j_lo, 0, 0, 0) K2_ISSUE_W(j_lo, 1, 0, 0)
    if (S > 1) {
        K2_ISSUE_W(J1, 0, k1, 1) K2_ISSUE_W(J1, 1, k1, 1) K2_ISSUE_A0(k1, 1)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 a[4][2], b0[2][2], b1[2][2];
    int J = j_lo, kt = 0;
    for (int s_ = 0; s_ < S; ++s_) {
        const int st = s_ & 1, ot = st ^ 1;
        // phase 1
        K2_READ_A(st, 0, a)
        K2_READ_W(st, 0, b0)
        if (s_ + 1 < S) K2_ISSUE_A1(k1, ot)
        __builtin_amdgcn_sched_barrier(0);
        K2_MFMA(0, a, 0, b0)
        __builtin_amdgcn_sched_barrier(0);
        // phase 2
        K2_READ_W(st, 1, b1)
        K2_LGKM0_BARRIER()      // every wave's W reads of this stage are done
        K2_MFMA(0, a, 1, b1)
        __builtin_amdgcn_sched_barrier(0);
        // phase 3
        K2_READ_A(st, 1, a)
        if (s_ + 2 < S) K2_ISSUE_W(J2, 0, k2, st)
        K2_LGKM0_BARRIER()      // every wave's A reads of this stage are done
        K2_MFMA(1, a, 1, b1)
        __builtin_amdgcn_sched_barrier(0);
        // phase 4
        if (s_ + 2 < S) {
            K2_ISSUE_W(J2, 1, k2, st) K2_ISSUE_A0(k2, st)
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // step s + 1 has landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        K2_MFMA(1, a, 0, b0)
        __builtin_amdgcn_sched_barrier(0);
        // advance the cursors
        k1 = k2; J1 = J2;
        k2 += KBK;
        if (k2 == nk * KBK) { k2 = 0; ++J2; }
        if (++kt < nk) continue;
        kt = 0;
        // ---- this column tile is complete: acc[i][j][r] = S~[m0 + wr*128 + i*16 + fr][n0 + wc*64 + j*16 + fq*4 + r]
        const int n0 = J * KT2;
        const bool mirror = I != J && I >= mirror_from;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float tr = trow[i];
            const int row = m0 + wr * 128 + i * 16 + fr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 v = acc[i][j];
                if (fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])) > tr) {
                    const int col = n0 + wc * 64 + j * 16 + fq * 4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (v[r] > tr && col + r < n) {
                            const unsigned pos = atomicAdd(&cnt[row], 1u);
                            if (pos < (unsigned)KNN_CAP)
                                buf[(int64_t)row * KNN_CAP + pos] =
                                    ((uint64_t)f32_to_ord(v[r]) << 32) | (uint64_t)(0xFFFFFFFFu - (unsigned)(col + r));
                        }
                    }
                }
            }
        }
        if (mirror) {  // entry (j, i) of the symmetric matrix: this tile's columns as rows, its rows as columns
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = n0 + wc * 64 + j * 16 + fq * 4;  // thr is padded to a tile multiple with +inf
                const f32x4 tc = *reinterpret_cast<const f32x4 *>(thr + col);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x4 v = acc[i][j];
                    const int row = m0 + wr * 128 + i * 16 + fr;
                    if (row < n && (v[0] > tc[0] || v[1] > tc[1] || v[2] > tc[2] || v[3] > tc[3])) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (v[r] > tc[r]) {
                                const unsigned pos = atomicAdd(&cnt[col + r], 1u);
                                if (pos < (unsigned)KNN_CAP)
                                    buf[(int64_t)(col + r) * KNN_CAP + pos] =
                                        ((uint64_t)f32_to_ord(v[r]) << 32) | (uint64_t)(0xFFFFFFFFu - (unsigned)row);
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        ++J;
    }
#undef K2_ISSUE_A0
#undef K2_ISSUE_A1
#undef K2_ISSUE_W
#undef K2_READ_A
#undef K2_READ_W
#undef K2_MFMA
#undef K2_LGKM0_BARRIER
}

// ---------------------------------------------------------------------------------------
// per row: keep the best M entries (score desc, column asc), raise the threshold
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_knn_compact(uint64_t *__restrict__ buf, unsigned *__restrict__ cnt,
                                                     float *__restrict__ thr, unsigned char *__restrict__ overflow,
                                                     int rows, int M) {
    __shared__ uint64_t keys[KNN_CAP];
    const int row = blockIdx.x;
    if (row >= rows) return;
    unsigned c = cnt[row];
    if (c > (unsigned)KNN_CAP) {  // entries were dropped: the row cannot be certified any more
        if (threadIdx.x == 0) overflow[row] = 1;
        c = KNN_CAP;
    }
    if (c <= 1) return;  // nothing to order (thr stays -inf below M entries)
    int P = 64;
    while (P < (int)c) P <<= 1;
    uint64_t *rb = buf + (int64_t)row * KNN_CAP;
    for (int i = threadIdx.x; i < P; i += 256) keys[i] = i < (int)c ? rb[i] : 0ull;
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride >= 1; stride >>= 1) {
            for (int i = threadIdx.x; i < P / 2; i += 256) {
                const int lo = 2 * i - (i & (stride - 1));  // index with bit `stride` clear
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const uint64_t a = keys[lo], b = keys[hi];
                if ((a < b) == desc) {
                    keys[lo] = b;
                    keys[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    const int keep = (int)c < M ? (int)c : M;
    for (int i = threadIdx.x; i < keep; i += 256) rb[i] = keys[i];
    if (threadIdx.x == 0) {
        cnt[row] = keep;
        if ((int)c >= M) thr[row] = ord_to_f32((uint32_t)(keys[M - 1] >> 32));
    }
}

struct KnnScratch {
    int32_t *perm = nullptr;
    float *norms = nullptr;
    unsigned *maxima = nullptr;
    f16 *Xh = nullptr;
    float *thr = nullptr;
    unsigned *cnt = nullptr;
    unsigned char *overflow = nullptr;
    uint64_t *buf = nullptr;
    int32_t *out_dst = nullptr;
    float *out_score = nullptr;
    unsigned char *out_cert = nullptr;
    void release() {
        for (void *p : {(void *)perm, (void *)norms, (void *)maxima, (void *)Xh, (void *)thr, (void *)cnt,
                        (void *)overflow, (void *)buf, (void *)out_dst, (void *)out_score, (void *)out_cert})
            (void)hipFree(p);
    }
};

inline uint64_t splitmix64(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

}  // namespace
}  // namespace ssw

using namespace ssw;

extern "C" ssw_status ssw_knn_build(ssw_index *index, int32_t k, uint64_t seed, int32_t *out_dst_host,
                                    float *out_score_host, uint8_t *out_certified_host) {
    SSW_REQUIRE(index != nullptr && out_dst_host != nullptr && out_score_host != nullptr &&
                    out_certified_host != nullptr,
                "NULL argument");
    int64_t n = 0, n_images = 0;
    int32_t D = 0;
    void *Xv = nullptr, *scores_unused = nullptr;
    SSW_TRY(ssw_index_shape(index, &n, &D, &n_images));
    SSW_TRY(ssw_index_device_ptrs(index, &Xv, &scores_unused));
    SSW_REQUIRE(n >= 1 && n < (int64_t)0x7fff0000, "ssw_knn_build: n=%lld out of range", (long long)n);
    SSW_REQUIRE(k >= 1 && k + 1 <= KNN_M_MAX / 2, "ssw_knn_build: k=%d out of range (1..%d)", k, KNN_M_MAX / 2 - 1);
    const int M = (k + 1 <= 16) ? 32 : 64;  // twice the list length: the margin the certificate lives on
    if (D != 256 && D != 512 && D != 768 && D != 1024) {
        set_error("ssw_knn_build: dim=%d unsupported", D);
        return SSW_ERR_UNSUPPORTED;
    }
    SSW_TRY(ssw_index_sync(index));
    int device = 0;
    SSW_HIP_TRY(hipGetDevice(&device));
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, Xv) == hipSuccess) device = attr.device;
    DeviceGuard guard(device);
    hipStream_t s = nullptr;  // one-off build: the default stream orders everything
    const float *X = static_cast<const float *>(Xv);
    const int k1 = k + 1;

    KnnScratch sc;
    ssw_status rc = SSW_OK;
    auto fail = [&](hipError_t e, const char *what) {
        set_error("ssw_knn_build: %s: %s", what, hipGetErrorString(e));
        sc.release();
        return e == hipErrorOutOfMemory ? SSW_ERR_NOMEM : SSW_ERR_HIP;
    };
#define KNN_HIP(expr)                                             \
    if (hipError_t _e = (expr); _e != hipSuccess) return fail(_e, #expr)

    // All rows at once (symmetric last level, half the matrix work) when their buffers fit comfortably,
    // otherwise batches of 131072 rows against every column (buffer 4.3 GB per batch).
    size_t free_b = 0, total_b = 0;
    KNN_HIP(hipMemGetInfo(&free_b, &total_b));
    const char *force = getenv("SSW_KNN_FORCE_BATCHED");
    const bool sym = n > 512 && n <= 8000000 && !(force && force[0] == '1') &&  // (one workgroup per row in k_knn_compact)
                     (double)n * (KNN_CAP * 8.0 + D * 2.0 + 64.0) < 0.6 * (double)free_b;
    const int64_t RB = sym ? n : std::min<int64_t>(n, 131072);
    const int64_t RBP = (RB + KT2 - 1) / KT2 * KT2;  // thresholds padded (+inf) to the larger tile edge
    KNN_HIP(hipMalloc((void **)&sc.perm, (size_t)n * 4));
    KNN_HIP(hipMalloc((void **)&sc.norms, (size_t)n * 4));
    KNN_HIP(hipMalloc((void **)&sc.maxima, 8));
    KNN_HIP(hipMalloc((void **)&sc.Xh, (size_t)n * D * 2));
    KNN_HIP(hipMalloc((void **)&sc.thr, (size_t)RBP * 4));
    KNN_HIP(hipMalloc((void **)&sc.cnt, (size_t)RBP * 4));
    KNN_HIP(hipMalloc((void **)&sc.overflow, (size_t)RBP));
    KNN_HIP(hipMalloc((void **)&sc.buf, (size_t)RBP * KNN_CAP * 8));
    KNN_HIP(hipMalloc((void **)&sc.out_dst, (size_t)n * k1 * 4));
    KNN_HIP(hipMalloc((void **)&sc.out_score, (size_t)n * k1 * 4));
    KNN_HIP(hipMalloc((void **)&sc.out_cert, (size_t)n));

    // random column order (Fisher-Yates, splitmix64)
    {
        std::vector<int32_t> perm((size_t)n);
        for (int64_t i = 0; i < n; ++i) perm[(size_t)i] = (int32_t)i;
        uint64_t st = seed ^ 0x5EE5A3D1ull;
        for (int64_t i = n - 1; i > 0; --i) {
            const uint64_t j = splitmix64(st) % (uint64_t)(i + 1);
            std::swap(perm[(size_t)i], perm[(size_t)j]);
        }
        KNN_HIP(hipMemcpy(sc.perm, perm.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    }
    KNN_HIP(hipMemsetAsync(sc.maxima, 0, 8, s));
    hipLaunchKernelGGL(k_knn_stats, dim3((unsigned)std::min<int64_t>((n + 3) / 4, 4096)), dim3(256), 0, s, X, n, (int)D, sc.norms,
                       sc.maxima);
    float maxima[2] = {0.f, 0.f};
    KNN_HIP(hipMemcpy(maxima, sc.maxima, 8, hipMemcpyDeviceToHost));
    const float maxabs = maxima[0], maxnorm = maxima[1];
    if (!(maxabs > 0.f) || !std::isfinite(maxabs)) {
        set_error("ssw_knn_build: the index holds no finite non-zero vector");
        sc.release();
        return SSW_ERR_NUMERIC;
    }
    int e2 = 0;
    (void)std::frexp(maxabs, &e2);                 // maxabs in [2^(e2-1), 2^e2)
    const float scale = std::ldexp(1.0f, 8 - e2);  // scale * maxabs in [128, 256)
    hipLaunchKernelGGL(k_knn_convert, dim3(4096), dim3(256), 0, s, X, sc.perm, n, (int)D, scale, sc.Xh);
    KNN_HIP(hipGetLastError());
    KNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_knn_gemm_filter<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 2 * K_STAGE));
    KNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_knn_gemm_filter<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 2 * K_STAGE));
    KNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_knn_gemm_filter256),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 2 * K2_STAGE));
    // The symmetric last level can run on 256 x 256 tiles (SSW_KNN_TILE256_FROM = smallest n that takes them; the
    // tests set 0).  It is NOT the default: measured at 1.56 M rows it takes 1.396 s against 1.369 s for the 128 x 128
    // kernel (both ~0.9 PFLOP/s of real MFMA work).  Counters for the 256 kernel: MFMA pipes busy 35 %, waves in
    // s_waitcnt / barriers 51 % of their cycles, L2 hit 91 % at 7.5 TB/s, HBM 0.5 TB/s; with the MFMAs removed the
    // same loop streams its operands 4.7x faster -- neither operand delivery nor the fragment reads bound it, the
    // barrier-synchronised phases do.  A third structure was measured and dropped: 128 x 256 x 32 tiles, four waves per
    // workgroup, two workgroups per CU (so that a SIMD's two waves drift apart), three 24-KB stages, one barrier per
    // K-step -- bit-exact as well, 1.553 s.
    const char *tile_env = getenv("SSW_KNN_TILE256_FROM");
    const int64_t big_from = tile_env ? atoll(tile_env) : (int64_t)1 << 62;
    const bool big_tiles = sym && n >= big_from;

    // level boundaries over the permuted columns (multiples of the tile edge): 512, then a constant
    // ratio <= 512 / M up to n: a level appends ~ M x ratio <= 512 entries per row (+- 90), the buffers hold 1024
    std::vector<int64_t> bounds;
    {
        const int64_t b0 = std::min<int64_t>(n, 512);
        bounds.push_back(b0);
        if (n > b0) {
            const double span = (double)n / (double)b0;
            const int levels = std::max(1, (int)std::ceil(std::log(span) / std::log(512.0 / M)));
            const double ratio = std::pow(span, 1.0 / levels);
            for (int l = 1; l < levels; ++l) {
                const int64_t b = (int64_t)std::llround((double)b0 * std::pow(ratio, l));
                bounds.push_back(std::min<int64_t>(n, (b + KT2 - 1) / KT2 * KT2));
            }
            bounds.push_back(n);
        }
    }
    for (int64_t r0 = 0; r0 < n && rc == SSW_OK; r0 += RB) {
        const int64_t r1 = std::min(n, r0 + RB);
        const int rows = (int)(r1 - r0);
        const int rows_padded = (rows + KT2 - 1) / KT2 * KT2;
        hipLaunchKernelGGL(k_knn_reset, dim3((rows_padded + 255) / 256), dim3(256), 0, s, sc.thr, sc.cnt, sc.overflow, rows,
                           rows_padded);
        int64_t c0 = 0;
        for (size_t li = 0; li < bounds.size(); ++li) {
            const int64_t c1 = bounds[li];
            if (c1 <= c0) continue;
            const bool last_sym = sym && li + 1 == bounds.size() && li > 0;
            if (last_sym && big_tiles && c0 % KT2 == 0) {
                const int i_tiles2 = (int)((n + KT2 - 1) / KT2);
                const int sj2 = (i_tiles2 + 7) / 8;
                const int64_t n_super2 = (int64_t)sj2 * (sj2 + 1) / 2;
                const int64_t blocks2 = ((n_super2 + 7) / 8) * 8 * 8;
                const int64_t gx2 = std::min<int64_t>(blocks2, 1 << 22), gy2 = (blocks2 + gx2 - 1) / gx2;
                auto kern256 = k_knn_gemm_filter256;
                hipEvent_t ev0 = nullptr, ev1 = nullptr;
                const bool timing = getenv("SSW_KNN_TIMING") != nullptr;
                if (timing) {
                    (void)hipEventCreate(&ev0);
                    (void)hipEventCreate(&ev1);
                    (void)hipEventRecord(ev0, s);
                }
                hipLaunchKernelGGL(kern256, dim3((unsigned)gx2, (unsigned)gy2), dim3(512), 2 * K2_STAGE, s, sc.Xh,
                                   (int)D, (int)n, sc.thr, sc.cnt, sc.buf, i_tiles2, sj2, (int)n_super2, (int)(c0 / KT2));
                if (timing) {
                    (void)hipEventRecord(ev1, s);
                    (void)hipEventSynchronize(ev1);
                    float ms = 0.f;
                    (void)hipEventElapsedTime(&ms, ev0, ev1);
                    const double tiles = 0.5 * (double)i_tiles2 * i_tiles2 - 0.5 * (double)(c0 / KT2) * (c0 / KT2);
                    fprintf(stderr, "knn: symmetric level on 256-tiles: %.3f ms, %.0f tiles, %.0f TFLOP/s of MFMA work\n", ms,
                            tiles, tiles * 2.0 * KT2 * KT2 * D / (ms * 1e-3) / 1e12);
                    (void)hipEventDestroy(ev0);
                    (void)hipEventDestroy(ev1);
                }
                hipLaunchKernelGGL(k_knn_compact, dim3(rows), dim3(256), 0, s, sc.buf, sc.cnt, sc.thr, sc.overflow, rows, M);
                c0 = c1;
                continue;
            }
            const int i_tiles = (rows + KT - 1) / KT;
            int j_tiles, sj;
            int64_t n_super;
            if (last_sym) {  // upper triangle of the whole square; columns below c0 were done by the levels above
                j_tiles = i_tiles;
                sj = (i_tiles + 7) / 8;
                n_super = (int64_t)sj * (sj + 1) / 2;
            } else {
                j_tiles = (int)((c1 - c0 + KT - 1) / KT);
                sj = (j_tiles + 7) / 8;
                n_super = (int64_t)((i_tiles + 7) / 8) * sj;
            }
            const int64_t blocks = ((n_super + 7) / 8) * 8 * 8;  // one workgroup per row-tile of a super-tile
            // a grid dimension holds < 2^32 work-items (8.4 M workgroups of 512): fold the rest into y
            const int64_t gx = std::min<int64_t>(blocks, 1 << 22), gy = (blocks + gx - 1) / gx;  // 2^22 x 512 < 2^32
            if (n_super >= (int64_t)0x7fffffff || gy > 65535) {
                set_error("ssw_knn_build: level of %lld x %lld tiles exceeds one launch", (long long)i_tiles,
                          (long long)j_tiles);
                rc = SSW_ERR_UNSUPPORTED;
                break;
            }
            const dim3 grid((unsigned)gx, (unsigned)gy);
            if (last_sym)
                hipLaunchKernelGGL(k_knn_gemm_filter<true>, grid, dim3(512), 2 * K_STAGE, s, sc.Xh, (int)D, 0,
                                   (int)n, 0, (int)n, sc.thr, sc.cnt, sc.buf, i_tiles, j_tiles, sj, (int)n_super,
                                   (int)(c0 / KT));
            else
                hipLaunchKernelGGL(k_knn_gemm_filter<false>, grid, dim3(512), 2 * K_STAGE, s, sc.Xh, (int)D,
                                   (int)r0, (int)r1, (int)c0, (int)c1, sc.thr, sc.cnt, sc.buf, i_tiles, j_tiles, sj,
                                   (int)n_super, 0);
            hipLaunchKernelGGL(k_knn_compact, dim3(rows), dim3(256), 0, s, sc.buf, sc.cnt, sc.thr, sc.overflow, rows, M);
            c0 = c1;
        }
        if (rc != SSW_OK) break;
        rc = launch_knn_rescore(X, D, sc.perm, (int)r0, rows, sc.buf, KNN_CAP, sc.cnt, sc.overflow, M, sc.norms, scale,
                                maxnorm, k1, sc.out_dst, sc.out_score, sc.out_cert, s);
        if (hipError_t e = hipGetLastError(); rc == SSW_OK && e != hipSuccess) return fail(e, "kernel launch");
    }
    if (rc != SSW_OK) {
        sc.release();
        return rc;
    }
    KNN_HIP(hipMemcpyAsync(out_dst_host, sc.out_dst, (size_t)n * k1 * 4, hipMemcpyDeviceToHost, s));
    KNN_HIP(hipMemcpyAsync(out_score_host, sc.out_score, (size_t)n * k1 * 4, hipMemcpyDeviceToHost, s));
    KNN_HIP(hipMemcpyAsync(out_certified_host, sc.out_cert, (size_t)n, hipMemcpyDeviceToHost, s));
    KNN_HIP(hipStreamSynchronize(s));
#undef KNN_HIP
    sc.release();
    return SSW_OK;
}
