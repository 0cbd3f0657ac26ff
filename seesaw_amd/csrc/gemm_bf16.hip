// bf16 GEMM for the CLIP towers:  C[M,N] = A[M,K] * W[N,K]^T  (+ fused epilogue), f32 accumulate.
//
// Both operands are K-contiguous ("NT" form: torch Linear keeps W as [out, in]), so the A and
// the W fragment of v_mfma_f32_16x16x32_bf16 are the same 16-row x 32-k slice read row-wise.
// Replaces the torch/cuBLAS linears inside transformers' CLIPModel that the reference calls
// (seesaw/models/embeddings.py:42-76 -> CLIPVisionModel / CLIPTextModel layers).
//
// variant 14 (default): LDS-DMA pipeline.  A 128x128 tile per 512-thread block (8 waves of 64 x 32: twice the
// waves of variant 2's 64 x 64 per wave for the same tile hide each other's waits, +2...6 %), BK = 64:
//   * staging is global_load_lds_dwordx4 only (no VGPR round trip): one wave-instruction fills
//     8 rows x 128 B of the LDS image, lane-linear, reading whole 128-B lines;
//   * the image is XOR-swizzled through the SOURCE address (16-B chunk c of row r sits at chunk
//     c ^ ((r >> 1) & 7)), the fragment ds_read_b128 applies the same XOR: every 16-lane read
//     group of ds_read_b128 then touches 16 distinct 16-B slots of the 256-B bank row;
//   * DEPTH-stage ring in one __shared__ array, one raw s_barrier per k-step, counted
//     s_waitcnt vmcnt so DEPTH-2 stages stay in flight across the barrier (DEPTH >= 3);
//   * the MFMA takes (W fragment, A fragment), so a lane owns 4 consecutive output COLUMNS of one
//     row and the epilogue stores 8 B (bf16) / 16 B (f32) per lane with float4 bias / residual;
//   * blocks that share an XCD (linear id % 8) walk whole row-tiles of A, so an A tile is pulled
//     into one L2 only.
// variant 0 is the first register-staged kernel, kept as the A/B reference of ssw_debug_gemm; variant 7 is the
// 256 x 128 tile, 8-wave, software-pipelined form (fragments of the next half K-step are read while the current
// one multiplies, barrier between the two MFMA blocks): +2 % at K = 3072, -7 % at K = 768, so not the default.
// Measured and dropped: one LDS buffer with two barriers (-5...-25 %), a 3-stage ring at one workgroup per CU
// (-20 %), 256-row tiles without the pipelined loop (-6 %), the pipelined loop on 128-row tiles (-7 %).
#include "ssw_common.h"
#include <cstdlib>

namespace ssw {
namespace {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ __forceinline__ bf16 to_bf16(float x) { return (bf16)x; }
// epilogue stores, non-temporal or not: bit 0 = the bf16 outputs that leave through LDS (QKV, fc1: 46 / 61 MB that the next
// kernel reads from whichever XCD it lands on), bit 1 = the producers' residual rows.  Measured at B = 200 (forward, f32 /
// bf16 rows, same box, two rounds): neither 2.625 / 2.467 ms, bit 0 2.56 / 2.436, bit 1 2.67-2.70 / 2.46, both 2.64 / 2.435.
#ifndef SSW_NT_STORES
#define SSW_NT_STORES 1
#endif
// the staggered K loop below (waves 4-7 half a K-step behind waves 0-3): bit-identical embeddings, B = 200 forward 2.592 ->
// 2.603 ms (f32 rows), 2.463 -> 2.476 (bf16 rows), three rounds -- with two workgroups a CU a SIMD's four waves drift apart
// by themselves; kept behind the switch
// LDS-DMA with the address as SGPR tile base + 32-bit lane offset (saddr form) instead of a 64-bit pointer per lane:
// identical embeddings, B = 200 forward 2.490 -> 2.486 ms over three rounds (noise); the piece's issue cost is not its
// address registers.  Kept behind the switch.
#ifndef SSW_GEMM_SADDR
#define SSW_GEMM_SADDR 0
#endif
#ifndef SSW_GEMM_STAGGER
#define SSW_GEMM_STAGGER 0
#endif
#ifndef SSW_ABL_LN
#define SSW_ABL_LN 0  // timing-only ablations of the LayerNorm-folded consumers (wrong results): bit 0 no statistics prologue, bit 1 no epilogue arithmetic
#endif
#ifndef SSW_EPI6_FULL_LINES
#define SSW_EPI6_FULL_LINES 1  // the f32-row producers' epilogue with four columns a lane (whole lines per instruction); 0: round 4's eight
#endif
#ifndef SSW_NT_LOADS
#define SSW_NT_LOADS 0  // the producers' residual reads non-temporal: measured 2.56 -> 2.64 ms (f32 rows), 2.43 -> 2.45 (bf16): off
#endif
template <typename V>
__device__ __forceinline__ V epi_load(const V *p) {
    if constexpr (SSW_NT_LOADS != 0) return __builtin_nontemporal_load(p);
    else return *p;
}
template <int BIT, typename V>
__device__ __forceinline__ void epi_store(V *p, V v) {
    if constexpr ((SSW_NT_STORES >> BIT) & 1) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// 4 / 5: LayerNorm folded into the product (GemmLn, ssw_common.h) -> bf16 [+ quick-GELU]; 6: +bias +residual -> f32, plus the
// bf16 copy of the new residual row and its partial LayerNorm statistics for the next product; 7: the same with the
// residual stream itself held in bf16 (read from and written back to GemmLn::xcopy, no f32 row)
enum Epilogue { EPI_F32 = 0, EPI_BF16_BIAS = 1, EPI_BF16_BIAS_GELU = 2, EPI_F32_BIAS_RESIDUAL = 3, EPI_BF16_LN = 4,
                EPI_BF16_LN_GELU = 5, EPI_F32_BIAS_RESIDUAL_STATS = 6, EPI_BF16_STREAM_STATS = 7 };
constexpr bool epi_ln(int e) { return e == EPI_BF16_LN || e == EPI_BF16_LN_GELU; }
constexpr bool epi_bf16_out(int e) { return e == EPI_BF16_BIAS || e == EPI_BF16_BIAS_GELU || epi_ln(e); }
constexpr bool epi_gelu(int e) { return e == EPI_BF16_BIAS_GELU || e == EPI_BF16_LN_GELU; }
constexpr bool epi_residual(int e) { return e == EPI_F32_BIAS_RESIDUAL || e == EPI_F32_BIAS_RESIDUAL_STATS; }
constexpr bool epi_stats(int e) { return e == EPI_F32_BIAS_RESIDUAL_STATS || e == EPI_BF16_STREAM_STATS; }

// LayerNorm folded into a product (no layernorm launch in the tile path).  With x the f32 residual row,
// LN(x) W^T + b = rstd * (x W'^T - mean * c1) + c2,  W' = gamma (.) W (folded once at load, bf16), c1_n = sum_k W'_nk,
// c2_n = sum_k beta_k W_nk + b_n: the product runs on bf16(x) -- which the kernel that produced x wrote next to the f32
// row -- and the row statistics arrive as per-column-tile partial sums (sum, sum of squares) from that kernel's
// epilogue (EPI_F32_BIAS_RESIDUAL_STATS), np_in of them per row, added here in ascending order.

// One epilogue for the tile kernels: acc holds C[row][col .. col + 3] (before bias); rows' LayerNorm statistics (mean,
// rstd) come from LDS (ln_stats[2 * local_row]).
template <int EPI>
__device__ __forceinline__ void epilogue_store(f32x4 v, int64_t o, int col, const float *__restrict__ bias,
                                               const float *__restrict__ residual, void *__restrict__ Cout,
                                               const GemmLn &ln, float mean, float rstd, float *sum, float *sq,
                                               const float *c1_lds = nullptr, const float *c2_lds = nullptr) {
    if constexpr (epi_ln(EPI)) {  // (the accumulators of these variants start from zero; c2 is in `bias`)
        // c1 / c2 of this tile's columns were requested before the K loop and parked in LDS: read from memory here they
        // were a round trip to L2 in every tile's epilogue (+8.5 us on the QKV and fc1 products of the B = 200 tower)
        const f32x4 c1 = *reinterpret_cast<const f32x4 *>(c1_lds), c2 = *reinterpret_cast<const f32x4 *>(c2_lds);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = rstd * (v[r] - mean * c1[r]) + c2[r];
    }
    (void)col;
    if constexpr (epi_gelu(EPI)) {
#pragma unroll
        for (int r = 0; r < 4; ++r)  // quick_gelu: x * sigmoid(1.702 x), v_exp + v_rcp
            v[r] *= __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.702f * 1.44269504f * v[r]));
    }
    if constexpr (epi_residual(EPI)) v += *reinterpret_cast<const f32x4 *>(residual + o);
    if constexpr (EPI == EPI_BF16_STREAM_STATS) {  // the stream's own element: read here, replaced below by this thread
        const bf16x4 rb = *reinterpret_cast<const bf16x4 *>(ln.xcopy + o);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)rb[r];
    } else if constexpr (epi_bf16_out(EPI)) {
        bf16x4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = to_bf16(v[r]);
        *reinterpret_cast<bf16x4 *>(reinterpret_cast<bf16 *>(Cout) + o) = h;
    } else {
        *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(Cout) + o) = v;
    }
    if constexpr (epi_stats(EPI)) {
        bf16x4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = to_bf16(v[r]);
        *reinterpret_cast<bf16x4 *>(ln.xcopy + o) = h;
        if constexpr (EPI == EPI_BF16_STREAM_STATS) {  // statistics of the row as stored: the next product normalises that
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (float)h[r];
        }
        *sum += (v[0] + v[1]) + (v[2] + v[3]);
        *sq += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
}

// bf16 outputs leave through LDS (round 3).  Straight from the accumulators a lane stores the 4 consecutive columns it
// holds -- 8 bytes -- and the 64 lanes of one store instruction touch 16 different rows: 8-byte write requests, 7.6 M
// of them for fc1's 61 MB.  A timing-only ablation with 16 bytes per lane took gemm_256 from 60 to 52 us on fc1 and
// from 55.5 to 47.3 on QKV: the request count, not the bytes, was what the epilogue cost.  So a wave parks RH x 16 rows
// of its sub-tile as bf16 in a private piece of the (by now idle) staging area, rows padded by 16 bytes so the 16
// rows of a write land on different banks, and reads them back row-major: every store instruction then writes whole
// 64- or 128-byte row segments, 16 bytes a lane.  Same values, same bits.
template <int EPI>
__device__ __forceinline__ f32x4 epilogue_value(f32x4 v, float mean, float rstd, const float *c1_lds, const float *c2_lds) {
    if constexpr (epi_ln(EPI) && !(SSW_ABL_LN & 2)) {
        const f32x4 c1 = *reinterpret_cast<const f32x4 *>(c1_lds), c2 = *reinterpret_cast<const f32x4 *>(c2_lds);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = rstd * (v[r] - mean * c1[r]) + c2[r];
    }
    if constexpr (epi_gelu(EPI)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.702f * 1.44269504f * v[r]));
    }
    return v;
}

// LayerNorm statistics of a tile's rows, AHEAD of the K loop (round 4): thread t < rows asks for row m0 + t's np_in
// partial pairs (np_in even, <= 8: four float4 requests, the last ones repeated when there are fewer), adds them in
// ascending order -- the order the producing epilogues' partials have always been added in -- and leaves mean / rstd
// in ln_lds[2 t], [2 t + 1]; the K loop's first barrier publishes them.  The loads go out before the ring's first
// pieces and are waited for behind them, so the wait is the first stage's own.  (They used to be parked in the idle
// staging bytes and summed BEHIND the loop: two barriers and an LDS round trip per tile in the exposed epilogue,
// 1.6 us of QKV's 44.3 and 1.9 of fc1's 62.5 at 10 000 rows.)
struct LnRowPre {
    f32x4 v[4];
};
__device__ __forceinline__ LnRowPre ln_row_request(const GemmLn &ln, int row) {
    LnRowPre r;
    const f32x4 *p = reinterpret_cast<const f32x4 *>(ln.stats_in + (int64_t)row * ln.np_in * 2);
    const int last = ln.np_in / 2 - 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.v[k] = p[min(k, last)];
    return r;
}
__device__ __forceinline__ void ln_row_finish(const GemmLn &ln, const LnRowPre &r, float *dst2) {
    float sm = 0.f, sq = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (2 * k < ln.np_in) {
            sm += r.v[k][0];
            sq += r.v[k][1];
            sm += r.v[k][2];
            sq += r.v[k][3];
        }
    }
    const float mean = sm * ln.inv_dim;
    const float var = fmaxf(sq * ln.inv_dim - mean * mean, 0.f);
    dst2[0] = mean;
    dst2[1] = rsqrtf(var + ln.eps);
}

// acc: [RH][NJ] accumulator tiles of this wave (rows row0 + i*16 + fr, columns col0 + j*16 + fq*4 + r); wl: the wave's
// RH*16 x (NJ*32 + 16) bytes of LDS; ln_rows: statistics of the tile's rows from local row lrow0 on; c1 / c2 at col0
template <int EPI, int RH, int NJ>
__device__ __forceinline__ void store_rows_via_lds(const f32x4 (*acc)[NJ], unsigned char *wl, int lane, int row0, int col0,
                                                   int M, int N, bf16 *__restrict__ C, const float *ln_rows,
                                                   const float *c1_lds, const float *c2_lds) {
    static_assert(epi_bf16_out(EPI), "bf16 outputs only");
    constexpr int RB = NJ * 32 + 16;
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int i = 0; i < RH; ++i) {
        float mean = 0.f, rstd = 1.f;
        if constexpr (epi_ln(EPI)) {
            mean = ln_rows[2 * (i * 16 + fr)];
            rstd = ln_rows[2 * (i * 16 + fr) + 1];
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const f32x4 v = epilogue_value<EPI>(acc[i][j], mean, rstd, c1_lds + j * 16 + fq * 4, c2_lds + j * 16 + fq * 4);
            bf16x4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = to_bf16(v[r]);
            *reinterpret_cast<bf16x4 *>(wl + (i * 16 + fr) * RB + j * 32 + fq * 8) = h;
        }
    }
    constexpr int CH = NJ * 2, RPI = 64 / CH;  // 16-byte chunks per row, rows per store instruction
    const int rr = lane / CH, cc = lane % CH;
#pragma unroll
    for (int it = 0; it < RH * 16 / RPI; ++it) {
        const int row = it * RPI + rr;
        const bf16x8 h = *reinterpret_cast<const bf16x8 *>(wl + row * RB + cc * 16);
        if (row0 + row < M) epi_store<0>(reinterpret_cast<bf16x8 *>(C + (int64_t)(row0 + row) * N + col0 + cc * 8), h);
    }
}


// ---------------------------------------------------------------------------------------
// variant 0: register-staged double buffer (global -> VGPR -> padded LDS), one tile of lookahead
// ---------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDS_STRIDE = BK + 8;  // bf16 elements per staged row (144 B: breaks the 128-B bank period)

// SPLITK (EPI_F32 only): blockIdx.z takes columns [z k_len, (z + 1) k_len) of K and writes its partial products to
// slab z of Cout ([gridDim.z][M][N] f32) -- launch_gemm_splitk_f32 below.
template <int EPI, bool SPLITK = false>
__global__ __launch_bounds__(256) void gemm_bf16_nt(const bf16 *__restrict__ A, const bf16 *__restrict__ W,
                                                    const float *__restrict__ bias,
                                                    const float *__restrict__ residual, void *__restrict__ Cout,
                                                    int M, int N, int K, int k_len = 0) {
    __shared__ __attribute__((aligned(16))) bf16 sA[2][BM * LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) bf16 sB[2][BN * LDS_STRIDE];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;  // 2 x 2 waves, 64 x 64 each
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    // staging: 1024 16-byte chunks per operand tile, 4 per thread (chunk c = t + 256 i:
    // row c/8, 8 bf16 at column (c%8)*8); kept in registers across the MFMA block
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    const int srow = t >> 3, scol = (t & 7) * 8;  // chunk i adds 32 rows
    static_assert(!SPLITK || EPI == EPI_F32, "partial products are plain f32");
    const int kb = SPLITK ? (int)blockIdx.z * k_len : 0;  // first column of this workgroup's K range
    if (SPLITK) Cout = reinterpret_cast<float *>(Cout) + (int64_t)blockIdx.z * M * N;
    const bf16 *a_ptr0 = A + (int64_t)min(m0 + srow, M - 1) * K + scol + kb;
    const bf16 *a_ptr1 = A + (int64_t)min(m0 + srow + 32, M - 1) * K + scol + kb;
    const bf16 *a_ptr2 = A + (int64_t)min(m0 + srow + 64, M - 1) * K + scol + kb;
    const bf16 *a_ptr3 = A + (int64_t)min(m0 + srow + 96, M - 1) * K + scol + kb;
    const bf16 *w_ptr = W + (int64_t)(n0 + srow) * K + scol + kb;
    const int64_t w_step = (int64_t)32 * K;
    const int lds_off = srow * LDS_STRIDE + scol;
#define SSW_LOAD_TILES(k0)                                                   \
    ra0 = *reinterpret_cast<const uint4 *>(a_ptr0 + (k0));                   \
    ra1 = *reinterpret_cast<const uint4 *>(a_ptr1 + (k0));                   \
    ra2 = *reinterpret_cast<const uint4 *>(a_ptr2 + (k0));                   \
    ra3 = *reinterpret_cast<const uint4 *>(a_ptr3 + (k0));                   \
    rb0 = *reinterpret_cast<const uint4 *>(w_ptr + (k0));                    \
    rb1 = *reinterpret_cast<const uint4 *>(w_ptr + w_step + (k0));           \
    rb2 = *reinterpret_cast<const uint4 *>(w_ptr + 2 * w_step + (k0));       \
    rb3 = *reinterpret_cast<const uint4 *>(w_ptr + 3 * w_step + (k0));
#define SSW_STORE_TILES(buf)                                                                  \
    *reinterpret_cast<uint4 *>(&sA[buf][lds_off]) = ra0;                                      \
    *reinterpret_cast<uint4 *>(&sA[buf][lds_off + 32 * LDS_STRIDE]) = ra1;                    \
    *reinterpret_cast<uint4 *>(&sA[buf][lds_off + 64 * LDS_STRIDE]) = ra2;                    \
    *reinterpret_cast<uint4 *>(&sA[buf][lds_off + 96 * LDS_STRIDE]) = ra3;                    \
    *reinterpret_cast<uint4 *>(&sB[buf][lds_off]) = rb0;                                      \
    *reinterpret_cast<uint4 *>(&sB[buf][lds_off + 32 * LDS_STRIDE]) = rb1;                    \
    *reinterpret_cast<uint4 *>(&sB[buf][lds_off + 64 * LDS_STRIDE]) = rb2;                    \
    *reinterpret_cast<uint4 *>(&sB[buf][lds_off + 96 * LDS_STRIDE]) = rb3;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (SPLITK ? k_len : K) / BK;
    SSW_LOAD_TILES(0)
    SSW_STORE_TILES(0)
    __syncthreads();
    const int fr = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) {  // global loads fly while this tile is multiplied
            const int k0 = (kt + 1) * BK;
            SSW_LOAD_TILES(k0)
        }
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = *reinterpret_cast<const bf16x8 *>(&sA[buf][(wm * 64 + i * 16 + fr) * LDS_STRIDE + ks * 32 + fq * 8]);
                b[i] = *reinterpret_cast<const bf16x8 *>(&sB[buf][(wn * 64 + i * 16 + fr) * LDS_STRIDE + ks * 32 + fq * 8]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            // the other buffer was last read one iteration ago, before the barrier below
            if (buf == 0) {
                SSW_STORE_TILES(1)
            } else {
                SSW_STORE_TILES(0)
            }
            __syncthreads();
        }
    }
#undef SSW_LOAD_TILES
#undef SSW_STORE_TILES

    // epilogue: lane holds C[row = fq*4 + r][col = fr] of each 16x16 tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wn * 64 + j * 16 + fr;
            const float bv = (EPI == EPI_F32) ? 0.f : bias[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 64 + i * 16 + fq * 4 + r;
                if (row >= M) continue;
                float v = acc[i][j][r] + bv;
                const int64_t o = (int64_t)row * N + col;
                if (EPI == EPI_BF16_BIAS_GELU) v = v / (1.f + __expf(-1.702f * v));  // quick_gelu
                if (EPI == EPI_F32_BIAS_RESIDUAL) v += residual[o];
                if (EPI == EPI_BF16_BIAS || EPI == EPI_BF16_BIAS_GELU)
                    reinterpret_cast<bf16 *>(Cout)[o] = to_bf16(v);
                else
                    reinterpret_cast<float *>(Cout)[o] = v;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------
// LDS-DMA ring kernel (DEPTH = 1 would keep one buffer and two barriers per k-step)
// ---------------------------------------------------------------------------------------
constexpr int G_WIMG = 16384;  // W image: 128 rows x 128 B

// glds16 (ssw_common.h): one LDS-DMA wave-instruction, issued from inline asm
#define SSW_GLDS16(gptr, lptr) glds16((gptr), (lptr))

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N == 0 || N == 6 || N == 8 || N == 12 || N == 16, "add the literal");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
}

// TM rows x 128 columns per block, TM / 64 x WN waves of 64 x (128 / WN) each (WN = 2: 64 x 64 per wave;
// WN = 4: 64 x 32 per wave, twice the waves for the same tile)
template <int EPI, int DEPTH, int TM, bool PIPE, int WN = 2>
__global__ __launch_bounds__(TM * WN) void gemm_glds(const bf16 *__restrict__ A, const bf16 *__restrict__ W,
                                                    const float *__restrict__ bias,
                                                    const float *__restrict__ residual, void *__restrict__ Cout,
                                                    int M, int N, int K, int m_tiles, int n_tiles, GemmLn ln) {
    constexpr int NW = (TM / 64) * WN;     // waves
    constexpr int NJ = 128 / WN / 16;      // 16-column accumulator tiles per wave
    constexpr int AP = (TM / 8) / NW;      // A pieces (8 rows x 128 B) per wave
    static_assert(!PIPE || WN == 2, "the pipelined loop is written for 64 x 64 per wave");
    constexpr int AIMG = TM * 128;         // A image bytes per stage
    constexpr int STAGE = AIMG + G_WIMG;   // ring stage
    constexpr int WP = 16 / NW;            // W pieces per wave
    constexpr int LOADS = AP + WP;         // LDS-DMA instructions per wave and stage
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // XCD-aware tile order: ids b and b + 8 share an XCD; XCD x takes row-tiles x, x + 8, ...
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int per_xcd = (m_tiles + 7) >> 3;
    const int mt = ln.xcd_contig ? xcd * per_xcd + idx / n_tiles : (idx / n_tiles) * 8 + xcd, nt = idx % n_tiles;
    if (mt >= m_tiles) return;
    const int m0 = mt * TM, n0 = nt * BN;
    const int wm = wave / WN, wn = wave % WN;

    // staging: a piece is 8 rows x 128 B (one wave-instruction); lane l lands at row l / 8, chunk
    // position l % 8 and therefore fetches chunk (l % 8) ^ swz(row)
#if SSW_GEMM_SADDR
    // saddr form: a wave-uniform tile base in SGPRs + a 32-bit byte offset per lane (a tile spans < 2^32 bytes)
    const bf16 *const a_base = A + (int64_t)m0 * K, *const w_base = W + (int64_t)n0 * K;
    unsigned a_off[AP], w_off[WP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int row = (wave * AP + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        a_off[i] = (unsigned)(((int64_t)(min(m0 + row, M - 1) - m0) * K + chunk * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int row = (wave * WP + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        w_off[i] = (unsigned)(((int64_t)row * K + chunk * 8) * 2);
    }
#else
    const bf16 *a_src[AP];
    const bf16 *w_src[WP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int row = (wave * AP + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        a_src[i] = A + (int64_t)min(m0 + row, M - 1) * K + chunk * 8;
    }
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int row = (wave * WP + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        w_src[i] = W + (int64_t)(n0 + row) * K + chunk * 8;
    }
#endif
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char *)smem);
    const unsigned a_dst = lds0 + wave * (AP * 1024);
    const unsigned w_dst = lds0 + AIMG + wave * (WP * 1024);
#if SSW_GEMM_SADDR
#define SSW_ISSUE(kt, buf)                                                                 \
    {                                                                                      \
        const bf16 *const ab = a_base + (kt) * BK, *const wb = w_base + (kt) * BK;         \
        _Pragma("unroll") for (int i = 0; i < AP; ++i)                                     \
            glds16s(a_off[i], ab, a_dst + (buf) * STAGE + i * 1024);                       \
        _Pragma("unroll") for (int i = 0; i < WP; ++i)                                     \
            glds16s(w_off[i], wb, w_dst + (buf) * STAGE + i * 1024);                       \
    }
#else
#define SSW_ISSUE(kt, buf)                                                                 \
    {                                                                                      \
        const int k0 = (kt) * BK;                                                          \
        _Pragma("unroll") for (int i = 0; i < AP; ++i)                                     \
            SSW_GLDS16(a_src[i] + k0, a_dst + (buf) * STAGE + i * 1024);                   \
        _Pragma("unroll") for (int i = 0; i < WP; ++i)                                     \
            SSW_GLDS16(w_src[i] + k0, w_dst + (buf) * STAGE + i * 1024);                   \
    }
#endif

    const int fr = lane & 15, fq = lane >> 4;
    // The accumulators start from the bias.  (Starting them from bias + residual as well was measured:
    // it moves the exposed residual read from the tail of a single-round launch to its head, -8 %.)
    f32x4 acc[4][NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int col = n0 + wn * (128 / WN) + j * 16 + fq * 4;
        f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (EPI != EPI_F32 && !epi_ln(EPI)) bv = *reinterpret_cast<const f32x4 *>(bias + col);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = bv;
    }
    float *const ln_lds = reinterpret_cast<float *>(smem + DEPTH * STAGE);  // [TM][2] statistics / [WN][TM][2] partials
    // LayerNorm-folded variants: thread t < TM requests the partial sums of row m0 + t NOW (ordinary loads, ahead of the
    // ring's first pieces) and turns them into the row's mean / rstd once the first stages are on their way (LnRowPre)
    LnRowPre ln_pre;
    f32x4 ln_cpre = f32x4{0.f, 0.f, 0.f, 0.f};  // threads 0-31: c1 of 4 of the tile's 128 columns, 32-63: c2
    float *const ln_c = ln_lds + 2 * TM;         // [2][128] behind the statistics
    if constexpr (epi_ln(EPI) && !(SSW_ABL_LN & 1)) {
        if (t < TM) ln_pre = ln_row_request(ln, min(m0 + t, M - 1));
        ln_cpre = *reinterpret_cast<const f32x4 *>(((t & 32) ? bias : ln.c1) + n0 + (t & 31) * 4);
    }

    // fragment byte offset inside an image: row (16-row block + fr), chunk (4 ks + fq) ^ (fr >> 1)
    const int frag0 = fr * 128 + ((fq ^ (fr >> 1)) << 4);
    const int a_frag = wm * 8192 + frag0, w_frag = AIMG + wn * (128 / WN) * 128 + frag0;
#define SSW_COMPUTE(buf)                                                                              \
    {                                                                                                 \
        const unsigned char *sb = smem + (buf) * STAGE;                                               \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                            \
            bf16x8 a[4], b[NJ];                                                                       \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                             \
                a[i] = *reinterpret_cast<const bf16x8 *>(sb + ((a_frag + i * 2048) ^ (ks * 64)));     \
            _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                            \
                b[j] = *reinterpret_cast<const bf16x8 *>(sb + ((w_frag + j * 2048) ^ (ks * 64)));     \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                             \
                _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                        \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0); \
        }                                                                                             \
    }
#define SSW_READ_FRAGS(buf, ks, a, b)                                                                 \
    {                                                                                                 \
        const unsigned char *sb = smem + (buf) * STAGE;                                               \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                               \
            a[i] = *reinterpret_cast<const bf16x8 *>(sb + ((a_frag + i * 2048) ^ ((ks) * 64)));       \
            b[i] = *reinterpret_cast<const bf16x8 *>(sb + ((w_frag + i * 2048) ^ ((ks) * 64)));       \
        }                                                                                             \
    }
#define SSW_MFMA_BLOCK(a, b)                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                     \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                 \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);

    // Have every kernel argument in registers before the loop: a scalar load still pending at loop
    // entry makes the compiler treat the LGKM counter as unordered and turn each counted wait for
    // fragment reads into lgkmcnt(0).
    asm volatile("" ::"s"(bias), "s"(residual), "s"(Cout), "s"(N));
    const int nk = K / BK;
    if constexpr (PIPE) {
        // Software-pipelined form (DEPTH >= 2): the fragments of the next 32-deep half step are
        // read into a second register set while the current half step multiplies, and the
        // barrier sits between the two MFMA blocks, so LDS latency never faces an idle MFMA pipe.
        static_assert(DEPTH >= 2, "pipelined loop needs a ring");
#pragma unroll
        for (int s = 0; s < DEPTH; ++s)
            if (s < nk) SSW_ISSUE(s, s)
        if (DEPTH - 1 < nk)
            wait_vmcnt<LOADS * (DEPTH - 1)>();
        else
            wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        bf16x8 a0[4], b0[4], a1[4], b1[4];
        SSW_READ_FRAGS(0, 0, a0, b0)
        int buf = 0;
        for (int kt = 0; kt + 1 < nk; ++kt) {
            const int nbuf = (buf + 1 == DEPTH) ? 0 : buf + 1;
            SSW_READ_FRAGS(buf, 1, a1, b1)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            SSW_MFMA_BLOCK(a0, b0)
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            // stage kt+1 must have landed; stages kt+2 .. kt+DEPTH-1 may stay in flight
            if (kt + DEPTH - 1 < nk)
                wait_vmcnt<LOADS * (DEPTH - 2)>();
            else
                wait_vmcnt<0>();
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): my reads of slot buf are done
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + DEPTH < nk) SSW_ISSUE(kt + DEPTH, buf)
            SSW_READ_FRAGS(nbuf, 0, a0, b0)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            SSW_MFMA_BLOCK(a1, b1)
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            buf = nbuf;
        }
        SSW_READ_FRAGS(buf, 1, a1, b1)
        SSW_MFMA_BLOCK(a0, b0)
        SSW_MFMA_BLOCK(a1, b1)
    } else if constexpr (DEPTH == 1) {
        for (int kt = 0; kt < nk; ++kt) {
            SSW_ISSUE(kt, 0)
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            SSW_COMPUTE(0)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    } else if (SSW_GEMM_STAGGER && DEPTH == 2 && NW == 8 && wave >= NW / 2) {
        // Stagger (MICROARCH guide, "Two waves per SIMD", item 9): waves 4-7 run half a K-step behind waves 0-3 -- the second
        // 32-deep half of stage kt - 1 is multiplied from registers (its fragments were read before the barrier) while waves
        // 0-3 read their first fragments of stage kt, and so on in anti-phase: one half's LDS reads beside the other half's
        // MFMAs instead of both reading, then both queueing on the matrix pipe.  Every accumulator still adds its halves in
        // the order (kt, 0), (kt, 1), (kt + 1, 0) ...: same bits.
        SSW_ISSUE(0, 0)
        int buf = 0, nxt = 1;
        bf16x8 a1[4], b1[NJ];
        for (int kt = 0; kt < nk; ++kt) {
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + 1 < nk) SSW_ISSUE(kt + 1, nxt)
            if (kt > 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j], a1[i], acc[i][j], 0, 0, 0);
            }
            const unsigned char *sb = smem + buf * STAGE;
            {
                bf16x8 a0[4], b0[NJ];
#pragma unroll
                for (int i = 0; i < 4; ++i) a0[i] = *reinterpret_cast<const bf16x8 *>(sb + (a_frag + i * 2048));
#pragma unroll
                for (int j = 0; j < NJ; ++j) b0[j] = *reinterpret_cast<const bf16x8 *>(sb + (w_frag + j * 2048));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[j], a0[i], acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) a1[i] = *reinterpret_cast<const bf16x8 *>(sb + ((a_frag + i * 2048) ^ 64));
#pragma unroll
            for (int j = 0; j < NJ; ++j) b1[j] = *reinterpret_cast<const bf16x8 *>(sb + ((w_frag + j * 2048) ^ 64));
            // the fragments of this stage's second half are in registers before the next barrier lets the stage be restaged
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
            asm volatile("" ::: "memory");
            buf ^= 1;
            nxt ^= 1;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j], a1[i], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
        for (int s = 0; s < DEPTH - 1; ++s)
            if (s < nk) SSW_ISSUE(s, s)
        if constexpr (epi_ln(EPI) && !(SSW_ABL_LN & 1)) {  // published by the loop's barriers; read behind the loop
            if (t < TM) ln_row_finish(ln, ln_pre, ln_lds + 2 * t);
            if (t < 64) *reinterpret_cast<f32x4 *>(ln_c + t * 4) = ln_cpre;
        }
        int buf = 0, nxt = DEPTH - 1;  // ring slots of stage kt and of stage kt + DEPTH - 1
        for (int kt = 0; kt < nk; ++kt) {
            // stage kt has landed once at most the DEPTH-2 younger stages are still in flight
            if (kt + DEPTH - 2 < nk)
                wait_vmcnt<LOADS * (DEPTH - 2)>();
            else
                wait_vmcnt<0>();
            // behind this barrier every wave's part of stage kt is visible and nobody still
            // reads slot nxt (it held stage kt - 1)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + DEPTH - 1 < nk) SSW_ISSUE(kt + DEPTH - 1, nxt)
            SSW_COMPUTE(buf)
            asm volatile("" ::: "memory");
            buf = (buf + 1 == DEPTH) ? 0 : buf + 1;
            nxt = (nxt + 1 == DEPTH) ? 0 : nxt + 1;
        }
    }
#undef SSW_ISSUE
#undef SSW_COMPUTE
#undef SSW_READ_FRAGS
#undef SSW_MFMA_BLOCK

    // epilogue: acc[i][j][r] = C[m0 + wm*64 + i*16 + fr][n0 + wn*(128/WN) + j*16 + fq*4 + r]
    static_assert(!(PIPE && epi_ln(EPI)), "the LayerNorm-folded epilogues run on the two-stage loop");
    if constexpr (epi_bf16_out(EPI) && !PIPE) {
        // another wave may still be reading its last fragments out of the bytes this wave is about to overwrite
        __syncthreads();
        store_rows_via_lds<EPI, 4, NJ>(acc, smem + wave * (64 * (NJ * 32 + 16)), lane, m0 + wm * 64, n0 + wn * (128 / WN), M, N,
                                       reinterpret_cast<bf16 *>(Cout), ln_lds + 2 * (wm * 64), ln_c + wn * (128 / WN),
                                       ln_c + 128 + wn * (128 / WN));
        return;
    }
    if constexpr (EPI == EPI_F32_BIAS_RESIDUAL_STATS && !PIPE && SSW_EPI6_FULL_LINES) {
        // f32 rows (round 5): the wave's 64 x CW sub-tile goes through LDS 32 rows at a time and comes back row-major with
        // CW / 4 lanes a row, FOUR consecutive columns a lane -- one store instruction writes whole 128-byte (CW = 32) row
        // segments of the f32 row, one load reads the residual row the same way, the bf16 copy leaves as 64-byte segments.
        // (Round 4's form gave a lane 8 columns as two 16-byte halves: every line was touched by two instructions, half of
        // it each -- the request count again, not the bytes.)  Partial sums: 4 per lane, the lanes of the row (butterfly),
        // then the WN waves in wave order.
        constexpr int CW = NJ * 16;
        constexpr int RBF = CW * 4 + 16;
        constexpr int LPR = CW / 4, RPI = 64 / LPR, NP = 64 / RPI;
        unsigned char *wl = smem + wave * (32 * RBF);
        const int rr = lane / LPR, cc = lane % LPR;
        const int col0 = n0 + wn * CW + cc * 4;
        const int64_t res_ld = ln.res_ld ? ln.res_ld : (int64_t)N;
        f32x4 res[NP];
#pragma unroll
        for (int it = 0; it < NP; ++it)
            res[it] = epi_load(reinterpret_cast<const f32x4 *>(residual + (int64_t)min(m0 + wm * 64 + it * RPI + rr, M - 1) * res_ld + col0));
        __syncthreads();  // another wave may still be reading its last fragments out of these bytes
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int il = 0; il < 2; ++il)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    *reinterpret_cast<f32x4 *>(wl + (il * 16 + fr) * RBF + j * 64 + fq * 16) = acc[2 * h + il][j];
#pragma unroll
            for (int ps = 0; ps < NP / 2; ++ps) {
                const int it = h * (NP / 2) + ps;
                const int lrow = wm * 64 + it * RPI + rr;
                f32x4 v = *reinterpret_cast<const f32x4 *>(wl + (ps * RPI + rr) * RBF + cc * 16);
                v += res[it];
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = to_bf16(v[r]);
                float ssum = (v[0] + v[1]) + (v[2] + v[3]);
                float ssq = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                if (m0 + lrow < M) {
                    const int64_t off = (int64_t)(m0 + lrow) * N + col0;
                    epi_store<1>(reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(Cout) + off), v);
                    epi_store<1>(reinterpret_cast<bf16x4 *>(ln.xcopy + off), o);
                }
#pragma unroll
                for (int sh = 1; sh < LPR; sh <<= 1) {
                    ssum += __shfl_xor(ssum, sh, 64);
                    ssq += __shfl_xor(ssq, sh, 64);
                }
                if (cc == 0) {
                    ln_lds[(wn * TM + lrow) * 2] = ssum;
                    ln_lds[(wn * TM + lrow) * 2 + 1] = ssq;
                }
            }
        }
    } else if constexpr (epi_stats(EPI) && !PIPE) {
        // The residual stream, the same way: the wave's 64 x CW sub-tile (CW = 32 or 64 columns) goes through LDS as
        // f32, 32 rows at a time (rows padded by 16 bytes: the 16 rows of a write on different banks), and comes back
        // row-major, CW / 8 lanes a row with 8 consecutive columns each.  bf16 stream (7): the row is read and written
        // back in 16-byte pieces instead of 8-byte ones, the sum is taken in f32 as before, the statistics are those of
        // the rounded values.  f32 stream (6): residual and output rows in 32-byte pieces (128- / 256-byte segments),
        // the bf16 copy in 16-byte ones, statistics of the f32 values.  Partial sums: 8 per lane, the lanes of the row
        // (butterfly), then the WN waves in wave order.
        constexpr bool BF = EPI == EPI_BF16_STREAM_STATS;
        constexpr int CW = NJ * 16;       // columns of this wave
        constexpr int RBF = CW * 4 + 16;  // bytes of a parked row
        constexpr int LPR = CW / 8;       // lanes per row on the way back
        constexpr int RPI = 64 / LPR;     // rows per pass
        constexpr int NP = 64 / RPI;      // passes over the wave's 64 rows
        unsigned char *wl = smem + wave * (32 * RBF);
        const int rr = lane / LPR, cc = lane % LPR;
        const int col0 = n0 + wn * CW + cc * 8;
        bf16x8 resb[BF ? NP : 1];
        f32x4 resf[BF ? 1 : NP][2];
        const int64_t res_ld = ln.res_ld ? ln.res_ld : (int64_t)N;
#pragma unroll
        for (int it = 0; it < NP; ++it) {  // every row group of the stream rows, requested before the LDS round trip
            const int64_t rrow = min(m0 + wm * 64 + it * RPI + rr, M - 1);
            if constexpr (BF) {
                resb[it] = epi_load(reinterpret_cast<const bf16x8 *>(ln.xcopy + rrow * N + col0));
            } else {
                resf[it][0] = epi_load(reinterpret_cast<const f32x4 *>(residual + rrow * res_ld + col0));
                resf[it][1] = epi_load(reinterpret_cast<const f32x4 *>(residual + rrow * res_ld + col0 + 4));
            }
        }
        __syncthreads();  // another wave may still be reading its last fragments out of these bytes
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int il = 0; il < 2; ++il)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    *reinterpret_cast<f32x4 *>(wl + (il * 16 + fr) * RBF + j * 64 + fq * 16) = acc[2 * h + il][j];
#pragma unroll
            for (int ps = 0; ps < NP / 2; ++ps) {
                const int it = h * (NP / 2) + ps;
                const int lrow = wm * 64 + it * RPI + rr;
                f32x4 lo = *reinterpret_cast<const f32x4 *>(wl + (ps * RPI + rr) * RBF + cc * 32);
                f32x4 hi = *reinterpret_cast<const f32x4 *>(wl + (ps * RPI + rr) * RBF + cc * 32 + 16);
                bf16x8 o;
                float ssum = 0.f, ssq = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (BF) {
                        lo[r] += (float)resb[it][r];
                        hi[r] += (float)resb[it][4 + r];
                    } else {
                        lo[r] += resf[it][0][r];
                        hi[r] += resf[it][1][r];
                    }
                    o[r] = to_bf16(lo[r]);
                    o[4 + r] = to_bf16(hi[r]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = BF ? (float)o[r] : lo[r], b = BF ? (float)o[4 + r] : hi[r];
                    ssum += a;
                    ssq += a * a;
                    ssum += b;
                    ssq += b * b;
                }
                if (m0 + lrow < M) {
                    const int64_t off = (int64_t)(m0 + lrow) * N + col0;
                    epi_store<1>(reinterpret_cast<bf16x8 *>(ln.xcopy + off), o);
                    if constexpr (!BF) {
                        epi_store<1>(reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(Cout) + off), lo);
                        epi_store<1>(reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(Cout) + off + 4), hi);
                    }
                }
#pragma unroll
                for (int sh = 1; sh < LPR; sh <<= 1) {
                    ssum += __shfl_xor(ssum, sh, 64);
                    ssq += __shfl_xor(ssq, sh, 64);
                }
                if (cc == 0) {
                    ln_lds[(wn * TM + lrow) * 2] = ssum;
                    ln_lds[(wn * TM + lrow) * 2 + 1] = ssq;
                }
            }
        }
    } else if constexpr ((EPI == EPI_F32 || EPI == EPI_F32_BIAS_RESIDUAL) && !PIPE) {
        // f32 outputs, the same way (a lane's 4 columns are 16 bytes here, but still 16 rows an instruction): 32 rows of
        // the wave's 64 x CW sub-tile at a time through LDS, back row-major, CW / 4 lanes a row -- 128- / 256-byte
        // segments for the store and for the residual row it adds
        constexpr int CW = NJ * 16;
        constexpr int RBF = CW * 4 + 16;
        constexpr int LPR = CW / 4, RPI = 64 / LPR, NP = 64 / RPI;
        unsigned char *wl = smem + wave * (32 * RBF);
        const int rr = lane / LPR, cc = lane % LPR;
        const int col0 = n0 + wn * CW + cc * 4;
        f32x4 res[EPI == EPI_F32_BIAS_RESIDUAL ? NP : 1];
        if constexpr (EPI == EPI_F32_BIAS_RESIDUAL) {
#pragma unroll
            for (int it = 0; it < NP; ++it)
                res[it] = *reinterpret_cast<const f32x4 *>(residual + (int64_t)min(m0 + wm * 64 + it * RPI + rr, M - 1) * N + col0);
        }
        __syncthreads();  // another wave may still be reading its last fragments out of these bytes
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int il = 0; il < 2; ++il)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    *reinterpret_cast<f32x4 *>(wl + (il * 16 + fr) * RBF + j * 64 + fq * 16) = acc[2 * h + il][j];
#pragma unroll
            for (int it = 0; it < NP / 2; ++it) {
                f32x4 v = *reinterpret_cast<const f32x4 *>(wl + (it * RPI + rr) * RBF + cc * 16);
                if constexpr (EPI == EPI_F32_BIAS_RESIDUAL) v += res[h * (NP / 2) + it];
                const int row = m0 + wm * 64 + h * 32 + it * RPI + rr;
                if (row < M) *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(Cout) + (int64_t)row * N + col0) = v;
            }
        }
    } else
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int lrow = wm * 64 + i * 16 + fr;
        const int row = m0 + lrow;
        float mean = 0.f, rstd = 1.f, ssum = 0.f, ssq = 0.f;
        if constexpr (epi_ln(EPI)) {
            mean = ln_lds[2 * lrow];
            rstd = ln_lds[2 * lrow + 1];
        }
        if (row < M) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int col = n0 + wn * (128 / WN) + j * 16 + fq * 4;
                epilogue_store<EPI>(acc[i][j], (int64_t)row * N + col, col, bias, residual, Cout, ln, mean, rstd, &ssum, &ssq,
                                    ln_c + (col - n0), ln_c + 128 + (col - n0));
            }
        }
        if constexpr (epi_stats(EPI)) {  // this wave's 128 / WN columns of the row: the four fq lanes
            ssum += __shfl_xor(ssum, 16, 64);
            ssq += __shfl_xor(ssq, 16, 64);
            ssum += __shfl_xor(ssum, 32, 64);
            ssq += __shfl_xor(ssq, 32, 64);
            if (fq == 0) {
                ln_lds[(wn * TM + lrow) * 2] = ssum;
                ln_lds[(wn * TM + lrow) * 2 + 1] = ssq;
            }
        }
    }
    if constexpr (epi_stats(EPI)) {
        __syncthreads();
        if (t < TM && m0 + t < M) {  // the tile's 128 columns of row t: the WN wave partials in wave order
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int w = 0; w < WN; ++w) {
                sm += ln_lds[(w * TM + t) * 2];
                sq += ln_lds[(w * TM + t) * 2 + 1];
            }
            float *o = ln.stats_out + ((int64_t)(m0 + t) * n_tiles + nt) * 2;
            o[0] = sm;
            o[1] = sq;
        }
    }
}

// ---------------------------------------------------------------------------------------
// 256 x 256 x 64 tile, 8 waves of 128 x 64 (2 x 4), phase-interleaved K loop  (variant 9)
// ---------------------------------------------------------------------------------------
// Why: in the 128 x 128 kernel above a wave owns 64 x 32 outputs and reads 12 fragments (ds_read_b128, 1 KiB each)
// per 16 MFMAs; at 16 waves per CU that is 1536 LDS clocks per K-step against 1024 MFMA clocks per SIMD -- the loop
// is bound by LDS read bandwidth (128 B/clk/CU), not by the matrix cores.  A wave that owns 128 x 64 reads 24
// fragments per 64 MFMAs (0.375 per MFMA): 1536 LDS clocks against 2048 MFMA clocks.
//
// One workgroup per CU (128 KiB of LDS: two stages of {A rows 0-127, A rows 128-255, W rows 0-127, W rows 128-255},
// each a 128-row x 128-B image in the swizzled layout of the kernel above).  Wave (wr, wc) reads only A half wr and
// W half wc >> 1.  A K-tile is four phases of 16 MFMAs (quadrants of the wave's 128 x 64: (lo,0) (lo,1) (hi,1)
// (hi,0)); the W fragments of both column halves stay in registers, so a stage's W halves are free after phase 2
// and its A halves after phase 3, and the next-but-one tile is restaged half by half while the current one
// multiplies:
//     phase 1: read A_lo, W_0          issue A1(kt+1)                                   MFMA (lo,0)
//     phase 2: read W_1                                           lgkm(0), barrier      MFMA (lo,1)
//     phase 3: read A_hi               issue W0(kt+2)             lgkm(0), barrier      MFMA (hi,1)
//     phase 4:                         issue W1(kt+2), A0(kt+2)   vmcnt(6), barrier     MFMA (hi,0)
// vmcnt(6) at phase 4 retires everything up to A1(kt+1) -- the three half-tiles issued after it stay in flight --
// and the barrier behind it makes tile kt+1 readable in the next phase 1 (read one phase AFTER the wait that
// retires the data).  WAR: a half is restaged only after a barrier that follows its last fragment reads
// (W: phases 1-2 -> barrier of phase 2; A: phases 1 and 3 -> barrier of phase 3).
constexpr int T256_HALF = 16384;            // one half-tile image: 128 rows x 128 B
constexpr int T256_STAGE = 4 * T256_HALF;   // A0 A1 W0 W1

template <int EPI>
__global__ __launch_bounds__(512) void gemm_256(const bf16 *__restrict__ A, const bf16 *__restrict__ W,
                                                const float *__restrict__ bias, const float *__restrict__ residual,
                                                void *__restrict__ Cout, int M, int N, int K, int m_tiles, int n_tiles,
                                                GemmLn ln) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    // workgroup ids b and b + 8 run on the same XCD: XCD x takes a contiguous eighth of the tiles in row-major order
    // (its workgroups share A row tiles through its L2), every XCD the same number to within one -- with whole row
    // tiles dealt round-robin (the 128 x 128 kernel's order) 28 row tiles x 9 put 36 tiles on the 32 CUs of four XCDs
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int n_all = m_tiles * n_tiles, per_xcd = (n_all + 7) >> 3;
    const int tile = xcd * per_xcd + idx;
    if (tile >= n_all) return;
    const int mt = tile / n_tiles, nt = tile % n_tiles;
    const int m0 = mt * 256, n0 = nt * 256;

    // staging: half-tile = 16 pieces of 8 rows x 128 B; wave w issues pieces 2w and 2w+1 of every half
    const bf16 *a_src[2], *w_src[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = (wave * 2 + p) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        a_src[p] = A + (int64_t)min(m0 + row, M - 1) * K + chunk * 8;
        w_src[p] = W + (int64_t)(n0 + row) * K + chunk * 8;
    }
    // the second half's rows are 128 further down; clamp the A rows of the last row tile
    const bf16 *a_src1[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = 128 + (wave * 2 + p) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        a_src1[p] = A + (int64_t)min(m0 + row, M - 1) * K + chunk * 8;
    }
    const int64_t w_half = (int64_t)128 * K;
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char *)smem);
    const unsigned piece_dst = lds0 + wave * 2048;
#define T256_ISSUE_A0(kt, st) { const int k0 = (kt) * BK; glds16(a_src[0] + k0, piece_dst + (st) * T256_STAGE); \
                                 glds16(a_src[1] + k0, piece_dst + (st) * T256_STAGE + 1024); }
#define T256_ISSUE_A1(kt, st) { const int k0 = (kt) * BK; glds16(a_src1[0] + k0, piece_dst + (st) * T256_STAGE + T256_HALF); \
                                 glds16(a_src1[1] + k0, piece_dst + (st) * T256_STAGE + T256_HALF + 1024); }
#define T256_ISSUE_W0(kt, st) { const int k0 = (kt) * BK; glds16(w_src[0] + k0, piece_dst + (st) * T256_STAGE + 2 * T256_HALF); \
                                 glds16(w_src[1] + k0, piece_dst + (st) * T256_STAGE + 2 * T256_HALF + 1024); }
#define T256_ISSUE_W1(kt, st) { const int k0 = (kt) * BK; glds16(w_src[0] + w_half + k0, piece_dst + (st) * T256_STAGE + 3 * T256_HALF); \
                                 glds16(w_src[1] + w_half + k0, piece_dst + (st) * T256_STAGE + 3 * T256_HALF + 1024); }

    const int fr = lane & 15, fq = lane >> 4;
    f32x4 acc[8][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wc * 64 + j * 16 + fq * 4;
        f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (EPI != EPI_F32 && !epi_ln(EPI)) bv = *reinterpret_cast<const f32x4 *>(bias + col);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i][j] = bv;
    }
    float *const ln_lds = reinterpret_cast<float *>(smem + 2 * T256_STAGE);  // [256][2] row statistics
    LnRowPre ln_pre;  // as in gemm_glds: row m0 + t's partial pairs, requested now, summed once the first tiles are on their way
    f32x4 ln_cpre = f32x4{0.f, 0.f, 0.f, 0.f};  // threads 0-63: c1 of 4 of the tile's 256 columns, 64-127: c2
    float *const ln_c = ln_lds + 2 * 256;        // [2][256] behind the statistics
    if constexpr (epi_ln(EPI)) {
        ln_cpre = *reinterpret_cast<const f32x4 *>(((t & 64) ? bias : ln.c1) + n0 + (t & 63) * 4);
        if (t < 256) ln_pre = ln_row_request(ln, min(m0 + t, M - 1));
    }
    const int frag0 = fr * 128 + ((fq ^ (fr >> 1)) << 4);
    const int a_frag = wr * T256_HALF + frag0;                                        // + i * 2048, i = 0..7
    const int w_frag = (2 + (wc >> 1)) * T256_HALF + (wc & 1) * 4 * 2048 + frag0;     // + j * 2048, j = 0..3
#define T256_READ_A(st, half, dst)                                                                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                        \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                 \
            dst[i][ks] = *reinterpret_cast<const bf16x8 *>(smem + (st) * T256_STAGE + ((a_frag + ((half) * 4 + i) * 2048) ^ (ks * 64)));
#define T256_READ_W(st, half, dst)                                                                       \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                        \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                 \
            dst[j][ks] = *reinterpret_cast<const bf16x8 *>(smem + (st) * T256_STAGE + ((w_frag + ((half) * 2 + j) * 2048) ^ (ks * 64)));
#define T256_MFMA(ahalf, a, whalf, b)                                                                    \
    __builtin_amdgcn_s_setprio(1);                                                                       \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                     \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                    \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                \
                acc[(ahalf) * 4 + i][(whalf) * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(         \
                    b[j][ks], a[i][ks], acc[(ahalf) * 4 + i][(whalf) * 2 + j], 0, 0, 0);                  \
    __builtin_amdgcn_s_setprio(0);
#define T256_LGKM0_BARRIER()                                                                             \
    __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) */                                                 \
    __builtin_amdgcn_s_barrier();                                                                        \
    asm volatile("" ::: "memory");

    asm volatile("" ::"s"(bias), "s"(residual), "s"(Cout), "s"(N));
    const int nk = K / BK;
    // prologue: tile 0 complete, then W0 W1 A0 of tile 1 (what phases 3-4 of a tile "-1" would have issued)
    T256_ISSUE_A0(0, 0) T256_ISSUE_A1(0, 0) T256_ISSUE_W0(0, 0) T256_ISSUE_W1(0, 0)
    if (nk > 1) {
        T256_ISSUE_W0(1, 1) T256_ISSUE_W1(1, 1) T256_ISSUE_A0(1, 1)
    }
    if constexpr (epi_ln(EPI)) {  // published by the barrier below; read behind the loop
        if (t < 256) ln_row_finish(ln, ln_pre, ln_lds + 2 * t);
        if (t < 128) *reinterpret_cast<f32x4 *>(ln_c + t * 4) = ln_cpre;
    }
    if (nk > 1) {
        wait_vmcnt<6>();
    } else {
        wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    bf16x8 a[4][2], b0[2][2], b1[2][2];
    for (int kt = 0; kt < nk; ++kt) {
        const int st = kt & 1, ot = st ^ 1;
        // phase 1
        T256_READ_A(st, 0, a)
        T256_READ_W(st, 0, b0)
        if (kt + 1 < nk) T256_ISSUE_A1(kt + 1, ot)
        __builtin_amdgcn_sched_barrier(0);
        T256_MFMA(0, a, 0, b0)
        __builtin_amdgcn_sched_barrier(0);
        // phase 2
        T256_READ_W(st, 1, b1)
        T256_LGKM0_BARRIER()      // every wave's W reads of this stage are done
        T256_MFMA(0, a, 1, b1)
        __builtin_amdgcn_sched_barrier(0);
        // phase 3
        T256_READ_A(st, 1, a)
        if (kt + 2 < nk) T256_ISSUE_W0(kt + 2, st)
        T256_LGKM0_BARRIER()      // every wave's A reads of this stage are done
        T256_MFMA(1, a, 1, b1)
        __builtin_amdgcn_sched_barrier(0);
        // phase 4
        if (kt + 2 < nk) {
            T256_ISSUE_W1(kt + 2, st) T256_ISSUE_A0(kt + 2, st)
            wait_vmcnt<6>();      // tile kt+1 has landed (its A1 was the oldest load still wanted)
        } else {
            wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        T256_MFMA(1, a, 0, b0)
        __builtin_amdgcn_sched_barrier(0);
    }
#undef T256_ISSUE_A0
#undef T256_ISSUE_A1
#undef T256_ISSUE_W0
#undef T256_ISSUE_W1
#undef T256_READ_A
#undef T256_READ_W
#undef T256_MFMA
#undef T256_LGKM0_BARRIER

    // epilogue: acc[i][j][r] = C[m0 + wr*128 + i*16 + fr][n0 + wc*64 + j*16 + fq*4 + r]
    static_assert(!epi_stats(EPI), "the statistics epilogue lives in gemm_glds (N = hidden width shapes)");
    if constexpr (epi_bf16_out(EPI)) {
        __syncthreads();  // as in gemm_glds: the staging bytes are about to be reused
#pragma unroll
        for (int h = 0; h < 2; ++h)  // 64 rows at a time: 9 KB of LDS per wave
            store_rows_via_lds<EPI, 4, 4>(acc + 4 * h, smem + wave * (64 * 144), lane, m0 + wr * 128 + h * 64, n0 + wc * 64, M, N,
                                          reinterpret_cast<bf16 *>(Cout), ln_lds + 2 * (wr * 128 + h * 64), ln_c + wc * 64,
                                          ln_c + 256 + wc * 64);
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int lrow = wr * 128 + i * 16 + fr;
        const int row = m0 + lrow;
        float mean = 0.f, rstd = 1.f, ssum = 0.f, ssq = 0.f;
        if constexpr (epi_ln(EPI)) {
            mean = ln_lds[2 * lrow];
            rstd = ln_lds[2 * lrow + 1];
        }
        if (row >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wc * 64 + j * 16 + fq * 4;
            epilogue_store<EPI>(acc[i][j], (int64_t)row * N + col, col, bias, residual, Cout, ln, mean, rstd, &ssum, &ssq,
                                ln_c + (col - n0), ln_c + 256 + (col - n0));
        }
    }
}

template <int EPI>
ssw_status launch_256(hipStream_t s, const bf16 *A, const bf16 *W, const float *bias, const float *res, void *C, int M,
                      int N, int K, const GemmLn &ln) {
    static bool attr_set[64] = {false};  // the attribute is per device
    constexpr int lds = 2 * T256_STAGE + 2048 + 2048;  // + the rows' LayerNorm statistics and the columns' c1 / c2 (EPI 4 / 5)
    int dev_id = 0;
    SSW_HIP_TRY(hipGetDevice(&dev_id));
    if (dev_id < 0 || dev_id >= 64 || !attr_set[dev_id]) {
        SSW_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_256<EPI>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        if (dev_id >= 0 && dev_id < 64) attr_set[dev_id] = true;
    }
    const int m_tiles = (M + 255) / 256, n_tiles = N / 256;
    const int grid = ((m_tiles * n_tiles + 7) / 8) * 8;
    hipLaunchKernelGGL((gemm_256<EPI>), dim3(grid), dim3(512), lds, s, A, W, bias, res, C, M, N, K, m_tiles, n_tiles, ln);
    SSW_HIP_TRY(hipGetLastError());
    return SSW_OK;
}

// (Round 2 also measured an anti-phase form of this kernel -- waves 0-3 multiply a whole K-tile from registers while
//  waves 4-7 read their fragments of the next one and issue the LDS-DMA, roles swapping at every barrier, halves issued
//  two intervals ahead of their first read: same results bit for bit, 511 TFLOP/s on fc1 against 800 for gemm_256 and
//  320 against 930 (128 x 128 kernel) on fc2 -- an interval took ~3 700 cycles against 1 024 of MFMA work, i.e. every
//  interval ended waiting for the DMA issued in the interval before.  Removed again)

// (And a 256 x 256 x 32 tile with a four-stage ring -- operands requested three K-steps ahead, 96 KB in flight per CU,
//  one barrier per step, 188 VGPRs: same results, 693 TFLOP/s on fc1 and 507 on fc2.  So the waits of the two-stage
//  kernels are not a shortage of operands in flight either.  Removed again.)

// (Staggering the two workgroups of a CU -- the second one sleeping 3, 5 or 8 us before its first load, so that their
//  +residual epilogues do not hit HBM together -- was measured on the three EPI 3 shapes: every variant adds about half
//  its delay to the launch (out-proj 21.5 -> 22.9 / 24.1 / 25.8 us).  A workgroup alone on its CU does not run faster
//  than one that shares it: the K loop is bound by its own latency chain, not by the shared matrix pipes.)

// (A sixth structure, same kernel template: 256 x 128 tiles, 16 waves of 64 x 32, three-stage ring, one workgroup per
//  CU -- a quarter less L2->LDS traffic per flop and operands requested two K-steps ahead.  fc2 886 vs 880 TFLOP/s,
//  out-proj 522 vs 523, patch 940 vs 926; qkv 651 vs 722, fc1 650 vs 700.  Not kept.)

SSW_TUNABLE int g_gemm_variant = 14;  // ssw_tune_gemm (lab build only)

template <int EPI, int DEPTH, int TM, bool PIPE = false, int WN = 2>
ssw_status launch_glds(hipStream_t s, const bf16 *A, const bf16 *W, const float *bias, const float *res, void *C,
                       int M, int N, int K, const GemmLn &ln) {
    static bool attr_set[64] = {false};  // the attribute is per device
    constexpr int lds = DEPTH * (TM * 128 + G_WIMG) + WN * TM * 8;  // + row statistics / their per-wave partials, c1 / c2
    int dev_id = 0;
    SSW_HIP_TRY(hipGetDevice(&dev_id));
    if (dev_id < 0 || dev_id >= 64 || !attr_set[dev_id]) {
        SSW_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_glds<EPI, DEPTH, TM, PIPE, WN>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        if (dev_id >= 0 && dev_id < 64) attr_set[dev_id] = true;
    }
    const int m_tiles = (M + TM - 1) / TM, n_tiles = N / BN;
    const int grid = ((m_tiles + 7) / 8) * 8 * n_tiles;
    hipLaunchKernelGGL((gemm_glds<EPI, DEPTH, TM, PIPE, WN>), dim3(grid), dim3(TM * WN), lds, s, A, W, bias, res, C, M, N, K,
                       m_tiles, n_tiles, ln);
    SSW_HIP_TRY(hipGetLastError());
    return SSW_OK;
}

// The default choice between the two tile kernels (variant 14 of ssw_tune_gemm; 15 = the 128 x 128 kernel only).
// The 256 x 256 tile (one workgroup per CU, 128 x 64 per wave) pays only when its grid fills whole rounds of the chip:
// at M = 10 000 it wins on fc1 (480 tiles, 1.9 rounds of 256 CUs) and loses on QKV (360 tiles: one round and 0.4 of a
// second) and on every N = 768 shape (120 tiles).
// (Round 3 also measured a split by rows for the grids in between -- the 7168 rows of QKV that make exactly one round
// on the 256 x 256 kernel, the other 2832 on the 128 x 128 kernel behind it: 48.7 us against 47.1 for the small tiles
// alone, text QKV 38.4 against 36.2, the B = 200 forward unchanged at 2.47 ms.  Removed.)
template <int EPI>
ssw_status launch_auto(hipStream_t s, const bf16 *A, const bf16 *W, const float *bias, const float *res, void *C, int M,
                       int N, int K, const GemmLn &ln) {
    // variants 16 / 17 (lab build): 256 x 128 tiles, 8 waves of 64 x 64 (16 fragment reads per 32 MFMAs where the 64 x 32
    // waves of the 128-square kernel read 12 per 16), one workgroup per CU, three- / two-stage ring
    if (ln.xcd_contig) return launch_glds<EPI, 2, 128, false, 4>(s, A, W, bias, res, C, M, N, K, ln);  // (the consumer counts on 128-row tiles)
    if (g_gemm_variant == 16) return launch_glds<EPI, 3, 256, false, 2>(s, A, W, bias, res, C, M, N, K, ln);
    if (g_gemm_variant == 17) return launch_glds<EPI, 2, 256, false, 2>(s, A, W, bias, res, C, M, N, K, ln);
    if constexpr (epi_bf16_out(EPI)) {  // (f32 outputs leave through LDS in the 128 x 128 kernel only)
        if (g_gemm_variant != 15 && N % 256 == 0) {
            int dev = 0;
            (void)hipGetDevice(&dev);
            // its tiles run in rounds of one per CU: it pays when the last round is nearly full.  Measured with the
            // row-segment epilogues, QKV / fc1 (N = 2304 / 3072, K = 768), 128-square vs 256-square kernel, us:
            //   M  2500: 15.0 / 21.8, 24.4 / 25.4      5000: 23.7 / 24.4, 30.5 / 28.6 (0.94 of a round)
            //      7500: 34.3 / 42.2, 44.2 / 50.3     10000: 39.2 / 43.0 (1.41 rounds), 56.1 / 51.5 (1.88)
            //     15000: 57.3 / 63.3 (2.07), 81.9 / 75.8 (2.77)    20000: 74.6 / 68.6 (2.78), 107.7 / 99.5
            const int64_t tiles256 = (int64_t)((M + 255) / 256) * (N / 256);
            const int64_t cus = num_cus(dev), rounds = (tiles256 + cus - 1) / cus;
            if (100 * tiles256 >= 85 * rounds * cus) return launch_256<EPI>(s, A, W, bias, res, C, M, N, K, ln);
        }
    }
    // Few tiles (round 5): a launch of at most one workgroup per CU lasts as long as ONE workgroup's K loop, and with two
    // stages that loop is a chain of memory round trips -- a K-step waits for the pieces requested one step earlier
    // (~1 us a step whatever the tile: the text tower's 40-tile fc2 took 31 us for 32 steps).  Such launches get a
    // four-stage ring (128 KB of the CU's LDS: nobody else wants it), three stages in flight; same sums in the same
    // order, so a row's result does not depend on which ring its launch took.
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const int64_t tiles128 = (int64_t)((M + 127) / 128) * (N / 128);
        static const bool no_deep = getenv("SSW_GEMM_NO_DEEP_RING") != nullptr;  // A/B
        if (!no_deep && g_gemm_variant == 14 && tiles128 <= num_cus(dev) && K >= 4 * BK)
            return launch_glds<EPI, 4, 128, false, 4>(s, A, W, bias, res, C, M, N, K, ln);
    }
    return launch_glds<EPI, 2, 128, false, 4>(s, A, W, bias, res, C, M, N, K, ln);
}

template <int EPI>
ssw_status launch_epi(hipStream_t s, const bf16 *A, const bf16 *W, const float *bias, const float *res, void *C, int M,
                      int N, int K) {
    const GemmLn none;
    switch (g_gemm_variant) {
        case 0:
            hipLaunchKernelGGL(gemm_bf16_nt<EPI>, dim3(N / BN, (M + BM - 1) / BM), dim3(256), 0, s, A, W, bias, res, C,
                               M, N, K);
            SSW_HIP_TRY(hipGetLastError());
            return SSW_OK;
        case 7: return launch_glds<EPI, 2, 256, true>(s, A, W, bias, res, C, M, N, K, none);
        case 9:
            if (N % 256 == 0) return launch_256<EPI>(s, A, W, bias, res, C, M, N, K, none);
            return launch_glds<EPI, 2, 128, false, 4>(s, A, W, bias, res, C, M, N, K, none);
        case 2: return launch_glds<EPI, 2, 128>(s, A, W, bias, res, C, M, N, K, none);
        default: return launch_auto<EPI>(s, A, W, bias, res, C, M, N, K, none);
    }
}

// the LayerNorm-folded / statistics-emitting epilogues exist on the default kernels only
template <int EPI>
ssw_status launch_epi_ln(hipStream_t s, const bf16 *A, const bf16 *W, const float *bias, const float *res, void *C, int M,
                         int N, int K, const GemmLn &ln) {
    return launch_auto<EPI>(s, A, W, bias, res, C, M, N, K, ln);
}

}  // namespace

#ifdef SSW_DEBUG_HOOKS
void tune_gemm(int variant) { g_gemm_variant = variant; }
int gemm_variant() { return g_gemm_variant; }
#endif

ssw_status launch_gemm_bf16_nt(int epi, hipStream_t s, const void *A_, const void *W_, const float *bias,
                               const float *res, void *C, int M, int N, int K) {
    if (N % BN != 0 || K % BK != 0 || M <= 0) {
        set_error("gemm_bf16_nt: shape M=%d N=%d K=%d unsupported (N %% 128, K %% 64)", M, N, K);
        return SSW_ERR_UNSUPPORTED;
    }
#ifdef SSW_DEBUG_HOOKS  // the four-wave experiment (gemm_pw4.hip) is part of the lab build only
    if (g_gemm_variant >= 20 && g_gemm_variant <= 23 && gemm_pw4_supports(M, N, K)) {
        static const int bn_of[4] = {0, 256, 192, 128};
        int bn = bn_of[g_gemm_variant - 20];
        if (bn != 0 && N % bn != 0) bn = 0;
        return launch_gemm_pw4(epi, s, A_, W_, bias, res, C, M, N, K, bn);
    }
#endif
    const bf16 *A = static_cast<const bf16 *>(A_), *W = static_cast<const bf16 *>(W_);
    switch (epi) {
        case EPI_F32: return launch_epi<EPI_F32>(s, A, W, bias, res, C, M, N, K);
        case EPI_BF16_BIAS: return launch_epi<EPI_BF16_BIAS>(s, A, W, bias, res, C, M, N, K);
        case EPI_BF16_BIAS_GELU: return launch_epi<EPI_BF16_BIAS_GELU>(s, A, W, bias, res, C, M, N, K);
        case EPI_F32_BIAS_RESIDUAL: return launch_epi<EPI_F32_BIAS_RESIDUAL>(s, A, W, bias, res, C, M, N, K);
    }
    set_error("gemm_bf16_nt: unknown epilogue %d", epi);
    return SSW_ERR_INVALID;
}

ssw_status launch_gemm_bf16_ln(int epi, hipStream_t s, const void *A_, const void *W_, const float *bias,
                               const float *res, void *C, int M, int N, int K, const GemmLn &ln) {
    if (N % BN != 0 || K % BK != 0 || M <= 0) {
        set_error("gemm_bf16_ln: shape M=%d N=%d K=%d unsupported (N %% 128, K %% 64)", M, N, K);
        return SSW_ERR_UNSUPPORTED;
    }
    const bf16 *A = static_cast<const bf16 *>(A_), *W = static_cast<const bf16 *>(W_);
    switch (epi) {
        case EPI_BF16_LN:
        case EPI_BF16_LN_GELU:
            if (!ln.stats_in || !ln.c1 || !bias || ln.np_in <= 0 || ln.np_in > 8 || (ln.np_in & 1)) {
                set_error("gemm_bf16_ln: the LayerNorm-folded product needs statistics (an even number of partial pairs, <= 8), c1 and c2");
                return SSW_ERR_INVALID;
            }
            return epi == EPI_BF16_LN ? launch_epi_ln<EPI_BF16_LN>(s, A, W, bias, res, C, M, N, K, ln)
                                      : launch_epi_ln<EPI_BF16_LN_GELU>(s, A, W, bias, res, C, M, N, K, ln);
        case EPI_F32_BIAS_RESIDUAL_STATS:
            if (!ln.xcopy || !ln.stats_out || !res || !bias) {
                set_error("gemm_bf16_ln: the statistics epilogue needs the bf16 copy, the partial-sum buffer, bias and residual");
                return SSW_ERR_INVALID;
            }
            return launch_epi_ln<EPI_F32_BIAS_RESIDUAL_STATS>(s, A, W, bias, res, C, M, N, K, ln);
        case EPI_BF16_STREAM_STATS:
            if (!ln.xcopy || !ln.stats_out || !bias) {
                set_error("gemm_bf16_ln: the bf16-stream epilogue needs the stream rows, the partial-sum buffer and bias");
                return SSW_ERR_INVALID;
            }
            return launch_epi_ln<EPI_BF16_STREAM_STATS>(s, A, W, bias, nullptr, nullptr, M, N, K, ln);
    }
    set_error("gemm_bf16_ln: epilogue %d is not a LayerNorm-folded one", epi);
    return SSW_ERR_INVALID;
}

// ---------------------------------------------------------------------------------------
// A product that is a fraction of one round of tiles lasts as long as ONE workgroup's K loop (~1 us a 64-wide step
// whatever the tile): the last layer's fc2 on the 200 pooled rows of a B = 200 call is 12 tiles of 48 steps.  Split over
// K it is `splits` times as many workgroups with a `splits`-th of the steps each, the partial products (f32) added up
// in ascending order by a second launch together with bias and residual.
// ---------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void k_splitk_reduce(const float *__restrict__ part, int splits, int64_t mn, int N,
                                                       const float *__restrict__ bias, const float *__restrict__ residual,
                                                       float *__restrict__ out) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= mn) return;
    f32x4 v = *reinterpret_cast<const f32x4 *>(part + i);
    for (int z = 1; z < splits; ++z) v += *reinterpret_cast<const f32x4 *>(part + (int64_t)z * mn + i);
    v += *reinterpret_cast<const f32x4 *>(bias + (int)(i % N));
    if (residual) v += *reinterpret_cast<const f32x4 *>(residual + i);
    *reinterpret_cast<f32x4 *>(out + i) = v;
}
// the producers' epilogue (EPI_F32_BIAS_RESIDUAL_STATS) behind a split-K product: a row per workgroup, four columns a
// thread -- partial products added in ascending order, + bias + residual -> the f32 row, its bf16 copy, and the (sum,
// sum of squares) of each 128-column tile (32 threads: a butterfly inside the half wave) for the next product's LayerNorm
__global__ __launch_bounds__(256) void k_splitk_reduce_stats(const float *__restrict__ part, int splits, int M, int N,
                                                             const float *__restrict__ bias, const float *__restrict__ residual,
                                                             float *__restrict__ out, bf16 *__restrict__ xcopy,
                                                             float *__restrict__ stats_out) {
    const int row = blockIdx.x, t = threadIdx.x, col = t * 4;
    if (col >= N) return;  // (N / 4 is a multiple of 32: whole half waves leave)
    const int64_t o = (int64_t)row * N + col, mn = (int64_t)M * N;
    f32x4 v = *reinterpret_cast<const f32x4 *>(part + o);
    for (int z = 1; z < splits; ++z) v += *reinterpret_cast<const f32x4 *>(part + (int64_t)z * mn + o);
    v += *reinterpret_cast<const f32x4 *>(bias + col);
    v += *reinterpret_cast<const f32x4 *>(residual + o);
    *reinterpret_cast<f32x4 *>(out + o) = v;
    bf16x4 hb;
#pragma unroll
    for (int r = 0; r < 4; ++r) hb[r] = to_bf16(v[r]);
    *reinterpret_cast<bf16x4 *>(xcopy + o) = hb;
    float sm = (v[0] + v[1]) + (v[2] + v[3]);
    float sq = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
#pragma unroll
    for (int sh = 1; sh < 32; sh <<= 1) {
        sm += __shfl_xor(sm, sh, 64);
        sq += __shfl_xor(sq, sh, 64);
    }
    if ((t & 31) == 0) {
        float *so = stats_out + ((int64_t)row * (N / 128) + (t >> 5)) * 2;
        so[0] = sm;
        so[1] = sq;
    }
}
}  // namespace

// the first launch alone: partials[z][M][N] = A[:, z K/splits ...] W[:, z K/splits ...]^T; the caller adds them up (ascending z)
ssw_status launch_gemm_splitk_partials(hipStream_t s, const void *A, const void *W, float *partials, int M, int N, int K,
                                       int splits) {
    if (N % BN != 0 || M <= 0 || splits < 1 || K % (splits * BK) != 0 || !partials) {
        set_error("gemm_splitk: shape M=%d N=%d K=%d in %d splits unsupported (N %% 128, K %% (64 splits))", M, N, K, splits);
        return SSW_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL((gemm_bf16_nt<EPI_F32, true>), dim3(N / BN, (M + BM - 1) / BM, splits), dim3(256), 0, s,
                       static_cast<const bf16 *>(A), static_cast<const bf16 *>(W), (const float *)nullptr, (const float *)nullptr,
                       (void *)partials, M, N, K, K / splits);
    SSW_HIP_TRY(hipGetLastError());
    return SSW_OK;
}

ssw_status launch_gemm_splitk_f32(hipStream_t s, const void *A, const void *W, const float *bias, const float *residual,
                                  float *out, float *partials, int M, int N, int K, int splits) {
    if (N % BN != 0 || M <= 0 || splits < 1 || K % (splits * BK) != 0 || !bias || !partials) {
        set_error("gemm_splitk: shape M=%d N=%d K=%d in %d splits unsupported (N %% 128, K %% (64 splits))", M, N, K, splits);
        return SSW_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL((gemm_bf16_nt<EPI_F32, true>), dim3(N / BN, (M + BM - 1) / BM, splits), dim3(256), 0, s,
                       static_cast<const bf16 *>(A), static_cast<const bf16 *>(W), (const float *)nullptr, (const float *)nullptr,
                       (void *)partials, M, N, K, K / splits);
    const int64_t mn = (int64_t)M * N;
    hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((mn / 4 + 255) / 256)), dim3(256), 0, s, partials, splits, mn, N, bias,
                       residual, out);
    SSW_HIP_TRY(hipGetLastError());
    return SSW_OK;
}

}  // namespace ssw
// A producer product (f32 row + bf16 copy + statistics) of FEW tiles, split over K (round 5: the text tower's N = 512
// products at 16 x 77 rows are 40 tiles -- one workgroup's 32-step K loop long as one launch)
namespace ssw {
int splitk_choice(int M, int N, int K, int cus) {
    const int64_t tiles = (int64_t)((M + BM - 1) / BM) * (N / BN);
    if (tiles * 4 > cus || K < 4 * BK) return 1;         // a good part of a round already, or nothing to split
    int splits = 1;
    while (splits < 8 && tiles * (splits * 2) <= 2 * cus && K % (splits * 2 * BK) == 0 && K / (splits * 2) >= 2 * BK) splits *= 2;
    return splits;
}
ssw_status launch_gemm_splitk_stats(hipStream_t s, const void *A, const void *W, const float *bias, const float *residual,
                                    float *out, float *partials, int M, int N, int K, int splits, const GemmLn &ln) {
    if (N % BN != 0 || N > 1024 || !bias || !residual || !ln.xcopy || !ln.stats_out) {
        set_error("gemm_splitk_stats: N=%d unsupported or a NULL operand", N);
        return SSW_ERR_UNSUPPORTED;
    }
    SSW_TRY(launch_gemm_splitk_partials(s, A, W, partials, M, N, K, splits));
    hipLaunchKernelGGL(k_splitk_reduce_stats, dim3(M), dim3(256), 0, s, partials, splits, M, N, bias, residual, out, ln.xcopy,
                       ln.stats_out);
    SSW_HIP_TRY(hipGetLastError());
    return SSW_OK;
}
}  // namespace ssw


