// clip.hip -- CLIP ViT-B/32 image and text towers, bf16 MFMA forward (gfx950 / MI355X)
//
// Replaces the reference's calls into transformers.CLIPModel:
//   get_text_features   seesaw/models/embeddings.py:441-455 (HGWrapper.from_string)
//   get_image_features + F.normalize   seesaw/models/model.py:50-57 (HGFaceWrapper.forward),
//                                      batch driver seesaw/indices/multiscale/multiscale_tools.py:187-202
// The arithmetic itself lives in a third-party dependency of the reference
// (transformers, pinned 4.19.0 in its poetry.lock); this file restates that published
// architecture: pre-LN transformer, quick-GELU x*sigmoid(1.702x), LN eps 1e-5,
// vision: conv 32x32/32 patch embedding (no bias) + class token + learned positions,
// pre-LN, 12 layers (d 768, 12 heads, MLP 3072), post-LN on the class token, 768->512
// projection; text: token + position embeddings, 12 causal layers (d 512, 8 heads, MLP 2048),
// final LN, hidden state at the first EOS token, 512->512 projection.
//
// Roofline: dense contraction -> MFMA.  8.82 GFLOP per 224x224 tile, 5.96 GFLOP per
// 77-token text.  Every matmul is one kernel, gemm_bf16_nt: C = A W^T with
// A [M,K] and W [N,K] both K-contiguous bf16 (nn.Linear's own weight layout, so no transposes),
// 128x128x64 workgroup tiles staged through LDS (register double buffer, padded rows), 4 waves
// each owning a 64x64 sub-tile = 4x4 v_mfma_f32_16x16x32_bf16 accumulators, f32 accumulate,
// and the epilogue fused: +bias, quick-GELU, +residual, bf16 or f32 store.  LayerNorm,
// softmax and the residual stream stay f32; attention (50 / <=77 tokens per head) runs whole
// heads out of LDS.
#include <cmath>
#include <vector>

#include "ssw_common.h"

namespace ssw {
namespace {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
enum Epilogue { EPI_F32 = 0, EPI_BF16_BIAS = 1, EPI_BF16_BIAS_GELU = 2, EPI_F32_BIAS_RESIDUAL = 3, EPI_BF16_LN = 4, EPI_BF16_LN_GELU = 5,
                EPI_F32_BIAS_RESIDUAL_STATS = 6, EPI_BF16_STREAM_STATS = 7 };  // as gemm_bf16.hip

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
__device__ __forceinline__ bf16 to_bf16(float x) { return (bf16)x; }

// ---------------------------------------------------------------------------------------
// LayerNorm over the last dim (one wave per row), f32 in -> bf16 or f32 out.
// row_index (optional): normalise only the listed rows (class token / EOS positions).
// ---------------------------------------------------------------------------------------
// One wave per row; the row (D <= 1024 floats = 4 float4 per lane) stays in registers, so x is
// read once with 16-byte loads and mean / variance / normalise are a single pass.
template <typename OutT>
__global__ __launch_bounds__(256) void layernorm_rows(const float *__restrict__ x, const int *__restrict__ row_index,
                                                      int n_rows, int D, const float *__restrict__ w,
                                                      const float *__restrict__ b, float eps, OutT *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const float4 *xr = reinterpret_cast<const float4 *>(x + (int64_t)(row_index ? row_index[r] : r) * D);
    const int nv = D >> 2;  // float4 per row (D % 4 == 0)
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < nv ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (lane + 64 * i < nv) {
            const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off, 64);
    const float rstd = rsqrtf(q / (float)D + eps);
    const float4 *w4 = reinterpret_cast<const float4 *>(w), *b4 = reinterpret_cast<const float4 *>(b);
    OutT *o = out + (int64_t)r * D;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const float4 ww = w4[c], bb = b4[c];
            const float y0 = (v[i].x - mean) * rstd * ww.x + bb.x, y1 = (v[i].y - mean) * rstd * ww.y + bb.y;
            const float y2 = (v[i].z - mean) * rstd * ww.z + bb.z, y3 = (v[i].w - mean) * rstd * ww.w + bb.w;
            o[4 * c + 0] = (OutT)y0;
            o[4 * c + 1] = (OutT)y1;
            o[4 * c + 2] = (OutT)y2;
            o[4 * c + 3] = (OutT)y3;
        }
    }
}

// The row that ENTERS the transformer stack of the tile path (vision: the pre-LayerNorm's output; text: the token +
// position embedding): besides the f32 residual row, the stack's first QKV product wants its bf16 copy and its
// LayerNorm statistics as `np` partial (sum, sum of squares) pairs (GemmLn, ssw_common.h) -- the whole row in partial 0.
// LN = true: y = LayerNorm(x) with (w, b) is the row (x is consumed); LN = false: y = x.
// With `pos` given, x is the patch embedding's output and the row is put together here, the image tower's input:
// token 0 of an image = cls + pos[0], token 1 + p = x[image * (T - 1) + p] + pos[1 + p].  y_out may be null (bf16 stream).
template <bool LN>
__global__ __launch_bounds__(256) void stack_input_rows(const float *__restrict__ x, int n_rows, int D,
                                                        const float *__restrict__ w, const float *__restrict__ b,
                                                        float eps, float *__restrict__ y_out, bf16 *__restrict__ y_bf16,
                                                        float *__restrict__ stats, int np,
                                                        const float *__restrict__ cls = nullptr,
                                                        const float *__restrict__ pos = nullptr, int T = 1) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const float4 *xr = reinterpret_cast<const float4 *>(x + (int64_t)r * D);
    const float4 *pr = nullptr;
    if (pos) {
        const int tk = r % T, img = r / T;
        xr = reinterpret_cast<const float4 *>(tk == 0 ? cls : x + ((int64_t)img * (T - 1) + tk - 1) * D);
        pr = reinterpret_cast<const float4 *>(pos + (int64_t)tk * D);
    }
    const int nv = D >> 2;
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < nv ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (pr && c < nv) {
            const float4 pp = pr[c];
            v[i].x += pp.x; v[i].y += pp.y; v[i].z += pp.z; v[i].w += pp.w;
        }
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    if (LN) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        const float mean = s / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (lane + 64 * i < nv) {
                const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
                q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off, 64);
        const float rstd = rsqrtf(q / (float)D + eps);
        const float4 *w4 = reinterpret_cast<const float4 *>(w), *b4 = reinterpret_cast<const float4 *>(b);
        s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                const float4 ww = w4[c], bb = b4[c];
                v[i].x = (v[i].x - mean) * rstd * ww.x + bb.x;
                v[i].y = (v[i].y - mean) * rstd * ww.y + bb.y;
                v[i].z = (v[i].z - mean) * rstd * ww.z + bb.z;
                v[i].w = (v[i].w - mean) * rstd * ww.w + bb.w;
                s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            }
        }
    }
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) q2 += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s += __shfl_xor(s, off, 64);
        q2 += __shfl_xor(q2, off, 64);
    }
    float4 *yo = reinterpret_cast<float4 *>(y_out + (int64_t)r * D);
    bf16x4 *yb = reinterpret_cast<bf16x4 *>(y_bf16 + (int64_t)r * D);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            if (LN && y_out) yo[c] = v[i];
            bf16x4 h;
            h[0] = (bf16)v[i].x; h[1] = (bf16)v[i].y; h[2] = (bf16)v[i].z; h[3] = (bf16)v[i].w;
            yb[c] = h;
        }
    }
    if (lane < np) {
        float *st = stats + ((int64_t)r * np + lane) * 2;
        st[0] = lane == 0 ? s : 0.f;
        st[1] = lane == 0 ? q2 : 0.f;
    }
}

// ---------------------------------------------------------------------------------------
// attention: one WORKGROUP per (batch, head), one wave per 16-query tile; S <= 16*NT tokens,
// head_dim 64.  The MFMA operands are ordered so that a lane owns ONE query:
//   scores^T = K Q^T     16x16x32 MFMA, K / Q fragments straight from global (K-contiguous rows
//                        of the fused qkv buffer, the GEMM's "NT" operand form); the lane holds
//                        keys 16 j + 4 fq + r of query fr -> row max / sum need two shuffles
//   softmax   in registers (f32), normalised before the bf16 rounding
//   out^T   = V^T P^T    P goes through a per-wave 16-row LDS tile (8-byte writes; 128-byte rows with the GEMM's
//                        chunk ^= (row>>1)&7 swizzle when KP = 64, so the ds_read_b128 lane groups hit distinct
//                        slots); V is copied row-major into LDS (16-byte writes) and its V^T fragments come out of
//                        ds_read_b64_tr_b16 (round 2: the 2-byte transposing writes were 0.73 conflict cycles per
//                        active LDS cycle); rows are 128 B, the 32-byte chunk of a key row is XORed with
//                        (key>>1 & 1) | (key>>3 & 1) << 1 so that the eight rows a 32-lane half reads land on
//                        disjoint banks; the lane ends with 4 consecutive head-dim values of its query
// qkv [R, 3D] bf16 (q | k | v), out [R, D] bf16
// ---------------------------------------------------------------------------------------
constexpr int ATT_MAX_S = 80;

// TPW: query tiles per wave (round 3).  With one tile per wave a (image, head) pair is four waves and a CU holds eight pairs;
// 2400 pairs on 2048 places run as a full round and a nearly empty one.  With two tiles per wave a pair is two waves, all
// pairs are resident at once, and a wave reads its K fragments once for both of its tiles.  Per pair the arithmetic is
// unchanged (same fragments, same MFMA order per tile): identical output.
template <int NT, int TPW = 1>
__global__ __launch_bounds__(NT / TPW * 64) void attention_mfma(const bf16 *__restrict__ qkv, bf16 *__restrict__ out, int S,
                                                          int D, int H, float scale, int causal) {
    constexpr int KP = ((NT * 16 + 31) / 32) * 32;  // keys padded to the MFMA K step
    constexpr bool PSW = KP == 64;                  // P rows of exactly 128 B: swizzled instead of padded
    constexpr int LDP = PSW ? 64 : KP + 8;          // LDS row stride of the P tile (bf16)
    static_assert(NT % TPW == 0, "whole waves");
    constexpr int NW = NT / TPW;                    // waves per workgroup
    constexpr int VCH = (KP * 8 + NW * 64 - 1) / (NW * 64);  // 16-byte V chunks per thread
    __shared__ __attribute__((aligned(16))) bf16 sV[KP * 64];  // [key][64 head dims], 32-byte chunks swizzled
    __shared__ __attribute__((aligned(16))) bf16 sP[NT][16 * LDP];  // one 16-row tile per query tile
    // element offset of column `col` (a multiple of 4) of P row `row`
    auto p_off = [](int row, int col) {
        return PSW ? row * 64 + ((((col >> 3) ^ ((row >> 1) & 7)) << 3) | (col & 7)) : row * LDP + col;
    };
    // element offset of 32-byte chunk `c32` (16 head dims) of V row `key`
    auto v_off = [](int key, int c32) { return key * 64 + ((c32 ^ (((key >> 1) & 1) | (((key >> 3) & 1) << 1))) << 4); };
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int fr = lane & 15, fq = lane >> 4;
    const bf16 *base = qkv + (int64_t)b * S * 3 * D + h * 64;

    // V rows of this head (whole workgroup, 128-byte rows); keys >= S are zero
    bf16x8 vreg[VCH];
#pragma unroll
    for (int c = 0; c < VCH; ++c) {
        const int ch = t + c * NW * 64, key = ch >> 3, d0 = (ch & 7) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) vreg[c][j] = (bf16)0.f;
        if (ch < KP * 8 && key < S) vreg[c] = *reinterpret_cast<const bf16x8 *>(base + (int64_t)key * 3 * D + 2 * D + d0);
    }
    const bool wave_live = wave * TPW * 16 < S;  // wave-uniform: its first tile has a live query
    if (wave_live) {
      // K fragments of every key tile, once for all of this wave's query tiles
      bf16x8 kf[NT][2];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
          const int key = min(j * 16 + fr, S - 1);  // clamped rows are masked below
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
              kf[j][ks] = *reinterpret_cast<const bf16x8 *>(base + (int64_t)key * 3 * D + D + ks * 32 + fq * 8);
      }
#pragma unroll
      for (int tt = 0; tt < TPW; ++tt) {
        const int i = wave * TPW + tt;  // query tile
        if (i * 16 >= S) break;         // wave-uniform
        bf16 *pt = sP[i];
        f32x4 sc[NT];
        const int qrow = min(i * 16 + fr, S - 1);
        bf16x8 qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf[ks] = *reinterpret_cast<const bf16x8 *>(base + (int64_t)qrow * 3 * D + ks * 32 + fq * 8);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            sc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                sc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[j][ks], qf[ks], sc[j], 0, 0, 0);
        }
        // this lane holds scores[query = 16 i + fr][key = 16 j + 4 fq + r]
        const int q = i * 16 + fr;
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = j * 16 + fq * 4 + r;
                float v = sc[j][r] * scale;
                if (key >= S || (causal && key > q)) v = -INFINITY;
                sc[j][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(sc[j][r] - mx);  // key 0 is never masked, so mx is finite
                sc[j][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        // P (bf16) -> this wave's LDS tile [16 queries][KP keys]
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            bf16x4 pv;
#pragma unroll
            for (int r = 0; r < 4; ++r) pv[r] = to_bf16(sc[j][r] * inv);
            *reinterpret_cast<bf16x4 *>(&pt[p_off(fr, j * 16 + fq * 4)]) = pv;
        }
        if (KP > NT * 16) {
            bf16x4 z;
#pragma unroll
            for (int r = 0; r < 4; ++r) z[r] = (bf16)0.f;
            *reinterpret_cast<bf16x4 *>(&pt[p_off(fr, NT * 16 + fq * 4)]) = z;
        }
      }
    }
    // V into LDS as it is: sV[key][d]
#pragma unroll
    for (int c = 0; c < VCH; ++c) {
        const int ch = t + c * NW * 64, key = ch >> 3, c16 = ch & 7;
        if (ch < KP * 8) *reinterpret_cast<bf16x8 *>(&sV[v_off(key, c16 >> 1) + (c16 & 1) * 8]) = vreg[c];
    }
    __syncthreads();
    if (!wave_live) return;  // whole waves leave: the transposed reads below need all 64 lanes of a wave
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
    const int i = wave * TPW + tt;
    if (i * 16 >= S) break;
    const bf16 *pt = sP[i];
    // out^T tile = V^T P^T : o[dt][r] = out[query fr][d = 16 dt + 4 fq + r]
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KP / 32; ++ks) {
        const bf16x8 pa = *reinterpret_cast<const bf16x8 *>(&pt[p_off(fr, ks * 32 + fq * 8)]);
        // V^T fragment (row d = 16 dt + fr, keys ks*32 + 8 fq + 0..7) = two transposed reads of 4 keys x 16 dims: in
        // each 16-lane group lane 4q + p supplies the address of key row q, dims 4p .. 4p+3, and receives dim `lane`
        const int kq = ks * 32 + 8 * fq + (fr >> 2), dp = (fr & 3) * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            typedef __attribute__((ext_vector_type(4))) short s16x4;
            typedef __attribute__((address_space(3))) s16x4 *lds_s16x4;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(&sV[v_off(kq, dt) + dp]));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(&sV[v_off(kq + 4, dt) + dp]));
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            const bf16x8 vb = __builtin_bit_cast(bf16x8, both);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb, pa, o[dt], 0, 0, 0);
        }
    }
    const int q = i * 16 + fr;
    if (q < S) {
        bf16 *orow = out + ((int64_t)b * S + q) * D + h * 64 + fq * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            bf16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = to_bf16(o[dt][r]);
            *reinterpret_cast<bf16x4 *>(orow + dt * 16) = ov;
        }
    }
    }
}

// Round 3, S <= 64 (the image tower): the same arithmetic with every global access a whole 128-byte row.  Above, a lane
// fetches its K / Q fragments straight from memory -- 16 bytes each from 16 different rows per instruction, K once per
// wave -- and stores the 4 head dims it ends with (8 bytes): 16- and 8-byte requests, and from B ~ 100 on the kernel ran
// at the rate of its requests, not of its 61 MB.  Here the workgroup's 128 threads fetch the K, Q and V rows of the
// (image, head) pair coalesced (8 lanes a row, K once), K and Q go to LDS in the GEMM's swizzled 128-byte rows and the
// fragments are read from there; a query tile's 2 KB of LDS holds first its Q rows, then its P tile, then its output
// tile, which leaves row-major (16 bytes a lane); V takes K's place once every wave has its K fragments.  16 KB of LDS
// as before.  Same fragments, same MFMA order: identical output.
__global__ __launch_bounds__(128) void attention_rows64(const bf16 *__restrict__ qkv, bf16 *__restrict__ out, int S, int D, int H,
                                                        float scale, int causal) {
    constexpr int NT = 4, TPW = 2, KP = 64;
    __shared__ __attribute__((aligned(16))) bf16 sKV[KP * 64];      // K rows (GEMM swizzle), later V rows (v_off)
    __shared__ __attribute__((aligned(16))) bf16 sQP[NT][16 * 64];  // per query tile: Q rows, then P, then the output
    // element offset of 16-byte chunk c16 of a 128-byte row
    auto g_off = [](int row, int c16) { return row * 64 + ((c16 ^ ((row >> 1) & 7)) << 3); };
    auto v_off = [](int key, int c32) { return key * 64 + ((c32 ^ (((key >> 1) & 1) | (((key >> 3) & 1) << 1))) << 4); };
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int fr = lane & 15, fq = lane >> 4;
    const bf16 *base = qkv + (int64_t)b * S * 3 * D + h * 64;

    bf16x8 kreg[4], qreg[4], vreg[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ch = t + c * 128, row = ch >> 3, d0 = (ch & 7) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) kreg[c][j] = qreg[c][j] = vreg[c][j] = (bf16)0.f;
        if (row < S) {
            const bf16 *r = base + (int64_t)row * 3 * D + d0;
            qreg[c] = *reinterpret_cast<const bf16x8 *>(r);
            kreg[c] = *reinterpret_cast<const bf16x8 *>(r + D);
            vreg[c] = *reinterpret_cast<const bf16x8 *>(r + 2 * D);
        }
    }
    bf16 *const sQ = &sQP[0][0];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ch = t + c * 128, row = ch >> 3, c16 = ch & 7;
        *reinterpret_cast<bf16x8 *>(&sKV[g_off(row, c16)]) = kreg[c];
        *reinterpret_cast<bf16x8 *>(&sQ[g_off(row, c16)]) = qreg[c];
    }
    __syncthreads();
    const bool wave_live = wave * TPW * 16 < S;  // wave-uniform: its first tile has a live query
    if (wave_live) {
        bf16x8 kf[NT][2];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) kf[j][ks] = *reinterpret_cast<const bf16x8 *>(&sKV[g_off(j * 16 + fr, ks * 4 + fq)]);
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            const int i = wave * TPW + tt;  // query tile
            if (i * 16 >= S) break;         // wave-uniform
            bf16 *pt = sQP[i];
            f32x4 sc[NT];
            bf16x8 qf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const bf16x8 *>(&pt[g_off(fr, ks * 4 + fq)]);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                sc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) sc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[j][ks], qf[ks], sc[j], 0, 0, 0);
            }
            // this lane holds scores[query = 16 i + fr][key = 16 j + 4 fq + r]
            const int q = i * 16 + fr;
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = j * 16 + fq * 4 + r;
                    float v = sc[j][r] * scale;
                    if (key >= S || (causal && key > q)) v = -INFINITY;
                    sc[j][r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __expf(sc[j][r] - mx);  // key 0 is never masked, so mx is finite
                    sc[j][r] = e;
                    sum += e;
                }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.f / sum;
            // P (bf16) over this tile's Q rows (its fragments are in registers): [16 queries][64 keys]
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                bf16x4 pv;
#pragma unroll
                for (int r = 0; r < 4; ++r) pv[r] = to_bf16(sc[j][r] * inv);
                *reinterpret_cast<bf16x4 *>(&pt[g_off(fr, j * 2 + (fq >> 1)) + (fq & 1) * 4]) = pv;
            }
        }
    }
    __syncthreads();  // every wave has its K fragments: V takes K's bytes
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ch = t + c * 128, key = ch >> 3, c16 = ch & 7;
        *reinterpret_cast<bf16x8 *>(&sKV[v_off(key, c16 >> 1) + (c16 & 1) * 8]) = vreg[c];
    }
    __syncthreads();
    if (!wave_live) return;  // whole waves leave: the transposed reads below need all 64 lanes of a wave
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        const int i = wave * TPW + tt;
        if (i * 16 >= S) break;
        bf16 *pt = sQP[i];
        // out^T tile = V^T P^T : o[dt][r] = out[query fr][d = 16 dt + 4 fq + r]
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KP / 32; ++ks) {
            const bf16x8 pa = *reinterpret_cast<const bf16x8 *>(&pt[g_off(fr, ks * 4 + fq)]);
            const int kq = ks * 32 + 8 * fq + (fr >> 2), dp = (fr & 3) * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                typedef __attribute__((ext_vector_type(4))) short s16x4;
                typedef __attribute__((address_space(3))) s16x4 *lds_s16x4;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(&sKV[v_off(kq, dt) + dp]));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(&sKV[v_off(kq + 4, dt) + dp]));
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                const bf16x8 vb = __builtin_bit_cast(bf16x8, both);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb, pa, o[dt], 0, 0, 0);
            }
        }
        // the output tile over the P tile (read above by this wave only), then out row-major: 8 lanes a row
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            bf16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = to_bf16(o[dt][r]);
            *reinterpret_cast<bf16x4 *>(&pt[g_off(fr, dt * 2 + (fq >> 1)) + (fq & 1) * 4]) = ov;
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int row = it * 8 + (lane >> 3), c16 = lane & 7;
            const bf16x8 h8 = *reinterpret_cast<const bf16x8 *>(&pt[g_off(row, c16)]);
            const int q = i * 16 + row;
            if (q < S) *reinterpret_cast<bf16x8 *>(out + ((int64_t)b * S + q) * D + h * 64 + c16 * 8) = h8;
        }
    }
}

// ---------------------------------------------------------------------------------------
// skinny linear: out[R <= 32, N] = epilogue(LN?(X)[R, K] W[N, K]^T) for the handful of rows a single text query has.
// The tile kernels above give such a product to N/128 workgroups that each walk all of K (a [8 x 2048] x [2048 x 512]
// fc2 takes 20 us on four workgroups); here a workgroup owns 16 output columns and has one wave per 128 of K: each wave
// reads its 16 x 128 slice of W straight from memory into four v_mfma_f32_16x16x32_bf16 fragments (X rows are the
// other operand, zero beyond R), every load of the kernel is in flight before the first wait, and the K/128 partial
// tiles are added through LDS in wave order.  With LN the layer norm of the rows (layernorm_rows's two-pass
// arithmetic, eight lanes a row) is redone by every workgroup from the f32 residual stream -- R x D floats out of L2,
// under the latency of the W loads -- instead of being a launch of its own.  Epilogues are gemm_bf16.hip's.
// ---------------------------------------------------------------------------------------
enum { SK_F32 = 0, SK_BF16_BIAS = 1, SK_BF16_BIAS_GELU = 2, SK_F32_BIAS_RESIDUAL = 3 };
constexpr int SK_MAX_ROWS = 32;

template <int CTRL>
__device__ __forceinline__ float sk_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// sum over each aligned group of eight lanes, in every lane of the group
__device__ __forceinline__ float sk_sum8(float v) {
    v += sk_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
    v += sk_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
    v += sk_dpp<0x141>(v);  // row_half_mirror: the other quad of the eight
    return v;
}

template <int EPI, bool LN, int RT, int NW>
__global__ __launch_bounds__(64 * NW) void skinny_linear(const void *__restrict__ Xv, const int *__restrict__ row_index,
                                                         const float *__restrict__ lnw, const float *__restrict__ lnb,
                                                         float eps, const bf16 *__restrict__ W,
                                                         const float *__restrict__ bias,
                                                         const float *__restrict__ residual, void *__restrict__ Cout,
                                                         int R, int N) {
    constexpr int K = NW * 128;
    __shared__ f32x4 part[NW][RT][64];
    __shared__ float stats[SK_MAX_ROWS][2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 16, kb = wave * 128 + fq * 8;
    // ---- every load of the kernel is issued here, before the first wait ----
    const int slot = tid >> 3, sub = tid & 7;  // layer-norm statistics: threads 0..255, eight lanes a row
    const bool stat_thread = LN && tid < 256 && slot < R;
    float4 sv[LN ? K / 32 : 1];
    if (stat_thread) {
        const float4 *xr = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(Xv) +
                                                             (int64_t)(row_index ? row_index[slot] : slot) * K);
#pragma unroll
        for (int i = 0; i < K / 32; ++i) sv[i] = xr[sub + 8 * i];
    }
    bf16x8 wf[4];
    {
        const bf16 *wrow = W + (int64_t)(n0 + fr) * K + kb;
#pragma unroll
        for (int u = 0; u < 4; ++u) wf[u] = *reinterpret_cast<const bf16x8 *>(wrow + 32 * u);
    }
    bool live[RT];
    bf16x8 xb[LN ? 1 : RT][4];         // !LN: the bf16 rows as they are
    float4 xf32[LN ? RT : 1][4][2];    // LN: the f32 rows, normalised below
    float4 lw[LN ? 4 : 1][2], lb[LN ? 4 : 1][2];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int r = t * 16 + fr;
        live[t] = r < R;
        const int64_t x0 = live[t] ? (int64_t)((LN && row_index) ? row_index[r] : r) * K + kb : 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (LN) {
                const float *xp = reinterpret_cast<const float *>(Xv) + x0 + 32 * u;
                xf32[LN ? t : 0][u][0] = live[t] ? *reinterpret_cast<const float4 *>(xp) : make_float4(0.f, 0.f, 0.f, 0.f);
                xf32[LN ? t : 0][u][1] = live[t] ? *reinterpret_cast<const float4 *>(xp + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else if (live[t]) {
                xb[LN ? 0 : t][u] = *reinterpret_cast<const bf16x8 *>(reinterpret_cast<const bf16 *>(Xv) + x0 + 32 * u);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) xb[LN ? 0 : t][u][e] = to_bf16(0.f);
            }
        }
    }
    if (LN) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            lw[u][0] = *reinterpret_cast<const float4 *>(lnw + kb + 32 * u);
            lw[u][1] = *reinterpret_cast<const float4 *>(lnw + kb + 32 * u + 4);
            lb[u][0] = *reinterpret_cast<const float4 *>(lnb + kb + 32 * u);
            lb[u][1] = *reinterpret_cast<const float4 *>(lnb + kb + 32 * u + 4);
        }
    }
    // the epilogue's operands (waves 0 .. RT-1 write rows wave*16 + fr, columns n0 + fq*4 ..)
    const int erow = wave * 16 + fr, ecol = n0 + fq * 4;
    const bool writer = wave < RT && erow < R;
    const int64_t eo = (int64_t)erow * N + ecol;
    f32x4 bias_v = f32x4{0.f, 0.f, 0.f, 0.f}, res_v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (writer) {
        if (EPI != SK_F32) bias_v = *reinterpret_cast<const f32x4 *>(bias + ecol);
        if (EPI == SK_F32_BIAS_RESIDUAL) res_v = *reinterpret_cast<const f32x4 *>(residual + eo);
    }
    // ---- layer-norm statistics (layernorm_rows's two passes over the registers) ----
    if (LN) {
        if (stat_thread) {
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < K / 32; ++i) sum += (sv[i].x + sv[i].y) + (sv[i].z + sv[i].w);
            const float mean = sk_sum8(sum) / (float)K;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < K / 32; ++i) {
                const float dx = sv[i].x - mean, dy = sv[i].y - mean, dz = sv[i].z - mean, dw = sv[i].w - mean;
                q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
            q = sk_sum8(q);
            if (sub == 0) {
                stats[slot][0] = mean;
                stats[slot][1] = rsqrtf(q / (float)K + eps);
            }
        }
        __syncthreads();
    }
    // ---- this wave's 128 of K ----
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        const int r = t * 16 + fr;
        const float mean = (LN && live[t]) ? stats[r][0] : 0.f, rstd = (LN && live[t]) ? stats[r][1] : 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            bf16x8 xf;
            if (!LN) {
                xf = xb[LN ? 0 : t][u];
            } else {
                const float4 x0 = xf32[LN ? t : 0][u][0], x1 = xf32[LN ? t : 0][u][1];
                const float4 w0 = lw[LN ? u : 0][0], w1 = lw[LN ? u : 0][1], b0 = lb[LN ? u : 0][0], b1 = lb[LN ? u : 0][1];
                // rows beyond R: x = 0, mean = rstd = 0 -> the bias b, multiplied into tile rows nobody writes
                xf[0] = to_bf16((x0.x - mean) * rstd * w0.x + b0.x);
                xf[1] = to_bf16((x0.y - mean) * rstd * w0.y + b0.y);
                xf[2] = to_bf16((x0.z - mean) * rstd * w0.z + b0.z);
                xf[3] = to_bf16((x0.w - mean) * rstd * w0.w + b0.w);
                xf[4] = to_bf16((x1.x - mean) * rstd * w1.x + b1.x);
                xf[5] = to_bf16((x1.y - mean) * rstd * w1.y + b1.y);
                xf[6] = to_bf16((x1.z - mean) * rstd * w1.z + b1.z);
                xf[7] = to_bf16((x1.w - mean) * rstd * w1.w + b1.w);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], xf, acc, 0, 0, 0);  // lane: out[row fr][col fq*4 + e]
        }
        part[wave][t][lane] = acc;
    }
    __syncthreads();
    if (!writer) return;
    f32x4 v = part[0][wave][lane];
#pragma unroll
    for (int w = 1; w < NW; ++w) v += part[w][wave][lane];
    v += bias_v;
    if (EPI == SK_BF16_BIAS_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] / (1.f + __expf(-1.702f * v[e]));  // quick_gelu
    }
    v += res_v;
    if (EPI == SK_BF16_BIAS || EPI == SK_BF16_BIAS_GELU) {
        bf16x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = to_bf16(v[e]);
        *reinterpret_cast<bf16x4 *>(reinterpret_cast<bf16 *>(Cout) + eo) = ov;
    } else {
        *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(Cout) + eo) = v;
    }
}

// ---------------------------------------------------------------------------------------
// skinny attention + out-projection: h2[R <= 32, 512] = h + bo + softmax(Q K^T / 8, block-causal) V  Wo^T straight from
// the fused qkv rows -- the attention of a short text query is a few MFMAs, a launch of its own costs more than it
// does.  Workgroup = 16 output columns, four waves; wave w owns heads 2w and 2w+1, i.e. the 128 of K = 512 that the
// out-projection's wave w multiplies, so every wave computes exactly the attention output it consumes and nothing goes
// through memory or LDS in between (each of the 32 workgroups redoes the 8 heads: 32 x a few microseconds of MFMA).
//   scores^T = K Q^T        lane (fr, fq) ends with scores[query 16i + fr][key 16j + 4fq + r]   (as attention_mfma)
//   out^T    = V^T P^T      the k index of an MFMA is only a summation label: k-slot 4c + r of lane group fq is
//                           key 16c + 4fq + r, which is where the softmaxed scores already are; the V^T operand is
//                           read to match (2-byte loads: V[key][d], eight keys per fragment)
//   h2       = att Wo^T     same trick: k-slot 4c + r of group fq is head dim 16(2v + c) + 4fq + r, where the
//                           attention output already is; the Wo fragment is two 8-byte loads to match
// Rows of several sequences (B x L <= 32) share the tiles: a key counts for a query of the same sequence at or before
// it (the text tower is causal; `causal = 0` keeps the whole sequence).
// ---------------------------------------------------------------------------------------
template <int RT>
__global__ __launch_bounds__(256) void skinny_attn_out(const bf16 *__restrict__ qkv, const bf16 *__restrict__ Wo,
                                                       const float *__restrict__ bias,
                                                       const float *__restrict__ residual, float *__restrict__ out,
                                                       int R, int L, float scale, int causal) {
    constexpr int D = 512, K3 = 3 * D;
    __shared__ f32x4 part[4][RT][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    // the epilogue's operands first: their latency runs under the attention
    const int erow = wave * 16 + fr, ecol = n0 + fq * 4;
    const bool writer = wave < RT && erow < R;
    const int64_t eo = (int64_t)erow * D + ecol;
    f32x4 bias_v = f32x4{0.f, 0.f, 0.f, 0.f}, res_v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (writer) {
        bias_v = *reinterpret_cast<const f32x4 *>(bias + ecol);
        res_v = *reinterpret_cast<const f32x4 *>(residual + eo);
    }
    f32x4 acc[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int hcol = (2 * wave + hh) * 64;  // this head's columns inside q, k, v and inside Wo's K
        // Wo fragments of the head's two k-steps (v = 0, 1): columns hcol + 32v + 16c + 4fq + r of row n0 + fr
        s16x8 wf[2];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const bf16 *wp = Wo + (int64_t)(n0 + fr) * D + hcol + 32 * v + 4 * fq;
            const s16x4 w0 = *reinterpret_cast<const s16x4 *>(wp), w1 = *reinterpret_cast<const s16x4 *>(wp + 16);
            wf[v] = __builtin_shufflevector(w0, w1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
        // Q and K fragments: rows 16t + fr (clamped; masked below), dims 32 ks + 8 fq ..
        bf16x8 qf[RT][2], kf[RT][2];
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const bf16 *rowp = qkv + (int64_t)min(t * 16 + fr, R - 1) * K3 + hcol + fq * 8;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                qf[t][ks] = *reinterpret_cast<const bf16x8 *>(rowp + ks * 32);
                kf[t][ks] = *reinterpret_cast<const bf16x8 *>(rowp + D + ks * 32);
            }
        }
        // V^T fragments: row d = 16 dt + fr, k-slot 4c + r = key 16c + 4fq + r
        s16x8 vt[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int key = 16 * (e >> 2) + 4 * fq + (e & 3);
                const bool live = key < R && (e >> 2) < RT;
                vt[dt][e] = live ? *reinterpret_cast<const short *>(qkv + (int64_t)key * K3 + 2 * D + hcol + 16 * dt + fr)
                                 : (short)0;
            }
        }
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            // scores of query 16i + fr against keys 16j + 4fq + r
            f32x4 sc[RT];
            const int q = i * 16 + fr, qseq = q / L;
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < RT; ++j) {
                sc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    sc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[j][ks], qf[i][ks], sc[j], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = j * 16 + fq * 4 + r;
                    float v = sc[j][r] * scale;
                    if (key >= R || key / L != qseq || (causal && key > q)) v = -INFINITY;
                    sc[j][r] = v;
                    mx = fmaxf(mx, v);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            if (q >= R) mx = 0.f;  // rows nobody writes: keep them finite
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < RT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __expf(sc[j][r] - mx);
                    sc[j][r] = e;
                    sum += e;
                }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = q < R ? 1.f / sum : 0.f;
            // P^T operand: k-slot 4c + r = key 16c + 4fq + r (zero beyond the RT key tiles)
            bf16x8 pb;
#pragma unroll
            for (int e = 0; e < 8; ++e) pb[e] = (e >> 2) < RT ? to_bf16(sc[(e >> 2) < RT ? (e >> 2) : 0][e & 3] * inv) : to_bf16(0.f);
            // attention output of the head: lane (fr = query, fq) holds dims 16 dt + 4fq + r
            f32x4 o[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vt[dt]), pb,
                                                                f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            // into the out-projection: k-step v takes dims 16(2v + c) + 4fq + r, rounded to bf16 as the att buffer was
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                bf16x8 ab;
#pragma unroll
                for (int e = 0; e < 8; ++e) ab[e] = to_bf16(o[2 * v + (e >> 2)][e & 3]);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[v]), ab, acc[i], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < RT; ++i) part[wave][i][lane] = acc[i];
    __syncthreads();
    if (!writer) return;
    f32x4 v = part[0][wave][lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) v += part[w][wave][lane];
    v += bias_v;
    v += res_v;
    *reinterpret_cast<f32x4 *>(out + eo) = v;
}

// ---------------------------------------------------------------------------------------
// embeddings
// ---------------------------------------------------------------------------------------
// pixels [B,3,224,224] f32 -> patches [B*49, 3072] bf16, column = c*1024 + py*32 + px
__global__ void im2col_patches(const float *__restrict__ px, bf16 *__restrict__ out, int B, int img, int patch) {
    const int g = img / patch;  // 7
    const int pp = patch * patch;
    const int cols8 = 3 * pp / 8;  // 8 consecutive pixels of one patch row per thread (patch % 8 == 0)
    const int64_t total = (int64_t)B * g * g * cols8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % cols8) * 8;
        const int64_t r = i / cols8;
        const int p = (int)(r % (g * g));
        const int b = (int)(r / (g * g));
        const int c = col / pp, py = (col % pp) / patch, pxx = col % patch;
        const int y = (p / g) * patch + py, x = (p % g) * patch + pxx;
        const float4 *src = reinterpret_cast<const float4 *>(px + (((int64_t)b * 3 + c) * img + y) * img + x);
        const float4 v0 = src[0], v1 = src[1];
        bf16x8 o;
        o[0] = to_bf16(v0.x); o[1] = to_bf16(v0.y); o[2] = to_bf16(v0.z); o[3] = to_bf16(v0.w);
        o[4] = to_bf16(v1.x); o[5] = to_bf16(v1.y); o[6] = to_bf16(v1.z); o[7] = to_bf16(v1.w);
        *reinterpret_cast<bf16x8 *>(out + i * 8) = o;
    }
}

// uint8 HWC tiles [B,224,224,3] (what the tiler produces, multiscale_tools.py:45-71) -> patches bf16,
// with batch_tx's arithmetic fused in (multiscale_tools.py:167-183: x / 255, then (x - mean) / std in
// f32): the f32 NCHW tensor is never materialised and the host hands over a quarter of the bytes.
__global__ void im2col_patches_u8(const uint8_t *__restrict__ tiles, bf16 *__restrict__ out, int B, int img, int patch,
                                  float m0, float m1, float m2, float s0, float s1, float s2) {
    const int g = img / patch;
    const int pp = patch * patch;
    const int per_row = patch / 8;  // 8-pixel groups per patch row
    const int64_t total = (int64_t)B * g * g * patch * per_row;
    const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xg = (int)(i % per_row);
        int64_t r = i / per_row;
        const int py = (int)(r % patch);
        r /= patch;
        const int p = (int)(r % (g * g));
        const int b = (int)(r / (g * g));
        const int y = (p / g) * patch + py, x = (p % g) * patch + xg * 8;
        const uint64_t *src = reinterpret_cast<const uint64_t *>(tiles + (((int64_t)b * img + y) * img + x) * 3);
        const uint64_t w0 = src[0], w1 = src[1], w2 = src[2];  // 8 pixels x RGB = 24 bytes
        uint8_t px[24];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            px[j] = (uint8_t)(w0 >> (8 * j));
            px[8 + j] = (uint8_t)(w1 >> (8 * j));
            px[16 + j] = (uint8_t)(w2 >> (8 * j));
        }
        bf16 *row = out + ((int64_t)b * g * g + p) * (3 * pp) + py * patch + xg * 8;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = to_bf16(((float)px[3 * j + c] / 255.0f - mean[c]) / stdv[c]);
            *reinterpret_cast<bf16x8 *>(row + c * pp) = o;
        }
    }
}

__global__ void text_embed(const int *__restrict__ ids, const float *__restrict__ tok, const float *__restrict__ pos,
                           float *__restrict__ hidden, int B, int L, int D) {
    const int64_t total = (int64_t)B * L * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const int64_t r = i / D;
        const int tk = (int)(r % L);
        hidden[i] = tok[(int64_t)ids[r] * D + d] + pos[tk * D + d];
    }
}

// out[b, :] = x[b, :] @ Wp^T  (Wp [P, D] bf16), optional L2 normalisation; one workgroup per row,
// one wave per output column (16-byte bf16 loads, wave reduction)
__global__ __launch_bounds__(256) void project_rows(const float *__restrict__ x, const bf16 *__restrict__ Wp, int D,
                                                    int P, int normalize, float *__restrict__ out) {
    __shared__ float sx[1024];
    __shared__ float so[1024];
    __shared__ float red[4];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int c = t; c < D; c += 256) sx[c] = x[(int64_t)b * D + c];
    __syncthreads();
    for (int p = wave; p < P; p += 4) {
        const bf16x8 *w8 = reinterpret_cast<const bf16x8 *>(Wp + (int64_t)p * D);
        float a = 0.f;
        for (int c = lane; c < D / 8; c += 64) {
            const bf16x8 wv = w8[c];
#pragma unroll
            for (int j = 0; j < 8; ++j) a = fmaf(sx[8 * c + j], (float)wv[j], a);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64);
        if (lane == 0) so[p] = a;
    }
    __syncthreads();
    float ss = 0.f;
    for (int p = t; p < P; p += 256) ss += so[p] * so[p];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
    const float inv = normalize ? 1.f / fmaxf(nrm, 1e-12f) : 1.f;  // F.normalize eps
    for (int p = t; p < P; p += 256) out[(int64_t)b * P + p] = so[p] * inv;
}

// in-place L2 normalisation of [B, P] rows (F.normalize, eps 1e-12), one wave per row
__global__ __launch_bounds__(256) void l2norm_rows(float *__restrict__ x, int B, int P) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= B) return;
    float *row = x + (int64_t)r * P;
    float ss = 0.f;
    for (int c = lane; c < P; c += 64) ss += row[c] * row[c];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    for (int c = lane; c < P; c += 64) row[c] *= inv;
}

__global__ void f32_to_bf16(const float *__restrict__ in, bf16 *__restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = to_bf16(in[i]);
}

// pooled position of every sequence -> row index into [B*L].  transformers' CLIPTextTransformer.forward:
// a config with the legacy eos_token_id == 2 (what the published openai/clip-vit-* checkpoints carry) pools at
// argmax(ids) -- the end-of-text token has the highest id --, any other value at the first id equal to it.
// (the host entry point has already checked that such a position exists)
__global__ void eos_rows(const int *__restrict__ ids, int B, int L, int eos, int *__restrict__ rows) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int pos = 0;
    if (eos == 2) {
        int best = ids[b * L];
        for (int j = 1; j < L; ++j)
            if (ids[b * L + j] > best) {
                best = ids[b * L + j];
                pos = j;
            }
    } else {
        for (int j = 0; j < L; ++j)
            if (ids[b * L + j] == eos) {
                pos = j;
                break;
            }
    }
    rows[b] = b * L + pos;
}

// the pooled rows of a bf16 residual stream, widened into the f32 rows the final LayerNorm reads
__global__ void widen_rows(const bf16 *__restrict__ x, const int *__restrict__ rows, int B, int D, float *__restrict__ out) {
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const int64_t o = (int64_t)rows[b] * D;
        for (int d = threadIdx.x; d < D; d += blockDim.x) out[o + d] = (float)x[o + d];
    }
}

// The pooled (first) row of every image out of the tile path's buffers, compacted: f32 residual row, its bf16 copy and
// its partial LayerNorm statistics (np pairs a row) -- the last layer's MLP runs on these rows only (run_tower)
__global__ __launch_bounds__(256) void gather_pooled_rows(const float *__restrict__ x, const bf16 *__restrict__ xb,
                                                          const float *__restrict__ st, int B, int S, int D, int np,
                                                          float *__restrict__ x_out, bf16 *__restrict__ xb_out,
                                                          float *__restrict__ st_out) {
    const int b = blockIdx.x;
    if (b >= B) return;
    const int64_t src = (int64_t)b * S;
    for (int k = threadIdx.x; k < D; k += 256) {
        const bf16 hb = xb[src * D + k];
        x_out[(int64_t)b * D + k] = x ? x[src * D + k] : (float)hb;  // bf16 rows: the stream row itself is the residual
        xb_out[(int64_t)b * D + k] = hb;
    }
    if ((int)threadIdx.x < 2 * np) st_out[(int64_t)b * 2 * np + threadIdx.x] = st[src * 2 * np + threadIdx.x];
}

// The last layer's attention for the pooled row of an image only (round 5).  The embedding reads row 0 of an image
// behind the last layer and nothing else (model.py:55-57, transformers' pooling), and attention mixes rows only
// through the keys and values: row 0's query against the image's S keys is all the last attention has to compute.
// One wave per (image, head) runs attention_rows64's query tile 0 -- the same fragments through the same MFMAs, the
// same softmax -- and keeps row 0 of the tile: the bf16 values attn_outproj_image / attention_rows64 leave for that row,
// bit for bit (a first version with plain f32 dot products differed from the full layer by 1e-4 on unit vectors: one
// P or output element rounding the other way to bf16 is 0.4 % of it).  Reads K and V once: 2 S D bf16 per image.
__global__ __launch_bounds__(64) void attn_pooled_rows(const bf16 *__restrict__ qkv, bf16 *__restrict__ out, int S, int D, int H,
                                                       float scale) {
    __shared__ __attribute__((aligned(16))) bf16 sKV[64 * 64];  // K rows (GEMM swizzle), later V rows (v_off)
    __shared__ __attribute__((aligned(16))) bf16 sQP[16 * 64];  // the query tile's Q rows, then P
    auto g_off = [](int row, int c16) { return row * 64 + ((c16 ^ ((row >> 1) & 7)) << 3); };
    auto v_off = [](int key, int c32) { return key * 64 + ((c32 ^ (((key >> 1) & 1) | (((key >> 3) & 1) << 1))) << 4); };
    const int lane = threadIdx.x;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int fr = lane & 15, fq = lane >> 4;
    const bf16 *base = qkv + (int64_t)b * S * 3 * D + h * 64;
    bf16x8 kreg[8], vreg[8], qreg[2];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int ch = lane + c * 64, row = ch >> 3, d0 = (ch & 7) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) kreg[c][j] = vreg[c][j] = (bf16)0.f;
        if (row < S) {
            const bf16 *r = base + (int64_t)row * 3 * D + d0;
            kreg[c] = *reinterpret_cast<const bf16x8 *>(r + D);
            vreg[c] = *reinterpret_cast<const bf16x8 *>(r + 2 * D);
        }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int ch = lane + c * 64, row = ch >> 3, d0 = (ch & 7) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) qreg[c][j] = (bf16)0.f;
        if (row < S) qreg[c] = *reinterpret_cast<const bf16x8 *>(base + (int64_t)row * 3 * D + d0);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int ch = lane + c * 64;
        *reinterpret_cast<bf16x8 *>(&sKV[g_off(ch >> 3, ch & 7)]) = kreg[c];
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int ch = lane + c * 64;
        *reinterpret_cast<bf16x8 *>(&sQP[g_off(ch >> 3, ch & 7)]) = qreg[c];
    }
    __syncthreads();
    bf16x8 kf[4][2], qf[2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) kf[j][ks] = *reinterpret_cast<const bf16x8 *>(&sKV[g_off(j * 16 + fr, ks * 4 + fq)]);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const bf16x8 *>(&sQP[g_off(fr, ks * 4 + fq)]);
    f32x4 sc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) sc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[j][ks], qf[ks], sc[j], 0, 0, 0);
    }
    float mx = -INFINITY;  // this lane holds scores[query fr][key = 16 j + 4 fq + r]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = j * 16 + fq * 4 + r;
            float v = sc[j][r] * scale;
            if (key >= S) v = -INFINITY;
            sc[j][r] = v;
            mx = fmaxf(mx, v);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = __expf(sc[j][r] - mx);
            sc[j][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    __syncthreads();  // the Q and K fragments are in registers: P takes Q's bytes, V takes K's
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bf16x4 pv;
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[r] = to_bf16(sc[j][r] * inv);
        *reinterpret_cast<bf16x4 *>(&sQP[g_off(fr, j * 2 + (fq >> 1)) + (fq & 1) * 4]) = pv;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int ch = lane + c * 64, key = ch >> 3, c16 = ch & 7;
        *reinterpret_cast<bf16x8 *>(&sKV[v_off(key, c16 >> 1) + (c16 & 1) * 8]) = vreg[c];
    }
    __syncthreads();
    f32x4 o[4];  // out^T tile = V^T P^T : o[dt][r] = out[query fr][d = 16 dt + 4 fq + r]
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 pa = *reinterpret_cast<const bf16x8 *>(&sQP[g_off(fr, ks * 4 + fq)]);
        const int kq = ks * 32 + 8 * fq + (fr >> 2), dp = (fr & 3) * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            typedef __attribute__((ext_vector_type(4))) short s16x4;
            typedef __attribute__((address_space(3))) s16x4 *lds_s16x4;
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(&sKV[v_off(kq, dt) + dp]));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(&sKV[v_off(kq + 4, dt) + dp]));
            const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, both), pa, o[dt], 0, 0, 0);
        }
    }
    if (fr == 0) {  // query row 0 of the image
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            bf16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = to_bf16(o[dt][r]);
            *reinterpret_cast<bf16x4 *>(out + (int64_t)b * D + h * 64 + dt * 16 + fq * 4) = ov;
        }
    }
}

__global__ void cls_rows(int B, int T, int *__restrict__ rows) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) rows[b] = b * T;
}

}  // namespace
}  // namespace ssw

using namespace ssw;

namespace {

struct Layer {
    float *ln1w, *ln1b, *ln2w, *ln2b, *bqkv, *bo, *b1, *b2;
    bf16 *wqkv, *wo, *w1, *w2;
    // LayerNorm folded into the tile path's QKV / fc1 products (GemmLn, ssw_common.h): W' = gamma (.) W in bf16,
    // c1_n = sum_k W'_nk, c2_n = sum_k beta_k W_nk + b_n
    bf16 *wqkv_ln = nullptr, *w1_ln = nullptr;
    bf16 *wo_pk = nullptr;  // wo in the fragment order of attn_out.hip (ViT-B/32 shapes only)
    float *c1qkv = nullptr, *c2qkv = nullptr, *c1fc1 = nullptr, *c2fc1 = nullptr;
};

struct Tower {
    int D = 0, L = 0, H = 0, M = 0, T = 0;
    std::vector<Layer> layers;
    float *lnf_w = nullptr, *lnf_b = nullptr;
    bf16 *proj = nullptr;
};

struct Header {
    char magic[8];
    int32_t v_hidden, v_layers, v_heads, v_mlp, image, patch;
    int32_t t_hidden, t_layers, t_heads, t_mlp, t_maxpos, vocab, eos;
    int32_t proj;
    float ln_eps;
    int32_t reserved;
};

}  // namespace

struct ssw_clip {
    int device = 0;
    Header hdr;
    hipStream_t stream = nullptr;
    std::vector<void *> allocs;
    Tower vis, txt;
    // vision extras
    bf16 *patch_w = nullptr;
    float *cls = nullptr, *vpos = nullptr, *pre_w = nullptr, *pre_b = nullptr;
    // text extras
    float *tok = nullptr, *tpos = nullptr;
    // workspace (sized for `cap_rows` token rows)
    int64_t cap_rows = 0, cap_mlp = 0, cap_batch = 0;
    float *hidden = nullptr, *hidden2 = nullptr, *pooled = nullptr, *patch_out = nullptr;
    bf16 *xn = nullptr, *qkv = nullptr, *att = nullptr, *h1 = nullptr, *patches = nullptr;
    bool stream_in_xn = false;  // run_tower left the residual stream in xn (bf16), not in hidden
    bool pooled_compact = false;  // run_tower left only the pooled rows of the last layer, as rows 0 .. B-1 of hidden
    float *stats_a = nullptr, *stats_b = nullptr;  // [rows][D / 128][2] partial LayerNorm statistics of hidden / hidden2
    float *pixels = nullptr, *out = nullptr;
    float *splitk = nullptr;  // [8][min(rows, SPLITK_MAX_ROWS)][D] f32: partial products of the text tower's split-K producers
    int *ids = nullptr, *rows = nullptr;
    int64_t rows_key = -1;  // what `rows` holds: (B << 32 | stride) of the image tower's pooled-row list, -1 = something else
    // ssw_clip_set_option: bit 0 = bf16 residual rows in the image tower's tile path, bit 1 = bf16 rows in the text tower's,
    // bit 2 = the tile path's attention with its K / Q fragments straight from memory (attention_mfma) for S <= 64 too,
    // bit 3 = attention and out-projection as two launches (round 3's layer) where attn_out.hip's one launch applies
    int flags = 0;
    // The lab build's tap (ssw_clip_debug_tap).  Declared in BOTH builds so that a handle has one layout whichever library
    // created it (a product-made handle handed to the lab build's tap wrote past the allocation); the product
    // never sets tap_layer and frees tap_buf, which only the lab build allocates, in its destroy as well.
    int tap_layer = -1, tap_tower = 0;  // the residual rows behind this layer of this tower (0 image, 1 text) ...
    float *tap_buf = nullptr;           // ... as f32 [tap_rows][tap_dim]
    int64_t tap_rows = 0, tap_dim = 0, tap_cap = 0;
};

namespace {

ssw_status dmalloc(ssw_clip *c, void **p, size_t bytes) {
    SSW_HIP_TRY(hipMalloc(p, bytes + 64));
    c->allocs.push_back(*p);
    return SSW_OK;
}

// upload a run of f32 values from the blob, as f32 or converted to bf16 on the device
ssw_status upload_f32(ssw_clip *c, const float *&cur, const float *end, size_t n, float **dst) {
    if (cur + n > end) {
        set_error("clip: weight blob truncated");
        return SSW_ERR_INVALID;
    }
    SSW_TRY(dmalloc(c, (void **)dst, n * sizeof(float)));
    SSW_HIP_TRY(hipMemcpy(*dst, cur, n * sizeof(float), hipMemcpyHostToDevice));
    cur += n;
    return SSW_OK;
}

ssw_status upload_bf16(ssw_clip *c, const float *&cur, const float *end, size_t n, bf16 **dst, float *scratch_dev) {
    if (cur + n > end) {
        set_error("clip: weight blob truncated");
        return SSW_ERR_INVALID;
    }
    SSW_TRY(dmalloc(c, (void **)dst, n * sizeof(bf16)));
    SSW_HIP_TRY(hipMemcpy(scratch_dev, cur, n * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(f32_to_bf16, dim3(1024), dim3(256), 0, 0, scratch_dev, *dst, (int64_t)n);
    SSW_HIP_TRY(hipDeviceSynchronize());
    cur += n;
    return SSW_OK;
}

// f32 -> bf16 -> f32 the way the device's conversion rounds (nearest even); the blob holds finite weights
inline float bf16_round_host(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    memcpy(&x, &u, 4);
    return x;
}

// W [N, D] (f32, host), gamma / beta [D], b [N]  ->  W' = gamma (.) W (bf16 on the device), c1, c2 (f32 on the device)
ssw_status fold_layernorm(ssw_clip *c, const float *W, const float *gamma, const float *beta, const float *b, size_t N,
                          size_t D, bf16 **w_ln, float **c1, float **c2, float *scratch) {
    std::vector<float> wf(N * D), v1(N), v2(N);
    for (size_t n = 0; n < N; ++n) {
        double s1 = 0.0, s2 = 0.0;
        for (size_t k = 0; k < D; ++k) {
            const float wp = W[n * D + k] * gamma[k];
            wf[n * D + k] = wp;
            s1 += (double)bf16_round_host(wp);  // what the matrix cores will multiply by
            s2 += (double)beta[k] * (double)W[n * D + k];
        }
        v1[n] = (float)s1;
        v2[n] = (float)(s2 + (double)b[n]);
    }
    const float *p = wf.data();
    SSW_TRY(upload_bf16(c, p, p + wf.size(), wf.size(), w_ln, scratch));
    p = v1.data();
    SSW_TRY(upload_f32(c, p, p + N, N, c1));
    p = v2.data();
    SSW_TRY(upload_f32(c, p, p + N, N, c2));
    return SSW_OK;
}

// q, k, v weights/biases are consecutive in the blob; fuse them into one [3D, D] operand
ssw_status load_tower(ssw_clip *c, Tower &tw, const float *&cur, const float *end, float *scratch) {
    const size_t D = tw.D, M = tw.M;
    tw.layers.resize(tw.L);
    for (int l = 0; l < tw.L; ++l) {
        Layer &ly = tw.layers[l];
        const float *ln1w_h = cur, *ln1b_h = cur + D;
        SSW_TRY(upload_f32(c, cur, end, D, &ly.ln1w));
        SSW_TRY(upload_f32(c, cur, end, D, &ly.ln1b));
        // blob order: q.w q.b k.w k.b v.w v.b -> gather into wqkv / bqkv on the host side
        if (cur + 3 * (D * D + D) > end) {
            set_error("clip: weight blob truncated");
            return SSW_ERR_INVALID;
        }
        std::vector<float> w(3 * D * D), b(3 * D);
        for (int p = 0; p < 3; ++p) {
            memcpy(w.data() + p * D * D, cur, D * D * sizeof(float));
            cur += D * D;
            memcpy(b.data() + p * D, cur, D * sizeof(float));
            cur += D;
        }
        const float *wc = w.data(), *bc = b.data();
        SSW_TRY(upload_bf16(c, wc, wc + w.size(), w.size(), &ly.wqkv, scratch));
        SSW_TRY(upload_f32(c, bc, bc + b.size(), b.size(), &ly.bqkv));
        SSW_TRY(fold_layernorm(c, w.data(), ln1w_h, ln1b_h, b.data(), 3 * D, D, &ly.wqkv_ln, &ly.c1qkv, &ly.c2qkv, scratch));
        SSW_TRY(upload_bf16(c, cur, end, D * D, &ly.wo, scratch));
        if (attn_outproj_supports(1, (int)D, tw.H)) {
            SSW_TRY(dmalloc(c, (void **)&ly.wo_pk, D * D * sizeof(bf16)));
            SSW_TRY(pack_attn_outproj_weight(0, ly.wo, ly.wo_pk));
            SSW_HIP_TRY(hipDeviceSynchronize());
        }
        SSW_TRY(upload_f32(c, cur, end, D, &ly.bo));
        const float *ln2w_h = cur, *ln2b_h = cur + D;
        SSW_TRY(upload_f32(c, cur, end, D, &ly.ln2w));
        SSW_TRY(upload_f32(c, cur, end, D, &ly.ln2b));
        if (cur + M * D + M > end) {
            set_error("clip: weight blob truncated");
            return SSW_ERR_INVALID;
        }
        SSW_TRY(fold_layernorm(c, cur, ln2w_h, ln2b_h, cur + M * D, M, D, &ly.w1_ln, &ly.c1fc1, &ly.c2fc1, scratch));
        SSW_TRY(upload_bf16(c, cur, end, M * D, &ly.w1, scratch));
        SSW_TRY(upload_f32(c, cur, end, M, &ly.b1));
        SSW_TRY(upload_bf16(c, cur, end, D * M, &ly.w2, scratch));
        SSW_TRY(upload_f32(c, cur, end, D, &ly.b2));
    }
    SSW_TRY(upload_f32(c, cur, end, D, &tw.lnf_w));
    SSW_TRY(upload_f32(c, cur, end, D, &tw.lnf_b));
    return SSW_OK;
}

template <int EPI>
ssw_status gemm(hipStream_t s, const bf16 *A, const bf16 *W, const float *bias, const float *res, void *C, int M,
                int N, int K) {
    return launch_gemm_bf16_nt(EPI, s, A, W, bias, res, C, M, N, K);
}

constexpr int64_t SPLITK_MAX_ROWS = 4096;  // rows up to which a producer product may be split over K (its partial-product buffer)

ssw_status reserve(ssw_clip *c, int64_t batch) {
    if (batch <= c->cap_batch) return SSW_OK;
    SSW_HIP_TRY(hipStreamSynchronize(c->stream));
    for (void *p : {(void *)c->hidden, (void *)c->hidden2, (void *)c->pooled, (void *)c->patch_out, (void *)c->xn,
                    (void *)c->qkv, (void *)c->att, (void *)c->h1, (void *)c->patches, (void *)c->pixels,
                    (void *)c->out, (void *)c->ids, (void *)c->rows, (void *)c->stats_a, (void *)c->stats_b, (void *)c->splitk})
        (void)hipFree(p);
    const Header &h = c->hdr;
    const int64_t Tv = (int64_t)(h.image / h.patch) * (h.image / h.patch) + 1;
    const int64_t rows = batch * std::max<int64_t>(Tv, h.t_maxpos);
    const int64_t D = std::max(h.v_hidden, h.t_hidden), Mm = std::max(h.v_mlp, h.t_mlp);
    const int64_t pcols = 3LL * h.patch * h.patch;
    SSW_HIP_TRY(hipMalloc((void **)&c->hidden, rows * D * sizeof(float)));
    SSW_HIP_TRY(hipMalloc((void **)&c->hidden2, rows * D * sizeof(float)));
    SSW_HIP_TRY(hipMalloc((void **)&c->pooled, batch * D * sizeof(float)));
    SSW_HIP_TRY(hipMalloc((void **)&c->patch_out, batch * (Tv - 1) * h.v_hidden * sizeof(float)));
    SSW_HIP_TRY(hipMalloc((void **)&c->xn, rows * D * sizeof(bf16)));
    SSW_HIP_TRY(hipMalloc((void **)&c->qkv, rows * 3 * D * sizeof(bf16)));
    SSW_HIP_TRY(hipMalloc((void **)&c->att, rows * D * sizeof(bf16)));
    SSW_HIP_TRY(hipMalloc((void **)&c->h1, rows * Mm * sizeof(bf16)));
    SSW_HIP_TRY(hipMalloc((void **)&c->patches, batch * (Tv - 1) * pcols * sizeof(bf16)));
    SSW_HIP_TRY(hipMalloc((void **)&c->pixels, batch * 3LL * h.image * h.image * sizeof(float)));
    SSW_HIP_TRY(hipMalloc((void **)&c->out, batch * h.proj * sizeof(float)));
    SSW_HIP_TRY(hipMalloc((void **)&c->ids, batch * h.t_maxpos * sizeof(int)));
    SSW_HIP_TRY(hipMalloc((void **)&c->rows, batch * sizeof(int)));
    c->rows_key = -1;
    SSW_HIP_TRY(hipMalloc((void **)&c->stats_a, rows * (D / 128 + 1) * 2 * sizeof(float)));
    SSW_HIP_TRY(hipMalloc((void **)&c->stats_b, rows * (D / 128 + 1) * 2 * sizeof(float)));
    SSW_HIP_TRY(hipMalloc((void **)&c->splitk, 8 * std::min<int64_t>(rows, SPLITK_MAX_ROWS) * D * sizeof(float)));
    c->cap_batch = batch;
    return SSW_OK;
}

// the transformer stack on `R = B*S` token rows held in c->hidden (f32)
template <int EPI, bool LN, int NW>
void launch_skinny_nw(hipStream_t s, const void *X, const int *row_index, const float *lnw, const float *lnb, float eps,
                      const bf16 *W, const float *bias, const float *residual, void *out, int R, int N) {
    if (R <= 16)
        hipLaunchKernelGGL((skinny_linear<EPI, LN, 1, NW>), dim3(N / 16), dim3(64 * NW), 0, s, X, row_index, lnw, lnb,
                           eps, W, bias, residual, out, R, N);
    else
        hipLaunchKernelGGL((skinny_linear<EPI, LN, 2, NW>), dim3(N / 16), dim3(64 * NW), 0, s, X, row_index, lnw, lnb,
                           eps, W, bias, residual, out, R, N);
}
template <int EPI, bool LN>
void launch_skinny(hipStream_t s, const void *X, const int *row_index, const float *lnw, const float *lnb, float eps,
                   const bf16 *W, const float *bias, const float *residual, void *out, int R, int N, int K) {
    // one wave per 128 of K (skinny_rows() admits nothing else); the fused layer norm is over D = 512
    if constexpr (LN) {
        launch_skinny_nw<EPI, LN, 4>(s, X, row_index, lnw, lnb, eps, W, bias, residual, out, R, N);
    } else {
        if (K == 512) launch_skinny_nw<EPI, LN, 4>(s, X, row_index, lnw, lnb, eps, W, bias, residual, out, R, N);
        else if (K == 1024) launch_skinny_nw<EPI, LN, 8>(s, X, row_index, lnw, lnb, eps, W, bias, residual, out, R, N);
        else launch_skinny_nw<EPI, LN, 16>(s, X, row_index, lnw, lnb, eps, W, bias, residual, out, R, N);
    }
}

// a handful of rows (one short text query): four launches a layer, every one over N/16 workgroups
bool skinny_rows(int R, int D, int M) {
    static const bool off = getenv("SSW_CLIP_NO_SKINNY") != nullptr;
    return !off && R <= SK_MAX_ROWS && D == 512 && (M == 512 || M == 1024 || M == 2048);  // (head dim 64: 8 heads)
}

// a handle's option word (ssw_clip::flags) starts from the environment
int clip_flags_from_env() {
    return (getenv("SSW_CLIP_BF16_STREAM") ? 1 : 0) | (getenv("SSW_CLIP_UNFUSED_ATTN") ? 8 : 0) | (getenv("SSW_CLIP_FULL_LAST_LAYER") ? 16 : 0);
}
bool unfused_ln_forced() {
    static const bool v = getenv("SSW_CLIP_UNFUSED_LN") != nullptr;  // A/B: the round-2 seven-launch layer
    return v;
}
// does this tower's tile path keep its residual rows in bf16?
bool bf16_rows(const ssw_clip *c, const Tower &tw, int causal) {
    return !unfused_ln_forced() && tw.D % 256 == 0 && (causal ? (c->flags & 2) != 0 : (c->flags & 1) != 0);
}

#ifdef SSW_DEBUG_HOOKS
__global__ void k_tap_widen(const bf16 *__restrict__ x, int64_t n, float *__restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (float)x[i];
}
// lab build: keep the residual rows behind layer l (f32 copy) when the handle asks for them
ssw_status tap_layer_rows(ssw_clip *c, const Tower &tw, int l, int causal, int64_t R, const float *rows_f32, const bf16 *rows_bf16) {
    if (c->tap_layer != l || c->tap_tower != (causal ? 1 : 0)) return SSW_OK;
    const int64_t n = R * tw.D;
    if (n > c->tap_cap) {
        if (c->tap_buf) (void)hipFree(c->tap_buf);
        c->tap_buf = nullptr;
        SSW_HIP_TRY(hipMalloc((void **)&c->tap_buf, (size_t)n * sizeof(float)));
        c->tap_cap = n;
    }
    if (rows_bf16) hipLaunchKernelGGL(k_tap_widen, dim3(1024), dim3(256), 0, c->stream, rows_bf16, n, c->tap_buf);
    else SSW_HIP_TRY(hipMemcpyAsync(c->tap_buf, rows_f32, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    c->tap_rows = R;
    c->tap_dim = tw.D;
    return SSW_OK;
}
#define SSW_TAP(l, f32p, bf16p) SSW_TRY(tap_layer_rows(c, tw, (l), causal, R, (f32p), (bf16p)))
#else
#define SSW_TAP(l, f32p, bf16p)
#endif

ssw_status run_tower(ssw_clip *c, const Tower &tw, int B, int S, int causal) {
    hipStream_t s = c->stream;
    const int R = B * S, D = tw.D, M = tw.M;
    const float eps = c->hdr.ln_eps;
    float *h = c->hidden, *h2 = c->hidden2;
    const bool skinny = skinny_rows(R, D, M) && tw.H * 64 == D;  // skinny_attn_out: 8 heads of 64
    for (int l = 0; l < tw.L && skinny; ++l) {
        const Layer &ly = tw.layers[l];
        launch_skinny<SK_BF16_BIAS, true>(s, h, nullptr, ly.ln1w, ly.ln1b, eps, ly.wqkv, ly.bqkv, nullptr, c->qkv, R,
                                          3 * D, D);
        const float att_scale = 1.0f / sqrtf((float)(D / tw.H));
        static const bool two_launches = getenv("SSW_CLIP_NO_FUSED_ATTN") != nullptr;  // A/B: attention as its own launch
        if (two_launches) {
            hipLaunchKernelGGL(attention_mfma<4>, dim3(B * tw.H), dim3(256), 0, s, c->qkv, c->att, S, D, tw.H, att_scale,
                               causal);
            launch_skinny<SK_F32_BIAS_RESIDUAL, false>(s, c->att, nullptr, nullptr, nullptr, eps, ly.wo, ly.bo, h, h2, R, D, D);
        } else if (R <= 16) {
            hipLaunchKernelGGL(skinny_attn_out<1>, dim3(D / 16), dim3(256), 0, s, c->qkv, ly.wo, ly.bo, h, h2, R, S,
                               att_scale, causal);
        } else {
            hipLaunchKernelGGL(skinny_attn_out<2>, dim3(D / 16), dim3(256), 0, s, c->qkv, ly.wo, ly.bo, h, h2, R, S,
                               att_scale, causal);
        }
        launch_skinny<SK_BF16_BIAS_GELU, true>(s, h2, nullptr, ly.ln2w, ly.ln2b, eps, ly.w1, ly.b1, nullptr, c->h1, R, M, D);
        launch_skinny<SK_F32_BIAS_RESIDUAL, false>(s, c->h1, nullptr, nullptr, nullptr, eps, ly.w2, ly.b2, h2, h, R, D, M);
        SSW_TAP(l, h, nullptr);
    }
    // ---- tile path: five launches a layer.  Both LayerNorms are folded into the products that consume them (GemmLn,
    // ssw_common.h): the out-projection / fc2 epilogues leave, next to the f32 residual row, its bf16 copy (c->xn) and
    // its partial statistics; QKV / fc1 multiply that copy by gamma (.) W and apply mean / rstd in their epilogue.
    // c->xn and stats_in arrive filled by whoever produced c->hidden (stack_input_rows).
    const bool unfused_ln = unfused_ln_forced();
    // the residual stream of the tile path in bf16: the producers' bf16 copy IS the stream (read, added to, written back
    // in place by EPI_BF16_STREAM_STATS), the f32 row and its 30 MB + 30 MB a product at B = 200 are gone
    // Measured against transformers' f32 towers (tools/clip_stream_error.py, 26 images): min cosine 0.999994 -> 0.999952,
    // max |delta| of a unit vector's component 5.0e-4 -> 1.5e-3 (bar: 0.999 / 5e-3); B = 200 forward 2.91 -> 2.75 ms.
    // On for the image tower (the batch path); the text tower -- the query side -- keeps its f32 rows.
    const bool bf16_stream = bf16_rows(c, tw, causal);
    c->stream_in_xn = false;
    c->pooled_compact = false;
    const int np = D / 128;
    float *st_h = c->stats_a, *st_h2 = c->stats_b;
    // Round 6 experiment (SSW_XCD_AFFINITY=1, A/B only): the QKV product deals CONTIGUOUS runs of row tiles to the XCDs and the
    // fused attention launch puts an image's workgroup on the XCD that produced its qkv rows (what it measured)
    static const bool xcd_affinity_env = getenv("SSW_XCD_AFFINITY") != nullptr;
    const bool xcd_affinity = xcd_affinity_env && !causal && !unfused_ln && D % 256 == 0 && (c->flags & 8) == 0 &&
                              attn_outproj_supports(S, D, tw.H);
    for (int l = 0; l < tw.L && !skinny; ++l) {
        const Layer &ly = tw.layers[l];
        GemmLn cons, prod;
        cons.np_in = np;
        cons.inv_dim = 1.0f / (float)D;
        cons.eps = eps;
        prod.xcopy = c->xn;
        if (unfused_ln || D % 256 != 0) {
            hipLaunchKernelGGL(layernorm_rows<bf16>, dim3((R + 3) / 4), dim3(256), 0, s, h, (const int *)nullptr, R, D,
                               ly.ln1w, ly.ln1b, eps, c->xn);
            SSW_TRY(gemm<EPI_BF16_BIAS>(s, c->xn, ly.wqkv, ly.bqkv, nullptr, c->qkv, R, 3 * D, D));
        } else {
            cons.stats_in = st_h;
            cons.c1 = ly.c1qkv;
            cons.xcd_contig = xcd_affinity ? 1 : 0;
            SSW_TRY(launch_gemm_bf16_ln(EPI_BF16_LN, s, c->xn, ly.wqkv_ln, ly.c2qkv, nullptr, c->qkv, R, 3 * D, D, cons));
            cons.xcd_contig = 0;
        }
        const int n_heads = B * tw.H;
        const float att_scale = 1.0f / sqrtf((float)(D / tw.H));
        // round 4: attention + out-projection + residual + statistics of the image tower in one launch, a workgroup per
        // image (attn_out.hip); the fc1 product then reads two partial pairs a row instead of D / 128
        const bool fused_attn = !unfused_ln && D % 256 == 0 && !causal && (c->flags & 8) == 0 &&
                                attn_outproj_supports(S, D, tw.H) && ly.wo_pk != nullptr;
        if (fused_attn) {
            // The last layer on the pooled rows only: the final LayerNorm and the projection read the first row of
            // every image and nothing else.  A row's MLP does not look at other rows (round 4: fc1 / fc2 on B rows, 113 us
            // of a B = 200 forward), and attention looks at other rows only through their keys and values (round 5): the
            // last attention runs row 0's query against the image's keys (attn_pooled_rows), the out-projection on those B
            // rows (split over K like the last fc2).  SSW_CLIP_OPT_FULL_LAST_LAYER runs every row as the reference's model does.
            const bool full_last_layer = (c->flags & 16) != 0;
            bool pooled_only = l == tw.L - 1 && !full_last_layer && S >= 8 && M % 512 == 0;  // (S: the partial products borrow the qkv buffer)
#ifdef SSW_DEBUG_HOOKS
            if (c->tap_layer == l && c->tap_tower == 0) pooled_only = false;  // the tap wants every row of this layer
#endif
            static const bool pooled_attn_off = getenv("SSW_CLIP_POOLED_MLP_ONLY") != nullptr;  // A/B: round 4's form
            cons.c1 = ly.c1fc1;
            cons.np_in = 2;
            if (pooled_only && !pooled_attn_off && !bf16_stream && D / tw.H == 64 && S <= 64) {
                // Two launches: row 0's attention, then the out-projection of those B rows as the tile GEMM's producer
                // form (bias-started accumulators, k ascending, + the residual row b S of the stack: GemmLn::res_ld) -- the
                // f32 row and the bf16 copy attn_outproj_image leaves for that row, bit for bit; only the partial
                // statistics are cut differently (six 128-column pairs instead of two halves).  Measured first and
                // dropped: the product split over K (another summation order moves the row by 1e-7, now and then its
                // bf16 copy by a whole ulp, and the vectors by 7e-5 where this form moves them by 2e-5), and separate
                // gather / finish passes (every launch of this tail is ~5 us on a slow box, whatever it does).
                bf16 *att_p = c->h1;                                    // [B][D]; fc1 writes h1 only after the product read it
                float *partials = reinterpret_cast<float *>(c->qkv);     // (fc2's partial products; free once the keys and values are read)
                hipLaunchKernelGGL(attn_pooled_rows, dim3(B * tw.H), dim3(64), 0, s, c->qkv, att_p, S, D, tw.H, att_scale);
                prod.xcopy = c->att;
                prod.stats_out = st_h;
                prod.res_ld = (int64_t)S * D;
                SSW_TRY(launch_gemm_bf16_ln(EPI_F32_BIAS_RESIDUAL_STATS, s, att_p, ly.wo, ly.bo, h, h2, B, D, D, prod));
                cons.np_in = np;
                cons.stats_in = st_h;
                SSW_TRY(launch_gemm_bf16_ln(EPI_BF16_LN_GELU, s, c->att, ly.w1_ln, ly.c2fc1, nullptr, c->h1, B, M, D, cons));
                SSW_TRY(launch_gemm_splitk_f32(s, c->h1, ly.w2, ly.b2, h2, h, partials, B, D, M, 8));
                c->pooled_compact = true;
                c->stream_in_xn = false;  // the pooled rows are f32 rows of hidden whatever the stream's precision
                continue;
            }
            if (bf16_stream) c->stream_in_xn = true;
            SSW_TRY(launch_attn_outproj(s, c->qkv, ly.wo_pk, ly.bo, c->xn, bf16_stream ? nullptr : h, h2, st_h2, B, S, D, tw.H,
                                        att_scale, xcd_affinity ? (R + 127) / 128 : 0));
            cons.stats_in = st_h2;
            // round 4's form (SSW_CLIP_POOLED_MLP_ONLY=1): attention and out-projection for every row, then the rows are
            // compacted (residual row, bf16 copy, statistics), fc1 runs on them as it is, fc2 -- 12 tiles with 48 K-steps
            // each -- split over K
            if (pooled_only) {
                float *res_p = h + (int64_t)B * D;  // hidden is free behind the out-projection: rows B .. 2B-1 take the residual rows
                hipLaunchKernelGGL(gather_pooled_rows, dim3(B), dim3(256), 0, s, bf16_stream ? (const float *)nullptr : h2, c->xn, st_h2,
                                   B, S, D, 2, res_p, c->att, st_h);
                cons.stats_in = st_h;
                SSW_TRY(launch_gemm_bf16_ln(EPI_BF16_LN_GELU, s, c->att, ly.w1_ln, ly.c2fc1, nullptr, c->h1, B, M, D, cons));
                SSW_TRY(launch_gemm_splitk_f32(s, c->h1, ly.w2, ly.b2, res_p, h, reinterpret_cast<float *>(c->qkv), B, D, M, 8));
                c->pooled_compact = true;
                c->stream_in_xn = false;  // the pooled rows are f32 rows of hidden whatever the stream's precision
                continue;
            }
            SSW_TRY(launch_gemm_bf16_ln(EPI_BF16_LN_GELU, s, c->xn, ly.w1_ln, ly.c2fc1, nullptr, c->h1, R, M, D, cons));
            prod.stats_out = st_h;
            if (bf16_stream)
                SSW_TRY(launch_gemm_bf16_ln(EPI_BF16_STREAM_STATS, s, c->h1, ly.w2, ly.b2, nullptr, nullptr, R, D, M, prod));
            else
                SSW_TRY(launch_gemm_bf16_ln(EPI_F32_BIAS_RESIDUAL_STATS, s, c->h1, ly.w2, ly.b2, h2, h, R, D, M, prod));
            SSW_TAP(l, bf16_stream ? nullptr : h, bf16_stream ? c->xn : nullptr);
            continue;
        }
        static const bool one_tile_waves = getenv("SSW_CLIP_ATTN_TPW1") != nullptr;  // A/B: the four-wave form
        // (one wave per pair with four tiles, attention_mfma<4, 4>: 21.6 us per layer against 17.1 -- measured, not kept)
        static const bool direct_env = getenv("SSW_CLIP_ATTN_DIRECT") != nullptr;  // A/B: fragments straight from memory
        const bool direct_frags = direct_env || (c->flags & 4) != 0;
        if (S <= 64 && !one_tile_waves && !direct_frags)
            hipLaunchKernelGGL(attention_rows64, dim3(n_heads), dim3(128), 0, s, c->qkv, c->att, S, D, tw.H, att_scale, causal);
        else if (S <= 64 && !one_tile_waves)
            hipLaunchKernelGGL((attention_mfma<4, 2>), dim3(n_heads), dim3(128), 0, s, c->qkv, c->att, S, D, tw.H, att_scale,
                               causal);
        else if (S <= 64)
            hipLaunchKernelGGL(attention_mfma<4>, dim3(n_heads), dim3(256), 0, s, c->qkv, c->att, S, D, tw.H, att_scale,
                               causal);
        else
            hipLaunchKernelGGL(attention_mfma<5>, dim3(n_heads), dim3(320), 0, s, c->qkv, c->att, S, D, tw.H, att_scale,
                               causal);
        if (unfused_ln || D % 256 != 0) {
            SSW_TRY(gemm<EPI_F32_BIAS_RESIDUAL>(s, c->att, ly.wo, ly.bo, h, h2, R, D, D));
            hipLaunchKernelGGL(layernorm_rows<bf16>, dim3((R + 3) / 4), dim3(256), 0, s, h2, (const int *)nullptr, R, D,
                               ly.ln2w, ly.ln2b, eps, c->xn);
            SSW_TRY(gemm<EPI_BF16_BIAS_GELU>(s, c->xn, ly.w1, ly.b1, nullptr, c->h1, R, M, D));
            SSW_TRY(gemm<EPI_F32_BIAS_RESIDUAL>(s, c->h1, ly.w2, ly.b2, h2, h, R, D, M));
        } else if (bf16_stream) {
            c->stream_in_xn = true;
            prod.stats_out = st_h2;
            SSW_TRY(launch_gemm_bf16_ln(EPI_BF16_STREAM_STATS, s, c->att, ly.wo, ly.bo, nullptr, nullptr, R, D, D, prod));
            cons.stats_in = st_h2;
            cons.c1 = ly.c1fc1;
            SSW_TRY(launch_gemm_bf16_ln(EPI_BF16_LN_GELU, s, c->xn, ly.w1_ln, ly.c2fc1, nullptr, c->h1, R, M, D, cons));
            prod.stats_out = st_h;
            SSW_TRY(launch_gemm_bf16_ln(EPI_BF16_STREAM_STATS, s, c->h1, ly.w2, ly.b2, nullptr, nullptr, R, D, M, prod));
        } else {
            // Round 5: the text tower's N = 512 products are 40 tiles at 16 x 77 rows -- a launch as long as ONE workgroup's
            // K loop (fc2: 32 steps, 31 us; 5 % of the MFMA peak on the whole tower).  Split over K in a fixed order
            // (launch_gemm_splitk_stats: partial products, then one pass that adds them in ascending order with bias and
            // residual and leaves the bf16 copy and the statistics) they are 160-320 workgroups of 2-4 steps.  Text tower
            // only: the image tower's vectors must not depend on how many tiles share a call (the split is chosen by the row
            // count), and its products fill the chip at the sizes that matter.
            static const bool no_splitk = getenv("SSW_CLIP_NO_SPLITK") != nullptr;  // A/B
            const int cus = num_cus(c->device);
            const int sp_o = 1;  // (the out-projection's 8 steps split 4 ways: 9.2 + 5.3 us for the two launches against 8.5)
            // (round 6, ADVICE r5: the split is a function of the product's shape alone -- 8 ways whenever fc2's K allows --
            //  not of the row count or the CU count: a query's vector then does not depend on the device or, up to
            //  SPLITK_MAX_ROWS rows, on what shares its call; splitk_choice would have said 8 at every size that matters)
            (void)cus;
            const int sp_2 = (causal && !no_splitk && R <= SPLITK_MAX_ROWS && M % (8 * 128) == 0) ? 8 : 1;
            prod.stats_out = st_h2;
            if (sp_o > 1) SSW_TRY(launch_gemm_splitk_stats(s, c->att, ly.wo, ly.bo, h, h2, c->splitk, R, D, D, sp_o, prod));
            else SSW_TRY(launch_gemm_bf16_ln(EPI_F32_BIAS_RESIDUAL_STATS, s, c->att, ly.wo, ly.bo, h, h2, R, D, D, prod));
            cons.stats_in = st_h2;
            cons.c1 = ly.c1fc1;
            SSW_TRY(launch_gemm_bf16_ln(EPI_BF16_LN_GELU, s, c->xn, ly.w1_ln, ly.c2fc1, nullptr, c->h1, R, M, D, cons));
            prod.stats_out = st_h;
            if (sp_2 > 1) SSW_TRY(launch_gemm_splitk_stats(s, c->h1, ly.w2, ly.b2, h2, h, c->splitk, R, D, M, sp_2, prod));
            else SSW_TRY(launch_gemm_bf16_ln(EPI_F32_BIAS_RESIDUAL_STATS, s, c->h1, ly.w2, ly.b2, h2, h, R, D, M, prod));
        }
        SSW_TAP(l, c->stream_in_xn ? nullptr : h, c->stream_in_xn ? c->xn : nullptr);
    }
    SSW_HIP_TRY(hipGetLastError());
    return SSW_OK;
}

// pooled rows -> final LayerNorm -> projection (+ optional L2 normalisation).  The projection is a
// [B, D] x [P, D]^T GEMM like every other linear; only odd shapes take the wave-per-column kernel.
ssw_status pool_and_project(ssw_clip *c, const Tower &tw, int B, int D, int normalize, float *out_dev) {
    hipStream_t s = c->stream;
    const Header &h = c->hdr;
    if (c->stream_in_xn) hipLaunchKernelGGL(widen_rows, dim3(B < 1024 ? B : 1024), dim3(256), 0, s, c->xn, c->rows, B, D, c->hidden);
    if (h.proj % 16 == 0 && skinny_rows(B, D, tw.M)) {
        launch_skinny<SK_F32, true>(s, c->hidden, c->rows, tw.lnf_w, tw.lnf_b, h.ln_eps, tw.proj, nullptr, nullptr,
                                    out_dev, B, h.proj, D);
        if (normalize) hipLaunchKernelGGL(l2norm_rows, dim3((B + 3) / 4), dim3(256), 0, s, out_dev, B, h.proj);
    } else if (h.proj % 128 == 0 && D % 64 == 0) {
        hipLaunchKernelGGL(layernorm_rows<bf16>, dim3((B + 3) / 4), dim3(256), 0, s, c->hidden, c->rows, B, D,
                           tw.lnf_w, tw.lnf_b, h.ln_eps, c->xn);
        SSW_TRY(gemm<EPI_F32>(s, c->xn, tw.proj, nullptr, nullptr, out_dev, B, h.proj, D));
        if (normalize) hipLaunchKernelGGL(l2norm_rows, dim3((B + 3) / 4), dim3(256), 0, s, out_dev, B, h.proj);
    } else {
        hipLaunchKernelGGL(layernorm_rows<float>, dim3((B + 3) / 4), dim3(256), 0, s, c->hidden, c->rows, B, D,
                           tw.lnf_w, tw.lnf_b, h.ln_eps, c->pooled);
        hipLaunchKernelGGL(project_rows, dim3(B), dim3(256), 0, s, c->pooled, tw.proj, D, h.proj, normalize, out_dev);
    }
    SSW_HIP_TRY(hipGetLastError());
    return SSW_OK;
}

// Tiles (or texts) a host-side call hands to the device at a time.  A row's result does not depend on its batch
// (tests/test_clip_gpu.py::test_more_tiles_than_one_device_chunk), so the size is a throughput choice: the image tower
// costs 12.2-12.3 us a tile at 200-256 tiles a call and 10.4-10.8 from 384 on (tile-count rounding of the 256-row GEMM
// tiles, launch ramps); 1024 tiles are ~2 GB of activations of the 288.
constexpr int HOST_CHUNK = 1024;

// c->patches holds the im2col'ed bf16 patches of B images
ssw_status image_forward_from_patches(ssw_clip *c, int B, int normalize, float *out_dev) {
    hipStream_t s = c->stream;
    const Header &h = c->hdr;
    const int g = h.image / h.patch, T = g * g + 1, D = h.v_hidden;
    const int pcols = 3 * h.patch * h.patch;
    SSW_TRY(gemm<EPI_F32>(s, c->patches, c->patch_w, nullptr, nullptr, c->patch_out, B * (T - 1), D, pcols));
    // cls / patch rows + position embedding, then the pre-LayerNorm = the stack's input row, in one pass: bf16 copy and
    // statistics for layer 0's QKV, and the f32 row into c->hidden unless the stack keeps its rows in bf16
    // (round 3: putting the rows together was a launch and a 30-MB round trip of its own at B = 200)
    hipLaunchKernelGGL(stack_input_rows<true>, dim3((B * T + 3) / 4), dim3(256), 0, s, c->patch_out, B * T, D, c->pre_w,
                       c->pre_b, h.ln_eps, bf16_rows(c, c->vis, 0) ? (float *)nullptr : c->hidden, c->xn, c->stats_a,
                       D / 128 > 0 ? D / 128 : 1, c->cls, c->vpos, T);
    SSW_TRY(run_tower(c, c->vis, B, T, 0));
    {   // the pooled rows' list depends on (B, stride) only: a launch (~5 us of a small forward) the first time, none after
        const int stride = c->pooled_compact ? 1 : T;
        const int64_t key = ((int64_t)B << 32) | (uint32_t)stride;
        if (c->rows_key != key) {
            hipLaunchKernelGGL(cls_rows, dim3((B + 255) / 256), dim3(256), 0, s, B, stride, c->rows);
            c->rows_key = key;
        }
    }
    return pool_and_project(c, c->vis, B, D, normalize, out_dev);
}

ssw_status image_forward(ssw_clip *c, const float *pixels_dev, int B, int normalize, float *out_dev) {
    const Header &h = c->hdr;
    hipLaunchKernelGGL(im2col_patches, dim3(2048), dim3(256), 0, c->stream, pixels_dev, c->patches, B, h.image, h.patch);
    return image_forward_from_patches(c, B, normalize, out_dev);
}

// CLIP's pixel statistics (make_clip_transform, embeddings.py:405-419; batch_tx, multiscale_tools.py:176-179)
ssw_status image_forward_u8(ssw_clip *c, const uint8_t *tiles_dev, int B, int normalize, float *out_dev) {
    const Header &h = c->hdr;
    hipLaunchKernelGGL(im2col_patches_u8, dim3(2048), dim3(256), 0, c->stream, tiles_dev, c->patches, B, h.image, h.patch,
                       0.48145466f, 0.4578275f, 0.40821073f, 0.26862954f, 0.26130258f, 0.27577711f);
    return image_forward_from_patches(c, B, normalize, out_dev);
}

// A single query is ~52 dependent launches of 4-6 us kernels (85 through the tile kernels).  Replaying them as a captured hipGraph was measured
// (round 2: 0.300 ms eager, 0.298 ms replayed; 0.645 / 0.631 before the skinny kernels) and dropped: the launches are
// not host-bound, the queue already holds them back to back.
ssw_status text_forward(ssw_clip *c, const int *ids_dev, int B, int L, int normalize, float *out_dev) {
    hipStream_t s = c->stream;
    const Header &h = c->hdr;
    const int D = h.t_hidden;
    hipLaunchKernelGGL(text_embed, dim3(1024), dim3(256), 0, s, ids_dev, c->tok, c->tpos, c->hidden, B, L, D);
    if (!(skinny_rows(B * L, D, c->txt.M) && c->txt.H * 64 == D))  // the tile path wants the row's bf16 copy + statistics
        hipLaunchKernelGGL(stack_input_rows<false>, dim3((B * L + 3) / 4), dim3(256), 0, s, c->hidden, B * L, D,
                           (const float *)nullptr, (const float *)nullptr, h.ln_eps, c->hidden, c->xn, c->stats_a,
                           D / 128 > 0 ? D / 128 : 1);
    SSW_TRY(run_tower(c, c->txt, B, L, 1));
    hipLaunchKernelGGL(eos_rows, dim3((B + 255) / 256), dim3(256), 0, s, ids_dev, B, L, h.eos, c->rows);
    c->rows_key = -1;
    return pool_and_project(c, c->txt, B, D, normalize, out_dev);
}

}  // namespace

extern "C" {

ssw_status ssw_clip_set_option(ssw_clip *c, int32_t option, int32_t value) {
    SSW_REQUIRE(c != nullptr, "clip is NULL");
    SSW_REQUIRE(option >= 0 && option <= 4, "ssw_clip_set_option: option %d unknown (SSW_CLIP_OPT_*)", option);
    DeviceGuard guard(c->device);
    SSW_HIP_TRY(hipStreamSynchronize(c->stream));  // a forward in flight keeps the form it started with
    if (value) c->flags |= 1 << option;
    else c->flags &= ~(1 << option);
    return SSW_OK;
}

#ifdef SSW_DEBUG_HOOKS
ssw_status ssw_clip_debug_tap(ssw_clip *c, int32_t tower, int32_t layer) {
    SSW_REQUIRE(c != nullptr, "clip is NULL");
    c->tap_tower = tower;
    c->tap_layer = layer;
    c->tap_rows = c->tap_dim = 0;
    return SSW_OK;
}

ssw_status ssw_clip_debug_tap_read(ssw_clip *c, float *out_host, int64_t cap_floats, int64_t *out_rows, int32_t *out_dim) {
    SSW_REQUIRE(c != nullptr && out_rows && out_dim, "NULL argument");
    DeviceGuard guard(c->device);
    SSW_HIP_TRY(hipStreamSynchronize(c->stream));
    *out_rows = c->tap_rows;
    *out_dim = (int32_t)c->tap_dim;
    if (out_host && c->tap_rows * c->tap_dim > 0) {
        SSW_REQUIRE(cap_floats >= c->tap_rows * c->tap_dim, "tap: %lld floats, buffer holds %lld",
                    (long long)(c->tap_rows * c->tap_dim), (long long)cap_floats);
        SSW_HIP_TRY(hipMemcpy(out_host, c->tap_buf, (size_t)(c->tap_rows * c->tap_dim) * sizeof(float), hipMemcpyDeviceToHost));
    }
    return SSW_OK;
}
#endif

ssw_status ssw_clip_destroy(ssw_clip *c) {
    if (!c) return SSW_OK;
    DeviceGuard guard(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (void *p : c->allocs) (void)hipFree(p);
    for (void *p : {(void *)c->hidden, (void *)c->hidden2, (void *)c->pooled, (void *)c->patch_out, (void *)c->xn,
                    (void *)c->qkv, (void *)c->att, (void *)c->h1, (void *)c->patches, (void *)c->pixels,
                    (void *)c->out, (void *)c->ids, (void *)c->rows, (void *)c->stats_a, (void *)c->stats_b, (void *)c->splitk})
        (void)hipFree(p);
    if (c->tap_buf) (void)hipFree(c->tap_buf);  // (allocated by the lab build's tap only; either build's destroy frees it)
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return SSW_OK;
}

ssw_status ssw_clip_create(int32_t device, const void *weight_blob, size_t bytes, ssw_clip **out) {
    SSW_REQUIRE(out != nullptr && weight_blob != nullptr, "NULL argument");
    *out = nullptr;
    SSW_REQUIRE(bytes > sizeof(Header), "clip: blob too small");
    Header h;
    memcpy(&h, weight_blob, sizeof(Header));
    SSW_REQUIRE(memcmp(h.magic, "SSWCLIP1", 8) == 0, "clip: bad blob magic");
    const int g = h.patch > 0 ? h.image / h.patch : 0;
    const int Tv = g * g + 1;
    if (h.v_hidden % 128 || h.t_hidden % 128 || h.v_mlp % 128 || h.t_mlp % 128 || h.proj % 4 ||
        (3 * h.patch * h.patch) % 64 || h.v_hidden / h.v_heads != 64 || h.t_hidden / h.t_heads != 64 ||
        Tv > ATT_MAX_S || h.t_maxpos > ATT_MAX_S || h.v_hidden > 1024 || h.t_hidden > 1024 || h.proj > 1024) {
        set_error("clip: configuration outside what the kernels support (head_dim 64, <= 80 tokens, dims %% 128)");
        return SSW_ERR_UNSUPPORTED;
    }
    DeviceGuard guard(device);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", device);
        return SSW_ERR_HIP;
    }
    ssw_clip *c = new (std::nothrow) ssw_clip();
    if (!c) return SSW_ERR_NOMEM;
    c->device = device;
    c->hdr = h;
    c->flags = clip_flags_from_env();
    auto bail = [&](ssw_status s) {
        ssw_clip_destroy(c);
        return s;
    };
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail(SSW_ERR_HIP);
    const float *cur = reinterpret_cast<const float *>(reinterpret_cast<const char *>(weight_blob) + sizeof(Header));
    const float *end = reinterpret_cast<const float *>(reinterpret_cast<const char *>(weight_blob) + bytes);
    // scratch large enough for the biggest single tensor (token embedding or a fused qkv)
    size_t big = (size_t)h.vocab * h.t_hidden;
    big = std::max(big, (size_t)3 * h.v_hidden * h.v_hidden);
    big = std::max(big, (size_t)h.v_mlp * h.v_hidden);
    big = std::max(big, (size_t)h.v_hidden * 3 * h.patch * h.patch);
    float *scratch = nullptr;
    if (hipMalloc((void **)&scratch, big * sizeof(float)) != hipSuccess) {
        set_error("clip: scratch allocation failed");
        return bail(SSW_ERR_NOMEM);
    }
    ssw_status st = SSW_OK;
    do {
        c->vis.D = h.v_hidden; c->vis.L = h.v_layers; c->vis.H = h.v_heads; c->vis.M = h.v_mlp; c->vis.T = Tv;
        c->txt.D = h.t_hidden; c->txt.L = h.t_layers; c->txt.H = h.t_heads; c->txt.M = h.t_mlp; c->txt.T = h.t_maxpos;
        if ((st = upload_f32(c, cur, end, h.v_hidden, &c->cls)) != SSW_OK) break;
        if ((st = upload_bf16(c, cur, end, (size_t)h.v_hidden * 3 * h.patch * h.patch, &c->patch_w, scratch)) != SSW_OK) break;
        if ((st = upload_f32(c, cur, end, (size_t)Tv * h.v_hidden, &c->vpos)) != SSW_OK) break;
        if ((st = upload_f32(c, cur, end, h.v_hidden, &c->pre_w)) != SSW_OK) break;
        if ((st = upload_f32(c, cur, end, h.v_hidden, &c->pre_b)) != SSW_OK) break;
        if ((st = load_tower(c, c->vis, cur, end, scratch)) != SSW_OK) break;
        if ((st = upload_bf16(c, cur, end, (size_t)h.proj * h.v_hidden, &c->vis.proj, scratch)) != SSW_OK) break;
        if ((st = upload_f32(c, cur, end, (size_t)h.vocab * h.t_hidden, &c->tok)) != SSW_OK) break;
        if ((st = upload_f32(c, cur, end, (size_t)h.t_maxpos * h.t_hidden, &c->tpos)) != SSW_OK) break;
        if ((st = load_tower(c, c->txt, cur, end, scratch)) != SSW_OK) break;
        if ((st = upload_bf16(c, cur, end, (size_t)h.proj * h.t_hidden, &c->txt.proj, scratch)) != SSW_OK) break;
        if (cur != end) {
            set_error("clip: %zu unread floats at the end of the weight blob", (size_t)(end - cur));
            st = SSW_ERR_INVALID;
        }
    } while (0);
    (void)hipFree(scratch);
    if (st != SSW_OK) return bail(st);
    *out = c;
    return SSW_OK;
}

ssw_status ssw_clip_embed_image(ssw_clip *c, const float *nchw_host, int32_t b, int32_t normalize,
                                float *out_host) {
    SSW_REQUIRE(c && nchw_host && out_host && b > 0, "bad argument");
    DeviceGuard guard(c->device);
    const Header &h = c->hdr;
    const int chunk = HOST_CHUNK;
    SSW_TRY(reserve(c, std::min<int64_t>(b, chunk)));
    const size_t per = (size_t)3 * h.image * h.image;
    for (int b0 = 0; b0 < b; b0 += chunk) {
        const int nb = std::min(chunk, b - b0);
        SSW_HIP_TRY(hipMemcpyAsync(c->pixels, nchw_host + (size_t)b0 * per, (size_t)nb * per * sizeof(float),
                                   hipMemcpyHostToDevice, c->stream));
        SSW_TRY(image_forward(c, c->pixels, nb, normalize, c->out));
        SSW_HIP_TRY(hipMemcpyAsync(out_host + (size_t)b0 * h.proj, c->out, (size_t)nb * h.proj * sizeof(float),
                                   hipMemcpyDeviceToHost, c->stream));
        SSW_HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return SSW_OK;
}

ssw_status ssw_clip_embed_tiles_u8(ssw_clip *c, const uint8_t *tiles_hwc_host, int32_t b, int32_t normalize,
                                   float *out_host) {
    SSW_REQUIRE(c && tiles_hwc_host && out_host && b > 0, "bad argument");
    DeviceGuard guard(c->device);
    const Header &h = c->hdr;
    SSW_REQUIRE(h.patch % 8 == 0, "clip: patch size %d is not a multiple of 8", h.patch);
    const int chunk = HOST_CHUNK;
    SSW_TRY(reserve(c, std::min<int64_t>(b, chunk)));
    const size_t per = (size_t)3 * h.image * h.image;  // bytes per tile; c->pixels (f32) is 4x that
    for (int b0 = 0; b0 < b; b0 += chunk) {
        const int nb = std::min(chunk, b - b0);
        SSW_HIP_TRY(hipMemcpyAsync(c->pixels, tiles_hwc_host + (size_t)b0 * per, (size_t)nb * per, hipMemcpyHostToDevice,
                                   c->stream));
        SSW_TRY(image_forward_u8(c, reinterpret_cast<const uint8_t *>(c->pixels), nb, normalize, c->out));
        SSW_HIP_TRY(hipMemcpyAsync(out_host + (size_t)b0 * h.proj, c->out, (size_t)nb * h.proj * sizeof(float),
                                   hipMemcpyDeviceToHost, c->stream));
        SSW_HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return SSW_OK;
}

ssw_status ssw_clip_embed_image_dev(ssw_clip *c, void *hip_stream, const float *nchw_dev, int32_t b,
                                    int32_t normalize, float *out_dev) {
    SSW_REQUIRE(c && nchw_dev && out_dev && b > 0, "bad argument");
    DeviceGuard guard(c->device);
    SSW_REQUIRE(b <= 1024, "clip: at most 1024 images per device-side call");
    SSW_TRY(reserve(c, b));
    hipStream_t saved = c->stream;
    if (hip_stream) c->stream = (hipStream_t)hip_stream;
    ssw_status st = image_forward(c, nchw_dev, b, normalize, out_dev);
    c->stream = saved;
    return st;
}

ssw_status ssw_clip_embed_text(ssw_clip *c, const int32_t *ids_host, int32_t b, int32_t seq_len, int32_t normalize,
                               float *out_host) {
    SSW_REQUIRE(c && ids_host && out_host && b > 0, "bad argument");
    const Header &h = c->hdr;
    SSW_REQUIRE(seq_len >= 1 && seq_len <= h.t_maxpos, "clip: sequence length %d outside [1, %d]", seq_len, h.t_maxpos);
    for (int64_t i = 0; i < (int64_t)b * seq_len; ++i)
        SSW_REQUIRE(ids_host[i] >= 0 && ids_host[i] < h.vocab, "clip: token id %d outside the vocabulary", ids_host[i]);
    if (h.eos != 2)  // legacy configs pool at argmax(ids), which always exists
        for (int r = 0; r < b; ++r) {
            bool found = false;
            for (int j = 0; j < seq_len && !found; ++j) found = ids_host[(int64_t)r * seq_len + j] == h.eos;
            SSW_REQUIRE(found, "clip: sequence %d holds no end-of-text token (id %d): nothing to pool", r, h.eos);
        }
    DeviceGuard guard(c->device);
    const int chunk = HOST_CHUNK;
    SSW_TRY(reserve(c, std::min<int64_t>(b, chunk)));
    for (int b0 = 0; b0 < b; b0 += chunk) {
        const int nb = std::min(chunk, b - b0);
        SSW_HIP_TRY(hipMemcpyAsync(c->ids, ids_host + (size_t)b0 * seq_len, (size_t)nb * seq_len * sizeof(int),
                                   hipMemcpyHostToDevice, c->stream));
        SSW_TRY(text_forward(c, c->ids, nb, seq_len, normalize, c->out));
        SSW_HIP_TRY(hipMemcpyAsync(out_host + (size_t)b0 * h.proj, c->out, (size_t)nb * h.proj * sizeof(float),
                                   hipMemcpyDeviceToHost, c->stream));
        SSW_HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return SSW_OK;
}

ssw_status ssw_clip_sync(ssw_clip *c) {
    SSW_REQUIRE(c != nullptr, "NULL argument");
    DeviceGuard guard(c->device);
    SSW_HIP_TRY(hipStreamSynchronize(c->stream));
    return SSW_OK;
}

}  // extern "C"
