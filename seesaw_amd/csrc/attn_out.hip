// Attention and the out-projection of a ViT-B/32 layer in one launch (round 4): a workgroup per image.
//
// The tile path ran attention (one workgroup per (image, head): 15.6 us a layer at B = 200) and then the
// out-projection as a [B*50, 768] x [768, 768] product (21 us: 11.8 GFLOP, mostly ramp, prologue and the exposed
// +residual epilogue of a single round of tiles), with the attention output -- 15 MB a layer -- written to memory and
// read back in between.  Here the image's 50 token rows (padded to 64) never leave the CU:
//   phase 1  attention, two heads at a time: the pair's Q / K / V rows come in coalesced (16 bytes a thread, the next
//            pair's requested before this pair is computed), go to LDS in attention_rows64's layouts, wave
//            (head of the pair, query tile) runs the same fragments through the same MFMAs (identical bf16 output),
//            and the 16 x 64 output tile lands in sO [64 rows][768] (LDS, 96 KB);
//   phase 2  out = sO Wo^T + bo: wave w owns output columns [96 w, 96 w + 96) of all 64 rows -- 4 x 6 accumulator tiles
//            -- its A fragments come from sO (ds_read_b128, conflict-free through a 16-chunk XOR by row), its Wo
//            fragments straight from memory (no other wave wants them: no LDS round trip), PD K-steps ahead; k ascends
//            exactly as in the tile GEMM, so the products are the GEMM's bit for bit;
//   epilogue the residual row is added and the new row written back 16 bytes a lane through LDS, with the row's
//            LayerNorm partial sums for the fc1 product (two partial pairs a row: columns [0, 384) and [384, 768)).
// Measured and not kept (round 4, in-kernel stamps): the two phases interleaved per head pair -- attention of pair p and
// the product's K-steps of pair p - 1 in one iteration, 32 KB of output tiles instead of 96 -- costs the SUM of the two
// per iteration whichever way the source orders them (7.2 k cycles sequential: the compiler does not move the product's
// LDS reads across the attention's LDS writes; 8.5-10 k with the product's steps woven between the attention's stages,
// 26 spilled registers; 30 k+ with the two wave halves in opposite orders, 740 spilled registers): same 72 k cycles a
// launch as the phases.  What would overlap them is a second instruction stream (waves specialised by role), which the
// 96 accumulator registers a product wave needs for its 96 columns do not leave room for at eight waves.
// Replaces, like the kernels it fuses, the attention + out_proj of transformers' CLIPEncoderLayer that the reference
// calls through HGFaceWrapper.forward (seesaw/models/model.py:50-57).
#include "ssw_common.h"
#include <algorithm>
#include <cstdlib>
#ifndef SSW_AO_NT
#define SSW_AO_NT 1  // non-temporal: bit 0 the qkv loads (each byte is read once), bit 1 the residual loads, bit 2 the row stores.
                     // B = 200 forward, f32 rows, two rounds: 0: 2.569 ms, 1: 2.559, 2: 2.63, 3: 2.62, 4: 2.63
#endif
template <int BIT, typename V>
__device__ __forceinline__ V ao_load(const V *p) {
    if constexpr ((SSW_AO_NT >> BIT) & 1) return __builtin_nontemporal_load(p);
    else return *p;
}
template <int BIT, typename V>
__device__ __forceinline__ void ao_store(V *p, V v) {
    if constexpr ((SSW_AO_NT >> BIT) & 1) __builtin_nontemporal_store(v, p);
    else *p = v;
}
#ifndef SSW_AO_WO_LATE
#define SSW_AO_WO_LATE 0  // the first Wo fragments (96 KB a workgroup, L2 hits) requested behind the first Q / K / V pairs instead of ahead of them
#endif
#ifndef SSW_AO_AHEAD
#define SSW_AO_AHEAD 3  // head pairs of Q / K / V requested ahead: 3 (2 is level) ends phase 1 3 k cycles earlier than 4, whose burst delays the first pair
#endif

namespace ssw {
namespace {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ __forceinline__ bf16 to_bf16(float x) { return (bf16)x; }

constexpr int AO_D = 768, AO_H = 12, AO_NW = 8;
constexpr int AO_NJ = AO_D / AO_NW / 16;           // 6 accumulator column tiles per wave
constexpr int AO_ROWB = AO_D * 2;                  // bytes of an sO row
constexpr int AO_SO = 64 * AO_ROWB;                // 98304
constexpr int AO_SK = AO_SO, AO_SV = AO_SK + 2 * 8192, AO_SQ = AO_SV + 2 * 8192;
constexpr int AO_LDS = AO_SQ + 2 * 4 * 2048;       // 147456
// epilogue reuse of the same bytes: per-wave f32 staging of 16 rows x 96 columns (400-byte rows), then the partial sums
constexpr int AO_EROW = 96 * 4 + 16;
constexpr int AO_ESTAGE = 16 * AO_EROW;            // 6400 per wave
constexpr int AO_EPART = AO_NW * AO_ESTAGE;        // partial (sum, sq): [wave][64 rows][12] float2
static_assert(AO_EPART + AO_NW * 64 * 12 * 8 <= AO_LDS, "epilogue scratch fits");

// byte offset in sO of the 16-byte chunk holding elements [k0, k0 + 8) of row `row` (k0 a multiple of 8): inside each
// 256-byte window the chunk index is XORed with the row, so the 16 rows a ds_read_b128 lane group touches -- and the 16
// rows of a ds_write_b64 group -- hit 16 different slots (rows are 1536 B apart: a multiple of the 256-byte bank row)
__device__ __forceinline__ int so_off(int row, int k0) {
    const int b = k0 * 2;
    return row * AO_ROWB + (b & ~255) + ((((b >> 4) & 15) ^ (row & 15)) << 4);
}

// lab build (SSW_DEBUG_HOOKS, SSW_AO_STAMPS=1): s_memtime at the phase boundaries, wave 0 of every workgroup
#ifdef SSW_DEBUG_HOOKS
__device__ unsigned long long g_ao_stamps[1024 * 32];
#define AO_STAMP(slot)                                                                                   \
    if (STAMP && t == 0 && blockIdx.x < 1024) g_ao_stamps[blockIdx.x * 32 + (slot)] = __builtin_amdgcn_s_memtime();
#else
#define AO_STAMP(slot)
#endif

template <bool BF, int PD, bool STAMP = false>
__global__ __launch_bounds__(512) void attn_outproj_image(const bf16 *__restrict__ qkv, const bf16 *__restrict__ Wo,
                                                          const float *__restrict__ bo, bf16 *__restrict__ xcopy,
                                                          const float *__restrict__ res_in, float *__restrict__ res_out,
                                                          float *__restrict__ stats_out, int S, float scale, int B, int aff_tiles) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    auto g_off = [](int row, int c16) { return row * 64 + ((c16 ^ ((row >> 1) & 7)) << 3); };
    auto v_off = [](int key, int c32) { return key * 64 + ((c32 ^ (((key >> 1) & 1) | (((key >> 3) & 1) << 1))) << 4); };
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    int b = blockIdx.x;
    if (aff_tiles > 0) {
        // round 6 experiment: workgroup ids b and b + 8 share an XCD, and XCD x produced row tiles [x per, (x + 1) per) of
        // the qkv rows (GemmLn::xcd_contig): it takes the images whose middle row lies there
        const int xcd = b & 7, idx = b >> 3, per = (aff_tiles + 7) >> 3;
        const int lo = xcd * per * 128, hi = min((xcd + 1) * per, aff_tiles) * 128;  // rows [lo, hi)
        const int half = S >> 1;
        const int i0 = lo <= half ? 0 : (lo - half + S - 1) / S;
        int i1 = (hi - 1 - half) / S;
        if (xcd == 7 || hi >= B * S) i1 = B - 1;
        b = i0 + idx;
        if (hi <= lo || b > i1 || b >= B) return;
    }
    const int64_t row_base = (int64_t)b * S;
    const int n0 = wave * (AO_NJ * 16);
    AO_STAMP(0)
#ifdef SSW_DEBUG_HOOKS
    if (STAMP && t == 0 && blockIdx.x < 1024) {  // which XCD this workgroup runs on, and which image it took
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_ao_stamps[blockIdx.x * 32 + 21] = xcc & 0xf;
        g_ao_stamps[blockIdx.x * 32 + 22] = (unsigned long long)b + 1;
    }
#endif

    // ---- Wo fragments of the first PD K-steps (64 deep), in flight under phase 1.  Wo arrives PACKED (pack_wo below):
    // the 1 KB a wave-instruction loads for one fragment is contiguous in lane order -- 8 whole 128-byte lines.  Straight
    // from the [out][in] matrix a fragment is 16 rows x 64 B, and the CU's L1 took 58 k cycles to look up the 18 k
    // half-lines of a workgroup's 1.18 MB (20 B per clock; in-kernel stamps), three times what the MFMAs need.
    const bf16 *w_lane = Wo + ((int64_t)wave * (AO_D / 64) * AO_NJ * 2 * 64 + lane) * 8;
    constexpr int W_STEP = AO_NJ * 2 * 64 * 8;  // elements per (wave, K-step)
    bf16x8 wf[PD][AO_NJ][2];
    auto fetch_wf = [&]() {
#pragma unroll
        for (int s = 0; s < PD; ++s)
#pragma unroll
            for (int j = 0; j < AO_NJ; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) wf[s][j][h] = *reinterpret_cast<const bf16x8 *>(w_lane + s * W_STEP + (j * 2 + h) * 512);
    };
    if constexpr (SSW_AO_WO_LATE == 0) fetch_wf();

    // ---- phase 1: attention, a pair of heads per iteration
    {
        const int hh = t >> 8, u = t & 255;           // staging: head of the pair, chunk index inside the head
        const int r0 = u >> 3, c16 = u & 7;           // rows r0 and r0 + 32, 16-byte chunk c16
        // rows beyond S read row S - 1: finite numbers that never reach a stored value (as keys they are masked, their
        // probabilities are exactly 0; as queries their rows are not stored)
        const bf16 *src0 = qkv + (row_base + min(r0, S - 1)) * 3 * AO_D + hh * 64 + c16 * 8;
        const bf16 *src1 = qkv + (row_base + min(r0 + 32, S - 1)) * 3 * AO_D + hh * 64 + c16 * 8;
        // A CU takes in ~11 B per clock with one pair (48 KB) in flight -- 38 k cycles for the image's 288 KB in a first
        // version; four pairs are requested up front, the other two as register sets come free
        bf16x8 rr[AO_H / 2][6];
        auto fetch = [&](int p) {
            rr[p][0] = ao_load<0>(reinterpret_cast<const bf16x8 *>(src0 + p * 128));
            rr[p][1] = ao_load<0>(reinterpret_cast<const bf16x8 *>(src0 + p * 128 + AO_D));
            rr[p][2] = ao_load<0>(reinterpret_cast<const bf16x8 *>(src0 + p * 128 + 2 * AO_D));
            rr[p][3] = ao_load<0>(reinterpret_cast<const bf16x8 *>(src1 + p * 128));
            rr[p][4] = ao_load<0>(reinterpret_cast<const bf16x8 *>(src1 + p * 128 + AO_D));
            rr[p][5] = ao_load<0>(reinterpret_cast<const bf16x8 *>(src1 + p * 128 + 2 * AO_D));
        };
        constexpr int AHEAD = SSW_AO_AHEAD;
#pragma unroll
        for (int p = 0; p < AHEAD; ++p) fetch(p);
        if constexpr (SSW_AO_WO_LATE != 0) fetch_wf();  // behind the first pairs in the memory queue, not ahead of them
        const int ah = wave >> 2, qt = wave & 3;      // attention: head of the pair, query tile
        bf16 *const sK = reinterpret_cast<bf16 *>(smem + AO_SK) + ah * 4096;
        bf16 *const sV = reinterpret_cast<bf16 *>(smem + AO_SV) + ah * 4096;
        bf16 *const pt = reinterpret_cast<bf16 *>(smem + AO_SQ) + (ah * 4 + qt) * 1024;
        bf16 *const stK = reinterpret_cast<bf16 *>(smem + AO_SK) + hh * 4096;
        bf16 *const stV = reinterpret_cast<bf16 *>(smem + AO_SV) + hh * 4096;
        bf16 *const stQ = reinterpret_cast<bf16 *>(smem + AO_SQ) + hh * 4096;  // 4 tiles x 16 rows = rows 0 .. 63 in g_off order
#pragma unroll
        for (int p = 0; p < AO_H / 2; ++p) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int row = r0 + 32 * c;
                // (a query tile's 16 rows are 2 KB of their own: tile row / 16, g_off on the row inside the tile -- the row's
                //  swizzle term (row >> 1) & 7 is the same for row and row % 16)
                *reinterpret_cast<bf16x8 *>(&stQ[(row >> 4) * 1024 + g_off(row & 15, c16)]) = rr[p][3 * c];
                *reinterpret_cast<bf16x8 *>(&stK[g_off(row, c16)]) = rr[p][3 * c + 1];
                *reinterpret_cast<bf16x8 *>(&stV[v_off(row, c16 >> 1) + (c16 & 1) * 8]) = rr[p][3 * c + 2];
            }
            __syncthreads();
            AO_STAMP(8 + 2 * p)
            if (p + AHEAD < AO_H / 2) fetch(p + AHEAD);
            const int head = 2 * p + ah;
            if (qt * 16 < S) {  // wave-uniform
                bf16x8 kf[4][2], qf[2];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) kf[j][ks] = *reinterpret_cast<const bf16x8 *>(&sK[g_off(j * 16 + fr, ks * 4 + fq)]);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const bf16x8 *>(&pt[g_off(fr, ks * 4 + fq)]);
                f32x4 sc[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    sc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) sc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[j][ks], qf[ks], sc[j], 0, 0, 0);
                }
                // this lane holds scores[query = 16 qt + fr][key = 16 j + 4 fq + r]
                float mx = -INFINITY;
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
                // P (bf16) over this tile's Q rows (its fragments are in registers)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bf16x4 pv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pv[r] = to_bf16(sc[j][r] * inv);
                    *reinterpret_cast<bf16x4 *>(&pt[g_off(fr, j * 2 + (fq >> 1)) + (fq & 1) * 4]) = pv;
                }
                // out^T tile = V^T P^T : o[dt][r] = out[query fr][d = 16 dt + 4 fq + r]
                f32x4 o[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 pa = *reinterpret_cast<const bf16x8 *>(&pt[g_off(fr, ks * 4 + fq)]);
                    const int kq = ks * 32 + 8 * fq + (fr >> 2), dp = (fr & 3) * 4;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        typedef __attribute__((ext_vector_type(4))) short s16x4;
                        typedef __attribute__((address_space(3))) s16x4 *lds_s16x4;
                        typedef __attribute__((ext_vector_type(8))) short s16x8;
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(&sV[v_off(kq, dt) + dp]));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(&sV[v_off(kq + 4, dt) + dp]));
                        const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, both), pa, o[dt], 0, 0, 0);
                    }
                }
                // the tile into sO: row = query, k = head * 64 + d; a lane's 4 consecutive d are 8 bytes
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    bf16x4 ov;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ov[r] = to_bf16(o[dt][r]);
                    const int k = head * 64 + dt * 16 + fq * 4;
                    *reinterpret_cast<bf16x4 *>(smem + so_off(qt * 16 + fr, k & ~7) + (k & 4) * 2) = ov;
                }
            } else {  // no live query in this tile: rows nobody stores, kept finite
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    bf16x4 z;
#pragma unroll
                    for (int r = 0; r < 4; ++r) z[r] = (bf16)0.f;
                    const int k = head * 64 + dt * 16 + fq * 4;
                    *reinterpret_cast<bf16x4 *>(smem + so_off(qt * 16 + fr, k & ~7) + (k & 4) * 2) = z;
                }
            }
            AO_STAMP(9 + 2 * p)
            __syncthreads();  // every wave is done with the pair's staging bytes (and, after the last pair, sO is whole)
        }
    }

    AO_STAMP(1)
    // ---- phase 2: out = sO Wo^T + bo, K = 768 in 24 steps of 32; acc[i][j][r] = out[16 i + fr][n0 + 16 j + 4 fq + r]
    f32x4 acc[4][AO_NJ];
#pragma unroll
    for (int j = 0; j < AO_NJ; ++j) {
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(bo + n0 + j * 16 + fq * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = bv;
    }
    // bf16 stream: the image's residual rows (this wave's 96 columns: 12 chunks of 16 bytes a lane) are requested now and
    // used behind the K loop -- requested there, 16 rows at a time, they were four exposed round trips (7.5 us of stores
    // phase in a first version)
    bf16x8 rb[4][3];
    if constexpr (BF) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int c = lane + 64 * it, rl = c / 12, cc = c - rl * 12;
                rb[i][it] = ao_load<1>(reinterpret_cast<const bf16x8 *>(xcopy + (row_base + min(i * 16 + rl, S - 1)) * AO_D + n0 + cc * 8));
            }
    }
    // f32 rows: 16 rows (this wave's 96 columns: 3 chunks of 32 bytes a lane) at a time, the first block requested here,
    // the next while a block is added and stored (requested block by block at its use they were four exposed round
    // trips: 21 k cycles of stores phase against 14 k)
    f32x4 rf[2][3][2];
    auto fetch_rf = [&](int i, int set) {
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int c = lane + 64 * it, rl = c / 12, cc = c - rl * 12;
            const int64_t off = (row_base + min(i * 16 + rl, S - 1)) * AO_D + n0 + cc * 8;
            rf[set][it][0] = ao_load<1>(reinterpret_cast<const f32x4 *>(res_in + off));
            rf[set][it][1] = ao_load<1>(reinterpret_cast<const f32x4 *>(res_in + off + 4));
        }
    };
    if constexpr (!BF) fetch_rf(0, 0);
    constexpr int NKS = AO_D / 64;
    static_assert(NKS % PD == 0, "the Wo ring is indexed statically");
    for (int ks0 = 0; ks0 < NKS; ks0 += PD) {
#pragma unroll
        for (int s = 0; s < PD; ++s) {
            const int ks = ks0 + s;
#pragma unroll
            for (int h = 0; h < 2; ++h) {  // k ascends in steps of 32 as in the tile GEMM
                bf16x8 a[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8 *>(smem + so_off(i * 16 + fr, ks * 64 + h * 32 + fq * 8));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < AO_NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][j][h], a[i], acc[i][j], 0, 0, 0);
            }
            if (ks + PD < NKS) {
#pragma unroll
                for (int j = 0; j < AO_NJ; ++j)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        wf[s][j][h] = *reinterpret_cast<const bf16x8 *>(w_lane + (ks + PD) * W_STEP + (j * 2 + h) * 512);
            }
        }
    }

    AO_STAMP(2)
    // ---- epilogue: + residual row, new row out (16 bytes a lane), partial LayerNorm sums of the row as stored
    __syncthreads();  // every wave has read its last sO fragments: the bytes become staging
    unsigned char *const wl = smem + wave * AO_ESTAGE;
    float *const part = reinterpret_cast<float *>(smem + AO_EPART) + (int64_t)wave * 64 * 12 * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // chunk c = lane + 64 it of the block's 16 rows x 12 chunks of 8 columns
        if constexpr (!BF) {
            if (i + 1 < 4) fetch_rf(i + 1, (i + 1) & 1);
        }
#pragma unroll
        for (int j = 0; j < AO_NJ; ++j) *reinterpret_cast<f32x4 *>(wl + fr * AO_EROW + j * 64 + fq * 16) = acc[i][j];
        // (same wave writes and reads: LDS operations of a wave complete in order)
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int c = lane + 64 * it, rl = c / 12, cc = c - rl * 12;
            f32x4 lo = *reinterpret_cast<const f32x4 *>(wl + rl * AO_EROW + cc * 32);
            f32x4 hi = *reinterpret_cast<const f32x4 *>(wl + rl * AO_EROW + cc * 32 + 16);
            bf16x8 o;
            float ssum = 0.f, ssq = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (BF) {
                    lo[r] += (float)rb[i][it][r];
                    hi[r] += (float)rb[i][it][4 + r];
                } else {
                    lo[r] += rf[i & 1][it][0][r];
                    hi[r] += rf[i & 1][it][1][r];
                }
                o[r] = to_bf16(lo[r]);
                o[4 + r] = to_bf16(hi[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = BF ? (float)o[r] : lo[r], y = BF ? (float)o[4 + r] : hi[r];
                ssum += x;
                ssq += x * x;
                ssum += y;
                ssq += y * y;
            }
            const int row = i * 16 + rl;
            if (row < S) {
                const int64_t off = (row_base + row) * AO_D + n0 + cc * 8;
                ao_store<2>(reinterpret_cast<bf16x8 *>(xcopy + off), o);
                if constexpr (!BF) {
                    ao_store<2>(reinterpret_cast<f32x4 *>(res_out + off), lo);
                    ao_store<2>(reinterpret_cast<f32x4 *>(res_out + off + 4), hi);
                }
            }
            part[(row * 12 + cc) * 2] = ssum;
            part[(row * 12 + cc) * 2 + 1] = ssq;
        }
    }
    AO_STAMP(3)
    __syncthreads();
    {   // (row, column half) = t >> 2: lane t & 3 adds one wave's twelve chunks in order, then the four waves pair up
        const int pair = t >> 2, sub = t & 3, row = pair & 63, half = pair >> 6;
        const float *pp = reinterpret_cast<const float *>(smem + AO_EPART) + (int64_t)((half * 4 + sub) * 64 + row) * 24;
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int cc = 0; cc < 12; ++cc) {
            sm += pp[cc * 2];
            sq += pp[cc * 2 + 1];
        }
        sm += __shfl_xor(sm, 1, 64);
        sq += __shfl_xor(sq, 1, 64);
        sm += __shfl_xor(sm, 2, 64);
        sq += __shfl_xor(sq, 2, 64);
        if (sub == 0 && row < S) {
            float *o = stats_out + ((row_base + row) * 2 + half) * 2;
            o[0] = sm;
            o[1] = sq;
        }
    }
    AO_STAMP(4)
}

// Wo [768][768] (out x in) -> the order the product's waves load it in: fragment (wave w, K-step ks of 64, column tile j,
// half h) is 1 KB, lane (fr, fq) holding Wo[96 w + 16 j + fr][64 ks + 32 h + 8 fq ..+8]
__global__ void k_pack_wo(const bf16 *__restrict__ Wo, bf16 *__restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte chunk
    if (idx >= AO_D * AO_D / 8) return;
    const int lane = idx & 63, f = idx >> 6;
    const int h = f & 1, j = (f >> 1) % AO_NJ, ks = (f / (2 * AO_NJ)) % (AO_D / 64), w = f / (2 * AO_NJ * (AO_D / 64));
    const int fr = lane & 15, fq = lane >> 4;
    *reinterpret_cast<bf16x8 *>(out + (int64_t)idx * 8) =
        *reinterpret_cast<const bf16x8 *>(Wo + (int64_t)(w * 96 + j * 16 + fr) * AO_D + ks * 64 + h * 32 + fq * 8);
}


template <bool BF, int PD>
ssw_status launch_ao(hipStream_t s, const bf16 *qkv, const bf16 *Wo, const float *bo, bf16 *xcopy, const float *res_in,
                     float *res_out, float *stats_out, int B, int S, float scale, int aff_tiles) {
    int grid = B;
    if (aff_tiles > 0) {  // 8 x the largest number of images an XCD takes (the kernel's own rule, restated)
        const int per = (aff_tiles + 7) >> 3, half = S >> 1;
        int most = 0, prev_i1 = -1;
        for (int xcd = 0; xcd < 8; ++xcd) {
            const int lo = xcd * per * 128, hi = std::min((xcd + 1) * per, aff_tiles) * 128;
            if (hi <= lo) continue;
            const int i0 = lo <= half ? 0 : (lo - half + S - 1) / S;
            int i1 = (hi - 1 - half) / S;
            if (xcd == 7 || hi >= B * S) i1 = B - 1;
            i1 = std::min(i1, B - 1);
            if (i0 != prev_i1 + 1 && i0 <= i1) {
                set_error("attn_outproj: affinity map leaves a gap at image %d", i0);
                return SSW_ERR_INVALID;
            }
            if (i1 >= i0) prev_i1 = i1;
            most = std::max(most, i1 - i0 + 1);
        }
        if (prev_i1 != B - 1) {
            set_error("attn_outproj: affinity map ends at image %d of %d", prev_i1, B);
            return SSW_ERR_INVALID;
        }
        grid = 8 * most;
    }
    static bool attr_set[64] = {false};
    int dev = 0;
    SSW_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        SSW_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(attn_outproj_image<BF, PD>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, AO_LDS));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
#ifdef SSW_DEBUG_HOOKS
    static const bool stamps = getenv("SSW_AO_STAMPS") != nullptr;
    if (stamps) {
        SSW_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(attn_outproj_image<BF, PD, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, AO_LDS));
        hipLaunchKernelGGL((attn_outproj_image<BF, PD, true>), dim3(grid), dim3(512), AO_LDS, s, qkv, Wo, bo, xcopy, res_in,
                           res_out, stats_out, S, scale, B, aff_tiles);
        SSW_HIP_TRY(hipGetLastError());
        return SSW_OK;
    }
#endif
    hipLaunchKernelGGL((attn_outproj_image<BF, PD>), dim3(grid), dim3(512), AO_LDS, s, qkv, Wo, bo, xcopy, res_in, res_out,
                       stats_out, S, scale, B, aff_tiles);
    SSW_HIP_TRY(hipGetLastError());
    return SSW_OK;
}

}  // namespace

#ifdef SSW_DEBUG_HOOKS
ssw_status read_ao_stamps(uint64_t *out, int n_words) {
    SSW_HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ao_stamps), (size_t)n_words * 8));
    return SSW_OK;
}
#endif
ssw_status pack_attn_outproj_weight(hipStream_t s, const void *Wo, void *out) {
    hipLaunchKernelGGL(k_pack_wo, dim3(AO_D * AO_D / 8 / 256), dim3(256), 0, s, static_cast<const bf16 *>(Wo), static_cast<bf16 *>(out));
    SSW_HIP_TRY(hipGetLastError());
    return SSW_OK;
}
bool attn_outproj_supports(int S, int D, int H) { return D == AO_D && H == AO_H && S >= 1 && S <= 64; }

// qkv [B*S, 3*768] bf16; Wo: the [768, 768] out-projection weight PACKED by pack_attn_outproj_weight; bo [768].  bf16 stream (res_in == nullptr): xcopy [B*S, 768] is read,
// added to and written back in place.  f32 stream: res_in -> res_out f32 rows, xcopy receives the bf16 copy.
// stats_out [B*S][2][2]: partial (sum, sum of squares) of the new row's columns [0, 384) and [384, 768).
ssw_status launch_attn_outproj(hipStream_t s, const void *qkv, const void *Wo, const float *bo, void *xcopy,
                               const float *res_in, float *res_out, float *stats_out, int B, int S, int D, int H,
                               float scale, int affinity_row_tiles) {
    if (!attn_outproj_supports(S, D, H) || B < 1) {
        set_error("attn_outproj: S=%d D=%d H=%d unsupported (ViT-B/32: S <= 64, D = 768, H = 12)", S, D, H);
        return SSW_ERR_UNSUPPORTED;
    }
    const bf16 *q = static_cast<const bf16 *>(qkv), *w = static_cast<const bf16 *>(Wo);
    bf16 *x = static_cast<bf16 *>(xcopy);
    const bool bf = res_in == nullptr;
    // (two K-steps of fragments ahead spill at the 256 registers of an eight-wave workgroup, and one is enough:
                    //  the product waits for the L1's fill rate, not for latency -- 27 k cycles either way)
    return bf ? launch_ao<true, 1>(s, q, w, bo, x, nullptr, nullptr, stats_out, B, S, scale, affinity_row_tiles)
              : launch_ao<false, 1>(s, q, w, bo, x, res_in, res_out, stats_out, B, S, scale, affinity_row_tiles);
}

}  // namespace ssw
