// rescore.hip -- second stage of the multiscale lookup: the `avg_score` aggregation of the candidate
// images' tile scores (gfx950 / MI355X).
//
// Replaces score_frame2 + box_join as rescore_candidates drives them
// (seesaw/indices/multiscale/multiscale_index.py:112-150, 379-403; seesaw/box_utils.py:336-372):
//   for every candidate image, all tile pairs (i, j) with IoU > 0 are joined, filtered by aug_larger
//   ('all' | 'greater': zoom_j >= zoom_i | 'adjacent': zoom_j == zoom_i), and tile i's score becomes the mean,
//   over the zoom levels z present among its partners, of the score of the level-z partner overlapping it most
//   (pandas idxmax: the first maximum).  The image is represented by its first tile with the highest
//   aggregated score.
// The reference does this with a pandas self-join + two groupbys per image (milliseconds per image, ~50
// images per query).  Here: one workgroup per candidate image, boxes / zoom levels / scores of its tiles in
// LDS, one thread per tile i walking the T partners once per zoom level.  Latency-bound (T ~ 13 .. 60 tiles,
// a few hundred IoUs per tile); what it removes is the host loop and the PCIe round trip of the tile scores:
// the scores are read from the buffer the scan left in HBM.
//
// Arithmetic is the reference's, op for op, so results are bit-identical to it given identical tile scores:
//   IoU in f32 exactly as torchvision's _box_inter_union forms it on float32 boxes
//     (area = (x2-x1)*(y2-y1); wh = max(min(rb) - max(lt), 0); inter = w*h; union = (a_i + a_j) - inter),
//   the mean as pandas' float32 group_mean: Kahan-compensated f32 sum in ascending zoom level, f32 division.
// aug_weight = 'cont_weighted' (bit 2 of `aug`): softmax-of-containment weights over all partners instead (f32 like
// scipy.special.softmax on float32 input; the weighted sum runs in partner order, numpy's dot in BLAS order: 1e-6).
// Explicitly rounded intrinsics keep the compiler from contracting w*h into the union's subtraction.
#include "ssw_common.h"

// Compiled with -ffp-contract=off (csrc/Makefile): results of this file are compared bit for bit with numpy /
// scipy / torch, so every product and sum must round on its own (hipcc would contract a * b + c into an fma).

namespace ssw {
namespace {

constexpr int RS_THREADS = 256;

__device__ __forceinline__ float iou_f32(const float4 a, float area_a, const float4 b, float area_b) {
    const float w = fmaxf(__fsub_rn(fminf(a.z, b.z), fmaxf(a.x, b.x)), 0.f);
    const float h = fmaxf(__fsub_rn(fminf(a.w, b.w), fmaxf(a.y, b.y)), 0.f);
    const float inter = __fmul_rn(w, h);
    const float uni = __fsub_rn(__fadd_rn(area_a, area_b), inter);
    return __fdiv_rn(inter, uni);
}

// boxes: x1, y1, x2, y2 per row.  cand_pos[c] = image position; its rows are [row_start[p], row_start[p+1]).
// minus (optional) holds one value per candidate row, in candidate order (cand_off[c] = first), subtracted from
// the resident score (the vector2 form of MultiscaleIndex.query).
// ST = the dtype of the score column: float for the scan's scores (pandas' float32 group mean), double for the graph
// loops, which hand rescore_candidates the label-propagation output as float64 (graph_based.py:100-108; pandas' float64
// group mean is the same Kahan sum in f64).  IoUs are f32 either way (the boxes are).
template <typename ST>
__global__ __launch_bounds__(RS_THREADS) void k_avg_score(const float4 *__restrict__ boxes,
                                                          const int32_t *__restrict__ zoom,
                                                          const ST *__restrict__ scores,
                                                          const ST *__restrict__ minus_or_null,
                                                          const int64_t *__restrict__ row_start,
                                                          const int64_t *__restrict__ cand_pos,
                                                          const int64_t *__restrict__ cand_off, int aug,
                                                          ST *__restrict__ out_score,
                                                          int64_t *__restrict__ out_row) {
    extern __shared__ float4 sh4[];
    const int c = blockIdx.x;
    const int64_t p = cand_pos[c];
    const int64_t r0 = row_start[p];
    const int T = (int)(row_start[p + 1] - r0);
    float4 *sbox = sh4;                                   // [T]
    ST *sscore = reinterpret_cast<ST *>(sbox + T);        // [T]  (8-byte types first: the base is 16-byte aligned)
    ST *sagg = sscore + T;                                // [T]
    float *sarea = reinterpret_cast<float *>(sagg + T);   // [T]
    int *szoom = reinterpret_cast<int *>(sarea + T);      // [T]
    __shared__ unsigned level_mask;
    if (threadIdx.x == 0) level_mask = 0u;
    __syncthreads();
    for (int i = threadIdx.x; i < T; i += RS_THREADS) {
        const float4 b = boxes[r0 + i];
        sbox[i] = b;
        sarea[i] = __fmul_rn(__fsub_rn(b.z, b.x), __fsub_rn(b.w, b.y));
        ST s = scores[r0 + i];
        if (minus_or_null) s = s - minus_or_null[cand_off[c] + i];
        sscore[i] = s;
        const int z = zoom[r0 + i];
        szoom[i] = z;
        atomicOr(&level_mask, 1u << z);
    }
    __syncthreads();
    const unsigned levels = level_mask;
    const bool cont_weighted = (aug & 4) != 0;
    aug &= 3;
    for (int i = threadIdx.x; i < T && cont_weighted; i += RS_THREADS) {
        // aug_weight = 'cont_weighted' (multiscale_index.py:133-145): over ALL joined partners j (IoU > 0, level
        // filter), weights = softmax(containment_ij), containment = inter / area_i (box_utils.py:347-349, f32);
        // score_i = weights . score_j.  scipy.special.softmax: exp(x - max) / sum(exp(x - max)), f32 on f32 input.
        const float4 bi = sbox[i];
        const float ai = sarea[i];
        const int zi = szoom[i];
        float cmax = -1.f;
        for (int j = 0; j < T; ++j) {
            const int z = szoom[j];
            if ((aug == 1 && z < zi) || (aug == 2 && z != zi)) continue;
            const float4 bj = sbox[j];
            const float w = fmaxf(__fsub_rn(fminf(bi.z, bj.z), fmaxf(bi.x, bj.x)), 0.f);
            const float h = fmaxf(__fsub_rn(fminf(bi.w, bj.w), fmaxf(bi.y, bj.y)), 0.f);
            const float inter = __fmul_rn(w, h);
            const float v = __fdiv_rn(inter, __fsub_rn(__fadd_rn(ai, sarea[j]), inter));
            if (!(v > 0.f)) continue;
            cmax = fmaxf(cmax, __fdiv_rn(inter, ai));
        }
        if (cmax < 0.f) {  // no partner at all (not even itself: a degenerate box)
            sagg[i] = (ST)__builtin_nanf("");
            continue;
        }
        float esum = 0.f;
        for (int pass = 0; pass < 2; ++pass) {
            ST acc = 0;
            for (int j = 0; j < T; ++j) {
                const int z = szoom[j];
                if ((aug == 1 && z < zi) || (aug == 2 && z != zi)) continue;
                const float4 bj = sbox[j];
                const float w = fmaxf(__fsub_rn(fminf(bi.z, bj.z), fmaxf(bi.x, bj.x)), 0.f);
                const float h = fmaxf(__fsub_rn(fminf(bi.w, bj.w), fmaxf(bi.y, bj.y)), 0.f);
                const float inter = __fmul_rn(w, h);
                const float v = __fdiv_rn(inter, __fsub_rn(__fadd_rn(ai, sarea[j]), inter));
                if (!(v > 0.f)) continue;
                const float e = expf(__fsub_rn(__fdiv_rn(inter, ai), cmax));
                if (pass == 0)
                    esum = __fadd_rn(esum, e);
                else
                    acc = acc + (ST)__fdiv_rn(e, esum) * sscore[j];
            }
            if (pass == 1) sagg[i] = acc;
        }
    }
    for (int i = threadIdx.x; i < T && !cont_weighted; i += RS_THREADS) {
        const float4 bi = sbox[i];
        const float ai = sarea[i];
        const int zi = szoom[i];
        ST sum = 0, comp = 0;  // Kahan pair
        int groups = 0;
        for (unsigned m = levels; m != 0u; m &= m - 1u) {
            const int z = __ffs(m) - 1;  // ascending zoom level
            if (aug == 1 && z < zi) continue;
            if (aug == 2 && z != zi) continue;
            float best = 0.f;  // only IoU > 0 joins
            int bj = -1;
            for (int j = 0; j < T; ++j) {
                if (szoom[j] != z) continue;
                const float v = iou_f32(bi, ai, sbox[j], sarea[j]);
                if (v > best) {  // strict: the first maximum wins, NaN never does
                    best = v;
                    bj = j;
                }
            }
            if (bj >= 0) {
                const ST y = sscore[bj] - comp;  // (this file is compiled without fma contraction)
                const ST t = sum + y;
                comp = (t - sum) - y;
                sum = t;
                ++groups;
            }
        }
        sagg[i] = groups > 0 ? sum / (ST)groups : (ST)__builtin_nanf("");
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // first tile with the highest aggregated score (NaN skipped, as pandas' max does)
        int bi = -1;
        ST bv = 0;
        for (int i = 0; i < T; ++i) {
            const ST v = sagg[i];
            if (v != v) continue;
            if (bi < 0 || v > bv) {
                bv = v;
                bi = i;
            }
        }
        out_score[c] = bi >= 0 ? bv : (ST)__builtin_nanf("");
        out_row[c] = r0 + (bi >= 0 ? bi : 0);
    }
}

}  // namespace

size_t avg_score_lds_bytes(int max_tiles, size_t score_bytes) {
    return (size_t)max_tiles * (sizeof(float4) + 2 * score_bytes + sizeof(float) + sizeof(int));
}

template <typename ST>
static ssw_status launch_avg_score_t(const float *boxes, const int32_t *zoom, const ST *scores, const ST *minus_or_null,
                                     const int64_t *row_start, const int64_t *cand_pos, const int64_t *cand_off, int32_t m,
                                     int32_t max_tiles, int32_t aug, ST *out_score, int64_t *out_row, hipStream_t stream) {
    if (m <= 0) return SSW_OK;
    if (max_tiles > SSW_RESCORE_MAX_TILES) {
        set_error("avg_score: an image with %d tiles exceeds the %d the kernel keeps in LDS", max_tiles,
                  SSW_RESCORE_MAX_TILES);
        return SSW_ERR_UNSUPPORTED;
    }
    const size_t lds = avg_score_lds_bytes(max_tiles, sizeof(ST));
    if (lds > (size_t)64 * 1024) {  // 2048 tiles of f64 scores: 80 KB
        // per device and cheap: set on every such launch instead of caching a process-wide flag
        SSW_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_avg_score<ST>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    }
    hipLaunchKernelGGL(k_avg_score<ST>, dim3((unsigned)m), dim3(RS_THREADS), lds, stream,
                       reinterpret_cast<const float4 *>(boxes), zoom, scores, minus_or_null, row_start, cand_pos,
                       cand_off, (int)aug, out_score, out_row);
    SSW_HIP_TRY(hipGetLastError());
    return SSW_OK;
}

ssw_status launch_avg_score(const float *boxes, const int32_t *zoom, const float *scores, const float *minus_or_null,
                            const int64_t *row_start, const int64_t *cand_pos, const int64_t *cand_off, int32_t m,
                            int32_t max_tiles, int32_t aug, float *out_score, int64_t *out_row, hipStream_t stream) {
    return launch_avg_score_t<float>(boxes, zoom, scores, minus_or_null, row_start, cand_pos, cand_off, m, max_tiles, aug,
                                     out_score, out_row, stream);
}

ssw_status launch_avg_score_f64(const float *boxes, const int32_t *zoom, const double *scores,
                                const int64_t *row_start, const int64_t *cand_pos, const int64_t *cand_off, int32_t m,
                                int32_t max_tiles, int32_t aug, double *out_score, int64_t *out_row, hipStream_t stream) {
    return launch_avg_score_t<double>(boxes, zoom, scores, (const double *)nullptr, row_start, cand_pos, cand_off, m,
                                      max_tiles, aug, out_score, out_row, stream);
}

}  // namespace ssw
