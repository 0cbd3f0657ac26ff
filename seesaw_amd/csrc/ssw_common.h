// Shared host-side helpers for libseesaw_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/seesaw_hip.h"
#ifdef SSW_DEBUG_HOOKS
#include "../../include/seesaw_hip_debug.h"
#endif

// Kernel-selection switches are file-scope constants in the product library; only the lab build (-DSSW_DEBUG_HOOKS ->
// libseesaw_hip_debug.so, entry points in include/seesaw_hip_debug.h) can change them.
#ifdef SSW_DEBUG_HOOKS
#define SSW_TUNABLE
#else
#define SSW_TUNABLE const
#endif

namespace ssw {

void set_error(const char *fmt, ...);

#define SSW_HIP_TRY(expr)                                                                   \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            ssw::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                           __LINE__);                                                       \
            return (_e == hipErrorOutOfMemory) ? SSW_ERR_NOMEM : SSW_ERR_HIP;               \
        }                                                                                   \
    } while (0)

#define SSW_REQUIRE(cond, ...)           \
    do {                                 \
        if (!(cond)) {                   \
            ssw::set_error(__VA_ARGS__); \
            return SSW_ERR_INVALID;      \
        }                                \
    } while (0)

#define SSW_TRY(expr)                   \
    do {                                \
        ssw_status _s = (expr);         \
        if (_s != SSW_OK) return _s;    \
    } while (0)

// Orderable key of an f32: ascending unsigned order == ascending float order
// (-inf lowest; used with -inf for excluded images).
__host__ __device__ inline uint32_t f32_to_ord(float f) {
    uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __float_as_uint(f);
#else
    memcpy(&u, &f, 4);
#endif
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ord_to_f32(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    memcpy(&f, &u, 4);
    return f;
#endif
}

// One LDS-DMA wave-instruction: 64 lanes x 16 B from per-lane global addresses to the LDS bytes
// [lds_dst, lds_dst + 1024) in lane order (lds_dst is wave-uniform).  Issued from inline asm so that the
// compiler's wait-count bookkeeping is not disturbed: for the builtin form it degrades every later
// s_waitcnt lgkmcnt / vmcnt to (0), which serialises fragment reads and MFMAs.  The vmcnt accounting for
// these loads is done by hand at the call sites (gemm_bf16.hip, knn.hip).
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// the same with the address split into a wave-uniform base (SGPR pair) and a 32-bit per-lane byte offset
__device__ __forceinline__ void glds16s(unsigned voff, const void *sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int num_cus(int device);

// Pinned host staging for small host->device inputs (query vectors, id lists): the
// caller's buffer is copied into pinned memory on the CPU, so the async H2D copy never
// reads memory the caller may free, and no stream synchronisation is needed.
struct PinnedStage {
    void *host = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    bool pending = false;
    ssw_status push(void *dev_dst, const void *src, size_t bytes, hipStream_t stream);
    void release();
};

// ---- launchers implemented in the kernel translation units -----------------

// capi_index.hip: the two halves of ssw_index_topk(q = NULL) on a stream of the caller's (ssw_labelprop_round)
ssw_status index_enqueue_topk_resident(ssw_index *idx, hipStream_t on_stream, const int64_t *excluded_images, int64_t n_excluded,
                                       int32_t k);
ssw_status index_collect_topk(ssw_index *idx, hipStream_t on_stream, int32_t k, int64_t *out_images, float *out_scores,
                              int64_t *out_best_rows, int32_t *out_count);
int index_device(const ssw_index *idx);

// scan.hip: scores[i] = dot(X[i,:], q) in the fixed kernel order (see scan.hip).
ssw_status launch_scan(const float *X, const float *q_dev, float *scores, int64_t n, int32_t dim,
                       int device, hipStream_t stream);
ssw_status launch_score_rows(const float *X, const float *q_dev, const int64_t *rows_dev, int64_t n,
                             int32_t dim, float *out, hipStream_t stream);
#ifdef SSW_DEBUG_HOOKS
void tune_scan(int variant, int blocks_per_cu);
#endif
// knn.hip's last stage (lives in scan.hip to share the scan's summation order)
ssw_status launch_knn_rescore(const float *X, int32_t dim, const int32_t *perm, int r0, int rows, const uint64_t *buf,
                              int cap, const unsigned *cnt, const unsigned char *overflow, int M, const float *norms,
                              float scale, float maxnorm, int k1, int32_t *out_dst, float *out_score,
                              unsigned char *out_cert, hipStream_t stream);
// gemm_bf16.hip: C[M,N] = A[M,K] W[N,K]^T (bf16 in, f32 accumulate) with fused epilogue
// epi: 0 f32 | 1 +bias -> bf16 | 2 +bias, quick-GELU -> bf16 | 3 +bias +residual -> f32
ssw_status launch_gemm_bf16_nt(int epi, hipStream_t stream, const void *A, const void *W, const float *bias,
                               const float *residual, void *C, int M, int N, int K);
#ifdef SSW_DEBUG_HOOKS
void tune_gemm(int variant);
#endif
// LayerNorm folded into a product (tile path of the CLIP towers; gemm_bf16.hip explains the algebra):
//   epi 4 / 5 (consumer):  C = rstd * (A W'^T - mean * c1) + c2 [quick-GELU] -> bf16, with A = bf16(x), W' = gamma (.) W,
//                          c2 passed as `bias`, the rows' statistics as np_in partial (sum, sum of squares) pairs
//   epi 6 (producer):      C = A W^T + bias + residual -> f32, plus its bf16 copy and the partial statistics of the
//                          128-column tile, for the next consumer
struct GemmLn {
    const float *stats_in = nullptr;  // [M][np_in][2]
    int np_in = 0;
    float inv_dim = 0.f, eps = 0.f;   // 1 / (row length of the normalised vector), LayerNorm epsilon
    const float *c1 = nullptr;        // [N]
    __bf16 *xcopy = nullptr;          // producer: [M][N] bf16 copy of the f32 output
    float *stats_out = nullptr;       // producer: [M][N / 128][2]
    int64_t res_ld = 0;               // producer (f32 rows): elements between consecutive residual rows, 0 = N (round 5: the
                                      // pooled last layer adds row b S of the stack to row b of the product)
    int xcd_contig = 0;               // 128 x 128 kernel: XCD x takes a CONTIGUOUS run of row tiles instead of x, x + 8, ...
                                      // (round 6 experiment: the rows of an image then sit in one XCD's L2 for the attention launch)
};
ssw_status launch_gemm_bf16_ln(int epi, hipStream_t stream, const void *A, const void *W, const float *bias,
                               const float *residual, void *C, int M, int N, int K, const GemmLn &ln);
// out[M][N] (f32) = A W^T + bias [+ residual] for a product of few tiles: K split over `splits` workgroups, partial
// products ([splits][M][N] f32 in `partials`) added in ascending order by a second launch
ssw_status launch_gemm_splitk_f32(hipStream_t stream, const void *A, const void *W, const float *bias, const float *residual,
                                  float *out, float *partials, int M, int N, int K, int splits);
// producer epilogue 6 (f32 row + bf16 copy + per-128-column statistics) behind a split-K product of few tiles;
// splitk_choice: how many ways such a product is worth splitting (1 = not at all)
int splitk_choice(int M, int N, int K, int cus);
ssw_status launch_gemm_splitk_stats(hipStream_t stream, const void *A, const void *W, const float *bias, const float *residual,
                                    float *out, float *partials, int M, int N, int K, int splits, const GemmLn &ln);
ssw_status launch_gemm_splitk_partials(hipStream_t stream, const void *A, const void *W, float *partials, int M, int N, int K,
                                       int splits);
// attn_out.hip: attention + out-projection (+ residual, + LayerNorm partial sums) of a ViT-B/32 layer, a workgroup per image
bool attn_outproj_supports(int S, int D, int H);
ssw_status pack_attn_outproj_weight(hipStream_t stream, const void *Wo_768x768, void *out_same_size);
// affinity_row_tiles > 0 (round 6 experiment): the qkv rows were produced by the 128-row tile kernel with
// GemmLn::xcd_contig over that many row tiles; the workgroup of image i is then launched on the XCD that produced its rows
ssw_status launch_attn_outproj(hipStream_t stream, const void *qkv, const void *Wo, const float *bo, void *xcopy,
                               const float *res_in, float *res_out, float *stats_out, int B, int S, int D, int H,
                               float scale, int affinity_row_tiles = 0);
#ifdef SSW_DEBUG_HOOKS
ssw_status read_ao_stamps(uint64_t *out, int n_words);
int gemm_variant();
// gemm_pw4.hip: the persistent four-wave kernel (256 x bn tiles, bn = 256 / 192 / 128, 0 = choose); N % 128, K % 128
bool gemm_pw4_supports(int M, int N, int K);
void gemm_pw4_set_mode(int mode);  // diagnostics of tools/perf_gemm.py (0 = the kernel)
ssw_status gemm_pw4_read_diag(unsigned long long out[6], bool reset);
ssw_status gemm_pw4_read_wg(unsigned long long *out);
ssw_status launch_gemm_pw4(int epi, hipStream_t stream, const void *A, const void *W, const float *bias,
                           const float *residual, void *C, int M, int N, int K, int bn);
#endif
// rng.hip: synthetic unit-norm rows.
ssw_status launch_fill_random(float *X, int64_t n, int32_t dim, uint64_t seed, int64_t first_row,
                              hipStream_t stream);

// side outputs / inputs of the selection's last kernel for the row-sharded exchange (select.hip, k_final)
struct FinalExchange {
    uint64_t *msg_out = nullptr;   // selection: this rank's message
    uint64_t image_offset = 0;     // subtracted from the keys (globalises the image position in the low word)
    int64_t row_offset = 0;        // added to the best rows
    int32_t k_max = 0, with_best = 0, msg_len = 0;
    int32_t from_msgs = 0;         // merge: the input lists are messages
    long long *flags_out = nullptr, *flags_seen = nullptr;
    // few images (<= the sort's capacity): the selection is this one kernel over all of them
    const float *values_all = nullptr;
    int64_t m_all = 0;
    const uint32_t *excl = nullptr;
    // ... which can also take the per-image maximum (values_all = row scores, row_start set; fills img_score / img_best),
    // the excluded ids as a list (device-visible pinned memory) and write the packed result into pinned host memory,
    // ending with the release of host_seq into its header word 3 (the host spins on it)
    const int64_t *row_start = nullptr;
    float *img_score = nullptr;
    uint32_t *img_best = nullptr;
    const int64_t *excl_ids = nullptr;
    int64_t n_excl = 0;
    unsigned host_seq = 0;
    int32_t sampled = 0;  // the threshold came from a sample: too few candidates is a failure, reported as overflow
};

// select.hip: exact top-k of per-image best scores.
struct SelectWorkspace {
    FinalExchange xchg;             // set by ssw_index_set_exchange_target
    // all device pointers
    uint32_t *hist1 = nullptr;      // [4096]
    uint32_t *hist2 = nullptr;      // [4096]
    uint32_t *state = nullptr;      // [16] see select.hip
    uint64_t *cand = nullptr;       // [SELECT_CAND_CAP]
    uint64_t *out_keys = nullptr;   // [SSW_MAX_TOPK]
    int32_t *out_count = nullptr;   // [1]
    uint32_t *out_best = nullptr;   // [SSW_MAX_TOPK]
    unsigned char *packed = nullptr;  // [16 + 12 k] host mirror layout: header, keys[k], best[k]
    float *img_score = nullptr;     // [n_images]   (only when row2image is set)
    uint32_t *img_best = nullptr;   // [n_images]
    uint32_t *excl_bits = nullptr;  // [(n_images+31)/32]
    int64_t *excl_ids = nullptr;    // device copy of the installed excluded-id list
    PinnedStage excl_stage;
    int64_t excl_ids_cap = 0;
    int64_t n_excluded_distinct = 0;  // distinct excluded images currently installed
    bool excl_dirty = false;          // bitmap currently has bits set
    std::vector<int64_t> excl_installed;  // the installed set, sorted: a new list costs its difference from this one
    // when set, the next selection writes its packed result here (pinned host memory, device view) and releases
    // host_seq into header word 3 instead of filling `packed`; cleared by the launch
    unsigned char *host_packed = nullptr;
    unsigned host_seq = 0;
};

#ifdef SSW_DEBUG_HOOKS
void tune_select(bool sampled);
#endif
ssw_status select_alloc(SelectWorkspace &ws, int64_t n_rows, int64_t n_images, bool has_map);
void select_free(SelectWorkspace &ws);
// install the excluded set (host ids) into ws.excl_bits; counts distinct ids.
ssw_status select_set_excluded(SelectWorkspace &ws, int64_t n_images, const int64_t *ids_host,
                               int64_t n, hipStream_t stream);
// per-image max over contiguous row ranges (row_start [n_images+1]).
ssw_status launch_image_max(const float *scores, const int64_t *row_start, int64_t n_images,
                            float *img_score, uint32_t *img_best, hipStream_t stream);
// top-k over values[m] (f32), skipping ids whose excl bit is set. Results in ws.out_*.
ssw_status launch_select_topk(SelectWorkspace &ws, const float *values, int64_t m,
                              const uint32_t *best_rows_or_null, int32_t k, int device,
                              hipStream_t stream);
// small index (n_images <= SELECT_SMALL_IMAGES): per-image max, exclusion by id list, selection and the packed result
// into pinned host memory in ONE launch; the host waits for header word 3 == seq (see FinalExchange)
constexpr int64_t SELECT_SMALL_IMAGES = 8192;
ssw_status launch_select_small(SelectWorkspace &ws, const float *row_scores, const int64_t *row_start_or_null,
                               int64_t n_images, const int64_t *excl_ids_mapped, int64_t n_excl, int32_t k,
                               unsigned char *packed_mapped, unsigned seq, hipStream_t stream);
// the fast path flags (out_count[1]) a 24-bit prefix bin with more candidates than the final
// sort can take (massive exact ties); the caller then reruns the selection on the deep path.
ssw_status launch_select_topk_deep(SelectWorkspace &ws, const float *values, int64_t m,
                                   const uint32_t *best_rows_or_null, int32_t k, int device,
                                   hipStream_t stream);
ssw_status launch_merge_topk(const uint64_t *keys_in, int32_t n_lists, int32_t list_stride,
                             const int32_t *counts, int32_t k, uint64_t *keys_out,
                             int32_t *count_out, hipStream_t stream);
ssw_status launch_merge_msgs(const uint64_t *msgs, int32_t world, int32_t k_max, int32_t with_best, int32_t k,
                             uint64_t *keys_out, int32_t *count_out, long long *flags_out, long long *flags_seen,
                             hipStream_t stream);
ssw_status launch_gather_f32(const float *src, const int64_t *idx_dev, int64_t n, float *dst,
                             hipStream_t stream);

constexpr int SSW_RANK_MAX_ITEMS = 65536;    // rank.hip: items of one counting launch (O(n^2) compares)

// rank.hip: quick zero-margin pairwise gradient on device-resident targets / scores (feedback engine)
ssw_status launch_rank_quick(const float *target_dev, const float *scores_dev, int n, float *grad_dev,
                             float *maxrev_dev_or_null, unsigned long long *total_dev_or_null, hipStream_t stream);

// rescore.hip: avg_score aggregation of candidate images' tiles (score_frame2 / box_join).
constexpr int SSW_RESCORE_MAX_TILES = 2048;  // tiles of one image held in LDS (28 B each)
constexpr int SSW_RESCORE_MAX_ZOOM = 31;     // zoom levels index a 32-bit presence mask
ssw_status launch_avg_score(const float *boxes, const int32_t *zoom, const float *scores, const float *minus_or_null,
                            const int64_t *row_start, const int64_t *cand_pos, const int64_t *cand_off, int32_t m,
                            int32_t max_tiles, int32_t aug, float *out_score, int64_t *out_row, hipStream_t stream);
// the same aggregation over float64 scores that live on the device (label-propagation output)
ssw_status launch_avg_score_f64(const float *boxes, const int32_t *zoom, const double *scores,
                                const int64_t *row_start, const int64_t *cand_pos, const int64_t *cand_off, int32_t m,
                                int32_t max_tiles, int32_t aug, double *out_score, int64_t *out_row, hipStream_t stream);

}  // namespace ssw
