/* seesaw_hip_debug.h -- lab-bench entry points of libseesaw_hip_debug.so.
 *
 * NOT part of the product ABI.  The product library (libseesaw_hip.so, include/seesaw_hip.h) is compiled without
 * -DSSW_DEBUG_HOOKS: its kernel-selection switches are constants and none of the symbols below exist in it.  The lab
 * build compiles the same sources with -DSSW_DEBUG_HOOKS and adds csrc/debug_hooks.hip and csrc/gemm_pw4.hip; tests and
 * tools that compare kernel variants, feed single kernels with chosen operands or read intermediate state load that
 * library instead (seesaw_amd._lib.debug_hooks()).  The switches are process-global and not thread-safe: one test at a
 * time.  Nothing here has a counterpart in the reference.
 */
#ifndef SEESAW_HIP_DEBUG_H
#define SEESAW_HIP_DEBUG_H

#include "seesaw_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* tuning hook (tools/sweep_scan.py): pick the scan kernel's schedule variant for dim=512
 * (0 u4, 1 u4+nt, 2 u8, 3 u8+nt, 4 u2+nt; -1 = default) and cap its resident blocks per CU
 * (0 = no cap, -1 = default).  Indexes under 65 536 rows run a latency-shaped kernel (8 rows in flight per wave, query
 * through LDS) unless a variant is named; -2 = the default streaming variant at every size.  All produce identical bits. */
ssw_status ssw_tune_scan(int32_t variant, int32_t blocks_per_cu);

/* ssw_index_topk on an index of <= 8192 images / 65536 rows and <= 8192 excluded ids runs as three launches (query staged
 * through a kernel argument; scan; per-image max + exclusion + selection in one workgroup) with the ids and the result in
 * pinned memory the device maps -- no copies, no stream wait (flag bit 0).  From 2^24 values on and k <= 2048 the selection's threshold comes from a
 * 1-in-16 sample instead of two full histogram passes (flag bit 1; exact all the same: a sample that leaves fewer than k
 * candidates raises the overflow word and the deep path runs).  Default 3; tests switch the forms off to compare. */
ssw_status ssw_tune_topk(int32_t flags);

/* Kernel A/B harness for the towers' bf16 GEMM (C[M,N] = A[M,K] W[N,K]^T + epilogue `epi`, see
 * csrc/gemm_bf16.hip): runs `variant` on seeded operands, reports ms per launch over `iters`
 * launches and the max |difference| to variant 0.  Not part of the reference's interface. */
ssw_status ssw_debug_gemm(int32_t M, int32_t N, int32_t K, int32_t epi, int32_t variant, int32_t iters,
                          float *out_ms, float *out_maxdiff);
/* Selects the GEMM variant the towers use (0 register-staged, 2 LDS-DMA ring with 4 waves per tile,
 * 14 the same with 8 waves per tile, 7 256-row pipelined, 9 the 8-wave 256 x 256 tile, 20-23 the persistent
 * four-wave kernel of csrc/gemm_pw4.hip with its column tile chosen / 256 / 192 / 128). */
ssw_status ssw_tune_gemm(int32_t variant);
/* Diagnostics of csrc/gemm_pw4.hip for tools/perf_gemm.py: mode 0 the kernel, 1 cycle stamps (out6 = cycles in the
 * mid-step wait + barrier, cycles in K-steps, K-steps, waves, s_memtime and s_memrealtime ticks per kernel; read and reset), 2-4 ablations (no LDS-DMA / no MFMA /
 * no fragment reads inside the loop: wrong results, timing only). */
ssw_status ssw_debug_gemm_pw4_mode(int32_t mode, uint64_t *out6_or_null);
/* mode 1's per-workgroup record of the last launch: [1024][4] = start, end (100-MHz ticks), HW_ID, XCC_ID; then mode 5's
 * [20] = cycles per interleave group (16), wait + barrier, sub-stages (summed over waves; read and reset) */
ssw_status ssw_debug_gemm_pw4_wg(uint64_t *out4116);

/* ONE product of the towers' bf16 GEMM on the caller's operands, everything it writes returned (tests/test_gemm_gpu.py
 * compares every shipped epilogue with a torch f32 matmul + the same epilogue).  All pointers are HOST memory; bf16
 * values travel as uint16 bit patterns.  epi (csrc/gemm_bf16.hip, enum Epilogue):
 *   0 C = A W^T                       -> C f32          4 LayerNorm folded (GemmLn): rstd (A W'^T - mean c1) + c2 -> C bf16
 *   1 + bias                          -> C bf16         5 the same + quick-GELU                                   -> C bf16
 *   2 + bias, quick-GELU              -> C bf16         6 + bias + residual -> C f32, xcopy = bf16(C), stats_out
 *   3 + bias + residual (f32)         -> C f32          7 xcopy += A W^T + bias in place (bf16 stream), stats_out
 * A [M,K], W [N,K] (K-contiguous, nn.Linear's layout), bias / c2 [N], residual [M,N] f32, xcopy [M,N] bf16,
 * stats_in [M][np_in][2] partial (sum, sum of squares) of the un-normalised f32 rows, c1 [N], stats_out [M][N/128][2].
 * variant: the kernel (ssw_tune_gemm's numbers; -1 = the library's default choice).
 * Three more forms take another meaning of `variant`: 8 = the split-K product (launch_gemm_splitk_f32; variant = splits);
 * 9 = epilogue 6 behind a split-K product (launch_gemm_splitk_stats: the text tower's fc2; variant = splits); 10 = epilogue 6
 * with the residual rows at a stride (GemmLn::res_ld: the pooled last layer's out-projection; variant = S, residual holds
 * M * S rows and row m * S is added to row m). */
ssw_status ssw_debug_gemm_run(int32_t epi, int32_t variant, int32_t M, int32_t N, int32_t K, const uint16_t *A_bf16,
                              const uint16_t *W_bf16, const float *bias_or_c2, const float *residual_or_null,
                              uint16_t *xcopy_inout_or_null, const float *stats_in_or_null, int32_t np_in,
                              const float *c1_or_null, float inv_dim, float eps, void *C_out_or_null,
                              float *stats_out_or_null);

/* The image tower's fused attention + out-projection launch (csrc/attn_out.hip) on the caller's operands.  qkv
 * [B*S, 3*768] bf16 (q | k | v), Wo [768,768] bf16 as nn.Linear holds it (packed here), bo [768]; bf16 stream
 * (res_in NULL): xcopy [B*S,768] bf16 is read, added to and returned; f32 stream: res_in -> res_out [B*S,768] f32 and
 * xcopy returns the bf16 copy.  stats_out [B*S][2][2]: partial (sum, sum of squares) of the new row's column halves. */
ssw_status ssw_debug_attn_out_run(int32_t B, int32_t S, const uint16_t *qkv_bf16, const uint16_t *Wo_bf16, const float *bo,
                                  uint16_t *xcopy_inout, const float *res_in_or_null, float *res_out_or_null,
                                  float *stats_out, float scale);
/* s_memtime stamps of the last launch of that kernel under SSW_AO_STAMPS=1: out[wg * 32 + slot], slots 0..4 = start, after
 * attention, after the product, after the stores, end; 8 + 2p / 9 + 2p = head pair p staged / computed */
ssw_status ssw_debug_attn_out_stamps(uint64_t *out, int32_t n_words);

/* Keep the residual rows behind transformer layer `layer` of tower `tower` (0 image, 1 text; layer -1 = off) of the
 * handle's next forward passes, as f32 [rows][hidden]; ssw_clip_debug_tap_read waits for the stream and returns them
 * (tests/test_clip_gpu.py compares every layer with transformers' output_hidden_states). */
ssw_status ssw_clip_debug_tap(ssw_clip *clip, int32_t tower, int32_t layer);
ssw_status ssw_clip_debug_tap_read(ssw_clip *clip, float *out_host_or_null, int64_t cap_floats, int64_t *out_rows,
                                   int32_t *out_dim);

#ifdef __cplusplus
}
#endif
#endif /* SEESAW_HIP_DEBUG_H */
