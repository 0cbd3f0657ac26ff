/*
 * seesaw_hip.h -- C-ABI of libseesaw_hip.so, the MI355X (gfx950) hot path behind
 * seesaw's Python API surface.
 *
 * The reference (orm011/seesaw) has no FFI: its boundary is a duck-typed Python
 * interface whose numeric work is delegated to numpy / scipy / torch-CPU.  Every
 * entry point below replaces one of those numeric call sites; the reference
 * file:line each one stands in for is cited next to it (paths relative to the
 * reference root).  The Python mirror of the reference interface that binds
 * these symbols lives in seesaw_amd/ (see INTEGRATION.md for the ctypes stub a
 * reference maintainer would add).
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / numpy types cross this boundary;
 *   - every function returns ssw_status: 0 = OK, <0 = error, and the failing
 *     thread can read a message with ssw_last_error();
 *   - "host" pointers are ordinary process memory, "dev" pointers are HIP device
 *     memory on the handle's device (e.g. torch.Tensor.data_ptr());
 *   - a handle is bound to one device and one HIP stream and is NOT thread-safe:
 *     the same serial-per-session model the reference uses
 *     (seesaw/web/web_session_actor.py:13-16);
 *   - functions with a host-side result synchronise the handle's stream before
 *     returning; the *_dev / *_async forms only enqueue work.
 */
#ifndef SEESAW_HIP_H
#define SEESAW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t ssw_status;
#define SSW_OK 0
#define SSW_ERR_INVALID (-1)     /* bad argument                                  */
#define SSW_ERR_HIP (-2)         /* a HIP runtime call failed                     */
#define SSW_ERR_NOMEM (-3)       /* device or host allocation failed              */
#define SSW_ERR_UNSUPPORTED (-4) /* shape / size outside what the kernels support */
#define SSW_ERR_NUMERIC (-5)     /* NaN / bound violation the reference asserts on */

#define SSW_ABI_VERSION 1
#define SSW_MAX_TOPK 4096        /* largest k ssw_index_topk accepts               */

int32_t ssw_abi_version(void);
const char *ssw_last_error(void);
ssw_status ssw_device_count(int32_t *out_count);
/* name / CU count / HBM bytes of a device, for bench reports. name_cap >= 64. */
ssw_status ssw_device_info(int32_t device, char *name, int32_t name_cap, int32_t *out_cus,
                           int64_t *out_hbm_bytes);

/* ------------------------------------------------------------------------- */
/* Vector index: resident [N, d] f32 matrix + brute-force cosine scan + top-k */
/* replaces: vectors @ q ; np.argsort(-scores)                                */
/*   seesaw/indices/multiscale/multiscale_index.py:170-175 (_get_top_exact)   */
/*   seesaw/indices/multiscale/multiscale_index.py:177-199 (_get_top_dbidxs)  */
/*   seesaw/indices/coarse/coarse_index.py:57-96 (CoarseIndex.query)          */
/*   seesaw/vector_index.py:44-60 (VectorIndex.query)                         */
/* ------------------------------------------------------------------------- */
typedef struct ssw_index ssw_index;

/* dev_vectors_or_null: borrow an existing device matrix (row-major, 16-byte
 * aligned, must outlive the handle); NULL -> the handle allocates n_rows*dim*4
 * bytes itself.  dim must be a multiple of 256 and <= 1024 (CLIP: 512). */
ssw_status ssw_index_create(int32_t device, int64_t n_rows, int32_t dim,
                            const float *dev_vectors_or_null, ssw_index **out);
ssw_status ssw_index_destroy(ssw_index *idx);
/* run all of this handle's work on an existing hipStream_t (e.g. torch's current
 * stream) instead of the handle's own (non-blocking) stream.  NULL restores the own stream; to name the
 * default stream -- whose handle IS NULL, e.g. torch.cuda.current_stream().cuda_stream == 0 -- pass
 * hipStreamLegacy ((hipStream_t)1). */
ssw_status ssw_index_set_stream(ssw_index *idx, void *hip_stream);
ssw_status ssw_index_sync(ssw_index *idx);
ssw_status ssw_index_shape(const ssw_index *idx, int64_t *n_rows, int32_t *dim, int64_t *n_images);
/* device pointers of the resident matrix and of the score buffer ([n_rows] f32,
 * valid after the last scan). */
ssw_status ssw_index_device_ptrs(ssw_index *idx, void **dev_vectors, void **dev_scores);

/* copy rows [first_row, first_row+n) from host memory into the index. */
ssw_status ssw_index_upload(ssw_index *idx, const float *host_rows, int64_t first_row, int64_t n);
/* copy rows back (tests / subset()). */
ssw_status ssw_index_download(ssw_index *idx, float *host_rows, int64_t first_row, int64_t n);
/* fill the whole index on the device with the counter-based synthetic generator
 * (unit-norm rows; bit-identical to oracle.synth_rows(seed, global_first_row+i)). */
ssw_status ssw_index_fill_random(ssw_index *idx, uint64_t seed, int64_t global_first_row);

/* row -> image map (vector_meta.dbidx of the reference, multiscale_index.py:192):
 * host array [n_rows] int32, non-decreasing, values are *positions* 0..n_images-1 of
 * the distinct images in ascending dbidx order.  NULL -> identity (one vector per
 * image, CoarseIndex). */
ssw_status ssw_index_set_row2image(ssw_index *idx, const int32_t *row2image_host, int64_t n_images);

/* scores = vectors @ q   (index.score(): multiscale_index.py:284-285, coarse_index.py:37-38).
 * Leaves the scores resident on the device; out_host may be NULL. */
ssw_status ssw_index_scan(ssw_index *idx, const float *q_host, float *out_scores_host_or_null);
/* same, q already on the device, nothing copied back, no synchronisation. */
ssw_status ssw_index_scan_dev(ssw_index *idx, const float *q_dev);

/* replace the resident per-row scores with caller-supplied ones ([n_rows] f32, -inf = skip the
 * row); a following ssw_index_topk(q_host = NULL) then ranks images by them.  Used to rank by
 * label-propagation scores (KnnProp2.next_batch, seesaw/loops/graph_based.py:88-109). */
ssw_status ssw_index_load_scores(ssw_index *idx, const float *scores_host);

/* Top-k distinct images by their best-scoring row, skipping excluded images:
 * _get_top_exact + _get_top_dbidxs (multiscale_index.py:170-199).  Order: score
 * descending, ties by ascending image position.  q_host==NULL reuses the resident
 * scores of the previous scan.  excluded = image positions (any order, may repeat).
 * Outputs (host, capacity k): image position, its max score, the row that attains it
 * (lowest row on ties).  *out_count = min(k, #non-excluded images). */
ssw_status ssw_index_topk(ssw_index *idx, const float *q_host, const int64_t *excluded_images,
                          int64_t n_excluded, int32_t k, int64_t *out_images, float *out_scores,
                          int64_t *out_best_rows, int32_t *out_count);
/* device-resident form used by bench.py and the sharded index: q_dev [dim] f32;
 * the excluded set is whatever the last ssw_index_set_excluded installed; results
 * stay on the device in the handle's result buffers (ssw_index_result_ptrs).
 * Enqueues only. */
ssw_status ssw_index_set_excluded(ssw_index *idx, const int64_t *excluded_images, int64_t n_excluded);
ssw_status ssw_index_topk_dev(ssw_index *idx, const float *q_dev, int32_t k);
/* device result buffers of the last topk: keys [SSW_MAX_TOPK] u64 =
 * (orderable(score) << 32) | (0xFFFFFFFF - image), sorted descending; count [2] i32 = {number of keys,
 * overflow flag}; best_rows [SSW_MAX_TOPK] u32.  overflow != 0: more than 8192 images share the 24-bit score
 * prefix of the k-th score (duplicated vectors) and the keys are NOT the exact top-k -- rerun with
 * ssw_index_select_deep_dev (ssw_index_topk / ssw_index_topk_fetch do that by themselves). */
ssw_status ssw_index_result_ptrs(ssw_index *idx, void **dev_keys, void **dev_count,
                                 void **dev_best_rows);
/* exact selection over the resident scores for the mass-tie case (host-synchronised radix descent over the
 * full 64-bit composite key); leaves keys / count / best_rows in the same buffers, overflow cleared.
 * No reference counterpart: np.argsort sorts all N keys (multiscale_index.py:171-172). */
ssw_status ssw_index_select_deep_dev(ssw_index *idx, int32_t k);
/* read the last device-side result back (synchronises). */
ssw_status ssw_index_topk_fetch(ssw_index *idx, int32_t k, int64_t *out_images, float *out_scores,
                                int64_t *out_best_rows, int32_t *out_count);

/* scores of arbitrary rows from the resident score buffer (stage-2 rescoring of
 * the candidate images' tiles, multiscale_index.py:341-345). */
ssw_status ssw_index_gather_scores(ssw_index *idx, const int64_t *rows_host, int64_t n,
                                   float *out_scores_host);

/* the vectors of arbitrary rows, [n, dim] f32 (`index.vectors[rows]`: the labelled tiles the fitting loops read,
 * seesaw/loops/multi_reg.py:204, loops/util.py:6,11), out of the resident matrix. */
ssw_status ssw_index_gather_rows(ssw_index *idx, const int64_t *rows_host, int64_t n, float *out_host);

/* scores of arbitrary rows against another vector, in the scan's summation order
 * (`vectors[ilocs] @ vector2`, multiscale_index.py:347-349). */
ssw_status ssw_index_score_rows(ssw_index *idx, const float *q_host, const int64_t *rows_host,
                                int64_t n, float *out_scores_host);

/* ------------------------------------------------------------------------- */
/* Second stage of the multiscale lookup: `avg_score` aggregation on the device */
/* replaces: score_frame2 (both aug_weight modes) + box_join as driven by        */
/*           rescore_candidates   seesaw/indices/multiscale/multiscale_index.py */
/*           :112-150, 379-403;   seesaw/box_utils.py:336-372                   */
/* ------------------------------------------------------------------------- */
/* tile geometry of every row: boxes [n_rows, 4] f32 = x1, y1, x2, y2 (vector_meta's columns, original-image
 * pixels) and zoom_level [n_rows] i32 in [0, 31]. */
ssw_status ssw_index_set_tile_meta(ssw_index *idx, const float *boxes_host, const int32_t *zoom_host);
/* for each of m candidate images (positions as returned by ssw_index_topk): tile i's score := mean over zoom
 * levels of the score of the best-overlapping (IoU > 0, first maximum) tile of that level, levels restricted by
 * aug_larger (0 'all', 1 'greater': partner level >= own, 2 'adjacent': partner level == own); the image is
 * represented by its first tile with the highest aggregated score.  Tile scores are the ones the last scan left
 * on the device, minus minus_scores (optional; one value per candidate tile, candidates in the given order, tiles
 * in row order: the `vector2` form, multiscale_index.py:347-349).  IoU in f32 as torchvision forms it, mean as
 * pandas' float32 group mean (Kahan sum): bit-identical to the reference on identical tile scores.
 * aug_larger + 4 selects aug_weight = 'cont_weighted' (multiscale_index.py:133-145): tile i's score := softmax over
 * ALL its joined partners of the containment inter / area_i, dotted with the partners' scores (f32; 1e-6 of the
 * reference, whose dot runs in BLAS order). */
ssw_status ssw_index_rescore_avg(ssw_index *idx, const int64_t *image_positions, int32_t m, int32_t aug_larger,
                                 const float *minus_scores_or_null, float *out_scores, int64_t *out_best_rows);
/* the same aggregation over float64 tile scores that live on the device, one per index row (the output of
 * ssw_labelprop_run_resident: ssw_labelprop_device_scores) -- what the graph loops' own rescoring does with the
 * propagated scores (`fullmeta.assign(score=scores[rows])` then rescore_candidates, seesaw/loops/graph_based.py:100-108):
 * mean = pandas' float64 group mean (Kahan sum in f64), IoU in f32 as above. */
ssw_status ssw_index_rescore_avg_f64(ssw_index *idx, const double *dev_scores, const int64_t *image_positions, int32_t m,
                                     int32_t aug_larger, double *out_scores, int64_t *out_best_rows);

/* merge several sorted key lists (e.g. the all-gathered per-shard top-k of a
 * row-sharded index; keys as in ssw_index_result_ptrs but with GLOBAL image ids
 * added by the caller via id_offsets) into the global top-k.  All pointers device. */
ssw_status ssw_topk_merge_dev(int32_t device, void *hip_stream, const uint64_t *dev_keys_in,
                              int32_t n_lists, int32_t list_stride, const int32_t *dev_counts,
                              int32_t k, uint64_t *dev_keys_out, int32_t *dev_count_out);

/* The sharded query as ONE device pipeline: scan -> select -> all-gather -> merge, with no elementwise kernels around
 * the collective.  ssw_index_set_exchange_target makes the selection's last kernel also write this rank's MESSAGE into
 * dev_msg [msg_len = (with_best ? 2 : 1) * k_max + 1 words]: keys - image_offset (the low word then holds
 * 0xFFFFFFFF - global image position) at [0, k), best rows + row_offset at [k_max, k_max + k) when with_best, and
 * count | overflow << 32 in the last word (NULL detaches).  ssw_topk_merge_msgs_dev merges `world` gathered messages:
 * a list's count is the low half of its last word; the overflow flags go to dev_flags [world] (i64) and are OR-ed into
 * dev_flags_seen [1].  No reference counterpart. */
ssw_status ssw_index_set_exchange_target(ssw_index *idx, uint64_t *dev_msg_or_null, int32_t k_max, int32_t with_best,
                                         int64_t image_offset, int64_t row_offset);
ssw_status ssw_topk_merge_msgs_dev(int32_t device, void *hip_stream, const uint64_t *dev_msgs, int32_t world,
                                   int32_t k_max, int32_t with_best, int32_t k, uint64_t *dev_keys_out,
                                   int32_t *dev_count_out, int64_t *dev_flags_or_null, int64_t *dev_flags_seen_or_null);

/* The collective itself (`ssw_topk_allgather(ssw_comm *, ...)`): RCCL bound at run time (dlopen; the
 * copy torch mapped, if any).  ssw_comm_unique_id on one rank -> 128 bytes handed to every rank by the caller's own
 * means -> ssw_comm_create on every rank (collective).  ssw_topk_allgather enqueues ncclAllGather of msg_len u64 words
 * per rank on hip_stream; dev_recv [world * msg_len] is in rank order. */
typedef struct ssw_comm ssw_comm;
ssw_status ssw_comm_unique_id(void *out_id128);
ssw_status ssw_comm_create(int32_t device, const void *unique_id128, int32_t rank, int32_t world, ssw_comm **out);
ssw_status ssw_comm_destroy(ssw_comm *comm);
ssw_status ssw_topk_allgather(ssw_comm *comm, void *hip_stream, const uint64_t *dev_send, uint64_t *dev_recv,
                              int32_t msg_len);

/* per-launch device time of the dominant (scan) kernel, measured with HIP events
 * on the handle's stream.  enable=1 starts recording one event pair per scan
 * launch (up to 4096 launches); ssw_index_profile_read synchronises and returns
 * the durations in milliseconds. */
ssw_status ssw_index_profile(ssw_index *idx, int32_t enable);
ssw_status ssw_index_profile_read(ssw_index *idx, float *out_ms, int32_t cap, int32_t *out_n);

/* ------------------------------------------------------------------------- */
/* k-NN-graph label propagation                                               */
/* replaces: LabelPropagation.fit_transform  seesaw/label_propagation.py:45-79 */
/*           (scipy csr_matvec sweeps, <= 300 per refine: knn_methods.py:186)  */
/* ------------------------------------------------------------------------- */
typedef struct ssw_lp ssw_lp;

/* W: n x n CSR with sorted column indices (get_weight_matrix, knn_graph.py:31-104);
 * host arrays indptr [n+1] i64, indices [nnz] i32, data [nnz] f64.
 * weight_sum = W.sum(0) [n] f64, or NULL to have it computed in scipy's order. */
ssw_status ssw_labelprop_create(int32_t device, int64_t n, const int64_t *indptr_host,
                                const int32_t *indices_host, const double *data_host,
                                const double *weight_sum_host_or_null, ssw_lp **out);
ssw_status ssw_labelprop_destroy(ssw_lp *lp);
/* one fit_transform: prior = reg_values (NULL only with reg_lambda == 0), start = the
 * start iterate (labels are clamped into it first), label ids/values, epsilon, max_iter.
 * out_f [n] f64 = what the reference returns: the iterate that entered the converging
 * sweep, or the last sweep's output without convergence.  *out_sweeps = sweeps run.
 * SSW_ERR_NUMERIC if a sweep leaves [min(0,prior.min()), max(1,prior.max())] -- the
 * reference's assert (label_propagation.py:36-40). */
ssw_status ssw_labelprop_run(ssw_lp *lp, const double *prior_host_or_null, const double *start_host,
                             const int64_t *label_ids, const double *label_vals, int64_t n_labels,
                             double reg_lambda, double eps, int32_t max_iter, double *out_f_host,
                             int32_t *out_sweeps, int32_t *out_converged);

/* A graph laid out in a LOCALITY ORDER (label_propagation.py:30-43 is unchanged arithmetic): node `old`
 * sits at position new_of_old[old]; row r of the CSR passed to create is the row of the node at position r, its columns
 * relabelled to positions but kept in ascending ORIGINAL column id -- scipy's summation order, which the sweep follows
 * entry by entry, so results stay bit-identical -- and weight_sum in position order.  Neighbours then sit near each other
 * in the iterate and the sweep's gathers hit cache.  Every entry point keeps taking and returning original ids.
 * ssw_labelprop_set_permutation must directly follow ssw_labelprop_create; ssw_labelprop_create_ordered does both and
 * skips the column-blocked copy. */
ssw_status ssw_labelprop_set_permutation(ssw_lp *lp, const int32_t *new_of_old_host);
ssw_status ssw_labelprop_create_ordered(int32_t device, int64_t n, const int64_t *indptr_host, const int32_t *indices_host,
                                        const double *data_host, const double *weight_sum_host,
                                        const int32_t *new_of_old_host, ssw_lp **out);

/* Device-resident chaining for the ranking loop (KnnProp2: the same prior is reg_values and start iterate
 * of every call, and the result only feeds a top-k): install the prior once, propagate without moving the
 * [n] f64 vectors over PCIe, hand the scores to the index's score buffer as f32 (labelled nodes at -inf
 * when mask_labeled != 0) for ssw_index_topk(q = NULL), and fetch the f64 result only when asked for. */
ssw_status ssw_labelprop_set_prior(ssw_lp *lp, const double *prior_host);
ssw_status ssw_labelprop_run_resident(ssw_lp *lp, const int64_t *label_ids, const double *label_vals,
                                      int64_t n_labels, double reg_lambda, double eps, int32_t max_iter,
                                      int32_t *out_sweeps, int32_t *out_converged);
/* Consecutive ssw_labelprop_run_resident calls of a ranking loop (knn_methods.py:176-199: every update() restarts from
 * the same prior with a few more labels) are INCREMENTAL: the handle keeps the previous call's iterates and, per sweep,
 * the block maxima of the convergence test; a call recomputes the rows within k hops of a changed label for sweep k and
 * nothing else -- same per-row arithmetic and order as the full sweeps, so values, sweep count and the returned iterate
 * are bit-identical to a from-scratch run (label_propagation.py:45-79), at a cost independent of n.  Falls back to full
 * sweeps when the change reaches n / 8 rows, when parameters change, or when more sweeps are needed than were kept (7).
 * out8: [0] 1 = the last propagation was an incremental update (0 = full sweeps, 2 = an incremental pass over the kept
 * iterates that did not converge within them, continued by full sweeps from there), [1] sweeps as the reference counts
 * them, [2] kernel launches, [3] host synchronisations, [4] rows recomputed over all sweeps, [5] iterates kept for the
 * next call, [6] nanoseconds of host time from entry until everything was enqueued, [7] nanoseconds waited for the device
 * (incremental runs).  Environment SSW_LP_NO_INCREMENTAL=1: every call runs the full sweeps (A/B, tests). */
ssw_status ssw_labelprop_last_run_info(ssw_lp *lp, int64_t *out8);
/* "nothing to propagate yet": the installed prior itself becomes the resident result (unchanged values) and the given
 * nodes are marked labelled, so that the first rounds of a graph loop -- BaseLabelPropagationRanker.update skips the
 * propagation until a negative label exists and serves the prior (research/knn_methods.py:62-75) -- go through the
 * same device path as the later ones. */
ssw_status ssw_labelprop_prior_as_result(ssw_lp *lp, const int64_t *label_ids, int64_t n_labels);
ssw_status ssw_labelprop_fetch(ssw_lp *lp, double *out_f_host);
/* out[i] = result[rows[i]] for m nodes (host arrays): the pseudo-labels makeXy takes from the propagated scores
 * (seesaw/loops/util.py:19, `lr.current_scores()[~is_labeled][randsel]`) -- 8 m bytes back instead of 8 n. */
ssw_status ssw_labelprop_gather(ssw_lp *lp, const int64_t *rows_host, int64_t m, double *out_host);
ssw_status ssw_labelprop_scores_to_index(ssw_lp *lp, ssw_index *index, int32_t mask_labeled);
/* One feedback round of a graph loop in ONE call -- KnnProp2.refine + next_batch (seesaw/loops/graph_based.py:73-121;
 * LabelPropagationRanker2.update, seesaw/research/knn_methods.py:176-199):
 *   propagate != 0: ssw_labelprop_run_resident(label_ids, label_vals, ...);  == 0: ssw_labelprop_prior_as_result(label_ids)
 *   then ssw_labelprop_scores_to_index(mask_labeled) and ssw_index_topk(index, q = NULL, excluded_images, n_excluded, k, out_*)
 * with the results of the three calls.  When the propagation is an incremental update (every round of a session after the
 * first that propagates) the whole round is enqueued on one stream -- the scores kernel takes the converged iterate's
 * number from the device-side control block -- and the host waits ONCE, on the selection's sequence word in pinned memory
 * (ssw_labelprop_last_run_info: [3] = 1).  info[0] = 3 for propagate == 0. */
ssw_status ssw_labelprop_round(ssw_lp *lp, ssw_index *index, int32_t propagate, const int64_t *label_ids,
                               const double *label_vals, int64_t n_labels, double reg_lambda, double eps, int32_t max_iter,
                               int32_t mask_labeled, const int64_t *excluded_images, int64_t n_excluded, int32_t k,
                               int64_t *out_images, float *out_scores, int64_t *out_best_rows, int32_t *out_count,
                               int32_t *out_sweeps, int32_t *out_converged);
/* device address of the [n] f64 result of the last propagation (valid until the next run on this handle). */
ssw_status ssw_labelprop_device_scores(ssw_lp *lp, const double **out_dev_scores);

/* The symmetric k-NN weight matrix on the device: get_weight_matrix(symmetric=True) (seesaw/knn_graph.py:31-104) from
 * the directed edge list src / dst [n_edges] i32 and w [n_edges] f64 = kfun(distance) (formed by the caller on the host:
 * exp() rounds as numpy's) -> CSR with sorted indices, W_ij = (sum over the directions present) / (their number), the
 * diagonal stored as 0.  Bit-identical to the reference's arrays (one commutative addition, one division per entry).
 * SSW_ERR_UNSUPPORTED: a vertex with more than 4096 incident edges, or a vertex pair with more than two edges (repeated
 * edges: the host path then decides the sum order).  ssw_wm_fetch copies indptr [n + 1] i64, indices [nnz] i32, data [nnz]
 * f64 to the host. */
typedef struct ssw_wm ssw_wm;
ssw_status ssw_wm_build_symmetric(int32_t device, int64_t n, int64_t n_edges, const int32_t *src_host,
                                  const int32_t *dst_host, const double *w_host, ssw_wm **out, int64_t *out_nnz);
ssw_status ssw_wm_fetch(ssw_wm *m, int64_t *indptr_host, int32_t *indices_host, double *data_host);
ssw_status ssw_wm_destroy(ssw_wm *m);

/* Exact k-NN graph over the resident matrix (rows as vertices, cosine / dot similarity):
 * replaces compute_exact_knn, seesaw/knn_graph.py:170-191 (`1 - X @ X.T`, argsort, first k+1).
 * out_dst / out_score are [n_rows, k+1]: per row the k+1 best rows INCLUDING the row itself,
 * ordered by (score descending, row id ascending); scores are the bits ssw_index_scan would
 * return for that row as the query (distance = 1 - score).  out_certified [n_rows]: 1 when the
 * row's list is proven exact; rows with 0 must be recomputed with ssw_index_topk (rare: the
 * fp16 candidate pass could not separate the k+1-th neighbour from the rest).  `seed` fixes the
 * random column order of the candidate pass; results do not depend on it.  1 <= k <= 31. */
ssw_status ssw_knn_build(ssw_index *index, int32_t k, uint64_t seed, int32_t *out_dst_host,
                         float *out_score_host, uint8_t *out_certified_host);

/* K6: out [dim, dim] f64 = X' L X for the resident matrix X of `index` and the CSR matrix L held
 * by `laplacian` (an ssw_lp created from the graph Laplacian, already divided by its trace):
 * replaces `X_vectors.T @ (L @ X_vectors)`, seesaw/loops/graph_based.py:45-49. */
ssw_status ssw_xlx(ssw_index *index, ssw_lp *laplacian, double *out_host);

/* ------------------------------------------------------------------------- */
/* Online relevance-feedback update: fused loss+gradient on the GPU, L-BFGS    */
/* (strong Wolfe) driver on the host.                                          */
/* replaces: LogisticRegressionPT.fit      seesaw/logistic_regression.py:350-406 */
/*           RegModule.fit (MultiReg)      seesaw/loops/multi_reg.py:139-162    */
/*           BasicTrainer.fit / torch.optim.LBFGS closure loop                  */
/*                                         seesaw/basic_trainer.py:11-69        */
/*           ref_pairwise_rank_loss / ref_pairwise_logistic_loss               */
/*                                         seesaw/rank_loss.py:34-95            */
/* ------------------------------------------------------------------------- */
typedef struct ssw_fb ssw_fb;

enum { SSW_FB_LOGREG = 0, SSW_FB_MULTIREG = 1, SSW_FB_RANKREG = 2 };
enum { SSW_FB_LOSS_CE = 0, SSW_FB_LOSS_PAIRWISE_HINGE = 1, SSW_FB_LOSS_PAIRWISE_LOGISTIC = 2 };
enum { SSW_FB_REG_NONE = 0, SSW_FB_REG_VECTOR = 1, SSW_FB_REG_NORM = 2, SSW_FB_REG_NORM1 = 3 };

typedef struct ssw_fb_objective {
    int32_t kind;          /* SSW_FB_LOGREG | SSW_FB_MULTIREG | SSW_FB_RANKREG (RankRegressionPT:  */
                           /* cheap pairwise rank loss + reg_kind / reg_weight as LOGREG)      */
    int32_t loss_type;     /* MULTIREG: label_loss_type (multi_reg.py:34-35)                   */
    int32_t fit_intercept; /* LOGREG: extra bias parameter after the dim weights               */
    int32_t reg_kind;      /* LOGREG: regulariser (logistic_regression.py:304-330)             */
    float pos_weight;      /* LOGREG: BCE pos_weight; MULTIREG ce_loss: < 0 means 'balanced'   */
    float reg_weight;      /* LOGREG: reg_lambda / n_examples (:366)                           */
    float margin;          /* MULTIREG: rank_loss_margin                                       */
    float reg_norm_lambda; /* MULTIREG                                                         */
    float reg_data_lambda; /* MULTIREG (needs ssw_fb_set_xlx when != 0)                        */
    float reg_query_lambda;/* MULTIREG                                                         */
} ssw_fb_objective;

/* Two linear outputs sharing the rows: MultiRegModule, the scorer of the multi_reg_neg loop
 * (seesaw/loops/multi_reg_module.py:40-165, loops/multi_reg_neg.py:26-109).  W [2, dim] raw weights (row 0 the target
 * query, row 1 the confusion class); logits use the L2-normalised rows.  loss = sum_i s_i [bce(z_i0, y_i0) +
 * bce(z_i1, y_i1)] + sum_{i: y_i0 + y_i1 > 0} s_i CE(z_i, y_i) + reg_norm_lambda sum_c (cosh(log |W_c|) - 1)
 * + reg_query_lambda sum_c (1 - <W_c / |W_c|, q^>) / 2, all f32 as in the reference.  Rows come from
 * ssw_fb_set_data[_from_device] (centred), the query from ssw_fb_set_query; y2 is [n, 2] row-major.
 * ssw_fb_fit2 = one torch.optim.LBFGS(strong_wolfe).step from W_inout (BasicTrainer.fit, max_epochs = 1).
 * out_parts5: loss_norm, loss_queryreg, loss_queryreg2, vertical, horizontal.  dim <= 512. */
ssw_status ssw_fb_set_targets2(ssw_fb *fb, const float *y2_host, const float *sample_weight_or_null);
ssw_status ssw_fb_lossgrad2(ssw_fb *fb, const float *W_host, float reg_norm_lambda, float reg_query_lambda,
                            float *out_loss, float *out_grad, float *out_parts5_or_null);
ssw_status ssw_fb_fit2(ssw_fb *fb, float *W_inout, float reg_norm_lambda, float reg_query_lambda, int32_t max_iter,
                       float lr, int32_t *out_iters, int32_t *out_evals, float *out_final_loss);

ssw_status ssw_fb_create(int32_t device, int32_t dim, ssw_fb **out);
ssw_status ssw_fb_destroy(ssw_fb *fb);
/* labelled vectors X [n, dim] f32 from the host; center != 0 subtracts the column means
 * (StandardScaler(with_std=False) / X - X.mean(0)). */
ssw_status ssw_fb_set_data(ssw_fb *fb, const float *X_host, int64_t n, int32_t center);
/* PseudoLR's training set assembled on the device -- makeXy (seesaw/loops/util.py:4-23) + the weights of
 * seesaw/loops/pseudo_lr.py:42-44: rows = [the labelled rows, then for every p in `drawn` the p-th unlabelled row
 * (np.nonzero(~is_labeled)[0][p])], targets = [their labels, then the propagated scores of the drawn rows (f64 -> f32)],
 * sample weights = [real_weight ..., 1 ...]; the vectors are gathered out of the resident matrix and centred as in
 * ssw_fb_set_data.  dev_scores: ssw_labelprop_device_scores of the propagation that has just run; `drawn`: the prefix of
 * np.random.permutation(#unlabelled) (ssw_np_permutation_prefix[_dev]).  Replaces the three host passes and four copies
 * of the same construction through ssw_labelprop_gather + ssw_fb_set_data_from_device + ssw_fb_set_targets; the targets
 * exist on the device only, so the set serves the logistic objective (SSW_FB_LOGREG), which reads no target on the host. */
ssw_status ssw_fb_set_pseudo_sample(ssw_fb *fb, const float *dev_matrix, int64_t n_matrix_rows, const double *dev_scores,
                                    const int64_t *labelled_rows_sorted, const float *labelled_y, int64_t n_lab,
                                    const int64_t *drawn, int64_t n_drawn, float real_weight, int32_t center);
/* same, but the rows are gathered on the device out of a resident matrix (e.g. the index:
 * ssw_index_device_ptrs) -- `index.vectors[matchdf.index.values]`, multi_reg.py:204. */
ssw_status ssw_fb_set_data_from_device(ssw_fb *fb, const float *dev_matrix, int64_t n_matrix_rows,
                                       const int64_t *rows_host, int64_t n, int32_t center);
/* targets y [n] f32 and optional per-item sample weights [n] f32. */
ssw_status ssw_fb_set_targets(ssw_fb *fb, const float *y_host, const float *sample_weight_or_null);
/* regulariser vector / query vector (normalised inside, F.normalize) and X'LX [dim, dim]. */
ssw_status ssw_fb_set_query(ssw_fb *fb, const float *q_host);
ssw_status ssw_fb_set_xlx(ssw_fb *fb, const float *xlx_host);
/* column means removed by the last set_data (the scaler's mean_). */
ssw_status ssw_fb_get_mean(ssw_fb *fb, float *out_mu_host);
/* one closure evaluation: loss and d loss / d params at w ([dim] or [dim+1] with intercept).
 * out_parts4 (MULTIREG) = loss_norm, loss_datareg, loss_queryreg, loss_labels. */
ssw_status ssw_fb_lossgrad(ssw_fb *fb, const ssw_fb_objective *obj, const float *w_host,
                           float *out_loss, float *out_grad, float *out_parts4_or_null);
/* logits X w (+ b) of the installed rows. */
ssw_status ssw_fb_scores(ssw_fb *fb, const float *w_host, int32_t has_bias, float *out_logits);
/* one optimizer.step(closure) of LBFGS(max_iter, lr, line_search_fn='strong_wolfe') starting
 * from w_inout; returns the fitted parameters in place.  SSW_ERR_NUMERIC on NaN/Inf loss
 * (the reference raises ValueError, logistic_regression.py:398-401). */
ssw_status ssw_fb_fit(ssw_fb *fb, const ssw_fb_objective *obj, float *w_inout, int32_t max_iter,
                      float lr, int32_t *out_iters, int32_t *out_evals, float *out_final_loss);
/* *out = 1 when the last ssw_fb_fit ran as ONE kernel launch (direction updates, line search and all closure
 * evaluations inside one workgroup: labelled sets of up to 1024 rows, every objective except SSW_FB_RANKREG),
 * 0 when the host drove one evaluation at a time (larger sets; or env SSW_FB_HOST_DRIVER).  Both walk the same
 * path bit for bit. */
ssw_status ssw_fb_last_fit_on_device(const ssw_fb *fb, int32_t *out);
/* forget the installed rows, targets, query and X'LX (the allocations stay): lets one engine serve the scorers the
 * loops build anew every refine (LogisticRegressionPT(...) per round, loops/pseudo_lr.py:33, log_reg.py:22) without
 * an allocate / free cycle per round. */
ssw_status ssw_fb_reset(ssw_fb *fb);

/* Host-only.  The first k entries of numpy's legacy `np.random.permutation(n)` drawn from the MT19937 state
 * (key[624], pos) of `np.random.get_state()`, which is advanced exactly as numpy's own call advances it (write it back
 * with `np.random.set_state`).  replaces `np.random.permutation(unl.shape[0])[:sample_size]`, seesaw/loops/util.py:13
 * (makeXy: PseudoLR's draw of pseudo-labelled rows) -- same sample, 15 ms -> 5 ms at 1.56 M rows. */
ssw_status ssw_np_permutation_prefix(uint32_t *mt_key624, int32_t *mt_pos, int64_t n, int64_t k,
                                     int64_t *out_prefix);
/* The same draw (same values, same stream position afterwards) with the walk through the swaps on GPU `device`: the host
 * makes the n - 1 draws into pinned memory, the device links the steps by target with one atomic exchange each and traces
 * the k wanted positions independently (~6 short list walks each) -- 2.4 ms -> 0.7 ms for 10 000 of 1.56 M.  device < 0,
 * n < 2^18 or k > n / 16: the host-only form above. */
ssw_status ssw_np_permutation_prefix_dev(int32_t device, uint32_t *mt_key624, int32_t *mt_pos, int64_t n, int64_t k,
                                         int64_t *out_prefix);

/* The pairwise rank losses on given scores (no data matrix): per-item column sums and d(sum)/d scores.
 * replaces ref_pairwise_rank_loss / ref_pairwise_logistic_loss(aggregate='sum') and
 * ref_pairwise_rank_loss_gradient, seesaw/rank_loss.py:34-106, and the per-item normalisation of
 * RegModule._step, seesaw/loops/multi_reg.py:106-121: out_item_loss[j] = coef_j / max_inversions_j *
 * sum_i loss_ij (coef NULL = all ones; coef = max_inversions gives the raw sums), out_grad = d sum_j / d scores. */
ssw_status ssw_rank_pairwise(int32_t device, int32_t logistic, const float *target_host, const float *scores_host,
                             const float *coef_host_or_null, int32_t n, float margin, double *out_item_loss,
                             float *out_grad);

/* The sort-based rank functions as counting kernels (exact integers, ties as the reference's stable sorts):
 * quick_pairwise_gradient_zero_margin(target, scores, return_max_inversions=True) -> gradient [n] (= 2 x net
 * position change between the (target, score) order and the (score, -target) order), max_reversals [n],
 * total_pairs; seesaw/rank_loss.py:109-161.  _CheapPairwiseRankingLoss (:164-187) is |gradient| / total_pairs. */
ssw_status ssw_rank_quick_gradient(int32_t device, const float *target_host, const float *scores_host, int32_t n,
                                   float *out_grad, float *out_max_reversals_or_null, int64_t *out_total_pairs_or_null);
/* compute_inversions(labs, scores), seesaw/pairwise_rank_loss.py:24-43: in descending score order (ties by index) a
 * positive counts the negatives before it, a negative the positives after it. */
ssw_status ssw_rank_inversions(int32_t device, const uint8_t *labels_host, const float *scores_host, int32_t n,
                               int64_t *out_inversions);

/* ------------------------------------------------------------------------- */
/* L-KNN active search: two-step look-ahead value of every node                 */
/* replaces: _top_sum / _opt_expected_utility_helper_lknn2                      */
/*           seesaw/research/active_search/efficient_nonmyopic_search.py:94-205 */
/*           (model state: seesaw/loops/LKNN_model.py:76-281)                   */
/* ------------------------------------------------------------------------- */
typedef struct ssw_lknn ssw_lknn;
/* neighbors_sorted: [n, D] i32, every row ascending (np.sort(matrix.indices.reshape(-1, D))), D <= 32. */
ssw_status ssw_lknn_create(int32_t device, int64_t n, int32_t D, const int32_t *neighbors_sorted_host, ssw_lknn **out);
ssw_status ssw_lknn_destroy(ssw_lknn *h);
/* numer = numerators + gamma with -inf at labelled nodes, denom = denominators + 1 (both [n] f64);
 * top_ids_desc = the K + D nodes of highest numer/denom, best first.  value[i] = s_i (1 + E1_i) + (1 - s_i) E0_i with
 * E_y = sum of the K best scores among the other nodes after labelling i with y (only i's neighbours change).
 * Returns np.nanargmax(value) and its value; out_values (optional) receives all n values.  K <= 128. */
ssw_status ssw_lknn_top_sum(ssw_lknn *h, const double *numer_host, const double *denom_host, const int32_t *top_ids_desc,
                            int32_t K, double *out_values_or_null, int64_t *out_best_idx, double *out_best_value);

/* ------------------------------------------------------------------------- */
/* CLIP ViT-B/32 image / text towers (bf16 MFMA forward)                       */
/* replaces: transformers.CLIPModel.get_text_features                          */
/*               seesaw/models/embeddings.py:441-455 (HGWrapper.from_string)   */
/*           get_image_features + F.normalize                                  */
/*               seesaw/models/model.py:50-57 (HGFaceWrapper.forward),         */
/*               seesaw/indices/multiscale/multiscale_tools.py:187-202          */
/* ------------------------------------------------------------------------- */
typedef struct ssw_clip ssw_clip;

/* weight_blob: 72-byte header ("SSWCLIP1", then int32 v_hidden, v_layers, v_heads, v_mlp,
 * image, patch, t_hidden, t_layers, t_heads, t_mlp, t_max_positions, vocab, eos_token_id,
 * projection_dim, float layer_norm_eps, int32 reserved) followed by the f32 tensors of a
 * transformers.CLIPModel state_dict in the order seesaw_amd/models/clip.py::pack_clip_weights
 * writes them.  GEMM weights are converted to bf16 on the device. */
ssw_status ssw_clip_create(int32_t device, const void *weight_blob, size_t bytes, ssw_clip **out);
ssw_status ssw_clip_destroy(ssw_clip *clip);
/* pixel_values [b, 3, image, image] f32 (already CLIP-normalised) -> [b, projection_dim] f32;
 * normalize != 0 L2-normalises every row (HGFaceWrapper). */
ssw_status ssw_clip_embed_image(ssw_clip *clip, const float *nchw_host, int32_t b, int32_t normalize,
                                float *out_host);
/* same with device pointers, enqueued on hip_stream (NULL: the handle's stream), no sync. */
ssw_status ssw_clip_embed_image_dev(ssw_clip *clip, void *hip_stream, const float *nchw_dev, int32_t b,
                                    int32_t normalize, float *out_dev);
/* tiles [b, image, image, 3] uint8 (HWC, as cut by the multiscale tiler): batch_tx's
 * `x / 255 -> (x - mean) / std` (seesaw/indices/multiscale/multiscale_tools.py:167-183) is fused into
 * the patch gather, then the same forward pass as ssw_clip_embed_image.  Replaces batch_tx +
 * InferenceActor.__call__ (multiscale_tools.py:187-202). */
ssw_status ssw_clip_embed_tiles_u8(ssw_clip *clip, const uint8_t *tiles_hwc_host, int32_t b, int32_t normalize,
                                   float *out_host);
/* input_ids [b, seq_len] int32 (BOS ... EOS [pad]); the feature is taken at the first EOS. */
ssw_status ssw_clip_embed_text(ssw_clip *clip, const int32_t *ids_host, int32_t b, int32_t seq_len,
                               int32_t normalize, float *out_host);
ssw_status ssw_clip_sync(ssw_clip *clip);

/* Per-handle options of the towers' tile path (csrc/clip.hip, run_tower); a handle starts from the environment
 * (SSW_CLIP_BF16_STREAM=1 sets option 0, SSW_CLIP_UNFUSED_ATTN=1 option 3).  Waits for the handle's stream first.
 *   SSW_CLIP_OPT_IMAGE_ROWS_BF16 (0): residual rows of the image tower in bf16 instead of the default f32 rows (SURVEY 8
 *       a-12's arithmetic) -- the precision / throughput choice of an ingest job: 6 % more tiles per second for 4x the score
 *       error (1.5e-3 against 4e-4 on a unit query, tests/test_clip_gpu.py retrieval test); both forms meet the parity bar;
 *   SSW_CLIP_OPT_TEXT_ROWS_BF16 (1): residual rows of the text tower's batched path in bf16 (default f32);
 *   SSW_CLIP_OPT_FULL_LAST_LAYER (4): the image tower's last layer over every row (as the reference's model runs it)
 *       instead of for the pooled rows only -- the embedding reads the first row of an image and nothing else, so the
 *       default runs the last attention for row 0's query only (all keys and values), and the out-projection, fc1 and
 *       fc2 on those B rows.  The vectors differ from the full layer's by 2e-5 ... 5e-5 on unit vectors (measured at
 *       200 tiles; test bar 1e-4, the tower itself is 4e-4 from transformers' f32 model): row 0's attention output and
 *       residual row are the full layer's bit for bit, but the products on B rows select other tile kernels and split
 *       K over workgroups, so sums are taken in another order and a bf16 hidden value may round the other way;
 *   SSW_CLIP_OPT_ATTN_DIRECT (2), SSW_CLIP_OPT_ATTN_OUT_UNFUSED (3): earlier kernel forms of the image tower's attention
 *       (fragments straight from memory; attention and out-projection as two launches) kept for A/B measurements.
 * Not part of the reference's interface (its precision choice is `.half()` on the whole model, embeddings.py:433-435). */
#define SSW_CLIP_OPT_IMAGE_ROWS_BF16 0
#define SSW_CLIP_OPT_TEXT_ROWS_BF16 1
#define SSW_CLIP_OPT_ATTN_DIRECT 2
#define SSW_CLIP_OPT_ATTN_OUT_UNFUSED 3
#define SSW_CLIP_OPT_FULL_LAST_LAYER 4
ssw_status ssw_clip_set_option(ssw_clip *clip, int32_t option, int32_t value);

#ifdef __cplusplus
}
#endif
#endif /* SEESAW_HIP_H */
