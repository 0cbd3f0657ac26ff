// labelprop.hip -- k-NN-graph label propagation sweeps (gfx950 / MI355X)
//
// Replaces LabelPropagation._step / fit_transform of the reference
// (seesaw/label_propagation.py:30-79):
//     weighted = W @ f_old + reg_lambda * reg_values          (scipy csr_matvec, f64)
//     f_new    = weighted / (W.sum(0) + reg_lambda)
//     f_new[label_ids] = label_values
//     stop when max((f_new - f_old)^2) < epsilon, RETURNING f_old (the iterate that
//     entered the converging sweep); without convergence, after max_iter sweeps, the
//     output of the last sweep.
//
// Roofline: HBM / L2-bound integer+f64 streaming.  Per sweep the CSR arrays are read once
// (12 B per non-zero: f64 weight + i32 column) plus ~40 B per node (f_old gather target,
// prior, normaliser, f_new); the f_old gathers hit L2 / Infinity Cache (N x 8 B <= 12.5 MB
// at 1.56 M nodes).  There is nothing to feed MFMA: ~20 non-zeros per row.
//
// Kernel shape ("CSR-stream"): a 256-thread workgroup owns 256 consecutive rows.  The
// workgroup's contiguous slice of non-zeros is streamed with fully coalesced loads, each
// product w * f_old[col] is formed once (one f64 rounding, no fma) and parked in LDS; then
// every lane adds up ITS row's products from LDS sequentially in ascending position.  That
// is exactly scipy's `sum += Ax[jj] * Xx[Aj[jj]]` order, so a sweep is BIT-EXACT against
// the CPU oracle (and iteration counts of the early exit match by construction), while
// global traffic stays coalesced.
// The convergence test runs on the device (block max -> u64 atomicMax of the non-negative
// f64 bit pattern; a 1-thread check kernel flips a `done` flag that turns the remaining
// enqueued sweeps into no-ops), so the host polls once per batch of sweeps, not per sweep.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <vector>

#include "ssw_common.h"

// Compiled with -ffp-contract=off (csrc/Makefile): results of this file are compared bit for bit with numpy /
// scipy / torch, so every product and sum must round on its own (hipcc would contract a * b + c into an fma).

namespace ssw {
namespace {

constexpr int LP_BLOCK = 256;
constexpr int LP_CHUNK = 4096;  // products staged in LDS per pass (32 KiB)
constexpr int LP_INFLIGHT = 16;  // index -> gather chains per thread in flight in the plain sweep (4: 0.140 ms, 16: 0.131 ms on the re-ordered 1.56 M-node graph)
constexpr int64_t LP_BLOCKED_ABOVE_BYTES = 5 << 20;  // iterates larger than this take the column-blocked sweep
constexpr int64_t LP_SLICE_BYTES = 2 << 20;  // f_old bytes one pass of the column-blocked sweep gathers from (L2 = 4 MiB per XCD)

struct LpState {           // device-resident control block
    unsigned long long maxdiff_bits;  // max (f_new - f_old)^2 of the current sweep
    int done;                         // 1 once converged
    int sweeps;                       // sweeps actually executed
    int result_buf;                   // which of the two f buffers holds the answer
    int bound_violation;              // reference asserts (label_propagation.py:36-40)
    int pad[3];
};

__global__ __launch_bounds__(LP_BLOCK) void k_lp_sweep(
    int64_t n, const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
    const double *__restrict__ data, const double *__restrict__ wsum,
    const double *__restrict__ prior, const double *__restrict__ f_old, double *__restrict__ f_new,
    const unsigned char *__restrict__ is_label, const double *__restrict__ label_val, double lambda,
    double low_bound, double high_bound, LpState *__restrict__ st) {
    __shared__ double prod[LP_CHUNK];
    __shared__ double red[LP_BLOCK / 64];
    if (st->done) return;
    const int t = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * LP_BLOCK;
    const int64_t row = row0 + t;
    const int64_t rend = min(row0 + LP_BLOCK, n);
    const int64_t p_begin = indptr[row0], p_end = indptr[rend];
    int64_t my_lo = 0, my_hi = 0;
    // the row's own operands of the last step are requested now, not after the sum: one round trip less per workgroup
    double e_prior = 0.0, e_wsum = 1.0, e_label = 0.0, e_old = 0.0;
    unsigned char e_is_label = 0;
    if (row < n) {
        my_lo = indptr[row];
        my_hi = indptr[row + 1];
        e_prior = prior[row];
        e_wsum = wsum[row];
        e_is_label = is_label[row];
        e_label = label_val[row];
        e_old = f_old[row];
    }
    double sum = 0.0;
    for (int64_t base = p_begin; base < p_end; base += LP_CHUNK) {
        const int64_t lim = min(base + LP_CHUNK, p_end);
        // LP_INFLIGHT independent index -> gather chains per thread in flight (the gathers of f_old are what the
        // sweep waits for: random 8-byte reads, one cache line each)
        for (int64_t p = base + t; p < lim; p += LP_INFLIGHT * LP_BLOCK) {
            int32_t col[LP_INFLIGHT];
            double w[LP_INFLIGHT], f[LP_INFLIGHT];
#pragma unroll
            for (int u = 0; u < LP_INFLIGHT; ++u) {
                const int64_t q = p + (int64_t)u * LP_BLOCK;
                col[u] = q < lim ? indices[q] : 0;
                w[u] = q < lim ? data[q] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < LP_INFLIGHT; ++u) f[u] = f_old[col[u]];
#pragma unroll
            for (int u = 0; u < LP_INFLIGHT; ++u) {
                const int64_t q = p + (int64_t)u * LP_BLOCK;
                if (q < lim) prod[q - base] = __dmul_rn(w[u], f[u]);
            }
        }
        __syncthreads();
        const int64_t lo = max(my_lo, base), hi = min(my_hi, lim);
        for (int64_t p = lo; p < hi; ++p) sum = __dadd_rn(sum, prod[p - base]);
        __syncthreads();
    }
    double d2 = 0.0;
    if (row < n) {
        const double weighted = __dadd_rn(sum, __dmul_rn(lambda, e_prior));
        double v = weighted / __dadd_rn(e_wsum, lambda);
        if (!(v >= low_bound) || !(v <= high_bound)) st->bound_violation = 1;
        if (e_is_label) v = e_label;
        f_new[row] = v;
        const double d = __dadd_rn(v, -e_old);
        d2 = __dmul_rn(d, d);
    }
    // block max of d2 (non-negative or NaN; NaN compares false and is caught by the bounds)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) d2 = fmax(d2, __shfl_xor(d2, off, 64));
    if ((t & 63) == 0) red[t >> 6] = d2;
    __syncthreads();
    if (t == 0) {
        double m = red[0];
        for (int i = 1; i < LP_BLOCK / 64; ++i) m = fmax(m, red[i]);
        atomicMax(&st->maxdiff_bits, (unsigned long long)__double_as_longlong(m));
    }
}

// Column-blocked form of the sweep for graphs whose iterate does not fit an XCD's L2 (4 MiB).
//
// Measured on the 1.56 M-node graph (profiles/r02_labelprop_pmc.csv): the plain sweep's 21.8 M gathers of f_old hit
// L2 27 % of the time (the 12.5 MB iterate against 4 MiB of L2) and the other 18 M leave as fabric requests
// (TCC_EA0_RDREQ), one line per 8 useful bytes: the sweep is bound by random lines from the Infinity Cache, not by
// its 300 MB of streams.  Here the columns are cut into slices of <= LP_SLICE_BYTES of f_old and the non-zeros are
// stored slice-major (slice, row, column): while a slice is being worked on, the gathers come from f_old[slice],
// which stays in every XCD's L2 after the first touch (hit rate 85 %, fabric requests / 4.8).  Since columns ascend
// within a row, "slice 0's terms, then slice 1's, ..." IS scipy's left-to-right order: the result is bit-identical
// to the unblocked kernel and to the CPU.  The streams use non-temporal loads so they do not evict the slice.
//
// ONE launch per sweep: a workgroup owns LP_GROUPS x 256 consecutive rows for the whole sweep and walks the slices
// in order, thread t keeping the running sums of its LP_GROUPS rows (t, t + 256, ...) in registers.  LP_GROUPS is
// chosen so that the whole grid is resident at once (<= 4 workgroups per CU): all workgroups do the same amount
// of work per slice, so the grid moves through the slices together and the slice being gathered from is the one
// in L2.  Measured alternatives at 1.56 M nodes (plain sweep 0.310 ms): this kernel 0.165 ms; one launch per
// slice with the running sums carried through HBM 0.21 ms; 4 / 8 / 12 row groups per workgroup 0.21 / 0.17 / 0.21 ms;
// a three-deep software pipeline of the stages (next stage's non-zeros and row offsets in flight during the
// gather, counted waits verified in the ISA) 0.20-0.22 ms -- SLOWER: the gathers already run at about half of the
// L2's line rate (21.8 M lines of 128 B in 165 us = 17 TB/s of its 34.5), more loads in flight only add pressure.
template <int LP_GROUPS, int FCHUNK>
__global__ __launch_bounds__(LP_BLOCK) void k_lp_sweep_fused(
    int64_t n, int nblk, const uint32_t *__restrict__ bptr /* [nblk][n+1] */, const int64_t *__restrict__ base,
    const int32_t *__restrict__ indices, const double *__restrict__ data, const double *__restrict__ wsum,
    const double *__restrict__ prior, const double *__restrict__ f_old, double *__restrict__ f_new,
    const unsigned char *__restrict__ is_label, const double *__restrict__ label_val, double lambda,
    double low_bound, double high_bound, LpState *__restrict__ st) {
    __shared__ double prod[FCHUNK];
    __shared__ double red[LP_BLOCK / 64];
    if (st->done) return;
    const int t = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * (LP_GROUPS * LP_BLOCK);
    double sum[LP_GROUPS];
#pragma unroll
    for (int g = 0; g < LP_GROUPS; ++g) sum[g] = 0.0;
    for (int b = 0; b < nblk; ++b) {
        const uint32_t *bp = bptr + (size_t)b * (size_t)(n + 1);
        const int32_t *ind = indices + base[b];
        const double *dat = data + base[b];
#pragma unroll
        for (int g = 0; g < LP_GROUPS; ++g) {
            const int64_t row0 = r0 + (int64_t)g * LP_BLOCK;
            if (row0 >= n) continue;  // uniform across the workgroup
            const int64_t row = row0 + t;
            const int64_t rend = min(row0 + LP_BLOCK, n);
            const int64_t p_begin = bp[row0], p_end = bp[rend];
            int64_t my_lo = 0, my_hi = 0;
            if (row < n) {
                my_lo = __builtin_nontemporal_load(&bp[row]);
                my_hi = __builtin_nontemporal_load(&bp[row + 1]);
            }
            double acc = sum[g];
            for (int64_t cb = p_begin; cb < p_end; cb += FCHUNK) {
                const int64_t lim = min(cb + FCHUNK, p_end);
                for (int64_t p = cb + t; p < lim; p += 4 * LP_BLOCK) {
                    int32_t col[4];
                    double w[4], f[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int64_t q = p + (int64_t)u * LP_BLOCK;
                        col[u] = q < lim ? __builtin_nontemporal_load(&ind[q]) : -1;
                        w[u] = q < lim ? __builtin_nontemporal_load(&dat[q]) : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) f[u] = col[u] >= 0 ? f_old[col[u]] : 0.0;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int64_t q = p + (int64_t)u * LP_BLOCK;
                        if (q < lim) prod[q - cb] = __dmul_rn(w[u], f[u]);
                    }
                }
                __syncthreads();
                const int64_t lo = max(my_lo, cb), hi = min(my_hi, lim);
                for (int64_t p = lo; p < hi; ++p) acc = __dadd_rn(acc, prod[p - cb]);
                __syncthreads();
            }
            sum[g] = acc;
        }
    }
    double d2 = 0.0;
#pragma unroll
    for (int g = 0; g < LP_GROUPS; ++g) {
        const int64_t row = r0 + (int64_t)g * LP_BLOCK + t;
        if (row < n) {
            const double weighted = __dadd_rn(sum[g], __dmul_rn(lambda, prior[row]));
            double v = weighted / __dadd_rn(wsum[row], lambda);
            if (!(v >= low_bound) || !(v <= high_bound)) st->bound_violation = 1;
            if (is_label[row]) v = label_val[row];
            f_new[row] = v;
            const double d = __dadd_rn(v, -f_old[row]);
            d2 = fmax(d2, __dmul_rn(d, d));
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) d2 = fmax(d2, __shfl_xor(d2, off, 64));
    if ((t & 63) == 0) red[t >> 6] = d2;
    __syncthreads();
    if (t == 0) {
        double m = red[0];
        for (int i = 1; i < LP_BLOCK / 64; ++i) m = fmax(m, red[i]);
        atomicMax(&st->maxdiff_bits, (unsigned long long)__double_as_longlong(m));
    }
}

// after sweep number `sweep_index` (1-based) that read buffer `src`: decide convergence
__global__ void k_lp_check(LpState *st, double eps, int src, int dst) {
    if (st->done) return;
    const double m = __longlong_as_double((long long)st->maxdiff_bits);
    st->sweeps += 1;
    if (m < eps) {
        st->done = 1;
        st->result_buf = src;  // converged: the reference returns the iterate that ENTERED this sweep
    } else {
        st->result_buf = dst;  // `old_fvalues = new_fvalues`: the output of this sweep
    }
    st->maxdiff_bits = 0ull;
}

__global__ void k_lp_apply_labels(double *f, unsigned char *is_label, double *label_val,
                                  const int64_t *ids, const double *vals, int64_t n_labels) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_labels) return;
    const int64_t r = ids[i];
    is_label[r] = 1;
    label_val[r] = vals[i];
    f[r] = vals[i];
}

// scores (f64) -> the index's f32 score buffer, labelled nodes at -inf: what the ranking loop hands to the
// top-k selection (graph_based.py:96-101: `scores[is_labeled] = -inf`, knn_methods.py:163-172)
__global__ void k_lp_scores_f32(const double *__restrict__ f, const unsigned char *__restrict__ is_label_or_null,
                                int64_t n, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (is_label_or_null && is_label_or_null[i]) ? -INFINITY : (float)f[i];
}

// ---------------------------------------------------------------------------------------
// Incremental propagation (round 5, ssw_labelprop_run_resident).  Every call of the ranking loop starts from the SAME
// installed prior and differs from the call before it in a handful of labels, so row i of iterate k can differ from
// the previous call's only if a label changed within k hops of i.  The handle keeps the previous call's iterates
// F[0..s] and, per sweep, the maximum of (F[k] - F[k-1])^2 over each block of 256 nodes; a call recomputes the rows
// whose inputs changed (the frontier is grown on the device through the transposed pattern: k_inc_expand), the maxima of
// the blocks they sit in, and the per-sweep maximum over all blocks -- the same per-row arithmetic in the same order as the full sweeps
// (products formed once, added in ascending position), so every value, the sweep count and the returned iterate
// are what the full sweeps give, bit for bit (tests/test_labelprop_gpu.py), at a cost that does not depend on n.
// ---------------------------------------------------------------------------------------
// Control block of an incremental call (device): list lengths per sweep and the overflow flag
struct LpInc {
    int total;                // rows in `mem` so far (grows as the frontier is expanded)
    int blocks;               // blocks in `blk` so far
    int overflow;             // the frontier did not fit `cap` rows: the call is redone with full sweeps
    int bound_violation;      // a recomputed value left the reference's bounds (label_propagation.py:36-40)
    int m[9];                 // m[k]: rows to recompute for sweep k = mem[0 .. m[k]); m[0] = the changed labels
    double level_max[8];      // per sweep: max over ALL blocks of 256 nodes of (F[k] - F[k-1])^2
};

// the label changes of an incremental call: ids [0, n_set) become labelled with vals, ids [n_set, n_set + n_unset) lose
// their label; every changed node enters the frontier list
__global__ void k_inc_seed(const int64_t *__restrict__ ids, const double *__restrict__ vals, int n_set, int n_unset,
                           const double *__restrict__ prior, double *__restrict__ f0, unsigned char *__restrict__ is_label,
                           double *__restrict__ label_val, uint32_t *__restrict__ stamp, uint32_t ep, int32_t *__restrict__ mem,
                           const int64_t *__restrict__ all_ids, const double *__restrict__ all_vals, int n_all,
                           int64_t *__restrict__ ids_out, double *__restrict__ vals_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_all) {  // the device-side list of every installed label (what the other entry points clear by)
        ids_out[i] = all_ids[i];
        vals_out[i] = all_vals[i];
    }
    if (i >= n_set + n_unset) return;
    const int64_t r = ids[i];
    if (i < n_set) {
        is_label[r] = 1;
        label_val[r] = vals[i];
        f0[r] = vals[i];
    } else {
        is_label[r] = 0;
        f0[r] = prior[r];
    }
    stamp[r] = ep;   // (the changed ids are distinct)
    mem[i] = (int32_t)r;
}

// frontier of sweep k: the rows whose sums read a value that changed in sweep k - 1, i.e. the in-neighbours (rows of the
// TRANSPOSED pattern) of the nodes that entered the list one sweep earlier.  A wave per source node (hub rows hold
// thousands of entries), first touch by atomic exchange on the node's stamp; the order of the list does not matter.
__global__ __launch_bounds__(256) void k_inc_expand(int k, const int64_t *__restrict__ ht_indptr, const int32_t *__restrict__ ht_indices,
                                                    uint32_t *__restrict__ stamp, uint32_t ep, int32_t *__restrict__ mem, int cap,
                                                    LpInc *__restrict__ ctl) {
    const int lane = threadIdx.x & 63;
    const int lo = k >= 2 ? ctl->m[k - 2] : 0, hi = ctl->m[k - 1];
    const int nwaves = gridDim.x * 4;
    for (int q = lo + blockIdx.x * 4 + (threadIdx.x >> 6); q < hi; q += nwaves) {
        const int64_t j = mem[q];
        for (int64_t p = ht_indptr[j] + lane, e = ht_indptr[j + 1]; p < e; p += 64) {
            const int32_t i = ht_indices[p];
            if (atomicExch(&stamp[i], ep) != ep) {
                const int pos = atomicAdd(&ctl->total, 1);
                if (pos < cap) mem[pos] = i;
                else ctl->overflow = 1;
            }
        }
    }
}

// Sweep k over the listed rows: a wave takes FOUR rows at a time, 16 lanes a row (a row of this graph holds ~12 entries;
// hub rows hold thousands and keep their 16 lanes busy for more trips while the other three groups idle).  A group takes
// its row 128 entries at a time -- coalesced index / weight loads, the gathers of f_old in flight together, each product
// formed once and parked in the group's 1 KB of LDS -- then the group's first lane adds them in ascending position: the
// order of the full sweeps and of scipy's csr_matvec.  Rows that are new in this sweep also enter their block of 256
// nodes in the block list (first touch by the block's stamp).
__global__ __launch_bounds__(256) void k_inc_rows(int k, const int32_t *__restrict__ mem, int cap, const int64_t *__restrict__ indptr,
                                                  const int32_t *__restrict__ indices, const double *__restrict__ data,
                                                  const double *__restrict__ wsum, const double *__restrict__ prior,
                                                  const double *__restrict__ f_old, double *__restrict__ f_new,
                                                  const unsigned char *__restrict__ is_label, const double *__restrict__ label_val,
                                                  double lambda, double low_bound, double high_bound, uint32_t *__restrict__ bstamp,
                                                  uint32_t ep, int32_t *__restrict__ blk, LpInc *__restrict__ ctl) {
    __shared__ double prod[4][4][128];
    const int lane = threadIdx.x & 63, g = lane >> 4, gl = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mk = min(ctl->total, cap), m_prev = ctl->m[k - 1];  // (the expansion of this sweep has finished: stream order)
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->m[k] = mk;
    const int nwaves = gridDim.x * 4;
    for (int q0 = (blockIdx.x * 4 + wave) * 4; q0 < mk; q0 += nwaves * 4) {  // (q0 is wave-uniform)
        const int q = q0 + g;
        const bool live = q < mk;
        const int64_t row = live ? (int64_t)mem[q] : 0;
        if (live && gl == 0 && (k == 1 || q >= m_prev)) {  // a row new to the list: its block joins the block list
            const uint32_t b = (uint32_t)(row >> 8);
            if (atomicExch(&bstamp[b], ep) != ep) blk[atomicAdd(&ctl->blocks, 1)] = (int32_t)b;
        }
        const int64_t p0 = live ? indptr[row] : 0, p1 = live ? indptr[row + 1] : 0;
        int64_t len = p1 - p0, longest = len;
        longest = max(longest, (int64_t)__shfl_xor((long long)longest, 16, 64));
        longest = max(longest, (int64_t)__shfl_xor((long long)longest, 32, 64));
        double sum = 0.0;
        for (int64_t base = 0; base < longest; base += 128) {  // (wave-uniform trip count: the longest of the four rows)
            const int cnt = (int)max((int64_t)0, min((int64_t)128, len - base));
            int32_t col[8];
            double w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int x = u * 16 + gl;
                col[u] = x < cnt ? indices[p0 + base + x] : -1;
                w[u] = x < cnt ? data[p0 + base + x] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int x = u * 16 + gl;
                if (col[u] >= 0) prod[wave][g][x] = __dmul_rn(w[u], f_old[col[u]]);
            }
            // (one wave writes and reads these bytes: its LDS operations complete in order; the fences keep the compiler
            //  from moving the adding lanes' reads ahead of the other lanes' writes, and the next trip's writes ahead of the reads)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (gl == 0)
                for (int p = 0; p < cnt; ++p) sum = __dadd_rn(sum, prod[wave][g][p]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (live && gl == 0) {
            const double weighted = __dadd_rn(sum, __dmul_rn(lambda, prior[row]));
            double v = weighted / __dadd_rn(wsum[row], lambda);
            if (!(v >= low_bound) || !(v <= high_bound)) ctl->bound_violation = 1;
            if (is_label[row]) v = label_val[row];
            f_new[row] = v;
        }
    }
}

// max over a block of 256 nodes of (f_new - f_old)^2.  blk == nullptr: every block (a full sweep's pass; skipped once
// the run has converged); otherwise the blocks of the incremental call's list (all touched so far)
__global__ __launch_bounds__(256) void k_lp_blockmax(const int32_t *__restrict__ blk, const LpInc *__restrict__ ctl, int64_t n,
                                                     const double *__restrict__ f_new, const double *__restrict__ f_old,
                                                     double *__restrict__ bmax, const LpState *__restrict__ st_or_null) {
    __shared__ double red[4];
    if (st_or_null && st_or_null->done) return;  // a sweep enqueued past convergence wrote nothing
    const int count = blk ? ctl->blocks : (int)gridDim.x;
    for (int q = blockIdx.x; q < count; q += gridDim.x) {
        const int64_t b = blk ? (int64_t)blk[q] : (int64_t)q;
        const int64_t row = b * 256 + threadIdx.x;
        double d2 = 0.0;
        if (row < n) {
            const double d = __dadd_rn(f_new[row], -f_old[row]);
            d2 = __dmul_rn(d, d);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) d2 = fmax(d2, __shfl_xor(d2, off, 64));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d2;
        __syncthreads();
        if (threadIdx.x == 0) bmax[b] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        __syncthreads();
    }
}

// per kept sweep (blockIdx.x + 1): the maximum over all its blocks
struct LpBmaxPtrs {
    const double *p[8];
};
__global__ __launch_bounds__(1024) void k_lp_levelmax(LpBmaxPtrs bm, int64_t nb, LpInc *__restrict__ ctl) {
    __shared__ double red[16];
    const int k = blockIdx.x + 1;
    const double *bmax = bm.p[k];
    double m = 0.0;
    for (int64_t i = threadIdx.x; i < nb; i += 1024) m = fmax(m, bmax[i]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) m = fmax(m, red[i]);
        ctl->level_max[k] = m;
    }
}

// The tail of a fused round (ssw_labelprop_round): the incremental pass's answer is iterate conv - 1, conv = the first
// sweep whose maximum over all blocks is below eps (the host's rule, lp_run_tracked) -- decided HERE from the control
// block, so the f32 scores for the selection can be written without the host looking at the control block first.  Thread 0
// leaves a copy of the block in pinned host memory (m[8] carries conv); when the pass did not converge / overflowed /
// left the bounds nothing is written and the host redoes the round the slow way.
struct LpFPtrs {
    const double *p[8];
};
__global__ void k_lp_scores_sel(LpFPtrs fp, const LpInc *__restrict__ ctl, int levels, double eps,
                                const unsigned char *__restrict__ is_label_or_null, const int32_t *__restrict__ perm_or_null,
                                int64_t n, float *__restrict__ out, LpInc *__restrict__ host_copy) {
    int conv = 0;
    if (!ctl->overflow && !ctl->bound_violation)
        for (int k = 1; k <= levels; ++k)
            if (ctl->level_max[k] < eps) {
                conv = k;
                break;
            }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        LpInc c = *ctl;
        c.m[8] = conv;
        *host_copy = c;
        __threadfence_system();
    }
    if (!conv) return;
    const double *__restrict__ f = fp.p[conv - 1];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t r = perm_or_null ? (int64_t)perm_or_null[i] : i;
    out[i] = (is_label_or_null && is_label_or_null[r]) ? -INFINITY : (float)f[r];
}

__global__ void k_lp_clear_labels(unsigned char *is_label, const int64_t *ids, int64_t n_labels) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_labels) is_label[ids[i]] = 0;
}
__global__ void k_lp_mark_labels(unsigned char *is_label, const int64_t *ids, int64_t n_labels) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_labels) is_label[ids[i]] = 1;
}


// ---------------------------------------------------------------------------------------
// K6: X' L X  (graph_based.py:45-49 -- the database-alignment matrix of MultiReg)
//   Y = L X          one wave per row, f64, row by row in storage order with separate
//                    multiply and add: scipy's csr_matvecs order (L f64, X upcast from f32)
//   P_z = X_z' Y_z   v_mfma_f64_16x16x4_f64 over row chunk z; a 256-thread workgroup owns a
//                    128 x 128 block of the D x D result (wave: 64 x 64 = 16 accumulators)
//   out = sum_z P_z  in ascending z (deterministic)
// Bound: MFMA f64 (2 N D^2 flop) for the product, L2 gathers (nnz x D x 4 B) for Y.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_xlx_spmm(const int64_t *__restrict__ indptr,
                                                  const int32_t *__restrict__ indices,
                                                  const double *__restrict__ data, const float *__restrict__ X,
                                                  int64_t n, int D, double *__restrict__ Y) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const int64_t lo = indptr[r], hi = indptr[r + 1];
    for (int c = lane; c < D; c += 64) {
        double acc = 0.0;
        for (int64_t jj = lo; jj < hi; ++jj)
            acc = __dadd_rn(acc, __dmul_rn(data[jj], (double)X[(int64_t)indices[jj] * D + c]));
        Y[r * D + c] = acc;
    }
}

typedef __attribute__((ext_vector_type(4))) double f64x4;

__global__ __launch_bounds__(256) void k_xlx_partial(const float *__restrict__ X, const double *__restrict__ Y,
                                                     int64_t n, int D, int64_t rows_per_chunk,
                                                     double *__restrict__ P) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int a0 = blockIdx.x * 128 + (wave >> 1) * 64, b0 = blockIdx.y * 128 + (wave & 1) * 64;
    const int64_t first = (int64_t)blockIdx.z * rows_per_chunk;
    const int64_t last = (first + rows_per_chunk < n) ? first + rows_per_chunk : n;
    const int m = lane & 15, kk = lane >> 4;
    f64x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
    for (int64_t i = first; i < last; i += 4) {
        const int64_t row = i + kk;
        const bool valid = row < last;
        const float *xr = X + (valid ? row : first) * D + a0 + m;
        const double *yr = Y + (valid ? row : first) * D + b0 + m;
        double a[4], b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a[t] = valid ? (double)xr[t * 16] : 0.0;
            b[t] = valid ? yr[t * 16] : 0.0;
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
    }
    // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
    double *Pz = P + (int64_t)blockIdx.z * D * D;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Pz[(int64_t)(a0 + mt * 16 + kk + 4 * r) * D + b0 + nt * 16 + m] = acc[mt][nt][r];
}

__global__ void k_xlx_reduce(const double *__restrict__ P, int n_chunks, int64_t dd, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= dd) return;
    double acc = 0.0;
    for (int z = 0; z < n_chunks; ++z) acc += P[(int64_t)z * dd + i];
    out[i] = acc;
}

}  // namespace
}  // namespace ssw

using namespace ssw;

namespace {
// a graph stored in a locality order (ssw_labelprop_set_permutation): perm[old id] = position in the device arrays
__global__ void k_lp_permute_in(const double *__restrict__ src_old, const int32_t *__restrict__ perm, int64_t n,
                                double *__restrict__ dst_new) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst_new[perm[i]] = src_old[i];
}
__global__ void k_lp_permute_out(const double *__restrict__ src_new, const int32_t *__restrict__ perm, int64_t n,
                                 double *__restrict__ dst_old) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst_old[i] = src_new[perm[i]];
}
__global__ void k_lp_scores_f32_perm(const double *__restrict__ f, const unsigned char *__restrict__ is_label_or_null,
                                     const int32_t *__restrict__ perm, int64_t n, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t r = perm[i];
    out[i] = (is_label_or_null && is_label_or_null[r]) ? -INFINITY : (float)f[r];
}

__global__ void k_lp_scores_f32_perm_or_plain(const double *__restrict__ f, const unsigned char *__restrict__ is_label_or_null,
                                              const int32_t *__restrict__ perm_or_null, int64_t n, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t r = perm_or_null ? (int64_t)perm_or_null[i] : i;
    out[i] = (is_label_or_null && is_label_or_null[r]) ? -INFINITY : (float)f[r];
}

__global__ void k_lp_gather(const double *__restrict__ f, const int64_t *__restrict__ rows, int64_t m,
                            double *__restrict__ out) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < m) out[i] = f[rows[i]];
}
}  // namespace

struct ssw_lp {
    int device = 0;
    int64_t n = 0, nnz = 0;
    int64_t *indptr = nullptr;
    int32_t *indices = nullptr;
    double *data = nullptr;
    double *wsum = nullptr;
    double *prior = nullptr;
    static constexpr int KEEP = 8;            // iterates a tracked run keeps: F[0 .. KEEP - 1], i.e. up to KEEP - 1 sweeps
    double *f[KEEP] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // [0], [1] always; the rest on demand
    unsigned char *is_label = nullptr;
    double *label_val = nullptr;
    int64_t *ids = nullptr;
    double *vals = nullptr;
    int64_t ids_cap = 0;
    int64_t n_labels_installed = 0;
    int sweeps_hint = 0;  // sweeps the previous converged run needed (sizes the next run's first batch)
    // locality order (ssw_labelprop_set_permutation): the device arrays are indexed by perm[original id]; every entry
    // point keeps speaking original ids
    int32_t *perm = nullptr;             // device [n]
    std::vector<int32_t> perm_host;      // host copy (label ids, gather rows)
    double *scratch = nullptr;           // [n] staging of uploads / the un-permuted result
    std::vector<int64_t> ids_tmp;
    // device-resident chaining: an installed prior (reg_values == start iterate of every call of the ranking
    // loop) and the buffer holding the last result
    bool prior_installed = false;
    double prior_lo = 0.0, prior_hi = 1.0;
    int last_result = -1;
    LpState *state = nullptr;
    hipStream_t stream = nullptr;
    // column-blocked copy of the matrix (nblk > 1 only): slice-major non-zeros, per-slice row offsets
    int nblk = 1;
    uint32_t *bl_ptr = nullptr;      // [nblk][n + 1]
    int32_t *bl_indices = nullptr;   // [nnz] sorted by (slice, row, column)
    double *bl_data = nullptr;       // [nnz]
    std::vector<int64_t> bl_base;    // [nblk + 1] first non-zero of every slice
    int64_t *bl_base_dev = nullptr;  // device copy
    int groups = 6;                  // row groups of 256 per workgroup of k_lp_sweep_fused (grid just resident)
    // ssw_labelprop_gather staging: device [cap] ids + values, pinned host mirror of both
    int64_t *g_rows = nullptr, *g_rows_host = nullptr;
    double *g_vals = nullptr, *g_vals_host = nullptr;
    int64_t g_cap = 0;
    // ---- incremental propagation (ssw_labelprop_run_resident): what the previous tracked run left behind
    struct Track {
        bool valid = false;               // f[0 .. levels], bmax[1 .. levels] and the labels below describe the previous call
        int levels = 0;                   // sweeps whose iterates are kept (the previous call converged within them)
        int sweeps = 0, result = 0;       // what the previous call returned (reused when the labels did not change)
        double lambda = 0.0, eps = 0.0;
        int max_iter = 0;
        std::vector<int64_t> ids;         // installed labels, device positions, ascending
        std::vector<double> vals;
    } trk;
    double *bmax[KEEP] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // [k][blocks of 256 nodes]: max (f[k] - f[k-1])^2
    int64_t *inc_dev = nullptr, *inc_host = nullptr;  // packed per-call label lists (device / pinned host), inc_cap words each
    int64_t inc_cap = 0;
    // the transposed pattern on the device (rows of W^T: who reads node j), per-node / per-block stamps of the call that
    // last touched them, the frontier's row and block lists and its control block
    int64_t *ht_indptr = nullptr;
    int32_t *ht_indices = nullptr;
    uint32_t *stamp = nullptr, *bstamp = nullptr;
    int32_t *mem = nullptr, *blk = nullptr;
    int cap_rows = 0;
    uint32_t epoch = 0;
    // what the last propagation did (ssw_labelprop_last_run_info)
    int64_t info[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    LpInc *round_host = nullptr;  // pinned, device-mapped: the control block as the fused round's scores kernel saw it
};

// the rest of a feedback round behind the propagation (ssw_labelprop_round): f32 scores into the index + the selection
struct LpRoundTail {
    ssw_index *index = nullptr;
    float *index_scores = nullptr;
    bool mask_labeled = true;
    const int64_t *excluded = nullptr;
    int64_t n_excluded = 0;
    int32_t k = 0;
    int64_t *out_images = nullptr;
    float *out_scores = nullptr;
    int64_t *out_best_rows = nullptr;
    int32_t *out_count = nullptr;
    bool valid = false;  // the selection's result was produced from the propagation's answer and sits in out_*
};

extern "C" {

static const int64_t *lp_map_ids(ssw_lp *lp, const int64_t *ids, int64_t m) {
    if (!lp->perm || m <= 0) return ids;
    lp->ids_tmp.resize((size_t)m);
    for (int64_t i = 0; i < m; ++i) lp->ids_tmp[(size_t)i] = lp->perm_host[(size_t)ids[i]];
    return lp->ids_tmp.data();
}

ssw_status ssw_labelprop_destroy(ssw_lp *lp) {
    if (!lp) return SSW_OK;
    DeviceGuard guard(lp->device);
    if (lp->stream) (void)hipStreamSynchronize(lp->stream);
    (void)hipFree(lp->indptr);
    (void)hipFree(lp->indices);
    (void)hipFree(lp->data);
    (void)hipFree(lp->wsum);
    (void)hipFree(lp->prior);
    for (int k = 0; k < ssw_lp::KEEP; ++k) {
        (void)hipFree(lp->f[k]);
        (void)hipFree(lp->bmax[k]);
    }
    for (void *q : {(void *)lp->ht_indptr, (void *)lp->ht_indices, (void *)lp->stamp, (void *)lp->bstamp, (void *)lp->mem,
                    (void *)lp->blk})
        (void)hipFree(q);
    (void)hipFree(lp->inc_dev);
    if (lp->inc_host) (void)hipHostFree(lp->inc_host);
    if (lp->round_host) (void)hipHostFree(lp->round_host);
    (void)hipFree(lp->is_label);
    (void)hipFree(lp->g_rows);
    (void)hipFree(lp->g_vals);
    if (lp->g_rows_host) (void)hipHostFree(lp->g_rows_host);
    if (lp->g_vals_host) (void)hipHostFree(lp->g_vals_host);
    (void)hipFree(lp->label_val);
    (void)hipFree(lp->ids);
    (void)hipFree(lp->vals);
    (void)hipFree(lp->state);
    (void)hipFree(lp->bl_ptr);
    (void)hipFree(lp->bl_indices);
    (void)hipFree(lp->bl_data);
    (void)hipFree(lp->bl_base_dev);
    (void)hipFree(lp->perm);
    (void)hipFree(lp->scratch);
    if (lp->stream) (void)hipStreamDestroy(lp->stream);
    delete lp;
    return SSW_OK;
}

static thread_local bool g_lp_skip_blocked = false;
static ssw_status lp_upload_transpose(ssw_lp *lp, const int64_t *indptr_host, const int32_t *indices_host);
static ssw_status lp_ensure_transpose(ssw_lp *lp);
static ssw_status lp_session_buffers(ssw_lp *lp);
static ssw_status lp_reserve_labels(ssw_lp *lp, const std::vector<int64_t> &ids);

ssw_status ssw_labelprop_create(int32_t device, int64_t n, const int64_t *indptr_host,
                                const int32_t *indices_host, const double *data_host,
                                const double *weight_sum_host_or_null, ssw_lp **out) {
    SSW_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    SSW_REQUIRE(n > 0 && indptr_host && indices_host && data_host, "labelprop: empty or NULL graph");
    SSW_REQUIRE(indptr_host[0] == 0, "labelprop: indptr[0] != 0");
    const int64_t nnz = indptr_host[n];
    for (int64_t i = 0; i < n; ++i)
        SSW_REQUIRE(indptr_host[i] <= indptr_host[i + 1], "labelprop: indptr not monotone at %lld",
                    (long long)i);
    for (int64_t p = 0; p < nnz; ++p)
        SSW_REQUIRE(indices_host[p] >= 0 && indices_host[p] < n, "labelprop: column %d out of range",
                    indices_host[p]);
    // W.sum(0) exactly as scipy accumulates it (rows ascending): label_propagation.py:25
    std::vector<double> wsum((size_t)n, 0.0);
    if (weight_sum_host_or_null) {
        memcpy(wsum.data(), weight_sum_host_or_null, (size_t)n * sizeof(double));
    } else {
        for (int64_t i = 0; i < n; ++i)
            for (int64_t p = indptr_host[i]; p < indptr_host[i + 1]; ++p)
                wsum[(size_t)indices_host[p]] += data_host[p];
    }
    DeviceGuard guard(device);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", device);
        return SSW_ERR_HIP;
    }
    ssw_lp *lp = new (std::nothrow) ssw_lp();
    if (!lp) return SSW_ERR_NOMEM;
    lp->device = device;
    lp->n = n;
    lp->nnz = nnz;
    auto bail = [&](ssw_status s) {
        ssw_labelprop_destroy(lp);
        return s;
    };
#define LP_ALLOC(ptr, bytes)                                              \
    if (hipMalloc((void **)&(ptr), (size_t)(bytes) + 16) != hipSuccess) { \
        set_error("labelprop: hipMalloc of %zu bytes failed", (size_t)(bytes)); \
        return bail(SSW_ERR_NOMEM);                                       \
    }
    LP_ALLOC(lp->indptr, (n + 1) * sizeof(int64_t));
    LP_ALLOC(lp->indices, nnz * sizeof(int32_t));
    LP_ALLOC(lp->data, nnz * sizeof(double));
    LP_ALLOC(lp->wsum, n * sizeof(double));
    LP_ALLOC(lp->prior, n * sizeof(double));
    LP_ALLOC(lp->f[0], n * sizeof(double));
    LP_ALLOC(lp->f[1], n * sizeof(double));
    LP_ALLOC(lp->is_label, n);
    LP_ALLOC(lp->label_val, n * sizeof(double));
    LP_ALLOC(lp->state, sizeof(LpState));
#undef LP_ALLOC
    if (hipStreamCreateWithFlags(&lp->stream, hipStreamNonBlocking) != hipSuccess) return bail(SSW_ERR_HIP);
    if (hipMemcpy(lp->indptr, indptr_host, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(lp->indices, indices_host, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(lp->data, data_host, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(lp->wsum, wsum.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(lp->is_label, 0, (size_t)n) != hipSuccess) {
        set_error("labelprop: graph upload failed");
        return bail(SSW_ERR_HIP);
    }
    // column-blocked copy when the iterate (8 B per node) exceeds what stays resident in one XCD's L2
    int64_t slice_bytes = LP_SLICE_BYTES;
    if (const char *e = getenv("SSW_LP_SLICE_KB")) slice_bytes = std::max<int64_t>(1, atoll(e)) * 1024;  // tuning / A-B runs / tests
    const int64_t cols_per_slice = std::max<int64_t>(1, slice_bytes / (int64_t)sizeof(double));
    int64_t nblk = (n + cols_per_slice - 1) / cols_per_slice;
    // an iterate that (nearly) fits L2 is better served by the plain kernel (measured: 400 k nodes = 3.2 MB plain
    // 50 us vs blocked 58 us; 800 k = 6.4 MB 120 vs 112 us; 1.56 M = 12.5 MB 305 vs 161 us; 3 M 693 vs 387 us)
    if (!getenv("SSW_LP_SLICE_KB") && n * (int64_t)sizeof(double) <= LP_BLOCKED_ABOVE_BYTES) nblk = 1;
    if (g_lp_skip_blocked) nblk = 1;  // ssw_labelprop_create_ordered: the columns do not ascend, the slices would be wrong
    if (nblk > 1 && nblk <= 4096 && nnz < (int64_t)0xffffffffll) {
        std::vector<uint32_t> bptr((size_t)nblk * (size_t)(n + 1), 0u);
        std::vector<int64_t> base((size_t)nblk + 1, 0);
        // per (slice, row) counts -> offsets inside the slice
        for (int64_t i = 0; i < n; ++i)
            for (int64_t p = indptr_host[i]; p < indptr_host[i + 1]; ++p)
                bptr[(size_t)(indices_host[p] / cols_per_slice) * (size_t)(n + 1) + (size_t)i + 1]++;
        for (int64_t b = 0; b < nblk; ++b) {
            uint32_t *row = bptr.data() + (size_t)b * (size_t)(n + 1);
            for (int64_t i = 0; i < n; ++i) row[i + 1] += row[i];
            base[(size_t)b + 1] = base[(size_t)b] + row[n];
        }
        std::vector<int32_t> bind((size_t)nnz);
        std::vector<double> bdat((size_t)nnz);
        std::vector<uint32_t> fill((size_t)nblk, 0u);
        for (int64_t i = 0; i < n; ++i) {
            for (int64_t b = 0; b < nblk; ++b) fill[(size_t)b] = bptr[(size_t)b * (size_t)(n + 1) + (size_t)i];
            for (int64_t p = indptr_host[i]; p < indptr_host[i + 1]; ++p) {  // columns ascend: order inside a slice kept
                const int64_t b = indices_host[p] / cols_per_slice;
                const size_t at = (size_t)base[(size_t)b] + fill[(size_t)b]++;
                bind[at] = indices_host[p];
                bdat[at] = data_host[p];
            }
        }
        if (hipMalloc((void **)&lp->bl_ptr, bptr.size() * sizeof(uint32_t) + 16) != hipSuccess ||
            hipMalloc((void **)&lp->bl_indices, (size_t)nnz * sizeof(int32_t) + 16) != hipSuccess ||
            hipMalloc((void **)&lp->bl_data, (size_t)nnz * sizeof(double) + 16) != hipSuccess ||
            hipMemcpy(lp->bl_ptr, bptr.data(), bptr.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(lp->bl_indices, bind.data(), (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(lp->bl_data, bdat.data(), (size_t)nnz * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
            set_error("labelprop: column-blocked copy of the graph failed");
            return bail(SSW_ERR_NOMEM);
        }
        if (hipMalloc((void **)&lp->bl_base_dev, base.size() * sizeof(int64_t)) != hipSuccess ||
            hipMemcpy(lp->bl_base_dev, base.data(), base.size() * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess) {
            set_error("labelprop: column-blocked copy of the graph failed");
            return bail(SSW_ERR_NOMEM);
        }
        lp->nblk = (int)nblk;
        lp->bl_base = std::move(base);
        // smallest row-group count whose grid is resident all at once: 4 workgroups of 32 KiB LDS per CU
        const int64_t resident = (int64_t)num_cus(device) * 4;
        lp->groups = 16;
        for (int g : {2, 3, 4, 6, 8, 12, 16})
            if ((n + (int64_t)g * LP_BLOCK - 1) / ((int64_t)g * LP_BLOCK) <= resident) {
                lp->groups = g;
                break;
            }
    }
    // (the TRANSPOSED pattern for the incremental runs' frontier sets is built by the first ssw_labelprop_set_prior -- the
    // per-session call -- so handles that only ever run ssw_labelprop_run pay neither its 0.1 s nor its memory: ADVICE r5)
    *out = lp;
    return SSW_OK;
}

// one sweep f[src] -> f[dst]: the plain kernel, or the column-blocked one (same result, bit for bit)
static void lp_launch_sweep(ssw_lp *lp, int src, int dst, double reg_lambda, double lo, double hi) {
    const int64_t n = lp->n;
    hipStream_t s = lp->stream;
    if (lp->nblk <= 1) {
        const unsigned grid = (unsigned)((n + LP_BLOCK - 1) / LP_BLOCK);
        hipLaunchKernelGGL(k_lp_sweep, dim3(grid), dim3(LP_BLOCK), 0, s, n, lp->indptr, lp->indices, lp->data, lp->wsum,
                           lp->prior, lp->f[src], lp->f[dst], lp->is_label, lp->label_val, reg_lambda, lo, hi, lp->state);
        return;
    }
#define LP_FUSED(G)                                                                                                        \
    hipLaunchKernelGGL((k_lp_sweep_fused<G, LP_CHUNK>), dim3((unsigned)((n + (int64_t)(G) * LP_BLOCK - 1) / ((int64_t)(G) * LP_BLOCK))), \
                       dim3(LP_BLOCK), 0, s, n, lp->nblk, lp->bl_ptr, lp->bl_base_dev, lp->bl_indices, lp->bl_data, lp->wsum,       \
                       lp->prior, lp->f[src], lp->f[dst], lp->is_label, lp->label_val, reg_lambda, lo, hi, lp->state)
    switch (lp->groups) {
        case 2: LP_FUSED(2); break;
        case 3: LP_FUSED(3); break;
        case 4: LP_FUSED(4); break;
        case 6: LP_FUSED(6); break;
        case 8: LP_FUSED(8); break;
        case 12: LP_FUSED(12); break;
        default: LP_FUSED(16); break;
    }
#undef LP_FUSED
}

// Shared body of ssw_labelprop_run / ssw_labelprop_run_resident.  prior: uploaded from
// prior_host_or_null, or (use_installed) the buffer ssw_labelprop_set_prior left on the device, or zeros.
// start: uploaded from start_host_or_null, or a device copy of the prior.  The result stays in
// lp->f[lp->last_result].
static ssw_status lp_run_core(ssw_lp *lp, const double *prior_host_or_null, bool use_installed,
                              const double *start_host_or_null, const int64_t *label_ids, const double *label_vals,
                              int64_t n_labels, double reg_lambda, double eps, int32_t max_iter, LpState *st_out) {
    SSW_REQUIRE(reg_lambda >= 0.0, "reg_lambda < 0");
    SSW_REQUIRE(max_iter >= 0, "max_iter < 0");
    SSW_REQUIRE(n_labels == 0 || (label_ids && label_vals), "NULL labels");
    for (int64_t i = 0; i < n_labels; ++i)
        SSW_REQUIRE(label_ids[i] >= 0 && label_ids[i] < lp->n, "label id %lld out of range",
                    (long long)label_ids[i]);
    const int64_t n = lp->n;
    hipStream_t s = lp->stream;
    lp->trk.valid = false;  // this path keeps two iterates and no block maxima: the next resident call starts over
    // bounds of the reference's sanity asserts: min(0, prior.min()) .. max(1, prior.max())
    double lo = 0.0, hi = 1.0;
    if (use_installed) {
        lo = lp->prior_lo;
        hi = lp->prior_hi;
    } else if (prior_host_or_null) {
        for (int64_t i = 0; i < n; ++i) {
            const double v = prior_host_or_null[i];
            if (v < lo) lo = v;
            if (v > hi) hi = v;
        }
        lp->prior_installed = false;
        if (lp->perm) {
            SSW_HIP_TRY(hipMemcpyAsync(lp->scratch, prior_host_or_null, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_lp_permute_in, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, lp->scratch, lp->perm, n,
                               lp->prior);
            SSW_HIP_TRY(hipStreamSynchronize(s));  // scratch is reused for the start iterate below
        } else {
            SSW_HIP_TRY(hipMemcpyAsync(lp->prior, prior_host_or_null, (size_t)n * sizeof(double),
                                       hipMemcpyHostToDevice, s));
        }
    } else {
        lp->prior_installed = false;
        SSW_HIP_TRY(hipMemsetAsync(lp->prior, 0, (size_t)n * sizeof(double), s));
    }
    if (start_host_or_null && lp->perm) {
        SSW_HIP_TRY(hipMemcpyAsync(lp->scratch, start_host_or_null, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_lp_permute_in, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, lp->scratch, lp->perm, n,
                           lp->f[0]);
    } else if (start_host_or_null)
        SSW_HIP_TRY(hipMemcpyAsync(lp->f[0], start_host_or_null, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
    else
        SSW_HIP_TRY(hipMemcpyAsync(lp->f[0], lp->prior, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
    // labels: clear the previous set, install the new one (also clamps the start iterate)
    if (lp->n_labels_installed > 0) {
        const int64_t m = lp->n_labels_installed;
        hipLaunchKernelGGL(k_lp_clear_labels, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s,
                           lp->is_label, lp->ids, m);
        lp->n_labels_installed = 0;
    }
    if (n_labels > 0) {
        if (n_labels > lp->ids_cap) {
            SSW_HIP_TRY(hipStreamSynchronize(s));
            (void)hipFree(lp->ids);
            (void)hipFree(lp->vals);
            lp->ids = nullptr;
            lp->vals = nullptr;
            int64_t cap = 1024;
            while (cap < n_labels) cap <<= 1;
            SSW_HIP_TRY(hipMalloc((void **)&lp->ids, (size_t)cap * sizeof(int64_t)));
            SSW_HIP_TRY(hipMalloc((void **)&lp->vals, (size_t)cap * sizeof(double)));
            lp->ids_cap = cap;
        }
        SSW_HIP_TRY(hipMemcpyAsync(lp->ids, lp_map_ids(lp, label_ids, n_labels), (size_t)n_labels * sizeof(int64_t),
                                   hipMemcpyHostToDevice, s));
        SSW_HIP_TRY(hipMemcpyAsync(lp->vals, label_vals, (size_t)n_labels * sizeof(double),
                                   hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_lp_apply_labels, dim3((unsigned)((n_labels + 255) / 256)), dim3(256), 0, s,
                           lp->f[0], lp->is_label, lp->label_val, lp->ids, lp->vals, n_labels);
        lp->n_labels_installed = n_labels;
    }
    SSW_HIP_TRY(hipMemsetAsync(lp->state, 0, sizeof(LpState), s));
    // the host buffers above must be consumed before this call returns: we synchronise below
    LpState st;
    memset(&st, 0, sizeof(st));
    int issued = 0, syncs = 0;
    // Sweeps are enqueued in batches and the host looks at the state once per batch; sweeps enqueued past convergence
    // are no-ops but still cost their launch and their check (~13 us a pair).  Consecutive rounds of a session converge
    // in about the same number of sweeps, so the first batch is what the previous call needed plus one (8 at first).
    int batch = lp->sweeps_hint > 0 ? std::min(8, lp->sweeps_hint + 1) : 8;
    while (issued < max_iter) {
        const int upto = (issued + batch < max_iter) ? issued + batch : max_iter;
        batch = 8;  // a second batch means the hint was too short: full size from here on
        for (; issued < upto; ++issued) {
            const int src = issued & 1;
            lp_launch_sweep(lp, src, src ^ 1, reg_lambda, lo, hi);
            hipLaunchKernelGGL(k_lp_check, dim3(1), dim3(1), 0, s, lp->state, eps, src, src ^ 1);
        }
        SSW_HIP_TRY(hipGetLastError());
        SSW_HIP_TRY(hipMemcpyAsync(&st, lp->state, sizeof(LpState), hipMemcpyDeviceToHost, s));
        SSW_HIP_TRY(hipStreamSynchronize(s));
        ++syncs;
        if (st.done) break;
    }
    if (max_iter == 0) SSW_HIP_TRY(hipStreamSynchronize(s));
    if (st.bound_violation) {
        set_error("label propagation: averaged scores left [%g, %g] (label_propagation.py:39-40)", lo, hi);
        return SSW_ERR_NUMERIC;
    }
    lp->last_result = (st.sweeps > 0) ? st.result_buf : 0;
    lp->sweeps_hint = st.done ? (int)st.sweeps : 0;
    lp->info[0] = 0, lp->info[1] = st.sweeps, lp->info[2] = 2 * (int64_t)issued + 2, lp->info[3] = syncs,
    lp->info[4] = (int64_t)issued * n, lp->info[5] = 0;
    *st_out = st;
    return SSW_OK;
}

ssw_status ssw_labelprop_run(ssw_lp *lp, const double *prior_host_or_null, const double *start_host,
                             const int64_t *label_ids, const double *label_vals, int64_t n_labels,
                             double reg_lambda, double eps, int32_t max_iter, double *out_f_host,
                             int32_t *out_sweeps, int32_t *out_converged) {
    SSW_REQUIRE(lp != nullptr && start_host != nullptr && out_f_host != nullptr, "NULL argument");
    SSW_REQUIRE(prior_host_or_null != nullptr || reg_lambda == 0.0,
                "reg_values is required when reg_lambda != 0 (label_propagation.py:50)");
    DeviceGuard guard(lp->device);
    LpState st;
    SSW_TRY(lp_run_core(lp, prior_host_or_null, false, start_host, label_ids, label_vals, n_labels, reg_lambda, eps,
                        max_iter, &st));
    const double *res = lp->f[lp->last_result];
    if (lp->perm) {
        hipLaunchKernelGGL(k_lp_permute_out, dim3((unsigned)((lp->n + 255) / 256)), dim3(256), 0, lp->stream, res, lp->perm,
                           lp->n, lp->scratch);
        res = lp->scratch;
    }
    SSW_HIP_TRY(hipMemcpyAsync(out_f_host, res, (size_t)lp->n * sizeof(double), hipMemcpyDeviceToHost, lp->stream));
    SSW_HIP_TRY(hipStreamSynchronize(lp->stream));
    if (out_sweeps) *out_sweeps = st.sweeps;
    if (out_converged) *out_converged = st.done;
    return SSW_OK;
}

// The graph passed to ssw_labelprop_create was laid out in a locality order: node `old` sits at position
// new_of_old[old]; row r of the CSR holds the entries of the node at position r with columns RELABELLED to positions but
// kept in ascending ORIGINAL column id -- the order scipy adds a row's products in, which the plain sweep (each lane adds
// its row's products in storage order) therefore still follows bit for bit.  Neighbours then sit near each other in the
// iterate and the sweep's gathers hit lines their neighbours' rows already pulled; the column-blocked copy (whose slices
// assume ascending columns) is dropped.  Every entry point keeps taking and returning ORIGINAL ids.
ssw_status ssw_labelprop_set_permutation(ssw_lp *lp, const int32_t *new_of_old_host) {
    SSW_REQUIRE(lp != nullptr && new_of_old_host != nullptr, "NULL argument");
    SSW_REQUIRE(lp->perm == nullptr && lp->last_result < 0 && !lp->prior_installed && lp->n_labels_installed == 0,
                "ssw_labelprop_set_permutation: call it right after ssw_labelprop_create");
    std::vector<unsigned char> seen((size_t)lp->n, 0);
    for (int64_t i = 0; i < lp->n; ++i) {
        const int32_t r = new_of_old_host[i];
        SSW_REQUIRE(r >= 0 && r < lp->n && !seen[(size_t)r], "not a permutation of [0, %lld) at %lld", (long long)lp->n,
                    (long long)i);
        seen[(size_t)r] = 1;
    }
    DeviceGuard guard(lp->device);
    SSW_HIP_TRY(hipStreamSynchronize(lp->stream));
    SSW_HIP_TRY(hipMalloc((void **)&lp->perm, (size_t)lp->n * sizeof(int32_t) + 16));
    SSW_HIP_TRY(hipMalloc((void **)&lp->scratch, (size_t)lp->n * sizeof(double) + 16));
    SSW_HIP_TRY(hipMemcpy(lp->perm, new_of_old_host, (size_t)lp->n * sizeof(int32_t), hipMemcpyHostToDevice));
    lp->perm_host.assign(new_of_old_host, new_of_old_host + lp->n);
    (void)hipFree(lp->bl_ptr);
    (void)hipFree(lp->bl_indices);
    (void)hipFree(lp->bl_data);
    (void)hipFree(lp->bl_base_dev);
    lp->bl_ptr = nullptr, lp->bl_indices = nullptr, lp->bl_data = nullptr, lp->bl_base_dev = nullptr;
    lp->nblk = 1;
    return SSW_OK;
}

// create + set_permutation in one call, without building a column-blocked copy that would be dropped again
ssw_status ssw_labelprop_create_ordered(int32_t device, int64_t n, const int64_t *indptr_host, const int32_t *indices_host,
                                        const double *data_host, const double *weight_sum_host, const int32_t *new_of_old_host,
                                        ssw_lp **out) {
    SSW_REQUIRE(weight_sum_host != nullptr, "create_ordered: the column sums must come from the matrix in its original order");
    g_lp_skip_blocked = true;
    const ssw_status st = ssw_labelprop_create(device, n, indptr_host, indices_host, data_host, weight_sum_host, out);
    g_lp_skip_blocked = false;
    if (st != SSW_OK) return st;
    const ssw_status st2 = ssw_labelprop_set_permutation(*out, new_of_old_host);
    if (st2 != SSW_OK) {
        ssw_labelprop_destroy(*out);
        *out = nullptr;
    }
    return st2;
}

ssw_status ssw_labelprop_set_prior(ssw_lp *lp, const double *prior_host) {
    SSW_REQUIRE(lp != nullptr && prior_host != nullptr, "NULL argument");
    DeviceGuard guard(lp->device);
    double lo = 0.0, hi = 1.0;
    for (int64_t i = 0; i < lp->n; ++i) {
        const double v = prior_host[i];
        if (v < lo) lo = v;
        if (v > hi) hi = v;
    }
    if (lp->perm) {
        SSW_HIP_TRY(hipMemcpyAsync(lp->scratch, prior_host, (size_t)lp->n * sizeof(double), hipMemcpyHostToDevice, lp->stream));
        hipLaunchKernelGGL(k_lp_permute_in, dim3((unsigned)((lp->n + 255) / 256)), dim3(256), 0, lp->stream, lp->scratch,
                           lp->perm, lp->n, lp->prior);
        SSW_HIP_TRY(hipGetLastError());
    } else {
        SSW_HIP_TRY(hipMemcpyAsync(lp->prior, prior_host, (size_t)lp->n * sizeof(double), hipMemcpyHostToDevice,
                                   lp->stream));
    }
    SSW_HIP_TRY(hipStreamSynchronize(lp->stream));
    SSW_TRY(lp_ensure_transpose(lp));  // first session on this handle: what the incremental updates walk (0.1 s at 18 M non-zeros)
    SSW_TRY(lp_session_buffers(lp));   // ... and what its first propagation would otherwise allocate inside a timed round
    lp->prior_lo = lo;
    lp->prior_hi = hi;
    lp->prior_installed = true;
    lp->trk.valid = false;  // the kept iterates belong to the prior that was just replaced
    return SSW_OK;
}

// ---- the tracked run behind ssw_labelprop_run_resident (see k_inc_seed / k_inc_expand / k_inc_rows above) ----------
static ssw_status lp_ensure_level(ssw_lp *lp, int k) {
    const int64_t nb = (lp->n + 255) / 256;
    if (!lp->f[k]) SSW_HIP_TRY(hipMalloc((void **)&lp->f[k], (size_t)lp->n * sizeof(double) + 16));
    if (!lp->bmax[k]) SSW_HIP_TRY(hipMalloc((void **)&lp->bmax[k], (size_t)nb * sizeof(double) + 16));
    return SSW_OK;
}

static ssw_status lp_inc_reserve(ssw_lp *lp, int64_t words) {
    if (words <= lp->inc_cap) return SSW_OK;
    SSW_HIP_TRY(hipStreamSynchronize(lp->stream));
    (void)hipFree(lp->inc_dev);
    if (lp->inc_host) (void)hipHostFree(lp->inc_host);
    lp->inc_dev = nullptr, lp->inc_host = nullptr, lp->inc_cap = 0;
    int64_t cap = 1 << 16;
    while (cap < words) cap <<= 1;
    SSW_HIP_TRY(hipMalloc((void **)&lp->inc_dev, (size_t)cap * sizeof(int64_t)));
    SSW_HIP_TRY(hipHostMalloc((void **)&lp->inc_host, (size_t)cap * sizeof(int64_t), hipHostMallocDefault));
    lp->inc_cap = cap;
    return SSW_OK;
}

// rows of the transposed pattern: ht row j lists the rows i with W[i][j] != 0, i.e. the rows whose sum reads f[j]
static ssw_status lp_upload_transpose(ssw_lp *lp, const int64_t *indptr_host, const int32_t *indices_host) {
    const int64_t n = lp->n, nnz = lp->nnz, nb = (n + 255) / 256;
    std::vector<int64_t> tp((size_t)n + 1, 0);
    for (int64_t p = 0; p < nnz; ++p) tp[(size_t)indices_host[p] + 1]++;
    for (int64_t j = 0; j < n; ++j) tp[(size_t)j + 1] += tp[(size_t)j];
    std::vector<int32_t> ti((size_t)nnz);
    {
        std::vector<int64_t> fill(tp.begin(), tp.end() - 1);
        for (int64_t i = 0; i < n; ++i)
            for (int64_t p = indptr_host[i]; p < indptr_host[i + 1]; ++p) ti[(size_t)fill[(size_t)indices_host[p]]++] = (int32_t)i;
    }
    lp->cap_rows = (int)std::min<int64_t>(std::max<int64_t>(4096, n / 8), (int64_t)1 << 30);
    SSW_HIP_TRY(hipMalloc((void **)&lp->ht_indptr, (size_t)(n + 1) * sizeof(int64_t) + 16));
    SSW_HIP_TRY(hipMalloc((void **)&lp->ht_indices, (size_t)nnz * sizeof(int32_t) + 16));
    SSW_HIP_TRY(hipMalloc((void **)&lp->stamp, (size_t)n * sizeof(uint32_t) + 16));
    SSW_HIP_TRY(hipMalloc((void **)&lp->bstamp, (size_t)nb * sizeof(uint32_t) + 16));
    SSW_HIP_TRY(hipMalloc((void **)&lp->mem, (size_t)lp->cap_rows * sizeof(int32_t) + 16));
    SSW_HIP_TRY(hipMalloc((void **)&lp->blk, (size_t)nb * sizeof(int32_t) + 16));
    SSW_HIP_TRY(hipMemcpy(lp->ht_indptr, tp.data(), (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    SSW_HIP_TRY(hipMemcpy(lp->ht_indices, ti.data(), (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice));
    SSW_HIP_TRY(hipMemset(lp->stamp, 0, (size_t)n * sizeof(uint32_t)));
    SSW_HIP_TRY(hipMemset(lp->bstamp, 0, (size_t)nb * sizeof(uint32_t)));
    lp->epoch = 0;
    return SSW_OK;
}

// Allocations a session's propagations need, made by ssw_labelprop_set_prior (the per-session call) instead of by the
// first round that propagates (round 5: that round was 3 ms of a 0.46 ms mean -- hipMalloc / hipHostMalloc of the kept
// iterates, the block maxima and the packed label lists): iterates 0 .. 4 (a run of the benchmark's graph needs 2-3
// sweeps and a run without a hint issues 4 before it looks; deeper levels still come on demand), the label-list staging, the fused round's pinned control block.
static ssw_status lp_session_buffers(ssw_lp *lp) {
    for (int k = 1; k <= 4; ++k) SSW_TRY(lp_ensure_level(lp, k));
    SSW_TRY(lp_inc_reserve(lp, 1 << 16));
    if (lp->ids_cap == 0) SSW_TRY(lp_reserve_labels(lp, std::vector<int64_t>(1)));
    if (!lp->round_host) {
        SSW_HIP_TRY(hipHostMalloc((void **)&lp->round_host, sizeof(LpInc), hipHostMallocMapped | hipHostMallocCoherent));
        memset(lp->round_host, 0, sizeof(LpInc));
    }
    return SSW_OK;
}

// built once per handle, from the pattern as the device holds it (device order under a permutation), and not at all
// when the incremental path is switched off
static ssw_status lp_ensure_transpose(ssw_lp *lp) {
    if (lp->ht_indptr || getenv("SSW_LP_NO_INCREMENTAL")) return SSW_OK;
    std::vector<int64_t> ip((size_t)lp->n + 1);
    std::vector<int32_t> ix((size_t)lp->nnz);
    SSW_HIP_TRY(hipMemcpy(ip.data(), lp->indptr, ip.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (lp->nnz) SSW_HIP_TRY(hipMemcpy(ix.data(), lp->indices, ix.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    return lp_upload_transpose(lp, ip.data(), ix.data());
}

static int lp_iter_buf(int k) {  // buffer of iterate k: kept one by one below KEEP, the last two alternate beyond
    return k < ssw_lp::KEEP ? k : ssw_lp::KEEP - 2 + ((k - (ssw_lp::KEEP - 2)) & 1);
}

// room for `ids.size()` installed labels in lp->ids / lp->vals (the device-side list the other entry points clear by)
static ssw_status lp_reserve_labels(ssw_lp *lp, const std::vector<int64_t> &ids) {
    const int64_t m = (int64_t)ids.size();
    if (m > lp->ids_cap) {
        // The installed list (n_labels_installed entries) is what the full path clears is_label[] by: it moves into the
        // larger buffers (freeing it here left the clear kernel reading uninitialised ids past 1024 labels).
        SSW_HIP_TRY(hipStreamSynchronize(lp->stream));
        int64_t cap = 1024;
        while (cap < m) cap <<= 1;
        int64_t *nids = nullptr;
        double *nvals = nullptr;
        SSW_HIP_TRY(hipMalloc((void **)&nids, (size_t)cap * sizeof(int64_t)));
        if (hipMalloc((void **)&nvals, (size_t)cap * sizeof(double)) != hipSuccess) {
            (void)hipFree(nids);
            set_error("label propagation: out of device memory for %lld labels", (long long)cap);
            return SSW_ERR_NOMEM;
        }
        const int64_t keep = std::min<int64_t>(lp->n_labels_installed, lp->ids_cap);
        if (keep > 0 && lp->ids && lp->vals) {
            SSW_HIP_TRY(hipMemcpy(nids, lp->ids, (size_t)keep * sizeof(int64_t), hipMemcpyDeviceToDevice));
            SSW_HIP_TRY(hipMemcpy(nvals, lp->vals, (size_t)keep * sizeof(double), hipMemcpyDeviceToDevice));
        }
        (void)hipFree(lp->ids);
        (void)hipFree(lp->vals);
        lp->ids = nids, lp->vals = nvals;
        lp->ids_cap = cap;
    }
    return SSW_OK;
}

static ssw_status lp_run_tracked(ssw_lp *lp, const int64_t *label_ids, const double *label_vals, int64_t n_labels,
                                 double reg_lambda, double eps, int32_t max_iter, LpState *st_out, LpRoundTail *tail = nullptr) {
    SSW_REQUIRE(reg_lambda >= 0.0, "reg_lambda < 0");
    SSW_REQUIRE(max_iter >= 0, "max_iter < 0");
    SSW_REQUIRE(n_labels == 0 || (label_ids && label_vals), "NULL labels");
    for (int64_t i = 0; i < n_labels; ++i)
        SSW_REQUIRE(label_ids[i] >= 0 && label_ids[i] < lp->n, "label id %lld out of range", (long long)label_ids[i]);
    const int64_t n = lp->n, nb = (n + 255) / 256;
    hipStream_t s = lp->stream;
    const double lo = lp->prior_lo, hi = lp->prior_hi;
    const auto t_entry = std::chrono::steady_clock::now();
    // the labels in device positions, ascending; duplicate ids (numpy: the last assignment wins) take the untracked path
    const int64_t *mapped = lp_map_ids(lp, label_ids, n_labels);
    std::vector<std::pair<int64_t, double>> lab((size_t)n_labels);
    for (int64_t i = 0; i < n_labels; ++i) lab[(size_t)i] = {mapped[i], label_vals[i]};
    std::sort(lab.begin(), lab.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    bool unique = true;
    for (size_t i = 1; i < lab.size(); ++i) unique = unique && lab[i].first != lab[i - 1].first;
    const bool inc_off = getenv("SSW_LP_NO_INCREMENTAL") != nullptr;  // A/B and tests: every call runs the full sweeps (read per call)
    if (!unique || max_iter == 0)
        return lp_run_core(lp, nullptr, true, nullptr, label_ids, label_vals, n_labels, reg_lambda, eps, max_iter, st_out);
    std::vector<int64_t> ids((size_t)n_labels);
    std::vector<double> vals((size_t)n_labels);
    for (size_t i = 0; i < lab.size(); ++i) ids[i] = lab[i].first, vals[i] = lab[i].second;
    SSW_TRY(lp_reserve_labels(lp, ids));
    ssw_lp::Track &tk = lp->trk;
    LpState st;
    memset(&st, 0, sizeof(st));

    bool inc = !inc_off && lp->ht_indptr != nullptr && tk.valid && tk.lambda == reg_lambda && tk.eps == eps && tk.max_iter == max_iter && tk.levels >= 1;
    int continue_from = 0;  // > 0: the incremental pass brought iterates 0 .. continue_from up to date without converging
    if (inc) {
        // ---- what changed: labels set (new, or another value) and labels removed
        std::vector<int64_t> set_ids, unset_ids;
        std::vector<double> set_vals;
        size_t a = 0, b = 0;
        while (a < ids.size() || b < tk.ids.size()) {
            if (b == tk.ids.size() || (a < ids.size() && ids[a] < tk.ids[b])) {
                set_ids.push_back(ids[a]), set_vals.push_back(vals[a]), ++a;
            } else if (a == ids.size() || tk.ids[b] < ids[a]) {
                unset_ids.push_back(tk.ids[b]), ++b;
            } else {
                if (memcmp(&vals[a], &tk.vals[b], sizeof(double)) != 0) set_ids.push_back(ids[a]), set_vals.push_back(vals[a]);
                ++a, ++b;
            }
        }
        if (set_ids.empty() && unset_ids.empty()) {  // the same labels: the kept iterates ARE the answer
            st.sweeps = tk.sweeps, st.done = 1, st.result_buf = tk.result;
            lp->last_result = tk.result;
            lp->info[0] = 1, lp->info[1] = st.sweeps, lp->info[2] = 0, lp->info[3] = 0, lp->info[4] = 0, lp->info[5] = tk.levels;
            *st_out = st;
            return SSW_OK;
        }
        // ---- frontier on the device: the changed labels seed the row list; sweep k first extends it by the in-neighbours
        // of the rows that joined one sweep earlier, then recomputes every listed row and the maxima of the touched blocks;
        // the nested sets make ONE list serve every sweep by its prefix.  Nothing but the label lists is uploaded.
        const int64_t n_set = (int64_t)set_ids.size(), n_unset = (int64_t)unset_ids.size();
        if (n_set + n_unset > lp->cap_rows) inc = false;
        if (inc) {
            if (++lp->epoch == 0) {  // (wrapped)
                SSW_HIP_TRY(hipMemsetAsync(lp->stamp, 0, (size_t)n * sizeof(uint32_t), s));
                SSW_HIP_TRY(hipMemsetAsync(lp->bstamp, 0, (size_t)nb * sizeof(uint32_t), s));
                lp->epoch = 1;
            }
            const uint32_t ep = lp->epoch;
            // one packed upload: [set ids | unset ids | set values | all ids | all values], then the control block
            const int64_t o_ch = 0, o_sv = n_set + n_unset, o_ids = o_sv + n_set, o_vals = o_ids + n_labels, words = o_vals + n_labels + 64;
            SSW_TRY(lp_inc_reserve(lp, words));
            int64_t *hb = lp->inc_host;
            for (int64_t i = 0; i < n_set; ++i) hb[o_ch + i] = set_ids[(size_t)i];
            for (int64_t i = 0; i < n_unset; ++i) hb[o_ch + n_set + i] = unset_ids[(size_t)i];
            memcpy(hb + o_sv, set_vals.data(), (size_t)n_set * sizeof(double));
            memcpy(hb + o_ids, ids.data(), (size_t)n_labels * sizeof(int64_t));
            memcpy(hb + o_vals, vals.data(), (size_t)n_labels * sizeof(double));
            const int64_t o_ctl = o_vals + n_labels;  // the control block travels with the lists (64 words of slack hold its 128 bytes)
            LpInc *hctl = reinterpret_cast<LpInc *>(hb + o_ctl);
            memset(hctl, 0, sizeof(LpInc));
            hctl->total = (int)(n_set + n_unset);
            hctl->m[0] = (int)(n_set + n_unset);
            SSW_HIP_TRY(hipMemcpyAsync(lp->inc_dev, hb, (size_t)(o_ctl + 16) * sizeof(int64_t), hipMemcpyHostToDevice, s));
            int64_t *db = lp->inc_dev;
            LpInc *dctl = reinterpret_cast<LpInc *>(db + o_ctl);
            hipLaunchKernelGGL(k_inc_seed, dim3((unsigned)((std::max<int64_t>(n_set + n_unset, n_labels) + 255) / 256)), dim3(256), 0, s,
                               db + o_ch, reinterpret_cast<const double *>(db + o_sv), (int)n_set, (int)n_unset, lp->prior, lp->f[0],
                               lp->is_label, lp->label_val, lp->stamp, ep, lp->mem, db + o_ids,
                               reinterpret_cast<const double *>(db + o_vals), (int)n_labels, lp->ids, lp->vals);
            // from here on the device holds THESE labels: whatever happens below (an error return included), the kept
            // state describes them or nothing -- it is valid again only where a branch says so
            const int kept_levels = tk.levels;
            tk.valid = false;
            tk.ids = ids, tk.vals = vals;
            LpBmaxPtrs bm;
            for (int k = 0; k < ssw_lp::KEEP; ++k) bm.p[k] = lp->bmax[k];
            for (int k = 1; k <= tk.levels; ++k) {
                hipLaunchKernelGGL(k_inc_expand, dim3(1024), dim3(256), 0, s, k, lp->ht_indptr, lp->ht_indices, lp->stamp, ep, lp->mem,
                                   lp->cap_rows, dctl);
                hipLaunchKernelGGL(k_inc_rows, dim3(2048), dim3(256), 0, s, k, lp->mem, lp->cap_rows, lp->indptr, lp->indices, lp->data,
                                   lp->wsum, lp->prior, lp->f[k - 1], lp->f[k], lp->is_label, lp->label_val, reg_lambda, lo, hi,
                                   lp->bstamp, ep, lp->blk, dctl);
                hipLaunchKernelGGL(k_lp_blockmax, dim3((unsigned)std::min<int64_t>(nb, 2048)), dim3(256), 0, s, lp->blk, dctl, n,
                                   lp->f[k], lp->f[k - 1], lp->bmax[k], (const LpState *)nullptr);
            }
            hipLaunchKernelGGL(k_lp_levelmax, dim3((unsigned)tk.levels), dim3(1024), 0, s, bm, nb, dctl);
            SSW_HIP_TRY(hipGetLastError());
            lp->n_labels_installed = n_labels;  // (k_inc_seed refreshed lp->ids / lp->vals)
            LpInc rctl;
            if (tail) {
                // fused round: the scores kernel picks the converged iterate itself and leaves the control block in pinned
                // memory, the selection follows on the same stream, and the host waits ONCE -- on the selection's word
                LpFPtrs fp;
                for (int k = 0; k < ssw_lp::KEEP; ++k) fp.p[k] = lp->f[k];
                LpInc *host_view = nullptr;
                SSW_HIP_TRY(hipHostGetDevicePointer((void **)&host_view, lp->round_host, 0));
                lp->round_host->m[8] = -1;
                hipLaunchKernelGGL(k_lp_scores_sel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, fp, dctl, tk.levels, eps,
                                   tail->mask_labeled ? lp->is_label : (const unsigned char *)nullptr, (const int32_t *)lp->perm, n,
                                   tail->index_scores, host_view);
                SSW_HIP_TRY(hipGetLastError());
                SSW_TRY(index_enqueue_topk_resident(tail->index, s, tail->excluded, tail->n_excluded, tail->k));
                const auto t_wait0 = std::chrono::steady_clock::now();
                lp->info[6] = (int64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t_wait0 - t_entry).count();
                SSW_TRY(index_collect_topk(tail->index, s, tail->k, tail->out_images, tail->out_scores, tail->out_best_rows,
                                           tail->out_count));
                lp->info[7] = (int64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_wait0).count();
                memcpy(&rctl, lp->round_host, sizeof(LpInc));
                if (rctl.m[8] < 0) {  // (the word the kernel always writes did not arrive: should not happen -- wait the plain way)
                    SSW_HIP_TRY(hipStreamSynchronize(s));
                    memcpy(&rctl, lp->round_host, sizeof(LpInc));
                    SSW_REQUIRE(rctl.m[8] >= 0, "labelprop round: the scores kernel did not publish its control block");
                }
                tail->valid = rctl.m[8] > 0;
            } else {
                SSW_HIP_TRY(hipMemcpyAsync(hctl, dctl, sizeof(LpInc), hipMemcpyDeviceToHost, s));  // ONE copy back: the control block
                const auto t_wait0 = std::chrono::steady_clock::now();
                lp->info[6] = (int64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t_wait0 - t_entry).count();
                SSW_HIP_TRY(hipStreamSynchronize(s));
                lp->info[7] = (int64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_wait0).count();
                memcpy(&rctl, hctl, sizeof(LpInc));
            }
            LpState dst;
            memset(&dst, 0, sizeof(dst));
            dst.bound_violation = rctl.bound_violation;
            if (rctl.overflow) {
                // the change reaches more than n / 8 rows: rows beyond the list were not recomputed, so the kept iterates
                // are no longer a run's -- start over with the full sweeps (from the labels just installed)
                inc = false;
            } else if (dst.bound_violation) {
                tk.valid = false;
                set_error("label propagation: averaged scores left [%g, %g] (label_propagation.py:39-40)", lo, hi);
                return SSW_ERR_NUMERIC;
            } else {
                int64_t rows_total = 0;
                for (int k = 1; k <= tk.levels; ++k) rows_total += rctl.m[k];
                int conv = 0;
                for (int k = 1; k <= tk.levels && !conv; ++k)
                    if (rctl.level_max[k] < eps) conv = k;
                if (conv) {
                    st.sweeps = conv, st.done = 1, st.result_buf = conv - 1;
                    tk.valid = true, tk.levels = kept_levels;
                    tk.sweeps = conv, tk.result = conv - 1;
                    lp->last_result = conv - 1;
                    lp->sweeps_hint = conv;
                    lp->info[0] = 1, lp->info[1] = conv, lp->info[2] = 2 + 3 * (int64_t)tk.levels, lp->info[3] = 1, lp->info[4] = rows_total,
                    lp->info[5] = tk.levels;
                    *st_out = st;
                    return SSW_OK;
                }
                // the new labels need more sweeps than were kept: iterates 0 .. levels ARE the new run's (every row either
                // unchanged or recomputed), so the full sweeps below continue from sweep levels + 1
                continue_from = tk.levels;
            }
        }
    }

    // ---- full sweeps, every iterate and its block maxima kept (the state the next call updates)
    tk.valid = false;
    if (continue_from > 0) {
        LpState init;
        memset(&init, 0, sizeof(init));
        init.sweeps = continue_from;
        init.result_buf = lp_iter_buf(continue_from);
        SSW_HIP_TRY(hipMemcpyAsync(lp->state, &init, sizeof(LpState), hipMemcpyHostToDevice, s));
        SSW_HIP_TRY(hipStreamSynchronize(s));  // (`init` is a stack object)
    } else {
        SSW_HIP_TRY(hipMemcpyAsync(lp->f[0], lp->prior, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
        if (lp->n_labels_installed > 0) {
            const int64_t m = lp->n_labels_installed;
            hipLaunchKernelGGL(k_lp_clear_labels, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, lp->is_label, lp->ids, m);
            lp->n_labels_installed = 0;
        }
        if (n_labels > 0) {
            SSW_TRY(lp_inc_reserve(lp, 2 * n_labels + 64));
            memcpy(lp->inc_host, ids.data(), (size_t)n_labels * sizeof(int64_t));
            memcpy(lp->inc_host + n_labels, vals.data(), (size_t)n_labels * sizeof(double));
            SSW_HIP_TRY(hipMemcpyAsync(lp->ids, lp->inc_host, (size_t)n_labels * sizeof(int64_t), hipMemcpyHostToDevice, s));
            SSW_HIP_TRY(hipMemcpyAsync(lp->vals, lp->inc_host + n_labels, (size_t)n_labels * sizeof(double), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_lp_apply_labels, dim3((unsigned)((n_labels + 255) / 256)), dim3(256), 0, s, lp->f[0], lp->is_label,
                               lp->label_val, lp->ids, lp->vals, n_labels);
            lp->n_labels_installed = n_labels;
        }
        SSW_HIP_TRY(hipMemsetAsync(lp->state, 0, sizeof(LpState), s));
    }
    int issued = continue_from, syncs = 0;
    if (continue_from >= max_iter) {
        // (max_iter <= kept levels: the incremental pass already produced iterate max_iter; the reference returns it
        // after max_iter sweeps, not converged -- ADVICE r5)
        st.sweeps = continue_from, st.done = 0, st.result_buf = lp_iter_buf(continue_from);
    }
    int batch = continue_from > 0 ? 2 : (lp->sweeps_hint > 0 ? std::min(8, lp->sweeps_hint + 1) : 4);
    while (issued < max_iter) {
        const int upto = (issued + batch < max_iter) ? issued + batch : max_iter;
        batch = 8;
        for (; issued < upto; ++issued) {
            const int k = issued + 1;  // this sweep produces iterate k
            const int src = lp_iter_buf(k - 1), dst = lp_iter_buf(k);
            if (k < ssw_lp::KEEP) SSW_TRY(lp_ensure_level(lp, k));
            lp_launch_sweep(lp, src, dst, reg_lambda, lo, hi);
            if (k < ssw_lp::KEEP)
                hipLaunchKernelGGL(k_lp_blockmax, dim3((unsigned)nb), dim3(256), 0, s, (const int32_t *)nullptr, (const LpInc *)nullptr, n,
                                   lp->f[dst], lp->f[src], lp->bmax[k], (const LpState *)lp->state);
            hipLaunchKernelGGL(k_lp_check, dim3(1), dim3(1), 0, s, lp->state, eps, src, dst);
        }
        SSW_HIP_TRY(hipGetLastError());
        SSW_HIP_TRY(hipMemcpyAsync(&st, lp->state, sizeof(LpState), hipMemcpyDeviceToHost, s));
        SSW_HIP_TRY(hipStreamSynchronize(s));
        ++syncs;
        if (st.done) break;
    }
    if (st.bound_violation) {
        set_error("label propagation: averaged scores left [%g, %g] (label_propagation.py:39-40)", lo, hi);
        return SSW_ERR_NUMERIC;
    }
    lp->last_result = (st.sweeps > 0) ? st.result_buf : 0;
    lp->sweeps_hint = st.done ? (int)st.sweeps : 0;
    if (st.done && st.sweeps >= 1 && st.sweeps < ssw_lp::KEEP) {
        tk.valid = true;
        tk.levels = st.sweeps, tk.sweeps = st.sweeps, tk.result = st.result_buf;
        tk.lambda = reg_lambda, tk.eps = eps, tk.max_iter = max_iter;
        tk.ids = ids, tk.vals = vals;
    }
    // (a run that continued an incremental pass reports mode 2; its pass's launches / rows are not added here)
    lp->info[0] = continue_from > 0 ? 2 : 0, lp->info[1] = st.sweeps, lp->info[2] = 3 * (int64_t)(issued - continue_from) + 3,
    lp->info[3] = syncs + (continue_from > 0 ? 2 : 0), lp->info[4] = (int64_t)(issued - continue_from) * n, lp->info[5] = tk.valid ? tk.levels : 0;
    *st_out = st;
    return SSW_OK;
}

ssw_status ssw_labelprop_run_resident(ssw_lp *lp, const int64_t *label_ids, const double *label_vals,
                                      int64_t n_labels, double reg_lambda, double eps, int32_t max_iter,
                                      int32_t *out_sweeps, int32_t *out_converged) {
    SSW_REQUIRE(lp != nullptr, "NULL argument");
    SSW_REQUIRE(lp->prior_installed, "ssw_labelprop_run_resident: no prior installed (ssw_labelprop_set_prior)");
    DeviceGuard guard(lp->device);
    LpState st;
    SSW_TRY(lp_run_tracked(lp, label_ids, label_vals, n_labels, reg_lambda, eps, max_iter, &st));
    if (out_sweeps) *out_sweeps = st.sweeps;
    if (out_converged) *out_converged = st.done;
    return SSW_OK;
}

/* what the last propagation of this handle did: out[0] = 1 if it was an incremental update (0: full sweeps; 2: an
 * incremental pass over the kept iterates that did not converge, continued by full sweeps from there),
 * [1] sweeps (as the reference counts them), [2] kernel launches, [3] host synchronisations, [4] rows recomputed over all
 * sweeps, [5] iterates kept for the next call, [6] ns of host time in the frontier walk, [7] ns waited for the device (incremental runs) */
ssw_status ssw_labelprop_last_run_info(ssw_lp *lp, int64_t *out8) {
    SSW_REQUIRE(lp != nullptr && out8 != nullptr, "NULL argument");
    memcpy(out8, lp->info, sizeof(lp->info));
    return SSW_OK;
}

static ssw_status lp_prior_as_result(ssw_lp *lp, const int64_t *label_ids, int64_t n_labels, bool wait) {
    SSW_REQUIRE(lp != nullptr, "NULL argument");
    SSW_REQUIRE(lp->prior_installed, "ssw_labelprop_prior_as_result: no prior installed (ssw_labelprop_set_prior)");
    SSW_REQUIRE(n_labels == 0 || label_ids, "NULL labels");
    for (int64_t i = 0; i < n_labels; ++i)
        SSW_REQUIRE(label_ids[i] >= 0 && label_ids[i] < lp->n, "label id %lld out of range", (long long)label_ids[i]);
    DeviceGuard guard(lp->device);
    hipStream_t s = lp->stream;
    lp->trk.valid = false;
    SSW_HIP_TRY(hipMemcpyAsync(lp->f[0], lp->prior, (size_t)lp->n * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (lp->n_labels_installed > 0) {
        const int64_t m = lp->n_labels_installed;
        hipLaunchKernelGGL(k_lp_clear_labels, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, lp->is_label, lp->ids, m);
        lp->n_labels_installed = 0;
    }
    if (n_labels > 0) {
        if (n_labels > lp->ids_cap) {
            SSW_HIP_TRY(hipStreamSynchronize(s));
            (void)hipFree(lp->ids);
            (void)hipFree(lp->vals);
            lp->ids = nullptr;
            lp->vals = nullptr;
            int64_t cap = 1024;
            while (cap < n_labels) cap <<= 1;
            SSW_HIP_TRY(hipMalloc((void **)&lp->ids, (size_t)cap * sizeof(int64_t)));
            SSW_HIP_TRY(hipMalloc((void **)&lp->vals, (size_t)cap * sizeof(double)));
            lp->ids_cap = cap;
        }
        SSW_HIP_TRY(hipMemcpyAsync(lp->ids, lp_map_ids(lp, label_ids, n_labels), (size_t)n_labels * sizeof(int64_t),
                                   hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_lp_mark_labels, dim3((unsigned)((n_labels + 255) / 256)), dim3(256), 0, s, lp->is_label,
                           lp->ids, n_labels);
        lp->n_labels_installed = n_labels;
    }
    SSW_HIP_TRY(hipGetLastError());
    if (wait) SSW_HIP_TRY(hipStreamSynchronize(s));
    lp->last_result = 0;
    return SSW_OK;
}

ssw_status ssw_labelprop_prior_as_result(ssw_lp *lp, const int64_t *label_ids, int64_t n_labels) {
    return lp_prior_as_result(lp, label_ids, n_labels, true);
}

ssw_status ssw_labelprop_fetch(ssw_lp *lp, double *out_f_host) {
    SSW_REQUIRE(lp != nullptr && out_f_host != nullptr, "NULL argument");
    SSW_REQUIRE(lp->last_result >= 0, "ssw_labelprop_fetch: nothing has been propagated yet");
    DeviceGuard guard(lp->device);
    const double *res = lp->f[lp->last_result];
    if (lp->perm) {
        hipLaunchKernelGGL(k_lp_permute_out, dim3((unsigned)((lp->n + 255) / 256)), dim3(256), 0, lp->stream, res, lp->perm,
                           lp->n, lp->scratch);
        res = lp->scratch;
    }
    SSW_HIP_TRY(hipMemcpyAsync(out_f_host, res, (size_t)lp->n * sizeof(double), hipMemcpyDeviceToHost, lp->stream));
    SSW_HIP_TRY(hipStreamSynchronize(lp->stream));
    return SSW_OK;
}

// f[rows[i]] for a short list of nodes: what makeXy (loops/util.py:19) needs of the propagated scores -- the
// pseudo-labels of `sample_size` nodes -- without moving the whole [n] f64 iterate over PCIe
ssw_status ssw_labelprop_gather(ssw_lp *lp, const int64_t *rows_host, int64_t m, double *out_host) {
    SSW_REQUIRE(lp != nullptr && m >= 0 && (m == 0 || (rows_host != nullptr && out_host != nullptr)), "bad argument");
    SSW_REQUIRE(lp->last_result >= 0, "ssw_labelprop_gather: nothing has been propagated yet");
    if (m == 0) return SSW_OK;
    for (int64_t i = 0; i < m; ++i)
        SSW_REQUIRE(rows_host[i] >= 0 && rows_host[i] < lp->n, "ssw_labelprop_gather: node %lld outside [0, %lld)",
                    (long long)rows_host[i], (long long)lp->n);
    DeviceGuard guard(lp->device);
    if (m > lp->g_cap) {
        SSW_HIP_TRY(hipStreamSynchronize(lp->stream));
        (void)hipFree(lp->g_rows);
        (void)hipFree(lp->g_vals);
        if (lp->g_rows_host) (void)hipHostFree(lp->g_rows_host);
        if (lp->g_vals_host) (void)hipHostFree(lp->g_vals_host);
        lp->g_rows = nullptr, lp->g_vals = nullptr, lp->g_rows_host = nullptr, lp->g_vals_host = nullptr, lp->g_cap = 0;
        int64_t cap = 16384;
        while (cap < m) cap <<= 1;
        SSW_HIP_TRY(hipMalloc((void **)&lp->g_rows, (size_t)cap * sizeof(int64_t)));
        SSW_HIP_TRY(hipMalloc((void **)&lp->g_vals, (size_t)cap * sizeof(double)));
        SSW_HIP_TRY(hipHostMalloc((void **)&lp->g_rows_host, (size_t)cap * sizeof(int64_t), hipHostMallocDefault));
        SSW_HIP_TRY(hipHostMalloc((void **)&lp->g_vals_host, (size_t)cap * sizeof(double), hipHostMallocDefault));
        lp->g_cap = cap;
    }
    if (lp->perm)
        for (int64_t i = 0; i < m; ++i) lp->g_rows_host[i] = lp->perm_host[(size_t)rows_host[i]];
    else
        memcpy(lp->g_rows_host, rows_host, (size_t)m * sizeof(int64_t));
    SSW_HIP_TRY(hipMemcpyAsync(lp->g_rows, lp->g_rows_host, (size_t)m * sizeof(int64_t), hipMemcpyHostToDevice, lp->stream));
    hipLaunchKernelGGL(k_lp_gather, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, lp->stream,
                       lp->f[lp->last_result], lp->g_rows, m, lp->g_vals);
    SSW_HIP_TRY(hipGetLastError());
    SSW_HIP_TRY(hipMemcpyAsync(lp->g_vals_host, lp->g_vals, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, lp->stream));
    SSW_HIP_TRY(hipStreamSynchronize(lp->stream));
    memcpy(out_host, lp->g_vals_host, (size_t)m * sizeof(double));
    return SSW_OK;
}

ssw_status ssw_labelprop_device_scores(ssw_lp *lp, const double **out_dev_scores) {
    SSW_REQUIRE(lp != nullptr && out_dev_scores != nullptr, "NULL argument");
    SSW_REQUIRE(lp->last_result >= 0, "ssw_labelprop_device_scores: nothing has been propagated yet");
    if (lp->perm) {  // callers index it by original node id: hand out an un-permuted copy (valid until the next call)
        DeviceGuard guard(lp->device);
        hipLaunchKernelGGL(k_lp_permute_out, dim3((unsigned)((lp->n + 255) / 256)), dim3(256), 0, lp->stream,
                           lp->f[lp->last_result], lp->perm, lp->n, lp->scratch);
        SSW_HIP_TRY(hipGetLastError());
        SSW_HIP_TRY(hipStreamSynchronize(lp->stream));
        *out_dev_scores = lp->scratch;
        return SSW_OK;
    }
    {   // the pointer is read by kernels of OTHER handles' streams (the index's re-scoring, the fit's sample): whatever this
        // handle still has queued -- sweeps turned into no-ops after convergence -- is behind the caller from here on
        DeviceGuard guard(lp->device);
        SSW_HIP_TRY(hipStreamSynchronize(lp->stream));
    }
    *out_dev_scores = lp->f[lp->last_result];
    return SSW_OK;
}

ssw_status ssw_labelprop_scores_to_index(ssw_lp *lp, ssw_index *index, int32_t mask_labeled) {
    SSW_REQUIRE(lp != nullptr && index != nullptr, "NULL argument");
    SSW_REQUIRE(lp->last_result >= 0, "ssw_labelprop_scores_to_index: nothing has been propagated yet");
    int64_t n = 0, n_images = 0;
    int32_t D = 0;
    void *Xv = nullptr, *scores = nullptr;
    SSW_TRY(ssw_index_shape(index, &n, &D, &n_images));
    SSW_TRY(ssw_index_device_ptrs(index, &Xv, &scores));
    SSW_REQUIRE(n == lp->n, "the index has %lld rows, the graph %lld nodes", (long long)n, (long long)lp->n);
    SSW_TRY(ssw_index_sync(index));  // nothing of the index's own stream still writes the score buffer
    DeviceGuard guard(lp->device);
    if (lp->perm)
        hipLaunchKernelGGL(k_lp_scores_f32_perm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, lp->stream,
                           lp->f[lp->last_result], mask_labeled ? lp->is_label : (const unsigned char *)nullptr, lp->perm, n,
                           (float *)scores);
    else
        hipLaunchKernelGGL(k_lp_scores_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, lp->stream, lp->f[lp->last_result],
                           mask_labeled ? lp->is_label : (const unsigned char *)nullptr, n, (float *)scores);
    SSW_HIP_TRY(hipGetLastError());
    SSW_HIP_TRY(hipStreamSynchronize(lp->stream));
    return SSW_OK;
}


/* One feedback round of a graph loop in ONE call (KnnProp2.refine + next_batch, seesaw/loops/graph_based.py:73-121;
 * LabelPropagationRanker2.update, research/knn_methods.py:176-199):
 *   propagate != 0:  ssw_labelprop_run_resident(labels)      propagate == 0:  ssw_labelprop_prior_as_result(label_ids)
 *   then ssw_labelprop_scores_to_index(mask_labeled) and ssw_index_topk(q = NULL, excluded_images, k)
 * with the same results as the three calls.  When the propagation is an incremental update (every round of a session but
 * its first), everything is enqueued back to back on the graph handle's stream -- the scores kernel reads the converged
 * iterate's number from the control block on the device -- and the host waits once, on the selection's sequence word. */
ssw_status ssw_labelprop_round(ssw_lp *lp, ssw_index *index, int32_t propagate, const int64_t *label_ids,
                               const double *label_vals, int64_t n_labels, double reg_lambda, double eps, int32_t max_iter,
                               int32_t mask_labeled, const int64_t *excluded_images, int64_t n_excluded, int32_t k,
                               int64_t *out_images, float *out_scores, int64_t *out_best_rows, int32_t *out_count,
                               int32_t *out_sweeps, int32_t *out_converged) {
    SSW_REQUIRE(lp != nullptr && index != nullptr && out_count != nullptr, "NULL argument");
    SSW_REQUIRE(lp->prior_installed, "ssw_labelprop_round: no prior installed (ssw_labelprop_set_prior)");
    SSW_REQUIRE(k >= 1 && k <= SSW_MAX_TOPK, "k=%d outside [1, %d]", k, SSW_MAX_TOPK);
    SSW_REQUIRE(n_excluded == 0 || excluded_images != nullptr, "excluded_images is NULL");
    *out_count = 0;
    int64_t n = 0, n_images = 0;
    int32_t D = 0;
    void *Xv = nullptr, *scores = nullptr;
    SSW_TRY(ssw_index_shape(index, &n, &D, &n_images));
    SSW_TRY(ssw_index_device_ptrs(index, &Xv, &scores));
    SSW_REQUIRE(n == lp->n, "the index has %lld rows, the graph %lld nodes", (long long)n, (long long)lp->n);
    SSW_REQUIRE(index_device(index) == lp->device, "ssw_labelprop_round: the index lives on device %d, the graph on device %d",
                index_device(index), lp->device);  // (one stream carries both halves of the round)
    for (int64_t i = 0; i < n_excluded; ++i)  // (checked before anything is enqueued)
        SSW_REQUIRE(excluded_images[i] >= 0 && excluded_images[i] < n_images, "excluded image %lld outside [0, %lld)",
                    (long long)excluded_images[i], (long long)n_images);
    SSW_TRY(ssw_index_sync(index));  // nothing of the index's own stream still touches the score buffer / the workspace
    DeviceGuard guard(lp->device);
    SSW_TRY(lp_session_buffers(lp));
    LpRoundTail tail;
    tail.index = index, tail.index_scores = (float *)scores, tail.mask_labeled = mask_labeled != 0;
    tail.excluded = excluded_images, tail.n_excluded = n_excluded, tail.k = k;
    tail.out_images = out_images, tail.out_scores = out_scores, tail.out_best_rows = out_best_rows, tail.out_count = out_count;
    LpState st;
    memset(&st, 0, sizeof(st));
    if (propagate) {
        SSW_TRY(lp_run_tracked(lp, label_ids, label_vals, n_labels, reg_lambda, eps, max_iter, &st, &tail));
    } else {
        lp->info[0] = 3, lp->info[1] = 0, lp->info[2] = 3, lp->info[3] = 0, lp->info[4] = 0, lp->info[5] = 0, lp->info[6] = 0, lp->info[7] = 0;
        SSW_TRY(lp_prior_as_result(lp, label_ids, n_labels, false));
    }
    if (out_sweeps) *out_sweeps = st.sweeps;
    if (out_converged) *out_converged = st.done;
    if (tail.valid) return SSW_OK;
    // the propagation took another path (a session's first run, more sweeps than were kept, ...): its answer is
    // lp->f[lp->last_result] by now -- scores and selection behind it, one more wait
    hipLaunchKernelGGL(k_lp_scores_f32_perm_or_plain, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, lp->stream,
                       lp->f[lp->last_result], mask_labeled ? lp->is_label : (const unsigned char *)nullptr, (const int32_t *)lp->perm, n,
                       (float *)scores);
    SSW_HIP_TRY(hipGetLastError());
    SSW_TRY(index_enqueue_topk_resident(index, lp->stream, excluded_images, n_excluded, k));
    return index_collect_topk(index, lp->stream, k, out_images, out_scores, out_best_rows, out_count);
}

ssw_status ssw_xlx(ssw_index *index, ssw_lp *lap, double *out_host) {
    SSW_REQUIRE(index != nullptr && lap != nullptr && out_host != nullptr, "NULL argument");
    SSW_REQUIRE(lap->perm == nullptr, "ssw_xlx: the Laplacian handle must be in original node order");
    int64_t n = 0, n_images = 0;
    int32_t D = 0;
    void *Xv = nullptr, *scores = nullptr;
    if (ssw_status rc = ssw_index_shape(index, &n, &D, &n_images); rc != SSW_OK) return rc;
    if (ssw_status rc = ssw_index_device_ptrs(index, &Xv, &scores); rc != SSW_OK) return rc;
    SSW_REQUIRE(n == lap->n, "ssw_xlx: the index has %lld rows, the matrix %lld", (long long)n, (long long)lap->n);
    if (D % 128 != 0) {
        set_error("ssw_xlx: dim %d unsupported (multiple of 128)", D);
        return SSW_ERR_UNSUPPORTED;
    }
    if (ssw_status rc = ssw_index_sync(index); rc != SSW_OK) return rc;  // uploads into X are complete
    DeviceGuard guard(lap->device);
    hipStream_t s = lap->stream;
    const int64_t dd = (int64_t)D * D;
    int64_t rows_per_chunk = (n + 127) / 128;
    rows_per_chunk = ((rows_per_chunk < 1024 ? 1024 : rows_per_chunk) + 3) / 4 * 4;
    const int n_chunks = (int)((n + rows_per_chunk - 1) / rows_per_chunk);
    double *Y = nullptr, *P = nullptr, *out = nullptr;
    auto release = [&]() {
        (void)hipFree(Y);
        (void)hipFree(P);
        (void)hipFree(out);
    };
    hipError_t e = hipMalloc((void **)&Y, (size_t)n * D * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&P, (size_t)n_chunks * dd * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&out, (size_t)dd * sizeof(double));
    if (e != hipSuccess) {
        release();
        set_error("ssw_xlx: %s", hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? SSW_ERR_NOMEM : SSW_ERR_HIP;
    }
    hipLaunchKernelGGL(k_xlx_spmm, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, lap->indptr, lap->indices, lap->data,
                       (const float *)Xv, n, (int)D, Y);
    hipLaunchKernelGGL(k_xlx_partial, dim3(D / 128, D / 128, n_chunks), dim3(256), 0, s, (const float *)Xv, Y, n, (int)D,
                       rows_per_chunk, P);
    hipLaunchKernelGGL(k_xlx_reduce, dim3((unsigned)((dd + 255) / 256)), dim3(256), 0, s, P, n_chunks, dd, out);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out_host, out, (size_t)dd * sizeof(double), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    release();
    if (e != hipSuccess) {
        set_error("ssw_xlx: %s", hipGetErrorString(e));
        return SSW_ERR_HIP;
    }
    return SSW_OK;
}

}  // extern "C"
