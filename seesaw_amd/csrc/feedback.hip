// feedback.hip -- online relevance-feedback update: fused loss + gradient kernels and the
// L-BFGS (strong Wolfe) driver that turns labelled tile vectors into the next query vector
// (gfx950 / MI355X).
//
// Replaces, for the two linear scorers the feedback loops fit every round:
//   LogisticRegressionPT / LogisticRegModule     seesaw/logistic_regression.py:68-124,270-421
//       mean_i BCEWithLogits(x_i.w + b, y_i; weight_i, pos_weight) + (lambda/n) R(w),
//       R = (|w|-1)^2 + |w/|w| - q/|q||^2   (vector regulariser, :304-330)
//   RegModule ("seesaw" / MultiReg)              seesaw/loops/multi_reg.py:24-134
//       sum_i sw_i item_i + l_norm (cosh(log w.w) - 1) + l_data w'(X'LX)w + l_query (1 - w^.q^)/2,
//       item = BCE (balanced re-weighting :90-105) | pairwise hinge (:106-112, rank_loss.py:63-95)
//              | pairwise logistic (:113-119, rank_loss.py:34-61)
//   torch.optim.LBFGS(line_search_fn="strong_wolfe") closure loop   seesaw/basic_trainer.py:11-69
//
// The reference evaluates the closure through Python autograd (with anomaly detection on)
// up to 1.25 x max_iter times per refine; the math per evaluation is two skinny GEMVs
// (n x 512, n = labelled tile vectors, up to ~10^4 with pseudo-labels) plus O(n^2) pairwise
// terms.  Two drivers over the same arithmetic (bit-identical fits, tests/test_feedback_gpu.py):
//   * n <= 1024 rows (every feedback session): k_fb_fit_wg -- the whole optimizer.step(closure), i.e. L-BFGS
//     direction updates, strong-Wolfe line search and all closure evaluations, in ONE launch of one workgroup;
//   * larger sets (PseudoLR's 10 000 pseudo-labelled rows) and the rank objective: the host walks the same
//     state machine and launches one evaluation at a time = 3-4 small kernels on rows that already sit in HBM:
//       fb_logits   z = Xc w (+ b)                        wave per row, coalesced 16-B loads
//       fb_pairwise per-item pairwise loss + dL/dz        one workgroup, z and targets in LDS
//       fb_elem     per-item BCE loss + dL/dz             elementwise
//       fb_grad     partial g = Xc' r per 32-row slab     thread per column, coalesced
//       fb_final    fixed-order reduction of the partials + regulariser terms -> [loss, grad]
// Bound: issue / launch latency, not HBM -- nothing here is GEMM-shaped enough for MFMA.  Compiled with
// -ffp-contract=off (Makefile): the host driver and the device driver must round alike, and the expressions that
// are meant to be fused say fmaf.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "ssw_common.h"

namespace ssw {
namespace {

constexpr int FB_SLAB = 32;          // rows per fb_grad workgroup
constexpr int FB_MAX_PAIRWISE = 4096;  // items the one-workgroup pairwise kernel takes

// ---- data preparation ---------------------------------------------------------------
__global__ void k_fb_gather_rows(const float *__restrict__ X, const int64_t *__restrict__ rows,
                                 int64_t n, int dim, float *__restrict__ out) {
    const int64_t i = blockIdx.x;
    const float4 *src = reinterpret_cast<const float4 *>(X + rows[i] * dim);
    float4 *dst = reinterpret_cast<float4 *>(out + i * dim);
    for (int c = threadIdx.x; c < dim / 4; c += blockDim.x) dst[c] = src[c];
}

// column means in f64 (fixed order), then subtract: StandardScaler(with_std=False) /
// `X - X.mean(axis=0)` (logistic_regression.py:353-355, multi_reg.py:168-169).  Three small launches: per-column sums
// of 256-row blocks, the block sums added in block order, the subtraction over all elements.  (One thread walking a
// whole column took 3.7 ms on PseudoLR's 10 000 rows.)
constexpr int FB_CENTER_ROWS = 256;
__global__ void k_fb_colsum_blocks(const float *__restrict__ X, int64_t n, int dim, double *__restrict__ partial) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= dim) return;
    const int64_t r0 = (int64_t)blockIdx.y * FB_CENTER_ROWS, r1 = min(r0 + FB_CENTER_ROWS, n);
    double s = 0.0;
    for (int64_t i0 = r0; i0 < r1; i0 += 32) {  // 32 rows in flight; the additions stay in row order
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = X[min(i0 + u, r1 - 1) * dim + c];
#pragma unroll
        for (int u = 0; u < 32; ++u)
            if (i0 + u < r1) s += (double)v[u];
    }
    partial[(int64_t)blockIdx.y * dim + c] = s;
}
__global__ void k_fb_colmean(const double *__restrict__ partial, int nblk, int64_t n, int dim, float *__restrict__ mu) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= dim) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += partial[(int64_t)b * dim + c];
    mu[c] = (float)(s / (double)n);
}
__global__ void k_fb_subtract_mean(float *__restrict__ X, int64_t n, int dim, const float *__restrict__ mu) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= dim) return;
    const float m = mu[c];
    const int64_t r0 = (int64_t)blockIdx.y * FB_CENTER_ROWS, r1 = min(r0 + FB_CENTER_ROWS, n);
    for (int64_t i0 = r0; i0 < r1; i0 += 32) {
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = X[min(i0 + u, r1 - 1) * dim + c];
#pragma unroll
        for (int u = 0; u < 32; ++u)
            if (i0 + u < r1) X[(i0 + u) * dim + c] = v[u] - m;
    }
}

// PseudoLR's sample on the device (ssw_fb_set_pseudo_sample): thread i takes the p-th unlabelled row for p = drawn[i] --
// p + #{j : th[j] <= p} with th[j] = labelled[j] - j, found by bisection (== np.nonzero(~is_labeled)[0][p]) -- and that row's
// propagated score as its target; rows / y / row ids land behind the n_lab labelled entries
__global__ void k_fb_pseudo_rows(const int64_t *__restrict__ th, int n_lab, const int64_t *__restrict__ drawn, int64_t n_drawn,
                                 const double *__restrict__ scores, int64_t n_scores, int64_t *__restrict__ rows_out,
                                 float *__restrict__ y_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_drawn) return;
    const int64_t p = drawn[i];
    int lo = 0, hi = n_lab;  // first j with th[j] > p
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (th[mid] <= p) lo = mid + 1;
        else hi = mid;
    }
    int64_t pick = p + lo;
    if (pick >= n_scores) pick = n_scores - 1;  // (the host checked p < #unlabelled: cannot happen)
    rows_out[n_lab + i] = pick;
    y_out[n_lab + i] = (float)scores[pick];
}

// ---- per-evaluation kernels -----------------------------------------------------------
// Parameters of one closure evaluation, passed BY VALUE in the kernel-argument segment: the driver
// changes them every evaluation, and a separate 2-KB host-to-device copy per evaluation costs more
// than the kernels it feeds.  (dim <= 768; larger dims take the device buffer.)
constexpr int FB_ARG_FLOATS = 772;
struct FbW {
    float v[FB_ARG_FLOATS];
};

// z_i = <x_i, w> + b ; one wave per row
__global__ __launch_bounds__(256) void k_fb_logits(const float *__restrict__ X,
                                                   const float *__restrict__ w, int64_t n, int dim,
                                                   int has_bias, float *__restrict__ z) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float4 *x4 = reinterpret_cast<const float4 *>(X + row * dim);
    const float4 *w4 = reinterpret_cast<const float4 *>(w);
    float a = 0.f;
    for (int c = lane; c < dim / 4; c += 64) {
        const float4 xv = x4[c], wv = w4[c];
        a = fmaf(xv.x, wv.x, a);
        a = fmaf(xv.y, wv.y, a);
        a = fmaf(xv.z, wv.z, a);
        a = fmaf(xv.w, wv.w, a);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64);
    if (lane == 0) z[row] = a + (has_bias ? w[dim] : 0.f);
}

// The reference's label losses are evaluated in f64 (its targets are float64, so torch
// promotes: logistic_regression.py:393, multi_reg.py:170) while logits and gradients stay
// f32.  L-BFGS stops on |loss - prev_loss| < 1e-9, which an f32 loss cannot resolve, so the
// per-item losses are f64 here too; dL/dz is rounded to f32 like torch's.
__device__ __forceinline__ double softplus_neg(double z) {  // log(1 + exp(-z)), stable
    return log1p(exp(-fabs(z))) + fmax(-z, 0.0);
}
__device__ __forceinline__ double sigmoidd(double z) { return 1.0 / (1.0 + exp(-z)); }
// One BCE-with-logits item the way torch forms it with f32 logits and f64 targets
// (binary_cross_entropy_with_logits: log_sigmoid(input) is an f32 tensor, multiplied IN PLACE by the f64
// log_weight -- so the product is rounded to f32 again -- and only then subtracted from the f64
// (1 - target) * input).  The values differ from the exact ones by ~1e-8 per item; L-BFGS's last line search
// and its |loss - prev_loss| < 1e-9 stop sit on exactly that noise, so the rounding is reproduced here.
__device__ __forceinline__ double bce_item(double z, double y, double lw, bool exact) {
    if (exact) return (1.0 - y) * z + lw * softplus_neg(z);  // diagnostic mode (SSW_FB_EXACT_LOSS)
    const float log_sig = (float)(-softplus_neg(z));
    const float weighted = (float)((double)log_sig * lw);
    return (1.0 - y) * z - (double)weighted;
}

__device__ __forceinline__ double bce_lw(float pw, double y) { return 1.0 + ((double)pw - 1.0) * y; }  // the item's log-weight
// d loss / d logit of one BCE item, rounded to f32 like torch's (one body for every site that forms it)
__device__ __forceinline__ float bce_dz(double z, double y, double c, double lw) {
    return (float)(c * ((1.0 - y) - lw * sigmoidd(-z)));
}

// elementwise BCE-with-logits with pos_weight pw and per-item coefficient c_i:
//   l = c [ (1-y) z + (1 + (pw-1) y) softplus(-z) ],  dl/dz = c [ (1-y) - (1 + (pw-1) y) sigmoid(-z) ]
__global__ void k_fb_elem(const float *__restrict__ z, const float *__restrict__ y,
                          const float *__restrict__ coef, float pw, int64_t n,
                          double *__restrict__ item_loss, float *__restrict__ r, int exact) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double zi = z[i], yi = y[i], ci = coef[i];
    const double lw = bce_lw(pw, yi);
    item_loss[i] = ci * bce_item(zi, yi, lw, exact != 0);
    r[i] = bce_dz(zi, yi, ci, lw);
}

// k_fb_logits with the parameters in the argument segment and, when `elem` is set, k_fb_elem folded in
// (same expressions, same order: the results are bit-identical to the two-kernel form)
__global__ __launch_bounds__(256) void k_fb_logits_arg(const float *__restrict__ X, FbW wv, int64_t n, int dim,
                                                       int has_bias, float *__restrict__ z, int elem,
                                                       const float *__restrict__ y, const float *__restrict__ coef,
                                                       float pw, double *__restrict__ item_loss,
                                                       float *__restrict__ r) {
    // a wave takes FOUR rows: their loads are in flight together and lanes 0-3 each finish one row's label term (the
    // f64 exp / log1p chain was one lane per wave: a quarter of the waves, the same latency each).  Per row the
    // arithmetic is unchanged: the lane's fma chain over its columns, the xor butterfly 32 ... 1.
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    if (row0 >= n) return;
    const float4 *w4 = reinterpret_cast<const float4 *>(wv.v);
    const float4 *x4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) x4[u] = reinterpret_cast<const float4 *>(X + (row0 + u < n ? row0 + u : row0) * dim);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = lane; c < dim / 4; c += 64) {
        float4 xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) xv[u] = x4[u][c];
        const float4 wq = w4[c];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = fmaf(xv[u].x, wq.x, a[u]);
            a[u] = fmaf(xv[u].y, wq.y, a[u]);
            a[u] = fmaf(xv[u].z, wq.z, a[u]);
            a[u] = fmaf(xv[u].w, wq.w, a[u]);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] += __shfl_xor(a[u], off, 64);
    }
    const float mine = lane == 0 ? a[0] : lane == 1 ? a[1] : lane == 2 ? a[2] : a[3];
    const int64_t row = row0 + lane;
    if (lane < 4 && row < n) {
        const float zf = mine + (has_bias ? wv.v[dim] : 0.f);
        z[row] = zf;
        if (elem) {
            const double zi = zf, yi = y[row], ci = coef[row];
            const double lw = bce_lw(pw, yi);
            item_loss[row] = ci * bce_item(zi, yi, lw, elem == 2);
            r[row] = bce_dz(zi, yi, ci, lw);
        }
    }
}

// pairwise losses over all ordered pairs (i, j), t_ij = sign(y_i - y_j), s_ij = z_i - z_j:
//   hinge    : max(0, m - t_ij s_ij) - m [t_ij == 0]        (rank_loss.py:63-95)
//   logistic : t_ij^2 log(1 + exp(-t_ij s_ij))              (rank_loss.py:34-61)
// item_j = coef_j / max_inv_j * sum_i loss_ij, max_inv_j = #{i : t_ij != 0};
// r_k = d(sum_j item_j)/dz_k.  One workgroup; z, y, c = coef/max_inv in LDS.
template <int LOGISTIC>
__device__ __forceinline__ void fb_pairwise_body(const float *__restrict__ z, const float *__restrict__ y,
                                                 const float *__restrict__ coef, float margin, int n,
                                                 double *__restrict__ item_loss, float *__restrict__ r,
                                                 float *sh /* LDS, 3 n floats */) {
    float *sz = sh, *sy = sh + n, *sc = sh + 2 * n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        sz[i] = z[i];
        sy[i] = y[i];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        int cnt = 0;
        const float yj = sy[j];
        for (int i = 0; i < n; ++i) cnt += (sy[i] != yj);
        sc[j] = cnt > 0 ? coef[j] / (float)cnt : 0.f;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const float zk = sz[k], yk = sy[k], ck = sc[k];
        double loss_k = 0.0;  // column sum over i of loss_ik
        double rk = 0.0;
        for (int i = 0; i < n; ++i) {
            const float d = sy[i] - yk;
            if (d == 0.f) continue;
            const double t = d > 0.f ? 1.0 : -1.0;  // t_ik
            const double s = (double)(sz[i] - zk);  // s_ik (f32 difference, as torch forms it)
            // pair (i, k): contributes to item_k (column k) with weight ck; z_k enters with -1
            // pair (k, i): t_ki = -t, s_ki = -s, contributes to item_i with weight sc[i]; z_k enters with +1
            if (LOGISTIC) {
                const double u = -t * s;  // -t_ik s_ik  (== -t_ki s_ki)
                loss_k += log(1.0 + exp(u));  // the reference's unguarded form (rank_loss.py:48)
                const double sg = sigmoidd(u);  // d/du log(1+exp(u))
                // d loss_ik / d z_k = sg * (-t) * (-1) = t sg ;  d loss_ki / d z_k = sg * (t)(+1)... see below
                rk += ck * (t * sg) + sc[i] * (t * sg);
            } else {
                const double h = (double)margin - t * s;
                if (h >= 0.0) {  // torch's clamp(min=0) passes the gradient at the kink
                    loss_k += h;
                    // loss_ik = m - t (z_i - z_k): d/dz_k = +t ; loss_ki = m - (-t)(z_k - z_i) = m + t z_k - t z_i: d/dz_k = +t
                    rk += ck * t + sc[i] * t;
                }
            }
        }
        item_loss[k] = (double)ck * loss_k;
        r[k] = (float)rk;
    }
}

template <int LOGISTIC>
__global__ __launch_bounds__(1024) void k_fb_pairwise(const float *__restrict__ z,
                                                      const float *__restrict__ y,
                                                      const float *__restrict__ coef, float margin,
                                                      int n, double *__restrict__ item_loss,
                                                      float *__restrict__ r) {
    extern __shared__ float sh[];
    fb_pairwise_body<LOGISTIC>(z, y, coef, margin, n, item_loss, r, sh);
}

// RankingRegModule (logistic_regression.py:16-65): per-item "loss" |g_i| / total_pairs and the pseudo-gradient
// g_i / total_pairs that _CheapPairwiseRankingLoss.backward hands to autograd (rank_loss.py:164-187), g = the
// quick zero-margin gradient of rank.hip
__global__ void k_fb_rank_items(const float *__restrict__ g, float factor, int64_t n, double *__restrict__ item_loss,
                                float *__restrict__ r) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = g[i] * factor;  // f32, as torch forms grads * factor
    item_loss[i] = (double)fabsf(v);
    r[i] = v;
}

// partial gradient of one slab of rows: thread c owns column c
__global__ void k_fb_grad(const float *__restrict__ X, const float *__restrict__ r, int64_t n,
                          int dim, float *__restrict__ partial /* [nslabs, dim] */) {
    const int c = threadIdx.x + blockIdx.y * blockDim.x;
    if (c >= dim) return;
    const int64_t r0 = (int64_t)blockIdx.x * FB_SLAB;
    const int cnt = (int)(min(r0 + FB_SLAB, n) - r0);
    // all of the slab's loads are issued before the (ordered, dependent) fma chain consumes them: the chain
    // itself is 32 steps, waiting for one row at a time made it 32 memory latencies
    float xv[FB_SLAB], rv[FB_SLAB];
#pragma unroll
    for (int i = 0; i < FB_SLAB; ++i) {
        const int64_t row = r0 + (i < cnt ? i : 0);
        xv[i] = X[row * dim + c];
        rv[i] = r[row];
    }
    float g = 0.f;
#pragma unroll
    for (int i = 0; i < FB_SLAB; ++i)
        if (i < cnt) g = fmaf(rv[i], xv[i], g);
    partial[(int64_t)blockIdx.x * dim + c] = g;
}

// ---- two linear outputs: MultiRegModule (seesaw/loops/multi_reg_module.py:40-165), the scorer of the multi_reg_neg
// loop.  Row i has logits z_ic = <x_i, W_c / |W_c|>, c = 0 (target) / 1 (confusion class), f32 targets y_ic and a
// sample weight s_i.  loss_labels = sum_i s_i [ bce(z_i0, y_i0) + bce(z_i1, y_i1) ]          ("vertical")
//                                 + sum_{i : y_i0 + y_i1 > 0} s_i * -(sum_c y_ic log_softmax(z_i)_c)  ("horizontal")
// Everything is f32 in the reference (targets are cast with .astype('float32'), multi_reg_neg.py:73), so it is here.
// z, y, r are stored as two planes of `plane` floats so the single-output gradient kernel serves both outputs.
__global__ __launch_bounds__(256) void k_fb2_logits_elem(const float *__restrict__ X, const float *__restrict__ nw /* [2, dim] */,
                                                         int64_t n, int dim, int64_t plane, const float *__restrict__ y2,
                                                         const float *__restrict__ sw, float *__restrict__ z2,
                                                         float *__restrict__ r2, float *__restrict__ item_v,
                                                         float *__restrict__ item_h) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float4 *x4 = reinterpret_cast<const float4 *>(X + row * dim);
    const float4 *w0 = reinterpret_cast<const float4 *>(nw), *w1 = reinterpret_cast<const float4 *>(nw + dim);
    float a = 0.f, b = 0.f;
    for (int c = lane; c < dim / 4; c += 64) {
        const float4 xv = x4[c], p = w0[c], q = w1[c];
        a = fmaf(xv.x, p.x, a); a = fmaf(xv.y, p.y, a); a = fmaf(xv.z, p.z, a); a = fmaf(xv.w, p.w, a);
        b = fmaf(xv.x, q.x, b); b = fmaf(xv.y, q.y, b); b = fmaf(xv.z, q.z, b); b = fmaf(xv.w, q.w, b);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        a += __shfl_xor(a, off, 64);
        b += __shfl_xor(b, off, 64);
    }
    if (lane != 0) return;
    const float y0 = y2[row], y1 = y2[plane + row], s = sw[row];
    // binary_cross_entropy_with_logits: (1 - y) z - log_sigmoid(z), log_sigmoid(z) = min(z, 0) - log1p(exp(-|z|))
    const float ls0 = fminf(a, 0.f) - log1pf(expf(-fabsf(a))), ls1 = fminf(b, 0.f) - log1pf(expf(-fabsf(b)));
    const float vert = ((1.f - y0) * a - ls0) + ((1.f - y1) * b - ls1);
    const float sg0 = 1.f / (1.f + expf(-a)), sg1 = 1.f / (1.f + expf(-b));
    float g0 = sg0 - y0, g1 = sg1 - y1, hor = 0.f;
    const float near = y0 + y1;
    if (near > 0.f) {  // cross_entropy with probability targets on the rows that carry any label
        const float m = fmaxf(a, b);
        const float lse = m + logf(expf(a - m) + expf(b - m));
        hor = -(y0 * (a - lse) + y1 * (b - lse));
        g0 += near * expf(a - lse) - y0;
        g1 += near * expf(b - lse) - y1;
    }
    z2[row] = a;
    z2[plane + row] = b;
    r2[row] = s * g0;
    r2[plane + row] = s * g1;
    item_v[row] = s * vert;
    item_h[row] = s * hor;
}

// out[0] = sum item_v, out[1] = sum item_h, out[2 + c*dim + k] = sum over slabs of partial_c[slab][k]; then the flag
// sum of a column over the slab partials, in slab order, with the loads of 32 slabs in flight at a time: the plain
// loop issued them a few at a time (322 slabs at 10 000 rows = ~80 trips to L2, 80 of the 113 us a closure evaluation
// cost there); the additions and their order are unchanged
__device__ __forceinline__ float slab_column_sum(const float *__restrict__ col, int nslabs, int dim) {
    float g = 0.f;
    for (int s0 = 0; s0 < nslabs; s0 += 32) {
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = col[(int64_t)min(s0 + u, nslabs - 1) * dim];
#pragma unroll
        for (int u = 0; u < 32; ++u)
            if (s0 + u < nslabs) g += v[u];
    }
    return g;
}

// The same sums ahead of k_fb_final, over several workgroups, for sets of thousands of rows: one workgroup adding 322
// slab partials of 512 columns reads 660 KB through one CU (23 us of a 48-us closure evaluation at 10 000 rows).  Here a
// workgroup takes 64 columns; its four thread groups fetch a quarter of a 128-slab chunk each into LDS (32 loads in
// flight per thread), then one thread per column adds the chunk in slab order -- the additions k_fb_final's own loop
// makes, in its order; k_fb_final then finds one "slab" per column.
__global__ __launch_bounds__(256) void k_fb_slabsum(const float *__restrict__ partial, int nslabs, int dim,
                                                    float *__restrict__ gsum) {
    __shared__ float buf[128][64];
    const int t = threadIdx.x, col = t & 63, part = t >> 6;
    const int c = blockIdx.x * 64 + col;
    const int cc = c < dim ? c : dim - 1;
    float g = 0.f;
    for (int s0 = 0; s0 < nslabs; s0 += 128) {
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = partial[(int64_t)min(s0 + part * 32 + u, nslabs - 1) * dim + cc];
#pragma unroll
        for (int u = 0; u < 32; ++u) buf[part * 32 + u][col] = v[u];
        __syncthreads();
        if (part == 0) {
            const int cnt = min(128, nslabs - s0);
            for (int i = 0; i < cnt; ++i) g += buf[i][col];
        }
        __syncthreads();
    }
    if (part == 0 && c < dim) gsum[c] = g;
}

__global__ __launch_bounds__(1024) void k_fb2_reduce(const float *__restrict__ partial /* [2][nslabs_cap][dim] */,
                                                     int64_t partial_plane, int nslabs, const float *__restrict__ item_v,
                                                     const float *__restrict__ item_h, int64_t n, int dim,
                                                     float *__restrict__ out, unsigned *__restrict__ done_flag,
                                                     unsigned seqno) {
    __shared__ float rv[1024], rh[1024];
    const int t = threadIdx.x;
    if (t < 2 * dim) {
        const float *pp = partial + (t / dim) * partial_plane + (t % dim);
        float g = 0.f;
        g = slab_column_sum(pp, nslabs, dim);
        out[2 + t] = g;
    }
    float v = 0.f, h = 0.f;
    for (int64_t i = t; i < n; i += 1024) {
        v += item_v[i];
        h += item_h[i];
    }
    rv[t] = v;
    rh[t] = h;
    __syncthreads();
    for (int off = 512; off >= 1; off >>= 1) {
        if (t < off) {
            rv[t] += rv[t + off];
            rh[t] += rh[t + off];
        }
        __syncthreads();
    }
    if (t == 0) {
        out[0] = rv[0];
        out[1] = rh[0];
    }
    __threadfence_system();
    __syncthreads();
    if (t == 0) __hip_atomic_store(done_flag, seqno, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct FbObjDev {
    int kind;          // 0 logreg, 1 multireg
    int has_bias;
    int reg_kind;      // logreg: 0 none, 1 vector, 2 norm (target 0), 3 norm1 (target 1)
    float scale;       // multiplies the data loss and its gradient (1/n for logreg, 1 for multireg)
    float reg_weight;  // logreg: lambda / n
    float l_norm, l_data, l_query;  // multireg
    int exact;         // diagnostic (env SSW_FB_EXACT_LOSS): exact f64 loss values instead of torch's mixed f32/f64 rounding
};

// ---- reductions inside one full wave: row_shr 1, 2, 4, 8 (lane l takes lane l - s of its 16-lane row, rows'
// first s lanes take 0), the four row totals (lanes 15, 31, 47, 63) added in ascending order
template <int CTRL>
__device__ __forceinline__ double dpp_take_d(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_d(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_d(double v) {
    v += dpp_take_d<0x111>(v);
    v += dpp_take_d<0x112>(v);
    v += dpp_take_d<0x114>(v);
    v += dpp_take_d<0x118>(v);
    return ((readlane_d(v, 15) + readlane_d(v, 31)) + readlane_d(v, 47)) + readlane_d(v, 63);
}
template <int CTRL>
__device__ __forceinline__ float dpp_take_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum_f(float v) {  // the same network in f32
    v += dpp_take_f<0x111>(v);
    v += dpp_take_f<0x112>(v);
    v += dpp_take_f<0x114>(v);
    v += dpp_take_f<0x118>(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 15));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 47));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
    return ((r0 + r1) + r2) + r3;
}
__device__ __forceinline__ double wave_max_d(double v) {  // v >= 0
    v = fmax(v, dpp_take_d<0x111>(v));
    v = fmax(v, dpp_take_d<0x112>(v));
    v = fmax(v, dpp_take_d<0x114>(v));
    v = fmax(v, dpp_take_d<0x118>(v));
    return fmax(fmax(readlane_d(v, 15), readlane_d(v, 31)), fmax(readlane_d(v, 47), readlane_d(v, 63)));
}

struct FbFinalLds {
    double *red;   // [64]  wave partials of the four-way sum
    double *red2;  // [16]  wave partials of the single sums
    float *sw_;    // [1024] parameters
};

// the body of the last evaluation step, shared by k_fb_final (one launch per evaluation, host-driven fit) and
// k_fb_fit_wg (whole fit in one launch): `g` = this column's data gradient summed over the slabs in slab order,
// `wc` = this column's parameter.  `out` / `out_loss` may point to LDS.
__device__ __forceinline__ void fb_final_body(float g, const float wc, const double *__restrict__ item_loss,
                                              const float *__restrict__ r, int64_t n, int dim,
                                              const float *__restrict__ qhat, const float *__restrict__ xlx,
                                              const FbObjDev &obj, float *out, double *out_loss, const FbFinalLds &L) {
    double *red = L.red, *red2 = L.red2;
    float *sw_ = L.sw_;
    const int c = threadIdx.x;
    const bool act = c < dim;
    if (act) sw_[c] = wc;
    g *= obj.scale;
    // block reductions: |w|^2, w.qhat, data loss, sum r.  Inside each wave the DPP network of wave_sum_d (no LDS
    // traffic), then the 16 wave totals added in ascending order by every thread; four sums share one barrier.
    // (The first version walked an LDS tree with six barriers and 48 ds_bpermute per sum: 5 of the 6 us this
    // step took.)
    const int lane = c & 63, wave = c >> 6;
    auto block_sum = [&](double v) -> double {
        const double wsum = wave_sum_d(v);
        if (lane == 0) red2[wave] = wsum;
        __syncthreads();
        double o = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) o += red2[w];
        __syncthreads();
        return o;
    };
    // Waves without a column (c >= dim) or without an item have zeros to sum: their wave totals are +0.0 without the DPP
    // network, and only the waves that use a total add the 16 wave totals up (every thread did, for all four: 64 f64
    // additions a thread, four waves deep on every SIMD, ~2000 cycles of this step).  The additions that remain are
    // the same, in the same order.
    const bool has_cols = wave * 64 < dim, has_items = (int64_t)wave * 64 < n || n > 1024;
    double ls = 0.0, rs = 0.0;
    if (n <= 1024) {  // one item a thread at most
        if (c < n) {
            ls += item_loss[c];
            rs += (double)r[c];
        }
    } else {
        for (int64_t i0 = c; i0 < n; i0 += 8 * 1024) {  // eight of this thread's items in flight; same order of additions
            double li[8];
            float ri[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t i = i0 + (int64_t)u * 1024;
                li[u] = item_loss[i < n ? i : c];
                ri[u] = r[i < n ? i : c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (i0 + (int64_t)u * 1024 < n) {
                    ls += li[u];
                    rs += (double)ri[u];
                }
            }
        }
    }
    double ww, wq, data_loss, rsum;
    {
        const double v0 = act ? (double)wc * wc : 0.0, v1 = act && qhat ? (double)wc * qhat[c] : 0.0;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (has_cols) {
            s0 = wave_sum_d(v0);
            s1 = wave_sum_d(v1);
        }
        if (has_items) {
            s2 = wave_sum_d(ls);
            s3 = wave_sum_d(rs);
        }
        if (lane == 0) {
            red[4 * wave + 0] = s0;
            red[4 * wave + 1] = s1;
            red[4 * wave + 2] = s2;
            red[4 * wave + 3] = s3;
        }
        __syncthreads();
        double t0 = 1.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;  // a wave without columns uses none of them (ww = 1 keeps its arithmetic finite)
        if (has_cols) {
            t0 = 0.0;
#pragma unroll
            for (int w = 0; w < 16; ++w) {
                t0 += red[4 * w + 0];
                t1 += red[4 * w + 1];
            }
        }
        if (wave == 0) {
#pragma unroll
            for (int w = 0; w < 16; ++w) {
                t2 += red[4 * w + 2];
                t3 += red[4 * w + 3];
            }
        }
        ww = t0;
        wq = t1;
        data_loss = t2 * obj.scale;
        rsum = t3 * obj.scale;
    }
    const double norm = sqrt(ww);
    const double nclamp = norm > 1e-12 ? norm : 1e-12;  // F.normalize eps
    // From here on the scalar terms of the loss are formed by thread 0 only and waves that own no column leave
    // once the last barrier is behind them: with all 16 waves walking the whole body its ~800 instructions, four
    // waves deep on every SIMD, were most of this step's time.
    float greg = 0.f;
    if (obj.kind == 0) {
        double d2 = 0.0;
        if (obj.reg_kind == 1) {
            // (|w| - 1)^2 + |w^ - q^|^2 ; |w^ - q^|^2 = w^.w^ - 2 w^.q^ + q^.q^  (q^ unit)
            const double what_c = wc / nclamp;
            d2 = block_sum(act ? (what_c - qhat[c]) * (what_c - qhat[c]) : 0.0);
        }
        if (wave != 0 && wave * 64 >= dim) return;
        double reg_loss = 0.0;
        if (obj.reg_kind == 1) {
            const double what_c = wc / nclamp;
            const double whq = wq / nclamp;
            reg_loss = (norm - 1.0) * (norm - 1.0) + d2;
            if (act) {
                const double diff = what_c - qhat[c];
                // J'(diff), J = (I - w^ w^')/|w| ; w^.diff = w^.w^ - w^.q^
                const double wd = ww / (nclamp * nclamp) - whq;
                greg = (float)(2.0 * (norm - 1.0) * (wc / nclamp) + 2.0 / nclamp * (diff - what_c * wd));
            }
        } else if (obj.reg_kind == 2 || obj.reg_kind == 3) {
            const double target = obj.reg_kind == 3 ? 1.0 : 0.0;
            reg_loss = (norm - target) * (norm - target);
            if (act) greg = (float)(2.0 * (norm - target) * (wc / nclamp));
        }
        reg_loss *= obj.reg_weight;
        greg *= obj.reg_weight;
        if (act) out[1 + c] = g + greg;
        if (c == 0) {
            out[0] = (float)(data_loss + reg_loss);
            *out_loss = data_loss + reg_loss;
            out[1 + dim] = obj.has_bias ? (float)rsum : 0.f;
            float *parts = out + 1 + dim + 1;
            parts[0] = 0.f;
            parts[1] = 0.f;
            parts[2] = 0.f;
            parts[3] = (float)data_loss;
        }
    } else {
        // data: l w'Mw ; d/dw = l (M + M') w
        double mw = 0.0, mtw = 0.0, p_data = 0.0;
        if (obj.l_data != 0.f) {
            if (act && xlx) {
                for (int k = 0; k < dim; ++k) {
                    mw += (double)xlx[(int64_t)c * dim + k] * sw_[k];
                    mtw += (double)xlx[(int64_t)k * dim + c] * sw_[k];
                }
            }
            p_data = obj.l_data * block_sum(act ? (double)wc * mw : 0.0);
        }
        if (wave != 0 && wave * 64 >= dim) return;
        // query: l (1 - w^.q^)/2 ; d/dw = -l/2 (q^ - (w^.q^) w^)/|w|
        const double whq = wq / nclamp;
        if (act) {
            const double what_c = wc / nclamp;
            // norm: l (cosh(log s) - 1), s = w.w ; d/dw = l (1 - 1/s^2) w
            greg = (float)(obj.l_norm * (1.0 - 1.0 / (ww * ww)) * wc + obj.l_data * (mw + mtw) -
                           obj.l_query * 0.5 * ((qhat ? qhat[c] : 0.f) - whq * what_c) / nclamp);
            out[1 + c] = g + greg;
        }
        if (c == 0) {
            // The three regulariser VALUES are rounded the way the reference's f32 tensors round them
            // (multi_reg.py:125-129): the label loss is f64 there (float64 targets promote) but these terms are
            // f32, and cosh(log s) - 1 near s = 1 moves in steps of 2^-23 -- times l_norm = 100 that is a 1.2e-5
            // staircase in the total loss.  torch's L-BFGS stops / accepts line-search points on that staircase
            // (|loss - prev_loss| < 1e-9, Armijo), so an exact f64 value here walks a different path and ends up to
            // 7e-4 (rank scores) away from the reference's fit -- measured on tests/golden/multireg.npz c4.
            // Gradients stay analytic (autograd's f32 sinh(log s)/s * 2w equals l (1 - 1/s^2) w to rounding).
            const double p_norm = obj.exact ? obj.l_norm * (0.5 * (ww + 1.0 / ww) - 1.0)
                                            : (double)(obj.l_norm * (coshf(logf((float)ww)) - 1.f));
            if (!obj.exact) p_data = (double)(float)p_data;
            const double p_query =
                obj.exact ? obj.l_query * (1.0 - whq) * 0.5 : (double)(obj.l_query * ((1.f - (float)whq) / 2.f));
            const double reg_loss = p_norm + p_data + p_query;
            out[0] = (float)(data_loss + reg_loss);
            *out_loss = data_loss + reg_loss;
            out[1 + dim] = obj.has_bias ? (float)rsum : 0.f;
            float *parts = out + 1 + dim + 1;
            parts[0] = (float)p_norm;
            parts[1] = (float)p_data;
            parts[2] = (float)p_query;
            parts[3] = (float)data_loss;
        }
    }
}

// one workgroup of 1024 threads (column c = thread c, dim <= 1024): reduce partials in slab order, add the
// regulariser terms, emit out[0] = loss, out[1 .. 1+P) = gradient, out[1+P ..] = parts
__global__ __launch_bounds__(1024) void k_fb_final(const float *__restrict__ partial, int nslabs,
                                                   const double *__restrict__ item_loss,
                                                   const float *__restrict__ r, int64_t n, int dim,
                                                   const float *__restrict__ w_or_null, FbW wv,
                                                   const float *__restrict__ qhat,
                                                   const float *__restrict__ xlx, FbObjDev obj,
                                                   float *__restrict__ out, double *__restrict__ out_loss,
                                                   unsigned *__restrict__ done_flag, unsigned seqno) {
    __shared__ double red[64], red2[16];
    __shared__ float sw_[1024];
    const int c = threadIdx.x;
    const bool act = c < dim;
    const float wc = act ? (w_or_null ? w_or_null[c] : wv.v[c < FB_ARG_FLOATS ? c : 0]) : 0.f;
    // data gradient
    float g = 0.f;
    if (act)
        g = slab_column_sum(partial + c, nslabs, dim);
    FbFinalLds L{red, red2, sw_};
    fb_final_body(g, wc, item_loss, r, n, dim, qhat, xlx, obj, out, out_loss, L);
    // completion signal for the host's spin-wait (cheaper than waking up from hipStreamSynchronize, which costs
    // ~10 us per closure evaluation): every thread's result stores are released to the system, then one flag word
    if (done_flag) {
        __threadfence_system();
        __syncthreads();
        if (c == 0) {
            __hip_atomic_store(done_flag, seqno, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}


// minFunc's polyinterp restricted to two points with derivatives (torch.optim.lbfgs._cubic_interpolate)
__host__ __device__ inline double cubic_interpolate_dev(double x1, double f1, double g1, double x2, double f2, double g2,
                                                        bool has_bounds, double lo, double hi) {
    double xmin = lo, xmax = hi;
    if (!has_bounds) {
        xmin = x1 <= x2 ? x1 : x2;
        xmax = x1 <= x2 ? x2 : x1;
    }
    const double d1 = g1 + g2 - 3.0 * (f1 - f2) / (x1 - x2);
    const double d2sq = d1 * d1 - g1 * g2;
    if (d2sq >= 0) {
        const double d2 = sqrt(d2sq);
        double mp;
        if (x1 <= x2)
            mp = x2 - (x2 - x1) * ((g2 + d2 - d1) / (g2 - g1 + 2 * d2));
        else
            mp = x1 - (x1 - x2) * ((g1 + d2 - d1) / (g1 - g2 + 2 * d2));
        return fmin(fmax(mp, xmin), xmax);
    }
    return (xmin + xmax) / 2.0;
}

// ---- the whole fit in one launch --------------------------------------------------------------
// k_fb_fit_wg: torch.optim.LBFGS(strong_wolfe).step(closure) -- direction update, line search and every closure
// evaluation -- inside ONE workgroup, for labelled sets of up to FIT_WG_MAX_ROWS rows (a feedback session has
// tens to hundreds).  The host-driven form (ssw_fb_fit's loop further down) pays three launches and a
// host <-> device round trip per closure evaluation, ~27 us each and 80 % of a fit.  Here:
//   * all 16 waves evaluate the closure: logits (a wave per row, eight rows in flight), label loss, gradient
//     (column per thread, the two thread halves take alternate 32-row slabs), regulariser terms -- the bodies of
//     the per-evaluation kernels, with z / r / item losses / targets / the query in LDS so that no phase waits
//     for a global store to become visible; the only global traffic is two passes over the rows, which sit in L2;
//   * wave 0 alone drives: the L-BFGS vectors live in LDS (element k in lane k % 64), reductions are DPP networks
//     inside the wave -- no barrier and no memory round trip in the two-loop recursion; the history is the one
//     structure too large for LDS (100 x 2 x 513 floats) and is prefetched from global one entry ahead.
// The driver is the host loop of ssw_fb_fit (torch.optim.LBFGS.step + _strong_wolfe) unrolled into a state machine
// around ONE evaluation site: ST_INIT (first closure call), ST_BRACKET (an evaluation of the bracketing phase),
// ST_ZOOM (an evaluation of the zoom phase); ST_NEW_ITER, ST_ZOOM_HEAD, ST_LS_END need no evaluation.
// The arithmetic is the host driver's, operation for operation (wave_sum_host restates the reduction network),
// so both drivers return bit-identical fits, which tests/test_feedback_gpu.py asserts.
constexpr int FIT_WG_MAX_ROWS = 1024;
constexpr int FIT_HISTORY = 100;
constexpr int FIT_ROUND_SLABS = 8;  // slabs whose partial gradients are staged in LDS at a time

struct FitWgArgs {
    const float *X, *y, *coef;   // [n, dim] centred rows, targets, per-item coefficients
    const float *qhat, *xlx;     // or null
    const float *w0_or_null;     // initial parameters when they do not fit the argument segment
    float *hist_dirs, *hist_stps;  // [FIT_HISTORY, 1024] (rows padded: the driving wave loads without bounds checks)
    float *out_w;                // mapped host memory: [dim + 1]
    double *out_loss;            // mapped host
    int *out_counts;             // mapped host: iterations, evaluations, status (0 ok, 1 loss diverged), diagnostics
    unsigned *done_flag;
    unsigned seqno;
    int n, dim, P;               // P = trainable parameters (dim or dim + 1)
    int label_mode;              // 0 elementwise BCE, 1 pairwise hinge, 2 pairwise logistic, 3 identically zero
    int onepass;                 // fit_eval_onepass instead of the logits / labels / gradient phases (same bits)
    float pw, margin;
    int max_iter;
    float lr;
    FbObjDev obj;
};

struct FitDriver {  // the L-BFGS / line-search scalars of k_fb_fit_wg's driving wave
    double loss, prev_loss, t, H_diag, gtd;
    double f0, d_norm, t_prev, f_prev, gtd_prev, gtd_new;
    double br0, br1, brf0, brf1, brg0, brg1;
    int st, status, n_iter, current_evals, evals, m, head;
    int nbr, low, ls_iter, ls_evals, done, insuf, first_ls_eval;
};

// ---- LDS map of k_fb_fit_wg (the launch sizes the allocation with fit_wg_lds_bytes)
__host__ __device__ inline size_t fit_wg_vec_stride(int dim) { return dim + 1 <= 9 * 64 ? 9 * 64 : 16 * 64; }  // FIT_EPL * 64
constexpr int FIT_OP_ROWS = 16, FIT_OP_BUFS = 3, FIT_OP_DIM = 512;  // fit_eval_onepass: units of rows staged in LDS
__host__ __device__ inline size_t fit_wg_stage_floats(int dim, bool onepass) {
    const size_t a = (size_t)FIT_ROUND_SLABS * dim, b = (size_t)3 * FIT_WG_MAX_ROWS;
    const size_t c = onepass ? (size_t)FIT_OP_BUFS * FIT_OP_ROWS * FIT_OP_DIM : 0;
    return a > b ? (a > c ? a : c) : (b > c ? b : c);
}
constexpr size_t FIT_LDS_DOUBLES = 64 + 16 + 2 * FIT_HISTORY + 2 + 2 + 32 + 8 + FIT_WG_MAX_ROWS + 32 /* FitWgArgs */;
__host__ __device__ inline size_t fit_wg_lds_bytes(int dim, bool onepass) {
    const size_t floats = 1024 /*sw*/ + 1040 /*gout*/ + 4 * FIT_WG_MAX_ROWS /*z r y coef*/ + 1024 /*qhat*/ +
                          7 * fit_wg_vec_stride(dim) + fit_wg_stage_floats(dim, onepass);
    return FIT_LDS_DOUBLES * sizeof(double) + floats * sizeof(float);
}
// pointers read back from the LDS copy of the arguments are generic to the compiler (flat loads, 64-bit address
// arithmetic per access); the phases cast them back to what they are
typedef const float __attribute__((address_space(1))) *fit_gcptr;
typedef float __attribute__((address_space(1))) *fit_gptr;
typedef float fit_v4f __attribute__((ext_vector_type(4)));
typedef const fit_v4f __attribute__((address_space(1))) *fit_g4ptr;

struct FitLds {
    double *red;        // [64]  wave partials of the four-way block sum (fb_final_body)
    double *red2;       // [16]
    double *ro, *al;    // [FIT_HISTORY] 1 / y.s of the history entries; the two-loop recursion's alphas
    double *loss_slot;  // [2]  loss of the last evaluation (f64)
    double *ctl;        // [2]  ctl[0] != 0: the driver has finished
    FitDriver *drv;     // [32 doubles]
    unsigned long long *tk;  // [8] diagnostic timers (100 MHz ticks)
    double *item;       // [FIT_WG_MAX_ROWS] per-item label losses
    FitWgArgs *args;    // [32 doubles] the kernel arguments, for the phases compiled as functions
    float *sw_;         // [1024] parameters of the evaluation
    float *gout;        // [1 + dim + 1 + 4] loss, gradient, parts (the layout k_fb_final emits)
    float *zl, *rl, *yl, *cl;  // [FIT_WG_MAX_ROWS] logits, d loss / d logit, targets, per-item coefficients
    float *qh;          // [dim] unit query
    float *vx, *vd, *vg, *vpg, *vgp, *vb0, *vb1;  // L-BFGS vectors, [FIT_EPL * 64] each, zero beyond element dim
    float *stage;       // pairwise staging [3 n] / partial gradients [FIT_ROUND_SLABS, dim] / fit_eval_onepass's row units
};
__device__ __forceinline__ FitLds fit_lds_map(double *base, int dim) {
    static_assert(sizeof(FitDriver) <= 32 * sizeof(double), "FitDriver outgrew its LDS slot");
    static_assert(sizeof(FitWgArgs) <= 32 * sizeof(double), "FitWgArgs outgrew its LDS slot");
    FitLds S;
    S.red = base;
    S.red2 = S.red + 64;
    S.ro = S.red2 + 16;
    S.al = S.ro + FIT_HISTORY;
    S.loss_slot = S.al + FIT_HISTORY;
    S.ctl = S.loss_slot + 2;
    S.drv = reinterpret_cast<FitDriver *>(S.ctl + 2);
    S.tk = reinterpret_cast<unsigned long long *>(S.ctl + 2 + 32);
    S.item = S.ctl + 2 + 32 + 8;
    S.args = reinterpret_cast<FitWgArgs *>(S.item + FIT_WG_MAX_ROWS);
    S.sw_ = reinterpret_cast<float *>(base + FIT_LDS_DOUBLES);
    S.gout = S.sw_ + 1024;
    S.zl = S.gout + 1040;
    S.rl = S.zl + FIT_WG_MAX_ROWS;
    S.yl = S.rl + FIT_WG_MAX_ROWS;
    S.cl = S.yl + FIT_WG_MAX_ROWS;
    S.qh = S.cl + FIT_WG_MAX_ROWS;
    const int VS = (int)fit_wg_vec_stride(dim);
    S.vx = S.qh + 1024;
    S.vd = S.vx + VS;
    S.vg = S.vd + VS;
    S.vpg = S.vg + VS;
    S.vgp = S.vpg + VS;
    S.vb0 = S.vgp + VS;
    S.vb1 = S.vb0 + VS;
    S.stage = S.vb1 + VS;
    return S;
}

// The phases of one closure evaluation and the driver step are separate (not inlined) functions: each gets the
// whole 128-register budget of a 1024-thread workgroup.  Inlined into one body the kernel spilled 130+ registers
// into scratch inside its hot loops, and every phase ran 3-5x slower than its memory traffic allows.

// z = X w (+ b): a wave per row, eight rows in flight
__device__ __noinline__ void fit_eval_logits(double *base, int dim_) {
    const FitLds S = fit_lds_map(base, dim_);
    const FitWgArgs &a = *S.args;
    const int c = threadIdx.x, lane = c & 63, wave = c >> 6;
    const int n = a.n, dim = a.dim;
    float *sw_ = S.sw_, *zl = S.zl;
    for (int row0 = wave; row0 < n; row0 += 128) {
        float acc[8];
        fit_g4ptr x4[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = row0 + 16 * u;
            x4[u] = (fit_g4ptr)(a.X + (int64_t)(row < n ? row : row0) * dim);
            acc[u] = 0.f;
        }
        const float4 *w4 = reinterpret_cast<const float4 *>(sw_);
        for (int k = lane; k < dim / 4; k += 64) {  // the eight rows' loads are issued together
            // (both column steps of a 512-wide row in one trip -- sixteen loads in flight per lane -- was measured:
            //  logits 7.3 -> 10.4 us per evaluation at 390 rows; eight it stays)
            fit_v4f xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) xv[u] = x4[u][k];
            const float4 wq = w4[k];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float s_ = acc[u];
                s_ = fmaf(xv[u].x, wq.x, s_);
                s_ = fmaf(xv[u].y, wq.y, s_);
                s_ = fmaf(xv[u].z, wq.z, s_);
                s_ = fmaf(xv[u].w, wq.w, s_);
                acc[u] = s_;
            }
        }
        // the xor butterfly of k_fb_logits (a += shfl_xor(a, off), off = 32 ... 1) for eight rows at once: at
        // off = 32, 16, 8 each lane keeps half of its rows and hands the other half to its partner, so a step
        // moves 4, 2, 1 values instead of 8, 8, 8 -- the sums formed are the butterfly's (a + b = b + a), bit for bit
        float b4[4], b2[2], b1;
        {
            const bool hi = (lane & 32) != 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float keep = hi ? acc[j + 4] : acc[j], send = hi ? acc[j] : acc[j + 4];
                b4[j] = keep + __shfl_xor(send, 32, 64);
            }
        }
        {
            const bool hi = (lane & 16) != 0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float keep = hi ? b4[j + 2] : b4[j], send = hi ? b4[j] : b4[j + 2];
                b2[j] = keep + __shfl_xor(send, 16, 64);
            }
        }
        {
            const bool hi = (lane & 8) != 0;
            const float keep = hi ? b2[1] : b2[0], send = hi ? b2[0] : b2[1];
            b1 = keep + __shfl_xor(send, 8, 64);
        }
        b1 += __shfl_xor(b1, 4, 64);
        b1 += __shfl_xor(b1, 2, 64);
        b1 += __shfl_xor(b1, 1, 64);
        {
            const int u = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
            const int row = row0 + 16 * u;
            if ((lane & 7) == 0 && row < n) zl[row] = b1 + (a.obj.has_bias ? sw_[dim] : 0.f);
        }
    }
}

// per-item label losses and d loss / d logit
__device__ __noinline__ void fit_eval_labels(double *base, int dim_) {
    const FitLds S = fit_lds_map(base, dim_);
    const FitWgArgs &a = *S.args;
    const int c = threadIdx.x;
    const int n = a.n;
    float *zl = S.zl, *yl = S.yl, *cl = S.cl, *rl = S.rl, *stage = S.stage;
    double *item = S.item;
    if (a.label_mode == 0) {
        for (int i = c; i < n; i += 1024) {
            const double zi = zl[i], yi = yl[i], ci = cl[i];
            const double lw = bce_lw(a.pw, yi);
            item[i] = ci * bce_item(zi, yi, lw, a.obj.exact != 0);
            rl[i] = bce_dz(zi, yi, ci, lw);
        }
    } else if (a.label_mode == 1) {
        fb_pairwise_body<0>(zl, yl, cl, a.margin, n, item, rl, stage);
    } else if (a.label_mode == 2) {
        fb_pairwise_body<1>(zl, yl, cl, a.margin, n, item, rl, stage);
    } else {
        for (int i = c; i < n; i += 1024) {
            item[i] = 0.0;
            rl[i] = 0.f;
        }
    }
}

// g = X' r: slabs of FB_SLAB rows, an ordered fma chain inside each (k_fb_grad), the slab sums added in slab
// order (k_fb_final); returns column threadIdx.x's sum
__device__ __noinline__ float fit_eval_grad(double *base, int dim_) {
    const FitLds S = fit_lds_map(base, dim_);
    const FitWgArgs &a = *S.args;
    const int c = threadIdx.x;
    const int n = a.n, dim = a.dim;
    float *rl = S.rl, *stage = S.stage;
    float gcol = 0.f;
    {
        // a thread takes four adjacent columns (16-byte loads); thread group grp = c / (dim / 4) takes slab s0 + grp
        // of every round of FIT_ROUND_SLABS slabs; a slab's chain runs in two halves of 16 rows in flight
        const int dq = dim / 4;
        const int groups = min(FIT_ROUND_SLABS, 1024 / dq);
        const int grp = c / dq, cq = c - grp * dq;
        const int nslabs = (n + FB_SLAB - 1) / FB_SLAB;
        for (int s0 = 0; s0 < nslabs; s0 += groups) {
            const int sl = s0 + grp;
            if (grp < groups && sl < nslabs) {
                const int r0 = sl * FB_SLAB;
                const int cnt = min(r0 + FB_SLAB, n) - r0;
                float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
                // quarters of 8 rows on two register sets: the next quarter's loads are in flight while this one's
                // chain runs (it was two halves of 16, the second requested only after the first had been consumed)
                static_assert(FB_SLAB == 32, "four quarters of eight rows");
                fit_v4f xq[2][8];
#pragma unroll
                for (int i = 0; i < 8; ++i) xq[0][i] = *(fit_g4ptr)(a.X + (int64_t)(r0 + (i < cnt ? i : 0)) * dim + 4 * cq);
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int h = qd * 8;
                    if (qd + 1 < 4) {
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            xq[(qd + 1) & 1][i] =
                                *(fit_g4ptr)(a.X + (int64_t)(r0 + (h + 8 + i < cnt ? h + 8 + i : 0)) * dim + 4 * cq);
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (h + i < cnt) {
                            const float ri = rl[r0 + h + i];
                            const fit_v4f xv = xq[qd & 1][i];
                            p.x = fmaf(ri, xv.x, p.x);
                            p.y = fmaf(ri, xv.y, p.y);
                            p.z = fmaf(ri, xv.z, p.z);
                            p.w = fmaf(ri, xv.w, p.w);
                        }
                }
                *reinterpret_cast<float4 *>(stage + (size_t)grp * dim + 4 * cq) = p;
            }
            __syncthreads();
            if (c < dim)
                for (int j = 0; j < groups && s0 + j < nslabs; ++j) gcol += stage[(size_t)j * dim + c];
            __syncthreads();
        }
    }
    return gcol;
}

// The three phases above in ONE pass over the rows (elementwise BCE, dim = 512): the rows are what an evaluation
// costs -- a CU pulls them from L2 at ~110 GB/s, and the logits and the gradient each walked them once.  Here a
// unit of 16 rows is loaded once, by the waves that take its logits (lane k the float4s k and k + 64 of a row --
// fit_eval_logits' assignment, so the fma chain and the butterfly's additions are the same), and copied to LDS on
// the way for the gradient chain, which needs it two barriers later.  The waves have roles:
//     stage s :  waves 4-11  loads of unit s + 2 issued; logits of unit s (two rows a wave), its rows to LDS
//                wave 12     d loss / d logit of unit s - 1 (one f64 exp + divide chain for its 16 rows; on the
//                            logit waves it was 16 wave-wide chains a stage and three times the rows' time)
//                waves 0-3   gradient chain of unit s - 2 from LDS, two columns a thread
// Three row units of 32 KB rotate through LDS.  Every sum keeps its order: a row's logit is the lane chain + xor
// butterfly (v_permlane32/16_swap and row_ror DPP adds form the butterfly's sums without its six LDS round trips);
// a column's slab sum is the fma chain over the slab's rows in row order (a thread keeps it across the slab's two
// units); the slab sums are added in slab order.  The per-item loss values (exp + log1p) are not on the gradient's
// path and are taken after the last unit for all rows at once.  Returns column threadIdx.x's gradient sum.
// Measured (390 rows, tools/perf_fit.py): 15.3 us an evaluation for what took 7.3 + 1.4 + 9.5 = 18.2; a stage is
// ~1260 cycles whatever is taken out of it (no row loads, no f64 chain, no butterfly: 33-35 k cycles a pass each
// time): each role is a latency chain of 800-1000 cycles -- exp + divide in f64, LDS round trips, the wait for the
// rows -- and the barrier adds ~250, i.e. 78 cycles a row against the two passes' 112 and the rows' own 45.
__device__ __forceinline__ float fit_xor_butterfly(float a) {  // a += shfl_xor(a, off), off = 32 ... 1: the same additions
    {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(a), false, false);
        a = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(a), false, false);
        a = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    a += dpp_take_f<0x128>(a);  // row_ror 8, 4, 2, 1: the values repeat with that period by then, so a rotation is the xor
    a += dpp_take_f<0x124>(a);
    a += dpp_take_f<0x122>(a);
    a += dpp_take_f<0x121>(a);
    return a;
}

#ifndef SSW_FIT_STAMPS
#define SSW_FIT_STAMPS 0  // diagnostic builds: 1 busy cycles per evaluation of a logit wave, the d loss / d logit wave and a gradient wave; 2 an idle wave's entry -> last stage -> return
#endif
__device__ __noinline__ float fit_eval_onepass(double *base, int dim_) {
    long long t_entry = 0;
    if (SSW_FIT_STAMPS == 2) t_entry = clock64();
    const FitLds S = fit_lds_map(base, dim_);
    const FitWgArgs &a = *S.args;
    const int c = threadIdx.x, lane = c & 63, wave = c >> 6;
    const int n = a.n;
    constexpr int D = FIT_OP_DIM, D4 = FIT_OP_DIM / 4;
    static_assert(FIT_OP_ROWS == 16 && FIT_OP_BUFS == 3 && FB_SLAB == 2 * FIT_OP_ROWS, "roles and rotation below");
    float *rows = S.stage;  // [FIT_OP_BUFS][FIT_OP_ROWS][D]
    float *zl = S.zl, *rl = S.rl, *yl = S.yl, *cl = S.cl;
    // (Spreading the roles over the SIMDs by their VALU cost -- the f64 wave alone with one logit wave -- measured
    // slower: 429 against 397 us of data phases per fit at 390 rows.)
    const bool gwave = wave < 4, zwave = wave >= 4 && wave < 12, rwave = wave == 12;
    const int zslot = wave - 4;
    const int gi = c;  // a gradient thread takes columns 2 gi, 2 gi + 1
    const int zrow = 2 * zslot;  // a logit wave's first row inside the unit
    const int H = (n + FIT_OP_ROWS - 1) / FIT_OP_ROWS;
    const fit_g4ptr X4 = (fit_g4ptr)a.X;
    long long busy = 0;
    float g0 = 0.f, g1 = 0.f;
    // A loop per role, H + 2 barriers each.  (One loop with the roles as branches made the compiler wait for every
    // load in flight before it issued the next unit's: the row loads must stay two stages ahead.)
    if (zwave) {
        // The row loads and their waits are written out (inline asm): left to the compiler, the wait-count pass drained
        // every load in flight at the loop header -- once per three stages no load was in flight, and the rows' stream
        // is what bounds a stage.  Four loads a stage, in order; a stage consumes the set issued two stages earlier,
        // i.e. it waits until at most eight newer loads are outstanding.
#ifndef SSW_FIT_PF
#define SSW_FIT_PF 2  // units of rows in flight ahead of the one a stage consumes
#endif
        constexpr int PF = SSW_FIT_PF, NS = PF + 1;
        fit_v4f xl[NS][2], xh[NS][2];
        auto load = [&](int s, fit_v4f(&lo)[2], fit_v4f(&hi)[2]) {  // s beyond the last unit: the last unit again, unused
            s = s < H ? s : H - 1;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                int row = FIT_OP_ROWS * s + zrow + u;
                row = row < n ? row : n - 1;
                const fit_g4ptr src = X4 + (int64_t)row * D4 + lane;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(lo[u]) : "v"(src) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(hi[u]) : "v"(src) : "memory");
            }
        };
#pragma unroll
        for (int q = 0; q < PF; ++q) load(q, xl[q], xh[q]);
        // The rows do not depend on the point of the evaluation: the first units are on their way while wave 0 still
        // drives (the caller enters here without waiting for it).  Behind this barrier sw_ and ctl are published.
        __syncthreads();
        if (S.ctl[0] != 0.0) {  // the driver has finished: nothing to evaluate
#pragma unroll
            for (int q = 0; q < PF; ++q)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(xl[q][0]), "+v"(xh[q][0]), "+v"(xl[q][1]), "+v"(xh[q][1])::"memory");
            return 0.f;
        }
        const float4 wa = reinterpret_cast<const float4 *>(S.sw_)[lane], wb = reinterpret_cast<const float4 *>(S.sw_)[lane + 64];
        const float bias = a.obj.has_bias ? S.sw_[D] : 0.f;
        for (int s0 = 0; s0 < H; s0 += NS) {
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const int s = s0 + j;
                if (s >= H) break;
                long long tb = 0;
                if (SSW_FIT_STAMPS == 1) tb = clock64();
                load(s + PF, xl[(j + PF) % NS], xh[(j + PF) % NS]);
                static_assert(PF == 2 || PF == 3, "the wait below names the count");
                if (PF == 2)
                    asm volatile("s_waitcnt vmcnt(8)" : "+v"(xl[j][0]), "+v"(xh[j][0]), "+v"(xl[j][1]), "+v"(xh[j][1])::"memory");
                else
                    asm volatile("s_waitcnt vmcnt(12)" : "+v"(xl[j][0]), "+v"(xh[j][0]), "+v"(xl[j][1]), "+v"(xh[j][1])::"memory");
                float acc[2];
                float *buf = rows + (size_t)(s % 3) * FIT_OP_ROWS * D;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const fit_v4f x0 = xl[j][u], x1 = xh[j][u];
                    float t = 0.f;
                    t = fmaf(x0.x, wa.x, t);
                    t = fmaf(x0.y, wa.y, t);
                    t = fmaf(x0.z, wa.z, t);
                    t = fmaf(x0.w, wa.w, t);
                    t = fmaf(x1.x, wb.x, t);
                    t = fmaf(x1.y, wb.y, t);
                    t = fmaf(x1.z, wb.z, t);
                    t = fmaf(x1.w, wb.w, t);
                    acc[u] = t;
                    float *dst = buf + (size_t)(zrow + u) * D;
                    *reinterpret_cast<fit_v4f *>(dst + 4 * lane) = x0;
                    *reinterpret_cast<fit_v4f *>(dst + 256 + 4 * lane) = x1;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float z = fit_xor_butterfly(acc[u]) + bias;
                    const int row = FIT_OP_ROWS * s + zrow + u;
                    if (lane == 0 && row < n) zl[row] = z;
                }
                if (SSW_FIT_STAMPS == 1) busy += clock64() - tb;
                __syncthreads();
            }
        }
        // the loads still in flight land in registers the compiler would otherwise hand out again
#pragma unroll
        for (int q = 0; q < NS; ++q)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(xl[q][0]), "+v"(xh[q][0]), "+v"(xl[q][1]), "+v"(xh[q][1])::"memory");
        __syncthreads();
        __syncthreads();
    } else if (rwave) {
        __syncthreads();
        if (S.ctl[0] != 0.0) return 0.f;
        for (int s = 0; s < H + 2; ++s) {
            long long tb = 0;
            if (SSW_FIT_STAMPS == 1) tb = clock64();
            const int row = FIT_OP_ROWS * (s - 1) + lane;
            if (s >= 1 && s - 1 < H && lane < FIT_OP_ROWS && row < n) {
                const double yi = yl[row], ci = cl[row];
                rl[row] = bce_dz((double)zl[row], yi, ci, bce_lw(a.pw, yi));
            }
            if (SSW_FIT_STAMPS == 1) busy += clock64() - tb;
            __syncthreads();
        }
    } else if (gwave) {
        __syncthreads();
        if (S.ctl[0] != 0.0) return 0.f;
        float p0 = 0.f, p1 = 0.f;
        for (int s = 0; s < H + 2; ++s) {
            long long tb = 0;
            if (SSW_FIT_STAMPS == 1) tb = clock64();
            if (s >= 2) {
                const int u = s - 2, r0 = FIT_OP_ROWS * u;
                const float *src = rows + (size_t)(u % 3) * FIT_OP_ROWS * D + 2 * gi;
                if (r0 + FIT_OP_ROWS <= n) {
                    float rr[FIT_OP_ROWS];  // the unit's sixteen rows in flight together: one LDS latency, not two
                    float2 xv[FIT_OP_ROWS];
#pragma unroll
                    for (int q = 0; q < FIT_OP_ROWS / 4; ++q) {
                        const float4 t = *reinterpret_cast<const float4 *>(rl + r0 + 4 * q);
                        rr[4 * q + 0] = t.x;
                        rr[4 * q + 1] = t.y;
                        rr[4 * q + 2] = t.z;
                        rr[4 * q + 3] = t.w;
                    }
#pragma unroll
                    for (int i = 0; i < FIT_OP_ROWS; ++i) xv[i] = *reinterpret_cast<const float2 *>(src + (size_t)i * D);
#pragma unroll
                    for (int i = 0; i < FIT_OP_ROWS; ++i) {
                        p0 = fmaf(rr[i], xv[i].x, p0);
                        p1 = fmaf(rr[i], xv[i].y, p1);
                    }
                } else {
                    for (int i = 0; r0 + i < n; ++i) {
                        const float2 xv = *reinterpret_cast<const float2 *>(src + (size_t)i * D);
                        const float ri = rl[r0 + i];
                        p0 = fmaf(ri, xv.x, p0);
                        p1 = fmaf(ri, xv.y, p1);
                    }
                }
                if ((u & 1) || u == H - 1) {  // the slab is complete
                    g0 += p0;
                    g1 += p1;
                    p0 = 0.f;
                    p1 = 0.f;
                }
            }
            if (SSW_FIT_STAMPS == 1) busy += clock64() - tb;
            __syncthreads();
        }
    } else {
        __syncthreads();
        if (S.ctl[0] != 0.0) return 0.f;
        for (int s = 0; s < H + 2; ++s) __syncthreads();
    }
    long long t_loop = 0;
    if (SSW_FIT_STAMPS == 2) t_loop = clock64();
    if (SSW_FIT_STAMPS == 1 && lane == 0) {
        if (wave == 4) S.tk[4] += (unsigned long long)busy;
        if (wave == 12) S.tk[6] += (unsigned long long)busy;
        if (wave == 0) S.tk[7] += (unsigned long long)busy;
    }
    // columns back to their threads (thread c: column c)
    if (gwave) *reinterpret_cast<float2 *>(rows + 2 * gi) = make_float2(g0, g1);
    // the loss values of all rows (exp + log1p: ~1700 cycles a chain -- as a wave's job inside the stages it was the
    // longest of them and cost 130 us a fit; here all rows' chains run side by side once, ~1 us)
    for (int i = c; i < n; i += 1024) {
        const double zi = zl[i], yi = yl[i], ci = cl[i];
        S.item[i] = ci * bce_item(zi, yi, bce_lw(a.pw, yi), a.obj.exact != 0);
    }
    __syncthreads();
    if (SSW_FIT_STAMPS == 2 && c == 1023) {  // an idle wave's view: entry -> loop end -> return
        S.tk[4] += (unsigned long long)(t_loop - t_entry);
        S.tk[6] += (unsigned long long)(clock64() - t_loop);
    }
    return c < D ? rows[c] : 0.f;
}

__device__ __noinline__ void fit_eval_final(double *base, int dim_, float gcol) {
    const FitLds S = fit_lds_map(base, dim_);
    const FitWgArgs &a = *S.args;
    const int c = threadIdx.x;
    const FbFinalLds L{S.red, S.red2, S.sw_};
    fb_final_body(gcol, c < a.dim ? S.sw_[c] : 0.f, S.item, S.rl, a.n, a.dim, a.qhat ? S.qh : (const float *)nullptr, a.xlx,
                  a.obj, S.gout, S.loss_slot, L);
}

// One step of the driver (wave 0): consumes the evaluation in gout / loss_slot, advances the L-BFGS / line-search
// state machine up to the next point that needs an evaluation, and publishes that point in sw_ (or ctl[0] = 1).
// FIT_EPL: vector elements per lane (9 covers dim + 1 <= 576, 16 covers 1024).
template <int FIT_EPL>
__device__ __noinline__ void fit_driver_step(double *base, int dim_) {
    const FitLds S = fit_lds_map(base, dim_);
    const FitWgArgs &a = *S.args;
    const int lane = threadIdx.x & 63;
    const int dim = a.dim, Pd = dim + 1;
    double *ro = S.ro, *al = S.al, *loss_slot = S.loss_slot, *ctl = S.ctl;
    float *sw_ = S.sw_, *gout = S.gout;
    float *vx = S.vx, *vd = S.vd, *vg = S.vg, *vpg = S.vpg, *vgp = S.vgp, *vb0 = S.vb0, *vb1 = S.vb1;
    enum { ST_INIT, ST_BRACKET, ST_ZOOM, ST_NEW_ITER, ST_ZOOM_HEAD, ST_LS_END };
    FitDriver D = *S.drv;  // a register copy for the duration of the step (LDS round trips between dependent scalar
                           // operations were most of a step's time); written back on the way out
    int &st = D.st, &status = D.status, &n_iter = D.n_iter, &current_evals = D.current_evals, &evals = D.evals;
    int &m = D.m, &head = D.head;  // history ring: logical entry i sits in slot (head + i) % FIT_HISTORY
    int &nbr = D.nbr, &low = D.low, &ls_iter = D.ls_iter, &ls_evals = D.ls_evals;
    int &done = D.done, &insuf = D.insuf, &first_ls_eval = D.first_ls_eval;
    double &loss = D.loss, &prev_loss = D.prev_loss, &t = D.t, &H_diag = D.H_diag, &gtd = D.gtd;
    double &f0 = D.f0, &d_norm = D.d_norm, &t_prev = D.t_prev, &f_prev = D.f_prev, &gtd_prev = D.gtd_prev;
    double &gtd_new = D.gtd_new;
    double &br0 = D.br0, &br1 = D.br1, &brf0 = D.brf0, &brf1 = D.brf1, &brg0 = D.brg0, &brg1 = D.brg1;  // bracket ends: t, f, g.d
    const double tol_grad = 1e-7, tol_change = 1e-9, c1 = 1e-4, c2 = 0.9;
    const int max_ls = 25;
    const int max_eval = a.max_iter * 5 / 4;
    // vector helpers: element k of a vector lives in lane k % 64.  The L-BFGS vectors are FIT_EPL * 64 long and zero
    // beyond element dim, so the loops are unrolled without bounds checks (the LDS reads overlap); `g_new` is the one
    // operand with live data behind element dim (the loss parts): it is masked when copied and only ever multiplied
    // with a padded vector otherwise.
    // dot products are f32 throughout, like torch's (a BLAS sdot there; here: per-lane products added in ascending
    // order, then wave_sum_f's network) -- one wave owns the whole recursion, so every instruction saved is latency
    auto vdot = [&](const float *u, const float *v) -> double {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < FIT_EPL; ++j) s += u[lane + 64 * j] * v[lane + 64 * j];
        return (double)wave_sum_f(s);
    };
    auto vabsmax = [&](const float *u) -> double {
        double mx = 0.0;
#pragma unroll
        for (int j = 0; j < FIT_EPL; ++j) mx = fmax(mx, (double)fabsf(u[lane + 64 * j]));
        return wave_max_d(mx);
    };
    auto vcopy = [&](float *dst, const float *src) {
        float v[FIT_EPL];
#pragma unroll
        for (int j = 0; j < FIT_EPL; ++j) v[j] = src[lane + 64 * j];
#pragma unroll
        for (int j = 0; j < FIT_EPL; ++j) dst[lane + 64 * j] = lane + 64 * j < Pd ? v[j] : 0.f;
    };
    const float *g_new = gout + 1;  // the gradient of the last evaluation (element dim is forced to 0 without intercept)

    evals++;
    if (a.P == dim && lane == 0) gout[1 + dim] = 0.f;  // no intercept: that slot is not a parameter
    const double f_new = loss_slot[0];
    bool finished = false;
    if (!isfinite(f_new)) {
        status = 1;
        finished = true;
    } else if (st == ST_INIT) {
        loss = f_new;
        vcopy(vg, g_new);
        current_evals = 1;
        prev_loss = loss;
        if (vabsmax(vg) <= tol_grad) finished = true;
        st = ST_NEW_ITER;
    } else if (st == ST_BRACKET) {
        if (first_ls_eval) {
            ls_evals = 1;
            first_ls_eval = 0;
        } else {
            ls_evals++;
            ls_iter++;
        }
        gtd_new = vdot(g_new, vd);
        bool next_eval = false;
        if (ls_iter < max_ls) {
            if (f_new > (f0 + c1 * t * gtd) || (ls_iter > 1 && f_new >= f_prev)) {
                br0 = t_prev; br1 = t; brf0 = f_prev; brf1 = f_new;
                vcopy(vb0, vgp); vcopy(vb1, g_new); brg0 = gtd_prev; brg1 = gtd_new; nbr = 2;
            } else if (fabs(gtd_new) <= -c2 * gtd) {
                br0 = t; brf0 = f_new; vcopy(vb0, g_new); nbr = 1;
                done = 1;
            } else if (gtd_new >= 0) {
                br0 = t_prev; br1 = t; brf0 = f_prev; brf1 = f_new;
                vcopy(vb0, vgp); vcopy(vb1, g_new); brg0 = gtd_prev; brg1 = gtd_new; nbr = 2;
            } else {
                const double min_step = t + 0.01 * (t - t_prev), max_step = t * 10;
                const double tmp = t;
                t = cubic_interpolate_dev(t_prev, f_prev, gtd_prev, t, f_new, gtd_new, true, min_step, max_step);
                t_prev = tmp; f_prev = f_new; vcopy(vgp, g_new); gtd_prev = gtd_new;
                next_eval = true;
            }
        } else {  // max_ls interpolations without a bracket: [0, t]
            br0 = 0; br1 = t; brf0 = f0; brf1 = f_new; vcopy(vb0, vg); vcopy(vb1, g_new); nbr = 2;
            brg0 = gtd; brg1 = gtd_new;
        }
        if (!next_eval) {
            insuf = 0;
            low = 0;
            if (nbr == 2) low = brf0 <= brf1 ? 0 : 1;
            st = ST_ZOOM_HEAD;
        }
    } else {  // ST_ZOOM
        ls_evals++;
        gtd_new = vdot(g_new, vd);
        ls_iter++;
        const double brf_low = low == 0 ? brf0 : brf1;
        if (f_new > (f0 + c1 * t * gtd) || f_new >= brf_low) {
            if (low == 0) { br1 = t; brf1 = f_new; vcopy(vb1, g_new); brg1 = gtd_new; }
            else          { br0 = t; brf0 = f_new; vcopy(vb0, g_new); brg0 = gtd_new; }
            low = brf0 <= brf1 ? 0 : 1;
        } else {
            if (fabs(gtd_new) <= -c2 * gtd) {
                done = 1;
            } else {
                const double br_high = low == 0 ? br1 : br0, br_low = low == 0 ? br0 : br1;
                if (gtd_new * (br_high - br_low) >= 0) {
                    if (low == 0) { br1 = br0; brf1 = brf0; vcopy(vb1, vb0); brg1 = brg0; }
                    else          { br0 = br1; brf0 = brf1; vcopy(vb0, vb1); brg0 = brg1; }
                }
            }
            if (low == 0) { br0 = t; brf0 = f_new; vcopy(vb0, g_new); brg0 = gtd_new; }
            else          { br1 = t; brf1 = f_new; vcopy(vb1, g_new); brg1 = gtd_new; }
        }
        st = ST_ZOOM_HEAD;
    }

    while (!finished && st != ST_BRACKET && st != ST_ZOOM) {
        if (st == ST_ZOOM_HEAD) {
            if (done || ls_iter >= max_ls || fabs(br1 - br0) * d_norm < tol_change) {
                st = ST_LS_END;
                continue;
            }
            t = cubic_interpolate_dev(br0, brf0, brg0, br1, brf1, brg1, false, 0, 0);
            const double bmax = fmax(br0, br1), bmin = fmin(br0, br1);
            const double eps = 0.1 * (bmax - bmin);
            if (fmin(bmax - t, t - bmin) < eps) {
                if (insuf || t >= bmax || t <= bmin) {
                    t = (fabs(t - bmax) < fabs(t - bmin)) ? bmax - eps : bmin + eps;
                    insuf = 0;
                } else {
                    insuf = 1;
                }
            } else {
                insuf = 0;
            }
            st = ST_ZOOM;
        } else if (st == ST_LS_END) {
            // accepted point: the bracket's low end
            loss = low == 0 ? brf0 : brf1;
            vcopy(vg, low == 0 ? vb0 : vb1);
            t = low == 0 ? br0 : br1;
            double dm = 0.0;  // max_i |d_i t| in f64, as the host forms it
            for (int k = lane; k < Pd; k += 64) {
                vx[k] += (float)t * vd[k];
                dm = fmax(dm, fabs((double)vd[k] * t));
            }
            dm = wave_max_d(dm);
            const bool opt_cond = vabsmax(vg) <= tol_grad;
            current_evals += ls_evals;
            if (n_iter == a.max_iter || current_evals >= max_eval || opt_cond || dm <= tol_change ||
                fabs(loss - prev_loss) < tol_change) {
                finished = true;
                break;
            }
            st = ST_NEW_ITER;
        } else {  // ST_NEW_ITER: the optimality test of the initial point / previous iteration has passed
            if (n_iter >= a.max_iter) {
                finished = true;
                break;
            }
            n_iter++;
            if (n_iter == 1) {
                for (int k = lane; k < Pd; k += 64) vd[k] = -vg[k];
                H_diag = 1.0;
            } else {
                const unsigned long long t2l = wall_clock64();
                // y = g - prev_g, s = d t
                float yv[FIT_EPL], sv[FIT_EPL];
                float pys = 0.f, pyy = 0.f;
#pragma unroll
                for (int j = 0; j < FIT_EPL; ++j) {
                    const int k = lane + 64 * j;
                    yv[j] = vg[k] - vpg[k];  // the vectors are zero beyond element dim
                    sv[j] = vd[k] * (float)t;
                    pys += yv[j] * sv[j];
                    pyy += yv[j] * yv[j];
                }
                const double ys = (double)wave_sum_f(pys);
                if (ys > 1e-10) {
                    if (m == FIT_HISTORY) {
                        head = (head + 1) % FIT_HISTORY;
                        m--;
                    }
                    const int slot = (head + m) % FIT_HISTORY;
#pragma unroll
                    for (int j = 0; j < FIT_EPL; ++j) {
                        ((fit_gptr)a.hist_dirs + (size_t)slot * 1024 + lane)[64 * j] = yv[j];
                        ((fit_gptr)a.hist_stps + (size_t)slot * 1024 + lane)[64 * j] = sv[j];
                    }
                    if (lane == 0) ro[slot] = 1.0 / ys;
                    m++;
                    H_diag = ys / (double)wave_sum_f(pyy);
                }
                // two-loop recursion, q in registers.  A step is ~60 instructions of one wave (~300 cycles) and needs
                // one history entry (2 x 513 floats in L2, ~800 cycles away): four register sets, entries loaded three
                // steps ahead, the trip unrolled by four so the sets rotate without copies.
                const fit_gcptr hst = (fit_gcptr)a.hist_stps + lane, hdr = (fit_gcptr)a.hist_dirs + lane;
                float q[FIT_EPL], sb[4][FIT_EPL], db[4][FIT_EPL];
#pragma unroll
                for (int j = 0; j < FIT_EPL; ++j) {
                    q[j] = -vg[lane + 64 * j];
#pragma unroll
                    for (int u = 0; u < 4; ++u) sb[u][j] = db[u][j] = 0.f;
                }
                auto load_entry = [&](int i, float *s_out, float *d_out) {
                    const int slot = (head + i) % FIT_HISTORY;
                    const fit_gcptr ps = hst + (size_t)slot * 1024, pd = hdr + (size_t)slot * 1024;
#pragma unroll
                    for (int j = 0; j < FIT_EPL; ++j) {
                        s_out[j] = ps[64 * j];
                        d_out[j] = pd[64 * j];
                    }
                };
                auto first_loop_step = [&](int i, const float *cs, const float *cd) {
                    const double roi = ro[(head + i) % FIT_HISTORY];  // read before the reduction, not after it
                    float p = 0.f;
#pragma unroll
                    for (int j = 0; j < FIT_EPL; ++j) p += cs[j] * q[j];
                    const double ali = (double)wave_sum_f(p) * roi;
                    if (lane == 0) al[i] = ali;
#pragma unroll
                    for (int j = 0; j < FIT_EPL; ++j) q[j] -= (float)ali * cd[j];
                };
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    if (u < m) load_entry(m - 1 - u, sb[u], db[u]);
                for (int k0 = 0; k0 < m; k0 += 4) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int k = k0 + u;  // step k works on entry m - 1 - k
                        if (k < m) {
                            if (k + 3 < m) load_entry(m - 1 - (k + 3), sb[(u + 3) & 3], db[(u + 3) & 3]);
                            first_loop_step(m - 1 - k, sb[u], db[u]);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < FIT_EPL; ++j) q[j] = q[j] * (float)H_diag;  // q is now d
                auto second_loop_step = [&](int i, const float *cs, const float *cd) {
                    const double roi = ro[(head + i) % FIT_HISTORY], ali = al[i];
                    float p = 0.f;
#pragma unroll
                    for (int j = 0; j < FIT_EPL; ++j) p += cd[j] * q[j];
                    const double be = (double)wave_sum_f(p) * roi;
                    const float cf = (float)(ali - be);
#pragma unroll
                    for (int j = 0; j < FIT_EPL; ++j) q[j] += cf * cs[j];
                };
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    if (u < m) load_entry(u, sb[u], db[u]);
                for (int k0 = 0; k0 < m; k0 += 4) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int k = k0 + u;
                        if (k < m) {
                            if (k + 3 < m) load_entry(k + 3, sb[(u + 3) & 3], db[(u + 3) & 3]);
                            second_loop_step(k, sb[u], db[u]);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < FIT_EPL; ++j) {
                    vd[lane + 64 * j] = q[j];
                }
                if (lane == 0) S.tk[5] += wall_clock64() - t2l;
            }
            vcopy(vpg, vg);
            prev_loss = loss;
            if (n_iter == 1) {
                double pa = 0.0;
                for (int k = lane; k < Pd; k += 64) pa += (double)fabsf(vg[k]);
                const double gs = wave_sum_d(pa);
                t = fmin(1.0, 1.0 / gs) * (double)a.lr;
            } else {
                t = (double)a.lr;
            }
            gtd = vdot(vg, vd);
            if (gtd > -tol_change) {
                finished = true;
                break;
            }
            // line-search set-up
            f0 = loss;
            d_norm = vabsmax(vd);
            t_prev = 0; f_prev = f0; gtd_prev = gtd;
            vcopy(vgp, vg);
            done = 0;
            ls_iter = 0;
            nbr = 0;
            first_ls_eval = 1;
            st = ST_BRACKET;
        }
    }
    if (finished) {
        if (lane == 0) ctl[0] = 1.0;
    } else {
#pragma unroll
        for (int j = 0; j < FIT_EPL; ++j) sw_[lane + 64 * j] = vx[lane + 64 * j] + (float)t * vd[lane + 64 * j];
    }
    if (lane == 0) *S.drv = D;
}

template <int FIT_EPL>
__global__ __launch_bounds__(1024) void k_fb_fit_wg(FitWgArgs a_in, FbW w0v) {
    extern __shared__ double fit_lds[];
    const int c = threadIdx.x, lane = c & 63, wave = c >> 6;
    const int dim = a_in.dim, n = a_in.n, Pd = dim + 1;
    const FitLds S = fit_lds_map(fit_lds, dim);
    const int VS = (int)fit_wg_vec_stride(dim);
    if (c == 0) {
        *S.args = a_in;
        FitDriver z0;
        memset(&z0, 0, sizeof(z0));
        z0.H_diag = 1.0;  // st = ST_INIT = 0
        *S.drv = z0;
        S.ctl[0] = 0.0;
        for (int i = 0; i < 8; ++i) S.tk[i] = 0;
    }
    for (int i = c; i < n; i += 1024) {
        S.yl[i] = a_in.y[i];
        S.cl[i] = a_in.coef[i];
    }
    if (c < dim) S.qh[c] = a_in.qhat ? a_in.qhat[c] : 0.f;
    for (int k = c; k < 1040; k += 1024) S.gout[k] = 0.f;  // the driver reads it FIT_EPL * 64 wide
    for (int k = c; k < VS; k += 1024) {
        float x0 = 0.f;
        if (k < a_in.P) x0 = a_in.w0_or_null ? a_in.w0_or_null[k] : w0v.v[k < FB_ARG_FLOATS ? k : 0];
        S.vx[k] = x0;
        S.vd[k] = 0.f;
        S.vg[k] = 0.f;
        S.vpg[k] = 0.f;
        S.vgp[k] = 0.f;
        S.vb0[k] = 0.f;
        S.vb1[k] = 0.f;
        if (k < 1024) S.sw_[k] = x0 + (float)0.0 * 0.f;  // the first closure call: f(x + 0 d)
    }
    const unsigned long long t_kernel0 = wall_clock64();
    const long long c_kernel0 = clock64();
    for (;;) {
        unsigned long long tq0 = wall_clock64();
        float gcol;
        if (a_in.onepass) {
            // the barrier that publishes sw_ / ctl is inside: the logit waves request their first rows ahead of it
            gcol = fit_eval_onepass(fit_lds, dim);
            if (S.ctl[0] != 0.0) break;
            __syncthreads();
            if (c == 0) { const unsigned long long q = wall_clock64(); S.tk[0] += q - tq0; tq0 = q; }
        } else {
            __syncthreads();  // sw_ and ctl are published
            if (S.ctl[0] != 0.0) break;
            tq0 = wall_clock64();
            fit_eval_logits(fit_lds, dim);
            __syncthreads();
            if (c == 0) { const unsigned long long q = wall_clock64(); S.tk[0] += q - tq0; tq0 = q; }
            fit_eval_labels(fit_lds, dim);
            __syncthreads();
            if (c == 0) { const unsigned long long q = wall_clock64(); S.tk[1] += q - tq0; tq0 = q; }
            gcol = fit_eval_grad(fit_lds, dim);
            if (c == 0) { const unsigned long long q = wall_clock64(); S.tk[2] += q - tq0; tq0 = q; }
        }
        fit_eval_final(fit_lds, dim, gcol);
        __syncthreads();
        if (c == 0) { const unsigned long long q = wall_clock64(); S.tk[3] += q - tq0; tq0 = q; }
        if (wave == 0) fit_driver_step<FIT_EPL>(fit_lds, dim);
    }
    // every wave is here; wave 0 publishes the result
    if (wave == 0) {
        const FitDriver &D = *S.drv;
        for (int k = lane; k < Pd; k += 64) a_in.out_w[k] = S.vx[k];
        if (lane == 0) {
            a_in.out_counts[0] = D.n_iter;
            a_in.out_counts[1] = D.evals;
            a_in.out_counts[2] = D.status;
            for (int i = 0; i < 4; ++i) a_in.out_counts[3 + i] = (int)S.tk[i];
            a_in.out_counts[7] = (int)(wall_clock64() - t_kernel0);
            a_in.out_counts[8] = (int)((clock64() - c_kernel0) >> 4);
            a_in.out_counts[9] = 0;
            a_in.out_counts[10] = (int)S.tk[5];
            a_in.out_counts[11] = (int)(S.tk[4] >> 4);
            a_in.out_counts[12] = (int)(S.tk[6] >> 4);
            a_in.out_counts[13] = (int)(S.tk[7] >> 4);
            *a_in.out_loss = D.loss;
        }
        __threadfence_system();
        if (lane == 0) __hip_atomic_store(a_in.done_flag, a_in.seqno, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace
}  // namespace ssw

using namespace ssw;

struct ssw_fb {
    int device = 0;
    int dim = 0;
    int64_t n = 0, cap = 0;
    float *X = nullptr;        // [cap, dim] centred rows
    float *mu = nullptr;       // [dim]
    float *y = nullptr, *coef = nullptr, *z = nullptr, *r = nullptr;  // [cap]
    double *item = nullptr;    // [cap] per-item label losses (f64)
    double *loss_dev = nullptr, *loss_host = nullptr;  // total loss in f64
    int64_t *rows = nullptr;   // gather staging
    float *partial = nullptr;  // [nslabs(cap), dim]
    float *gsum = nullptr;     // [1024] ordered column sums of the slab partials (k_fb_slabsum, sets of >= 2048 rows)
    float *rankg = nullptr;    // [cap] net position changes of the rank objective (SSW_FB_RANKREG)
    // two-output objective (ssw_fb_*2): planes of cap floats / [2][nslabs(cap)][dim] partial gradients
    float *y2 = nullptr, *sw2 = nullptr, *z2 = nullptr, *r2 = nullptr, *itemv = nullptr, *itemh = nullptr;
    float *partial2 = nullptr, *nw2 = nullptr;
    float *out2_host = nullptr, *out2_host_dev = nullptr;  // mapped pinned [2 + 2 dim]
    int64_t cap2 = 0;
    bool has_targets2 = false;
    double *colsum = nullptr;  // [cap / FB_CENTER_ROWS, dim] block sums of the centring step
    float *w = nullptr;        // [dim + 1]
    float *qhat = nullptr;     // [dim]
    float *xlx = nullptr;      // [dim, dim]
    bool has_q = false, has_xlx = false;
    float *out = nullptr;      // device [1 + dim + 1 + 4]
    float *out_host = nullptr;  // pinned mirror
    float *out_host_dev = nullptr;     // the same memory through the device's address space
    double *loss_host_dev = nullptr;
    unsigned *flag_host = nullptr, *flag_host_dev = nullptr;  // mapped pinned completion word of the closure evaluation
    unsigned seqno = 0;
    float *w_host = nullptr;    // pinned staging
    hipStream_t stream = nullptr;
    // targets (host copies, for the objective set-up)
    std::vector<float> y_host, sw_host;
    // pinned staging of the per-round uploads (row ids, targets, query): queued on the stream without a wait -- the fit
    // that follows is stream-ordered behind them (a refine spent ~35 us in three synchronisations here)
    PinnedStage rows_stage, y_stage, q_stage, coef_stage, th_stage, drawn_stage;
    int64_t *th = nullptr, *drawn = nullptr;  // ssw_fb_set_pseudo_sample: labelled[j] - j, the draw
    int64_t th_cap = 0, drawn_cap = 0;
    bool targets_on_device = false;           // the targets were formed on the device: y_host holds only the labelled part
    std::vector<float> qhat_host;  // the normalised query (host copy, for the two-output objective's host part)
    // diagnostics of the last fit
    int last_iters = 0, last_evals = 0;
    float rank_factor = 0.f;   // 1 / total_pairs of the installed targets (SSW_FB_RANKREG)
    float *hist = nullptr;     // [2, FIT_HISTORY, 1024] L-BFGS history of the single-launch fit
    bool last_fit_on_device = false;
};

static ssw_status fb_reserve(ssw_fb *fb, int64_t n) {
    if (n <= fb->cap) return SSW_OK;
    SSW_HIP_TRY(hipStreamSynchronize(fb->stream));
    (void)hipFree(fb->X);
    (void)hipFree(fb->y);
    (void)hipFree(fb->coef);
    (void)hipFree(fb->z);
    (void)hipFree(fb->item);
    (void)hipFree(fb->r);
    (void)hipFree(fb->rows);
    (void)hipFree(fb->partial);
    (void)hipFree(fb->rankg);
    (void)hipFree(fb->colsum);
    fb->colsum = nullptr;
    fb->X = fb->y = fb->coef = fb->z = fb->r = fb->partial = fb->rankg = nullptr;
    fb->item = nullptr;
    fb->rows = nullptr;
    fb->cap = 0;
    int64_t cap = 256;
    while (cap < n) cap <<= 1;
    const int64_t nslabs = (cap + FB_SLAB - 1) / FB_SLAB;
    SSW_HIP_TRY(hipMalloc((void **)&fb->X, (size_t)cap * fb->dim * sizeof(float)));
    SSW_HIP_TRY(hipMalloc((void **)&fb->y, (size_t)cap * sizeof(float)));
    SSW_HIP_TRY(hipMalloc((void **)&fb->coef, (size_t)cap * sizeof(float)));
    SSW_HIP_TRY(hipMalloc((void **)&fb->z, (size_t)cap * sizeof(float)));
    SSW_HIP_TRY(hipMalloc((void **)&fb->item, (size_t)cap * sizeof(double)));
    SSW_HIP_TRY(hipMalloc((void **)&fb->r, (size_t)cap * sizeof(float)));
    SSW_HIP_TRY(hipMalloc((void **)&fb->rows, (size_t)cap * sizeof(int64_t)));
    SSW_HIP_TRY(hipMalloc((void **)&fb->partial, (size_t)nslabs * fb->dim * sizeof(float)));
    SSW_HIP_TRY(hipMalloc((void **)&fb->rankg, (size_t)cap * sizeof(float)));
    SSW_HIP_TRY(hipMalloc((void **)&fb->colsum, (size_t)((cap + FB_CENTER_ROWS - 1) / FB_CENTER_ROWS) * fb->dim * sizeof(double)));
    fb->cap = cap;
    return SSW_OK;
}

static ssw_status fb_center(ssw_fb *fb, int center) {
    if (center && fb->n > 0) {
        const int nblk = (int)((fb->n + FB_CENTER_ROWS - 1) / FB_CENTER_ROWS);
        const dim3 grid((fb->dim + 63) / 64, nblk);
        hipLaunchKernelGGL(k_fb_colsum_blocks, grid, dim3(64), 0, fb->stream, fb->X, fb->n, fb->dim, fb->colsum);
        hipLaunchKernelGGL(k_fb_colmean, dim3((fb->dim + 63) / 64), dim3(64), 0, fb->stream, fb->colsum, nblk, fb->n,
                           fb->dim, fb->mu);
        hipLaunchKernelGGL(k_fb_subtract_mean, grid, dim3(64), 0, fb->stream, fb->X, fb->n, fb->dim, fb->mu);
        SSW_HIP_TRY(hipGetLastError());
    } else {
        SSW_HIP_TRY(hipMemsetAsync(fb->mu, 0, (size_t)fb->dim * sizeof(float), fb->stream));
    }
    return SSW_OK;
}

// effective per-item coefficients + pos_weight + data-loss scale for an objective
static ssw_status fb_prepare(ssw_fb *fb, const ssw_fb_objective *o, FbObjDev *dev, float *pw_out,
                             bool *pairwise_active) {
    const int64_t n = fb->n;
    std::vector<float> coef((size_t)n);
    *pairwise_active = false;
    memset(dev, 0, sizeof(*dev));
    dev->kind = o->kind;
    dev->exact = getenv("SSW_FB_EXACT_LOSS") != nullptr;
    if (fb->targets_on_device && o->kind != SSW_FB_LOGREG) {
        set_error("feedback: targets set by ssw_fb_set_pseudo_sample serve the logistic objective only");
        return SSW_ERR_INVALID;
    }
    if (o->kind == SSW_FB_LOGREG) {
        // mean over items of weight_i * bce(.; pos_weight)   (logistic_regression.py:98-105)
        for (int64_t i = 0; i < n; ++i) coef[(size_t)i] = fb->sw_host.empty() ? 1.f : fb->sw_host[(size_t)i];
        *pw_out = o->pos_weight;
        dev->has_bias = o->fit_intercept ? 1 : 0;
        dev->reg_kind = o->reg_kind;
        dev->scale = n > 0 ? 1.f / (float)n : 0.f;
        dev->reg_weight = o->reg_weight;
        if (o->reg_kind == 1 && !fb->has_q) {
            set_error("feedback: vector regulariser needs ssw_fb_set_query first");
            return SSW_ERR_INVALID;
        }
    } else if (o->kind == SSW_FB_MULTIREG) {
        dev->scale = 1.f;
        dev->l_norm = o->reg_norm_lambda;
        dev->l_data = o->reg_data_lambda;
        dev->l_query = o->reg_query_lambda;
        if (!fb->has_q) {
            set_error("feedback: multireg needs ssw_fb_set_query first");
            return SSW_ERR_INVALID;
        }
        if (o->reg_data_lambda != 0.f && !fb->has_xlx) {
            set_error("feedback: reg_data_lambda != 0 needs ssw_fb_set_xlx first");
            return SSW_ERR_INVALID;
        }
        // sample_weight bookkeeping of RegModule._step (multi_reg.py:85-105), in f32 like torch
        float orig_sum = 0.f, pos_total = 0.f;
        for (int64_t i = 0; i < n; ++i) {
            const float s = fb->sw_host.empty() ? 1.f : fb->sw_host[(size_t)i];
            coef[(size_t)i] = s;
            orig_sum += s;
            if (fb->y_host[(size_t)i] == 1.f) pos_total += s;
        }
        const float neg_total = orig_sum - pos_total;
        *pw_out = 1.f;
        if (o->loss_type == SSW_FB_LOSS_CE) {
            const float positive_weight = o->pos_weight < 0.f ? (neg_total + 1.f) / (pos_total + 1.f) : o->pos_weight;
            float new_sum = 0.f;
            for (int64_t i = 0; i < n; ++i) {
                if (fb->y_host[(size_t)i] == 1.f) coef[(size_t)i] *= positive_weight;
                new_sum += coef[(size_t)i];
            }
            const float f = n > 0 ? orig_sum / new_sum : 0.f;
            for (int64_t i = 0; i < n; ++i) coef[(size_t)i] *= f;
        } else {
            *pairwise_active = (pos_total > 0.f && neg_total > 0.f);
            if (n > FB_MAX_PAIRWISE) {
                set_error("feedback: pairwise losses take at most %d items, got %lld", FB_MAX_PAIRWISE, (long long)n);
                return SSW_ERR_UNSUPPORTED;
            }
        }
    } else if (o->kind == SSW_FB_RANKREG) {
        // sum_i |g_i| / total_pairs + (lambda / n) R(w): RankRegressionPT.fit (logistic_regression.py:216-255)
        dev->kind = SSW_FB_LOGREG;  // the regulariser terms are LogisticRegressionPT's
        dev->reg_kind = o->reg_kind;
        dev->scale = 1.f;
        dev->reg_weight = o->reg_weight;
        if (o->reg_kind == 1 && !fb->has_q) {
            set_error("feedback: vector regulariser needs ssw_fb_set_query first");
            return SSW_ERR_INVALID;
        }
        if (n > SSW_RANK_MAX_ITEMS) {
            set_error("feedback: the rank objective takes at most %d items, got %lld", SSW_RANK_MAX_ITEMS, (long long)n);
            return SSW_ERR_UNSUPPORTED;
        }
        // total_pairs = n^2 - sum over distinct targets of (class size)^2 (rank_loss.py:152)
        std::vector<float> ys(fb->y_host.begin(), fb->y_host.begin() + n);
        std::sort(ys.begin(), ys.end());
        double total = (double)n * (double)n;
        for (int64_t i = 0; i < n;) {
            int64_t j = i;
            while (j < n && ys[(size_t)j] == ys[(size_t)i]) ++j;
            total -= (double)(j - i) * (double)(j - i);
            i = j;
        }
        fb->rank_factor = total > 0 ? (float)(1.0 / total) : 0.f;
        *pw_out = 1.f;
        return SSW_OK;  // no per-item coefficients
    } else {
        set_error("feedback: unknown objective kind %d", o->kind);
        return SSW_ERR_INVALID;
    }
    // through pinned staging (coef is a local), without a wait: the evaluations are stream-ordered behind the copy
    if (n > 0) SSW_TRY(fb->coef_stage.push(fb->coef, coef.data(), (size_t)n * sizeof(float), fb->stream));
    return SSW_OK;
}

// one closure evaluation at the parameters in fb->w_host; results land in fb->out_host
static ssw_status fb_eval(ssw_fb *fb, const ssw_fb_objective *o, const FbObjDev &dev, float pw,
                          bool pairwise_active, int P) {
    hipStream_t s = fb->stream;
    const int64_t n = fb->n;
    const int dim = fb->dim;
    const bool by_arg = dim + 1 <= FB_ARG_FLOATS;
    FbW wv;
    if (by_arg) {
        memcpy(wv.v, fb->w_host, (size_t)P * sizeof(float));
        for (int i = P; i < FB_ARG_FLOATS; ++i) wv.v[i] = 0.f;
    } else {
        SSW_HIP_TRY(hipMemcpyAsync(fb->w, fb->w_host, (size_t)P * sizeof(float), hipMemcpyHostToDevice, s));
    }
    int nslabs = 0;
    if (n > 0) {
        const bool rankreg = o->kind == SSW_FB_RANKREG;
        const bool pairwise = rankreg || (o->kind == SSW_FB_MULTIREG && o->loss_type != SSW_FB_LOSS_CE);
        nslabs = (int)((n + FB_SLAB - 1) / FB_SLAB);
        if (by_arg)
            hipLaunchKernelGGL(k_fb_logits_arg, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, s, fb->X, wv, n, dim,
                               dev.has_bias, fb->z, pairwise ? 0 : (dev.exact ? 2 : 1), fb->y, fb->coef, pw, fb->item, fb->r);
        else
            hipLaunchKernelGGL(k_fb_logits, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, fb->X, fb->w, n, dim,
                               dev.has_bias, fb->z);
        if (!pairwise) {
            if (!by_arg)
                hipLaunchKernelGGL(k_fb_elem, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, fb->z, fb->y, fb->coef,
                                   pw, n, fb->item, fb->r, dev.exact);
        } else if (rankreg) {
            // z is ready: net position changes by counting (rank.hip), then |g| / total_pairs and g / total_pairs
            SSW_TRY(launch_rank_quick(fb->y, fb->z, (int)n, fb->rankg, nullptr, nullptr, s));
            hipLaunchKernelGGL(k_fb_rank_items, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, fb->rankg,
                               fb->rank_factor, n, fb->item, fb->r);
        } else if (pairwise_active) {
            const size_t lds = (size_t)3 * n * sizeof(float);
            if (o->loss_type == SSW_FB_LOSS_PAIRWISE_LOGISTIC)
                hipLaunchKernelGGL(k_fb_pairwise<1>, dim3(1), dim3(1024), lds, s, fb->z, fb->y, fb->coef, o->margin,
                                   (int)n, fb->item, fb->r);
            else
                hipLaunchKernelGGL(k_fb_pairwise<0>, dim3(1), dim3(1024), lds, s, fb->z, fb->y, fb->coef, o->margin,
                                   (int)n, fb->item, fb->r);
        } else {  // only one class labelled so far: the label loss is identically 0 (multi_reg.py:107)
            SSW_HIP_TRY(hipMemsetAsync(fb->item, 0, (size_t)n * sizeof(double), s));
            SSW_HIP_TRY(hipMemsetAsync(fb->r, 0, (size_t)n * sizeof(float), s));
        }
        // (folding this into the logits kernel, one 32-row slab per workgroup, was measured: fewer, longer
        // workgroups cost more than the launch they save -- 48 vs 36 us per evaluation)
        const int tx = dim < 256 ? dim : 256;
        hipLaunchKernelGGL(k_fb_grad, dim3((unsigned)nslabs, (unsigned)((dim + tx - 1) / tx)), dim3(tx), 0, s, fb->X,
                           fb->r, n, dim, fb->partial);
    }
    const float *final_partial = fb->partial;
    int final_slabs = nslabs;
    if (nslabs >= 64 && !getenv("SSW_FB_NO_SLABSUM")) {  // thousands of rows: the slab sums on several CUs first (same additions, same order; the variable is the tests' A/B switch)
        if (!fb->gsum) SSW_HIP_TRY(hipMalloc((void **)&fb->gsum, 1024 * sizeof(float)));
        hipLaunchKernelGGL(k_fb_slabsum, dim3((unsigned)((dim + 63) / 64)), dim3(256), 0, s, fb->partial, nslabs, dim, fb->gsum);
        final_partial = fb->gsum;
        final_slabs = 1;
    }
    // the last kernel writes loss + gradient straight into the pinned host mirror (mapped into the
    // device's address space): no device-to-host copies are queued behind it
    hipLaunchKernelGGL(k_fb_final, dim3(1), dim3(1024), 0, s, final_partial, final_slabs, fb->item, fb->r, n, dim,
                       by_arg ? (const float *)nullptr : (const float *)fb->w, wv,
                       fb->has_q ? fb->qhat : (const float *)nullptr, fb->has_xlx ? fb->xlx : (const float *)nullptr,
                       dev, fb->out_host_dev, fb->loss_host_dev, fb->flag_host_dev, ++fb->seqno);
    SSW_HIP_TRY(hipGetLastError());
    {   // spin on the completion word (the results are in host memory when it flips); the stream wait is the fallback
        const unsigned want = fb->seqno;
        bool seen = false;
        if (!getenv("SSW_FB_NO_SPIN")) {
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned it = 0;; ++it) {
                if (__atomic_load_n(fb->flag_host, __ATOMIC_ACQUIRE) == want) {
                    seen = true;
                    break;
                }
                if ((it & 1023u) == 1023u &&
                    std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2))
                    break;  // long evaluation (thousands of rows): sleep in the runtime instead
            }
        }
        if (!seen) SSW_HIP_TRY(hipStreamSynchronize(s));
    }
    fb->last_evals++;
    return SSW_OK;
}

// ---- two-output objective: device buffers, one closure evaluation ----------------------------
static ssw_status fb2_reserve(ssw_fb *fb) {
    if (fb->cap2 >= fb->cap && fb->cap2 > 0) return SSW_OK;
    SSW_HIP_TRY(hipStreamSynchronize(fb->stream));
    for (float **p : {&fb->y2, &fb->sw2, &fb->z2, &fb->r2, &fb->itemv, &fb->itemh, &fb->partial2}) {
        (void)hipFree(*p);
        *p = nullptr;
    }
    fb->cap2 = 0;
    const int64_t cap = fb->cap > 0 ? fb->cap : 256;
    const int64_t nslabs = (cap + FB_SLAB - 1) / FB_SLAB;
    SSW_HIP_TRY(hipMalloc((void **)&fb->y2, (size_t)2 * cap * sizeof(float)));
    SSW_HIP_TRY(hipMalloc((void **)&fb->sw2, (size_t)cap * sizeof(float)));
    SSW_HIP_TRY(hipMalloc((void **)&fb->z2, (size_t)2 * cap * sizeof(float)));
    SSW_HIP_TRY(hipMalloc((void **)&fb->r2, (size_t)2 * cap * sizeof(float)));
    SSW_HIP_TRY(hipMalloc((void **)&fb->itemv, (size_t)cap * sizeof(float)));
    SSW_HIP_TRY(hipMalloc((void **)&fb->itemh, (size_t)cap * sizeof(float)));
    SSW_HIP_TRY(hipMalloc((void **)&fb->partial2, (size_t)2 * nslabs * fb->dim * sizeof(float)));
    if (!fb->nw2) SSW_HIP_TRY(hipMalloc((void **)&fb->nw2, (size_t)2 * fb->dim * sizeof(float)));
    if (!fb->out2_host) {
        SSW_HIP_TRY(hipHostMalloc((void **)&fb->out2_host, (size_t)(2 + 2 * fb->dim) * sizeof(float), hipHostMallocMapped));
        memset(fb->out2_host, 0, (size_t)(2 + 2 * fb->dim) * sizeof(float));
        SSW_HIP_TRY(hipHostGetDevicePointer((void **)&fb->out2_host_dev, fb->out2_host, 0));
    }
    fb->cap2 = cap;
    return SSW_OK;
}

// One evaluation of MultiRegModule._step (multi_reg_module.py:64-128) at the raw weights W = fb->w_host [2, dim]:
// the data part (logits, per-row losses, X' r per output) on the device; the O(dim) part -- F.normalize and its
// chain rule, the norm and query regularisers -- here, in f32 like the reference's tensors.
// Results: fb->loss_host[0] = total loss, fb->out_host[1 .. 1 + 2 dim) = d loss / d W.  parts5 (optional):
// loss_norm, loss_queryreg, loss_queryreg2, vertical, horizontal.
static ssw_status fb_eval2(ssw_fb *fb, float l_norm, float l_query, float *parts5) {
    hipStream_t s = fb->stream;
    const int64_t n = fb->n;
    const int dim = fb->dim;
    const float *W = fb->w_host;
    float nrm[2], nw[2][1024];
    for (int c = 0; c < 2; ++c) {
        float ss = 0.f;
        for (int k = 0; k < dim; ++k) ss += W[c * dim + k] * W[c * dim + k];
        nrm[c] = std::sqrt(ss);
        const float den = std::fmax(nrm[c], 1e-12f);  // F.normalize's eps
        for (int k = 0; k < dim; ++k) nw[c][k] = W[c * dim + k] / den;
    }
    float vsum = 0.f, hsum = 0.f;
    const float *g = nullptr;  // [2, dim] d labels / d (normalised weights)
    std::vector<float> gz;
    if (n > 0) {
        SSW_REQUIRE(fb->has_targets2 && fb->cap2 >= fb->cap, "feedback: ssw_fb_set_targets2 has not been called for these rows");
        const int64_t plane = fb->cap2;
        const int nslabs = (int)((n + FB_SLAB - 1) / FB_SLAB);
        const int64_t pplane = ((fb->cap2 + FB_SLAB - 1) / FB_SLAB) * dim;
        SSW_HIP_TRY(hipMemcpyAsync(fb->nw2, &nw[0][0], (size_t)dim * sizeof(float), hipMemcpyHostToDevice, s));
        SSW_HIP_TRY(hipMemcpyAsync(fb->nw2 + dim, &nw[1][0], (size_t)dim * sizeof(float), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_fb2_logits_elem, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, fb->X, fb->nw2, n, dim, plane,
                           fb->y2, fb->sw2, fb->z2, fb->r2, fb->itemv, fb->itemh);
        const int tx = dim < 256 ? dim : 256;
        for (int c = 0; c < 2; ++c)
            hipLaunchKernelGGL(k_fb_grad, dim3((unsigned)nslabs, (unsigned)((dim + tx - 1) / tx)), dim3(tx), 0, s, fb->X,
                               fb->r2 + c * plane, n, dim, fb->partial2 + c * pplane);
        hipLaunchKernelGGL(k_fb2_reduce, dim3(1), dim3(1024), 0, s, fb->partial2, pplane, nslabs, fb->itemv, fb->itemh, n,
                           dim, fb->out2_host_dev, fb->flag_host_dev, ++fb->seqno);
        SSW_HIP_TRY(hipGetLastError());
        const unsigned want = fb->seqno;
        bool seen = false;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned it = 0;; ++it) {
            if (__atomic_load_n(fb->flag_host, __ATOMIC_ACQUIRE) == want) {
                seen = true;
                break;
            }
            if ((it & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
        }
        if (!seen) SSW_HIP_TRY(hipStreamSynchronize(s));
        vsum = fb->out2_host[0];
        hsum = fb->out2_host[1];
        g = fb->out2_host + 2;
    } else {
        gz.assign((size_t)2 * dim, 0.f);
        g = gz.data();
    }
    SSW_REQUIRE(fb->has_q, "feedback: the two-output objective needs ssw_fb_set_query first");
    const std::vector<float> &qh = fb->qhat_host;
    // regularisers (multi_reg_module.py:112-117), f32
    float loss_norm = 0.f, lq[2];
    float *grad = fb->out_host + 1;
    for (int c = 0; c < 2; ++c) {
        const float lg = std::log(nrm[c]);
        loss_norm += std::cosh(lg) - 1.f;
        float nq = 0.f;
        for (int k = 0; k < dim; ++k) nq += nw[c][k] * qh[(size_t)k];
        lq[c] = l_query * ((1.f - nq) / 2.f);
        // G = d total / d nw_c ; d nw / d W = (I - nw nw') / |W|
        float dotg = 0.f;
        for (int k = 0; k < dim; ++k) dotg += nw[c][k] * (g[c * dim + k] - 0.5f * l_query * qh[(size_t)k]);
        const float den = std::fmax(nrm[c], 1e-12f);
        const float radial = l_norm * std::sinh(lg) / den;  // d [l_norm (cosh(log |W|) - 1)] / d|W|, times d|W|/dW = nw
        for (int k = 0; k < dim; ++k) {
            const float G = g[c * dim + k] - 0.5f * l_query * qh[(size_t)k];
            grad[c * dim + k] = (G - nw[c][k] * dotg) / den + radial * nw[c][k];
        }
    }
    loss_norm *= l_norm;
    const float labels = vsum + hsum;
    const float total = ((labels + loss_norm) + lq[0]) + lq[1];
    fb->loss_host[0] = (double)total;
    if (parts5) {
        parts5[0] = loss_norm; parts5[1] = lq[0]; parts5[2] = lq[1]; parts5[3] = vsum; parts5[4] = hsum;
    }
    fb->last_evals++;
    return SSW_OK;
}

static double g_fit_eval_s = 0;  // diagnostic (SSW_FB_TIMING): seconds of the last fit spent inside closure evaluations

// ---- L-BFGS with strong-Wolfe line search ----------------------------------------------
// Restatement of the algorithm torch.optim.LBFGS implements (minFunc's lbfgs / lswolfe with
// cubic interpolation; defaults history 100, tolerance_grad 1e-7, tolerance_change 1e-9,
// c1 1e-4, c2 0.9, max_ls 25), which is what the reference drives through
// BasicTrainer.fit -> opt.step(closure) (basic_trainer.py:59-63).
namespace {

typedef std::vector<float> Vec;

// Sums in the association k_fb_fit_wg's driving wave uses (element i in lane i % 64, each lane adding its elements
// in ascending order; then inside every 16-lane row lane l takes lane l - s for s = 1, 2, 4, 8; then the four row
// totals in ascending order), so that the host-driven and the single-launch fit agree bit for bit.  (torch's own
// dot is a BLAS call with an unspecified order.)
double wave_sum_host(const double *p, size_t n) {
    double b[64], nb[64];
    for (size_t l = 0; l < 64; ++l) {
        double s = 0.0;
        for (size_t i = l; i < n; i += 64) s += p[i];
        b[l] = s;
    }
    for (size_t s = 1; s <= 8; s <<= 1) {
        for (size_t l = 0; l < 64; ++l) nb[l] = b[l] + ((l & 15) >= s ? b[l - s] : 0.0);
        for (size_t l = 0; l < 64; ++l) b[l] = nb[l];
    }
    return ((b[15] + b[31]) + b[47]) + b[63];
}
float wave_sum_host_f(const float *p, size_t n) {  // wave_sum_f's network over per-lane sums in ascending order
    float b[64], nb[64];
    for (size_t l = 0; l < 64; ++l) {
        float s = 0.f;
        for (size_t i = l; i < n; i += 64) s += p[i];
        b[l] = s;
    }
    for (size_t s = 1; s <= 8; s <<= 1) {
        for (size_t l = 0; l < 64; ++l) nb[l] = b[l] + ((l & 15) >= s ? b[l - s] : 0.f);
        for (size_t l = 0; l < 64; ++l) b[l] = nb[l];
    }
    return ((b[15] + b[31]) + b[47]) + b[63];
}
double vdot(const Vec &a, const Vec &b) {
    float p[1032];
    for (size_t i = 0; i < a.size(); ++i) p[i] = a[i] * b[i];
    return (double)wave_sum_host_f(p, a.size());
}
double vabsmax(const Vec &a) {
    float m = 0.f;
    for (float v : a) m = std::fmax(m, std::fabs(v));
    return m;
}

struct Evaluator {
    ssw_fb *fb;
    const ssw_fb_objective *o;
    FbObjDev dev;
    float pw;
    bool pairwise_active;
    int P;
    bool two = false;            // the two-output objective (fb_eval2) instead of fb_eval
    float l_norm2 = 0.f, l_query2 = 0.f;
    ssw_status status = SSW_OK;
    // f(x + t d) and its gradient
    bool eval(const Vec &x, double t, const Vec &d, double *f, Vec *g) {
        for (int i = 0; i < P; ++i) fb->w_host[i] = x[i] + (float)t * d[i];
        const auto t0 = std::chrono::steady_clock::now();
        status = two ? fb_eval2(fb, l_norm2, l_query2, nullptr) : fb_eval(fb, o, dev, pw, pairwise_active, P);
        g_fit_eval_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (status != SSW_OK) return false;
        *f = fb->loss_host[0];
        g->assign(fb->out_host + 1, fb->out_host + 1 + P);
        if (!std::isfinite(*f)) {
            set_error("feedback: loss diverged (%g) -- regression training failed with a nan", *f);
            status = SSW_ERR_NUMERIC;  // logistic_regression.py:398-401
            return false;
        }
        return true;
    }
};

// returns false on evaluator failure
bool strong_wolfe(Evaluator &E, const Vec &x, double t, const Vec &d, double f, const Vec &g, double gtd,
                  double *f_out, Vec *g_out, double *t_out, int *evals, double c1 = 1e-4, double c2 = 0.9,
                  double tol_change = 1e-9, int max_ls = 25) {
    const double d_norm = vabsmax(d);
    double f_new;
    Vec g_new;
    if (!E.eval(x, t, d, &f_new, &g_new)) return false;
    int ls_evals = 1;
    double gtd_new = vdot(g_new, d);
    double t_prev = 0, f_prev = f, gtd_prev = gtd;
    Vec g_prev = g;
    bool done = false;
    int ls_iter = 0;
    double br[2] = {0, 0}, br_f[2] = {0, 0}, br_gtd[2] = {0, 0};
    Vec br_g[2];
    int nbr = 0;
    while (ls_iter < max_ls) {
        if (f_new > (f + c1 * t * gtd) || (ls_iter > 1 && f_new >= f_prev)) {
            br[0] = t_prev; br[1] = t; br_f[0] = f_prev; br_f[1] = f_new;
            br_g[0] = g_prev; br_g[1] = g_new; br_gtd[0] = gtd_prev; br_gtd[1] = gtd_new; nbr = 2;
            break;
        }
        if (std::fabs(gtd_new) <= -c2 * gtd) {
            br[0] = t; br_f[0] = f_new; br_g[0] = g_new; nbr = 1;
            done = true;
            break;
        }
        if (gtd_new >= 0) {
            br[0] = t_prev; br[1] = t; br_f[0] = f_prev; br_f[1] = f_new;
            br_g[0] = g_prev; br_g[1] = g_new; br_gtd[0] = gtd_prev; br_gtd[1] = gtd_new; nbr = 2;
            break;
        }
        const double min_step = t + 0.01 * (t - t_prev), max_step = t * 10;
        const double tmp = t;
        t = cubic_interpolate_dev(t_prev, f_prev, gtd_prev, t, f_new, gtd_new, true, min_step, max_step);
        t_prev = tmp; f_prev = f_new; g_prev = g_new; gtd_prev = gtd_new;
        if (!E.eval(x, t, d, &f_new, &g_new)) return false;
        ls_evals++;
        gtd_new = vdot(g_new, d);
        ls_iter++;
    }
    if (ls_iter == max_ls) {
        br[0] = 0; br[1] = t; br_f[0] = f; br_f[1] = f_new; br_g[0] = g; br_g[1] = g_new; nbr = 2;
        br_gtd[0] = gtd; br_gtd[1] = gtd_new;
    }
    bool insuf = false;
    int low = 0, high = 0;
    if (nbr == 2) {
        low = br_f[0] <= br_f[1] ? 0 : 1;
        high = 1 - low;
    }
    while (!done && ls_iter < max_ls) {
        if (std::fabs(br[1] - br[0]) * d_norm < tol_change) break;
        t = cubic_interpolate_dev(br[0], br_f[0], br_gtd[0], br[1], br_f[1], br_gtd[1], false, 0, 0);
        const double bmax = std::fmax(br[0], br[1]), bmin = std::fmin(br[0], br[1]);
        const double eps = 0.1 * (bmax - bmin);
        if (std::fmin(bmax - t, t - bmin) < eps) {
            if (insuf || t >= bmax || t <= bmin) {
                t = (std::fabs(t - bmax) < std::fabs(t - bmin)) ? bmax - eps : bmin + eps;
                insuf = false;
            } else {
                insuf = true;
            }
        } else {
            insuf = false;
        }
        if (!E.eval(x, t, d, &f_new, &g_new)) return false;
        ls_evals++;
        gtd_new = vdot(g_new, d);
        ls_iter++;
        if (f_new > (f + c1 * t * gtd) || f_new >= br_f[low]) {
            br[high] = t; br_f[high] = f_new; br_g[high] = g_new; br_gtd[high] = gtd_new;
            low = br_f[0] <= br_f[1] ? 0 : 1;
            high = 1 - low;
        } else {
            if (std::fabs(gtd_new) <= -c2 * gtd) {
                done = true;
            } else if (gtd_new * (br[high] - br[low]) >= 0) {
                br[high] = br[low]; br_f[high] = br_f[low]; br_g[high] = br_g[low]; br_gtd[high] = br_gtd[low];
            }
            br[low] = t; br_f[low] = f_new; br_g[low] = g_new; br_gtd[low] = gtd_new;
        }
    }
    *t_out = br[low];
    *f_out = br_f[low];
    *g_out = br_g[low];
    *evals = ls_evals;
    return true;
}

}  // namespace

// torch.optim.LBFGS.step(closure) driven from the host: x [E.P] in/out; `zero_slot` (or -1) is a parameter slot the
// objective does not use (the bias of a fit without intercept) whose gradient is held at zero
static ssw_status lbfgs_host(Evaluator &E, Vec &x, int zero_slot, int max_iter, float lr, int *n_iter_out,
                             double *loss_out) {
    const int history = 100;
    const double tol_grad = 1e-7, tol_change = 1e-9;
    const int max_eval = max_iter * 5 / 4;
    Vec d(E.P, 0.f), g, prev_g, zero(E.P, 0.f);
    double loss;
    if (!E.eval(x, 0.0, zero, &loss, &g)) return E.status;
    if (zero_slot >= 0) g[zero_slot] = 0.f;
    int current_evals = 1, n_iter = 0;
    double t = 0, prev_loss = loss, H_diag = 1.0;
    std::vector<Vec> old_dirs, old_stps;
    std::vector<double> ro;
    bool opt_cond = vabsmax(g) <= tol_grad;
    while (!opt_cond && n_iter < max_iter) {
        n_iter++;
        if (n_iter == 1) {
            for (int i = 0; i < E.P; ++i) d[i] = -g[i];
            H_diag = 1.0;
        } else {
            Vec yv(E.P), sv(E.P);
            for (int i = 0; i < E.P; ++i) {
                yv[i] = g[i] - prev_g[i];
                sv[i] = d[i] * (float)t;
            }
            const double ys = vdot(yv, sv);
            if (ys > 1e-10) {
                if ((int)old_dirs.size() == history) {
                    old_dirs.erase(old_dirs.begin());
                    old_stps.erase(old_stps.begin());
                    ro.erase(ro.begin());
                }
                old_dirs.push_back(yv);
                old_stps.push_back(sv);
                ro.push_back(1.0 / ys);
                H_diag = ys / vdot(yv, yv);
            }
            const int m = (int)old_dirs.size();
            std::vector<double> al((size_t)m);
            Vec q(E.P);
            for (int i = 0; i < E.P; ++i) q[i] = -g[i];
            for (int i = m - 1; i >= 0; --i) {
                al[(size_t)i] = vdot(old_stps[(size_t)i], q) * ro[(size_t)i];
                for (int j = 0; j < E.P; ++j) q[j] -= (float)al[(size_t)i] * old_dirs[(size_t)i][j];
            }
            for (int j = 0; j < E.P; ++j) d[j] = q[j] * (float)H_diag;
            for (int i = 0; i < m; ++i) {
                const double be = vdot(old_dirs[(size_t)i], d) * ro[(size_t)i];
                for (int j = 0; j < E.P; ++j) d[j] += (float)(al[(size_t)i] - be) * old_stps[(size_t)i][j];
            }
        }
        prev_g = g;
        prev_loss = loss;
        if (n_iter == 1) {
            double pa[1032];
            for (size_t i = 0; i < g.size(); ++i) pa[i] = (double)std::fabs(g[i]);
            const double gs = wave_sum_host(pa, g.size());
            t = std::fmin(1.0, 1.0 / gs) * lr;
        } else {
            t = lr;
        }
        const double gtd = vdot(g, d);
        if (gtd > -tol_change) break;
        double f_new, t_new;
        Vec g_new;
        int ls_evals = 0;
        if (!strong_wolfe(E, x, t, d, loss, g, gtd, &f_new, &g_new, &t_new, &ls_evals)) return E.status;
        loss = f_new;
        g = g_new;
        if (zero_slot >= 0) g[zero_slot] = 0.f;
        t = t_new;
        for (int i = 0; i < E.P; ++i) x[i] += (float)t * d[i];
        opt_cond = vabsmax(g) <= tol_grad;
        current_evals += ls_evals;
        if (n_iter == max_iter) break;
        if (current_evals >= max_eval) break;
        if (opt_cond) break;
        double dm = 0;
        for (int i = 0; i < E.P; ++i) dm = std::fmax(dm, std::fabs((double)d[i] * t));
        if (dm <= tol_change) break;
        if (std::fabs(loss - prev_loss) < tol_change) break;
    }
    *n_iter_out = n_iter;
    *loss_out = loss;
    return SSW_OK;
}

extern "C" {

ssw_status ssw_fb_destroy(ssw_fb *fb) {
    if (!fb) return SSW_OK;
    DeviceGuard guard(fb->device);
    if (fb->stream) (void)hipStreamSynchronize(fb->stream);
    fb->th_stage.release();
    fb->drawn_stage.release();
    (void)hipFree(fb->th);
    (void)hipFree(fb->drawn);
    fb->rows_stage.release();
    fb->y_stage.release();
    fb->q_stage.release();
    fb->coef_stage.release();
    (void)hipFree(fb->X);
    (void)hipFree(fb->mu);
    (void)hipFree(fb->y);
    (void)hipFree(fb->coef);
    (void)hipFree(fb->z);
    (void)hipFree(fb->item);
    (void)hipFree(fb->r);
    (void)hipFree(fb->rows);
    (void)hipFree(fb->partial);
    (void)hipFree(fb->gsum);
    (void)hipFree(fb->rankg);
    for (float *p2 : {fb->y2, fb->sw2, fb->z2, fb->r2, fb->itemv, fb->itemh, fb->partial2, fb->nw2}) (void)hipFree(p2);
    if (fb->out2_host) (void)hipHostFree(fb->out2_host);
    (void)hipFree(fb->colsum);
    (void)hipFree(fb->w);
    (void)hipFree(fb->qhat);
    (void)hipFree(fb->xlx);
    (void)hipFree(fb->out);
    (void)hipFree(fb->loss_dev);
    (void)hipFree(fb->hist);
    if (fb->loss_host) (void)hipHostFree(fb->loss_host);
    if (fb->out_host) (void)hipHostFree(fb->out_host);
    if (fb->w_host) (void)hipHostFree(fb->w_host);
    if (fb->flag_host) (void)hipHostFree(fb->flag_host);
    if (fb->stream) (void)hipStreamDestroy(fb->stream);
    delete fb;
    return SSW_OK;
}

ssw_status ssw_fb_create(int32_t device, int32_t dim, ssw_fb **out) {
    SSW_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    if (dim <= 0 || dim % 4 != 0 || dim > 1024) {
        set_error("feedback: dim=%d unsupported (multiple of 4, <= 1024)", dim);
        return SSW_ERR_UNSUPPORTED;
    }
    DeviceGuard guard(device);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", device);
        return SSW_ERR_HIP;
    }
    ssw_fb *fb = new (std::nothrow) ssw_fb();
    if (!fb) return SSW_ERR_NOMEM;
    fb->device = device;
    fb->dim = dim;
    const size_t outn = (size_t)(1 + 2 * dim + 8);  // [loss, gradient (dim + 1, or 2 dim for the two-output objective), parts]
    if (hipStreamCreateWithFlags(&fb->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void **)&fb->mu, (size_t)dim * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&fb->w, (size_t)(dim + 1) * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&fb->qhat, (size_t)dim * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&fb->xlx, (size_t)dim * dim * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&fb->out, outn * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&fb->loss_dev, sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&fb->hist, (size_t)2 * FIT_HISTORY * 1024 * sizeof(float)) != hipSuccess ||
        hipHostMalloc((void **)&fb->loss_host, sizeof(double), hipHostMallocMapped) != hipSuccess ||
        hipHostMalloc((void **)&fb->out_host, outn * sizeof(float), hipHostMallocMapped) != hipSuccess ||
        hipHostMalloc((void **)&fb->flag_host, 128, hipHostMallocMapped) != hipSuccess ||
        hipHostMalloc((void **)&fb->w_host, (size_t)(2 * dim + 1) * sizeof(float), hipHostMallocDefault) != hipSuccess) {
        set_error("feedback: allocation failed");
        ssw_fb_destroy(fb);
        return SSW_ERR_NOMEM;
    }
    // hipHostMalloc does not zero: a stale completion word equal to the first sequence number would end the
    // first wait before the kernel has written anything
    memset(fb->flag_host, 0, 128);
    memset(fb->out_host, 0, outn * sizeof(float));
    memset(fb->loss_host, 0, sizeof(double));
    if (hipHostGetDevicePointer((void **)&fb->out_host_dev, fb->out_host, 0) != hipSuccess ||
        hipHostGetDevicePointer((void **)&fb->loss_host_dev, fb->loss_host, 0) != hipSuccess ||
        hipHostGetDevicePointer((void **)&fb->flag_host_dev, fb->flag_host, 0) != hipSuccess) {
        set_error("feedback: pinned host memory is not mapped into the device address space");
        ssw_fb_destroy(fb);
        return SSW_ERR_HIP;
    }
    *out = fb;
    return SSW_OK;
}

ssw_status ssw_fb_set_data(ssw_fb *fb, const float *X_host, int64_t n, int32_t center) {
    SSW_REQUIRE(fb != nullptr && n >= 0 && (n == 0 || X_host != nullptr), "bad argument");
    DeviceGuard guard(fb->device);
    SSW_TRY(fb_reserve(fb, n));
    fb->n = n;
    fb->has_targets2 = false;
    fb->targets_on_device = false;  // a new row set: the pseudo-sample's device-made targets (if any) went with the old one
    if (n > 0)
        SSW_HIP_TRY(hipMemcpyAsync(fb->X, X_host, (size_t)n * fb->dim * sizeof(float), hipMemcpyHostToDevice, fb->stream));
    SSW_TRY(fb_center(fb, center));
    SSW_HIP_TRY(hipStreamSynchronize(fb->stream));
    return SSW_OK;
}

ssw_status ssw_fb_set_data_from_device(ssw_fb *fb, const float *dev_matrix, int64_t n_matrix_rows,
                                       const int64_t *rows_host, int64_t n, int32_t center) {
    SSW_REQUIRE(fb != nullptr && dev_matrix != nullptr && n >= 0 && (n == 0 || rows_host != nullptr), "bad argument");
    for (int64_t i = 0; i < n; ++i)
        SSW_REQUIRE(rows_host[i] >= 0 && rows_host[i] < n_matrix_rows, "row %lld outside [0, %lld)",
                    (long long)rows_host[i], (long long)n_matrix_rows);
    DeviceGuard guard(fb->device);
    SSW_TRY(fb_reserve(fb, n));
    fb->n = n;
    fb->has_targets2 = false;
    fb->targets_on_device = false;
    if (n > 0) {
        SSW_TRY(fb->rows_stage.push(fb->rows, rows_host, (size_t)n * sizeof(int64_t), fb->stream));
        hipLaunchKernelGGL(k_fb_gather_rows, dim3((unsigned)n), dim3(128), 0, fb->stream, dev_matrix, fb->rows, n,
                           fb->dim, fb->X);
        SSW_HIP_TRY(hipGetLastError());
    }
    SSW_TRY(fb_center(fb, center));
    return SSW_OK;  // no wait: every consumer (the fit, lossgrad, get_mean, scores) is ordered behind this on the stream
}

ssw_status ssw_fb_set_pseudo_sample(ssw_fb *fb, const float *dev_matrix, int64_t n_matrix_rows, const double *dev_scores,
                                    const int64_t *labelled_rows_sorted, const float *labelled_y, int64_t n_lab,
                                    const int64_t *drawn, int64_t n_drawn, float real_weight, int32_t center) {
    SSW_REQUIRE(fb != nullptr && dev_matrix != nullptr && dev_scores != nullptr && n_lab >= 0 && n_drawn >= 0, "bad argument");
    SSW_REQUIRE((n_lab == 0 || (labelled_rows_sorted && labelled_y)) && (n_drawn == 0 || drawn), "bad argument");
    const int64_t n = n_lab + n_drawn, n_unl = n_matrix_rows - n_lab;
    for (int64_t j = 0; j < n_lab; ++j) {
        SSW_REQUIRE(labelled_rows_sorted[j] >= 0 && labelled_rows_sorted[j] < n_matrix_rows &&
                    (j == 0 || labelled_rows_sorted[j] > labelled_rows_sorted[j - 1]), "labelled rows must ascend inside [0, %lld)",
                    (long long)n_matrix_rows);
        SSW_REQUIRE(std::isfinite(labelled_y[j]), "target %lld is not finite", (long long)j);
    }
    for (int64_t i = 0; i < n_drawn; ++i)
        SSW_REQUIRE(drawn[i] >= 0 && drawn[i] < n_unl, "draw %lld outside the %lld unlabelled rows", (long long)drawn[i], (long long)n_unl);
    DeviceGuard guard(fb->device);
    SSW_TRY(fb_reserve(fb, n));
    if (n_lab + 1 > fb->th_cap) {
        SSW_HIP_TRY(hipStreamSynchronize(fb->stream));
        (void)hipFree(fb->th);
        fb->th = nullptr;
        int64_t cap = 1024;
        while (cap < n_lab + 1) cap <<= 1;
        SSW_HIP_TRY(hipMalloc((void **)&fb->th, (size_t)cap * sizeof(int64_t)));
        fb->th_cap = cap;
    }
    if (n_drawn + 1 > fb->drawn_cap) {
        SSW_HIP_TRY(hipStreamSynchronize(fb->stream));
        (void)hipFree(fb->drawn);
        fb->drawn = nullptr;
        int64_t cap = 16384;
        while (cap < n_drawn + 1) cap <<= 1;
        SSW_HIP_TRY(hipMalloc((void **)&fb->drawn, (size_t)cap * sizeof(int64_t)));
        fb->drawn_cap = cap;
    }
    fb->n = n;
    fb->has_targets2 = false;
    fb->targets_on_device = true;
    // host copies for the objective's set-up: the logistic objective reads the weights only
    fb->y_host.assign((size_t)n, 0.f);
    fb->sw_host.assign((size_t)n, 1.f);
    std::vector<int64_t> th((size_t)std::max<int64_t>(n_lab, 1));
    for (int64_t j = 0; j < n_lab; ++j) {
        fb->y_host[(size_t)j] = labelled_y[j];
        fb->sw_host[(size_t)j] = real_weight;
        th[(size_t)j] = labelled_rows_sorted[j] - j;
    }
    if (n > 0) {
        // labelled part: row ids and targets as ssw_fb_set_data_from_device / ssw_fb_set_targets upload them
        if (n_lab > 0) {
            SSW_TRY(fb->rows_stage.push(fb->rows, labelled_rows_sorted, (size_t)n_lab * sizeof(int64_t), fb->stream));
            SSW_TRY(fb->y_stage.push(fb->y, fb->y_host.data(), (size_t)n_lab * sizeof(float), fb->stream));
            SSW_TRY(fb->th_stage.push(fb->th, th.data(), (size_t)n_lab * sizeof(int64_t), fb->stream));
        }
        if (n_drawn > 0) {
            SSW_TRY(fb->drawn_stage.push(fb->drawn, drawn, (size_t)n_drawn * sizeof(int64_t), fb->stream));
            hipLaunchKernelGGL(k_fb_pseudo_rows, dim3((unsigned)((n_drawn + 255) / 256)), dim3(256), 0, fb->stream, fb->th, (int)n_lab,
                               fb->drawn, n_drawn, dev_scores, n_matrix_rows, fb->rows, fb->y);
        }
        hipLaunchKernelGGL(k_fb_gather_rows, dim3((unsigned)n), dim3(128), 0, fb->stream, dev_matrix, fb->rows, n, fb->dim, fb->X);
        SSW_HIP_TRY(hipGetLastError());
    }
    SSW_TRY(fb_center(fb, center));
    return SSW_OK;  // no wait: the fit is ordered behind this on the stream
}

ssw_status ssw_fb_set_targets(ssw_fb *fb, const float *y_host, const float *sample_weight_or_null) {
    SSW_REQUIRE(fb != nullptr && (fb->n == 0 || y_host != nullptr), "bad argument");
    DeviceGuard guard(fb->device);
    const int64_t n = fb->n;
    fb->targets_on_device = false;
    fb->y_host.assign(y_host, y_host + n);
    if (sample_weight_or_null)
        fb->sw_host.assign(sample_weight_or_null, sample_weight_or_null + n);
    else
        fb->sw_host.clear();
    for (int64_t i = 0; i < n; ++i)
        SSW_REQUIRE(std::isfinite(y_host[i]), "target %lld is not finite", (long long)i);
    if (n > 0) {
        SSW_TRY(fb->y_stage.push(fb->y, fb->y_host.data(), (size_t)n * sizeof(float), fb->stream));
    }
    return SSW_OK;
}

ssw_status ssw_fb_set_query(ssw_fb *fb, const float *q_host) {
    SSW_REQUIRE(fb != nullptr && q_host != nullptr, "bad argument");
    DeviceGuard guard(fb->device);
    double nn = 0;
    for (int i = 0; i < fb->dim; ++i) nn += (double)q_host[i] * q_host[i];
    SSW_REQUIRE(nn > 0 && std::isfinite(nn), "query vector has zero or non-finite norm");
    const double inv = 1.0 / std::fmax(std::sqrt(nn), 1e-12);
    std::vector<float> qh((size_t)fb->dim);
    for (int i = 0; i < fb->dim; ++i) qh[(size_t)i] = (float)(q_host[i] * inv);  // F.normalize
    if (fb->has_q && fb->qhat_host.size() == qh.size() && memcmp(fb->qhat_host.data(), qh.data(), qh.size() * sizeof(float)) == 0)
        return SSW_OK;  // a session hands over the same text vector every round: already installed
    SSW_TRY(fb->q_stage.push(fb->qhat, qh.data(), (size_t)fb->dim * sizeof(float), fb->stream));
    fb->qhat_host = qh;
    fb->has_q = true;
    return SSW_OK;
}

ssw_status ssw_fb_set_xlx(ssw_fb *fb, const float *xlx_host) {
    SSW_REQUIRE(fb != nullptr && xlx_host != nullptr, "bad argument");
    DeviceGuard guard(fb->device);
    SSW_HIP_TRY(hipMemcpy(fb->xlx, xlx_host, (size_t)fb->dim * fb->dim * sizeof(float), hipMemcpyHostToDevice));
    fb->has_xlx = true;
    return SSW_OK;
}

ssw_status ssw_fb_get_mean(ssw_fb *fb, float *out_mu_host) {
    SSW_REQUIRE(fb != nullptr && out_mu_host != nullptr, "bad argument");
    DeviceGuard guard(fb->device);
    SSW_HIP_TRY(hipMemcpyAsync(out_mu_host, fb->mu, (size_t)fb->dim * sizeof(float), hipMemcpyDeviceToHost, fb->stream));
    SSW_HIP_TRY(hipStreamSynchronize(fb->stream));  // behind the centring kernels of set_data on the same stream
    return SSW_OK;
}

ssw_status ssw_fb_lossgrad(ssw_fb *fb, const ssw_fb_objective *obj, const float *w_host, float *out_loss,
                           float *out_grad, float *out_parts4_or_null) {
    SSW_REQUIRE(fb && obj && w_host && out_loss && out_grad, "NULL argument");
    DeviceGuard guard(fb->device);
    const int P = fb->dim + ((obj->kind == SSW_FB_LOGREG && obj->fit_intercept) ? 1 : 0);
    FbObjDev dev;
    float pw = 1.f;
    bool pa = false;
    SSW_TRY(fb_prepare(fb, obj, &dev, &pw, &pa));
    memcpy(fb->w_host, w_host, (size_t)P * sizeof(float));
    if (P == fb->dim) fb->w_host[fb->dim] = 0.f;
    SSW_TRY(fb_eval(fb, obj, dev, pw, pa, fb->dim + 1));
    *out_loss = (float)fb->loss_host[0];
    memcpy(out_grad, fb->out_host + 1, (size_t)P * sizeof(float));
    if (out_parts4_or_null) memcpy(out_parts4_or_null, fb->out_host + 1 + fb->dim + 1, 4 * sizeof(float));
    return SSW_OK;
}

ssw_status ssw_fb_scores(ssw_fb *fb, const float *w_host, int32_t has_bias, float *out_logits) {
    SSW_REQUIRE(fb && w_host && (fb->n == 0 || out_logits), "NULL argument");
    DeviceGuard guard(fb->device);
    if (fb->n == 0) return SSW_OK;
    memcpy(fb->w_host, w_host, (size_t)(fb->dim + (has_bias ? 1 : 0)) * sizeof(float));
    SSW_HIP_TRY(hipMemcpyAsync(fb->w, fb->w_host, (size_t)(fb->dim + 1) * sizeof(float), hipMemcpyHostToDevice, fb->stream));
    hipLaunchKernelGGL(k_fb_logits, dim3((unsigned)((fb->n + 3) / 4)), dim3(256), 0, fb->stream, fb->X, fb->w, fb->n,
                       fb->dim, has_bias, fb->z);
    SSW_HIP_TRY(hipMemcpyAsync(out_logits, fb->z, (size_t)fb->n * sizeof(float), hipMemcpyDeviceToHost, fb->stream));
    SSW_HIP_TRY(hipStreamSynchronize(fb->stream));
    return SSW_OK;
}

ssw_status ssw_fb_fit(ssw_fb *fb, const ssw_fb_objective *obj, float *w_inout, int32_t max_iter, float lr,
                      int32_t *out_iters, int32_t *out_evals, float *out_final_loss) {
    SSW_REQUIRE(fb && obj && w_inout, "NULL argument");
    SSW_REQUIRE(max_iter >= 1, "max_iter < 1");
    DeviceGuard guard(fb->device);
    const int P = fb->dim + ((obj->kind == SSW_FB_LOGREG && obj->fit_intercept) ? 1 : 0);
    Evaluator E;
    E.fb = fb;
    E.o = obj;
    E.P = fb->dim + 1;  // the device always sees dim+1 parameters; slot dim is the (maybe unused) bias
    const auto t_fit0 = std::chrono::steady_clock::now();
    SSW_TRY(fb_prepare(fb, obj, &E.dev, &E.pw, &E.pairwise_active));
    const double prep_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_fit0).count();
    g_fit_eval_s = 0;
    fb->last_evals = 0;
    fb->last_fit_on_device = false;
    // ---- the whole step(closure) in one launch (k_fb_fit_wg) when the labelled set fits one workgroup's reach
    if (fb->n <= FIT_WG_MAX_ROWS && fb->dim + 1 <= 1024 && obj->kind != SSW_FB_RANKREG && !getenv("SSW_FB_HOST_DRIVER")) {
        for (int i = 0; i < P; ++i) SSW_REQUIRE(std::isfinite(w_inout[i]), "initial weight %d is not finite", i);
        FitWgArgs a;
        memset(&a, 0, sizeof(a));
        a.X = fb->X; a.y = fb->y; a.coef = fb->coef;
        a.qhat = fb->has_q ? fb->qhat : nullptr;
        a.xlx = fb->has_xlx ? fb->xlx : nullptr;
        a.hist_dirs = fb->hist;
        a.hist_stps = fb->hist + (size_t)FIT_HISTORY * 1024;
        a.out_w = fb->out_host_dev + 1;
        a.out_loss = fb->loss_host_dev;
        a.out_counts = reinterpret_cast<int *>(fb->flag_host_dev) + 4;
        a.done_flag = fb->flag_host_dev;
        a.seqno = ++fb->seqno;
        a.n = (int)fb->n; a.dim = fb->dim; a.P = P;
        const bool pairwise = obj->kind == SSW_FB_MULTIREG && obj->loss_type != SSW_FB_LOSS_CE;
        a.label_mode = !pairwise ? 0 : !E.pairwise_active ? 3 : obj->loss_type == SSW_FB_LOSS_PAIRWISE_LOGISTIC ? 2 : 1;
        a.pw = E.pw; a.margin = obj->margin; a.max_iter = max_iter; a.lr = lr; a.obj = E.dev;
        FbW w0v;
        memset(&w0v, 0, sizeof(w0v));
        if (fb->dim + 1 <= FB_ARG_FLOATS) {
            memcpy(w0v.v, w_inout, (size_t)P * sizeof(float));
        } else {
            memcpy(fb->w_host, w_inout, (size_t)P * sizeof(float));
            if (P == fb->dim) fb->w_host[fb->dim] = 0.f;
            SSW_HIP_TRY(hipMemcpyAsync(fb->w, fb->w_host, (size_t)(fb->dim + 1) * sizeof(float), hipMemcpyHostToDevice, fb->stream));
            a.w0_or_null = fb->w;
        }
        a.onepass = a.label_mode == 0 && fb->dim == FIT_OP_DIM && fb->n >= 1 && !getenv("SSW_FB_TWO_PASS");  // the variable is the tests' A/B switch
        const size_t lds = fit_wg_lds_bytes(fb->dim, a.onepass != 0);
        auto kern = fb->dim + 1 <= 9 * 64 ? k_fb_fit_wg<9> : k_fb_fit_wg<16>;
        // the attribute is per device: one flag per (device, kernel)
        static bool attr_done[64][2] = {};
        const int dslot = fb->device >= 0 && fb->device < 64 ? fb->device : -1, kslot = fb->dim + 1 <= 9 * 64 ? 0 : 1;
        if (dslot < 0 || !attr_done[dslot][kslot]) {
            SSW_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            if (dslot >= 0) attr_done[dslot][kslot] = true;
        }
        hipLaunchKernelGGL(kern, dim3(1), dim3(1024), lds, fb->stream, a, w0v);
        SSW_HIP_TRY(hipGetLastError());
        {
            const unsigned want = fb->seqno;
            bool seen = false;
            if (!getenv("SSW_FB_NO_SPIN")) {
                const auto t0 = std::chrono::steady_clock::now();
                for (unsigned it = 0;; ++it) {
                    if (__atomic_load_n(fb->flag_host, __ATOMIC_ACQUIRE) == want) {
                        seen = true;
                        break;
                    }
                    if ((it & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
                }
            }
            if (!seen) SSW_HIP_TRY(hipStreamSynchronize(fb->stream));
        }
        const int *counts = reinterpret_cast<const int *>(fb->flag_host) + 4;
        fb->last_iters = counts[0];
        fb->last_evals = counts[1];
        fb->last_fit_on_device = true;
        if (counts[2] != 0) {
            set_error("feedback: loss diverged -- regression training failed with a nan");
            return SSW_ERR_NUMERIC;  // logistic_regression.py:398-401
        }
        for (int i = 0; i < P; ++i) {
            if (!std::isfinite(fb->out_host[1 + i])) {
                set_error("feedback: weights diverged");
                return SSW_ERR_NUMERIC;
            }
            w_inout[i] = fb->out_host[1 + i];
        }
        if (getenv("SSW_FB_TIMING"))
            fprintf(stderr, "fit (one launch): n=%lld iters=%d evals=%d total %.1f us, prepare %.1f; kernel %.1f us = logits %.1f + "
                    "labels %.1f + gradient %.1f + final %.1f + driver %.1f (%.0f MHz)\n", (long long)fb->n, counts[0], counts[1],
                    1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_fit0).count(), 1e6 * prep_s,
                    counts[7] * 0.01, counts[3] * 0.01, counts[4] * 0.01, counts[5] * 0.01, counts[6] * 0.01,
                    (counts[7] - counts[3] - counts[4] - counts[5] - counts[6]) * 0.01,
                    counts[7] > 0 ? 16.0 * counts[8] / (counts[7] * 0.01) : 0.0);
        if (getenv("SSW_FB_TIMING"))
            fprintf(stderr, "   of the driver: two-loop recursion %.1f us\n", counts[10] * 0.01);
        if (getenv("SSW_FB_TIMING") && counts[11])
            fprintf(stderr, "   one pass (SSW_FIT_STAMPS build), cycles per evaluation: %.0f, %.0f, %.0f\n",
                    16.0 * counts[11] / counts[1], 16.0 * counts[12] / counts[1], 16.0 * counts[13] / counts[1]);
        if (out_iters) *out_iters = counts[0];
        if (out_evals) *out_evals = counts[1];
        if (out_final_loss) *out_final_loss = (float)fb->loss_host[0];
        return SSW_OK;
    }
    Vec x(E.P, 0.f);
    for (int i = 0; i < P; ++i) {
        SSW_REQUIRE(std::isfinite(w_inout[i]), "initial weight %d is not finite", i);
        x[i] = w_inout[i];
    }
    int n_iter = 0;
    double loss = 0;
    SSW_TRY(lbfgs_host(E, x, P == fb->dim ? fb->dim : -1, max_iter, lr, &n_iter, &loss));
    for (int i = 0; i < P; ++i) {
        if (!std::isfinite(x[i])) {
            set_error("feedback: weights diverged");
            return SSW_ERR_NUMERIC;
        }
        w_inout[i] = x[i];
    }
    fb->last_iters = n_iter;
    if (getenv("SSW_FB_TIMING"))
        fprintf(stderr, "fit: n=%lld iters=%d evals=%d total %.1f us, prepare %.1f, evaluations %.1f\n", (long long)fb->n,
                n_iter, fb->last_evals,
                1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_fit0).count(), 1e6 * prep_s,
                1e6 * g_fit_eval_s);
    if (out_iters) *out_iters = n_iter;
    if (out_evals) *out_evals = fb->last_evals;
    if (out_final_loss) *out_final_loss = (float)loss;
    return SSW_OK;
}

// ---- MultiRegModule (multi_reg_neg loop): two-output objective -------------------------------
ssw_status ssw_fb_set_targets2(ssw_fb *fb, const float *y2_host, const float *sample_weight_or_null) {
    SSW_REQUIRE(fb != nullptr && (fb->n == 0 || y2_host != nullptr), "bad argument");
    DeviceGuard guard(fb->device);
    SSW_REQUIRE(2 * fb->dim <= 1024, "feedback: the two-output objective takes dim <= 512, got %d", fb->dim);
    SSW_TRY(fb2_reserve(fb));
    const int64_t n = fb->n, plane = fb->cap2;
    std::vector<float> buf((size_t)3 * n);
    for (int64_t i = 0; i < n; ++i) {
        SSW_REQUIRE(std::isfinite(y2_host[2 * i]) && std::isfinite(y2_host[2 * i + 1]), "target %lld is not finite", (long long)i);
        buf[(size_t)i] = y2_host[2 * i];
        buf[(size_t)(n + i)] = y2_host[2 * i + 1];
        buf[(size_t)(2 * n + i)] = sample_weight_or_null ? sample_weight_or_null[i] : 1.f;
        SSW_REQUIRE(std::isfinite(buf[(size_t)(2 * n + i)]), "sample weight %lld is not finite", (long long)i);
    }
    if (n > 0) {
        SSW_HIP_TRY(hipMemcpyAsync(fb->y2, buf.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice, fb->stream));
        SSW_HIP_TRY(hipMemcpyAsync(fb->y2 + plane, buf.data() + n, (size_t)n * sizeof(float), hipMemcpyHostToDevice, fb->stream));
        SSW_HIP_TRY(hipMemcpyAsync(fb->sw2, buf.data() + 2 * n, (size_t)n * sizeof(float), hipMemcpyHostToDevice, fb->stream));
        SSW_HIP_TRY(hipStreamSynchronize(fb->stream));
    }
    fb->has_targets2 = true;
    return SSW_OK;
}

ssw_status ssw_fb_lossgrad2(ssw_fb *fb, const float *W_host, float reg_norm_lambda, float reg_query_lambda,
                            float *out_loss, float *out_grad, float *out_parts5_or_null) {
    SSW_REQUIRE(fb && W_host && out_loss && out_grad, "NULL argument");
    DeviceGuard guard(fb->device);
    SSW_REQUIRE(2 * fb->dim <= 1024, "feedback: the two-output objective takes dim <= 512, got %d", fb->dim);
    memcpy(fb->w_host, W_host, (size_t)2 * fb->dim * sizeof(float));
    SSW_TRY(fb_eval2(fb, reg_norm_lambda, reg_query_lambda, out_parts5_or_null));
    *out_loss = (float)fb->loss_host[0];
    memcpy(out_grad, fb->out_host + 1, (size_t)2 * fb->dim * sizeof(float));
    return SSW_OK;
}

ssw_status ssw_fb_fit2(ssw_fb *fb, float *W_inout, float reg_norm_lambda, float reg_query_lambda, int32_t max_iter,
                       float lr, int32_t *out_iters, int32_t *out_evals, float *out_final_loss) {
    SSW_REQUIRE(fb && W_inout, "NULL argument");
    SSW_REQUIRE(max_iter >= 1, "max_iter < 1");
    DeviceGuard guard(fb->device);
    SSW_REQUIRE(2 * fb->dim <= 1024, "feedback: the two-output objective takes dim <= 512, got %d", fb->dim);
    Evaluator E;
    E.fb = fb;
    E.o = nullptr;
    E.P = 2 * fb->dim;
    E.two = true;
    E.l_norm2 = reg_norm_lambda;
    E.l_query2 = reg_query_lambda;
    E.pw = 1.f;
    E.pairwise_active = false;
    memset(&E.dev, 0, sizeof(E.dev));
    fb->last_evals = 0;
    fb->last_fit_on_device = false;
    Vec x((size_t)E.P);
    for (int i = 0; i < E.P; ++i) {
        SSW_REQUIRE(std::isfinite(W_inout[i]), "initial weight %d is not finite", i);
        x[(size_t)i] = W_inout[i];
    }
    int n_iter = 0;
    double loss = 0;
    SSW_TRY(lbfgs_host(E, x, -1, max_iter, lr, &n_iter, &loss));
    for (int i = 0; i < E.P; ++i) {
        if (!std::isfinite(x[(size_t)i])) {
            set_error("feedback: weights diverged");
            return SSW_ERR_NUMERIC;
        }
        W_inout[i] = x[(size_t)i];
    }
    fb->last_iters = n_iter;
    if (out_iters) *out_iters = n_iter;
    if (out_evals) *out_evals = fb->last_evals;
    if (out_final_loss) *out_final_loss = (float)loss;
    return SSW_OK;
}

ssw_status ssw_fb_reset(ssw_fb *fb) {
    SSW_REQUIRE(fb != nullptr, "NULL argument");
    DeviceGuard guard(fb->device);
    SSW_HIP_TRY(hipStreamSynchronize(fb->stream));
    fb->n = 0;
    fb->has_q = fb->has_xlx = false;
    fb->has_targets2 = false;
    fb->targets_on_device = false;
    fb->y_host.clear();
    fb->sw_host.clear();
    fb->last_iters = fb->last_evals = 0;
    fb->last_fit_on_device = false;
    return SSW_OK;
}

ssw_status ssw_fb_last_fit_on_device(const ssw_fb *fb, int32_t *out) {
    SSW_REQUIRE(fb && out, "NULL argument");
    *out = fb->last_fit_on_device ? 1 : 0;
    return SSW_OK;
}

// The pairwise losses evaluated directly on given scores: the kernel the fit uses (k_fb_pairwise), without a
// data matrix in front of it.  The reference's functions normalise nothing (rank_loss.py:34-95); RegModule
// divides column j by max_inversions_j and multiplies by sample_weight_j (multi_reg.py:106-121):
// coef_j is that per-item factor times max_inversions_j, i.e. coef = max_inversions reproduces the raw column
// sums / gradient of rank_loss.py and coef = sample_weight what RegModule optimises.
ssw_status ssw_rank_pairwise(int32_t device, int32_t logistic, const float *target_host, const float *scores_host,
                             const float *coef_host_or_null, int32_t n, float margin, double *out_item_loss,
                             float *out_grad) {
    SSW_REQUIRE(n >= 0 && n <= FB_MAX_PAIRWISE, "rank_pairwise: n = %d outside [0, %d]", n, FB_MAX_PAIRWISE);
    if (n == 0) return SSW_OK;
    SSW_REQUIRE(target_host && scores_host && out_item_loss && out_grad, "NULL argument");
    DeviceGuard guard(device);
    float *buf = nullptr;  // z | y | coef | r   then f64 item losses
    double *item = nullptr;
    SSW_HIP_TRY(hipMalloc((void **)&buf, (size_t)4 * n * sizeof(float)));
    if (hipMalloc((void **)&item, (size_t)n * sizeof(double)) != hipSuccess) {
        (void)hipFree(buf);
        set_error("rank_pairwise: out of device memory");
        return SSW_ERR_HIP;
    }
    std::vector<float> ones;
    if (!coef_host_or_null) ones.assign((size_t)n, 1.f);
    ssw_status st = SSW_OK;
    auto run = [&]() -> ssw_status {
        SSW_HIP_TRY(hipMemcpy(buf, scores_host, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
        SSW_HIP_TRY(hipMemcpy(buf + n, target_host, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
        SSW_HIP_TRY(hipMemcpy(buf + 2 * n, coef_host_or_null ? coef_host_or_null : ones.data(), (size_t)n * sizeof(float),
                              hipMemcpyHostToDevice));
        const size_t lds = (size_t)3 * n * sizeof(float);
        if (logistic)
            hipLaunchKernelGGL(k_fb_pairwise<1>, dim3(1), dim3(1024), lds, 0, buf, buf + n, buf + 2 * n, margin, n, item,
                               buf + 3 * n);
        else
            hipLaunchKernelGGL(k_fb_pairwise<0>, dim3(1), dim3(1024), lds, 0, buf, buf + n, buf + 2 * n, margin, n, item,
                               buf + 3 * n);
        SSW_HIP_TRY(hipGetLastError());
        SSW_HIP_TRY(hipMemcpy(out_grad, buf + 3 * n, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
        SSW_HIP_TRY(hipMemcpy(out_item_loss, item, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
        return SSW_OK;
    };
    st = run();
    (void)hipFree(buf);
    (void)hipFree(item);
    return st;
}

}  // extern "C"
