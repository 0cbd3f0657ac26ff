// select.hip -- exact top-k of the per-image best scores (gfx950 / MI355X)
//
// Replaces the reference's full sort + pandas passes:
//   np.argsort(-scores)                      seesaw/indices/multiscale/multiscale_index.py:172
//   dbidx gather / ~isin(exclude) / np.unique(return_index) / sort / [:topk]
//                                            seesaw/indices/multiscale/multiscale_index.py:177-199
//   np.argsort(-scores)[:topk]               seesaw/indices/coarse/coarse_index.py:75
// Only the first `shortlist_size` distinct images of that order are ever used, so the
// N-key sort is replaced by an exact radix SELECT:
//   1. image_max    : rows are stored sorted by image, so "first occurrence of each image
//                     in descending score order" == per-image max over a contiguous row
//                     range (lowest row on ties).            [skipped for 1 vector/image]
//   2. hist<1>/pick : 4096-bin histogram of the top 12 bits of the order-preserving u32
//                     key of the image score; find the bin holding the k-th largest.
//   3. hist<2>/pick : same on the next 12 bits, only inside that bin, only when the bin
//                     is too full to sort directly.
//   4. collect      : every image at or above the (12- or 24-bit) prefix is appended as a
//                     64-bit composite key (score_key << 32 | ~image) -- unique per image,
//                     so there are no ties left to break.
//   5. final        : one workgroup bitonic-sorts the <= 8192 candidates in LDS and emits
//                     the k largest = (score desc, image asc).
// Excluded images (InteractiveQuery.returned, query_interface.py:43-48) are skipped in
// steps 2-4 through a device bitmap.  Bound: reads 4 B/image three times -- < 0.6 % of
// the scan's traffic at dim=512.
#include <algorithm>
#include <iterator>

#include "ssw_common.h"

namespace ssw {

namespace {

constexpr int NBINS = 4096;
constexpr int FINAL_CAP = 8192;  // candidates the final sort can take (64 KiB of LDS)
constexpr size_t FINAL_LDS_BYTES = (size_t)(FINAL_CAP + FINAL_CAP / 2) * sizeof(uint64_t);  // + the halvings' second buffer
// The second histogram level is taken as soon as the first leaves more than this many candidates: collecting them is
// one global atomic each and the final sort is a one-workgroup bitonic network, so 8192 candidates cost 17 + 30 us
// where the 9 us of a second level leave a few hundred (1.56 M rows / 120 000 images: 99 -> 60 us of selection).
constexpr int LEVEL2_FROM = 1024;
// sampled threshold (launch_select_topk): from this many values on, one 16-element block in SAMPLE_R, the sample's
// rank-th largest value as threshold, rank = max(SAMPLE_RANK / 2, 3 k / SAMPLE_R) (expected candidates
// max(1536 +- 310, 3 k); 3072 +- 450 at most: under the 4096 the final selection handles with four keys a thread), for
// k up to SAMPLE_MAX_K
constexpr int64_t SAMPLE_FROM = (int64_t)1 << 24;
constexpr int64_t SAMPLE_R = 64;
constexpr int SAMPLE_RANK = 48;
constexpr int SAMPLE_MAX_K = 1024;
SSW_TUNABLE bool g_select_sampled = true;  // ssw_tune_topk bit 1
enum StateSlot : int {
    ST_B1 = 0, ST_ABOVE1, ST_CNT1, ST_B2, ST_ABOVE2, ST_CNT2, ST_MODE, ST_NCAND, ST_OVERFLOW, ST_K,
    ST_WORDS = 16
};

__device__ __forceinline__ bool is_excluded(const uint32_t *__restrict__ excl, int64_t i) {
    return excl != nullptr && ((excl[i >> 5] >> (i & 31)) & 1u);
}

// a short difference between two excluded sets, in the kernel-argument segment: ids[0, n_set) gain their bit,
// ids[n_set, n_set + n_clear) lose it
constexpr int EXCL_ARG_IDS = 60;
struct ExclDelta {
    int32_t ids[EXCL_ARG_IDS];
};
__global__ void k_excl_delta_arg(uint32_t *bits, ExclDelta d, int n_set, int n_clear) {
    const int i = threadIdx.x;
    if (i >= n_set + n_clear) return;
    const int32_t id = d.ids[i];
    if (i < n_set)
        atomicOr(&bits[id >> 5], 1u << (id & 31));
    else
        atomicAnd(&bits[id >> 5], ~(1u << (id & 31)));
}
__global__ void k_excl_delta(uint32_t *bits, const int64_t *ids, int64_t n_set, int64_t n_clear) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_set + n_clear) return;
    const int64_t id = ids[i];
    if (i < n_set)
        atomicOr(&bits[id >> 5], 1u << (id & 31));
    else
        atomicAnd(&bits[id >> 5], ~(1u << (id & 31)));
}

__global__ void k_image_max(const float *__restrict__ scores, const int64_t *__restrict__ row_start,
                            int64_t n_images, float *__restrict__ img_score,
                            uint32_t *__restrict__ img_best) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_images) return;
    const int64_t r0 = row_start[m], r1 = row_start[m + 1];
    float best = -INFINITY;
    int64_t br = r0;
    for (int64_t r = r0; r < r1; ++r) {
        const float s = scores[r];
        if (s > best) {  // strict: lowest row wins ties (score_frame2 `.head(n=1)`)
            best = s;
            br = r;
        }
    }
    img_score[m] = best;
    img_best[m] = (uint32_t)br;
}

// LEVEL 1: bins = key >> 20.  LEVEL 2: bins = (key >> 8) & 0xfff for keys with key>>20 == b1.
// sample_r > 1: the histogram of a SAMPLE -- blocks of 16 consecutive elements (one 64-byte segment), one block in
// every sample_r (see launch_select_topk's sampled threshold)
template <int LEVEL>
__global__ __launch_bounds__(256) void k_hist(const float *__restrict__ values, int64_t m,
                                              const uint32_t *__restrict__ excl,
                                              uint32_t *__restrict__ hist,
                                              const uint32_t *__restrict__ state, int64_t sample_r) {
    __shared__ uint32_t lh[NBINS];
    uint32_t b1 = 0;
    if (LEVEL == 2) {
        if (state[ST_MODE] == 0) return;  // level 1 was enough
        b1 = state[ST_B1];
    }
    for (int i = threadIdx.x; i < NBINS; i += 256) lh[i] = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t total = sample_r > 1 ? (m / (16 * sample_r)) * 16 : m;
    auto tally = [&](float v, int64_t i) {
        if (is_excluded(excl, i)) return;
        const uint32_t key = f32_to_ord(v);
        if (LEVEL == 1) {
            atomicAdd(&lh[key >> 20], 1u);
        } else if ((key >> 20) == b1) {
            atomicAdd(&lh[(key >> 8) & 0xfffu], 1u);
        }
    };
    // 16-byte loads, two in flight per thread (the buffers are hipMalloc'ed: 16-byte aligned)
    const float4 *v4 = reinterpret_cast<const float4 *>(values);
    const int64_t total4 = total >> 2;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < total4; j += 2 * stride) {
        const int64_t j2 = j + stride;
        const int64_t q0 = sample_r > 1 ? (((j >> 2) * sample_r) << 2) + (j & 3) : j;
        const int64_t q1 = sample_r > 1 ? (((j2 >> 2) * sample_r) << 2) + (j2 & 3) : j2;
        const float4 a = v4[q0];
        const float4 b = j2 < total4 ? v4[q1] : make_float4(0.f, 0.f, 0.f, 0.f);
        tally(a.x, 4 * q0);
        tally(a.y, 4 * q0 + 1);
        tally(a.z, 4 * q0 + 2);
        tally(a.w, 4 * q0 + 3);
        if (j2 < total4) {
            tally(b.x, 4 * q1);
            tally(b.y, 4 * q1 + 1);
            tally(b.z, 4 * q1 + 2);
            tally(b.w, 4 * q1 + 3);
        }
    }
    if (sample_r <= 1)  // the last m % 4 values
        for (int64_t i = (total4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += stride) tally(values[i], i);
    __syncthreads();
    for (int i = threadIdx.x; i < NBINS; i += 256) {
        const uint32_t c = lh[i];
        if (c) atomicAdd(&hist[i], c);
    }
}

// one block of 1024 threads: find the bin that holds the kk-th largest element.
template <int LEVEL>
__global__ __launch_bounds__(1024) void k_pick(const uint32_t *__restrict__ hist,
                                               uint32_t *__restrict__ state, int k, uint32_t scale) {
    __shared__ uint32_t sums[1024];
    const int t = threadIdx.x;
    uint32_t kk = (uint32_t)k;  // LEVEL 0 (deep path): k is already the rank inside the prefix
    if (LEVEL == 2) {
        if (state[ST_MODE] == 0) return;
        kk = (uint32_t)k - state[ST_ABOVE1];  // rank inside bin b1 (>= 1 by construction)
    }
    const uint32_t h0 = hist[4 * t], h1 = hist[4 * t + 1], h2 = hist[4 * t + 2],
                   h3 = hist[4 * t + 3];
    const uint32_t local = h0 + h1 + h2 + h3;
    sums[t] = local;
    __syncthreads();
    // inclusive suffix sum over threads (from the top bin downwards)
    for (int d = 1; d < 1024; d <<= 1) {
        const uint32_t v = (t + d < 1024) ? sums[t + d] : 0u;
        __syncthreads();
        sums[t] += v;
        __syncthreads();
    }
    const uint32_t total = sums[0];
    const uint32_t above_t = sums[t] - local;  // elements in bins above this thread's 4
    uint32_t b = 0, above = 0, cnt = 0;
    bool found = false;
    if (total < kk) {
        // fewer than k elements in all: everything is a candidate
        if (t == 0) {
            found = true;
            b = 0;
            cnt = h0;
            above = total - h0;
        }
    } else if (above_t < kk && kk <= above_t + local) {
        const uint32_t hh[4] = {h0, h1, h2, h3};
        uint32_t c = above_t;
#pragma unroll
        for (int j = 3; j >= 0; --j) {
            if (!found && c + hh[j] >= kk) {
                found = true;
                b = 4 * t + j;
                above = c;
                cnt = hh[j];
            }
            c += hh[j];
        }
    }
    if (found) {
        if (LEVEL != 2) {
            state[ST_B1] = b;
            state[ST_ABOVE1] = above;
            state[ST_CNT1] = cnt;
            // `scale` = SAMPLE_R when the histogram is that of a sample: its counts stand for scale times as many values
            // (a 12-bit prefix that holds 24 of the sample holds ~1536 of the values, not 24), so the decision is taken on
            // the scaled count -- in sampled mode the second level always runs and the candidates stay near rank x scale
            state[ST_MODE] = ((uint64_t)(above + cnt) * scale > (uint64_t)LEVEL2_FROM) ? 1u : 0u;
        } else {
            state[ST_B2] = b;
            state[ST_ABOVE2] = above;
            state[ST_CNT2] = cnt;
            if ((uint64_t)(state[ST_ABOVE1] + above + cnt) * scale > (uint64_t)FINAL_CAP) state[ST_OVERFLOW] = 1u;
        }
    }
}

__global__ __launch_bounds__(256) void k_collect(const float *__restrict__ values, int64_t m,
                                                 const uint32_t *__restrict__ excl,
                                                 uint32_t *__restrict__ state,
                                                 uint64_t *__restrict__ cand) {
    const uint32_t mode = state[ST_MODE];
    const uint32_t prefix = mode ? ((state[ST_B1] << 12) | state[ST_B2]) : state[ST_B1];
    const int shift = mode ? 8 : 20;
    const int64_t stride = (int64_t)gridDim.x * 256;
    auto take = [&](float v, int64_t i) {
        const uint32_t key = f32_to_ord(v);
        if ((key >> shift) >= prefix && !is_excluded(excl, i)) {
            const uint32_t slot = atomicAdd(&state[ST_NCAND], 1u);
            if (slot < (uint32_t)FINAL_CAP)
                cand[slot] = ((uint64_t)key << 32) | (uint64_t)(0xffffffffu - (uint32_t)i);
        }
    };
    // 16-byte loads, four in flight per thread (the buffers are hipMalloc'ed: 16-byte aligned)
    const float4 *v4 = reinterpret_cast<const float4 *>(values);
    const int64_t m4 = m >> 2;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < m4; j += 4 * stride) {
        float4 a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t ju = j + u * stride;
            a[u] = v4[ju < m4 ? ju : j];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t ju = j + u * stride;
            if (ju < m4) {
                take(a[u].x, 4 * ju);
                take(a[u].y, 4 * ju + 1);
                take(a[u].z, 4 * ju + 2);
                take(a[u].w, 4 * ju + 3);
            }
        }
    }
    for (int64_t i = (m4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += stride) take(values[i], i);
}


// ---- deep path: only when more than FINAL_CAP candidates share a 24-bit score prefix
// (massive exact ties, e.g. duplicated vectors).  Radix select on the full 64-bit
// composite key, one digit per host-synchronised round; composite keys are unique, so
// the loop always ends.
__device__ __forceinline__ uint64_t composite_key(float v, int64_t i) {
    return ((uint64_t)f32_to_ord(v) << 32) | (uint64_t)(0xffffffffu - (uint32_t)i);
}

__global__ __launch_bounds__(256) void k_hist_deep(const float *__restrict__ values, int64_t m,
                                                   const uint32_t *__restrict__ excl,
                                                   uint32_t *__restrict__ hist, uint64_t prefix,
                                                   int prefix_shift, int digit_shift,
                                                   uint32_t digit_mask) {
    __shared__ uint32_t lh[NBINS];
    for (int i = threadIdx.x; i < NBINS; i += 256) lh[i] = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += stride) {
        if (is_excluded(excl, i)) continue;
        const uint64_t ck = composite_key(values[i], i);
        const bool match = (prefix_shift >= 64) || ((ck >> prefix_shift) == prefix);
        if (match) atomicAdd(&lh[(uint32_t)(ck >> digit_shift) & digit_mask], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NBINS; i += 256) {
        const uint32_t c = lh[i];
        if (c) atomicAdd(&hist[i], c);
    }
}

__global__ __launch_bounds__(256) void k_collect_deep(const float *__restrict__ values, int64_t m,
                                                      const uint32_t *__restrict__ excl,
                                                      uint32_t *__restrict__ state,
                                                      uint64_t *__restrict__ cand,
                                                      uint64_t threshold) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += stride) {
        if (is_excluded(excl, i)) continue;
        const uint64_t ck = composite_key(values[i], i);
        if (ck >= threshold) {
            const uint32_t slot = atomicAdd(&state[ST_NCAND], 1u);
            if (slot < (uint32_t)FINAL_CAP) cand[slot] = ck;
        }
    }
}

// ---- the final selection: the c largest of p keys in LDS, sorted descending, 1024 threads ------------------------------
// A full bitonic sort of the candidates in LDS cost a barrier per stage -- 66 stages for 2048 keys, ~30 us of a ~35 us
// kernel for an LVIS-subset index, where the selection is most of a feedback round.  This form:
//   * a compare-exchange at distance j < 64 stays in registers: the partner's key comes through DPP (j = 1, 2, 4, 8) or
//     v_permlane16_swap / v_permlane32_swap (j = 16, 32) -- VALU latency, no LDS queue, no barrier; only j >= 64 goes
//     through LDS;
//   * only the c = max(64, 2^ceil(log2 k)) largest are wanted: blocks of c keys are sorted (alternating direction, the
//     ordinary network up to k2 = c), then pairs of blocks are merged by keeping the element-wise maximum of a descending
//     and an ascending block -- the c largest of both, as a bitonic sequence -- and running the half-cleaners on it,
//     until one block is left.  For k = 60 of 2048 keys: 21 stages over 2048 keys, then 7 over 1024, 512, ... 64,
//     one barrier per halving (the halves ping-pong between two LDS buffers).
// Keys are unique (score, position) apart from the 0 padding, so the order of equal keys does not matter.

template <int J>
__device__ __forceinline__ uint32_t lane_xor_u32(uint32_t x, int lane) {
    if constexpr (J == 1) {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
    } else if constexpr (J == 2) {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
    } else if constexpr (J == 4) {
        // lanes 0-3, 8-11 of a row read lane + 4 (row_shl:4, banks 0 and 2), the others lane - 4 (row_shr:4)
        int r = __builtin_amdgcn_update_dpp((int)x, (int)x, 0x104, 0xF, 0x5, false);
        r = __builtin_amdgcn_update_dpp(r, (int)x, 0x114, 0xF, 0xA, false);
        return (uint32_t)r;
    } else if constexpr (J == 8) {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x128, 0xF, 0xF, true);  // row_ror:8
    } else if constexpr (J == 16) {
        // (a, b) = (x, x): the swap exchanges a's odd rows with b's even rows -> a = [r0 r0 r2 r2], b = [r1 r1 r3 r3]
        const auto ab = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        return (lane & 16) ? ab[0] : ab[1];
    } else {
        // a's upper half <-> b's lower half -> a = [lo lo], b = [hi hi]
        const auto ab = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        return (lane & 32) ? ab[0] : ab[1];
    }
}

// one compare-exchange stage at lane distance J of the merge with direction bit k2, key of element index i
template <int J>
__device__ __forceinline__ uint64_t lane_stage(uint64_t v, int i, int k2, int lane) {
    const uint32_t lo = lane_xor_u32<J>((uint32_t)v, lane), hi = lane_xor_u32<J>((uint32_t)(v >> 32), lane);
    const uint64_t o = ((uint64_t)hi << 32) | lo;
    const bool lower = (lane & J) == 0, desc = (i & k2) == 0;
    return ((lower == desc) == (o > v)) ? o : v;  // the larger key to the descending side's lower lane
}

// stages j = min(jstart, 32) .. 1 on NE register-resident keys, key e being element e * 1024 + t (live when < pcur)
template <int NE>
__device__ __forceinline__ void lane_stages(uint64_t (&v)[NE], int pcur, int k2, int jstart, int t) {
    const int lane = t & 63;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int i = e * 1024 + t;
        if (e * 1024 < pcur && i < pcur) {  // uniform per wave: pcur is a multiple of 64
            if (jstart >= 32) v[e] = lane_stage<32>(v[e], i, k2, lane);
            if (jstart >= 16) v[e] = lane_stage<16>(v[e], i, k2, lane);
            if (jstart >= 8) v[e] = lane_stage<8>(v[e], i, k2, lane);
            if (jstart >= 4) v[e] = lane_stage<4>(v[e], i, k2, lane);
            if (jstart >= 2) v[e] = lane_stage<2>(v[e], i, k2, lane);
            v[e] = lane_stage<1>(v[e], i, k2, lane);
        }
    }
}

// stages j = jstart .. 1 of the merge whose direction bit is k2, over pcur keys (a multiple of 64) in LDS
__device__ __forceinline__ void bitonic_stages(uint64_t *s, int pcur, int k2, int jstart, int t) {
    int j = jstart;
    for (; j >= 64; j >>= 1) {
        for (int pr = t; pr < (pcur >> 1); pr += 1024) {
            const int i = ((pr & ~(j - 1)) << 1) | (pr & (j - 1));  // pair number -> lower index (a 0 at bit log2 j)
            const int ixj = i | j;
            const bool desc = ((i & k2) == 0);
            const uint64_t a = s[i], b = s[ixj];
            if ((a < b) == desc) {
                s[i] = b;
                s[ixj] = a;
            }
        }
        __syncthreads();
    }
    uint64_t v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (e * 1024 < pcur && e * 1024 + t < pcur) v[e] = s[e * 1024 + t];
    lane_stages<8>(v, pcur, k2, j, t);
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (e * 1024 < pcur && e * 1024 + t < pcur) s[e * 1024 + t] = v[e];
    __syncthreads();
}

// s: p keys (p a power of two in [64, FINAL_CAP]) followed by a second buffer of FINAL_CAP / 2 keys at s + FINAL_CAP;
// c: power of two in [64, p].  Returns where the c largest keys lie, descending (s or the second buffer).
__device__ __forceinline__ uint64_t *select_sorted_desc(uint64_t *s, int p, int c, int t) {
    {   // k2 = 2 .. 64 without leaving the registers
        uint64_t v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (e * 1024 < p && e * 1024 + t < p) v[e] = s[e * 1024 + t];
        lane_stages<8>(v, p, 2, 1, t);
        lane_stages<8>(v, p, 4, 2, t);
        lane_stages<8>(v, p, 8, 4, t);
        lane_stages<8>(v, p, 16, 8, t);
        lane_stages<8>(v, p, 32, 16, t);
        lane_stages<8>(v, p, 64, 32, t);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (e * 1024 < p && e * 1024 + t < p) s[e * 1024 + t] = v[e];
        __syncthreads();
    }
    for (int k2 = 128; k2 <= c; k2 <<= 1) bitonic_stages(s, p, k2, k2 >> 1, t);
    const int cshift = __builtin_ctz(c);
    uint64_t *src = s, *dst = s + FINAL_CAP;
    for (int pcur = p; pcur > c;) {
        const int half = pcur >> 1;
        uint64_t r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int o = e * 1024 + t;
            if (e * 1024 < half && o < half) {
                const int m = o >> cshift, i = o & (c - 1);
                const uint64_t a = src[((2 * m) << cshift) + i], b = src[((2 * m + 1) << cshift) + i];
                r[e] = a > b ? a : b;
            }
        }
        pcur = half;
        if (c == 64) {  // the half-cleaners are lane stages: straight on the registers, one barrier per halving
            lane_stages<4>(r, pcur, c, 32, t);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e * 1024 < half && e * 1024 + t < half) dst[e * 1024 + t] = r[e];
            __syncthreads();
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e * 1024 < half && e * 1024 + t < half) dst[e * 1024 + t] = r[e];
            __syncthreads();
            bitonic_stages(dst, pcur, c, c >> 1, t);
        }
        uint64_t *tmp = src;
        src = dst;
        dst = tmp;
    }
    return src;
}

// candidates (n_lists lists of counts[l] keys at keys_in + l*stride; or one list whose
// length is *ncand_ptr) -> k largest, sorted descending.
// `x` (sharded exchange, no extra launch on either side of the all-gather):
//   selection: msg_out = this rank's message -- keys made global (key - image_offset: the low word is
//     0xFFFFFFFF - id) at [0, k), best rows + row_offset at [k_max, k_max + k) when with_best, and the word
//     count | overflow << 32 at [msg_len - 1];
//   merge (from_msgs): the lists ARE such messages (stride msg_len): a list's count is the low word of its last entry,
//     its overflow flag goes to flags_out[l] and is OR-ed into flags_seen.
__global__ __launch_bounds__(1024) void k_final(const uint64_t *__restrict__ keys_in, int n_lists,
                                                int list_stride, const int32_t *__restrict__ counts,
                                                const uint32_t *__restrict__ ncand_ptr,
                                                const uint32_t *__restrict__ state_or_null, int k,
                                                const uint32_t *__restrict__ best_src,
                                                uint64_t *__restrict__ keys_out,
                                                int32_t *__restrict__ count_out,
                                                uint32_t *__restrict__ best_out,
                                                unsigned char *__restrict__ packed_or_null, FinalExchange x) {
    extern __shared__ uint64_t s[];  // FINAL_CAP entries + FINAL_CAP / 2 for the halvings
    const int t = threadIdx.x;
    int n = 0;
    if (x.values_all != nullptr) {
        // few images (m <= FINAL_CAP: an LVIS-category subset has ~1 100): no histograms, no candidate pass -- every
        // image's composite key goes straight into the selection, excluded ones as 0 (below every real key)
        n = (int)x.m_all;
        // the id list lives in host memory: ask for it first, use it once the keys are in LDS
        const int64_t ex0 = (t < x.n_excl) ? x.excl_ids[t] : -1;
        for (int i = t; i < n; i += 1024) {
            float v;
            if (x.row_start != nullptr) {  // k_image_max, here; eight rows in flight
                const int64_t r0 = x.row_start[i], r1 = x.row_start[i + 1];
                float best = -INFINITY;
                int64_t br = r0;
                for (int64_t r = r0; r < r1; r += 8) {
                    float sc[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) sc[u] = x.values_all[min(r + u, r1 - 1)];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (r + u < r1 && sc[u] > best) {  // strict: lowest row wins ties
                            best = sc[u];
                            br = r + u;
                        }
                    }
                }
                x.img_score[i] = best;
                x.img_best[i] = (uint32_t)br;
                v = best;
            } else {
                v = x.values_all[i];
            }
            s[i] = is_excluded(x.excl, i) ? 0ull : composite_key(v, i);
        }
        __syncthreads();
        if (ex0 >= 0) s[ex0] = 0ull;  // ids range-checked by the host
        for (int64_t j = t + 1024; j < x.n_excl; j += 1024) s[x.excl_ids[j]] = 0ull;
    } else if (ncand_ptr != nullptr) {
        n = (int)min(*ncand_ptr, (uint32_t)FINAL_CAP);
        for (int i = t; i < n; i += 1024) s[i] = keys_in[i];
    } else {
        for (int l = 0; l < n_lists; ++l) {
            // lists arrive sorted (descending): only a list's first k keys can reach the global top-k
            const int have = x.from_msgs ? (int)(keys_in[(int64_t)l * list_stride + list_stride - 1] & 0xffffffffull)
                                         : counts[l];
            const int c = max(0, min(min(have, x.from_msgs ? x.k_max : list_stride), k));
            if (n + c > FINAL_CAP) break;  // launcher guarantees n_lists * min(stride, k) <= FINAL_CAP
            for (int i = t; i < c; i += 1024) s[n + i] = keys_in[(int64_t)l * list_stride + i];
            n += c;
        }
    }
    int p = 64;
    while (p < n) p <<= 1;
    int c = 64;  // keys wanted, rounded up to a block of the selection
    while (c < k) c <<= 1;
    c = min(c, p);
    for (int i = n + t; i < p; i += 1024) s[i] = 0ull;  // below every real key
    __syncthreads();
    const uint64_t *res = select_sorted_desc(s, p, c, t);
    // res[0, c): descending, real keys (never 0) first.  The thread that sees the end of the real keys among the first
    // min(k, c) reports the count; every entry before it goes out.
    const int kk = min(k, c);
    // packed mirror for the host: [count, overflow, k, 0][keys x k][best x k], one D2H copy
    uint64_t *pk_keys = packed_or_null ? reinterpret_cast<uint64_t *>(packed_or_null + 16) : nullptr;
    uint32_t *pk_best = packed_or_null ? reinterpret_cast<uint32_t *>(packed_or_null + 16 + (size_t)k * 8) : nullptr;
    for (int i = t; i < kk; i += 1024) {
        const uint64_t key = res[i];
        const bool last = key != 0ull && (i + 1 == kk || res[i + 1] == 0ull);
        if (last || (i == 0 && key == 0ull)) {
            const int out = key != 0ull ? i + 1 : 0;
            // overflow of the fast path (more candidates than the final sort takes), for the host
            // ... or of the sampled threshold: fewer candidates than asked for although the threshold left some out
            int ovf = (state_or_null && (state_or_null[ST_OVERFLOW] != 0 ||
                                         state_or_null[ST_NCAND] > (uint32_t)FINAL_CAP)) ? 1 : 0;
            if (x.sampled && state_or_null && state_or_null[ST_NCAND] < (uint32_t)k &&
                (state_or_null[ST_MODE] != 0 || state_or_null[ST_B1] != 0))
                ovf = 1;
            count_out[0] = out;
            if (state_or_null || x.values_all) count_out[1] = ovf;
            if (x.msg_out) x.msg_out[x.msg_len - 1] = (uint64_t)(uint32_t)out | ((uint64_t)(uint32_t)ovf << 32);
            if (packed_or_null) {
                int32_t *hdr = reinterpret_cast<int32_t *>(packed_or_null);
                hdr[0] = out;
                hdr[1] = ovf;
                hdr[2] = k;
                if (x.host_seq == 0) hdr[3] = 0;
            }
        }
        if (key == 0ull) continue;
        keys_out[i] = key;
        if (pk_keys) pk_keys[i] = key;
        uint32_t br = 0;
        if (best_out != nullptr) {
            const uint32_t id = 0xffffffffu - (uint32_t)(key & 0xffffffffull);
            br = best_src ? best_src[id] : id;
            best_out[i] = br;
            if (pk_best) pk_best[i] = br;
        }
        if (x.msg_out) {
            x.msg_out[i] = key - x.image_offset;
            if (x.with_best) x.msg_out[x.k_max + i] = (uint64_t)((int64_t)br + x.row_offset);
        }
    }
    if (t == 0 && x.from_msgs) {
        long long any = 0;
        for (int l = 0; l < n_lists; ++l) {
            const long long f = (long long)(keys_in[(int64_t)l * list_stride + list_stride - 1] >> 32);
            if (x.flags_out) x.flags_out[l] = f;
            any |= f;
        }
        if (x.flags_seen) *x.flags_seen |= any;
    }
    if (x.host_seq != 0) {  // the packed block is host memory: release everything, then the word the host spins on
        __threadfence_system();
        __syncthreads();
        if (t == 0)
            __hip_atomic_store(reinterpret_cast<unsigned *>(packed_or_null) + 3, x.host_seq, __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ void k_gather_f32(const float *__restrict__ src, const int64_t *__restrict__ idx,
                             int64_t n, float *__restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

template <typename T>
ssw_status dev_alloc(T **p, size_t count) {
    SSW_HIP_TRY(hipMalloc((void **)p, count * sizeof(T) + 16));
    return SSW_OK;
}

int grid_for(int64_t m, int device) {
    // >= 4096 elements per workgroup: zeroing / flushing the 4096-bin LDS histogram is the
    // fixed cost of a block (at 1 M images 2048 blocks spent most of k_hist on it)
    int64_t g = (m + 4095) / 4096;
    const int64_t cap = (int64_t)num_cus(device) * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

#ifdef SSW_DEBUG_HOOKS
void tune_select(bool sampled) { g_select_sampled = sampled; }
#endif

ssw_status select_alloc(SelectWorkspace &ws, int64_t n_rows, int64_t n_images, bool has_map) {
    SSW_TRY(dev_alloc(&ws.hist1, 2 * NBINS + ST_WORDS));
    ws.hist2 = ws.hist1 + NBINS;
    ws.state = ws.hist1 + 2 * NBINS;
    SSW_TRY(dev_alloc(&ws.cand, FINAL_CAP));
    SSW_TRY(dev_alloc(&ws.out_keys, SSW_MAX_TOPK));
    SSW_TRY(dev_alloc(&ws.out_count, 4));
    SSW_TRY(dev_alloc(&ws.out_best, SSW_MAX_TOPK));
    SSW_TRY(dev_alloc(&ws.packed, 16 + (size_t)SSW_MAX_TOPK * 12));
    if (has_map) {
        SSW_TRY(dev_alloc(&ws.img_score, n_images));
        SSW_TRY(dev_alloc(&ws.img_best, n_images));
    }
    const int64_t words = (n_images + 31) / 32;
    SSW_TRY(dev_alloc(&ws.excl_bits, words));
    SSW_HIP_TRY(hipMemset(ws.excl_bits, 0, words * 4));
    SSW_HIP_TRY(hipMemset(ws.out_count, 0, 16));
    (void)n_rows;
    return SSW_OK;
}

void select_free(SelectWorkspace &ws) {
    (void)hipFree(ws.hist1);
    (void)hipFree(ws.cand);
    (void)hipFree(ws.out_keys);
    (void)hipFree(ws.out_count);
    (void)hipFree(ws.out_best);
    (void)hipFree(ws.packed);
    (void)hipFree(ws.img_score);
    (void)hipFree(ws.img_best);
    (void)hipFree(ws.excl_bits);
    (void)hipFree(ws.excl_ids);
    ws.excl_stage.release();
    ws = SelectWorkspace();
}

ssw_status select_set_excluded(SelectWorkspace &ws, int64_t n_images, const int64_t *ids_host,
                               int64_t n, hipStream_t stream) {
    // A session's list grows by a batch per round (InteractiveQuery.returned): what reaches the device is the
    // difference from the installed set -- a few ids in the kernel-argument segment, no copy -- not the whole list.
    static thread_local std::vector<int64_t> neu, delta;
    neu.assign(ids_host, ids_host + (n > 0 ? n : 0));
    if (!std::is_sorted(neu.begin(), neu.end())) std::sort(neu.begin(), neu.end());
    neu.erase(std::unique(neu.begin(), neu.end()), neu.end());
    const std::vector<int64_t> &cur = ws.excl_installed;
    delta.clear();
    std::set_difference(neu.begin(), neu.end(), cur.begin(), cur.end(), std::back_inserter(delta));
    const int64_t n_set = (int64_t)delta.size();
    std::set_difference(cur.begin(), cur.end(), neu.begin(), neu.end(), std::back_inserter(delta));
    const int64_t n_clear = (int64_t)delta.size() - n_set;
    if (n_set + n_clear > 0) {
        if (n_set + n_clear <= EXCL_ARG_IDS) {
            ExclDelta d;
            memset(&d, 0, sizeof(d));
            for (int64_t i = 0; i < n_set + n_clear; ++i) d.ids[i] = (int32_t)delta[(size_t)i];
            hipLaunchKernelGGL(k_excl_delta_arg, dim3(1), dim3(64), 0, stream, ws.excl_bits, d, (int)n_set, (int)n_clear);
        } else {
            const int64_t m = n_set + n_clear;
            if (m > ws.excl_ids_cap) {
                SSW_HIP_TRY(hipStreamSynchronize(stream));  // a kernel may still be reading the old buffer
                (void)hipFree(ws.excl_ids);
                ws.excl_ids = nullptr;
                ws.excl_ids_cap = 0;
                int64_t cap = 1024;
                while (cap < m) cap <<= 1;
                SSW_TRY(dev_alloc(&ws.excl_ids, cap));
                ws.excl_ids_cap = cap;
            }
            SSW_TRY(ws.excl_stage.push(ws.excl_ids, delta.data(), (size_t)m * sizeof(int64_t), stream));
            hipLaunchKernelGGL(k_excl_delta, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream, ws.excl_bits,
                               ws.excl_ids, n_set, n_clear);
        }
        SSW_HIP_TRY(hipGetLastError());
    }
    ws.excl_installed.swap(neu);
    ws.excl_dirty = !ws.excl_installed.empty();
    ws.n_excluded_distinct = (int64_t)ws.excl_installed.size();
    (void)n_images;  // the ids were range-checked by the caller
    return SSW_OK;
}

ssw_status launch_image_max(const float *scores, const int64_t *row_start, int64_t n_images,
                            float *img_score, uint32_t *img_best, hipStream_t stream) {
    if (n_images <= 0) return SSW_OK;
    hipLaunchKernelGGL(k_image_max, dim3((unsigned)((n_images + 255) / 256)), dim3(256), 0, stream,
                       scores, row_start, n_images, img_score, img_best);
    SSW_HIP_TRY(hipGetLastError());
    return SSW_OK;
}

// k_final's LDS (96 KiB) is above the 64 KiB a kernel gets by default: raise the limit once per device
static ssw_status final_lds_ready() {
    static bool done[64] = {false};
    int dev = 0;
    SSW_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !done[dev]) {
        SSW_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_final), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)FINAL_LDS_BYTES));
        if (dev >= 0 && dev < 64) done[dev] = true;
    }
    return SSW_OK;
}

// With an exchange target attached (ssw_index_set_exchange_target) k_final writes k keys -- and k best rows from slot
// k_max on -- into the rank's message: a k beyond k_max would run over the count word and past the send buffer.
static ssw_status exchange_fits(const SelectWorkspace &ws, int32_t k) {
    if (ws.xchg.msg_out != nullptr && k > ws.xchg.k_max) {
        set_error("topk: k=%d exceeds the exchange target's k_max=%d", k, ws.xchg.k_max);
        return SSW_ERR_INVALID;
    }
    return SSW_OK;
}

ssw_status launch_select_small(SelectWorkspace &ws, const float *row_scores, const int64_t *row_start_or_null,
                               int64_t n_images, const int64_t *excl_ids_mapped, int64_t n_excl, int32_t k,
                               unsigned char *packed_mapped, unsigned seq, hipStream_t stream) {
    if (n_images < 1 || n_images > SELECT_SMALL_IMAGES || n_images > FINAL_CAP || k < 1 || k > SSW_MAX_TOPK || seq == 0) {
        set_error("select_small: n_images=%lld k=%d outside the one-launch path", (long long)n_images, k);
        return SSW_ERR_INVALID;
    }
    SSW_TRY(exchange_fits(ws, k));
    FinalExchange x = ws.xchg;
    x.values_all = row_scores;
    x.m_all = n_images;
    x.row_start = row_start_or_null;
    x.img_score = ws.img_score;
    x.img_best = ws.img_best;
    x.excl_ids = excl_ids_mapped;
    x.n_excl = n_excl;
    x.host_seq = seq;
    SSW_TRY(final_lds_ready());
    hipLaunchKernelGGL(k_final, dim3(1), dim3(1024), FINAL_LDS_BYTES, stream, (const uint64_t *)nullptr, 0, 0,
                       (const int32_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr, (int)k,
                       row_start_or_null ? (const uint32_t *)ws.img_best : (const uint32_t *)nullptr, ws.out_keys,
                       ws.out_count, ws.out_best, packed_mapped, x);
    SSW_HIP_TRY(hipGetLastError());
    return SSW_OK;
}

ssw_status launch_select_topk(SelectWorkspace &ws, const float *values, int64_t m,
                              const uint32_t *best_rows_or_null, int32_t k, int device,
                              hipStream_t stream) {
    if (k < 1 || k > SSW_MAX_TOPK) {
        set_error("topk: k=%d outside [1, %d]", k, SSW_MAX_TOPK);
        return SSW_ERR_INVALID;
    }
    if (m >= (int64_t)0xffffffffll) {
        set_error("topk: %lld images exceed the 32-bit id space of one shard", (long long)m);
        return SSW_ERR_UNSUPPORTED;
    }
    SSW_TRY(exchange_fits(ws, k));
    const uint32_t *excl = ws.excl_dirty ? ws.excl_bits : nullptr;
    unsigned char *packed = ws.host_packed ? ws.host_packed : ws.packed;
    FinalExchange xg = ws.xchg;
    xg.host_seq = ws.host_packed ? ws.host_seq : 0u;
    ws.host_packed = nullptr;
    if (m <= FINAL_CAP) {  // one launch: the sort takes every image
        FinalExchange x = xg;
        x.values_all = values;
        x.m_all = m;
        x.excl = excl;
        SSW_TRY(final_lds_ready());
        hipLaunchKernelGGL(k_final, dim3(1), dim3(1024), FINAL_LDS_BYTES, stream, (const uint64_t *)nullptr,
                           0, 0, (const int32_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr, (int)k,
                           best_rows_or_null, ws.out_keys, ws.out_count, ws.out_best, packed, x);
        SSW_HIP_TRY(hipGetLastError());
        return SSW_OK;
    }
    SSW_HIP_TRY(hipMemsetAsync(ws.hist1, 0, (2 * NBINS + ST_WORDS) * sizeof(uint32_t), stream));
    const int g = grid_for(m, device);
    // Large m: the two histogram passes over all m values (147 us at 100 M) only serve to find a threshold that leaves
    // a few thousand candidates.  A 1-in-64 sample finds one as well: the 24-bit prefix of the sample's 48th largest
    // value leaves ~3072 +- 450 of the m (blocks of 16 neighbours: wider for clustered scores), always every value at
    // or above it -- so the result is exact whenever at least k candidates came out; if fewer did (or more than the
    // final sort takes) the overflow word is raised and the caller reruns the deep path, as for mass ties.
    const bool sampled = g_select_sampled && m >= SAMPLE_FROM && k <= SAMPLE_MAX_K;
    const int64_t sr = sampled ? SAMPLE_R : 1;
    // expected candidates max(1536, 3 k): rank 24 ... 48 of the sample
    const int rank = sampled ? std::max(SAMPLE_RANK / 2, (int)((3 * (int64_t)k + SAMPLE_R - 1) / SAMPLE_R)) : (int)k;
    const int gs = sampled ? grid_for(m / sr, device) : g;
    xg.sampled = sampled ? 1 : 0;
    hipLaunchKernelGGL(k_hist<1>, dim3(gs), dim3(256), 0, stream, values, m, excl, ws.hist1, ws.state, sr);
    hipLaunchKernelGGL(k_pick<1>, dim3(1), dim3(1024), 0, stream, ws.hist1, ws.state, rank, (uint32_t)sr);
    hipLaunchKernelGGL(k_hist<2>, dim3(gs), dim3(256), 0, stream, values, m, excl, ws.hist2, ws.state, sr);
    hipLaunchKernelGGL(k_pick<2>, dim3(1), dim3(1024), 0, stream, ws.hist2, ws.state, rank, (uint32_t)sr);
    hipLaunchKernelGGL(k_collect, dim3(g), dim3(256), 0, stream, values, m, excl, ws.state, ws.cand);
    SSW_TRY(final_lds_ready());
    hipLaunchKernelGGL(k_final, dim3(1), dim3(1024), FINAL_LDS_BYTES, stream, ws.cand,
                       0, 0, (const int32_t *)nullptr, ws.state + ST_NCAND, (const uint32_t *)ws.state, (int)k,
                       best_rows_or_null, ws.out_keys, ws.out_count, ws.out_best, packed, xg);
    SSW_HIP_TRY(hipGetLastError());
    return SSW_OK;
}


ssw_status launch_select_topk_deep(SelectWorkspace &ws, const float *values, int64_t m,
                                   const uint32_t *best_rows_or_null, int32_t k, int device,
                                   hipStream_t stream) {
    if (k < 1 || k > SSW_MAX_TOPK) {
        set_error("topk: k=%d outside [1, %d]", k, SSW_MAX_TOPK);
        return SSW_ERR_INVALID;
    }
    SSW_TRY(exchange_fits(ws, k));
    const uint32_t *excl = ws.excl_dirty ? ws.excl_bits : nullptr;
    const int g = grid_for(m, device);
    static const int digit_shift[6] = {52, 40, 32, 20, 8, 0};
    static const int digit_bits[6] = {12, 12, 8, 12, 12, 8};
    uint64_t prefix = 0;
    int prefix_shift = 64;
    uint32_t kk = (uint32_t)k;
    uint64_t threshold = 0;
    for (int lvl = 0; lvl < 6; ++lvl) {
        SSW_HIP_TRY(hipMemsetAsync(ws.hist1, 0, (2 * NBINS + ST_WORDS) * sizeof(uint32_t), stream));
        hipLaunchKernelGGL(k_hist_deep, dim3(g), dim3(256), 0, stream, values, m, excl, ws.hist1,
                           prefix, prefix_shift, digit_shift[lvl],
                           (uint32_t)((1u << digit_bits[lvl]) - 1u));
        hipLaunchKernelGGL(k_pick<0>, dim3(1), dim3(1024), 0, stream, ws.hist1, ws.state, (int)kk, 1u);
        SSW_HIP_TRY(hipGetLastError());
        uint32_t st[ST_WORDS];
        SSW_HIP_TRY(hipMemcpyAsync(st, ws.state, sizeof(st), hipMemcpyDeviceToHost, stream));
        SSW_HIP_TRY(hipStreamSynchronize(stream));
        const uint32_t b = st[ST_B1], above = st[ST_ABOVE1], cnt = st[ST_CNT1];
        prefix = (prefix << digit_bits[lvl]) | b;
        prefix_shift = digit_shift[lvl];
        threshold = prefix << prefix_shift;
        // everything above the prefix is fewer than k <= SSW_MAX_TOPK elements in total
        const uint32_t above_total = (uint32_t)k - kk + above;
        if (above_total + cnt <= (uint32_t)FINAL_CAP) break;
        kk -= above;
    }
    SSW_HIP_TRY(hipMemsetAsync(ws.state, 0, ST_WORDS * sizeof(uint32_t), stream));
    hipLaunchKernelGGL(k_collect_deep, dim3(g), dim3(256), 0, stream, values, m, excl, ws.state,
                       ws.cand, threshold);
    SSW_TRY(final_lds_ready());
    unsigned char *packed = ws.host_packed ? ws.host_packed : ws.packed;
    FinalExchange xg = ws.xchg;
    xg.host_seq = ws.host_packed ? ws.host_seq : 0u;
    ws.host_packed = nullptr;
    hipLaunchKernelGGL(k_final, dim3(1), dim3(1024), FINAL_LDS_BYTES, stream, ws.cand,
                       0, 0, (const int32_t *)nullptr, ws.state + ST_NCAND, (const uint32_t *)ws.state, (int)k,
                       best_rows_or_null, ws.out_keys, ws.out_count, ws.out_best, packed, xg);
    SSW_HIP_TRY(hipGetLastError());
    return SSW_OK;
}

ssw_status launch_merge_topk(const uint64_t *keys_in, int32_t n_lists, int32_t list_stride,
                             const int32_t *counts, int32_t k, uint64_t *keys_out,
                             int32_t *count_out, hipStream_t stream) {
    if (k < 1 || k > SSW_MAX_TOPK) {
        set_error("merge: k=%d outside [1, %d]", k, SSW_MAX_TOPK);
        return SSW_ERR_INVALID;
    }
    // every list contributes its first min(stride, k) keys (lists are sorted): 8 shards x k = 1024 just fit
    if (n_lists < 1 || list_stride < 1 || (int64_t)n_lists * std::min(list_stride, k) > FINAL_CAP) {
        set_error("merge: %d lists x %d keys exceed %d candidates", n_lists, std::min(list_stride, k), FINAL_CAP);
        return SSW_ERR_INVALID;
    }
    SSW_TRY(final_lds_ready());
    hipLaunchKernelGGL(k_final, dim3(1), dim3(1024), FINAL_LDS_BYTES, stream, keys_in,
                       (int)n_lists, (int)list_stride, counts, (const uint32_t *)nullptr,
                       (const uint32_t *)nullptr, (int)k, (const uint32_t *)nullptr, keys_out, count_out,
                       (uint32_t *)nullptr, (unsigned char *)nullptr, FinalExchange());
    SSW_HIP_TRY(hipGetLastError());
    return SSW_OK;
}

// the merge side of the sharded exchange: the gathered messages themselves are the lists
ssw_status launch_merge_msgs(const uint64_t *msgs, int32_t world, int32_t k_max, int32_t with_best, int32_t k,
                             uint64_t *keys_out, int32_t *count_out, long long *flags_out, long long *flags_seen,
                             hipStream_t stream) {
    if (k < 1 || k > SSW_MAX_TOPK || k_max < k) {
        set_error("merge: k=%d outside [1, min(%d, k_max = %d)]", k, SSW_MAX_TOPK, k_max);
        return SSW_ERR_INVALID;
    }
    if (world < 1 || (int64_t)world * k > FINAL_CAP) {
        set_error("merge: %d lists x %d keys exceed %d candidates", world, k, FINAL_CAP);
        return SSW_ERR_INVALID;
    }
    FinalExchange x;
    x.from_msgs = 1;
    x.k_max = k_max;
    x.with_best = with_best;
    x.msg_len = (with_best ? 2 : 1) * k_max + 1;
    x.flags_out = flags_out;
    x.flags_seen = flags_seen;
    SSW_TRY(final_lds_ready());
    hipLaunchKernelGGL(k_final, dim3(1), dim3(1024), FINAL_LDS_BYTES, stream, msgs, (int)world,
                       (int)x.msg_len, (const int32_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr,
                       (int)k, (const uint32_t *)nullptr, keys_out, count_out, (uint32_t *)nullptr,
                       (unsigned char *)nullptr, x);
    SSW_HIP_TRY(hipGetLastError());
    return SSW_OK;
}

ssw_status launch_gather_f32(const float *src, const int64_t *idx_dev, int64_t n, float *dst,
                             hipStream_t stream) {
    if (n <= 0) return SSW_OK;
    hipLaunchKernelGGL(k_gather_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src,
                       idx_dev, n, dst);
    SSW_HIP_TRY(hipGetLastError());
    return SSW_OK;
}

}  // namespace ssw
