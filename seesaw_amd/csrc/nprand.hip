// nprand.hip -- host-only: numpy's legacy `np.random.permutation(n)[:k]`, on numpy's own MT19937 state.
//
// PseudoLR draws its pseudo-labelled sample with `np.random.permutation(n_unlabelled)[:sample_size]`
// (seesaw/loops/util.py:11-14): which rows are drawn decides the fit, so a drop-in has to reproduce the draw, i.e.
// consume the global RandomState's stream exactly as numpy does.  numpy shuffles np.arange(n) with a generic
// memcpy-swap per element (15 ms for the 1.56 M unlabelled rows of the LVIS-scale bench, most of a PseudoLR round);
// this is the same algorithm, with the draws made word-wise (draw_targets) and only the k wanted positions followed
// through the swaps (PrefixTrace):
//     for i = n-1 ... 1:  j = random_interval(i);  swap(a[i], a[j])          (mtrand.pyx _shuffle_raw)
//     random_interval(max): mask = 2^ceil(log2(max+1)) - 1; draw 32-bit words until (word & mask) <= max
//                                                                          (distributions.c random_interval)
//     word = MT19937 genrand with numpy's (randomkit's) reload and tempering    (mt19937.c)
// The caller passes the 624-word key and position of np.random.get_state() and writes them back with set_state():
// the stream continues exactly where numpy's own call would have left it (tests/test_nprand_cpu.py).
#include <algorithm>
#include <cstdint>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifndef __HIP_DEVICE_COMPILE__
#include <immintrin.h>  // host pass only: the AVX-512 variants below
#endif

#include "ssw_common.h"

namespace {

constexpr int MT_N = 624, MT_M = 397;
constexpr uint32_t MT_MATRIX_A = 0x9908b0dfu, MT_UPPER = 0x80000000u, MT_LOWER = 0x7fffffffu;

inline void mt_reload(uint32_t *key) {
    int i = 0;
    for (; i < MT_N - MT_M; ++i) {
        const uint32_t y = (key[i] & MT_UPPER) | (key[i + 1] & MT_LOWER);
        key[i] = key[i + MT_M] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1u)) & MT_MATRIX_A);
    }
    for (; i < MT_N - 1; ++i) {
        const uint32_t y = (key[i] & MT_UPPER) | (key[i + 1] & MT_LOWER);
        key[i] = key[i + (MT_M - MT_N)] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1u)) & MT_MATRIX_A);
    }
    const uint32_t y = (key[MT_N - 1] & MT_UPPER) | (key[0] & MT_LOWER);
    key[MT_N - 1] = key[MT_M - 1] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1u)) & MT_MATRIX_A);
}

struct Mt {
    uint32_t *key;
    int pos;
    uint32_t out[MT_N];  // tempered words of the current block (filled from `pos` on)
    inline void temper_from(int from) {
        for (int i = from; i < MT_N; ++i) {
            uint32_t y = key[i];
            y ^= y >> 11;
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= y >> 18;
            out[i] = y;
        }
    }
    inline uint32_t next32() {
        if (pos == MT_N) {
            mt_reload(key);
            pos = 0;
            temper_from(0);
        }
        return out[pos++];
    }
    inline uint64_t next64() {  // mt19937_next64: high word first
        const uint64_t hi = next32();
        return (hi << 32) | next32();
    }
    // numpy's random_interval for max <= 0xffffffff: words are drawn until (word & mask) <= max.  Two candidates are
    // looked at per trip and the choice made without a branch (the first one is rejected 0-50 % of the time, which a
    // predictor cannot learn); the stream position advances by exactly the number of words numpy would have drawn.
    inline uint32_t interval32(uint32_t max, uint32_t mask) {
        for (;;) {
            if (pos + 2 <= MT_N) {
                const uint32_t v1 = out[pos] & mask, v2 = out[pos + 1] & mask;
                const bool ok1 = v1 <= max;
                const uint32_t v = ok1 ? v1 : v2;
                pos += ok1 ? 1 : 2;
                if (ok1 || v2 <= max) return v;
            } else {
                const uint32_t v = next32() & mask;
                if (v <= max) return v;
            }
        }
    }
    inline uint64_t interval(uint64_t max) {
        if (max == 0) return 0;
        uint64_t mask = max;
        mask |= mask >> 1;
        mask |= mask >> 2;
        mask |= mask >> 4;
        mask |= mask >> 8;
        mask |= mask >> 16;
        mask |= mask >> 32;
        if (max <= 0xffffffffull) return interval32((uint32_t)max, (uint32_t)mask);
        uint64_t v;
        while ((v = (next64() & mask)) > max) {
        }
        return v;
    }
};

// Draw targets j_i = random_interval(i) for i = n-1 ... 1, written in draw order: Jd[t] = j_(n-1-t).  The stream is
// walked word by word, not draw by draw: every word is consumed whether it is accepted or not, and it is accepted iff
// (word & mask) <= i for the i current at that word -- so the only loop-carried dependency is `i -= accepted` (two
// cycles), instead of the load -> compare -> advance chain of a draw-by-draw loop (14 cycles a draw, 10 ms of the
// former 14).  All i of one bit length share a mask, so the loop runs per bit length; a rejected word's value is stored
// too and overwritten by the next one.  Jd has 16 entries of slack past n.
void draw_targets(Mt &mt, int64_t n, uint32_t *Jd) {
    int64_t i = n - 1;
    uint32_t *o = Jd;
    while (i >= 1) {
        const int b = 31 - __builtin_clz((uint32_t)i);
        const int64_t lo = (int64_t)1 << b;                      // the smallest i of this bit length
        const uint32_t mask = 0xffffffffu >> (31 - b);           // numpy's smallest all-ones mask covering i
        while (i >= lo) {
            if (mt.pos == MT_N) {
                mt_reload(mt.key);
                mt.pos = 0;
                mt.temper_from(0);
            }
            const uint32_t *w = mt.out + mt.pos;
            const int avail = MT_N - mt.pos;
            int t = 0;
            while (t < avail && i >= lo) {
                const uint32_t v = w[t++] & mask;
                *o = v;
                const int64_t ok = (int64_t)(v <= (uint32_t)i);
                o += ok;
                i -= ok;
            }
            mt.pos += t;
        }
    }
}

#ifndef __HIP_DEVICE_COMPILE__
#define SSW_AVX512 __attribute__((target("avx512f,avx512bw,avx512dq,avx512vl,popcnt")))
// The generator itself, sixteen words a step (the host pass of hipcc compiles for baseline x86-64: its scalar reload +
// tempering took 360 ns per 624-word block, 1.2 of the 1.5 ms the draws cost on the build host; 140 ns here).  A step of
// the first loop reads key[i .. i+16] and key[i+397 .. i+412], none written yet; a step of the second reads
// key[i-227 .. i-212], all written at least 211 words earlier: the same values as the scalar recurrence.
SSW_AVX512 static inline void mt_step16(uint32_t *key, int i, int off) {
    const __m512i U = _mm512_set1_epi32((int)MT_UPPER), L = _mm512_set1_epi32((int)MT_LOWER);
    const __m512i A = _mm512_set1_epi32((int)MT_MATRIX_A), one = _mm512_set1_epi32(1), zero = _mm512_setzero_si512();
    const __m512i k0 = _mm512_loadu_si512(key + i), k1 = _mm512_loadu_si512(key + i + 1);
    const __m512i km = _mm512_loadu_si512(key + i + off);
    const __m512i y = _mm512_or_si512(_mm512_and_si512(k0, U), _mm512_and_si512(k1, L));
    const __m512i mag = _mm512_and_si512(_mm512_sub_epi32(zero, _mm512_and_si512(y, one)), A);
    _mm512_storeu_si512(key + i, _mm512_xor_si512(_mm512_xor_si512(km, _mm512_srli_epi32(y, 1)), mag));
}
static inline void mt_step1(uint32_t *key, int i, int from) {
    const uint32_t y = (key[i] & MT_UPPER) | (key[(i + 1) % MT_N] & MT_LOWER);
    key[i] = key[from] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1u)) & MT_MATRIX_A);
}
SSW_AVX512 void mt_refill_avx512(uint32_t *key, uint32_t *out) {
    int i = 0;
    for (; i + 16 <= MT_N - MT_M; i += 16) mt_step16(key, i, MT_M);
    for (; i < MT_N - MT_M; ++i) mt_step1(key, i, i + MT_M);
    for (; i + 16 <= MT_N - 1; i += 16) mt_step16(key, i, MT_M - MT_N);
    for (; i < MT_N - 1; ++i) mt_step1(key, i, i + (MT_M - MT_N));
    mt_step1(key, MT_N - 1, MT_M - 1);
    for (int j = 0; j < MT_N; j += 16) {  // 624 = 39 x 16
        __m512i y = _mm512_loadu_si512(key + j);
        y = _mm512_xor_si512(y, _mm512_srli_epi32(y, 11));
        y = _mm512_xor_si512(y, _mm512_and_si512(_mm512_slli_epi32(y, 7), _mm512_set1_epi32((int)0x9d2c5680u)));
        y = _mm512_xor_si512(y, _mm512_and_si512(_mm512_slli_epi32(y, 15), _mm512_set1_epi32((int)0xefc60000u)));
        y = _mm512_xor_si512(y, _mm512_srli_epi32(y, 18));
        _mm512_storeu_si512(out + j, y);
    }
}
// The same walk sixteen words at a time: a word is surely accepted if (word & mask) <= i - 16 and surely rejected if
// it is > i, whatever the fifteen words before it did; a block with a word in between (16 / 2^bits of them) or that
// crosses a bit length is walked by the scalar loop.  Accepted values are compressed in a register and stored whole
// (the tail is overwritten by the next store).
SSW_AVX512 void draw_targets_avx512(Mt &mt, int64_t n, uint32_t *Jd) {
    int64_t i = n - 1;
    uint32_t *o = Jd;
    while (i >= 1) {
        const int b = 31 - __builtin_clz((uint32_t)i);
        const int64_t lo = (int64_t)1 << b;
        const uint32_t mask = 0xffffffffu >> (31 - b);
        const __m512i vmask = _mm512_set1_epi32((int)mask);
        while (i >= lo) {
            if (mt.pos == MT_N) {
                mt_refill_avx512(mt.key, mt.out);
                mt.pos = 0;
            }
            // sixty-four words a trip while every one of them is decided by the bounds (accepted if <= i - 64,
            // rejected if > i): the loop-carried chain (popcount -> i -> broadcast -> compares) is paid once per 64
            while (mt.pos + 64 <= MT_N && i - 64 >= lo) {
                const __m512i ta = _mm512_set1_epi32((int)(i - 64)), tr = _mm512_set1_epi32((int)i);
                const __m512i v0 = _mm512_and_si512(_mm512_loadu_si512(mt.out + mt.pos), vmask);
                const __m512i v1 = _mm512_and_si512(_mm512_loadu_si512(mt.out + mt.pos + 16), vmask);
                const __m512i v2 = _mm512_and_si512(_mm512_loadu_si512(mt.out + mt.pos + 32), vmask);
                const __m512i v3 = _mm512_and_si512(_mm512_loadu_si512(mt.out + mt.pos + 48), vmask);
                const __mmask16 a0 = _mm512_cmple_epu32_mask(v0, ta), a1 = _mm512_cmple_epu32_mask(v1, ta);
                const __mmask16 a2 = _mm512_cmple_epu32_mask(v2, ta), a3 = _mm512_cmple_epu32_mask(v3, ta);
                const __mmask16 r0 = _mm512_cmpgt_epu32_mask(v0, tr), r1 = _mm512_cmpgt_epu32_mask(v1, tr);
                const __mmask16 r2 = _mm512_cmpgt_epu32_mask(v2, tr), r3 = _mm512_cmpgt_epu32_mask(v3, tr);
                if ((__mmask16)((a0 | r0) & (a1 | r1) & (a2 | r2) & (a3 | r3)) != (__mmask16)0xffff) break;
                const int c0 = __builtin_popcount((unsigned)a0), c1 = __builtin_popcount((unsigned)a1);
                const int c2 = __builtin_popcount((unsigned)a2), c3 = __builtin_popcount((unsigned)a3);
                _mm512_storeu_si512(o, _mm512_maskz_compress_epi32(a0, v0));
                _mm512_storeu_si512(o + c0, _mm512_maskz_compress_epi32(a1, v1));
                _mm512_storeu_si512(o + c0 + c1, _mm512_maskz_compress_epi32(a2, v2));
                _mm512_storeu_si512(o + c0 + c1 + c2, _mm512_maskz_compress_epi32(a3, v3));
                const int c = c0 + c1 + c2 + c3;
                o += c;
                i -= c;
                mt.pos += 64;
            }
            while (mt.pos + 16 <= MT_N && i - 16 >= lo) {
                const __m512i v = _mm512_and_si512(_mm512_loadu_si512(mt.out + mt.pos), vmask);
                const __mmask16 acc = _mm512_cmple_epu32_mask(v, _mm512_set1_epi32((int)(i - 16)));
                const __mmask16 rej = _mm512_cmpgt_epu32_mask(v, _mm512_set1_epi32((int)i));
                if ((__mmask16)(acc | rej) != (__mmask16)0xffff) break;
                _mm512_storeu_si512(o, _mm512_maskz_compress_epi32(acc, v));
                const int c = __builtin_popcount((unsigned)acc);
                o += c;
                i -= c;
                mt.pos += 16;
            }
            const uint32_t *w = mt.out + mt.pos;
            const int avail = MT_N - mt.pos < 16 ? MT_N - mt.pos : 16;
            int t = 0;
            while (t < avail && i >= lo) {
                const uint32_t v = w[t++] & mask;
                *o = v;
                const int64_t ok = (int64_t)(v <= (uint32_t)i);
                o += ok;
                i -= ok;
            }
            mt.pos += t;
        }
    }
}
#endif

// The first k entries of the shuffled arange(n), given every swap target.  The shuffle applies swap(a[i], a[J[i]]) for
// i = n-1 ... 1; what ends at position p is found by walking the swaps backwards (i = 1 ... n-1) and moving a pointer
// that starts at p: at i it jumps to J[i], at J[i] it jumps to i, and where it stands after the last step is the
// value (a starts as the identity).  All k pointers are walked at once: a bitmap of the occupied positions (n/8 bytes,
// cache-resident, one random bit test per step) says whether a step touches any of them -- about 4 % of the steps do
// for 10 000 of 1.56 M -- and `slot_at[p]` names the pointer at an occupied position (read only where the bit is set,
// so it is never initialised).  No n-sized array is shuffled.
struct Scratch {  // per thread, kept between calls: fresh 6-MB allocations cost their page faults every round
    std::vector<uint32_t> Jd;
    std::vector<int32_t> slot_at;
    std::vector<uint64_t> bits;
};

class PrefixTrace {
   public:
    PrefixTrace(int64_t n, int64_t k, Scratch &sc) : where_((size_t)k) {
        if ((int64_t)sc.slot_at.size() < n) sc.slot_at.resize((size_t)n);
        sc.bits.assign((size_t)((n + 63) / 64), 0);
        bits_ = sc.bits.data();
        slot_at_ = sc.slot_at.data();
        for (int64_t s = 0; s < k; ++s) place((uint32_t)s, (int32_t)s);
    }
    inline bool occupied(uint32_t p) const { return (bits_[p >> 6] >> (p & 63)) & 1u; }
    inline void swap_positions(uint32_t i, uint32_t j) {  // at least one of them is occupied, i != j
        const bool oi = occupied(i), oj = occupied(j);
        const int32_t si = slot_at_[i], sj = slot_at_[j];  // meaningful where occupied
        if (oi && oj) {
            place(j, si);
            place(i, sj);
        } else if (oi) {
            bits_[i >> 6] &= ~((uint64_t)1 << (i & 63));
            place(j, si);
        } else {
            bits_[j >> 6] &= ~((uint64_t)1 << (j & 63));
            place(i, sj);
        }
    }
    const std::vector<int32_t> &where() const { return where_; }

   private:
    inline void place(uint32_t p, int32_t s) {
        where_[(size_t)s] = (int32_t)p;
        slot_at_[p] = s;
        bits_[p >> 6] |= (uint64_t)1 << (p & 63);
    }
    uint64_t *bits_ = nullptr;
    int32_t *slot_at_ = nullptr;
    std::vector<int32_t> where_;
};

// steps i0 ... i1-1 of the backward walk
inline void trace_steps(PrefixTrace &tr, int64_t n, const uint32_t *Jd, int64_t i0, int64_t i1) {
    for (int64_t i = i0; i < i1; ++i) {
        const uint32_t j = Jd[(size_t)(n - 1 - i)];
        if ((tr.occupied((uint32_t)i) || tr.occupied(j)) && j != (uint32_t)i) tr.swap_positions((uint32_t)i, j);
    }
}

#ifndef __HIP_DEVICE_COMPILE__
bool have_avx512() {
    static const bool yes = !getenv("SSW_NPRAND_NO_AVX512") && __builtin_cpu_supports("avx512f") &&
                            __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512dq") &&
                            __builtin_cpu_supports("avx512vl");
    return yes;
}
#else
bool have_avx512() { return false; }
void draw_targets_avx512(Mt &, int64_t, uint32_t *) {}
#endif


// ---- the walk on the device ------------------------------------------------------------------------------------------
// The host walk above follows all k pointers through the 1.56 M steps in one serial pass (2 ms on the EPYC host, most
// of a draw).  Traced one pointer at a time the walk needs no pass at all: what ends at position p came, at step p
// (swap(a[p], a[J[p]]), the last step to touch p), from position J[p]; before that, the value at a position q below the
// current step can only have come from a step i with J[i] == q -- the LATEST executed, i.e. the SMALLEST i above where
// the trace stands -- and then sat at position i, and so on until no step targets it: that position is the value
// (a starts as the identity).  So:   cur = p, t = 1;  if p >= 1: cur = J[p], t = p + 1;
//                                    while some i >= t has J[i] == cur: take the smallest, cur = i, t = i + 1.
// The lists {i : J[i] == q} are built with one atomic exchange per step (their order is the atomics' order; the
// minimum over a list does not depend on it) and are short: ln(n / q) entries on average.  ~6 hops per pointer.
__global__ void k_np_build_lists(const uint32_t *__restrict__ Jd, int64_t n, int32_t *__restrict__ head,
                                 int32_t *__restrict__ nxt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;  // steps 1 .. n-1
    if (i >= n) return;
    const uint32_t j = Jd[n - 1 - i];
    nxt[i] = atomicExch(&head[j], (int32_t)i);
}

__global__ void k_np_trace(const uint32_t *__restrict__ Jd, int64_t n, int64_t k, const int32_t *__restrict__ head,
                           const int32_t *__restrict__ nxt, int64_t *__restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= k) return;
    int64_t cur = p, t = 1;
    if (p >= 1) {
        cur = Jd[n - 1 - p];
        t = p + 1;
    }
    for (;;) {
        int32_t best = 0x7fffffff;
        for (int32_t i = head[cur]; i >= 0; i = nxt[i])
            if (i >= t && i < best) best = i;
        if (best == 0x7fffffff) break;
        cur = best;
        t = (int64_t)best + 1;
    }
    out[p] = cur;
}

struct DevWalk {  // per device, kept between calls
    int device = -1;
    hipStream_t stream = nullptr;
    uint32_t *Jd_host = nullptr;  // pinned: the draws are written straight into it
    uint32_t *Jd = nullptr;
    int32_t *head = nullptr, *nxt = nullptr;
    int64_t *out = nullptr, *out_host = nullptr;
    int64_t cap_n = 0, cap_k = 0;
};

ssw_status dev_walk_reserve(DevWalk &w, int device, int64_t n, int64_t k) {
    if (w.device != device) {
        if (w.device >= 0) {
            ssw::set_error("np_permutation_prefix: one device per thread (was %d, now %d)", w.device, device);
            return SSW_ERR_INVALID;
        }
        w.device = device;
        SSW_HIP_TRY(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
    }
    if (n + 16 > w.cap_n) {
        if (w.Jd_host) (void)hipHostFree(w.Jd_host);
        (void)hipFree(w.Jd);
        (void)hipFree(w.head);
        (void)hipFree(w.nxt);
        w.Jd_host = nullptr; w.Jd = nullptr; w.head = nullptr; w.nxt = nullptr; w.cap_n = 0;
        const int64_t cap = n + n / 8 + 64;
        SSW_HIP_TRY(hipHostMalloc((void **)&w.Jd_host, (size_t)cap * sizeof(uint32_t), hipHostMallocDefault));
        SSW_HIP_TRY(hipMalloc((void **)&w.Jd, (size_t)cap * sizeof(uint32_t)));
        SSW_HIP_TRY(hipMalloc((void **)&w.head, (size_t)cap * sizeof(int32_t)));
        SSW_HIP_TRY(hipMalloc((void **)&w.nxt, (size_t)cap * sizeof(int32_t)));
        w.cap_n = cap;
    }
    if (k > w.cap_k) {
        if (w.out_host) (void)hipHostFree(w.out_host);
        (void)hipFree(w.out);
        w.out = nullptr; w.out_host = nullptr; w.cap_k = 0;
        const int64_t cap = k + k / 8 + 64;
        SSW_HIP_TRY(hipHostMalloc((void **)&w.out_host, (size_t)cap * sizeof(int64_t), hipHostMallocDefault));
        SSW_HIP_TRY(hipMalloc((void **)&w.out, (size_t)cap * sizeof(int64_t)));
        w.cap_k = cap;
    }
    return SSW_OK;
}

}  // namespace

extern "C" ssw_status ssw_np_permutation_prefix(uint32_t *mt_key624, int32_t *mt_pos, int64_t n, int64_t k,
                                                int64_t *out_prefix) {
    SSW_REQUIRE(mt_key624 && mt_pos, "NULL state");
    SSW_REQUIRE(*mt_pos >= 0 && *mt_pos <= MT_N, "MT19937 position %d outside [0, 624]", *mt_pos);
    SSW_REQUIRE(n >= 0 && k >= 0, "negative size");
    if (k > n) k = n;
    SSW_REQUIRE(k == 0 || out_prefix, "NULL output");
    Mt mt;
    mt.key = mt_key624;
    mt.pos = *mt_pos;
    mt.temper_from(mt.pos);
    if (n <= 0x7fffffffll) {
        static thread_local Scratch sc;
        if ((int64_t)sc.Jd.size() < n + 16) sc.Jd.resize((size_t)n + 16);
        std::vector<uint32_t> &Jd = sc.Jd;  // Jd[t] = the target of step i = n-1-t
        const bool wide = have_avx512();
        static const bool timing = getenv("SSW_NPRAND_TIMING") != nullptr;
        auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t0 = timing ? now_ms() : 0.0;
        if (wide) draw_targets_avx512(mt, n, Jd.data());
        else draw_targets(mt, n, Jd.data());  // either way the stream is consumed for the whole shuffle, whatever k is
        const double t1 = timing ? now_ms() : 0.0;
        if (k * 16 <= n) {
            PrefixTrace tr(n, k, sc);
            // (sixteen steps tested at once with a gather of their bitmap words was measured on the EPYC host: 2.16 ms
            // against 2.03 ms for this loop -- the walk's time is the 60 000 pointer moves, not the 1.56 M tests)
            if (k > 0) trace_steps(tr, n, Jd.data(), 1, n);
            for (int64_t s = 0; s < k; ++s) out_prefix[s] = tr.where()[(size_t)s];
            if (timing) fprintf(stderr, "[nprand] draws %.3f ms, walk %.3f ms (avx512 %d)\n", t1 - t0, now_ms() - t1, (int)wide);
        } else {  // a long prefix: shuffle the array itself
            std::vector<int32_t> a((size_t)n);
            for (int64_t i = 0; i < n; ++i) a[(size_t)i] = (int32_t)i;
            constexpr int AHEAD = 32;
            for (int64_t i = n - 1; i >= 1; --i) {
                if (i > AHEAD) __builtin_prefetch(&a[Jd[(size_t)(n - 1 - (i - AHEAD))]], 1);
                const uint32_t j = Jd[(size_t)(n - 1 - i)];
                const int32_t t = a[j];
                a[j] = a[(size_t)i];
                a[(size_t)i] = t;
            }
            for (int64_t i = 0; i < k; ++i) out_prefix[i] = a[(size_t)i];
        }
    } else {
        std::vector<int64_t> a((size_t)n);
        for (int64_t i = 0; i < n; ++i) a[(size_t)i] = i;
        for (int64_t i = n - 1; i >= 1; --i) {
            const int64_t j = (int64_t)mt.interval((uint64_t)i);
            const int64_t t = a[(size_t)j];
            a[(size_t)j] = a[(size_t)i];
            a[(size_t)i] = t;
        }
        for (int64_t i = 0; i < k; ++i) out_prefix[i] = a[(size_t)i];
    }
    *mt_pos = mt.pos;
    return SSW_OK;
}

// The same draw with the walk on a GPU (see k_np_trace): the draws are made on the host, into pinned memory, and cross
// the link once (4 n bytes).  For the shapes the host walk is good at (n under 2^18, or a long prefix) this is the host
// entry point.
extern "C" ssw_status ssw_np_permutation_prefix_dev(int32_t device, uint32_t *mt_key624, int32_t *mt_pos, int64_t n,
                                                    int64_t k, int64_t *out_prefix) {
    SSW_REQUIRE(mt_key624 && mt_pos, "NULL state");
    SSW_REQUIRE(*mt_pos >= 0 && *mt_pos <= MT_N, "MT19937 position %d outside [0, 624]", *mt_pos);
    SSW_REQUIRE(n >= 0 && k >= 0, "negative size");
    if (k > n) k = n;
    if (device < 0 || n < ((int64_t)1 << 18) || n > 0x7fffffffll || k * 16 > n || k == 0)
        return ssw_np_permutation_prefix(mt_key624, mt_pos, n, k, out_prefix);
    SSW_REQUIRE(out_prefix, "NULL output");
    ssw::DeviceGuard guard(device);
    static thread_local DevWalk w;
    SSW_TRY(dev_walk_reserve(w, device, n, k));
    Mt mt;
    mt.key = mt_key624;
    mt.pos = *mt_pos;
    mt.temper_from(mt.pos);
    if (have_avx512()) draw_targets_avx512(mt, n, w.Jd_host);
    else draw_targets(mt, n, w.Jd_host);
    *mt_pos = mt.pos;
    SSW_HIP_TRY(hipMemcpyAsync(w.Jd, w.Jd_host, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, w.stream));
    SSW_HIP_TRY(hipMemsetAsync(w.head, 0xff, (size_t)n * sizeof(int32_t), w.stream));
    hipLaunchKernelGGL(k_np_build_lists, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, w.stream, w.Jd, n, w.head, w.nxt);
    hipLaunchKernelGGL(k_np_trace, dim3((unsigned)((k + 63) / 64)), dim3(64), 0, w.stream, w.Jd, n, k, w.head, w.nxt, w.out);
    SSW_HIP_TRY(hipGetLastError());
    SSW_HIP_TRY(hipMemcpyAsync(w.out_host, w.out, (size_t)k * sizeof(int64_t), hipMemcpyDeviceToHost, w.stream));
    SSW_HIP_TRY(hipStreamSynchronize(w.stream));
    memcpy(out_prefix, w.out_host, (size_t)k * sizeof(int64_t));
    return SSW_OK;
}
