// nprand.hip -- host-only: numpy's legacy `np.random.permutation(n)[:k]`, on numpy's own MT19937 state.
//
// PseudoLR draws its pseudo-labelled sample with `np.random.permutation(n_unlabelled)[:sample_size]`
// (seesaw/loops/util.py:11-14): which rows are drawn decides the fit, so a drop-in has to reproduce the draw, i.e.
// consume the global RandomState's stream exactly as numpy does.  numpy shuffles np.arange(n) with a generic
// memcpy-swap per element (15 ms for the 1.56 M unlabelled rows of the LVIS-scale bench, most of a PseudoLR round);
// this is the same algorithm on an int32 array:
//     for i = n-1 ... 1:  j = random_interval(i);  swap(a[i], a[j])          (mtrand.pyx _shuffle_raw)
//     random_interval(max): mask = 2^ceil(log2(max+1)) - 1; draw 32-bit words until (word & mask) <= max
//                                                                          (distributions.c random_interval)
//     word = MT19937 genrand with numpy's (randomkit's) reload and tempering    (mt19937.c)
// The caller passes the 624-word key and position of np.random.get_state() and writes them back with set_state():
// the stream continues exactly where numpy's own call would have left it (tests/test_nprand_cpu.py).
#include <cstdint>
#include <vector>

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
    // The draws do not depend on the array, only on i: they are made a batch ahead and their targets prefetched, so
    // the swaps (random accesses into a 6-MB array) do not wait for memory one at a time.
    constexpr int BATCH = 64;
    int64_t js[BATCH];
    if (n <= 0x7fffffffll) {
        std::vector<int32_t> a((size_t)n);
        for (int64_t i = 0; i < n; ++i) a[(size_t)i] = (int32_t)i;
        for (int64_t i0 = n - 1; i0 >= 1; i0 -= BATCH) {
            const int cnt = (int)(i0 < BATCH ? i0 : BATCH);  // i = i0, i0 - 1, ..., i0 - cnt + 1  (all >= 1)
            for (int b = 0; b < cnt; ++b) {
                const uint32_t mx = (uint32_t)(i0 - b);  // >= 1: the smallest all-ones mask covering it
                js[b] = (int64_t)mt.interval32(mx, 0xffffffffu >> __builtin_clz(mx));
                __builtin_prefetch(&a[(size_t)js[b]], 1);
            }
            for (int b = 0; b < cnt; ++b) {
                const int64_t i = i0 - b, j = js[b];
                const int32_t t = a[(size_t)j];
                a[(size_t)j] = a[(size_t)i];
                a[(size_t)i] = t;
            }
        }
        for (int64_t i = 0; i < k; ++i) out_prefix[i] = a[(size_t)i];
    } else {
        std::vector<int64_t> a((size_t)n);
        for (int64_t i = 0; i < n; ++i) a[(size_t)i] = i;
        for (int64_t i0 = n - 1; i0 >= 1; i0 -= BATCH) {
            const int cnt = (int)(i0 < BATCH ? i0 : BATCH);
            for (int b = 0; b < cnt; ++b) {
                js[b] = (int64_t)mt.interval((uint64_t)(i0 - b));
                __builtin_prefetch(&a[(size_t)js[b]], 1);
            }
            for (int b = 0; b < cnt; ++b) {
                const int64_t i = i0 - b, j = js[b];
                const int64_t t = a[(size_t)j];
                a[(size_t)j] = a[(size_t)i];
                a[(size_t)i] = t;
            }
        }
        for (int64_t i = 0; i < k; ++i) out_prefix[i] = a[(size_t)i];
    }
    *mt_pos = mt.pos;
    return SSW_OK;
}
