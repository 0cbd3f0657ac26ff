// rng.hip -- synthetic unit-norm rows generated on the device (gfx950 / MI355X)
//
// BASELINE.json's configs use "synthetic d=512 vectors"; 100M x 512 f32 (204.8 GB) can
// not be produced on, or uploaded from, the host in reasonable time, so the index is
// filled in place by a counter-based generator whose every element is a pure function
// of (seed, global row, column).  All arithmetic is integer or correctly-rounded IEEE
// (f64 sqrt / divide, one f64->f32 rounding), so oracle/seesaw_oracle.py::synth_rows
// reproduces the same BITS on the CPU -- the parity tests use that to check scans of
// device-generated shards against the oracle without moving the shard.
//
//   mix32(x): x ^= x>>16; x *= 0x7feb352d; x ^= x>>15; x *= 0x846ca68b; x ^= x>>16
//   rowkey   = mix32(mix32(lo32(row) ^ mix32(hi32(row) ^ lo32(seed))) ^ hi32(seed))
//   h1       = mix32(rowkey ^ (col * 0x9E3779B9));   h2 = mix32(h1 ^ 0x85EBCA6B)
//   x_int    = lo16(h1) + hi16(h1) + lo16(h2) + hi16(h2) - 131070      (Irwin-Hall, n=4)
//   value    = f32( f64(x_int) / sqrt(f64(sum_c x_int^2)) )
#include "ssw_common.h"

namespace ssw {

namespace {

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

__device__ __forceinline__ int synth_int(uint32_t rowkey, uint32_t col) {
    const uint32_t h1 = mix32(rowkey ^ (col * 0x9E3779B9u));
    const uint32_t h2 = mix32(h1 ^ 0x85EBCA6Bu);
    return (int)((h1 & 0xffffu) + (h1 >> 16) + (h2 & 0xffffu) + (h2 >> 16)) - 131070;
}

// one wave per row; lane l produces columns 256*c + 4*l .. +3 (the scan's load layout)
template <int C>
__global__ __launch_bounds__(256) void k_fill_random(float *__restrict__ X, int64_t n,
                                                     uint64_t seed, int64_t first_row) {
    const int lane = threadIdx.x & 63;
    const int64_t gwave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r = gwave; r < n; r += nwaves) {
        const uint64_t grow = (uint64_t)(first_row + r);
        const uint32_t rowkey =
            mix32(mix32((uint32_t)grow ^ mix32((uint32_t)(grow >> 32) ^ (uint32_t)seed)) ^
                  (uint32_t)(seed >> 32));
        int xi[C * 4];
        long long ss = 0;
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int v = synth_int(rowkey, (uint32_t)(256 * c + 4 * lane + j));
                xi[c * 4 + j] = v;
                ss += (long long)v * v;
            }
        // exact integer sum over the wave (< 2^53, carried in f64)
        double tot = (double)ss;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) tot += __shfl_xor(tot, off, 64);
        const bool degenerate = !(tot > 0.0);
        const double norm = degenerate ? 1.0 : sqrt(tot);
        float4 *dst = reinterpret_cast<float4 *>(X + r * (int64_t)(C * 256)) + lane;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float4 o;
            o.x = (float)((double)xi[c * 4 + 0] / norm);
            o.y = (float)((double)xi[c * 4 + 1] / norm);
            o.z = (float)((double)xi[c * 4 + 2] / norm);
            o.w = (float)((double)xi[c * 4 + 3] / norm);
            if (degenerate && c == 0 && lane == 0) o.x = 1.0f;
            dst[c * 64] = o;
        }
    }
}

}  // namespace

ssw_status launch_fill_random(float *X, int64_t n, int32_t dim, uint64_t seed, int64_t first_row,
                              hipStream_t stream) {
    if (n <= 0) return SSW_OK;
    int64_t grid = (n + 3) / 4;
    if (grid > 256 * 8 * 4) grid = 256 * 8 * 4;
    switch (dim) {
        case 256:
            hipLaunchKernelGGL(k_fill_random<1>, dim3((unsigned)grid), dim3(256), 0, stream, X, n,
                               seed, first_row);
            break;
        case 512:
            hipLaunchKernelGGL(k_fill_random<2>, dim3((unsigned)grid), dim3(256), 0, stream, X, n,
                               seed, first_row);
            break;
        case 768:
            hipLaunchKernelGGL(k_fill_random<3>, dim3((unsigned)grid), dim3(256), 0, stream, X, n,
                               seed, first_row);
            break;
        case 1024:
            hipLaunchKernelGGL(k_fill_random<4>, dim3((unsigned)grid), dim3(256), 0, stream, X, n,
                               seed, first_row);
            break;
        default:
            set_error("fill_random: dim=%d unsupported", dim);
            return SSW_ERR_UNSUPPORTED;
    }
    SSW_HIP_TRY(hipGetLastError());
    return SSW_OK;
}

}  // namespace ssw
