// micro-benchmark: what one workgroup on an otherwise idle MI355X pays for a dependent global load (L2 hit),
// a wall_clock64() read, a __syncthreads() of 16 waves, and a streaming pass of one CU over an L2-resident buffer.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/latency.hip -o tools/micro/latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_chase(const int *next, int steps, long long *out, int *sink) {
    int p = 0;
    for (int i = 0; i < 64; ++i) p = next[p];  // warm
    const long long c0 = clock64();
    const unsigned long long w0 = wall_clock64();
    for (int i = 0; i < steps; ++i) p = next[p];
    const unsigned long long w1 = wall_clock64();
    const long long c1 = clock64();
    out[0] = c1 - c0;
    out[1] = (long long)(w1 - w0);
    *sink = p;
}
__global__ void k_wallclock(long long *out) {
    const long long c0 = clock64();
    unsigned long long s = 0;
    for (int i = 0; i < 100; ++i) s += wall_clock64();
    const long long c1 = clock64();
    out[0] = c1 - c0;
    out[1] = (long long)s;
}
__global__ __launch_bounds__(1024) void k_barrier(long long *out) {
    const long long c0 = clock64();
    for (int i = 0; i < 100; ++i) __syncthreads();
    const long long c1 = clock64();
    if (threadIdx.x == 0) out[0] = c1 - c0;
}
// one workgroup streams `bytes` (float4 per lane, R loads in flight per thread)
template <int R>
__global__ __launch_bounds__(1024) void k_stream(const float4 *buf, size_t n4, long long *out, float *sink) {
    float acc = 0.f;
    // warm pass (brings the buffer into this XCD's L2)
    for (size_t i = threadIdx.x; i < n4; i += blockDim.x) acc += buf[i].x;
    __syncthreads();
    const long long c0 = clock64();
    for (size_t i0 = threadIdx.x; i0 < n4; i0 += (size_t)blockDim.x * R) {
        float4 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const size_t i = i0 + (size_t)r * blockDim.x;
            v[r] = i < n4 ? buf[i] : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) acc += v[r].x + v[r].y + v[r].z + v[r].w;
    }
    __syncthreads();
    const long long c1 = clock64();
    if (threadIdx.x == 0) out[0] = c1 - c0;
    sink[threadIdx.x] = acc;
}

int main() {
    const int N = 1 << 18;  // 1 MiB of ints
    std::vector<int> h(N);
    const int stride = 1024 + 16;  // ~4 KB apart, co-prime walk
    for (int i = 0; i < N; ++i) h[i] = (int)(((long long)i + stride) % N);
    int *d_next, *d_sink;
    long long *d_out, h_out[2];
    float *d_fsink;
    hipMalloc(&d_next, N * sizeof(int));
    hipMalloc(&d_sink, sizeof(int));
    hipMalloc(&d_out, 2 * sizeof(long long));
    hipMalloc(&d_fsink, 1024 * sizeof(float));
    hipMemcpy(d_next, h.data(), N * sizeof(int), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, d_next, 2000, d_out, d_sink);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("dependent global load (1 MiB footprint, L2): %.0f cycles = %.0f ns each (clock64 %.0f MHz)\n",
               h_out[0] / 2000.0, h_out[1] * 10.0 / 2000.0, h_out[0] / (h_out[1] * 0.01));
    }
    hipLaunchKernelGGL(k_wallclock, dim3(1), dim3(64), 0, 0, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("wall_clock64(): %.0f cycles each\n", h_out[0] / 100.0);
    hipLaunchKernelGGL(k_barrier, dim3(1), dim3(1024), 0, 0, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("__syncthreads() of 16 waves: %.0f cycles each\n", h_out[0] / 100.0);
    float4 *d_buf;
    const size_t bytes = 800 * 1024;
    hipMalloc(&d_buf, bytes);
    hipMemset(d_buf, 0, bytes);
    for (int threads : {512, 1024}) {
        hipLaunchKernelGGL(k_stream<2>, dim3(1), dim3(threads), 0, 0, d_buf, bytes / 16, d_out, d_fsink);
        hipMemcpy(h_out, d_out, sizeof(long long), hipMemcpyDeviceToHost);
        printf("one workgroup (%d threads) streams 800 KB from L2, 2 x 16 B in flight per thread: %.0f cycles = %.1f B/clk\n",
               threads, (double)h_out[0], bytes / (double)h_out[0]);
        hipLaunchKernelGGL(k_stream<8>, dim3(1), dim3(threads), 0, 0, d_buf, bytes / 16, d_out, d_fsink);
        hipMemcpy(h_out, d_out, sizeof(long long), hipMemcpyDeviceToHost);
        printf("one workgroup (%d threads) streams 800 KB from L2, 8 x 16 B in flight per thread: %.0f cycles = %.1f B/clk\n",
               threads, (double)h_out[0], bytes / (double)h_out[0]);
    }
    return 0;
}
