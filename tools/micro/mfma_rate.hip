// micro-benchmark: what the matrix cores of an MI355X deliver to ONE resident wave per SIMD (and to two) in a bare
// loop of bf16 MFMAs on random operands held in registers -- the ceiling any GEMM K-loop of that occupancy sits under --
// together with the clock the chip holds meanwhile (s_memtime ticks / s_memrealtime 100-MHz ticks).
// Variants: v_mfma_f32_16x16x32_bf16 on 64 / 32 independent accumulators, through the builtin and through inline asm
// with the accumulators tied in place in the accumulator file (what csrc/gemm_pw4.hip issues); 32x32x16 on 16.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_rate.hip -o tools/micro/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ unsigned long long g_ticks[3];

template <int NACC, bool ASM>
__global__ __launch_bounds__(256) void k_mfma16(const bf16x8 *src, float *sink, int iters) {
    bf16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = src[(threadIdx.x * 16 + i) & 4095];
        b[i] = src[(threadIdx.x * 16 + 8 + i) & 4095];
    }
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if constexpr (ASM)
                asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(b[i & 7]), "v"(a[(i >> 3) & 7]));
            else
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[i & 7], a[(i >> 3) & 7], acc[i], 0, 0, 0);
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long c1 = clock64(), r1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        g_ticks[0] = c1 - c0;
        g_ticks[1] = r1 - r0;
    }
}

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma32(const bf16x8 *src, float *sink, int iters) {
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = src[(threadIdx.x * 8 + i) & 4095];
        b[i] = src[(threadIdx.x * 8 + 4 + i) & 4095];
    }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
    const unsigned long long c0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[i & 3], a[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = clock64(), r1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int k = 0; k < 16; ++k) s += acc[i][k];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        g_ticks[0] = c1 - c0;
        g_ticks[1] = r1 - r0;
    }
}

// the same MFMA streams with fragment reads from LDS between them, as a GEMM K loop has: per "sub-step" 64 (32) MFMAs,
// NRD ds_read_b128 spread over the first groups, then s_waitcnt lgkmcnt(0) -- does the dense MFMA stream of a single wave
// per SIMD let the LDS returns through, and what does the closing wait cost?
template <int NRD, bool BIG>
__global__ __launch_bounds__(256) void k_mfma_lds(const bf16x8 *src, float *sink, int iters) {
    __shared__ bf16x8 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[i];
    __syncthreads();
    bf16x8 a[8], b[8], na[8], nb[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = src[(threadIdx.x * 16 + i) & 4095];
        b[i] = src[(threadIdx.x * 16 + 8 + i) & 4095];
        na[i] = a[i];
        nb[i] = b[i];
    }
    f32x4 acc[64];
    f32x16 accb[16];
    for (int i = 0; i < 64; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 16; ++i)
        for (int k = 0; k < 16; ++k) accb[i][k] = 0.f;
    const bf16x8 *base = lds + (threadIdx.x & 63);  // conflict-free: consecutive 16-byte slots
    const unsigned long long c0 = clock64(), r0 = wall_clock64();
    unsigned long long tail = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (g < NRD) {
                if (g & 1) na[g >> 1] = base[((it + g) & 31) * 64];
                else nb[g >> 1] = base[((it + g) & 31) * 64];
            } else if (g - 8 < NRD - 8 && g >= 8) {
            }
            if constexpr (!BIG) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int q = g * 4 + m;
                    asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[q]) : "v"(b[q & 7]), "v"(a[(q >> 3) & 7]));
                }
            } else {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int q = (g * 2 + m) & 15;
                    asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(accb[q]) : "v"(b[q & 7]), "v"(a[(q >> 2) & 7]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const unsigned long long t0 = clock64();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        tail += clock64() - t0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {  // the fragments just read become the operands of the next sub-step
            bf16x8 t = a[i]; a[i] = na[i]; na[i] = t;
            t = b[i]; b[i] = nb[i]; nb[i] = t;
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long c1 = clock64(), r1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 64; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += accb[i][0] + accb[i][15];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        g_ticks[0] = c1 - c0;
        g_ticks[1] = r1 - r0;
        g_ticks[2] = tail;
    }
}

template <typename F>
static void run(const char *name, F launch, double flop_per_wave_iter, int waves_per_cu, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch(iters / 8);  // warm
    hipDeviceSynchronize();
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        launch(iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    unsigned long long t[3];
    hipMemcpyFromSymbol(t, HIP_SYMBOL(g_ticks), sizeof(t));
    const double flop = flop_per_wave_iter * iters * waves_per_cu * 256.0;
    printf("%-52s %8.3f ms  %7.1f TFLOP/s  clock %.0f MHz  cycles/iter %.0f (closing wait %.0f)\n", name, best,
           flop / (best * 1e-3) / 1e12, t[1] ? (double)t[0] / (double)t[1] * 100.0 : 0.0, (double)t[0] / iters,
           (double)t[2] / iters);
    unsigned long long z[3] = {0, 0, 0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_ticks), z, sizeof(z));
}

int main() {
    std::vector<unsigned short> h(4096 * 8);
    srand(1);
    for (auto &v : h) {  // random bf16 in about [-2, 2)
        const float f = (float)rand() / RAND_MAX * 4.f - 2.f;
        unsigned u;
        memcpy(&u, &f, 4);
        v = (unsigned short)(u >> 16);
    }
    bf16x8 *src;
    float *sink;
    hipMalloc(&src, h.size() * 2);
    hipMalloc(&sink, 4);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int iters = 20000;
    const double f16 = 2.0 * 16 * 16 * 32, f32 = 2.0 * 32 * 32 * 16;
    run("16x16x32, 64 acc, asm in place, 1 wave/SIMD", [&](int n) { hipLaunchKernelGGL((k_mfma16<64, true>), dim3(256), dim3(256), 0, 0, src, sink, n); }, 64 * f16, 4, iters);
    run("16x16x32, 64 acc, builtin,      1 wave/SIMD", [&](int n) { hipLaunchKernelGGL((k_mfma16<64, false>), dim3(256), dim3(256), 0, 0, src, sink, n); }, 64 * f16, 4, iters);
    run("16x16x32, 32 acc, asm in place, 1 wave/SIMD", [&](int n) { hipLaunchKernelGGL((k_mfma16<32, true>), dim3(256), dim3(256), 0, 0, src, sink, n); }, 32 * f16, 4, iters);
    run("16x16x32, 32 acc, asm in place, 2 waves/SIMD", [&](int n) { hipLaunchKernelGGL((k_mfma16<32, true>), dim3(512), dim3(256), 0, 0, src, sink, n); }, 32 * f16, 8, iters);
    run("16x16x32, 32 acc, builtin,      2 waves/SIMD", [&](int n) { hipLaunchKernelGGL((k_mfma16<32, false>), dim3(512), dim3(256), 0, 0, src, sink, n); }, 32 * f16, 8, iters);
    run("32x32x16, 16 acc, builtin,      1 wave/SIMD", [&](int n) { hipLaunchKernelGGL((k_mfma32<16>), dim3(256), dim3(256), 0, 0, src, sink, n); }, 16 * f32, 4, iters);
    run("32x32x16,  8 acc, builtin,      2 waves/SIMD", [&](int n) { hipLaunchKernelGGL((k_mfma32<8>), dim3(512), dim3(256), 0, 0, src, sink, n); }, 8 * f32, 8, iters);
    const int it2 = 20000;
    run("16x16x32 x64 + 0 ds_read_b128 / sub-step, 1 wave/SIMD", [&](int n) { hipLaunchKernelGGL((k_mfma_lds<0, false>), dim3(256), dim3(256), 0, 0, src, sink, n); }, 64 * f16, 4, it2);
    run("16x16x32 x64 + 8 ds_read_b128 / sub-step, 1 wave/SIMD", [&](int n) { hipLaunchKernelGGL((k_mfma_lds<8, false>), dim3(256), dim3(256), 0, 0, src, sink, n); }, 64 * f16, 4, it2);
    run("16x16x32 x64 + 16 ds_read_b128 / sub-step, 1 wave/SIMD", [&](int n) { hipLaunchKernelGGL((k_mfma_lds<16, false>), dim3(256), dim3(256), 0, 0, src, sink, n); }, 64 * f16, 4, it2);
    run("32x32x16 x32 + 0 ds_read_b128 / sub-step, 1 wave/SIMD", [&](int n) { hipLaunchKernelGGL((k_mfma_lds<0, true>), dim3(256), dim3(256), 0, 0, src, sink, n); }, 32 * f32, 4, it2);
    run("32x32x16 x32 + 16 ds_read_b128 / sub-step, 1 wave/SIMD", [&](int n) { hipLaunchKernelGGL((k_mfma_lds<16, true>), dim3(256), dim3(256), 0, 0, src, sink, n); }, 32 * f32, 4, it2);
    return 0;
}
