// Lab-bench entry points (include/seesaw_hip_debug.h): compiled into libseesaw_hip_debug.so only.
#ifndef SSW_DEBUG_HOOKS
#error "debug_hooks.hip belongs to the lab build (-DSSW_DEBUG_HOOKS)"
#endif
#include "ssw_common.h"

// ---------------------------------------------------------------------------------------
// A/B harness (tools/perf_gemm.py): time one variant on seeded operands and compare its
// output with variant 0 in the same process.
// ---------------------------------------------------------------------------------------
namespace {
__global__ void k_debug_fill(__bf16 *x, int64_t n, uint32_t seed, float scale) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + seed;
        h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
        x[i] = (__bf16)(((int)(h & 0xffff) - 32768) * (scale / 32768.f));
    }
}
__global__ void k_debug_fill_f32(float *x, int64_t n, uint32_t seed, float scale) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + seed;
        h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
        x[i] = ((int)(h & 0xffff) - 32768) * (scale / 32768.f);
    }
}
template <typename T>
__global__ void k_debug_maxdiff(const T *a, const T *b, int64_t n, float *out) {
    float m = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = fabsf((float)a[i] - (float)b[i]);
        m = fmaxf(m, d == d ? d : 3.0e38f);
    }
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<int *>(out), __float_as_int(m));
}
}  // namespace

extern "C" ssw_status ssw_tune_gemm(int32_t variant) {
    if (variant != 0 && variant != 2 && variant != 7 && variant != 9 && variant != 14 && variant != 15 && variant != 16 && variant != 17 &&
        !(variant >= 20 && variant <= 23)) {
        ssw::set_error("ssw_tune_gemm: variant %d unknown (0, 2, 7)", variant);
        return SSW_ERR_INVALID;
    }
    ssw::tune_gemm(variant);
    return SSW_OK;
}

// diagnostics of the persistent kernel (gemm_pw4.hip): mode 1 accumulates cycle stamps, read back here as
// out4 = {cycles in the mid-step wait + barrier, cycles in K-steps, K-steps, waves}; modes 2-4 are ablations
extern "C" ssw_status ssw_debug_gemm_pw4_mode(int32_t mode, uint64_t *out6_or_null) {
    ssw::gemm_pw4_set_mode(mode);
    if (out6_or_null) return ssw::gemm_pw4_read_diag(reinterpret_cast<unsigned long long *>(out6_or_null), true);
    return SSW_OK;
}

extern "C" ssw_status ssw_debug_gemm_pw4_wg(uint64_t *out4096) {
    return ssw::gemm_pw4_read_wg(reinterpret_cast<unsigned long long *>(out4096));
}

extern "C" ssw_status ssw_debug_gemm(int32_t M, int32_t N, int32_t K, int32_t epi, int32_t variant, int32_t iters,
                                     float *out_ms, float *out_maxdiff) {
    using namespace ssw;
    if (M <= 0 || iters <= 0 || epi < 0 || epi > 7) {
        set_error("ssw_debug_gemm: bad arguments");
        return SSW_ERR_INVALID;
    }
    if (epi >= 4) {  // the LayerNorm-folded consumers (4, 5) and the row producers (6, 7): timing only, filled operands
        __bf16 *A = nullptr, *W = nullptr, *xc = nullptr;
        float *bias = nullptr, *c1 = nullptr, *st_in = nullptr, *st_out = nullptr;
        void *res = nullptr, *C = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        const int np_in = K / 128;
        auto cleanup = [&]() {
            for (void *p : {(void *)A, (void *)W, (void *)xc, (void *)bias, (void *)c1, (void *)st_in, (void *)st_out, res, C}) (void)hipFree(p);
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        };
        bool ok = hipMalloc(&A, (size_t)M * K * 2) == hipSuccess && hipMalloc(&W, (size_t)N * K * 2) == hipSuccess &&
                  hipMalloc(&xc, (size_t)M * N * 2) == hipSuccess && hipMalloc(&bias, (size_t)N * 4) == hipSuccess &&
                  hipMalloc(&c1, (size_t)N * 4) == hipSuccess && hipMalloc(&st_in, (size_t)M * np_in * 8) == hipSuccess &&
                  hipMalloc(&st_out, (size_t)M * (N / 128 + 1) * 8) == hipSuccess && hipMalloc(&res, (size_t)M * N * 4) == hipSuccess &&
                  hipMalloc(&C, (size_t)M * N * 4) == hipSuccess && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
        if (!ok) {
            cleanup();
            set_error("ssw_debug_gemm: allocation failed");
            return SSW_ERR_HIP;
        }
        hipLaunchKernelGGL(k_debug_fill, dim3(2048), dim3(256), 0, 0, A, (int64_t)M * K, 0x1234u, 1.0f);
        hipLaunchKernelGGL(k_debug_fill, dim3(2048), dim3(256), 0, 0, W, (int64_t)N * K, 0x9876u, 0.05f);
        hipLaunchKernelGGL(k_debug_fill_f32, dim3(64), dim3(256), 0, 0, bias, (int64_t)N, 0x4242u, 0.5f);
        hipLaunchKernelGGL(k_debug_fill_f32, dim3(64), dim3(256), 0, 0, c1, (int64_t)N, 0x4243u, 0.5f);
        hipLaunchKernelGGL(k_debug_fill_f32, dim3(2048), dim3(256), 0, 0, (float *)res, (int64_t)M * N, 0x7777u, 1.0f);
        (void)hipMemsetAsync(st_in, 0, (size_t)M * np_in * 8, 0);  // mean 0, variance 0: rstd = 1 / sqrt(eps)
        GemmLn ln;
        ln.stats_in = st_in; ln.np_in = np_in; ln.inv_dim = 1.f / K; ln.eps = 1.f; ln.c1 = c1;
        ln.xcopy = xc; ln.stats_out = st_out;
        const int keep = gemm_variant();
        tune_gemm(variant);
        int rc = launch_gemm_bf16_ln(epi, 0, A, W, bias, (const float *)res, C, M, N, K, ln);
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < iters && rc == SSW_OK; ++i) rc = launch_gemm_bf16_ln(epi, 0, A, W, bias, (const float *)res, C, M, N, K, ln);
        (void)hipEventRecord(e1, 0);
        tune_gemm(keep);
        float ms = 0.f;
        if (rc == SSW_OK && (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess)) rc = SSW_ERR_HIP;
        cleanup();
        if (out_ms) *out_ms = ms / iters;
        if (out_maxdiff) *out_maxdiff = 0.f;
        return (ssw_status)rc;
    }
    const bool out_bf16 = (epi == 1 || epi == 2);
    const size_t out_bytes = (size_t)M * N * (out_bf16 ? 2 : 4);
    __bf16 *A = nullptr, *W = nullptr;
    float *bias = nullptr, *res = nullptr, *diff = nullptr;
    void *c_ref = nullptr, *c_var = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = SSW_OK;
    auto cleanup = [&]() {
        for (void *p : {(void *)A, (void *)W, (void *)bias, (void *)res, (void *)diff, c_ref, c_var}) (void)hipFree(p);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
#define SSW_DBG_TRY(expr)                                                      \
    if (hipError_t _e = (expr); _e != hipSuccess) {                            \
        set_error("%s failed: %s", #expr, hipGetErrorString(_e));              \
        cleanup();                                                             \
        return SSW_ERR_HIP;                                                    \
    }
    SSW_DBG_TRY(hipMalloc(&A, (size_t)M * K * 2));
    SSW_DBG_TRY(hipMalloc(&W, (size_t)N * K * 2));
    SSW_DBG_TRY(hipMalloc(&bias, (size_t)N * 4));
    SSW_DBG_TRY(hipMalloc(&res, (size_t)M * N * 4));
    SSW_DBG_TRY(hipMalloc(&diff, 4));
    SSW_DBG_TRY(hipMalloc(&c_ref, out_bytes));
    SSW_DBG_TRY(hipMalloc(&c_var, out_bytes));
    hipLaunchKernelGGL(k_debug_fill, dim3(2048), dim3(256), 0, 0, A, (int64_t)M * K, 0x1234u, 1.0f);
    hipLaunchKernelGGL(k_debug_fill, dim3(2048), dim3(256), 0, 0, W, (int64_t)N * K, 0x9876u, 0.05f);
    hipLaunchKernelGGL(k_debug_fill_f32, dim3(64), dim3(256), 0, 0, bias, (int64_t)N, 0x4242u, 0.5f);
    hipLaunchKernelGGL(k_debug_fill_f32, dim3(2048), dim3(256), 0, 0, res, (int64_t)M * N, 0x7777u, 1.0f);
    SSW_DBG_TRY(hipMemsetAsync(diff, 0, 4, 0));
    SSW_DBG_TRY(hipEventCreate(&e0));
    SSW_DBG_TRY(hipEventCreate(&e1));
    const int keep = gemm_variant();
    tune_gemm(0);
    rc = launch_gemm_bf16_nt(epi, 0, A, W, bias, res, c_ref, M, N, K);
    tune_gemm(variant);
    if (rc == SSW_OK) rc = launch_gemm_bf16_nt(epi, 0, A, W, bias, res, c_var, M, N, K);  // warm-up + checked run
    if (rc == SSW_OK) {
        if (out_bf16)
            hipLaunchKernelGGL(k_debug_maxdiff<__bf16>, dim3(1024), dim3(256), 0, 0, (const __bf16 *)c_ref,
                               (const __bf16 *)c_var, (int64_t)M * N, diff);
        else
            hipLaunchKernelGGL(k_debug_maxdiff<float>, dim3(1024), dim3(256), 0, 0, (const float *)c_ref,
                               (const float *)c_var, (int64_t)M * N, diff);
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < iters && rc == SSW_OK; ++i) rc = launch_gemm_bf16_nt(epi, 0, A, W, bias, res, c_var, M, N, K);
        (void)hipEventRecord(e1, 0);
    }
    tune_gemm(keep);
    if (rc != SSW_OK) {
        cleanup();
        return rc;
    }
    SSW_DBG_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    SSW_DBG_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (out_ms) *out_ms = ms / iters;
    if (out_maxdiff) SSW_DBG_TRY(hipMemcpy(out_maxdiff, diff, 4, hipMemcpyDeviceToHost));
#undef SSW_DBG_TRY
    cleanup();
    return SSW_OK;
}


// ---------------------------------------------------------------------------------------
// one product / one fused launch on the caller's operands (tests/test_gemm_gpu.py)
// ---------------------------------------------------------------------------------------
namespace {
struct DevBufs {
    std::vector<void *> ptrs;
    ~DevBufs() {
        for (void *p : ptrs) (void)hipFree(p);
    }
    ssw_status up(const void *host, size_t bytes, void **dev, bool copy = true) {
        *dev = nullptr;
        SSW_HIP_TRY(hipMalloc(dev, bytes ? bytes : 16));
        ptrs.push_back(*dev);
        if (host && copy) SSW_HIP_TRY(hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice));
        else SSW_HIP_TRY(hipMemset(*dev, 0xFF, bytes));  // NaN patterns: an element the kernel skips shows
        return SSW_OK;
    }
};
}  // namespace

extern "C" ssw_status ssw_debug_gemm_run(int32_t epi, int32_t variant, int32_t M, int32_t N, int32_t K, const uint16_t *A_bf16,
                                         const uint16_t *W_bf16, const float *bias_or_c2, const float *residual_or_null,
                                         uint16_t *xcopy_inout_or_null, const float *stats_in_or_null, int32_t np_in,
                                         const float *c1_or_null, float inv_dim, float eps, void *C_out_or_null,
                                         float *stats_out_or_null) {
    using namespace ssw;
    SSW_REQUIRE(epi >= 0 && epi <= 10 && M > 0 && N > 0 && K > 0 && A_bf16 && W_bf16, "ssw_debug_gemm_run: bad arguments");
    if (epi == 9 || epi == 10) {
        // 9: the producer epilogue behind a split-K product (launch_gemm_splitk_stats: the text tower's fc2), `variant` = splits;
        // 10: the producer epilogue with the residual rows at a stride (GemmLn::res_ld: the pooled last layer's out-projection),
        //     `variant` = S: residual_or_null holds M * S rows of N, row m S is added to row m of the product
        SSW_REQUIRE(bias_or_c2 && residual_or_null && xcopy_inout_or_null && C_out_or_null && stats_out_or_null && variant >= 1,
                    "ssw_debug_gemm_run: the producer forms need bias, residual, the copy, the output, the statistics and splits / S >= 1");
        DevBufs d9;
        void *A9, *W9, *b9, *r9, *x9, *C9, *s9, *P9 = nullptr;
        const int64_t res_rows = epi == 10 ? (int64_t)M * variant : M;
        SSW_TRY(d9.up(A_bf16, (size_t)M * K * 2, &A9));
        SSW_TRY(d9.up(W_bf16, (size_t)N * K * 2, &W9));
        SSW_TRY(d9.up(bias_or_c2, (size_t)N * 4, &b9));
        SSW_TRY(d9.up(residual_or_null, (size_t)res_rows * N * 4, &r9));
        SSW_TRY(d9.up(nullptr, (size_t)M * N * 2, &x9));
        SSW_TRY(d9.up(nullptr, (size_t)M * N * 4, &C9));
        SSW_TRY(d9.up(nullptr, (size_t)M * (N / 128) * 2 * 4, &s9));
        GemmLn ln;
        ln.xcopy = (__bf16 *)x9;
        ln.stats_out = (float *)s9;
        if (epi == 9) {
            SSW_TRY(d9.up(nullptr, (size_t)variant * M * N * 4, &P9));
            SSW_TRY(launch_gemm_splitk_stats(0, A9, W9, (const float *)b9, (const float *)r9, (float *)C9, (float *)P9, M, N, K, variant, ln));
        } else {
            ln.res_ld = (int64_t)variant * N;
            SSW_TRY(launch_gemm_bf16_ln(6, 0, A9, W9, (const float *)b9, (const float *)r9, C9, M, N, K, ln));
        }
        SSW_HIP_TRY(hipDeviceSynchronize());
        SSW_HIP_TRY(hipMemcpy(C_out_or_null, C9, (size_t)M * N * 4, hipMemcpyDeviceToHost));
        SSW_HIP_TRY(hipMemcpy(xcopy_inout_or_null, x9, (size_t)M * N * 2, hipMemcpyDeviceToHost));
        SSW_HIP_TRY(hipMemcpy(stats_out_or_null, s9, (size_t)M * (N / 128) * 2 * 4, hipMemcpyDeviceToHost));
        return SSW_OK;
    }
    if (epi == 8) {  // the split-K product of few-tile shapes (launch_gemm_splitk_f32): `variant` = number of splits
        SSW_REQUIRE(bias_or_c2 && C_out_or_null && variant >= 1, "ssw_debug_gemm_run: split-K needs bias, an output and the split count");
        DevBufs d8;
        void *A8, *W8, *b8, *r8 = nullptr, *C8, *P8;
        SSW_TRY(d8.up(A_bf16, (size_t)M * K * 2, &A8));
        SSW_TRY(d8.up(W_bf16, (size_t)N * K * 2, &W8));
        SSW_TRY(d8.up(bias_or_c2, (size_t)N * 4, &b8));
        if (residual_or_null) SSW_TRY(d8.up(residual_or_null, (size_t)M * N * 4, &r8));
        SSW_TRY(d8.up(nullptr, (size_t)M * N * 4, &C8));
        SSW_TRY(d8.up(nullptr, (size_t)variant * M * N * 4, &P8));
        SSW_TRY(launch_gemm_splitk_f32(0, A8, W8, (const float *)b8, (const float *)r8, (float *)C8, (float *)P8, M, N, K, variant));
        SSW_HIP_TRY(hipDeviceSynchronize());
        SSW_HIP_TRY(hipMemcpy(C_out_or_null, C8, (size_t)M * N * 4, hipMemcpyDeviceToHost));
        return SSW_OK;
    }
    const bool c_bf16 = epi == 1 || epi == 2 || epi == 4 || epi == 5, c_f32 = epi == 0 || epi == 3 || epi == 6;
    const size_t mn = (size_t)M * N;
    const int n_tiles = N / 128;
    DevBufs d;
    void *A, *W, *bias = nullptr, *res = nullptr, *xc = nullptr, *sin = nullptr, *c1 = nullptr, *C = nullptr, *sout = nullptr;
    SSW_TRY(d.up(A_bf16, (size_t)M * K * 2, &A));
    SSW_TRY(d.up(W_bf16, (size_t)N * K * 2, &W));
    if (bias_or_c2) SSW_TRY(d.up(bias_or_c2, (size_t)N * 4, &bias));
    if (residual_or_null) SSW_TRY(d.up(residual_or_null, mn * 4, &res));
    if (xcopy_inout_or_null) SSW_TRY(d.up(xcopy_inout_or_null, mn * 2, &xc, epi == 7));
    if (stats_in_or_null) SSW_TRY(d.up(stats_in_or_null, (size_t)M * np_in * 2 * 4, &sin));
    if (c1_or_null) SSW_TRY(d.up(c1_or_null, (size_t)N * 4, &c1));
    if (c_bf16 || c_f32) SSW_TRY(d.up(nullptr, mn * (c_bf16 ? 2 : 4), &C));
    if (epi >= 6) SSW_TRY(d.up(nullptr, (size_t)M * n_tiles * 2 * 4, &sout));
    const int keep = gemm_variant();
    if (variant >= 0) tune_gemm(variant);
    ssw_status rc;
    if (epi <= 3) {
        rc = launch_gemm_bf16_nt(epi, 0, A, W, (const float *)bias, (const float *)res, C, M, N, K);
    } else {
        GemmLn ln;
        ln.stats_in = (const float *)sin;
        ln.np_in = np_in;
        ln.inv_dim = inv_dim;
        ln.eps = eps;
        ln.c1 = (const float *)c1;
        ln.xcopy = (__bf16 *)xc;
        ln.stats_out = (float *)sout;
        rc = launch_gemm_bf16_ln(epi, 0, A, W, (const float *)bias, (const float *)res, C, M, N, K, ln);
    }
    tune_gemm(keep);
    if (rc != SSW_OK) return rc;
    SSW_HIP_TRY(hipDeviceSynchronize());
    if (C && C_out_or_null) SSW_HIP_TRY(hipMemcpy(C_out_or_null, C, mn * (c_bf16 ? 2 : 4), hipMemcpyDeviceToHost));
    if (xc && epi >= 6) SSW_HIP_TRY(hipMemcpy(xcopy_inout_or_null, xc, mn * 2, hipMemcpyDeviceToHost));
    if (sout && stats_out_or_null) SSW_HIP_TRY(hipMemcpy(stats_out_or_null, sout, (size_t)M * n_tiles * 2 * 4, hipMemcpyDeviceToHost));
    return SSW_OK;
}

extern "C" ssw_status ssw_debug_attn_out_run(int32_t B, int32_t S, const uint16_t *qkv_bf16, const uint16_t *Wo_bf16,
                                             const float *bo, uint16_t *xcopy_inout, const float *res_in_or_null,
                                             float *res_out_or_null, float *stats_out, float scale) {
    using namespace ssw;
    SSW_REQUIRE(B > 0 && qkv_bf16 && Wo_bf16 && bo && xcopy_inout && stats_out, "ssw_debug_attn_out_run: NULL argument");
    SSW_REQUIRE((res_in_or_null == nullptr) == (res_out_or_null == nullptr), "f32 stream: both residual pointers");
    const int D = 768, H = 12;
    const size_t rows = (size_t)B * S;
    DevBufs d;
    void *qkv, *wo, *wo_pk, *b, *xc, *rin = nullptr, *rout = nullptr, *st;
    SSW_TRY(d.up(qkv_bf16, rows * 3 * D * 2, &qkv));
    SSW_TRY(d.up(Wo_bf16, (size_t)D * D * 2, &wo));
    SSW_TRY(d.up(nullptr, (size_t)D * D * 2, &wo_pk));
    SSW_TRY(d.up(bo, (size_t)D * 4, &b));
    SSW_TRY(d.up(xcopy_inout, rows * D * 2, &xc, res_in_or_null == nullptr));
    if (res_in_or_null) {
        SSW_TRY(d.up(res_in_or_null, rows * D * 4, &rin));
        SSW_TRY(d.up(nullptr, rows * D * 4, &rout));
    }
    SSW_TRY(d.up(nullptr, rows * 4 * 4, &st));
    SSW_TRY(pack_attn_outproj_weight(0, wo, wo_pk));
    SSW_TRY(launch_attn_outproj(0, qkv, wo_pk, (const float *)b, xc, (const float *)rin, (float *)rout, (float *)st, B, S, D, H, scale));
    SSW_HIP_TRY(hipDeviceSynchronize());
    SSW_HIP_TRY(hipMemcpy(xcopy_inout, xc, rows * D * 2, hipMemcpyDeviceToHost));
    if (rout) SSW_HIP_TRY(hipMemcpy(res_out_or_null, rout, rows * D * 4, hipMemcpyDeviceToHost));
    SSW_HIP_TRY(hipMemcpy(stats_out, st, rows * 4 * 4, hipMemcpyDeviceToHost));
    return SSW_OK;
}

extern "C" ssw_status ssw_debug_attn_out_stamps(uint64_t *out, int32_t n_words) {
    SSW_REQUIRE(out && n_words > 0 && n_words <= 32 * 1024, "ssw_debug_attn_out_stamps: bad arguments");
    return ssw::read_ao_stamps(out, n_words);
}
