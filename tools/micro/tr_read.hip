// ds_read_b64_tr_b16 semantics check (development aid): per 16-lane group, lane 4q+p supplies the address of row q,
// columns 4p..4p+3 of a 4 x 16 block of 16-bit elements; lane i receives column i, row e in element e.
// hipcc --offload-arch=gfx950 -O2 tr_read.hip -o tr_read && ./tr_read
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(const short *in, short *out) {
    __shared__ __attribute__((aligned(16))) short s[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) s[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, l = lane & 15, q = l >> 2, p = l & 3;
    auto ptr = (__attribute__((address_space(3))) s16x4 *)(&s[(4 * g + q) * 64 + 4 * p]);
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
    short h[64 * 64], o[256];
    for (int r = 0; r < 64; ++r)
        for (int c = 0; c < 64; ++c) h[r * 64 + c] = (short)(r * 64 + c);
    short *din, *dout;
    hipMalloc(&din, sizeof(h));
    hipMalloc(&dout, sizeof(o));
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 4; ++e) {
            const int g = lane >> 4, i = lane & 15;
            const int want = (4 * g + e) * 64 + i;  // row 4g+e, column i
            if (o[lane * 4 + e] != want) ++bad;
        }
    printf("lane 0: %d %d %d %d | lane 1: %d %d %d %d | lane 17: %d %d %d %d | mismatches %d\n", o[0], o[1], o[2], o[3],
           o[4], o[5], o[6], o[7], o[68], o[69], o[70], o[71], bad);
    return bad != 0;
}
