// gemm_pw4: persistent bf16 GEMM for the CLIP towers,  C[M,N] = A[M,K] * W[N,K]^T (+ fused epilogue), f32 accumulate.
//
// Round 3 structure.  Measured first (tools/micro/mfma_rate.hip, profiles/r03_gemm_notes.txt): one wave
// per SIMD issuing v_mfma_f32_16x16x32_bf16 from inline asm with the accumulators tied in place runs the matrix pipes at
// 2.1 PFLOP/s (2.16 GHz; the builtin form reaches 1.57: hipcc renames every result and shuffles accumulators), and an
// LDS-DMA piece takes ~2 us from issue to landed when every CU streams its operands (4 000 cycles; 250-400 on an idle
// chip).  A 256 x 256 tile needs 64 KB per 64-deep K-step, i.e. ~1 us of MFMA work per 64 KB: with a two-stage ring
// (first version of this file: <= 64 KB in flight, a K-step took 3.3 us against 1.0 of MFMA issue) the loop is bound by
// that latency, not by bandwidth, LDS reads or the matrix pipes.  Hence:
//   * ONE workgroup per CU, four waves, one per SIMD; a wave owns 128 x (BN / 2) of a 256 x BN macro tile (BN = 256 /
//     192 / 128: accumulators 256 / 192 / 128 registers of the SIMD's 512): 0.25 fragment reads per MFMA and the fewest
//     L2 -> LDS bytes per flop a CU's 160 KB of LDS allows;
//   * the operands travel in 32-deep SUB-STAGES (256 + BN rows x 64 B) through a ring of D = 5 (6 for BN = 128) slots
//     that fills the whole LDS: sub-stage u + D is requested as soon as sub-stage u's fragments are in registers, so
//     three to four sub-stages (96-128 KB per CU) are in flight at any time; one raw s_barrier per sub-stage behind a
//     COUNTED s_waitcnt vmcnt that leaves the D - 2 younger sub-stages in flight;
//   * the fragments of the next sub-stage are read while the current one multiplies (two register sets); LDS-DMA pieces,
//     fragment reads and MFMAs are interleaved one piece / one read per four MFMAs, pinned with sched_barrier;
//   * the MFMAs are inline asm with tied accumulators (see pw_mfma) and the K loop has no branch: the load cursor
//     crosses tile boundaries by scalar selects;
//   * PERSISTENT: a workgroup walks its tiles (ids g, g + G, ...) as one stream of sub-stages, so the next tile's first
//     sub-stages are in flight / in registers while the current tile's epilogue stores drain;
//   * tile ids that share an XCD (id % 8) walk the column tiles of whole row tiles.
// LDS image of a sub-stage: rows of 64 B (four 16-byte chunks); chunk c of row r sits at chunk position c ^ f(r),
// f(r) = (4 - (r >> 2)) & 3 -- for ds_read_b128's lane groups ({0-3, 12-15, 20-27}, ...) the 16 lanes of a group then hit
// 16 distinct 16-byte slots of the 256-byte bank row.  The permutation is applied to the SOURCE address of the LDS-DMA
// (the destination is lane-linear) and again by the fragment reads.
// Replaces (with gemm_bf16.hip) the torch / cuBLAS linears inside transformers' CLIPModel that the reference calls
// (seesaw/models/model.py:50-57, seesaw/models/embeddings.py:433-455).
#include "ssw_common.h"

namespace ssw {
namespace {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
enum Epilogue { EPI_F32 = 0, EPI_BF16_BIAS = 1, EPI_BF16_BIAS_GELU = 2, EPI_F32_BIAS_RESIDUAL = 3 };

constexpr int PW_BM = 256, PW_BKS = 32;   // rows of a tile; depth of a sub-stage
constexpr int PW_ASUB = PW_BM * 64;        // A image of one sub-stage: 256 rows x 64 B

// acc += W-fragment x A-fragment, accumulating IN PLACE in the accumulator file.  Through the builtin hipcc renames
// every MFMA result (destination != source in more than half of them) and shuffles the 256 accumulators around with
// v_accvgpr_read / write / mov inside the K loop; the tied "+a" operand leaves it no choice.  The operands come from
// ds_read (the compiler's lgkmcnt bookkeeping covers asm inputs); an accumulate chain on the same registers needs no
// wait states; readers of the accumulators other than the next MFMA are fenced by pw_mfma_drain() below.
__device__ __forceinline__ void pw_mfma(f32x4 &acc, const bf16x8 &w, const bf16x8 &a) {
    asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
}
// MFMA result -> any other reader / writer of those registers: the hazard hipcc would pad for its own MFMAs
__device__ __forceinline__ void pw_mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

template <int N>
__device__ __forceinline__ void pw_wait_vmcnt() {
    static_assert(N == 0 || N == 21 || N == 24 || N == 28 || N == 30 || N == 32, "add the literal");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (N == 21) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
    if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    if constexpr (N == 28) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
    if constexpr (N == 30) asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
    if constexpr (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
}

// tile id -> (row tile, column tile); ids t and t + 8 share an XCD, XCD x takes row tiles x, x + 8, ...
__device__ __forceinline__ bool pw_tile(int t, int m_tiles, int n_tiles, int tiles_all, int *mt, int *nt) {
    if (t >= tiles_all) return false;
    const int idx = t >> 3;
    *mt = (idx / n_tiles) * 8 + (t & 7);
    *nt = idx % n_tiles;
    return *mt < m_tiles;
}

// MODE: 0 the kernel; diagnostics (tools/perf_gemm.py --pw-diag, wrong results except 1): 1 = cycle stamps around the
// wait + barrier, 2 = no LDS-DMA after the prologue, 3 = no MFMA, 4 = no fragment reads after the prologue
__device__ unsigned long long g_pw_diag[6];  // sum over waves: cycles in wait + barrier, cycles in sub-steps, sub-steps, waves,
                                             // s_memtime ticks of the kernel, s_memrealtime ticks (100 MHz) of the kernel
__device__ unsigned long long g_pw_wg[1024][4];  // MODE 1, per workgroup: start / end (100-MHz ticks), HW_ID, XCC_ID
__device__ unsigned long long g_pw_grp[20];      // MODE 5: cycles per interleave group (16), wait + barrier, bodies

template <int BN>
struct PwGeom {
    static constexpr int NJ = BN / 32;               // 16-column accumulator blocks per wave (the wave owns BN / 2 columns)
    static constexpr int SUB = PW_ASUB + BN * 64;    // bytes of one sub-stage
    static constexpr int D = (160 * 1024) / SUB > 6 ? 6 : (160 * 1024) / SUB;  // ring slots
    static constexpr int NWP = BN / 64;              // W pieces (16 rows x 64 B) per wave and sub-stage
    static constexpr int NL = 4 + NWP;               // LDS-DMA pieces per wave and sub-stage
    static constexpr int LDS = D * SUB;
};

template <int EPI, int BN, int MODE = 0>
__global__ __launch_bounds__(256) void gemm_pw4(const bf16 *__restrict__ A, const bf16 *__restrict__ W,
                                                const float *__restrict__ bias, const float *__restrict__ residual,
                                                void *__restrict__ Cout, int M, int N, int K, int m_tiles, int n_tiles,
                                                int tiles_all) {
    using Gm = PwGeom<BN>;
    constexpr int NJ = Gm::NJ, SUB = Gm::SUB, D = Gm::D, NWP = Gm::NWP, NL = Gm::NL;
    constexpr int NF = 8 + NJ;   // fragment reads per wave and sub-stage
    constexpr int MPG = NJ / 2;  // MFMAs per interleave group (16 groups per sub-stage)
    static_assert(NL * D <= 63 && NF <= 16, "vmcnt is six bits; one read per group");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int G = gridDim.x;
    const int nks = K / PW_BKS;  // sub-stages per tile (even: K % 64 == 0)

    // ---- this workgroup's tiles: ids g, g + G, ... (invalid ids -- row tiles past the matrix -- are skipped)
    int my_tiles = 0;
    for (int t = blockIdx.x; t < tiles_all; t += G) {
        int a, b;
        my_tiles += pw_tile(t, m_tiles, n_tiles, tiles_all, &a, &b) ? 1 : 0;
    }
    if (my_tiles == 0) return;

    // ---- load cursor: the sub-stage to request next = sub-stage l_ks of the tile whose A rows start at l_m0 and whose W
    // rows start at l_wbase; its ring slot is l_slot.  It runs D sub-stages ahead of the multiplication, so it crosses
    // into the next tile once per computed tile: the next tile's values wait in n_* and are swapped in by scalar selects
    // -- the K loop has no branch (hipcc's register allocation of 256 accumulators + two fragment sets does not survive
    // control flow there).  A piece is 16 rows x 64 B; lane l lands at row l / 4, chunk position l % 4 and fetches chunk
    // (l % 4) ^ f(row), f = (4 - (l >> 4)) & 3: one VGPR serves every piece of both operands, the rest of the address is
    // scalar (saddr form of the LDS-DMA).  A piece that would start past the last 16 rows of A is redirected to them
    // (M % 16 == 0: no piece straddles M; its products are never stored).
    const unsigned voff = (unsigned)((lane >> 2) * K * 2 + (((lane & 3) ^ ((4 - (lane >> 4)) & 3)) << 4));
    int l_ks = 0, l_slot = 0, l_m0, n_m0;
    const bf16 *l_wbase, *n_wbase;
    bool in_loop = false;  // (diagnostic MODE 2 / 4: the prologue still loads and reads)
    unsigned long long d_wait = 0, d_steps = 0, d_n = 0, d_t1 = 0;
    const unsigned long long d_c0 = MODE == 1 ? clock64() : 0, d_r0 = MODE == 1 ? wall_clock64() : 0;
    unsigned long long d_ts[17], d_grp[18];
    if (MODE == 5)
        for (int g = 0; g < 18; ++g) d_grp[g] = 0;
    unsigned long long d_prev_end = 0;
    int c_tile = blockIdx.x, c_mt, c_nt;
    while (!pw_tile(c_tile, m_tiles, n_tiles, tiles_all, &c_mt, &c_nt)) c_tile += G;
    l_m0 = n_m0 = c_mt * PW_BM + wave * 64;
    l_wbase = n_wbase = W + (int64_t)(c_nt * BN + wave * (NWP * 16)) * K;
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char *)smem);
    const unsigned a_dst = lds0 + wave * 4096;
    const unsigned w_dst = lds0 + PW_ASUB + wave * (NWP * 1024);
#define PW_ISSUE_A(p)                                                                                           \
    if (MODE != 2 || !in_loop)                                                                                  \
    glds16s(voff, A + (int64_t)min(l_m0 + (p) * 16, M - 16) * K + l_ks * PW_BKS, a_dst + l_slot * SUB + (p) * 1024)
#define PW_ISSUE_W(p)                                                                                           \
    if (MODE != 2 || !in_loop)                                                                                  \
    glds16s(voff, l_wbase + (int64_t)((p) * 16) * K + l_ks * PW_BKS, w_dst + l_slot * SUB + (p) * 1024)
#define PW_ADVANCE()                                                                                            \
    {                                                                                                           \
        const bool wrap = (l_ks + 1 == nks);                                                                    \
        l_ks = wrap ? 0 : l_ks + 1;                                                                             \
        l_m0 = wrap ? n_m0 : l_m0;                                                                              \
        l_wbase = wrap ? n_wbase : l_wbase;                                                                     \
        l_slot = (l_slot + 1 == D) ? 0 : l_slot + 1;                                                            \
    }

    // ---- fragments: lane (fr, fq) reads row fr of a 16-row block, chunk fq at position fq ^ f(fr)
    const int fr = lane & 15, fq = lane >> 4;
    const int frag0 = fr * 64 + ((fq ^ ((4 - (fr >> 2)) & 3)) << 4);
    const unsigned char *a_rd0 = smem + wr * 8192 + frag0;                        // + slot * SUB + i * 1024
    const unsigned char *w_rd0 = smem + PW_ASUB + wc * (NJ * 1024) + frag0;       // + slot * SUB + j * 1024
    int r_slot = 0;  // ring slot of the sub-stage whose fragments are read next
#define PW_RD_A(i) (*reinterpret_cast<const bf16x8 *>(a_rd + (i) * 1024))
#define PW_RD_W(j) (*reinterpret_cast<const bf16x8 *>(w_rd + (j) * 1024))

    f32x4 acc[8][NJ];
    bf16x8 fa0[8], fb0[NJ], fa1[8], fb1[NJ];

    // kernel arguments into registers before the loops (a scalar load pending at loop entry degrades the counted waits)
    asm volatile("" ::"s"(bias), "s"(residual), "s"(Cout), "s"(N), "s"(M));

    // ---- prologue: the first D sub-stages of the stream, then the fragments of sub-stage 0
#pragma unroll
    for (int u = 0; u < D; ++u) {
#pragma unroll
        for (int p = 0; p < 4; ++p) { PW_ISSUE_A(p); }
#pragma unroll
        for (int p = 0; p < NWP; ++p) { PW_ISSUE_W(p); }
        PW_ADVANCE()
    }
    pw_wait_vmcnt<NL *(D - 1)>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
        const unsigned char *a_rd = a_rd0, *w_rd = w_rd0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb0[j] = PW_RD_W(j);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa0[i] = PW_RD_A(i);
    }
    r_slot = 1;

    // One sub-stage: every wave has seen its own pieces of the NEXT sub-stage land (the D - 2 younger ones stay in
    // flight) and has this sub-stage's fragments in registers; behind the barrier it multiplies them while it reads the
    // next sub-stage's fragments into the other register set and requests sub-stage + D into the slot just vacated.
    // (After a tile's epilogue its stores are the youngest entries of the vmcnt queue: the counted wait is then
    // stricter than needed for a few sub-stages, never too lax.)
#define PW_SUBSTEP(FA, FB, FA_N, FB_N)                                                                             \
    {                                                                                                              \
        __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0): this sub-stage's fragments are in registers */          \
        if (MODE == 1) {                                                                                           \
            const unsigned long long t1 = clock64();                                                               \
            if (d_t1) d_steps += t1 - d_t1, ++d_n;                                                                 \
            d_t1 = t1;                                                                                             \
        }                                                                                                          \
        pw_wait_vmcnt<NL *(D - 2)>();                                                                              \
        __builtin_amdgcn_s_barrier();                                                                              \
        asm volatile("" ::: "memory");                                                                             \
        if (MODE == 1) d_wait += clock64() - d_t1;                                                                 \
        const unsigned char *a_rd = a_rd0 + r_slot * SUB, *w_rd = w_rd0 + r_slot * SUB;                            \
        _Pragma("unroll") for (int g = 0; g < 16; ++g) {                                                           \
            if (MODE == 5) asm volatile("s_memtime %0" : "=s"(d_ts[g]));                                           \
            /* fragment reads two per group in the first groups (their latency has the rest of the sub-stage),  */ \
            /* LDS-DMA pieces one per group in the last ones                                                     */ \
            if (MODE != 4) {                                                                                       \
                _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                    \
                    const int f = 2 * g + h;                                                                       \
                    if (f < NJ) FB_N[f] = PW_RD_W(f);                                                              \
                    else if (f < NF) FA_N[f - NJ] = PW_RD_A(f - NJ);                                               \
                }                                                                                                  \
            }                                                                                                      \
            if (g >= 16 - NL) {                                                                                    \
                const int pc = g - (16 - NL);                                                                      \
                if (pc < 4) { PW_ISSUE_A(pc); }                                                                    \
                else { PW_ISSUE_W(pc - 4); }                                                                       \
            }                                                                                                      \
            _Pragma("unroll") for (int m = 0; m < MPG; ++m) {                                                      \
                const int q = g * MPG + m, i = q / NJ, j = q % NJ;                                                 \
                if (MODE != 3) pw_mfma(acc[i][j], FB[j], FA[i]);                                                   \
            }                                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
        }                                                                                                          \
        if (MODE == 5) {                                                                                           \
            asm volatile("s_memtime %0" : "=s"(d_ts[16]));                                                         \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
            _Pragma("unroll") for (int g = 0; g < 16; ++g) d_grp[g] += d_ts[g + 1] - d_ts[g];                       \
            if (d_prev_end) d_grp[16] += d_ts[0] - d_prev_end;                                                     \
            d_prev_end = d_ts[16];                                                                                 \
            d_grp[17] += 1;                                                                                        \
        }                                                                                                          \
        PW_ADVANCE()                                                                                               \
        r_slot = (r_slot + 1 == D) ? 0 : r_slot + 1;                                                               \
    }

    // ---- the stream: tile after tile
    for (int done = 0; done < my_tiles; ++done) {
        const int m0 = c_mt * PW_BM, n0 = c_nt * BN;
        // the tile after this one (the last tile stands in for itself: the stream's surplus requests re-load bytes that
        // are already in place)
        if (done + 1 < my_tiles) {
            c_tile += G;
            while (!pw_tile(c_tile, m_tiles, n_tiles, tiles_all, &c_mt, &c_nt)) c_tile += G;
        }
        n_m0 = c_mt * PW_BM + wave * 64;
        n_wbase = W + (int64_t)(c_nt * BN + wave * (NWP * 16)) * K;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        pw_mfma_drain();  // (the zeroing writes above -> the first MFMA reading them as C)
        in_loop = true;
        for (int ks = 0; ks < nks; ks += 2) {
            PW_SUBSTEP(fa0, fb0, fa1, fb1)
            PW_SUBSTEP(fa1, fb1, fa0, fb0)
        }
        pw_mfma_drain();
        // epilogue: acc[i][j][r] = C[m0 + wr*128 + i*16 + fr][n0 + wc*BN/2 + j*16 + fq*4 + r].  The bias comes through
        // the scalar cache (16 floats per column block, the lane picks its four by fq): an ordinary vector load here would
        // make the compiler wait for every LDS-DMA piece in flight, which its vmcnt bookkeeping does not see.
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int colb = n0 + wc * (BN / 2) + j * 16;
            f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
            if (EPI != EPI_F32) {
                const f32x4 *bp = reinterpret_cast<const f32x4 *>(bias + colb);  // wave-uniform address
                const f32x4 b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
                bv = fq == 0 ? b0 : fq == 1 ? b1 : fq == 2 ? b2 : b3;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = m0 + wr * 128 + i * 16 + fr;
                if (row < M) {
                    const int64_t o = (int64_t)row * N + colb + fq * 4;
                    f32x4 v = acc[i][j] + bv;
                    if (EPI == EPI_BF16_BIAS_GELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)  // quick_gelu: x * sigmoid(1.702 x), v_exp + v_rcp
                            v[r] *= __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.702f * 1.44269504f * v[r]));
                    }
                    if (EPI == EPI_F32_BIAS_RESIDUAL) v += *reinterpret_cast<const f32x4 *>(residual + o);
                    if (EPI == EPI_BF16_BIAS || EPI == EPI_BF16_BIAS_GELU) {
                        bf16x4 h;
#pragma unroll
                        for (int r = 0; r < 4; ++r) h[r] = (bf16)v[r];
                        *reinterpret_cast<bf16x4 *>(reinterpret_cast<bf16 *>(Cout) + o) = h;
                    } else {
                        *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(Cout) + o) = v;
                    }
                }
            }
        }
    }
    pw_wait_vmcnt<0>();  // the surplus requests of the stream's end
    if (MODE == 5 && lane == 0)
        for (int g = 0; g < 18; ++g) atomicAdd(&g_pw_grp[g], d_grp[g]);
    if (MODE == 1 && lane == 0) {
        atomicAdd(&g_pw_diag[0], d_wait);
        atomicAdd(&g_pw_diag[1], d_steps);
        atomicAdd(&g_pw_diag[2], d_n);
        atomicAdd(&g_pw_diag[3], 1ull);
        atomicAdd(&g_pw_diag[4], (unsigned long long)(clock64() - d_c0));
        atomicAdd(&g_pw_diag[5], (unsigned long long)(wall_clock64() - d_r0));
        if (wave == 0 && blockIdx.x < 1024) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
            g_pw_wg[blockIdx.x][0] = d_r0;
            g_pw_wg[blockIdx.x][1] = wall_clock64();
            g_pw_wg[blockIdx.x][2] = hw;
            g_pw_wg[blockIdx.x][3] = xcc;
        }
    }
#undef PW_SUBSTEP
#undef PW_ISSUE_A
#undef PW_ISSUE_W
#undef PW_ADVANCE
#undef PW_RD_A
#undef PW_RD_W
}

int g_pw_mode = 0;

template <int EPI, int BN, int MODE = 0>
ssw_status launch_pw4_bn(hipStream_t s, const bf16 *A, const bf16 *W, const float *bias, const float *res, void *C, int M,
                         int N, int K) {
    if constexpr (MODE == 0 && ((EPI == EPI_BF16_BIAS_GELU && BN == 256) || (EPI == EPI_F32 && BN == 128))) {
        switch (g_pw_mode) {  // diagnostic builds exist for these two instantiations only
            case 1: return launch_pw4_bn<EPI, BN, 1>(s, A, W, bias, res, C, M, N, K);
            case 2: return launch_pw4_bn<EPI, BN, 2>(s, A, W, bias, res, C, M, N, K);
            case 3: return launch_pw4_bn<EPI, BN, 3>(s, A, W, bias, res, C, M, N, K);
            case 4: return launch_pw4_bn<EPI, BN, 4>(s, A, W, bias, res, C, M, N, K);
            case 5: return launch_pw4_bn<EPI, BN, 5>(s, A, W, bias, res, C, M, N, K);
        }
    }
    constexpr int lds = PwGeom<BN>::LDS;  // the ring
    static bool attr_set[64] = {};
    int dev = 0;
    SSW_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        SSW_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pw4<EPI, BN, MODE>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const int m_tiles = (M + PW_BM - 1) / PW_BM, n_tiles = N / BN;
    const int tiles_all = ((m_tiles + 7) / 8) * 8 * n_tiles;
    int grid = num_cus(dev) / 8 * 8;
    if (grid <= 0) grid = 256;
    if (grid > tiles_all) grid = tiles_all;  // tiles_all is a multiple of 8
    hipLaunchKernelGGL((gemm_pw4<EPI, BN, MODE>), dim3(grid), dim3(256), lds, s, A, W, bias, res, C, M, N, K, m_tiles,
                       n_tiles, tiles_all);
    SSW_HIP_TRY(hipGetLastError());
    return SSW_OK;
}

template <int EPI>
ssw_status launch_pw4_epi(hipStream_t s, const bf16 *A, const bf16 *W, const float *bias, const float *res, void *C, int M,
                          int N, int K, int bn) {
    switch (bn) {
        case 256: return launch_pw4_bn<EPI, 256>(s, A, W, bias, res, C, M, N, K);
        case 192: return launch_pw4_bn<EPI, 192>(s, A, W, bias, res, C, M, N, K);
        case 128: return launch_pw4_bn<EPI, 128>(s, A, W, bias, res, C, M, N, K);
    }
    set_error("gemm_pw4: column tile %d unknown", bn);
    return SSW_ERR_INVALID;
}

}  // namespace

void gemm_pw4_set_mode(int mode) { g_pw_mode = mode; }
ssw_status gemm_pw4_read_wg(unsigned long long *out /* [1024][4] then [20] group counters (read and reset) */) {
    SSW_HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pw_wg), sizeof(unsigned long long) * 4096));
    SSW_HIP_TRY(hipMemcpyFromSymbol(out + 4096, HIP_SYMBOL(g_pw_grp), sizeof(unsigned long long) * 20));
    unsigned long long z[20] = {};
    SSW_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_pw_grp), z, sizeof(z)));
    return SSW_OK;
}
ssw_status gemm_pw4_read_diag(unsigned long long out[6], bool reset) {
    SSW_HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pw_diag), 6 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long z[6] = {0, 0, 0, 0, 0, 0};
        SSW_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_pw_diag), z, sizeof(z)));
    }
    return SSW_OK;
}

// bn = 0: pick the column tile (fewest idle tile slots on this device's CUs)
bool gemm_pw4_supports(int M, int N, int K) { return M >= 16 && M % 16 == 0 && K % 64 == 0 && K >= 64 * 6 && N % 128 == 0; }

ssw_status launch_gemm_pw4(int epi, hipStream_t s, const void *A_, const void *W_, const float *bias, const float *res,
                           void *C, int M, int N, int K, int bn) {
    if (!gemm_pw4_supports(M, N, K)) {
        set_error("gemm_pw4: shape M=%d N=%d K=%d unsupported (M %% 16, N %% 128, K %% 64, K >= 384)", M, N, K);
        return SSW_ERR_UNSUPPORTED;
    }
    if (bn == 0) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const int cus = num_cus(dev) > 0 ? num_cus(dev) : 256;
        double best = 1e30;
        for (int cand : {256, 192, 128}) {
            if (N % cand) continue;
            const int64_t tiles = (int64_t)((M + PW_BM - 1) / PW_BM) * (N / cand);
            const int64_t rounds = (tiles + cus - 1) / cus;
            // time ~ rounds x (tile work + a fixed per-tile cost of about a quarter of a 256-wide tile's K loop at K = 768)
            const double cost = (double)rounds * (cand + 48.0);
            if (cost < best) {
                best = cost;
                bn = cand;
            }
        }
    }
    if (N % bn != 0) {
        set_error("gemm_pw4: N=%d is not a multiple of the column tile %d", N, bn);
        return SSW_ERR_UNSUPPORTED;
    }
    const bf16 *A = static_cast<const bf16 *>(A_), *W = static_cast<const bf16 *>(W_);
    switch (epi) {
        case EPI_F32: return launch_pw4_epi<EPI_F32>(s, A, W, bias, res, C, M, N, K, bn);
        case EPI_BF16_BIAS: return launch_pw4_epi<EPI_BF16_BIAS>(s, A, W, bias, res, C, M, N, K, bn);
        case EPI_BF16_BIAS_GELU: return launch_pw4_epi<EPI_BF16_BIAS_GELU>(s, A, W, bias, res, C, M, N, K, bn);
        case EPI_F32_BIAS_RESIDUAL: return launch_pw4_epi<EPI_F32_BIAS_RESIDUAL>(s, A, W, bias, res, C, M, N, K, bn);
    }
    set_error("gemm_pw4: unknown epilogue %d", epi);
    return SSW_ERR_INVALID;
}

}  // namespace ssw
