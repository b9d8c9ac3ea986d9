// K2/K3/K8: nn.Linear as an exact-fp32 MFMA GEMM with fused bias / column scale / erf-GELU / residual epilogue.
//   Cout[m,n] = act((sum_k A[m,k] W[n,k] + bias[n]) * (n < scale_cols ? scale : 1)) + residual[m,n]
// Reference call sites: modules.py:760-766 (q,k proj + q scaling), :794,:799 (v, out proj), :896-905, :923 (column),
// :424-426 (fc1 + GELU, fc2), :396 (residual add).
//
// Roofline: MFMA-bound.  2*M*N*K flops against v_mfma_f32_32x32x2_f32's 157.3 TFLOP/s; per 128x128x32 K tile a CU
// issues 256 MFMAs (4096 cycles per SIMD) while 32 KB arrive from L2 (8 B/clk/CU).  Two blocks are resident per CU
// (73.7 KB LDS, <=128 VGPRs each) so one block's barrier / staging bubbles are covered by the other's MFMAs.
#include "mma_core.h"

namespace rnamsm {

constexpr int GEMM_LDS_BYTES = 2 * (TILE_KC + TILE_KC) * 4;   // double-buffered A and W tiles

template <int ACT, bool HAS_RES, bool ZROWS>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_f32_kernel(
    const float* __restrict__ A, int64_t lda, const float* __restrict__ W, const float* __restrict__ bias,
    const float* residual, int64_t ldr, float* Cout, int64_t ldc,
    int M, int N, int K, float scale, int scale_cols, const uint8_t* __restrict__ zero_rows, int group) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                    // [2][BM][LDK]
    float* Ws = smem + 2 * TILE_KC;      // [2][BN][LDK]

    const unsigned nb = N / BN, mp = (M + BM - 1) / BM;
    unsigned mpanel, nblk;
    if (!xcd_panel_map_grouped(blockIdx.x, mp, nb, (unsigned)group, mpanel, nblk)) return;
    const int m0 = mpanel * BM, n0 = nblk * BN;

    const WaveCoord w = wave_coord();
    const int c4 = threadIdx.x & 7, r0 = threadIdx.x >> 3;

    // per-thread global row pointers (A rows clamped: a clamped row only feeds its own discarded output row)
    const float* ap[4];
    const float* wp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + r0 + 32 * i;
        m = m < M ? m : M - 1;
        ap[i] = A + (int64_t)m * lda + c4 * 4;
        wp[i] = W + (int64_t)(n0 + r0 + 32 * i) * K + c4 * 4;
    }

    f32x16 acc[2][2];
    zero_acc(acc);

    StageKC sa[1], sw[1];
    pipelined_kloop<true, 8, 1>(
        K / BK, As, Ws, TILE_KC, TILE_KC, acc, w,
        [&](int kt, auto set) {
            constexpr int S = decltype(set)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sa[S].v[i] = *reinterpret_cast<const f32x4*>(ap[i] + kt * BK);
                sw[S].v[i] = *reinterpret_cast<const f32x4*>(wp[i] + kt * BK);
            }
        },
        [&](int buf, auto set) {
            constexpr int S = decltype(set)::value;
            stage_store_kc(As + buf * TILE_KC, sa[S]);
            stage_store_kc(Ws + buf * TILE_KC, sw[S]);
        });

    // ---- epilogue.  The accumulator layout (one row x 32 columns per register and lane half) would give 64
    // 4-byte-per-lane stores per wave, and store tails are issue-bound; instead each wave transposes its 64x64 tile
    // through its own slice of the (now idle) LDS and moves whole 256-B row segments: 16 float4 stores, and 16 float4
    // residual loads that are issued BEFORE the transpose so their latency hides behind it.
    constexpr int LDE = 64 + 4;                                   // padded row stride (floats) of the staging tile
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int er = lane >> 4, ec = (lane & 15) * 4;               // lane -> (row er + 4*i, columns ec..ec+3)
    const int gm0 = m0 + w.wm * 64, gn = n0 + w.wn * 64 + ec;
    f32x4 res[16];
    if (HAS_RES) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = min(gm0 + er + 4 * i, M - 1);
            res[i] = *reinterpret_cast<const f32x4*>(residual + (int64_t)row * ldr + gn);
        }
    }
    __syncthreads();                                              // every wave has finished reading operand tiles
    float* stage = smem + wv * (64 * LDE);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int col = n0 + acc_col(w, nt);
        const float b = bias ? bias[col] : 0.f;
        const float sc = col < scale_cols ? scale : 1.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                float v = (acc[mt][nt][t] + b) * sc;
                if (ACT == RNAMSM_ACT_GELU_ERF) v = gelu_erf(v);
                stage[(mt * 32 + (t & 3) + 8 * (t >> 2) + 4 * w.lh) * LDE + nt * 32 + w.li] = v;
            }
    }
    // same-wave LDS write -> read: ordered by the hardware queue, the compiler inserts the lgkmcnt wait
    // all 16 LDS reads first, then the stores; full tiles (the common case) carry no per-row bounds branch -- hipcc
    // otherwise sinks each read into its row's branch and serialises read -> wait -> store sixteen times
    f32x4 ov[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = er + 4 * i;
        ov[i] = *reinterpret_cast<const f32x4*>(&stage[r * LDE + ec]);
        if (HAS_RES) ov[i] += res[i];
        // f2: q *= 1 - padding_mask (modules.py:767-772): padded tokens get q = 0 (the scaled columns are q)
        if (ZROWS && gn < scale_cols && zero_rows[min(gm0 + r, M - 1)]) ov[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (m0 + BM <= M) {
#pragma unroll
        for (int i = 0; i < 16; ++i) *reinterpret_cast<f32x4*>(Cout + (int64_t)(gm0 + er + 4 * i) * ldc + gn) = ov[i];
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (gm0 + er + 4 * i < M) *reinterpret_cast<f32x4*>(Cout + (int64_t)(gm0 + er + 4 * i) * ldc + gn) = ov[i];
    }
}

template <int ACT, bool HAS_RES, bool ZROWS = false>
static int launch_gemm(const float* A, int64_t lda, const float* W, const float* bias, const float* residual,
                       int64_t ldr, float* Cout, int64_t ldc, int M, int N, int K, float scale, int scale_cols,
                       const uint8_t* zero_rows, hipStream_t stream) {
    static bool configured = false;
    auto kern = gemm_f32_kernel<ACT, HAS_RES, ZROWS>;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
        if (e != hipSuccess) return fail(RNAMSM_ERR_HIP, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
        configured = true;
    }
    // Block order (speed/traffic only): 64 blocks are resident per XCD.  With more than 8 column blocks per row panel a
    // whole-panel order keeps only 64/nb panels in flight and re-streams W (7-9 MB > the 4 MB L2) for each of them;
    // groups of 8 panels x 8 column blocks halve the fabric reads (PMC, cfg3: QKV 4.9 -> 2.9 GB, fc1 8.0 -> 3.6 GB per
    // launch; same speed, the kernel is MFMA-bound).  N = 768 (6 column blocks) is already balanced and stays ungrouped.
    const int group = tuning().gemm_group > 0 ? tuning().gemm_group : (N / BN > 8 ? 8 : 1);
    const unsigned grid = xcd_panel_grid_grouped((M + BM - 1) / BM, N / BN, (unsigned)group);
    // algorithmic work: 2MNK flops; bytes = A + W + C once (+ residual read)
    KernelTimer timer(TC_GEMM, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N * (HAS_RES ? 2 : 1)), stream);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(GEMM_THREADS), GEMM_LDS_BYTES, stream, A, lda, W, bias, residual, ldr,
                       Cout, ldc, M, N, K, scale, scale_cols, zero_rows, group);
    RNAMSM_CHECK_LAUNCH("gemm_f32");
    return RNAMSM_OK;
}

}  // namespace rnamsm

using namespace rnamsm;

extern "C" int rnamsm_gemm_bias_act_res(const float* A, int64_t lda, const float* W, const float* bias,
                                        const float* residual, int64_t ldr, float* Cout, int64_t ldc, int64_t M,
                                        int N, int K, int act, float scale, int scale_cols, const uint8_t* zero_rows,
                                        int dtype, void* stream) {
    if (dtype != RNAMSM_F32) return fail(RNAMSM_ERR_UNSUPPORTED, "gemm: only RNAMSM_F32 is implemented");
    RNAMSM_CHECK_ARG(A && W && Cout, "gemm: null pointer");
    RNAMSM_CHECK_ARG(M > 0 && M <= INT32_MAX && N > 0 && K > 0, "gemm: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
    RNAMSM_CHECK_ARG(N % BN == 0 && K % BK == 0, "gemm: need N %% 128 == 0 and K %% 32 == 0 (N=%d K=%d)", N, K);
    RNAMSM_CHECK_ARG(lda >= K && lda % 4 == 0 && ldc >= N, "gemm: bad leading dimension lda=%lld ldc=%lld",
                     (long long)lda, (long long)ldc);
    RNAMSM_CHECK_ARG(aligned16(A) && aligned16(W), "gemm: A and W must be 16-byte aligned");
    RNAMSM_CHECK_ARG(!residual || (ldr >= N && ldr % 4 == 0 && aligned16(residual)), "gemm: bad residual stride/alignment");
    RNAMSM_CHECK_ARG(ldc % 4 == 0 && aligned16(Cout), "gemm: Cout must be 16-byte aligned with ldc %% 4 == 0");
    RNAMSM_CHECK_ARG(act == RNAMSM_ACT_NONE || act == RNAMSM_ACT_GELU_ERF, "gemm: unknown activation %d", act);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int m = (int)M;
#define RNAMSM_GEMM_DISPATCH(ACT_, RES_) \
    launch_gemm<ACT_, RES_>(A, lda, W, bias, residual, ldr, Cout, ldc, m, N, K, scale, scale_cols, zero_rows, s)
    if (zero_rows) {   // f2: only the QKV projection of row attention uses it (no activation, no residual)
        RNAMSM_CHECK_ARG(act == RNAMSM_ACT_NONE && !residual, "gemm: zero_rows is supported without activation / residual");
        return launch_gemm<RNAMSM_ACT_NONE, false, true>(A, lda, W, bias, residual, ldr, Cout, ldc, m, N, K, scale,
                                                         scale_cols, zero_rows, s);
    }
    if (act == RNAMSM_ACT_GELU_ERF) return residual ? RNAMSM_GEMM_DISPATCH(RNAMSM_ACT_GELU_ERF, true) : RNAMSM_GEMM_DISPATCH(RNAMSM_ACT_GELU_ERF, false);
    return residual ? RNAMSM_GEMM_DISPATCH(RNAMSM_ACT_NONE, true) : RNAMSM_GEMM_DISPATCH(RNAMSM_ACT_NONE, false);
#undef RNAMSM_GEMM_DISPATCH
}
