// Linear GEMM on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA rate), fp32 accumulate, same
// fused epilogue as gemm_f32.hip.  Two operand modes, both with fp32 activations in HBM:
//   SPLIT = 1  "bf16"   : A and W rounded to bf16 (the mixed-precision mode of BASELINE config 4; error ~2^-9 per operand)
//   SPLIT = 3  "bf16x3" : a = a_hi + a_lo, w = w_hi + w_lo with hi = bf16(x), lo = bf16(x - hi); the product is
//                         a_hi w_hi + a_hi w_lo + a_lo w_hi (3 MFMAs), dropping a_lo w_lo: operands carry ~17 bits
//                         (relative error <= ~2^-17 per product), accumulation is fp32.  An opt-in fast mode: the
//                         exact-fp32 kernel (gemm_f32.hip) stays the default and the reference for parity.
// W is pre-split once into bf16 planes (rnamsm_split_bf16); A is split on the fly while it is staged into LDS
// (v_cvt_pk_bf16_f32), so no extra HBM pass exists for activations.
//
// Tile: 128x128 block, 4 waves (2x2), wave 64x64 = 2x2 MFMA tiles, K tile = 64 bf16.  An LDS plane is [128 rows][64 bf16]
// with the row stride padded to 144 B -- byte-for-byte the layout of the fp32 kernel's [128][32 f32] tile, so the
// conflict-free ds_read_b128 pattern carries over: 16 B at row*144 + 32*kk + 16*half = k 16kk + 8*half + j, exactly
// the operand map of the 32x32x16 MFMA (lane (r, half) holds k = 8*half + j).
#include "common.h"

namespace rnamsm {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 16-bit operand format: FMT 0 = bf16 (8-bit mantissa, fp32 range), FMT 1 = fp16 (11-bit mantissa, |x| < 65504).
// An fp16 hi/lo pair carries ~22 mantissa bits (fp32: 24): "f16x3" is fp32-grade arithmetic at the bf16 MFMA rate for
// operands inside fp16 range -- true for this model's GEMM inputs (LayerNorm outputs, attention contexts, GELU
// activations, 0.04-scale weights); values below 2^-24 * 2^11 of an element's magnitude fall into fp16 subnormals of
// the lo plane, an ABSOLUTE error <= 3e-8 per element.
template <int FMT> struct Half16;
template <> struct Half16<0> {
    typedef __bf16 T;
    typedef __bf16 V8 __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Half16<1> {
    typedef _Float16 T;
    typedef _Float16 V8 __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

constexpr int HB_BM = 128, HB_BN = 128, HB_BK = 64;
constexpr int HB_LDB = HB_BK * 2 + 16;             // bytes per LDS row (64 bf16 + 16 B pad)
constexpr int HB_PLANE = 128 * HB_LDB;             // bytes per operand plane
constexpr int HB_THREADS = 256;

template <int SPLIT>
struct HbCfg {
    static constexpr int NPL = SPLIT == 3 ? 2 : 1;             // planes per operand (hi [, lo])
    static constexpr int BUF = 2 * NPL * HB_PLANE;             // A planes then W planes
    static constexpr int LDS = 2 * BUF;                        // double buffered
    static constexpr int NM = 4 * (SPLIT == 3 ? 3 : 1);        // MFMAs per k16 group per wave
};

template <int SPLIT, int FMT>
struct HbFrag {
    typename Half16<FMT>::V8 a[HbCfg<SPLIT>::NPL][2], b[HbCfg<SPLIT>::NPL][2];
};

template <int SPLIT, int FMT>
__device__ __forceinline__ void hb_frag_load(const char* buf, int kk, int wm, int wn, int li, int lh, HbFrag<SPLIT, FMT>& f) {
    typedef typename Half16<FMT>::V8 bf16x8;
    constexpr int NPL = HbCfg<SPLIT>::NPL;
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f.a[p][t] = *reinterpret_cast<const bf16x8*>(buf + p * HB_PLANE + (wm * 64 + t * 32 + li) * HB_LDB + kk * 32 + 16 * lh);
            f.b[p][t] = *reinterpret_cast<const bf16x8*>(buf + (NPL + p) * HB_PLANE + (wn * 64 + t * 32 + li) * HB_LDB + kk * 32 + 16 * lh);
        }
    }
}

template <int SPLIT, int FMT>
__device__ __forceinline__ void hb_frag_mma(const HbFrag<SPLIT, FMT>& f, f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            if (SPLIT == 3) {      // small cross terms first, the leading term last
                acc[mt][nt] = Half16<FMT>::mfma(f.a[1][mt], f.b[0][nt], acc[mt][nt]);
                acc[mt][nt] = Half16<FMT>::mfma(f.a[0][mt], f.b[1][nt], acc[mt][nt]);
            }
            acc[mt][nt] = Half16<FMT>::mfma(f.a[0][mt], f.b[0][nt], acc[mt][nt]);
        }
}

// 8 consecutive f32 -> 8 halves (hi) and the halves of the remainders (lo)
template <int FMT>
__device__ __forceinline__ void split8(const f32x4& x, const f32x4& y, typename Half16<FMT>::V8& hi,
                                       typename Half16<FMT>::V8& lo, bool want_lo) {
    typedef typename Half16<FMT>::T H;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        hi[i] = (H)x[i];
        hi[4 + i] = (H)y[i];
    }
    if (want_lo) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lo[i] = (H)(x[i] - (float)hi[i]);
            lo[4 + i] = (H)(y[i] - (float)hi[4 + i]);
        }
    }
}

template <int ACT, bool HAS_RES, int SPLIT, int FMT>
__global__ __launch_bounds__(HB_THREADS, SPLIT == 3 ? 1 : 2) void gemm_bf16_kernel(
    const float* __restrict__ A, int64_t lda, const uint16_t* __restrict__ Whi, const uint16_t* __restrict__ Wlo,
    const float* __restrict__ bias, const float* residual, int64_t ldr, float* Cout, int64_t ldc, int M, int N, int K,
    float scale, int scale_cols) {
    using Cfg = HbCfg<SPLIT>;
    typedef typename Half16<FMT>::V8 bf16x8;
    constexpr int NPL = Cfg::NPL;
    extern __shared__ __attribute__((aligned(16))) char smem_b[];

    const unsigned nb = N / HB_BN, mp = (M + HB_BM - 1) / HB_BM;
    unsigned mpanel, nblk;
    if (!xcd_panel_map(blockIdx.x, mp, nb, mpanel, nblk)) return;
    const int m0 = mpanel * HB_BM, n0 = nblk * HB_BN;

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wm = wv >> 1, wn = wv & 1, li = lane & 31, lh = lane >> 5;
    const int c8 = threadIdx.x & 7, r0 = threadIdx.x >> 3;        // staging: rows r0 + 32 i, 8-element chunk c8

    const float* ap[4];
    int64_t woff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + r0 + 32 * i;
        m = m < M ? m : M - 1;
        ap[i] = A + (int64_t)m * lda + c8 * 8;
        woff[i] = (int64_t)(n0 + r0 + 32 * i) * K + c8 * 8;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[mt][nt][t] = 0.f;

    f32x4 sa[4][2];
    u32x4 sw[NPL][4];
    auto load = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sa[i][0] = *reinterpret_cast<const f32x4*>(ap[i] + kt * HB_BK);
            sa[i][1] = *reinterpret_cast<const f32x4*>(ap[i] + kt * HB_BK + 4);
            sw[0][i] = *reinterpret_cast<const u32x4*>(Whi + woff[i] + kt * HB_BK);
            if (SPLIT == 3) sw[NPL - 1][i] = *reinterpret_cast<const u32x4*>(Wlo + woff[i] + kt * HB_BK);
        }
    };
    auto store = [&](int buf) {
        char* base = smem_b + buf * Cfg::BUF;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int off = (r0 + 32 * i) * HB_LDB + c8 * 16;
            bf16x8 hi, lo;
            split8<FMT>(sa[i][0], sa[i][1], hi, lo, SPLIT == 3);
            *reinterpret_cast<bf16x8*>(base + off) = hi;
            if (SPLIT == 3) *reinterpret_cast<bf16x8*>(base + HB_PLANE + off) = lo;
#pragma unroll
            for (int p = 0; p < NPL; ++p) *reinterpret_cast<u32x4*>(base + (NPL + p) * HB_PLANE + off) = sw[p][i];
        }
    };

    // K loop: same software pipeline as the fp32 kernel (mma_core.h): per tile four k16 groups; the LDS writes of
    // tile t+1 and the global loads of tile t+2 ride between the MFMAs of groups 0/1, the barrier sits before the last
    // group, whose MFMAs cover the first fragment read of the next tile.
    const int nk = K / HB_BK;
    load(0);
    store(0);
    if (nk > 1) load(1);
    __syncthreads();
    HbFrag<SPLIT, FMT> f0, f1;
    hb_frag_load<SPLIT, FMT>(smem_b, 0, wm, wn, li, lh, f0);
    for (int kt = 0; kt < nk; ++kt) {
        const char* cur = smem_b + (kt & 1) * Cfg::BUF;
        const bool has_next = kt + 1 < nk;
        hb_frag_load<SPLIT, FMT>(cur, 1, wm, wn, li, lh, f1);
        if (has_next) store((kt & 1) ^ 1);
        hb_frag_mma<SPLIT, FMT>(f0, acc);
        __builtin_amdgcn_sched_barrier(0);
        hb_frag_load<SPLIT, FMT>(cur, 2, wm, wn, li, lh, f0);
        if (kt + 2 < nk) load(kt + 2);
        hb_frag_mma<SPLIT, FMT>(f1, acc);
        __builtin_amdgcn_sched_barrier(0);
        hb_frag_load<SPLIT, FMT>(cur, 3, wm, wn, li, lh, f1);
        hb_frag_mma<SPLIT, FMT>(f0, acc);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (has_next) hb_frag_load<SPLIT, FMT>(smem_b + ((kt & 1) ^ 1) * Cfg::BUF, 0, wm, wn, li, lh, f0);
        hb_frag_mma<SPLIT, FMT>(f1, acc);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: identical to gemm_f32.hip (the 32x32 accumulator map does not depend on the operand dtype)
    constexpr int LDE = 64 + 4;
    const int er = lane >> 4, ec = (lane & 15) * 4;
    const int gm0 = m0 + wm * 64, gn = n0 + wn * 64 + ec;
    f32x4 res[16];
    if (HAS_RES) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = min(gm0 + er + 4 * i, M - 1);
            res[i] = *reinterpret_cast<const f32x4*>(residual + (int64_t)row * ldr + gn);
        }
    }
    __syncthreads();
    float* stage = reinterpret_cast<float*>(smem_b) + wv * (64 * LDE);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int col = n0 + wn * 64 + nt * 32 + li;
        const float b = bias ? bias[col] : 0.f;
        const float sc = col < scale_cols ? scale : 1.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                float v = (acc[mt][nt][t] + b) * sc;
                if (ACT == RNAMSM_ACT_GELU_ERF) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
                stage[(mt * 32 + (t & 3) + 8 * (t >> 2) + 4 * lh) * LDE + nt * 32 + li] = v;
            }
    }
    f32x4 ov[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        ov[i] = *reinterpret_cast<const f32x4*>(&stage[(er + 4 * i) * LDE + ec]);
        if (HAS_RES) ov[i] += res[i];
    }
    if (m0 + HB_BM <= M) {
#pragma unroll
        for (int i = 0; i < 16; ++i) *reinterpret_cast<f32x4*>(Cout + (int64_t)(gm0 + er + 4 * i) * ldc + gn) = ov[i];
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (gm0 + er + 4 * i < M) *reinterpret_cast<f32x4*>(Cout + (int64_t)(gm0 + er + 4 * i) * ldc + gn) = ov[i];
    }
}

template <int FMT>
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ hi,
                                                         uint16_t* __restrict__ lo, int64_t n) {
    typedef typename Half16<FMT>::T H;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float x = src[i];
        const H h = (H)x;
        hi[i] = __builtin_bit_cast(uint16_t, h);
        if (lo) lo[i] = __builtin_bit_cast(uint16_t, (H)(x - (float)h));
    }
}

template <int ACT, bool HAS_RES, int SPLIT, int FMT>
static int launch_hb(const float* A, int64_t lda, const uint16_t* Whi, const uint16_t* Wlo, const float* bias,
                     const float* residual, int64_t ldr, float* Cout, int64_t ldc, int M, int N, int K, float scale,
                     int scale_cols, hipStream_t stream) {
    static bool configured = false;
    auto kern = gemm_bf16_kernel<ACT, HAS_RES, SPLIT, FMT>;
    constexpr int lds = HbCfg<SPLIT>::LDS;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return fail(RNAMSM_ERR_HIP, "gemm_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e));
        configured = true;
    }
    const unsigned grid = xcd_panel_grid((M + HB_BM - 1) / HB_BM, N / HB_BN);
    KernelTimer timer(TC_GEMM, 2.0 * M * N * K, 4.0 * ((double)M * K + 0.5 * HbCfg<SPLIT>::NPL * (double)N * K + (double)M * N * (HAS_RES ? 2 : 1)), stream);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(HB_THREADS), lds, stream, A, lda, Whi, Wlo, bias, residual, ldr, Cout, ldc, M,
                       N, K, scale, scale_cols);
    RNAMSM_CHECK_LAUNCH("gemm_bf16");
    return RNAMSM_OK;
}

}  // namespace rnamsm

using namespace rnamsm;

extern "C" int rnamsm_split_bf16(const float* src, uint16_t* hi, uint16_t* lo, int64_t n, int fmt, void* stream) {
    RNAMSM_CHECK_ARG(src && hi && n > 0 && (fmt == 0 || fmt == 1), "split_bf16: bad arguments");
    const int64_t blocks = (n + 255) / 256;
    const dim3 grid((unsigned)(blocks < 8192 ? blocks : 8192));
    if (fmt == 1)
        hipLaunchKernelGGL(split_bf16_kernel<1>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), src, hi, lo, n);
    else
        hipLaunchKernelGGL(split_bf16_kernel<0>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), src, hi, lo, n);
    RNAMSM_CHECK_LAUNCH("split_bf16");
    return RNAMSM_OK;
}

extern "C" int rnamsm_gemm_bf16(const float* A, int64_t lda, const uint16_t* W_hi, const uint16_t* W_lo, const float* bias,
                                const float* residual, int64_t ldr, float* Cout, int64_t ldc, int64_t M, int N, int K,
                                int act, float scale, int scale_cols, int split, int fmt, void* stream) {
    RNAMSM_CHECK_ARG(A && W_hi && Cout, "gemm_bf16: null pointer");
    RNAMSM_CHECK_ARG(split == 1 || (split == 3 && W_lo), "gemm_bf16: split must be 1, or 3 with a lo plane");
    RNAMSM_CHECK_ARG((fmt == 0) || (fmt == 1 && split == 3), "gemm_bf16: fmt 0 (bf16) or 1 (fp16, split 3 only)");
    RNAMSM_CHECK_ARG(M > 0 && M <= INT32_MAX && N > 0 && K > 0, "gemm_bf16: bad shape");
    RNAMSM_CHECK_ARG(N % HB_BN == 0 && K % HB_BK == 0, "gemm_bf16: need N %% 128 == 0 and K %% 64 == 0 (N=%d K=%d)", N, K);
    RNAMSM_CHECK_ARG(lda >= K && lda % 4 == 0 && ldc >= N && ldc % 4 == 0, "gemm_bf16: bad leading dimension");
    RNAMSM_CHECK_ARG(aligned16(A) && aligned16(W_hi) && aligned16(Cout) && (!W_lo || aligned16(W_lo)), "gemm_bf16: 16-byte alignment");
    RNAMSM_CHECK_ARG(!residual || (ldr >= N && ldr % 4 == 0 && aligned16(residual)), "gemm_bf16: bad residual");
    RNAMSM_CHECK_ARG(act == RNAMSM_ACT_NONE || act == RNAMSM_ACT_GELU_ERF, "gemm_bf16: unknown activation %d", act);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int m = (int)M;
#define HB_GO(ACT_, RES_, SP_, FMT_) \
    launch_hb<ACT_, RES_, SP_, FMT_>(A, lda, W_hi, W_lo, bias, residual, ldr, Cout, ldc, m, N, K, scale, scale_cols, s)
#define HB_ACT_RES(SP_, FMT_)                                                                                     \
    do {                                                                                                          \
        if (act == RNAMSM_ACT_GELU_ERF)                                                                           \
            return residual ? HB_GO(RNAMSM_ACT_GELU_ERF, true, SP_, FMT_) : HB_GO(RNAMSM_ACT_GELU_ERF, false, SP_, FMT_); \
        return residual ? HB_GO(RNAMSM_ACT_NONE, true, SP_, FMT_) : HB_GO(RNAMSM_ACT_NONE, false, SP_, FMT_);       \
    } while (0)
    if (fmt == 1) HB_ACT_RES(3, 1);
    if (split == 3) HB_ACT_RES(3, 0);
    HB_ACT_RES(1, 0);
#undef HB_ACT_RES
#undef HB_GO
}
