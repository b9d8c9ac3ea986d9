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
#include "half16.h"

namespace rnamsm {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

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
        const float xv = pinned(x[i]), yv = pinned(y[i]);
        hi[i] = (H)xv;
        hi[4 + i] = (H)yv;
        if (want_lo) {
            lo[i] = (H)(xv - (float)hi[i]);
            lo[4 + i] = (H)(yv - (float)hi[4 + i]);
        }
    }
}

// K1 folded into the 16-bit GEMMs (the scheme of gemm_f32.hip FOLD / STATS).  Runtime switches of the two
// 256x256 kernels (null pointers = the plain epilogue):
//   consumer (QKV / fc1; A planes = the RAW residual stream, W planes = W * gamma): stats [M, 2] = (mean, rstd) per row,
//     c [N] = row sums of the Wg planes, `bias` carries d [N]:  out = act((rstd (acc - mean c) + d) * colscale)
//   producer (out_proj / fc2 + residual): besides the fp32 residual stream it writes the same values as 16-bit planes
//     xhi / xlo [M, ldx] (the next consumer's A operand: LayerNorm never runs as a launch) and leaves the slab sums
//     partials [N/32, pld, 2] (sum, sum of squares about the slab mean) for rnamsm_row_stats_from_partials.
struct Fold16 {
    const float* c = nullptr;
    const float2* stats = nullptr;
    float2* partials = nullptr;
    int64_t pld = 0;
    uint16_t* xhi = nullptr;
    uint16_t* xlo = nullptr;
    int64_t ldx = 0;
};
__device__ __forceinline__ float sum8_dpp16(float v) {      // sum over aligned groups of 8 lanes (i^1, i^2, 7-i), no LDS
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    return v;
}
// second half of both epilogues (row-per-lane layout: lane -> row er + 4 i, columns ec .. ec + 3 of the wave's 64x64 slab)
template <int ACT, int SPLIT, int FMT>
__device__ __forceinline__ void fold16_rows(f32x4 (&ov)[16], const Fold16& fa, const float2 (&st)[16], const f32x4& c4,
                                            const f32x4& d4, float fs) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[i][e] = fmaf(st[i].y, fmaf(-st[i].x, c4[e], ov[i][e]), d4[e]) * fs;
        if (ACT == RNAMSM_ACT_GELU_ERF) {
            const f32x2 g0 = gelu_erf2(f32x2{ov[i][0], ov[i][1]}), g1 = gelu_erf2(f32x2{ov[i][2], ov[i][3]});
            ov[i] = f32x4{g0[0], g0[1], g1[0], g1[1]};
        }
    }
}
template <int SPLIT, int FMT>
__device__ __forceinline__ void fold16_produce(const f32x4 (&ov)[16], const Fold16& fa, float* stage, int lane, int er, int ec,
                                               int gm0, int gn, int gnb, int M) {
    constexpr int LDE = 64 + 4;
    typedef typename Half16<FMT>::T H;
    typedef H H4 __attribute__((ext_vector_type(4)));
    // the stored residual stream once more as planes
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (gm0 + er + 4 * i < M) {
            H4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = pinned(ov[i][e]);
                hi[e] = (H)x;
                lo[e] = (H)(x - (float)hi[e]);
            }
            const int64_t o = (int64_t)(gm0 + er + 4 * i) * fa.ldx + gn;
            epi_store(reinterpret_cast<H4*>(fa.xhi + o), hi);
            if (SPLIT == 3) epi_store(reinterpret_cast<H4*>(fa.xlo + o), lo);
        }
    }
    // slab sums of the fp32 values (8 lanes = 32 columns of a row), parked in the staging rows' padding, then lane = row
    float ps[16], pq[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) ps[i] = sum8_dpp16((ov[i][0] + ov[i][1]) + (ov[i][2] + ov[i][3]));
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float mb = ps[i] * (1.f / 32.f);
        const float d0 = ov[i][0] - mb, d1 = ov[i][1] - mb, d2 = ov[i][2] - mb, d3 = ov[i][3] - mb;
        pq[i] = sum8_dpp16(fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, d3 * d3))));
    }
    if ((lane & 7) == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) *reinterpret_cast<float2*>(&stage[(er + 4 * i) * LDE + 64 + 2 * (ec / 32)]) = float2{ps[i], pq[i]};
    }
    float2* pout = fa.partials + (int64_t)(gnb / 32) * fa.pld + gm0 + lane;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        const float2 pr = *reinterpret_cast<const float2*>(&stage[lane * LDE + 64 + 2 * sl]);
        if (gm0 + lane < M) pout[(int64_t)sl * fa.pld] = pr;
    }
}

// Epilogue shared by the 16-bit GEMM kernels: identical to gemm_f32.hip (the 32x32 accumulator map does not depend on
// the operand dtype); O_PL writes 16-bit planes for the next GEMM instead of fp32.
template <int ACT, bool HAS_RES, int SPLIT, int FMT, bool O_PL>
__device__ __forceinline__ void hb_epilogue(f32x16 (&acc)[2][2], char* smem_b, int gm0, int gnb, int wv,
                                            int lane, int li, int lh, const float* __restrict__ bias,
                                            const float* residual, int64_t ldr, float* Cout, int64_t ldc, int M,
                                            float scale, int scale_cols, uint16_t* __restrict__ Ohi,
                                            uint16_t* __restrict__ Olo, const Fold16& fa = Fold16{}) {
    constexpr int LDE = 64 + 4;
    const int er = lane >> 4, ec = (lane & 15) * 4;
    const int gn = gnb + ec;                                  // gm0 / gnb: global origin of this wave's 64x64 slab
    const bool fold = !HAS_RES && fa.stats != nullptr;        // uniform
    f32x4 res[16];
    if (HAS_RES) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = min(gm0 + er + 4 * i, M - 1);
            res[i] = epi_load(reinterpret_cast<const f32x4*>(residual + (int64_t)row * ldr + gn));
        }
    }
    float2 st[16];
    f32x4 c4, d4;
    if (fold) {
#pragma unroll
        for (int i = 0; i < 16; ++i) st[i] = fa.stats[min(gm0 + er + 4 * i, M - 1)];
        c4 = *reinterpret_cast<const f32x4*>(fa.c + gn);
        d4 = *reinterpret_cast<const f32x4*>(bias + gn);
    }
    __syncthreads();
    float* stage = reinterpret_cast<float*>(smem_b) + wv * (64 * LDE);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int col = gnb + nt * 32 + li;
        const float b = (!fold && bias) ? bias[col] : 0.f;
        const float sc = (!fold && col < scale_cols) ? scale : 1.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < 16; t += 2) {                    // pairs: the GELU runs on the packed-fp32 VALU
                f32x2 v = f32x2{(acc[mt][nt][t] + b) * sc, (acc[mt][nt][t + 1] + b) * sc};   // fold: the raw sums
                if (ACT == RNAMSM_ACT_GELU_ERF && !fold) v = gelu_erf2(v);
                stage[(mt * 32 + (t & 3) + 8 * (t >> 2) + 4 * lh) * LDE + nt * 32 + li] = v[0];
                stage[(mt * 32 + ((t + 1) & 3) + 8 * ((t + 1) >> 2) + 4 * lh) * LDE + nt * 32 + li] = v[1];
            }
    }
    f32x4 ov[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        ov[i] = *reinterpret_cast<const f32x4*>(&stage[(er + 4 * i) * LDE + ec]);
        if (HAS_RES) ov[i] += res[i];
    }
    if (fold) fold16_rows<ACT, SPLIT, FMT>(ov, fa, st, c4, d4, gn < scale_cols ? scale : 1.f);
    if (HAS_RES && fa.partials) fold16_produce<SPLIT, FMT>(ov, fa, stage, lane, er, ec, gm0, gn, gnb, M);
    if (O_PL) {      // 4 values -> 4 halves hi (+ 4 halves lo): 8-byte stores into the planes
        typedef typename Half16<FMT>::T H;
        typedef H H4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (gm0 + er + 4 * i < M) {
                H4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = pinned(ov[i][e]);
                    hi[e] = (H)x;
                    lo[e] = (H)(x - (float)hi[e]);
                }
                const int64_t o = (int64_t)(gm0 + er + 4 * i) * ldc + gn;
                epi_store(reinterpret_cast<H4*>(Ohi + o), hi);
                if (SPLIT == 3) epi_store(reinterpret_cast<H4*>(Olo + o), lo);
            }
        }
    } else if (gm0 + 64 <= M) {
#pragma unroll
        for (int i = 0; i < 16; ++i) epi_store(reinterpret_cast<f32x4*>(Cout + (int64_t)(gm0 + er + 4 * i) * ldc + gn), ov[i]);
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (gm0 + er + 4 * i < M) epi_store(reinterpret_cast<f32x4*>(Cout + (int64_t)(gm0 + er + 4 * i) * ldc + gn), ov[i]);
    }
}

// A_PL: the A operand arrives pre-split as 16-bit planes (Ahi/Alo, row stride lda halves) written by its producer
//       (rnamsm_layernorm_split, or this kernel's O_PL epilogue) -- staging is then a plain 16-B copy, no conversion.
// O_PL: the epilogue writes the result as 16-bit planes (Ohi/Olo, row stride ldc halves) for the next GEMM.
template <int ACT, bool HAS_RES, int SPLIT, int FMT, bool A_PL, bool O_PL>
__global__ __launch_bounds__(HB_THREADS, SPLIT == 3 ? 1 : 2) void gemm_bf16_kernel(
    const float* __restrict__ A, int64_t lda, const uint16_t* __restrict__ Whi, const uint16_t* __restrict__ Wlo,
    const float* __restrict__ bias, const float* residual, int64_t ldr, float* Cout, int64_t ldc, int M, int N, int K,
    float scale, int scale_cols, const uint16_t* __restrict__ Ahi, const uint16_t* __restrict__ Alo,
    uint16_t* __restrict__ Ohi, uint16_t* __restrict__ Olo) {
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
    int64_t woff[4], aoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + r0 + 32 * i;
        m = m < M ? m : M - 1;
        ap[i] = A + (int64_t)m * lda + c8 * 8;
        aoff[i] = (int64_t)m * lda + c8 * 8;
        woff[i] = (int64_t)(n0 + r0 + 32 * i) * K + c8 * 8;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[mt][nt][t] = 0.f;

    f32x4 sa[A_PL ? 1 : 4][2];
    u32x4 sah[NPL][A_PL ? 4 : 1];
    u32x4 sw[NPL][4];
    auto load = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (A_PL) {
                sah[0][i] = *reinterpret_cast<const u32x4*>(Ahi + aoff[i] + kt * HB_BK);
                if (SPLIT == 3) sah[NPL - 1][i] = *reinterpret_cast<const u32x4*>(Alo + aoff[i] + kt * HB_BK);
            } else {
                sa[i][0] = *reinterpret_cast<const f32x4*>(ap[i] + kt * HB_BK);
                sa[i][1] = *reinterpret_cast<const f32x4*>(ap[i] + kt * HB_BK + 4);
            }
            sw[0][i] = *reinterpret_cast<const u32x4*>(Whi + woff[i] + kt * HB_BK);
            if (SPLIT == 3) sw[NPL - 1][i] = *reinterpret_cast<const u32x4*>(Wlo + woff[i] + kt * HB_BK);
        }
    };
    auto store = [&](int buf) {
        char* base = smem_b + buf * Cfg::BUF;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int off = (r0 + 32 * i) * HB_LDB + c8 * 16;
            if (A_PL) {
#pragma unroll
                for (int p = 0; p < NPL; ++p) *reinterpret_cast<u32x4*>(base + p * HB_PLANE + off) = sah[p][i];
            } else {
                bf16x8 hi, lo;
                split8<FMT>(sa[i][0], sa[i][1], hi, lo, SPLIT == 3);
                *reinterpret_cast<bf16x8*>(base + off) = hi;
                if (SPLIT == 3) *reinterpret_cast<bf16x8*>(base + HB_PLANE + off) = lo;
            }
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

    hb_epilogue<ACT, HAS_RES, SPLIT, FMT, O_PL>(acc, smem_b, m0 + wm * 64, n0 + wn * 64, wv, lane, li, lh, bias, residual, ldr,
                                                Cout, ldc, M, scale, scale_cols, Ohi, Olo);
}

// Algorithmic HBM bytes of a plane-input 16-bit GEMM launch: both operands once as `npl` 16-bit planes, the output once
// (planes, or the fp32 residual stream) and the residual read.
static inline double gemm16_bytes(double M, double N, double K, int npl, bool has_res, bool out_planes) {
    return 2.0 * npl * (M * K + N * K) + (out_planes ? 2.0 * npl : 4.0) * M * N + (has_res ? 4.0 * M * N : 0.0);
}

// ---- plane-input GEMM with LDS-DMA staging -------------------------------------------------------------------------
// Both operands are 16-bit planes in HBM (A from rnamsm_layernorm_split / an O_PL epilogue, W from rnamsm_split_bf16), so
// they go global -> LDS by global_load_lds_dwordx4 with no VGPR and no ds_write: the register-staged kernel above is
// bound by the LDS write path (64 KB per K tile per CU at ~79 B/clk) plus its fragment reads.  A DMA writes 64 lanes x 16 B
// linearly, so tiles are unpadded [128 rows][128 B] and the bank-conflict fix is an XOR swizzle applied on the SOURCE
// address and on the READ (cdna_hip_programming.md rule 21): physical 16-B chunk = logical chunk ^ ((row >> 1) & 7).  For
// any ds_read_b128 lane group (16 rows, one logical chunk) that yields 16 distinct slots of the 256-B bank row.
constexpr int HD_PLANE = 128 * 128;                 // bytes, unpadded
template <int SPLIT>
struct HdCfg {
    static constexpr int NPL = SPLIT == 3 ? 2 : 1;
    static constexpr int BUF = 2 * NPL * HD_PLANE;
    static constexpr int LDS = (2 * BUF) > 4 * 64 * 68 * 4 ? (2 * BUF) : 4 * 64 * 68 * 4;    // >= epilogue staging
};

template <int SPLIT, int FMT>
__device__ __forceinline__ void hd_frag_load(const char* buf, int kk, int wm, int wn, int li, int lh, HbFrag<SPLIT, FMT>& f) {
    typedef typename Half16<FMT>::V8 V8;
    constexpr int NPL = HdCfg<SPLIT>::NPL;
    const int chunk = ((2 * kk + lh) ^ ((li >> 1) & 7)) * 16;       // tile rows are multiples of 32 + li: (row>>1)&7 = (li>>1)&7
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f.a[p][t] = *reinterpret_cast<const V8*>(buf + p * HD_PLANE + (wm * 64 + t * 32 + li) * 128 + chunk);
            f.b[p][t] = *reinterpret_cast<const V8*>(buf + (NPL + p) * HD_PLANE + (wn * 64 + t * 32 + li) * 128 + chunk);
        }
    }
}

template <int ACT, bool HAS_RES, int SPLIT, int FMT, bool O_PL>
__global__ __launch_bounds__(HB_THREADS, SPLIT == 3 ? 1 : 2) void gemm16_dma_kernel(
    const uint16_t* __restrict__ Ahi, const uint16_t* __restrict__ Alo, int64_t lda, const uint16_t* __restrict__ Whi,
    const uint16_t* __restrict__ Wlo, const float* __restrict__ bias, const float* residual, int64_t ldr, float* Cout,
    int64_t ldc, int M, int N, int K, float scale, int scale_cols, uint16_t* __restrict__ Ohi, uint16_t* __restrict__ Olo) {
    using Cfg = HdCfg<SPLIT>;
    constexpr int NPL = Cfg::NPL;
    extern __shared__ __attribute__((aligned(16))) char smem_b[];

    const unsigned nb = N / HB_BN, mp = (M + HB_BM - 1) / HB_BM;
    unsigned mpanel, nblk;
    if (!xcd_panel_map(blockIdx.x, mp, nb, mpanel, nblk)) return;
    const int m0 = mpanel * HB_BM, n0 = nblk * HB_BN;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wv >> 1, wn = wv & 1, li = lane & 31, lh = lane >> 5;

    // DMA map: one wave instruction = 8 rows x 128 B; wave w moves row groups w, w+4, w+8, w+12 of every plane.
    // lane -> (row R0 + lane/8, physical chunk lane%8) which must hold logical chunk (lane%8) ^ ((row>>1)&7).
    const int drow = lane >> 3;
    const int dchunk = (lane & 7) ^ ((4 * (wv & 1) + (lane >> 4)) & 7);       // (row>>1)&7 does not depend on j
    int64_t aoff[4], woff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 8 * (wv + 4 * j) + drow;
        int m = m0 + row;
        m = m < M ? m : M - 1;
        aoff[j] = (int64_t)m * lda + dchunk * 8;
        woff[j] = (int64_t)(n0 + row) * K + dchunk * 8;
    }
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto issue = [&](int kt, int buf) {
        char* base = smem_b + buf * Cfg::BUF;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int loff = (8 * (wv + 4 * j)) * 128;                          // wave-uniform LDS row-group offset
            __builtin_amdgcn_global_load_lds((gptr_t)(Ahi + aoff[j] + kt * HB_BK), (lptr_t)(base + loff), 16, 0, 0);
            if (SPLIT == 3)
                __builtin_amdgcn_global_load_lds((gptr_t)(Alo + aoff[j] + kt * HB_BK), (lptr_t)(base + HD_PLANE + loff), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(Whi + woff[j] + kt * HB_BK), (lptr_t)(base + NPL * HD_PLANE + loff), 16, 0, 0);
            if (SPLIT == 3)
                __builtin_amdgcn_global_load_lds((gptr_t)(Wlo + woff[j] + kt * HB_BK), (lptr_t)(base + (NPL + 1) * HD_PLANE + loff), 16, 0, 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[mt][nt][t] = 0.f;

    const int nk = K / HB_BK;
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        wait_dma_then_barrier<0>();      // tile kt has landed (every wave's share), the other buffer is free again
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        const char* cur = smem_b + (kt & 1) * Cfg::BUF;
        HbFrag<SPLIT, FMT> f0, f1;
        hd_frag_load<SPLIT, FMT>(cur, 0, wm, wn, li, lh, f0);
        hd_frag_load<SPLIT, FMT>(cur, 1, wm, wn, li, lh, f1);
        hb_frag_mma<SPLIT, FMT>(f0, acc);
        __builtin_amdgcn_sched_barrier(0);
        hd_frag_load<SPLIT, FMT>(cur, 2, wm, wn, li, lh, f0);
        hb_frag_mma<SPLIT, FMT>(f1, acc);
        __builtin_amdgcn_sched_barrier(0);
        hd_frag_load<SPLIT, FMT>(cur, 3, wm, wn, li, lh, f1);
        hb_frag_mma<SPLIT, FMT>(f0, acc);
        __builtin_amdgcn_sched_barrier(0);
        hb_frag_mma<SPLIT, FMT>(f1, acc);
    }
    hb_epilogue<ACT, HAS_RES, SPLIT, FMT, O_PL>(acc, smem_b, m0 + wm * 64, n0 + wn * 64, wv, lane, li, lh, bias, residual, ldr,
                                                Cout, ldc, M, scale, scale_cols, Ohi, Olo);
}

template <int ACT, bool HAS_RES, int SPLIT, int FMT, bool O_PL>
static int launch_hd(const uint16_t* Whi, const uint16_t* Wlo, const float* bias, const float* residual, int64_t ldr,
                     float* Cout, int64_t ldc, int64_t lda, int M, int N, int K, float scale, int scale_cols,
                     const uint16_t* a_hi, const uint16_t* a_lo, uint16_t* o_hi, uint16_t* o_lo, hipStream_t stream) {
    static DeviceOnce configured;
    auto kern = gemm16_dma_kernel<ACT, HAS_RES, SPLIT, FMT, O_PL>;
    constexpr int lds = HdCfg<SPLIT>::LDS;
    if (configured.pending()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return fail(RNAMSM_ERR_HIP, "gemm16_dma: hipFuncSetAttribute: %s", hipGetErrorString(e));
        configured.mark();
    }
    const unsigned grid = xcd_panel_grid((M + HB_BM - 1) / HB_BM, N / HB_BN);
    KernelTimer timer(TC_GEMM, 2.0 * M * N * K, gemm16_bytes(M, N, K, HdCfg<SPLIT>::NPL, HAS_RES, O_PL), stream, PEAK_F16_MFMA_TFLOPS, SPLIT);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(HB_THREADS), lds, stream, a_hi, a_lo, lda, Whi, Wlo, bias, residual, ldr, Cout,
                       ldc, M, N, K, scale, scale_cols, o_hi, o_lo);
    RNAMSM_CHECK_LAUNCH("gemm16_dma");
    return RNAMSM_OK;
}

// ---- 256x256 tile, 8 waves: the same LDS-DMA scheme sized for the DMA latency --------------------------------------
// Little's law: a CU consumes operand bytes at (bytes per K tile) / (MFMA cycles per K tile) and a DMA takes ~3000 cycles
// to land, so bytes-in-flight must cover rate x latency.  The 128x128 tile eats 42 B/clk (126 KB needed, 64 KB in
// flight); a 256x256 tile with BK = 32 eats 21 B/clk (64 KB needed = exactly the one 64 KB buffer in flight).
// 8 waves as 2(M) x 4(N), wave tile 128x64 = 4x2 MFMA tiles (128 accumulator VGPRs), two waves per SIMD.
// Rows are 64 B (32 halves): physical 16-B chunk = logical chunk ^ ((row >> 2) & 3), again 16 distinct slots per
// ds_read_b128 lane group.
constexpr int HX_BM = 256, HX_BN = 256, HX_BK = 32, HX_THREADS = 512;
constexpr unsigned GEMM16_PERSISTENT_BLOCKS = 256;     // one per CU; a multiple of 8 keeps the XCD affinity of xcd_panel_map_grouped
constexpr int HX_ROWB = HX_BK * 2;                     // 64 bytes per row
constexpr int HX_PLANE = 256 * HX_ROWB;                // 16 KB
template <int SPLIT>
struct HxCfg {
    static constexpr int NPL = SPLIT == 3 ? 2 : 1;
    static constexpr int BUF = 2 * NPL * HX_PLANE;
    static constexpr int EPI = 8 * 64 * 68 * 4;        // epilogue staging: 8 waves x [64][68] f32
    static constexpr int LDS = (2 * BUF) > EPI ? (2 * BUF) : EPI;
};
template <int SPLIT, int FMT>
struct HxFrag {
    typename Half16<FMT>::V8 a[HxCfg<SPLIT>::NPL][4], b[HxCfg<SPLIT>::NPL][2];
};
template <int SPLIT, int FMT>
__device__ __forceinline__ void hx_frag_mma(const HxFrag<SPLIT, FMT>& f, f32x16 (&acc)[4][2]) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            if (SPLIT == 3) {
                acc[mt][nt] = Half16<FMT>::mfma(f.a[1][mt], f.b[0][nt], acc[mt][nt]);
                acc[mt][nt] = Half16<FMT>::mfma(f.a[0][mt], f.b[1][nt], acc[mt][nt]);
            }
            acc[mt][nt] = Half16<FMT>::mfma(f.a[0][mt], f.b[0][nt], acc[mt][nt]);
        }
}

// ---- 256x256 tile, two buffers, software-pipelined fragments -------------------------------------------------------
// A kernel that reads a k-step's fragments and waits for them right before the MFMAs that use them (round 1-3: gemm16_dma256_kernel, removed in round 4) has its two
// waves per SIMD are barrier-synchronised, so they tend to sit in those waits together and the matrix pipe idles
// (PMC, split 3 QKV: 56% MFMA busy at the actual clock, waves 32% of their time in s_waitcnt).  Here a k-step's
// fragments are requested while the PREVIOUS step's MFMAs issue (two fragment sets, interleave pinned with
// sched_group_barrier), also across the tile boundary: the barrier sits in the middle of a tile, right after the tile's
// last fragment read -- at that point every wave is done READING the buffer, so the DMA of tile kt+2 can start
// overwriting it while the MFMAs of tile kt are still issuing, and the first fragments of tile kt+1 load under them.
// Measured against the kernel above in one process (cfg3 shapes): +3..5% for split 3 (QKV 337 -> 356 TF algorithmic,
// matrix pipe 56% -> 60% busy at the ~1.7 GHz the chip sustains here) and +4..10% for split 1.  What remains is not
// fragment latency: a 4-stage DMA pipeline with counted vmcnt (three tiles in flight, BK = 16 for split 3) was also
// built and measured -- split 1 +5..10%, split 3 -7..12% (twice the barriers for the same bytes) -- and dropped.
// What-if builds: skipping the B fragment reads gains <= 2% (LDS read bandwidth is not the limiter);
// skipping the DMA gains 23-45%; a DMA-only loop takes 60-75% of the kernel time.  Data movement and MFMA each need
// most of the time and overlap imperfectly -- contention of the LDS-DMA path or the power budget, still open.
// BK = 32 (64-B tile rows) or 64 (128-B rows = whole cache lines per DMA row, twice the bytes in flight, half the
// barriers; only split 1 fits: 2 buffers x 64 KB).  Split 1 with BK = 64 against BK = 32, one process, cfg3 shapes:
// QKV 775 -> 882 TF, out_proj 515 -> 553, fc1 714 -> 759, fc2 830 -> 996 TF ("gemm16_dma" = 4 forces BK = 32).
template <int SPLIT, int BK>
struct HsCfg {
    static constexpr int NPL = SPLIT == 3 ? 2 : 1;
    static constexpr int ROWB = BK * 2;
    static constexpr int PLANE = 256 * ROWB;
    static constexpr int BUF = 2 * NPL * PLANE;
    static constexpr int EPI = 8 * 64 * 68 * 4;
    static constexpr int LDS = (2 * BUF) > EPI ? (2 * BUF) : EPI;
    static constexpr int RPI = 1024 / ROWB;                 // tile rows per wave DMA instruction
    static constexpr int IPW = 256 / RPI / 8;               // DMA instructions per wave per plane tile
    static constexpr int KS = BK / 16;                      // MFMA k steps per tile
};

template <int ACT, bool HAS_RES, int SPLIT, int FMT, bool O_PL, int BK>
__global__ __launch_bounds__(HX_THREADS, 1) void gemm16_swp_kernel(
    const uint16_t* __restrict__ Ahi, const uint16_t* __restrict__ Alo, int64_t lda, const uint16_t* __restrict__ Whi,
    const uint16_t* __restrict__ Wlo, const float* __restrict__ bias, const float* residual, int64_t ldr, float* Cout,
    int64_t ldc, int M, int N, int K, float scale, int scale_cols, uint16_t* __restrict__ Ohi, uint16_t* __restrict__ Olo,
    int group, unsigned total_tiles, Fold16 fa) {
    using Cfg = HsCfg<SPLIT, BK>;
    constexpr int NPL = Cfg::NPL, ROWB = Cfg::ROWB, PLANE = Cfg::PLANE, KS = Cfg::KS;
    constexpr int NMF = 8 * (SPLIT == 3 ? 3 : 1);          // MFMAs per k step per wave
    constexpr int NDS = 6 * NPL;                           // fragment reads per k step per wave
    typedef typename Half16<FMT>::V8 V8;
    static_assert(BK == 32 || BK == 64, "tile depth");
    extern __shared__ __attribute__((aligned(16))) char smem_b[];

    const unsigned nb = N / HX_BN, mp = (M + HX_BM - 1) / HX_BM;
    const int lane_k = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wv >> 2, wn = wv & 3;
    // Persistent form: gridDim.x blocks (one per CU) walk the output tiles in launch order, block b taking virtual ids
    // b, b + gridDim.x, ... (gridDim.x % 8 == 0 keeps the XCD affinity of xcd_panel_map_grouped).  A tile's stores then
    // drain while the block already accumulates its next tile.  (A start offset per block, to spread the CUs' store bursts over the
    // tile period, was measured at 3 k .. 20 k cycles and changed nothing: EXPERIMENTS, round 2; removed in round 6.)
    for (unsigned vid = blockIdx.x; vid < total_tiles; vid += gridDim.x) {
    unsigned mpanel, nblk;
    if (!xcd_panel_map_grouped(vid, mp, nb, (unsigned)group, mpanel, nblk)) continue;
    const int m0 = mpanel * HX_BM, n0 = nblk * HX_BN;
    __syncthreads();                 // every wave has finished reading the previous tile's epilogue staging

    // Per-tile lane geometry from an OPAQUE copy of the lane id (see the epilogue below): nothing lane-derived is loop-invariant,
    // so nothing is hoisted to kernel entry and carried -- spilled -- across the tiles.
    int lane = lane_k, li = lane_k & 31, lh = lane_k >> 5;
    asm volatile("" : "+v"(lane));
    li = lane & 31;
    lh = lane >> 5;
    // DMA map: a wave instruction covers RPI rows; lane -> (row RPI*g + lane / chunks-per-row, physical chunk lane %
    // chunks-per-row), fetching the logical chunk the read-side swizzle expects there.  g = wv + 8j.
    constexpr int CPR = ROWB / 16;
    const int drow = lane / CPR;
    const int dchunk = BK == 32 ? ((lane & 3) ^ ((lane >> 4) & 3))                      // (row >> 2) & 3
                                : ((lane & 7) ^ ((4 * (wv & 1) + (lane >> 4)) & 7));    // (row >> 1) & 7, row = 8g + lane/8
    // source offsets: one 64-bit base per operand and lane; a lane's further rows lie 8 RPI rows apart -- a uniform stride for W
    // (N % 256 == 0), a 32-bit delta per row for A (rows past M are clamped to the last one: <= 255 rows, fits an int).  Eight 64-bit
    // offsets per lane used to live here: with 128 accumulators and two fragment sets that spilled 1-7 registers.
    const int row0 = Cfg::RPI * wv + drow;
    const int ma0 = min(m0 + row0, M - 1);
    const int64_t aoff0 = (int64_t)ma0 * lda + dchunk * 8, woff0 = (int64_t)(n0 + row0) * K + dchunk * 8;
    const int64_t wstride = (int64_t)(8 * Cfg::RPI) * K;
    int adelta[Cfg::IPW];
#pragma unroll
    for (int j = 0; j < Cfg::IPW; ++j) adelta[j] = (min(m0 + row0 + 8 * Cfg::RPI * j, M - 1) - ma0) * (int)lda;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto issue = [&](int kt, int buf) {
        char* base = smem_b + buf * Cfg::BUF;
#pragma unroll
        for (int j = 0; j < Cfg::IPW; ++j) {
            const int loff = (Cfg::RPI * (wv + 8 * j)) * ROWB;
            const int64_t ao = aoff0 + adelta[j] + kt * BK, wo = woff0 + j * wstride + kt * BK;
            __builtin_amdgcn_global_load_lds((gptr_t)(Ahi + ao), (lptr_t)(base + loff), 16, 0, 0);
            if (SPLIT == 3)
                __builtin_amdgcn_global_load_lds((gptr_t)(Alo + ao), (lptr_t)(base + PLANE + loff), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(Whi + wo), (lptr_t)(base + NPL * PLANE + loff), 16, 0, 0);
            if (SPLIT == 3)
                __builtin_amdgcn_global_load_lds((gptr_t)(Wlo + wo), (lptr_t)(base + (NPL + 1) * PLANE + loff), 16, 0, 0);
        }
    };
    auto frag_load = [&](const char* buf, int kk, HxFrag<SPLIT, FMT>& f) {
        const int chunk = (BK == 32 ? ((2 * kk + lh) ^ ((li >> 2) & 3)) : ((2 * kk + lh) ^ ((li >> 1) & 7))) * 16;
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                f.a[p][t] = *reinterpret_cast<const V8*>(buf + p * PLANE + (wm * 128 + t * 32 + li) * ROWB + chunk);
#pragma unroll
            for (int t = 0; t < 2; ++t)
                f.b[p][t] = *reinterpret_cast<const V8*>(buf + (NPL + p) * PLANE + (wn * 64 + t * 32 + li) * ROWB + chunk);
        }
    };
    // pin "NMF MFMAs with NDS fragment reads between them": one read per MFMA from the start, so the last read has the
    // rest of the run to land before the next phase waits for it
    auto interleave = [&]() {
#pragma unroll
        for (int i = 0; i < NDS; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NMF - NDS, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[mt][nt][t] = 0.f;

    const int nk = K / BK;
    issue(0, 0);
    wait_dma_then_barrier<0>();                               // tile 0 landed
    issue(nk > 1 ? 1 : 0, 1);                                 // (a redundant reload when nk == 1: never read)
    HxFrag<SPLIT, FMT> f[2];                                  // step kk lives in f[kk & 1]; KS is even
    frag_load(smem_b, 0, f[0]);
    for (int kt = 0; kt + 1 < nk; ++kt) {
        const char* cur = smem_b + (kt & 1) * Cfg::BUF;
        const char* nxt = smem_b + ((kt & 1) ^ 1) * Cfg::BUF;
#pragma unroll
        for (int kk = 0; kk + 1 < KS; ++kk) {                 // step kk on f[kk&1], step kk+1's fragments arriving
            frag_load(cur, kk + 1, f[(kk + 1) & 1]);
            hx_frag_mma<SPLIT, FMT>(f[kk & 1], acc);
            interleave();
        }
        // every wave is done reading `cur` once its last fragments have arrived; tile kt+1 (issued one tile ago) must
        // have landed
        wait_dma_then_barrier<0>();
        const int k2 = kt + 2 < nk ? kt + 2 : nk - 1;         // clamped: the last reload is never read
        // DEPHASED ISSUE: a wave is stuck for ~100 cycles per LDS-DMA request it issues (8 per tile); when the two waves of a SIMD issue
        // theirs at the same moment -- right after this barrier -- nobody feeds the matrix pipe meanwhile, so the upper wave group issues
        // one step later.  (Staging by operand as in gemm16_q16s_kernel -- row-half-major micro-steps,
        // W requested half a tile before A -- was built for this kernel too and measured no better than this delay: f16x3 six
        // GEMMs x1.026 against x1.035, plain bf16 QKV x1.01 and fc1 x0.93; EXPERIMENTS.md R3.3.)
        if (wm == 0) issue(k2, kt & 1);
        __builtin_amdgcn_sched_barrier(0);
        // last step of tile kt, step 0 of tile kt+1 arriving
        frag_load(nxt, 0, f[0]);
        hx_frag_mma<SPLIT, FMT>(f[(KS - 1) & 1], acc);
        interleave();
        if (wm == 1) issue(k2, kt & 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    {   // last tile
        const char* cur = smem_b + ((nk - 1) & 1) * Cfg::BUF;
#pragma unroll
        for (int kk = 0; kk + 1 < KS; ++kk) {
            frag_load(cur, kk + 1, f[(kk + 1) & 1]);
            hx_frag_mma<SPLIT, FMT>(f[kk & 1], acc);
            interleave();
        }
        hx_frag_mma<SPLIT, FMT>(f[(KS - 1) & 1], acc);
    }
    wait_dma_then_barrier<0>();                               // the clamped reload has landed: LDS is free for the epilogue
    // The epilogue's lane geometry is re-derived from an OPAQUE copy of the lane id inside the tile loop: values computed from
    // the loop-invariant `lane` are hoisted to kernel entry, live across the whole K loop next to 128 accumulators and two
    // fragment sets, and were spilled (7 registers in the plain-bf16 residual instance, reloaded once per tile).
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
#pragma unroll
    for (int p = 0; p < 2; ++p)
        hb_epilogue<ACT, HAS_RES, SPLIT, FMT, O_PL>(reinterpret_cast<f32x16(&)[2][2]>(acc[2 * p]), smem_b,
                                                    m0 + wm * 128 + p * 64, n0 + wn * 64, wv, lane_e, lane_e & 31, lane_e >> 5, bias, residual,
                                                    ldr, Cout, ldc, M, scale, scale_cols, Ohi, Olo, fa);
    }   // persistent tile loop
}

template <int ACT, bool HAS_RES, int SPLIT, int FMT, bool O_PL, int BK>
static int launch_hs(const uint16_t* Whi, const uint16_t* Wlo, const float* bias, const float* residual, int64_t ldr,
                     float* Cout, int64_t ldc, int64_t lda, int M, int N, int K, float scale, int scale_cols,
                     const uint16_t* a_hi, const uint16_t* a_lo, uint16_t* o_hi, uint16_t* o_lo, hipStream_t stream,
                     const Fold16& fa = Fold16{}) {
    static DeviceOnce configured;
    auto kern = gemm16_swp_kernel<ACT, HAS_RES, SPLIT, FMT, O_PL, BK>;
    constexpr int lds = HsCfg<SPLIT, BK>::LDS;
    if (configured.pending()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return fail(RNAMSM_ERR_HIP, "gemm16_swp: hipFuncSetAttribute: %s", hipGetErrorString(e));
        configured.mark();
    }
    // block order as in gemm_f32.hip: 32 blocks are resident per XCD; for the wide GEMMs (QKV 9, fc1 12 column blocks)
    // groups of 8 row panels keep a W slab shared by 8 panels instead of ~3: +5..6% (split 3), +7..10% (split 1),
    // measured in one process; of the 3-column-block GEMMs fc2 (K = 3072) is neutral to slightly worse and stays ungrouped, out_proj
    // (K = 768, HBM-bound by its fp32 residual traffic) takes groups of 4 panels since round 5: +2.3 % (bf16) / +2.5 % (f16x3) at
    // T = 131072, +4.8 % / +1.0 % at T = 1048576, bit-identical (docs/history/profiles_r05/r05_gemm16_group_n768.log)
    const int group = tuning().gemm_group > 0 ? tuning().gemm_group
                                              : (N / HX_BN > 4 ? (int)xcd_group_for_persistent((M + HX_BM - 1) / HX_BM, 8)
                                                               : (K <= 1024 ? (int)xcd_group_for_persistent((M + HX_BM - 1) / HX_BM, 4) : 1));
    const unsigned total = xcd_panel_grid_grouped((M + HX_BM - 1) / HX_BM, N / HX_BN, (unsigned)group);
    // 256 persistent blocks (one per CU) walk the tiles: +1-5 % over one block per tile (round 2; the knob went in round 6)
    const unsigned grid = GEMM16_PERSISTENT_BLOCKS < total ? GEMM16_PERSISTENT_BLOCKS : total;
    // (a folded-LayerNorm producer also writes the new residual stream as planes: the next consumer's A operand)
    KernelTimer timer(TC_GEMM, 2.0 * M * N * K,
                      gemm16_bytes(M, N, K, HxCfg<SPLIT>::NPL, HAS_RES, O_PL) + (fa.xhi ? 2.0 * HxCfg<SPLIT>::NPL * (double)M * N : 0.0),
                      stream, PEAK_F16_MFMA_TFLOPS, SPLIT);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(HX_THREADS), lds, stream, a_hi, a_lo, lda, Whi, Wlo, bias, residual, ldr, Cout,
                       ldc, M, N, K, scale, scale_cols, o_hi, o_lo, group, total, fa);
    RNAMSM_CHECK_LAUNCH("gemm16_swp");
    return RNAMSM_OK;
}

// ---- 256x256 tile, plain bf16, 16x16x32 MFMAs ------------------------------------------------------------------------
// The loop of gemm16_swp_kernel<.., SPLIT 1, BK 64> with v_mfma_f32_16x16x32_bf16 instead of 32x32x16: the same flops per
// cycle on paper, but the chip holds a higher clock on the 16x16 shape under sustained matrix load (MI355X_MICROARCH.md,
// DVFS give-back item 7: ~1.12-1.15x the FLOP/s of the 32x32 loop at equal cycles) -- and everything else in the kernel
// (DMA issue, LDS reads) speeds up with the clock.  Same tiles, same DMA map and swizzle (the 16 lanes of a ds_read_b128
// group hold rows distinct mod 16 and two k-groups: 16 distinct 16-B slots), wave tile 128x64 = 8x4 MFMA tiles (128
// accumulator VGPRs), a k-step is 32 deep: 32 MFMAs of 16 cycles against 12 fragment reads.  Products are identical to the
// 32x32 kernel's and are summed in a different order inside a k-step: results agree to fp32 rounding (exact on integers).
typedef float f32x4a __attribute__((ext_vector_type(4)));
// Epilogue of the 16x16x32 kernel, straight from the accumulators (no LDS transpose, no barrier).  With the MFMA operands
// swapped the tile comes out transposed in the registers, and with the W rows permuted by the DMA map lane (fr, fq) holds,
// of row 16 mt + fr of the wave's 128 x 64 tile, the columns 8 fq .. 8 fq + 7 (tiles t = 0, 1) and 32 + 8 fq .. + 7
// (t = 2, 3): two 16-byte (bf16) or four 16-byte (fp32) stores per row, the four fq lanes of a row writing 64 contiguous
// bytes (128 for fp32) per instruction.  A what-if had put the LDS-transposed epilogue at 27 % (QKV) / 34 % (fc1) of the
// launch.
struct HqEpiRegs {               // what the epilogue reads from memory besides the residual: requested before the stores
    f32x4 b4[4], c4[4];
    float2 st[8];
};
template <bool HAS_RES>
__device__ __forceinline__ void hq_epilogue_loads(HqEpiRegs& r, int gm0, int gnb, int lane, const float* __restrict__ bias,
                                                  int M, const Fold16& fa) {
    const int fr = lane & 15, fq = lane >> 4;
    const bool fold = !HAS_RES && fa.stats != nullptr;        // uniform
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int col = gnb + 32 * (t >> 1) + 8 * fq + 4 * (t & 1);
        r.b4[t] = bias ? *reinterpret_cast<const f32x4*>(bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};      // fold: d[n]
        r.c4[t] = fold ? *reinterpret_cast<const f32x4*>(fa.c + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) r.st[mt] = fold ? fa.stats[min(gm0 + mt * 16 + fr, M - 1)] : float2{0.f, 1.f};
}
template <int ACT, bool HAS_RES, bool O_PL>
__device__ __forceinline__ void hq_epilogue(f32x4a (&acc)[8][4], const HqEpiRegs& r, int gm0, int gnb, int lane,
                                            const float* residual, int64_t ldr, float* Cout, int64_t ldc, int M, float scale,
                                            int scale_cols, uint16_t* __restrict__ Ohi) {
    const int fr = lane & 15, fq = lane >> 4;
    // column of accumulator element e of tile t: gnb + 32 (t >> 1) + 8 fq + 4 (t & 1) + e
    float fs[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) fs[t] = gnb + 32 * (t >> 1) + 8 * fq + 4 * (t & 1) < scale_cols ? scale : 1.f;   // scale_cols % 4 == 0
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const int row = gm0 + mt * 16 + fr;
        const int rowc = min(row, M - 1);
        const float2 st = r.st[mt];                           // (0, 1) without the fold: v = (acc - 0 c) 1 + b
        f32x4 v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[t][e] = fmaf(st.y, fmaf(-st.x, r.c4[t][e], acc[mt][t][e]), r.b4[t][e]) * fs[t];
            if (ACT == RNAMSM_ACT_GELU_ERF) {
                const f32x2 g0 = gelu_erf2(f32x2{v[t][0], v[t][1]}), g1 = gelu_erf2(f32x2{v[t][2], v[t][3]});
                v[t] = f32x4{g0[0], g0[1], g1[0], g1[1]};
            }
        }
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {                   // columns 8 fq .. + 7 of the tile's first / second 32
            const int64_t o = (int64_t)row * ldc + gnb + 32 * hlf + 8 * fq;
            if (HAS_RES) {
                const float* rp = residual + (int64_t)rowc * ldr + gnb + 32 * hlf + 8 * fq;
                v[2 * hlf] += epi_load(reinterpret_cast<const f32x4*>(rp));
                v[2 * hlf + 1] += epi_load(reinterpret_cast<const f32x4*>(rp + 4));
            }
            if (row < M) {
                if (O_PL) {
                    typedef typename Half16<0>::T H;
                    typedef H H8 __attribute__((ext_vector_type(8)));
                    H8 hi;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hi[e] = (H)v[2 * hlf][e];
                        hi[4 + e] = (H)v[2 * hlf + 1][e];
                    }
                    epi_store(reinterpret_cast<H8*>(Ohi + o), hi);
                } else {
                    epi_store(reinterpret_cast<f32x4*>(Cout + o), v[2 * hlf]);
                    epi_store(reinterpret_cast<f32x4*>(Cout + o + 4), v[2 * hlf + 1]);
                }
            }
        }
    }
}

template <int ACT, bool HAS_RES, bool O_PL>
__global__ __launch_bounds__(HX_THREADS, 1) void gemm16_q16s_kernel(
    const uint16_t* __restrict__ Ahi, int64_t lda, const uint16_t* __restrict__ Whi, const float* __restrict__ bias,
    const float* residual, int64_t ldr, float* Cout, int64_t ldc, int M, int N, int K, float scale, int scale_cols,
    uint16_t* __restrict__ Ohi, int group, unsigned total_tiles, Fold16 fa) {
    using Cfg = HsCfg<1, 64>;
    constexpr int BK = 64, ROWB = Cfg::ROWB, PLANE = Cfg::PLANE;
    typedef typename Half16<0>::V8 V8;
    extern __shared__ __attribute__((aligned(16))) char smem_b[];

    const unsigned nb = N / HX_BN, mp = (M + HX_BM - 1) / HX_BM;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wv >> 2, wn = wv & 3, fr = lane & 15, fq = lane >> 4;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    // Persistent tile walk with the epilogue OFF the memory critical path.  The epilogue needs no LDS (hq_epilogue), so the
    // next tile's first operand tile is requested BEFORE the current tile's stores: vector memory operations retire in
    // order, and the first wait of the next tile then allows exactly the stores to stay in flight (vmcnt(HQ_STORES)) instead
    // of draining them (a what-if: 27 % of the QKV launch, 34 % of fc1's, was the wait for the previous tile's stores).
    constexpr int HQ_STORES = O_PL ? 16 : 32;                 // store instructions of one epilogue, per lane
    // STAGING BY OPERAND.  A wave is stuck for ~100 cycles per LDS-DMA request it issues, and while both waves of a SIMD issue
    // theirs at the same moment nobody feeds the matrix pipe.  Here the lower wave group (wm = 0) moves the whole W tile and
    // the upper one the whole A tile -- 8 requests per wave and K tile either way -- and the micro-steps run in the order
    // (ks 0, h 0), (ks 1, h 0), (ks 0, h 1), (ks 1, h 1): the W tile has been read completely after the FIRST micro-step's
    // fragment reads, the A tile after the third's.  So W of tile kt+2 is requested behind a barrier at the start of
    // micro-step 1 and A of tile kt+2 behind the barrier at the start of micro-step 3: the two bursts lie half a tile apart,
    // each overlapped by the other group's MFMAs, and both have at least a whole tile of flight time.
    const int w4 = wv & 3;
    const int drow = lane >> 3;
    const int dchunk = (lane & 7) ^ ((4 * (w4 & 1) + (lane >> 4)) & 7);      // (row >> 1) & 7, row = 8 (w4 + 4 j) + lane / 8
    const int r0 = 8 * w4 + drow;                             // the wave's rows: r0 + 32 j, j = 0..7
    const uint16_t* __restrict__ src = wm ? Ahi : Whi;        // uniform
    int64_t off[8];
    auto find_tile = [&](unsigned& vid, int& m0, int& n0) -> bool {
        for (; vid < total_tiles; vid += gridDim.x) {
            unsigned mpanel, nblk;
            if (xcd_panel_map_grouped(vid, mp, nb, (unsigned)group, mpanel, nblk)) {
                m0 = mpanel * HX_BM;
                n0 = nblk * HX_BN;
                return true;
            }
        }
        return false;
    };
    auto set_offsets = [&](int m0, int n0) {
        // W rows are PERMUTED on their way into LDS so that the (transposed) accumulators of a lane are 8 + 8 consecutive output
        // columns (hq_epilogue): LDS row 64 g + 16 t + 4 a + b  <-  weight row
        // 64 g + 32 (t >> 1) + 8 a + 4 (t & 1) + b; for LDS row r0 + 32 j (r0 < 32) that is weight row wrow(r0) + 32 j
        const int tt = (r0 >> 4) & 1, aa = (r0 >> 2) & 3;
        const int wrow0 = 8 * aa + 4 * tt + (r0 & 3);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int m = m0 + r0 + 32 * j;
            m = m < M ? m : M - 1;
            off[j] = wm ? (int64_t)m * lda + dchunk * 8 : (int64_t)(n0 + wrow0 + 32 * j) * K + dchunk * 8;
        }
    };
    auto issue_mine = [&](int kt, int buf) {                 // this wave group's operand of K tile kt
        char* base = smem_b + buf * Cfg::BUF + (wm ? 0 : PLANE) + 1024 * w4;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + off[j] + kt * BK), (lptr_t)(base + 4096 * j), 16, 0, 0);
    };
    unsigned vid = blockIdx.x;
    int m0, n0;
    if (!find_tile(vid, m0, n0)) return;
    set_offsets(m0, n0);
    issue_mine(0, 0);
    bool stores_in_flight = false;                            // uniform
    for (;;) {
    // lane (row fr, k-group fq) of a 16-row tile reads logical chunk 4*ks + fq of its row; (row >> 1) & 7 = (fr >> 1) & 7.
    // A tile is consumed in four micro-steps u = (k-step ks = u >> 1, row half h = u & 1) of 16 MFMAs: the A fragments of
    // one half (4 x V8) ping-pong by micro-step, the B fragments of a k-step (4 x V8) by k-step -- 64 fragment registers
    // instead of the 96 of two whole k-step sets, which with 128 accumulators would not fit 256.
    V8 ah[2][4], bq[2][4];
    auto load_a = [&](const char* buf, int ks, int h, V8 (&a)[4]) {
        const int chunk = ((4 * ks + fq) ^ ((fr >> 1) & 7)) * 16;
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = *reinterpret_cast<const V8*>(buf + (wm * 128 + (4 * h + t) * 16 + fr) * ROWB + chunk);
    };
    auto load_b = [&](const char* buf, int ks, V8 (&b)[4]) {
        const int chunk = ((4 * ks + fq) ^ ((fr >> 1) & 7)) * 16;
#pragma unroll
        for (int t = 0; t < 4; ++t) b[t] = *reinterpret_cast<const V8*>(buf + PLANE + (wn * 64 + t * 16 + fr) * ROWB + chunk);
    };
    f32x4a acc[8][4];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4a{0.f, 0.f, 0.f, 0.f};
    auto mma = [&](int h, const V8 (&a)[4], const V8 (&b)[4]) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                // operands swapped: the tile comes out TRANSPOSED in the registers -- lane (fr, fq) holds row fr, columns
                // 4 fq .. 4 fq + 3 of the 16x16 tile -- so the epilogue stores row pieces straight from the accumulators
                acc[4 * h + mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[nt], a[mt], acc[4 * h + mt][nt], 0, 0, 0);
    };

    const int nk = K / BK;
    // tile 0 was requested before the previous tile's stores (or at kernel start): it has landed once at most those
    // stores are still outstanding
    if (stores_in_flight) wait_dma_then_barrier<HQ_STORES>();
    else wait_dma_then_barrier<0>();
    issue_mine(nk > 1 ? 1 : 0, 1);
    load_a(smem_b, 0, 0, ah[0]);
    load_b(smem_b, 0, bq[0]);
#define HQ_PIN(NDS_)                                                                  \
    _Pragma("unroll") for (int i_ = 0; i_ < NDS_; ++i_) {                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                            \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                            \
    }                                                                                 \
    __builtin_amdgcn_sched_group_barrier(0x008, 16 - NDS_, 0);                        \
    __builtin_amdgcn_sched_barrier(0)
    for (int kt = 0; kt + 1 < nk; ++kt) {
        const char* cur = smem_b + (kt & 1) * Cfg::BUF;
        const char* nxt = smem_b + ((kt & 1) ^ 1) * Cfg::BUF;
        const int k2 = kt + 2 < nk ? kt + 2 : nk - 1;         // clamped: the last reload is never read (keeps the counts below fixed)
        load_a(cur, 1, 0, ah[1]);                             // u0 = (ks 0, h 0) computes; u1 = (ks 1, h 0) arriving: the last reads of W
        load_b(cur, 1, bq[1]);
        mma(0, ah[0], bq[0]);
        HQ_PIN(8);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // every wave has read the W tile of `cur`
        if (wm == 0) issue_mine(k2, kt & 1);                  // W of tile kt+2 -> the W plane of `cur`
        __builtin_amdgcn_sched_barrier(0);
        load_a(cur, 0, 1, ah[0]);                             // u1 computes; u2 = (ks 0, h 1) arriving
        mma(0, ah[1], bq[1]);
        HQ_PIN(4);
        load_a(cur, 1, 1, ah[1]);                             // u2 computes; u3 = (ks 1, h 1) arriving: the last reads of A
        mma(1, ah[0], bq[0]);
        HQ_PIN(4);
        // every wave has read the A tile of `cur`, and tile kt+1 has landed: the A group waits for all of its requests, the W
        // group leaves its newest 8 (W of tile kt+2, requested half a tile ago) in flight
        if (wm == 0) wait_dma_then_barrier<8>();
        else wait_dma_then_barrier<0>();
        if (wm == 1) issue_mine(k2, kt & 1);                  // A of tile kt+2 -> the A plane of `cur`
        __builtin_amdgcn_sched_barrier(0);
        load_a(nxt, 0, 0, ah[0]);                             // u3 computes; the next tile's u0 operands arriving
        load_b(nxt, 0, bq[0]);
        mma(1, ah[1], bq[1]);
        HQ_PIN(8);
    }
    {
        const char* cur = smem_b + ((nk - 1) & 1) * Cfg::BUF;
        load_a(cur, 1, 0, ah[1]);
        load_b(cur, 1, bq[1]);
        mma(0, ah[0], bq[0]);
        HQ_PIN(8);
        load_a(cur, 0, 1, ah[0]);
        mma(0, ah[1], bq[1]);
        HQ_PIN(4);
        load_a(cur, 1, 1, ah[1]);
        mma(1, ah[0], bq[0]);
        HQ_PIN(4);
        mma(1, ah[1], bq[1]);
    }
#undef HQ_PIN
    wait_dma_then_barrier<0>();                               // every wave is done with LDS; the clamped reload has landed
    // what the epilogue reads (bias / fold vectors, row statistics) first, then the next tile's tile 0, then the stores
    HqEpiRegs er;
    hq_epilogue_loads<HAS_RES>(er, m0 + wm * 128, n0 + wn * 64, lane, bias, M, fa);
    // ... and waited for HERE (a use, as far as hipcc can tell): with an LDS-DMA in flight it would otherwise wait vmcnt(0) at
    // their first real use and drain the next tile's operands inside the epilogue
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        asm volatile("" : "+v"(er.b4[t]));
        asm volatile("" : "+v"(er.c4[t]));
    }
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) asm volatile("" : "+v"(er.st[mt].x), "+v"(er.st[mt].y));
    asm volatile("" ::: "memory");                            // ... and the requests below stay below
    __builtin_amdgcn_sched_barrier(0);
    const int em0 = m0 + wm * 128, en0 = n0 + wn * 64;
    unsigned nvid = vid + gridDim.x;
    int m1 = 0, n1 = 0;
    const bool more = find_tile(nvid, m1, n1);
    if (more) {
        set_offsets(m1, n1);
        issue_mine(0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    hq_epilogue<ACT, HAS_RES, O_PL>(acc, er, em0, en0, lane, residual, ldr, Cout, ldc, M, scale, scale_cols, Ohi);
    if (!more) break;
    vid = nvid;
    m0 = m1;
    n0 = n1;
    // the hoisted request pays only if exactly HQ_STORES vector memory instructions follow it: a residual adds loads, a
    // ragged last row panel drops stores -- those tiles drain (vmcnt(0)) as before
    stores_in_flight = !HAS_RES && em0 + 128 <= M;
    if (!stores_in_flight) __builtin_amdgcn_s_waitcnt(0x0f70);           // vmcnt(0), keep expcnt / lgkmcnt
    }   // persistent tile loop
}

// (a three-product 16x16x32 kernel staged by operand, "gemm16_x3q", was measured at +1.8 % and removed in round 4: its k order
// moved the f16x3 embedding error -- EXPERIMENTS.md R3.3b, git history)

template <int ACT, bool HAS_RES, bool O_PL>
static int launch_hq(const uint16_t* Whi, const float* bias, const float* residual, int64_t ldr, float* Cout, int64_t ldc,
                     int64_t lda, int M, int N, int K, float scale, int scale_cols, const uint16_t* a_hi, uint16_t* o_hi,
                     hipStream_t stream, const Fold16& fa = Fold16{}) {
    static DeviceOnce configured;
    auto kern = gemm16_q16s_kernel<ACT, HAS_RES, O_PL>;        // staging split by operand between the wave groups
    constexpr int lds = HsCfg<1, 64>::LDS;
    if (configured.pending()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return fail(RNAMSM_ERR_HIP, "gemm16_q16s: hipFuncSetAttribute: %s", hipGetErrorString(e));
        configured.mark();
    }
    const int group = tuning().gemm_group > 0 ? tuning().gemm_group
                                              : (N / HX_BN > 4 ? (int)xcd_group_for_persistent((M + HX_BM - 1) / HX_BM, 8) : 1);
    const unsigned total = xcd_panel_grid_grouped((M + HX_BM - 1) / HX_BM, N / HX_BN, (unsigned)group);
    const unsigned grid = GEMM16_PERSISTENT_BLOCKS < total ? GEMM16_PERSISTENT_BLOCKS : total;
    KernelTimer timer(TC_GEMM, 2.0 * M * N * K, gemm16_bytes(M, N, K, 1, HAS_RES, O_PL), stream, PEAK_F16_MFMA_TFLOPS, 1.0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(HX_THREADS), lds, stream, a_hi, lda, Whi, bias, residual, ldr, Cout, ldc, M, N, K,
                       scale, scale_cols, o_hi, group, total, fa);
    RNAMSM_CHECK_LAUNCH("gemm16_q16");
    return RNAMSM_OK;
}

// LayerNorm whose output goes straight into 16-bit hi/lo planes (the A operand of the following matrix-core GEMM):
// same arithmetic as layernorm_kernel (elementwise.hip), only the store differs.
template <int FMT>
__global__ __launch_bounds__(256) void layernorm_split_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, uint16_t* __restrict__ hi,
                                                              uint16_t* __restrict__ lo, int64_t T, int D, float eps) {
    typedef typename Half16<FMT>::T H;
    typedef H H4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, nvec = D / 4;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < T; row += stride) {
        f32x4 v[4];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (lane + 64 * e < nvec) {
                v[e] = *reinterpret_cast<const f32x4*>(x + row * D + 4 * (lane + 64 * e));
                s += (v[e][0] + v[e][1]) + (v[e][2] + v[e][3]);
            }
        const float mean = wave_sum(s) / (float)D;
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (lane + 64 * e < nvec) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float d = v[e][i] - mean;
                    ss += d * d;
                }
            }
        const float rstd = rsqrtf(wave_sum(ss) / (float)D + eps);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int vi = lane + 64 * e;
            if (vi < nvec) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * vi);
                const f32x4 b = *reinterpret_cast<const f32x4*>(beta + 4 * vi);
                H4 h, l;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float y = pinned((v[e][i] - mean) * rstd * g[i] + b[i]);
                    h[i] = (H)y;
                    l[i] = (H)(y - (float)h[i]);
                }
                *reinterpret_cast<H4*>(hi + row * D + 4 * vi) = h;
                if (lo) *reinterpret_cast<H4*>(lo + row * D + 4 * vi) = l;
            }
        }
    }
}

template <int FMT>
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ hi,
                                                         uint16_t* __restrict__ lo, int64_t n) {
    typedef typename Half16<FMT>::T H;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float x = pinned(src[i]);
        const H h = (H)x;
        hi[i] = __builtin_bit_cast(uint16_t, h);
        if (lo) lo[i] = __builtin_bit_cast(uint16_t, (H)(x - (float)h));
    }
}

struct HbPlanes {
    const uint16_t *a_hi, *a_lo;
    uint16_t *o_hi, *o_lo;
};

template <int ACT, bool HAS_RES, int SPLIT, int FMT, bool A_PL = false, bool O_PL = false>
static int launch_hb(const float* A, int64_t lda, const uint16_t* Whi, const uint16_t* Wlo, const float* bias,
                     const float* residual, int64_t ldr, float* Cout, int64_t ldc, int M, int N, int K, float scale,
                     int scale_cols, HbPlanes pl, hipStream_t stream) {
    static DeviceOnce configured;
    auto kern = gemm_bf16_kernel<ACT, HAS_RES, SPLIT, FMT, A_PL, O_PL>;
    constexpr int lds = HbCfg<SPLIT>::LDS;
    if (configured.pending()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return fail(RNAMSM_ERR_HIP, "gemm_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e));
        configured.mark();
    }
    const unsigned grid = xcd_panel_grid((M + HB_BM - 1) / HB_BM, N / HB_BN);
    KernelTimer timer(TC_GEMM, 2.0 * M * N * K,
                      (A_PL ? 2.0 * HbCfg<SPLIT>::NPL : 4.0) * (double)M * K + gemm16_bytes(M, N, 0, HbCfg<SPLIT>::NPL, HAS_RES, O_PL) +
                          2.0 * HbCfg<SPLIT>::NPL * (double)N * K, stream, PEAK_F16_MFMA_TFLOPS, SPLIT);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(HB_THREADS), lds, stream, A, lda, Whi, Wlo, bias, residual, ldr, Cout, ldc, M,
                       N, K, scale, scale_cols, pl.a_hi, pl.a_lo, pl.o_hi, pl.o_lo);
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
                                int act, float scale, int scale_cols, int split, int fmt, const uint16_t* A_hi,
                                const uint16_t* A_lo, uint16_t* O_hi, uint16_t* O_lo, void* stream) {
    RNAMSM_CHECK_ARG((A || A_hi) && W_hi && (Cout || O_hi), "gemm_bf16: null pointer");
    RNAMSM_CHECK_ARG(!A_hi || (split == 1 || A_lo), "gemm_bf16: split 3 needs the A lo plane");
    RNAMSM_CHECK_ARG(!O_hi || (!residual && A_hi && (split == 1 || O_lo)),
                     "gemm_bf16: plane output is built for the fc1 / QKV shapes (no residual, plane input)");
    RNAMSM_CHECK_ARG(!A_hi || O_hi || act == RNAMSM_ACT_NONE, "gemm_bf16: plane input with fp32 output supports act none only");
    RNAMSM_CHECK_ARG(split == 1 || (split == 3 && W_lo), "gemm_bf16: split must be 1, or 3 with a lo plane");
    RNAMSM_CHECK_ARG((fmt == 0) || (fmt == 1 && split == 3), "gemm_bf16: fmt 0 (bf16) or 1 (fp16, split 3 only)");
    RNAMSM_NO_BF16X3(split == 3 && fmt == 0, "gemm_bf16");
    RNAMSM_CHECK_ARG(M > 0 && M <= INT32_MAX && N > 0 && K > 0, "gemm_bf16: bad shape");
    RNAMSM_CHECK_ARG(N % HB_BN == 0 && K % HB_BK == 0, "gemm_bf16: need N %% 128 == 0 and K %% 64 == 0 (N=%d K=%d)", N, K);
    RNAMSM_CHECK_ARG(lda >= K && lda % 8 == 0 && ldc >= N && ldc % 4 == 0, "gemm_bf16: bad leading dimension");
    RNAMSM_CHECK_ARG((A_hi ? aligned16(A_hi) && (!A_lo || aligned16(A_lo)) : aligned16(A)) && aligned16(W_hi) &&
                     (O_hi ? (reinterpret_cast<uintptr_t>(O_hi) & 7u) == 0 : aligned16(Cout)) && (!W_lo || aligned16(W_lo)),
                     "gemm_bf16: alignment");
    RNAMSM_CHECK_ARG(!residual || (ldr >= N && ldr % 4 == 0 && aligned16(residual)), "gemm_bf16: bad residual");
    RNAMSM_CHECK_ARG(act == RNAMSM_ACT_NONE || act == RNAMSM_ACT_GELU_ERF, "gemm_bf16: unknown activation %d", act);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int m = (int)M;
    const HbPlanes pl{A_hi, A_lo, O_hi, O_lo};
#define HB_GO(ACT_, RES_, SP_, FMT_, APL_, OPL_) \
    launch_hb<ACT_, RES_, SP_, FMT_, APL_, OPL_>(A, lda, W_hi, W_lo, bias, residual, ldr, Cout, ldc, m, N, K, scale, scale_cols, pl, s)
#define HS_GO(ACT_, RES_, SP_, FMT_, OPL_) \
    ((SP_ == 1 && tuning().gemm16_dma != 4 && K % 64 == 0)                                                             \
         ? launch_hs<ACT_, RES_, SP_, FMT_, OPL_, (SP_ == 1 ? 64 : 32)>(W_hi, W_lo, bias, residual, ldr, Cout, ldc, lda, m, N, K, scale, scale_cols, A_hi, A_lo, O_hi, O_lo, s) \
         : launch_hs<ACT_, RES_, SP_, FMT_, OPL_, 32>(W_hi, W_lo, bias, residual, ldr, Cout, ldc, lda, m, N, K, scale, scale_cols, A_hi, A_lo, O_hi, O_lo, s))
#define HQ_GO(ACT_, RES_, OPL_) \
    launch_hq<ACT_, RES_, OPL_>(W_hi, bias, residual, ldr, Cout, ldc, lda, m, N, K, scale, scale_cols, A_hi, O_hi, s)
    // rows from which the 256x256-tile kernels take over from the 128x128 one (rnamsm_forward raises it, see there)
    const int64_t big_rows = gemm16_big_rows_now();
#define HD_GO(ACT_, RES_, SP_, FMT_, OPL_) \
    launch_hd<ACT_, RES_, SP_, FMT_, OPL_>(W_hi, W_lo, bias, residual, ldr, Cout, ldc, lda, m, N, K, scale, scale_cols, A_hi, A_lo, O_hi, O_lo, s)
#define HB_ACT_RES(SP_, FMT_)                                                                                       \
    do {                                                                                                            \
        /* 16x16x32 MFMAs: round 2 +6.5 % on QKV, +1.4 % on fc1, -2.5 % on out_proj, 0 on fc2 (one process, cfg3 shapes): wide N only;   \
           round 3, staged by operand (gemm16_q16s_kernel): QKV +18 %, fc1 +14 %, fc2 +6.6 % (long K), out_proj -14 %           */ \
        if (SP_ == 1 && A_hi && (tuning().gemm16_mfma16 == 1 ? (N / HX_BN > 4 || K >= 2048) : tuning().gemm16_mfma16 == 2) && tuning().gemm16_dma >= 3 && N % HX_BN == 0 && m >= big_rows && K % 64 == 0) { \
            if (O_hi) return act == RNAMSM_ACT_GELU_ERF ? HQ_GO(RNAMSM_ACT_GELU_ERF, false, true)                    \
                                                        : HQ_GO(RNAMSM_ACT_NONE, false, true);                       \
            return residual ? HQ_GO(RNAMSM_ACT_NONE, true, false) : HQ_GO(RNAMSM_ACT_NONE, false, false);            \
        }                                                                                                           \
        if (A_hi && tuning().gemm16_dma >= 3 && N % HX_BN == 0 && m >= big_rows) {   /* software-pipelined fragments */    \
            if (O_hi) return act == RNAMSM_ACT_GELU_ERF ? HS_GO(RNAMSM_ACT_GELU_ERF, false, SP_, FMT_, true)        \
                                                        : HS_GO(RNAMSM_ACT_NONE, false, SP_, FMT_, true);           \
            return residual ? HS_GO(RNAMSM_ACT_NONE, true, SP_, FMT_, false) : HS_GO(RNAMSM_ACT_NONE, false, SP_, FMT_, false); \
        }                                                                                                           \
        if (A_hi && tuning().gemm16_dma) {                                                                          \
            if (O_hi) return act == RNAMSM_ACT_GELU_ERF ? HD_GO(RNAMSM_ACT_GELU_ERF, false, SP_, FMT_, true)        \
                                                        : HD_GO(RNAMSM_ACT_NONE, false, SP_, FMT_, true);           \
            return residual ? HD_GO(RNAMSM_ACT_NONE, true, SP_, FMT_, false) : HD_GO(RNAMSM_ACT_NONE, false, SP_, FMT_, false); \
        }                                                                                                           \
        if (O_hi) return act == RNAMSM_ACT_GELU_ERF ? HB_GO(RNAMSM_ACT_GELU_ERF, false, SP_, FMT_, true, true)      \
                                                    : HB_GO(RNAMSM_ACT_NONE, false, SP_, FMT_, true, true);         \
        if (A_hi) return residual ? HB_GO(RNAMSM_ACT_NONE, true, SP_, FMT_, true, false)                            \
                                  : HB_GO(RNAMSM_ACT_NONE, false, SP_, FMT_, true, false);                          \
        if (act == RNAMSM_ACT_GELU_ERF)                                                                             \
            return residual ? HB_GO(RNAMSM_ACT_GELU_ERF, true, SP_, FMT_, false, false)                             \
                            : HB_GO(RNAMSM_ACT_GELU_ERF, false, SP_, FMT_, false, false);                           \
        return residual ? HB_GO(RNAMSM_ACT_NONE, true, SP_, FMT_, false, false)                                     \
                        : HB_GO(RNAMSM_ACT_NONE, false, SP_, FMT_, false, false);                                   \
    } while (0)
    if (fmt == 1) HB_ACT_RES(3, 1);
    HB_ACT_RES(1, 0);
#undef HB_ACT_RES
#undef HD_GO
#undef HQ_GO
#undef HS_GO
#undef HB_GO
}

// K1 folded, 16-bit modes: the two entry points of the 256x256 kernels with Fold16 set.
extern "C" int rnamsm_gemm16_lnfold(const uint16_t* X_hi, const uint16_t* X_lo, int64_t ldx, const uint16_t* Wg_hi,
                                    const uint16_t* Wg_lo, const float* cvec, const float* dvec, const float* row_stats,
                                    uint16_t* O_hi, uint16_t* O_lo, int64_t ldo, int64_t M, int N, int K, int act, float scale,
                                    int scale_cols, int split, int fmt, void* stream) {
    RNAMSM_CHECK_ARG(X_hi && Wg_hi && cvec && dvec && row_stats && O_hi, "gemm16_lnfold: null pointer");
    RNAMSM_CHECK_ARG(split == 1 || (split == 3 && X_lo && Wg_lo && O_lo), "gemm16_lnfold: split must be 1, or 3 with lo planes");
    RNAMSM_CHECK_ARG((fmt == 0) || (fmt == 1 && split == 3), "gemm16_lnfold: fmt 0 (bf16) or 1 (fp16, split 3 only)");
    RNAMSM_CHECK_ARG(M >= 2048 && M <= INT32_MAX && N > 0 && N % HX_BN == 0 && K > 0 && K % 64 == 0,
                     "gemm16_lnfold: the 256x256 kernels need M >= 2048, N %% 256 == 0, K %% 64 == 0 (M=%lld N=%d K=%d)", (long long)M, N, K);
    RNAMSM_CHECK_ARG(ldx >= K && ldx % 8 == 0 && ldo >= N && ldo % 4 == 0 && scale_cols >= 0 && scale_cols % 4 == 0,
                     "gemm16_lnfold: bad leading dimension / scale_cols");
    RNAMSM_CHECK_ARG(aligned16(X_hi) && (!X_lo || aligned16(X_lo)) && aligned16(Wg_hi) && (!Wg_lo || aligned16(Wg_lo)) &&
                     aligned16(cvec) && aligned16(dvec) && (reinterpret_cast<uintptr_t>(row_stats) & 7u) == 0 &&
                     (reinterpret_cast<uintptr_t>(O_hi) & 7u) == 0 && (!O_lo || (reinterpret_cast<uintptr_t>(O_lo) & 7u) == 0),
                     "gemm16_lnfold: alignment");
    RNAMSM_CHECK_ARG(act == RNAMSM_ACT_NONE || act == RNAMSM_ACT_GELU_ERF, "gemm16_lnfold: unknown activation %d", act);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int m = (int)M;
    Fold16 fa;
    fa.c = cvec;
    fa.stats = reinterpret_cast<const float2*>(row_stats);
#define LF_HS(ACT_, SP_, FMT_, BK_) \
    launch_hs<ACT_, false, SP_, FMT_, true, BK_>(Wg_hi, Wg_lo, dvec, nullptr, 0, nullptr, ldo, ldx, m, N, K, scale, scale_cols, X_hi, X_lo, O_hi, O_lo, s, fa)
#define LF_HQ(ACT_) launch_hq<ACT_, false, true>(Wg_hi, dvec, nullptr, 0, nullptr, ldo, ldx, m, N, K, scale, scale_cols, X_hi, O_hi, s, fa)
    const bool gelu = act == RNAMSM_ACT_GELU_ERF;
    if (split == 1) {
        if (tuning().gemm16_mfma16 == 1 ? N / HX_BN > 4 : tuning().gemm16_mfma16 == 2)
            return gelu ? LF_HQ(RNAMSM_ACT_GELU_ERF) : LF_HQ(RNAMSM_ACT_NONE);
        return gelu ? LF_HS(RNAMSM_ACT_GELU_ERF, 1, 0, 64) : LF_HS(RNAMSM_ACT_NONE, 1, 0, 64);
    }
    RNAMSM_NO_BF16X3(fmt == 0, "gemm16_lnfold");
    return gelu ? LF_HS(RNAMSM_ACT_GELU_ERF, 3, 1, 32) : LF_HS(RNAMSM_ACT_NONE, 3, 1, 32);
#undef LF_HQ
#undef LF_HS
}

extern "C" int rnamsm_gemm16_residual_stats(const uint16_t* A_hi, const uint16_t* A_lo, int64_t lda, const uint16_t* W_hi,
                                            const uint16_t* W_lo, const float* bias, float* x, int64_t ldx, int64_t M, int N,
                                            int K, int split, int fmt, uint16_t* X_hi, uint16_t* X_lo, int64_t ldp,
                                            float* row_partials, int64_t partials_ld, void* stream) {
    RNAMSM_CHECK_ARG(A_hi && W_hi && x && X_hi && row_partials, "gemm16_residual_stats: null pointer");
    RNAMSM_CHECK_ARG(split == 1 || (split == 3 && A_lo && W_lo && X_lo), "gemm16_residual_stats: split must be 1, or 3 with lo planes");
    RNAMSM_CHECK_ARG((fmt == 0) || (fmt == 1 && split == 3), "gemm16_residual_stats: fmt 0 (bf16) or 1 (fp16, split 3 only)");
    RNAMSM_CHECK_ARG(M >= 2048 && M <= INT32_MAX && N > 0 && N % HX_BN == 0 && K > 0 && K % 64 == 0,
                     "gemm16_residual_stats: the 256x256 kernel needs M >= 2048, N %% 256 == 0, K %% 64 == 0 (M=%lld N=%d K=%d)", (long long)M, N, K);
    RNAMSM_CHECK_ARG(lda >= K && lda % 8 == 0 && ldx >= N && ldx % 4 == 0 && ldp >= N && ldp % 4 == 0 && partials_ld >= M,
                     "gemm16_residual_stats: bad leading dimension");
    RNAMSM_CHECK_ARG(aligned16(A_hi) && (!A_lo || aligned16(A_lo)) && aligned16(W_hi) && (!W_lo || aligned16(W_lo)) && aligned16(x) &&
                     (reinterpret_cast<uintptr_t>(X_hi) & 7u) == 0 && (!X_lo || (reinterpret_cast<uintptr_t>(X_lo) & 7u) == 0) &&
                     (reinterpret_cast<uintptr_t>(row_partials) & 7u) == 0, "gemm16_residual_stats: alignment");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int m = (int)M;
    Fold16 fa;
    fa.partials = reinterpret_cast<float2*>(row_partials);
    fa.pld = partials_ld;
    fa.xhi = X_hi;
    fa.xlo = X_lo;
    fa.ldx = ldp;
#define RS_HS(SP_, FMT_, BK_) \
    launch_hs<RNAMSM_ACT_NONE, true, SP_, FMT_, false, BK_>(W_hi, W_lo, bias, x, ldx, x, ldx, lda, m, N, K, 1.f, 0, A_hi, A_lo, nullptr, nullptr, s, fa)
    if (split == 1) return RS_HS(1, 0, 64);
    RNAMSM_NO_BF16X3(fmt == 0, "gemm16_residual_stats");
    return RS_HS(3, 1, 32);
#undef RS_HS
}

extern "C" int rnamsm_layernorm_split(const float* x, const float* gamma, const float* beta, uint16_t* hi, uint16_t* lo,
                                      int64_t T, int D, float eps, int fmt, void* stream) {
    RNAMSM_CHECK_ARG(x && gamma && beta && hi, "layernorm_split: null pointer");
    RNAMSM_CHECK_ARG(T > 0 && D > 0 && D % 4 == 0 && D <= 1024 && (fmt == 0 || fmt == 1), "layernorm_split: bad shape / fmt");
    RNAMSM_CHECK_ARG(aligned16(x) && aligned16(gamma) && aligned16(beta) && (reinterpret_cast<uintptr_t>(hi) & 7u) == 0 &&
                     (!lo || (reinterpret_cast<uintptr_t>(lo) & 7u) == 0), "layernorm_split: alignment");
    const int64_t blocks = (T + 3) / 4;
    const dim3 grid((unsigned)(blocks < 4096 ? blocks : 4096));
    KernelTimer timer(TC_LAYERNORM, 0.0, (4.0 + (lo ? 4.0 : 2.0)) * T * D, static_cast<hipStream_t>(stream), PEAK_F16_MFMA_TFLOPS);
    if (fmt == 1)
        hipLaunchKernelGGL(layernorm_split_kernel<1>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, gamma, beta, hi, lo, T, D, eps);
    else
        hipLaunchKernelGGL(layernorm_split_kernel<0>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, gamma, beta, hi, lo, T, D, eps);
    RNAMSM_CHECK_LAUNCH("layernorm_split");
    return RNAMSM_OK;
}
