// K4' / K6': tied row attention on the 16-bit matrix cores (the 16-bit modes of RowSelfAttention, modules.py:688-821).
//
// Same contractions as row_attn.hip, on operands that already live in HBM as 16-bit planes: the QKV GEMM epilogue
// writes q (unscaled) | k | v as hi (+lo) planes [T, 3D], the softmax kernel writes P as hi (+lo) planes [H*C, ldp].
// SPLIT 1 = one bf16 MFMA per product; SPLIT 3 = hi*hi + hi*lo + lo*hi on hi/lo pairs (bf16x3 / f16x3, see half16.h).
//
//  K4' row_logits16 : S[h,i,j] = scale * sum_{r,d} q[r,i,h,d] k[r,j,h,d].  128x128 output tile, K tile = one alignment row r
//                     (64 head dims = one 128-B run per alignment column): both operands are "k" tiles of tile16.h,
//                     DMA-staged, read with ds_read_b128 -- the loop of gemm16_dma_kernel with a different address map.
//  K6' row_apply16  : ctx[r,i,h,:] = out_scale * sum_j P[h,i,j] v[r,j,h,:].  A = P rows ("k" tile, k = j); B = v for two alignment
//                     rows, staged as it lies in memory ([key j][64 d], a "t" tile) and read TRANSPOSED by
//                     ds_read_b64_tr_b16, so no transposed copy of V is ever made.
// Scaling happens on the fp32 accumulators, not on the operands: q scaled by dh^-0.5 / sqrt(R) (~1e-2) would push its
// fp16 lo plane into subnormals (absolute step 6e-8, only ~2^-18 of q), and probabilities are stored as P * 2^12 for the
// same reason (exact power of two, undone by out_scale).
// Roofline: MFMA-bound like the fp32 kernels (same flops, 16-bit rate x 1 or / 3).
#include <type_traits>

#include "row_split.h"
#include "tile16.h"

namespace rnamsm {

constexpr int R16_THREADS = 256;
constexpr int R16_PLANE = 128 * T16_ROWB;          // one [128][64 halves] operand plane tile = 16 KB
constexpr int R16_STAGE = 4 * 64 * 68 * 4;         // slab_store_64x64 staging (row_apply16 epilogue)
template <int SPLIT>
struct R16Cfg {
    static constexpr int NPL = SPLIT == 3 ? 2 : 1;
    static constexpr int BUF = 2 * NPL * R16_PLANE;                       // A planes then B planes
    static constexpr int LDS = 2 * BUF > R16_STAGE ? 2 * BUF : R16_STAGE;
};

template <int SPLIT, int FMT>
struct R16Frag {
    typename Half16<FMT>::V8 a[2][SPLIT == 3 ? 2 : 1], b[2][SPLIT == 3 ? 2 : 1];
};

template <int SPLIT, int FMT>
__device__ __forceinline__ void r16_mma(const R16Frag<SPLIT, FMT>& f, f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mma16<SPLIT, FMT>(f.a[mt], f.b[nt], acc[mt][nt]);
}

// ---------------------------------------------------------------------------------------------- K4'
template <int SPLIT, int FMT>
__global__ __launch_bounds__(R16_THREADS, SPLIT == 3 ? 1 : 2) void row_logits16_kernel(
    const uint16_t* __restrict__ qhi, const uint16_t* __restrict__ qlo, const uint16_t* __restrict__ khi,
    const uint16_t* __restrict__ klo, int64_t ld, float* __restrict__ partial, int R, int C, int H, int nsplit,
    int rows_per_split, float scale, int64_t qk_bstride, int64_t part_bstride, const int* __restrict__ true_rows) {
    // batched launch (rnamsm_forward_batch, 16-bit modes): MSA blockIdx.y; a ragged batch scales every MSA's logits by its own depth
    qhi += blockIdx.y * qk_bstride;
    khi += blockIdx.y * qk_bstride;
    if (qlo) qlo += blockIdx.y * qk_bstride;
    if (klo) klo += blockIdx.y * qk_bstride;
    partial += blockIdx.y * part_bstride;
    if (true_rows) scale = scale / sqrtf((float)max(true_rows[blockIdx.y], 1));
    using Cfg = R16Cfg<SPLIT>;
    constexpr int NPL = Cfg::NPL;
    extern __shared__ __attribute__((aligned(16))) char smem_b[];

    const unsigned tiles_c = (C + 127) / 128;
    unsigned panel, tile;
    if (!xcd_panel_map(blockIdx.x, (unsigned)(H * nsplit), tiles_c * tiles_c, panel, tile)) return;
    const int h = panel / nsplit, split = panel % nsplit;
    const int i0 = (tile / tiles_c) * 128, j0 = (tile % tiles_c) * 128;
    const int r_begin = split * rows_per_split;
    const int r_end = min(R, r_begin + rows_per_split);

    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wv >> 1, wn = wv & 1, li = lane & 31, lh = lane >> 5;

    // DMA map: wave wv moves row groups wv, wv+4, wv+8, wv+12 (8 rows each) of every plane tile; a clamped alignment
    // column only feeds discarded outputs.
    const int drow = lane >> 3;
    const int dchunk = dma_chunk_k(lane, wv);                  // 4*(wv+4j) = 4*wv (mod 8)
    int64_t qoff[4], koff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 8 * (wv + 4 * j) + drow;
        qoff[j] = (int64_t)min(i0 + row, C - 1) * ld + h * 64 + dchunk * 8;
        koff[j] = (int64_t)min(j0 + row, C - 1) * ld + h * 64 + dchunk * 8;
    }
    auto issue = [&](int kt, int buf) {
        char* base = smem_b + buf * Cfg::BUF;
        const int64_t rb = (int64_t)(r_begin + kt) * C * ld;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int loff = (8 * (wv + 4 * j)) * T16_ROWB;
            dma16(qhi + rb + qoff[j], base + loff);
            if (SPLIT == 3) dma16(qlo + rb + qoff[j], base + R16_PLANE + loff);
            dma16(khi + rb + koff[j], base + NPL * R16_PLANE + loff);
            if (SPLIT == 3) dma16(klo + rb + koff[j], base + (NPL + 1) * R16_PLANE + loff);
        }
    };
    auto frags = [&](const char* cur, int kk, R16Frag<SPLIT, FMT>& f) {
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f.a[t][p] = frag_k<FMT>(cur + p * R16_PLANE, wm * 64 + t * 32 + li, kk, lh);
                f.b[t][p] = frag_k<FMT>(cur + (NPL + p) * R16_PLANE, wn * 64 + t * 32 + li, kk, lh);
            }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[mt][nt][t] = 0.f;

    const int nk = r_end - r_begin;          // K tile kt = alignment row r_begin + kt
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        wait_dma_then_barrier<0>();          // tile kt has landed (every wave's share), the other buffer is free again
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        const char* cur = smem_b + (kt & 1) * Cfg::BUF;
        R16Frag<SPLIT, FMT> f0, f1;
        frags(cur, 0, f0);
        frags(cur, 1, f1);
        r16_mma<SPLIT, FMT>(f0, acc);
        __builtin_amdgcn_sched_barrier(0);
        frags(cur, 2, f0);
        r16_mma<SPLIT, FMT>(f1, acc);
        __builtin_amdgcn_sched_barrier(0);
        frags(cur, 3, f1);
        r16_mma<SPLIT, FMT>(f0, acc);
        __builtin_amdgcn_sched_barrier(0);
        r16_mma<SPLIT, FMT>(f1, acc);
    }

    float* out = partial + ((int64_t)split * H + h) * C * C;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int j = j0 + wn * 64 + nt * 32 + li;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int i = i0 + wm * 64 + mt * 32 + (t & 3) + 8 * (t >> 2) + 4 * lh;
                if (i < C && j < C) out[(int64_t)i * C + j] = acc[mt][nt][t] * scale;
            }
    }
}

// ---------------------------------------------------------------------------------------------- K6'
// grid.x = xcd-mapped (panel = (head, pair of alignment rows), inner = tiles_i) as in row_apply_kernel.
// The B region of a buffer is [2 rows r][64 keys][128 B] per plane: wave column wn reads alignment row rr0 + wn.
template <int SPLIT, int FMT, int OUT>
__global__ __launch_bounds__(R16_THREADS, SPLIT == 3 ? 1 : 2) void row_apply16_kernel(
    const uint16_t* __restrict__ phi, const uint16_t* __restrict__ plo, int64_t ldp, const uint16_t* __restrict__ vhi,
    const uint16_t* __restrict__ vlo, int64_t ld, float* __restrict__ ctx, int64_t ldc, int R, int C, int H,
    uint16_t* __restrict__ ctx_hi, uint16_t* __restrict__ ctx_lo, float out_scale, int64_t p_bstride, int64_t v_bstride,
    int64_t ctx_bstride) {
    phi += blockIdx.y * p_bstride;                           // batched launch: MSA blockIdx.y
    vhi += blockIdx.y * v_bstride;
    if (plo) plo += blockIdx.y * p_bstride;
    if (vlo) vlo += blockIdx.y * v_bstride;
    if (ctx) ctx += blockIdx.y * ctx_bstride;
    if (ctx_hi) ctx_hi += blockIdx.y * ctx_bstride;
    if (ctx_lo) ctx_lo += blockIdx.y * ctx_bstride;
    using Cfg = R16Cfg<SPLIT>;
    constexpr int NPL = Cfg::NPL;
    extern __shared__ __attribute__((aligned(16))) char smem_b[];

    const unsigned tiles_i = (C + 127) / 128, tiles_n = (R + 1) / 2;
    unsigned panel, ti;
    if (!xcd_panel_map(blockIdx.x, (unsigned)H * tiles_n, tiles_i, panel, ti)) return;
    const int h = panel / tiles_n, rr0 = (panel % tiles_n) * 2;
    const int i0 = ti * 128;

    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wv >> 1, wn = wv & 1, li = lane & 31, lh = lane >> 5;

    const int drow = lane >> 3;
    const int ck = dma_chunk_k(lane, wv), ct = dma_chunk_t(lane);
    int64_t poff[4], voff[4];
    int vkey[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 8 * (wv + 4 * j) + drow;                       // P: tile row i; V: (r_local = row / 64, key = row % 64)
        poff[j] = ((int64_t)h * C + min(i0 + row, C - 1)) * ldp + ck * 8;
        const int r = min(rr0 + (row >> 6), R - 1);                    // second row of an odd R is discarded
        voff[j] = (int64_t)r * C * ld + h * 64 + ct * 8;
        vkey[j] = row & 63;
    }
    auto issue = [&](int kt, int buf) {
        char* base = smem_b + buf * Cfg::BUF;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int loff = (8 * (wv + 4 * j)) * T16_ROWB;
            // keys past C: P is zero-padded there, V is clamped (finite) -> contributes exactly 0
            const int64_t vo = voff[j] + (int64_t)min(kt * 64 + vkey[j], C - 1) * ld;
            dma16(phi + poff[j] + kt * 64, base + loff);
            if (SPLIT == 3) dma16(plo + poff[j] + kt * 64, base + R16_PLANE + loff);
            dma16(vhi + vo, base + NPL * R16_PLANE + loff);
            if (SPLIT == 3) dma16(vlo + vo, base + (NPL + 1) * R16_PLANE + loff);
        }
    };
    // transposed-read geometry of this lane (tile16.h): addresses row q of the 4-row block, columns 4p..4p+3
    const int tq = (lane & 15) >> 2, tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    // B pieces are transposed reads issued as asm (tile16.h): usable after the counted wait in mma_t
    struct FragT {
        R16Frag<SPLIT, FMT> f;
        TrPieces bp[2][NPL];
    };
    auto frags = [&](const char* cur, int kk, FragT& ft) {
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ft.f.a[t][p] = frag_k<FMT>(cur + p * R16_PLANE, wm * 64 + t * 32 + li, kk, lh);
                // B[k = key][n = d]: lane (d = t*32 + li, lh) needs keys 16kk + 8lh + 0..7
                const int ka = 16 * kk + 8 * lh + tq;
                ft.bp[t][p] = tr16_issue(cur + (NPL + p) * R16_PLANE + wn * 64 * T16_ROWB, ka, ka + 4, t * 32 + tcol);
            }
    };
    // YOUNGER: transposed reads requested after this set's (4 per plane per set) -- LDS answers in order
    auto mma_t = [&](FragT& ft, f32x16 (&acc)[2][2], auto younger_tag) {
        constexpr int YOUNGER = decltype(younger_tag)::value;
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
            tr16_wait2<YOUNGER>(ft.bp[0][p], ft.bp[1][p]);
#pragma unroll
            for (int t = 0; t < 2; ++t) ft.f.b[t][p] = tr16_frag<FMT>(ft.bp[t][p]);
        }
        r16_mma<SPLIT, FMT>(ft.f, acc);
    };
    typedef std::integral_constant<int, 4 * NPL> next_set_t;
    typedef std::integral_constant<int, 0> none_t;

    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[mt][nt][t] = 0.f;

    const int nk = (C + 63) / 64;
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        wait_dma_then_barrier<0>();
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        const char* cur = smem_b + (kt & 1) * Cfg::BUF;
        FragT f0, f1;
        frags(cur, 0, f0);
        frags(cur, 1, f1);
        mma_t(f0, acc, next_set_t());
        __builtin_amdgcn_sched_barrier(0);
        frags(cur, 2, f0);
        mma_t(f1, acc, next_set_t());
        __builtin_amdgcn_sched_barrier(0);
        frags(cur, 3, f1);
        mma_t(f0, acc, next_set_t());
        __builtin_amdgcn_sched_barrier(0);
        mma_t(f1, acc, none_t());
    }

#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[mt][nt][t] *= out_scale;
    // wave slab = alignment row rr0 + wn, alignment columns i0 + wm*64 .. +63, 64 head dims -> 256-B / 128-B segments
    wait_dma_then_barrier<0>();
    const int r = rr0 + wn;
    const int ibase = i0 + wm * 64;
    auto rowoff = [&](int row) -> int64_t {
        const int i = ibase + row;
        return (r < R && i < C) ? ((int64_t)r * C + i) * ldc + h * 64 : (int64_t)-1;
    };
    slab_store_64x64<OUT>(acc, reinterpret_cast<float*>(smem_b) + wv * (64 * 68), li, lh, lane, rowoff, ctx, ctx_hi, ctx_lo);
}

// ---------------------------------------------------------------------------------------------- K4' (large C)
// 256x256 tile version of row_logits16 for C >= 256 (half the operand bytes per flop, see K6' (large C) below): both
// operands are "k" tiles, i.e. exactly the loop of gemm16_swp_kernel with q / k planes for A / W.  One block per CU, so the
// row split is chosen for 256 slots.  K tile: BK = 32 -- half of one alignment row's head dims, 64-B tile rows (the hi/lo
// modes: four planes per stage) -- or BK = 64 -- one whole alignment row, 128-B tile rows = whole cache lines per DMA row,
// half the barriers (plain bf16 only: two 64 KB stages; round 3).
template <int SPLIT, int BK>
struct R16LCfg {
    static constexpr int NPL = SPLIT == 3 ? 2 : 1;
    static constexpr int ROWB = BK * 2;
    static constexpr int PLANE = 256 * ROWB;            // [256 rows][BK halves]
    static constexpr int BUF = 2 * NPL * PLANE;
    static constexpr int LDS = 2 * BUF;
    static constexpr int RPI = 1024 / ROWB;              // tile rows per wave DMA instruction
    static constexpr int IPW = 256 / RPI / 8;            // DMA instructions per wave per plane tile
    static constexpr int KS = BK / 16;                   // MFMA k steps per tile
};
template <int SPLIT, int FMT>
struct R16LFrag {
    typename Half16<FMT>::V8 a[SPLIT == 3 ? 2 : 1][4], b[SPLIT == 3 ? 2 : 1][2];
};

template <int SPLIT, int FMT, int BK>
__global__ __launch_bounds__(512, 1) void row_logits16x_kernel(
    const uint16_t* __restrict__ qhi, const uint16_t* __restrict__ qlo, const uint16_t* __restrict__ khi,
    const uint16_t* __restrict__ klo, int64_t ld, float* __restrict__ partial, int R, int C, int H, int nsplit,
    int rows_per_split, float scale, int64_t qk_bstride, int64_t part_bstride, const int* __restrict__ true_rows) {
    // batched launch (rnamsm_forward_batch, 16-bit modes): MSA blockIdx.y; a ragged batch scales every MSA's logits by its own depth
    qhi += blockIdx.y * qk_bstride;
    khi += blockIdx.y * qk_bstride;
    if (qlo) qlo += blockIdx.y * qk_bstride;
    if (klo) klo += blockIdx.y * qk_bstride;
    partial += blockIdx.y * part_bstride;
    if (true_rows) scale = scale / sqrtf((float)max(true_rows[blockIdx.y], 1));
    using Cfg = R16LCfg<SPLIT, BK>;
    constexpr int NPL = Cfg::NPL, ROWB = Cfg::ROWB, PLANE = Cfg::PLANE, KS = Cfg::KS;
    constexpr int NMF = 8 * (SPLIT == 3 ? 3 : 1);
    constexpr int NDS = 6 * NPL;
    static_assert(BK == 32 || (BK == 64 && SPLIT == 1), "64-deep K tiles: two 64 KB stages fit for one plane per operand only");
    typedef typename Half16<FMT>::V8 V8;
    extern __shared__ __attribute__((aligned(16))) char smem_b[];

    const unsigned tiles_c = (C + 255) / 256;
    unsigned panel, tile;
    if (!xcd_panel_map(blockIdx.x, (unsigned)(H * nsplit), tiles_c * tiles_c, panel, tile)) return;
    const int h = panel / nsplit, split = panel % nsplit;
    const int i0 = (tile / tiles_c) * 256, j0 = (tile % tiles_c) * 256;
    const int r_begin = split * rows_per_split;
    const int r_end = min(R, r_begin + rows_per_split);

    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wv >> 2, wn = wv & 3, li = lane & 31, lh = lane >> 5;

    // DMA map: a wave instruction covers RPI tile rows (16 of 64 B, or 8 of 128 B); lane -> (row RPI g + lane / chunks-per-row,
    // physical chunk lane % chunks-per-row) fetching the logical chunk the read-side swizzle expects there; g = wv + 8 j
    constexpr int CPR = ROWB / 16;
    const int dchunk = BK == 32 ? ((lane & 3) ^ ((lane >> 4) & 3))                      // (row >> 2) & 3
                                : ((lane & 7) ^ ((4 * (wv & 1) + (lane >> 4)) & 7));    // (row >> 1) & 7, row = 8 g + lane / 8
    int64_t qoff[Cfg::IPW], koff[Cfg::IPW];
#pragma unroll
    for (int j = 0; j < Cfg::IPW; ++j) {
        const int row = Cfg::RPI * (wv + 8 * j) + lane / CPR;
        qoff[j] = (int64_t)min(i0 + row, C - 1) * ld + h * 64 + dchunk * 8;     // clamped columns feed discarded outputs
        koff[j] = (int64_t)min(j0 + row, C - 1) * ld + h * 64 + dchunk * 8;
    }
    // K tile kt: BK = 32 -> (alignment row r_begin + kt/2, d half kt&1); BK = 64 -> alignment row r_begin + kt
    auto issue = [&](int kt, int buf) {
        char* base = smem_b + buf * Cfg::BUF;
        const int64_t rb = BK == 32 ? (int64_t)(r_begin + (kt >> 1)) * C * ld + (kt & 1) * 32 : (int64_t)(r_begin + kt) * C * ld;
#pragma unroll
        for (int j = 0; j < Cfg::IPW; ++j) {
            const int loff = (Cfg::RPI * (wv + 8 * j)) * ROWB;
            dma16(qhi + rb + qoff[j], base + loff);
            if (SPLIT == 3) dma16(qlo + rb + qoff[j], base + PLANE + loff);
            dma16(khi + rb + koff[j], base + NPL * PLANE + loff);
            if (SPLIT == 3) dma16(klo + rb + koff[j], base + (NPL + 1) * PLANE + loff);
        }
    };
    auto frag_load = [&](const char* buf, int kk, R16LFrag<SPLIT, FMT>& f) {
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
    auto frag_mma = [&](const R16LFrag<SPLIT, FMT>& f, f32x16 (&acc)[4][2]) {
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
    };
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

    const int nk = (BK == 32 ? 2 : 1) * (r_end - r_begin);
    issue(0, 0);
    wait_dma_then_barrier<0>();
    issue(nk > 1 ? 1 : 0, 1);                                // (a redundant reload when nk == 1: never read)
    R16LFrag<SPLIT, FMT> f[2];                               // step kk lives in f[kk & 1]; KS is even
    frag_load(smem_b, 0, f[0]);
    for (int kt = 0; kt + 1 < nk; ++kt) {
        const char* cur = smem_b + (kt & 1) * Cfg::BUF;
        const char* nxt = smem_b + ((kt & 1) ^ 1) * Cfg::BUF;
#pragma unroll
        for (int kk = 0; kk + 1 < KS; ++kk) {
            frag_load(cur, kk + 1, f[(kk + 1) & 1]);
            frag_mma(f[kk & 1], acc);
            interleave();
        }
        // every wave is done reading `cur` once its last fragments have arrived; tile kt+1 (issued one tile ago) must have landed
        wait_dma_then_barrier<0>();
        const int k2 = kt + 2 < nk ? kt + 2 : nk - 1;        // clamped: the last reload is never read
        if (wm == 0) issue(k2, kt & 1);                      // dephased issue: the upper wave group issues one step later (gemm_bf16.hip)
        __builtin_amdgcn_sched_barrier(0);
        frag_load(nxt, 0, f[0]);
        frag_mma(f[(KS - 1) & 1], acc);
        interleave();
        if (wm == 1) issue(k2, kt & 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    {
        const char* cur = smem_b + ((nk - 1) & 1) * Cfg::BUF;
#pragma unroll
        for (int kk = 0; kk + 1 < KS; ++kk) {
            frag_load(cur, kk + 1, f[(kk + 1) & 1]);
            frag_mma(f[kk & 1], acc);
            interleave();
        }
        frag_mma(f[(KS - 1) & 1], acc);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the clamped reload has landed before the block exits

    float* out = partial + ((int64_t)split * H + h) * C * C;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int j = j0 + wn * 64 + nt * 32 + li;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int i = i0 + wm * 128 + mt * 32 + (t & 3) + 8 * (t >> 2) + 4 * lh;
                if (i < C && j < C) out[(int64_t)i * C + j] = acc[mt][nt][t] * scale;
            }
    }
}

// ---------------------------------------------------------------------------------------------- K4' (large C, plain bf16)
// The K loop of gemm16_q16s_kernel (gemm_bf16.hip) on the tied-row logits: per head a [C x R*64] . [R*64 x C] product whose K
// tile is one alignment row r (64 head dims = one 128-byte run per alignment column).  256x256 output tile, 8 waves (2 x 4),
// wave tile 128 x 64 = 8 x 4 tiles of v_mfma_f32_16x16x32_bf16, two 64 KB stages.  STAGING BY OPERAND: the lower wave group
// moves the k tile (the "W" operand), the upper one the q tile, half a tile apart, each burst under the other group's MFMAs
// and with a whole tile of flight time (see gemm16_q16s_kernel for the reasoning and the measurements); persistent blocks walk
// (MSA, head, row slab, tile) and request the next tile's first stage before the current tile's stores.  The k rows are
// permuted on their way into LDS so that a lane's (transposed) accumulators are 8 + 8 consecutive output columns: the
// epilogue stores 16-byte pieces straight from the registers.  Needs C % 8 == 0 (16-byte stores into rows of C floats); other
// alignments keep the 128x128 kernel above.  Round 4: the 128x128 kernel reached 790 TFLOP/s at M = L = 1024 (matrix pipe 47 %
// busy, waves parked in the DMA wait / barrier 41 % of their time).
constexpr int R16Q_ROWB = 128, R16Q_PLANE = 256 * R16Q_ROWB, R16Q_BUF = 2 * R16Q_PLANE, R16Q_LDS = 2 * R16Q_BUF;
typedef float f32x4q __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 1) void row_logits16q_kernel(
    const uint16_t* __restrict__ qhi, const uint16_t* __restrict__ khi, int64_t ld, float* __restrict__ partial, int R, int C, int H,
    int nsplit, int rows_per_split, float scale, int64_t qk_bstride, int64_t part_bstride, const int* __restrict__ true_rows,
    unsigned num_panels, unsigned total_tiles) {
    constexpr int ROWB = R16Q_ROWB, PLANE = R16Q_PLANE, BUF = R16Q_BUF;
    typedef typename Half16<0>::V8 V8;
    extern __shared__ __attribute__((aligned(16))) char smem_b[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wv >> 2, wn = wv & 3, fr = lane & 15, fq = lane >> 4;
    const unsigned tiles_c = (C + 255) / 256;
    const int64_t kstride = (int64_t)C * ld;                  // halves from alignment row r to r + 1
    constexpr int NSTORES = 32;                               // store instructions of one epilogue, per lane

    const int w4 = wv & 3;
    const int drow = lane >> 3;
    const int dchunk = (lane & 7) ^ ((4 * (w4 & 1) + (lane >> 4)) & 7);      // (row >> 1) & 7, row = 8 (w4 + 4 j) + lane / 8
    const int r0 = 8 * w4 + drow;                             // the wave's LDS rows: r0 + 32 j, j = 0..7
    const uint16_t* src = nullptr;                            // this wave group's operand of the current tile
    int64_t off[8];
    struct Tile { int b, h, split, i0, j0, r_begin, nk; };
    auto find_tile = [&](unsigned& vid, Tile& t) -> bool {
        for (; vid < total_tiles; vid += gridDim.x) {
            unsigned panel, tl;
            if (xcd_panel_map(vid, num_panels, tiles_c * tiles_c, panel, tl)) {
                t.b = panel / (H * nsplit);
                const int rest = panel % (H * nsplit);
                t.h = rest / nsplit;
                t.split = rest % nsplit;
                t.i0 = (tl / tiles_c) * 256;
                t.j0 = (tl % tiles_c) * 256;
                t.r_begin = t.split * rows_per_split;
                t.nk = min(R, t.r_begin + rows_per_split) - t.r_begin;
                return true;
            }
        }
        return false;
    };
    auto set_offsets = [&](const Tile& t) {
        // k rows are PERMUTED on their way into LDS (gemm16_q16s_kernel): LDS row 64 g + 16 t + 4 a + b  <-  tile column
        // 64 g + 32 (t >> 1) + 8 a + 4 (t & 1) + b; for LDS row r0 + 32 j (r0 < 32) that is column wrow(r0) + 32 j
        const int tt = (r0 >> 4) & 1, aa = (r0 >> 2) & 3;
        const int wrow0 = 8 * aa + 4 * tt + (r0 & 3);
        src = (wm ? qhi : khi) + (int64_t)t.b * qk_bstride + (int64_t)t.r_begin * kstride + t.h * 64;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = wm ? t.i0 + r0 + 32 * j : t.j0 + wrow0 + 32 * j;     // clamped columns feed discarded outputs
            off[j] = (int64_t)min(row, C - 1) * ld + dchunk * 8;
        }
    };
    auto issue_mine = [&](int kt, int buf) {                 // this wave group's operand of K tile kt = alignment row r_begin + kt
        char* base = smem_b + buf * BUF + (wm ? 0 : PLANE) + 1024 * w4;
        const uint16_t* s = src + (int64_t)kt * kstride;
#pragma unroll
        for (int j = 0; j < 8; ++j) dma16(s + off[j], base + 4096 * j);
    };
    unsigned vid = blockIdx.x;
    Tile t;
    if (!find_tile(vid, t)) return;
    set_offsets(t);
    issue_mine(0, 0);
    bool stores_in_flight = false;                            // uniform
    for (;;) {
    // lane (row fr, k-group fq) of a 16-row tile reads logical chunk 4*ks + fq of its row; (row >> 1) & 7 = (fr >> 1) & 7.
    // A tile is consumed in four micro-steps (k-step, row half) of 16 MFMAs in the order (0,0) (1,0) (0,1) (1,1): the k tile has
    // been read completely after the first micro-step's fragment reads, the q tile after the third's (gemm16_q16s_kernel).
    V8 ah[2][4], bq[2][4];
    auto load_a = [&](const char* buf, int ks, int hh, V8 (&a)[4]) {
        const int chunk = ((4 * ks + fq) ^ ((fr >> 1) & 7)) * 16;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) a[tt] = *reinterpret_cast<const V8*>(buf + (wm * 128 + (4 * hh + tt) * 16 + fr) * ROWB + chunk);
    };
    auto load_b = [&](const char* buf, int ks, V8 (&b)[4]) {
        const int chunk = ((4 * ks + fq) ^ ((fr >> 1) & 7)) * 16;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) b[tt] = *reinterpret_cast<const V8*>(buf + PLANE + (wn * 64 + tt * 16 + fr) * ROWB + chunk);
    };
    f32x4q acc[8][4];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4q{0.f, 0.f, 0.f, 0.f};
    auto mma = [&](int hh, const V8 (&a)[4], const V8 (&b)[4]) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                // operands swapped: the tile comes out TRANSPOSED in the registers -- lane (fr, fq) holds row fr, columns
                // 4 fq .. 4 fq + 3 of the 16x16 tile
                acc[4 * hh + mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[nt], a[mt], acc[4 * hh + mt][nt], 0, 0, 0);
    };
    const int nk = t.nk;
    if (stores_in_flight) wait_dma_then_barrier<NSTORES>();
    else wait_dma_then_barrier<0>();
    issue_mine(nk > 1 ? 1 : 0, 1);
    load_a(smem_b, 0, 0, ah[0]);
    load_b(smem_b, 0, bq[0]);
#define RQ_PIN(NDS_)                                                                  \
    _Pragma("unroll") for (int i_ = 0; i_ < NDS_; ++i_) {                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                            \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                            \
    }                                                                                 \
    __builtin_amdgcn_sched_group_barrier(0x008, 16 - NDS_, 0);                        \
    __builtin_amdgcn_sched_barrier(0)
    for (int kt = 0; kt + 1 < nk; ++kt) {
        const char* cur = smem_b + (kt & 1) * BUF;
        const char* nxt = smem_b + ((kt & 1) ^ 1) * BUF;
        const int k2 = kt + 2 < nk ? kt + 2 : nk - 1;         // clamped: the last reload is never read (keeps the counts below fixed)
        load_a(cur, 1, 0, ah[1]);                             // (ks 0, h 0) computes; (ks 1, h 0) arriving: the last reads of the k tile
        load_b(cur, 1, bq[1]);
        mma(0, ah[0], bq[0]);
        RQ_PIN(8);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // every wave has read the k tile of `cur`
        if (wm == 0) issue_mine(k2, kt & 1);                  // k of tile kt+2 -> the k plane of `cur`
        __builtin_amdgcn_sched_barrier(0);
        load_a(cur, 0, 1, ah[0]);
        mma(0, ah[1], bq[1]);
        RQ_PIN(4);
        load_a(cur, 1, 1, ah[1]);                             // the last reads of the q tile
        mma(1, ah[0], bq[0]);
        RQ_PIN(4);
        // every wave has read the q tile of `cur`, and tile kt+1 has landed: the q group waits for all of its requests, the k
        // group leaves its newest 8 (k of tile kt+2, requested half a tile ago) in flight
        if (wm == 0) wait_dma_then_barrier<8>();
        else wait_dma_then_barrier<0>();
        if (wm == 1) issue_mine(k2, kt & 1);                  // q of tile kt+2 -> the q plane of `cur`
        __builtin_amdgcn_sched_barrier(0);
        load_a(nxt, 0, 0, ah[0]);
        load_b(nxt, 0, bq[0]);
        mma(1, ah[1], bq[1]);
        RQ_PIN(8);
    }
    {
        const char* cur = smem_b + ((nk - 1) & 1) * BUF;
        load_a(cur, 1, 0, ah[1]);
        load_b(cur, 1, bq[1]);
        mma(0, ah[0], bq[0]);
        RQ_PIN(8);
        load_a(cur, 0, 1, ah[0]);
        mma(0, ah[1], bq[1]);
        RQ_PIN(4);
        load_a(cur, 1, 1, ah[1]);
        mma(1, ah[0], bq[0]);
        RQ_PIN(4);
        mma(1, ah[1], bq[1]);
    }
#undef RQ_PIN
    wait_dma_then_barrier<0>();                               // every wave is done with LDS; the clamped reload has landed
    // the epilogue's own data first (the scale of a ragged batch: every MSA's logits by ITS depth), then the next tile's first
    // stage, then the stores
    float sc = scale;
    if (true_rows) sc = scale / sqrtf((float)max(true_rows[t.b], 1));
    asm volatile("" : "+v"(sc));
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const Tile done = t;
    unsigned nvid = vid + gridDim.x;
    Tile nt;
    const bool more = find_tile(nvid, nt);
    if (more) {
        set_offsets(nt);
        issue_mine(0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        float* out = partial + (int64_t)done.b * part_bstride + ((int64_t)done.split * H + done.h) * C * C;
        const int gm0 = done.i0 + wm * 128, gnb = done.j0 + wn * 64;
        // lane (fr, fq): row 16 mt + fr, columns 8 fq .. + 7 (tiles 0, 1) and 32 + 8 fq .. + 7 (tiles 2, 3) of the wave's 128 x 64
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const int i = gm0 + mt * 16 + fr;
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                const int j = gnb + 32 * hlf + 8 * fq;
                if (i < C && j < C) {                         // C % 8 == 0: a group of 8 columns is inside or outside as a whole
                    float* o = out + (int64_t)i * C + j;
                    epi_store(reinterpret_cast<f32x4*>(o), f32x4{acc[mt][2 * hlf][0] * sc, acc[mt][2 * hlf][1] * sc, acc[mt][2 * hlf][2] * sc, acc[mt][2 * hlf][3] * sc});
                    epi_store(reinterpret_cast<f32x4*>(o + 4), f32x4{acc[mt][2 * hlf + 1][0] * sc, acc[mt][2 * hlf + 1][1] * sc, acc[mt][2 * hlf + 1][2] * sc, acc[mt][2 * hlf + 1][3] * sc});
                }
            }
        }
    }
    if (!more) break;
    vid = nvid;
    t = nt;
    // the hoisted request pays only if exactly NSTORES vector memory instructions follow it: a ragged tile drops stores and drains
    stores_in_flight = done.i0 + 256 <= C && done.j0 + 256 <= C && !true_rows;
    if (!stores_in_flight) __builtin_amdgcn_s_waitcnt(0x0f70);           // vmcnt(0), keep expcnt / lgkmcnt
    }   // persistent tile loop
}

// ---------------------------------------------------------------------------------------------- K6' (large C)
// 256x256 tile version of row_apply16 for C >= 256: M = 256 alignment columns i, N = 4 alignment rows x 64 head dims,
// K tile = 32 keys.  Per flop it moves half the operand bytes of the 128x128 kernel above -- the 16-bit kernels lose
// matrix-pipe time in proportion to the bytes they pull into the CU, so the tile size is the lever.
// 8 waves as 2 (M) x 4 (N): wave tile 128 x 64 = one alignment row; the loop is gemm16_swp_kernel's (two fragment sets,
// one LDS read per MFMA, barrier in the middle of a tile), with the B fragments coming from "t" tiles by transposed read.
// Throw-away what-if builds of row_apply16x_kernel (wrong results, timing only; tools/whatif_row_apply16.sh): 1 no MFMAs, 2 no LDS-DMA
// inside the K loop, 4 one K tile per block, 8 no epilogue stores.  0 in the shipped library (every use folds away).
#ifndef R16X_WHATIF
#define R16X_WHATIF 0
#endif
constexpr int R16X_THREADS = 512;
// KT = keys per K tile: 32 (64-B P rows; the hi/lo modes: four planes per stage) or 64 (128-B P rows = whole cache lines per
// DMA row, half the barriers; plain bf16 only: two 64 KB stages)
template <int SPLIT, int KT>
struct R16XCfg {
    static constexpr int NPL = SPLIT == 3 ? 2 : 1;
    static constexpr int AROWB = KT * 2;
    static constexpr int PA = 256 * AROWB;                 // A plane tile: [256 rows i][KT keys]
    static constexpr int PB = 4 * KT * T16_ROWB;           // B plane tile: [4 rows r][KT keys][64 d]
    static constexpr int BUF = NPL * (PA + PB);
    static constexpr int EPI = 8 * 64 * 68 * 4;
    static constexpr int LDS = 2 * BUF > EPI ? 2 * BUF : EPI;
    static constexpr int RPI = 1024 / AROWB;               // A tile rows per wave DMA instruction
    static constexpr int IPA = 256 / RPI / 8;              // DMA instructions per wave per A plane tile
    static constexpr int IPB = 4 * KT / 8 / 8;             // ... per B plane tile (8 rows of 128 B each)
    static constexpr int KS = KT / 16;
};
template <int SPLIT, int FMT>
struct R16XFrag {
    typename Half16<FMT>::V8 a[SPLIT == 3 ? 2 : 1][4];
    TrPieces bp[SPLIT == 3 ? 2 : 1][2];                   // transposed-read pieces (asm requests, tile16.h), usable after tr16_wait2
};

template <int SPLIT, int FMT, int OUT, int KT>
__global__ __launch_bounds__(R16X_THREADS, 1) void row_apply16x_kernel(
    const uint16_t* __restrict__ phi, const uint16_t* __restrict__ plo, int64_t ldp, const uint16_t* __restrict__ vhi,
    const uint16_t* __restrict__ vlo, int64_t ld, float* __restrict__ ctx, int64_t ldc, int R, int C, int H,
    uint16_t* __restrict__ ctx_hi, uint16_t* __restrict__ ctx_lo, float out_scale, int64_t p_bstride, int64_t v_bstride,
    int64_t ctx_bstride) {
    phi += blockIdx.y * p_bstride;                           // batched launch: MSA blockIdx.y
    vhi += blockIdx.y * v_bstride;
    if (plo) plo += blockIdx.y * p_bstride;
    if (vlo) vlo += blockIdx.y * v_bstride;
    if (ctx) ctx += blockIdx.y * ctx_bstride;
    if (ctx_hi) ctx_hi += blockIdx.y * ctx_bstride;
    if (ctx_lo) ctx_lo += blockIdx.y * ctx_bstride;
    using Cfg = R16XCfg<SPLIT, KT>;
    constexpr int NPL = Cfg::NPL, AROWB = Cfg::AROWB, PA = Cfg::PA, PB = Cfg::PB, KS = Cfg::KS;
    constexpr int NMF = 8 * (SPLIT == 3 ? 3 : 1);          // MFMAs per k step per wave
    constexpr int NDS = 4 * NPL;                           // ds_read_b128 per k step per wave that the pinned interleave places (the 4 transposed reads per plane are asm)
    static_assert(KT == 32 || (KT == 64 && SPLIT == 1), "64-key tiles: two 64 KB stages fit for one plane per operand only");
    typedef typename Half16<FMT>::V8 V8;
    extern __shared__ __attribute__((aligned(16))) char smem_b[];

    const unsigned tiles_i = (C + 255) / 256, tiles_n = (R + 3) / 4;
    unsigned panel, ti;
    if (!xcd_panel_map(blockIdx.x, (unsigned)H * tiles_n, tiles_i, panel, ti)) return;
    const int h = panel / tiles_n, rr0 = (panel % tiles_n) * 4;
    const int i0 = ti * 256;

    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wv >> 2, wn = wv & 3, li = lane & 31, lh = lane >> 5;

    // DMA maps.  A (64- or 128-B rows): wave w moves row groups w + 8 j (RPI rows each), lane -> (row RPI g + lane / CPR,
    // physical chunk lane % CPR) fetching the logical chunk the read-side swizzle expects there.  B (128-B rows, rows =
    // r_local * KT + key): row groups w + 8 j (8 rows each), lane -> (row 8g + lane/8, physical chunk lane%8) fetching
    // (lane%8) ^ swz_t(row).
    constexpr int CPR = AROWB / 16;
    const int ca = KT == 32 ? ((lane & 3) ^ ((lane >> 4) & 3)) : ((lane & 7) ^ ((4 * (wv & 1) + (lane >> 4)) & 7));
    const int ct = dma_chunk_t(lane);
    int64_t poff[Cfg::IPA], voff[Cfg::IPB];
    int vkey[Cfg::IPB];
#pragma unroll
    for (int j = 0; j < Cfg::IPA; ++j) {
        const int rowa = Cfg::RPI * (wv + 8 * j) + lane / CPR;
        poff[j] = ((int64_t)h * C + min(i0 + rowa, C - 1)) * ldp + ca * 8;
    }
#pragma unroll
    for (int j = 0; j < Cfg::IPB; ++j) {
        const int rowb = 8 * (wv + 8 * j) + (lane >> 3);
        const int r = min(rr0 + rowb / KT, R - 1);                     // rows past R are discarded at the store
        voff[j] = (int64_t)r * C * ld + h * 64 + ct * 8;
        vkey[j] = rowb % KT;
    }
    auto issue = [&](int kt, int buf) {
        char* base = smem_b + buf * Cfg::BUF;
#pragma unroll
        for (int j = 0; j < Cfg::IPA; ++j) {
            const int la = (Cfg::RPI * (wv + 8 * j)) * AROWB;
            dma16(phi + poff[j] + kt * KT, base + la);
            if (SPLIT == 3) dma16(plo + poff[j] + kt * KT, base + PA + la);
        }
#pragma unroll
        for (int j = 0; j < Cfg::IPB; ++j) {
            const int lb = (8 * (wv + 8 * j)) * T16_ROWB;
            // keys past C: P is zero-padded there, V is clamped (finite) -> contributes exactly 0
            const int64_t vo = voff[j] + (int64_t)min(kt * KT + vkey[j], C - 1) * ld;
            dma16(vhi + vo, base + NPL * PA + lb);
            if (SPLIT == 3) dma16(vlo + vo, base + NPL * PA + PB + lb);
        }
    };
    const int tq = (lane & 15) >> 2, tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);    // transposed-read geometry
    auto frag_load = [&](const char* buf, int kk, R16XFrag<SPLIT, FMT>& f) {
        const int chunk = (KT == 32 ? ((2 * kk + lh) ^ ((li >> 2) & 3)) : ((2 * kk + lh) ^ ((li >> 1) & 7))) * 16;
        const int ka = 16 * kk + 8 * lh + tq;
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                f.a[p][t] = *reinterpret_cast<const V8*>(buf + p * PA + (wm * 128 + t * 32 + li) * AROWB + chunk);
            const char* bt = buf + NPL * PA + p * PB + wn * KT * T16_ROWB;    // this wave's alignment row
#pragma unroll
            for (int t = 0; t < 2; ++t) f.bp[p][t] = tr16_issue(bt, ka, ka + 4, t * 32 + tcol);
        }
    };
    // YOUNGER: transposed reads requested after this set's (the next set's, 4 per plane) -- LDS answers in order
    auto frag_mma = [&](R16XFrag<SPLIT, FMT>& f, f32x16 (&acc)[4][2], auto younger_tag) {
        constexpr int YOUNGER = decltype(younger_tag)::value;
        V8 b[NPL][2];
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
            tr16_wait2<YOUNGER>(f.bp[p][0], f.bp[p][1]);
#pragma unroll
            for (int t = 0; t < 2; ++t) b[p][t] = tr16_frag<FMT>(f.bp[p][t]);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                if (R16X_WHATIF & 1) {
                    asm volatile("" :: "v"(f.a[0][mt][0]), "v"(b[0][nt][0]));
                    continue;
                }
                if (SPLIT == 3) {
                    acc[mt][nt] = Half16<FMT>::mfma(f.a[NPL - 1][mt], b[0][nt], acc[mt][nt]);
                    acc[mt][nt] = Half16<FMT>::mfma(f.a[0][mt], b[NPL - 1][nt], acc[mt][nt]);
                }
                acc[mt][nt] = Half16<FMT>::mfma(f.a[0][mt], b[0][nt], acc[mt][nt]);
            }
    };
    typedef std::integral_constant<int, 4 * NPL> next_set_t;
    typedef std::integral_constant<int, 0> none_t;
    auto interleave = [&]() {
#pragma unroll
        for (int i = 0; i < (NDS < NMF ? NDS : NMF); ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if (NMF > NDS) __builtin_amdgcn_sched_group_barrier(0x008, NMF - NDS, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[mt][nt][t] = 0.f;

    const int nk = (R16X_WHATIF & 4) ? 1 : (C + KT - 1) / KT;
    issue(0, 0);
    wait_dma_then_barrier<0>();
    issue(nk > 1 ? 1 : 0, 1);
    R16XFrag<SPLIT, FMT> f[2];                                 // step kk lives in f[kk & 1]; KS is even
    frag_load(smem_b, 0, f[0]);
    for (int kt = 0; kt + 1 < nk; ++kt) {
        const char* cur = smem_b + (kt & 1) * Cfg::BUF;
        const char* nxt = smem_b + ((kt & 1) ^ 1) * Cfg::BUF;
#pragma unroll
        for (int kk = 0; kk + 1 < KS; ++kk) {
            frag_load(cur, kk + 1, f[(kk + 1) & 1]);
            frag_mma(f[kk & 1], acc, next_set_t());
            interleave();
        }
        // every wave is done reading `cur` once its last fragments have arrived; tile kt+1 (issued one tile ago) must have landed
        wait_dma_then_barrier<0>();
        const int k2 = kt + 2 < nk ? kt + 2 : nk - 1;              // clamped: the last reload is never read
        if (wm == 0 && !(R16X_WHATIF & 2)) issue(k2, kt & 1);                              // dephased issue: the upper wave group issues one step later
        __builtin_amdgcn_sched_barrier(0);
        frag_load(nxt, 0, f[0]);
        frag_mma(f[(KS - 1) & 1], acc, next_set_t());
        interleave();
        if (wm == 1 && !(R16X_WHATIF & 2)) issue(k2, kt & 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    {
        const char* cur = smem_b + ((nk - 1) & 1) * Cfg::BUF;
#pragma unroll
        for (int kk = 0; kk + 1 < KS; ++kk) {
            frag_load(cur, kk + 1, f[(kk + 1) & 1]);
            frag_mma(f[kk & 1], acc, next_set_t());
            interleave();
        }
        frag_mma(f[(KS - 1) & 1], acc, none_t());
    }
    wait_dma_then_barrier<0>();   // LDS is free for the epilogue staging

    const int r = rr0 + wn;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        f32x16 (&a2)[2][2] = reinterpret_cast<f32x16(&)[2][2]>(acc[2 * p]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int t = 0; t < 16; ++t) a2[mt][nt][t] *= out_scale;
        const int ibase = i0 + wm * 128 + p * 64;
        auto rowoff = [&](int row) -> int64_t {
            const int i = ibase + row;
            if ((R16X_WHATIF & 8) && i >= 0) return (int64_t)-1;
            return (r < R && i < C) ? ((int64_t)r * C + i) * ldc + h * 64 : (int64_t)-1;
        };
        slab_store_64x64<OUT>(a2, reinterpret_cast<float*>(smem_b) + wv * (64 * 68), li, lh, lane, rowoff, ctx, ctx_hi, ctx_lo);
    }
}

template <typename K>
static int set_lds16(K kern, int bytes, const char* name) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return fail(RNAMSM_ERR_HIP, "%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e));
    return RNAMSM_OK;
}

}  // namespace rnamsm

using namespace rnamsm;

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// The 256x256 kernel (and the 256-slot row split that goes with it) is used for C >= 256 in the hi/lo modes: measured at
// cfg3 in one process, split 3 0.40 -> 0.34 ms, but plain bf16 0.166 -> 0.185 ms (its 128x128 kernel already runs
// 64-deep tiles of whole cache lines with two blocks per CU), so split 1 stays on 128x128.  "attn16" = 2 forces 128x128.
// Round 3: the 256x256 kernel with 64-deep K tiles (whole cache lines per DMA row) for plain bf16 (a knob until round 6):
// 0.165-0.170 ms against 0.167-0.170 ms on the 128x128 kernel (cfg3, one process) -- no gain, stays off.  The same tile
// depth in row_apply16x (64 keys per tile) is what plain bf16 runs: 0.215 -> 0.192 ms.
// Round 4: plain bf16 at C >= 384 (at C = 256 the 128x128 kernel's 2 blocks per CU win: 0.047 vs 0.054 ms at 128 x 256) with C % 8 == 0 takes row_logits16q_kernel (256x256 tiles on the 16x16x32 MFMA, staged by
// operand): the same 256-slot row split as the other 256x256 kernel.
static inline bool row_logits16_q16(int C, bool split3) {
    return !split3 && C >= 384 && C % 8 == 0 && tuning().attn16 != 2;
}
static inline bool row_logits16_big(int C, bool split3) {
    return (split3 && C >= 256 && tuning().attn16 != 2) || row_logits16_q16(C, split3);
}
// row split of the 16-bit logits kernels; the hi/lo modes cap a slab's rows ("row16_max_rows", see row_split.h and DESIGN 3.2)
static inline RowSplit row_split16(int R, int C, int H, bool big, bool split3) {
    const int cap = split3 ? tuning().row16_max_rows : 0;
    if (row_logits16_q16(C, split3)) return choose_row_split(R, C, H, 256, 256, 0, 0.01);
    return big ? choose_row_split(R, C, H, 256, 256, cap) : choose_row_split(R, C, H, 128, 512, cap);
}

extern "C" int rnamsm_row_logits16_nsplit(int R, int C, int H, int split) {
    if (R <= 0 || C <= 0 || H <= 0) return 0;
    return row_split16(R, C, H, row_logits16_big(C, split == 3), split == 3).nsplit;
}

extern "C" size_t rnamsm_row_logits16_workspace_bytes(int R, int C, int H) {
    if (R <= 0 || C <= 0 || H <= 0) return 0;
    // the larger of the two tilings, so a workspace stays valid when the "attn16" knob flips
    int n = 1;
    for (int big = 0; big < 2; ++big)
        for (int s3 = 0; s3 < 2; ++s3) {
            const int v = row_split16(R, C, H, big != 0, s3 != 0).nsplit;
            n = v > n ? v : n;
        }
    return (size_t)n * H * C * C * sizeof(float);
}

static int row_logits16_launch(const uint16_t* q_hi, const uint16_t* q_lo, const uint16_t* k_hi, const uint16_t* k_lo, int64_t ld,
                               float* partial, int R, int C, int H, int head_dim, float scale, int fmt, void* stream, int batch,
                               int64_t qk_bstride, int64_t part_bstride, const int* true_rows) {
    RNAMSM_CHECK_ARG(q_hi && k_hi && partial, "row_logits16: null pointer");
    RNAMSM_CHECK_ARG((q_lo == nullptr) == (k_lo == nullptr), "row_logits16: q_lo and k_lo must both be given (x3) or both be null");
    RNAMSM_CHECK_ARG(head_dim == 64, "row_logits16: head_dim must be 64 (got %d)", head_dim);
    RNAMSM_CHECK_ARG(R > 0 && R <= 1024 && C > 0 && H > 0 && batch >= 1 && batch <= 65535, "row_logits16: bad shape R=%d C=%d H=%d batch=%d", R, C, H, batch);
    RNAMSM_CHECK_ARG(fmt == 0 || (fmt == 1 && q_lo), "row_logits16: fmt must be 0 (bf16) or 1 (fp16, hi/lo only)");
    RNAMSM_CHECK_ARG(ld >= (int64_t)H * 64 && ld % 8 == 0 && qk_bstride % 8 == 0 && al16(q_hi) && al16(k_hi) && al16(q_lo) && al16(k_lo),
                     "row_logits16: planes must be 16-byte aligned with ld %% 8 == 0");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool big = row_logits16_big(C, q_lo != nullptr);
    const RowSplit sp = row_split16(R, C, H, big, q_lo != nullptr);
    if (row_logits16_q16(C, q_lo != nullptr)) {
        const unsigned tc = (C + 255) / 256;
        const unsigned panels = (unsigned)batch * H * sp.nsplit;
        const unsigned total = xcd_panel_grid(panels, tc * tc);
        const unsigned pgrid = total < 256u ? total : 256u;               // persistent: one block per CU
        static DeviceOnce cfgq;
        if (cfgq.pending()) {
            int rc = set_lds16(row_logits16q_kernel, R16Q_LDS, "row_logits16q");
            if (rc) return rc;
            cfgq.mark();
        }
        KernelTimer timer(TC_ROW_LOGITS, 2.0 * batch * H * C * C * R * 64,
                          batch * (2.0 * 2.0 * R * C * H * 64 + 4.0 * (double)sp.nsplit * H * C * C), s, PEAK_F16_MFMA_TFLOPS, 1.0);
        hipLaunchKernelGGL(row_logits16q_kernel, dim3(pgrid), dim3(512), R16Q_LDS, s, q_hi, k_hi, ld, partial, R, C, H, sp.nsplit,
                           sp.rows_per_split, scale, qk_bstride, part_bstride, true_rows, panels, total);
        RNAMSM_CHECK_LAUNCH("row_logits16q");
        return RNAMSM_OK;
    }
    const unsigned tiles_c = big ? (C + 255) / 256 : (C + 127) / 128;
    const unsigned grid = xcd_panel_grid((unsigned)(H * sp.nsplit), tiles_c * tiles_c);
    KernelTimer timer(TC_ROW_LOGITS, 2.0 * batch * H * C * C * R * 64,
                      batch * ((q_lo ? 4.0 : 2.0) * 2.0 * R * C * H * 64 + 4.0 * (double)sp.nsplit * H * C * C), s, PEAK_F16_MFMA_TFLOPS,
                      q_lo ? 3.0 : 1.0);
#define RL_BIG(SP_, FMT_)                                                                                           \
    do {                                                                                                            \
        constexpr int BKX_ = SP_ == 1 ? 64 : 32;                                                                    \
        static DeviceOnce cfgx_;                                                                                    \
        if (cfgx_.pending()) {                                                                                      \
            int rc = set_lds16(row_logits16x_kernel<SP_, FMT_, BKX_>, R16LCfg<SP_, BKX_>::LDS, "row_logits16x");    \
            if (rc) return rc;                                                                                      \
            cfgx_.mark();                                                                                           \
        }                                                                                                           \
        hipLaunchKernelGGL((row_logits16x_kernel<SP_, FMT_, BKX_>), dim3(grid, batch), dim3(512), (R16LCfg<SP_, BKX_>::LDS), s, q_hi, q_lo,  \
                           k_hi, k_lo, ld, partial, R, C, H, sp.nsplit, sp.rows_per_split, scale, qk_bstride, part_bstride, true_rows); \
    } while (0)
#define RL_GO(SP_, FMT_)                                                                                            \
    do {                                                                                                            \
        static DeviceOnce cfg_;                                                                                     \
        if (cfg_.pending()) {                                                                                       \
            int rc = set_lds16(row_logits16_kernel<SP_, FMT_>, R16Cfg<SP_>::LDS, "row_logits16");                  \
            if (rc) return rc;                                                                                      \
            cfg_.mark();                                                                                            \
        }                                                                                                           \
        hipLaunchKernelGGL((row_logits16_kernel<SP_, FMT_>), dim3(grid, batch), dim3(R16_THREADS), R16Cfg<SP_>::LDS, s, q_hi, \
                           q_lo, k_hi, k_lo, ld, partial, R, C, H, sp.nsplit, sp.rows_per_split, scale, qk_bstride, part_bstride, true_rows); \
    } while (0)
    RNAMSM_NO_BF16X3(q_lo && fmt == 0, "row_logits16");
    // (plain bf16 on 256x256 tiles is row_logits16q_kernel, launched above; the 64-deep 32x32x16 variant measured level and went in round 6)
    if (!q_lo) RL_GO(1, 0);
    else if (big) RL_BIG(3, 1);
    else RL_GO(3, 1);
#undef RL_BIG
#undef RL_GO
    RNAMSM_CHECK_LAUNCH("row_logits16");
    return RNAMSM_OK;
}

extern "C" int rnamsm_row_logits16(const uint16_t* q_hi, const uint16_t* q_lo, const uint16_t* k_hi, const uint16_t* k_lo,
                                   int64_t ld, float* partial, int R, int C, int H, int head_dim, float scale, int fmt,
                                   void* stream) {
    return row_logits16_launch(q_hi, q_lo, k_hi, k_lo, ld, partial, R, C, H, head_dim, scale, fmt, stream, 1, 0, 0, nullptr);
}
namespace rnamsm {
int row_logits16_batched(const uint16_t* q_hi, const uint16_t* q_lo, const uint16_t* k_hi, const uint16_t* k_lo, int64_t ld,
                         float* partial, int R, int C, int H, float scale, int fmt, int batch, int64_t qk_bstride,
                         int64_t part_bstride, const int* true_rows, void* stream) {
    return row_logits16_launch(q_hi, q_lo, k_hi, k_lo, ld, partial, R, C, H, 64, scale, fmt, stream, batch, qk_bstride, part_bstride,
                               true_rows);
}
}  // namespace rnamsm

template <int SP, int FMT, int OUT, int KT>
static int launch_apply16x(unsigned grid, int batch, hipStream_t s, const uint16_t* p_hi, const uint16_t* p_lo, int64_t ldp,
                           const uint16_t* v_hi, const uint16_t* v_lo, int64_t ld, float* ctx, int64_t ldc, int R, int C, int H,
                           uint16_t* ctx_hi, uint16_t* ctx_lo, float out_scale, int64_t p_bstride, int64_t v_bstride,
                           int64_t ctx_bstride) {
    static DeviceOnce cfg;
    auto kern = row_apply16x_kernel<SP, FMT, OUT, KT>;
    constexpr int lds = R16XCfg<SP, KT>::LDS;
    if (cfg.pending()) {
        int rc = set_lds16(kern, lds, "row_apply16x");
        if (rc) return rc;
        cfg.mark();
    }
    hipLaunchKernelGGL(kern, dim3(grid, batch), dim3(R16X_THREADS), lds, s, p_hi, p_lo, ldp, v_hi, v_lo, ld, ctx, ldc, R, C, H,
                       ctx_hi, ctx_lo, out_scale, p_bstride, v_bstride, ctx_bstride);
    return RNAMSM_OK;
}

static int row_apply16_launch(const uint16_t* p_hi, const uint16_t* p_lo, int64_t ldp, const uint16_t* v_hi, const uint16_t* v_lo,
                              int64_t ld, float* ctx, int64_t ldc, int R, int C, int H, int head_dim, float out_scale,
                              uint16_t* ctx_hi, uint16_t* ctx_lo, int fmt, void* stream, int batch, int64_t p_bstride,
                              int64_t v_bstride, int64_t ctx_bstride) {
    RNAMSM_CHECK_ARG(p_hi && v_hi && (ctx || ctx_hi), "row_apply16: null pointer");
    RNAMSM_CHECK_ARG((p_lo == nullptr) == (v_lo == nullptr), "row_apply16: p_lo and v_lo must both be given (x3) or both be null");
    RNAMSM_CHECK_ARG(!ctx_hi || (ctx_lo == nullptr) == (p_lo == nullptr), "row_apply16: ctx_lo must match the operand split");
    RNAMSM_CHECK_ARG(head_dim == 64, "row_apply16: head_dim must be 64 (got %d)", head_dim);
    RNAMSM_CHECK_ARG(R > 0 && R <= 1024 && C > 0 && H > 0 && batch >= 1 && batch <= 65535, "row_apply16: bad shape R=%d C=%d H=%d batch=%d", R, C, H, batch);
    RNAMSM_CHECK_ARG(fmt == 0 || (fmt == 1 && p_lo), "row_apply16: fmt must be 0 (bf16) or 1 (fp16, hi/lo only)");
    RNAMSM_CHECK_ARG(ldp >= C && ldp % 64 == 0, "row_apply16: P plane stride must be C rounded up to a multiple of 64");
    RNAMSM_CHECK_ARG(ld >= (int64_t)H * 64 && ld % 8 == 0 && p_bstride % 8 == 0 && v_bstride % 8 == 0 && ctx_bstride % 4 == 0 &&
                     al16(p_hi) && al16(p_lo) && al16(v_hi) && al16(v_lo),
                     "row_apply16: planes must be 16-byte aligned with ld %% 8 == 0");
    RNAMSM_CHECK_ARG(ldc >= (int64_t)H * 64 && ldc % 4 == 0 && (ctx_hi ? (reinterpret_cast<uintptr_t>(ctx_hi) & 7u) == 0 : al16(ctx)),
                     "row_apply16: output alignment");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // (Round 4: the same contraction on the operand-staged 16x16x32 loop of row_logits16q_kernel -- 256x256 tiles, V fragments by
    // asm transposed reads, register-direct epilogue through v_permlane16_swap -- was built, passed the tests and measured 2.09 ms
    // against 2.00 ms at M = L = 1024 and 0.22 against 0.18 ms at 256 x 512: with only C / 64 = 16 K tiles per output tile the
    // 128 KB store burst of the epilogue (10 k cycles with nothing resident to cover it, in-kernel stamps) is 15 % of a tile
    // whatever the loop does.  Not kept; EXPERIMENTS.md R4.3.)
    const bool big = C >= 256 && R >= 4 && tuning().attn16 != 2;      // 256x256 tiles ("attn16" = 2 forces 128x128: A/B)
    const unsigned tiles_i = big ? (C + 255) / 256 : (C + 127) / 128, tiles_n = big ? (R + 3) / 4 : (R + 1) / 2;
    const unsigned grid = xcd_panel_grid((unsigned)H * tiles_n, tiles_i);
    KernelTimer timer(TC_ROW_APPLY, 2.0 * batch * H * C * C * R * 64,
                      batch * ((p_lo ? 4.0 : 2.0) * ((double)R * C * H * 64 + (double)H * C * ldp) + (ctx_hi ? (p_lo ? 4.0 : 2.0) : 4.0) * R * C * H * 64),
                      s, PEAK_F16_MFMA_TFLOPS, p_lo ? 3.0 : 1.0);
#define RA_GO(SP_, FMT_, OUT_)                                                                                      \
    do {                                                                                                            \
        if (big) {                                                                                                  \
            int rc = launch_apply16x<SP_, FMT_, OUT_, (SP_ == 1 ? 64 : 32)>(grid, batch, s, p_hi, p_lo, ldp, v_hi, v_lo, ld, ctx, ldc, R, C, H, ctx_hi, ctx_lo, out_scale, p_bstride, v_bstride, ctx_bstride); \
            if (rc) return rc;                                                                                      \
            break;                                                                                                  \
        }                                                                                                           \
        static DeviceOnce cfg_;                                                                                   \
        if (cfg_.pending()) {                                                                                                \
            int rc = set_lds16(row_apply16_kernel<SP_, FMT_, OUT_>, R16Cfg<SP_>::LDS, "row_apply16");              \
            if (rc) return rc;                                                                                      \
            cfg_.mark();                                                                                            \
        }                                                                                                           \
        hipLaunchKernelGGL((row_apply16_kernel<SP_, FMT_, OUT_>), dim3(grid, batch), dim3(R16_THREADS), R16Cfg<SP_>::LDS, s, p_hi, \
                           p_lo, ldp, v_hi, v_lo, ld, ctx, ldc, R, C, H, ctx_hi, ctx_lo, out_scale, p_bstride, v_bstride, ctx_bstride); \
    } while (0)
    RNAMSM_NO_BF16X3(p_lo && fmt == 0, "row_apply16");
    if (!p_lo) {
        if (ctx_hi) RA_GO(1, 0, 1); else RA_GO(1, 0, 0);
    } else {
        if (ctx_hi) RA_GO(3, 1, 2); else RA_GO(3, 1, 0);
    }
#undef RA_GO
    RNAMSM_CHECK_LAUNCH("row_apply16");
    return RNAMSM_OK;
}

extern "C" int rnamsm_row_apply16(const uint16_t* p_hi, const uint16_t* p_lo, int64_t ldp, const uint16_t* v_hi,
                                  const uint16_t* v_lo, int64_t ld, float* ctx, int64_t ldc, int R, int C, int H,
                                  int head_dim, float out_scale, uint16_t* ctx_hi, uint16_t* ctx_lo, int fmt, void* stream) {
    return row_apply16_launch(p_hi, p_lo, ldp, v_hi, v_lo, ld, ctx, ldc, R, C, H, head_dim, out_scale, ctx_hi, ctx_lo, fmt, stream, 1, 0, 0, 0);
}
namespace rnamsm {
int row_apply16_batched(const uint16_t* p_hi, const uint16_t* p_lo, int64_t ldp, const uint16_t* v_hi, const uint16_t* v_lo, int64_t ld,
                        int64_t ldc, int R, int C, int H, float out_scale, uint16_t* ctx_hi, uint16_t* ctx_lo, int fmt, int batch,
                        int64_t p_bstride, int64_t v_bstride, int64_t ctx_bstride, void* stream) {
    return row_apply16_launch(p_hi, p_lo, ldp, v_hi, v_lo, ld, nullptr, ldc, R, C, H, 64, out_scale, ctx_hi, ctx_lo, fmt, stream, batch,
                              p_bstride, v_bstride, ctx_bstride);
}
}  // namespace rnamsm
