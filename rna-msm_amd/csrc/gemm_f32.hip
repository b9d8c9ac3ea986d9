// K2/K3/K8: nn.Linear as an exact-fp32 MFMA GEMM with fused bias / column scale / erf-GELU / residual epilogue.
//   Cout[m,n] = act((sum_k A[m,k] W[n,k] + bias[n]) * (n < scale_cols ? scale : 1)) + residual[m,n]
// Reference call sites: modules.py:760-766 (q,k proj + q scaling), :794,:799 (v, out proj), :896-905, :923 (column),
// :424-426 (fc1 + GELU, fc2), :396 (residual add).
//
// FOLD (K1 fused, modules.py:387 + the Linear that consumes it): the GEMM reads the residual stream x itself and
// LayerNorm is applied to the ACCUMULATORS,
//   LN(x) W^T + b = rstd_m * ( sum_k x[m,k] Wg[n,k]  -  mean_m * c[n] ) + d[n],
//   Wg = W * gamma (columnwise),  c[n] = sum_k Wg[n,k],  d[n] = b[n] + sum_k W[n,k] beta[k]   (rnamsm_ln_fold_weights)
// so the normalised copy of x is never written or read (806 MB per LayerNorm at cfg3) and LayerNorm is no launch at all:
// the statistics come from whoever WROTE x.  STATS: the residual epilogue (out_proj / fc2, which produce the residual
// stream) leaves per row and 32-column slab the partial sums of what it stores: row_partials [N/32, M, 2]
// (slab-major: a wave's 64 rows of one slab are 512 contiguous bytes -- row-major 8-byte pieces cost a read-modify-write
// each and made this epilogue 50 us per launch slower);
// (sum x, sum (x - slab mean)^2); rnamsm_row_stats_from_partials (elementwise.hip, one thread per row) combines a row's
// K/32 partials into (mean, rstd) -- Chan et al.: no difference of large numbers anywhere -- and FOLD = 2, the consuming
// block, reads its 128 rows' pairs into LDS for the epilogue.  FOLD = 1 is the self-contained
// form (no partials given): the block sums x and x^2 of the rows it stages while the tiles go to LDS -- measured: those
// ~50 VALU instructions per K tile are NOT hidden under the MFMAs (+2.6 % kernel time, as much as the LayerNorm launch
// they replace), which is why the forward uses the partial sums.
//
// Roofline: MFMA-bound.  2*M*N*K flops against v_mfma_f32_32x32x2_f32's 157.3 TFLOP/s; per 128x128x32 K tile a CU
// issues 256 MFMAs (4096 cycles per SIMD) while 32 KB arrive from L2 (8 B/clk/CU).  Two blocks are resident per CU
// (73.7 KB LDS, <=128 VGPRs each) so one block's barrier / staging bubbles are covered by the other's MFMAs.
#include "mma_core.h"

namespace rnamsm {

// NT = MFMA tiles per wave along N: 2 -> block tile 128x128 (default), 1 -> 128x64.  All tiles of a GEMM take the same
// time and two blocks share a CU, so a launch costs ceil(blocks / 512) rounds; on small problems the last round is
// mostly empty (T = 8192, N = 768: 384 blocks = 75 % of one round).  Half-width tiles are ~3.5 % less efficient per flop
// (a wave's A fragments feed one B tile instead of two, half-size epilogue stores) but twice as many, which evens the
// rounds out; launch_gemm picks whichever its time model says finishes first.
template <int NT>
struct GemmCfg {
    static constexpr int BN_ = 64 * NT;
    static constexpr int TILE_W = BN_ * LDK;                         // floats in one W tile
    static constexpr int LDS_BYTES = 2 * (TILE_KC + TILE_W) * 4;     // double-buffered A and W tiles
    static constexpr int LDS_FOLD = LDS_BYTES + BM * 8;              // FOLD: + (mean, rstd) of the block's 128 rows
    static constexpr int LDE = 32 * NT + 4;                          // padded row stride of the epilogue staging tile
    static_assert(4 * 64 * LDE * 4 <= LDS_BYTES, "epilogue staging must fit the operand buffers");
};

// lanes i and i^1, i^2, 7-i: the three steps of a sum over aligned groups of 8 lanes on the DPP path (no LDS traffic)
__device__ __forceinline__ float sum8_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    return v;
}

// ZROWS: 0 = none; 1 = zero_rows is a uint8 mask, the scaled (q) columns of flagged rows are zeroed (f2: q *= 1 - padding_mask);
// 2 = zero_rows is a float per-row factor multiplying the scaled columns after `scale` (ragged batches: 0 at <pad>,
// 1/sqrt(true depth) elsewhere -- the general form of 1).
// One output tile [m0, m0 + 128) x [n0, n0 + 64 NT): K loop over kc (W rows keep their stride K) and the epilogue.
template <int ACT, bool HAS_RES, int ZROWS, int NT, int FOLD, bool STATS>
__device__ __forceinline__ void gemm_f32_tile(
    float* smem, const float* __restrict__ A, int64_t lda, const float* __restrict__ W, const float* __restrict__ bias,
    const float* residual, int64_t ldr, float* Cout, int64_t ldc,
    int M, int K, int kc, float scale, int scale_cols, const void* __restrict__ zero_rows,
    const float* __restrict__ fold_c, float ln_eps, float* row_partials, int64_t pld, int* fold_flag, int m0, int n0) {
    using Cfg = GemmCfg<NT>;
    constexpr int TILE_W = Cfg::TILE_W;
    float* As = smem;                    // [2][BM][LDK]
    float* Ws = smem + 2 * TILE_KC;      // [2][BN_][LDK]

    const WaveCoord w = wave_coord();
    const int c4 = threadIdx.x & 7, r0 = threadIdx.x >> 3;

    // FOLD = 2: (mean, rstd) of row m0 + tid (threads 0..127) from rnamsm_row_stats_from_partials, requested first and put
    // into LDS after tile 0 has been staged (the latency hides behind the first operand tile's)
    float2 pst = float2{0.f, 0.f};
    if (FOLD == 2 && threadIdx.x < BM) pst = reinterpret_cast<const float2*>(row_partials)[min(m0 + (int)threadIdx.x, M - 1)];

    // per-thread global row pointers (A rows clamped: a clamped row only feeds its own discarded output row)
    const float* ap[4];
    const float* wp[2 * NT];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + r0 + 32 * i;
        m = m < M ? m : M - 1;
        ap[i] = A + (int64_t)m * lda + c4 * 4;
    }
#pragma unroll
    for (int i = 0; i < 2 * NT; ++i) wp[i] = W + (int64_t)(n0 + r0 + 32 * i) * K + c4 * 4;

    f32x16 acc[2][NT];
    zero_acc<NT>(acc);

    f32x4 sa[4], sw[2 * NT];             // staging registers of one K tile: thread -> (row tid/8 + 32 i, 16-B chunk tid%8)
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};   // FOLD: sum x, sum x^2 of this thread's part of 4 rows
    pipelined_kloop<true, 4 + 2 * NT, 1, NT>(
        kc / BK, As, Ws, TILE_KC, TILE_W, acc, w,
        [&](int kt, auto) {
#pragma unroll
            for (int i = 0; i < 4; ++i) sa[i] = *reinterpret_cast<const f32x4*>(ap[i] + kt * BK);
#pragma unroll
            for (int i = 0; i < 2 * NT; ++i) sw[i] = *reinterpret_cast<const f32x4*>(wp[i] + kt * BK);
        },
        [&](int buf, auto) {
            float* at = As + buf * TILE_KC;
            float* wt = Ws + buf * TILE_W;
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(&at[(r0 + 32 * i) * LDK + c4 * 4]) = sa[i];
#pragma unroll
            for (int i = 0; i < 2 * NT; ++i) *reinterpret_cast<f32x4*>(&wt[(r0 + 32 * i) * LDK + c4 * 4]) = sw[i];
            if (FOLD == 1) {                                       // every K tile passes here exactly once
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s1[i] += (sa[i][0] + sa[i][1]) + (sa[i][2] + sa[i][3]);
                    s2[i] = fmaf(sa[i][0], sa[i][0], fmaf(sa[i][1], sa[i][1], fmaf(sa[i][2], sa[i][2], fmaf(sa[i][3], sa[i][3], s2[i]))));
                }
            }
        },
        [&]() {
            if (FOLD == 2 && threadIdx.x < BM) reinterpret_cast<float2*>(smem + Cfg::LDS_BYTES / 4)[threadIdx.x] = pst;
        });
    if (FOLD == 1) {
        // the 8 lanes tid%8 = 0..7 hold the eight 96-feature parts of a row: butterfly over them, biased variance as
        // E[x^2] - mean^2 (fp32; fine while |mean| is not >> the row's spread, as for a residual stream)
        float2* sst = reinterpret_cast<float2*>(smem + Cfg::LDS_BYTES / 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float a = s1[i], b = s2[i];
#pragma unroll
            for (int off = 1; off < 8; off <<= 1) {
                a += __shfl_xor(a, off, 64);
                b += __shfl_xor(b, off, 64);
            }
            const float mean = a / (float)K;
            const float var = fmaxf(b / (float)K - mean * mean, 0.f);
            if (c4 == 0) sst[r0 + 32 * i] = float2{mean, rsqrtf(var + ln_eps)};
            if (c4 == 0 && fold_flag && n0 == 0 && m0 + r0 + 32 * i < M && mean * mean > 1024.f * (var + ln_eps)) atomicOr(fold_flag, 2);
        }
    }

    // ---- epilogue.  The accumulator layout (one row x 32 columns per register and lane half) would give 32 NT
    // 4-byte-per-lane stores per wave, and store tails are issue-bound; instead each wave transposes its 64 x 32NT tile
    // through its own slice of the (now idle) LDS and moves whole row segments (256 B / 128 B): 8 NT float4 stores, and
    // as many float4 residual loads that are issued BEFORE the transpose so their latency hides behind it.
    constexpr int LDE = Cfg::LDE;
    constexpr int LPR = 8 * NT;                                   // lanes per staged row (float4 each)
    constexpr int RPP = 64 / LPR;                                 // rows per pass
    constexpr int NP = 64 / RPP;                                  // passes
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int er = lane / LPR, ec = (lane % LPR) * 4;             // lane -> (row er + RPP*i, columns ec..ec+3)
    const int gm0 = m0 + w.wm * 64, gn = n0 + w.wn * 32 * NT + ec;
    f32x4 res[NP];
    if (HAS_RES) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int row = min(gm0 + er + RPP * i, M - 1);
            res[i] = epi_load(reinterpret_cast<const f32x4*>(residual + (int64_t)row * ldr + gn));
        }
    }
    __syncthreads();                                              // every wave has finished reading operand tiles
    // FOLD: (mean, rstd) of the 32 rows this lane's accumulators belong to (acc_row: (t&3) + 8(t>>2) + 4 lh + 32 mt); the
    // two lane halves read two addresses per instruction (broadcast), and the fold runs where the bias add does, ahead
    // of the GELU and of the transpose
    float2 st[FOLD ? 32 : 1];
    if (FOLD) {
        const float2* sst = reinterpret_cast<const float2*>(smem + Cfg::LDS_BYTES / 4) + w.wm * 64 + 4 * w.lh;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < 16; ++t) st[mt * 16 + t] = sst[mt * 32 + (t & 3) + 8 * (t >> 2)];
    }
    float* stage = smem + wv * (64 * LDE);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = n0 + w.wn * 32 * NT + nt * 32 + w.li;
        const float b = bias ? bias[col] : 0.f;                   // FOLD: d[n]
        const float sc = col < scale_cols ? scale : 1.f;
        const float fc = FOLD ? fold_c[col] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < 16; t += 2) {                    // pairs: the GELU runs on the packed-fp32 VALU
                f32x2 v;
                if (FOLD) {
                    const float2 s0 = st[mt * 16 + t], s1 = st[mt * 16 + t + 1];
                    v = f32x2{fmaf(s0.y, fmaf(-s0.x, fc, acc[mt][nt][t]), b) * sc,
                              fmaf(s1.y, fmaf(-s1.x, fc, acc[mt][nt][t + 1]), b) * sc};
                } else {
                    v = f32x2{(acc[mt][nt][t] + b) * sc, (acc[mt][nt][t + 1] + b) * sc};
                }
                if (ACT == RNAMSM_ACT_GELU_ERF) v = gelu_erf2(v);
                stage[(mt * 32 + (t & 3) + 8 * (t >> 2) + 4 * w.lh) * LDE + nt * 32 + w.li] = v[0];
                stage[(mt * 32 + ((t + 1) & 3) + 8 * ((t + 1) >> 2) + 4 * w.lh) * LDE + nt * 32 + w.li] = v[1];
            }
    }
    // same-wave LDS write -> read: ordered by the hardware queue, the compiler inserts the lgkmcnt wait
    // all LDS reads first, then the stores; full tiles (the common case) carry no per-row bounds branch -- hipcc
    // otherwise sinks each read into its row's branch and serialises read -> wait -> store every time
    f32x4 ov[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int r = er + RPP * i;
        ov[i] = *reinterpret_cast<const f32x4*>(&stage[r * LDE + ec]);
        if (HAS_RES) ov[i] += res[i];
        // f2: q *= 1 - padding_mask (modules.py:767-772): padded tokens get q = 0 (the scaled columns are q)
        if (ZROWS == 1 && gn < scale_cols && static_cast<const uint8_t*>(zero_rows)[min(gm0 + r, M - 1)]) ov[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ZROWS == 2 && gn < scale_cols) ov[i] *= static_cast<const float*>(zero_rows)[min(gm0 + r, M - 1)];
    }
    if (STATS) {
        // What this lane stores of row r: 4 of the 32 columns its group of 8 lanes covers.  A second pass over the finished
        // values (an LDS write inside the read loop above would order every later read of the staging tile behind it); the
        // group sums are parked in the 4 padding columns of the row's staging slot (2 floats per 32-column slab, never
        // read as data) ...
        float ps[NP], pq[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) ps[i] = sum8_dpp((ov[i][0] + ov[i][1]) + (ov[i][2] + ov[i][3]));      // slab sum
#pragma unroll
        for (int i = 0; i < NP; ++i) {                            // sum of squares about the SLAB's mean: nothing to cancel
            const float mb = ps[i] * (1.f / 32.f);
            const float d0 = ov[i][0] - mb, d1 = ov[i][1] - mb, d2 = ov[i][2] - mb, d3 = ov[i][3] - mb;
            pq[i] = sum8_dpp(fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, d3 * d3))));
        }
        if ((lane & 7) == 0) {
#pragma unroll
            for (int i = 0; i < NP; ++i)
                *reinterpret_cast<float2*>(&stage[(er + RPP * i) * LDE + 32 * NT + 2 * (ec / 32)]) = float2{ps[i], pq[i]};
        }
        // ... and leave lane = row: one 512-byte run per slab (slab-major [N/32, M, 2]) instead of 8-byte pieces
        float2* pout = reinterpret_cast<float2*>(row_partials) + (int64_t)((n0 + w.wn * 32 * NT) / 32) * pld + gm0 + lane;
#pragma unroll
        for (int sl = 0; sl < NT; ++sl) {
            const float2 pr = *reinterpret_cast<const float2*>(&stage[lane * LDE + 32 * NT + 2 * sl]);
            if (gm0 + lane < M) pout[(int64_t)sl * pld] = pr;
        }
    }
    if (m0 + BM <= M) {
#pragma unroll
        for (int i = 0; i < NP; ++i) epi_store(reinterpret_cast<f32x4*>(Cout + (int64_t)(gm0 + er + RPP * i) * ldc + gn), ov[i]);
    } else {
#pragma unroll
        for (int i = 0; i < NP; ++i)
            if (gm0 + er + RPP * i < M) epi_store(reinterpret_cast<f32x4*>(Cout + (int64_t)(gm0 + er + RPP * i) * ldc + gn), ov[i]);
    }
}

template <int ACT, bool HAS_RES, int ZROWS, int NT, int FOLD = 0, bool STATS = false>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_f32_kernel(
    const float* __restrict__ A, int64_t lda, const float* __restrict__ W, const float* __restrict__ bias,
    const float* residual, int64_t ldr, float* Cout, int64_t ldc,
    int M, int N, int K, float scale, int scale_cols, const void* __restrict__ zero_rows, int group,
    const float* __restrict__ fold_c, float ln_eps, float* row_partials, int64_t pld, int* fold_flag) {
    constexpr int BN_ = GemmCfg<NT>::BN_;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned nb = N / BN_, mp = (M + BM - 1) / BM;
    // split-K (group >> 16 = number of K ranges; launch_gemm_splitk): the grid holds that many copies of the tile grid, copy s
    // accumulates K range s (W keeps its row stride K) and stores its tile into partial slab s of Cout ([ks][M][ldc])
    const int ksplit = group >> 16;
    unsigned bid = blockIdx.x;
    int kc = K;
    if (ksplit) {
        const unsigned per = gridDim.x / ksplit, s_ = bid / per;
        bid -= s_ * per;
        kc = K / ksplit;
        A += (int64_t)s_ * kc;
        W += (int64_t)s_ * kc;
        Cout += (int64_t)s_ * M * ldc;
    }
    unsigned mpanel, nblk;
    if (!xcd_panel_map_ragged(bid, mp, nb, (unsigned)(group & 0xffff), mpanel, nblk)) return;
    gemm_f32_tile<ACT, HAS_RES, ZROWS, NT, FOLD, STATS>(smem, A, lda, W, bias, residual, ldr, Cout, ldc, M, K, kc, scale, scale_cols, zero_rows,
                                                        fold_c, ln_eps, row_partials, pld, fold_flag, (int)(mpanel * BM), (int)(nblk * BN_));
}

// MIXED tiles (round 5): a launch costs ceil(blocks / 512) rounds of co-resident pairs, and on a mid-size problem the last round is
// mostly empty -- T = 18432, N = 2304: 2592 tiles = 5 full rounds + 32 tiles that cost a sixth (all-half-width tiles: 5184 = 10
// rounds + 64, no better).  Here the first `full_blocks` block ids (whole rounds) are 128 x 128 tiles in the usual XCD order and
// every tile position after them is cut into its two 128 x 64 halves, consecutive block ids: the last round then holds twice as
// many blocks of half the length.  A tile's shape never touches an element's K order (one accumulator, k ascending): the result
// is bit-identical to any other tiling -- unlike a split of K over the last round (stream-K), which would make an element's
// rounding depend on how many rows the launch has, i.e. on the batch an alignment travels in.
template <int ACT, bool HAS_RES, int ZROWS, int FOLD = 0, bool STATS = false>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_f32_mixed_kernel(
    const float* __restrict__ A, int64_t lda, const float* __restrict__ W, const float* __restrict__ bias,
    const float* residual, int64_t ldr, float* Cout, int64_t ldc,
    int M, int N, int K, float scale, int scale_cols, const void* __restrict__ zero_rows, int group, unsigned full_blocks,
    const float* __restrict__ fold_c, float ln_eps, float* row_partials, int64_t pld, int* fold_flag) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned nb = N / BN, mp = (M + BM - 1) / BM;
    unsigned mpanel, nblk;
    if (blockIdx.x < full_blocks) {
        if (!xcd_panel_map_ragged(blockIdx.x, mp, nb, (unsigned)group, mpanel, nblk)) return;
        gemm_f32_tile<ACT, HAS_RES, ZROWS, 2, FOLD, STATS>(smem, A, lda, W, bias, residual, ldr, Cout, ldc, M, K, K, scale, scale_cols, zero_rows,
                                                           fold_c, ln_eps, row_partials, pld, fold_flag, (int)(mpanel * BM), (int)(nblk * BN));
    } else {
        const unsigned u = blockIdx.x - full_blocks;
        if (!xcd_panel_map_ragged(full_blocks + (u >> 1), mp, nb, (unsigned)group, mpanel, nblk)) return;
        gemm_f32_tile<ACT, HAS_RES, ZROWS, 1, FOLD, STATS>(smem, A, lda, W, bias, residual, ldr, Cout, ldc, M, K, K, scale, scale_cols, zero_rows,
                                                           fold_c, ln_eps, row_partials, pld, fold_flag, (int)(mpanel * BM),
                                                           (int)(nblk * BN + (u & 1u) * 64u));
    }
}

// A GEMM of at most this many tiles (the chip's block slots) is dealt flat (tile = block id) instead of XCD-aware.  Larger values win
// the stand-alone GEMM A/B up to ~19 k tokens (T = 2064 QKV +28 %, 18944 fc2 +14 %) but LOSE 1-2 % inside the forward, where A was just
// written by the previous kernel and the XCD-aware order keeps each panel on one XCD (a knob until round 6; EXPERIMENTS R4.8).
constexpr int64_t GEMM_FLAT_TILES = 512;
static inline int gemm_group_for(int mp, int nb, int NT, int ksplit) {
    return tuning().gemm_group > 0 ? tuning().gemm_group
                                   : ((int64_t)mp * nb * (ksplit > 1 ? ksplit : 1) <= GEMM_FLAT_TILES ? 0 : (nb > 8 * (3 - NT) ? 8 : 1));
}

template <int ACT, bool HAS_RES, int ZROWS, int NT, int FOLD = 0, bool STATS = false>
static int launch_gemm_nt(const float* A, int64_t lda, const float* W, const float* bias, const float* residual,
                          int64_t ldr, float* Cout, int64_t ldc, int M, int N, int K, float scale, int scale_cols,
                          const void* zero_rows, hipStream_t stream, const float* fold_c = nullptr,
                          float ln_eps = 0.f, float* row_partials = nullptr, int64_t pld = 0, int* fold_flag = nullptr,
                          int ksplit = 0) {
    using Cfg = GemmCfg<NT>;
    static DeviceOnce configured;
    auto kern = gemm_f32_kernel<ACT, HAS_RES, ZROWS, NT, FOLD, STATS>;
    if (configured.pending()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, FOLD ? Cfg::LDS_FOLD : Cfg::LDS_BYTES);
        if (e != hipSuccess) return fail(RNAMSM_ERR_HIP, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
        configured.mark();
    }
    // Block order (speed/traffic only): 64 blocks are resident per XCD.  With more than 8 column blocks per row panel a
    // whole-panel order keeps only 64/nb panels in flight and re-streams W (7-9 MB > the 4 MB L2) for each of them;
    // groups of 8 panels x 8 column blocks halve the fabric reads (PMC, cfg3: QKV 4.9 -> 2.9 GB, fc1 8.0 -> 3.6 GB per
    // launch; same speed, the kernel is MFMA-bound).  N = 768 (6 column blocks) is already balanced and stays ungrouped.
    const int nb = N / Cfg::BN_;
    // (round 4) no padding groups: an XCD's last panels form a smaller group (xcd_panel_map_ragged); and a GEMM with no more tiles
    // than the chip has block slots (GEMM_FLAT_TILES, 512) is dealt FLAT (group 0: tile = block id), so that a lone small
    // alignment's 18-24 column tiles run on as many CUs of all XCDs instead of on one XCD's
    const int mp_ = (M + BM - 1) / BM;
    const int group = gemm_group_for(mp_, nb, NT, ksplit);
    const unsigned grid = xcd_panel_grid_ragged(mp_, nb, (unsigned)group) * (ksplit > 1 ? ksplit : 1);
    // algorithmic work: 2MNK flops; bytes = A + W + C once (+ residual read)
    KernelTimer timer(TC_GEMM, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N * (HAS_RES ? 2 : 1)), stream);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(GEMM_THREADS), FOLD ? Cfg::LDS_FOLD : Cfg::LDS_BYTES, stream, A, lda, W, bias,
                       residual, ldr, Cout, ldc, M, N, K, scale, scale_cols, zero_rows, group | (ksplit > 1 ? ksplit << 16 : 0), fold_c, ln_eps,
                       row_partials, pld, fold_flag);
    RNAMSM_CHECK_LAUNCH("gemm_f32");
    return RNAMSM_OK;
}

template <int ACT, bool HAS_RES, int ZROWS, int FOLD = 0, bool STATS = false>
static int launch_gemm_mixed(const float* A, int64_t lda, const float* W, const float* bias, const float* residual,
                             int64_t ldr, float* Cout, int64_t ldc, int M, int N, int K, float scale, int scale_cols,
                             const void* zero_rows, hipStream_t stream, unsigned full_blocks, const float* fold_c = nullptr,
                             float ln_eps = 0.f, float* row_partials = nullptr, int64_t pld = 0, int* fold_flag = nullptr) {
    using Cfg = GemmCfg<2>;
    static DeviceOnce configured;
    auto kern = gemm_f32_mixed_kernel<ACT, HAS_RES, ZROWS, FOLD, STATS>;
    if (configured.pending()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, FOLD ? Cfg::LDS_FOLD : Cfg::LDS_BYTES);
        if (e != hipSuccess) return fail(RNAMSM_ERR_HIP, "gemm (mixed tiles): hipFuncSetAttribute: %s", hipGetErrorString(e));
        configured.mark();
    }
    const int nb = N / BN, mp_ = (M + BM - 1) / BM;
    const int group = gemm_group_for(mp_, nb, 2, 1);
    const unsigned all = xcd_panel_grid_ragged(mp_, nb, (unsigned)group);       // block ids of the all-full-tile order (incl. spare ids at its end)
    const unsigned grid = full_blocks + 2u * (all - full_blocks);
    KernelTimer timer(TC_GEMM, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N * (HAS_RES ? 2 : 1)), stream);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(GEMM_THREADS), FOLD ? Cfg::LDS_FOLD : Cfg::LDS_BYTES, stream, A, lda, W, bias,
                       residual, ldr, Cout, ldc, M, N, K, scale, scale_cols, zero_rows, group, full_blocks, fold_c, ln_eps,
                       row_partials, pld, fold_flag);
    RNAMSM_CHECK_LAUNCH("gemm_f32 (mixed tiles)");
    return RNAMSM_OK;
}

// Tile width from a per-CU time model calibrated on the box (tools/gemm_ab.py gemm_tile=1,2 at T = 8192 .. 131072):
// blocks are dealt evenly, a CU holding b of them runs them two at a time; a co-resident pair costs 2 units of matrix
// pipe time, a block running alone 1.1 (it nearly saturates the pipe by itself), a half-width block 0.5175 of a full
// one (3.5 % per-flop penalty).  Half-width tiles are taken when the model gains more than 3 %: e.g. T = 8192, N = 768:
// 384 full blocks -> 2 per busiest CU = 2.0 units, 768 half blocks -> 3 per CU = 1.6 units (measured 0.101 -> 0.082 ms);
// cfg3 (exact multiples of 512 blocks) stays on full tiles.
// Round 5: MIXED (see gemm_f32_mixed_kernel): whole rounds of full tiles, then the remaining tile positions as halves; taken
// when the model gains another 2 % over the better uniform choice.  T = 18432: QKV 11.1 (full) / 10.9 (half) -> 10.6 units.
struct TilePlan {
    int kind;                 // 0 = 128 x 128 tiles, 1 = 128 x 64 tiles, 2 = mixed
    unsigned full_blocks;     // mixed: block ids that are full tiles
};
static inline TilePlan tile_plan(int M, int N) {
    const int t = tuning().gemm_tile;
    const long mp = (M + BM - 1) / BM, nb = N / BN, tiles = mp * nb;
    auto cu_time = [](long blocks) {
        const long b = (blocks + 255) / 256;
        return 2.0 * (double)(b / 2) + 1.1 * (double)(b % 2);
    };
    // block ids below full_blocks must all be real tiles: whole rounds of 512 inside the panels every XCD owns
    long full = tiles / 512 * 512;
    const long common = 8 * (mp / 8) * nb;
    while (full > common) full -= 512;
    const bool can_mix = full > 0 && full < tiles && tiles > GEMM_FLAT_TILES && tuning().gemm_group <= 0;
    if (t == 1) return {0, 0};
    if (t == 2) return {1, 0};
    if (t == 3) return can_mix ? TilePlan{2, (unsigned)full} : TilePlan{0, 0};
    const double c_full = cu_time(tiles), c_half = 0.5175 * cu_time(2 * tiles);
    TilePlan best{0, 0};
    double c_best = c_full;
    if (c_half < 0.97 * c_full) {
        best = {1, 0};
        c_best = c_half;
    }
    if (t != 4 && can_mix && (double)(full / 256) + 0.5175 * cu_time(2 * (tiles - full)) < 0.98 * c_best) best = {2, (unsigned)full};
    return best;
}

template <int ACT, bool HAS_RES, int ZROWS = 0>
static int launch_gemm(const float* A, int64_t lda, const float* W, const float* bias, const float* residual,
                       int64_t ldr, float* Cout, int64_t ldc, int M, int N, int K, float scale, int scale_cols,
                       const void* zero_rows, hipStream_t stream) {
    const TilePlan plan = ZROWS == 1 ? TilePlan{0, 0} : tile_plan(M, N);      // (the uint8-mask form: padded single forwards, full tiles)
    if (plan.kind == 2)
        return launch_gemm_mixed<ACT, HAS_RES, ZROWS>(A, lda, W, bias, residual, ldr, Cout, ldc, M, N, K, scale, scale_cols, zero_rows,
                                                      stream, plan.full_blocks);
    if (plan.kind == 1)
        return launch_gemm_nt<ACT, HAS_RES, ZROWS, 1>(A, lda, W, bias, residual, ldr, Cout, ldc, M, N, K, scale, scale_cols,
                                                      zero_rows, stream);
    return launch_gemm_nt<ACT, HAS_RES, ZROWS, 2>(A, lda, W, bias, residual, ldr, Cout, ldc, M, N, K, scale, scale_cols,
                                                  zero_rows, stream);
}

template <int ACT, int FOLD>
static int launch_gemm_fold(const float* X, int64_t ldx, const float* Wg, const float* cvec, const float* dvec,
                            float ln_eps, const float* row_partials, int64_t pld, int* fold_flag, float* Cout, int64_t ldc,
                            int M, int N, int K, float scale, int scale_cols, hipStream_t stream) {
    float* rp = const_cast<float*>(row_partials);                  // read-only in the FOLD kernels
    const TilePlan plan = tile_plan(M, N);
    if (plan.kind == 2)
        return launch_gemm_mixed<ACT, false, 0, FOLD>(X, ldx, Wg, dvec, nullptr, 0, Cout, ldc, M, N, K, scale, scale_cols, nullptr, stream,
                                                      plan.full_blocks, cvec, ln_eps, rp, pld, fold_flag);
    if (plan.kind == 1)
        return launch_gemm_nt<ACT, false, 0, 1, FOLD>(X, ldx, Wg, dvec, nullptr, 0, Cout, ldc, M, N, K, scale, scale_cols,
                                                          nullptr, stream, cvec, ln_eps, rp, pld, fold_flag);
    return launch_gemm_nt<ACT, false, 0, 2, FOLD>(X, ldx, Wg, dvec, nullptr, 0, Cout, ldc, M, N, K, scale, scale_cols,
                                                      nullptr, stream, cvec, ln_eps, rp, pld, fold_flag);
}

// residual GEMM that also leaves the row partial sums of what it stores (the producer side of the folded LayerNorm)
static int launch_gemm_res_stats(const float* A, int64_t lda, const float* W, const float* bias, const float* residual,
                                 int64_t ldr, float* Cout, int64_t ldc, int M, int N, int K, float* row_partials,
                                 int64_t pld, hipStream_t stream) {
    const TilePlan plan = tile_plan(M, N);
    if (plan.kind == 2)
        return launch_gemm_mixed<RNAMSM_ACT_NONE, true, 0, 0, true>(A, lda, W, bias, residual, ldr, Cout, ldc, M, N, K, 1.f, 0, nullptr, stream,
                                                                    plan.full_blocks, nullptr, 0.f, row_partials, pld);
    if (plan.kind == 1)
        return launch_gemm_nt<RNAMSM_ACT_NONE, true, 0, 1, 0, true>(A, lda, W, bias, residual, ldr, Cout, ldc, M, N, K, 1.f, 0,
                                                                        nullptr, stream, nullptr, 0.f, row_partials, pld);
    return launch_gemm_nt<RNAMSM_ACT_NONE, true, 0, 2, 0, true>(A, lda, W, bias, residual, ldr, Cout, ldc, M, N, K, 1.f, 0,
                                                                    nullptr, stream, nullptr, 0.f, row_partials, pld);
}

// ---- split-K for small token counts (rnamsm_forward, fc2).  Below ~1.4 k tokens the K = 3072 GEMM has far fewer tiles than
// the chip has CUs (T = 1024: 48 tiles of 96 K steps each): four copies of the tile grid each take K / 4 and leave fp32
// partial tiles, and one elementwise pass adds the four slabs IN SLAB ORDER (reruns stay bit-identical), the bias and the
// residual.  Measured per forward (tools/splitk_ab.py): 256 .. 1024 tokens 5.92 -> 5.52 ms (+7 %); with more tiles than
// CUs / 4 the copies share CUs and the gain is gone (2048 tokens -1 %, 4096 -1 %), and two ranges never paid -- so: four
// ranges while tiles * 4 <= 256, else none.  Knob "gemm_splitk": 0 = never, 1 = that rule, 2 / 4 / 8 = that many ranges
// whenever tiles * ks <= 512 (A/B).
//
// Round 4: the K = 768 GEMMs (QKV, out_proj, fc1) of a LONE small alignment.  At 520 tokens a forward is 61 GEMM launches of ~80 us
// each -- one tile's serial K loop on 90 of 256 CUs (tools/lone_small_profile.py: 5.0 of 5.5 ms) -- so the same split applies:
// knob "gemm_splitk_short" K ranges (0 = off) while tiles * ranges <= 512 and the GEMM has at most SPLITK_SHORT_MAX_TILES tiles; the
// reduction pass then also applies the column scale, the activation and the q = 0 rows of the GEMM it completes.
constexpr int64_t SPLITK_SHORT_MAX_TILES = 192;
int gemm_f32_splitk_factor(int64_t M, int N, int K, bool by_shape_only) {
    const int knob = by_shape_only ? 8 : tuning().gemm_splitk;      // by_shape_only: the most the workspace may be asked for
    if (N % BN) return 1;
    if (K < 2048) {
        const int ks = by_shape_only ? 4 : tuning().gemm_splitk_short;
        const int64_t tiles = ((M + BM - 1) / BM) * (N / BN);
        if (ks < 2 || tiles > SPLITK_SHORT_MAX_TILES) return 1;
        for (int k2 = ks; k2 >= 2; k2 >>= 1)
            if (tiles * k2 <= 512 && K % (k2 * BK) == 0) return k2;
        return 1;
    }
    if (knob == 0) return 1;
    const int64_t tiles = ((M + BM - 1) / BM) * (N / BN);
    if (knob == 1) return (tiles * 4 <= 256 && K % (4 * BK) == 0) ? 4 : 1;
    for (int ks = knob; ks >= 2; ks >>= 1)
        if (tiles * ks <= 512 && K % (ks * BK) == 0) return ks;
    return 1;
}

// the epilogue of gemm_f32_kernel (same order: + bias, column scale / zeroed rows on the scaled columns, activation, + residual)
// on the ordered sum of the partial slabs
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partials, int ks, int64_t slab,
                                                            const float* __restrict__ bias, const float* residual, int64_t ldr,
                                                            float* out, int64_t ldc, int64_t M, int N, int act, float scale,
                                                            int scale_cols, const uint8_t* __restrict__ zero_rows) {
    const int n4 = N / 4;
    const int64_t total = M * n4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / n4;
        const int n = (int)(i - m * n4) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(partials + m * N + n);
        for (int s = 1; s < ks; ++s) v += *reinterpret_cast<const f32x4*>(partials + s * slab + m * N + n);
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
        if (n < scale_cols) v *= scale;            // scale_cols % 4 == 0: the four columns lie on one side
        if (act == RNAMSM_ACT_GELU_ERF) {
            const f32x2 a = gelu_erf2(f32x2{v[0], v[1]}), b = gelu_erf2(f32x2{v[2], v[3]});
            v = f32x4{a[0], a[1], b[0], b[1]};
        }
        if (residual) v += *reinterpret_cast<const f32x4*>(residual + m * ldr + n);
        if (zero_rows && n < scale_cols && zero_rows[m]) v = f32x4{0.f, 0.f, 0.f, 0.f};      // q *= 1 - padding_mask (modules.py:767-772)
        *reinterpret_cast<f32x4*>(out + m * ldc + n) = v;
    }
}

int gemm_f32_splitk(const float* A, int64_t lda, const float* W, const float* bias, const float* residual, int64_t ldr,
                    float* Cout, int64_t ldc, int64_t M, int N, int K, int ks, float* partials, hipStream_t stream, int act,
                    float scale, int scale_cols, const uint8_t* zero_rows) {
    RNAMSM_CHECK_ARG(scale_cols % 4 == 0, "gemm_splitk: scale_cols must be a multiple of 4 (got %d)", scale_cols);
    RNAMSM_CHECK_ARG(ks >= 2 && K % (ks * BK) == 0 && N % BN == 0 && partials, "gemm_splitk: bad split %d for K=%d N=%d", ks, K, N);
    const int rc = launch_gemm_nt<RNAMSM_ACT_NONE, false, false, 2>(A, lda, W, nullptr, nullptr, 0, partials, N, (int)M, N, K, 1.f, 0,
                                                                    nullptr, stream, nullptr, 0.f, nullptr, 0, nullptr, ks);
    if (rc != RNAMSM_OK) return rc;
    const int64_t total = M * (N / 4);
    const unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, stream, partials, ks, M * (int64_t)N, bias, residual, ldr,
                       Cout, ldc, M, N, act, scale, scale_cols, zero_rows);
    RNAMSM_CHECK_LAUNCH("splitk_reduce");
    return RNAMSM_OK;
}

}  // namespace rnamsm


using namespace rnamsm;

extern "C" int rnamsm_gemm_residual_stats(const float* A, int64_t lda, const float* W, const float* bias,
                                          const float* residual, int64_t ldr, float* Cout, int64_t ldc, int64_t M, int N,
                                          int K, float* row_partials, int64_t partials_ld, int dtype, void* stream) {
    if (dtype != RNAMSM_F32) return fail(RNAMSM_ERR_UNSUPPORTED, "gemm_residual_stats: only RNAMSM_F32 is implemented");
    RNAMSM_CHECK_ARG(A && W && Cout && residual && row_partials, "gemm_residual_stats: null pointer");
    RNAMSM_CHECK_ARG(M > 0 && M <= INT32_MAX && N > 0 && K > 0, "gemm_residual_stats: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
    RNAMSM_CHECK_ARG(N % BN == 0 && K % BK == 0, "gemm_residual_stats: need N %% 128 == 0 and K %% 32 == 0 (N=%d K=%d)", N, K);
    RNAMSM_CHECK_ARG(lda >= K && lda % 4 == 0 && ldc >= N && ldc % 4 == 0 && ldr >= N && ldr % 4 == 0,
                     "gemm_residual_stats: bad leading dimension");
    RNAMSM_CHECK_ARG(aligned16(A) && aligned16(W) && aligned16(Cout) && aligned16(residual) && aligned16(row_partials),
                     "gemm_residual_stats: 16-byte alignment");
    RNAMSM_CHECK_ARG(partials_ld >= M, "gemm_residual_stats: partials_ld (rows per slab of row_partials) must be >= M");
    return launch_gemm_res_stats(A, lda, W, bias, residual, ldr, Cout, ldc, (int)M, N, K, row_partials, partials_ld,
                                 static_cast<hipStream_t>(stream));
}

extern "C" int rnamsm_gemm_lnfold(const float* X, int64_t ldx, const float* Wg, const float* cvec, const float* dvec,
                                  float ln_eps, const float* row_stats, int* cond_flag, float* Cout, int64_t ldc, int64_t M,
                                  int N, int K, int act, float scale, int scale_cols, int dtype, void* stream) {
    if (dtype != RNAMSM_F32) return fail(RNAMSM_ERR_UNSUPPORTED, "gemm_lnfold: only RNAMSM_F32 is implemented");
    RNAMSM_CHECK_ARG(X && Wg && cvec && dvec && Cout, "gemm_lnfold: null pointer");
    RNAMSM_CHECK_ARG(M > 0 && M <= INT32_MAX && N > 0 && K > 0, "gemm_lnfold: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
    RNAMSM_CHECK_ARG(N % BN == 0 && K % BK == 0, "gemm_lnfold: need N %% 128 == 0 and K %% 32 == 0 (N=%d K=%d)", N, K);
    RNAMSM_CHECK_ARG(ldx >= K && ldx % 4 == 0 && ldc >= N && ldc % 4 == 0, "gemm_lnfold: bad leading dimension ldx=%lld ldc=%lld",
                     (long long)ldx, (long long)ldc);
    RNAMSM_CHECK_ARG(aligned16(X) && aligned16(Wg) && aligned16(cvec) && aligned16(dvec) && aligned16(Cout),
                     "gemm_lnfold: 16-byte alignment");
    RNAMSM_CHECK_ARG(ln_eps >= 0.f, "gemm_lnfold: negative eps");
    RNAMSM_CHECK_ARG(!row_stats || (reinterpret_cast<uintptr_t>(row_stats) & 7u) == 0, "gemm_lnfold: row_stats must be 8-byte aligned");
    RNAMSM_CHECK_ARG(scale_cols >= 0 && scale_cols % 4 == 0, "gemm_lnfold: scale_cols must be a multiple of 4");
    RNAMSM_CHECK_ARG(act == RNAMSM_ACT_NONE || act == RNAMSM_ACT_GELU_ERF, "gemm_lnfold: unknown activation %d", act);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define RNAMSM_FOLD_DISPATCH(ACT_, FOLD_) \
    launch_gemm_fold<ACT_, FOLD_>(X, ldx, Wg, cvec, dvec, ln_eps, row_stats, 0, cond_flag, Cout, ldc, (int)M, N, K, scale, scale_cols, s)
    if (act == RNAMSM_ACT_GELU_ERF) return row_stats ? RNAMSM_FOLD_DISPATCH(RNAMSM_ACT_GELU_ERF, 2) : RNAMSM_FOLD_DISPATCH(RNAMSM_ACT_GELU_ERF, 1);
    return row_stats ? RNAMSM_FOLD_DISPATCH(RNAMSM_ACT_NONE, 2) : RNAMSM_FOLD_DISPATCH(RNAMSM_ACT_NONE, 1);
#undef RNAMSM_FOLD_DISPATCH
}

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
        return launch_gemm<RNAMSM_ACT_NONE, false, 1>(A, lda, W, bias, residual, ldr, Cout, ldc, m, N, K, scale,
                                                      scale_cols, zero_rows, s);
    }
    if (act == RNAMSM_ACT_GELU_ERF) return residual ? RNAMSM_GEMM_DISPATCH(RNAMSM_ACT_GELU_ERF, true) : RNAMSM_GEMM_DISPATCH(RNAMSM_ACT_GELU_ERF, false);
    return residual ? RNAMSM_GEMM_DISPATCH(RNAMSM_ACT_NONE, true) : RNAMSM_GEMM_DISPATCH(RNAMSM_ACT_NONE, false);
#undef RNAMSM_GEMM_DISPATCH
}

// K2 with a per-row factor on the scaled columns: Cout[m, n] = ((A W^T + bias)[m, n] * scale) * row_factor[m] for n < scale_cols,
// (A W^T + bias)[m, n] elsewhere -- the QKV projection of a RAGGED batch (rnamsm_forward_batch with true_rows), where a token's q
// is scaled by dh^-1/2 / sqrt(the TRUE depth of its own MSA) and zeroed at <pad> (align_scaling and q *= 1 - padding_mask,
// modules.py:713-715, 767-772): the general form of zero_rows.
extern "C" int rnamsm_gemm_row_scaled(const float* A, int64_t lda, const float* W, const float* bias, float* Cout, int64_t ldc,
                                      int64_t M, int N, int K, float scale, int scale_cols, const float* row_factor, int dtype,
                                      void* stream) {
    if (dtype != RNAMSM_F32) return fail(RNAMSM_ERR_UNSUPPORTED, "gemm_row_scaled: only RNAMSM_F32 is implemented");
    RNAMSM_CHECK_ARG(A && W && Cout && row_factor, "gemm_row_scaled: null pointer");
    RNAMSM_CHECK_ARG(M > 0 && M <= INT32_MAX && N > 0 && K > 0, "gemm_row_scaled: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
    RNAMSM_CHECK_ARG(N % BN == 0 && K % BK == 0, "gemm_row_scaled: need N %% 128 == 0 and K %% 32 == 0 (N=%d K=%d)", N, K);
    RNAMSM_CHECK_ARG(lda >= K && lda % 4 == 0 && ldc >= N && ldc % 4 == 0, "gemm_row_scaled: bad leading dimension");
    RNAMSM_CHECK_ARG(scale_cols >= 0 && scale_cols <= N && scale_cols % 4 == 0, "gemm_row_scaled: scale_cols must be a multiple of 4 in [0, N]");
    RNAMSM_CHECK_ARG(aligned16(A) && aligned16(W) && aligned16(Cout), "gemm_row_scaled: A, W and Cout must be 16-byte aligned");
    return launch_gemm<RNAMSM_ACT_NONE, false, 2>(A, lda, W, bias, nullptr, 0, Cout, ldc, (int)M, N, K, scale, scale_cols,
                                                  row_factor, static_cast<hipStream_t>(stream));
}
