// K4 / K5 / K6: tied row attention (RowSelfAttention, modules.py:688-821).
//
//  K4 row_logits : S[h,i,j] = sum_{r,d} q[r,i,h,d] k[r,j,h,d]     per head a [C x R*64] x [R*64 x C] GEMM
//  K5 softmax    : P[h,i,:] = softmax_j(sum_s S_s[h,i,:])         wave-per-row reduction; P is the atp slab
//  K6 row_apply  : ctx[r,i,h,:] = sum_j P[h,i,j] v[r,j,h,:]       per head a [C x C] x [C x R*64] GEMM
//
// Both contractions run on the 128x128x32 fp32 MFMA tile of mma_core.h.  Their operands live in the fused QKV
// activation [T, 3D] written by K2, addressed in place: element (r,i,h,d) = ptr[(r*C+i)*ld + h*64 + d], so one K tile
// of K4 is "row r, half of head_dim" (32 contiguous floats per alignment column) and one B tile of K6 is 32 key
// columns x (2 rows r x 64 d) read as 256-B runs.  Roofline: MFMA-bound (K = R*64 resp. C deep); K4's only HBM
// output is nsplit*H*C*C partials.
//
// Tied attention sums over all R rows before the softmax; the reference does that in max_tokens-sized row chunks
// (_batched_forward, modules.py:717-750).  Here the rows are cut into nsplit contiguous ranges (to fill 256 CUs: one
// head has only ceil(C/128)^2 output tiles) and K5 adds the partials in range order -> fixed summation order.
#include "half16.h"
#include "mma_core.h"
#include "row_split.h"

namespace rnamsm {

constexpr int HEAD_DIM = 64;
constexpr int ROW_NARROW_MAX_C = 64;     // alignments this narrow take the *_narrow kernels (K4, K6)
constexpr int ROWLOGITS_LDS_BYTES = 2 * (TILE_KC + TILE_KC) * 4;
constexpr int ROWAPPLY_LDS_BYTES = 2 * (TILE_KC + TILE_NC) * 4;
constexpr int ROWAPPLY_VT_LDS_BYTES = 2 * (TILE_KC + TILE_KC) * 4;

// ---------------------------------------------------------------------------------------------- K4
// grid.x = xcd-mapped (panel = (head, split), inner = tiles_i * tiles_j): the 16 tiles of one (head, split) share
// the same q/k row range, so they are placed on one XCD and re-read it from that L2.
// chain_tiles: after every chain_tiles K tiles (= chain_tiles / 2 alignment rows) the MFMA accumulators are added into a
// second register set and restarted, so no fp32 accumulation chain is longer than chain_tiles * 32 terms however many
// rows the block's slab covers (row_split.h: the accuracy of the exact path hangs on it); the chain sums are added in
// order.  A slab of 32 rows in four chains replaces four slabs of 8 rows: a quarter of the slab writes here and of the
// slab reads in K5, and four times the K loop per block prologue / epilogue.
__global__ __launch_bounds__(GEMM_THREADS, 2) void row_logits_kernel(
    const float* __restrict__ q, const float* __restrict__ k, int64_t ld, float* __restrict__ partial,
    int R, int C, int H, int nsplit, int rows_per_split, int chain_tiles, int64_t qk_bstride, int64_t part_bstride,
    const PackedMsa* __restrict__ pk, int skip_narrow) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem;
    float* Ks = smem + 2 * TILE_KC;
    if (pk) {
        // token-packed batch (rnamsm_forward_packed): alignment blockIdx.y has its own shape, slabs and token offset
        const PackedMsa& m = pk[blockIdx.y];
        if (skip_narrow && m.C <= ROW_NARROW_MAX_C) return;      // that alignment is row_logits_narrow_kernel's
        R = m.R; C = m.C; nsplit = m.nsplit; rows_per_split = m.rows_per_split;
        q += m.tok0 * ld;
        k += m.tok0 * ld;
        partial += m.part_off;
    } else {
        // batched launch (gridDim.y MSAs of the same shape, rnamsm_forward_batch): MSA blockIdx.y's operands and slabs
        q += blockIdx.y * qk_bstride;
        k += blockIdx.y * qk_bstride;
        partial += blockIdx.y * part_bstride;
    }

    const unsigned tiles_c = (C + BM - 1) / BM;
    unsigned panel, tile;
    if (!xcd_panel_map(blockIdx.x, (unsigned)(H * nsplit), tiles_c * tiles_c, panel, tile)) return;
    const int h = panel / nsplit, split = panel % nsplit;
    const int i0 = (tile / tiles_c) * BM, j0 = (tile % tiles_c) * BN;
    const int r_begin = split * rows_per_split;
    const int r_end = min(R, r_begin + rows_per_split);

    const WaveCoord w = wave_coord();
    const int c4 = threadIdx.x & 7, r0 = threadIdx.x >> 3;

    // column (alignment position) handled by each staging slot, clamped; a clamped column only feeds discarded outputs
    int64_t qoff[4], koff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int i = min(i0 + r0 + 32 * s, C - 1), j = min(j0 + r0 + 32 * s, C - 1);
        qoff[s] = (int64_t)i * ld + h * HEAD_DIM + c4 * 4;
        koff[s] = (int64_t)j * ld + h * HEAD_DIM + c4 * 4;
    }

    f32x16 acc[2][2], total[2][2];
    zero_acc(acc);
    zero_acc(total);

    const int nk = (r_end - r_begin) * (HEAD_DIM / BK);      // K tile kt = (row r_begin + kt/2, d half kt&1)
    auto tile_base = [&](int kt) -> int64_t {
        return (int64_t)(r_begin + (kt >> 1)) * C * ld + (kt & 1) * BK;
    };

    StageKC sq, sk;
    pipelined_kloop<true, 8, 1>(
        nk, Qs, Ks, TILE_KC, TILE_KC, acc, w,
        [&](int kt, auto) {
            const int64_t b = tile_base(kt);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                sq.v[s] = *reinterpret_cast<const f32x4*>(q + b + qoff[s]);
                sk.v[s] = *reinterpret_cast<const f32x4*>(k + b + koff[s]);
            }
        },
        [&](int buf, auto) {
            stage_store_kc(Qs + buf * TILE_KC, sq);
            stage_store_kc(Ks + buf * TILE_KC, sk);
        },
        NoHook{},
        [&](int done) {
            if (done % chain_tiles == 0 || done == nk) {      // chain_tiles is even, `done` walks the even numbers
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        total[mt][nt] += acc[mt][nt];
#pragma unroll
                        for (int t = 0; t < 16; ++t) acc[mt][nt][t] = 0.f;
                    }
            }
        });

    float* out = partial + ((int64_t)split * H + h) * C * C;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int j = j0 + acc_col(w, nt);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int i = i0 + acc_row(w, mt, t);
                if (i < C && j < C) out[(int64_t)i * C + j] = total[mt][nt][t];
            }
    }
}

// K4 for alignments of at most 64 columns (short RNAs: the shipped 2DRB_1 example is 36 wide).  The 128 x 128 tile spends
// (C/128)^2 of its matrix-pipe time on real logits -- 8 % at C = 36, 118 us per layer at 512 x 36 -- so this kernel drops the
// tile and the LDS altogether: block = (head, slab) as before, its four waves own the four 32 x 32 quadrants of the
// [64, 64] logit block, and a lane loads ITS query / key token's 16-byte pieces straight into the MFMA operand registers
// (lane (li, lh) -> token quadrant*32 + li, floats 8 kk + 4 lh ..+3: the operand layout of mma_core.h without the detour).
// Two half-rows are in flight per wave: the d < 32 pieces of row r+1 are requested when row r's have been consumed.
// Same slabs, same K order (row, kk, s), same chain folding as row_logits_kernel: the two kernels agree BIT FOR BIT, so an
// alignment's maps do not depend on which one a batch routed it to.
__global__ __launch_bounds__(256) void row_logits_narrow_kernel(
    const float* __restrict__ q, const float* __restrict__ k, int64_t ld, float* __restrict__ partial,
    int R, int C, int H, int nsplit, int rows_per_split, int chain_rows, int64_t qk_bstride, int64_t part_bstride,
    const PackedMsa* __restrict__ pk) {
    if (pk) {
        const PackedMsa& m = pk[blockIdx.y];
        if (m.C > ROW_NARROW_MAX_C) return;                      // row_logits_kernel's
        R = m.R; C = m.C; nsplit = m.nsplit; rows_per_split = m.rows_per_split;
        q += m.tok0 * ld;
        k += m.tok0 * ld;
        partial += m.part_off;
    } else {
        q += blockIdx.y * qk_bstride;
        k += blockIdx.y * qk_bstride;
        partial += blockIdx.y * part_bstride;
    }
    if ((int)blockIdx.x >= H * nsplit) return;
    const int h = blockIdx.x / nsplit, split = blockIdx.x % nsplit;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const int i0 = (wave >> 1) * 32, j0 = (wave & 1) * 32;
    if (i0 >= C || j0 >= C) return;                              // C <= 32: one quadrant holds it all (no barrier in this kernel)
    const int r_begin = split * rows_per_split;
    const int nrows = min(R, r_begin + rows_per_split) - r_begin;
    // wave-uniform base (scalar registers) + one 32-bit lane offset per operand: the 16 loads of a row share two address registers
    const float* qp = q + (int64_t)r_begin * C * ld + h * HEAD_DIM;
    const float* kp = k + (int64_t)r_begin * C * ld + h * HEAD_DIM;
    const unsigned qlane = (unsigned)(min(i0 + li, C - 1) * (int)ld + 4 * lh), klane = (unsigned)(min(j0 + li, C - 1) * (int)ld + 4 * lh);
    const int64_t row_step = (int64_t)C * ld;

    f32x16 acc, total;
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = total[t] = 0.f;
    f32x4 a0[4], b0[4], a1[4], b1[4];                            // d < 32 and d >= 32 of the row in flight
    auto load_half = [&](f32x4 (&a)[4], f32x4 (&b)[4], int r, int half) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            a[kk] = *reinterpret_cast<const f32x4*>(qp + r * row_step + half * 32 + kk * 8 + qlane);
            b[kk] = *reinterpret_cast<const f32x4*>(kp + r * row_step + half * 32 + kk * 8 + klane);
        }
    };
    auto mma_half = [&](const f32x4 (&a)[4], const f32x4 (&b)[4]) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = mfma32(a[kk][s], b[kk][s], acc);
    };
    load_half(a0, b0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);                           // same request order as in the loop: the wait counts at its head stay exact
    load_half(a1, b1, 0, 1);
    __builtin_amdgcn_sched_barrier(0);
    for (int r = 0; r < nrows; ++r) {
        const int rn = r + 1 < nrows ? r + 1 : r;               // the last row's prefetch re-reads itself (no branch in the loop)
        mma_half(a0, b0);
        __builtin_amdgcn_sched_barrier(0);                       // (hipcc otherwise sinks both requests to the end of the row:
        load_half(a0, b0, rn, 0);                                //  no flight time left)
        __builtin_amdgcn_sched_barrier(0);
        mma_half(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        load_half(a1, b1, rn, 1);
        __builtin_amdgcn_sched_barrier(0);
        if ((r + 1) % chain_rows == 0 || r + 1 == nrows) {       // <= 512-term accumulation chains, added in order (row_split.h)
            total += acc;
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = 0.f;
        }
    }
    float* out = partial + ((int64_t)split * H + h) * C * C;
    const int j = j0 + li;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int i = i0 + (t & 3) + 8 * (t >> 2) + 4 * lh;
        if (i < C && j < C) out[(int64_t)i * C + j] = total[t];
    }
}

// ---------------------------------------------------------------------------------------------- K5
// One wave per (h, i) row: sum the nsplit partial slabs in slab order, then softmax over j in fp32
// (attn_weights.softmax(-1), modules.py:818/739).  C <= 1024 + 1 fits 17 values per lane.
// PL: 0 = fp32 probabilities only; 1 / 2 = additionally bf16 / fp16 hi(+lo) planes [rows][ldp] (ldp = C rounded up
// to 64, the tail zero-filled) holding P * plane_scale: the k-contiguous A operand of the 16-bit row_apply.
// NE = values per lane (C <= 64 NE): 17 covers the maximum width; narrower alignments take the 1-, 4- or 8-value instance, whose slab
// loop is 1 / 4 loads per slab instead of 17 predicated ones (23 -> 8 us per launch at 512 x 36, 40 slabs).  Same arithmetic: the
// values a wider instance would carry beyond C are -inf / 0 and add nothing.
constexpr int SOFTMAX_MAX_PER_LANE = 17;
template <int PL, int NE = SOFTMAX_MAX_PER_LANE>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ partial, int nsplit,
                                                           float* __restrict__ probs, int64_t rows, int C,
                                                           const uint8_t* __restrict__ key_mask,
                                                           uint16_t* __restrict__ p_hi, uint16_t* __restrict__ p_lo,
                                                           int64_t ldp, float plane_scale, int64_t mask_slab_stride,
                                                           int64_t part_bstride, int64_t probs_bstride, int64_t mask_bstride,
                                                           int64_t plane_bstride, const PackedMsa* __restrict__ pk, int layer, int H,
                                                           float logit_scale) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    float packed_scale = logit_scale;
    if (pk) {       // token-packed batch: alignment blockIdx.y's own width, slabs, maps and depth factor (fp32 maps only)
        const PackedMsa& m = pk[blockIdx.y];
        C = m.C; rows = (int64_t)H * C; nsplit = m.nsplit;
        partial += m.part_off;
        probs += m.probs_off + (int64_t)layer * H * C * C;
        packed_scale = m.logit_scale;
    } else {
        partial += blockIdx.y * part_bstride;        // batched launch: MSA blockIdx.y
        probs += blockIdx.y * probs_bstride;
    }
    if (row >= rows) return;
    if (PL != 0) {
        p_hi += blockIdx.y * plane_bstride;
        if (p_lo) p_lo += blockIdx.y * plane_bstride;
    }
    if (key_mask) key_mask += blockIdx.y * mask_bstride;
    const int lane = threadIdx.x & 63;
    const int64_t slab = rows * C;
    float v[NE];
    float mx = -INFINITY;
    // slab-major: every slab iteration issues up to 17 independent loads per lane (the fp32 path sums 32+ slabs: one
    // dependent load per iteration left the kernel latency-bound at 3 TB/s); each element still adds its slabs in slab order
    const float* prow = partial + row * C;
    if (mask_slab_stride == 0) {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int j = lane + 64 * e;
            v[e] = j < C ? prow[j] : -INFINITY;
        }
        // (four slabs' loads in flight per lane: PMC showed the waves 94 % of their time in s_waitcnt, one HBM round
        // trip per slab; the adds stay in slab order)
        constexpr int SLAB_UNROLL = NE == 1 ? 8 : 4;
#pragma unroll SLAB_UNROLL
        for (int sp = 1; sp < nsplit; ++sp) {
            const float* ps = prow + (int64_t)sp * slab;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int j = lane + 64 * e;
                if (j < C) v[e] += ps[j];
            }
        }
        // exact path without padding -- alone, in a same-shape batch or token-packed alike (ONE arithmetic per alignment, whatever
        // the batch: VERDICT r04 item 4): q carries dh^-1/2 only and the alignment's 1/sqrt(R) meets the summed logits here.
        // (1.0 elsewhere: the masked / chunked / 16-bit callers keep their factor where the reference has it)
#pragma unroll
        for (int e = 0; e < NE; ++e) v[e] *= packed_scale;
        // f2: masked_fill(padding_mask[:, 0], -10000) on the key axis (modules.py:781-785)
        if (key_mask) {
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int j = lane + 64 * e;
                if (j < C && key_mask[j]) v[e] = -10000.f;
            }
        }
    } else {
        // f2 on the reference's chunked path (_batched_forward, modules.py:717-750): every slab is one row chunk,
        // filled with -10000 where the chunk's OWN first row is padded (:727-737), then `attns += attn_weights`
        // in chunk order -- masks of slab sp start at key_mask + sp * mask_slab_stride
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int j = lane + 64 * e;
            v[e] = j < C ? (key_mask[j] ? -10000.f : prow[j]) : -INFINITY;
        }
        for (int sp = 1; sp < nsplit; ++sp) {
            const float* ps = prow + (int64_t)sp * slab;
            const uint8_t* ms = key_mask + (int64_t)sp * mask_slab_stride;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int j = lane + 64 * e;
                if (j < C) v[e] += ms[j] ? -10000.f : ps[j];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) mx = fmaxf(mx, v[e]);
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int j = lane + 64 * e;
        v[e] = j < C ? expf(v[e] - mx) : 0.f;
        sum += v[e];
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int j = lane + 64 * e;
        if (j < C) probs[row * C + j] = v[e] * inv;
        if (PL != 0 && j < ldp) {
            typedef typename Half16<(PL > 0 ? PL - 1 : 0)>::T Hh;
            const float p = pinned(j < C ? v[e] * inv * plane_scale : 0.f);
            const Hh hi = (Hh)p;
            reinterpret_cast<Hh*>(p_hi)[row * ldp + j] = hi;
            if (p_lo) reinterpret_cast<Hh*>(p_lo)[row * ldp + j] = (Hh)(p - (float)hi);
        }
    }
}

// ---------------------------------------------------------------------------------------------- K6
// out[i, (r,d)] = sum_j P[h][i][j] * V[j, (r,d)].  A = P_h rows (k = j contiguous), B = v read as [j][n] with
// n = (r_local, d): 2 alignment rows x 64 head dims per 128-wide N tile.
// grid.x = xcd-mapped (panel = (head, n tile), inner = tiles_i): the i tiles of one V panel share an L2.
// ALIGNED: C % 4 == 0 and probs 16-B aligned -> P rows can be read as float4.
// OUT: 0 = fp32 context; 1 / 2 = bf16 / fp16 hi(+lo) planes, the pre-split A operand of the following out_proj GEMM.
// VT: the V tile is transposed WHILE it is staged ([n][k], k contiguous, the layout of the GEMM's W tile), so its
// fragments are ds_read_b128 and the loop is the B_KC one that row_logits and the GEMM run at 90 % matrix-pipe busy;
// without it the tile stays [k][n] and every fragment costs four ds_read_b32 (70 % busy).  The transpose is free of
// bank conflicts because the lanes of a wave are mapped key-fastest: lane -> (key = lane/2, 16-B half = lane%2), two
// lanes fetch one 32-B sector of a V row and write 4 + 4 scalars to LDS rows n..n+3 at column key (each half-wave
// touches 32 distinct banks).
template <bool ALIGNED, int OUT, bool VT>
__global__ __launch_bounds__(GEMM_THREADS, 2) void row_apply_kernel(
    const float* __restrict__ probs, const float* __restrict__ v, int64_t ld, float* __restrict__ ctx, int64_t ldc,
    int R, int C, int H, uint16_t* __restrict__ ctx_hi, uint16_t* __restrict__ ctx_lo, int64_t probs_bstride, int64_t v_bstride,
    int64_t ctx_bstride, const PackedMsa* __restrict__ pk, int layer, int skip_narrow) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ps = smem;                    // [2][BM][LDK]
    if (pk) {                            // token-packed batch (fp32 context only): alignment blockIdx.y's own shape and offsets
        const PackedMsa& m = pk[blockIdx.y];
        if (skip_narrow && m.C <= ROW_NARROW_MAX_C) return;      // row_apply_narrow_kernel's
        R = m.R; C = m.C;
        probs += m.probs_off + (int64_t)layer * H * C * C;
        v += m.tok0 * ld;
        if (OUT == 0) ctx += m.tok0 * ldc;
    } else {
        probs += blockIdx.y * probs_bstride;         // batched launch: MSA blockIdx.y (fp32 context only)
        v += blockIdx.y * v_bstride;
        if (OUT == 0) ctx += blockIdx.y * ctx_bstride;
    }
    float* Vs = smem + 2 * TILE_KC;      // [2][BK][LDN], or [2][BN][LDK] when VT
    constexpr int TILE_V = VT ? TILE_KC : TILE_NC;

    const unsigned tiles_i = (C + BM - 1) / BM, tiles_n = (R + 1) / 2;
    unsigned panel, ti;
    if (!xcd_panel_map(blockIdx.x, (unsigned)H * tiles_n, tiles_i, panel, ti)) return;
    const int h = panel / tiles_n, rr0 = (panel % tiles_n) * 2;      // alignment rows rr0, rr0+1
    const int i0 = ti * BM;

    const WaveCoord w = wave_coord();
    // A staging: thread -> (row = tid/8 + 32*s, chunk tid%8) of the [128][32] P tile
    const int c4 = threadIdx.x & 7, r0 = threadIdx.x >> 3;
    // B staging: thread -> (k = tid/32 + 8*s, n4 = tid%32) of the [32][128] V tile; n4 -> (r_local = n4/16, d4 = n4%16)
    // VT staging: thread -> key k = lane/2, n chunk q = 2*(4*wave + s) + lane%2 (n = 4q..4q+3; waves 0,1 hold
    // alignment row rr0, waves 2,3 row rr0+1)
    const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;
    const int bk0 = VT ? (lane_ >> 1) : (threadIdx.x >> 5), n4 = VT ? (8 * wave_ + (lane_ & 1)) : (threadIdx.x & 31);
    const int br = min(rr0 + (n4 >> 4), R - 1);                      // clamped: second row of an odd R is discarded
    const float* pbase = probs + (int64_t)h * C * C;
    const float* vbase = v + (int64_t)br * C * ld + h * HEAD_DIM + (n4 & 15) * 4;      // VT: + 8*s floats per slot s

    f32x16 acc[2][2];
    zero_acc(acc);
    const int nk = (C + BK - 1) / BK;

    f32x4 sp[4], sv[4];
    int staged_j0 = 0;                   // first key of the tile sitting in sp / sv
    // Loads are branch-free (clamped addresses) and NOTHING consumes the loaded values here: the zero-fill of keys
    // j >= C (they must contribute exactly 0 in BOTH operands, clamped data could be NaN) happens a tile later, when the
    // registers are written to LDS -- a select at load time would make the wave wait for its own loads immediately
    // (that was 23 % of the kernel's wave time in s_waitcnt).
    auto load_tiles = [&](int kt, auto) {
        const int j0 = kt * BK;
        staged_j0 = j0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int i = min(i0 + r0 + 32 * s, C - 1);
            const int j = j0 + c4 * 4;
            const float* prow = pbase + (int64_t)i * C;
            if (ALIGNED) {
                sp[s] = *reinterpret_cast<const f32x4*>(prow + min(j, C - 4));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) sp[s][e] = prow[min(j + e, C - 1)];
            }
            const int jj = VT ? j0 + bk0 : j0 + bk0 + 8 * s;
            sv[s] = *reinterpret_cast<const f32x4*>(vbase + (int64_t)min(jj, C - 1) * ld + (VT ? 8 * s : 0));
        }
    };
    auto store_tiles = [&](int buf, auto) {
        float* pt = Ps + buf * TILE_KC;
        float* vt = Vs + buf * TILE_V;
        const int j = staged_j0 + c4 * 4;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 t = sp[s];
            if (ALIGNED) {
                if (j >= C) t = f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = (j + e < C) ? t[e] : 0.f;
            }
            *reinterpret_cast<f32x4*>(&pt[(r0 + 32 * s) * LDK + c4 * 4]) = t;
            const int jj = VT ? staged_j0 + bk0 : staged_j0 + bk0 + 8 * s;
            f32x4 u = sv[s];
            if (jj >= C) u = f32x4{0.f, 0.f, 0.f, 0.f};
            if (VT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) vt[((n4 + 2 * s) * 4 + e) * LDK + bk0] = u[e];
            } else {
                *reinterpret_cast<f32x4*>(&vt[(bk0 + 8 * s) * LDN + n4 * 4]) = u;
            }
        }
    };

    if (VT)
        pipelined_kloop<true, ALIGNED ? 8 : 20, 1, 2, 20>(nk, Ps, Vs, TILE_KC, TILE_V, acc, w, load_tiles, store_tiles);
    else
        pipelined_kloop<false, ALIGNED ? 8 : 20, 1>(nk, Ps, Vs, TILE_KC, TILE_V, acc, w, load_tiles, store_tiles);

    // The wave's 64 columns are exactly one alignment row (r = rr0 + wn) x 64 head dims, its 64 rows are alignment
    // columns i: the slab leaves as 256-B context segments ctx[(r*C + i), h*64 .. h*64+63] through LDS.
    __syncthreads();                                              // every wave has finished reading operand tiles
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = rr0 + w.wn;
    const int ibase = i0 + w.wm * 64;
    auto rowoff = [&](int row) -> int64_t {
        const int i = ibase + row;
        return (r < R && i < C) ? ((int64_t)r * C + i) * ldc + h * HEAD_DIM : (int64_t)-1;
    };
    slab_store_64x64<OUT>(acc, smem + wv * (64 * 68), w.li, w.lh, lane, rowoff, ctx, ctx_hi, ctx_lo);
}

// K6 for alignments of at most 64 columns.  The map P_h (<= 64 x 64) is the same for every alignment row, so a block keeps it:
// staged once through LDS (zero-padded to [64][64]: keys >= C contribute exactly 0), each lane then holds ITS fragments of it
// in registers for the whole block (<= 64 VGPRs) and the K loop has no LDS traffic at all.  A wave walks alignment rows
// r = chunk start + wave, +4, ...; V of one row goes global -> registers in the MFMA B layout (lane (li, lh) reads
// v[r, 8 kk + 4 lh + s, h, 32 nt + li]: two full 128-byte lines per instruction), the d >= 32 half requested while the
// d < 32 half is being multiplied, and the context leaves from the accumulator registers the same way (128-byte runs).
// Addressing: buffer loads / stores over the alignment's own V and context ranges -- ONE 32-bit lane offset for all 32 loads of
// a half row (the key's row offset is wave-uniform and rides in the scalar offset), and the hardware range check instead of
// branches: a key row past the alignment's last token reads 0, a context row >= C gets an out-of-range lane offset and is dropped.
// K order (kk, s over the keys) as in row_apply_kernel, whose trailing all-zero key groups add +0: the results agree bit for bit.
// NKK = key groups of 8 (C in (8 NKK - 8, 8 NKK]): compile-time, so the row loop is branch-free.
template <int NKK>
__device__ __forceinline__ void row_apply_narrow_rows(const float* Ps, const float* v, int64_t ld, float* ctx, int64_t ldc, int R, int C,
                                                      int h, int r, int r_end, int li, int lh) {
    constexpr int LDP = 68;
    constexpr bool TWO = NKK > 4;                                // query rows 32..63 exist
    constexpr int MT = TWO ? 2 : 1;
    f32x4 pa[MT][NKK];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) pa[mt][kk] = *reinterpret_cast<const f32x4*>(&Ps[(mt * 32 + li) * LDP + kk * 8 + 4 * lh]);
    // (R * C <= 65536 tokens here: the byte ranges fit the descriptor's 32 bits)
    const unsigned OOB = 0xffffffffu;
    const auto vsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(v + h * HEAD_DIM), 0, (int)((int64_t)R * C * ld * 4 - h * HEAD_DIM * 4), 0x00020000);
    const auto cdst = __builtin_amdgcn_make_buffer_rsrc(ctx + h * HEAD_DIM, 0, (int)((int64_t)R * C * ldc * 4 - h * HEAD_DIM * 4), 0x00020000);
    const unsigned vlane = (unsigned)(4 * lh * (int)ld + li) * 4u;        // bytes: key + 4 lh, head dim li
    const unsigned clane = (unsigned)(4 * lh * (int)ldc + li) * 4u;       // bytes: query row + 4 lh
    const unsigned clane_lo = lh ? OOB : clane;                           // a pair of rows (ib, ib + 4) of which only ib < C
    const int ld4 = (int)ld * 4, ldc4 = (int)ldc * 4;
    // the last key group: a lane whose key is >= C loads through an out-of-range offset, i.e. reads 0 (no select after the load,
    // which would make the wave wait for its newest request first)
    unsigned vlast[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) vlast[s] = (NKK - 1) * 8 + 4 * lh + s >= C ? OOB : vlane;
    f32x4 vb0[NKK], vb1[NKK];                                    // d < 32, d >= 32 of the row in flight: [kk][s]
    auto load_half = [&](f32x4 (&vb)[NKK], int row, int nt) {
        const int base = row * C * ld4 + nt * 128;               // wave-uniform byte offset of key 0
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                vb[kk][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(vsrc, kk == NKK - 1 ? vlast[s] : vlane,
                                                                                           base + (kk * 8 + s) * ld4, 0));
    };
    auto run_half = [&](const f32x4 (&vb)[NKK], int row, int nt) {
        f32x16 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[mt][t] = 0.f;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma32(pa[mt][kk][s], vb[kk][s], acc[mt]);
        const int base = row * C * ldc4 + nt * 128;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int ib = mt * 32 + (t & 3) + 8 * (t >> 2);  // rows ib (lh = 0) and ib + 4 (lh = 1) in one instruction
                const float val = acc[mt][t];                     // (a bit_cast of the vector ELEMENT expression stores element 0: hipcc 7.2)
                if (ib + 4 < 8 * (NKK - 1)) {                     // both below C whatever C is
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), cdst, clane, base + ib * ldc4, 0);
                } else if (ib < 8 * NKK) {                        // the range check drops what lies at or past row C
                    const unsigned off = ib + 4 < C ? clane : (ib < C ? clane_lo : OOB);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), cdst, off, base + ib * ldc4, 0);
                }
            }
    };
    load_half(vb0, r, 0);
    __builtin_amdgcn_sched_barrier(0);                           // request order = the loop's: the wait counts at its head stay exact
    load_half(vb1, r, 1);
    __builtin_amdgcn_sched_barrier(0);
    for (; r < r_end; r += 4) {
        const bool more = r + 4 < r_end;                         // wave-uniform
        run_half(vb0, r, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) load_half(vb0, r + 4, 0);
        __builtin_amdgcn_sched_barrier(0);
        run_half(vb1, r, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (more) load_half(vb1, r + 4, 1);
        __builtin_amdgcn_sched_barrier(0);
    }
}

__global__ __launch_bounds__(256) void row_apply_narrow_kernel(
    const float* __restrict__ probs, const float* __restrict__ v, int64_t ld, float* __restrict__ ctx, int64_t ldc,
    int R, int C, int H, int rows_per_block, int64_t probs_bstride, int64_t v_bstride, int64_t ctx_bstride,
    const PackedMsa* __restrict__ pk, int layer) {
    constexpr int LDP = 68;
    __shared__ __attribute__((aligned(16))) float Ps[64 * LDP];
    if (pk) {
        const PackedMsa& m = pk[blockIdx.y];
        if (m.C > ROW_NARROW_MAX_C) return;                      // row_apply_kernel's
        R = m.R; C = m.C;
        probs += m.probs_off + (int64_t)layer * H * C * C;
        v += m.tok0 * ld;
        ctx += m.tok0 * ldc;
    } else {
        probs += blockIdx.y * probs_bstride;
        v += blockIdx.y * v_bstride;
        ctx += blockIdx.y * ctx_bstride;
    }
    const int chunks = (R + rows_per_block - 1) / rows_per_block;
    if ((int)blockIdx.x >= H * chunks) return;
    const int h = blockIdx.x % H, chunk = blockIdx.x / H;        // the heads of one row chunk read the same token rows: neighbours
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;

    const float* pbase = probs + (int64_t)h * C * C;
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int i = e >> 6, j = e & 63;
        Ps[i * LDP + j] = (i < C && j < C) ? pbase[i * C + j] : 0.f;
    }
    __syncthreads();
    const int r_end = min(R, (chunk + 1) * rows_per_block);
    const int r = chunk * rows_per_block + wave;
    if (r >= r_end) return;
    switch ((C + 7) >> 3) {
        case 1: row_apply_narrow_rows<1>(Ps, v, ld, ctx, ldc, R, C, h, r, r_end, li, lh); break;
        case 2: row_apply_narrow_rows<2>(Ps, v, ld, ctx, ldc, R, C, h, r, r_end, li, lh); break;
        case 3: row_apply_narrow_rows<3>(Ps, v, ld, ctx, ldc, R, C, h, r, r_end, li, lh); break;
        case 4: row_apply_narrow_rows<4>(Ps, v, ld, ctx, ldc, R, C, h, r, r_end, li, lh); break;
        case 5: row_apply_narrow_rows<5>(Ps, v, ld, ctx, ldc, R, C, h, r, r_end, li, lh); break;
        case 6: row_apply_narrow_rows<6>(Ps, v, ld, ctx, ldc, R, C, h, r, r_end, li, lh); break;
        case 7: row_apply_narrow_rows<7>(Ps, v, ld, ctx, ldc, R, C, h, r, r_end, li, lh); break;
        default: row_apply_narrow_rows<8>(Ps, v, ld, ctx, ldc, R, C, h, r, r_end, li, lh); break;
    }
}

// rows of one block of row_apply_narrow_kernel (a multiple of 4: one per wave and step): enough blocks for ~4 per CU, and
// as many rows per block as that allows (the block's map staging is paid once).  Speed only: every row is computed alone.
static inline bool narrow_fits(int R, int C, int64_t ld, int64_t ldc) {
    const int64_t lim = (int64_t)1 << 31;
    return (int64_t)R * C * ld * 4 < lim && (int64_t)R * C * ldc * 4 < lim;
}
static inline int narrow_rows_per_block(int64_t rows_total, int H) {
    int64_t rpb = (rows_total * H + 1023) / 1024;
    rpb = (rpb + 3) & ~(int64_t)3;
    return (int)(rpb < 4 ? 4 : rpb > 32 ? 32 : rpb);
}

template <typename K>
static int set_lds(K kern, int bytes, const char* name) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return fail(RNAMSM_ERR_HIP, "%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e));
    return RNAMSM_OK;
}

}  // namespace rnamsm

using namespace rnamsm;

extern "C" int rnamsm_row_logits_nsplit(int R, int C, int H) {
    if (R <= 0 || C <= 0 || H <= 0) return 0;
    return choose_row_split(R, C, H, 128, 512, ROW_LOGITS_F32_MAX_ROWS).nsplit;
}

extern "C" size_t rnamsm_row_logits_workspace_bytes(int R, int C, int H) {
    if (R <= 0 || C <= 0 || H <= 0) return 0;
    return (size_t)choose_row_split(R, C, H, 128, 512, ROW_LOGITS_F32_MAX_ROWS).nsplit * H * C * C * sizeof(float);
}

static int row_logits_launch(const float* q, const float* k, int64_t ld, float* partial, int R, int C, int H,
                             int head_dim, int dtype, void* stream, int rows_per_chunk, int batch = 1, int64_t qk_bstride = 0,
                             int64_t part_bstride = 0);

extern "C" int rnamsm_row_logits(const float* q, const float* k, int64_t ld, float* partial, int R, int C, int H,
                                 int head_dim, int dtype, void* stream) {
    return row_logits_launch(q, k, ld, partial, R, C, H, head_dim, dtype, stream, 0);
}

extern "C" int rnamsm_row_chunks(int R, int C, int max_tokens_per_msa) {
    if (R <= 0 || C <= 0 || max_tokens_per_msa <= 0 || (int64_t)R * C <= max_tokens_per_msa) return 0;
    const int max_rows = max_tokens_per_msa / C > 1 ? max_tokens_per_msa / C : 1;      // modules.py:724
    return (R + max_rows - 1) / max_rows;
}

extern "C" int rnamsm_row_logits_chunked(const float* q, const float* k, int64_t ld, float* partial, int R, int C, int H,
                                         int head_dim, int rows_per_chunk, int dtype, void* stream) {
    RNAMSM_CHECK_ARG(rows_per_chunk >= 1, "row_logits_chunked: rows_per_chunk must be >= 1 (got %d)", rows_per_chunk);
    return row_logits_launch(q, k, ld, partial, R, C, H, head_dim, dtype, stream, rows_per_chunk);
}

static int row_logits_launch(const float* q, const float* k, int64_t ld, float* partial, int R, int C, int H,
                             int head_dim, int dtype, void* stream, int rows_per_chunk, int batch, int64_t qk_bstride,
                             int64_t part_bstride) {
    if (dtype != RNAMSM_F32) return fail(RNAMSM_ERR_UNSUPPORTED, "row_logits: only RNAMSM_F32 is implemented");
    RNAMSM_CHECK_ARG(q && k && partial, "row_logits: null pointer");
    RNAMSM_CHECK_ARG(head_dim == HEAD_DIM, "row_logits: head_dim must be 64 (got %d)", head_dim);
    RNAMSM_CHECK_ARG(R > 0 && R <= 1024 && C > 0 && H > 0, "row_logits: bad shape R=%d C=%d H=%d", R, C, H);
    RNAMSM_CHECK_ARG(ld >= (int64_t)H * HEAD_DIM && ld % 4 == 0 && aligned16(q) && aligned16(k),
                     "row_logits: q/k must be 16-byte aligned with ld %% 4 == 0");
    static DeviceOnce configured;
    if (configured.pending()) {
        int rc = set_lds(row_logits_kernel, ROWLOGITS_LDS_BYTES, "row_logits");
        if (rc) return rc;
        configured.mark();
    }
    RowSplit sp = choose_row_split(R, C, H, 128, 512, ROW_LOGITS_F32_MAX_ROWS);
    if (rows_per_chunk > 0) {       // slabs = the reference's row chunks (f2 on the chunked path)
        sp.rows_per_split = rows_per_chunk;
        sp.nsplit = (R + rows_per_chunk - 1) / rows_per_chunk;
    }
    const unsigned tiles_c = (C + BM - 1) / BM;
    const unsigned grid = xcd_panel_grid((unsigned)(H * sp.nsplit), tiles_c * tiles_c);
    KernelTimer timer(TC_ROW_LOGITS, 2.0 * batch * H * C * C * R * HEAD_DIM,
                      4.0 * batch * (2.0 * R * C * H * HEAD_DIM + (double)sp.nsplit * H * C * C), static_cast<hipStream_t>(stream));
    if (C <= ROW_NARROW_MAX_C && tuning().row_narrow)       // same slabs, same bits (see the kernel)
        hipLaunchKernelGGL(row_logits_narrow_kernel, dim3((unsigned)(H * sp.nsplit), batch), dim3(256), 0, static_cast<hipStream_t>(stream),
                           q, k, ld, partial, R, C, H, sp.nsplit, sp.rows_per_split, ROW_LOGITS_F32_CHAIN_ROWS, qk_bstride, part_bstride,
                           (const PackedMsa*)nullptr);
    else
        hipLaunchKernelGGL(row_logits_kernel, dim3(grid, batch), dim3(GEMM_THREADS), ROWLOGITS_LDS_BYTES,
                           static_cast<hipStream_t>(stream), q, k, ld, partial, R, C, H, sp.nsplit, sp.rows_per_split,
                           ROW_LOGITS_F32_CHAIN_ROWS * (HEAD_DIM / BK), qk_bstride, part_bstride, (const PackedMsa*)nullptr, 0);
    RNAMSM_CHECK_LAUNCH("row_logits");
    return RNAMSM_OK;
}

static int softmax_rows_launch(const float* partial, int nsplit, float* probs, int H, int C, const uint8_t* key_mask,
                               uint16_t* p_hi, uint16_t* p_lo, int64_t ldp, float plane_scale, int fmt, void* stream,
                               int64_t mask_slab_stride = 0, int batch = 1, int64_t part_bstride = 0, int64_t probs_bstride = 0,
                               int64_t mask_bstride = 0, int64_t plane_bstride = 0, float logit_scale = 1.f) {
    RNAMSM_CHECK_ARG(logit_scale > 0.f && logit_scale <= 1.f && (logit_scale == 1.f || mask_slab_stride == 0),
                     "softmax_rows: logit_scale must be in (0, 1] (and 1 on the chunked path)");
    RNAMSM_CHECK_ARG(partial && probs, "softmax_rows: null pointer");
    RNAMSM_CHECK_ARG(nsplit >= 1 && H > 0 && C > 0 && C <= 64 * SOFTMAX_MAX_PER_LANE,
                     "softmax_rows: bad shape nsplit=%d H=%d C=%d (C <= %d)", nsplit, H, C, 64 * SOFTMAX_MAX_PER_LANE);
    RNAMSM_CHECK_ARG(!p_hi || (ldp >= C && ldp % 64 == 0 && ldp <= 64 * SOFTMAX_MAX_PER_LANE && (fmt == 0 || fmt == 1)),
                     "softmax_rows: plane stride must be C rounded up to a multiple of 64, fmt 0 (bf16) or 1 (fp16)");
    const int64_t rows = (int64_t)H * C;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((rows + 3) / 4), batch);
    KernelTimer timer(TC_SOFTMAX, 0.0, batch * (4.0 * (double)(nsplit + 1) * H * C * C + (p_hi ? (p_lo ? 4.0 : 2.0) * rows * ldp : 0.0)), s);
#define SM_GO(PL_, NE_)                                                                                                            \
    hipLaunchKernelGGL((softmax_rows_kernel<PL_, NE_>), grid, dim3(256), 0, s, partial, nsplit, probs, rows, C, key_mask, p_hi, p_lo, ldp, \
                       plane_scale, mask_slab_stride, part_bstride, probs_bstride, mask_bstride, plane_bstride, (const PackedMsa*)nullptr, \
                       0, H, logit_scale)
#define SM_PICK(PL_)                                                                                                               \
    do {                                                                                                                           \
        if (C <= 64) SM_GO(PL_, 1); else if (C <= 256) SM_GO(PL_, 4); else if (C <= 512) SM_GO(PL_, 8);                            \
        else SM_GO(PL_, SOFTMAX_MAX_PER_LANE);                                                                                     \
    } while (0)
    if (!p_hi) SM_PICK(0); else if (fmt == 0) SM_PICK(1); else SM_PICK(2);
#undef SM_PICK
#undef SM_GO
    RNAMSM_CHECK_LAUNCH("softmax_rows");
    return RNAMSM_OK;
}

extern "C" int rnamsm_softmax_rows(const float* partial, int nsplit, float* probs, int H, int C,
                                   const uint8_t* key_mask, void* stream) {
    return softmax_rows_launch(partial, nsplit, probs, H, C, key_mask, nullptr, nullptr, 0, 1.f, 0, stream);
}

extern "C" int rnamsm_softmax_rows_scaled(const float* partial, int nsplit, float* probs, int H, int C, const uint8_t* key_mask,
                                          float logit_scale, void* stream) {
    return softmax_rows_launch(partial, nsplit, probs, H, C, key_mask, nullptr, nullptr, 0, 1.f, 0, stream, 0, 1, 0, 0, 0, 0, logit_scale);
}

extern "C" int rnamsm_softmax_rows_chunked(const float* partial, int nchunks, float* probs, int H, int C,
                                           const uint8_t* pad_mask, int rows_per_chunk, void* stream) {
    RNAMSM_CHECK_ARG(pad_mask && rows_per_chunk >= 1, "softmax_rows_chunked: pad_mask [R, C] and rows_per_chunk >= 1 are required");
    return softmax_rows_launch(partial, nchunks, probs, H, C, pad_mask, nullptr, nullptr, 0, 1.f, 0, stream,
                               (int64_t)rows_per_chunk * C);
}

extern "C" int rnamsm_softmax_rows_planes(const float* partial, int nsplit, float* probs, uint16_t* p_hi, uint16_t* p_lo,
                                          int64_t ldp, float plane_scale, int H, int C, const uint8_t* key_mask, int fmt,
                                          void* stream) {
    RNAMSM_CHECK_ARG(p_hi, "softmax_rows_planes: null plane pointer");
    RNAMSM_CHECK_ARG(plane_scale > 0.f && plane_scale <= 32768.f, "softmax_rows_planes: plane_scale must be in (0, 32768]");
    return softmax_rows_launch(partial, nsplit, probs, H, C, key_mask, p_hi, p_lo, ldp, plane_scale, fmt, stream);
}

static int row_apply_launch(const float* probs, const float* v, int64_t ld, float* ctx, int64_t ldc, int R, int C,
                            int H, int head_dim, uint16_t* ctx_hi, uint16_t* ctx_lo, int plane_fmt, int dtype,
                            void* stream, int batch, int64_t probs_bstride, int64_t v_bstride, int64_t ctx_bstride) {
    if (dtype != RNAMSM_F32) return fail(RNAMSM_ERR_UNSUPPORTED, "row_apply: only RNAMSM_F32 is implemented");
    RNAMSM_CHECK_ARG(probs && v && (ctx || ctx_hi), "row_apply: null pointer");
    RNAMSM_CHECK_ARG(head_dim == HEAD_DIM, "row_apply: head_dim must be 64 (got %d)", head_dim);
    RNAMSM_CHECK_ARG(R > 0 && R <= 1024 && C > 0 && H > 0, "row_apply: bad shape R=%d C=%d H=%d", R, C, H);
    RNAMSM_CHECK_ARG(ld >= (int64_t)H * HEAD_DIM && ld % 4 == 0 && aligned16(v) && ldc >= (int64_t)H * HEAD_DIM && ldc % 4 == 0,
                     "row_apply: v must be 16-byte aligned with ld, ldc %% 4 == 0");
    RNAMSM_CHECK_ARG(ctx_hi ? ((reinterpret_cast<uintptr_t>(ctx_hi) & 7u) == 0 && (plane_fmt == 0 || plane_fmt == 1)) : aligned16(ctx),
                     "row_apply: output alignment / plane format");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned tiles_i = (C + BM - 1) / BM, tiles_n = (R + 1) / 2;
    const unsigned grid = xcd_panel_grid((unsigned)H * tiles_n, tiles_i);
    RNAMSM_CHECK_ARG(batch == 1 || (!ctx_hi && probs_bstride % 4 == 0), "row_apply: a batched launch writes fp32 context from 16-byte aligned maps");
    const bool al = C % 4 == 0 && C >= 4 && aligned16(probs);
    const int out = ctx_hi ? 1 + plane_fmt : 0;
    KernelTimer timer(TC_ROW_APPLY, 2.0 * batch * H * C * C * R * HEAD_DIM, 4.0 * batch * (2.0 * R * C * H * HEAD_DIM + (double)H * C * C), s);
    // (the narrow kernel addresses V and the context through 32-bit buffer offsets: the alignment's byte ranges must fit them)
    if (out == 0 && C <= ROW_NARROW_MAX_C && tuning().row_narrow && narrow_fits(R, C, ld, ldc)) {
        const int rpb = narrow_rows_per_block((int64_t)R * batch, H);
        hipLaunchKernelGGL(row_apply_narrow_kernel, dim3((unsigned)(H * ((R + rpb - 1) / rpb)), batch), dim3(256), 0, s, probs, v, ld, ctx, ldc,
                           R, C, H, rpb, probs_bstride, v_bstride, ctx_bstride, (const PackedMsa*)nullptr, 0);
        RNAMSM_CHECK_LAUNCH("row_apply (narrow)");
        return RNAMSM_OK;
    }
#define RA_GO2(AL_, OUT_, VT_)                                                                                    \
    do {                                                                                                          \
        static DeviceOnce cfg_;                                                                                 \
        constexpr int lds_ = VT_ ? ROWAPPLY_VT_LDS_BYTES : ROWAPPLY_LDS_BYTES;                                     \
        if (cfg_.pending()) {                                                                                              \
            int rc = set_lds(row_apply_kernel<AL_, OUT_, VT_>, lds_, "row_apply");                               \
            if (rc) return rc;                                                                                    \
            cfg_.mark();                                                                                          \
        }                                                                                                         \
        hipLaunchKernelGGL((row_apply_kernel<AL_, OUT_, VT_>), dim3(grid, batch), dim3(GEMM_THREADS), lds_, s, probs, \
                           v, ld, ctx, ldc, R, C, H, ctx_hi, ctx_lo, probs_bstride, v_bstride, ctx_bstride, (const PackedMsa*)nullptr, 0, 0); \
    } while (0)
#define RA_GO(AL_, OUT_)                                                                                          \
    do {                                                                                                          \
        RA_GO2(AL_, OUT_, true);       /* V tile transposed while staged (b128 fragments); the [k][n] tile went in round 6 */ \
    } while (0)
    if (al) {
        if (out == 0) RA_GO(true, 0); else if (out == 1) RA_GO(true, 1); else RA_GO(true, 2);
    } else {
        if (out == 0) RA_GO(false, 0); else if (out == 1) RA_GO(false, 1); else RA_GO(false, 2);
    }
#undef RA_GO2
#undef RA_GO
    RNAMSM_CHECK_LAUNCH("row_apply");
    return RNAMSM_OK;
}

extern "C" int rnamsm_row_apply(const float* probs, const float* v, int64_t ld, float* ctx, int64_t ldc, int R, int C,
                                int H, int head_dim, uint16_t* ctx_hi, uint16_t* ctx_lo, int plane_fmt, int dtype,
                                void* stream) {
    return row_apply_launch(probs, v, ld, ctx, ldc, R, C, H, head_dim, ctx_hi, ctx_lo, plane_fmt, dtype, stream, 1, 0, 0, 0);
}

// K4-K6 for `batch` same-shape MSAs in one launch each (rnamsm_forward_batch): MSA b's operands lie b * stride elements on
namespace rnamsm {
int row_logits_batched(const float* q, const float* k, int64_t ld, float* partial, int R, int C, int H, int batch,
                       int64_t qk_bstride, int64_t part_bstride, void* stream) {
    return row_logits_launch(q, k, ld, partial, R, C, H, HEAD_DIM, RNAMSM_F32, stream, 0, batch, qk_bstride, part_bstride);
}
int softmax_rows_batched(const float* partial, int nsplit, float* probs, int H, int C, int batch, int64_t part_bstride,
                         int64_t probs_bstride, const uint8_t* key_mask, int64_t mask_bstride, void* stream, float logit_scale) {
    return softmax_rows_launch(partial, nsplit, probs, H, C, key_mask, nullptr, nullptr, 0, 1.f, 0, stream, 0, batch, part_bstride,
                               probs_bstride, mask_bstride, 0, logit_scale);
}
int softmax_rows_planes_batched(const float* partial, int nsplit, float* probs, uint16_t* p_hi, uint16_t* p_lo, int64_t ldp,
                                float plane_scale, int H, int C, const uint8_t* key_mask, int fmt, int batch, int64_t part_bstride,
                                int64_t probs_bstride, int64_t mask_bstride, int64_t plane_bstride, void* stream) {
    RNAMSM_CHECK_ARG(p_hi && plane_scale > 0.f && plane_scale <= 32768.f, "softmax_rows_planes_batched: bad plane arguments");
    return softmax_rows_launch(partial, nsplit, probs, H, C, key_mask, p_hi, p_lo, ldp, plane_scale, fmt, stream, 0, batch, part_bstride,
                               probs_bstride, mask_bstride, plane_bstride);
}
int row_apply_batched(const float* probs, const float* v, int64_t ld, float* ctx, int64_t ldc, int R, int C, int H, int batch,
                      int64_t probs_bstride, int64_t v_bstride, int64_t ctx_bstride, void* stream) {
    return row_apply_launch(probs, v, ld, ctx, ldc, R, C, H, HEAD_DIM, nullptr, nullptr, 0, RNAMSM_F32, stream, batch, probs_bstride,
                            v_bstride, ctx_bstride);
}

// ---- K4-K6 of a token-packed batch (rnamsm_forward_packed): one launch each, gridDim.y = alignment, gridDim.x sized for the
// alignment that needs most blocks; every alignment keeps the slab split of its own forward (PackedMsa::nsplit)
int row_logits_packed(const float* q, const float* k, int64_t ld, float* partial, int H, const PackedMsa* pk, const PackedMsa* host,
                      int B, void* stream) {
    static DeviceOnce configured;
    if (configured.pending()) {
        int rc = set_lds(row_logits_kernel, ROWLOGITS_LDS_BYTES, "row_logits");
        if (rc) return rc;
        configured.mark();
    }
    // alignments of <= 64 columns go to the narrow kernel, the others to the tile kernel: two launches over the same descriptor
    // table, each leaving the other's alignments alone (the kernels agree bit for bit, the split is speed only)
    const bool narrow_on = tuning().row_narrow != 0;
    unsigned grid = 0, grid_narrow = 0;
    double flops = 0.0, bytes = 0.0;
    for (int b = 0; b < B; ++b) {
        const PackedMsa& m = host[b];
        if (narrow_on && m.C <= ROW_NARROW_MAX_C) {
            const unsigned g = (unsigned)(H * m.nsplit);
            grid_narrow = g > grid_narrow ? g : grid_narrow;
        } else {
            const unsigned tiles_c = (m.C + BM - 1) / BM;
            const unsigned g = xcd_panel_grid((unsigned)(H * m.nsplit), tiles_c * tiles_c);
            grid = g > grid ? g : grid;
        }
        flops += 2.0 * H * m.C * m.C * (double)m.R * HEAD_DIM;
        bytes += 4.0 * (2.0 * m.R * m.C * H * HEAD_DIM + (double)m.nsplit * H * m.C * m.C);
    }
    KernelTimer timer(TC_ROW_LOGITS, flops, bytes, static_cast<hipStream_t>(stream));
    if (grid_narrow)
        hipLaunchKernelGGL(row_logits_narrow_kernel, dim3(grid_narrow, B), dim3(256), 0, static_cast<hipStream_t>(stream), q, k, ld, partial,
                           0, 0, H, 0, 0, ROW_LOGITS_F32_CHAIN_ROWS, (int64_t)0, (int64_t)0, pk);
    if (grid)
        hipLaunchKernelGGL(row_logits_kernel, dim3(grid, B), dim3(GEMM_THREADS), ROWLOGITS_LDS_BYTES, static_cast<hipStream_t>(stream), q, k,
                           ld, partial, 0, 0, H, 0, 0, ROW_LOGITS_F32_CHAIN_ROWS * (HEAD_DIM / BK), (int64_t)0, (int64_t)0, pk, narrow_on ? 1 : 0);
    RNAMSM_CHECK_LAUNCH("row_logits (packed)");
    return RNAMSM_OK;
}
int softmax_rows_packed(const float* partial, float* row_attn, int layer, int H, const PackedMsa* pk, const PackedMsa* host, int B,
                        void* stream) {
    int max_C = 0;
    double bytes = 0.0;
    for (int b = 0; b < B; ++b) {
        max_C = host[b].C > max_C ? host[b].C : max_C;
        bytes += 4.0 * (double)(host[b].nsplit + 1) * H * host[b].C * host[b].C;
    }
    RNAMSM_CHECK_ARG(max_C <= 64 * SOFTMAX_MAX_PER_LANE, "softmax_rows (packed): C <= %d", 64 * SOFTMAX_MAX_PER_LANE);
    hipStream_t s = static_cast<hipStream_t>(stream);
    KernelTimer timer(TC_SOFTMAX, 0.0, bytes, s);
#define SM_PK(NE_)                                                                                                                  \
    hipLaunchKernelGGL((softmax_rows_kernel<0, NE_>), dim3((unsigned)(((int64_t)H * max_C + 3) / 4), B), dim3(256), 0, s, partial, 0, row_attn, \
                       (int64_t)0, 0, (const uint8_t*)nullptr, (uint16_t*)nullptr, (uint16_t*)nullptr, (int64_t)0, 1.f, (int64_t)0,    \
                       (int64_t)0, (int64_t)0, (int64_t)0, (int64_t)0, pk, layer, H, 1.f)
    if (max_C <= 64) SM_PK(1); else if (max_C <= 256) SM_PK(4); else if (max_C <= 512) SM_PK(8); else SM_PK(SOFTMAX_MAX_PER_LANE);
#undef SM_PK
    RNAMSM_CHECK_LAUNCH("softmax_rows (packed)");
    return RNAMSM_OK;
}
int row_apply_packed(const float* row_attn, int layer, const float* v, int64_t ld, float* ctx, int64_t ldc, int H, const PackedMsa* pk,
                     const PackedMsa* host, int B, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    bool narrow_on = tuning().row_narrow != 0;
    for (int b = 0; b < B && narrow_on; ++b)                     // (32-bit buffer offsets: never an issue at this model's row strides)
        if (host[b].C <= ROW_NARROW_MAX_C && !narrow_fits(host[b].R, host[b].C, ld, ldc)) narrow_on = false;
    unsigned grid = 0, grid_narrow = 0;
    double flops = 0.0, bytes = 0.0;
    int64_t narrow_rows = 0;
    int narrow_max_r = 0;
    for (int b = 0; b < B; ++b) {
        const PackedMsa& m = host[b];
        if (narrow_on && m.C <= ROW_NARROW_MAX_C) {
            narrow_rows += m.R;
            narrow_max_r = m.R > narrow_max_r ? m.R : narrow_max_r;
        } else {
            const unsigned g = xcd_panel_grid((unsigned)H * ((m.R + 1) / 2), (m.C + BM - 1) / BM);
            grid = g > grid ? g : grid;
        }
        flops += 2.0 * H * m.C * m.C * (double)m.R * HEAD_DIM;
        bytes += 4.0 * (2.0 * m.R * m.C * H * HEAD_DIM + (double)H * m.C * m.C);
    }
    KernelTimer timer(TC_ROW_APPLY, flops, bytes, s);
    if (narrow_rows) {
        const int rpb = narrow_rows_per_block(narrow_rows, H);
        grid_narrow = (unsigned)(H * ((narrow_max_r + rpb - 1) / rpb));
        hipLaunchKernelGGL(row_apply_narrow_kernel, dim3(grid_narrow, B), dim3(256), 0, s, row_attn, v, ld, ctx, ldc, 0, 0, H, rpb,
                           (int64_t)0, (int64_t)0, (int64_t)0, pk, layer);
        RNAMSM_CHECK_LAUNCH("row_apply (packed, narrow)");
    }
    if (!grid) return RNAMSM_OK;
    // (maps of odd widths are not 16-byte aligned: the scalar-load instance for every alignment of the batch)
#define RA_PK(VT_)                                                                                                \
    do {                                                                                                          \
        static DeviceOnce cfg_;                                                                                   \
        constexpr int lds_ = VT_ ? ROWAPPLY_VT_LDS_BYTES : ROWAPPLY_LDS_BYTES;                                     \
        if (cfg_.pending()) {                                                                                     \
            int rc = set_lds(row_apply_kernel<false, 0, VT_>, lds_, "row_apply");                                 \
            if (rc) return rc;                                                                                    \
            cfg_.mark();                                                                                          \
        }                                                                                                         \
        hipLaunchKernelGGL((row_apply_kernel<false, 0, VT_>), dim3(grid, B), dim3(GEMM_THREADS), lds_, s, row_attn, v, ld, ctx, ldc, 0, 0, \
                           H, (uint16_t*)nullptr, (uint16_t*)nullptr, (int64_t)0, (int64_t)0, (int64_t)0, pk, layer, narrow_on ? 1 : 0);  \
    } while (0)
    RA_PK(true);
#undef RA_PK
    RNAMSM_CHECK_LAUNCH("row_apply (packed)");
    return RNAMSM_OK;
}
}  // namespace rnamsm
