// K7: fused column attention (ColumnSelfAttention.compute_attention_update, modules.py:875-924).
//
// For every alignment column c and head h:  ctx[:,c,h,:] = softmax_j(q[:,c,h,:] k[:,c,h,:]^T) v[:,c,h,:]  with the
// R x R score matrix kept in registers -- the reference materialises [H,C,R,R] probabilities (1.6 GB at R=256,
// C=512) only to discard them.
//
// Structure (exact-fp32 MFMA 32x32x2, online softmax):
//   block = 4 waves = 128 query rows of one (c, h); wave = 32 query rows, its Q fragment lives in 32 VGPRs.  Two blocks
//   are resident per CU (the kernel needs ~256 VGPRs): one block's prologue/epilogue (strided q/k/v rows, ~10 us)
//   hides behind the other's MFMAs, and the two waves of a SIMD belong to different blocks, so their softmax (VALU)
//   and MFMA phases interleave instead of coinciding.  The i-blocks of one (c, h) share an XCD and re-read K/V from L2.
//   Keys/values stream through LDS in 64-row chunks (double buffered, register-staged global loads issued before
//   the MFMAs that hide them).
//   Scores are computed TRANSPOSED, S^T = K Q^T (A = K rows, B = Q): the accumulator then holds the query row on the
//   lane and the keys in the 16 registers, so (a) softmax max/sum are per-lane reductions plus one lane-half
//   exchange, (b) exp(S^T) is, register for register, the B operand of O^T += V^T P^T with no data movement
//   (the k index of MFMA step t is key (t&3)+8(t>>2)+4*half -- the accumulator's own row map), and (c) O^T also has
//   the query row on the lane, so the online-softmax rescale is a per-lane scalar multiply.
// Roofline: MFMA-bound: 4*R*R*64 flops per (c,h) vs 4*R*256 B of q,k,v,ctx (64 flop/B at R=256).
#include "tile16.h"
#include <type_traits>

namespace rnamsm {

constexpr int CA_THREADS = 256;
constexpr int CA_ROWS = 128;          // query rows per block
constexpr int CA_JC = 64;             // keys per chunk
constexpr int CA_HD = 64;             // head dim
constexpr int CA_LDD = CA_HD + 4;     // padded LDS row stride (floats): 16 rows -> 16 distinct 16-B slots
constexpr int CA_TILE = CA_JC * CA_LDD;
constexpr int CA_LDS_BYTES = 2 * 2 * CA_TILE * 4;

// MASKED: a padding mask is applied (f2); the un-masked instance carries no mask code at all.
// OUT: 0 = fp32 context; 1 / 2 = bf16 / fp16 hi(+lo) planes, the pre-split A operand of the following out_proj GEMM.
template <bool MASKED, int OUT>
__global__ __launch_bounds__(CA_THREADS, 2) void col_attn_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int64_t ld,
    float* __restrict__ ctx, int64_t ldc, int R, int C, int H, const uint8_t* __restrict__ pad_mask,
    uint16_t* __restrict__ ctx_hi, uint16_t* __restrict__ ctx_lo, int q_rows, int64_t qkv_bstride, int64_t ctx_bstride,
    const PackedMsa* /* packed batches run on col_attn_dma_kernel */, int, int) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    q += blockIdx.y * qkv_bstride;        // batched launch (rnamsm_forward_batch): MSA blockIdx.y, fp32 context
    k += blockIdx.y * qkv_bstride;
    v += blockIdx.y * qkv_bstride;
    if (OUT == 0) ctx += blockIdx.y * ctx_bstride;
    if (MASKED) pad_mask += blockIdx.y * ((int64_t)R * C);         // its own [R, C] mask
    float* Ks = smem;                    // [2][CA_JC][CA_LDD]
    float* Vs = smem + 2 * CA_TILE;      // [2][CA_JC][CA_LDD]

    const unsigned iblocks = (q_rows + CA_ROWS - 1) / CA_ROWS;       // query rows [0, q_rows) only (q_rows == R: all)
    unsigned prob, ib;
    if (!xcd_panel_map(blockIdx.x, (unsigned)C * H, iblocks, prob, ib)) return;
    const int c = prob / H, h = prob % H;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int irow0 = ib * CA_ROWS + wave * 32;              // first query row of this wave
    const bool active = irow0 < q_rows;                      // wave-uniform
    const int64_t col_off = (int64_t)c * ld + h * CA_HD;     // + r*C*ld selects the alignment row

    // Q fragment: lane (i, half) holds q[i][8kk + 4*half + s], kk = 0..7, s = 0..3 (B operand of S^T = K Q^T)
    f32x4 qf[8];
    {
        const int qi = min(irow0 + li, R - 1);
        const float* qp = q + (int64_t)qi * C * ld + col_off + 4 * lh;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) qf[kk] = *reinterpret_cast<const f32x4*>(qp + 8 * kk);
    }

    // staging map: thread -> (key rows tid/16 + 16*s, 16-B chunk tid%16) of the [64][64] K and V chunks
    constexpr int NST = CA_JC * 16 / CA_THREADS, JSTEP = CA_THREADS / 16;
    const int d4 = threadIdx.x & 15, jr = threadIdx.x >> 4;
    f32x4 sk[NST], sv[NST];
    auto load_chunk = [&](int ch) {
#pragma unroll
        for (int s = 0; s < NST; ++s) {
            // keys past R are clamped to the last key (branch-free): their scores are masked to -inf and their
            // (finite) values only ever meet P = 0
            const int j = min(ch * CA_JC + jr + JSTEP * s, R - 1);
            const int64_t off = (int64_t)j * C * ld + col_off + d4 * 4;
            sk[s] = *reinterpret_cast<const f32x4*>(k + off);
            sv[s] = *reinterpret_cast<const f32x4*>(v + off);
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int s = 0; s < NST; ++s) {
            *reinterpret_cast<f32x4*>(&Ks[buf * CA_TILE + (jr + JSTEP * s) * CA_LDD + d4 * 4]) = sk[s];
            *reinterpret_cast<f32x4*>(&Vs[buf * CA_TILE + (jr + JSTEP * s) * CA_LDD + d4 * 4]) = sv[s];
        }
    };

    f32x16 o0, o1;                       // O^T tiles: head dims [0,32) and [32,64) x 32 query rows
#pragma unroll
    for (int t = 0; t < 16; ++t) { o0[t] = 0.f; o1[t] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    // K fragments of one 32-key tile: lane (key j, half) holds K[j][8kk + 4*half + s] (A operand of S^T = K Q^T)
    f32x4 kf[8];
    auto read_kfrags = [&](const float* Kc, int jt) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
            kf[kk] = *reinterpret_cast<const f32x4*>(&Kc[(jt * 32 + li) * CA_LDD + 8 * kk + 4 * lh]);
    };

    // One 32-key tile, one basic block, order pinned (hipcc otherwise reuses one register quad for every K fragment
    // and waits lgkmcnt(0) after each ds_read: the LDS latency was exposed every 4 MFMAs):
    //   S^T = K Q^T   32 MFMAs on fragments already in registers, the 32 V values of this tile requested between them
    //   softmax       VALU on the accumulator (query row = lane, keys = registers x lane half), hardware exp2
    //   O^T += V^T P^T 32 MFMAs on the V registers, the NEXT tile's 8 K-fragment reads requested between them
    auto tile = [&](const float* Kc, const float* Vc, int jt, int jbase, bool next_in_chunk) {
        f32x16 s;
#pragma unroll
        for (int t = 0; t < 16; ++t) s[t] = 0.f;
        float vv[32];
#pragma unroll
        for (int t = 0; t < 16; ++t) {       // MFMA step t of P.V contracts key (t&3)+8(t>>2)+4*half
            const int jl = jt * 32 + (t & 3) + 8 * (t >> 2) + 4 * lh;
            vv[2 * t] = Vc[jl * CA_LDD + li];
            vv[2 * t + 1] = Vc[jl * CA_LDD + 32 + li];
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int e = 0; e < 4; ++e) s = mfma32(kf[kk][e], qf[kk][e], s);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);     // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- online softmax; keys >= R (last tile only) are masked branch-free.  __expf = v_exp_f32(x*log2e):
        // relative error <= ~2e-6 for the |x| <= 20 that matter, 100x inside the parity bar.
        const int limit = R - jbase;
        if (limit < 32) {        // block-uniform: only a ragged last tile holds keys >= R (32 compare/select per lane saved elsewhere)
#pragma unroll
            for (int t = 0; t < 16; ++t)
                s[t] = ((t & 3) + 8 * (t >> 2) + 4 * lh < limit) ? s[t] : -INFINITY;
        }
        if (MASKED) {       // f2: masked_fill(padding_mask, -10000) on padded keys of this column (modules.py:911-915)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int j = jbase + (t & 3) + 8 * (t >> 2) + 4 * lh;
                if (j < R && pad_mask[(int64_t)j * C + c]) s[t] = -10000.f;
            }
        }
        float mx = fmaxf(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])), fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7])));
        mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(s[8], s[9]), fmaxf(s[10], s[11])), fmaxf(fmaxf(s[12], s[13]), fmaxf(s[14], s[15]))));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);                    // finite: key jbase is valid
        const float alpha = __expf(m_run - m_new);               // 0 on the first tile
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            s[t] = __expf(s[t] - m_new);
            psum += s[t];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < 16; ++t) { o0[t] *= alpha; o1[t] *= alpha; }
        __builtin_amdgcn_sched_barrier(0);
        // ---- O^T += V^T P^T: register t of P, as it stands, is the B operand of step t
        if (next_in_chunk) read_kfrags(Kc, jt + 1);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            o0 = mfma32(vv[2 * t], s[t], o0);
            o1 = mfma32(vv[2 * t + 1], s[t], o1);
        }
        if (next_in_chunk) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    const int nch = (R + CA_JC - 1) / CA_JC;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    if (active) read_kfrags(Ks, 0);

    for (int ch = 0; ch < nch; ++ch) {
        const int cur = ch & 1;
        const bool more = ch + 1 < nch;
        if (more) load_chunk(ch + 1);
        if (active) {
            const float* Kc = Ks + cur * CA_TILE;
            const float* Vc = Vs + cur * CA_TILE;
            const int jbase = ch * CA_JC;
            if (jbase + 32 < R) {                                // block-uniform: both tiles of the chunk hold keys
                tile(Kc, Vc, 0, jbase, true);
                tile(Kc, Vc, 1, jbase + 32, false);
            } else {
                tile(Kc, Vc, 0, jbase, false);
            }
        }
        if (more) store_chunk(cur ^ 1);
        __syncthreads();
        if (more && active) read_kfrags(Ks + (cur ^ 1) * CA_TILE, 0);
    }

    if (active) {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.f / l_tot;
        const int i = irow0 + li;
        if (i < q_rows) {
            const int64_t ooff = ((int64_t)i * C + c) * ldc + h * CA_HD + 4 * lh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {      // registers 4g..4g+3 are head dims 8g + 4*half + {0..3}
                const f32x4 a = f32x4{o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv};
                const f32x4 b = f32x4{o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv};
                if (OUT == 0) {
                    *reinterpret_cast<f32x4*>(ctx + ooff + 8 * g) = a;
                    *reinterpret_cast<f32x4*>(ctx + ooff + 32 + 8 * g) = b;
                } else {
                    typedef typename Half16<(OUT > 0 ? OUT - 1 : 0)>::T Hh;
                    typedef Hh H4 __attribute__((ext_vector_type(4)));
                    H4 ah, al, bh, bl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float av = pinned(a[e]), bv = pinned(b[e]);
                        ah[e] = (Hh)av; al[e] = (Hh)(av - (float)ah[e]);
                        bh[e] = (Hh)bv; bl[e] = (Hh)(bv - (float)bh[e]);
                    }
                    *reinterpret_cast<H4*>(ctx_hi + ooff + 8 * g) = ah;
                    *reinterpret_cast<H4*>(ctx_hi + ooff + 32 + 8 * g) = bh;
                    if (ctx_lo) {
                        *reinterpret_cast<H4*>(ctx_lo + ooff + 8 * g) = al;
                        *reinterpret_cast<H4*>(ctx_lo + ooff + 32 + 8 * g) = bl;
                    }
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------- K7, LDS-DMA variant
// Same arithmetic as col_attn_kernel, tile for tile (32 keys, the same MFMA and softmax sequence: outputs are
// bit-identical), different data path.  A block's main loop at R = 256 is shorter than the fixed cost around it (strided
// q rows, first chunk, output stores), which only OTHER resident blocks hide -- so this variant is built for occupancy:
//   * K/V chunks are staged by global_load_lds_dwordx4 (no staging registers, no ds_write, no address VALU per element);
//     a DMA writes lanes linearly, so rows are unpadded 256-B lines and the bank spread is an XOR swizzle applied to the
//     SOURCE chunk and to the READ: physical 16-B chunk = logical ^ (row & 15) -- the 16 lanes of a ds_read_b128 group
//     hold rows distinct mod 16 (K fragments), and a 32-lane ds_read_b32 group covers 8 chunks that share bit 3 (V);
//   * 32-key chunks (one tile per barrier): 2 x 16 KB of LDS per block instead of 70 KB;
//   * <= 168 registers: three blocks per CU instead of two.
constexpr int CD_JC = 32;
constexpr int CD_ROWB = 256;                     // bytes per LDS row (64 floats)
constexpr int CD_TILE = CD_JC * CD_ROWB;         // one operand chunk
constexpr int CD_BUF = 2 * CD_TILE;              // K chunk then V chunk
constexpr int CD_LDS_BYTES = 2 * CD_BUF;         // double buffered: 32 KB

// index of the V lane base for MFMA step t / head-dim tile dt (the swizzle constant K = j0 ^ 8 dt takes the values
// {0..3, 8..11}), and the row of step t inside the 32-key tile without its lane-half part
__host__ __device__ constexpr int cd_vidx(int t, int dt) {
    return ((((t & 3) + 8 * ((t >> 2) & 1)) ^ (8 * dt)) & 3) + 4 * (((((t & 3) + 8 * ((t >> 2) & 1)) ^ (8 * dt)) >> 3) & 1);
}
__host__ __device__ constexpr int cd_vrow(int t) { return (t & 3) + 8 * (t >> 2); }

// PRE (round 4): q arrives PRESCALED by dh^-1/2 * log2(e) (the QKV GEMM's epilogue), scores are in log2 units and the kernel
// first runs a FAST loop with NO running maximum -- p = v_exp_f32(s) on the raw score, no max chain, no half exchange, no
// rescale of the accumulators (softmax is invariant to the reference point; fp32 p and sums keep their relative precision
// whatever the scale) -- and looks at the row sums afterwards: outside [2^-64, 2^100] or not finite, some score left the
// exponent range; the block votes and redoes its column on the TRACKED loop (the online softmax below, in log2 units).
template <bool MASKED, int OUT, bool PRE = false>
__global__ __launch_bounds__(CA_THREADS, MASKED ? 2 : 3) void col_attn_dma_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int64_t ld,
    float* __restrict__ ctx, int64_t ldc, int R, int C, int H, const uint8_t* __restrict__ pad_mask,
    uint16_t* __restrict__ ctx_hi, uint16_t* __restrict__ ctx_lo, int q_rows, int64_t qkv_bstride, int64_t ctx_bstride,
    const PackedMsa* __restrict__ pk, int pk_skip_shallow, int pre_tracked_only) {
    extern __shared__ __attribute__((aligned(16))) char smem_b[];
    if (pk) {
        // token-packed batch (rnamsm_forward_packed; unmasked, fp32 context): alignment blockIdx.y's own shape and token offset;
        // the shallow ones (R <= 16) belong to col_attn_small_kernel's launch when pk_skip_shallow is set
        const PackedMsa& m = pk[blockIdx.y];
        if (pk_skip_shallow && m.R <= 16) return;
        R = m.R; C = m.C; q_rows = m.R;
        q += m.tok0 * ld;
        k += m.tok0 * ld;
        v += m.tok0 * ld;
        if (OUT == 0) ctx += m.tok0 * ldc;
    } else {
        q += blockIdx.y * qkv_bstride;        // batched launch (rnamsm_forward_batch): MSA blockIdx.y, fp32 context
        k += blockIdx.y * qkv_bstride;
        v += blockIdx.y * qkv_bstride;
        if (OUT == 0) ctx += blockIdx.y * ctx_bstride;
        if (MASKED) pad_mask += blockIdx.y * ((int64_t)R * C);         // its own [R, C] mask
    }

    const unsigned iblocks = (q_rows + CA_ROWS - 1) / CA_ROWS;       // query rows [0, q_rows) only (q_rows == R: all)
    unsigned prob, ib;
    if (!xcd_panel_map(blockIdx.x, (unsigned)C * H, iblocks, prob, ib)) return;
    const int c = prob / H, h = prob % H;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int irow0 = ib * CA_ROWS + wave * 32;
    const bool active = irow0 < q_rows;                      // wave-uniform
    bool computing = active;                                 // (the fallback pass narrows it to the waves that asked for it)
    const int64_t col_off = (int64_t)c * ld + h * CA_HD;

    f32x4 qf[8];
    {
        const int qi = min(irow0 + li, R - 1);
        const float* qp = q + (int64_t)qi * C * ld + col_off + 4 * lh;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) qf[kk] = *reinterpret_cast<const f32x4*>(qp + 8 * kk);
    }

    // DMA map: one wave instruction = 4 key rows x 256 B (lane -> row lane/16, physical chunk lane%16); a chunk is 8 such
    // groups per operand, wave w moves groups w and w+4.  Keys past R are clamped (scores masked to -inf, V meets P = 0).
    // The per-lane source offsets of chunk 0 are kept; a full chunk adds ch * 32 rows to them (one 64-bit mad each).
    const int drow = lane >> 4, dchunk = lane & 15;
    const int64_t row_bytes = (int64_t)C * ld;                           // elements between consecutive alignment rows
    int64_t doff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 4 * (wave + 4 * j) + drow;
        doff[j] = (int64_t)row * row_bytes + col_off + ((dchunk ^ (row & 15)) << 2);
    }
    auto issue = [&](int ch, int buf) {
        char* base = smem_b + buf * CD_BUF;
        if ((ch + 1) * CD_JC <= R) {                                     // block-uniform: every key of the chunk exists
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int64_t off = doff[j] + (int64_t)(ch * CD_JC) * row_bytes;
                const int g = wave + 4 * j;
                __builtin_amdgcn_global_load_lds((gptr_t)(k + off), (lptr_t)(base + g * 4 * CD_ROWB), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)(v + off), (lptr_t)(base + CD_TILE + g * 4 * CD_ROWB), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int g = wave + 4 * j, row = 4 * g + drow;
                const int64_t off = (int64_t)min(ch * CD_JC + row, R - 1) * row_bytes + col_off + ((dchunk ^ (row & 15)) << 2);
                __builtin_amdgcn_global_load_lds((gptr_t)(k + off), (lptr_t)(base + g * 4 * CD_ROWB), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)(v + off), (lptr_t)(base + CD_TILE + g * 4 * CD_ROWB), 16, 0, 0);
            }
        }
    };

    f32x16 o0, o1;
#pragma unroll
    for (int t = 0; t < 16; ++t) { o0[t] = 0.f; o1[t] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    // LDS read addresses.  The swizzle XORs a lane term with a per-read constant, which no immediate offset can express; the
    // constants take only 8 values per operand, so 8 + 8 lane bases are formed once and every read of the loop is
    // base + immediate (row, buffer) -- the address arithmetic was a third of the tile's VALU instructions.
    //   K fragment kk of row li:  chunk (2kk + lh) ^ (li & 15) = [(li & 15) ^ lh] ^ 2kk
    //   V value (step t, d tile): row jl = (t&3) + 8(t>>2) + 4 lh, chunk [(li>>2) ^ 4 lh] ^ [j0 ^ 8 dt], j0 = (t&3) + 8((t>>2)&1)
    const char* kb[8];
    const char* vb8[8];
    {
        const int lk = (li & 15) ^ lh, lv = (li >> 2) ^ (lh << 2);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) kb[kk] = smem_b + li * CD_ROWB + ((lk ^ (2 * kk)) << 4);
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const int K = (x & 3) + 8 * (x >> 2);
            vb8[x] = smem_b + CD_TILE + lh * 4 * CD_ROWB + (li & 3) * 4 + ((lv ^ K) << 4);
        }
    }
    // V value of MFMA step t (key (t&3)+8(t>>2)+4*half), head-dim tile dt, buffer BUF: base + immediate
#define CD_VREAD(BUF_, t_, dt_) \
    (*reinterpret_cast<const float*>(vb8[cd_vidx(t_, dt_)] + (BUF_) * CD_BUF + cd_vrow(t_) * CD_ROWB))

    // One 32-key tile, order pinned.  V values travel in quarters of 8 (steps 4q..4q+3, both head-dim tiles) that ping-pong:
    // quarter 0 under the QK^T MFMAs, quarter q+1 under the PV MFMAs of quarter q -- 16 live V registers instead of 32.
    auto tile = [&](auto bufc, int jbase, auto trk_tag) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value;
        constexpr bool TRK = decltype(trk_tag)::value;
        f32x4 kf[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) kf[kk] = *reinterpret_cast<const f32x4*>(kb[kk] + BUF * CD_BUF);
        f32x16 s;
#pragma unroll
        for (int t = 0; t < 16; ++t) s[t] = 0.f;
        float vq[2][8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            vq[0][2 * t] = CD_VREAD(BUF, t, 0);
            vq[0][2 * t + 1] = CD_VREAD(BUF, t, 1);
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int e = 0; e < 4; ++e) s = mfma32(kf[kk][e], qf[kk][e], s);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);     // 4 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
        }
        __builtin_amdgcn_sched_barrier(0);
        const int limit = R - jbase;
        if (limit < 32) {        // block-uniform: only a ragged last tile holds keys >= R
#pragma unroll
            for (int t = 0; t < 16; ++t)
                s[t] = ((t & 3) + 8 * (t >> 2) + 4 * lh < limit) ? s[t] : -INFINITY;
        }
        if (MASKED) {       // f2: masked_fill(padding_mask, -10000) on padded keys of this column (modules.py:911-915)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int j = jbase + (t & 3) + 8 * (t >> 2) + 4 * lh;
                if (j < R && pad_mask[(int64_t)j * C + c]) s[t] = -10000.f;
            }
        }
        if (TRK) {
            float mx = fmaxf(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])), fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7])));
            mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(s[8], s[9]), fmaxf(s[10], s[11])), fmaxf(fmaxf(s[12], s[13]), fmaxf(s[14], s[15]))));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = PRE ? __builtin_amdgcn_exp2f(m_run - m_new) : __expf(m_run - m_new);
            float psum = 0.f;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                s[t] = PRE ? __builtin_amdgcn_exp2f(s[t] - m_new) : __expf(s[t] - m_new);
                psum += s[t];
            }
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int t = 0; t < 16; ++t) { o0[t] *= alpha; o1[t] *= alpha; }
        } else {
            float psum = 0.f;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                s[t] = __builtin_amdgcn_exp2f(s[t]);          // keys past R hold -inf: 0
                psum += s[t];
            }
            l_run += psum;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            if (qd < 3) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    vq[(qd + 1) & 1][2 * t] = CD_VREAD(BUF, 4 * (qd + 1) + t, 0);
                    vq[(qd + 1) & 1][2 * t + 1] = CD_VREAD(BUF, 4 * (qd + 1) + t, 1);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                o0 = mfma32(vq[qd & 1][2 * t], s[4 * qd + t], o0);
                o1 = mfma32(vq[qd & 1][2 * t + 1], s[4 * qd + t], o1);
            }
            if (qd < 3) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const int nch = (R + CD_JC - 1) / CD_JC;
    auto run = [&](auto trk_tag) __attribute__((always_inline)) {
        issue(0, 0);
        for (int ch = 0; ch < nch; ch += 2) {                     // unrolled by two: the buffer index is a compile-time constant
            wait_dma_then_barrier<0>();      // chunk ch has landed (every wave's share) and the other buffer is free again
            if (ch + 1 < nch) issue(ch + 1, 1);
            if (computing) tile(std::integral_constant<int, 0>{}, ch * CD_JC, trk_tag);
            if (ch + 1 < nch) {
                wait_dma_then_barrier<0>();
                if (ch + 2 < nch) issue(ch + 2, 0);
                if (computing) tile(std::integral_constant<int, 1>{}, (ch + 1) * CD_JC, trk_tag);
            }
        }
    };
    if (PRE && !pre_tracked_only) {
        run(std::integral_constant<bool, false>{});
        // the block shares the ring, so a second pass over the keys is the whole block's -- but only the WAVES that asked for it
        // recompute (round 5; before, one bad row sent all 128 through the TRACKED loop): a row's arithmetic is a function of its
        // own wave's 32 rows, whatever block they sit in.  (That made cutting the blocks of a mostly empty last round into 64-query
        // halves a bit-identical change; measured 0.5-1 % SLOWER on deep, narrow alignments and removed: EXPERIMENTS R5.11.)
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        // lower bound 2^-64 (not 2^-100): v_exp_f32 flushes below 2^-126, so a row whose LARGEST weight is 2^-w silently drops keys
        // 2^-(126-w) below it; at 2^-64 (less log2 R <= 10 of spread) every dropped key is < 2^-52 of the sum, under fp32 rounding
        const bool bad = active && !(l_tot < 0x1p100f && l_tot > 0x1p-64f);
        __shared__ int flags[4];
        const int wave_bad = __builtin_amdgcn_ballot_w64(bad) != 0;
        if (lane == 0) flags[wave] = wave_bad;
        wait_dma_then_barrier<0>();
        const int any_bad = flags[0] | flags[1] | flags[2] | flags[3];       // block-uniform
        if (any_bad) {
            wait_dma_then_barrier<0>();      // every wave has read the flags and is done with the ring
            computing = wave_bad != 0;
            if (computing) {
#pragma unroll
                for (int t = 0; t < 16; ++t) { o0[t] = 0.f; o1[t] = 0.f; }
                m_run = -INFINITY; l_run = 0.f;
            }
            run(std::integral_constant<bool, true>{});
        }
    } else {
        run(std::integral_constant<bool, true>{});
    }
#undef CD_VREAD

    if (active) {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.f / l_tot;
        const int i = irow0 + li;
        if (i < q_rows) {
            const int64_t ooff = ((int64_t)i * C + c) * ldc + h * CA_HD + 4 * lh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 a = f32x4{o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv};
                const f32x4 b = f32x4{o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv};
                if (OUT == 0) {
                    *reinterpret_cast<f32x4*>(ctx + ooff + 8 * g) = a;
                    *reinterpret_cast<f32x4*>(ctx + ooff + 32 + 8 * g) = b;
                } else {
                    typedef typename Half16<(OUT > 0 ? OUT - 1 : 0)>::T Hh;
                    typedef Hh H4 __attribute__((ext_vector_type(4)));
                    H4 ah, al, bh, bl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float av = pinned(a[e]), bv = pinned(b[e]);
                        ah[e] = (Hh)av; al[e] = (Hh)(av - (float)ah[e]);
                        bh[e] = (Hh)bv; bl[e] = (Hh)(bv - (float)bh[e]);
                    }
                    *reinterpret_cast<H4*>(ctx_hi + ooff + 8 * g) = ah;
                    *reinterpret_cast<H4*>(ctx_hi + ooff + 32 + 8 * g) = bh;
                    if (ctx_lo) {
                        *reinterpret_cast<H4*>(ctx_lo + ooff + 8 * g) = al;
                        *reinterpret_cast<H4*>(ctx_lo + ooff + 32 + 8 * g) = bl;
                    }
                }
            }
        }
    }
}

}  // namespace rnamsm

using namespace rnamsm;

static int col_attn_launch(const float* q, const float* k, const float* v, int64_t ld, float* ctx, int64_t ldc, int R, int C,
                           int H, int head_dim, const uint8_t* pad_mask, uint16_t* ctx_hi, uint16_t* ctx_lo, int plane_fmt,
                           int dtype, void* stream, int q_rows, int batch = 1, int64_t qkv_bstride = 0, int64_t ctx_bstride = 0,
                           bool prescaled = false);

extern "C" int rnamsm_col_attn_fused(const float* q, const float* k, const float* v, int64_t ld, float* ctx,
                                     int64_t ldc, int R, int C, int H, int head_dim, const uint8_t* pad_mask,
                                     uint16_t* ctx_hi, uint16_t* ctx_lo, int plane_fmt, int dtype, void* stream) {
    return col_attn_launch(q, k, v, ld, ctx, ldc, R, C, H, head_dim, pad_mask, ctx_hi, ctx_lo, plane_fmt, dtype, stream, R);
}

extern "C" int rnamsm_col_attn_fused_queries(const float* q, const float* k, const float* v, int64_t ld, float* ctx,
                                             int64_t ldc, int R, int C, int H, int head_dim, int q_rows,
                                             const uint8_t* pad_mask, int dtype, void* stream) {
    RNAMSM_CHECK_ARG(q_rows >= 1 && q_rows <= R, "col_attn_queries: q_rows must be in [1, R] (got %d, R=%d)", q_rows, R);
    return col_attn_launch(q, k, v, ld, ctx, ldc, R, C, H, head_dim, pad_mask, nullptr, nullptr, 0, dtype, stream, q_rows);
}

extern "C" int rnamsm_col_attn_fused_prescaled(const float* q, const float* k, const float* v, int64_t ld, float* ctx, int64_t ldc,
                                               int R, int C, int H, int head_dim, int q_rows, void* stream) {
    RNAMSM_CHECK_ARG(q_rows >= 1 && q_rows <= R, "col_attn_prescaled: q_rows must be in [1, R] (got %d, R=%d)", q_rows, R);
    return col_attn_launch(q, k, v, ld, ctx, ldc, R, C, H, head_dim, nullptr, nullptr, nullptr, 0, RNAMSM_F32, stream, q_rows, 1, 0, 0, true);
}

namespace rnamsm {
// ---- K7 for shallow alignments (R <= 16): ONE WAVE per (column, head) problem, no LDS, no barrier.
// The kernels above give a (column, head) a block of 128 query rows: at R = 8 such a block is 6 % full and the launch is
// bound by block turnover (batches of small alignments spend 5 % of their time here: 24576 blocks of ~4 us at B = 32, R = 8,
// C = 64).  With R <= 16 the whole problem is one 16x16 tile of the exact-fp32 v_mfma_f32_16x16x4_f32 (lane = (fr, fq),
// fr = lane & 15 the tile row / column it feeds, fq = lane >> 4 its k-slot):
//   S^T = K Q^T   A = K (key fr), B = Q^T (query fr).  The 64 head dims are contracted in the order the loads deliver them: lane
//                 (fr, fq) fetches the four float4 at d = 16 g + 4 fq of its row, and step (g, e) pairs k-slot fq with
//                 d = 16 g + 4 fq + e in BOTH operands (any order is a valid contraction order as long as A and B agree).
//                 The accumulator holds S^T[key 4 fq + t][query fr], t = 0..3: the softmax over keys is 4 registers + two
//                 lane exchanges (xor 16, xor 32).
//   O^T = V^T P^T at step s the k-slot fq stands for key 4 fq + s, so the lane's OWN probability register s is its B operand
//                 (no data movement, the trick of the big kernels).  A = V^T with the head dims dealt over four tiles as
//                 d = 4 row + tile: lane (fr, fq) fetches ONE float4 V[key 4 fq + s][4 fr .. 4 fr + 3] per step and feeds
//                 element tile to tile `tile`; register t of the four accumulators is then d = 16 fq + 4 t + tile, i.e. four
//                 consecutive head dims of query fr: float4 stores.
// 12 sixteen-byte loads per lane straight from global memory (L1/L2-served), 32 MFMAs of 32 cycles, ~45 registers: 8 waves
// per SIMD hide the latency.  Same arithmetic as the big kernels up to the summation order.  q_rows < R: only those query
// rows are stored (rnamsm_col_attn_fused_queries), bit-identical to the full launch's.
template <bool MASKED>
__global__ __launch_bounds__(256) void col_attn_small_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                             const float* __restrict__ v, int64_t ld, float* __restrict__ ctx,
                                                             int64_t ldc, int R, int C, int H,
                                                             const uint8_t* __restrict__ pad_mask, int q_rows,
                                                             int64_t qkv_bstride, int64_t ctx_bstride,
                                                             const PackedMsa* __restrict__ pk, int log2_domain) {
    typedef float f32x4s __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int prob = blockIdx.x * 4 + wave;
    if (pk) {                                                    // token-packed batch: the shallow alignments only
        const PackedMsa& m = pk[blockIdx.y];
        if (m.R > 16) return;
        R = m.R; C = m.C; q_rows = m.R;
        q += m.tok0 * ld;
        k += m.tok0 * ld;
        v += m.tok0 * ld;
        ctx += m.tok0 * ldc;
    } else {
        q += blockIdx.y * qkv_bstride;
        k += blockIdx.y * qkv_bstride;
        v += blockIdx.y * qkv_bstride;
        ctx += blockIdx.y * ctx_bstride;
        if (MASKED) pad_mask += blockIdx.y * ((int64_t)R * C);
    }
    if (prob >= C * H) return;                                   // wave-uniform
    const int c = prob / H, h = prob % H;
    const int fr = lane & 15, fq = lane >> 4;
    const int rr = min(fr, R - 1);                               // rows past R are clamped: masked as keys, not stored as queries
    const int64_t ro = ((int64_t)rr * C + c) * ld + h * CA_HD + 4 * fq;
    f32x4s k4[4], q4[4], v4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        k4[g] = *reinterpret_cast<const f32x4s*>(k + ro + 16 * g);
        q4[g] = *reinterpret_cast<const f32x4s*>(q + ro + 16 * g);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
        v4[s] = *reinterpret_cast<const f32x4s*>(v + ((int64_t)min(4 * fq + s, R - 1) * C + c) * ld + h * CA_HD + 4 * fr);
    f32x4s s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};   // two chains: the 16x16x4 MFMA's dependent latency exceeds its issue time
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(k4[g][0], q4[g][0], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(k4[g][1], q4[g][1], s1, 0, 0, 0);
        s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(k4[g][2], q4[g][2], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(k4[g][3], q4[g][3], s1, 0, 0, 0);
    }
    float p[4];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int j = 4 * fq + t;
        float sc = s0[t] + s1[t];
        if (MASKED && j < R && pad_mask[(int64_t)j * C + c]) sc = -10000.f;         // modules.py:911-915
        p[t] = j < R ? sc : -INFINITY;
        mx = fmaxf(mx, p[t]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        p[t] = log2_domain ? __builtin_amdgcn_exp2f(p[t] - mx) : __expf(p[t] - mx);      // keys past R: exp(-inf) = 0; log2_domain: q carries log2(e)
        l += p[t];
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.f / l;
    f32x4s o[4];
#pragma unroll
    for (int tile = 0; tile < 4; ++tile) {
        o[tile] = f32x4s{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) o[tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(v4[s][tile], p[s], o[tile], 0, 0, 0);
    }
    if (fr < q_rows) {                                           // register t of the four tiles = head dims 16 fq + 4 t + 0..3 of query fr
        float* orow = ctx + ((int64_t)fr * C + c) * ldc + h * CA_HD + 16 * fq;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            *reinterpret_cast<f32x4s*>(orow + 4 * t) = f32x4s{o[0][t], o[1][t], o[2][t], o[3][t]} * inv;
    }
}
}  // namespace rnamsm

static int col_attn_launch(const float* q, const float* k, const float* v, int64_t ld, float* ctx, int64_t ldc, int R, int C,
                           int H, int head_dim, const uint8_t* pad_mask, uint16_t* ctx_hi, uint16_t* ctx_lo, int plane_fmt,
                           int dtype, void* stream, int q_rows, int batch, int64_t qkv_bstride, int64_t ctx_bstride, bool prescaled) {
    if (dtype != RNAMSM_F32) return fail(RNAMSM_ERR_UNSUPPORTED, "col_attn: only RNAMSM_F32 is implemented");
    RNAMSM_CHECK_ARG(!prescaled || (!pad_mask && !ctx_hi), "col_attn: prescaled q has no masked / plane-output variant");
    RNAMSM_CHECK_ARG(batch == 1 || (!ctx_hi && qkv_bstride % 4 == 0 && ctx_bstride % 4 == 0),
                     "col_attn: a batched launch writes fp32 context");
    RNAMSM_CHECK_ARG(q && k && v && (ctx || ctx_hi), "col_attn: null pointer");
    RNAMSM_CHECK_ARG(head_dim == CA_HD, "col_attn: head_dim must be 64 (got %d)", head_dim);
    RNAMSM_CHECK_ARG(R > 0 && R <= 1024 && C > 0 && H > 0, "col_attn: bad shape R=%d C=%d H=%d", R, C, H);
    RNAMSM_CHECK_ARG(ld >= (int64_t)H * CA_HD && ld % 4 == 0 && ldc >= (int64_t)H * CA_HD && ldc % 4 == 0,
                     "col_attn: ld/ldc must be multiples of 4 and >= H*64");
    RNAMSM_CHECK_ARG(aligned16(q) && aligned16(k) && aligned16(v), "col_attn: 16-byte alignment");
    RNAMSM_CHECK_ARG(ctx_hi ? (!pad_mask && (reinterpret_cast<uintptr_t>(ctx_hi) & 7u) == 0 && (plane_fmt == 0 || plane_fmt == 1))
                            : aligned16(ctx), "col_attn: output alignment / plane format (plane output has no masked variant)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned iblocks = (q_rows + CA_ROWS - 1) / CA_ROWS;
    const unsigned grid = xcd_panel_grid((unsigned)C * H, iblocks);
    KernelTimer timer(TC_COL_ATTN, 4.0 * batch * C * H * (double)q_rows * R * CA_HD, 4.0 * batch * (2.0 * R + 2.0 * q_rows) * C * H * CA_HD, s);
    // shallow alignments: one wave per (column, head), no LDS ("col_small" = 0 keeps the 128-query blocks: A/B)
    if (R <= 16 && !ctx_hi && tuning().col_small != 0) {
        const dim3 sgrid(((unsigned)C * H + 3) / 4, batch);
        if (pad_mask)
            hipLaunchKernelGGL(col_attn_small_kernel<true>, sgrid, dim3(256), 0, s, q, k, v, ld, ctx, ldc, R, C, H, pad_mask, q_rows, qkv_bstride, ctx_bstride, (const PackedMsa*)nullptr, prescaled ? 1 : 0);
        else
            hipLaunchKernelGGL(col_attn_small_kernel<false>, sgrid, dim3(256), 0, s, q, k, v, ld, ctx, ldc, R, C, H, pad_mask, q_rows, qkv_bstride, ctx_bstride, (const PackedMsa*)nullptr, prescaled ? 1 : 0);
        RNAMSM_CHECK_LAUNCH("col_attn_small");
        return RNAMSM_OK;
    }
#define CA_GO2(KERN_, LDS_, M_, OUT_)                                                                               \
    do {                                                                                                            \
        static DeviceOnce cfg_;                                                                                     \
        if (cfg_.pending()) {                                                                                       \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(KERN_<M_, OUT_>),                      \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_);                   \
            if (e != hipSuccess) return fail(RNAMSM_ERR_HIP, "col_attn: hipFuncSetAttribute: %s", hipGetErrorString(e)); \
            cfg_.mark();                                                                                            \
        }                                                                                                           \
        hipLaunchKernelGGL((KERN_<M_, OUT_>), dim3(grid, batch), dim3(CA_THREADS), LDS_, s, q, k, v, ld, ctx, ldc, R, C, H, \
                           pad_mask, ctx_hi, ctx_lo, q_rows, qkv_bstride, ctx_bstride, (const PackedMsa*)nullptr, 0, 0); \
    } while (0)
    // "col_dma": 1 = the LDS-DMA, three-blocks-per-CU variant, 0 = the register-staged kernel, -1 (default) = the former.
    // Measured in one process (tools/col_attn_ab.py, after the key-range masking was confined to the ragged last tile):
    // R=256 C=512 +11 %, R=128 C=256 +5 %, R=1024 +2 %, R=100 C=300 +3 %, R=64 / R=512 C=36 equal; the masked instance
    // +10-20 % at every shape (the register-staged one spills).  Outputs bit-identical.
    const bool use_dma = tuning().col_dma != 0;
#define CA_GO(M_, OUT_)                                                                                             \
    do {                                                                                                            \
        if (use_dma) CA_GO2(col_attn_dma_kernel, CD_LDS_BYTES, M_, OUT_);                                           \
        else CA_GO2(col_attn_kernel, CA_LDS_BYTES, M_, OUT_);                                                       \
    } while (0)
    if (prescaled) {        // log2-domain scores: FAST loop first ("col_fast" = 0: the TRACKED loop only, A/B)
        static DeviceOnce cfgp;
        if (cfgp.pending()) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(col_attn_dma_kernel<false, 0, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, CD_LDS_BYTES);
            if (e != hipSuccess) return fail(RNAMSM_ERR_HIP, "col_attn: hipFuncSetAttribute: %s", hipGetErrorString(e));
            cfgp.mark();
        }
        hipLaunchKernelGGL((col_attn_dma_kernel<false, 0, true>), dim3(grid, batch), dim3(CA_THREADS), CD_LDS_BYTES, s, q, k, v, ld, ctx, ldc,
                           R, C, H, pad_mask, ctx_hi, ctx_lo, q_rows, qkv_bstride, ctx_bstride, (const PackedMsa*)nullptr, 0,
                           tuning().col_fast ? 0 : 1);
        RNAMSM_CHECK_LAUNCH("col_attn (prescaled)");
        return RNAMSM_OK;
    }
    if (pad_mask) CA_GO2(col_attn_dma_kernel, CD_LDS_BYTES, true, 0);      // (the register-staged kernel has no masked instance: it spilled)
    else if (!ctx_hi) CA_GO(false, 0);
    else if (plane_fmt == 0) CA_GO(false, 1);
    else CA_GO(false, 2);
#undef CA_GO2
#undef CA_GO
    RNAMSM_CHECK_LAUNCH("col_attn");
    return RNAMSM_OK;
}

namespace rnamsm {
// K7 for `batch` same-shape MSAs in one launch (rnamsm_forward_batch): MSA b's q / k / v lie b * qkv_bstride elements on,
// its padding mask (if any) b * R * C bytes on
int col_attn_batched(const float* q, const float* k, const float* v, int64_t ld, float* ctx, int64_t ldc, int R, int C, int H,
                     int batch, int64_t qkv_bstride, int64_t ctx_bstride, const uint8_t* pad_mask, void* stream, bool prescaled) {
    return col_attn_launch(q, k, v, ld, ctx, ldc, R, C, H, CA_HD, pad_mask, nullptr, nullptr, 0, RNAMSM_F32, stream, R, batch,
                           qkv_bstride, ctx_bstride, prescaled);
}

// K7 of a token-packed batch (rnamsm_forward_packed): the LDS-DMA kernel with gridDim.y = alignment, gridDim.x sized for the
// largest alignment; with "col_small" on, the alignments of R <= 16 go to a second launch of the one-wave-per-problem kernel
int col_attn_packed(const float* q, const float* k, const float* v, int64_t ld, float* ctx, int64_t ldc, int H, const PackedMsa* pk,
                    const PackedMsa* host, int B, void* stream, bool prescaled) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool split_shallow = tuning().col_small != 0;
    unsigned grid = 0, sgrid = 0;
    double flops = 0.0, bytes = 0.0;
    for (int b = 0; b < B; ++b) {
        const PackedMsa& m = host[b];
        if (split_shallow && m.R <= 16) {
            const unsigned g = ((unsigned)m.C * H + 3) / 4;
            sgrid = g > sgrid ? g : sgrid;
        } else {
            const unsigned g = xcd_panel_grid((unsigned)m.C * H, (m.R + CA_ROWS - 1) / CA_ROWS);
            grid = g > grid ? g : grid;
        }
        flops += 4.0 * m.C * H * (double)m.R * m.R * CA_HD;
        bytes += 4.0 * 4.0 * m.R * m.C * H * CA_HD;
    }
    KernelTimer timer(TC_COL_ATTN, flops, bytes, s);
    if (grid) {
        static DeviceOnce cfg;
        if (cfg.pending()) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(col_attn_dma_kernel<false, 0>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, CD_LDS_BYTES);
            if (e != hipSuccess) return fail(RNAMSM_ERR_HIP, "col_attn (packed): hipFuncSetAttribute: %s", hipGetErrorString(e));
            cfg.mark();
        }
        if (prescaled) {
            static DeviceOnce cfgp;
            if (cfgp.pending()) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(col_attn_dma_kernel<false, 0, true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, CD_LDS_BYTES);
                if (e != hipSuccess) return fail(RNAMSM_ERR_HIP, "col_attn (packed): hipFuncSetAttribute: %s", hipGetErrorString(e));
                cfgp.mark();
            }
            hipLaunchKernelGGL((col_attn_dma_kernel<false, 0, true>), dim3(grid, B), dim3(CA_THREADS), CD_LDS_BYTES, s, q, k, v, ld, ctx, ldc, 0,
                               0, H, (const uint8_t*)nullptr, (uint16_t*)nullptr, (uint16_t*)nullptr, 0, (int64_t)0, (int64_t)0, pk,
                               split_shallow ? 1 : 0, tuning().col_fast ? 0 : 1);
        } else
        hipLaunchKernelGGL((col_attn_dma_kernel<false, 0>), dim3(grid, B), dim3(CA_THREADS), CD_LDS_BYTES, s, q, k, v, ld, ctx, ldc, 0, 0, H,
                           (const uint8_t*)nullptr, (uint16_t*)nullptr, (uint16_t*)nullptr, 0, (int64_t)0, (int64_t)0, pk,
                           split_shallow ? 1 : 0, 0);
        RNAMSM_CHECK_LAUNCH("col_attn (packed)");
    }
    if (sgrid) {
        hipLaunchKernelGGL(col_attn_small_kernel<false>, dim3(sgrid, B), dim3(256), 0, s, q, k, v, ld, ctx, ldc, 0, 0, H,
                           (const uint8_t*)nullptr, 0, (int64_t)0, (int64_t)0, pk, prescaled ? 1 : 0);
        RNAMSM_CHECK_LAUNCH("col_attn_small (packed)");
    }
    return RNAMSM_OK;
}
}  // namespace rnamsm
