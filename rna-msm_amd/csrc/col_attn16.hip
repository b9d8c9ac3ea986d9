// K7': fused column attention on the 16-bit matrix cores (16-bit modes of ColumnSelfAttention, modules.py:875-924).
//
// Same algorithm as col_attn.hip (S^T = K Q^T with the query on the lane, softmax in fp32 registers, exp(S^T) used in place as
// the B operand of O^T += V^T P^T), with q/k/v arriving as 16-bit hi(+lo) planes [T, 3D] from the QKV GEMM epilogue:
//   K chunk [JC keys][64 d]  "k" tile of tile16.h, DMA-staged, read with ds_read_b128 (A operand of S^T, k = d);
//   V chunk [JC keys][64 d]  "t" tile, staged exactly as it lies in memory and read with ds_read_b64_tr_b16 as the
//                            A operand V^T[d][key] of O^T -- the hardware transposed read replaces a transposing copy;
//   P                        the fp32 accumulator registers 8s..8s+7 of S^T, converted pairwise to halves, ARE the B
//                            fragment of k-step s; its k order is key 16s + 8(e>>2) + 4*half + (e&3) for element e, so
//                            the V fragment is built from the two 4-key blocks 16s + 4*half and 16s + 8 + 4*half.
// scale (dh^-0.5) multiplies the fp32 scores, not q: an unscaled q keeps its fp16 lo plane clear of subnormals.
// SPLIT 1: bf16 operands, one MFMA per product.  SPLIT 3: hi/lo pairs, 3 MFMAs (P = hi + lo with |lo| <= 2^-11 P).
//
// What bounds it (round 4).  Per 32-key tile and 32-query block a wave issues (4 + 4) * SPLIT MFMAs of 32 cycles -- 256 cycles
// of matrix pipe in plain bf16 -- against the softmax of 16 scores per lane on the VALU, whose ISSUE cycles (4 per plain
// instruction, 8 per v_exp_f32; MI355X_MICROARCH.md cycle constants) are the real limit at head_dim 64: the round-3 kernel
// spent ~150 instructions (~690 issue cycles) per tile.  Two loops now:
//   FAST (bf16 operands, no padding mask): NO running maximum.  The reference m of a query is fixed once, from its scores
//     against the first 32 keys (a pre-pass of four MFMAs), and p = v_exp_f32(fma(s, scale*log2e, -m)) -- one fma and the raw
//     transcendental per score, 8 v_cvt_pk for the operand, nothing else: no max chain, no exchange between the lane halves, no
//     rescale of O, no branch.  A bf16 P (like the fp32 sums) keeps its relative precision at ANY magnitude, so an m that is
//     stale by up to 2^96 costs no accuracy; the key that set m contributes P = 1, so the sum cannot underflow.  Only a score
//     more than 96 log2-units (66 nats) above every score of the first tile overflows: then the row sum comes out >= 2^96 / non-
//     finite, the block votes, and ALL its waves redo the column with the TRACKED loop.  tests/test_gpu_attn16.py forces that.
//   TRACKED (fp16 operands -- P must stay below 2^16 --, padding masks, and the fallback): the classic online softmax in its
//     cheapest form: v_maximum3_f32 chain on the raw scores (scale > 0), halves exchanged by v_permlane32_swap, log2 domain.
//   PRESCALED q (rnamsm_col_attn16_prescaled, what the forward runs in the bf16 modes): the QKV epilogue multiplies the q columns by
//     scale * log2(e) before it rounds them, so the scores arrive in the log2 domain and p = v_exp_f32(s): no fma, no reference, no
//     pre-pass -- 16 exp, 16 adds and 8 v_cvt_pk per 32-key tile and query block.
//   QB = 2 (plain bf16, R >= 256): a wave owns two 32-query blocks, so every K / V fragment read from LDS feeds two blocks (half
//     the LDS reads, barriers and DMA requests per flop) and one block's softmax can issue under the other's MFMAs.
//   (Row sums on the matrix pipe -- L^T += 1 P^T, one all-ones MFMA per k-step instead of 16 v_add_f32 per tile -- were built and
//     measured: +8 MFMAs on the pipe that is already the busier one (60 %) and 32 more registers; 4.25 against 3.83 ms at
//     M = L = 1024 once the addressing was trimmed.  Dropped; EXPERIMENTS.md R4.1.)
// K / V chunks ride a three-deep LDS ring two chunks ahead of the compute (counted vmcnt).  The V^T reads are inline asm
// (tile16.h: tr16_issue): behind the builtin hipcc drains every LDS-DMA in flight, which had made the round-3 ring synchronous.
// The context leaves through LDS as whole 128-byte rows (16-byte stores, 8 lanes per row) instead of 8-byte pieces at a row stride.
#include <type_traits>

#include "tile16.h"

namespace rnamsm {

// Throw-away what-if builds (wrong results, timing only; tools/whatif_col_attn16.sh, docs/history/profiles_r05/r05_whatif_col_attn16.log): 1 no v_exp,
// 2 no row sums, 4 no LDS-DMA inside the loop, 8 no P V MFMAs, 16 no S MFMAs, 32 never fall back, 64 one chunk per block.  0 in the
// shipped library (every use folds away).
#ifndef C16_WHATIF
#define C16_WHATIF 0
#endif
constexpr int C16_THREADS = 256;
constexpr int C16_NST = 3;                       // LDS ring depth (chunks)
constexpr float C16_LOG2E = 1.4426950408889634f;
constexpr float C16_OVERFLOW = 7.9228163e28f;    // 2^96: a FAST row sum at or above it (or not finite) sends the block to the TRACKED loop

// maximum over the two lanes (l, l ^ 32) that share a query.  v_permlane32_swap as asm: this hipcc folds element 1 of
// __builtin_amdgcn_permlane32_swap's result to element 0 (the IR keeps `extractvalue 0` only), which silently turned the
// exchange into a copy; the s_nop covers the VALU-write -> permlane-read hazard hipcc does not pad inside an asm body.
__device__ __forceinline__ float half_pair_max(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));      // a = {lo, lo}, b = {hi, hi}
    return __builtin_elementwise_maximum(a, b);
}
// v_maximum3_f32 chain (NaN-propagating; no canonicalising v_max x, x in front of MFMA outputs as fmaxf gets)
__device__ __forceinline__ float max16(const f32x16& s) {
    float mx = s[0];
#pragma unroll
    for (int t = 1; t < 16; ++t) mx = __builtin_elementwise_maximum(mx, s[t]);
    return mx;
}

// acc += the eight bf16 values of a P fragment, in fp32: four v_dot2c_f32_bf16 against (1, 1) in ONE asm statement (hipcc cannot
// select the builtin on gfx950, and pads every asm boundary).  Plain bf16 sums the ROUNDED P with it -- the values that enter
// O^T += V^T P^T -- so that O / l stays exact where one key takes all the weight (a P of 2^k (1 + eps) rounds by up to 2^-9; with
// the running maximum gone P is no longer 1.0 there, and a sum of the unrounded values showed as 3e-4 relative in
// test_col_attention_16bit_selects_the_right_value_rows) -- in half the instructions of sixteen v_add_f32.
typedef unsigned u32x4c __attribute__((ext_vector_type(4)));
template <class V8T>
__device__ __forceinline__ float sum8_bf16(const V8T& p, float acc) {
    const u32x4c u = __builtin_bit_cast(u32x4c, p);
    const unsigned ones = 0x3f803f80u;
    asm("v_dot2c_f32_bf16 %0, %1, %5\n\tv_dot2c_f32_bf16 %0, %2, %5\n\tv_dot2c_f32_bf16 %0, %3, %5\n\tv_dot2c_f32_bf16 %0, %4, %5"
        : "+v"(acc)
        : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]), "v"(ones));
    return acc;
}

template <int SPLIT, int QB>
struct C16Cfg {
    static constexpr int NPL = SPLIT == 3 ? 2 : 1;
    static constexpr int JC = SPLIT == 3 ? 32 : 64;                 // keys per chunk
    static constexpr int TILE = JC * T16_ROWB;                      // bytes per plane tile
    static constexpr int BUF = 2 * NPL * TILE;                      // K planes then V planes
    static constexpr int ROWS = 128 * QB;                           // query rows per block
    static constexpr int EPI = 4 * 9216;                            // epilogue staging: 32 rows x 272 B (fp32) / 2 planes x 32 x 144 B per wave
    static constexpr int RING = C16_NST * BUF > EPI ? C16_NST * BUF : EPI;
    static constexpr int LDS = RING + 16;                           // + the four waves' overflow flags
    static constexpr int NDMA = (JC / 32) * NPL * 2;                // LDS-DMA requests per wave per chunk
};

// MASKED: f2 padding mask (scores of padded keys of this column become -10000 after scaling, modules.py:911-915); the
// un-masked instances carry no mask code.  FAST: see the header (requires bf16 operands and no mask).
// PRE (FAST only): q arrives PRESCALED by scale * log2(e) (the QKV GEMM's epilogue multiplies the q columns before it rounds them:
// no second rounding) -- the scores are log2-domain already and p = v_exp_f32(s): not even the fma, no reference, no pre-pass.  A
// row sum outside [2^-96, 2^96] (or not finite) sends the block to the TRACKED loop as before.
template <int SPLIT, int FMT, int OUT, int QB, bool MASKED, bool FAST, bool PRE>
__global__ __launch_bounds__(C16_THREADS, QB == 1 ? 3 : 2) void col_attn16_kernel(
    const uint16_t* __restrict__ qhi, const uint16_t* __restrict__ qlo, const uint16_t* __restrict__ khi,
    const uint16_t* __restrict__ klo, const uint16_t* __restrict__ vhi, const uint16_t* __restrict__ vlo, int64_t ld,
    float* __restrict__ ctx, int64_t ldc, int R, int C, int H, uint16_t* __restrict__ ctx_hi, uint16_t* __restrict__ ctx_lo,
    float scale, const uint8_t* __restrict__ pad_mask, int64_t qkv_bstride, int64_t ctx_bstride, int64_t mask_bstride,
    int force_tracked) {
    static_assert(!FAST || (FMT == 0 && !MASKED), "the FAST loop needs bf16's exponent range and no -10000 scores");
    static_assert(!PRE || FAST, "prescaled q is the FAST loop's input");
    using Cfg = C16Cfg<SPLIT, QB>;
    constexpr int NPL = Cfg::NPL, JC = Cfg::JC, TILE = Cfg::TILE, BUF = Cfg::BUF;
    // TRACKED loop: P is kept as 2^12 exp(s - m) so that the fp16 lo plane of every P that matters stays normal
    constexpr float SHIFT = FMT == 1 ? 12.f : 0.f;
    // batched launch (rnamsm_forward_batch, 16-bit modes): MSA blockIdx.y
    qhi += blockIdx.y * qkv_bstride; khi += blockIdx.y * qkv_bstride; vhi += blockIdx.y * qkv_bstride;
    if (qlo) { qlo += blockIdx.y * qkv_bstride; klo += blockIdx.y * qkv_bstride; vlo += blockIdx.y * qkv_bstride; }
    if (ctx) ctx += blockIdx.y * ctx_bstride;
    if (ctx_hi) ctx_hi += blockIdx.y * ctx_bstride;
    if (ctx_lo) ctx_lo += blockIdx.y * ctx_bstride;
    if (MASKED) pad_mask += blockIdx.y * mask_bstride;
    typedef typename Half16<FMT>::T Hh;
    typedef typename Half16<FMT>::V8 V8;
    extern __shared__ __attribute__((aligned(16))) char smem_b[];

    const unsigned iblocks = (R + Cfg::ROWS - 1) / Cfg::ROWS;
    unsigned prob, ib;
    if (!xcd_panel_map(blockIdx.x, (unsigned)C * H, iblocks, prob, ib)) return;
    const int c = prob / H, h = prob % H;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int irow0 = ib * Cfg::ROWS + wave * 32 * QB;         // this wave's queries: irow0 + 32 qb + li
    const bool active = irow0 < R;                             // wave-uniform
    const int64_t col_off = (int64_t)c * ld + h * 64;          // + r*C*ld selects the alignment row

    const uint16_t* qpl[2] = {qhi, qlo};
    const uint16_t* kpl[2] = {khi, klo};
    const uint16_t* vpl[2] = {vhi, vlo};

    // Q fragments (B operand of S^T = K Q^T): lane (query i, half) holds q[i][16kk + 8*half + 0..7]; rows past R are clamped
    // (computed, never stored)
    V8 qf[QB][4][NPL];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qi = min(irow0 + 32 * qb + li, R - 1);
        const int64_t qo = (int64_t)qi * C * ld + col_off + 8 * lh;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int p = 0; p < NPL; ++p) qf[qb][kk][p] = *reinterpret_cast<const V8*>(qpl[p] + qo + 16 * kk);
    }

    // DMA map: a plane tile is JC / 8 groups of 8 key rows; wave w moves groups w and w+4.  Keys past R are clamped to the
    // last key: their scores are masked to -inf and their V values only meet P = 0.
    const int drow = lane >> 3;
    const int ck = dma_chunk_k(lane, wave), ct = dma_chunk_t(lane);
    // a lane's source offsets within chunk 0 (halves), a chunk on = JC alignment rows: one 64-bit add per request in the loop;
    // only the last chunk of a ragged column clamps its rows (a multiply per request, once per block)
    int64_t krow[JC / 32];
#pragma unroll
    for (int j = 0; j < JC / 32; ++j) krow[j] = (int64_t)(8 * (wave + 4 * j) + drow) * C * ld + col_off;
    const int64_t chunk_stride = (int64_t)JC * C * ld;
    auto issue = [&](int ch, int buf) __attribute__((always_inline)) {
        char* base = smem_b + buf * BUF;
        const bool ragged = (ch + 1) * JC > R;                 // block-uniform
#pragma unroll
        for (int j = 0; j < JC / 32; ++j) {
            const int row = 8 * (wave + 4 * j) + drow;
            const int64_t ko = ragged ? (int64_t)min(ch * JC + row, R - 1) * C * ld + col_off : krow[j] + ch * chunk_stride;
            const int loff = (8 * (wave + 4 * j)) * T16_ROWB;
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                dma16(kpl[p] + ko + ck * 8, base + p * TILE + loff);
                dma16(vpl[p] + ko + ct * 8, base + (NPL + p) * TILE + loff);
            }
        }
    };

    f32x16 o0[QB], o1[QB];               // O^T tiles: head dims [0,32) and [32,64) x 32 query rows
    float m_run[QB], l_run[QB];          // log2 units (score * scale * log2e); FAST: m_run is the fixed reference
    const float c2 = PRE ? 1.f : scale * C16_LOG2E;          // raw score -> log2 units (PRE: q carries the factor already)
    const float zmask = -10000.f / scale;                    // a masked score in raw units (-10000 after scaling)
    const int tq = (lane & 15) >> 2, tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);    // transposed-read geometry
    // byte offset of this lane's transposed-read address inside a V tile for rows 4 lh + tq (+ 16 ks, + 8, + 32 jt: multiples of 8
    // rows leave the row swizzle alone) and d tile dt
    uint32_t vbase[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        const int row = 4 * lh + tq, col = dt * 32 + tcol;
        vbase[dt] = row * T16_ROWB + (((col >> 3) ^ swz_t(row)) << 4) + (col & 7) * 2;
    }

    // one 32-key tile.  RAG (compile-time): the ragged last tile, whose keys >= R are masked out.  TRK: TRACKED arithmetic.
    auto tile = [&](const char* Kc, const char* Vc, int jt, int jbase, auto rag_tag, auto trk_tag) __attribute__((always_inline)) {
        constexpr bool RAG = decltype(rag_tag)::value, TRK = decltype(trk_tag)::value;
        // the masked three-product instances are one register over their budget: their per-lane K addresses are re-derived per
        // tile (a few VALU ops) instead of being held from kernel entry
        int lane_t = lane;
        if (MASKED && SPLIT == 3) asm volatile("" : "+v"(lane_t));
        const int li = lane_t & 31, lh = lane_t >> 5;
        V8 kf[4][NPL];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int p = 0; p < NPL; ++p) kf[kk][p] = frag_k<FMT>(Kc + p * TILE, jt * 32 + li, kk, lh);
        // V^T fragments of this tile, requested now (asm: tile16.h) and waited for after the softmax.  A lane's address =
        // (its row / swizzled column within an 8-row group: vbase[d tile], fixed for the kernel) + slot + a compile-time offset
        // (tile, k-step, second 4-key block, plane): one v_add per d tile and chunk instead of one per request
        TrPieces vp[2][2][NPL];          // [d tile][k step][plane]
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const uint32_t va = lds_addr_of(Vc) + vbase[dt];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int p = 0; p < NPL; ++p) {
                    if (jt == 0) tr16_issue_at<0>(va, ks * 16 * T16_ROWB + p * TILE, vp[dt][ks][p]);
                    else tr16_issue_at<32 * T16_ROWB>(va, ks * 16 * T16_ROWB + p * TILE, vp[dt][ks][p]);
                }
        }
        // ---- S^T = K Q^T: 32 keys x 32 queries per block, k = 64 head dims
        f32x16 s[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
            for (int t = 0; t < 16; ++t) s[qb][t] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (C16_WHATIF & 16) s[qb][kk] += (float)kf[kk][0][0] + (float)qf[qb][kk][0][0];
                else s[qb] = mma16<SPLIT, FMT>(kf[kk], qf[qb][kk], s[qb]);
            }
        }
        unsigned mbits = 0;              // MASKED: bit t = key of accumulator register t is padded in this column
        if (MASKED) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int j = jbase + (t & 3) + 8 * (t >> 2) + 4 * lh;
                if (j < R && pad_mask[(int64_t)j * C + c]) mbits |= 1u << t;
                if ((t & 3) == 3) asm volatile("" : "+v"(mbits));      // four byte loads in flight, not sixteen (registers)
            }
        }
        V8 pf[QB][2][NPL];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            if (RAG) {
                const int limit = R - jbase;
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    if ((t & 3) + 8 * (t >> 2) + 4 * lh >= limit) s[qb][t] = -INFINITY;
            }
            if (MASKED) {
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    if ((mbits >> t) & 1u) s[qb][t] = zmask;
            }
            float nm;
            if (TRK) {
                // online softmax: tile maximum on the raw scores (scale > 0), both halves of the query; finite: key jbase is valid
                const float m_new = __builtin_elementwise_maximum(m_run[qb], half_pair_max(max16(s[qb])) * c2);
                const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);     // 0 on the first tile
                m_run[qb] = m_new;
#pragma unroll
                for (int t = 0; t < 16; ++t) { o0[qb][t] *= alpha; o1[qb][t] *= alpha; }
                l_run[qb] *= alpha;
                nm = SHIFT - m_new;
            } else {
                nm = -m_run[qb];
            }
            float psum = 0.f;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                if (!(C16_WHATIF & 1))
                    s[qb][t] = (PRE && !TRK) ? __builtin_amdgcn_exp2f(s[qb][t]) : __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][t], c2, nm));
                if (SPLIT == 3) psum += s[qb][t];           // hi + lo carries P to 2^-17 / 2^-22: the fp32 values serve
            }
            if (SPLIT == 3) l_run[qb] += psum;
            // ---- P fragments: registers 8ks..8ks+7 -> halves (hi, and lo = P - hi)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float pv = SPLIT == 3 ? pinned(s[qb][8 * ks + e]) : s[qb][8 * ks + e];
                    const Hh hi = (Hh)pv;
                    pf[qb][ks][0][e] = hi;
                    if (SPLIT == 3) pf[qb][ks][NPL - 1][e] = (Hh)(pv - (float)hi);
                }
            if (SPLIT == 1 && !(C16_WHATIF & 2)) l_run[qb] = sum8_bf16(pf[qb][1][0], sum8_bf16(pf[qb][0][0], l_run[qb]));     // the rounded P (see sum8_bf16)
        }
        // ---- O^T += V^T P^T
#pragma unroll
        for (int p = 0; p < NPL; ++p) tr16_wait4(vp[0][0][p], vp[0][1][p], vp[1][0][p], vp[1][1][p]);
        V8 vf[2][2][NPL];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int p = 0; p < NPL; ++p) vf[dt][ks][p] = tr16_frag<FMT>(vp[dt][ks][p]);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (C16_WHATIF & 8) {
                    asm volatile("" :: "v"(pf[qb][ks][0]), "v"(vf[0][ks][0]), "v"(vf[1][ks][0]));
                } else {
                    o0[qb] = mma16<SPLIT, FMT>(vf[0][ks], pf[qb][ks], o0[qb]);
                    o1[qb] = mma16<SPLIT, FMT>(vf[1][ks], pf[qb][ks], o1[qb]);
                }
            }
    };
    typedef std::integral_constant<bool, false> no_t;
    typedef std::integral_constant<bool, true> yes_t;

    // FAST pre-pass: the reference of every query = its largest score against the first min(32, R) keys
    auto reference_from_first_tile = [&](const char* Kc) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            f32x16 s;
#pragma unroll
            for (int t = 0; t < 16; ++t) s[t] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                V8 kf[NPL];
#pragma unroll
                for (int p = 0; p < NPL; ++p) kf[p] = frag_k<FMT>(Kc + p * TILE, li, kk, lh);
                s = mma16<SPLIT, FMT>(kf, qf[qb][kk], s);
            }
            if (R < 32) {
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    if ((t & 3) + 8 * (t >> 2) + 4 * lh >= R) s[t] = -INFINITY;
            }
            m_run[qb] = half_pair_max(max16(s)) * c2;
        }
    };

    // The whole key loop: three-deep ring, two chunks ahead.  The request of a chunk past the end is clamped to the last chunk
    // (a redundant reload into a free slot, never read) so that every wave has exactly NDMA requests per iteration in flight
    // behind the awaited one.
    auto run = [&](auto trk_tag) __attribute__((always_inline)) {
        constexpr bool TRK = decltype(trk_tag)::value;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
            for (int t = 0; t < 16; ++t) { o0[qb][t] = 0.f; o1[qb][t] = 0.f; }
            m_run[qb] = -INFINITY;
            l_run[qb] = 0.f;
        }
        const int nch = (C16_WHATIF & 64) ? 1 : (R + JC - 1) / JC, nfull = (C16_WHATIF & 64) ? 1 : R / JC;
        issue(0, 0);
        issue(min(1, nch - 1), 1);
        if (!TRK && !PRE) {
            wait_dma_then_barrier<Cfg::NDMA>();      // chunk 0 has landed
            reference_from_first_tile(smem_b);
        }
        // Full chunks: a straight-line body.  (Any branch around the tiles -- an `if (active)`, a ragged variant -- makes hipcc
        // merge the 512-bit accumulator tuples of its paths through copies: 48 v_mov per iteration.  A wave whose queries all
        // lie past R therefore computes on clamped rows and stores nothing.)
        int slot = 0;                        // ring slot of chunk ch
        for (int ch = 0; ch < nfull; ++ch) {
            if (C16_WHATIF & 4) wait_dma_then_barrier<0>();
            else wait_dma_then_barrier<Cfg::NDMA>();      // chunk ch has landed (every wave's share); everyone is done with chunk ch - 1
            if (!(C16_WHATIF & 4)) issue(min(ch + 2, nch - 1), slot == 0 ? 2 : slot - 1);
            const char* Kc = smem_b + slot * BUF;
            const char* Vc = Kc + NPL * TILE;
            tile(Kc, Vc, 0, ch * JC, no_t(), trk_tag);
            if (JC == 64) {
                __builtin_amdgcn_sched_barrier(0);
                tile(Kc, Vc, 1, ch * JC + 32, no_t(), trk_tag);
            }
            slot = slot == 2 ? 0 : slot + 1;
        }
        if (nfull < nch) {                   // the partial last chunk: 1 .. JC - 1 keys
            wait_dma_then_barrier<Cfg::NDMA>();
            issue(nch - 1, slot == 0 ? 2 : slot - 1);
            const char* Kc = smem_b + slot * BUF;
            const char* Vc = Kc + NPL * TILE;
            const int jbase = nfull * JC, rem = R - jbase;
            if (JC == 64 && rem >= 32) {
                tile(Kc, Vc, 0, jbase, no_t(), trk_tag);
                if (rem > 32) tile(Kc, Vc, 1, jbase + 32, yes_t(), trk_tag);
            } else {
                tile(Kc, Vc, 0, jbase, yes_t(), trk_tag);
            }
        }
        wait_dma_then_barrier<0>();          // the redundant reloads have landed, every wave is done with the ring
    };
    auto row_sum = [&](int qb) -> float { return l_run[qb] + __shfl_xor(l_run[qb], 32, 64); };

    // (the context store is a lambda called on each path's own exit: merging the two loops' accumulators at a common epilogue
    // would cost a second copy of all of them in registers)
    auto store_context = [&]() __attribute__((always_inline)) {
        if (!active) return;
        // lane geometry re-derived from an opaque copy: values computed at kernel entry and used only here would otherwise be
        // held (or spilled) across the whole key loop
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int li_e = lane_e & 31, lh_e = lane_e >> 5;
        // wave-private staging: [32 queries][64 d] as 16-bit rows of 144 B (hi, then lo) or fp32 rows of 272 B
        char* stg = smem_b + wave * 9216;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const float inv = 1.f / row_sum(qb);
            if (qb) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the previous block's staging has been read
#pragma unroll
            for (int g = 0; g < 4; ++g) {      // registers 4g..4g+3 are head dims 8g + 4*half + {0..3}
                const f32x4 a = f32x4{o0[qb][4 * g] * inv, o0[qb][4 * g + 1] * inv, o0[qb][4 * g + 2] * inv, o0[qb][4 * g + 3] * inv};
                const f32x4 b = f32x4{o1[qb][4 * g] * inv, o1[qb][4 * g + 1] * inv, o1[qb][4 * g + 2] * inv, o1[qb][4 * g + 3] * inv};
                const int d = 8 * g + 4 * lh_e;
                if (OUT == 0) {
                    *reinterpret_cast<f32x4*>(stg + li_e * 272 + d * 4) = a;
                    *reinterpret_cast<f32x4*>(stg + li_e * 272 + (32 + d) * 4) = b;
                } else {
                    typedef typename Half16<(OUT > 0 ? OUT - 1 : 0)>::T Ho;
                    typedef Ho H4 __attribute__((ext_vector_type(4)));
                    H4 ah, al, bh, bl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float av = pinned(a[e]), bv = pinned(b[e]);
                        ah[e] = (Ho)av; al[e] = (Ho)(av - (float)ah[e]);
                        bh[e] = (Ho)bv; bl[e] = (Ho)(bv - (float)bh[e]);
                    }
                    *reinterpret_cast<H4*>(stg + li_e * 144 + d * 2) = ah;
                    *reinterpret_cast<H4*>(stg + li_e * 144 + (32 + d) * 2) = bh;
                    if (SPLIT == 3) {
                        *reinterpret_cast<H4*>(stg + 4608 + li_e * 144 + d * 2) = al;
                        *reinterpret_cast<H4*>(stg + 4608 + li_e * 144 + (32 + d) * 2) = bl;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // wave-private: the wave's own writes are visible to its reads
            const int i0 = irow0 + 32 * qb;
            if (OUT == 0) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {       // 4 rows x 256 B per instruction
                    const int row = 4 * it + (lane_e >> 4), chunk = lane_e & 15;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * 272 + chunk * 16);
                    if (i0 + row < R) *reinterpret_cast<f32x4*>(ctx + ((int64_t)(i0 + row) * C + c) * ldc + h * 64 + chunk * 4) = v;
                }
            } else {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
                for (int it = 0; it < 4; ++it) {       // 8 rows x 128 B per instruction
                    const int row = 8 * it + (lane_e >> 3), chunk = lane_e & 7;
                    const int64_t ooff = ((int64_t)(i0 + row) * C + c) * ldc + h * 64 + chunk * 8;
                    const u32x4 vh = *reinterpret_cast<const u32x4*>(stg + row * 144 + chunk * 16);
                    if (i0 + row < R) *reinterpret_cast<u32x4*>(ctx_hi + ooff) = vh;
                    if (SPLIT == 3) {
                        const u32x4 vl = *reinterpret_cast<const u32x4*>(stg + 4608 + row * 144 + chunk * 16);
                        if (i0 + row < R && ctx_lo) *reinterpret_cast<u32x4*>(ctx_lo + ooff) = vl;
                    }
                }
            }
        }
    };

    if (FAST && !force_tracked) {
        run(no_t());
        // a row sum that reached 2^96 (or is not finite): some score lay > 96 log2 units above the query's first-tile scores.
        // The block's waves share the ring, so they vote and redo the column together.
        bool bad = false;
        if (active) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) bad = bad || !(row_sum(qb) < C16_OVERFLOW) || (PRE && !(row_sum(qb) > 1.f / C16_OVERFLOW));
        }
        int* flags = reinterpret_cast<int*>(smem_b + Cfg::RING);
        const int wave_bad = __builtin_amdgcn_ballot_w64(bad) != 0;
        if (lane == 0) flags[wave] = wave_bad;
        wait_dma_then_barrier<0>();
        const int any_bad = (C16_WHATIF & 32) ? 0 : (flags[0] | flags[1] | flags[2] | flags[3]);       // block-uniform
        if (!any_bad) {
            store_context();
            return;
        }
        wait_dma_then_barrier<0>();          // every wave has read the flags before the ring is refilled
    }
    run(yes_t());
    store_context();
}

}  // namespace rnamsm

using namespace rnamsm;

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <int SP, int FMT, int OUT, int QB, bool MASK, bool FAST, bool PRE>
static int col16_go(unsigned grid, int batch, hipStream_t s, const uint16_t* q_hi, const uint16_t* q_lo, const uint16_t* k_hi,
                    const uint16_t* k_lo, const uint16_t* v_hi, const uint16_t* v_lo, int64_t ld, float* ctx, int64_t ldc, int R, int C,
                    int H, uint16_t* ctx_hi, uint16_t* ctx_lo, float scale, const uint8_t* pad_mask, int64_t qkv_bstride,
                    int64_t ctx_bstride, int64_t mask_bstride, int force_tracked) {
    static DeviceOnce cfg;
    auto kern = col_attn16_kernel<SP, FMT, OUT, QB, MASK, FAST, PRE>;
    constexpr int lds = C16Cfg<SP, QB>::LDS;
    if (cfg.pending()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return fail(RNAMSM_ERR_HIP, "col_attn16: hipFuncSetAttribute: %s", hipGetErrorString(e));
        cfg.mark();
    }
    hipLaunchKernelGGL(kern, dim3(grid, batch), dim3(C16_THREADS), lds, s, q_hi, q_lo, k_hi, k_lo, v_hi, v_lo, ld, ctx, ldc, R, C, H,
                       ctx_hi, ctx_lo, scale, pad_mask, qkv_bstride, ctx_bstride, mask_bstride, force_tracked);
    return RNAMSM_OK;
}

static int col_attn16_launch(const uint16_t* q_hi, const uint16_t* q_lo, const uint16_t* k_hi, const uint16_t* k_lo,
                             const uint16_t* v_hi, const uint16_t* v_lo, int64_t ld, float* ctx, int64_t ldc, int R, int C, int H,
                             int head_dim, float scale, const uint8_t* pad_mask, uint16_t* ctx_hi, uint16_t* ctx_lo, int fmt,
                             void* stream, int batch, int64_t qkv_bstride, int64_t ctx_bstride, int64_t mask_bstride, bool prescaled = false) {
    RNAMSM_CHECK_ARG(batch >= 1 && batch <= 65535 && qkv_bstride % 8 == 0 && ctx_bstride % 4 == 0, "col_attn16: bad batch / strides");
    RNAMSM_CHECK_ARG(!prescaled || (fmt == 0 && !pad_mask), "col_attn16_prescaled: bf16 operand formats without a padding mask only");
    RNAMSM_CHECK_ARG(q_hi && k_hi && v_hi && (ctx || ctx_hi), "col_attn16: null pointer");
    RNAMSM_CHECK_ARG((q_lo == nullptr) == (k_lo == nullptr) && (q_lo == nullptr) == (v_lo == nullptr),
                     "col_attn16: the lo planes must all be given (x3) or all be null");
    RNAMSM_CHECK_ARG(!ctx_hi || (ctx_lo == nullptr) == (q_lo == nullptr), "col_attn16: ctx_lo must match the operand split");
    RNAMSM_CHECK_ARG(head_dim == 64, "col_attn16: head_dim must be 64 (got %d)", head_dim);
    RNAMSM_CHECK_ARG(R > 0 && R <= 1024 && C > 0 && H > 0, "col_attn16: bad shape R=%d C=%d H=%d", R, C, H);
    RNAMSM_CHECK_ARG(scale > 0.f && scale < 1e30f, "col_attn16: scale must be positive and finite");
    RNAMSM_CHECK_ARG(fmt == 0 || (fmt == 1 && q_lo), "col_attn16: fmt must be 0 (bf16) or 1 (fp16, hi/lo only)");
    RNAMSM_CHECK_ARG(ld >= (int64_t)H * 64 && ld % 8 == 0 && al16(q_hi) && al16(k_hi) && al16(v_hi) && al16(q_lo) && al16(k_lo) && al16(v_lo),
                     "col_attn16: planes must be 16-byte aligned with ld %% 8 == 0");
    RNAMSM_CHECK_ARG(ldc >= (int64_t)H * 64 && (ctx_hi ? ldc % 8 == 0 && ctx_bstride % 8 == 0 && al16(ctx_hi) && al16(ctx_lo) : ldc % 4 == 0 && al16(ctx)),
                     "col_attn16: the context must be 16-byte aligned (ldc %% 8 == 0 for planes, %% 4 for fp32)");
    RNAMSM_NO_BF16X3(q_lo && fmt == 0, "col_attn16");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int npl = q_lo ? 2 : 1;
    const int var = tuning().attn16;
    // vector-ALU instructions per score (a transcendental counts 4): exp 4 + 1/2 v_cvt_pk + 1/2 v_dot2c (row sum) = 5, + 1 fma when q
    // is not prescaled; TRACKED adds the maximum (v_maximum3: 1/2) and the rescale of O (2); the hi/lo modes the lo half of P
    // (sub + 1/2 cvt) and an fp32 add instead of the dot2c (+ 1/2)
    const bool fast_loop = fmt == 0 && !pad_mask && var != 5;
    const double per_score = (prescaled ? 5.0 : 6.0) + (fast_loop ? 0.0 : 2.5) + (q_lo ? 2.0 : 0.0);
    KernelTimer timer(TC_COL_ATTN, 4.0 * batch * C * H * (double)R * R * 64, batch * (2.0 * npl * 3.0 + (ctx_hi ? 2.0 * npl : 4.0)) * R * C * H * 64, s,
                      PEAK_F16_MFMA_TFLOPS, q_lo ? 3.0 : 1.0, per_score * batch * C * H * (double)R * R);
    // two query blocks per wave (256-query blocks) for plain bf16 without a mask when that adds no idle query rows; bf16
    // operands without a mask take the FAST loop ("attn16" = 4: one query block per wave, 5: TRACKED loop only; A/B and tests)
    const bool qb2 = !q_lo && !pad_mask && var != 4 && (R + 255) / 256 * 256 <= (R + 127) / 128 * 128;
    const int force_tracked = var == 5;
    const unsigned iblocks = qb2 ? (R + 255) / 256 : (R + 127) / 128;
    const unsigned grid = xcd_panel_grid((unsigned)C * H, iblocks);
    int rc;
#define CA_ARGS grid, batch, s, q_hi, q_lo, k_hi, k_lo, v_hi, v_lo, ld, ctx, ldc, R, C, H, ctx_hi, ctx_lo, scale, pad_mask, qkv_bstride, ctx_bstride, mask_bstride, force_tracked
    // (SP, FMT, OUT, QB) x mask x {plain, prescaled}
#define CA_GO(SP_, FMT_, OUT_, QB_)                                                                                   \
    do {                                                                                                              \
        constexpr bool F_ = FMT_ == 0;                                                                                \
        if (pad_mask) rc = col16_go<SP_, FMT_, OUT_, QB_, true, false, false>(CA_ARGS);                               \
        else rc = prescaled ? col16_go<SP_, FMT_, OUT_, QB_, false, F_, F_>(CA_ARGS)                                  \
                            : col16_go<SP_, FMT_, OUT_, QB_, false, F_, false>(CA_ARGS);                              \
    } while (0)
    if (!q_lo) {
        if (qb2) { if (ctx_hi) CA_GO(1, 0, 1, 2); else CA_GO(1, 0, 0, 2); }
        else { if (ctx_hi) CA_GO(1, 0, 1, 1); else CA_GO(1, 0, 0, 1); }
    } else {
        if (ctx_hi) CA_GO(3, 1, 2, 1); else CA_GO(3, 1, 0, 1);
    }
#undef CA_GO
#undef CA_ARGS
    if (rc) return rc;
    RNAMSM_CHECK_LAUNCH("col_attn16");
    return RNAMSM_OK;
}

extern "C" int rnamsm_col_attn16(const uint16_t* q_hi, const uint16_t* q_lo, const uint16_t* k_hi, const uint16_t* k_lo,
                                 const uint16_t* v_hi, const uint16_t* v_lo, int64_t ld, float* ctx, int64_t ldc, int R,
                                 int C, int H, int head_dim, float scale, const uint8_t* pad_mask, uint16_t* ctx_hi, uint16_t* ctx_lo, int fmt,
                                 void* stream) {
    return col_attn16_launch(q_hi, q_lo, k_hi, k_lo, v_hi, v_lo, ld, ctx, ldc, R, C, H, head_dim, scale, pad_mask, ctx_hi, ctx_lo, fmt, stream,
                             1, 0, 0, 0);
}
extern "C" int rnamsm_col_attn16_prescaled(const uint16_t* q_hi, const uint16_t* q_lo, const uint16_t* k_hi, const uint16_t* k_lo,
                                           const uint16_t* v_hi, const uint16_t* v_lo, int64_t ld, float* ctx, int64_t ldc, int R,
                                           int C, int H, int head_dim, uint16_t* ctx_hi, uint16_t* ctx_lo, void* stream) {
    return col_attn16_launch(q_hi, q_lo, k_hi, k_lo, v_hi, v_lo, ld, ctx, ldc, R, C, H, head_dim, 1.f, nullptr, ctx_hi, ctx_lo, 0, stream,
                             1, 0, 0, 0, true);
}
namespace rnamsm {
int col_attn16_batched(const uint16_t* q_hi, const uint16_t* q_lo, const uint16_t* k_hi, const uint16_t* k_lo, const uint16_t* v_hi,
                       const uint16_t* v_lo, int64_t ld, int64_t ldc, int R, int C, int H, float scale, const uint8_t* pad_mask,
                       uint16_t* ctx_hi, uint16_t* ctx_lo, int fmt, int batch, int64_t qkv_bstride, int64_t ctx_bstride, int64_t mask_bstride,
                       void* stream, bool prescaled) {
    return col_attn16_launch(q_hi, q_lo, k_hi, k_lo, v_hi, v_lo, ld, nullptr, ldc, R, C, H, 64, scale, pad_mask, ctx_hi, ctx_lo, fmt, stream,
                             batch, qkv_bstride, ctx_bstride, mask_bstride, prescaled);
}
}  // namespace rnamsm
