// K7': fused column attention on the 16-bit matrix cores (16-bit modes of ColumnSelfAttention, modules.py:875-924).
//
// Same algorithm as col_attn.hip (S^T = K Q^T with the query on the lane, online softmax in fp32 registers, exp(S^T)
// used in place as the B operand of O^T += V^T P^T), with q/k/v arriving as 16-bit hi(+lo) planes [T, 3D] from the QKV
// GEMM epilogue:
//   K chunk [64 keys][64 d]  "k" tile of tile16.h, DMA-staged, read with ds_read_b128 (A operand of S^T, k = d);
//   V chunk [64 keys][64 d]  "t" tile, staged exactly as it lies in memory and read with ds_read_b64_tr_b16 as the
//                            A operand V^T[d][key] of O^T -- the hardware transposed read replaces a transposing copy;
//   P                        the fp32 accumulator registers 8s..8s+7 of S^T, converted pairwise to halves, ARE the B
//                            fragment of k-step s; its k order is key 16s + 8(e>>2) + 4*half + (e&3) for element e, so
//                            the V fragment is built from the two 4-key blocks 16s + 4*half and 16s + 8 + 4*half.
// scale (dh^-0.5) multiplies the fp32 scores, not q: an unscaled q keeps its fp16 lo plane clear of subnormals.
// SPLIT 1: bf16 operands, one MFMA per product.  SPLIT 3: hi/lo pairs, 3 MFMAs (P = hi + lo with |lo| <= 2^-11 P).
// Per 32-key tile and wave: (4 + 4) * SPLIT MFMAs of 32 cycles against ~150 VALU for the softmax -> VALU/exp-bound;
// two blocks per CU interleave one block's softmax with the other's MFMAs.
#include "tile16.h"

namespace rnamsm {

constexpr int C16_THREADS = 256;
constexpr int C16_ROWS = 128;                  // query rows per block (4 waves x 32)
// keys per chunk JC: 64 (two 32-key tiles per barrier) for plain bf16, 32 for the hi/lo modes -- their four planes per
// chunk would otherwise cost 64 KB of LDS per block and cap the CU at two blocks; at 32 KB three fit (the registers'
// limit), and this kernel lives on occupancy: one problem's loop is shorter than the fixed cost around it (strided q
// rows, first chunk, output stores), which only other resident blocks can hide.

// MASKED: f2 padding mask (scores of padded keys of this column become -10000 after scaling, modules.py:911-915); the
// un-masked instances carry no mask code.
template <int SPLIT, int FMT, int OUT, int JC, bool MASKED>
__global__ __launch_bounds__(C16_THREADS, JC == 32 ? 3 : 2) void col_attn16_kernel(
    const uint16_t* __restrict__ qhi, const uint16_t* __restrict__ qlo, const uint16_t* __restrict__ khi,
    const uint16_t* __restrict__ klo, const uint16_t* __restrict__ vhi, const uint16_t* __restrict__ vlo, int64_t ld,
    float* __restrict__ ctx, int64_t ldc, int R, int C, int H, uint16_t* __restrict__ ctx_hi, uint16_t* __restrict__ ctx_lo,
    float scale, const uint8_t* __restrict__ pad_mask, int64_t qkv_bstride, int64_t ctx_bstride, int64_t mask_bstride) {
    constexpr int NPL = SPLIT == 3 ? 2 : 1;
    // batched launch (rnamsm_forward_batch, 16-bit modes): MSA blockIdx.y
    qhi += blockIdx.y * qkv_bstride; khi += blockIdx.y * qkv_bstride; vhi += blockIdx.y * qkv_bstride;
    if (qlo) { qlo += blockIdx.y * qkv_bstride; klo += blockIdx.y * qkv_bstride; vlo += blockIdx.y * qkv_bstride; }
    if (ctx) ctx += blockIdx.y * ctx_bstride;
    if (ctx_hi) ctx_hi += blockIdx.y * ctx_bstride;
    if (ctx_lo) ctx_lo += blockIdx.y * ctx_bstride;
    if (MASKED) pad_mask += blockIdx.y * mask_bstride;
    constexpr int C16_JC = JC;
    constexpr int C16_TILE = JC * T16_ROWB;            // bytes per plane tile
    constexpr int BUF = 2 * NPL * C16_TILE;            // K planes then V planes
    typedef typename Half16<FMT>::T Hh;
    typedef typename Half16<FMT>::V8 V8;
    extern __shared__ __attribute__((aligned(16))) char smem_b[];

    const unsigned iblocks = (R + C16_ROWS - 1) / C16_ROWS;
    unsigned prob, ib;
    if (!xcd_panel_map(blockIdx.x, (unsigned)C * H, iblocks, prob, ib)) return;
    const int c = prob / H, h = prob % H;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int irow0 = ib * C16_ROWS + wave * 32;
    const bool active = irow0 < R;                             // wave-uniform
    const int64_t col_off = (int64_t)c * ld + h * 64;          // + r*C*ld selects the alignment row

    const uint16_t* qpl[2] = {qhi, qlo};
    const uint16_t* kpl[2] = {khi, klo};
    const uint16_t* vpl[2] = {vhi, vlo};

    // Q fragment (B operand of S^T = K Q^T): lane (query i, half) holds q[i][16kk + 8*half + 0..7]
    V8 qf[4][NPL];
    {
        const int qi = min(irow0 + li, R - 1);
        const int64_t qo = (int64_t)qi * C * ld + col_off + 8 * lh;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int p = 0; p < NPL; ++p) qf[kk][p] = *reinterpret_cast<const V8*>(qpl[p] + qo + 16 * kk);
    }

    // DMA map: a plane tile is 8 groups of 8 key rows; wave w moves groups w and w+4.  Keys past R are clamped to the
    // last key: their scores are masked to -inf and their V values only meet P = 0.
    const int drow = lane >> 3;
    const int ck = dma_chunk_k(lane, wave), ct = dma_chunk_t(lane);
    auto issue = [&](int ch, int buf) {
        char* base = smem_b + buf * BUF;
#pragma unroll
        for (int j = 0; j < JC / 32; ++j) {
            const int row = 8 * (wave + 4 * j) + drow;
            const int64_t ko = (int64_t)min(ch * C16_JC + row, R - 1) * C * ld + col_off;
            const int loff = (8 * (wave + 4 * j)) * T16_ROWB;
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                dma16(kpl[p] + ko + ck * 8, base + p * C16_TILE + loff);
                dma16(vpl[p] + ko + ct * 8, base + (NPL + p) * C16_TILE + loff);
            }
        }
    };

    f32x16 o0, o1;                       // O^T tiles: head dims [0,32) and [32,64) x 32 query rows
#pragma unroll
    for (int t = 0; t < 16; ++t) { o0[t] = 0.f; o1[t] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    const int tq = (lane & 15) >> 2, tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);    // transposed-read geometry

    auto tile = [&](const char* Kc, const char* Vc, int jt, int jbase) {
        // ---- S^T = K Q^T: 32 keys x 32 queries, k = 64 head dims
        f32x16 s;
#pragma unroll
        for (int t = 0; t < 16; ++t) s[t] = 0.f;
        V8 kf[4][NPL];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int p = 0; p < NPL; ++p) kf[kk][p] = frag_k<FMT>(Kc + p * C16_TILE, jt * 32 + li, kk, lh);
        // V^T fragments of this tile, requested before the MFMAs that hide them
        V8 vf[2][2][NPL];                // [d tile][k step][plane]
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ra = jt * 32 + 16 * ks + 4 * lh + tq;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int p = 0; p < NPL; ++p)
                    vf[dt][ks][p] = frag_t<FMT>(Vc + p * C16_TILE, ra, ra + 8, dt * 32 + tcol);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) s = mma16<SPLIT, FMT>(kf[kk], qf[kk], s);
        // ---- online softmax (fp32): keys >= R masked branch-free; __expf = v_exp_f32(x * log2e)
        const int limit = R - jbase;
        if (limit < 32) {        // block-uniform: only a ragged last tile holds keys >= R
#pragma unroll
            for (int t = 0; t < 16; ++t)
                s[t] = ((t & 3) + 8 * (t >> 2) + 4 * lh < limit) ? s[t] * scale : -INFINITY;
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) s[t] *= scale;
        }
        if (MASKED) {
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
        // P is kept as exp(s - m) * 2^12 (the max shifted by 12 ln 2): the running sum carries the same factor, so
        // O / l is unchanged, and the fp16 lo plane of P stays out of subnormals for every P that matters.
        const float m_shift = m_new - 8.317766167f;
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            s[t] = __expf(s[t] - m_shift);
            psum += s[t];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < 16; ++t) { o0[t] *= alpha; o1[t] *= alpha; }
        // ---- P fragments: registers 8ks..8ks+7 -> halves (hi, and lo = P - hi)
        V8 pf[2][NPL];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float pv = s[8 * ks + e];
                const Hh hi = (Hh)pv;
                pf[ks][0][e] = hi;
                if (SPLIT == 3) pf[ks][NPL - 1][e] = (Hh)(pv - (float)hi);
            }
        // ---- O^T += V^T P^T
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            o0 = mma16<SPLIT, FMT>(vf[0][ks], pf[ks], o0);
            o1 = mma16<SPLIT, FMT>(vf[1][ks], pf[ks], o1);
        }
    };

    const int nch = (R + C16_JC - 1) / C16_JC;
    issue(0, 0);
    for (int ch = 0; ch < nch; ++ch) {
        wait_dma_then_barrier<0>();      // chunk ch has landed (every wave's share), the other buffer is free again
        if (ch + 1 < nch) issue(ch + 1, (ch + 1) & 1);
        if (active) {
            const char* Kc = smem_b + (ch & 1) * BUF;
            const char* Vc = Kc + NPL * C16_TILE;
            const int jbase = ch * C16_JC;
            tile(Kc, Vc, 0, jbase);
            if (JC == 64 && jbase + 32 < R) tile(Kc, Vc, 1, jbase + 32);     // block-uniform
        }
    }

    if (active) {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.f / l_tot;
        const int i = irow0 + li;
        if (i < R) {
            const int64_t ooff = ((int64_t)i * C + c) * ldc + h * 64 + 4 * lh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {      // registers 4g..4g+3 are head dims 8g + 4*half + {0..3}
                const f32x4 a = f32x4{o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv};
                const f32x4 b = f32x4{o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv};
                if (OUT == 0) {
                    *reinterpret_cast<f32x4*>(ctx + ooff + 8 * g) = a;
                    *reinterpret_cast<f32x4*>(ctx + ooff + 32 + 8 * g) = b;
                } else {
                    typedef typename Half16<(OUT > 0 ? OUT - 1 : 0)>::T Ho;
                    typedef Ho H4 __attribute__((ext_vector_type(4)));
                    H4 ah, al, bh, bl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ah[e] = (Ho)a[e]; al[e] = (Ho)(a[e] - (float)ah[e]);
                        bh[e] = (Ho)b[e]; bl[e] = (Ho)(b[e] - (float)bh[e]);
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

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static int col_attn16_launch(const uint16_t* q_hi, const uint16_t* q_lo, const uint16_t* k_hi, const uint16_t* k_lo,
                             const uint16_t* v_hi, const uint16_t* v_lo, int64_t ld, float* ctx, int64_t ldc, int R, int C, int H,
                             int head_dim, float scale, const uint8_t* pad_mask, uint16_t* ctx_hi, uint16_t* ctx_lo, int fmt,
                             void* stream, int batch, int64_t qkv_bstride, int64_t ctx_bstride, int64_t mask_bstride) {
    RNAMSM_CHECK_ARG(batch >= 1 && batch <= 65535 && qkv_bstride % 8 == 0 && ctx_bstride % 4 == 0, "col_attn16: bad batch / strides");
    RNAMSM_CHECK_ARG(q_hi && k_hi && v_hi && (ctx || ctx_hi), "col_attn16: null pointer");
    RNAMSM_CHECK_ARG((q_lo == nullptr) == (k_lo == nullptr) && (q_lo == nullptr) == (v_lo == nullptr),
                     "col_attn16: the lo planes must all be given (x3) or all be null");
    RNAMSM_CHECK_ARG(!ctx_hi || (ctx_lo == nullptr) == (q_lo == nullptr), "col_attn16: ctx_lo must match the operand split");
    RNAMSM_CHECK_ARG(head_dim == 64, "col_attn16: head_dim must be 64 (got %d)", head_dim);
    RNAMSM_CHECK_ARG(R > 0 && R <= 1024 && C > 0 && H > 0, "col_attn16: bad shape R=%d C=%d H=%d", R, C, H);
    RNAMSM_CHECK_ARG(fmt == 0 || (fmt == 1 && q_lo), "col_attn16: fmt must be 0 (bf16) or 1 (fp16, hi/lo only)");
    RNAMSM_CHECK_ARG(ld >= (int64_t)H * 64 && ld % 8 == 0 && al16(q_hi) && al16(k_hi) && al16(v_hi) && al16(q_lo) && al16(k_lo) && al16(v_lo),
                     "col_attn16: planes must be 16-byte aligned with ld %% 8 == 0");
    RNAMSM_CHECK_ARG(ldc >= (int64_t)H * 64 && ldc % 4 == 0 && (ctx_hi ? (reinterpret_cast<uintptr_t>(ctx_hi) & 7u) == 0 : al16(ctx)),
                     "col_attn16: output alignment");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned iblocks = (R + C16_ROWS - 1) / C16_ROWS;
    const unsigned grid = xcd_panel_grid((unsigned)C * H, iblocks);
    const int npl = q_lo ? 2 : 1;
    const int jc = q_lo ? 32 : 64;
    const int lds = 2 * 2 * npl * jc * T16_ROWB;
    KernelTimer timer(TC_COL_ATTN, 4.0 * batch * C * H * (double)R * R * 64, batch * (2.0 * npl * 3.0 + (ctx_hi ? 2.0 * npl : 4.0)) * R * C * H * 64, s,
                      PEAK_F16_MFMA_TFLOPS, q_lo ? 3.0 : 1.0);
#define CA_GO(SP_, FMT_, OUT_)                                                                                      \
    do {                                                                                                            \
        if (pad_mask) CA_GO2(SP_, FMT_, OUT_, true); else CA_GO2(SP_, FMT_, OUT_, false);                           \
    } while (0)
#define CA_GO2(SP_, FMT_, OUT_, MASK_)                                                                              \
    do {                                                                                                            \
        static DeviceOnce cfg_;                                                                                   \
        if (cfg_.pending()) {                                                                                                \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(col_attn16_kernel<SP_, FMT_, OUT_, (SP_ == 3 ? 32 : 64), MASK_>),   \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);                    \
            if (e != hipSuccess) return fail(RNAMSM_ERR_HIP, "col_attn16: hipFuncSetAttribute: %s", hipGetErrorString(e)); \
            cfg_.mark();                                                                                            \
        }                                                                                                           \
        hipLaunchKernelGGL((col_attn16_kernel<SP_, FMT_, OUT_, (SP_ == 3 ? 32 : 64), MASK_>), dim3(grid, batch), dim3(C16_THREADS), lds, s, q_hi, q_lo, k_hi, \
                           k_lo, v_hi, v_lo, ld, ctx, ldc, R, C, H, ctx_hi, ctx_lo, scale, pad_mask, qkv_bstride, ctx_bstride, mask_bstride); \
    } while (0)
    if (!q_lo) {
        if (ctx_hi) CA_GO(1, 0, 1); else CA_GO(1, 0, 0);
    } else if (fmt == 0) {
        if (ctx_hi) CA_GO(3, 0, 1); else CA_GO(3, 0, 0);
    } else {
        if (ctx_hi) CA_GO(3, 1, 2); else CA_GO(3, 1, 0);
    }
#undef CA_GO
#undef CA_GO2
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
namespace rnamsm {
int col_attn16_batched(const uint16_t* q_hi, const uint16_t* q_lo, const uint16_t* k_hi, const uint16_t* k_lo, const uint16_t* v_hi,
                       const uint16_t* v_lo, int64_t ld, int64_t ldc, int R, int C, int H, float scale, const uint8_t* pad_mask,
                       uint16_t* ctx_hi, uint16_t* ctx_lo, int fmt, int batch, int64_t qkv_bstride, int64_t ctx_bstride, int64_t mask_bstride,
                       void* stream) {
    return col_attn16_launch(q_hi, q_lo, k_hi, k_lo, v_hi, v_lo, ld, nullptr, ldc, R, C, H, 64, scale, pad_mask, ctx_hi, ctx_lo, fmt, stream,
                             batch, qkv_bstride, ctx_bstride, mask_bstride);
}
}  // namespace rnamsm
