// Whole-forward driver: one C call enqueues K0..K10 for one MSA on the caller's stream
// (MSATransformer.forward, model.py:338-416, with AxialTransformerLayer.forward, modules.py:242-267, and
// NormalizedResidualBlock.forward, modules.py:385-401, unrolled into 13 launches per layer, 10 with
// LayerNorm folded into the GEMMs).  The order of a layer's steps is written once (LayerSteps / run_layers below); the three
// drivers -- one alignment, a same-shape batch, a token-packed batch -- fill the steps for their layout and arithmetic.
//
// HBM layout of the workspace (T = R*C tokens, D = embed dim, F = 4D):
//   x      [T, D]    residual stream, updated in place by the out_proj / fc2 epilogues (K8)
//   xn     [T, D]    LayerNorm output, the A operand of the next GEMM
//   wide   [T, 4D]   fused QKV activation [T, 3D] (row stride 3D) followed by the attention context [T, D];
//                    the FFN hidden activation [T, F] overlays both (they are never live together)
//   part   [nsplit, H, C, C]  row-logit partial slabs (K4 -> K5)
//   pplanes [2][H*C, ldp]     row-attention probabilities as 16-bit hi / lo planes (16-bit modes, K5' -> K6')
//   rowsum [D/32, T, 2]  (sum x, sum (x - slab mean)^2) of every token per 32-feature slab (slab-major), left by whatever wrote x last (K0 via
//                    rnamsm_row_partials, then the out_proj / fc2 epilogues): with LayerNorm folded into the consuming
//                    GEMM (ln_folded given, exact path, no padding) xn is never touched -- the QKV / fc1 GEMMs read x and
//                    take each row's (mean, rstd) from these sums
// In the 16-bit modes the same regions hold 16-bit planes instead: xn = LayerNorm hi|lo, the QKV slot of wide = q|k|v
// hi plane [T,3D] then lo plane [T,3D] (2 x 2 B = the fp32 footprint), ctx = context hi|lo, hidden = GELU hi|lo.
// cfg3 (R=256, C=512): 403 MB + 403 MB + 1.61 GB + 75 MB; cfg5 (R=C=1024): 19.3 GB -- one 288 GB HBM3E stack set
// holds every activation of the largest supported MSA, so nothing is chunked or recomputed.
#include <algorithm>
#include <cstdint>
#include <functional>

#include "common.h"

using namespace rnamsm;

namespace {
constexpr float LOG2E = 1.4426950408889634f;
struct Layout {
    size_t x, xn, wide, part, mask, pplanes, rowsum, stats, splitk, total;
};
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
Layout make_layout(const rnamsm_model_dims& d, int R, int C, int nchunks) {
    const size_t T = (size_t)R * C, D = d.embed_dim;
    Layout l;
    size_t off = 0;
    l.x = off;    off += align256(T * D * 4);
    l.xn = off;   off += align256(T * D * 4);
    l.wide = off; off += align256(T * (size_t)(3 * D + D > (size_t)d.ffn_dim ? 4 * D : d.ffn_dim) * 4);
    {
        const size_t a = rnamsm_row_logits_workspace_bytes(R, C, d.num_heads), b = rnamsm_row_logits16_workspace_bytes(R, C, d.num_heads);
        const size_t c = (size_t)nchunks * d.num_heads * C * C * sizeof(float);      // one slab per reference row chunk (f2)
        const size_t ab = a > b ? a : b;
        l.part = off; off += align256(ab > c ? ab : c);
    }
    l.mask = off; off += align256(T);
    l.pplanes = off; off += align256((size_t)d.num_heads * C * (size_t)((C + 63) / 64 * 64) * 4);   // P hi + lo planes (K5' -> K6')
    l.rowsum = off; off += align256(T * (D / 32) * 2 * sizeof(float));   // (sum, centred sum of squares) per token and 32-feature slab
    l.stats = off; off += align256(T * 2 * sizeof(float));            // (mean, rstd) per token, combined from rowsum
    {   // split-K partial tiles of fc2 at small token counts (gemm_f32_splitk): [ks][T][D] -- sized by shape alone, for the most
        // ranges any knob value can ask for, so that a knob changed between the two calls cannot outgrow it
        l.splitk = off; off += align256(rnamsm::splitk_workspace_floats((int64_t)T, (int)D, d.ffn_dim) * sizeof(float));
    }
    l.total = off;
    return l;
}
}  // namespace

extern "C" size_t rnamsm_forward_workspace_bytes(const rnamsm_model_dims* dims, int R, int C, int has_padding,
                                                 int max_tokens_per_msa) {
    if (!dims || R <= 0 || C <= 0) return 0;
    return make_layout(*dims, R, C, has_padding ? rnamsm_row_chunks(R, C, max_tokens_per_msa) : 0).total;
}

#define FWD(call)                   \
    do {                            \
        int rc_ = (call);           \
        if (rc_ != RNAMSM_OK) return rc_; \
    } while (0)

namespace {
// The token-parallel launches of the EXACT path -- LayerNorm (or its fold), the Linear that consumes it, the Linear that writes the
// residual stream -- written once for the three drivers (rnamsm_forward, _forward_batch, _forward_packed; round 5: they had been
// three copies of the same three lambdas).  T = tokens of the launch set and of the K-split decisions (the whole alignment's /
// batch's count also where only the first `rows` tokens are computed: the outputs-only tail), so that a part of a Linear sums in
// the order of the whole.
struct ExactPath {
    const rnamsm_model_dims& d;
    float* x;                     // residual stream [T, D]
    float* xn;                    // LayerNorm output [T, D] (unused when folded)
    float* rowsum;                // slab sums of x, left by whoever wrote it (folded)
    float* stats;                 // (mean, rstd) per token (folded)
    float* splitk;                // split-K partial tiles
    int64_t T;
    bool fold;                    // LayerNorm applied inside the consuming GEMM
    bool fold_sums;               // ... with the row statistics from the producers' epilogues (false: every GEMM sums its own rows, knob ln_fold = 2)
    const float* const* ln_folded;
    int* err_flag;
    void* stream;

    int norm(const float* g, const float* b, int64_t rows) const {
        if (fold) return RNAMSM_OK;
        return rnamsm_layernorm(x, g, b, xn, rows, d.embed_dim, d.ln_eps, stream);
    }
    // x[:rows] += A W^T + bias; folded: + the rows' slab sums, combined into (mean, rstd) right away
    int res_linear(const float* A, int64_t lda, const float* Wf, const float* bias, int64_t rows, int K) const {
        const int D = d.embed_dim;
        if (fold_sums) {
            FWD(rnamsm_gemm_residual_stats(A, lda, Wf, bias, x, D, x, D, rows, D, K, rowsum, T, RNAMSM_F32, stream));
            return rnamsm_row_stats_from_partials(rowsum, T, rows, D, d.ln_eps, stats, err_flag, stream);
        }
        // > 1 only under the "gemm_splitk" knob (fc2 of small MSAs; off by default since round 5), decided by the WHOLE token count
        const int ks = rnamsm::gemm_f32_splitk_factor(T, D, K);
        if (ks > 1) return rnamsm::gemm_f32_splitk(A, lda, Wf, bias, x, D, x, D, rows, D, K, ks, splitk, static_cast<hipStream_t>(stream));
        return rnamsm_gemm_bias_act_res(A, lda, Wf, bias, x, D, x, D, rows, D, K, RNAMSM_ACT_NONE, 1.f, 0, nullptr, RNAMSM_F32, stream);
    }
    // columns n_ofs .. n_ofs + N - 1 of the Linear in folded slot fslot (0 row QKV, 1 column QKV, 2 fc1) over `rows` tokens of LN(x)
    int lin_normed(int layer, int fslot, const float* Wf, const float* bias, int n_ofs, float* out, int64_t ldc, int64_t rows, int N,
                   int act, float scale, int scale_cols) const {
        const int D = d.embed_dim;
        if (fold) {
            const float* const* Fp = ln_folded + (size_t)layer * RNAMSM_FOLDED_PER_LAYER + 3 * fslot;
            return rnamsm_gemm_lnfold(x, D, Fp[0] + (size_t)n_ofs * D, Fp[1] + n_ofs, Fp[2] + n_ofs, d.ln_eps, fold_sums ? stats : nullptr,
                                      err_flag, out, ldc, rows, N, D, act, scale, scale_cols, RNAMSM_F32, stream);
        }
        // (knob "gemm_splitk_short": the K = 768 GEMM of a lone small alignment split over idle CUs; by the FULL width of the Linear)
        const int ks = rnamsm::gemm_f32_splitk_factor(T, fslot == 2 ? d.ffn_dim : 3 * D, D);
        if (ks > 1 && N % 128 == 0)
            return rnamsm::gemm_f32_splitk(xn, D, Wf + (size_t)n_ofs * D, bias + n_ofs, nullptr, 0, out, ldc, rows, N, D, ks, splitk,
                                           static_cast<hipStream_t>(stream), act, scale, scale_cols);
        return rnamsm_gemm_bias_act_res(xn, D, Wf + (size_t)n_ofs * D, bias + n_ofs, nullptr, 0, out, ldc, rows, N, D, act, scale,
                                        scale_cols, nullptr, RNAMSM_F32, stream);
    }
};
}  // namespace

namespace {
// ONE layer skeleton for every driver and arithmetic mode (round 6): AxialTransformerLayer.forward
// (modules.py:242-267) with its three NormalizedResidualBlocks (modules.py:385-401) unrolled -- row block, column block, feed-forward
// block, each "LayerNorm (or its fold) + the Linear it feeds", "the block's middle", "the Linear that adds into the residual stream".
// A driver describes HOW each step runs on its layout (one alignment / a same-shape batch / a token-packed batch; exact fp32 / 16-bit
// planes) by filling the steps below; WHAT runs in which order is written here once.  `last_layer_tail` (optional) may finish the
// forward after the last layer's row block (rnamsm_forward's outputs-only tail): it sets `finished`.
struct LayerSteps {
    typedef std::function<int(int layer, const float* const* W)> Step;
    Step row_qkv;            // K1 (or fold) + K2/K3: q | k | v of the tied row attention
    Step row_attention;      // K4 logits, K5 softmax (writes the layer's maps), K6 apply
    Step row_out;            // K2 + K8: x += ctx Wo^T + bo
    std::function<int(int layer, const float* const* W, bool& finished)> last_layer_tail;
    Step col_qkv;            // K1 (or fold) + K2/K3
    Step col_attention;      // K7
    Step col_out;            // K2 + K8
    Step ffn;                // K1 (or fold) + fc1 + erf-GELU, fc2 + residual
};
int run_layers(const LayerSteps& s, const float* const* weights, int num_layers, bool& finished) {
    finished = false;
    for (int l = 0; l < num_layers; ++l) {
        const float* const* W = weights + RNAMSM_W_GLOBAL_COUNT + (size_t)l * RNAMSM_W_LAYER_COUNT;
        FWD(s.row_qkv(l, W));
        FWD(s.row_attention(l, W));
        FWD(s.row_out(l, W));
        if (l == num_layers - 1 && s.last_layer_tail) {
            FWD(s.last_layer_tail(l, W, finished));
            if (finished) return RNAMSM_OK;
        }
        FWD(s.col_qkv(l, W));
        FWD(s.col_attention(l, W));
        FWD(s.col_out(l, W));
        FWD(s.ffn(l, W));
    }
    return RNAMSM_OK;
}
}  // namespace

extern "C" int rnamsm_forward(const rnamsm_model_dims* dims, const float* const* weights, const int64_t* tokens,
                              int R, int C, void* workspace, size_t workspace_bytes, float* row_attn, float* repr,
                              float* emb, float* atp, int* err_flag, int has_padding, int max_tokens_per_msa, int outputs,
                              int dtype, const uint16_t* const* weight_planes, const float* const* ln_folded,
                              const void* const* ln_folded16, void* stream) {
    RNAMSM_CHECK_ARG(dtype >= RNAMSM_F32 && dtype <= RNAMSM_F16X3, "forward: unknown dtype %d", dtype);
    RNAMSM_NO_BF16X3(dtype == RNAMSM_BF16X3_REMOVED, "forward");
    ForwardScope in_flight;                                  // rnamsm_set_param refuses until the launches are enqueued (common.h)
    RNAMSM_CHECK_ARG(dtype == RNAMSM_F32 || weight_planes, "forward: bf16 modes need weight_planes");
    RNAMSM_CHECK_ARG(dims && weights && tokens && workspace && row_attn && repr && emb && atp, "forward: null pointer");
    const rnamsm_model_dims& d = *dims;
    const int D = d.embed_dim, H = d.num_heads, F = d.ffn_dim, NL = d.num_layers;
    RNAMSM_CHECK_ARG(D > 0 && H > 0 && D == H * 64, "forward: embed_dim must be num_heads * 64 (D=%d H=%d)", D, H);
    RNAMSM_CHECK_ARG(D % 128 == 0 && F % 128 == 0 && NL > 0, "forward: embed_dim and ffn_dim must be multiples of 128");
    RNAMSM_CHECK_ARG(d.row_pos_dim == 0 || d.row_pos_dim == 1 || d.row_pos_dim == D, "forward: row_pos_dim must be 0, 1 or embed_dim (got %d)", d.row_pos_dim);
    RNAMSM_CHECK_ARG(C >= 2 && R >= 1, "forward: need R >= 1 and C >= 2 (got R=%d C=%d)", R, C);
    if (R > 1024)   // model.py:355-359
        return fail(RNAMSM_ERR_INVALID,
                    "Using model with MSA position embedding trained on maximum MSA depth of 1024, but received %d alignments.", R);
    RNAMSM_CHECK_ARG(C <= d.num_positions - d.pad_idx - 1, "forward: C=%d exceeds the positional table", C);
    // padded AND above the reference's token budget: its chunked path fills the key mask per row chunk (modules.py:717-750)
    const int nchunks = has_padding ? rnamsm_row_chunks(R, C, max_tokens_per_msa) : 0;
    const int rows_per_chunk = nchunks ? (max_tokens_per_msa / C > 1 ? max_tokens_per_msa / C : 1) : 0;
    const Layout lay = make_layout(d, R, C, nchunks);
    RNAMSM_CHECK_ARG(workspace_bytes >= lay.total, "forward: workspace too small (%zu < %zu)", workspace_bytes, lay.total);
    RNAMSM_CHECK_ARG(aligned16(workspace), "forward: workspace must be 16-byte aligned");

    char* ws = static_cast<char*>(workspace);
    float* x = reinterpret_cast<float*>(ws + lay.x);
    float* xn = reinterpret_cast<float*>(ws + lay.xn);
    float* wide = reinterpret_cast<float*>(ws + lay.wide);
    float* part = reinterpret_cast<float*>(ws + lay.part);
    const int64_t T = (int64_t)R * C;
    const int64_t ldq = 3 * (int64_t)D;
    float* qkv = wide;                 // [T, 3D]: q | k | v per token
    float* ctx = wide + T * ldq;       // [T, D]
    float* hidden = wide;              // [T, F]
    const float* const* G = weights;
    const int nsplit = rnamsm_row_logits_nsplit(R, C, H);
    // Linear dispatch: exact-fp32 MFMA, or the bf16 matrix cores on pre-split weight planes (slot = index into the
    // layer's plane table).  The masked QKV projection (f2) always takes the fp32 kernel.
    const int f32 = RNAMSM_F32;
    auto linear = [&](int layer, int slot, const float* A, int64_t lda, const float* Wf, const float* bias,
                      const float* res, int64_t ldr, float* out, int64_t ldc, int N, int K, int act, float scale,
                      int scale_cols, const uint8_t* zero_rows) -> int {
        if (dtype == RNAMSM_F32 || zero_rows) {
            // a lone small alignment: split over idle CUs (gemm_f32_splitk_factor; the reduction pass carries the epilogue)
            const int ks = rnamsm::gemm_f32_splitk_factor(T, N, K);
            if (ks > 1)
                return rnamsm::gemm_f32_splitk(A, lda, Wf, bias, res, ldr, out, ldc, T, N, K, ks, reinterpret_cast<float*>(ws + lay.splitk),
                                               static_cast<hipStream_t>(stream), act, scale, scale_cols, zero_rows);
            return rnamsm_gemm_bias_act_res(A, lda, Wf, bias, res, ldr, out, ldc, T, N, K, act, scale, scale_cols,
                                            zero_rows, f32, stream);
        }
        const uint16_t* const* P = weight_planes + (size_t)layer * RNAMSM_PLANES_PER_LAYER + 2 * slot;
        return rnamsm_gemm_bf16(A, lda, P[0], P[1], bias, res, ldr, out, ldc, T, N, K, act, scale, scale_cols,
                                dtype == RNAMSM_BF16 ? 1 : 3, dtype == RNAMSM_F16X3 ? 1 : 0, nullptr, nullptr, nullptr,
                                nullptr, stream);
    };
    // 16-bit modes without padding: LayerNorm and the fc1 epilogue write their outputs directly as hi/lo planes (the
    // A operands of the QKV / fc1 / fc2 GEMMs), so those GEMMs stage plain 16-B copies.  The planes overlay the fp32
    // buffers they replace (2 x 2 B per element).
    // (a padded batch with the "attn16" knob off keeps the fp32 attention kernels and the masked fp32 QKV GEMM)
    // (per-chunk mask fills exist in the fp32 row kernels only: such an MSA takes the exact path as a whole)
    if (nchunks) dtype = RNAMSM_F32;
    BigRowsScope big_rows_scope(dtype == RNAMSM_BF16);       // the 16-bit GEMMs' tile by this MSA's token count (common.h)
    const bool planes = dtype != RNAMSM_F32 && (!has_padding || tuning().attn16 != 0);
    const int split = dtype == RNAMSM_BF16 ? 1 : 3, fmt = dtype == RNAMSM_F16X3 ? 1 : 0;
    uint16_t* xn_hi = reinterpret_cast<uint16_t*>(xn);
    uint16_t* xn_lo = split == 3 ? xn_hi + T * D : nullptr;
    uint16_t* ctx_hi = planes ? reinterpret_cast<uint16_t*>(ctx) : nullptr;     // context planes overlay the fp32 ctx
    uint16_t* ctx_lo = planes && split == 3 ? ctx_hi + T * D : nullptr;
    // 16-bit attention (K4'..K7'): the QKV epilogue writes q|k|v planes over the fp32 qkv slot
    const bool attn16 = planes && tuning().attn16 != 0;
    uint16_t* qkv_hi = reinterpret_cast<uint16_t*>(qkv);
    uint16_t* qkv_lo = split == 3 ? qkv_hi + T * ldq : nullptr;
    auto lo_at = [&](int64_t off) -> uint16_t* { return qkv_lo ? qkv_lo + off : nullptr; };
    const int64_t ldp = (C + 63) / 64 * 64;
    uint16_t* p_hi = reinterpret_cast<uint16_t*>(ws + lay.pplanes);
    uint16_t* p_lo = split == 3 ? p_hi + (int64_t)H * C * ldp : nullptr;
    uint16_t* hid_hi = reinterpret_cast<uint16_t*>(hidden);
    uint16_t* hid_lo = split == 3 ? hid_hi + T * (int64_t)F : nullptr;
    auto ln_for_gemm = [&](const float* g, const float* b) -> int {
        if (planes) return rnamsm_layernorm_split(x, g, b, xn_hi, xn_lo, T, D, d.ln_eps, fmt, stream);
        return rnamsm_layernorm(x, g, b, xn, T, D, d.ln_eps, stream);
    };
    auto linear_pl = [&](int layer, int slot, const uint16_t* ahi, const uint16_t* alo, int64_t lda, const float* bias,
                         const float* res, int64_t ldr, float* out, uint16_t* ohi, uint16_t* olo, int64_t ldc, int N,
                         int K, int act, float scale, int scale_cols) -> int {
        const uint16_t* const* P = weight_planes + (size_t)layer * RNAMSM_PLANES_PER_LAYER + 2 * slot;
        return rnamsm_gemm_bf16(nullptr, lda, P[0], P[1], bias, res, ldr, out, ldc, T, N, K, act, scale, scale_cols, split,
                                fmt, ahi, alo, ohi, olo, stream);
    };
    // Exact path without padding, folded weights given: the three LayerNorms of a layer are applied inside the GEMMs they
    // feed (gemm_f32.hip FOLD) -- norm() is then no launch at all, lin_normed() reads x itself and res_linear() (the GEMMs
    // that write x) leaves the row sums the next lin_normed() normalises with.
    // Measured (tools/ln_fold_ab.py, whole forward): -1.4 % at M = L = 1024, -1.25 % at M=256 L=512, -0.5 % at 128x256 and at
    // 512x36 (18432 tokens).  Below that it used to LOSE (+3 % at 128x128, +1.8 % at 64x128: the 30 small statistics launches sat
    // on the critical path at 10-15 us each) and mode 1 folded from 18432 tokens; since round 5 those launches take ~3 us
    // (row_stats_from_partials keeps a row's slabs in flight) and the fold wins from 4096 tokens up: -1.1 .. -1.6 % at 64x64 .. 128x128
    // and 256x36 .. 400x36, -0.4 % at 32x64 -- LN_FOLD_MIN_TOKENS (common.h).
    const int fold_mode = tuning().ln_fold;          // 0 off, 1 by shape, 2 GEMMs sum their own rows, 3 always
    const bool fold = ln_folded && dtype == RNAMSM_F32 && !has_padding &&
                      (fold_mode >= 2 || (fold_mode == 1 && (int64_t)R * C >= LN_FOLD_MIN_TOKENS));
    const bool fold_sums = fold && fold_mode != 2;   // row sums travel from the residual epilogues to the consumers
    float* rowsum = reinterpret_cast<float*>(ws + lay.rowsum);
    float* stats = reinterpret_cast<float*>(ws + lay.stats);
    float* splitk = reinterpret_cast<float*>(ws + lay.splitk);
    // 16-bit modes (planes end to end, 16-bit attention, no padding): the same fold on the 256x256 matrix-core kernels --
    // the residual GEMMs also write the new x as planes and its slab sums, the QKV / fc1 GEMMs read those planes
    // (rnamsm_gemm16_residual_stats / rnamsm_gemm16_lnfold); rnamsm_layernorm_split is no launch at all.
    // Measured (tools/ln_fold_ab.py with DTYPE=bf16 / f16x3): the LayerNorm launches disappear (cfg3 bf16: 3.2 -> 0.25 ms,
    // 1024^2: 26.9 -> 2.2 ms) and the GEMMs take exactly that much longer (28.6 -> 30.9 ms, 223.6 -> 245.7 ms; f16x3 the
    // same picture): with one 256x256 block per CU nothing hides an epilogue, and the extra plane writes sit in the
    // HBM-bound out_proj.  Net +-0.2 %, so the 16-bit modes fold only on request (knob "ln_fold" = 3).
    const bool fold16 = ln_folded16 && attn16 && !has_padding && D % 256 == 0 && F % 256 == 0 && T >= 2048 && fold_mode == 3;
    auto lin16_fold = [&](int layer, int fslot, uint16_t* ohi, uint16_t* olo, int64_t ldo, int N, int act) -> int {
        const void* const* Fp = ln_folded16 + (size_t)layer * RNAMSM_FOLDED16_PER_LAYER + 4 * fslot;
        return rnamsm_gemm16_lnfold(xn_hi, xn_lo, D, static_cast<const uint16_t*>(Fp[0]), static_cast<const uint16_t*>(Fp[1]),
                                    static_cast<const float*>(Fp[2]), static_cast<const float*>(Fp[3]), stats, ohi, olo, ldo, T, N,
                                    D, act, 1.f, 0, split, fmt, stream);
    };
    // x += A W^T + bias, the new x once more as the xn planes, its slab sums -> (mean, rstd)
    auto res16_fold = [&](int layer, int slot, const uint16_t* ahi, const uint16_t* alo, int64_t lda, const float* bias, int K) -> int {
        const uint16_t* const* P = weight_planes + (size_t)layer * RNAMSM_PLANES_PER_LAYER + 2 * slot;
        FWD(rnamsm_gemm16_residual_stats(ahi, alo, lda, P[0], P[1], bias, x, D, T, D, K, split, fmt, xn_hi, xn_lo, D, rowsum, T, stream));
        return rnamsm_row_stats_from_partials(rowsum, T, T, D, d.ln_eps, stats, err_flag, stream);
    };
    // the exact path's token-parallel launches (ExactPath above): LayerNorm or its fold, the Linear it feeds, the residual Linear
    const ExactPath ex{d, x, xn, rowsum, stats, splitk, T, fold, fold_sums, ln_folded, err_flag, stream};
    auto norm = [&](const float* g, const float* b, int64_t rows) -> int { return ex.norm(g, b, rows); };
    auto res_linear = [&](const float* A, int64_t lda, const float* Wf, const float* bias, int64_t rows, int K) -> int {
        return ex.res_linear(A, lda, Wf, bias, rows, K);
    };
    auto lin_normed = [&](int layer, int fslot, const float* Wf, const float* bias, int n_ofs, float* out, int64_t ldc,
                          int64_t rows, int N, int act, float scale, int scale_cols) -> int {
        return ex.lin_normed(layer, fslot, Wf, bias, n_ofs, out, ldc, rows, N, act, scale, scale_cols);
    };
    const float col_scale = 1.0f / sqrtf(64.0f);                         // modules.py:839
    // align_scaling (modules.py:713-715) = dh^-1/2 / sqrt(R).  ONE arithmetic per alignment, whatever the batch it is computed
    // in: on the exact path without padding q carries dh^-1/2 only and 1/sqrt(R) multiplies the summed
    // logits in K5 -- exactly what rnamsm_forward_packed and rnamsm_forward_batch do, so an alignment's emb / atp are the same
    // bits alone, in a same-shape batch or token-packed.  With padding (the reference's masked / chunked semantics) and in
    // the 16-bit modes the factor stays where it was.
    const bool canon = dtype == RNAMSM_F32 && !has_padding;
    const float depth_scale = 1.0f / sqrtf((float)R);
    const float row_scale = canon ? col_scale : col_scale * depth_scale;

    uint8_t* mask = nullptr;           // [R, C]; row 0 of it is the tied-attention key mask
    if (has_padding) {
        mask = reinterpret_cast<uint8_t*>(ws + lay.mask);
        FWD(rnamsm_pad_mask(tokens, mask, T, d.pad_idx, stream));
    }
    FWD(rnamsm_embed_ln_rows(tokens, G[RNAMSM_W_EMBED_TOKENS], G[RNAMSM_W_EMBED_POSITIONS], G[RNAMSM_W_ROW_POS], d.row_pos_dim,
                             G[RNAMSM_W_LN_BEFORE_G], G[RNAMSM_W_LN_BEFORE_B], x, R, C, D, d.vocab, d.num_positions,
                             d.pad_idx, d.ln_eps, err_flag, stream));
    if (fold16) {                                    // K0's output as planes, with its statistics
        FWD(rnamsm_split_bf16(x, xn_hi, xn_lo, T * D, fmt, stream));
        FWD(rnamsm_row_partials(x, rowsum, T, D, stream));
        FWD(rnamsm_row_stats_from_partials(rowsum, T, T, D, d.ln_eps, stats, err_flag, stream));
    }
    if (fold_sums) {
        FWD(rnamsm_row_partials(x, rowsum, T, D, stream));
        FWD(rnamsm_row_stats_from_partials(rowsum, T, T, D, d.ln_eps, stats, err_flag, stream));
    }
    LayerSteps st;
    // ---- tied row attention block
    st.row_qkv = [&](int l, const float* const* W) -> int {
        if (!fold && !fold16) FWD(ln_for_gemm(W[RNAMSM_WL_ROW_LN_G], W[RNAMSM_WL_ROW_LN_B]));
        if (attn16) {
            // q stays unscaled in the planes; the scaling multiplies the fp32 logits (see include/rnamsm.h)
            if (fold16)
                FWD(lin16_fold(l, 0, qkv_hi, qkv_lo, ldq, 3 * D, RNAMSM_ACT_NONE));
            else
                FWD(linear_pl(l, 0, xn_hi, xn_lo, D, W[RNAMSM_WL_ROW_BQKV], nullptr, 0, nullptr, qkv_hi, qkv_lo, ldq, 3 * D, D,
                              RNAMSM_ACT_NONE, 1.f, 0));
            if (mask) FWD(rnamsm_zero_plane_rows(qkv_hi, qkv_lo, mask, T, D, ldq, stream));      // q *= 1 - padding_mask
            return RNAMSM_OK;
        }
        if (planes)
            return linear_pl(l, 0, xn_hi, xn_lo, D, W[RNAMSM_WL_ROW_BQKV], nullptr, 0, qkv, nullptr, nullptr, ldq, 3 * D, D,
                             RNAMSM_ACT_NONE, row_scale, D);
        if (fold) return lin_normed(l, 0, nullptr, nullptr, 0, qkv, ldq, T, 3 * D, RNAMSM_ACT_NONE, row_scale, D);
        return linear(l, 0, xn, D, W[RNAMSM_WL_ROW_WQKV], W[RNAMSM_WL_ROW_BQKV], nullptr, 0, qkv, ldq, 3 * D, D, RNAMSM_ACT_NONE, row_scale, D,
                      mask);
    };
    st.row_attention = [&](int l, const float* const*) -> int {
        float* probs = row_attn + (int64_t)l * H * C * C;
        if (attn16) {
            FWD(rnamsm_row_logits16(qkv_hi, qkv_lo, qkv_hi + D, lo_at(D), ldq, part, R, C, H, 64, row_scale, fmt, stream));
            FWD(rnamsm_softmax_rows_planes(part, rnamsm_row_logits16_nsplit(R, C, H, split), probs, p_hi, p_lo, ldp, 4096.f, H, C,
                                           mask, fmt, stream));
            return rnamsm_row_apply16(p_hi, p_lo, ldp, qkv_hi + 2 * D, lo_at(2 * D), ldq, nullptr, D, R, C, H, 64, 1.f / 4096.f,
                                      ctx_hi, ctx_lo, fmt, stream);
        }
        if (nchunks) {
            FWD(rnamsm_row_logits_chunked(qkv, qkv + D, ldq, part, R, C, H, 64, rows_per_chunk, f32, stream));
            FWD(rnamsm_softmax_rows_chunked(part, nchunks, probs, H, C, mask, rows_per_chunk, stream));
        } else {
            FWD(rnamsm_row_logits(qkv, qkv + D, ldq, part, R, C, H, 64, f32, stream));
            FWD(rnamsm::softmax_rows_batched(part, nsplit, probs, H, C, 1, 0, 0, mask, 0, stream, canon ? depth_scale : 1.f));
        }
        return rnamsm_row_apply(probs, qkv + 2 * D, ldq, ctx, D, R, C, H, 64, ctx_hi, ctx_lo, fmt, f32, stream);
    };
    // a block's last Linear, adding into the residual stream: slot 1 (row out_proj) / 3 (column out_proj)
    auto out_linear = [&](int l, int slot, const float* Wf, const float* bias) -> int {
        if (fold16) return res16_fold(l, slot, ctx_hi, ctx_lo, D, bias, D);
        if (planes) return linear_pl(l, slot, ctx_hi, ctx_lo, D, bias, x, D, x, nullptr, nullptr, D, D, D, RNAMSM_ACT_NONE, 1.f, 0);
        return dtype == RNAMSM_F32 ? res_linear(ctx, D, Wf, bias, T, D)
                                   : linear(l, slot, ctx, D, Wf, bias, x, D, x, D, D, D, RNAMSM_ACT_NONE, 1.f, 0, nullptr);
    };
    st.row_out = [&](int l, const float* const* W) -> int { return out_linear(l, 1, W[RNAMSM_WL_ROW_WO], W[RNAMSM_WL_ROW_BO]); };
    if (!(outputs & RNAMSM_OUT_REPR) && dtype == RNAMSM_F32 && !has_padding && R > 1)
        st.last_layer_tail = [&](int l, const float* const* W, bool& finished) -> int {
            // Only emb (alignment row 0 of the final representation) and the maps are wanted, and the maps are complete:
            // from here on every row but row 0 is dead.  The last column attention still needs K and V of all rows
            // (LayerNorm + the k|v two thirds of the QKV GEMM over all T tokens), but its queries, its out_proj, the FFN
            // and the final LayerNorm run on row 0's C tokens (the first C rows of every [T, .] buffer).  Same kernels,
            // same per-element arithmetic: emb is bit-identical to the full forward's.
            const int64_t Tq = C;
            FWD(norm(W[RNAMSM_WL_COL_LN_G], W[RNAMSM_WL_COL_LN_B], T));
            FWD(lin_normed(l, 1, W[RNAMSM_WL_COL_WQKV], W[RNAMSM_WL_COL_BQKV], D, qkv + D, ldq, T, 2 * D, RNAMSM_ACT_NONE, 1.f, 0));   // k | v
            FWD(lin_normed(l, 1, W[RNAMSM_WL_COL_WQKV], W[RNAMSM_WL_COL_BQKV], 0, qkv, ldq, Tq, D, RNAMSM_ACT_NONE, col_scale * LOG2E, D));  // q, row 0 (log2 units, as below)
            FWD(rnamsm_col_attn_fused_prescaled(qkv, qkv + D, qkv + 2 * D, ldq, ctx, D, R, C, H, 64, 1, stream));
            FWD(res_linear(ctx, D, W[RNAMSM_WL_COL_WO], W[RNAMSM_WL_COL_BO], Tq, D));
            FWD(norm(W[RNAMSM_WL_FFN_LN_G], W[RNAMSM_WL_FFN_LN_B], Tq));
            FWD(lin_normed(l, 2, W[RNAMSM_WL_FC1_W], W[RNAMSM_WL_FC1_B], 0, hidden, F, Tq, F, RNAMSM_ACT_GELU_ERF, 1.f, 0));
            FWD(res_linear(hidden, F, W[RNAMSM_WL_FC2_W], W[RNAMSM_WL_FC2_B], Tq, F));
            FWD(rnamsm_layernorm(x, G[RNAMSM_W_LN_AFTER_G], G[RNAMSM_W_LN_AFTER_B], repr, Tq, D, d.ln_eps, stream));
            FWD(rnamsm::pack_outputs_batched(repr, row_attn, emb, atp, C, D, NL, H, 1, 0, 0, err_flag, static_cast<hipStream_t>(stream)));
            finished = true;
            return RNAMSM_OK;
        };
    // ---- column attention block
    // 16-bit, bf16 operand formats, no padding: q leaves the QKV epilogue PRESCALED by dh^-0.5 log2(e) (multiplied in before the
    // rounding to 16 bits) and the column kernel exponentiates the scores as they come (rnamsm_col_attn16_prescaled)
    const bool pre16 = fmt == 0 && !mask && !fold16;
    // exact path without padding: q leaves the QKV epilogue in log2 units (dh^-1/2 * log2(e)) and the column kernel's first
    // pass runs without a running maximum (rnamsm_col_attn_fused_prescaled)
    const bool pre32 = dtype == RNAMSM_F32 && !mask;
    st.col_qkv = [&](int l, const float* const* W) -> int {
        if (!fold && !fold16) FWD(ln_for_gemm(W[RNAMSM_WL_COL_LN_G], W[RNAMSM_WL_COL_LN_B]));
        if (attn16) {
            if (fold16) return lin16_fold(l, 1, qkv_hi, qkv_lo, ldq, 3 * D, RNAMSM_ACT_NONE);
            return linear_pl(l, 2, xn_hi, xn_lo, D, W[RNAMSM_WL_COL_BQKV], nullptr, 0, nullptr, qkv_hi, qkv_lo, ldq, 3 * D, D,
                             RNAMSM_ACT_NONE, pre16 ? col_scale * 1.4426950408889634f : 1.f, pre16 ? D : 0);
        }
        const float cs = pre32 ? col_scale * LOG2E : col_scale;
        if (planes)
            return linear_pl(l, 2, xn_hi, xn_lo, D, W[RNAMSM_WL_COL_BQKV], nullptr, 0, qkv, nullptr, nullptr, ldq, 3 * D, D,
                             RNAMSM_ACT_NONE, col_scale, D);
        if (fold) return lin_normed(l, 1, nullptr, nullptr, 0, qkv, ldq, T, 3 * D, RNAMSM_ACT_NONE, cs, D);
        return linear(l, 2, xn, D, W[RNAMSM_WL_COL_WQKV], W[RNAMSM_WL_COL_BQKV], nullptr, 0, qkv, ldq, 3 * D, D, RNAMSM_ACT_NONE, cs, D, nullptr);
    };
    st.col_attention = [&](int, const float* const*) -> int {
        if (attn16) {
            if (pre16)
                return rnamsm_col_attn16_prescaled(qkv_hi, qkv_lo, qkv_hi + D, lo_at(D), qkv_hi + 2 * D, lo_at(2 * D), ldq, nullptr, D, R, C,
                                                   H, 64, ctx_hi, ctx_lo, stream);
            return rnamsm_col_attn16(qkv_hi, qkv_lo, qkv_hi + D, lo_at(D), qkv_hi + 2 * D, lo_at(2 * D), ldq, nullptr, D, R, C,
                                     H, 64, col_scale, mask, ctx_hi, ctx_lo, fmt, stream);
        }
        if (pre32) return rnamsm_col_attn_fused_prescaled(qkv, qkv + D, qkv + 2 * D, ldq, ctx, D, R, C, H, 64, R, stream);
        return rnamsm_col_attn_fused(qkv, qkv + D, qkv + 2 * D, ldq, ctx, D, R, C, H, 64, mask, ctx_hi, ctx_lo, fmt, f32, stream);
    };
    st.col_out = [&](int l, const float* const* W) -> int { return out_linear(l, 3, W[RNAMSM_WL_COL_WO], W[RNAMSM_WL_COL_BO]); };
    // ---- feed-forward block
    st.ffn = [&](int l, const float* const* W) -> int {
        if (!fold && !fold16) FWD(ln_for_gemm(W[RNAMSM_WL_FFN_LN_G], W[RNAMSM_WL_FFN_LN_B]));
        if (fold16) {
            FWD(lin16_fold(l, 2, hid_hi, hid_lo, F, F, RNAMSM_ACT_GELU_ERF));
            if (l + 1 < NL) return res16_fold(l, 5, hid_hi, hid_lo, F, W[RNAMSM_WL_FC2_B], F);
            // the last fc2 feeds the final LayerNorm kernel, which reads the fp32 stream
            return linear_pl(l, 5, hid_hi, hid_lo, F, W[RNAMSM_WL_FC2_B], x, D, x, nullptr, nullptr, D, D, F, RNAMSM_ACT_NONE, 1.f, 0);
        }
        if (planes) {
            FWD(linear_pl(l, 4, xn_hi, xn_lo, D, W[RNAMSM_WL_FC1_B], nullptr, 0, nullptr, hid_hi, hid_lo, F, F, D,
                          RNAMSM_ACT_GELU_ERF, 1.f, 0));
            return linear_pl(l, 5, hid_hi, hid_lo, F, W[RNAMSM_WL_FC2_B], x, D, x, nullptr, nullptr, D, D, F, RNAMSM_ACT_NONE, 1.f, 0);
        }
        if (fold)
            FWD(lin_normed(l, 2, nullptr, nullptr, 0, hidden, F, T, F, RNAMSM_ACT_GELU_ERF, 1.f, 0));
        else
            FWD(linear(l, 4, xn, D, W[RNAMSM_WL_FC1_W], W[RNAMSM_WL_FC1_B], nullptr, 0, hidden, F, F, D, RNAMSM_ACT_GELU_ERF, 1.f, 0, nullptr));
        return dtype == RNAMSM_F32 ? res_linear(hidden, F, W[RNAMSM_WL_FC2_W], W[RNAMSM_WL_FC2_B], T, F)
                                   : linear(l, 5, hidden, F, W[RNAMSM_WL_FC2_W], W[RNAMSM_WL_FC2_B], x, D, x, D, D, F, RNAMSM_ACT_NONE, 1.f, 0, nullptr);
    };
    bool finished = false;
    FWD(run_layers(st, weights, NL, finished));
    if (finished) return RNAMSM_OK;
    FWD(rnamsm_layernorm(x, G[RNAMSM_W_LN_AFTER_G], G[RNAMSM_W_LN_AFTER_B], repr, T, D, d.ln_eps, stream));
    FWD(rnamsm::pack_outputs_batched(repr, row_attn, emb, atp, C, D, NL, H, 1, 0, 0, err_flag, static_cast<hipStream_t>(stream)));
    return RNAMSM_OK;
}

// ---- B same-shape MSAs (ragged ones padded to one shape with <pad>) through one set of token-parallel launches (exact path).
// Why: below ~4 k tokens a forward costs 5.5-6 ms whatever the alignment holds -- each of its ~140 dependent launches lasts a
// block's serial time while most CUs idle.  LayerNorm, the six Linear GEMMs of a layer and the final LayerNorm
// are per token: for a batch [B, R, C] they run ONCE over the B*R*C tokens (MSA-major, which is also the layout of the
// representation the caller gets back); the attention kernels, which couple the tokens of one MSA, take the MSA
// index from gridDim.y (one launch each for the whole batch); only the embedding (its row-position table is per MSA) and
// the output packing are launched per MSA.  Same kernels, same per-element
// arithmetic as rnamsm_forward on each MSA alone -- up to the two shape-dependent choices that change the rounding (split-K
// of fc2 and the folded LayerNorm are decided by the batch's token count).
namespace {
struct BatchLayout {
    size_t x, xn, wide, part, rowsum, stats, splitk, mask, qscale, pplanes, total;
};
BatchLayout make_batch_layout(const rnamsm_model_dims& d, int B, int R, int C) {
    const size_t T = (size_t)B * R * C, D = d.embed_dim;
    BatchLayout l;
    size_t off = 0;
    l.x = off;      off += align256(T * D * 4);
    l.xn = off;     off += align256(T * D * 4);
    l.wide = off;   off += align256(T * (size_t)(4 * D > (size_t)d.ffn_dim ? 4 * D : d.ffn_dim) * 4);
    {   // every MSA's logit slabs, for either arithmetic
        const size_t a = rnamsm_row_logits_workspace_bytes(R, C, d.num_heads), b16 = rnamsm_row_logits16_workspace_bytes(R, C, d.num_heads);
        l.part = off;   off += align256((size_t)B * (a > b16 ? a : b16));
    }
    l.rowsum = off; off += align256(T * (D / 32) * 2 * sizeof(float));
    l.stats = off;  off += align256(T * 2 * sizeof(float));
    l.splitk = off; off += align256(rnamsm::splitk_workspace_floats((int64_t)T, (int)D, d.ffn_dim) * sizeof(float));
    l.mask = off;   off += align256(T);              // padding mask uint8 [B, R, C]
    l.qscale = off; off += align256(T * sizeof(float));   // ragged batches: per-token q scale
    l.pplanes = off; off += align256((size_t)B * d.num_heads * C * (size_t)((C + 63) / 64 * 64) * 4);   // 16-bit modes: P hi + lo planes of every MSA
    l.total = off;
    return l;
}
}  // namespace

extern "C" size_t rnamsm_forward_batch_workspace_bytes(const rnamsm_model_dims* dims, int B, int R, int C) {
    if (!dims || B <= 0 || R <= 0 || C <= 0) return 0;
    return make_batch_layout(*dims, B, R, C).total;
}

extern "C" int rnamsm_forward_batch(const rnamsm_model_dims* dims, const float* const* weights, const int64_t* tokens, int B,
                                    int R, int C, void* workspace, size_t workspace_bytes, float* row_attn, float* repr,
                                    float* emb, float* atp, int* err_flag, int has_padding, const int* true_rows,
                                    const float* const* ln_folded, int dtype, const uint16_t* const* weight_planes, void* stream) {
    RNAMSM_CHECK_ARG(dims && weights && tokens && workspace && row_attn && repr && emb && atp, "forward_batch: null pointer");
    RNAMSM_CHECK_ARG(dtype >= RNAMSM_F32 && dtype <= RNAMSM_F16X3, "forward_batch: unknown dtype %d", dtype);
    RNAMSM_NO_BF16X3(dtype == RNAMSM_BF16X3_REMOVED, "forward_batch");
    ForwardScope in_flight;
    RNAMSM_CHECK_ARG(dtype == RNAMSM_F32 || (weight_planes && tuning().attn16 != 0),
                     "forward_batch: the 16-bit modes need weight_planes (and the attn16 knob on: planes end to end)");
    const rnamsm_model_dims& d = *dims;
    const int D = d.embed_dim, H = d.num_heads, F = d.ffn_dim, NL = d.num_layers;
    RNAMSM_CHECK_ARG(D > 0 && H > 0 && D == H * 64, "forward_batch: embed_dim must be num_heads * 64 (D=%d H=%d)", D, H);
    RNAMSM_CHECK_ARG(D % 128 == 0 && F % 128 == 0 && NL > 0, "forward_batch: embed_dim and ffn_dim must be multiples of 128");
    RNAMSM_CHECK_ARG(d.row_pos_dim == 0 || d.row_pos_dim == 1 || d.row_pos_dim == D, "forward_batch: row_pos_dim must be 0, 1 or embed_dim (got %d)", d.row_pos_dim);
    RNAMSM_CHECK_ARG(B >= 1 && C >= 2 && R >= 1, "forward_batch: need B >= 1, R >= 1 and C >= 2 (got B=%d R=%d C=%d)", B, R, C);
    if (R > 1024)   // model.py:355-359
        return fail(RNAMSM_ERR_INVALID,
                    "Using model with MSA position embedding trained on maximum MSA depth of 1024, but received %d alignments.", R);
    RNAMSM_CHECK_ARG(C <= d.num_positions - d.pad_idx - 1, "forward_batch: C=%d exceeds the positional table", C);
    const int64_t Tm = (int64_t)R * C, T = (int64_t)B * Tm;          // tokens of one MSA / of the batch
    RNAMSM_CHECK_ARG(T <= INT32_MAX, "forward_batch: %lld tokens exceed the GEMM row range", (long long)T);
    const BatchLayout lay = make_batch_layout(d, B, R, C);
    RNAMSM_CHECK_ARG(workspace_bytes >= lay.total, "forward_batch: workspace too small (%zu < %zu)", workspace_bytes, lay.total);
    RNAMSM_CHECK_ARG(aligned16(workspace), "forward_batch: workspace must be 16-byte aligned");
    hipStream_t hs = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    float* x = reinterpret_cast<float*>(ws + lay.x);
    float* xn = reinterpret_cast<float*>(ws + lay.xn);
    float* wide = reinterpret_cast<float*>(ws + lay.wide);
    float* part = reinterpret_cast<float*>(ws + lay.part);
    float* rowsum = reinterpret_cast<float*>(ws + lay.rowsum);
    float* stats = reinterpret_cast<float*>(ws + lay.stats);
    float* splitk = reinterpret_cast<float*>(ws + lay.splitk);
    const int64_t ldq = 3 * (int64_t)D;
    float* qkv = wide;                 // [T, 3D]
    float* ctx = wide + T * ldq;       // [T, D]
    float* hidden = wide;              // [T, F]
    const float* const* G = weights;
    const int f32 = RNAMSM_F32;
    const int nsplit = rnamsm_row_logits_nsplit(R, C, H);
    BigRowsScope big_rows_scope(dtype == RNAMSM_BF16);       // (common.h)
    const int fold_mode = tuning().ln_fold;
    // statistics from the producers only; a ragged batch (true_rows) keeps the LayerNorm launches: its QKV GEMM carries the
    // per-token q factor in the epilogue slot the fold would need
    // (by the MEMBER's token count, as its own forward decides: the batch must not change an alignment's rounding)
    const bool fold = ln_folded && !has_padding && !true_rows && (fold_mode >= 2 || (fold_mode == 1 && Tm >= LN_FOLD_MIN_TOKENS));
    const bool fold_sums = fold && fold_mode != 2;           // mode 2 (every GEMM sums its own rows): as in rnamsm_forward
    // f2: the batch contains <pad> (ragged MSAs padded to one shape): the reference's direct-path mask semantics as in
    // rnamsm_forward -- zeroed embeddings (K0) and q (QKV epilogue) at padded tokens, -10000 on keys whose first-row token is
    // <pad> (tied rows) and on padded keys (columns); every MSA reads its own [R, C] slice of the mask
    uint8_t* mask = nullptr;
    if (has_padding) {
        mask = reinterpret_cast<uint8_t*>(ws + lay.mask);
        FWD(rnamsm_pad_mask(tokens, mask, T, d.pad_idx, stream));
    }
    // true_rows (device int32 [B], ragged batches): every MSA's tied logits are scaled by ITS depth, so that a padded element
    // comes out as its unpadded forward would (see ragged_row_scale); the q scaling then happens per token after the GEMM
    const float col_scale = 1.0f / sqrtf(64.0f);
    // exact path, no padding, one common shape: the arithmetic of rnamsm_forward on each member (q carries dh^-1/2, K5 applies
    // 1/sqrt(R)) -- every member's outputs are the bits of its own forward
    const bool canon = dtype == RNAMSM_F32 && !has_padding && !true_rows;
    const float row_scale = (true_rows || canon) ? col_scale : col_scale / sqrtf((float)R);
    float* qscale = reinterpret_cast<float*>(ws + lay.qscale);
    if (true_rows) FWD(rnamsm::ragged_row_scale(tokens, d.pad_idx, true_rows, qscale, T, Tm, hs));

    if (dtype != RNAMSM_F32) {
        // ---- 16-bit modes: the plane data flow of rnamsm_forward (LayerNorm and the QKV / fc1 epilogues write 16-bit planes, every
        // contraction on the 16-bit matrix cores, fp32 softmax / statistics / residual stream), batched the same way: token-parallel
        // launches once over the B*R*C tokens, K4'..K7' with the MSA on gridDim.y.  q stays unscaled in the planes; the tied logits are
        // scaled on the fp32 accumulators -- by every MSA's own depth in a ragged batch (true_rows).
        const int split = dtype == RNAMSM_BF16 ? 1 : 3, fmt = dtype == RNAMSM_F16X3 ? 1 : 0;
        uint16_t* xn_hi = reinterpret_cast<uint16_t*>(xn);
        uint16_t* xn_lo = split == 3 ? xn_hi + T * D : nullptr;
        uint16_t* qkv_hi = reinterpret_cast<uint16_t*>(qkv);
        uint16_t* qkv_lo = split == 3 ? qkv_hi + T * ldq : nullptr;
        auto lo_at = [&](int64_t off) -> uint16_t* { return qkv_lo ? qkv_lo + off : nullptr; };
        uint16_t* ctx_hi = reinterpret_cast<uint16_t*>(ctx);
        uint16_t* ctx_lo = split == 3 ? ctx_hi + T * D : nullptr;
        uint16_t* hid_hi = reinterpret_cast<uint16_t*>(hidden);
        uint16_t* hid_lo = split == 3 ? hid_hi + T * (int64_t)F : nullptr;
        const int64_t ldp = (C + 63) / 64 * 64, plane_bs = (int64_t)H * C * ldp;
        uint16_t* p_hi = reinterpret_cast<uint16_t*>(ws + lay.pplanes);
        uint16_t* p_lo = split == 3 ? p_hi + (int64_t)B * plane_bs : nullptr;
        const int nsplit16 = rnamsm_row_logits16_nsplit(R, C, H, split);
        const int64_t part_bs = (int64_t)nsplit16 * H * C * C, probs_bs = (int64_t)NL * H * C * C;
        auto linear_pl = [&](int layer, int slot, const uint16_t* ahi, const uint16_t* alo, int64_t lda, const float* bias, const float* res,
                             float* out, uint16_t* ohi, uint16_t* olo, int64_t ldc, int N, int K, int act, float scale = 1.f,
                             int scale_cols = 0) -> int {
            const uint16_t* const* P = weight_planes + (size_t)layer * RNAMSM_PLANES_PER_LAYER + 2 * slot;
            return rnamsm_gemm_bf16(nullptr, lda, P[0], P[1], bias, res, D, out, ldc, T, N, K, act, scale, scale_cols, split, fmt, ahi, alo,
                                    ohi, olo, stream);
        };
        // bf16 operand formats, no padding: the column kernel takes q prescaled by dh^-0.5 log2(e) (see rnamsm_forward)
        const bool pre = fmt == 0 && !mask;
        FWD(rnamsm::embed_ln_batched(tokens, G[RNAMSM_W_EMBED_TOKENS], G[RNAMSM_W_EMBED_POSITIONS], G[RNAMSM_W_ROW_POS],
                                     G[RNAMSM_W_LN_BEFORE_G], G[RNAMSM_W_LN_BEFORE_B], x, B, R, C, D, d.vocab, d.num_positions, d.pad_idx,
                                     d.ln_eps, err_flag, hs, d.row_pos_dim));
        auto ln16 = [&](const float* g, const float* b_) -> int { return rnamsm_layernorm_split(x, g, b_, xn_hi, xn_lo, T, D, d.ln_eps, fmt, stream); };
        LayerSteps st;
        st.row_qkv = [&](int l, const float* const* W) -> int {
            FWD(ln16(W[RNAMSM_WL_ROW_LN_G], W[RNAMSM_WL_ROW_LN_B]));
            FWD(linear_pl(l, 0, xn_hi, xn_lo, D, W[RNAMSM_WL_ROW_BQKV], nullptr, nullptr, qkv_hi, qkv_lo, ldq, 3 * D, D, RNAMSM_ACT_NONE));
            if (mask) FWD(rnamsm_zero_plane_rows(qkv_hi, qkv_lo, mask, T, D, ldq, stream));             // q *= 1 - padding_mask
            return RNAMSM_OK;
        };
        st.row_attention = [&](int l, const float* const*) -> int {
            float* probs = row_attn + (int64_t)l * H * C * C;
            FWD(rnamsm::row_logits16_batched(qkv_hi, qkv_lo, qkv_hi + D, lo_at(D), ldq, part, R, C, H, row_scale, fmt, B, Tm * ldq, part_bs,
                                             true_rows, stream));
            FWD(rnamsm::softmax_rows_planes_batched(part, nsplit16, probs, p_hi, p_lo, ldp, 4096.f, H, C, mask, fmt, B, part_bs, probs_bs, Tm,
                                                    plane_bs, stream));
            return rnamsm::row_apply16_batched(p_hi, p_lo, ldp, qkv_hi + 2 * D, lo_at(2 * D), ldq, D, R, C, H, 1.f / 4096.f, ctx_hi, ctx_lo, fmt, B,
                                               plane_bs, Tm * ldq, Tm * D, stream);
        };
        st.row_out = [&](int l, const float* const* W) -> int {
            return linear_pl(l, 1, ctx_hi, ctx_lo, D, W[RNAMSM_WL_ROW_BO], x, x, nullptr, nullptr, D, D, D, RNAMSM_ACT_NONE);
        };
        st.col_qkv = [&](int l, const float* const* W) -> int {
            FWD(ln16(W[RNAMSM_WL_COL_LN_G], W[RNAMSM_WL_COL_LN_B]));
            return linear_pl(l, 2, xn_hi, xn_lo, D, W[RNAMSM_WL_COL_BQKV], nullptr, nullptr, qkv_hi, qkv_lo, ldq, 3 * D, D, RNAMSM_ACT_NONE,
                             pre ? col_scale * 1.4426950408889634f : 1.f, pre ? D : 0);
        };
        st.col_attention = [&](int, const float* const*) -> int {
            return rnamsm::col_attn16_batched(qkv_hi, qkv_lo, qkv_hi + D, lo_at(D), qkv_hi + 2 * D, lo_at(2 * D), ldq, D, R, C, H, col_scale,
                                              R > 1 ? mask : nullptr, ctx_hi, ctx_lo, fmt, B, Tm * ldq, Tm * D, Tm, stream, pre);
        };
        st.col_out = [&](int l, const float* const* W) -> int {
            return linear_pl(l, 3, ctx_hi, ctx_lo, D, W[RNAMSM_WL_COL_BO], x, x, nullptr, nullptr, D, D, D, RNAMSM_ACT_NONE);
        };
        st.ffn = [&](int l, const float* const* W) -> int {
            FWD(ln16(W[RNAMSM_WL_FFN_LN_G], W[RNAMSM_WL_FFN_LN_B]));
            FWD(linear_pl(l, 4, xn_hi, xn_lo, D, W[RNAMSM_WL_FC1_B], nullptr, nullptr, hid_hi, hid_lo, F, F, D, RNAMSM_ACT_GELU_ERF));
            return linear_pl(l, 5, hid_hi, hid_lo, F, W[RNAMSM_WL_FC2_B], x, x, nullptr, nullptr, D, D, F, RNAMSM_ACT_NONE);
        };
        bool finished = false;
        FWD(run_layers(st, weights, NL, finished));
        FWD(rnamsm_layernorm(x, G[RNAMSM_W_LN_AFTER_G], G[RNAMSM_W_LN_AFTER_B], repr, T, D, d.ln_eps, stream));
        FWD(rnamsm::pack_outputs_batched(repr, row_attn, emb, atp, C, D, NL, H, B, Tm * D, (int64_t)NL * H * C * C, err_flag, hs));
        return RNAMSM_OK;
    }

    const ExactPath ex{d, x, xn, rowsum, stats, splitk, T, fold, fold_sums, ln_folded, err_flag, stream};
    auto norm = [&](const float* g, const float* b) -> int { return ex.norm(g, b, T); };
    auto lin_normed = [&](int layer, int fslot, const float* Wf, const float* bias, float* out, int64_t ldc, int N, int act,
                          float scale, int scale_cols) -> int {
        return ex.lin_normed(layer, fslot, Wf, bias, 0, out, ldc, T, N, act, scale, scale_cols);
    };
    auto res_linear = [&](const float* A, int64_t lda, const float* Wf, const float* bias, int K) -> int {
        return ex.res_linear(A, lda, Wf, bias, T, K);
    };

    // K0 of the whole batch in one launch (the row-position table restarts with every alignment: row index mod R)
    FWD(rnamsm::embed_ln_batched(tokens, G[RNAMSM_W_EMBED_TOKENS], G[RNAMSM_W_EMBED_POSITIONS], G[RNAMSM_W_ROW_POS],
                                 G[RNAMSM_W_LN_BEFORE_G], G[RNAMSM_W_LN_BEFORE_B], x, B, R, C, D, d.vocab, d.num_positions, d.pad_idx,
                                 d.ln_eps, err_flag, hs, d.row_pos_dim));
    if (fold_sums) {
        FWD(rnamsm_row_partials(x, rowsum, T, D, stream));
        FWD(rnamsm_row_stats_from_partials(rowsum, T, T, D, d.ln_eps, stats, err_flag, stream));
    }
    LayerSteps st;
    // ---- tied row attention: projections over the batch, K4-K6 per MSA
    st.row_qkv = [&](int l, const float* const* W) -> int {
        FWD(norm(W[RNAMSM_WL_ROW_LN_G], W[RNAMSM_WL_ROW_LN_B]));
        if (true_rows)         // q = ((x Wq^T + bq) dh^-1/2) * (0 at <pad>, 1/sqrt(true depth) elsewhere), per token, in the epilogue
            return rnamsm_gemm_row_scaled(xn, D, W[RNAMSM_WL_ROW_WQKV], W[RNAMSM_WL_ROW_BQKV], qkv, ldq, T, 3 * D, D, row_scale, D, qscale,
                                          f32, stream);
        if (mask)              // q *= 1 - padding_mask (modules.py:767-772) in the epilogue
            return rnamsm_gemm_bias_act_res(xn, D, W[RNAMSM_WL_ROW_WQKV], W[RNAMSM_WL_ROW_BQKV], nullptr, 0, qkv, ldq, T, 3 * D, D,
                                            RNAMSM_ACT_NONE, row_scale, D, mask, f32, stream);
        return lin_normed(l, 0, W[RNAMSM_WL_ROW_WQKV], W[RNAMSM_WL_ROW_BQKV], qkv, ldq, 3 * D, RNAMSM_ACT_NONE, row_scale, D);
    };
    st.row_attention = [&](int l, const float* const*) -> int {
        // K4-K6 of all B MSAs in one launch each (gridDim.y = B); the maps land in row_attn [B, NL, H, C, C]
        const int64_t part_bs = (int64_t)nsplit * H * C * C, probs_bs = (int64_t)NL * H * C * C;
        float* probs = row_attn + (int64_t)l * H * C * C;
        FWD(rnamsm::row_logits_batched(qkv, qkv + D, ldq, part, R, C, H, B, Tm * ldq, part_bs, stream));
        FWD(rnamsm::softmax_rows_batched(part, nsplit, probs, H, C, B, part_bs, probs_bs, mask, Tm, stream,
                                         canon ? 1.0f / sqrtf((float)R) : 1.f));
        return rnamsm::row_apply_batched(probs, qkv + 2 * D, ldq, ctx, D, R, C, H, B, probs_bs, Tm * ldq, Tm * D, stream);
    };
    st.row_out = [&](int, const float* const* W) -> int { return res_linear(ctx, D, W[RNAMSM_WL_ROW_WO], W[RNAMSM_WL_ROW_BO], D); };
    // ---- column attention
    st.col_qkv = [&](int l, const float* const* W) -> int {
        FWD(norm(W[RNAMSM_WL_COL_LN_G], W[RNAMSM_WL_COL_LN_B]));
        return lin_normed(l, 1, W[RNAMSM_WL_COL_WQKV], W[RNAMSM_WL_COL_BQKV], qkv, ldq, 3 * D, RNAMSM_ACT_NONE, mask ? col_scale : col_scale * LOG2E, D);
    };
    st.col_attention = [&](int, const float* const*) -> int {
        return rnamsm::col_attn_batched(qkv, qkv + D, qkv + 2 * D, ldq, ctx, D, R, C, H, B, Tm * ldq, Tm * D, mask, stream, !mask);
    };
    st.col_out = [&](int, const float* const* W) -> int { return res_linear(ctx, D, W[RNAMSM_WL_COL_WO], W[RNAMSM_WL_COL_BO], D); };
    // ---- feed-forward
    st.ffn = [&](int l, const float* const* W) -> int {
        FWD(norm(W[RNAMSM_WL_FFN_LN_G], W[RNAMSM_WL_FFN_LN_B]));
        FWD(lin_normed(l, 2, W[RNAMSM_WL_FC1_W], W[RNAMSM_WL_FC1_B], hidden, F, F, RNAMSM_ACT_GELU_ERF, 1.f, 0));
        return res_linear(hidden, F, W[RNAMSM_WL_FC2_W], W[RNAMSM_WL_FC2_B], F);
    };
    bool finished = false;
    FWD(run_layers(st, weights, NL, finished));
    FWD(rnamsm_layernorm(x, G[RNAMSM_W_LN_AFTER_G], G[RNAMSM_W_LN_AFTER_B], repr, T, D, d.ln_eps, stream));
    // K10 of the whole batch in one launch (gridDim.y = alignment)
    FWD(rnamsm::pack_outputs_batched(repr, row_attn, emb, atp, C, D, NL, H, B, Tm * D, (int64_t)NL * H * C * C, err_flag, hs));
    return RNAMSM_OK;
}

// ---- B alignments of DIFFERENT shapes, token-packed (no padding, exact path) -----------------------------------------------
// The reference's real workload is many short RNAs of unlike length and depth (RNA_MSM_Inference.py:141-148 feeds them one by
// one); rnamsm_forward_batch needs one common [R, C] frame, and frames of unlike alignments hold 1.4-1.5x their true tokens.
// Here the alignments lie back to back on the token axis: tokens [T] = MSA 0's [R0, C0] row-major, then MSA 1's, ...  The
// token-parallel launches (K0's LayerNorm, the six GEMMs of a layer, the LayerNorms) run once over the T real tokens; K0, K4-K7
// and K10 take the alignment from gridDim.y and its shape / offsets from a PackedMsa descriptor (common.h).  Every alignment
// keeps the slab split of its own forward; q is scaled by dh^-1/2 in the QKV epilogue and the alignment's 1/sqrt(R) meets the
// summed tied logits in K5 (one rounding apart from rnamsm_forward, which folds both into q).  No <pad> inside a packed batch
// (no masks are built; K0 reports it: bit 3 of *err_flag); outputs are the concatenation of what rnamsm_forward returns per
// alignment: row_attn [NL,H,C_b,C_b], repr [R_b*C_b, D], emb [C_b-1, D], atp [NL*H, C_b-1, C_b-1].
#include <vector>
#include "row_split.h"
namespace {
struct PackedLayout {
    size_t x, xn, wide, part, rowsum, stats, splitk, desc, total;
};
// fills `host` (B descriptors) from shapes; returns false on a bad shape
bool make_packed(const rnamsm_model_dims& d, int B, const int* shapes, std::vector<PackedMsa>& host, PackedLayout& l, int64_t& T_out) {
    host.resize(B);
    int64_t tok = 0, part = 0, probs = 0, emb = 0, atp = 0;
    const int H = d.num_heads, NL = d.num_layers, D = d.embed_dim;
    for (int b = 0; b < B; ++b) {
        const int R = shapes[2 * b], C = shapes[2 * b + 1];
        if (R < 1 || R > 1024 || C < 2 || C > d.num_positions - d.pad_idx - 1) return false;
        const RowSplit sp = choose_row_split(R, C, H, 128, 512, ROW_LOGITS_F32_MAX_ROWS);
        PackedMsa& m = host[b];
        m.R = R; m.C = C; m.nsplit = sp.nsplit; m.rows_per_split = sp.rows_per_split;
        m.tok0 = tok; m.part_off = part; m.probs_off = probs; m.emb_off = emb; m.atp_off = atp;
        m.logit_scale = 1.0f / sqrtf((float)R);
        m.pad_ = 0;
        tok += (int64_t)R * C;
        part += (int64_t)sp.nsplit * H * C * C;
        probs += (int64_t)NL * H * C * C;
        emb += (int64_t)(C - 1) * D;
        atp += (int64_t)NL * H * (C - 1) * (C - 1);
    }
    T_out = tok;
    const size_t T = (size_t)tok;
    size_t off = 0;
    l.x = off;      off += align256(T * D * 4);
    l.xn = off;     off += align256(T * D * 4);
    l.wide = off;   off += align256(T * (size_t)(4 * D > d.ffn_dim ? 4 * D : d.ffn_dim) * 4);
    l.part = off;   off += align256((size_t)part * 4);
    l.rowsum = off; off += align256(T * (D / 32) * 2 * sizeof(float));
    l.stats = off;  off += align256(T * 2 * sizeof(float));
    l.splitk = off; off += align256(rnamsm::splitk_workspace_floats((int64_t)T, D, d.ffn_dim) * sizeof(float));
    l.desc = off;   off += align256((size_t)B * sizeof(PackedMsa));
    l.total = off;
    return true;
}
}  // namespace

extern "C" size_t rnamsm_forward_packed_workspace_bytes(const rnamsm_model_dims* dims, int B, const int* shapes) {
    if (!dims || B <= 0 || !shapes) return 0;
    std::vector<PackedMsa> host;
    PackedLayout lay;
    int64_t T;
    return make_packed(*dims, B, shapes, host, lay, T) ? lay.total : 0;
}

extern "C" int rnamsm_forward_packed(const rnamsm_model_dims* dims, const float* const* weights, const int64_t* tokens, int B,
                                     const int* shapes, void* workspace, size_t workspace_bytes, float* row_attn, float* repr,
                                     float* emb, float* atp, int* err_flag, const float* const* ln_folded, int dtype,
                                     const uint16_t* const* weight_planes, void* stream) {
    RNAMSM_CHECK_ARG(dims && weights && tokens && shapes && workspace && row_attn && repr && emb && atp, "forward_packed: null pointer");
    RNAMSM_CHECK_ARG(dtype >= RNAMSM_F32 && dtype <= RNAMSM_F16X3, "forward_packed: unknown dtype %d", dtype);
    RNAMSM_NO_BF16X3(dtype == RNAMSM_BF16X3_REMOVED, "forward_packed");
    RNAMSM_CHECK_ARG(dtype == RNAMSM_F32 || weight_planes, "forward_packed: the 16-bit modes need weight_planes");
    ForwardScope in_flight;
    const rnamsm_model_dims& d = *dims;
    const int D = d.embed_dim, H = d.num_heads, F = d.ffn_dim, NL = d.num_layers;
    RNAMSM_CHECK_ARG(D > 0 && H > 0 && D == H * 64, "forward_packed: embed_dim must be num_heads * 64 (D=%d H=%d)", D, H);
    RNAMSM_CHECK_ARG(D % 128 == 0 && F % 128 == 0 && NL > 0, "forward_packed: embed_dim and ffn_dim must be multiples of 128");
    RNAMSM_CHECK_ARG(d.row_pos_dim == 0 || d.row_pos_dim == 1 || d.row_pos_dim == D, "forward_packed: row_pos_dim must be 0, 1 or embed_dim (got %d)", d.row_pos_dim);
    RNAMSM_CHECK_ARG(B >= 1 && B <= 65535, "forward_packed: need 1 <= B <= 65535 alignments (got %d)", B);
    std::vector<PackedMsa> host;
    PackedLayout lay;
    int64_t T = 0;
    if (!make_packed(d, B, shapes, host, lay, T))
        return fail(RNAMSM_ERR_INVALID, "forward_packed: every alignment needs 1 <= R <= 1024 and 2 <= C <= %d", d.num_positions - d.pad_idx - 1);
    RNAMSM_CHECK_ARG(T <= INT32_MAX, "forward_packed: %lld tokens exceed the GEMM row range", (long long)T);
    RNAMSM_CHECK_ARG(workspace_bytes >= lay.total, "forward_packed: workspace too small (%zu < %zu)", workspace_bytes, lay.total);
    RNAMSM_CHECK_ARG(aligned16(workspace), "forward_packed: workspace must be 16-byte aligned");
    hipStream_t hs = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    float* x = reinterpret_cast<float*>(ws + lay.x);
    float* xn = reinterpret_cast<float*>(ws + lay.xn);
    float* wide = reinterpret_cast<float*>(ws + lay.wide);
    float* part = reinterpret_cast<float*>(ws + lay.part);
    float* rowsum = reinterpret_cast<float*>(ws + lay.rowsum);
    float* stats = reinterpret_cast<float*>(ws + lay.stats);
    float* splitk = reinterpret_cast<float*>(ws + lay.splitk);
    PackedMsa* desc = reinterpret_cast<PackedMsa*>(ws + lay.desc);
    const int64_t ldq = 3 * (int64_t)D;
    float* qkv = wide;                 // [T, 3D]
    float* ctx = wide + T * ldq;       // [T, D]
    float* hidden = wide;              // [T, F]
    const float* const* G = weights;
    const int fold_mode = tuning().ln_fold;
    // folded where every member's own forward would fold (>= LN_FOLD_MIN_TOKENS each): an alignment's rounding must not depend on
    // its company.  One launch set is one class: a table that MIXES the two classes under the by-shape rule (knob ln_fold = 1) is
    // refused -- it could only run unfolded, and its large members would silently differ from rnamsm_forward by a rounding (ADVICE
    // r05); the Python mirror (MSATransformer.forward_packed, plan_packed_groups) hands such a list over as two batches.
    int64_t min_member = INT64_MAX, max_member = 0;
    for (int b = 0; b < B; ++b) {
        min_member = std::min<int64_t>(min_member, (int64_t)host[b].R * host[b].C);
        max_member = std::max<int64_t>(max_member, (int64_t)host[b].R * host[b].C);
    }
    if (ln_folded && dtype == RNAMSM_F32 && fold_mode == 1 && min_member < LN_FOLD_MIN_TOKENS && max_member >= LN_FOLD_MIN_TOKENS)
        return fail(RNAMSM_ERR_INVALID, "forward_packed: the batch mixes alignments below and from %lld tokens (LayerNorm folded from there): "
                                        "pass the two classes as two batches, so that every alignment keeps the bits of its own forward",
                    (long long)LN_FOLD_MIN_TOKENS);
    const bool fold = ln_folded && (fold_mode >= 2 || (fold_mode == 1 && min_member >= LN_FOLD_MIN_TOKENS));
    const bool fold_sums = fold && fold_mode != 2;
    const float qk_scale = 1.0f / sqrtf(64.0f);
    const PackedMsa* hp = host.data();

    const ExactPath ex{d, x, xn, rowsum, stats, splitk, T, fold, fold_sums, ln_folded, err_flag, stream};
    auto norm = [&](const float* g, const float* b) -> int { return ex.norm(g, b, T); };
    auto lin_normed = [&](int layer, int fslot, const float* Wf, const float* bias, float* out, int64_t ldc, int N, int act,
                          float scale, int scale_cols) -> int {
        return ex.lin_normed(layer, fslot, Wf, bias, 0, out, ldc, T, N, act, scale, scale_cols);
    };
    auto res_linear = [&](const float* A, int64_t lda, const float* Wf, const float* bias, int K) -> int {
        return ex.res_linear(A, lda, Wf, bias, T, K);
    };

    // the attention steps of BOTH arithmetic modes of this driver: the exact descriptor kernels K4-K6 / K7 (member from gridDim.y)
    auto row_attention_packed = [&](int l, const float* const*) -> int {
        FWD(rnamsm::row_logits_packed(qkv, qkv + D, ldq, part, H, desc, hp, B, stream));
        FWD(rnamsm::softmax_rows_packed(part, row_attn, l, H, desc, hp, B, stream));
        return rnamsm::row_apply_packed(row_attn, l, qkv + 2 * D, ldq, ctx, D, H, desc, hp, B, stream);
    };
    auto col_attention_packed = [&](int, const float* const*) -> int {
        return rnamsm::col_attn_packed(qkv, qkv + D, qkv + 2 * D, ldq, ctx, D, H, desc, hp, B, stream, true);
    };

    FWD(rnamsm::packed_descriptors_upload(hp, B, desc, hs));
    FWD(rnamsm::embed_ln_packed(tokens, G[RNAMSM_W_EMBED_TOKENS], G[RNAMSM_W_EMBED_POSITIONS], G[RNAMSM_W_ROW_POS], G[RNAMSM_W_LN_BEFORE_G],
                                G[RNAMSM_W_LN_BEFORE_B], x, desc, B, T, D, d.vocab, d.num_positions, d.pad_idx, d.ln_eps, err_flag, hs,
                                d.row_pos_dim));
    if (dtype != RNAMSM_F32) {
        // ---- 16-bit modes (round 5): every Linear -- 85-93 % of a packed batch -- on the 16-bit matrix cores over the T packed
        // tokens, in the plane data flow of rnamsm_forward (LayerNorm and fc1 write planes, QKV / fc1 / fc2 stage plain copies);
        // the attention contractions stay on the descriptor-driven fp32 kernels K4-K7 of the exact path: for the small alignments a
        // packed batch is made of they are a few per cent of the time, they already take the member from gridDim.y, and their
        // results are the exact path's (the 16-bit attention kernels would need descriptors in six kernel families for that
        // few per cent).  q is scaled in the QKV epilogue (fp32 output), 1/sqrt(R_b) in K5, as on the exact path.
        BigRowsScope big_rows_scope(dtype == RNAMSM_BF16);
        const int split = dtype == RNAMSM_BF16 ? 1 : 3, fmt = dtype == RNAMSM_F16X3 ? 1 : 0;
        uint16_t* xn_hi = reinterpret_cast<uint16_t*>(xn);
        uint16_t* xn_lo = split == 3 ? xn_hi + T * D : nullptr;
        uint16_t* hid_hi = reinterpret_cast<uint16_t*>(hidden);
        uint16_t* hid_lo = split == 3 ? hid_hi + T * (int64_t)F : nullptr;
        auto planes_of = [&](int layer, int slot) { return weight_planes + (size_t)layer * RNAMSM_PLANES_PER_LAYER + 2 * slot; };
        // plane A -> fp32 out (QKV) / planes out (fc1) / fp32 residual (fc2)
        auto lin_pl = [&](int layer, int slot, const uint16_t* ahi, const uint16_t* alo, int64_t lda, const float* bias, const float* res,
                          float* out, uint16_t* ohi, uint16_t* olo, int64_t ldc, int N, int K, int act, float scale, int scale_cols) -> int {
            const uint16_t* const* P = planes_of(layer, slot);
            return rnamsm_gemm_bf16(nullptr, lda, P[0], P[1], bias, res, D, out, ldc, T, N, K, act, scale, scale_cols, split, fmt, ahi, alo,
                                    ohi, olo, stream);
        };
        // fp32 A (the attention context), split while it is staged: x += ctx W^T + b
        auto out_proj = [&](int layer, int slot, const float* bias) -> int {
            const uint16_t* const* P = planes_of(layer, slot);
            return rnamsm_gemm_bf16(ctx, D, P[0], P[1], bias, x, D, x, D, T, D, D, RNAMSM_ACT_NONE, 1.f, 0, split, fmt, nullptr, nullptr,
                                    nullptr, nullptr, stream);
        };
        auto ln16 = [&](const float* g, const float* b_) -> int { return rnamsm_layernorm_split(x, g, b_, xn_hi, xn_lo, T, D, d.ln_eps, fmt, stream); };
        LayerSteps st;
        st.row_qkv = [&](int l, const float* const* W) -> int {
            FWD(ln16(W[RNAMSM_WL_ROW_LN_G], W[RNAMSM_WL_ROW_LN_B]));
            return lin_pl(l, 0, xn_hi, xn_lo, D, W[RNAMSM_WL_ROW_BQKV], nullptr, qkv, nullptr, nullptr, ldq, 3 * D, D, RNAMSM_ACT_NONE, qk_scale, D);
        };
        st.row_attention = row_attention_packed;
        st.row_out = [&](int l, const float* const* W) -> int { return out_proj(l, 1, W[RNAMSM_WL_ROW_BO]); };
        st.col_qkv = [&](int l, const float* const* W) -> int {
            FWD(ln16(W[RNAMSM_WL_COL_LN_G], W[RNAMSM_WL_COL_LN_B]));
            return lin_pl(l, 2, xn_hi, xn_lo, D, W[RNAMSM_WL_COL_BQKV], nullptr, qkv, nullptr, nullptr, ldq, 3 * D, D, RNAMSM_ACT_NONE,
                          qk_scale * LOG2E, D);
        };
        st.col_attention = col_attention_packed;
        st.col_out = [&](int l, const float* const* W) -> int { return out_proj(l, 3, W[RNAMSM_WL_COL_BO]); };
        st.ffn = [&](int l, const float* const* W) -> int {
            FWD(ln16(W[RNAMSM_WL_FFN_LN_G], W[RNAMSM_WL_FFN_LN_B]));
            FWD(lin_pl(l, 4, xn_hi, xn_lo, D, W[RNAMSM_WL_FC1_B], nullptr, nullptr, hid_hi, hid_lo, F, F, D, RNAMSM_ACT_GELU_ERF, 1.f, 0));
            return lin_pl(l, 5, hid_hi, hid_lo, F, W[RNAMSM_WL_FC2_B], x, x, nullptr, nullptr, D, D, F, RNAMSM_ACT_NONE, 1.f, 0);
        };
        bool finished = false;
        FWD(run_layers(st, weights, NL, finished));
        FWD(rnamsm_layernorm(x, G[RNAMSM_W_LN_AFTER_G], G[RNAMSM_W_LN_AFTER_B], repr, T, D, d.ln_eps, stream));
        int max_C16 = 0;
        double out_floats16 = 0.0;
        for (int b = 0; b < B; ++b) {
            max_C16 = host[b].C > max_C16 ? host[b].C : max_C16;
            out_floats16 += (double)(host[b].C - 1) * D + (double)NL * H * (host[b].C - 1) * (host[b].C - 1);
        }
        FWD(rnamsm::pack_outputs_packed(repr, row_attn, emb, atp, desc, B, max_C16, D, NL, H, out_floats16, err_flag, hs));
        return RNAMSM_OK;
    }
    if (fold_sums) {
        FWD(rnamsm_row_partials(x, rowsum, T, D, stream));
        FWD(rnamsm_row_stats_from_partials(rowsum, T, T, D, d.ln_eps, stats, err_flag, stream));
    }
    LayerSteps st;
    // ---- tied row attention (q carries dh^-1/2; each alignment's 1/sqrt(R) is applied to its summed logits in K5)
    st.row_qkv = [&](int l, const float* const* W) -> int {
        FWD(norm(W[RNAMSM_WL_ROW_LN_G], W[RNAMSM_WL_ROW_LN_B]));
        return lin_normed(l, 0, W[RNAMSM_WL_ROW_WQKV], W[RNAMSM_WL_ROW_BQKV], qkv, ldq, 3 * D, RNAMSM_ACT_NONE, qk_scale, D);
    };
    st.row_attention = row_attention_packed;
    st.row_out = [&](int, const float* const* W) -> int { return res_linear(ctx, D, W[RNAMSM_WL_ROW_WO], W[RNAMSM_WL_ROW_BO], D); };
    // ---- column attention
    st.col_qkv = [&](int l, const float* const* W) -> int {
        FWD(norm(W[RNAMSM_WL_COL_LN_G], W[RNAMSM_WL_COL_LN_B]));
        return lin_normed(l, 1, W[RNAMSM_WL_COL_WQKV], W[RNAMSM_WL_COL_BQKV], qkv, ldq, 3 * D, RNAMSM_ACT_NONE, qk_scale * LOG2E, D);
    };
    st.col_attention = col_attention_packed;
    st.col_out = [&](int, const float* const* W) -> int { return res_linear(ctx, D, W[RNAMSM_WL_COL_WO], W[RNAMSM_WL_COL_BO], D); };
    // ---- feed-forward
    st.ffn = [&](int l, const float* const* W) -> int {
        FWD(norm(W[RNAMSM_WL_FFN_LN_G], W[RNAMSM_WL_FFN_LN_B]));
        FWD(lin_normed(l, 2, W[RNAMSM_WL_FC1_W], W[RNAMSM_WL_FC1_B], hidden, F, F, RNAMSM_ACT_GELU_ERF, 1.f, 0));
        return res_linear(hidden, F, W[RNAMSM_WL_FC2_W], W[RNAMSM_WL_FC2_B], F);
    };
    bool finished = false;
    FWD(run_layers(st, weights, NL, finished));
    FWD(rnamsm_layernorm(x, G[RNAMSM_W_LN_AFTER_G], G[RNAMSM_W_LN_AFTER_B], repr, T, D, d.ln_eps, stream));
    int max_C = 0;
    double out_floats = 0.0;
    for (int b = 0; b < B; ++b) {
        max_C = host[b].C > max_C ? host[b].C : max_C;
        out_floats += (double)(host[b].C - 1) * D + (double)NL * H * (host[b].C - 1) * (host[b].C - 1);
    }
    FWD(rnamsm::pack_outputs_packed(repr, row_attn, emb, atp, desc, B, max_C, D, NL, H, out_floats, err_flag, hs));
    return RNAMSM_OK;
}
