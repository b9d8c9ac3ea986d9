/*
 * rnamsm.h -- C ABI of librnamsm_hip.so: the MI355X (gfx950) implementation of RNA-MSM's
 * axial-attention forward path.
 *
 * The reference (yikunpku/RNA-MSM) has no FFI: the path is reached through nn.Module.forward
 * calls that bottom out in ATen ops.  Each entry point below replaces one
 * group of those ATen call sites; the citation names the reference lines (relative to the
 * reference root) whose result it reproduces.  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - plain C, no torch types; every pointer is DEVICE memory owned by the caller;
 *   - one MSA per call (B = 1, as the reference CLI runs it: RNA_MSM_Inference.py:141-148);
 *     a token (r, c) of the [R rows, C columns] alignment is row t = r*C + c of a [T, D] matrix;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing
 *     synchronises, nothing allocates (scratch is caller-provided, sized by *_workspace_bytes);
 *   - return value: RNAMSM_OK or a negative rnamsm_status; rnamsm_last_error() gives the text
 *     for the calling thread;
 *   - dtype: the per-kernel entry points implement RNAMSM_F32 (exact-fp32 MFMA, v_mfma_f32_32x32x2_f32) and return
 *     RNAMSM_ERR_UNSUPPORTED otherwise; rnamsm_gemm_bf16 and rnamsm_forward's dtype select the bf16 matrix-core modes.
 */
#ifndef RNAMSM_H_
#define RNAMSM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RNAMSM_VERSION 600 /* major*10000 + minor*100 + patch; 6.0 (round 6): same 63 entry points and signatures; ten knob names of
                              * rnamsm_set_param are gone (INVALID), rnamsm_forward_packed refuses a table mixing the two LayerNorm-fold classes,
                              * rnamsm_pack_outputs requires D % 4 == 0 and 16-byte-aligned x_final / emb.  5.0: rnamsm_forward_packed takes dtype +
                              * weight_planes, dtype value 2 (bf16x3) answers UNSUPPORTED, + rnamsm_softmax_rows_scaled; 4.0: row_pos_dim, rnamsm_embed_ln_rows */

typedef enum {
    RNAMSM_OK = 0,
    RNAMSM_ERR_INVALID = -1,     /* bad shape / null pointer / misaligned argument */
    RNAMSM_ERR_UNSUPPORTED = -2, /* valid request this build does not implement */
    RNAMSM_ERR_HIP = -3          /* a HIP runtime call failed (launch, attribute) */
} rnamsm_status;

/* Arithmetic of the Linear GEMMs inside rnamsm_forward (everything else -- attention contractions, softmax,
 * LayerNorm, residual stream, outputs -- is fp32 in every mode):
 *   RNAMSM_F32     exact-fp32 MFMA (default; the parity path)
 *   RNAMSM_BF16    bf16 MFMA on bf16-rounded operands, fp32 accumulate (mixed precision; BASELINE config 4)
 *   (value 2, RNAMSM_BF16X3_REMOVED: hi/lo bf16 pairs, 3 products -- removed in round 5: the same three MFMAs per product as
 *                  F16X3 with 17 instead of 22 operand bits; every entry point answers RNAMSM_ERR_UNSUPPORTED for it)
 *   RNAMSM_F16X3   fp16 MFMA on hi/lo-split operands, 3 products, fp32 accumulate (~2^-22 operand error: fp32-grade
 *                  for the fp16-range operands this model feeds its Linear layers) */
typedef enum { RNAMSM_F32 = 0, RNAMSM_BF16 = 1, RNAMSM_BF16X3_REMOVED = 2, RNAMSM_F16X3 = 3 } rnamsm_dtype;
typedef enum { RNAMSM_ACT_NONE = 0, RNAMSM_ACT_GELU_ERF = 1 } rnamsm_act;

int rnamsm_version(void);
const char* rnamsm_last_error(void);
/* Number of visible HIP devices (0 on a CPU-only box); never initialises a context. */
int rnamsm_device_count(void);

/* K0 -- MSATransformer.forward embedding section (model.py:349-362) with
 * LearnedPositionalEmbedding.forward (modules.py:286-300):
 *   x[r,c,:] = LayerNorm( E_tok[tok[r,c]] + E_pos[pos(r,c)] + row_pos[r] ),
 *   pos(r,c) = (#non-pad tokens in tok[r,0..c]) * (tok[r,c] != pad) + pad_idx.
 * tokens int64 [R,C]; embed_tokens [V,D]; embed_positions [P,D]; row_pos [>=R] (the
 * (1,1024,1,1) msa_position_embedding); out [R*C, D].  Token ids outside [0,V) or positions
 * outside [0,P) set bit 0 of *err_flag (device int, may be NULL; value 1) and are clamped. */
int rnamsm_embed_ln(const int64_t* tokens, const float* embed_tokens, const float* embed_positions,
                    const float* row_pos, const float* gamma, const float* beta, float* out,
                    int R, int C, int D, int vocab, int num_positions, int pad_idx, float eps,
                    int* err_flag, void* stream);
/* The same with the per-channel alignment-row embedding of msm/model.py:289-292: row_pos_dim = D reads row_pos as
 * [>=R, D] (x[r,c,:] += row_pos[r,:]); row_pos_dim = 0 or 1 is rnamsm_embed_ln. */
int rnamsm_embed_ln_rows(const int64_t* tokens, const float* embed_tokens, const float* embed_positions,
                         const float* row_pos, int row_pos_dim, const float* gamma, const float* beta, float* out,
                         int R, int C, int D, int vocab, int num_positions, int pad_idx, float eps,
                         int* err_flag, void* stream);

/* K1 -- nn.LayerNorm over the last dim (modules.py:383,387; model.py:396): y = (x-mean)/sqrt(var+eps)*gamma+beta,
 * biased variance.  x, y [T, D] contiguous (y may alias x). */
int rnamsm_layernorm(const float* x, const float* gamma, const float* beta, float* y,
                     int64_t T, int D, float eps, void* stream);

/* K1 folded into the Linear that consumes it (NormalizedResidualBlock, modules.py:385-401: layer(LayerNorm(x)) with
 * layer's first op a Linear -- q/k/v projections modules.py:760-766, 794, 896-905; fc1 modules.py:424):
 *   LN(x) W^T + b  =  rstd[m] * ( sum_k x[m,k] Wg[n,k]  -  mean[m] * c[n] ) + d[n]
 *   Wg = W * gamma (per input feature k),  c[n] = sum_k Wg[n,k],  d[n] = b[n] + sum_k W[n,k] * beta[k]
 * so the GEMM reads the residual stream itself: no normalised copy of it is written or re-read (806 MB per LayerNorm at
 * M=256 L=512) and LayerNorm is no launch at all.  The statistics come from whoever wrote x: `row_partials`
 * [K/32, M, 2] (slab-major) holds, per row and 32-feature slab, the slab's sum and its sum of squares about the slab's own
 * mean; rnamsm_row_stats_from_partials combines them into (mean, rstd) per row as in Chan et al. (M2 = sum of slab M2 +
 * sum of 32 (slab mean - mean)^2), so the biased variance never comes from a difference of large numbers.  What the fold itself costs is the subtraction of
 * mean * c[n] from the accumulated x.Wg: a row whose |mean| is much larger than its spread loses log2(|mean| / spread)
 * bits there (a residual stream is nowhere near: the synthetic model's rows sit below 1).  `cond_flag` (device int, may be
 * NULL) gets bit 1 (value 2) OR-ed in when a row has mean^2 > 1024 (var + eps) -- by rnamsm_row_stats_from_partials, or by
 * rnamsm_gemm_lnfold when it sums the rows itself; rnamsm_forward passes its err_flag and
 * the Python mirror then redoes that MSA with separate LayerNorm launches.  Results agree with LayerNorm -> Linear to fp32
 * rounding (different association; against an fp64 truth the two are equally far from it:
 * tests/analysis/ln_fold_numerics.py, tests/test_gpu_fullsize.py).
 *   rnamsm_ln_fold_weights     one-time preparation: Wg [N,K], c [N], d [N] from W [N,K], bias [N] (or NULL), gamma, beta [K]
 *                              (c and d are accumulated in double; c sums the ROUNDED Wg the GEMM will multiply)
 *   rnamsm_row_partials        row_partials of x [T, D] as it lies in memory (D % 32 == 0, D <= 1024): for an x that did
 *                              not come out of rnamsm_gemm_residual_stats (the embedding)
 *   rnamsm_gemm_residual_stats Cout = A W^T + bias + residual (K8: out_proj / fc2 + the residual add, modules.py:396) and
 *                              row_partials [N/32, M, 2] of the Cout it stores; requirements of rnamsm_gemm_bias_act_res
 *   rnamsm_row_stats_from_partials  row_stats [M, 2] = (mean, rstd) of the first M rows from row_partials [K/32, partials_ld, 2]
 *   rnamsm_gemm_lnfold         Cout[m,n] = act( (rstd[m] * (sum_k X[m,k] Wg[n,k] - mean[m] c[n]) + d[n]) * (n < scale_cols ? scale : 1) )
 *                              X [M,K] row stride ldx; row_stats [M, 2] or NULL = the block sums the rows it stages itself
 *                              (self-contained, variance as E[x^2] - mean^2, ~2.6 % slower); eps = ln_eps; requirements of
 *                              rnamsm_gemm_bias_act_res, scale_cols % 4 == 0
 * partials_ld = rows per slab of the row_partials buffer (>= M: a GEMM over the first M rows of a longer stream, as the
 * outputs-only forward runs them, addresses the buffer with its own slab stride) */
int rnamsm_ln_fold_weights(const float* W, const float* bias, const float* gamma, const float* beta, float* Wg,
                           float* cvec, float* dvec, int N, int K, void* stream);
int rnamsm_row_partials(const float* x, float* row_partials, int64_t T, int D, void* stream);
int rnamsm_gemm_residual_stats(const float* A, int64_t lda, const float* W, const float* bias, const float* residual,
                               int64_t ldr, float* Cout, int64_t ldc, int64_t M, int N, int K, float* row_partials,
                               int64_t partials_ld, int dtype, void* stream);
int rnamsm_row_stats_from_partials(const float* row_partials, int64_t partials_ld, int64_t M, int K, float eps,
                                   float* row_stats, int* cond_flag, void* stream);
int rnamsm_gemm_lnfold(const float* X, int64_t ldx, const float* Wg, const float* cvec, const float* dvec,
                       float ln_eps, const float* row_stats, int* cond_flag, float* Cout, int64_t ldc, int64_t M,
                       int N, int K, int act, float scale, int scale_cols, int dtype, void* stream);

/* K2/K3/K8 -- nn.Linear with fused epilogue (modules.py:760-766, 794-799, 896-905, 923, 424-426, 396):
 *   Cout[m,n] = act( (sum_k A[m,k]*W[n,k] + bias[n]) * (n < scale_cols ? scale : 1) ) + residual[m,n]
 * A [M,K] row stride lda; W [N,K] contiguous (torch Linear layout); bias [N] or NULL;
 * residual [M,N] row stride ldr or NULL; Cout [M,N] row stride ldc (may alias residual).
 * Requires N % 128 == 0, K % 32 == 0, lda/ldr/ldc % 4 == 0 and 16-byte aligned pointers.
 * zero_rows (uint8 [M], may be NULL): rows flagged 1 get 0 in the scaled columns -- the reference's
 * `q *= 1 - padding_mask` (modules.py:767-772).
 * dtype: RNAMSM_F32 only (RNAMSM_ERR_UNSUPPORTED otherwise); the 16-bit modes of the same Linear are rnamsm_gemm_bf16. */
int rnamsm_gemm_bias_act_res(const float* A, int64_t lda, const float* W, const float* bias,
                             const float* residual, int64_t ldr, float* Cout, int64_t ldc,
                             int64_t M, int N, int K, int act, float scale, int scale_cols,
                             const uint8_t* zero_rows, int dtype, void* stream);

/* K2 with a per-row factor on the scaled columns (the general form of zero_rows):
 *   Cout[m, n] = ((A W^T + bias)[m, n] * scale) * row_factor[m]   for n < scale_cols,   (A W^T + bias)[m, n] elsewhere.
 * The QKV projection of a RAGGED batch (rnamsm_forward_batch with true_rows): a token's q is scaled by dh^-1/2 and by
 * 1/sqrt(the true depth of ITS alignment) (align_scaling, modules.py:713-715) and zeroed at <pad> (q *= 1 - padding_mask,
 * modules.py:767-772) -- row_factor (float [M], device) carries both.  No activation, no residual.  dtype: RNAMSM_F32 only. */
int rnamsm_gemm_row_scaled(const float* A, int64_t lda, const float* W, const float* bias, float* Cout, int64_t ldc, int64_t M,
                           int N, int K, float scale, int scale_cols, const float* row_factor, int dtype, void* stream);

/* Linear on the bf16 matrix cores (fp32 accumulate, fp32 activations in HBM), same epilogue as above.
 *   split = 1: operands rounded to bf16 (mixed-precision mode, BASELINE config 4);
 *   split = 3: both operands as hi + lo pairs, product = hi*hi + hi*lo + lo*hi, fp32 accumulation; opt-in fast mode, the
 *              exact-fp32 kernel stays the default.
 *   fmt   = 0: bf16 halves (split 1 only: split 3 with fmt 0, "bf16x3", was removed -> RNAMSM_ERR_UNSUPPORTED);
 *   fmt   = 1 (split 3 only): fp16 halves, "f16x3": a hi/lo fp16 pair carries ~22 mantissa bits (fp32: 24) for operands
 *              inside fp16 range (|x| < 65504).
 * W_hi / W_lo [N, K] are bf16 planes produced once by rnamsm_split_bf16 (lo = bf16(w - hi); may be NULL there
 * and here when split = 1); A is split while it is staged.  Requires N % 128 == 0, K % 64 == 0. */
int rnamsm_split_bf16(const float* src, uint16_t* hi, uint16_t* lo, int64_t n, int fmt, void* stream);
int rnamsm_gemm_bf16(const float* A, int64_t lda, const uint16_t* W_hi, const uint16_t* W_lo, const float* bias,
                     const float* residual, int64_t ldr, float* Cout, int64_t ldc, int64_t M, int N, int K,
                     int act, float scale, int scale_cols, int split, int fmt,
                     const uint16_t* A_hi, const uint16_t* A_lo, uint16_t* O_hi, uint16_t* O_lo, void* stream);
/* K1 folded in the 16-bit modes (same scheme as rnamsm_gemm_lnfold / rnamsm_gemm_residual_stats above; 256x256-tile kernels:
 * M >= 2048, N % 256 == 0, K % 64 == 0):
 *   rnamsm_gemm16_lnfold          O planes = act( (rstd[m] (sum_k X[m,k] Wg[n,k] - mean[m] c[n]) + d[n]) * colscale ): X planes hold
 *                                 the RAW residual stream, Wg planes = rnamsm_split_bf16(W * gamma), c[n] = the row sums of the
 *                                 Wg planes' values (hi + lo), d = bias + W beta; row_stats [M,2] from
 *                                 rnamsm_row_stats_from_partials.  Output as planes (the QKV / fc1 shapes).
 *   rnamsm_gemm16_residual_stats  x[m,:] += A W^T + bias in place (fp32 residual stream, out_proj / fc2), the new x written
 *                                 once more as 16-bit planes X_hi/X_lo [M, ldp] (the next rnamsm_gemm16_lnfold's operand) and
 *                                 its slab sums left in row_partials [N/32, partials_ld, 2]. */
int rnamsm_gemm16_lnfold(const uint16_t* X_hi, const uint16_t* X_lo, int64_t ldx, const uint16_t* Wg_hi,
                         const uint16_t* Wg_lo, const float* cvec, const float* dvec, const float* row_stats,
                         uint16_t* O_hi, uint16_t* O_lo, int64_t ldo, int64_t M, int N, int K, int act, float scale,
                         int scale_cols, int split, int fmt, void* stream);
int rnamsm_gemm16_residual_stats(const uint16_t* A_hi, const uint16_t* A_lo, int64_t lda, const uint16_t* W_hi,
                                 const uint16_t* W_lo, const float* bias, float* x, int64_t ldx, int64_t M, int N,
                                 int K, int split, int fmt, uint16_t* X_hi, uint16_t* X_lo, int64_t ldp,
                                 float* row_partials, int64_t partials_ld, void* stream);
/* LayerNorm (K1) whose output is written as 16-bit hi/lo planes [T, D] (lo may be NULL), the pre-split A operand of
 * rnamsm_gemm_bf16: with A_hi/A_lo given (row stride lda halves) A is ignored and no conversion runs in the GEMM;
 * with O_hi/O_lo given (fc1: GELU; QKV: no activation; never with a residual) the result is written as planes with row stride ldc for the next
 * GEMM instead of fp32 Cout. */
int rnamsm_layernorm_split(const float* x, const float* gamma, const float* beta, uint16_t* hi, uint16_t* lo,
                           int64_t T, int D, float eps, int fmt, void* stream);

/* K4 -- tied row-attention logits, RowSelfAttention.compute_attention_weights (modules.py:752-786):
 *   S[h,i,j] = sum_{r,d} q[r,i,h,d] * k[r,j,h,d]      (q already scaled by dh^-0.5/sqrt(R), K3)
 * q, k: element (r,i,h,d) at ptr[(r*C+i)*ld + h*64 + d]  (head_dim is 64).
 * The sum over rows is split into `nsplit` contiguous row ranges (rnamsm_row_logits_nsplit);
 * partial [nsplit, H, C, C] holds one slab per range and K5 adds them in range order, so the
 * result does not depend on scheduling.  This replaces the reference's chunk-and-sum
 * _batched_forward (modules.py:717-750): same sum, fixed order. */
int rnamsm_row_logits_nsplit(int R, int C, int H);
size_t rnamsm_row_logits_workspace_bytes(int R, int C, int H);
int rnamsm_row_logits(const float* q, const float* k, int64_t ld, float* partial,
                      int R, int C, int H, int head_dim, int dtype, void* stream);

/* K5 -- softmax over the last axis of the summed logits (modules.py:818 / 739):
 *   probs[h,i,:] = softmax_j( sum_s partial[s,h,i,:] ).  probs [H, C, C] is the layer's
 *   row_attentions slab (model.py:392).  key_mask (uint8 [C], may be NULL): keys flagged 1 have their summed
 *   logit replaced by -10000 first (masked_fill with padding_mask[:, 0], modules.py:781-785). */
int rnamsm_softmax_rows(const float* partial, int nsplit, float* probs, int H, int C,
                        const uint8_t* key_mask, void* stream);
/* The same with the summed logits multiplied by logit_scale (0 < logit_scale <= 1) before the mask fill and the softmax.  This is
 * where the exact path applies align_scaling's 1/sqrt(R) (modules.py:713-715) since round 5: q leaves the QKV epilogue scaled by
 * dh^-1/2 only, in rnamsm_forward, rnamsm_forward_batch and rnamsm_forward_packed alike, so that an alignment's outputs are the
 * same bits whatever batch it is computed in (the reference's loop is list-independent, RNA_MSM_Inference.py:141-166). */
int rnamsm_softmax_rows_scaled(const float* partial, int nsplit, float* probs, int H, int C, const uint8_t* key_mask,
                               float logit_scale, void* stream);

/* K4 / K5 with padding on the reference's CHUNKED path (RowSelfAttention._batched_forward, modules.py:717-750, taken when
 * R*C > max_tokens_per_msa): the rows are processed in chunks of max_rows = max(1, max_tokens_per_msa / C) (:724), every
 * chunk's logits are filled with -10000 where the chunk's OWN first row is <pad> (:727-737 slice the mask per chunk,
 * :781-785 use its row 0) and the filled slabs are added in chunk order (:738).  With padding this differs from the direct
 * path: a pad on a chunk-starting row masks that key column, and an all-padded chunk start shifts every logit by -10000.
 *   rnamsm_row_chunks            number of chunks, 0 when the reference takes the direct path (R*C <= max_tokens_per_msa)
 *   rnamsm_row_logits_chunked    rnamsm_row_logits with one partial slab per chunk: partial [nchunks, H, C, C]
 *   rnamsm_softmax_rows_chunked  probs = softmax_j( sum_c (pad_mask[c*rows_per_chunk, j] ? -10000 : partial[c,h,i,j]) );
 *                                pad_mask uint8 [R, C] (rnamsm_pad_mask). */
int rnamsm_row_chunks(int R, int C, int max_tokens_per_msa);
int rnamsm_row_logits_chunked(const float* q, const float* k, int64_t ld, float* partial, int R, int C, int H,
                              int head_dim, int rows_per_chunk, int dtype, void* stream);
int rnamsm_softmax_rows_chunked(const float* partial, int nchunks, float* probs, int H, int C,
                                const uint8_t* pad_mask, int rows_per_chunk, void* stream);

/* K6 -- RowSelfAttention.compute_attention_update's contraction (modules.py:797-798):
 *   ctx[r,i,h,:] = sum_j probs[h,i,j] * v[r,j,h,:]
 * v addressed like q/k above; ctx element (r,i,h,d) at ctx[(r*C+i)*ldc + h*64 + d].
 * With ctx_hi given the context is written instead as 16-bit hi (+ lo if non-NULL) planes with the same indexing
 * (plane_fmt 0 = bf16, 1 = fp16): the pre-split A operand of rnamsm_gemm_bf16 for the following out_proj.
 * dtype: RNAMSM_F32 only; the 16-bit modes are rnamsm_row_apply16 (and rnamsm_row_logits16 / rnamsm_softmax_rows_planes for
 * K4 / K5, whose fp32 entry points rnamsm_row_logits[_chunked] likewise take RNAMSM_F32 only). */
int rnamsm_row_apply(const float* probs, const float* v, int64_t ld, float* ctx, int64_t ldc,
                     int R, int C, int H, int head_dim, uint16_t* ctx_hi, uint16_t* ctx_lo, int plane_fmt,
                     int dtype, void* stream);

/* K7 -- ColumnSelfAttention.compute_attention_update's attention (modules.py:905-921), fused:
 *   for every column c and head h: ctx[:,c,h,:] = softmax_j( q[:,c,h,:] k[:,c,h,:]^T ) v[:,c,h,:]
 * (q already scaled by dh^-0.5).  The [H,C,R,R] probabilities are never written (the reference
 * computes and discards them).  R == 1 degenerates to ctx = v (modules.py:882-894).
 * pad_mask (uint8 [R, C], may be NULL): scores of keys flagged 1 are replaced by -10000 (modules.py:911-915).
 * ctx_hi / ctx_lo / plane_fmt: optional 16-bit plane output as for rnamsm_row_apply (not with pad_mask).
 * dtype: RNAMSM_F32 only; the 16-bit modes are rnamsm_col_attn16. */
int rnamsm_col_attn_fused(const float* q, const float* k, const float* v, int64_t ld,
                          float* ctx, int64_t ldc, int R, int C, int H, int head_dim,
                          const uint8_t* pad_mask, uint16_t* ctx_hi, uint16_t* ctx_lo, int plane_fmt,
                          int dtype, void* stream);

/* K7 restricted to the query rows [0, q_rows) of every column (keys / values: all R rows): what the LAST layer needs when
 * only alignment row 0 of the final representation is wanted (rnamsm_forward without RNAMSM_OUT_REPR).  Row i of ctx is
 * written for i < q_rows only and equals rnamsm_col_attn_fused's row i bit for bit.
 * dtype: RNAMSM_F32 only. */
int rnamsm_col_attn_fused_queries(const float* q, const float* k, const float* v, int64_t ld, float* ctx, int64_t ldc,
                                  int R, int C, int H, int head_dim, int q_rows, const uint8_t* pad_mask, int dtype,
                                  void* stream);

/* K7 on PRESCALED q (round 4): q already carries dh^-1/2 * log2(e) (the scale / scale_cols of the QKV GEMM's epilogue), so the
 * scores are in log2 units and the kernel's first pass needs no running maximum -- p = exp2(s) on the raw score, no max chain, no
 * rescale of the accumulators (softmax does not depend on the reference point and fp32 keeps its relative precision at any
 * scale); a wave (32 query rows) whose row sums leave [2^-64, 2^100] (a score outside fp32's exponent range) redoes its rows with
 * the online softmax in log2 units (round 5: per wave, not per 128-query block -- a row's arithmetic does not depend on its block mates).  No mask, fp32 context; q_rows as in rnamsm_col_attn_fused_queries (q_rows == R: all rows).  Results
 * equal rnamsm_col_attn_fused on q / log2(e) to fp32 rounding.  What rnamsm_forward / _batch / _packed call on the exact path
 * when the MSA has no padding.  Knob "col_fast" = 0 keeps the online softmax only (A/B). */
int rnamsm_col_attn_fused_prescaled(const float* q, const float* k, const float* v, int64_t ld, float* ctx, int64_t ldc,
                                    int R, int C, int H, int head_dim, int q_rows, void* stream);

/* K7's probabilities, materialised on request -- ColumnSelfAttention's second return value (modules.py:905-917 builds
 * attn_probs [H, C, B, R, R]; :926-945 returns it; AxialTransformerLayer.forward hands it on, :253-267):
 *   probs[((h*C + c)*R + i)*R + j] = softmax_j( scale * q[i,c,h,:] . k[j,c,h,:]   [-10000 where pad_mask[j*C + c]] )
 * i.e. the reference's tensor for B = 1, contiguous.  The fused kernels above never form it (H*C*R*R floats: 1.6 GB per
 * layer at R = 256, C = 512); callers that read it (rnamsm.modules.ColumnSelfAttention(return_probs=True)) pay for it
 * here.  q, k addressed like rnamsm_col_attn_fused's; the fp32 form takes q as the QKV GEMM left it (already scaled by
 * dh^-0.5: pass scale = 1), the plane form takes the 16-bit modes' unscaled q planes (scale = dh^-0.5; *_lo may be NULL
 * together; fmt 0 = bf16, 1 = fp16; ld in halves).  R == 1 gives ones, as modules.py:882-891.
 * dtype: RNAMSM_F32 only. */
int rnamsm_col_attn_probs(const float* q, const float* k, int64_t ld, float* probs, int R, int C, int H, int head_dim,
                          const uint8_t* pad_mask, float scale, int dtype, void* stream);
int rnamsm_col_attn_probs16(const uint16_t* q_hi, const uint16_t* q_lo, const uint16_t* k_hi, const uint16_t* k_lo,
                            int64_t ld, float* probs, int R, int C, int H, int head_dim, const uint8_t* pad_mask, int fmt,
                            float scale, void* stream);

/* K4' / K5' / K6' / K7' -- the attention contractions of the 16-bit modes (same reference lines as K4..K7).  Operands
 * are 16-bit planes in HBM, addressed like their fp32 counterparts (element (r,c,h,d) = plane[(r*C+c)*ld + h*64 + d],
 * ld in halves): q/k/v as written by rnamsm_gemm_bf16's plane epilogue (O_hi/O_lo over the fused [T,3D] QKV output),
 * P as written by rnamsm_softmax_rows_planes.  All *_lo NULL = bf16 operands, one MFMA per product (fmt must be 0);
 * all *_lo given = hi/lo pairs, three MFMAs per product (fmt must be 1 = f16x3, fp32-grade; fmt 0 with lo planes -- bf16x3 -- was removed).
 * Accumulation, softmax and the partial-slab sum stay fp32.  Padding masks (f2): rnamsm_zero_plane_rows on the q
 * planes, key_mask in rnamsm_softmax_rows_planes, pad_mask in rnamsm_col_attn16. */
/* Scaling is applied to fp32 accumulators, never to 16-bit operands (a q scaled by ~1e-2 or a probability ~1e-3 would
 * push its fp16 lo plane into subnormals): q arrives UNSCALED and `scale` multiplies the logits; P planes hold
 * P * plane_scale (a power of two, 4096 in rnamsm_forward) and out_scale = 1 / plane_scale undoes it. */
/* Row split of rnamsm_row_logits16 for split = 1 (hi planes only) or 3 (hi/lo pairs); it may differ from
 * rnamsm_row_logits_nsplit (large C in the hi/lo modes uses a 256x256 tile with its own split), and the size of a
 * partial buffer that fits either tiling. */
int rnamsm_row_logits16_nsplit(int R, int C, int H, int split);
size_t rnamsm_row_logits16_workspace_bytes(int R, int C, int H);
int rnamsm_row_logits16(const uint16_t* q_hi, const uint16_t* q_lo, const uint16_t* k_hi, const uint16_t* k_lo,
                        int64_t ld, float* partial, int R, int C, int H, int head_dim, float scale, int fmt,
                        void* stream);
/* rnamsm_softmax_rows that also writes P * plane_scale as planes [H*C, ldp] (ldp = C rounded up to 64, tail zeroed). */
int rnamsm_softmax_rows_planes(const float* partial, int nsplit, float* probs, uint16_t* p_hi, uint16_t* p_lo,
                               int64_t ldp, float plane_scale, int H, int C, const uint8_t* key_mask, int fmt,
                               void* stream);
/* ctx = out_scale * P v as fp32 (ctx_hi NULL) or as ctx_hi(+ctx_lo) planes in the operands' format. */
int rnamsm_row_apply16(const uint16_t* p_hi, const uint16_t* p_lo, int64_t ldp, const uint16_t* v_hi,
                       const uint16_t* v_lo, int64_t ld, float* ctx, int64_t ldc, int R, int C, int H, int head_dim,
                       float out_scale, uint16_t* ctx_hi, uint16_t* ctx_lo, int fmt, void* stream);
/* ctx = softmax_j(scale * q k^T) v per (column, head); pad_mask (uint8 [R, C], may be NULL): scaled scores of keys
 * flagged 1 are replaced by -10000 (modules.py:911-915). */
int rnamsm_col_attn16(const uint16_t* q_hi, const uint16_t* q_lo, const uint16_t* k_hi, const uint16_t* k_lo,
                      const uint16_t* v_hi, const uint16_t* v_lo, int64_t ld, float* ctx, int64_t ldc, int R, int C,
                      int H, int head_dim, float scale, const uint8_t* pad_mask, uint16_t* ctx_hi, uint16_t* ctx_lo,
                      int fmt, void* stream);
/* The same for bf16 operand formats (plain bf16: lo planes NULL, or bf16 hi/lo pairs) without a padding mask, with q PRESCALED:
 * the q planes hold q * scale * log2(e) (scale = dh^-0.5), multiplied in BEFORE the rounding to 16 bits -- rnamsm_gemm_bf16's
 * epilogue does that for the q columns (scale, scale_cols) -- so the kernel's softmax is p = exp2(q'.k) with no multiply, no
 * reference subtraction and no pre-pass; a row sum outside [2^-96, 2^96] sends the block to the online-softmax loop.  This is
 * what rnamsm_forward runs in the bf16 modes. */
int rnamsm_col_attn16_prescaled(const uint16_t* q_hi, const uint16_t* q_lo, const uint16_t* k_hi, const uint16_t* k_lo,
                                const uint16_t* v_hi, const uint16_t* v_lo, int64_t ld, float* ctx, int64_t ldc, int R,
                                int C, int H, int head_dim, uint16_t* ctx_hi, uint16_t* ctx_lo, void* stream);
/* f2 in the 16-bit modes: zero the first ncols halves of the plane rows flagged in mask (uint8 [T]) -- q *= 1 -
 * padding_mask (modules.py:767-772) applied to the q planes after the QKV GEMM. */
int rnamsm_zero_plane_rows(uint16_t* hi, uint16_t* lo, const uint8_t* mask, int64_t T, int ncols, int64_t ld, void* stream);

/* f2 -- padding_mask = tokens.eq(pad_idx) (model.py:346): mask uint8 [n]. */
int rnamsm_pad_mask(const int64_t* tokens, uint8_t* mask, int64_t n, int pad_idx, void* stream);

/* K10 -- extract_feat's output section (RNA_MSM_Inference.py:151-166):
 *   emb[c-1, :]            = x_final[row 0, c, :]                c = 1..C-1      -> [C-1, D]
 *   atp[l*H+h, i-1, j-1]   = probs_all[l, h, i, j]               i,j = 1..C-1    -> [NL*H, C-1, C-1]
 * x_final [R*C, D] is the output of emb_layer_norm_after; probs_all [NL, H, C, C].
 * The embedding rows move as 16-byte vectors: D % 4 == 0, x_final and emb 16-byte aligned (refused with RNAMSM_ERR_INVALID
 * otherwise); probs_all / atp have no alignment requirement. */
int rnamsm_pack_outputs(const float* x_final, const float* probs_all, float* emb, float* atp,
                        int C, int D, int num_layers, int H, void* stream);

/* f1 -- contact head on the stacked row attentions (ContactPredictionHead.forward, modules.py:344-366 with
 * symmetrize/apc, utils/tensor.py:98-113; model.py:412-414): row_attn [nch = L*H, C, C] (with <cls>) ->
 * contacts [C-1, C-1] = sigmoid(regression(apc(symmetrize(row_attn[:, 1:, 1:])))).  weight [nch], bias [1]. */
size_t rnamsm_contact_head_workspace_bytes(int C, int nch);
int rnamsm_contact_head(const float* row_attn, const float* weight, const float* bias, float* contacts,
                        void* workspace, size_t workspace_bytes, int C, int nch, void* stream);

/* a7 -- the residual add of NormalizedResidualBlock around a layer that is NOT one of this library's (modules.py:396,
 * `x = residual + x`; around the library's own layers the add is fused into the layer's last GEMM): out[i] = a[i] + b[i], fp32,
 * out may alias a or b. */
int rnamsm_add(const float* a, const float* b, float* out, int64_t n, void* stream);

/* a10 -- head-averaged attention weights of the generic MultiheadAttention (msm/multihead_attention.py:389-397,
 * need_weights=True without need_head_weights): out[i] = mean_h probs[h, i], probs [H, n] fp32, out [n]. */
int rnamsm_head_mean(const float* probs, float* out, int H, int64_t n, void* stream);

/* f3 -- greedy max/min mean-Hamming row sub-sampling (MSA.greedy_select, utils/align.py:128-148, via
 * select_diverse "diversity-max" / "diversity-min", :165-181): msa uint8 [N, L] (any per-character code, e.g. the raw
 * bytes or token ids), keeps row 0, returns the num_seqs selected row indices in ascending order.  Performs the
 * reference's float64 operations in the reference's order (numpy's pairwise summation of every candidate's distance
 * history), so the indices are identical to numpy's.  L < 65536, num_seqs <= 2048. */
size_t rnamsm_greedy_select_workspace_bytes(int N, int L, int num_seqs);
int rnamsm_greedy_select(const uint8_t* msa, int N, int L, int num_seqs, int minimise, int* out_indices,
                         void* workspace, size_t workspace_bytes, void* stream);

/* f3 -- sequence weights for `sample-pretrained` sub-sampling (MSA.weights, utils/align.py:250-253, consumed by
 * MSA.sample_weights, :150-163): weights[i] = 1 / #{ j : hamming(msa[i], msa[j]) / L < seqid_cutoff } as float64 (the row
 * itself counts), msa uint8 [N, L], L <= 32768.  The random draw itself stays on the host (numpy's generator is the
 * contract).  The reference compares raw character bytes; any one-to-one recoding (the tokenizer's ids for A/C/G/U/X/-)
 * gives the same distances. */
int rnamsm_msa_weights(const uint8_t* msa, int N, int L, double seqid_cutoff, double* weights, void* stream);

/* Whole forward, K0..K10 for one MSA, driven from C++ so that one call enqueues every launch
 * (MSATransformer.forward, model.py:338-416, with repr_layers=[num_layers], need_head_weights=True,
 * lm_head / contact head omitted -- their results are unused by the CLI).
 *
 * Weights are passed as a table of device pointers in the order documented for
 * rnamsm_weight_index (fused QKV is packed by the caller: Wqkv [3D, D] = q;k;v rows). */
typedef struct {
    int num_layers, embed_dim, num_heads, ffn_dim, vocab, num_positions, pad_idx;
    float ln_eps;
    /* msa_position_embedding: 0 or 1 = one scalar per alignment row, weights[RNAMSM_W_ROW_POS] = [1024] (the RNA-MSM
     * model, model.py:293-296: shape (1,1024,1,1)); embed_dim = one VECTOR per alignment row, [1024, embed_dim] (the msm/
     * variant of the shell, msm/model.py:289-292: shape (1,1024,1,emb_dim)).  A trailing field: an initialiser written for
     * ABI 3.x leaves it 0. */
    int row_pos_dim;
} rnamsm_model_dims;

enum {
    RNAMSM_W_EMBED_TOKENS = 0, RNAMSM_W_EMBED_POSITIONS, RNAMSM_W_ROW_POS,
    RNAMSM_W_LN_BEFORE_G, RNAMSM_W_LN_BEFORE_B, RNAMSM_W_LN_AFTER_G, RNAMSM_W_LN_AFTER_B,
    RNAMSM_W_GLOBAL_COUNT
};
enum { /* per layer, offset RNAMSM_W_GLOBAL_COUNT + layer * RNAMSM_W_LAYER_COUNT */
    RNAMSM_WL_ROW_LN_G = 0, RNAMSM_WL_ROW_LN_B, RNAMSM_WL_ROW_WQKV, RNAMSM_WL_ROW_BQKV,
    RNAMSM_WL_ROW_WO, RNAMSM_WL_ROW_BO,
    RNAMSM_WL_COL_LN_G, RNAMSM_WL_COL_LN_B, RNAMSM_WL_COL_WQKV, RNAMSM_WL_COL_BQKV,
    RNAMSM_WL_COL_WO, RNAMSM_WL_COL_BO,
    RNAMSM_WL_FFN_LN_G, RNAMSM_WL_FFN_LN_B, RNAMSM_WL_FC1_W, RNAMSM_WL_FC1_B,
    RNAMSM_WL_FC2_W, RNAMSM_WL_FC2_B,
    RNAMSM_W_LAYER_COUNT
};

/* max_tokens_per_msa: the reference's chunking budget (MSATransformer.max_tokens_per_msa_, model.py:418-428).  Without
 * padding chunking only re-orders sums and the value is ignored; with has_padding and R*C > max_tokens_per_msa the row
 * attention reproduces the chunked path's per-chunk mask fill (above; the partial-slab region grows to one slab per
 * chunk and, in the 16-bit modes, that MSA runs on the exact-fp32 kernels).  0 = never chunk. */
size_t rnamsm_forward_workspace_bytes(const rnamsm_model_dims* dims, int R, int C, int has_padding, int max_tokens_per_msa);
/* tokens int64 [R,C]; weights: host array of (RNAMSM_W_GLOBAL_COUNT + L*RNAMSM_W_LAYER_COUNT)
 * device pointers; workspace >= rnamsm_forward_workspace_bytes; row_attn [L,H,C,C] (full, with
 * <cls>), repr [R*C, D] (after emb_layer_norm_after), emb [C-1, D], atp [L*H, C-1, C-1]; all four are
 * required outputs.  has_padding != 0: the MSA contains <pad> tokens; the driver builds the padding mask and applies
 * the reference's mask semantics (§8 f2): zeroed embeddings and q at padded tokens, -10000 on keys whose first-row token
 * is <pad> (row attention; per row chunk when R*C > max_tokens_per_msa) and on padded keys (column attention). */
/* weight_planes (host array, may be NULL when dtype == RNAMSM_F32): for every layer 12 device pointers to bf16 planes
 * from rnamsm_split_bf16, in the order {row_wqkv, row_wo, col_wqkv, col_wo, fc1_w, fc2_w} x {hi, lo} (lo may be
 * NULL for RNAMSM_BF16). */
#define RNAMSM_PLANES_PER_LAYER 12
/* ln_folded (host array, may be NULL = separate LayerNorm launches): for every layer 9 device pointers from
 * rnamsm_ln_fold_weights, {Wg, c, d} for {row QKV [3D,D] with the row LayerNorm, column QKV with the column LayerNorm,
 * fc1 [F,D] with the FFN LayerNorm}.  Used by the exact path on MSAs without padding (knob "ln_fold"). */
#define RNAMSM_FOLDED_PER_LAYER 9
/* ln_folded16 (host array, may be NULL): the same for the 16-bit modes (planes end to end, "attn16" on, no padding): for
 * every layer 12 device pointers {Wg_hi, Wg_lo (NULL for RNAMSM_BF16), c, d} for {row QKV, column QKV, fc1}; Wg planes =
 * rnamsm_split_bf16(W * gamma) in the mode's format, c = row sums of the plane values (rnamsm_gemm16_lnfold). */
#define RNAMSM_FOLDED16_PER_LAYER 12
/* outputs: RNAMSM_OUT_REPR = the whole final representation repr [R*C, D] is wanted (MSATransformer.forward's
 * representations[num_layers]).  Without it only what extract_feat writes is produced -- emb (alignment row 0) and atp
 * (RNA_MSM_Inference.py:151-166) -- and the last layer stops computing the other rows once its tied row attention is done:
 * K and V of the last column attention still cover every row, but its queries, out_proj, the last FFN and the final
 * LayerNorm run on row 0's C tokens only (~5 % of the forward at M=256 L=512).  emb and atp are bit-identical either
 * way; repr then holds valid data in its first C rows only.  (Exact path without padding; otherwise the flag is ignored
 * and everything is computed.) */
#define RNAMSM_OUT_REPR 1
/* err_flag (device int, may be NULL) collects, OR-ed by the kernels: bit 0 a token / position index outside its table (K0,
 * clamped); bit 1 the folded LayerNorm's precondition failed for some row (K1 folded; redo with ln_folded = NULL); bit 2
 * (RNAMSM_ERR_NONFINITE) an emb / atp value is inf or NaN (K10 inspects every value it packs) -- in the 16-bit modes that is
 * how an operand outside fp16 range shows, and the caller should redo the MSA with dtype RNAMSM_F32.  One word, one read. */
#define RNAMSM_ERR_NONFINITE 4
int rnamsm_forward(const rnamsm_model_dims* dims, const float* const* weights, const int64_t* tokens,
                   int R, int C, void* workspace, size_t workspace_bytes,
                   float* row_attn, float* repr, float* emb, float* atp,
                   int* err_flag, int has_padding, int max_tokens_per_msa, int outputs, int dtype,
                   const uint16_t* const* weight_planes, const float* const* ln_folded,
                   const void* const* ln_folded16, void* stream);

/* B same-shape MSAs -- ragged ones padded to one shape with <pad> -- through one set of token-parallel launches:
 * MSATransformer.forward(tokens[B,R,C]) (model.py:338-416), exact path.  Why it exists: below ~4 k tokens a forward costs 5.5-6 ms whatever the
 * alignment holds (each of its ~140 dependent launches lasts one block's serial time on a mostly idle chip); LayerNorm, the
 * Linear GEMMs and the final LayerNorm are per token and run ONCE over the B*R*C tokens of the batch, only the embedding
 * (row positions restart per MSA) and the attention kernels (they couple the tokens of one MSA) are launched per MSA.
 * tokens int64 [B,R,C]; row_attn [B,L,H,C,C]; repr [B,R*C,D]; emb [B,C-1,D]; atp [B,L*H,C-1,C-1]; err_flag as in
 * rnamsm_forward; has_padding != 0: the batch contains <pad> and the reference's direct-path mask semantics apply (as in
 * rnamsm_forward; the chunked path's per-chunk fill is not offered here -- callers keep R*C <= max_tokens_per_msa or use
 * rnamsm_forward per MSA); ln_folded as in rnamsm_forward or NULL (ignored with padding).
 * true_rows (device int32 [B], or NULL = the reference's batch semantics): RAGGED batches -- alignments of different shapes
 * padded into one [R, C] frame (rows and columns appended) that must come out as each would ALONE.  A padded element already
 * equals its unpadded forward in everything (masked keys get probability exactly 0, padded queries are zeroed, padded values
 * never reach a real token) except one number: the reference scales the tied logits by 1/sqrt(R) of the PADDED depth
 * (align_scaling, modules.py:713-715).  With true_rows[b] = the real depth of MSA b that factor is applied per MSA, and
 * emb[b, :C_b-1], atp[b, :, :C_b-1, :C_b-1] equal rnamsm_forward's outputs on the unpadded MSA b to fp32 rounding.  Every MSA's outputs equal rnamsm_forward's on that MSA alone up
 * to fp32 rounding (bit-identical when the two shape-dependent choices agree: fc2 split-K and the folded LayerNorm are
 * decided by the batch's token count).  B*R*C must fit 31 bits.
 * dtype / weight_planes as in rnamsm_forward (round 3): the 16-bit modes run the same plane data flow -- LayerNorm-split, the
 * plane GEMMs, K4'..K7' with the MSA on gridDim.y; in a ragged batch the tied logits are scaled per MSA on the fp32
 * accumulators (q stays unscaled in the planes) -- and need the "attn16" knob on; which GEMM kernel runs depends on the
 * batch's token count, so an element agrees with its lone forward to the mode's rounding, not bit for bit. */
size_t rnamsm_forward_batch_workspace_bytes(const rnamsm_model_dims* dims, int B, int R, int C);
int rnamsm_forward_batch(const rnamsm_model_dims* dims, const float* const* weights, const int64_t* tokens, int B, int R, int C,
                         void* workspace, size_t workspace_bytes, float* row_attn, float* repr, float* emb, float* atp,
                         int* err_flag, int has_padding, const int* true_rows, const float* const* ln_folded, int dtype,
                         const uint16_t* const* weight_planes, void* stream);

/* B alignments of DIFFERENT shapes as one TOKEN-PACKED batch (round 4): no frame, no padding.  The reference
 * feeds short RNAs of unlike length and depth one by one (RNA_MSM_Inference.py:141-148, model.py:338-416 per MSA); a padded
 * frame of such alignments (rnamsm_forward_batch with true_rows) holds 1.4-1.5x their real tokens.  Here the alignments lie
 * back to back on the token axis and every output is the concatenation of what rnamsm_forward returns per alignment:
 *   tokens   int64 [T], T = sum R_b*C_b: alignment 0's [R_0, C_0] row-major, then alignment 1's, ... (device)
 *   shapes   int32 [B][2] = {R_b, C_b} on the HOST (read during the call; 1 <= R_b <= 1024, 2 <= C_b)
 *   row_attn [sum L*H*C_b*C_b]   repr [T, D]   emb [sum (C_b-1)*D]   atp [sum L*H*(C_b-1)^2]   (device, fp32)
 * The per-token launches (LayerNorm, the six GEMMs of a layer) run once over the T real tokens; K0, K4-K7 and K10 take the
 * alignment from gridDim.y and its shape / offsets from a descriptor table the call writes into the workspace.  Every
 * alignment keeps the tied-logit slab split of its own forward; q carries dh^-1/2 and the alignment's 1/sqrt(R_b)
 * (align_scaling, modules.py:713-715) multiplies its summed logits -- the arithmetic of rnamsm_forward itself since round 5 --
 * fc2 is never split and LayerNorm is folded only where every member's own forward folds it (>= 4096 tokens each; with ln_folded given
 * and the by-shape rule in force a table that mixes the two classes is REFUSED with RNAMSM_ERR_INVALID: hand it over as two
 * batches, one per class, as the Python mirror does): with
 * RNAMSM_F32 every alignment's outputs are rnamsm_forward's BIT FOR BIT (tests/test_gpu_forward.py).  A packed batch builds no
 * masks: <pad> inside it sets bit 3 (value 8) of *err_flag and the caller reruns that batch framed.
 * dtype RNAMSM_BF16 / RNAMSM_F16X3 (round 5; weight_planes as in rnamsm_forward, NULL for RNAMSM_F32): every Linear runs on the
 * 16-bit matrix cores over the T packed tokens (LayerNorm and fc1 write operand planes); the attention contractions stay on
 * the exact-fp32 descriptor kernels K4-K7 (a few per cent of a batch of small alignments).  An alignment then equals its own
 * 16-bit forward to the mode's rounding.  ln_folded as in rnamsm_forward or NULL (fp32 only).
 * rnamsm_forward_packed_workspace_bytes returns 0 for a shape outside the limits. */
#define RNAMSM_ERR_PAD_IN_PACKED 8
size_t rnamsm_forward_packed_workspace_bytes(const rnamsm_model_dims* dims, int B, const int* shapes);
int rnamsm_forward_packed(const rnamsm_model_dims* dims, const float* const* weights, const int64_t* tokens, int B,
                          const int* shapes, void* workspace, size_t workspace_bytes, float* row_attn, float* repr, float* emb,
                          float* atp, int* err_flag, const float* const* ln_folded, int dtype,
                          const uint16_t* const* weight_planes, void* stream);

/* Per-kernel timing with HIP events recorded on the launch stream (measurement aid for bench.py's
 * roofline block; adds two event records per launch while enabled, nothing when disabled).
 *   rnamsm_timing_enable(1) ... enqueue work ... (caller synchronises the stream) ... rnamsm_timing_collect()
 * collect folds every completed (start, stop) pair into per-category totals and returns the number of
 * categories; rnamsm_timing_get reads one: name, launches, total milliseconds, total algorithmic flops and bytes
 * (the figures DESIGN.md derives per kernel).  rnamsm_timing_reset clears the totals. */
int rnamsm_timing_enable(int on);
int rnamsm_timing_collect(void);
int rnamsm_timing_get(int category, const char** name, long long* launches, double* ms, double* flops, double* bytes);
/* Roofline terms of the timed launches of a category: bound_ms = sum over launches of max(executed matrix flops / the dense
 * peak of the launch's MFMA family, algorithmic bytes / the achievable HBM rate), plus both terms summed separately
 * (peaks from MI355X_MICROARCH.md: 157.3 TFLOP/s fp32 MFMA, 2.5 PFLOP/s bf16 / f16 MFMA, 6.3 TB/s HBM).  bound_ms / ms is
 * the fraction of its own roofline a kernel family reached (bench.py's `roofline` of the 16-bit modes). */
int rnamsm_timing_get_bound(int category, double* bound_ms, double* mfma_ms, double* hbm_ms);
/* Third roofline term, for the kernels whose softmax can out-weigh their contractions (the fused column attention at head_dim
 * 64): the launches' vector-ALU issue time, (plain instructions + 4 x transcendentals) per score x scores / (1024 SIMDs x
 * 32 lanes x 2.4 GHz).  bound_ms of rnamsm_timing_get_bound is the per-launch max over all three terms. */
int rnamsm_timing_get_valu_bound(int category, double* valu_ms);
void rnamsm_timing_reset(void);

/* Knobs: process-global selectors between SHIPPED kernels / arithmetic rules, for in-process A/B measurements and for the tests that
 * compare two kernels on the same input.  14 names since round 6 (round 5 had 24: the ten whose A/B was closed -- gemm16_persist,
 * gemm16_stagger, gemm16_dephase, gemm16_big_rows, gemm16_big_rows_fwd, row16_q16, row16_bk64, row_narrow_rows, gemm_flat_tiles, row_vt --
 * are constants now, the losing code paths are gone; rnamsm_set_param answers RNAMSM_ERR_INVALID for them).  rnamsm_set_param takes
 * the knob table's lock exclusively and REFUSES (RNAMSM_ERR_INVALID) while a forward driver is enqueuing on another thread.
 *   "gemm16_dma"  staging of the plane-input 16-bit GEMMs: 0 register-staged, 1 (or 2) LDS-DMA 128x128 tile,
 *                 3 (default) = LDS-DMA 256x256 tile when the problem allows, software-pipelined fragment
 *                 reads and a mid-tile barrier (K tile 64 deep for plain bf16, 32 for the hi/lo modes), 4 = 3 with
 *                 32-deep K tiles for every mode.  Speed only.
 *   "gemm16_mfma16"  plain-bf16 256x256 GEMM: 1 (default) = the 16x16x32-MFMA kernel for N > 1024 or K >= 2048, 2 = for every shape,
 *                 0 = the 32x32x16 kernel.  Results agree to fp32 rounding (the k order inside a step differs).
 *   "gemm_group"  GEMM block order (fp32 kernel and the 256x256 16-bit kernel): row panels per XCD group (0 = chosen
 *                 from the shape, default: 8 for N > 1024, else 1 -- and the fp32 kernel deals a GEMM of <= 512 tiles flat,
 *                 tile = block id; 1 = whole panels).  An XCD's panels beyond its full groups form one smaller group: no padding
 *                 groups.  Changes HBM-side traffic and speed, never results.
 *   "gemm_tile"   fp32 GEMM block tile: 0 (default) = 128x128, or 128x64 where that evens out the last round of blocks
 *                 on a small problem, or MIXED (whole rounds of 128x128 tiles, the tile positions of the last,
 *                 partly empty round cut into their two 128x64 halves); 1 = always 128x128; 2 = always 128x64; 3 = mixed
 *                 wherever a launch has both whole rounds and a tail; 4 = by shape among the two uniform tilings only.
 *                 Results are bit-identical under every tiling: each output element sums its K products in the same order.
 *   "gemm_splitk_short"  rnamsm_forward*, the K = 768 GEMMs of a lone small alignment (<= 192 tiles): 0 (default) = off, 2 / 4 = that
 *                 many K ranges with the epilogue applied by the reduction pass (A/B: no gain once the block order was fixed).
 *   "row_narrow"  fp32 rnamsm_row_logits / rnamsm_row_apply at C <= 64: 1 (default) = the LDS-free narrow kernels, 0 = the
 *                 128x128 tile kernels.  Speed only, results bit-identical.
 *   "col_small"   fp32 rnamsm_col_attn_fused at R <= 16: 1 (default) = one wave per (column, head) on v_mfma_f32_16x16x4_f32,
 *                 no LDS; 0 = the 128-query-block kernels.  Results agree to fp32 rounding.
 *   "col_fast"    rnamsm_col_attn_fused_prescaled: 1 (default) = first pass without a running maximum, the online softmax as the
 *                 fallback of a block whose row sums leave [2^-64, 2^100]; 0 = the online softmax only (results agree to rounding).
 *   "col_dma"     fp32 rnamsm_col_attn_fused: 1 = K/V chunks staged by LDS-DMA, 32-key chunks, three blocks per CU;
 *                 0 = register-staged 64-key chunks, two blocks per CU; -1 (default) = chosen from the shape.  Speed only
 *                 (the two kernels run the same arithmetic per 32-key tile; results agree to fp32 rounding).
 *   "row16_max_rows"  hi/lo modes of rnamsm_row_logits16: cap on the rows of one partial slab (default 32, 0 = none).
 *                 Shorter fp32 accumulation chains; changes results at the rounding level (and the slab count).
 *   "greedy_fused"  rnamsm_greedy_select: 1 (default) = one launch per step (one wave per row) for alignments of up to 3072
 *                 rows, three launches per step (one thread per row) above; 2 = always one; 0 = always three.  Same indices.
 *   "ln_fold"     rnamsm_forward with ln_folded given: 1 (default) = LayerNorm folded into the QKV / fc1 GEMMs, row sums
 *                 left by the out_proj / fc2 epilogues, for MSAs of R*C >= 4096 tokens (rnamsm_get_param("ln_fold_min_tokens"); below that the separate launches
 *                 are faster); 3 = for every shape, and in the 16-bit modes too (ln_folded16; measured neutral there, hence
 *                 not the default); 2 = folded, every GEMM sums the rows it stages itself; 0 = separate LayerNorm launches
 *                 (all agree to fp32 rounding, resp. to the 16-bit mode's rounding).
 *   "gemm_splitk" exact path of rnamsm_forward, fc2 (K = ffn_dim >= 2048) of MSAs with at most 64 output tiles (below ~1.4 k
 *                 tokens): 0 (default since round 5) = never; 1 = four K ranges computed side by side into partial tiles of the
 *                 workspace and added in range order with the bias and the residual (bit-identical reruns; results differ
 *                 from the unsplit GEMM at the fp32 rounding level -- and the choice follows the token count of whatever
 *                 batch the alignment is computed in, which is why it is off by default: an alignment's emb / atp are the
 *                 same bits alone, in a same-shape batch and token-packed); 2 / 4 / 8 = that many ranges whenever
 *                 tiles x ranges <= 512 (A/B).
 *   "attn16"      16-bit modes of rnamsm_forward: 1 (default) = the attention contractions also run on the 16-bit
 *                 matrix cores in the mode's operand format (K4'..K7'), 0 = they stay on the exact-fp32 kernels,
 *                 2 = as 1 but the row kernels keep 128x128 tiles for every C (A/B of the 256x256-tile kernels);
 *                 rnamsm_col_attn16 only: 4 = one 32-query block per wave for every R, 5 = the TRACKED (online-softmax) loop for
 *                 every format (A/B, and the reference the FAST loop's fallback is tested against).
 *                 The RNAMSM_F32 path is not affected by either.
 * Threading: the knobs are process-global and are read on the host while a call enqueues its launches.  Change them only between
 * calls; while any thread is inside rnamsm_forward / _forward_batch / _forward_packed, rnamsm_set_param changes nothing and returns
 * RNAMSM_ERR_INVALID (the one per-forward choice that used to be a knob write, the 16-bit GEMM tile threshold, is thread-local). */
int rnamsm_set_param(const char* name, int value);
int rnamsm_get_param(const char* name);

#ifdef __cplusplus
}
#endif
#endif /* RNAMSM_H_ */
