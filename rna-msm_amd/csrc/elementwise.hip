// K0 / K1 / K10: the HBM-bound kernels of the path.  One wave (64 lanes) per token row, 16-B loads and stores,
// wavefront shuffles for the reductions; nothing is staged through LDS because no value is reused across lanes.
//   K0 embed_ln      model.py:349-362 + modules.py:286-300    gather 2 table rows + row scalar, LayerNorm
//   K1 layernorm     modules.py:383,387 ; model.py:396          read 4*D B, write 4*D B per token
//   K10 pack_outputs RNA_MSM_Inference.py:151-166               strip <cls>, keep MSA row 0
#include "common.h"

namespace rnamsm {

constexpr int LN_MAX_VEC = 4;     // float4 per lane -> D <= 1024

// LayerNorm of one row held as up to LN_MAX_VEC float4 per lane (biased variance, two-pass like ATen's CPU kernel).
__device__ __forceinline__ void ln_row(f32x4 (&x)[LN_MAX_VEC], int nvec, int lane, int D, float eps,
                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                       float* __restrict__ out_row) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < LN_MAX_VEC; ++e)
        if (lane + 64 * e < nvec) s += (x[e][0] + x[e][1]) + (x[e][2] + x[e][3]);
    const float mean = wave_sum(s) / (float)D;
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < LN_MAX_VEC; ++e)
        if (lane + 64 * e < nvec) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d = x[e][i] - mean;
                ss += d * d;
            }
        }
    const float rstd = rsqrtf(wave_sum(ss) / (float)D + eps);
#pragma unroll
    for (int e = 0; e < LN_MAX_VEC; ++e) {
        const int vi = lane + 64 * e;
        if (vi < nvec) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * vi);
            const f32x4 b = *reinterpret_cast<const f32x4*>(beta + 4 * vi);
            f32x4 y;
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = (x[e][i] - mean) * rstd * g[i] + b[i];
            *reinterpret_cast<f32x4*>(out_row + 4 * vi) = y;
        }
    }
}

__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* y, int64_t T, int D,
                                                        float eps) {
    const int lane = threadIdx.x & 63;
    const int nvec = D / 4;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < T; row += stride) {
        f32x4 v[LN_MAX_VEC];
#pragma unroll
        for (int e = 0; e < LN_MAX_VEC; ++e)
            if (lane + 64 * e < nvec) v[e] = *reinterpret_cast<const f32x4*>(x + row * D + 4 * (lane + 64 * e));
        ln_row(v, nvec, lane, D, eps, gamma, beta, y + row * D);
    }
}

// K1 folded into the consuming GEMM (gemm_f32.hip FOLD = 2): the row partial sums [D/32, T, 2] = (sum x, sum (x - slab mean)^2)
// per 32-column slab (slab-major), in the format the residual GEMM epilogues leave them -- for a residual stream that did not come out of
// one (the embedding, K0).  One wave per row; the 8 lanes of a slab add up on the DPP path.
__device__ __forceinline__ float sum8_dpp_e(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    return v;
}
__global__ __launch_bounds__(256) void row_partials_kernel(const float* __restrict__ x, float2* __restrict__ partials,
                                                           int64_t T, int D) {
    const int lane = threadIdx.x & 63;
    const int nvec = D / 4;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < T; row += stride) {
#pragma unroll
        for (int e = 0; e < LN_MAX_VEC; ++e) {
            const int vi = lane + 64 * e;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (vi < nvec) v = *reinterpret_cast<const f32x4*>(x + row * D + 4 * vi);
            const float ps = sum8_dpp_e((v[0] + v[1]) + (v[2] + v[3]));
            const float mb = ps * (1.f / 32.f);                          // second moment about the slab's own mean
            const float d0 = v[0] - mb, d1 = v[1] - mb, d2 = v[2] - mb, d3 = v[3] - mb;
            const float pq = sum8_dpp_e(fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, d3 * d3))));
            if ((lane & 7) == 0 && vi < nvec) partials[(int64_t)(vi / 8) * T + row] = float2{ps, pq};
        }
    }
}

// (mean, rstd) of every row from its slab partials, one thread per row (the K/32 loads of consecutive rows coalesce):
// Chan et al.'s combination -- M2 = sum of the slabs' M2 + sum of 32 (slab mean - mean)^2 -- so the variance never comes
// from a difference of large numbers.  Rows whose |mean| is more than 32x their spread are reported in *cond_flag (bit 1):
// the fold's own subtraction of mean * c[n] loses more than 5 bits on them.
__global__ __launch_bounds__(256) void row_stats_from_partials_kernel(const float2* __restrict__ partials, int64_t pld, int64_t M,
                                                                      int K, float eps, float2* __restrict__ stats,
                                                                      int* cond_flag) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= M) return;
    const int ns = K / 32;
    float a = 0.f, m2 = 0.f, mean;
    if (ns <= 32) {
        // all slabs of the row requested at once and kept (K = 768: 24 pairs): the two loops below used to wait for one load per
        // iteration, 48 L2 round trips in a row = 9.7 us per launch, 31 launches per forward; the adds keep their order
        float2 p[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) p[j] = j < ns ? partials[(int64_t)j * pld + row] : float2{0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 32; ++j)
            if (j < ns) a += p[j].x;
        mean = a / (float)K;
#pragma unroll
        for (int j = 0; j < 32; ++j)
            if (j < ns) {
                const float dm = fmaf(p[j].x, 1.f / 32.f, -mean);
                m2 += fmaf(32.f * dm, dm, p[j].y);
            }
    } else {
        for (int j = 0; j < ns; ++j) a += partials[(int64_t)j * pld + row].x;
        mean = a / (float)K;
        for (int j = 0; j < ns; ++j) {
            const float2 p = partials[(int64_t)j * pld + row];
            const float dm = fmaf(p.x, 1.f / 32.f, -mean);
            m2 += fmaf(32.f * dm, dm, p.y);
        }
    }
    const float var = m2 / (float)K;
    stats[row] = float2{mean, rsqrtf(var + eps)};
    if (cond_flag && mean * mean > 1024.f * (var + eps)) atomicOr(cond_flag, 2);
}

// K1 folded into the consuming GEMM (gemm_f32.hip FOLD): one-time weight preparation, one wave per output feature n.
//   Wg[n,k] = W[n,k] * gamma[k]   (one fp32 rounding)
//   c[n]    = sum_k Wg[n,k]       (of the ROUNDED products, in double: it has to cancel what the GEMM accumulates)
//   d[n]    = bias[n] + sum_k W[n,k] * beta[k]   (double)
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__global__ __launch_bounds__(256) void ln_fold_weights_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ Wg,
                                                              float* __restrict__ cvec, float* __restrict__ dvec, int N,
                                                              int K) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    double c = 0.0, d = 0.0;
    for (int k = lane; k < K; k += 64) {
        const float w = W[(int64_t)n * K + k];
        const float wg = w * gamma[k];
        Wg[(int64_t)n * K + k] = wg;
        c += (double)wg;
        d += (double)w * (double)beta[k];
    }
    c = wave_sum_f64(c);
    d = wave_sum_f64(d);
    if (lane == 0) {
        cvec[n] = (float)c;
        dvec[n] = (float)((bias ? (double)bias[n] : 0.0) + d);
    }
}

constexpr int EMBED_RUN = 8;
__global__ __launch_bounds__(256) void embed_ln_kernel(const int64_t* __restrict__ tokens,
                                                       const float* __restrict__ embed_tokens,
                                                       const float* __restrict__ embed_positions,
                                                       const float* __restrict__ row_pos,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ out, int R, int C, int D, int vocab,
                                                       int num_positions, int pad_idx, float eps, int* err_flag, int B,
                                                       int row_pos_ld, const PackedMsa* __restrict__ pk, int64_t packed_T) {
    // tokens [B, R, C] -> out [B*R*C, D]: the row-position table restarts with every alignment (r = global row mod R)
    const int lane = threadIdx.x & 63;
    const int nvec = D / 4;
    const int64_t T = pk ? packed_T : (int64_t)B * R * C;
    const int64_t stride = (int64_t)gridDim.x * 4;
    // A wave takes RUNS of EMBED_RUN consecutive tokens: the position of a token is a prefix count over its alignment row, and
    // inside a run it is the previous token's count + 1 bit -- one wave per token recounted its whole row prefix for every token
    // (up to C / 64 dependent load + ballot steps each: 155 us where the output write takes 64, round 4's 0.39 of the HBM rate)
    int count = 0;
    for (int64_t run = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); run * EMBED_RUN < T; run += stride)
    for (int t_ = 0; t_ < EMBED_RUN; ++t_) {
        const int64_t row = run * EMBED_RUN + t_;
        if (row >= T) break;
        int r, c;
        const int64_t* trow;
        if (pk) {
            // token-packed batch: alignments of different shapes back to back; the token's alignment = the last descriptor
            // whose first token is <= row (wave-uniform binary search over B entries)
            int lo = 0, hi = B - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (pk[mid].tok0 <= row) lo = mid; else hi = mid - 1;
            }
            const PackedMsa& m = pk[lo];
            const int64_t local = row - m.tok0;
            r = (int)(local / m.C); c = (int)(local % m.C);
            trow = tokens + m.tok0 + (int64_t)r * m.C;
        } else {
            const int64_t rg = row / C;
            r = (int)(rg % R); c = (int)(row % C);
            trow = tokens + rg * C;
        }
        // pos = cumsum(tok != pad)[c] * (tok[c] != pad) + pad   (modules.py:288-290)
        int64_t tok = trow[c];
        if (t_ == 0 || c == 0) {          // first token of the run or of an alignment row: count the prefix [0, c]
            count = 0;
            for (int base = 0; base <= c; base += 64) {
                const int cc = base + lane;
                const bool nonpad = cc <= c && trow[cc] != pad_idx;
                count += __popcll(__ballot(nonpad));
            }
        } else {                          // same alignment row as the token before (c - 1): one more position
            count += tok != pad_idx ? 1 : 0;
        }
        int pos = (tok != pad_idx) ? count + pad_idx : pad_idx;
        if (tok < 0 || tok >= vocab || pos >= num_positions) {
            if (lane == 0 && err_flag) *err_flag = 1;
            tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
            pos = pos >= num_positions ? num_positions - 1 : pos;
        }
        // msa_position_embedding: one scalar per alignment row (model.py:293-296), or -- row_pos_ld = D -- a vector per row
        // (msm/model.py:289-292); the scalar keeps its broadcast in a register
        const float rp = row_pos_ld ? 0.f : row_pos[r];
        const float* rpv = row_pos + (int64_t)r * row_pos_ld;
        f32x4 v[LN_MAX_VEC];
#pragma unroll
        for (int e = 0; e < LN_MAX_VEC; ++e) {
            const int vi = lane + 64 * e;
            if (vi < nvec) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(embed_tokens + tok * D + 4 * vi);
                const f32x4 b = *reinterpret_cast<const f32x4*>(embed_positions + (int64_t)pos * D + 4 * vi);
                const f32x4 rv = row_pos_ld ? *reinterpret_cast<const f32x4*>(rpv + 4 * vi) : f32x4{rp, rp, rp, rp};
#pragma unroll
                for (int i = 0; i < 4; ++i) v[e][i] = (a[i] + b[i]) + rv[i];     // same association as model.py:349-360
            }
        }
        // a packed batch carries no masks: <pad> inside one is reported (bit 3 of the error word), the caller reruns it framed
        if (pk && trow[c] == pad_idx && lane == 0 && err_flag) atomicOr(err_flag, 8);
        if (trow[c] == pad_idx) {      // x * (1 - padding_mask) after emb_layer_norm_before (model.py:366-367)
#pragma unroll
            for (int e = 0; e < LN_MAX_VEC; ++e)
                if (lane + 64 * e < nvec)
                    *reinterpret_cast<f32x4*>(out + row * D + 4 * (lane + 64 * e)) = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            ln_row(v, nvec, lane, D, eps, gamma, beta, out + row * D);
        }
    }
}

// f2: padding_mask = tokens.eq(pad) (model.py:346), one byte per token
__global__ __launch_bounds__(256) void pad_mask_kernel(const int64_t* __restrict__ tokens, uint8_t* __restrict__ mask,
                                                       int64_t n, int pad_idx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) mask[i] = tokens[i] == pad_idx;
}

// a10: head-averaged attention weights, attn_weights.mean(dim=0) over [H, ...] (msm/multihead_attention.py:394-397)
__global__ __launch_bounds__(256) void head_mean_kernel(const float* __restrict__ probs, float* __restrict__ out, int H,
                                                        int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = probs[i];
    for (int h = 1; h < H; ++h) s += probs[(int64_t)h * n + i];
    out[i] = s / (float)H;
}

// f2 in the 16-bit modes: q *= 1 - padding_mask (modules.py:767-772) on q planes the QKV GEMM has already written: one
// wave per flagged token row zeroes its first `ncols` halves in the hi (and lo) plane; unflagged rows are not touched.
__global__ __launch_bounds__(256) void zero_plane_rows_kernel(uint16_t* __restrict__ hi, uint16_t* __restrict__ lo,
                                                              const uint8_t* __restrict__ mask, int64_t T, int ncols,
                                                              int64_t ld) {
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T || !mask[t]) return;
    const int lane = threadIdx.x & 63;
    for (int c = lane * 4; c < ncols; c += 256) {            // 8-byte stores (ncols % 4 == 0)
        *reinterpret_cast<uint2*>(hi + t * ld + c) = uint2{0u, 0u};
        if (lo) *reinterpret_cast<uint2*>(lo + t * ld + c) = uint2{0u, 0u};
    }
}

// blockIdx.y = alignment of a batch (operands b * stride further on).  err_flag (may be null): bit 2 is set when an output
// value is not finite -- in the 16-bit modes an operand outside fp16 range surfaces as inf / NaN here, and the host reads
// this one word instead of reducing over the outputs (RNAMSM_ERR_NONFINITE).
__global__ __launch_bounds__(256) void pack_outputs_kernel(const float* __restrict__ x_final,
                                                           const float* __restrict__ probs_all,
                                                           float* __restrict__ emb, float* __restrict__ atp, int C,
                                                           int D, int64_t n_emb, int64_t n_atp, int64_t x_bstride,
                                                           int64_t probs_bstride, int* err_flag, const PackedMsa* __restrict__ pk,
                                                           int channels) {
    if (pk) {       // token-packed batch: alignment blockIdx.y's own width and offsets; channels = num_layers * H
        const PackedMsa& m = pk[blockIdx.y];
        C = m.C;
        n_emb = (int64_t)(C - 1) * D;
        n_atp = (int64_t)channels * (C - 1) * (C - 1);
        x_final += m.tok0 * D;
        probs_all += m.probs_off;
        emb += m.emb_off;
        atp += m.atp_off;
    } else {
        x_final += blockIdx.y * x_bstride;
        probs_all += blockIdx.y * probs_bstride;
        emb += blockIdx.y * n_emb;
        atp += blockIdx.y * n_atp;
    }
    const int L = C - 1;
    bool bad = false;
    // emb[c - 1, d] = x_final[row 0, c, d]: the run x_final[D, C * D) as it lies, float4 (D % 4 == 0; round 4's kernel paid two
    // 64-bit divisions per 4-byte element: 0.43 of the HBM rate)
    {
        const int64_t n4 = n_emb / 4, stride = (int64_t)gridDim.x * blockDim.x;
        const f32x4* src = reinterpret_cast<const f32x4*>(x_final + D);
        f32x4* dst = reinterpret_cast<f32x4*>(emb);
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            const f32x4 v = src[i];
            dst[i] = v;
#pragma unroll
            for (int e = 0; e < 4; ++e) bad |= !(fabsf(v[e]) <= 3.4028234663852886e38f);     // inf or NaN
        }
    }
    // atp[ch, i, :] = probs_all[ch, i + 1, 1:]: one wave per map row, 32-bit indices inside the row
    {
        const int lane = threadIdx.x & 63;
        const int64_t rows = (int64_t)(n_atp / ((int64_t)L > 0 ? L : 1)), wstride = (int64_t)gridDim.x * 4;
        for (int64_t rw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); L > 0 && rw < rows; rw += wstride) {
            const int64_t ch = rw / L;
            const int i = (int)(rw - ch * L);
            const float* src = probs_all + (ch * C + (i + 1)) * C + 1;
            float* dst = atp + rw * L;
            for (int j = lane; j < L; j += 64) {
                const float v = src[j];
                dst[j] = v;
                bad |= !(fabsf(v) <= 3.4028234663852886e38f);
            }
        }
    }
    if (err_flag && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(err_flag, 4);
}

static unsigned rows_grid(int64_t rows) {
    const int64_t blocks = (rows + 3) / 4;
    return (unsigned)(blocks < 4096 ? (blocks > 0 ? blocks : 1) : 4096);     // grid-stride beyond 16 blocks per CU
}

}  // namespace rnamsm

using namespace rnamsm;

extern "C" int rnamsm_layernorm(const float* x, const float* gamma, const float* beta, float* y, int64_t T, int D,
                                float eps, void* stream) {
    RNAMSM_CHECK_ARG(x && gamma && beta && y, "layernorm: null pointer");
    RNAMSM_CHECK_ARG(T > 0 && D > 0 && D % 4 == 0 && D <= 256 * LN_MAX_VEC, "layernorm: need D %% 4 == 0, D <= 1024 (D=%d)", D);
    RNAMSM_CHECK_ARG(aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta), "layernorm: 16-byte alignment");
    KernelTimer timer(TC_LAYERNORM, 0.0, 8.0 * T * D, static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(layernorm_kernel, dim3(rows_grid(T)), dim3(256), 0, static_cast<hipStream_t>(stream), x, gamma,
                       beta, y, T, D, eps);
    RNAMSM_CHECK_LAUNCH("layernorm");
    return RNAMSM_OK;
}

extern "C" int rnamsm_row_partials(const float* x, float* row_partials, int64_t T, int D, void* stream) {
    RNAMSM_CHECK_ARG(x && row_partials, "row_partials: null pointer");
    RNAMSM_CHECK_ARG(T > 0 && D > 0 && D % 32 == 0 && D <= 256 * LN_MAX_VEC, "row_partials: need D %% 32 == 0, D <= 1024 (D=%d)", D);
    RNAMSM_CHECK_ARG(aligned16(x) && aligned16(row_partials), "row_partials: 16-byte alignment");
    KernelTimer timer(TC_LAYERNORM, 0.0, 4.0 * T * D + 0.25 * T * D, static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(row_partials_kernel, dim3(rows_grid(T)), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                       reinterpret_cast<float2*>(row_partials), T, D);
    RNAMSM_CHECK_LAUNCH("row_partials");
    return RNAMSM_OK;
}

extern "C" int rnamsm_row_stats_from_partials(const float* row_partials, int64_t partials_ld, int64_t M, int K, float eps,
                                              float* row_stats, int* cond_flag, void* stream) {
    RNAMSM_CHECK_ARG(row_partials && row_stats, "row_stats_from_partials: null pointer");
    RNAMSM_CHECK_ARG(M > 0 && K > 0 && K % 32 == 0 && partials_ld >= M && eps >= 0.f,
                     "row_stats_from_partials: need K %% 32 == 0 and partials_ld >= M (M=%lld K=%d)", (long long)M, K);
    RNAMSM_CHECK_ARG((reinterpret_cast<uintptr_t>(row_partials) & 7u) == 0 && (reinterpret_cast<uintptr_t>(row_stats) & 7u) == 0,
                     "row_stats_from_partials: 8-byte alignment");
    hipLaunchKernelGGL(row_stats_from_partials_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), reinterpret_cast<const float2*>(row_partials), partials_ld, M, K, eps,
                       reinterpret_cast<float2*>(row_stats), cond_flag);
    RNAMSM_CHECK_LAUNCH("row_stats_from_partials");
    return RNAMSM_OK;
}

extern "C" int rnamsm_ln_fold_weights(const float* W, const float* bias, const float* gamma, const float* beta, float* Wg,
                                      float* cvec, float* dvec, int N, int K, void* stream) {
    RNAMSM_CHECK_ARG(W && gamma && beta && Wg && cvec && dvec, "ln_fold_weights: null pointer");
    RNAMSM_CHECK_ARG(N > 0 && K > 0, "ln_fold_weights: bad shape N=%d K=%d", N, K);
    hipLaunchKernelGGL(ln_fold_weights_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), W,
                       bias, gamma, beta, Wg, cvec, dvec, N, K);
    RNAMSM_CHECK_LAUNCH("ln_fold_weights");
    return RNAMSM_OK;
}

extern "C" int rnamsm_embed_ln(const int64_t* tokens, const float* embed_tokens, const float* embed_positions,
                               const float* row_pos, const float* gamma, const float* beta, float* out, int R, int C,
                               int D, int vocab, int num_positions, int pad_idx, float eps, int* err_flag,
                               void* stream) {
    RNAMSM_CHECK_ARG(tokens && embed_tokens && embed_positions && row_pos && gamma && beta && out, "embed_ln: null pointer");
    RNAMSM_CHECK_ARG(R > 0 && C > 0 && D > 0 && D % 4 == 0 && D <= 256 * LN_MAX_VEC, "embed_ln: bad shape R=%d C=%d D=%d", R, C, D);
    if (R > 1024)   // model.py:355-359
        return fail(RNAMSM_ERR_INVALID,
                    "Using model with MSA position embedding trained on maximum MSA depth of 1024, but received %d alignments.", R);
    RNAMSM_CHECK_ARG(aligned16(embed_tokens) && aligned16(embed_positions) && aligned16(out) && aligned16(gamma) && aligned16(beta),
                     "embed_ln: 16-byte alignment");
    return embed_ln_batched(tokens, embed_tokens, embed_positions, row_pos, gamma, beta, out, 1, R, C, D, vocab, num_positions,
                            pad_idx, eps, err_flag, static_cast<hipStream_t>(stream), 0);
}

extern "C" int rnamsm_embed_ln_rows(const int64_t* tokens, const float* embed_tokens, const float* embed_positions,
                                    const float* row_pos, int row_pos_dim, const float* gamma, const float* beta, float* out,
                                    int R, int C, int D, int vocab, int num_positions, int pad_idx, float eps, int* err_flag,
                                    void* stream) {
    RNAMSM_CHECK_ARG(row_pos_dim == 0 || row_pos_dim == 1 || row_pos_dim == D, "embed_ln_rows: row_pos_dim must be 0, 1 or D (got %d)", row_pos_dim);
    if (row_pos_dim <= 1)
        return rnamsm_embed_ln(tokens, embed_tokens, embed_positions, row_pos, gamma, beta, out, R, C, D, vocab, num_positions, pad_idx,
                               eps, err_flag, stream);
    RNAMSM_CHECK_ARG(tokens && embed_tokens && embed_positions && row_pos && gamma && beta && out, "embed_ln_rows: null pointer");
    RNAMSM_CHECK_ARG(R > 0 && C > 0 && D > 0 && D % 4 == 0 && D <= 256 * LN_MAX_VEC, "embed_ln_rows: bad shape R=%d C=%d D=%d", R, C, D);
    if (R > 1024)   // msm/model.py:341-345
        return fail(RNAMSM_ERR_INVALID,
                    "Using model with MSA position embedding trained on maximum MSA depth of 1024, but received %d alignments.", R);
    RNAMSM_CHECK_ARG(aligned16(embed_tokens) && aligned16(embed_positions) && aligned16(out) && aligned16(gamma) && aligned16(beta) && aligned16(row_pos),
                     "embed_ln_rows: 16-byte alignment");
    return embed_ln_batched(tokens, embed_tokens, embed_positions, row_pos, gamma, beta, out, 1, R, C, D, vocab, num_positions,
                            pad_idx, eps, err_flag, static_cast<hipStream_t>(stream), D);
}

namespace rnamsm {
int embed_ln_batched(const int64_t* tokens, const float* embed_tokens, const float* embed_positions, const float* row_pos,
                     const float* gamma, const float* beta, float* out, int B, int R, int C, int D, int vocab, int num_positions,
                     int pad_idx, float eps, int* err_flag, hipStream_t stream, int row_pos_dim) {
    const int64_t T = (int64_t)B * R * C;
    // algorithmic HBM bytes: the output rows and the token ids; the two embedding tables (3.2 MB) are L2-resident, their rows are
    // not HBM reads (counting them had given this launch a "fraction of roofline" above 1)
    KernelTimer timer(TC_EMBED, 0.0, 4.0 * T * D + 8.0 * T, stream);
    hipLaunchKernelGGL(embed_ln_kernel, dim3(rows_grid((T + EMBED_RUN - 1) / EMBED_RUN)), dim3(256), 0, stream, tokens, embed_tokens, embed_positions, row_pos,
                       gamma, beta, out, R, C, D, vocab, num_positions, pad_idx, eps, err_flag, B, row_pos_dim > 1 ? row_pos_dim : 0,
                       (const PackedMsa*)nullptr, (int64_t)0);
    RNAMSM_CHECK_LAUNCH("embed_ln");
    return RNAMSM_OK;
}
int embed_ln_packed(const int64_t* tokens, const float* embed_tokens, const float* embed_positions, const float* row_pos,
                    const float* gamma, const float* beta, float* out, const PackedMsa* pk, int B, int64_t T, int D, int vocab,
                    int num_positions, int pad_idx, float eps, int* err_flag, hipStream_t stream, int row_pos_dim) {
    KernelTimer timer(TC_EMBED, 0.0, 4.0 * T * D + 8.0 * T, stream);
    hipLaunchKernelGGL(embed_ln_kernel, dim3(rows_grid((T + EMBED_RUN - 1) / EMBED_RUN)), dim3(256), 0, stream, tokens, embed_tokens, embed_positions, row_pos,
                       gamma, beta, out, 0, 0, D, vocab, num_positions, pad_idx, eps, err_flag, B, row_pos_dim > 1 ? row_pos_dim : 0, pk, T);
    RNAMSM_CHECK_LAUNCH("embed_ln (packed)");
    return RNAMSM_OK;
}
// PackedMsa descriptors host -> device WITHOUT a host buffer that has to outlive the call: they travel as kernel arguments,
// 32 (2 KB) per launch
struct PackedChunk { PackedMsa m[32]; };
__global__ void packed_descriptors_kernel(PackedChunk chunk, int n, PackedMsa* __restrict__ dev) {
    if ((int)threadIdx.x < n) dev[threadIdx.x] = chunk.m[threadIdx.x];
}
int packed_descriptors_upload(const PackedMsa* host, int B, PackedMsa* dev, hipStream_t stream) {
    for (int b0 = 0; b0 < B; b0 += 32) {
        PackedChunk chunk;
        const int n = B - b0 < 32 ? B - b0 : 32;
        for (int i = 0; i < 32; ++i) chunk.m[i] = host[b0 + (i < n ? i : 0)];
        hipLaunchKernelGGL(packed_descriptors_kernel, dim3(1), dim3(32), 0, stream, chunk, n, dev + b0);
        RNAMSM_CHECK_LAUNCH("packed_descriptors");
    }
    return RNAMSM_OK;
}
int pack_outputs_packed(const float* x_final, const float* row_attn, float* emb, float* atp, const PackedMsa* pk, int B, int max_C, int D,
                        int num_layers, int H, double total_out_floats, int* err_flag, hipStream_t stream) {
    const int64_t L = max_C - 1;
    const int64_t blocks = (L * D + (int64_t)num_layers * H * L * L + 255) / 256;
    KernelTimer timer(TC_PACK, 0.0, 8.0 * total_out_floats, stream);
    hipLaunchKernelGGL(pack_outputs_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048), (unsigned)B), dim3(256), 0, stream, x_final,
                       row_attn, emb, atp, 0, D, (int64_t)0, (int64_t)0, (int64_t)0, (int64_t)0, err_flag, pk, num_layers * H);
    RNAMSM_CHECK_LAUNCH("pack_outputs (packed)");
    return RNAMSM_OK;
}
int pack_outputs_batched(const float* x_final, const float* probs_all, float* emb, float* atp, int C, int D, int num_layers, int H,
                         int B, int64_t x_bstride, int64_t probs_bstride, int* err_flag, hipStream_t stream) {
    const int64_t L = C - 1;
    const int64_t n_emb = L * D, n_atp = (int64_t)num_layers * H * L * L;
    const int64_t blocks = (n_emb + n_atp + 255) / 256;
    const int64_t cap = B > 1 ? 2048 : 8192;
    KernelTimer timer(TC_PACK, 0.0, 8.0 * B * (n_emb + n_atp), stream);
    hipLaunchKernelGGL(pack_outputs_kernel, dim3((unsigned)(blocks < cap ? blocks : cap), (unsigned)B), dim3(256), 0, stream, x_final,
                       probs_all, emb, atp, C, D, n_emb, n_atp, x_bstride, probs_bstride, err_flag, (const PackedMsa*)nullptr, 0);
    RNAMSM_CHECK_LAUNCH("pack_outputs");
    return RNAMSM_OK;
}
}  // namespace rnamsm

namespace rnamsm {
// Ragged batches (rnamsm_forward_batch with true_rows): the q scale of the tied row attention per token --
// 0 at <pad> (q *= 1 - padding_mask, modules.py:767-772), 1/sqrt(R_b) elsewhere, R_b = the TRUE depth of the token's MSA
// (align_scaling, modules.py:713-715, as the MSA alone would get it; the reference itself would use the padded depth).
__global__ __launch_bounds__(256) void ragged_row_scale_kernel(const int64_t* __restrict__ tokens, int pad_idx,
                                                               const int* __restrict__ true_rows, float* __restrict__ out,
                                                               int64_t n, int64_t tokens_per_msa) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int r = true_rows[t / tokens_per_msa];
    out[t] = tokens[t] == pad_idx ? 0.f : 1.0f / sqrtf((float)(r > 0 ? r : 1));
}
int ragged_row_scale(const int64_t* tokens, int pad_idx, const int* true_rows, float* out, int64_t n, int64_t tokens_per_msa,
                     hipStream_t stream) {
    hipLaunchKernelGGL(ragged_row_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, tokens, pad_idx, true_rows,
                       out, n, tokens_per_msa);
    RNAMSM_CHECK_LAUNCH("ragged_row_scale");
    return RNAMSM_OK;
}
}  // namespace rnamsm

extern "C" int rnamsm_pad_mask(const int64_t* tokens, uint8_t* mask, int64_t n, int pad_idx, void* stream) {
    RNAMSM_CHECK_ARG(tokens && mask && n > 0, "pad_mask: bad arguments");
    hipLaunchKernelGGL(pad_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       tokens, mask, n, pad_idx);
    RNAMSM_CHECK_LAUNCH("pad_mask");
    return RNAMSM_OK;
}

namespace rnamsm {
// (no __restrict__: include/rnamsm.h lets `out` alias `a` or `b`; grid-stride: the grid is capped, any n is covered)
__global__ __launch_bounds__(256) void add_kernel(const float* a, const float* b, float* out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = a[i] + b[i];
}
}  // namespace rnamsm

extern "C" int rnamsm_add(const float* a, const float* b, float* out, int64_t n, void* stream) {
    RNAMSM_CHECK_ARG(a && b && out && n > 0, "add: bad arguments");
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(rnamsm::add_kernel, dim3((unsigned)(blocks < (1 << 20) ? blocks : (1 << 20))), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a, b, out, n);
    RNAMSM_CHECK_LAUNCH("add");
    return RNAMSM_OK;
}

extern "C" int rnamsm_head_mean(const float* probs, float* out, int H, int64_t n, void* stream) {
    RNAMSM_CHECK_ARG(probs && out && H > 0 && n > 0, "head_mean: bad arguments");
    hipLaunchKernelGGL(head_mean_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       probs, out, H, n);
    RNAMSM_CHECK_LAUNCH("head_mean");
    return RNAMSM_OK;
}

extern "C" int rnamsm_zero_plane_rows(uint16_t* hi, uint16_t* lo, const uint8_t* mask, int64_t T, int ncols, int64_t ld,
                                      void* stream) {
    RNAMSM_CHECK_ARG(hi && mask, "zero_plane_rows: null pointer");
    RNAMSM_CHECK_ARG(T > 0 && ncols > 0 && ncols % 4 == 0 && ld >= ncols && ld % 4 == 0 &&
                     (reinterpret_cast<uintptr_t>(hi) & 7u) == 0 && (reinterpret_cast<uintptr_t>(lo) & 7u) == 0,
                     "zero_plane_rows: ncols and ld must be multiples of 4, planes 8-byte aligned");
    hipLaunchKernelGGL(zero_plane_rows_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       hi, lo, mask, T, ncols, ld);
    RNAMSM_CHECK_LAUNCH("zero_plane_rows");
    return RNAMSM_OK;
}

extern "C" int rnamsm_pack_outputs(const float* x_final, const float* probs_all, float* emb, float* atp, int C, int D,
                                   int num_layers, int H, void* stream) {
    RNAMSM_CHECK_ARG(x_final && probs_all && emb && atp, "pack_outputs: null pointer");
    RNAMSM_CHECK_ARG(C >= 2 && D > 0 && num_layers > 0 && H > 0, "pack_outputs: bad shape C=%d D=%d", C, D);
    // the embedding rows are copied as 16-byte vectors (x_final + D is the first copied row): D % 4 and both pointers 16-byte aligned
    RNAMSM_CHECK_ARG(D % 4 == 0, "pack_outputs: D=%d must be a multiple of 4", D);
    RNAMSM_CHECK_ARG(aligned16(x_final) && aligned16(emb), "pack_outputs: x_final and emb must be 16-byte aligned");
    return pack_outputs_batched(x_final, probs_all, emb, atp, C, D, num_layers, H, 1, 0, 0, nullptr, static_cast<hipStream_t>(stream));
}
