// Column-attention probabilities, materialised on request -- ColumnSelfAttention's second return value
// (reference modules.py:905-917 builds attn_probs [H, C, B, R, R]; :926-945 returns it; AxialTransformerLayer hands it on,
// modules.py:253-267).  The fused kernels (col_attn.hip, col_attn16.hip) never form these (1.6 GB per layer at R=256,
// C=512; the reference's only caller discards them); this kernel is the opt-in path for a caller that does
// read them.  HBM-bound by its output (H*C*R*R floats written once, q and k read through L2), no matrix cores: the
// scores are 64-term fp32 FMA chains per (query, key).
#include "common.h"
#include "half16.h"

namespace rnamsm {
namespace {

constexpr int CP_HD = 64;
constexpr int CP_WAVES = 4;
constexpr int CP_QPW = 4;                       // queries per wave
constexpr int CP_MAXT = 16;                     // ceil(1024 / 64) key tiles

// FMT -1: q / k are fp32 (ld in floats; q already carries its scale, `scale` multiplies on top);
// FMT 0 / 1: bf16 / fp16 planes, hi (+ lo if non-null), ld in halves, q unscaled.
template <int FMT>
struct Operand {
    const void* hi;
    const void* lo;
    __device__ __forceinline__ float at(int64_t idx) const {
        if constexpr (FMT < 0) {
            return static_cast<const float*>(hi)[idx];
        } else {
            typedef typename Half16<(FMT < 0 ? 0 : FMT)>::T H;
            float v = (float)static_cast<const H*>(hi)[idx];
            if (lo) v += (float)static_cast<const H*>(lo)[idx];
            return v;
        }
    }
};

// grid (C*H, ceil(R / 16)); wave w of a block takes queries [16*by + 4*w, +4) of column c, head h.
// probs[((h*C + c)*R + i)*R + j] = softmax_j( scale * q_i . k_j  [-10000 where pad_mask[j*C + c]] )
template <int FMT>
__global__ void __launch_bounds__(CP_WAVES * WAVE) col_probs_kernel(Operand<FMT> q, Operand<FMT> k, int64_t ld,
                                                                    float* __restrict__ probs, int R, int C, int H,
                                                                    const uint8_t* __restrict__ pad_mask, float scale) {
    __shared__ float qs[CP_WAVES][CP_HD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x / H, h = blockIdx.x % H;
    const int nt = (R + 63) >> 6;
    for (int qi = 0; qi < CP_QPW; ++qi) {
        const int i = (blockIdx.y * CP_WAVES + wave) * CP_QPW + qi;        // wave-uniform
        if (i >= R) break;
        qs[wave][lane] = q.at(((int64_t)i * C + c) * ld + h * CP_HD + lane) * scale;
        __builtin_amdgcn_wave_barrier();
        float s[CP_MAXT];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < CP_MAXT; ++t) {
            s[t] = -INFINITY;
            if (t < nt) {
                const int j = t * 64 + lane;
                if (j < R) {
                    const int64_t base = ((int64_t)j * C + c) * ld + h * CP_HD;
                    float acc = 0.f;
#pragma unroll 16
                    for (int d = 0; d < CP_HD; ++d) acc = fmaf(qs[wave][d], k.at(base + d), acc);
                    if (pad_mask && pad_mask[(int64_t)j * C + c]) acc = -10000.f;          // modules.py:911-915
                    s[t] = acc;
                }
                mx = fmaxf(mx, s[t]);
            }
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < CP_MAXT; ++t)
            if (t < nt) {
                s[t] = expf(s[t] - mx);                                                   // exp(-inf) = 0 beyond R
                sum += s[t];
            }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        float* out = probs + (((int64_t)h * C + c) * R + i) * R;
#pragma unroll
        for (int t = 0; t < CP_MAXT; ++t)
            if (t < nt) {
                const int j = t * 64 + lane;
                if (j < R) out[j] = s[t] * inv;
            }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int FMT>
int launch(const void* q_hi, const void* q_lo, const void* k_hi, const void* k_lo, int64_t ld, float* probs, int R, int C,
           int H, const uint8_t* pad_mask, float scale, hipStream_t s) {
    const dim3 grid((unsigned)C * H, (unsigned)((R + CP_WAVES * CP_QPW - 1) / (CP_WAVES * CP_QPW)));
    KernelTimer timer(TC_COL_ATTN, 2.0 * C * H * (double)R * R * CP_HD, 4.0 * C * H * (double)R * R, s);
    hipLaunchKernelGGL(col_probs_kernel<FMT>, grid, dim3(CP_WAVES * WAVE), 0, s, Operand<FMT>{q_hi, q_lo},
                       Operand<FMT>{k_hi, k_lo}, ld, probs, R, C, H, pad_mask, scale);
    RNAMSM_CHECK_LAUNCH("col_probs");
    return RNAMSM_OK;
}

int check_shape(const char* who, int R, int C, int H, int head_dim, int64_t ld) {
    RNAMSM_CHECK_ARG(head_dim == CP_HD, "%s: head_dim must be 64 (got %d)", who, head_dim);
    RNAMSM_CHECK_ARG(R > 0 && R <= 1024 && C > 0 && H > 0 && (int64_t)C * H <= 0x7fffffff, "%s: bad shape R=%d C=%d H=%d", who, R, C, H);
    RNAMSM_CHECK_ARG(ld >= (int64_t)H * CP_HD, "%s: ld must be >= H*64", who);
    return RNAMSM_OK;
}

}  // namespace
}  // namespace rnamsm

using namespace rnamsm;

extern "C" int rnamsm_col_attn_probs(const float* q, const float* k, int64_t ld, float* probs, int R, int C, int H,
                                     int head_dim, const uint8_t* pad_mask, float scale, int dtype, void* stream) {
    if (dtype != RNAMSM_F32) return fail(RNAMSM_ERR_UNSUPPORTED, "col_attn_probs: only RNAMSM_F32 (planes: rnamsm_col_attn_probs16)");
    RNAMSM_CHECK_ARG(q && k && probs, "col_attn_probs: null pointer");
    if (int rc = check_shape("col_attn_probs", R, C, H, head_dim, ld)) return rc;
    return launch<-1>(q, nullptr, k, nullptr, ld, probs, R, C, H, pad_mask, scale, static_cast<hipStream_t>(stream));
}

extern "C" int rnamsm_col_attn_probs16(const uint16_t* q_hi, const uint16_t* q_lo, const uint16_t* k_hi,
                                       const uint16_t* k_lo, int64_t ld, float* probs, int R, int C, int H, int head_dim,
                                       const uint8_t* pad_mask, int fmt, float scale, void* stream) {
    RNAMSM_CHECK_ARG(q_hi && k_hi && probs, "col_attn_probs16: null pointer");
    RNAMSM_CHECK_ARG((q_lo == nullptr) == (k_lo == nullptr), "col_attn_probs16: q_lo and k_lo must both be given or both be NULL");
    RNAMSM_CHECK_ARG(fmt == 0 || fmt == 1, "col_attn_probs16: fmt must be 0 (bf16) or 1 (fp16)");
    if (int rc = check_shape("col_attn_probs16", R, C, H, head_dim, ld)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return fmt == 0 ? launch<0>(q_hi, q_lo, k_hi, k_lo, ld, probs, R, C, H, pad_mask, scale, s)
                    : launch<1>(q_hi, q_lo, k_hi, k_lo, ld, probs, R, C, H, pad_mask, scale, s);
}
