// f1: contact head on the row attentions (ContactPredictionHead.forward, modules.py:344-366; symmetrize / apc,
// utils/tensor.py:98-113; called from model.py:412-414):
//   X_ch = attn[ch, 1:, 1:]            (strip <cls>; the RNA alphabet appends no <eos>)
//   S_ch = X_ch + X_ch^T
//   N_ch = S_ch - rowsum(S_ch) colsum(S_ch) / sum(S_ch)
//   contacts[i,j] = sigmoid(b + sum_ch w[ch] N_ch[i,j])
// HBM-bound: the 120 x C x C maps (126 MB at C = 512) are read twice (once for the sums, once -- plus the transposed
// tile -- for the regression); S is symmetric so colsum == rowsum.
#include "common.h"

namespace rnamsm {

// rs[ch, i] = sum_j (X[i,j] + X[j,i]),  tot[ch] = sum_i rs[ch, i];  one block per channel, T = C - 1
__global__ __launch_bounds__(256) void contact_sums_kernel(const float* __restrict__ attn, float* __restrict__ rs,
                                                           float* __restrict__ tot, int C) {
    __shared__ float red[256];
    const int ch = blockIdx.x, T = C - 1;
    const float* X = attn + (int64_t)ch * C * C;
    float total = 0.f;
    for (int i = threadIdx.x; i < T; i += 256) {
        float r = 0.f, c = 0.f;
        for (int j = 0; j < T; ++j) {
            r += X[(int64_t)(i + 1) * C + (j + 1)];      // row i of X      (lanes stride rows: L2-resident re-reads)
            c += X[(int64_t)(j + 1) * C + (i + 1)];      // column i of X   (coalesced across lanes)
        }
        const float s = r + c;
        rs[(int64_t)ch * T + i] = s;
        total += s;
    }
    red[threadIdx.x] = total;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) tot[ch] = red[0];
}

// 32x32 output tile per block; the transposed operand X[J, I] goes through LDS so both global reads are coalesced.
__global__ __launch_bounds__(256) void contact_regress_kernel(const float* __restrict__ attn,
                                                              const float* __restrict__ rs,
                                                              const float* __restrict__ tot,
                                                              const float* __restrict__ weight,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              int C, int nch) {
    __shared__ float tr[32][33];
    const int T = C - 1;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 8 rows of 32 per pass
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < nch; ++ch) {
        const float* X = attn + (int64_t)ch * C * C;
        const float w = weight[ch], inv = 1.f / tot[ch];
        const float* r = rs + (int64_t)ch * T;
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 4; ++p) {                               // tr[a][b] = X[j0 + a, i0 + b]
            const int a = ty + 8 * p, jj = j0 + a, ii = i0 + tx;
            tr[a][tx] = (jj < T && ii < T) ? X[(int64_t)(jj + 1) * C + (ii + 1)] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int a = ty + 8 * p, i = i0 + a, j = j0 + tx;
            if (i < T && j < T) {
                const float s = X[(int64_t)(i + 1) * C + (j + 1)] + tr[tx][a];
                const float avg = (r[i] * r[j]) * inv;              // a1 * a2 / a12, same order as tensor.py:107-108
                acc[p] += w * (s - avg);
            }
        }
    }
    const float b = bias[0];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int i = i0 + ty + 8 * p, j = j0 + tx;
        if (i < T && j < T) out[(int64_t)i * T + j] = 1.f / (1.f + expf(-(acc[p] + b)));
    }
}

}  // namespace rnamsm

using namespace rnamsm;

extern "C" size_t rnamsm_contact_head_workspace_bytes(int C, int nch) {
    if (C < 2 || nch <= 0) return 0;
    return ((size_t)nch * (C - 1) + nch) * sizeof(float);
}

extern "C" int rnamsm_contact_head(const float* row_attn, const float* weight, const float* bias, float* contacts,
                                   void* workspace, size_t workspace_bytes, int C, int nch, void* stream) {
    RNAMSM_CHECK_ARG(row_attn && weight && bias && contacts && workspace, "contact_head: null pointer");
    RNAMSM_CHECK_ARG(C >= 2 && nch > 0, "contact_head: bad shape C=%d nch=%d", C, nch);
    RNAMSM_CHECK_ARG(workspace_bytes >= rnamsm_contact_head_workspace_bytes(C, nch), "contact_head: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int T = C - 1;
    float* rs = static_cast<float*>(workspace);
    float* tot = rs + (size_t)nch * T;
    hipLaunchKernelGGL(contact_sums_kernel, dim3(nch), dim3(256), 0, s, row_attn, rs, tot, C);
    RNAMSM_CHECK_LAUNCH("contact_sums");
    const unsigned tiles = (T + 31) / 32;
    hipLaunchKernelGGL(contact_regress_kernel, dim3(tiles, tiles), dim3(256), 0, s, row_attn, rs, tot, weight, bias,
                       contacts, C, nch);
    RNAMSM_CHECK_LAUNCH("contact_regress");
    return RNAMSM_OK;
}
