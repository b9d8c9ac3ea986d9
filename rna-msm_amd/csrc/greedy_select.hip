// f3: greedy max/min mean-Hamming row sub-sampling of an alignment on the device
// (MSA.greedy_select, utils/align.py:128-148, reached through select_diverse(method="diversity-max|min"),
//  utils/align.py:165-181) -- the hhfilter-free way to cut an MSA to max_seqs_per_msa rows.
//
// Reference algorithm: start from row 0; each step computes the normalised Hamming distance of every row to the LAST
// chosen row (scipy cdist 'hamming' = mismatches / L in float64), keeps the running per-row sum over the chosen rows,
// and takes argmax (argmin) of sum / step over the rows not chosen yet -- first index on ties; finally the indices are
// sorted.  The device version performs the same IEEE-754 double operations in the same order (m / L, running +=,
// / step), so the selected indices are bit-identical to numpy's, not just "equally diverse".
// Roofline: HBM/L2-bound byte compares, N*L bytes per step; two small launches per step.
#include "common.h"

namespace rnamsm {

// one wave per row: mismatches vs the last chosen row -> running sum; taken rows are skipped
__global__ __launch_bounds__(256) void greedy_dist_kernel(const uint8_t* __restrict__ msa, int N, int L,
                                                          const int* __restrict__ chosen, int step,
                                                          const uint8_t* __restrict__ taken,
                                                          double* __restrict__ dist_sum, double* __restrict__ score,
                                                          int minimise) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const int lane = threadIdx.x & 63;
    const uint8_t* a = msa + (int64_t)row * L;
    const uint8_t* b = msa + (int64_t)chosen[step - 1] * L;
    int m = 0;
    for (int c = lane; c < L; c += 64) m += a[c] != b[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m += __shfl_xor(m, off, 64);
    if (lane == 0) {
        const double d = dist_sum[row] + (double)m / (double)L;          // cdist hamming, then the column-wise sum
        dist_sum[row] = d;
        const double mean = d / (double)step;                             // .mean(0) over the chosen rows
        // argmin is done as argmax of the negated value; taken rows can never win
        score[row] = taken[row] ? -INFINITY : (minimise ? -mean : mean);
    }
}

// single block: argmax with first-index tie-break, records the pick
__global__ __launch_bounds__(1024) void greedy_pick_kernel(const double* __restrict__ score, int N, int* chosen,
                                                           int step, uint8_t* taken) {
    __shared__ double sv[1024];
    __shared__ int si[1024];
    double best = -INFINITY;
    int bi = 0x7fffffff;
    for (int n = threadIdx.x; n < N; n += 1024) {
        const double v = score[n];
        if (v > best || (v == best && n < bi)) { best = v; bi = n; }
    }
    sv[threadIdx.x] = best;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const double v = sv[threadIdx.x + off];
            const int i = si[threadIdx.x + off];
            if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && i < si[threadIdx.x])) { sv[threadIdx.x] = v; si[threadIdx.x] = i; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        chosen[step] = si[0];
        taken[si[0]] = 1;
    }
}

// single block: taken flags -> ascending index list (indices = sorted(indices), utils/align.py:146)
__global__ __launch_bounds__(1024) void greedy_compact_kernel(const uint8_t* __restrict__ taken, int N, int* out) {
    __shared__ int base;
    __shared__ int cnt[1024];
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int start = 0; start < N; start += 1024) {
        const int n = start + threadIdx.x;
        const int f = (n < N && taken[n]) ? 1 : 0;
        cnt[threadIdx.x] = f;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {                        // inclusive scan
            const int v = (int)threadIdx.x >= off ? cnt[threadIdx.x - off] : 0;
            __syncthreads();
            cnt[threadIdx.x] += v;
            __syncthreads();
        }
        if (f) out[base + cnt[threadIdx.x] - 1] = n;
        __syncthreads();
        if (threadIdx.x == 1023) base += cnt[1023];
        __syncthreads();
    }
}

__global__ void greedy_init_kernel(double* dist_sum, uint8_t* taken, int* chosen, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) {
        dist_sum[n] = 0.0;
        taken[n] = n == 0;
    }
    if (n == 0) chosen[0] = 0;
}

}  // namespace rnamsm

using namespace rnamsm;

static size_t greedy_ws(int N, int num_seqs, size_t* o_score, size_t* o_chosen, size_t* o_taken) {
    size_t off = (size_t)N * 8;            // dist_sum
    *o_score = off;  off += (size_t)N * 8;
    *o_chosen = off; off += ((size_t)num_seqs * 4 + 7) & ~(size_t)7;
    *o_taken = off;  off += ((size_t)N + 7) & ~(size_t)7;
    return off;
}

extern "C" size_t rnamsm_greedy_select_workspace_bytes(int N, int num_seqs) {
    if (N <= 0 || num_seqs <= 0) return 0;
    size_t a, b, c;
    return greedy_ws(N, num_seqs, &a, &b, &c);
}

extern "C" int rnamsm_greedy_select(const uint8_t* msa, int N, int L, int num_seqs, int minimise, int* out_indices,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    RNAMSM_CHECK_ARG(msa && out_indices && workspace, "greedy_select: null pointer");
    RNAMSM_CHECK_ARG(N > 0 && L > 0 && num_seqs > 0 && num_seqs <= N, "greedy_select: bad shape N=%d L=%d num_seqs=%d", N, L, num_seqs);
    size_t o_score, o_chosen, o_taken;
    const size_t need = greedy_ws(N, num_seqs, &o_score, &o_chosen, &o_taken);
    RNAMSM_CHECK_ARG(workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 7u) == 0, "greedy_select: workspace too small or misaligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    double* dist_sum = reinterpret_cast<double*>(ws);
    double* score = reinterpret_cast<double*>(ws + o_score);
    int* chosen = reinterpret_cast<int*>(ws + o_chosen);
    uint8_t* taken = reinterpret_cast<uint8_t*>(ws + o_taken);
    hipLaunchKernelGGL(greedy_init_kernel, dim3((N + 255) / 256), dim3(256), 0, s, dist_sum, taken, chosen, N);
    for (int step = 1; step < num_seqs; ++step) {
        hipLaunchKernelGGL(greedy_dist_kernel, dim3((N + 3) / 4), dim3(256), 0, s, msa, N, L, chosen, step, taken,
                           dist_sum, score, minimise);
        hipLaunchKernelGGL(greedy_pick_kernel, dim3(1), dim3(1024), 0, s, score, N, chosen, step, taken);
    }
    hipLaunchKernelGGL(greedy_compact_kernel, dim3(1), dim3(1024), 0, s, taken, N, out_indices);
    RNAMSM_CHECK_LAUNCH("greedy_select");
    return RNAMSM_OK;
}
