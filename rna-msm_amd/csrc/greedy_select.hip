// f3: greedy max/min mean-Hamming row sub-sampling of an alignment on the device
// (MSA.greedy_select, utils/align.py:128-148, reached through select_diverse(method="diversity-max|min"),
//  utils/align.py:165-181) -- the hhfilter-free way to cut an MSA to max_seqs_per_msa rows.
//
// Reference algorithm: start from row 0; each step computes the normalised Hamming distance of every row to the LAST
// chosen row (scipy cdist 'hamming' = mismatches / L in float64), appends it to a [steps, N] matrix, and takes argmax
// (argmin) of that matrix's column means over the rows not chosen yet -- first index on ties; finally the indices are
// sorted.  Ties in the mismatch TOTAL are common in real alignments and m / L is inexact, so the winner depends on the
// order of the additions: numpy reduces along the contiguous axis here (np.delete(.., axis=1) returns that layout), i.e.
// its pairwise summation -- n < 8: sequential; n <= 128: 8 interleaved accumulators, ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)),
// remainder added sequentially; n > 128: split at n/2 rounded down to a multiple of 8, recursively.  The device keeps
// the mismatch COUNT of every (step, row) (uint16, step-major so lanes read neighbouring rows) and re-reduces the
// history of every candidate at every step in exactly that order (pairwise_mean below), from a (L+1)-entry table of
// m / L, so the selected indices are bit-identical to the reference's (tests: the shipped 1176-row 2DRB_1 alignment ->
// 512 rows, where a plain running sum goes a different way at step 46).
// Roofline: HBM/L2-bound; per step N*L bytes of compares + 2*step*N bytes of history, three small launches.
#include "common.h"

namespace rnamsm {

// one wave per row: mismatches vs the last chosen row -> this step's row of the history
__global__ __launch_bounds__(256) void greedy_dist_kernel(const uint8_t* __restrict__ msa, int N, int L,
                                                          const int* __restrict__ chosen, int step,
                                                          uint16_t* __restrict__ hist) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const int lane = threadIdx.x & 63;
    const uint8_t* a = msa + (int64_t)row * L;
    const uint8_t* b = msa + (int64_t)chosen[step - 1] * L;
    int m = 0;
    for (int c = lane; c < L; c += 64) m += a[c] != b[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m += __shfl_xor(m, off, 64);
    if (lane == 0) hist[(int64_t)(step - 1) * N + row] = (uint16_t)m;
}

// numpy's pairwise summation (DOUBLE_pairwise_sum) over h[0], h[stride], .., n terms, term = lut[count]
template <int DEPTH>
__device__ __forceinline__ double pairwise_sum(const uint16_t* __restrict__ h, int64_t stride, int n,
                                               const double* __restrict__ lut) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += lut[h[i * stride]];
        return res;
    }
    if (n <= 128 || DEPTH == 0) {
        double r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = lut[h[j * stride]];
        int i = 8;
        for (; i < n - (n % 8); i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] += lut[h[(i + j) * stride]];
        }
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += lut[h[i * stride]];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    constexpr int D1 = DEPTH > 0 ? DEPTH - 1 : 0;
    const double lo = pairwise_sum<D1>(h, stride, n2, lut);
    const double hi = pairwise_sum<D1>(h + n2 * stride, stride, n - n2, lut);
    return lo + hi;
}

// one thread per row: mean of its distance history in numpy's order; taken rows can never win
__global__ __launch_bounds__(256) void greedy_score_kernel(const uint16_t* __restrict__ hist, int N, int step,
                                                           const double* __restrict__ lut,
                                                           const uint8_t* __restrict__ taken,
                                                           double* __restrict__ score, int minimise) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= N) return;
    if (taken[row]) { score[row] = -INFINITY; return; }
    // 128 * 2^4 = 2048 >= the 1023 steps the model's 1024-row limit allows (checked by the entry point)
    const double mean = pairwise_sum<4>(hist + row, N, step, lut) / (double)step;
    score[row] = minimise ? -mean : mean;          // argmin as argmax of the negated value
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

__global__ void greedy_init_kernel(double* lut, int L, uint8_t* taken, int* chosen, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) taken[n] = n == 0;
    if (n <= L) lut[n] = (double)n / (double)L;      // cdist 'hamming': mismatches / L, one IEEE division
    if (n == 0) chosen[0] = 0;
}

// MSA.weights (utils/align.py:250-253) for `sample-pretrained` sub-sampling (:150-163):
//   weights[i] = 1 / #{ j : hamming(i, j) / L < seqid_cutoff }   (pdist "hamming" in float64, the row itself included).
// One block per row i (staged in LDS); each group of G lanes (G = power of two >= min(L, 64)) counts the mismatches of one
// row j, so short alignments keep all lanes busy; the comparison is the reference's: (double)m / (double)L < cutoff.
// O(N^2 L) byte compares out of L2 -- 1e4 rows x 512 columns is ~50 G compares, well under a second.
__global__ __launch_bounds__(256) void msa_weights_kernel(const uint8_t* __restrict__ msa, int N, int L, double cutoff,
                                                          int G, double* __restrict__ weights) {
    extern __shared__ __attribute__((aligned(16))) uint8_t srow[];
    __shared__ int total;
    const int i = blockIdx.x;
    for (int c = threadIdx.x; c < L; c += 256) srow[c] = msa[(int64_t)i * L + c];
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    const int groups = 256 / G, g = threadIdx.x / G, gl = threadIdx.x % G;
    int mine = 0;
    for (int j = g; j < N; j += groups) {
        const uint8_t* b = msa + (int64_t)j * L;
        int m = 0;
        for (int c = gl; c < L; c += G) m += srow[c] != b[c];
        for (int off = G >> 1; off > 0; off >>= 1) m += __shfl_xor(m, off, 64);
        if (gl == 0 && (double)m / (double)L < cutoff) ++mine;
    }
    if (gl == 0 && mine) atomicAdd(&total, mine);        // integer count: order-independent
    __syncthreads();
    if (threadIdx.x == 0) weights[i] = 1.0 / (double)total;
}

}  // namespace rnamsm

using namespace rnamsm;

extern "C" int rnamsm_msa_weights(const uint8_t* msa, int N, int L, double seqid_cutoff, double* weights, void* stream) {
    RNAMSM_CHECK_ARG(msa && weights, "msa_weights: null pointer");
    RNAMSM_CHECK_ARG(N > 0 && L > 0 && L <= 65536, "msa_weights: bad shape N=%d L=%d", N, L);
    int G = 1;
    while (G < L && G < 64) G <<= 1;
    hipLaunchKernelGGL(msa_weights_kernel, dim3((unsigned)N), dim3(256), (size_t)L, static_cast<hipStream_t>(stream), msa, N,
                       L, seqid_cutoff, G, weights);
    RNAMSM_CHECK_LAUNCH("msa_weights");
    return RNAMSM_OK;
}

static size_t greedy_ws(int N, int L, int num_seqs, size_t* o_score, size_t* o_chosen, size_t* o_taken, size_t* o_hist) {
    size_t off = ((size_t)L + 1) * 8;      // lut: m / L
    *o_score = off;  off += (size_t)N * 8;
    *o_chosen = off; off += ((size_t)num_seqs * 4 + 7) & ~(size_t)7;
    *o_taken = off;  off += ((size_t)N + 7) & ~(size_t)7;
    *o_hist = off;   off += ((size_t)(num_seqs > 1 ? num_seqs - 1 : 0) * (size_t)N * 2 + 7) & ~(size_t)7;
    return off;
}

extern "C" size_t rnamsm_greedy_select_workspace_bytes(int N, int L, int num_seqs) {
    if (N <= 0 || L <= 0 || num_seqs <= 0) return 0;
    size_t a, b, c, d;
    return greedy_ws(N, L, num_seqs, &a, &b, &c, &d);
}

extern "C" int rnamsm_greedy_select(const uint8_t* msa, int N, int L, int num_seqs, int minimise, int* out_indices,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    RNAMSM_CHECK_ARG(msa && out_indices && workspace, "greedy_select: null pointer");
    RNAMSM_CHECK_ARG(N > 0 && L > 0 && num_seqs > 0 && num_seqs <= N, "greedy_select: bad shape N=%d L=%d num_seqs=%d", N, L, num_seqs);
    RNAMSM_CHECK_ARG(L < 65536 && num_seqs <= 2048, "greedy_select: L=%d must be < 65536 and num_seqs=%d <= 2048", L, num_seqs);
    size_t o_score, o_chosen, o_taken, o_hist;
    const size_t need = greedy_ws(N, L, num_seqs, &o_score, &o_chosen, &o_taken, &o_hist);
    RNAMSM_CHECK_ARG(workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 7u) == 0, "greedy_select: workspace too small or misaligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    double* lut = reinterpret_cast<double*>(ws);
    double* score = reinterpret_cast<double*>(ws + o_score);
    int* chosen = reinterpret_cast<int*>(ws + o_chosen);
    uint8_t* taken = reinterpret_cast<uint8_t*>(ws + o_taken);
    uint16_t* hist = reinterpret_cast<uint16_t*>(ws + o_hist);
    const int n_init = N > L + 1 ? N : L + 1;
    hipLaunchKernelGGL(greedy_init_kernel, dim3((n_init + 255) / 256), dim3(256), 0, s, lut, L, taken, chosen, N);
    for (int step = 1; step < num_seqs; ++step) {
        hipLaunchKernelGGL(greedy_dist_kernel, dim3((N + 3) / 4), dim3(256), 0, s, msa, N, L, chosen, step, hist);
        hipLaunchKernelGGL(greedy_score_kernel, dim3((N + 255) / 256), dim3(256), 0, s, hist, N, step, lut, taken, score,
                           minimise);
        hipLaunchKernelGGL(greedy_pick_kernel, dim3(1), dim3(1024), 0, s, score, N, chosen, step, taken);
    }
    hipLaunchKernelGGL(greedy_compact_kernel, dim3(1), dim3(1024), 0, s, taken, N, out_indices);
    RNAMSM_CHECK_LAUNCH("greedy_select");
    return RNAMSM_OK;
}
