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
// Roofline: HBM/L2-bound; per step N*L bytes of compares + 2*step*N bytes of history.  The steps are sequential (step s
// compares against the row step s-1 picked), so for alignments of a few thousand rows the cost is the per-step latency:
// there ONE launch per step (greedy_step_kernel: distances, re-reduced means, block argmax, and the grid-wide argmax by
// whichever block finishes last) replaces three; deeper alignments keep the three kernels (one thread per row).
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

// ---- one launch per step ---------------------------------------------------------------------------------------
// One WAVE per row (four rows per block): the distance to the last pick over 64 lanes, then the row's score -- numpy's
// pairwise sum over its distance history, evaluated in numpy's exact order but with its independent pieces side by side:
// the recursion's leaves (<= 128 terms each, <= 32 of them) go to the wave's eight 8-lane groups, a leaf's eight
// interleaved accumulators r[0..7] to the eight lanes of a group, and the leaf sums are then added along the recursion
// tree.  The history is kept row-major here ([N][S], one row's steps contiguous: a wave reads 128-byte runs).
__device__ __forceinline__ bool greedy_better(double v, int i, double bv, int bi) { return v > bv || (v == bv && i < bi); }

template <int DEPTH>
__device__ __forceinline__ double greedy_combine(int n, const double* __restrict__ leaf, int& k) {
    if (n <= 128 || DEPTH == 0) return leaf[k++];
    int n2 = n / 2;
    n2 -= n2 % 8;
    constexpr int D1 = DEPTH > 0 ? DEPTH - 1 : 0;
    const double lo = greedy_combine<D1>(n2, leaf, k);
    const double hi = greedy_combine<D1>(n - n2, leaf, k);
    return lo + hi;
}

// grid = ceil(N / 4).  blk_v / blk_i: one candidate per block; *counter: blocks finished this step (reset by the last).
__global__ __launch_bounds__(256) void greedy_step_kernel(const uint8_t* __restrict__ msa, int N, int L, int S, int* chosen,
                                                          int step, uint16_t* __restrict__ hist,
                                                          const double* __restrict__ lut, uint8_t* taken, int minimise,
                                                          double* blk_v, int* blk_i, unsigned* counter) {
    __shared__ int s_off[4][32], s_len[4][32];       // n <= 2047 terms: a split piece has >= 64 terms, so <= 32 leaves
    __shared__ double s_leaf[4][32];
    __shared__ double sv[4];
    __shared__ int si[4];
    __shared__ int last_block;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wv;
    double v = -INFINITY;
    int bi = 0x7fffffff;
    if (row < N) {                                                   // wave-uniform
        // ---- Hamming distance to the row picked last
        const uint8_t* a = msa + (int64_t)row * L;
        const uint8_t* b = msa + (int64_t)chosen[step - 1] * L;
        int m = 0;
        if ((L & 3) == 0 && ((reinterpret_cast<uintptr_t>(msa)) & 3u) == 0) {
            const uint32_t* a4 = reinterpret_cast<const uint32_t*>(a);
            const uint32_t* b4 = reinterpret_cast<const uint32_t*>(b);
            for (int c = lane; c < L / 4; c += 64) {
                const uint32_t x = a4[c] ^ b4[c];
                m += __popc((((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u);      // bytes that differ
            }
        } else {
            for (int c = lane; c < L; c += 64) m += a[c] != b[c];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m += __shfl_xor(m, off, 64);
        uint16_t* h = hist + (int64_t)row * S;
        if (lane == 0) h[step - 1] = (uint16_t)m;
        if (!taken[row]) {
            // term i of the row's history; the newest one comes from the register (the store above need not be visible)
            auto term = [&](int i) -> double { return lut[i == step - 1 ? m : (int)h[i]]; };
            // ---- leaves of numpy's recursion over n = step terms, in depth-first (= left to right) order
            int nleaf = 0, maxlen = 0;
            {
                int so[8], sn[8], sp = 1;
                so[0] = 0; sn[0] = step;
                while (sp) {
                    --sp;
                    const int o = so[sp], n = sn[sp];
                    if (n <= 128) {
                        if (lane == 0) { s_off[wv][nleaf] = o; s_len[wv][nleaf] = n; }
                        maxlen = n > maxlen ? n : maxlen;
                        ++nleaf;
                    } else {
                        int n2 = n / 2;
                        n2 -= n2 % 8;
                        so[sp] = o + n2; sn[sp] = n - n2; ++sp;      // right half is popped after ...
                        so[sp] = o; sn[sp] = n2; ++sp;               // ... the left half
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            const int g = lane >> 3, j = lane & 7;
            for (int base = 0; base < nleaf; base += 8) {
                const int k = base + g;
                const int o = k < nleaf ? s_off[wv][k] : 0, n = k < nleaf ? s_len[wv][k] : 0;
                const int full = n >= 8 ? n - n % 8 : 0;
                double r = full ? term(o + j) : 0.0;                 // r[j] = a[j]
                for (int i = 8; i < maxlen; i += 8)
                    if (i < full) r += term(o + i + j);              // r[j] += a[i + j]
                double res = r + __shfl_xor(r, 1, 64);               // (r0+r1), (r2+r3), ...
                res = res + __shfl_xor(res, 2, 64);                  // (r0+r1)+(r2+r3), ...
                res = res + __shfl_xor(res, 4, 64);                  // ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7))
                if (n < 8) res = 0.0;                                // n < 8: plain sequential sum from 0
                for (int i = 0; i < 7; ++i)
                    if (full + i < n) res += term(o + full + i);     // the n % 8 trailing terms, one after the other
                if (j == 0 && k < nleaf) s_leaf[wv][k] = res;
            }
            __builtin_amdgcn_wave_barrier();
            int k = 0;
            const double mean = greedy_combine<5>(step, s_leaf[wv], k) / (double)step;
            v = minimise ? -mean : mean;                             // argmin as argmax of the negated value
            bi = row;
        }
    }
    if (lane == 0) { sv[wv] = v; si[wv] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < 4; ++q)
            if (greedy_better(sv[q], si[q], v, bi)) { v = sv[q]; bi = si[q]; }
        __hip_atomic_store(&blk_v[blockIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&blk_i[blockIdx.x], bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // release the candidate, take a ticket; whoever draws the last one has (acquire) every block's candidate
        const unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last_block = ticket == gridDim.x - 1;
    }
    __syncthreads();
    if (!last_block) return;
    v = -INFINITY;
    bi = 0x7fffffff;
    for (int q = threadIdx.x; q < (int)gridDim.x; q += 256) {
        const double ov = __hip_atomic_load(&blk_v[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int oi = __hip_atomic_load(&blk_i[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (greedy_better(ov, oi, v, bi)) { v = ov; bi = oi; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(v, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (greedy_better(ov, oi, v, bi)) { v = ov; bi = oi; }
    }
    __syncthreads();                                   // sv / si of the block phase have been consumed
    if (lane == 0) { sv[wv] = v; si[wv] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < 4; ++q)
            if (greedy_better(sv[q], si[q], v, bi)) { v = sv[q]; bi = si[q]; }
        chosen[step] = bi;
        taken[bi] = 1;
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

__global__ void greedy_init_kernel(double* lut, int L, uint8_t* taken, int* chosen, int N, unsigned* counter) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n == 0 && counter) *counter = 0u;
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
    // the row under comparison is held in L bytes of dynamic LDS next to a static counter: the default 64 KB limit is never
    // raised, so L is capped where that fits with room to spare (alignments are cropped to 1024 columns upstream)
    RNAMSM_CHECK_ARG(N > 0 && L > 0 && L <= 32768, "msa_weights: bad shape N=%d L=%d (L <= 32768)", N, L);
    int G = 1;
    while (G < L && G < 64) G <<= 1;
    hipLaunchKernelGGL(msa_weights_kernel, dim3((unsigned)N), dim3(256), (size_t)L, static_cast<hipStream_t>(stream), msa, N,
                       L, seqid_cutoff, G, weights);
    RNAMSM_CHECK_LAUNCH("msa_weights");
    return RNAMSM_OK;
}

struct GreedyWs {
    size_t score, chosen, taken, hist, blk_v, blk_i, counter, total;
};
static GreedyWs greedy_ws(int N, int L, int num_seqs) {
    GreedyWs w;
    size_t off = ((size_t)L + 1) * 8;      // lut: m / L
    w.score = off;  off += (size_t)N * 8;
    w.chosen = off; off += ((size_t)num_seqs * 4 + 7) & ~(size_t)7;
    w.taken = off;  off += ((size_t)N + 7) & ~(size_t)7;
    w.hist = off;   off += ((size_t)(num_seqs > 1 ? num_seqs - 1 : 0) * (size_t)N * 2 + 7) & ~(size_t)7;
    const size_t nblk = ((size_t)N + 3) / 4;
    w.blk_v = off;  off += nblk * 8;
    w.blk_i = off;  off += (nblk * 4 + 7) & ~(size_t)7;
    w.counter = off; off += 8;
    w.total = off;
    return w;
}

extern "C" size_t rnamsm_greedy_select_workspace_bytes(int N, int L, int num_seqs) {
    if (N <= 0 || L <= 0 || num_seqs <= 0) return 0;
    return greedy_ws(N, L, num_seqs).total;
}

extern "C" int rnamsm_greedy_select(const uint8_t* msa, int N, int L, int num_seqs, int minimise, int* out_indices,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    RNAMSM_CHECK_ARG(msa && out_indices && workspace, "greedy_select: null pointer");
    RNAMSM_CHECK_ARG(N > 0 && L > 0 && num_seqs > 0 && num_seqs <= N, "greedy_select: bad shape N=%d L=%d num_seqs=%d", N, L, num_seqs);
    RNAMSM_CHECK_ARG(L < 65536 && num_seqs <= 2048, "greedy_select: L=%d must be < 65536 and num_seqs=%d <= 2048", L, num_seqs);
    const GreedyWs w = greedy_ws(N, L, num_seqs);
    RNAMSM_CHECK_ARG(workspace_bytes >= w.total && (reinterpret_cast<uintptr_t>(workspace) & 7u) == 0, "greedy_select: workspace too small or misaligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    double* lut = reinterpret_cast<double*>(ws);
    double* score = reinterpret_cast<double*>(ws + w.score);
    int* chosen = reinterpret_cast<int*>(ws + w.chosen);
    uint8_t* taken = reinterpret_cast<uint8_t*>(ws + w.taken);
    uint16_t* hist = reinterpret_cast<uint16_t*>(ws + w.hist);
    unsigned* counter = reinterpret_cast<unsigned*>(ws + w.counter);
    const int n_init = N > L + 1 ? N : L + 1;
    hipLaunchKernelGGL(greedy_init_kernel, dim3((n_init + 255) / 256), dim3(256), 0, s, lut, L, taken, chosen, N, counter);
    // one wave per row pays while the rows are few (measured, L = 1024: N = 2048 19 ms vs 32 ms; N = 8192 50 vs 37 ms;
    // 2DRB_1's 1176 x 35 -> 512: 7.1 vs 9.6 ms): beyond ~3000 rows one THREAD per row re-reduces the history cheaper
    if (tuning().greedy_fused == 2 || (tuning().greedy_fused == 1 && N <= 3072)) {
        const int S = num_seqs - 1;                    // history row-major [N][S] on this path
        double* blk_v = reinterpret_cast<double*>(ws + w.blk_v);
        int* blk_i = reinterpret_cast<int*>(ws + w.blk_i);
        for (int step = 1; step < num_seqs; ++step)
            hipLaunchKernelGGL(greedy_step_kernel, dim3((N + 3) / 4), dim3(256), 0, s, msa, N, L, S, chosen, step, hist, lut,
                               taken, minimise, blk_v, blk_i, counter);
    } else {
        for (int step = 1; step < num_seqs; ++step) {
            hipLaunchKernelGGL(greedy_dist_kernel, dim3((N + 3) / 4), dim3(256), 0, s, msa, N, L, chosen, step, hist);
            hipLaunchKernelGGL(greedy_score_kernel, dim3((N + 255) / 256), dim3(256), 0, s, hist, N, step, lut, taken, score,
                               minimise);
            hipLaunchKernelGGL(greedy_pick_kernel, dim3(1), dim3(1024), 0, s, score, N, chosen, step, taken);
        }
    }
    hipLaunchKernelGGL(greedy_compact_kernel, dim3(1), dim3(1024), 0, s, taken, N, out_indices);
    RNAMSM_CHECK_LAUNCH("greedy_select");
    return RNAMSM_OK;
}
