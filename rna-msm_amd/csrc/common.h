// Shared host/device helpers for librnamsm_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>
#include <shared_mutex>

#include "../../include/rnamsm.h"

namespace rnamsm {

// ---- error plumbing (thread-local text behind rnamsm_last_error) -------------------------------
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);

#define RNAMSM_CHECK_ARG(cond, ...)                                   \
    do {                                                              \
        if (!(cond)) return ::rnamsm::fail(RNAMSM_ERR_INVALID, __VA_ARGS__); \
    } while (0)

// The hi/lo-bf16 three-product mode ("bf16x3") was removed in round 5: it costs the same three MFMAs per product as f16x3 with
// 17 instead of 22 operand bits (tests/analysis/README.md held a recorded exceedance of it).  Entry points that take (split, fmt)
// or lo planes with fmt 0 answer RNAMSM_ERR_UNSUPPORTED.
#define RNAMSM_NO_BF16X3(cond, name)                                                                                     \
    do {                                                                                                                 \
        if (cond)                                                                                                        \
            return ::rnamsm::fail(RNAMSM_ERR_UNSUPPORTED, name ": hi/lo bf16 pairs (bf16x3) were removed; use fmt 1 (f16x3)"); \
    } while (0)

#define RNAMSM_CHECK_LAUNCH(name)                                                         \
    do {                                                                                  \
        hipError_t e_ = hipGetLastError();                                                \
        if (e_ != hipSuccess)                                                             \
            return ::rnamsm::fail(RNAMSM_ERR_HIP, "%s launch: %s", name, hipGetErrorString(e_)); \
    } while (0)

// ---- per-kernel timing (api.hip); categories index rnamsm_timing_get -----------------------------
enum TimingCategory {
    TC_GEMM = 0, TC_ROW_LOGITS, TC_SOFTMAX, TC_ROW_APPLY, TC_COL_ATTN, TC_LAYERNORM, TC_EMBED, TC_PACK, TC_COUNT
};
// Roofline terms of one launch (MI355X_MICROARCH.md): dense matrix peak of the instruction family it issues and the
// HBM rate a streaming kernel can reach.  A launch's bound is max(executed flops / matrix peak, algorithmic bytes / HBM
// rate); rnamsm_timing_get_bound sums it per category, so a mode's kernels are each priced against their OWN limit
// (in the 16-bit modes out_proj, the LayerNorms and the attention kernels are HBM-bound, QKV / fc1 matrix-bound).
constexpr double PEAK_F32_MFMA_TFLOPS = 157.3;     // v_mfma_f32_32x32x2_f32: 64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz
constexpr double PEAK_F16_MFMA_TFLOPS = 2500.0;    // dense bf16 / f16 MFMA
constexpr double PEAK_HBM_TBPS = 6.3;              // measured float4 copy (8 TB/s spec)
// Vector ALU: a wave64 instruction occupies its SIMD for 2 cycles (32 lanes per clock), a transcendental (v_exp_f32, v_rcp_f32)
// for four times that; 1024 SIMDs x 32 lanes x 2.4 GHz lane-operations per second.  The softmax of the fused attention
// kernels is priced with it: `valu_lane_ops` of a launch = scores x (plain instructions + 4 x transcendentals per score).
constexpr double PEAK_VALU_LANE_OPS = 1024.0 * 32.0 * 2.4e9;
bool timing_enabled();
void timing_begin(int category, double flops, double bytes, double mfma_s, double hbm_s, double valu_s, hipStream_t stream);
void timing_end(hipStream_t stream);
struct KernelTimer {   // brackets one launch with two hipEventRecord calls when timing is on
    hipStream_t s;
    bool on;
    // flops / bytes: ALGORITHMIC work of the launch; peak_tflops: matrix peak of its MFMA family; flop_mult: executed MFMA
    // flops per algorithmic flop (3 in the hi/lo-split modes)
    // valu_lane_ops: vector-ALU work of the launch in lane-operations (0 = not priced), see PEAK_VALU_LANE_OPS
    KernelTimer(int category, double flops, double bytes, hipStream_t stream, double peak_tflops = PEAK_F32_MFMA_TFLOPS,
                double flop_mult = 1.0, double valu_lane_ops = 0.0)
        : s(stream), on(timing_enabled()) {
        if (on)
            timing_begin(category, flops, bytes, flop_mult * flops / (peak_tflops * 1e12), bytes / (PEAK_HBM_TBPS * 1e12),
                         valu_lane_ops / PEAK_VALU_LANE_OPS, s);
    }
    ~KernelTimer() {
        if (on) timing_end(s);
    }
};

// ---- tuning knobs (api.hip) -------------------------------------------------------------------------
// The folded LayerNorm (forward.hip, knob "ln_fold" = 1) is taken by an alignment of at least this many tokens, in every exact
// driver alike: the decision follows the MEMBER, never the batch.  Round 5: 18432 -> 4096 (tools/forward_knob_ab.py ln_fold=0,3 after
// row_stats_from_partials lost its serial loads: +1.1 .. +1.6 % from 4096 tokens up, +0.4 % at 2048: docs/history/profiles_r05/r05_ln_fold_threshold_ab.log).
constexpr int64_t LN_FOLD_MIN_TOKENS = 4096;
struct Tuning {
    int gemm16_dma = 3;            // plane-input 16-bit GEMMs: 0 register staging, 1 / 2 LDS-DMA 128x128,
                                   // 3 = LDS-DMA 256x256 with software-pipelined fragments, BK = 64 for split 1 (default),
                                   // 4 = the same with BK = 32 for every split
    int gemm16_mfma16 = 1;         // plain-bf16 256x256 GEMM: 1 = v_mfma_f32_16x16x32_bf16 (gemm16_q16s_kernel) for wide N, 2 = always, 0 = 32x32x16
    int gemm_group = 0;            // fp32 GEMM: row panels per XCD group of the block order (xcd_panel_map_grouped); 0 = by shape
    int gemm_tile = 0;             // fp32 GEMM block tile: 0 = by shape, 1 = always 128x128, 2 = always 128x64, 3 = mixed wherever a launch has whole rounds and a tail, 4 = by shape among the uniform tilings only (round 4's rule, A/B)
    int gemm_splitk_short = 0;     // rnamsm_forward*, the K = 768 GEMMs of a lone small alignment (<= 192 tiles): K ranges (0 = off: the default -- measured no gain once the block order was fixed; 2, 4), gemm_f32_splitk_factor
    int gemm_splitk = 0;           // rnamsm_forward, fc2 below ~1.4 k tokens: 0 = never (default since round 5: a split chosen by the BATCH's token count made an alignment's bits depend on its company; costs a lone <= 1024-token alignment 0.4 of 2.5 ms, docs/history/profiles_r05/r05_splitk.log), 1 = four K ranges + an ordered reduction (gemm_f32_splitk_factor), 2 / 4 / 8 = forced (A/B)
    int row_narrow = 1;            // fp32 K4 / K6 at C <= 64: 1 = the LDS-free narrow kernels (row_logits_narrow / row_apply_narrow; bit-identical), 0 = the 128 x 128 tile kernels (A/B)
    int col_small = 1;             // fp32 col_attn at R <= 16: 1 = one wave per (column, head), no LDS (col_attn_small_kernel), 0 = the 128-query blocks
    int col_fast = 1;              // fp32 col_attn on prescaled q (rnamsm_col_attn_fused_prescaled): 1 = FAST loop (no running maximum) with the TRACKED loop as fallback, 0 = TRACKED only (A/B)
    int col_dma = -1;              // fp32 col_attn: 1 = LDS-DMA staging, 32-key chunks, 3 blocks/CU; 0 = register-staged kernel; -1 = by shape
    int row16_max_rows = 32;       // hi/lo modes: cap on the rows of one row_logits16 slab (0 = none): accuracy, DESIGN 3.2
    int ln_fold = 1;               // rnamsm_forward with ln_folded: 1 = LayerNorm applied inside the consuming GEMM (row sums from the producers'
                                   // epilogues) when R*C >= LN_FOLD_MIN_TOKENS (4096), 3 = for every shape, 2 = every GEMM sums its rows itself (A/B), 0 = separate launches
    int greedy_fused = 1;          // rnamsm_greedy_select: 1 = one launch per step (fused distance / score / argmax) up to 3072 rows, 2 = always, 0 = three launches
    int attn16 = 1;                // 16-bit modes of rnamsm_forward: 1 = attention contractions on the 16-bit matrix cores too, 0 = fp32 attention
};
Tuning& tuning();
// The knobs are process-global A/B instruments, read on the HOST while a driver enqueues its launches.  Writing one while another
// thread is inside a forward driver would let that forward mix two settings (and, for the arithmetic knobs, two roundings).  One
// reader-writer lock closes that (the earlier counter was check-then-act): a driver holds it SHARED for as long as it
// enqueues, rnamsm_set_param takes it EXCLUSIVE with try_lock and refuses (RNAMSM_ERR_INVALID) when a driver holds it -- it never
// blocks inside the library, and a driver that starts while a knob is being written waits the few nanoseconds the write takes.
std::shared_mutex& tuning_lock();
struct ForwardScope {
    ForwardScope() { tuning_lock().lock_shared(); }
    ~ForwardScope() { tuning_lock().unlock_shared(); }
    ForwardScope(const ForwardScope&) = delete;
    ForwardScope& operator=(const ForwardScope&) = delete;
};
// rnamsm_forward's choice of the 16-bit GEMM tile by the MSA's token count: below ~9-10 k tokens the 256x256 kernels leave most
// CUs without a tile (out_proj at 8192 tokens: 96 tiles for 256 CUs) and the 128x128 kernel is faster -- whole forward, one
// process (round 3's mid-size tile A/B, in the history): 2048 tokens x1.33 (bf16) / x1.59 (f16x3), 4096 x1.14 / x1.32, 8192 x1.11 / x1.10, level at
// 10 k (bf16) / 9 k (f16x3), 256x256 ahead from there (x0.92 at 12 k, x0.81-0.88 at 60 k).  The hi/lo modes' two kernels sum
// every output element in the same order: bit-identical either way.
// The choice is handed down as a THREAD-LOCAL override read by rnamsm_gemm_bf16 (gemm16_big_rows_now): nothing process-wide is
// written by a forward, so concurrent forwards from several threads / devices cannot leak their temporary threshold
// into each other or into later direct rnamsm_gemm_bf16 calls.
int& gemm16_big_rows_override();                 // api.hip: thread_local, 0 = none
struct BigRowsScope {
    int saved;
    explicit BigRowsScope(bool plain_bf16) : saved(gemm16_big_rows_override()) {
        gemm16_big_rows_override() = plain_bf16 ? 10752 : 8960;
    }
    ~BigRowsScope() { gemm16_big_rows_override() = saved; }
};
// rows from which rnamsm_gemm_bf16 uses the 256x256-tile kernels: the calling forward's choice, else 2048
inline int64_t gemm16_big_rows_now() {
    const int o = gemm16_big_rows_override();
    return o > 0 ? o : 2048;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) belongs to the (function, DEVICE) pair, so "already configured" is
// remembered per device of the calling thread, not per process (a process may drive several GPUs: data.device=cuda:1,
// or a helper thread whose current device differs).  Racing threads may both configure -- the call is idempotent.
struct DeviceOnce {
    std::atomic<uint64_t> mask{0};
    static int current() {
        int d = -1;
        return hipGetDevice(&d) == hipSuccess ? d : -1;
    }
    bool pending() const {
        const int d = current();
        return d < 0 || d >= 64 || !((mask.load(std::memory_order_acquire) >> d) & 1ull);
    }
    void mark() {
        const int d = current();
        if (d >= 0 && d < 64) mask.fetch_or(1ull << d, std::memory_order_release);
    }
};

// Epilogue traffic policy: a GEMM's outputs and its residual tile are touched once per kernel, its operand panels are
// re-read by the other tiles that share them -- so the former are marked non-temporal and do not evict the latter from
// the 4 MB L2.  A/B as two builds of the library on one box (tools/nt_epi_ab.sh, -DRNAMSM_NT_EPI=0 for the plain policy):
// bf16 GEMMs +3.4 %, f16x3 +1.8 %, fp32 unchanged (MFMA-bound); results bit-identical.
#ifndef RNAMSM_NT_EPI
#define RNAMSM_NT_EPI 1
#endif
template <class V>
__device__ __forceinline__ void epi_store(V* p, V v) {
#if RNAMSM_NT_EPI
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
template <class V>
__device__ __forceinline__ V epi_load(const V* p) {
#if RNAMSM_NT_EPI
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- device helpers ----------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// Exact-fp32 matrix core step: D(32x32) += A(32x2) * B(2x32).
// Lane l supplies A[row = l&31][k = l>>5] and B[k = l>>5][col = l&31]; accumulator register t of lane l is
// D[row = (t&3) + 8*(t>>2) + 4*(l>>5)][col = l&31]  (cdna_hip_programming.md §3).
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// GELU in its exact-erf form (nn.GELU(), modules.py:416; gelu(), modules.py:11-20): 0.5 v (1 + erf(v / sqrt 2)).
// erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7), branch-free: one v_rcp, five FMAs, one v_exp -- half the
// VALU issue slots of the library erff, whose two |x| ranges both execute under lane divergence.  In float32 the result
// is as close to the fp64 GELU as the erff formula itself (max abs error 4.6e-7 vs 4.5e-7 over [-12, 12], rel-L2 2.5e-8):
// the last bits are set by the fp32 products, not by the erf approximation.
__device__ __forceinline__ float gelu_erf(float v) {
    const float z = v * 0.70710678118654752440f;
    const float a = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.f));
    float p = 1.061405429f;
    p = fmaf(p, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(a * a * -1.4426950408889634f);
    const float er = copysignf(fmaf(-p, e, 1.f), z);
    return 0.5f * v * (1.f + er);
}

// Two elements at once on the packed-fp32 VALU (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth per issue slot); the same
// operations in the same order as gelu_erf, so results are bit-identical to it.  The GEMM epilogues run this form: the
// fc1 epilogue is ~1400 VALU instructions per wave in scalar form, 3.5 % of the kernel.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 v) {
    const f32x2 z = v * 0.70710678118654752440f;
    f32x2 a;
    a[0] = fabsf(z[0]);
    a[1] = fabsf(z[1]);
    const f32x2 den = __builtin_elementwise_fma(a, (f32x2)(0.3275911f), (f32x2)(1.f));
    f32x2 t;
    t[0] = __builtin_amdgcn_rcpf(den[0]);
    t[1] = __builtin_amdgcn_rcpf(den[1]);
    f32x2 p = (f32x2)(1.061405429f);
    p = __builtin_elementwise_fma(p, t, (f32x2)(-1.453152027f));
    p = __builtin_elementwise_fma(p, t, (f32x2)(1.421413741f));
    p = __builtin_elementwise_fma(p, t, (f32x2)(-0.284496736f));
    p = __builtin_elementwise_fma(p, t, (f32x2)(0.254829592f));
    p = p * t;
    const f32x2 ea = a * a * -1.4426950408889634f;
    f32x2 e;
    e[0] = __builtin_amdgcn_exp2f(ea[0]);
    e[1] = __builtin_amdgcn_exp2f(ea[1]);
    f32x2 er = __builtin_elementwise_fma(-p, e, (f32x2)(1.f));
    er[0] = copysignf(er[0], z[0]);
    er[1] = copysignf(er[1], z[1]);
    return 0.5f * v * (1.f + er);
}

// Barrier of the LDS-DMA staged loops: "my LDS reads are done (lgkmcnt 0), my global_load_lds older than the N
// youngest have landed (vmcnt N), then s_barrier" -- after it EVERY wave's share of the awaited tile is in LDS.
// __syncthreads() must not be used for this: its workgroup-scope fence does not wait for vmcnt, and whether hipcc adds a
// vmcnt wait for an LDS-DMA depends on its alias guess about later ds_reads (col_attn16 got none before its K-fragment
// reads: a wave could read rows another wave's DMA had not delivered yet -- a race that 2250 bit-identical soak reruns
// never hit with 64-key chunks and that showed up at once with 32-key chunks and three blocks per CU).
template <int N>
__device__ __forceinline__ void wait_dma_then_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// XCD-aware block remap for panel-sharing tiled kernels: hardware deals consecutive block ids round-robin over the
// 8 XCDs (each with a private 4 MiB L2), so blocks b and b+8 share an L2.  Blocks that read the same operand panel
// are given ids of equal (b % 8): `inner` consecutive logical tiles (one panel) per XCD slot.
// Returns false for padding blocks.  Speed only -- results never depend on placement.
__device__ __forceinline__ bool xcd_panel_map(unsigned bid, unsigned num_panels, unsigned inner,
                                              unsigned& panel, unsigned& in_panel) {
    const unsigned xcd = bid & 7u, idx = bid >> 3;
    panel = (idx / inner) * 8u + xcd;
    in_panel = idx % inner;
    return panel < num_panels;
}
static inline unsigned xcd_panel_grid(unsigned num_panels, unsigned inner) {
    return ((num_panels + 7u) / 8u) * 8u * inner;
}

// Same, with the XCD's panels taken in groups of G and the tiles of a group ordered in_panel-major: the blocks
// resident on an XCD at one time then cover G panels x (resident / G) consecutive in_panel values instead of
// (resident / inner) whole panels, so the operand indexed by in_panel (the weight column block of a GEMM) is
// re-streamed from the fabric once per G panels.  G == 1 is xcd_panel_map.
__device__ __forceinline__ bool xcd_panel_map_grouped(unsigned bid, unsigned num_panels, unsigned inner, unsigned G,
                                                      unsigned& panel, unsigned& in_panel) {
    const unsigned xcd = bid & 7u, idx = bid >> 3;
    const unsigned per_group = G * inner;
    const unsigned group = idx / per_group, rem = idx % per_group;
    in_panel = rem / G;
    panel = (group * G + rem % G) * 8u + xcd;
    return panel < num_panels;
}
// The same order WITHOUT padding groups (gemm_f32_kernel, round 4): the panels an XCD has left after its full groups of G form one
// last, smaller group.  Why it matters: hardware hands an XCD's consecutive workgroups to its CUs in turn, and a padded group
// holds its real blocks at a stride of G -- 1 real panel in a group of 8 put all its column blocks on 4 of the XCD's 32 CUs (a
// 82-token QKV GEMM: 18 tiles, 117 us instead of 26; the same at the ragged end of every mid-size GEMM: tools/gemm_f32_small_m.py).
__device__ __forceinline__ bool xcd_panel_map_ragged(unsigned bid, unsigned num_panels, unsigned inner, unsigned G,
                                                     unsigned& panel, unsigned& in_panel) {
    if (G == 0u) {          // FLAT: fewer tiles than block slots -- tile = block id, so the tiles spread over all XCDs and CUs
        panel = bid / inner;
        in_panel = bid % inner;
        return panel < num_panels;
    }
    const unsigned xcd = bid & 7u, idx = bid >> 3;
    // THIS XCD's panels (xcd, xcd + 8, ...): where the panel count is no multiple of 8 the higher XCDs own one fewer, and their
    // spare block ids lie at the END of their sequence, not strided through a group
    const unsigned local = (num_panels + 7u - xcd) / 8u, full = local / G, per_group = G * inner;
    if (idx >= local * inner) return false;
    unsigned local_panel;
    if (idx < full * per_group) {
        const unsigned group = idx / per_group, rem = idx % per_group;
        in_panel = rem / G;
        local_panel = group * G + rem % G;
    } else {
        const unsigned tail = local - full * G, r = idx - full * per_group;      // tail >= 1 here
        in_panel = r / tail;
        local_panel = full * G + r % tail;
    }
    panel = local_panel * 8u + xcd;
    return true;
}
static inline unsigned xcd_panel_grid_ragged(unsigned num_panels, unsigned inner, unsigned G) {
    return G == 0u ? num_panels * inner : ((num_panels + 7u) / 8u) * 8u * inner;
}
// Group size for PERSISTENT walks (block b takes virtual ids b, b + gridDim, ...): padding ids of a group that is not full
// recur with the period of the walk, so whole blocks would own nothing but padding (16 row panels in groups of 8: a quarter of
// the blocks did all the work -- 432 instead of 1136 TFLOP/s on a 4096^3 bf16 GEMM, round 4's square-shape A/B).  The
// largest G <= want that divides the panels per XCD leaves no padding inside the groups.
static inline unsigned xcd_group_for_persistent(unsigned num_panels, unsigned want) {
    const unsigned local = (num_panels + 7u) / 8u;
    unsigned g = want < 1u ? 1u : want;
    while (g > 1u && local % g) --g;
    return g;
}
static inline unsigned xcd_panel_grid_grouped(unsigned num_panels, unsigned inner, unsigned G) {
    const unsigned local = (num_panels + 7u) / 8u;                 // panels per XCD
    return ((local + G - 1u) / G) * G * 8u * inner;
}

// split-K form of the fp32 residual GEMM for small M (gemm_f32.hip): ks = gemm_f32_splitk_factor(M, N, K) > 1 K ranges,
// partial tiles in `partials` ([ks][M][N] floats), then out = sum of the slabs in order + bias + residual.
int gemm_f32_splitk_factor(int64_t M, int N, int K, bool by_shape_only = false);
int gemm_f32_splitk(const float* A, int64_t lda, const float* W, const float* bias, const float* residual, int64_t ldr,
                    float* Cout, int64_t ldc, int64_t M, int N, int K, int ks, float* partials, hipStream_t stream,
                    int act = RNAMSM_ACT_NONE, float scale = 1.f, int scale_cols = 0, const uint8_t* zero_rows = nullptr);
// floats the split-K partial slabs of a forward over T tokens may need (sized by shape alone, for every knob value)
static inline size_t splitk_workspace_floats(int64_t T, int D, int F) {
    const size_t long_k = (size_t)gemm_f32_splitk_factor(T, D, F, true);              // fc2: [ks][T][D]
    size_t n = long_k > 1 ? long_k * (size_t)T * D : 0;
    // QKV / fc1 / out_proj / a part of them (K = D): the factor depends on the tile count of the GEMM at hand, so every width the
    // forward uses is asked (a narrower GEMM has fewer tiles and may take MORE ranges than the widest one: sizing by the widest
    // alone overran the region -- found by the forward fuzz with the knob on, seed 73)
    const int widths[5] = {D, 2 * D, 3 * D, F, 4 * D};
    for (int N : widths) {
        const size_t ks = (size_t)gemm_f32_splitk_factor(T, N, D, true);
        if (ks > 1 && ks * (size_t)T * N > n) n = ks * (size_t)T * N;
    }
    return n;
}

// the attention kernels of `batch` same-shape, unpadded MSAs in one launch each (gridDim.y = batch; MSA b's operands lie
// b * stride elements further on): row_attn.hip, col_attn.hip; used by rnamsm_forward_batch
int row_logits_batched(const float* q, const float* k, int64_t ld, float* partial, int R, int C, int H, int batch,
                       int64_t qk_bstride, int64_t part_bstride, void* stream);
// logit_scale multiplies the summed logits before the mask fill and the softmax (the exact path's 1/sqrt(R); batch = 1 with zero
// strides = one alignment)
int softmax_rows_batched(const float* partial, int nsplit, float* probs, int H, int C, int batch, int64_t part_bstride,
                         int64_t probs_bstride, const uint8_t* key_mask, int64_t mask_bstride, void* stream, float logit_scale = 1.f);
int row_apply_batched(const float* probs, const float* v, int64_t ld, float* ctx, int64_t ldc, int R, int C, int H, int batch,
                      int64_t probs_bstride, int64_t v_bstride, int64_t ctx_bstride, void* stream);
int col_attn_batched(const float* q, const float* k, const float* v, int64_t ld, float* ctx, int64_t ldc, int R, int C, int H,
                     int batch, int64_t qkv_bstride, int64_t ctx_bstride, const uint8_t* pad_mask, void* stream, bool prescaled = false);

// the same for the 16-bit modes (row_attn16.hip, row_attn.hip, col_attn16.hip): plane operands, MSA b's planes b * stride halves
// further on; true_rows (device int32 [batch], may be null): every MSA's tied logits are scaled by its own depth
int row_logits16_batched(const uint16_t* q_hi, const uint16_t* q_lo, const uint16_t* k_hi, const uint16_t* k_lo, int64_t ld,
                         float* partial, int R, int C, int H, float scale, int fmt, int batch, int64_t qk_bstride,
                         int64_t part_bstride, const int* true_rows, void* stream);
int softmax_rows_planes_batched(const float* partial, int nsplit, float* probs, uint16_t* p_hi, uint16_t* p_lo, int64_t ldp,
                                float plane_scale, int H, int C, const uint8_t* key_mask, int fmt, int batch, int64_t part_bstride,
                                int64_t probs_bstride, int64_t mask_bstride, int64_t plane_bstride, void* stream);
int row_apply16_batched(const uint16_t* p_hi, const uint16_t* p_lo, int64_t ldp, const uint16_t* v_hi, const uint16_t* v_lo, int64_t ld,
                        int64_t ldc, int R, int C, int H, float out_scale, uint16_t* ctx_hi, uint16_t* ctx_lo, int fmt, int batch,
                        int64_t p_bstride, int64_t v_bstride, int64_t ctx_bstride, void* stream);
int col_attn16_batched(const uint16_t* q_hi, const uint16_t* q_lo, const uint16_t* k_hi, const uint16_t* k_lo, const uint16_t* v_hi,
                       const uint16_t* v_lo, int64_t ld, int64_t ldc, int R, int C, int H, float scale, const uint8_t* pad_mask,
                       uint16_t* ctx_hi, uint16_t* ctx_lo, int fmt, int batch, int64_t qkv_bstride, int64_t ctx_bstride, int64_t mask_bstride,
                       void* stream, bool prescaled = false);

// ragged batches (elementwise.hip): per-token q factor (0 at <pad>, 1/sqrt(true depth of the token's MSA) elsewhere), applied
// to the q columns in the QKV GEMM's epilogue (rnamsm_gemm_row_scaled)
int ragged_row_scale(const int64_t* tokens, int pad_idx, const int* true_rows, float* out, int64_t n, int64_t tokens_per_msa,
                     hipStream_t stream);

// K0 / K10 of a batch of B same-shape alignments in ONE launch each (elementwise.hip; rnamsm_forward_batch): tokens [B,R,C] ->
// x [B*R*C, D] (row positions restart per alignment); emb / atp of alignment b from x_final + b * x_bstride and
// probs_all + b * probs_bstride.  pack_outputs sets bit 2 of *err_flag (RNAMSM_ERR_NONFINITE) when an output is inf / NaN.
int embed_ln_batched(const int64_t* tokens, const float* embed_tokens, const float* embed_positions, const float* row_pos,
                     const float* gamma, const float* beta, float* out, int B, int R, int C, int D, int vocab, int num_positions,
                     int pad_idx, float eps, int* err_flag, hipStream_t stream, int row_pos_dim = 0);
int pack_outputs_batched(const float* x_final, const float* probs_all, float* emb, float* atp, int C, int D, int num_layers, int H,
                         int B, int64_t x_bstride, int64_t probs_bstride, int* err_flag, hipStream_t stream);


// ---- token-packed batches (rnamsm_forward_packed): alignments of DIFFERENT shapes concatenated along the token axis, no padding.
// The token-parallel launches (GEMMs, LayerNorm) see one [T, D] matrix; K0, K4-K7 and K10 take the alignment from blockIdx.y and
// its shape / offsets from this descriptor (device array of B entries, written by packed_descriptors_upload), gridDim.x sized
// for the largest alignment of the batch (blocks past an alignment's own need return at once).
struct PackedMsa {           // 64 bytes
    int32_t R, C;            // the alignment's own shape
    int32_t nsplit, rows_per_split;      // its tied-logits slabs (choose_row_split of ITS shape: the summation order of its own forward)
    int64_t tok0;            // tokens of the alignments before it
    int64_t part_off;        // floats: its slabs [nsplit, H, C, C] inside the slab workspace
    int64_t probs_off;       // floats: its maps [NL, H, C, C] inside row_attn
    int64_t emb_off;         // floats: its [C-1, D] inside emb
    int64_t atp_off;         // floats: its [NL*H, C-1, C-1] inside atp
    float logit_scale;       // 1 / sqrt(R): align_scaling's depth factor, applied to the summed tied logits (modules.py:713-715)
    int32_t pad_;
};
static_assert(sizeof(PackedMsa) == 64, "PackedMsa layout");
int packed_descriptors_upload(const PackedMsa* host, int B, PackedMsa* dev, hipStream_t stream);
int embed_ln_packed(const int64_t* tokens, const float* embed_tokens, const float* embed_positions, const float* row_pos,
                    const float* gamma, const float* beta, float* out, const PackedMsa* pk, int B, int64_t T, int D, int vocab,
                    int num_positions, int pad_idx, float eps, int* err_flag, hipStream_t stream, int row_pos_dim);
int pack_outputs_packed(const float* x_final, const float* row_attn, float* emb, float* atp, const PackedMsa* pk, int B, int max_C, int D,
                        int num_layers, int H, double total_out_floats, int* err_flag, hipStream_t stream);
// K4-K7, fp32: `host` = the same descriptors on the host (grid sizing, timer bytes); `layer` selects the maps inside an alignment's probs block
int row_logits_packed(const float* q, const float* k, int64_t ld, float* partial, int H, const PackedMsa* pk, const PackedMsa* host,
                      int B, void* stream);
int softmax_rows_packed(const float* partial, float* row_attn, int layer, int H, const PackedMsa* pk, const PackedMsa* host, int B,
                        void* stream);
int row_apply_packed(const float* row_attn, int layer, const float* v, int64_t ld, float* ctx, int64_t ldc, int H, const PackedMsa* pk,
                     const PackedMsa* host, int B, void* stream);
int col_attn_packed(const float* q, const float* k, const float* v, int64_t ld, float* ctx, int64_t ldc, int H, const PackedMsa* pk,
                    const PackedMsa* host, int B, void* stream, bool prescaled = false);

}  // namespace rnamsm
