// 128x128x32 exact-fp32 MFMA block tile shared by the Linear GEMM (K2), tied-row logits (K4) and row apply (K6).
//
// Block = 256 threads = 4 waves as 2(M) x 2(N); each wave owns a 64x64 output tile = 2x2 MFMA tiles of 32x32
// (64 accumulator registers).  Operands are staged through LDS in [row][k] order with the k stride padded to 36
// floats (144 B): a 16-lane ds_read_b128 group then touches 16 distinct 16-B slots of the 256-B bank row
// (conflict-free, checked against MI355X_MICROARCH.md §LDS lane groups).  One ds_read_b128 gives a lane the four
// k-steps {8kk+4h+s, s=0..3} of its MFMA half h, so a 32-deep K tile costs 4 A + 4 B reads per 32x32 tile pair and
// 16 MFMAs -- fp32 MFMA (64 cycles/instruction) is 16x gentler on LDS than bf16, the pipe to keep busy is the
// matrix core.  The k order inside a tile is permuted identically for A and B, which is legal for a dot product
// and fixed, so results are deterministic.
#pragma once
#include <type_traits>

#include "common.h"

namespace rnamsm {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDK = BK + 4;                 // padded k stride (floats) of a [row][k] LDS tile
constexpr int LDN = BN + 4;                 // padded n stride (floats) of a [k][n] LDS tile (row apply's V operand)
constexpr int TILE_KC = BM * LDK;           // floats in one k-contiguous operand tile
constexpr int TILE_NC = BK * LDN;           // floats in one n-contiguous operand tile
constexpr int GEMM_THREADS = 256;

struct WaveCoord {
    int wm, wn;      // wave position in the 2x2 grid
    int li, lh;      // lane & 31, lane >> 5
};
__device__ __forceinline__ WaveCoord wave_coord() {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    return WaveCoord{w >> 1, w & 1, lane & 31, lane >> 5};
}

// One staged K tile: acc[mt][nt] += A(64 x 32) * B(32 x 64) for this wave.
// B_KC: B tile is [n][k] (k contiguous, torch Linear weight / K of QK^T); else [k][n] (V of P.V).
template <bool B_KC>
__device__ __forceinline__ void mma_ktile(const float* __restrict__ As, const float* __restrict__ Bs,
                                          f32x16 (&acc)[2][2], const WaveCoord& w) {
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
        f32x4 a[2], b[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            a[mt] = *reinterpret_cast<const f32x4*>(&As[(w.wm * 64 + mt * 32 + w.li) * LDK + kk * 8 + 4 * w.lh]);
        if (B_KC) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                b[nt] = *reinterpret_cast<const f32x4*>(&Bs[(w.wn * 64 + nt * 32 + w.li) * LDK + kk * 8 + 4 * w.lh]);
        } else {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    b[nt][s] = Bs[(kk * 8 + 4 * w.lh + s) * LDN + w.wn * 64 + nt * 32 + w.li];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma32(a[mt][s], b[nt][s], acc[mt][nt]);
    }
}

// ---- software-pipelined K loop ----------------------------------------------------------------------------------
// Fragments of one 8-deep k group: 2 A + NT B ds_read_b128 (B_KC) per lane.  NT = MFMA tiles per wave along N:
// 2 (wave tile 64x64, block tile 128x128) or 1 (wave tile 64x32, block tile 128x64 -- the half-width GEMM tile that
// evens out the last round of blocks on small problems).
template <int NT>
struct FragT {
    f32x4 a[2], b[NT];
};
using Frag = FragT<2>;
template <bool B_KC, int NT>
__device__ __forceinline__ void frag_load(const float* __restrict__ As, const float* __restrict__ Bs, int kk,
                                          const WaveCoord& w, FragT<NT>& f) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
        f.a[mt] = *reinterpret_cast<const f32x4*>(&As[(w.wm * 64 + mt * 32 + w.li) * LDK + kk * 8 + 4 * w.lh]);
    if (B_KC) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            f.b[nt] = *reinterpret_cast<const f32x4*>(&Bs[(w.wn * 32 * NT + nt * 32 + w.li) * LDK + kk * 8 + 4 * w.lh]);
    } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int s = 0; s < 4; ++s) f.b[nt][s] = Bs[(kk * 8 + 4 * w.lh + s) * LDN + w.wn * 32 * NT + nt * 32 + w.li];
    }
}
template <int NT>
__device__ __forceinline__ void frag_mma(const FragT<NT>& f, f32x16 (&acc)[2][NT]) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma32(f.a[mt][s], f.b[nt][s], acc[mt][nt]);
}

// The K loop every tile kernel runs.  `load(kt)` issues the global loads of K tile kt into the caller's staging
// registers (NV loads per thread), `store(buf)` writes those registers into LDS buffer `buf` (8 ds_write_b128).
// One iteration = four groups of 16 MFMAs (1024 matrix-pipe cycles each), scheduled so the pipe only ever waits at the
// barrier itself:
//   group 0 : fragments of group 1 requested first; then, BETWEEN its MFMAs, the 8 LDS writes of tile t+1 (whose
//             global loads were issued a whole iteration earlier) and the global loads of tile t+2
//   group 1 : fragments of group 2 in flight          group 2 : fragments of group 3 in flight
//   barrier : every wave has its group-3 fragments in registers, tile t+1 is complete in the other buffer
//   group 3 : covers the LDS latency of tile t+1's group-0 fragments, requested right after the barrier
// hipcc otherwise sinks each ds_read next to its use and exposes the LDS latency four times per tile, so the order is
// pinned with sched_group_barrier (interleave inside a group) and sched_barrier (between groups); the steady-state
// body is branch-free (tail iterations are peeled) because those directives act per basic block.
// Buffer hazards: tile t+1's buffer was last read before the previous iteration's barrier; nobody reads the current
// buffer after this iteration's barrier.
#define RNAMSM_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
constexpr int SG_MFMA = 0x8, SG_VMEM_READ = 0x20, SG_DS_READ = 0x100, SG_DS_WRITE = 0x200;

template <int S>
using SetTag = std::integral_constant<int, S>;

// One K tile.  S = staging register set holding tile kt+1 (and, once stored, reloaded with tile kt+1+DEPTH).
// STEADY: all three actions happen and the schedule is pinned; otherwise (the last DEPTH+1 tiles) they are runtime
// flags and the compiler's own order is accepted.
template <bool B_KC, int NV, int DEPTH, int S, bool STEADY, int NT, int NWR_, class LoadFn, class StoreFn>
__device__ __forceinline__ void kstep(int kt, bool do_store, bool do_load, bool do_next, float* As, float* Bs,
                                      int a_tile, int b_tile, f32x16 (&acc)[2][NT], const WaveCoord& w, FragT<NT>& f0,
                                      FragT<NT>& f1, LoadFn& load, StoreFn& store) {
    constexpr int NR = B_KC ? 2 + NT : 2 + 4 * NT;     // ds_reads per fragment group
    constexpr int NM = 8 * NT;                          // MFMAs per group
    constexpr int NWR = NWR_ > 0 ? NWR_ : 4 + 2 * NT;   // LDS writes per staged tile (default: A 128 rows + B 64*NT rows, b128)
    constexpr int WPS = NWR > NM - 2 ? 2 : 1;           // writes per MFMA slot of group 0
    constexpr int WSL = (NWR + WPS - 1) / WPS;          // MFMA slots that carry writes
    constexpr int LG = (NM - WSL) < 4 ? (NM - WSL) : 4; // MFMA slots that carry the global loads
    const int cur = kt & 1, nxt = cur ^ 1;
    const float* Ac = As + cur * a_tile;
    const float* Bc = Bs + cur * b_tile;
    // ---- group 0
    frag_load<B_KC, NT>(Ac, Bc, 1, w, f1);
    if (STEADY || do_store) store(nxt, SetTag<S>{});
    if (STEADY || do_load) load(kt + 1 + DEPTH, SetTag<S>{});
    frag_mma<NT>(f0, acc);
    if (STEADY) {
        RNAMSM_SGB(SG_DS_READ, NR);
#pragma unroll
        for (int i = 0; i < WSL; ++i) {
            RNAMSM_SGB(SG_MFMA, 1);
            RNAMSM_SGB(SG_DS_WRITE, WPS);
        }
#pragma unroll
        for (int i = 0; i < LG; ++i) {
            RNAMSM_SGB(SG_MFMA, 1);
            RNAMSM_SGB(SG_VMEM_READ, (NV + LG - 1) / LG);
        }
        RNAMSM_SGB(SG_MFMA, NM - WSL - LG);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- group 1
    frag_load<B_KC, NT>(Ac, Bc, 2, w, f0);
    frag_mma<NT>(f1, acc);
    RNAMSM_SGB(SG_DS_READ, NR);
    RNAMSM_SGB(SG_MFMA, NM);
    __builtin_amdgcn_sched_barrier(0);
    // ---- group 2
    frag_load<B_KC, NT>(Ac, Bc, 3, w, f1);
    frag_mma<NT>(f0, acc);
    RNAMSM_SGB(SG_DS_READ, NR);
    RNAMSM_SGB(SG_MFMA, NM);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    // ---- group 3
    if (STEADY || do_next) frag_load<B_KC, NT>(As + nxt * a_tile, Bs + nxt * b_tile, 0, w, f0);
    frag_mma<NT>(f1, acc);
    __builtin_amdgcn_sched_barrier(0);
}

// DEPTH = 1: one staging register set, tile t+2 is requested during tile t (distance ~52 MFMAs to its LDS write).
// DEPTH = 2: two sets (tile t+1 lives in set (t+1)&1), tile t+3 is requested during tile t: a whole extra tile
//            (64 MFMAs) of latency cover for first-touch HBM reads at +32 VGPRs.
// `load(kt, SetTag<S>)` / `store(buf, SetTag<S>)` address the caller's staging registers by a compile-time set index
// (runtime-indexed register arrays would go to scratch), hence the x2 unrolled loops.
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
    __device__ __forceinline__ void operator()(int) const {}
};
// `after_first_tile()` runs once, after tile 0 is in LDS and tile 1's loads are in flight and before the first barrier: the
// place for one-time work whose own global loads were issued before the loop (they return ahead of tile 0's) and whose
// LDS results the epilogue needs.
// `after_tiles(done)` runs after every pair of K tiles (done = tiles finished so far, even, or nk at the very end): the
// place to fold the accumulators into a second set at fixed K boundaries (row_logits: bounded fp32 accumulation chains).
template <bool B_KC, int NV, int DEPTH, int NT = 2, int NWR = 0, class LoadFn, class StoreFn, class HookFn = NoHook,
          class TilesFn = NoHook>
__device__ __forceinline__ void pipelined_kloop(int nk, float* As, float* Bs, int a_tile, int b_tile,
                                                f32x16 (&acc)[2][NT], const WaveCoord& w, LoadFn load, StoreFn store,
                                                HookFn after_first_tile = HookFn{}, TilesFn after_tiles = TilesFn{}) {
    static_assert(DEPTH == 1 || DEPTH == 2, "prefetch depth");
    constexpr int S_ODD = DEPTH == 2 ? 1 : 0;       // set of odd tiles
    load(0, SetTag<0>{});
    store(0, SetTag<0>{});
    if (nk > 1) load(1, SetTag<S_ODD>{});
    if (DEPTH == 2 && nk > 2) load(2, SetTag<0>{});
    after_first_tile();
    __syncthreads();
    FragT<NT> f0, f1;
    frag_load<B_KC, NT>(As, Bs, 0, w, f0);
    int kt = 0;
    for (; kt + 2 + DEPTH < nk; kt += 2) {           // both steps are full: (kt + 1) + 1 + DEPTH < nk
        kstep<B_KC, NV, DEPTH, S_ODD, true, NT, NWR>(kt, true, true, true, As, Bs, a_tile, b_tile, acc, w, f0, f1, load, store);
        kstep<B_KC, NV, DEPTH, 0, true, NT, NWR>(kt + 1, true, true, true, As, Bs, a_tile, b_tile, acc, w, f0, f1, load, store);
        after_tiles(kt + 2);
    }
#pragma unroll 1
    for (; kt < nk; kt += 2) {                       // kt even: tile kt+1 is odd -> set S_ODD
        kstep<B_KC, NV, DEPTH, S_ODD, false, NT, NWR>(kt, kt + 1 < nk, kt + 1 + DEPTH < nk, kt + 1 < nk, As, Bs, a_tile, b_tile,
                                                 acc, w, f0, f1, load, store);
        if (kt + 1 < nk)
            kstep<B_KC, NV, DEPTH, 0, false, NT, NWR>(kt + 1, kt + 2 < nk, kt + 2 + DEPTH < nk, kt + 2 < nk, As, Bs, a_tile,
                                                 b_tile, acc, w, f0, f1, load, store);
        after_tiles(kt + 2 < nk ? kt + 2 : nk);
    }
}

// Register staging of one [128 rows][32 k] tile: thread -> (row = tid/8 + 32*i, 16-B chunk tid%8); 8 lanes cover
// one row's 128 contiguous bytes.
struct StageKC {
    f32x4 v[4];
};
__device__ __forceinline__ void stage_store_kc(float* tile, const StageKC& s) {
    const int c4 = threadIdx.x & 7, r0 = threadIdx.x >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(&tile[(r0 + 32 * i) * LDK + c4 * 4]) = s.v[i];
}

template <int NT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][NT]) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[mt][nt][t] = 0.f;
}

// Accumulator register t of lane (li, lh) in MFMA tile (mt, nt) of wave (wm, wn) -> tile-local (row, col).
__device__ __forceinline__ int acc_row(const WaveCoord& w, int mt, int t) {
    return w.wm * 64 + mt * 32 + (t & 3) + 8 * (t >> 2) + 4 * w.lh;
}
__device__ __forceinline__ int acc_col(const WaveCoord& w, int nt) { return w.wn * 64 + nt * 32 + w.li; }

}  // namespace rnamsm
