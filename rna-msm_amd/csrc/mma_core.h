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

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[mt][nt][t] = 0.f;
}

// Accumulator register t of lane (li, lh) in MFMA tile (mt, nt) of wave (wm, wn) -> tile-local (row, col).
__device__ __forceinline__ int acc_row(const WaveCoord& w, int mt, int t) {
    return w.wm * 64 + mt * 32 + (t & 3) + 8 * (t >> 2) + 4 * w.lh;
}
__device__ __forceinline__ int acc_col(const WaveCoord& w, int nt) { return w.wn * 64 + nt * 32 + w.li; }

}  // namespace rnamsm
