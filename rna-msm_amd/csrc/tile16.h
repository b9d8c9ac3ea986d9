// LDS tiles of 16-bit operand planes for the 16-bit attention kernels: DMA staging, swizzles, fragment reads.
//
// A tile is [rows][128 B] (64 halves per row), filled by global_load_lds_dwordx4: one wave instruction moves 8 rows
// x 128 B and writes LDS linearly (lane l -> row l/8, physical 16-B chunk l%8), so a tile is unpadded and the bank
// spread comes from an XOR swizzle applied to the SOURCE chunk and to the READ (cdna_hip_programming.md rule 21).
// Two read patterns, two swizzles:
//   "k" tiles  rows = the operand's m/n index, 64 consecutive k per row; read with ds_read_b128 (lane = row, 8 k);
//              physical chunk = logical ^ ((row >> 1) & 7): the 16 rows of a lane group hit 16 distinct 16-B slots.
//   "t" tiles  rows = k, columns = the operand's m/n index (V: rows = keys, columns = head dims); read with
//              ds_read_b64_tr_b16, which hands lane (col, half) four consecutive ROWS of its column -- the transposed
//              fragment with no LDS transpose pass.  Lane semantics verified on hardware (tools/probes/tr16_probe.hip):
//              per 16-lane group, lane 4q+p supplies the address of row q, columns 4p..4p+3 of a 4x16 block and lane i
//              receives column i, row q in element q.  A half-wave touches 4 rows x 64 B; rows r and r+2 share banks,
//              so physical chunk = logical ^ (((row >> 1) & 1) << 2) moves them to different chunk quads.
#pragma once
#include "half16.h"

namespace rnamsm {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int T16_ROWB = 128;                       // bytes per tile row

__device__ __forceinline__ int swz_k(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int swz_t(int row) { return ((row >> 1) & 1) << 2; }

// per-lane DMA geometry of a wave instruction covering tile rows [8g, 8g+8): row = 8g + lane/8, and the LOGICAL chunk
// this lane must fetch so that physical chunk lane%8 holds it.  (row>>1)&7 = (4g + lane/16) & 7, (row>>1)&1 = (lane/16)&1.
__device__ __forceinline__ int dma_chunk_k(int lane, int g) { return (lane & 7) ^ ((4 * g + (lane >> 4)) & 7); }
__device__ __forceinline__ int dma_chunk_t(int lane) { return (lane & 7) ^ (((lane >> 4) & 1) << 2); }

__device__ __forceinline__ void dma16(const uint16_t* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

// k tile -> MFMA 32x32x16 operand: lane (row, half lh) gets k = 16 kk + 8 lh + 0..7
template <int FMT>
__device__ __forceinline__ typename Half16<FMT>::V8 frag_k(const char* tile, int row, int kk, int lh) {
    return *reinterpret_cast<const typename Half16<FMT>::V8*>(tile + row * T16_ROWB + (((2 * kk + lh) ^ swz_k(row)) << 4));
}

// t tile -> MFMA 32x32x16 operand: elements 0..3 = rows row_a - q .. +3, elements 4..7 = rows row_b - q .. +3 of the
// lane's column, where q = (lane & 15) >> 2 is already folded into row_a / row_b and col = the 4-column group this
// lane ADDRESSES (block column base + 4 * (lane & 3)); the lane RECEIVES column base + (lane & 15).
template <int FMT>
__device__ __forceinline__ typename Half16<FMT>::V8 frag_t(const char* tile, int row_a, int row_b, int col) {
    typedef __attribute__((address_space(3))) s16x4* lp;
    const char* pa = tile + row_a * T16_ROWB + (((col >> 3) ^ swz_t(row_a)) << 4) + (col & 7) * 2;
    const char* pb = tile + row_b * T16_ROWB + (((col >> 3) ^ swz_t(row_b)) << 4) + (col & 7) * 2;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)pa);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)pb);
    const s16x8 r = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(typename Half16<FMT>::V8, r);
}

// The same transposed read as inline asm, in two steps.  hipcc treats the ds_read_tr16 builtin as a possible LDS STORE and puts
// an s_waitcnt vmcnt(0) in front of it whenever an LDS-DMA may be in flight (found in round 4 in the ISA of every kernel that
// mixes frag_t with a DMA ring: the prefetched tile was drained right after it had been requested, so no transfer ever
// overlapped the MFMAs).  An asm statement is invisible to that bookkeeping: tr16_issue() requests the 4-row pieces, and
// tr16_wait*() -- an s_waitcnt lgkmcnt(0) naming every piece as "+v", placed before the first consumer -- makes them usable
// (cdna_hip_programming.md 5.7 item 1, form (ii)).  LDS returns data in order, so hipcc's own counted lgkmcnt waits can only
// over-wait because of these requests, never under-wait.
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
struct TrPieces { s16x4 a, b; };
__device__ __forceinline__ TrPieces tr16_issue(const char* tile, int row_a, int row_b, int col) {
    const uint32_t pa = lds_addr_of(tile + row_a * T16_ROWB + (((col >> 3) ^ swz_t(row_a)) << 4) + (col & 7) * 2);
    const uint32_t pb = lds_addr_of(tile + row_b * T16_ROWB + (((col >> 3) ^ swz_t(row_b)) << 4) + (col & 7) * 2);
    TrPieces r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r.a) : "v"(pa));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r.b) : "v"(pb));
    return r;
}
// the same from a per-lane LDS byte address `va` (row / swizzled column inside an 8-row group, computed once) plus offsets that are
// multiples of 8 rows: CT compile-time, `off` a constant after unrolling; the second 4-row... block of the fragment lies 8 rows on
template <int CT>
__device__ __forceinline__ void tr16_issue_at(uint32_t va, int off, TrPieces& r) {
    const uint32_t a = va + (uint32_t)(CT + off);
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r.a) : "v"(a));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(r.b) : "v"(a));
}
__device__ __forceinline__ void tr16_wait4(TrPieces& p0, TrPieces& p1, TrPieces& p2, TrPieces& p3) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(p0.a), "+v"(p0.b), "+v"(p1.a), "+v"(p1.b), "+v"(p2.a), "+v"(p2.b), "+v"(p3.a), "+v"(p3.b));
}
// counted form for software-pipelined fragment sets: N = transposed reads (asm) requested AFTER these pieces and before this wait
template <int N>
__device__ __forceinline__ void tr16_wait2(TrPieces& p0, TrPieces& p1) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(p0.a), "+v"(p0.b), "+v"(p1.a), "+v"(p1.b) : "n"(N));
}
template <int FMT>
__device__ __forceinline__ typename Half16<FMT>::V8 tr16_frag(const TrPieces& p) {
    const s16x8 r = __builtin_shufflevector(p.a, p.b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(typename Half16<FMT>::V8, r);
}

// acc += A.B with hi/lo operand pairs (SPLIT 3: small cross terms first, the leading term last) or hi only (SPLIT 1)
template <int SPLIT, int FMT>
__device__ __forceinline__ f32x16 mma16(const typename Half16<FMT>::V8 (&a)[SPLIT == 3 ? 2 : 1],
                                        const typename Half16<FMT>::V8 (&b)[SPLIT == 3 ? 2 : 1], f32x16 acc) {
    if (SPLIT == 3) {
        acc = Half16<FMT>::mfma(a[1], b[0], acc);
        acc = Half16<FMT>::mfma(a[0], b[1], acc);
    }
    return Half16<FMT>::mfma(a[0], b[0], acc);
}

}  // namespace rnamsm
