// Plain-bf16 Linear GEMM with the epilogue HIDDEN under the next tile's K loop ("ping-pong accumulators").
//
// Why: the 256x256 kernels of gemm_bf16.hip keep one block of 8 waves x 256 registers
// per CU -- the whole register file holds ONE output tile, so nothing is resident to overlap a tile's epilogue: 19-45 % of a
// bf16 launch is an exposed conversion + store burst, with all 256 CUs bursting at once (the burst itself runs at the
// chip's HBM write rate).  AMD hardware gives every wave of a kernel the same register allocation, so "store waves that own
// no accumulators" cannot be had inside one block.  What can be had is a wave that owns TWO accumulator sets:
//
//   block = 4 waves, ONE per SIMD, up to 512 registers per wave; block tile 256 x 128, wave tile 128 x 64 (128 accumulator
//   registers); set `cur` accumulates tile t while set `prev` -- the finished tile t-1 -- leaves in eight 16-row units, one per
//   K iteration: a unit's bias / residual pieces are requested at iteration u, its conversion + stores are issued at
//   iteration u+2, between the MFMAs.  The stores never burst (2-4 per wave per ~1000 cycles) and no wave ever waits for them.
//
// K loop: v_mfma_f32_16x16x32_bf16 with swapped operands and the W rows permuted on their way into LDS (the register-direct
// epilogue layout of gemm16_q16_kernel), BK = 64, a THREE-slot LDS-DMA ring (3 x 48 KB) with two operand tiles in flight,
// one barrier per K tile placed after the tile's last fragment read (as gemm16_swp_kernel), and the operand stream runs on
// across tile boundaries (the next tile's first K tiles are requested while the current tile finishes), so a block never
// waits for a first operand tile after its first.
//
// Counting (MI355X_MICROARCH.md: loads, stores and LDS-DMA retire in issue order on one counter): every barrier waits with a
// LITERAL vmcnt = the number of vector-memory instructions issued after the DMA it needs; the epilogue's loads are inline
// asm (hipcc would otherwise drain the DMA ring with vmcnt(0) at their first use) into VGPR temporaries that only the
// statement following the covering barrier wait names ("+v"), and everything between request and use is straight-line code.
// Results: the same products in the same order per accumulator as gemm16_q16_kernel, the same epilogue arithmetic -> bit-identical
// to it.
#include <type_traits>

#include "half16.h"

namespace rnamsm {

// PP_SPREAD 0: the LDS-DMA requests of a K tile go out in one burst after the barrier; 1: between the MFMAs of that half
// (measured no faster -- a request's issue time is matrix-pipe idle time wherever it sits, see the header -- and fc2 slower)
#ifndef PP_SPREAD
#define PP_SPREAD 0
#endif

namespace pp {
constexpr int BM = 256, BN = 128, BK = 64, THREADS = 256;
constexpr int ROWB = BK * 2;                    // 128-byte tile rows (whole cache lines)
constexpr int A_BYTES = BM * ROWB;              // 32 KB
constexpr int W_BYTES = BN * ROWB;              // 16 KB
constexpr int STAGE = A_BYTES + W_BYTES;        // 48 KB
constexpr int NSLOT = 3;
constexpr int LDS = NSLOT * STAGE;              // 144 KB
constexpr int NDMA = (A_BYTES + W_BYTES) / (4 * 1024);   // LDS-DMA instructions per wave per K tile (1 KB each, 4 waves) = 12
constexpr int NSPECIAL = 12;                    // unrolled iterations at the head of every tile's K loop
constexpr int UNITS = 8;                        // 16-row units of a wave's 128-row tile

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 V8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// epilogue memory operations of iteration i of a tile's K loop while the previous tile drains: loads of unit i (and the
// bias at i == 0) at i = 0..7, stores of unit i - 2 at i = 2..9
template <bool HAS_RES>
constexpr int n_loads(int i, bool draining) { return !draining ? 0 : ((i == 0 ? 4 : 0) + ((HAS_RES && i >= 0 && i < UNITS) ? 4 : 0)); }
template <bool O_PL>
constexpr int n_stores(int i, bool draining) { return (draining && i >= 2 && i < UNITS + 2) ? (O_PL ? 2 : 4) : 0; }
// vmcnt literal of iteration kt's barrier: it needs the operand tile requested in iteration kt-2 (order inside an
// iteration: loads, DMA, stores); younger than that DMA are the stores of kt-2 and everything of kt-1
template <bool HAS_RES, bool O_PL>
constexpr int wait_count(int kt, bool draining) {
    return NDMA + n_stores<O_PL>(kt - 2, draining) + n_loads<HAS_RES>(kt - 1, draining) + n_stores<O_PL>(kt - 1, draining);
}

typedef const __attribute__((address_space(3))) char* lds_cptr;
typedef const __attribute__((address_space(3))) V8* lds_v8ptr;

// One LDS-DMA request of a wave (64 lanes x 16 B, written linearly from the wave-uniform LDS byte address lds_dst): global
// address = wave-uniform 64-bit base (scalar registers) + 32-bit per-lane byte offset -- no vector address arithmetic.  M0
// (the destination base) is compiler-reserved: saved, written and restored inside the one statement
// (cdna_hip_programming.md 5.7).  Invisible to hipcc's s_waitcnt bookkeeping: completion is counted by hand.
__device__ __forceinline__ void dma16_sv(const char* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
}
__device__ __forceinline__ void asm_load16(f32x4v& dst, const float* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm_lgkm_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}
}  // namespace pp

__device__ f32x4 g_pp_sink[128];               // where the lanes of rows past M store (never read)

template <int ACT, bool HAS_RES, bool O_PL>
__global__ __launch_bounds__(pp::THREADS, 1) void gemm16_pp_kernel(
    const uint16_t* __restrict__ Ahi, int64_t lda, const uint16_t* __restrict__ Whi, const float* __restrict__ bias,
    const float* residual, int64_t ldr, float* Cout, int64_t ldc, int M, int N, int K, float scale, int scale_cols,
    uint16_t* __restrict__ Ohi, int group, unsigned total_tiles) {
    using namespace pp;
    extern __shared__ __attribute__((aligned(16))) char smem_b[];

    const unsigned nb = N / BN, mp = (M + BM - 1) / BM;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wv >> 1, wn = wv & 1, fr = lane & 15, fq = lane >> 4;
    const int nk = K / BK;                                       // multiple of 3, >= NSPECIAL (checked by the launcher)

    auto find_tile = [&](unsigned& vid, int& m0, int& n0) __attribute__((always_inline)) -> bool {
        for (; vid < total_tiles; vid += gridDim.x) {
            unsigned mpanel, nblk;
            if (xcd_panel_map_grouped(vid, mp, nb, (unsigned)group, mpanel, nblk)) {
                m0 = mpanel * BM;
                n0 = nblk * BN;
                return true;
            }
        }
        return false;
    };

    // ---- operand stream: K tiles of consecutive output tiles, requested NSLOT-1 .. NSLOT ahead of their use
    // DMA map (BK = 64): a wave instruction covers 8 tile rows of 128 B; wave w moves row groups w, w+4, ...; lane -> (row 8g +
    // lane/8, physical chunk lane%8) fetching logical chunk (lane%8) ^ ((row>>1)&7), (row>>1)&7 = (4 (w&1) + lane/16) & 7
    const int drow = lane >> 3;
    const int dchunk = (lane & 7) ^ ((4 * (wv & 1) + (lane >> 4)) & 7);
    // Addresses = a wave-uniform 64-bit base (the tile's operand panel at the stream's K position: scalar registers, advanced
    // by scalar adds) + a 32-bit per-lane byte offset inside the panel (< 2 MB): no 64-bit vector arithmetic per request.
    // W rows are PERMUTED on their way into LDS so that the (transposed) accumulators of a lane are 8 + 8 consecutive output
    // columns: LDS row 64 g + 16 t + 4 a + b  <-  weight row 64 g + 32 (t >> 1) + 8 a + 4 (t & 1) + b  (tile-independent).
    unsigned avoff[8], wvoff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 8 * (wv + 4 * j) + drow;
        const int tt = (row >> 4) & 3, aa = (row >> 2) & 3;
        const int wrow = (row & ~63) + 32 * (tt >> 1) + 8 * aa + 4 * (tt & 1) + (row & 3);
        wvoff[j] = (unsigned)wrow * (unsigned)K * 2u + dchunk * 16;
    }
    const char* a_stream = nullptr;                              // uniform: Ahi + (m0 * lda + ikt * BK) halves
    const char* w_stream = nullptr;                              // uniform: Whi + (n0 * K + ikt * BK) halves
    int ikt = 0;                                                 // the stream's next K tile within its output tile
    // the output tile AFTER the one being accumulated (found once per tile): where the stream goes next
    int nm0 = 0, nn0 = 0;
    bool have_next = false;
    auto set_offsets = [&](int m0, int n0) __attribute__((always_inline)) {
        const int last = M - 1 - m0;                             // rows past M are clamped: they only feed discarded output rows
#pragma unroll
        for (int j = 0; j < 8; ++j) avoff[j] = (unsigned)min(8 * (wv + 4 * j) + drow, last) * (unsigned)lda * 2u + dchunk * 16;
        a_stream = reinterpret_cast<const char*>(Ahi) + (int64_t)m0 * lda * 2;
        w_stream = reinterpret_cast<const char*>(Whi) + (int64_t)n0 * K * 2;
    };
    // request the stream's next K tile into ring slot `slot` (always NDMA instructions).  SWITCH: this request may be the
    // last K tile of the stream's output tile (it is at iteration nk - 4 of the tile being accumulated: iteration 8 for
    // nk = 12, an iteration = 2 mod 3 of the steady loop for nk >= 18) -- the stream then moves to the next output tile, or,
    // past the block's last tile, keeps requesting that last K tile (never read) so that the literal wait counts stay exact.
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem_b;      // LDS byte address of the ring
    auto issue_piece = [&](int slot, int j) __attribute__((always_inline)) {     // piece j of NDMA: 8 of the A tile, then 4 of W
        const unsigned base = lds0 + slot * STAGE;
        if (j < 8) dma16_sv(a_stream, avoff[j], base + (8 * (wv + 4 * j)) * ROWB);
        else dma16_sv(w_stream, wvoff[j - 8], base + A_BYTES + (8 * (wv + 4 * (j - 8))) * ROWB);
    };
    auto stream_advance = [&](auto SWITCH_) __attribute__((always_inline)) {
        if (decltype(SWITCH_)::value && ikt + 1 == nk) {
            if (have_next) {
                ikt = 0;
                set_offsets(nm0, nn0);
            }
        } else if (ikt + 1 < nk) {
            ++ikt;
            a_stream += ROWB;
            w_stream += ROWB;
        }
    };
    auto issue_next = [&](int slot, auto SWITCH_) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NDMA; ++j) issue_piece(slot, j);
        stream_advance(SWITCH_);
    };

    // ---- fragments: lane (row fr, k-group fq) of a 16-row tile reads logical chunk 4 ks + fq of its row
    struct Frags {
        V8 a[8], b[4];
    };
    // Read addresses = one lane-dependent LDS pointer per (operand, k-step, ring half) + an IMMEDIATE (slot, 16-row tile): the
    // pointers are made opaque once per iteration so that hipcc folds the constants into the ds_read offset fields instead of
    // hoisting 72 loop-invariant full addresses into registers (which spilled).  Immediates stay below 64 KB: slots 0 / 1 from
    // the low pointer (<= 49152 + 14336), slot 2 from the high one.
    lds_cptr pa[2][2], pb[2][2];                                 // [k-step][0: slots 0, 1 | 1: slot 2]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int chunk = ((4 * ks + fq) ^ ((fr >> 1) & 7)) * 16;
        pa[ks][0] = (lds_cptr)smem_b + (wm * 128 + fr) * ROWB + chunk;
        pa[ks][1] = pa[ks][0] + 2 * STAGE;
        pb[ks][0] = (lds_cptr)smem_b + A_BYTES + (wn * 64 + fr) * ROWB + chunk;
        pb[ks][1] = pb[ks][0] + 2 * STAGE;
    }
    auto load_frags = [&](auto SLOT_, auto KS_, Frags& f) __attribute__((always_inline)) {
        constexpr int slot = decltype(SLOT_)::value, ks = decltype(KS_)::value;
        constexpr int hi = slot == 2, sofs = slot == 1 ? STAGE : 0;
#pragma unroll
        for (int t = 0; t < 8; ++t) f.a[t] = *reinterpret_cast<lds_v8ptr>(pa[ks][hi] + sofs + t * 16 * ROWB);
#pragma unroll
        for (int t = 0; t < 4; ++t) f.b[t] = *reinterpret_cast<lds_v8ptr>(pb[ks][hi] + sofs + t * 16 * ROWB);
    };
    auto opaque_bases = [&]() __attribute__((always_inline)) {
        asm volatile("" : "+v"(pa[0][0]), "+v"(pa[0][1]), "+v"(pa[1][0]), "+v"(pa[1][1]), "+v"(pb[0][0]), "+v"(pb[0][1]),
                     "+v"(pb[1][0]), "+v"(pb[1][1]));
    };
    typedef f32x4v Acc[8][4];
    // operands swapped: the tile comes out TRANSPOSED in the registers -- lane (fr, fq) holds row fr, columns 4 fq .. + 3 of
    // each 16x16 tile -- so the epilogue stores row pieces straight from the accumulators
    auto mma = [&](const Frags& f, Acc& acc) __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.b[nt], f.a[mt], acc[mt][nt], 0, 0, 0);
    };
    auto mma_first = [&](const Frags& f, Acc& acc) __attribute__((always_inline)) {             // first k-step of a tile: C = 0, no zeroing pass
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.b[nt], f.a[mt], f32x4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    };

    // ---- epilogue of one 16-row unit, straight from the accumulators (the arithmetic of hq_epilogue)
    // column of accumulator element e of tile t: gnb + 32 (t >> 1) + 8 fq + 4 (t & 1) + e
    struct EpiCtx {
        int gm0, gnb;                                            // origin of the wave's 128 x 64 tile
    };
    auto unit_request = [&](const EpiCtx& c, int u, f32x4v (&b4)[4], f32x4v (&r4)[4]) __attribute__((always_inline)) {      // inline-asm loads: see the header
        if (u == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t) asm_load16(b4[t], bias + c.gnb + 32 * (t >> 1) + 8 * fq + 4 * (t & 1));
        }
        if (HAS_RES) {
            const int rowc = min(c.gm0 + u * 16 + fr, M - 1);
            const float* rp = residual + (int64_t)rowc * ldr + c.gnb + 8 * fq;
            asm_load16(r4[0], rp);
            asm_load16(r4[1], rp + 4);
            asm_load16(r4[2], rp + 32);
            asm_load16(r4[3], rp + 36);
        }
    };
    auto unit_finish = [&](const EpiCtx& c, int u, const f32x4v (&acc)[4], const f32x4v (&b4)[4], const f32x4v (&r4)[4]) __attribute__((always_inline)) {
        const int row = c.gm0 + u * 16 + fr;
        f32x4v v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float fs = c.gnb + 32 * (t >> 1) + 8 * fq + 4 * (t & 1) < scale_cols ? scale : 1.f;          // scale_cols % 4 == 0
#pragma unroll
            for (int e = 0; e < 4; ++e) v[t][e] = (acc[t][e] + b4[t][e]) * fs;                  // hq_epilogue's arithmetic without the fold
            if (ACT == RNAMSM_ACT_GELU_ERF) {
                const f32x2 g0 = gelu_erf2(f32x2{v[t][0], v[t][1]}), g1 = gelu_erf2(f32x2{v[t][2], v[t][3]});
                v[t] = f32x4v{g0[0], g0[1], g1[0], g1[1]};
            }
            if (HAS_RES) v[t] += r4[t];
        }
        // Rows past M (ragged last row panel) must STILL issue their store instructions: the barriers wait with literal
        // vmcnt counts, and an exec-masked-out store that the compiler branches around would make a count too large (a wait too
        // short: a race that showed as rare wrong tiles at M = 9000).  Such lanes write to a sink instead.
        {
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                const int64_t o = (int64_t)row * ldc + c.gnb + 32 * hlf + 8 * fq;
                if (O_PL) {
                    typedef __bf16 H8 __attribute__((ext_vector_type(8)));
                    H8 hi;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hi[e] = (__bf16)v[2 * hlf][e];
                        hi[4 + e] = (__bf16)v[2 * hlf + 1][e];
                    }
                    H8* dst = row < M ? reinterpret_cast<H8*>(Ohi + o) : reinterpret_cast<H8*>(g_pp_sink) + lane;
                    epi_store(dst, hi);
                } else {
                    f32x4* dst = row < M ? reinterpret_cast<f32x4*>(Cout + o) : g_pp_sink + 2 * lane;
                    epi_store(dst, (f32x4)v[2 * hlf]);
                    epi_store(dst + 1, (f32x4)v[2 * hlf + 1]);
                }
            }
        }
    };

    // ---- one K iteration of the tile being accumulated (slot = kt % 3, static).  F0 holds k-step 0 of this K tile on entry
    // and k-step 0 of the next one on exit.  DRAIN: the previous tile's unit I - 2 leaves, unit I's inputs are requested.
    Frags F0, F1;
    f32x4v b4[4], r4[3][4];
    auto k_iter = [&](auto I_, auto DRAIN_, auto SLOT_, Acc& cur, Acc& prev, const EpiCtx& pc) __attribute__((always_inline)) {
        constexpr int I = decltype(I_)::value;                   // position in the tile's K loop (I >= NSPECIAL: steady state)
        constexpr bool DRAIN = decltype(DRAIN_)::value;
        constexpr int S = decltype(SLOT_)::value;                // ring slot
        opaque_bases();
        load_frags(std::integral_constant<int, S>{}, std::integral_constant<int, 1>{}, F1);
        if constexpr (I == 0) mma_first(F0, cur); else mma(F0, cur);
        __builtin_amdgcn_sched_barrier(0);
        // all waves have finished reading slot S once F1 has arrived; the K tile after this one (requested two iterations
        // ago) has landed -- every wave's share
        wait_vm_lgkm_barrier<wait_count<HAS_RES, O_PL>(I < NSPECIAL ? I : NSPECIAL, DRAIN)>();
        if constexpr (DRAIN && I < UNITS) unit_request(pc, I, b4, r4[I % 3]);
        load_frags(std::integral_constant<int, (S + 1) % NSLOT>{}, std::integral_constant<int, 0>{}, F0);
        // The NDMA requests of the next-but-two K tile go out BETWEEN the MFMAs of this half (one per 2-3 MFMAs), not in a
        // burst ahead of them: with one wave per SIMD a request's issue time is matrix-pipe idle time, and back-to-back
        // requests queue behind each other in the memory pipeline (round 3, first version: 3500 cycles per iteration for 1024
        // cycles of MFMA).
        if constexpr (PP_SPREAD) {
#pragma unroll
            for (int j = 0; j < NDMA; ++j) {
                issue_piece(S, j);
#pragma unroll
                for (int q = (32 * j) / NDMA; q < (32 * (j + 1)) / NDMA; ++q)
                    cur[q >> 2][q & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F1.b[q & 3], F1.a[q >> 2], cur[q >> 2][q & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            stream_advance(std::integral_constant<bool, I == 8 || I == 14>{});
        } else {
            issue_next(S, std::integral_constant<bool, I == 8 || I == 14>{});
            __builtin_amdgcn_sched_barrier(0);
            mma(F1, cur);
        }
        if constexpr (DRAIN && I >= 2 && I < UNITS + 2) {
            constexpr int U = I - 2;
            // the unit's inputs were requested two iterations ago, before that iteration's DMA: this iteration's barrier wait
            // covered them.  From here on the compiler may read them.
            if constexpr (U == 0) asm volatile("" : "+v"(b4[0]), "+v"(b4[1]), "+v"(b4[2]), "+v"(b4[3]));
            if constexpr (HAS_RES) asm volatile("" : "+v"(r4[U % 3][0]), "+v"(r4[U % 3][1]), "+v"(r4[U % 3][2]), "+v"(r4[U % 3][3]));
            unit_finish(pc, U, prev[U], b4, r4[U % 3]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto tile_body = [&](auto DRAIN_, Acc& cur, Acc& prev, const EpiCtx& pc) __attribute__((always_inline)) {
#define PP_IT(I_) k_iter(std::integral_constant<int, I_>{}, DRAIN_, std::integral_constant<int, (I_) % 3>{}, cur, prev, pc)
        PP_IT(0); PP_IT(1); PP_IT(2); PP_IT(3); PP_IT(4); PP_IT(5); PP_IT(6); PP_IT(7); PP_IT(8); PP_IT(9); PP_IT(10); PP_IT(11);
        for (int g = NSPECIAL / 3; g < nk / 3; ++g) {
            PP_IT(12); PP_IT(13); PP_IT(14);
        }
#undef PP_IT
    };
    // the block's last tile has no successor to hide under: its units leave one after the other (compiler-counted loads)
    auto tail_epilogue = [&](Acc& acc, const EpiCtx& c) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        f32x4v tb[4], tr[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) tb[t] = *reinterpret_cast<const f32x4v*>(bias + c.gnb + 32 * (t >> 1) + 8 * fq + 4 * (t & 1));
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            if (HAS_RES) {
                const int rowc = min(c.gm0 + u * 16 + fr, M - 1);
                const float* rp = residual + (int64_t)rowc * ldr + c.gnb + 8 * fq;
                tr[0] = epi_load(reinterpret_cast<const f32x4v*>(rp));
                tr[1] = epi_load(reinterpret_cast<const f32x4v*>(rp + 4));
                tr[2] = epi_load(reinterpret_cast<const f32x4v*>(rp + 32));
                tr[3] = epi_load(reinterpret_cast<const f32x4v*>(rp + 36));
            }
            unit_finish(c, u, acc[u], tb, tr);
        }
    };

    // ---- the block's tile walk
    unsigned vid = blockIdx.x;
    int m0, n0;
    if (!find_tile(vid, m0, n0)) return;
    set_offsets(m0, n0);
    issue_next(0, std::false_type{});
    issue_next(1, std::false_type{});
    issue_next(2, std::false_type{});
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * NDMA) : "memory");     // K tile 0 has landed
    load_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, F0);
    Acc accA, accB;
    EpiCtx cc{m0 + wm * 128, n0 + wn * 64}, pc{0, 0};
    // look one output tile ahead (the stream enters it four iterations before this tile's K loop ends)
    auto look_ahead = [&]() __attribute__((always_inline)) {
        unsigned nv = vid + gridDim.x;
        have_next = find_tile(nv, nm0, nn0);
        if (have_next) vid = nv;
    };
    auto advance = [&]() __attribute__((always_inline)) {        // the tile just looked ahead to becomes the current one
        pc = cc;
        cc = EpiCtx{nm0 + wm * 128, nn0 + wn * 64};
    };
    look_ahead();
    tile_body(std::false_type{}, accA, accB, pc);
    for (;;) {
        if (!have_next) {
            tail_epilogue(accA, cc);
            break;
        }
        advance();
        look_ahead();
        tile_body(std::true_type{}, accB, accA, pc);
        if (!have_next) {
            tail_epilogue(accB, cc);
            break;
        }
        advance();
        look_ahead();
        tile_body(std::true_type{}, accA, accB, pc);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the stream's last (unread) requests land before the block exits
}

template <int ACT, bool HAS_RES, bool O_PL>
static int launch_pp(const uint16_t* Whi, const float* bias, const float* residual, int64_t ldr, float* Cout, int64_t ldc,
                     int64_t lda, int M, int N, int K, float scale, int scale_cols, const uint16_t* a_hi, uint16_t* o_hi,
                     hipStream_t stream) {
    static DeviceOnce configured;
    auto kern = gemm16_pp_kernel<ACT, HAS_RES, O_PL>;
    if (configured.pending()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, pp::LDS);
        if (e != hipSuccess) return fail(RNAMSM_ERR_HIP, "gemm16_pp: hipFuncSetAttribute: %s", hipGetErrorString(e));
        configured.mark();
    }
    const int nb = N / pp::BN;
    const int group = tuning().gemm_group > 0 ? tuning().gemm_group : (nb > 8 ? (int)xcd_group_for_persistent((M + pp::BM - 1) / pp::BM, 8) : 1);
    const unsigned total = xcd_panel_grid_grouped((M + pp::BM - 1) / pp::BM, nb, (unsigned)group);
    const unsigned pb = tuning().gemm16_persist > 0 ? (unsigned)tuning().gemm16_persist : 256u;
    const unsigned grid = pb < total ? pb : total;
    KernelTimer timer(TC_GEMM, 2.0 * M * N * K,
                      2.0 * ((double)M * K + (double)N * K) + (O_PL ? 2.0 : 4.0) * (double)M * N + (HAS_RES ? 4.0 * (double)M * N : 0.0),
                      stream, PEAK_F16_MFMA_TFLOPS, 1.0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(pp::THREADS), pp::LDS, stream, a_hi, lda, Whi, bias, residual, ldr, Cout, ldc, M, N, K,
                       scale, scale_cols, o_hi, group, total);
    RNAMSM_CHECK_LAUNCH("gemm16_pp");
    return RNAMSM_OK;
}

// Eligibility: plain bf16, plane input, N % 128 == 0, K a multiple of 192 with at least 12 K tiles (the drain schedule is
// 10 iterations long), enough rows to fill the chip, a bias (every Linear of the model has one).
bool gemm16_pp_eligible(int64_t M, int N, int K, const float* bias) {
    const int nk = K / pp::BK;          // the stream's tile switch sits at iteration nk - 4: iteration 8 (nk = 12) or a steady one = 2 mod 3
    return tuning().gemm16_pp != 0 && bias && M >= 2048 && N % pp::BN == 0 && K % (3 * pp::BK) == 0 && (nk == pp::NSPECIAL || nk >= 18);
}

int gemm16_pp(const uint16_t* a_hi, int64_t lda, const uint16_t* Whi, const float* bias, const float* residual, int64_t ldr,
              float* Cout, int64_t ldc, int M, int N, int K, int act, float scale, int scale_cols, uint16_t* o_hi,
              hipStream_t stream) {
    if (o_hi)
        return act == RNAMSM_ACT_GELU_ERF
                   ? launch_pp<RNAMSM_ACT_GELU_ERF, false, true>(Whi, bias, nullptr, 0, nullptr, ldc, lda, M, N, K, scale, scale_cols, a_hi, o_hi, stream)
                   : launch_pp<RNAMSM_ACT_NONE, false, true>(Whi, bias, nullptr, 0, nullptr, ldc, lda, M, N, K, scale, scale_cols, a_hi, o_hi, stream);
    return residual ? launch_pp<RNAMSM_ACT_NONE, true, false>(Whi, bias, residual, ldr, Cout, ldc, lda, M, N, K, scale, scale_cols, a_hi, nullptr, stream)
                    : launch_pp<RNAMSM_ACT_NONE, false, false>(Whi, bias, nullptr, 0, Cout, ldc, lda, M, N, K, scale, scale_cols, a_hi, nullptr, stream);
}

}  // namespace rnamsm
