// 16-bit operand formats of the matrix-core modes and the shared "64x64 slab through LDS" store.
#pragma once
#include "common.h"

namespace rnamsm {

// 16-bit operand format: FMT 0 = bf16 (8-bit mantissa, fp32 range), FMT 1 = fp16 (11-bit mantissa, |x| < 65504).
// An fp16 hi/lo pair carries ~22 mantissa bits (fp32: 24): "f16x3" is fp32-grade arithmetic at the bf16 MFMA rate for
// operands inside fp16 range -- true for this model's GEMM inputs (LayerNorm outputs, attention contexts, GELU
// activations, 0.04-scale weights); values below 2^-24 * 2^11 of an element's magnitude fall into fp16 subnormals of
// the lo plane, an ABSOLUTE error <= 3e-8 per element.
// A value about to be split into hi = round16(x), lo = round16(x - hi) is made OPAQUE to the optimiser first.  Found in round 4
// (tests/test_gpu_attn16.py, plane outputs): with the default -ffp-contract=fast hipcc contracts the multiplication that
// produced x into the conversions -- the STORED hi comes from v_cvt_pk(fl32(a * b)), the hi inside the lo term from
// v_fma_mixlo_f16(a, b, 0) on the exact product -- and at a near-tie of the 16-bit grid the two round to different neighbours:
// hi + lo is then off by a whole 16-bit ulp (2^-11 relative in fp16; one element in ~30 000, ~3e-6 of relative L2 error, the
// size of the f16x3 mode's whole error budget).  An empty asm with the value as a read-write operand costs nothing and
// pins ONE fp32 value for both conversions.
__device__ __forceinline__ float pinned(float x) {
    asm("" : "+v"(x));
    return x;
}

template <int FMT> struct Half16;
template <> struct Half16<0> {
    typedef __bf16 T;
    typedef __bf16 V8 __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Half16<1> {
    typedef _Float16 T;
    typedef _Float16 V8 __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};


// One wave's 64x64 accumulator slab (2x2 MFMA tiles, the map of mma_core.h) written as whole 256-B row segments:
// the accumulator layout would give 64 four-byte stores per lane (issue-bound), so the slab goes through a
// wave-private [64][68] f32 LDS staging tile and leaves as 16 x 16-B stores per lane (each instruction = 4 rows x 256 B).
// rowoff(row 0..63) -> element offset of that row's 64-column segment in the output, or -1 to skip the row.
// OUT: 0 = fp32 to out_f32; 1 / 2 = bf16 / fp16 hi (+ lo if out_lo) planes (the pre-split A operand of a 16-bit GEMM).
// The caller guarantees every wave is done with the LDS region (barrier) before calling.
template <int OUT, class RowOff>
__device__ __forceinline__ void slab_store_64x64(const f32x16 (&acc)[2][2], float* stage, int li, int lh, int lane,
                                                 RowOff rowoff, float* out_f32, uint16_t* out_hi, uint16_t* out_lo) {
    constexpr int LDE = 64 + 4;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < 16; ++t)
                stage[(mt * 32 + (t & 3) + 8 * (t >> 2) + 4 * lh) * LDE + nt * 32 + li] = acc[mt][nt][t];
    const int er = lane >> 4, ec = (lane & 15) * 4;
    f32x4 ov[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) ov[i] = *reinterpret_cast<const f32x4*>(&stage[(er + 4 * i) * LDE + ec]);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t off = rowoff(er + 4 * i);
        if (off >= 0) {
            if (OUT == 0) {
                *reinterpret_cast<f32x4*>(out_f32 + off + ec) = ov[i];
            } else {
                typedef typename Half16<(OUT > 0 ? OUT - 1 : 0)>::T H;
                typedef H H4 __attribute__((ext_vector_type(4)));
                H4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = pinned(ov[i][e]);
                    hi[e] = (H)x;
                    lo[e] = (H)(x - (float)hi[e]);
                }
                *reinterpret_cast<H4*>(out_hi + off + ec) = hi;
                if (out_lo) *reinterpret_cast<H4*>(out_lo + off + ec) = lo;
            }
        }
    }
}

}  // namespace rnamsm
