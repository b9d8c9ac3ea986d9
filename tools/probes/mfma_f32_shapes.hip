// Bare fp32 MFMA loops on random operands, every CU busy: does the chip hold a different clock (hence a different
// FLOP/s at equal cycles per FLOP) on v_mfma_f32_16x16x4_f32 than on v_mfma_f32_32x32x2_f32?  (MI355X_MICROARCH.md, DVFS
// give-back item 7 reports 1.12-1.15x for the bf16 16x16x32 shape over 32x32x16.)  Same accumulator footprint per wave
// (64 registers), operands in registers, one wave per SIMD and two.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_f32_shapes.bin mfma_f32_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void loop(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(t * 8 + i) & 0xFFFFF]; b[i] = in[(t * 8 + i + 4096) & 0xFFFFF]; }
    if (SHAPE == 32) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[(k + i) & 7], acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
        out[t] = s;
    } else {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(k + i) & 7], b[(k + 2 * i) & 7], acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
        out[t] = s;
    }
}

int main() {
    const int N = 1 << 20;
    std::vector<float> h(N);
    unsigned x = 12345;
    for (int i = 0; i < N; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((x >> 8) & 0xFFFF) / 32768.f - 1.f; }
    float *in, *out;
    hipMalloc(&in, N * 4); hipMalloc(&out, 4 << 20);
    hipMemcpy(in, h.data(), N * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
        const int blocks = 256 * waves_per_simd, iters = 20000;
        for (int rep = 0; rep < 3; ++rep)
            for (int shape : {32, 16}) {
                // per iteration and wave: 32 MFMAs of 32x32x2 (4096 flop each) or 64 of 16x16x4 (2048 flop each): equal flops
                hipEventRecord(e0);
                if (shape == 32) hipLaunchKernelGGL(loop<32>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
                else hipLaunchKernelGGL(loop<16>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double flop = (double)blocks * 4 * iters * 32 * 4096.0;
                printf("waves/SIMD %d  shape %2d: %.2f ms  %.1f TFLOP/s\n", waves_per_simd, shape, ms, flop / ms / 1e9);
            }
    }
    return 0;
}
