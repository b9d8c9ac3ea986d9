// Does v_mfma_f32_32x32x2_f32 pay for srcA and srcB sitting in the same VGPR bank (register number mod 4)?
// Two loops of independent MFMAs on 4 accumulator tiles: operands a[s], b[s] (same index of two 4-aligned tuples: same bank)
// against a[s], b[(s+1)&3] / b[(s+2)&3].  Prints cycles per MFMA (s_memtime) for 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_bank_probe.hip -o /tmp/mfma_bank_probe && /tmp/mfma_bank_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ROT>
__global__ __launch_bounds__(256) void probe(const float* in, float* out, long long* cycles, int iters) {
    f32x4 a0 = *reinterpret_cast<const f32x4*>(in + threadIdx.x * 4);
    f32x4 a1 = *reinterpret_cast<const f32x4*>(in + 1024 + threadIdx.x * 4);
    f32x4 b0 = *reinterpret_cast<const f32x4*>(in + 2048 + threadIdx.x * 4);
    f32x4 b1 = *reinterpret_cast<const f32x4*>(in + 3072 + threadIdx.x * 4);
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.f;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[(s + ROT) & 3], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b1[(s + ROT) & 3], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b0[(s + ROT) & 3], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[(s + ROT) & 3], acc[1][1], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int t = 0; t < 16; ++t) r += acc[i][j][t];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int ROT>
static void run(int threads, const float* in, float* out, long long* cyc, const char* tag) {
    const int iters = 20000, blocks = 256;
    hipLaunchKernelGGL(probe<ROT>, dim3(blocks), dim3(threads), 0, 0, in, out, cyc, 100);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<ROT>, dim3(blocks), dim3(threads), 0, 0, in, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double mfmas_per_wave = 16.0 * iters;
    // s_memtime ticks at 100 MHz: use wall time instead; MFMAs per SIMD = waves per SIMD * mfmas_per_wave
    const double waves_per_simd = threads / 64 / 4.0;
    const double tf = 256.0 * 4 * waves_per_simd * mfmas_per_wave * (2.0 * 32 * 32 * 2) / (ms * 1e-3) / 1e12;
    printf("%-28s threads/block %4d (%.0f wave/SIMD): %.3f ms, %.1f TFLOP/s, memtime ticks %lld\n", tag, threads, waves_per_simd, ms, tf, h[0]);
}

int main() {
    float *in, *out;
    long long* cyc;
    hipMalloc(&in, 4096 * 4 * 4);
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 256 * 8);
    float* h = new float[4096 * 4];
    for (int i = 0; i < 4096 * 4; ++i) h[i] = (float)((i * 7 + 3) % 13 - 6) * 0.01f;
    hipMemcpy(in, h, 4096 * 4 * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        for (int threads : {256, 512}) {
            run<0>(threads, in, out, cyc, "a[s], b[s]    (same bank)");
            run<1>(threads, in, out, cyc, "a[s], b[s+1]  (bank + 1)");
            run<2>(threads, in, out, cyc, "a[s], b[s+2]  (bank + 2)");
        }
    }
    return 0;
}
