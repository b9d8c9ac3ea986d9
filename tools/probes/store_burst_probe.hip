// How fast can ONE CU drain a 128 KB output tile (16 x global_store_dwordx4 per lane, 8 waves), and how does that change
// with the number of CUs storing at the same time?  The 16-bit GEMM epilogue (256x256 bf16 tile per CU) measures 10.7 k
// cycles; 256 CUs x 128 KB in 10.7 k cycles is ~6 TB/s, the chip's plain-store rate -- so is the epilogue bound per CU
// (store issue) or by the whole chip storing at once?  One 512-thread block per CU (100 KB of LDS keeps a second one
// out); blocks with blockIdx >= active leave at once.  Prints cycles (s_memtime) from the first store to vmcnt(0).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/store_burst_probe.hip -o /tmp/store_burst_probe && /tmp/store_burst_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(512) void burst(f32x4* out, int active, int stride_blocks, int tiles, long long* cycles) {
    extern __shared__ char lds[];
    if ((int)blockIdx.x % stride_blocks != 0 || (int)blockIdx.x / stride_blocks >= active) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x4 v = {(float)threadIdx.x, 1.f, 2.f, 3.f};
    long long worst = 0;
    for (int t = 0; t < tiles; ++t) {
        f32x4* base = out + ((size_t)(t * gridDim.x + blockIdx.x) * 8192);       // 8192 x 16 B = 128 KB per tile
        __syncthreads();
        const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            f32x4* p = base + (i * 8 + wave) * 64 + lane;
            if (NT) __builtin_nontemporal_store(v, p);
            else *p = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const long long t1 = __builtin_amdgcn_s_memtime();
        worst = t1 - t0 > worst ? t1 - t0 : worst;
        v[1] += 1.f;
    }
    if (threadIdx.x == 0) cycles[blockIdx.x] = worst;
}

int main() {
    const int grid = 256, tiles = 8;
    f32x4* out;
    long long* cyc;
    hipMalloc(&out, (size_t)grid * tiles * 8192 * 16);
    hipMalloc(&cyc, grid * sizeof(long long));
    hipFuncSetAttribute(reinterpret_cast<const void*>(burst<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(burst<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    std::vector<long long> h(grid);
    for (int nt = 0; nt < 2; ++nt)
        for (int stride : {1, 8})                     // stride 8: the active blocks sit on ONE XCD (blocks are dealt round-robin over 8)
            for (int active : {256, 128, 64, 32, 16, 8, 1}) {
                if (active * stride > grid) continue;
                for (int rep = 0; rep < 3; ++rep) {
                    hipMemset(cyc, 0, grid * sizeof(long long));
                    if (nt) hipLaunchKernelGGL(burst<true>, dim3(grid), dim3(512), 100 * 1024, 0, out, active, stride, tiles, cyc);
                    else hipLaunchKernelGGL(burst<false>, dim3(grid), dim3(512), 100 * 1024, 0, out, active, stride, tiles, cyc);
                    hipDeviceSynchronize();
                }
                hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
                std::vector<long long> a;
                for (long long c : h) if (c > 0) a.push_back(c);
                std::sort(a.begin(), a.end());
                printf("%s stores, %3d CUs storing (%s): slowest tile of a CU: median %lld cycles, max %lld  (%.1f B/clk/CU)\n",
                       nt ? "nontemporal" : "plain      ", (int)a.size(), stride == 1 ? "all XCDs" : "one XCD ", a[a.size() / 2], a.back(),
                       131072.0 / a[a.size() / 2]);
            }
    return 0;
}
