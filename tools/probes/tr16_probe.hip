// Probe of ds_read_b64_tr_b16 lane semantics (gfx950): prints, for every lane, which (row, col) elements it received.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(const short* g, short* o) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = g[i];
    __syncthreads();
    const int lane = threadIdx.x;
    const int grp = lane >> 4, pos = lane & 15, q = pos >> 2, p = pos & 3;
    // group grp: 4x16 block at rows 8*(grp>>1).., cols 16*(grp&1)..; lane 4q+p supplies &lds[row q][cols 4p..4p+3]
    const short* addr = &lds[(8 * (grp >> 1) + q) * 64 + 16 * (grp & 1) + 4 * p];
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)addr);
    for (int e = 0; e < 4; ++e) o[lane * 4 + e] = v[e];
}
int main() {
    std::vector<short> h(64 * 64);
    for (int r = 0; r < 64; ++r) for (int c = 0; c < 64; ++c) h[r * 64 + c] = (short)(r * 100 + c);
    short *g, *o; hipMalloc(&g, 64 * 64 * 2); hipMalloc(&o, 64 * 4 * 2);
    hipMemcpy(g, h.data(), 64 * 64 * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o);
    std::vector<short> r(256); hipMemcpy(r.data(), o, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) {
            printf(" (%d,%d)", r[l * 4 + e] / 100, r[l * 4 + e] % 100);
            const int exp_row = 8 * (l >> 5) + e, exp_col = (l & 31);
            if (r[l * 4 + e] != exp_row * 100 + exp_col) ++bad;
        }
        printf("\n");
    }
    printf("expected-model mismatches: %d\n", bad);
    return 0;
}
