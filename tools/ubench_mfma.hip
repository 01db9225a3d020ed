// Microbenchmark (diagnostic): effective clock and cost of the MFMA / LDS-read step used by the chain kernels.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define AS3 __attribute__((address_space(3)))
__global__ __launch_bounds__(256) void k_mfma(long long* out, float* sink, int mode, int iters) {
    extern __shared__ float smem_g[];
    AS3 float* sm = (AS3 float*)smem_g;
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * 132; i += 256) sm[i] = 0.001f * i;
    __syncthreads();
    f32x4 a0 = {0,0,0,0}, a1 = {0,0,0,0};
    f32x4 b = {1.f, 2.f, 3.f, 4.f};
    const long long t0 = wall_clock64();
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        f32x4 a;
        if (mode == 0) a = f32x4{1.f, 1.f, 1.f, 1.f};                                     // registers only
        else a = *(const AS3 f32x4*)(sm + (lane & 15) * 132 + 16 * (it & 7) + 4 * (lane >> 4));   // + LDS fragment read
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], a1, 0, 0, 0);
        }
    }
    const long long c1 = clock64();
    const long long t1 = wall_clock64();
    if (a0.x + a1.y == 12345.f) sink[threadIdx.x] = a0.x;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = c1 - c0; }
}
int main() {
    const int grid = 256;
    long long* out; float* sink;
    hipMalloc(&out, grid * 2 * sizeof(long long)); hipMalloc(&sink, 4096);
    std::vector<long long> h(2 * grid);
    for (int mode = 0; mode < 2; ++mode)
    for (int iters : {8, 64, 1024}) {
        for (int rep = 0; rep < 3; ++rep) k_mfma<<<grid, 256, 16 * 132 * 4>>>(out, sink, mode, iters);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), out, 2 * grid * sizeof(long long), hipMemcpyDeviceToHost);
        std::vector<double> us, cyc;
        for (int i = 0; i < grid; ++i) { us.push_back(h[2*i] / 100.0); cyc.push_back((double)h[2*i+1]); }
        std::sort(us.begin(), us.end()); std::sort(cyc.begin(), cyc.end());
        const int nm = iters * 8;
        printf("mode=%d iters=%4d (%5d MFMA/wave): median %.2f us = %.1f ns/MFMA ; clock64 delta %.0f -> %.1f per MFMA\n", mode, iters, nm,
               us[grid/2], us[grid/2] * 1000.0 / nm, cyc[grid/2], cyc[grid/2] / nm);
    }
    return 0;
}
