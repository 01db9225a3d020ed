// Microbenchmark (diagnostic): issue rate of INDEPENDENT v_mfma_f32_16x16x4_f32 (16 accumulators, the
// k_wgrad inner step) with 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(512) void k(long long* out, float* sink, int iters) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x;
    const long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = wall_clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i].x;
    if (s == 12345.f) sink[threadIdx.x] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
template <int NACC>
void run(int threads, int grid, long long* out, float* sink) {
    const int iters = 512 / NACC * 4;
    std::vector<long long> h(grid);
    for (int rep = 0; rep < 3; ++rep) k<NACC><<<grid, threads>>>(out, sink, iters);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), out, grid * sizeof(long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double us = h[grid / 2] / 100.0;
    const int n = iters * NACC;
    printf("NACC=%2d threads=%3d grid=%d: %d MFMA/wave in %.2f us = %.1f ns per MFMA per wave; per SIMD %.1f ns\n", NACC, threads, grid, n, us,
           us * 1000 / n, us * 1000 / n / (threads / 256.0));
}
int main() {
    long long* out; float* sink;
    hipMalloc(&out, 4096 * sizeof(long long)); hipMalloc(&sink, 4096);
    for (int grid : {1, 256}) {
        run<16>(256, grid, out, sink);
        run<16>(512, grid, out, sink);
        run<4>(256, grid, out, sink);
        run<2>(256, grid, out, sink);
        run<1>(256, grid, out, sink);
    }
    return 0;
}
