// Microbenchmark: how fast can ONE workgroup (4 waves) pull a KB-sized weight block that every
// workgroup of the grid also reads, as wave-contiguous 1 KB float4 loads?  (diagnostic only)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define AS1 __attribute__((address_space(1)))

// rot (round 4): every workgroup walks the SAME shared block, but piece p is read as p ^ ((blockIdx.x >> 3) & rot): the
// workgroups that share an XCD (round-robin placement: blocks b, b + 8, ...) ask for DIFFERENT kilobytes at any one time
// instead of all for the same one (same L2 channel).
template <int NL>   // loads in flight per lane
__global__ __launch_bounds__(256) void k_fetch(const float* buf, int kb_per_wave, int distinct, long long* out, float* sink, int rot = 0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rx = (blockIdx.x >> 3) & rot;
    const float* base = buf + (size_t)(distinct ? blockIdx.x : 0) * kb_per_wave * 4 * 256 + (size_t)wave * kb_per_wave * 256 + lane * 4;
    __syncthreads();
    const long long t0 = wall_clock64();
    f32x4 acc = {0, 0, 0, 0};
    for (int k = 0; k < kb_per_wave; k += NL) {
        f32x4 v[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) v[j] = *(const AS1 f32x4*)(base + (size_t)((k + j) ^ rx) * 256);
#pragma unroll
        for (int j = 0; j < NL; ++j) acc += v[j];
    }
    if (acc.x == 12345.f) sink[threadIdx.x] = acc.y + acc.z + acc.w;
    __syncthreads();
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
__global__ void k_touch(float* buf, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = 1.0f;
}
int main() {
    const int grid = 256;
    float *buf, *sink; long long* out;
    const size_t nfl = (size_t)grid * 96 * 256 * 4;     // enough for distinct mode at 96 KB/wave... 
    hipMalloc(&buf, nfl * sizeof(float)); hipMalloc(&sink, 4096); hipMalloc(&out, grid * sizeof(long long));
    std::vector<long long> h(grid);
    for (int distinct = 0; distinct < 2; ++distinct)
    for (int fresh = 0; fresh < 2; ++fresh)
    for (int kbw : {24, 96}) {
        for (int rep = 0; rep < 3; ++rep) {
            if (fresh) k_touch<<<1024, 256>>>(buf, nfl);
            k_fetch<24><<<grid, 256>>>(buf, kbw, distinct, out, sink);
        }
        hipDeviceSynchronize();
        hipMemcpy(h.data(), out, grid * sizeof(long long), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double us_med = h[grid / 2] / 100.0, us_max = h[grid - 1] / 100.0;
        printf("distinct=%d fresh=%d KB/WG=%3d : median %.2f us (%.1f GB/s per CU), max %.2f us\n", distinct, fresh, kbw * 4,
               us_med, kbw * 4 * 1024 / us_med / 1e3, us_max);
    }
    // shared block, rotated piece order per workgroup (rot = 7: eight orders, 31: thirty-two) against the same order
    for (int rot : {0, 7, 31})
    for (int kbw : {24, 96}) {
        for (int rep = 0; rep < 3; ++rep) k_fetch<8><<<grid, 256>>>(buf, kbw, 0, out, sink, rot);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), out, grid * sizeof(long long), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double us_med = h[grid / 2] / 100.0, us_max = h[grid - 1] / 100.0;
        printf("shared, 8 in flight, rot=%2d KB/WG=%3d : median %.2f us (%.1f GB/s per CU), max %.2f us\n", rot, kbw * 4,
               us_med, kbw * 4 * 1024 / us_med / 1e3, us_max);
    }
    return 0;
}
