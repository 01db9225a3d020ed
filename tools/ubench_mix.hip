// Microbenchmark: do a wave's requests for SHARED (L2-resident, every workgroup reads the same) weight fragments wait behind
// another wave's requests for DISTINCT (per-workgroup, written by the previous kernel) rows in the CU's memory path?
// Per wave class: time until the last request has been ISSUED, time until the data is there.  (diagnostic only)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define AS1 __attribute__((address_space(1)))
constexpr int MAXL = 32;

// waves < NSHW: first ND0 distinct then NS shared loads; the others: ND1 distinct loads (all counts compile-time: no branches)
template <int ND, int NS>
__device__ __forceinline__ void leg(const float* db, const float* sb, long long& t0, long long& t1, long long& t2, float* sink) {
    f32x4 v[ND + 1], w[NS + 1];
    t0 = wall_clock64();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < ND; ++j) v[j] = *(const AS1 f32x4*)(db + (size_t)j * 256);
#pragma unroll
    for (int j = 0; j < NS; ++j) w[j] = *(const AS1 f32x4*)(sb + (size_t)j * 256);
    asm volatile("" ::: "memory");
    t1 = wall_clock64();
    f32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < ND; ++j) acc += v[j];
#pragma unroll
    for (int j = 0; j < NS; ++j) acc += w[j];
    if (acc.x == 12345.f) sink[threadIdx.x] = acc.y + acc.z + acc.w;
    asm volatile("" ::: "memory");
    t2 = wall_clock64();
}
template <int NS, int ND0, int ND1, int NSHW>
__global__ __launch_bounds__(256) void k_mix(const float* shared_buf, const float* distinct_buf, long long* out, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* sb = shared_buf + (size_t)wave * MAXL * 256 + lane * 4;
    const float* db = distinct_buf + ((size_t)blockIdx.x * 4 + wave) * MAXL * 256 + lane * 4;
    long long t0 = 0, t1 = 0, t2 = 0;
    for (int pass = 0; pass < 3; ++pass) {                 // the last pass is timed: instruction cache warm, data of its own
        db += (size_t)gridDim.x * 4 * MAXL * 256;
        __syncthreads();
        if (wave < NSHW) leg<ND0, NS>(db, sb, t0, t1, t2, sink);
        else leg<ND1, 0>(db, sb, t0, t1, t2, sink);
    }
    if (lane == 0) { out[((size_t)blockIdx.x * 4 + wave) * 2] = t1 - t0; out[((size_t)blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0; }
}
__global__ void k_touch(float* buf, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = 1.0f;
}
int main() {
    const int grid = 256;
    float *sh, *di, *sink; long long* out;
    const size_t nsh = (size_t)4 * MAXL * 256, ndi = (size_t)grid * 4 * MAXL * 256 * 4;
    (void)hipMalloc(&sh, nsh * 4); (void)hipMalloc(&di, ndi * 4); (void)hipMalloc(&sink, 4096); (void)hipMalloc(&out, grid * 4 * 2 * sizeof(long long));
    k_touch<<<64, 256>>>(sh, nsh);
    std::vector<long long> h(grid * 8);
    auto report = [&](const char* name) {
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), out, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        for (int cls = 0; cls < 2; ++cls) {
            std::vector<double> iss, done;
            for (int b = 0; b < grid; ++b) for (int w = cls * 2; w < cls * 2 + 2; ++w) { iss.push_back(h[(b * 4 + w) * 2] / 100.0); done.push_back(h[(b * 4 + w) * 2 + 1] / 100.0); }
            std::sort(iss.begin(), iss.end()); std::sort(done.begin(), done.end());
            printf("%s waves %d-%d: issued %.2f us (max %.2f), data %.2f us (max %.2f)\n", name, cls * 2, cls * 2 + 1, iss[iss.size() / 2],
                   iss.back(), done[done.size() / 2], done.back());
        }
    };
#define CASE(name, NS, ND0, ND1, NSHW) for (int rep = 0; rep < 3; ++rep) { k_touch<<<1024, 256>>>(di, ndi); \
        k_mix<NS, ND0, ND1, NSHW><<<grid, 256>>>(sh, di, out, sink); } report(name);
    CASE("A  w0-1: 18 shared              | w2-3: idle        ", 18, 0, 0, 2)
    CASE("B  w0-1: 18 shared              | w2-3: 12 distinct ", 18, 0, 12, 2)
    CASE("C  w0-3: 6 distinct + 18 shared |                   ", 18, 6, 0, 4)
    CASE("D  w0-1: 18 shared              | w2-3: 24 distinct ", 18, 0, 24, 2)
    CASE("E  w0-3: 24 shared              |                   ", 24, 0, 0, 4)
    CASE("F  w0-3: 6 distinct             |                   ", 0, 6, 0, 4)
    CASE("G  w0-1: 6 distinct + 18 shared | w2-3: 6 distinct  ", 18, 6, 6, 2)
    CASE("H  w0-3: 12 shared              |                   ", 12, 0, 0, 4)
    CASE("I  w0-3: 12 distinct            |                   ", 0, 12, 0, 4)
    return 0;
}
