// Diagnostic: does data a kernel wrote stay readable from the SAME XCD's L2 by the next kernel?
// Kernel A: workgroup j writes its own slice.  Kernel B: workgroup j reads the slice of workgroup (j + shift) % n.
// shift 0 = same position in the grid = same XCD (round-robin dispatch, grids that are multiples of 8); shift 1 = the
// neighbouring XCD.  Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_xcd_reuse.hip -o tools/ab/ubench_xcd_reuse
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void k_write(float* buf, int floats_per_wg, float v) {
    f4* p = (f4*)(buf + (size_t)blockIdx.x * floats_per_wg);
    for (int i = threadIdx.x; i < floats_per_wg / 4; i += 256) {
        f4 x = {v, v + 1, v + 2, v + 3};
        if (NT) __builtin_nontemporal_store(x, p + i); else p[i] = x;
    }
}
__global__ __launch_bounds__(256) void k_read(const float* buf, int floats_per_wg, int shift, float* out) {
    const int src = (blockIdx.x + shift) % gridDim.x;
    const f4* p = (const f4*)(buf + (size_t)src * floats_per_wg);
    f4 acc = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < floats_per_wg / 4; i += 256) acc += p[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
}
int main() {
    const int n = 256;
    for (int kb : {32, 64, 108}) {
        const int fl = kb * 256;                       // floats per workgroup
        float *buf, *out;
        CHECK(hipMalloc(&buf, (size_t)n * fl * 4)); CHECK(hipMalloc(&out, n * 4));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int nt = 0; nt < 2; ++nt)
            for (int shift : {0, 8, 1, 4}) {
                float best = 1e9f;
                for (int rep = 0; rep < 20; ++rep) {
                    if (nt) hipLaunchKernelGGL(k_write<true>, dim3(n), dim3(256), 0, 0, buf, fl, (float)rep);
                    else hipLaunchKernelGGL(k_write<false>, dim3(n), dim3(256), 0, 0, buf, fl, (float)rep);
                    hipEventRecord(e0, 0);
                    hipLaunchKernelGGL(k_read, dim3(n), dim3(256), 0, 0, buf, fl, shift, out);
                    hipEventRecord(e1, 0);
                    hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (rep > 3 && ms < best) best = ms;
                }
                printf("%3d KB per workgroup (%5.1f MB), %s stores, reader shift %d: read kernel %.2f us\n", kb, n * fl * 4 / 1e6,
                       nt ? "streaming" : "plain    ", shift, best * 1e3);
            }
        hipFree(buf); hipFree(out);
    }
    return 0;
}
