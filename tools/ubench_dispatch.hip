// Diagnostic: where does the dispatcher put the workgroups of a k_wgrad-shaped launch
// (N blocks x 256 threads, L bytes of LDS each)?  Prints how many CUs got 0 / 1 / 2 / ... blocks.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <map>
__global__ __launch_bounds__(256) void k(unsigned* out, int spin) {
    extern __shared__ float sm[];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { sm[threadIdx.x] += 1.f; }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 352;
    const int lds = argc > 2 ? atoi(argv[2]) : 70 * 1024;
    unsigned* out; hipMalloc(&out, N * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    k<<<N, 256, lds>>>(out, 1000);   // 10 us
    hipDeviceSynchronize();
    std::vector<unsigned> h(2 * N);
    hipMemcpy(h.data(), out, N * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> by_cu;
    for (int i = 0; i < N; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        by_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu].push_back(i);
    }
    std::map<int, int> hist;
    for (auto& kv : by_cu) hist[(int)kv.second.size()]++;
    printf("N=%d lds=%d: distinct CUs used %zu;", N, lds, by_cu.size());
    for (auto& kv : hist) printf("  %d CUs x %d blocks", kv.second, kv.first);
    printf("\n first CUs:");
    int c = 0;
    for (auto& kv : by_cu) { if (c++ >= 6) break; printf(" [%05x:", kv.first); for (int b : kv.second) printf(" %d", b); printf("]"); }
    printf("\n");
    return 0;
}
