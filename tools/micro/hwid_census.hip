// Where does the dispatcher put the workgroups that are resident together when a grid starts?  1024 workgroups of 256 threads
// with 52 KB of LDS (3 per CU, like pn_chain_kernel); each records HW_REG_HW_ID / XCC_ID of its wave 0 and then waits until
// every workgroup that can be resident has checked in.   hipcc --offload-arch=gfx950 -O3 -o hwid_census.bin hwid_census.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned* counter, int expect) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const unsigned order = atomicAdd(counter, 1u);
        out[blockIdx.x * 3] = hw; out[blockIdx.x * 3 + 1] = xcc; out[blockIdx.x * 3 + 2] = order;
        lds[0] = (float)hw;
        for (int i = 0; i < 20000 && atomicAdd(counter, 0u) < (unsigned)expect; ++i) __builtin_amdgcn_s_sleep(32);  // bounded wait
    }
    __syncthreads();
}
int main() {
    const int grid = 1024, expect = 768;
    unsigned *out, *cnt;
    hipMalloc(&out, grid * 3 * sizeof(unsigned)); hipMalloc(&cnt, 4); hipMemset(cnt, 0, 4);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 52 * 1024, 0, out, cnt, expect);
    std::vector<unsigned> h(grid * 3);
    hipMemcpy(h.data(), out, grid * 3 * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> per_cu;
    for (int b = 0; b < grid; ++b) {
        const unsigned hw = h[b * 3], xcc = h[b * 3 + 1] & 0xf, order = h[b * 3 + 2];
        const unsigned wave = hw & 0xf, simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        if (b < 32 || (b % 97) == 0) printf("block %4d arrival %4u: xcc %u se %u sh %u cu %2u simd %u wave_slot %u\n", b, order, xcc, se, sh, cu, simd, wave);
        if (order < (unsigned)expect) per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b * 16 + wave);
    }
    int shown = 0;
    for (auto& kv : per_cu) {
        if (shown++ >= 12) break;
        printf("CU %05x:", kv.first);
        for (int v : kv.second) printf(" block %d (slot %d)", v / 16, v % 16);
        printf("\n");
    }
    printf("distinct CUs among the first %d arrivals: %zu\n", expect, per_cu.size());
    return 0;
}
