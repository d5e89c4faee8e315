// Micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate vs number of independent accumulators
// and waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate tools/mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x16){0};
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][7];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu, float* out) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, 100, 1.f, 1.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)grid * 4 * iters * 16;
    const double tf = mfma * 4096 / (ms * 1e-3) / 1e12;
    // cycles per MFMA per SIMD assuming 2.4 GHz
    printf("NACC=%d waves/SIMD=%d: %.2f ms, %.1f TFLOP/s (%.1f%% of 157.3)\n", NACC, blocks_per_cu, ms, tf, tf / 157.3 * 100);
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int w = 1; w <= 3; ++w) { run<1>(w, out); run<2>(w, out); run<4>(w, out); }
    return 0;
}
