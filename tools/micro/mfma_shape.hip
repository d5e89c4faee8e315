// Micro-benchmark: does the fp32 MFMA shape change the clock the chip holds?  Same FLOP per cycle on paper
// (32x32x2: 4096 FLOP / 64 cyc, 16x16x4: 2048 FLOP / 32 cyc); operands are random and change every iteration so the
// data path toggles like a real kernel's.  hipcc --offload-arch=gfx950 -O3 -o mfma_shape.bin mfma_shape.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline float rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return (float)(int)(s >> 8) * (1.f / 8388608.f) - 1.f; }

__global__ __launch_bounds__(256) void k32(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x16){0};
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = rnd(s); b[i] = rnd(s); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(r + i) & 7], b[(r * 3 + i) & 7], acc[i], 0, 0, 0);
        a[it & 7] = a[(it + 3) & 7] * 0.999f + 0.001f * b[it & 7];   // keep the operands moving
    }
    float t = 0;
    for (int i = 0; i < 4; ++i) t += acc[i][0] + acc[i][9];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0};
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = rnd(s); b[i] = rnd(s); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(r + i) & 7], b[(r * 3 + i) & 7], acc[i], 0, 0, 0);
        a[it & 7] = a[(it + 3) & 7] * 0.999f + 0.001f * b[it & 7];
    }
    float t = 0;
    for (int i = 0; i < 16; ++i) t += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}
template <typename K>
void run(const char* name, K kern, double flop_per_iter_per_wave, int waves_per_simd, float* out) {
    const int iters = 40000, grid = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 2000);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = (double)grid * 4 * iters * flop_per_iter_per_wave / (ms * 1e-3) / 1e12;
    printf("%s, %d wave(s)/SIMD: %.2f ms, %.1f TFLOP/s (%.1f %% of 157.3)\n", name, waves_per_simd, ms, tf, tf / 157.3 * 100);
}
int main() {
    float* out; if (hipMalloc(&out, 256 * 4 * 256 * sizeof(float)) != hipSuccess) return 1;
    for (int rep = 0; rep < 2; ++rep)
        for (int w = 1; w <= 2; ++w) {
            run("v_mfma_f32_32x32x2_f32 (32 per iteration)", k32, 32 * 4096.0, w, out);
            run("v_mfma_f32_16x16x4_f32 (64 per iteration)", k16, 64 * 2048.0, w, out);
        }
    return 0;
}
