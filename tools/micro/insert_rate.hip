// Microbenchmark: cost of one sorted top-20 insert per wave, (a) packed f64 min/max, (b) f32 med3 + index cndmask.
// hipcc --offload-arch=gfx950 -O3 -o insert_rate.bin insert_rate.hip && ./insert_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int K = 20;
__global__ void k_f64(const double* in, double* out, int iters) {
    double v[K];
    for (int i = 0; i < K; ++i) v[i] = -1e300;
    double c = in[threadIdx.x];
    for (int it = 0; it < iters; ++it) {
        c = c * 1.0000001 + 1e-9;
#pragma unroll
        for (int i = K - 1; i >= 1; --i) v[i] = fmax(fmin(c, v[i - 1]), v[i]);
        v[0] = fmax(c, v[0]);
    }
    double s = 0; for (int i = 0; i < K; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_f32(const float* in, float* out, int iters) {
    float v[K]; int id[K];
    for (int i = 0; i < K; ++i) { v[i] = -1e30f; id[i] = 0; }
    float c = in[threadIdx.x]; int ci = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        c = c * 1.0000001f + 1e-9f; ci += 3;
        bool b[K];
#pragma unroll
        for (int i = 0; i < K; ++i) b[i] = c > v[i];
#pragma unroll
        for (int i = K - 1; i >= 1; --i) {
            id[i] = b[i - 1] ? id[i - 1] : (b[i] ? ci : id[i]);
            v[i] = __builtin_amdgcn_fmed3f(c, v[i - 1], v[i]);
        }
        id[0] = b[0] ? ci : id[0];
        v[0] = fmaxf(c, v[0]);
    }
    float s = 0; for (int i = 0; i < K; ++i) s += v[i] + id[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double *din, *dout; float *fin, *fout;
    hipMalloc(&din, 64 * 8); hipMalloc(&dout, 1024 * 64 * 8 * 4); hipMalloc(&fin, 64 * 4); hipMalloc(&fout, 1024 * 64 * 4 * 4);
    hipMemset(din, 0, 64 * 8); hipMemset(fin, 0, 64 * 4);
    const int iters = 20000;
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
        const int blocks = 1024 * waves_per_simd;   // 64-thread blocks: one wave each
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms;
        hipLaunchKernelGGL(k_f64, dim3(blocks), dim3(64), 0, 0, din, dout, 10);
        hipEventRecord(e0); hipLaunchKernelGGL(k_f64, dim3(blocks), dim3(64), 0, 0, din, dout, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("f64 packed insert, %d wave(s)/SIMD: %.1f ns per insert per wave (%.0f cycles at 2.4 GHz)\n", waves_per_simd, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
        hipLaunchKernelGGL(k_f32, dim3(blocks), dim3(64), 0, 0, fin, fout, 10);
        hipEventRecord(e0); hipLaunchKernelGGL(k_f32, dim3(blocks), dim3(64), 0, 0, fin, fout, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("f32 med3 + index insert, %d wave(s)/SIMD: %.1f ns per insert per wave (%.0f cycles)\n", waves_per_simd, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
    }
    return 0;
}
