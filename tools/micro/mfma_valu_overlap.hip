// Micro-benchmark: how VALU work overlaps with a saturating fp32 MFMA stream on gfx950.
//   S  one wave per SIMD: a dependent chain of v_mfma_f32_32x32x2_f32 with N independent VALU instructions (v_fma_f32 on other
//      registers) written between consecutive MFMAs: cycles per MFMA as N grows (64 = the MFMA pipe is full)
//   X  two waves per SIMD: one issues only MFMAs, the other only VALU instructions: cycles per VALU instruction of the second
//      wave and cycles per MFMA of the first (alone: 4 and 64)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap.bin mfma_valu_overlap.hip && ./mfma_valu_overlap.bin
#include <hip/hip_runtime.h>

#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline unsigned long long clk() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

template <int N>
__global__ __launch_bounds__(64) void same_wave(float* out, unsigned long long* cyc, int iters, float seed) {
    f32x16 acc = {0};
    float a = seed + threadIdx.x, b = seed * 0.5f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed * i;
    const unsigned long long t0 = clk();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int n = 0; n < N; ++n) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[n & 7]) : "v"(b));
        }
    }
    const unsigned long long t1 = clk();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// blocks 0..1023: MFMA only; blocks 1024..2047: VALU only (with 2048 blocks of one wave every SIMD holds one of each kind as
// long as the dispatcher places them round-robin; the per-kind medians are reported)
__global__ __launch_bounds__(64) void two_waves(float* out, unsigned long long* cyc, int iters, float seed, int split) {
    f32x16 acc = {0};
    float a = seed + threadIdx.x, b = seed * 0.5f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed * i;
    const bool mfma_wave = (int)blockIdx.x < split;
    const unsigned long long t0 = clk();
    if (mfma_wave) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int n = 0; n < 128; ++n) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[n & 7]) : "v"(b));
        }
    }
    const unsigned long long t1 = clk();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static unsigned long long median(unsigned long long* h, int n) {
    for (int i = 1; i < n; ++i)
        for (int j = i; j > 0 && h[j - 1] > h[j]; --j) { unsigned long long t = h[j]; h[j] = h[j - 1]; h[j - 1] = t; }
    return h[n / 2];
}

int main() {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 2048 * 64 * 4);
    hipMalloc(&cyc, 2048 * 8);
    static unsigned long long h[2048];
    const int iters = 4000;
#define RUN_S(N)                                                                                                         \
    hipLaunchKernelGGL(same_wave<N>, dim3(1024), dim3(64), 0, 0, out, cyc, iters, 1.25f);                                \
    hipLaunchKernelGGL(same_wave<N>, dim3(1024), dim3(64), 0, 0, out, cyc, iters, 1.25f);                                \
    hipDeviceSynchronize();                                                                                              \
    hipMemcpy(h, cyc, 1024 * 8, hipMemcpyDeviceToHost);                                                                  \
    printf("S  one wave/SIMD, %2d VALU between MFMAs: %.1f cycles per MFMA\n", N, (double)median(h, 1024) / (iters * 8.0));
    RUN_S(0) RUN_S(2) RUN_S(4) RUN_S(8) RUN_S(12) RUN_S(14) RUN_S(16) RUN_S(24)
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(two_waves, dim3(2048), dim3(64), 0, 0, out, cyc, iters, 1.25f, 1024);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, 2048 * 8, hipMemcpyDeviceToHost);
    printf("X  MFMA wave next to a VALU wave: %.1f cycles per MFMA; the VALU wave: %.1f cycles per VALU instruction\n",
           (double)median(h, 1024) / (iters * 8.0), (double)median(h + 1024, 1024) / (iters * 128.0));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(two_waves, dim3(1024), dim3(64), 0, 0, out, cyc, iters, 1.25f, 0);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, 1024 * 8, hipMemcpyDeviceToHost);
    printf("X  VALU waves alone (one per SIMD): %.1f cycles per VALU instruction\n", (double)median(h, 1024) / (iters * 128.0));
    return 0;
}
