// Micro-benchmark: what does the bf16 matrix pipe SUSTAIN under the board's power cap?  Pure register loops of
// v_mfma_f32_32x32x16_bf16 and v_mfma_f32_16x16x32_bf16 (four independent accumulators per wave, operands with random
// mantissas so that the datapath toggles), 1 or 2 waves per SIMD, ~2 s per run; reports TFLOP/s, the fraction of the
// 2 516.6 TFLOP/s dense peak, and the shader clock the run held (s_memtime ticks per wall second).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_bf16_power.bin tools/micro/mfma_bf16_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ unsigned long long g_clk[2];

__device__ inline bf16x8 operand(unsigned seed) {   // eight bf16 values in [1, 2) with random mantissas, random signs
    u32x4 v;
    for (int i = 0; i < 4; ++i) {
        seed = seed * 1664525u + 1013904223u;
        const unsigned lo = 0x3f80u | ((seed >> 9) & 0x7fu) | ((seed >> 3) & 0x8000u);
        seed = seed * 1664525u + 1013904223u;
        const unsigned hi = 0x3f80u | ((seed >> 9) & 0x7fu) | ((seed >> 3) & 0x8000u);
        v[i] = lo | (hi << 16);
    }
    return __builtin_bit_cast(bf16x8, v);
}

template <int SHAPE>   // 0: 32x32x16, 1: 16x16x32
__global__ __launch_bounds__(256) void k(float* out, int iters, int seed0) {
    const bf16x8 a0 = operand(seed0 + threadIdx.x * 7 + blockIdx.x), a1 = operand(seed0 * 3 + threadIdx.x * 11 + blockIdx.x);
    const bf16x8 b0 = operand(seed0 * 5 + threadIdx.x * 13), b1 = operand(seed0 * 9 + threadIdx.x * 17);
    unsigned long long t0 = 0;
    if (threadIdx.x == 0 && blockIdx.x == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    float s = 0.f;
    if (SHAPE == 0) {
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c3, 0, 0, 0);
            }
        }
        s = c0[0] + c1[5] + c2[9] + c3[15];
    } else {
        f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, c3, 0, 0, 0);
            }
        }
        s = c0[0] + c1[1] + c2[2] + c3[3];
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned long long t1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        g_clk[0] = t1 - t0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE>
void run(int waves_per_simd, float* out, double seconds) {
    const int grid = 256 * waves_per_simd;
    const double flop_per_mfma = SHAPE == 0 ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 20000;
    float ms = 0.f;
    for (int pass = 0; pass < 2; ++pass) {   // pass 0 calibrates the iteration count for ~`seconds`
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<SHAPE>, dim3(grid), dim3(256), 0, 0, out, iters, 12345 + pass);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (pass == 0) iters = (int)(iters * seconds * 1e3 / ms);
    }
    unsigned long long clk[2];
    hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk));
    const double mfma = (double)grid * 4 * (double)iters * 16;
    const double tf = mfma * flop_per_mfma / (ms * 1e-3) / 1e12;
    printf("%-28s waves/SIMD %d: %7.1f ms  %7.1f TFLOP/s = %.3f of 2516.6 dense peak, shader clock %.3f GHz (first wave)\n",
           SHAPE == 0 ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_16x16x32_bf16", waves_per_simd, ms, tf, tf / 2516.6,
           (double)clk[0] / (ms * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
    float* out;
    hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
    for (int w = 1; w <= 2; ++w) { run<0>(w, out, seconds); run<1>(w, out, seconds); }
    return 0;
}
