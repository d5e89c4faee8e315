// Micro-benchmark: issue cost (shader cycles per wave instruction, one wave per SIMD) of the VALU operations the kNN
// selection is built from: v_max_f64 / v_min_f64 (the packed sorted insert of iq_topk.h), v_med3_f32, v_cmp + v_cndmask.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate.bin valu_rate.hip && ./valu_rate.bin
#include <hip/hip_runtime.h>

#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, unsigned long long* cyc, int iters, double seed) {
    double v[20];
    float f[20];
    int ix[20];
    for (int i = 0; i < 20; ++i) { v[i] = seed * (20 - i) + threadIdx.x; f[i] = (float)v[i]; ix[i] = i; }
    double c = seed * 7.5;
    float cf = (float)c;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // the fp64 sorted insert: 2 ops per slot
#pragma unroll
            for (int i = 19; i >= 1; --i) v[i] = fmax(fmin(c, v[i - 1]), v[i]);
            v[0] = fmax(c, v[0]);
            c += 0.37;
        } else if (MODE == 1) {   // fp32 values by med3, indices by compare + 2 selects
            bool t[21];
#pragma unroll
            for (int i = 0; i < 20; ++i) t[i] = f[i] < cf;
            t[20] = true;
#pragma unroll
            for (int i = 19; i >= 1; --i) {
                ix[i] = t[i - 1] ? ix[i - 1] : (t[i] ? it : ix[i]);
                f[i] = __builtin_amdgcn_fmed3f(cf, f[i - 1], f[i]);
            }
            ix[0] = t[0] ? it : ix[0];
            f[0] = fmaxf(cf, f[0]);
            cf += 0.37f;
        } else {                  // fp32 values only
#pragma unroll
            for (int i = 19; i >= 1; --i) f[i] = __builtin_amdgcn_fmed3f(cf, f[i - 1], f[i]);
            f[0] = fmaxf(cf, f[0]);
            cf += 0.37f;
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    double s = 0;
    for (int i = 0; i < 20; ++i) s += v[i] + f[i] + ix[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    double* out;
    unsigned long long* cyc;
    hipMalloc(&out, 2048 * 64 * 8);
    hipMalloc(&cyc, 2048 * 8);
    const int iters = 20000;
    const char* names[3] = {"fp64 insert (20 x v_min_f64 + v_max_f64)", "fp32 med3 + cmp + 2 cndmask per slot", "fp32 med3 only"};
    for (int waves = 1; waves <= 2; ++waves)
        for (int m = 0; m < 3; ++m) {
            const int grid = 1024 * waves;   // `waves` waves per SIMD
            for (int rep = 0; rep < 2; ++rep) {
                if (m == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, out, cyc, iters, 1.25);
                if (m == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, out, cyc, iters, 1.25);
                if (m == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(64), 0, 0, out, cyc, iters, 1.25);
            }
            hipDeviceSynchronize();
            unsigned long long h[8];
            hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            printf("%-44s %d wave(s)/SIMD: %.1f cycles per insert of 20 slots (per wave)\n", names[m], waves, (double)h[0] / iters);
        }
    return 0;
}
