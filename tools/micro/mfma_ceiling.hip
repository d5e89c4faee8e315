// Micro-benchmark: what fp32 MFMA rate does THIS chip sustain for >= 1 s on random operands, and at which
// in-kernel clock?  Settles the ceiling the PointNet chain kernel is priced against.
//   A  operands in registers (random, fixed per lane; consecutive MFMAs use different registers, so the operand buses toggle)
//   B  A operand re-read from LDS with ds_read_b128 every K-block (the chain kernel's L3 pattern), B operand in registers
//   Z  the same as A on all-zero operands (the clock the chip holds when nothing toggles)
// each at 1, 2 and 3 waves per SIMD (the chain kernel runs 3 workgroups of 4 waves per CU).  Every configuration runs back to
// back for >= 1.5 s; the rate is taken over the last second with HIP events, the in-kernel clock as
// d(s_memtime) / d(s_memrealtime) x 100 MHz around the loop (MI355X_MICROARCH.md, DVFS give-back item 6), median over workgroups.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_ceiling.bin mfma_ceiling.hip && ./mfma_ceiling.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline float rnd(unsigned& s) {
    s = s * 1664525u + 1013904223u;
    return (float)(int)(s >> 8) * (1.f / 8388608.f) - 1.f;
}

struct Stamp { unsigned long long clk, rt; };

__device__ inline Stamp now() {
    Stamp s;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(s.clk), "=s"(s.rt)::"memory");
    return s;
}

// MODE 0: random register operands, 1: A from LDS, 2: zero operands
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, Stamp* stamps, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 132];
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 17u;
    for (int i = threadIdx.x; i < 64 * 132; i += 256) lds[i] = MODE == 2 ? 0.f : rnd(s);
    f32x4 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { a[i][j] = MODE == 2 ? 0.f : rnd(s); b[i][j] = MODE == 2 ? 0.f : rnd(s); }
    f32x16 acc0 = {0}, acc1 = {0};
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const float* abase = lds + (lane & 31) * 132 + 4 * (lane >> 5);
    const Stamp t0 = now();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {   // one "n-tile" of the chain kernel's L3: 16 K-blocks x (2 m-tiles x 4 MFMAs)
            f32x4 a0 = a[kb & 3], a1 = a[(kb + 1) & 3];
            if (MODE == 1) {
                a0 = *reinterpret_cast<const f32x4*>(abase + 8 * kb);
                a1 = *reinterpret_cast<const f32x4*>(abase + 32 * 132 + 8 * kb);
            }
            const f32x4 bk = b[kb & 3];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], bk[j], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], bk[j], acc1, 0, 0, 0);
            }
        }
        if ((it & 63) == 63) {   // keep the accumulators bounded (random walk): fold them back every 64 n-tiles
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc0[i] *= 0.001f; acc1[i] *= 0.001f; }
        }
    }
    const Stamp t1 = now();
    float t = 0;
    for (int i = 0; i < 16; ++i) t += acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
    if (threadIdx.x == 0) stamps[blockIdx.x] = Stamp{t1.clk - t0.clk, t1.rt - t0.rt};
}

template <typename K>
void run(const char* name, K kern, int wps, float* out, Stamp* stamps) {
    const int grid = 256 * wps;
    const int iters = 12000 / wps;    // ~55 ms per launch at every occupancy
    const double flop = (double)grid * 4 * iters * 128.0 * 4096.0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, stamps, iters);  // ~0.5 s warm-up
    hipEventRecord(e0);
    const int reps = 20;                                                                                    // ~1.1 s measured
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, stamps, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h(grid);
    hipMemcpy(h.data(), stamps, grid * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> ghz, mfma_cyc;
    for (const Stamp& s : h) {
        ghz.push_back((double)s.clk / (double)s.rt * 0.1);
        mfma_cyc.push_back((double)s.clk / ((double)iters * 128.0) / wps);   // shader cycles per MFMA per SIMD
    }
    std::sort(ghz.begin(), ghz.end());
    std::sort(mfma_cyc.begin(), mfma_cyc.end());
    const double tf = flop * reps / (ms * 1e-3) / 1e12;
    printf("%-44s %d wave(s)/SIMD: %7.1f ms/launch  %6.1f TFLOP/s (%5.1f %% of 157.3)  in-kernel clock %.3f GHz  "
           "%.1f cycles per MFMA per SIMD (64 = pipe full)\n",
           name, wps, ms / reps, tf, tf / 157.3 * 100, ghz[ghz.size() / 2], mfma_cyc[mfma_cyc.size() / 2]);
}

int main() {
    float* out;
    Stamp* stamps;
    if (hipMalloc(&out, 256 * 3 * 256 * sizeof(float)) != hipSuccess) return 1;
    if (hipMalloc(&stamps, 256 * 3 * sizeof(Stamp)) != hipSuccess) return 1;
    for (int wps = 1; wps <= 3; ++wps) {
        run("A  random operands in registers", k<0>, wps, out, stamps);
        run("B  A operand from LDS (ds_read_b128), random", k<1>, wps, out, stamps);
        run("Z  all-zero operands", k<2>, wps, out, stamps);
    }
    return 0;
}
