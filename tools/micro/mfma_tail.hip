// Micro-test: can a 16-row tail tile computed with v_mfma_f32_16x16x4_f32 reproduce, BIT FOR BIT, the rows a
// v_mfma_f32_32x32x2_f32 tile computes?  The chain kernel feeds the 32x32x2 instruction fragments in which MFMA step j of
// k-block kb multiplies k = 8 kb + j (half h = 0) and k = 8 kb + 4 + j (h = 1); the 16x16x4 instruction takes four k per
// step.  If both accumulate their k's as one sequential fma chain in operand order, the 16x16x4 steps must be fed
// (k0, k4, k1, k5) and (k2, k6, k3, k7) per k-block to match.  This program runs both on random data (K = 128, as L3 of the
// chain) and counts mismatching bits; it also tries the plain order (k0..k3), (k4..k7) to show the test has power.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_tail.bin mfma_tail.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 128;

// A (32 x K) and B (K x 32) row-major in global memory; out32 (32 x 32) by 32x32x2, out16a / out16b (16 x 32) by 16x16x4
__global__ __launch_bounds__(64) void tail_kernel(const float* A, const float* B, float* out32, float* out16a, float* out16b) {
    const int lane = threadIdx.x, col = lane & 31, h = lane >> 5;
    f32x16 acc = {0};
    for (int kb = 0; kb < K / 8; ++kb)
        for (int j = 0; j < 4; ++j) {
            const int k = 8 * kb + 4 * h + j;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[col * K + k], B[k * 32 + col], acc, 0, 0, 0);   // A row = lane & 31
        }
    for (int i = 0; i < 16; ++i) out32[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + col] = acc[i];
    // 16x16x4: lane (r16 = lane & 15, kq = lane >> 4): A[r16][k(kq)], B[k(kq)][c16]; C[4 (lane >> 4) + i][lane & 15]
    const int r16 = lane & 15, kq = lane >> 4;
    for (int variant = 0; variant < 2; ++variant)
        for (int half = 0; half < 2; ++half) {
            f32x4 c = {0};
            for (int kb = 0; kb < K / 8; ++kb)
                for (int step = 0; step < 2; ++step) {
                    // variant 0: steps (k0,k4,k1,k5), (k2,k6,k3,k7); variant 1: (k0..k3), (k4..k7)
                    const int k = variant == 0 ? 8 * kb + 4 * (kq & 1) + (kq >> 1) + 2 * step : 8 * kb + 4 * step + kq;
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r16 * K + k], B[k * 32 + 16 * half + r16], c, 0, 0, 0);
                }
            float* o = variant == 0 ? out16a : out16b;
            for (int i = 0; i < 4; ++i) o[(4 * kq + i) * 32 + 16 * half + r16] = c[i];
        }
}

int main() {
    std::vector<float> A(32 * K), B(K * 32);
    float *dA, *dB, *d32, *d16a, *d16b;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&d32, 32 * 32 * 4); hipMalloc(&d16a, 16 * 32 * 4); hipMalloc(&d16b, 16 * 32 * 4);
    long bad_a = 0, bad_b = 0, total = 0;
    srand(1);
    for (int trial = 0; trial < 200; ++trial) {
        for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
        for (auto& v : B) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * (trial % 3 == 0 ? 100.f : 1.f);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(tail_kernel, dim3(1), dim3(64), 0, 0, dA, dB, d32, d16a, d16b);
        float o32[32 * 32], oa[16 * 32], ob[16 * 32];
        hipMemcpy(o32, d32, sizeof(o32), hipMemcpyDeviceToHost); hipMemcpy(oa, d16a, sizeof(oa), hipMemcpyDeviceToHost); hipMemcpy(ob, d16b, sizeof(ob), hipMemcpyDeviceToHost);
        for (int i = 0; i < 16 * 32; ++i) {
            bad_a += memcmp(&o32[i], &oa[i], 4) != 0;
            bad_b += memcmp(&o32[i], &ob[i], 4) != 0;
            ++total;
        }
    }
    printf("16x16x4 vs 32x32x2 over %ld outputs: interleaved k order (k0,k4,k1,k5 | k2,k6,k3,k7): %ld differ; plain k order: %ld differ\n", total, bad_a, bad_b);
    return 0;
}
