// Minimal victim for the round-4 finding (profiles/r04_shared_gpu_determinism.txt): do PACKED float32 VALU instructions give
// different bits while another process runs the bf16x3 PointNet chain kernel on the same GPU?
// One wave per workgroup (like the smoothness kernel), registers only - no LDS, no memory inside the loop.  Every lane iterates
// the chaotic map x <- 1 - a x^2 on TWO packed states and, in the same loop, on two scalar twins with the same start values:
//   packed : v_pk_mul_f32 + v_pk_fma_f32          (variant 0)   or  v_pk_mul_f32, v_pk_mul_f32, v_pk_add_f32   (variant 1)
//   scalar : v_mul_f32 + v_fma_f32                              or  v_mul_f32, v_mul_f32, v_add_f32
// Both are IEEE operations: packed == scalar bit for bit, and every launch equals the first.  The map doubles an error per
// step, so ONE wrong ulp anywhere in the loop changes the final value completely.  Variant 2: the scalar loop only (control).
// Prints, per run: launches, launches whose packed result differs from the first launch, launches whose scalar result differs,
// launches where packed != scalar inside one launch, and the number of lanes involved.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/pk_victim.bin tools/micro/pk_victim.hip
// Run  : tools/micro/pk_victim.bin <variant 0|1|2|3> <seconds> [workgroups=64] [iterations=100000]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int VARIANT>
__global__ __launch_bounds__(64) void victim(float* out, int iters) {
    const int gid = blockIdx.x * 64 + threadIdx.x;
    unsigned s = 0x9e3779b9u * (gid + 1);
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f); };   // [0, 1)
    f32x2 x = {rnd() * 1.6f - 0.8f, rnd() * 1.6f - 0.8f};
    float y0 = x[0], y1 = x[1];
    const float a0 = 1.85f + 0.1f * rnd(), a1 = 1.85f + 0.1f * rnd();
    const f32x2 nega = {-a0, -a1}, one = {1.0f, 1.0f};
    const float na0 = -a0, na1 = -a1, o = 1.0f;
    for (int it = 0; it < iters; ++it) {
        if (VARIANT == 0) {
            f32x2 t;
            asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(t) : "v"(x));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(x) : "v"(t), "v"(nega), "v"(one));
            float t0, t1;
            asm volatile("v_mul_f32 %0, %1, %1" : "=v"(t0) : "v"(y0));
            asm volatile("v_mul_f32 %0, %1, %1" : "=v"(t1) : "v"(y1));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(y0) : "v"(t0), "v"(na0), "v"(o));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(y1) : "v"(t1), "v"(na1), "v"(o));
        } else if (VARIANT == 1) {
            f32x2 t, u;
            asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(t) : "v"(x));
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(u) : "v"(t), "v"(nega));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(x) : "v"(u), "v"(one));
            float t0, t1, u0, u1;
            asm volatile("v_mul_f32 %0, %1, %1" : "=v"(t0) : "v"(y0));
            asm volatile("v_mul_f32 %0, %1, %1" : "=v"(t1) : "v"(y1));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u0) : "v"(t0), "v"(na0));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u1) : "v"(t1), "v"(na1));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(y0) : "v"(u0), "v"(o));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(y1) : "v"(u1), "v"(o));
        } else {
            float t0, t1;
            asm volatile("v_mul_f32 %0, %1, %1" : "=v"(t0) : "v"(y0));
            asm volatile("v_mul_f32 %0, %1, %1" : "=v"(t1) : "v"(y1));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(y0) : "v"(t0), "v"(na0), "v"(o));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(y1) : "v"(t1), "v"(na1), "v"(o));
        }
    }
    out[gid * 4 + 0] = x[0];
    out[gid * 4 + 1] = x[1];
    out[gid * 4 + 2] = y0;
    out[gid * 4 + 3] = y1;
}

// Variant 3 (round 5): the OPERAND FORMS of packed float32 - op_sel / op_sel_hi picking the other half of a register pair, as hipcc's
// SLP code does to broadcast one float to both halves (e.g. `v_pk_mul_f32 v[24:25], v[6:7], v[12:13] op_sel:[0,1] op_sel_hi:[1,0]`).
// One modifier at a time, on the multiply (forms 0-3), the add (4-7) or the fma's addend (8-9); the half of the pair that the
// modifier leaves UNREAD holds a chosen bit pattern `garb`; the scalar twin computes the same map.
//   0 mul hi <- src1.lo (op_sel_hi:[1,0])   1 mul lo <- src1.hi (op_sel:[0,1])   2 mul lo <- src0.hi (op_sel:[1,0])   3 mul hi <- src0.lo (op_sel_hi:[0,1])
//   4-7 the same four on v_pk_add_f32       8 fma src2 lo <- hi (op_sel:[0,0,1])  9 fma src2 hi <- lo (op_sel_hi:[1,1,0])
template <int FORM>
__global__ __launch_bounds__(64) void victim_garbage(float* out, int iters, unsigned garb) {
    const int gid = blockIdx.x * 64 + threadIdx.x;
    unsigned s = 0x9e3779b9u * (gid + 1);
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f); };
    f32x2 x = {rnd() * 1.6f - 0.8f, rnd() * 1.6f - 0.8f};
    float y0 = x[0], y1 = x[1];
    const float a = 1.85f + 0.1f * rnd(), na = -a, o = 1.0f;
    const float g = __builtin_bit_cast(float, garb);
    const f32x2 c_lo = {na, g}, c_hi = {g, na}, cc = {na, na}, one_lo = {o, g}, one_hi = {g, o}, one = {o, o};
    for (int it = 0; it < iters; ++it) {
        f32x2 t, u;
        asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(t) : "v"(x));
        if (FORM == 0) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(u) : "v"(t), "v"(c_lo));
        else if (FORM == 1) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(u) : "v"(t), "v"(c_hi));
        else if (FORM == 2) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(u) : "v"(c_hi), "v"(t));
        else if (FORM == 3) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(u) : "v"(c_lo), "v"(t));
        else if (FORM < 8) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(u) : "v"(t), "v"(cc));
        if (FORM < 4) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(x) : "v"(u), "v"(one));
        else if (FORM == 4) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(x) : "v"(u), "v"(one_lo));
        else if (FORM == 5) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(x) : "v"(u), "v"(one_hi));
        else if (FORM == 6) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(x) : "v"(one_hi), "v"(u));
        else if (FORM == 7) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(x) : "v"(one_lo), "v"(u));
        else if (FORM == 8) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(x) : "v"(t), "v"(cc), "v"(one_hi));
        else asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(x) : "v"(t), "v"(cc), "v"(one_lo));
        float t0, t1, u0, u1;
        asm volatile("v_mul_f32 %0, %1, %1" : "=v"(t0) : "v"(y0));
        asm volatile("v_mul_f32 %0, %1, %1" : "=v"(t1) : "v"(y1));
        if (FORM >= 8) {
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(y0) : "v"(t0), "v"(na), "v"(o));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(y1) : "v"(t1), "v"(na), "v"(o));
        } else {
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u0) : "v"(t0), "v"(na));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u1) : "v"(t1), "v"(na));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(y0) : "v"(u0), "v"(o));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(y1) : "v"(u1), "v"(o));
        }
    }
    out[gid * 4 + 0] = x[0];
    out[gid * 4 + 1] = x[1];
    out[gid * 4 + 2] = y0;
    out[gid * 4 + 3] = y1;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 2; } } while (0)

int main(int argc, char** argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    const double seconds = argc > 2 ? atof(argv[2]) : 10.0;
    const int wgs = argc > 3 ? atoi(argv[3]) : 64;
    const int iters = argc > 4 ? atoi(argv[4]) : 100000;
    const size_t n = (size_t)wgs * 64 * 4;
    float* d;
    CK(hipMalloc(&d, n * sizeof(float)));
    if (variant == 3) {   // operand forms x patterns of the unread half: packed against scalar inside each launch, looped for `seconds`
        const unsigned pats[] = {0u, 0x3f800000u, 0x7fc00000u, 0x7f800001u, 0xffffffffu, 0x00000001u, 0xdeadbeefu};
        const char* names[10] = {"mul hi<-src1.lo", "mul lo<-src1.hi", "mul lo<-src0.hi", "mul hi<-src0.lo", "add hi<-src1.lo", "add lo<-src1.hi",
                                 "add lo<-src0.hi", "add hi<-src0.lo", "fma src2 lo<-hi", "fma src2 hi<-lo"};
        std::vector<float> g(n);
        long launches[10] = {0}, bad_launches[10] = {0}, quarter[10][4] = {{0}}, lo_bad[10] = {0}, hi_bad[10] = {0};
        const auto t0 = std::chrono::steady_clock::now();
        do
            for (int form = 0; form < 10; ++form)
                for (unsigned pat : pats) {
                    switch (form) {
                        case 0: hipLaunchKernelGGL(victim_garbage<0>, dim3(wgs), dim3(64), 0, 0, d, iters, pat); break;
                        case 1: hipLaunchKernelGGL(victim_garbage<1>, dim3(wgs), dim3(64), 0, 0, d, iters, pat); break;
                        case 2: hipLaunchKernelGGL(victim_garbage<2>, dim3(wgs), dim3(64), 0, 0, d, iters, pat); break;
                        case 3: hipLaunchKernelGGL(victim_garbage<3>, dim3(wgs), dim3(64), 0, 0, d, iters, pat); break;
                        case 4: hipLaunchKernelGGL(victim_garbage<4>, dim3(wgs), dim3(64), 0, 0, d, iters, pat); break;
                        case 5: hipLaunchKernelGGL(victim_garbage<5>, dim3(wgs), dim3(64), 0, 0, d, iters, pat); break;
                        case 6: hipLaunchKernelGGL(victim_garbage<6>, dim3(wgs), dim3(64), 0, 0, d, iters, pat); break;
                        case 7: hipLaunchKernelGGL(victim_garbage<7>, dim3(wgs), dim3(64), 0, 0, d, iters, pat); break;
                        case 8: hipLaunchKernelGGL(victim_garbage<8>, dim3(wgs), dim3(64), 0, 0, d, iters, pat); break;
                        default: hipLaunchKernelGGL(victim_garbage<9>, dim3(wgs), dim3(64), 0, 0, d, iters, pat); break;
                    }
                    CK(hipMemcpy(g.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
                    bool b = false;
                    for (size_t i = 0; i < n; i += 4) {
                        const bool lo = memcmp(&g[i], &g[i + 2], 4) != 0, hi = memcmp(&g[i + 1], &g[i + 3], 4) != 0;
                        if (lo || hi) { b = true; ++quarter[form][((i / 4) & 63) >> 4]; lo_bad[form] += lo; hi_bad[form] += hi; }
                    }
                    ++launches[form]; bad_launches[form] += b;
                }
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds);
        int any = 0;
        for (int f = 0; f < 10; ++f) {
            printf("form %d %-16s: %4ld launches (x %d waves x %d iterations), %4ld with packed != scalar; lanes 0-15 / 16-31 / 32-47 / 48-63: %ld %ld %ld %ld; "
                   "low half %ld, high half %ld\n", f, names[f], launches[f], wgs, iters, bad_launches[f], quarter[f][0], quarter[f][1], quarter[f][2],
                   quarter[f][3], lo_bad[f], hi_bad[f]);
            any |= bad_launches[f] != 0;
        }
        return any;
    }
    std::vector<float> first(n), got(n);
    long launches = 0, bad_packed = 0, bad_scalar = 0, bad_pair = 0, lanes = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        if (variant == 0) hipLaunchKernelGGL(victim<0>, dim3(wgs), dim3(64), 0, 0, d, iters);
        else if (variant == 1) hipLaunchKernelGGL(victim<1>, dim3(wgs), dim3(64), 0, 0, d, iters);
        else hipLaunchKernelGGL(victim<2>, dim3(wgs), dim3(64), 0, 0, d, iters);
        CK(hipMemcpy(launches ? got.data() : first.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
        const std::vector<float>& g = launches ? got : first;
        bool bp = false, bs = false, bq = false;
        for (size_t i = 0; i < n; i += 4) {
            const bool p = memcmp(&g[i], &first[i], 8) != 0, sc = memcmp(&g[i + 2], &first[i + 2], 8) != 0;
            const bool q = variant != 2 && memcmp(&g[i], &g[i + 2], 8) != 0;
            bp |= p; bs |= sc; bq |= q;
            lanes += p || sc || q;
        }
        bad_packed += bp; bad_scalar += bs; bad_pair += bq;
        ++launches;
    }
    printf("variant %d (%s) wgs %d iters %d: launches %ld, packed differs from launch 0: %ld, scalar differs: %ld, packed != scalar: %ld, lanes %ld\n",
           variant, variant == 0 ? "v_pk_mul+v_pk_fma" : variant == 1 ? "v_pk_mul+v_pk_mul+v_pk_add" : "scalar only", wgs, iters, launches,
           bad_packed, bad_scalar, bad_pair, lanes);
    return (bad_packed || bad_scalar || bad_pair) ? 1 : 0;
}
