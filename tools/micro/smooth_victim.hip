// Reduced clone of the smoothness enumeration's gradient loop (interpret_quality_amd/csrc/iq_smooth.hip) for BISECTING round 4's
// shared-GPU effect: which part of the loop, compiled WITH packed float32 instructions, changes its bits beside the bf16x3
// chain kernel of a second process?  One wave per workgroup, the region's points in LDS, three code regions that can each be
// compiled with or without the packed-fp32 subtarget feature (noinline wrappers around the same source):
//   A  variances()           LDS reads -> projections -> wave sums (ds_bpermute) -> float32 divisions
//   B  gradient step         LDS reads -> gradient -> wave sum -> sqrt -> per-coordinate division -> LDS writes
//   C  distance bound count  LDS reads of cur and org -> sqrt -> compare -> integer wave sum
// variant = 3 bits (A | B<<1 | C<<2): bit set = that region may use packed float32.  Every launch is compared with the first.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/micro/smooth_victim.bin tools/micro/smooth_victim.hip
// Run  : tools/micro/smooth_victim.bin <variant 0..7 | 100 + bits of the variances() parts> <seconds> [workgroups=96] [steps=400]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#if defined(__HIP_DEVICE_COMPILE__)
#define NOPK __attribute__((target("no-packed-fp32-ops")))
#else
#define NOPK
#endif
constexpr int kWave = 64, kMaxS = 128;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
struct Vars { float var[3]; float mean[3]; };
struct Ori { float o[3][3]; };

__device__ __forceinline__ Vars variances_impl(const float* pts, int S, int lane, const Ori& O) {
    Vars r;
    float s[3] = {0.f, 0.f, 0.f};
    for (int i = lane; i < S; i += kWave) {
        const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
#pragma unroll
        for (int k = 0; k < 3; ++k) s[k] += x * O.o[k][0] + y * O.o[k][1] + z * O.o[k][2];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) r.mean[k] = wave_sum(s[k]) / (float)S;
    float q[3] = {0.f, 0.f, 0.f};
    for (int i = lane; i < S; i += kWave) {
        const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float d = (x * O.o[k][0] + y * O.o[k][1] + z * O.o[k][2]) - r.mean[k];
            q[k] += d * d;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) r.var[k] = wave_sum(q[k]) / (float)(S - 1);
    return r;
}
__device__ __forceinline__ void gradient_impl(float* cur, int S, int lane, const Ori& O, const Vars& v, const float ck[3], float step) {
    const float two_over = 2.f / (float)(S - 1);
    float n2 = 0.f;
    for (int i = lane; i < S; i += kWave) {
        const float x = cur[3 * i], y = cur[3 * i + 1], z = cur[3 * i + 2];
        float g[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < 3; ++k) {
            const float gp = ck[k] * (two_over * ((x * O.o[k][0] + y * O.o[k][1] + z * O.o[k][2]) - v.mean[k]));
            for (int c = 0; c < 3; ++c) g[c] += gp * O.o[k][c];
        }
        n2 += g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
    }
    const float norm = sqrtf(wave_sum(n2));
    for (int i = lane; i < S; i += kWave) {
        const float x = cur[3 * i], y = cur[3 * i + 1], z = cur[3 * i + 2];
        float g[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < 3; ++k) {
            const float gp = ck[k] * (two_over * ((x * O.o[k][0] + y * O.o[k][1] + z * O.o[k][2]) - v.mean[k]));
            for (int c = 0; c < 3; ++c) g[c] += gp * O.o[k][c];
        }
        for (int c = 0; c < 3; ++c) {
            const float delta = (norm != 0.f) ? (step * g[c]) / norm : 1e-8f;
            cur[3 * i + c] = cur[3 * i + c] + delta;
        }
    }
}
__device__ __forceinline__ int bound_impl(const float* cur, const float* org, int S, int lane, float dth) {
    int count = 0;
    for (int i = lane; i < S; i += kWave) {
        const float dx = cur[3 * i] - org[3 * i], dy = cur[3 * i + 1] - org[3 * i + 1], dz = cur[3 * i + 2] - org[3 * i + 2];
        const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
        if (dist > dth) ++count;
    }
    return wave_sum_i(count);
}
__device__ __noinline__ Vars variances_pk(const float* p, int S, int lane, const Ori& O) { return variances_impl(p, S, lane, O); }
__device__ __noinline__ NOPK Vars variances_np(const float* p, int S, int lane, const Ori& O) { return variances_impl(p, S, lane, O); }
__device__ __noinline__ void gradient_pk(float* c, int S, int l, const Ori& O, const Vars& v, const float ck[3], float st) { gradient_impl(c, S, l, O, v, ck, st); }
__device__ __noinline__ NOPK void gradient_np(float* c, int S, int l, const Ori& O, const Vars& v, const float ck[3], float st) { gradient_impl(c, S, l, O, v, ck, st); }
__device__ __noinline__ int bound_pk(const float* c, const float* o, int S, int l, float d) { return bound_impl(c, o, S, l, d); }
__device__ __noinline__ NOPK int bound_np(const float* c, const float* o, int S, int l, float d) { return bound_impl(c, o, S, l, d); }

template <int V>
__global__ NOPK __launch_bounds__(kWave) void victim(float* out, int steps) {
    __shared__ float cur[kMaxS * 3];
    __shared__ float org[kMaxS * 3];
    const int lane = threadIdx.x, r = blockIdx.x;
    unsigned s = 0x9e3779b9u * (r + 1);
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f); };
    const int S = 20 + (int)(rnd() * 90.f);                  // 20 .. 109 points, as the regions of a 1024-point cloud
    Ori O;                                                   // some fixed orientation (orthonormal up to rounding is not needed)
    for (int k = 0; k < 3; ++k)
        for (int c = 0; c < 3; ++c) O.o[k][c] = (k == c ? 0.9f : 0.f) + 0.3f * (rnd() - 0.5f);
    unsigned t = s ^ (0x85ebca6bu * (lane + 1));
    for (int i = lane; i < S; i += kWave)
        for (int c = 0; c < 3; ++c) {
            t = t * 1664525u + 1013904223u;
            const float v = 0.2f * ((float)(t >> 8) * (1.0f / 16777216.0f) - 0.5f);
            cur[3 * i + c] = v;
            org[3 * i + c] = v;
        }
    __syncthreads();
    Vars v{};
    int counts = 0;
    for (int it = 0; it < steps; ++it) {
        v = (V & 1) ? variances_pk(cur, S, lane, O) : variances_np(cur, S, lane, O);
        const float smax = fmaxf(v.var[0], fmaxf(v.var[1], v.var[2]));
        const float ck[3] = {-1.f / smax, 1.f / smax - (smax - v.var[1]) / (smax * smax), 0.5f / smax};
        if (V & 2) gradient_pk(cur, S, lane, O, v, ck, 0.003f); else gradient_np(cur, S, lane, O, v, ck, 0.003f);
        counts += (V & 4) ? bound_pk(cur, org, S, lane, 0.05f) : bound_np(cur, org, S, lane, 0.05f);
    }
    float* o = out + (size_t)r * (kMaxS * 3 + 8);
    for (int i = lane; i < kMaxS * 3; i += kWave) o[i] = i < 3 * S ? cur[i] : 0.f;
    if (lane == 0) {
        for (int k = 0; k < 3; ++k) { o[kMaxS * 3 + k] = v.var[k]; o[kMaxS * 3 + 3 + k] = v.mean[k]; }
        o[kMaxS * 3 + 6] = (float)counts;
        o[kMaxS * 3 + 7] = (float)S;
    }
}

// ---- second level: variances() cut into four parts, each with or without packed float32 (gradient and bound never packed) ----
//   bit 0  P1  first loop: LDS reads -> projections -> per-lane sums s[k]
//   bit 1  P2  wave sums of s[k] (ds_bpermute) + the divisions -> mean[k]
//   bit 2  P3  second loop: LDS reads -> (projection - mean)^2 -> per-lane sums q[k]
//   bit 3  P4  wave sums of q[k] + the divisions -> var[k]
struct F3 { float v[3]; };
template <int MODE = 0>
__device__ __forceinline__ F3 p1_impl(const float* pts, int S, int lane, const Ori& O) {
    F3 s{{0.f, 0.f, 0.f}};
    for (int i = lane; i < S; i += kWave) {
        float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        if (MODE == 1) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(x), "+v"(y), "+v"(z));   // the loads have landed (s_waitcnt before this), then 16 idle cycles
#pragma unroll
        for (int k = 0; k < 3; ++k) s.v[k] += x * O.o[k][0] + y * O.o[k][1] + z * O.o[k][2];
    }
    return s;
}
__device__ __forceinline__ F3 p2_impl(const F3& s, float denom) {
    F3 r;
#pragma unroll
    for (int k = 0; k < 3; ++k) r.v[k] = wave_sum(s.v[k]) / denom;
    return r;
}
__device__ __forceinline__ F3 p3_impl(const float* pts, int S, int lane, const Ori& O, const F3& mean) {
    F3 q{{0.f, 0.f, 0.f}};
    for (int i = lane; i < S; i += kWave) {
        const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float d = (x * O.o[k][0] + y * O.o[k][1] + z * O.o[k][2]) - mean.v[k];
            q.v[k] += d * d;
        }
    }
    return q;
}
__device__ __noinline__ F3 p1_pk(const float* p, int S, int l, const Ori& O) { return p1_impl(p, S, l, O); }
__device__ __noinline__ F3 p1_pk_nop(const float* p, int S, int l, const Ori& O) { return p1_impl<1>(p, S, l, O); }
__device__ __noinline__ NOPK F3 p1_np(const float* p, int S, int l, const Ori& O) { return p1_impl(p, S, l, O); }
__device__ __noinline__ F3 p2_pk(const F3& s, float d) { return p2_impl(s, d); }
__device__ __noinline__ NOPK F3 p2_np(const F3& s, float d) { return p2_impl(s, d); }
__device__ __noinline__ F3 p3_pk(const float* p, int S, int l, const Ori& O, const F3& m) { return p3_impl(p, S, l, O, m); }
__device__ __noinline__ NOPK F3 p3_np(const float* p, int S, int l, const Ori& O, const F3& m) { return p3_impl(p, S, l, O, m); }

template <int V>
__global__ NOPK __launch_bounds__(kWave) void victim2(float* out, int steps) {
    __shared__ float cur[kMaxS * 3];
    __shared__ float org[kMaxS * 3];
    const int lane = threadIdx.x, r = blockIdx.x;
    unsigned s = 0x9e3779b9u * (r + 1);
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f); };
    const int S = 20 + (int)(rnd() * 90.f);
    Ori O;
    for (int k = 0; k < 3; ++k)
        for (int c = 0; c < 3; ++c) O.o[k][c] = (k == c ? 0.9f : 0.f) + 0.3f * (rnd() - 0.5f);
    unsigned t = s ^ (0x85ebca6bu * (lane + 1));
    for (int i = lane; i < S; i += kWave)
        for (int c = 0; c < 3; ++c) {
            t = t * 1664525u + 1013904223u;
            const float v = 0.2f * ((float)(t >> 8) * (1.0f / 16777216.0f) - 0.5f);
            cur[3 * i + c] = v;
            org[3 * i + c] = v;
        }
    __syncthreads();
    Vars v{};
    int counts = 0;
    for (int it = 0; it < steps; ++it) {
        const F3 s1 = (V & 1) ? p1_pk(cur, S, lane, O) : p1_np(cur, S, lane, O);
        const F3 mean = (V & 2) ? p2_pk(s1, (float)S) : p2_np(s1, (float)S);
        const F3 q = (V & 4) ? p3_pk(cur, S, lane, O, mean) : p3_np(cur, S, lane, O, mean);
        const F3 var = (V & 8) ? p2_pk(q, (float)(S - 1)) : p2_np(q, (float)(S - 1));
        for (int k = 0; k < 3; ++k) { v.mean[k] = mean.v[k]; v.var[k] = var.v[k]; }
        const float smax = fmaxf(v.var[0], fmaxf(v.var[1], v.var[2]));
        const float ck[3] = {-1.f / smax, 1.f / smax - (smax - v.var[1]) / (smax * smax), 0.5f / smax};
        gradient_np(cur, S, lane, O, v, ck, 0.003f);
        counts += bound_np(cur, org, S, lane, 0.05f);
    }
    float* o = out + (size_t)r * (kMaxS * 3 + 8);
    for (int i = lane; i < kMaxS * 3; i += kWave) o[i] = i < 3 * S ? cur[i] : 0.f;
    if (lane == 0) {
        for (int k = 0; k < 3; ++k) { o[kMaxS * 3 + k] = v.var[k]; o[kMaxS * 3 + 3 + k] = v.mean[k]; }
        o[kMaxS * 3 + 6] = (float)counts;
        o[kMaxS * 3 + 7] = (float)S;
    }
}

// ---- third level: the packed and the scalar build of loop 1 (P1) on the SAME LDS contents, inside one launch ----------------
// Every step calls p1_pk and p1_np back to back on the same points and compares the three per-lane sums bit for bit; the points
// then move a little (as the gradient step moves them) so that a STALE value - a register or LDS word of the step before - would
// show as a small deviation.  out (per workgroup): [0] steps with a mismatch, [1] first such step, [2..4] packed sums, [5..7]
// scalar sums of the first mismatching lane at that step, [8] that lane, [9] S, [10..12] the sums the scalar loop gave on the
// step BEFORE (what a stale operand would reproduce), [13] mismatches where packed == previous step's scalar sums (bitwise).
__global__ NOPK __launch_bounds__(kWave) void victim3(float* out, int steps, int order, float* gcopy) {
    __shared__ float cur[kMaxS * 3];
    const int lane = threadIdx.x, r = blockIdx.x;
    unsigned s = 0x9e3779b9u * (r + 1);
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f); };
    const int S = 20 + (int)(rnd() * 90.f);
    Ori O;
    for (int k = 0; k < 3; ++k)
        for (int c = 0; c < 3; ++c) O.o[k][c] = (k == c ? 0.9f : 0.f) + 0.3f * (rnd() - 0.5f);
    unsigned t = s ^ (0x85ebca6bu * (lane + 1));
    for (int i = lane; i < S; i += kWave)
        for (int c = 0; c < 3; ++c) {
            t = t * 1664525u + 1013904223u;
            cur[3 * i + c] = 0.2f * ((float)(t >> 8) * (1.0f / 16777216.0f) - 0.5f);
        }
    __syncthreads();
    int nbad = 0, first = -1, stale_hits = 0;
    F3 keep_pk{}, keep_np{}, keep_prev{}, prev{};
    for (int it = 0; it < steps; ++it) {
        F3 a, b;
        if (order == 0) { a = p1_pk(cur, S, lane, O); b = p1_np(cur, S, lane, O); }
        else if (order == 1) { b = p1_np(cur, S, lane, O); a = p1_pk(cur, S, lane, O); }
        else if (order == 2) { a = p1_pk_nop(cur, S, lane, O); b = p1_np(cur, S, lane, O); }
        else {   // the same points through GLOBAL memory (a per-workgroup copy): the packed loop without LDS
            float* g = gcopy + (size_t)r * kMaxS * 3;
            for (int i = lane; i < 3 * S; i += kWave) g[i] = cur[i];
            __threadfence_block();
            a = p1_pk(g, S, lane, O); b = p1_np(g, S, lane, O);
        }
        bool bad = false;
        for (int k = 0; k < 3; ++k) bad |= __float_as_uint(a.v[k]) != __float_as_uint(b.v[k]);
        if (bad) {
            bool st = true;
            for (int k = 0; k < 3; ++k) st &= __float_as_uint(a.v[k]) == __float_as_uint(prev.v[k]);
            stale_hits += st;
            if (first < 0) { first = it; keep_pk = a; keep_np = b; keep_prev = prev; }
            ++nbad;
        }
        prev = b;
        for (int i = lane; i < S; i += kWave)                       // the points move a little, as under the gradient step
            for (int c = 0; c < 3; ++c) cur[3 * i + c] += 1e-4f * (b.v[c] + 0.01f);
    }
    // one record per workgroup: the lane with the earliest mismatch
    unsigned long long key = first < 0 ? ~0ull : ((unsigned long long)first << 8 | lane);
    unsigned long long best = key;
    for (int o = 32; o >= 1; o >>= 1) { const unsigned long long other = __shfl_xor(best, o, kWave); best = other < best ? other : best; }
    int total = nbad, stale = stale_hits;
    for (int o = 32; o >= 1; o >>= 1) { total += __shfl_xor(total, o, kWave); stale += __shfl_xor(stale, o, kWave); }
    float* o = out + (size_t)r * (kMaxS * 3 + 8);
    if (key == best && first >= 0) {
        o[1] = (float)first;
        for (int k = 0; k < 3; ++k) { o[2 + k] = keep_pk.v[k]; o[5 + k] = keep_np.v[k]; o[10 + k] = keep_prev.v[k]; }
        o[8] = (float)lane;
    }
    if (lane == 0) { o[0] = (float)total; o[9] = (float)S; o[13] = (float)stale; if (best == ~0ull) o[1] = -1.f; }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 2; } } while (0)
template <int V> void launch(float* d, int wgs, int steps) { hipLaunchKernelGGL(victim<V>, dim3(wgs), dim3(kWave), 0, 0, d, steps); }
template <int V> void launch2(float* d, int wgs, int steps) { hipLaunchKernelGGL(victim2<V>, dim3(wgs), dim3(kWave), 0, 0, d, steps); }

int main(int argc, char** argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 7;
    const double seconds = argc > 2 ? atof(argv[2]) : 10.0;
    const int wgs = argc > 3 ? atoi(argv[3]) : 96;
    const int steps = argc > 4 ? atoi(argv[4]) : 400;
    const size_t per = kMaxS * 3 + 8, n = (size_t)wgs * per;
    float* d;
    CK(hipMalloc(&d, n * sizeof(float)));
    if (variant >= 200) {      // 200 / 201: packed against scalar inside one launch (packed first / scalar first); 202: 16 idle cycles
        std::vector<float> g(n);   // between the LDS data's arrival and its first use; 203: the points through global memory, not LDS
        float* gcopy;
        CK(hipMalloc(&gcopy, (size_t)wgs * kMaxS * 3 * sizeof(float)));
        long launches = 0, bad_launches = 0, bad_steps = 0, stale = 0, shown = 0;
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
            CK(hipMemset(d, 0, n * sizeof(float)));
            hipLaunchKernelGGL(victim3, dim3(wgs), dim3(kWave), 0, 0, d, steps, variant - 200, gcopy);
            CK(hipMemcpy(g.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
            bool b = false;
            for (int w = 0; w < wgs; ++w) {
                const float* o = &g[w * per];
                if (o[0] > 0) {
                    b = true; bad_steps += (long)o[0]; stale += (long)o[13];
                    if (shown < 12) {
                        ++shown;
                        printf("  launch %ld wg %d (S=%g): %g lane-steps differ, first at step %g lane %g: packed %.9g %.9g %.9g | scalar %.9g %.9g %.9g | "
                               "scalar of the step before %.9g %.9g %.9g\n", launches, w, o[9], o[0], o[1], o[8], o[2], o[3], o[4], o[5], o[6], o[7],
                               o[10], o[11], o[12]);
                    }
                }
            }
            bad_launches += b;
            ++launches;
        }
        printf("variant %d (%s) wgs %d steps %d: launches %ld, with packed != scalar: %ld, lane-steps %ld, of which packed == the scalar sums of "
               "the step before: %ld\n", variant, variant == 200 ? "packed first" : variant == 201 ? "scalar first" : variant == 202 ?
               "packed, 16 idle cycles after the LDS data arrived" : "packed, points read from global memory", wgs, steps, launches, bad_launches,
               bad_steps, stale);
        return bad_launches ? 1 : 0;
    }
    std::vector<float> first(n), got(n);
    long launches = 0, bad = 0, bad_wgs = 0;
    double worst = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        if (variant >= 100) switch (variant - 100) {
            case 0: launch2<0>(d, wgs, steps); break; case 1: launch2<1>(d, wgs, steps); break;
            case 2: launch2<2>(d, wgs, steps); break; case 4: launch2<4>(d, wgs, steps); break;
            case 8: launch2<8>(d, wgs, steps); break; case 5: launch2<5>(d, wgs, steps); break;
            case 10: launch2<10>(d, wgs, steps); break; default: launch2<15>(d, wgs, steps); break;
        } else
        switch (variant) {
            case 0: launch<0>(d, wgs, steps); break; case 1: launch<1>(d, wgs, steps); break;
            case 2: launch<2>(d, wgs, steps); break; case 3: launch<3>(d, wgs, steps); break;
            case 4: launch<4>(d, wgs, steps); break; case 5: launch<5>(d, wgs, steps); break;
            case 6: launch<6>(d, wgs, steps); break; default: launch<7>(d, wgs, steps); break;
        }
        CK(hipMemcpy(launches ? got.data() : first.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
        if (launches) {
            bool b = false;
            for (int w = 0; w < wgs; ++w)
                if (memcmp(&got[w * per], &first[w * per], per * sizeof(float)) != 0) {
                    b = true; ++bad_wgs;
                    for (size_t i = 0; i < per; ++i) { const double e = fabs((double)got[w * per + i] - first[w * per + i]); if (e > worst) worst = e; }
                }
            bad += b;
        }
        ++launches;
    }
    if (variant >= 100)
        printf("variant %d = variances() parts with packed float32:%s%s%s%s%s | ", variant, (variant - 100) & 1 ? " P1(loop 1)" : "",
               (variant - 100) & 2 ? " P2(wave sums + div -> mean)" : "", (variant - 100) & 4 ? " P3(loop 2)" : "",
               (variant - 100) & 8 ? " P4(wave sums + div -> var)" : "", variant == 100 ? " none" : "");
    if (variant >= 100) printf("wgs %d steps %d: launches %ld, differ from launch 0: %ld (%ld workgroups, max |d| %.3g)\n", wgs, steps, launches, bad, bad_wgs, worst);
    else
    printf("variant %d (packed float32 allowed in:%s%s%s%s) wgs %d steps %d: launches %ld, differ from launch 0: %ld (%ld workgroups, max |d| %.3g); "
           "first var %.9g counts %g\n", variant, variant & 1 ? " variances" : "", variant & 2 ? " gradient" : "", variant & 4 ? " bound" : "",
           variant ? "" : " nothing", wgs, steps, launches, bad, bad_wgs, worst, first[kMaxS * 3], first[kMaxS * 3 + 6]);
    return bad ? 1 : 0;
}
