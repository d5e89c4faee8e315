// Geometry kernels: nearest-centre region assignment (K6) and farthest point sampling (K7).
//
// Both are index-valued, so the floating-point expressions are written with explicitly rounded
// operations (__fmul_rn/__fadd_rn: no FMA contraction) in the order the reference's PyTorch
// expressions evaluate them; ties resolve to the lowest index like the CPU argmin/argmax.
#include "iq_common.h"
#include "iq_mfma.h"

// Index-valued results depend on individually rounded operations: forbid the compiler from fusing
// a*b+c into an fma anywhere in this file (explicit fmaf / MFMA calls are unaffected).
#pragma clang fp contract(off)

namespace {

// ---- K6: tools/final_util.py:134-147 + final_shapley_value.py:29-31 -------------------------
__global__ __launch_bounds__(256) void region_assign_kernel(const float* __restrict__ cloud,
                                                            const int32_t* __restrict__ fps_idx,
                                                            int32_t* __restrict__ region_id, int N, int R) {
    __shared__ float cs[IQ_MAX_REGIONS * 4];
    if (threadIdx.x < R) {
        const int c = min(max(fps_idx[threadIdx.x], 0), N - 1);  // out-of-range indices: iq_check_index_range
        const float x = cloud[c * 3], y = cloud[c * 3 + 1], z = cloud[c * 3 + 2];
        cs[threadIdx.x * 4 + 0] = x;
        cs[threadIdx.x * 4 + 1] = y;
        cs[threadIdx.x * 4 + 2] = z;
        cs[threadIdx.x * 4 + 3] = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
    }
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const float x = cloud[p * 3], y = cloud[p * 3 + 1], z = cloud[p * 3 + 2];
    const float sx = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
    float best = INFINITY;
    int arg = 0;
    for (int r = 0; r < R; ++r) {
        // matmul row (K = 3) as an fma chain, then * -2, + |src|^2, + |dst|^2
        float dot = __fmul_rn(x, cs[r * 4]);
        dot = __fmaf_rn(y, cs[r * 4 + 1], dot);
        dot = __fmaf_rn(z, cs[r * 4 + 2], dot);
        float d = __fmul_rn(-2.f, dot);
        d = __fadd_rn(d, sx);
        d = __fadd_rn(d, cs[r * 4 + 3]);
        if (d < best) { best = d; arg = r; }
    }
    region_id[p] = arg;
}

// ---- K7: final_save_fps.py:10-31 --------------------------------------------------------------
// One workgroup per cloud; coordinates and running min-distance live in LDS; per iteration one
// wave-shuffle arg-max + one cross-wave step (2 barriers).
constexpr int kFpsThreads = 256;

__device__ inline void argmax_combine(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}

// n_unique (optional): number of sampled points before the running max distance reaches 0, i.e.
// before every remaining point coincides with an already sampled one; from then on arg-max returns
// index 0 forever (CPU tie rule), so the loop stops and the tail is filled with 0.
__global__ __launch_bounds__(kFpsThreads) void fps_kernel(const float* __restrict__ xyz,
                                                          int32_t* __restrict__ idx,
                                                          int32_t* __restrict__ n_unique, int N, int S) {
    extern __shared__ float lds[];
    float* px = lds;           // N
    float* py = px + N;        // N
    float* pz = py + N;        // N
    float* mind = pz + N;      // N
    __shared__ float wv[kFpsThreads / 64];
    __shared__ int wi[kFpsThreads / 64];
    __shared__ int far_s;

    const float* src = xyz + (size_t)blockIdx.x * N * 3;
    for (int p = threadIdx.x; p < N; p += kFpsThreads) {
        px[p] = src[p * 3];
        py[p] = src[p * 3 + 1];
        pz[p] = src[p * 3 + 2];
        mind[p] = 1e10f;
    }
    if (threadIdx.x == 0) far_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int it = 0; it < S; ++it) {
        const int far = far_s;
        if (threadIdx.x == 0) idx[(size_t)blockIdx.x * S + it] = far;
        const float cx = px[far], cy = py[far], cz = pz[far];
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int p = threadIdx.x; p < N; p += kFpsThreads) {
            const float dx = __fsub_rn(px[p], cx), dy = __fsub_rn(py[p], cy), dz = __fsub_rn(pz[p], cz);
            const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            float m = mind[p];
            if (d < m) { m = d; mind[p] = d; }
            if (m > bv) { bv = m; bi = p; }  // increasing p: strict > keeps the lowest index
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(bi, off);
            argmax_combine(bv, bi, ov, oi);
        }
        __syncthreads();  // everyone has read far_s
        if (lane == 0) { wv[wave] = bv; wi[wave] = bi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float v = wv[0];
            int i = wi[0];
            for (int w = 1; w < kFpsThreads / 64; ++w) argmax_combine(v, i, wv[w], wi[w]);
            far_s = (v == 0.f) ? -1 - it : i;  // negative = exhausted after `it + 1` samples
        }
        __syncthreads();
        if (far_s < 0) {
            const int done = it + 1;
            for (int j = done + threadIdx.x; j < S; j += kFpsThreads) idx[(size_t)blockIdx.x * S + j] = 0;
            if (threadIdx.x == 0 && n_unique) n_unique[blockIdx.x] = done;
            return;
        }
    }
    if (threadIdx.x == 0 && n_unique) n_unique[blockIdx.x] = S;
}

// One WAVE per cloud, the points in registers (lane l owns points l, l + 64, ...): no LDS, no barrier.  FPS is a chain of S
// dependent arg-max rounds, so a launch is latency-bound per cloud and throughput comes from how many clouds are in flight;
// the workgroup form above keeps 2-3 clouds per CU busy with two barriers per round, this one a wave per cloud on every SIMD.
// Same arithmetic per point and the same tie rule (largest distance, then lowest index), so the picks are identical.
// The arg-max runs on a 64-bit key (float bits of the running min-distance, which is >= +0, above the complemented index);
// the winner's coordinates are read from the lane that owns it (its local best IS the winner).
template <int PPL>
__global__ __launch_bounds__(64) void fps_wave_kernel(const float* __restrict__ xyz, int32_t* __restrict__ idx,
                                                      int32_t* __restrict__ n_unique, int N, int S) {
    const int lane = threadIdx.x;
    const float* src = xyz + (size_t)blockIdx.x * N * 3;
    float px[PPL], py[PPL], pz[PPL], md[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        const int p = lane + 64 * j;
        const bool live = p < N;
        px[j] = live ? src[p * 3] : 0.f;
        py[j] = live ? src[p * 3 + 1] : 0.f;
        pz[j] = live ? src[p * 3 + 2] : 0.f;
        md[j] = live ? 1e10f : -1.f;           // a point that does not exist never wins (keys are built from md >= 0 only)
    }
    int far = 0;
    float cx, cy, cz;
    cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(px[0]), 0));
    cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(py[0]), 0));
    cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pz[0]), 0));
    int32_t* out = idx + (size_t)blockIdx.x * S;
    for (int it = 0; it < S; ++it) {
        if (lane == 0) out[it] = far;
        unsigned long long best = 0ull;         // (float bits of min-distance) << 32 | ~index ; 0 = nothing
        float bx = 0.f, by = 0.f, bz = 0.f;
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const float dx = __fsub_rn(px[j], cx), dy = __fsub_rn(py[j], cy), dz = __fsub_rn(pz[j], cz);
            const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            float m = md[j];
            if (d < m && m >= 0.f) { m = d; md[j] = d; }
            if (m >= 0.f) {
                const unsigned long long key = ((unsigned long long)__float_as_uint(m) << 32) | (unsigned)(~(lane + 64 * j));
                if (key > best) { best = key; bx = px[j]; by = py[j]; bz = pz[j]; }
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const unsigned lo = __shfl_xor((unsigned)(best & 0xffffffffu), off);
            const unsigned hi = __shfl_xor((unsigned)(best >> 32), off);
            const unsigned long long other = ((unsigned long long)hi << 32) | lo;
            best = other > best ? other : best;
        }
        const float v = __uint_as_float((unsigned)(best >> 32));
        const int win = (int)~(unsigned)(best & 0xffffffffu);
        if (v == 0.f) {                         // every remaining point coincides with a sampled one: arg-max returns 0 from now on
            for (int j2 = it + 1 + lane; j2 < S; j2 += 64) out[j2] = 0;
            if (lane == 0 && n_unique) n_unique[blockIdx.x] = it + 1;
            return;
        }
        far = win;
        const int owner = __builtin_amdgcn_readfirstlane(win) & 63;
        cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bx), owner));
        cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(by), owner));
        cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bz), owner));
    }
    if (lane == 0 && n_unique) n_unique[blockIdx.x] = S;
}

}  // namespace

extern "C" int iq_region_assign(const float* cloud, const int32_t* fps_idx, int32_t* region_id,
                                int N, int R, iq_stream_t stream) {
    IQ_REQUIRE(cloud && fps_idx && region_id, "iq_region_assign: null pointer");
    IQ_REQUIRE(N > 0 && R >= 1 && R <= IQ_MAX_REGIONS, "iq_region_assign: N=%d R=%d", N, R);
    hipLaunchKernelGGL(region_assign_kernel, dim3((N + 255) / 256), dim3(256), 0, iq::as_stream(stream),
                       cloud, fps_idx, region_id, N, R);
    return iq::check_launch("region_assign_kernel");
}

int iq::launch_fps(const float* xyz, int32_t* idx, int32_t* n_unique, int B, int N, int S, hipStream_t st) {
    IQ_REQUIRE(B >= 0 && N > 0 && N <= 8192 && S >= 1, "iq_fps: B=%d N=%d S=%d", B, N, S);
    if (B == 0) return IQ_OK;
    IQ_REQUIRE(xyz && idx, "iq_fps: null pointer");
    // many clouds of at most 1024 points: one wave per cloud (points in registers); few or larger clouds: one workgroup per cloud
    if (B >= 64 && N <= 1024) {
        if (N <= 128) hipLaunchKernelGGL(fps_wave_kernel<2>, dim3(B), dim3(64), 0, st, xyz, idx, n_unique, N, S);
        else if (N <= 512) hipLaunchKernelGGL(fps_wave_kernel<8>, dim3(B), dim3(64), 0, st, xyz, idx, n_unique, N, S);
        else hipLaunchKernelGGL(fps_wave_kernel<16>, dim3(B), dim3(64), 0, st, xyz, idx, n_unique, N, S);
        return iq::check_launch("fps_wave_kernel");
    }
    const size_t lds = (size_t)N * 4 * sizeof(float);
    if (lds > 48 * 1024) {  // above the default dynamic-LDS limit (N > 3072): opt in, up to 128 KB at N = 8192
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(fps_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return iq::fail(IQ_ELAUNCH, "iq_fps: cannot reserve %zu bytes of LDS for N=%d", lds, N);
    }
    hipLaunchKernelGGL(fps_kernel, dim3(B), dim3(kFpsThreads), lds, st, xyz, idx, n_unique, N, S);
    return iq::check_launch("fps_kernel");
}

extern "C" int iq_fps(const float* xyz, int32_t* idx, int B, int N, int S, iq_stream_t stream) {
    return iq::launch_fps(xyz, idx, nullptr, B, N, S, iq::as_stream(stream));
}
