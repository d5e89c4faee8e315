// Coalition masking kernels: HBM-write-bound streaming.
//
// One workgroup writes ROWS_PER_WG = 4 consecutive output clouds (48 KB at N = 1024).  The source
// cloud (12 KB), region ids and centre are re-read from L2 by every workgroup - algorithmic traffic
// is the 12 288 B written per coalition.  Each lane stores 16 B (float4), consecutive lanes store
// consecutive addresses.
#include <algorithm>

#include "iq_common.h"

namespace {

constexpr int kRowsPerWg = 4;
constexpr int kThreads = 256;

enum KeepMode { kExplicit = 0, kShapleyPrefix = 1, kInteraction = 2 };

struct MaskArgs {
    const float* cloud;        // (N,3)
    const int32_t* region_id;  // (N)
    const float* center;       // (3)
    float* out;
    const uint64_t* keep;      // explicit: (rows)
    const int32_t* orders;     // shapley: (bs,R)
    const int32_t* pairs;      // interaction: (nb,2)
    const uint64_t* ctx;       // interaction: (nb)
    int N, R, rows, channel_first, mode;
};

__device__ inline uint64_t row_keep(const MaskArgs& a, int g) {
    if (a.mode == kExplicit) return a.keep[g];
    if (a.mode == kShapleyPrefix) {
        // row i of order o keeps orders[o][0..i-1]  (tools/final_common.py:56-60)
        const int o = g / (a.R + 1), i = g % (a.R + 1);
        uint64_t m = 0;
        for (int j = 0; j < i; ++j) m |= 1ull << (a.orders[o * a.R + j] & 63);
        return m;
    }
    // interaction: rows 4k..4k+3 = S+{i,j}, S+{i}, S+{j}, S
    const int k = g >> 2, which = g & 3;
    uint64_t m = a.ctx[k];
    const uint64_t bi = 1ull << (a.pairs[2 * k] & 63), bj = 1ull << (a.pairs[2 * k + 1] & 63);
    if (which == 0) m |= bi | bj;
    if (which == 1) m |= bi;
    if (which == 2) m |= bj;
    return m;
}

__global__ __launch_bounds__(kThreads) void mask_rows_kernel(MaskArgs a) {
    __shared__ uint64_t keep_s[kRowsPerWg];
    const int g0 = blockIdx.x * kRowsPerWg;
    if (threadIdx.x < kRowsPerWg) {
        const int g = g0 + threadIdx.x;
        keep_s[threadIdx.x] = g < a.rows ? row_keep(a, g) : 0ull;
    }
    __syncthreads();
    const size_t cloud_floats = (size_t)a.N * 3;
    if (a.N & 3) {   // a cloud size that is no multiple of 4: the output clouds are not 16-byte aligned - element by element
        for (int e = threadIdx.x; e < a.N * 3; e += kThreads) {
            int p, ch;
            if (a.channel_first) { ch = e / a.N; p = e - ch * a.N; }
            else                 { p = e / 3;    ch = e - p * 3; }
            const float x = a.cloud[p * 3 + ch], c = a.center[ch];
            const int rid = a.region_id[p];
            for (int r = 0; r < kRowsPerWg && g0 + r < a.rows; ++r)
                a.out[(size_t)(g0 + r) * cloud_floats + e] = iq::keep_bit(keep_s[r], rid) ? x : c;
        }
        return;
    }
    const int nvec = a.N * 3 / 4;
    for (int e4 = threadIdx.x; e4 < nvec; e4 += kThreads) {
        float x[4], c[4];
        int rid[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = e4 * 4 + q;
            int p, ch;
            if (a.channel_first) { ch = e / a.N; p = e - ch * a.N; }
            else                 { p = e / 3;    ch = e - p * 3; }
            x[q] = a.cloud[p * 3 + ch];
            c[q] = a.center[ch];
            rid[q] = a.region_id[p];
        }
#pragma unroll
        for (int r = 0; r < kRowsPerWg; ++r) {
            const int g = g0 + r;
            if (g >= a.rows) break;
            const uint64_t m = keep_s[r];
            float4 o;
            o.x = iq::keep_bit(m, rid[0]) ? x[0] : c[0];
            o.y = iq::keep_bit(m, rid[1]) ? x[1] : c[1];
            o.z = iq::keep_bit(m, rid[2]) ? x[2] : c[2];
            o.w = iq::keep_bit(m, rid[3]) ? x[3] : c[3];
            reinterpret_cast<float4*>(a.out + (size_t)g * cloud_floats)[e4] = o;
        }
    }
}

// first position whose value lies outside [lo, hi), or 0xffffffff
__global__ __launch_bounds__(256) void index_range_kernel(const int32_t* __restrict__ idx, size_t count, int lo, int hi,
                                                          uint32_t* __restrict__ first_bad) {
    uint32_t bad = 0xffffffffu;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        const int v = idx[i];
        if (v < lo || v >= hi) { bad = (uint32_t)min(i, (size_t)0xfffffffeu); break; }
    }
    if (bad != 0xffffffffu) atomicMin(first_bad, bad);
}

int launch(MaskArgs a, iq_stream_t stream) {
    if (a.rows == 0) return IQ_OK;
    IQ_REQUIRE(a.cloud && a.region_id && a.center && a.out, "mask: null pointer");
    IQ_REQUIRE(a.N > 0 && a.N <= IQ_MAX_POINTS, "mask: N=%d not in (0,%d]", a.N, IQ_MAX_POINTS);
    IQ_REQUIRE(a.R >= 0 && a.R <= IQ_MAX_REGIONS, "mask: R=%d out of range", a.R);
    const int grid = (a.rows + kRowsPerWg - 1) / kRowsPerWg;
    hipLaunchKernelGGL(mask_rows_kernel, dim3(grid), dim3(kThreads), 0, iq::as_stream(stream), a);
    return iq::check_launch("mask_rows_kernel");
}

}  // namespace

extern "C" int iq_mask_shapley(const float* cloud, const int32_t* region_id, const int32_t* orders,
                               const float* center, float* out, int N, int R, int bs,
                               int channel_first, iq_stream_t stream) {
    IQ_REQUIRE(bs >= 0 && R >= 1, "iq_mask_shapley: bs=%d R=%d", bs, R);
    IQ_REQUIRE(orders || bs == 0, "iq_mask_shapley: null orders");
    MaskArgs a{cloud, region_id, center, out, nullptr, orders, nullptr, nullptr,
               N, R, bs * (R + 1), channel_first, kShapleyPrefix};
    return launch(a, stream);
}

extern "C" int iq_mask_interaction(const float* cloud, const int32_t* region_id, const int32_t* pairs,
                                   const uint64_t* ctx_mask, const float* center, float* out,
                                   int N, int R, int nb, iq_stream_t stream) {
    IQ_REQUIRE(nb >= 0, "iq_mask_interaction: nb=%d", nb);
    IQ_REQUIRE((pairs && ctx_mask) || nb == 0, "iq_mask_interaction: null pairs/ctx");
    MaskArgs a{cloud, region_id, center, out, nullptr, nullptr, pairs, ctx_mask,
               N, R, 4 * nb, 1, kInteraction};
    return launch(a, stream);
}

extern "C" int iq_mask_coalitions(const float* cloud, const int32_t* region_id, const uint64_t* keep,
                                  const float* center, float* out, int N, int B, int channel_first,
                                  iq_stream_t stream) {
    IQ_REQUIRE(B >= 0, "iq_mask_coalitions: B=%d", B);
    IQ_REQUIRE(keep || B == 0, "iq_mask_coalitions: null keep");
    MaskArgs a{cloud, region_id, center, out, keep, nullptr, nullptr, nullptr,
               N, IQ_MAX_REGIONS, B, channel_first, kExplicit};
    return launch(a, stream);
}

extern "C" int iq_check_index_range(const int32_t* idx, size_t count, int lo, int hi, uint32_t* scratch, iq_stream_t stream) {
    IQ_REQUIRE(scratch && (idx || count == 0), "iq_check_index_range: null pointer");
    if (count == 0) return IQ_OK;
    hipStream_t st = iq::as_stream(stream);
    uint32_t bad = 0xffffffffu;
    if (hipMemcpyAsync(scratch, &bad, sizeof(bad), hipMemcpyHostToDevice, st) != hipSuccess)
        return iq::fail(IQ_ELAUNCH, "iq_check_index_range: copy failed");
    const int grid = (int)std::min<size_t>((count + 255) / 256, 1024);
    hipLaunchKernelGGL(index_range_kernel, dim3(grid), dim3(256), 0, st, idx, count, lo, hi, scratch);
    int rc = iq::check_launch("index_range_kernel");
    if (rc) return rc;
    if (hipMemcpyAsync(&bad, scratch, sizeof(bad), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return iq::fail(IQ_ELAUNCH, "iq_check_index_range: read-back failed");
    if (bad != 0xffffffffu) return iq::fail(IQ_EINVAL, "index at position %u is outside [%d, %d)", bad, lo, hi);
    return IQ_OK;
}
