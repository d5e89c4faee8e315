// Coalition sampling on the device (SURVEY.md §8 a4 / K1-K2 inputs; north star: "coalition sampling").
//
// * mt_permutations_kernel: the reference draws its region permutations from NumPy's GLOBAL legacy generator
//   (final_shapley_value.py:59-72: np.random.permutation per sample after tools/final_util.py:113-120 seeded it).  That
//   stream is MT19937 + Fisher-Yates from the back with masked rejection sampling (numpy/random: RandomState.shuffle ->
//   _shuffle_raw -> random_interval).  The kernel continues it ON THE DEVICE from a given generator state (624 key words +
//   position) and hands the advanced state back, so the permutations - and everything the host draws afterwards - are
//   bit-identical to the reference's for the same seed.  The stream is sequential by construction (a rejection shifts
//   every later draw), so ONE workgroup runs it: 256 lanes regenerate and temper 624 words at a time (3 barriers per
//   twist), wave 0 consumes them with scalar control flow - the candidate words of 64 draws sit in one VGPR and are
//   read with v_readlane, the permutation under construction is one VGPR across the lanes and is swapped with
//   two v_readlane and two compare-selects.
// * prefix_keep_kernel / context_keep_kernel: permutations -> the R+1 prefix coalitions of each
//   (tools/final_common.py:56-60), (pair, context) -> the 4 coalitions of each context
//   (final_point_binary_interaction_logits.py:45-52), as uint64 region bit masks, the form every coalition entry point
//   of this library takes.
#include "iq_common.h"

namespace {

constexpr int kMtN = 624, kMtM = 397;

__device__ inline uint32_t mt_mix(uint32_t hi, uint32_t lo, uint32_t far) {
    const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__device__ inline uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

__global__ __launch_bounds__(256) void mt_permutations_kernel(uint32_t* __restrict__ state, int32_t* __restrict__ orders, int S, int R) {
    __shared__ uint32_t key[2][kMtN];
    __shared__ uint32_t word[kMtN + 64];
    __shared__ int ctl[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int cur = 0;
    for (int k = tid; k < kMtN; k += 256) key[0][k] = state[k];
    int pos = (int)state[kMtN];
    // consumer state (meaningful in wave 0 only; all of it wave-uniform except arr)
    int s = 0, i = R - 1, arr = lane;
    __syncthreads();
    while (true) {
        if (pos >= kMtN) {  // regenerate the 624 words: three dependent thirds, each element-parallel
            const uint32_t* o = key[cur];
            uint32_t* n = key[cur ^ 1];
            for (int k = tid; k < kMtN - kMtM; k += 256) n[k] = mt_mix(o[k], o[k + 1], o[k + kMtM]);
            __syncthreads();
            for (int k = kMtN - kMtM + tid; k < 2 * (kMtN - kMtM); k += 256) n[k] = mt_mix(o[k], o[k + 1], n[k - (kMtN - kMtM)]);
            __syncthreads();
            for (int k = 2 * (kMtN - kMtM) + tid; k < kMtN; k += 256)
                n[k] = mt_mix(o[k], k + 1 < kMtN ? o[k + 1] : n[0], n[k - (kMtN - kMtM)]);
            __syncthreads();
            cur ^= 1;
            pos = 0;
        }
        for (int k = tid; k < kMtN + 64; k += 256) word[k] = k < kMtN ? mt_temper(key[cur][k]) : 0u;
        __syncthreads();
        if (wave == 0) {
            int p = pos;
            while (p < kMtN && s < S) {
                const uint32_t w = word[p + lane];
                const int nvalid = min(64, kMtN - p);
                int k = 0;
                for (; k < nvalid && s < S; ++k) {
                    const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)w, k);
                    const uint32_t mask = (2u << (31 - __clz(i))) - 1u;  // smallest 2^b - 1 >= i  (i >= 1)
                    const int v = (int)(x & mask);
                    if (v <= i) {  // accepted: swap positions i and v, move on
                        const int a_i = __builtin_amdgcn_readlane(arr, i), a_v = __builtin_amdgcn_readlane(arr, v);
                        arr = lane == i ? a_v : lane == v ? a_i : arr;
                        if (--i == 0) {
                            if (lane < R) orders[(size_t)s * R + lane] = arr;
                            arr = lane;
                            i = R - 1;
                            ++s;
                        }
                    }
                }
                p += k;
            }
            if (lane == 0) { ctl[0] = p; ctl[1] = s >= S; }
        }
        __syncthreads();
        pos = ctl[0];
        if (ctl[1]) break;
    }
    for (int k = tid; k < kMtN; k += 256) state[k] = key[cur][k];
    if (tid == 0) state[kMtN] = (uint32_t)pos;
}

// R == 1: a permutation of one region draws nothing
__global__ void zero_orders_kernel(int32_t* __restrict__ orders, int n) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) orders[t] = 0;
}

// one lane per permutation: running OR over its entries; R + 1 coalesced-enough 8-byte stores per lane
__global__ __launch_bounds__(256) void prefix_keep_kernel(const int32_t* __restrict__ orders, uint64_t* __restrict__ keep, int S, int R) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    uint64_t m = 0;
    uint64_t* out = keep + (size_t)s * (R + 1);
    out[0] = 0;
    for (int j = 0; j < R; ++j) {
        const int r = orders[(size_t)s * R + j];
        if ((unsigned)r < 64u) m |= 1ull << r;  // an out-of-range entry is ignored (iq_check_index_range names it)
        out[j + 1] = m;
    }
}

// one lane per (pair, context)
__global__ __launch_bounds__(256) void context_keep_kernel(const int32_t* __restrict__ pairs, const int32_t* __restrict__ ctx,
                                                           uint64_t* __restrict__ keep, int P, int C, int m) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)P * C) return;
    const int p = (int)(t / C);
    uint64_t sset = 0;
    for (int j = 0; j < m; ++j) {
        const int r = ctx[t * m + j];
        if ((unsigned)r < 64u) sset |= 1ull << r;
    }
    const int i = pairs[2 * p], j = pairs[2 * p + 1];
    const uint64_t bi = (unsigned)i < 64u ? 1ull << i : 0ull, bj = (unsigned)j < 64u ? 1ull << j : 0ull;
    uint64_t* out = keep + 4 * t;  // rows 4k: S+{i,j}, 4k+1: S+{i}, 4k+2: S+{j}, 4k+3: S
    out[0] = sset | bi | bj;
    out[1] = sset | bi;
    out[2] = sset | bj;
    out[3] = sset;
}

}  // namespace

extern "C" int iq_sample_permutations(uint32_t* mt_state, int32_t* orders, int S, int R, iq_stream_t stream) {
    IQ_REQUIRE(S >= 0 && R >= 1 && R <= IQ_MAX_REGIONS, "iq_sample_permutations: S=%d R=%d", S, R);
    if (S == 0) return IQ_OK;
    IQ_REQUIRE(mt_state && orders, "iq_sample_permutations: null pointer");
    hipStream_t st = iq::as_stream(stream);
    if (R == 1) {
        hipLaunchKernelGGL(zero_orders_kernel, dim3((S + 255) / 256), dim3(256), 0, st, orders, S);
        return iq::check_launch("zero_orders_kernel");
    }
    hipLaunchKernelGGL(mt_permutations_kernel, dim3(1), dim3(256), 0, st, mt_state, orders, S, R);
    return iq::check_launch("mt_permutations_kernel");
}

extern "C" int iq_prefix_keep_masks(const int32_t* orders, uint64_t* keep, int S, int R, iq_stream_t stream) {
    IQ_REQUIRE(S >= 0 && R >= 1 && R <= IQ_MAX_REGIONS, "iq_prefix_keep_masks: S=%d R=%d", S, R);
    if (S == 0) return IQ_OK;
    IQ_REQUIRE(orders && keep, "iq_prefix_keep_masks: null pointer");
    hipLaunchKernelGGL(prefix_keep_kernel, dim3((S + 255) / 256), dim3(256), 0, iq::as_stream(stream), orders, keep, S, R);
    return iq::check_launch("prefix_keep_kernel");
}

extern "C" int iq_context_keep_masks(const int32_t* pairs, const int32_t* contexts, uint64_t* keep, int P, int C, int m,
                                     iq_stream_t stream) {
    IQ_REQUIRE(P >= 0 && C >= 0 && m >= 0 && m <= IQ_MAX_REGIONS, "iq_context_keep_masks: P=%d C=%d m=%d", P, C, m);
    if ((size_t)P * C == 0) return IQ_OK;
    IQ_REQUIRE(pairs && keep && (contexts || m == 0), "iq_context_keep_masks: null pointer");
    const size_t n = (size_t)P * C;
    hipLaunchKernelGGL(context_keep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, iq::as_stream(stream), pairs, contexts, keep,
                       P, C, m);
    return iq::check_launch("context_keep_kernel");
}
