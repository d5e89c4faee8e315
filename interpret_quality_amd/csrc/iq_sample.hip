// Coalition sampling on the device (SURVEY.md §8 a4 / K1-K2 inputs; north star: "coalition sampling").
//
// * mt_permutations_kernel: the reference draws its region permutations from NumPy's GLOBAL legacy generator
//   (final_shapley_value.py:59-72: np.random.permutation per sample after tools/final_util.py:113-120 seeded it).  That
//   stream is MT19937 + Fisher-Yates from the back with masked rejection sampling (numpy/random: RandomState.shuffle ->
//   _shuffle_raw -> random_interval).  The kernel continues it ON THE DEVICE from a given generator state (624 key words +
//   position) and hands the advanced state back, so the permutations - and everything the host draws afterwards - are
//   bit-identical to the reference's for the same seed.
//   The stream looks sequential - a rejected draw shifts every later one - but where a permutation STARTS in the word stream
//   is all that couples two permutations, and how many words a permutation consumes is a function of its start offset alone.
//   One workgroup of 1024 lanes therefore works in batches of 8 x 624 words: (1) regenerate and temper the words (three
//   element-parallel thirds per 624-word block); (2) every lane simulates the draws of a permutation starting at EVERY word
//   offset of the batch (no swaps, just the rejection loop) -> next[o] = offset behind it; (3) one lane follows
//   next[] from the current position: the start offsets of the real permutations; (4) one lane per real permutation replays
//   its draws with the swaps and writes the row.  1000 permutations of 32 regions take ~0.2 ms instead of the 4 ms of a
//   draw-by-draw scalar loop (which is what the first version of this kernel was).
// * prefix_keep_kernel / context_keep_kernel: permutations -> the R+1 prefix coalitions of each
//   (tools/final_common.py:56-60), (pair, context) -> the 4 coalitions of each context
//   (final_point_binary_interaction_logits.py:45-52), as uint64 region bit masks, the form every coalition entry point
//   of this library takes.
#include "iq_common.h"

namespace {

constexpr int kMtN = 624, kMtM = 397;
constexpr int kMtBlocks = 8;                      // 624-word blocks per batch
constexpr int kMtWords = kMtBlocks * kMtN;        // 4992 words per batch
constexpr int kMtThreads = 1024;
constexpr int kMtMaxStarts = kMtWords + 1;        // (R = 2 draws at least one word per permutation)
constexpr unsigned kMtInvalid = 0xffffu;
constexpr unsigned kMtPoison = 0x7fffffffu;        // position word of a state no draw may continue from (a failed / refused call)

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

// smallest 2^b - 1 >= i (i >= 1): random_interval's mask
__device__ inline uint32_t interval_mask(int i) { return (2u << (31 - __clz(i))) - 1u; }

__global__ __launch_bounds__(kMtThreads) void mt_permutations_kernel(uint32_t* __restrict__ state, int32_t* __restrict__ orders, int S, int R) {
    __shared__ uint32_t key[kMtBlocks][kMtN];      // block 0: the carried generator state; block k = twist of block k - 1
    __shared__ uint32_t word[kMtWords];            // tempered outputs of the batch
    __shared__ uint16_t nxt[kMtWords + 1];         // offset behind a permutation that starts at offset o (kMtInvalid: runs out of the batch)
    __shared__ uint16_t start[kMtMaxStarts];       // start offsets of the real permutations of this batch
    __shared__ int ctl[3];                         // permutations found in this batch, position behind the last of them, no-progress flag
    const int tid = threadIdx.x;
    for (int k = tid; k < kMtN; k += kMtThreads) key[0][k] = state[k];
    int pos = (int)state[kMtN];                    // 0..624
    int done = 0;                                  // permutations written so far (uniform)
    __syncthreads();
    if ((unsigned)pos > (unsigned)kMtN) {
        // a malformed state (np.random.set_state accepts any position) or one a failed call poisoned: draw nothing, write rows
        // that every range check rejects, keep the state poisoned so that the host sees it (hip_ops.mt_state_to_host raises)
        for (int e = tid; e < S * R; e += kMtThreads) orders[e] = -1;
        if (tid == 0) state[kMtN] = kMtPoison;
        return;
    }
    while (done < S) {
        // (1) blocks 1..7 by twisting (three dependent thirds, each element-parallel); temper all
        for (int blk = 1; blk < kMtBlocks; ++blk) {
            const uint32_t* o = key[blk - 1];
            uint32_t* n = key[blk];
            if (tid < kMtN - kMtM) n[tid] = mt_mix(o[tid], o[tid + 1], o[tid + kMtM]);
            __syncthreads();
            if (tid < kMtN - kMtM) { const int k = kMtN - kMtM + tid; n[k] = mt_mix(o[k], o[k + 1], n[tid]); }
            __syncthreads();
            if (tid < kMtM - (kMtN - kMtM)) {
                const int k = 2 * (kMtN - kMtM) + tid;
                n[k] = mt_mix(o[k], k + 1 < kMtN ? o[k + 1] : n[0], n[k - (kMtN - kMtM)]);
            }
            __syncthreads();
        }
        for (int k = tid; k < kMtWords; k += kMtThreads) word[k] = mt_temper(key[k / kMtN][k % kMtN]);
        __syncthreads();
        // (2) length of a permutation that would start at offset o, for every o.  The words a simulation consumes are
        // consecutive whatever it accepts, so four are read ahead per trip and the four decisions run branch-free: one flat loop
        // (nested accept / reject loops diverge lane by lane and wait on every LDS read: 5x slower)
        for (int o = pos + tid; o <= kMtWords; o += kMtThreads) {
            int i = R - 1, p = o;
            bool fail = false;
            while (i >= 1) {
                uint32_t w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) w[u] = word[min(p + u, kMtWords - 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool active = i >= 1, inb = p < kMtWords;
                    fail = fail || (active && !inb);
                    const bool acc = active && inb && (w[u] & interval_mask(max(i, 1))) <= (uint32_t)i;
                    p += (active && inb) ? 1 : 0;
                    i -= acc ? 1 : 0;
                }
                if (fail) i = 0;
            }
            nxt[o] = fail ? (uint16_t)kMtInvalid : (uint16_t)p;
        }
        __syncthreads();
        // (3) the chain of real starts
        if (tid == 0) {
            int cur = pos, n = 0;
            while (done + n < S) {
                const unsigned nx = nxt[cur];
                if (nx == kMtInvalid) break;
                start[n++] = (uint16_t)cur;
                cur = (int)nx;
            }
            ctl[0] = n;
            ctl[1] = cur;
        }
        __syncthreads();
        const int n = ctl[0], cur = ctl[1];
        // (4) replay with the swaps: one lane per permutation, its row as bytes in nxt[]'s storage (free once the chain is
        // known; n rows of R bytes never exceed it: n <= 4992 / (R - 1) + 1), then one coalesced copy to global memory
        uint8_t* rowbytes = reinterpret_cast<uint8_t*>(nxt);
        for (int s = tid; s < n; s += kMtThreads) {
            uint8_t* row = rowbytes + s * R;
            for (int j = 0; j < R; ++j) row[j] = (uint8_t)j;
            int p = start[s], i = R - 1;
            while (i >= 1) {                         // same flat form; the words of a real permutation all lie inside the batch
                uint32_t w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) w[u] = word[min(p + u, kMtWords - 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (i >= 1) {
                        const uint32_t v = w[u] & interval_mask(i);
                        ++p;
                        if (v <= (uint32_t)i) {
                            const uint8_t a_i = row[i], a_v = row[v];
                            row[i] = a_v;
                            row[v] = a_i;
                            --i;
                        }
                    }
                }
            }
        }
        __syncthreads();
        for (int e = tid; e < n * R; e += kMtThreads) orders[(size_t)done * R + e] = rowbytes[e];
        // carry the state: the block that holds `cur` becomes block 0 (cur == 624 k stays at the END of block k - 1, as NumPy
        // leaves its position at 624 until the next draw)
        const int kb = cur == 0 ? 0 : min((cur - 1) / kMtN, kMtBlocks - 1);
        __syncthreads();
        if (kb > 0) {
            uint32_t carry = tid < kMtN ? key[kb][tid] : 0u;
            __syncthreads();
            if (tid < kMtN) key[0][tid] = carry;
        }
        pos = cur - kb * kMtN;
        done += n;
        __syncthreads();
        if (n == 0 && kb == 0) break;   // no progress is only possible if one permutation needs more than 4368 words (p < 2^-4000)
    }
    for (int k = tid; k < kMtN; k += kMtThreads) state[k] = key[0][k];
    if (done < S) {                     // gave up: the rows not drawn are marked, and so is the state (see the entry check)
        for (int e = done * R + tid; e < S * R; e += kMtThreads) orders[e] = -1;
        pos = (int)kMtPoison;
    }
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
    hipLaunchKernelGGL(mt_permutations_kernel, dim3(1), dim3(kMtThreads), 0, st, mt_state, orders, S, R);
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
