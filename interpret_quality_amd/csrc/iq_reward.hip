// Reward (K3), per-region Shapley accumulation (K4) and interaction reduction (K5).
// All tiny next to the forward; written for bit-stability (fixed summation order), not speed.
#include "iq_common.h"

namespace {

constexpr int kMaxClasses = 64;

// tools/final_common.py:19-24.  torch.logsumexp(x) = log(sum(exp(x - max))) + max.
__global__ void reward_kernel(const float* __restrict__ logits, int label, int modified,
                              float* __restrict__ v, int B, int C) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* z = logits + (size_t)b * C;
    float zy = z[label];
    float m = -INFINITY;
    for (int c = 0; c < C; ++c)
        if (!modified || c != label) m = fmaxf(m, z[c]);
    const float ms = (fabsf(m) == INFINITY) ? 0.f : m;
    float s = 0.f;
    for (int c = 0; c < C; ++c)
        if (!modified || c != label) s += expf(z[c] - ms);
    // modified: z_y - logsumexp(others);  normal: log_softmax = (z_y - max) - log(sum)
    v[b] = modified ? zy - (logf(s) + ms) : (zy - ms) - logf(s);
}

// dv = v[i+1] - v[i] in float32, widened to float64 and scattered by the permutation
// (tools/final_common.py:94-96; the float64 target is np.zeros((R,))).
__global__ void shapley_scatter_kernel(const float* __restrict__ v, const int32_t* __restrict__ orders,
                                       double* __restrict__ sv_rows, int R, int S) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S * R) return;
    const int o = t / R, j = t - o * R;
    const float* vo = v + (size_t)o * (R + 1);
    const float dv = vo[j + 1] - vo[j];
    const int r = orders[t];
    if ((unsigned)r < (unsigned)R) sv_rows[(size_t)o * R + r] = (double)dv;  // entries outside [0,R): iq_check_index_range
}

// Sum over permutations in permutation order, one lane per region: the same sequence of float64
// adds the reference's host loop performs, hence bit-identical for identical v.
__global__ void shapley_sum_kernel(const double* __restrict__ sv_rows, double* __restrict__ phi_sum,
                                   const int32_t* __restrict__ snap_counts, int n_snap,
                                   double* __restrict__ snaps, int R, int S) {
    const int r = threadIdx.x;
    if (r >= R) return;
    double acc = 0.0;
    int k = 0;
    // the adds stay strictly in permutation order; the loads run 8 rows ahead of them (a plain loop paid one memory
    // latency per permutation: 260 us for 1000 permutations)
    constexpr int U = 8;
    int o = 0;
    for (; o + U <= S; o += U) {
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = sv_rows[(size_t)(o + u) * R + r];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc += v[u];
            while (k < n_snap && snap_counts[k] == o + u + 1) { snaps[(size_t)k * R + r] = acc; ++k; }
        }
    }
    for (; o < S; ++o) {
        acc += sv_rows[(size_t)o * R + r];
        while (k < n_snap && snap_counts[k] == o + 1) { snaps[(size_t)k * R + r] = acc; ++k; }
    }
    phi_sum[r] = acc;
}

__global__ void interaction_reduce_kernel(const float* __restrict__ v, float* __restrict__ out, int n) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float4 q = reinterpret_cast<const float4*>(v)[k];
    // v[4k] + v[4k+3] - v[4k+1] - v[4k+2], left to right (final_cal_interactions.py:33)
    out[k] = __fsub_rn(__fsub_rn(__fadd_rn(q.x, q.w), q.y), q.z);
}

}  // namespace

extern "C" int iq_reward(const float* logits, int label, int modified, float* v, int B, int C,
                         iq_stream_t stream) {
    IQ_REQUIRE(B >= 0 && C >= 2 && C <= kMaxClasses, "iq_reward: B=%d C=%d", B, C);
    IQ_REQUIRE(label >= 0 && label < C, "iq_reward: label %d not in [0,%d)", label, C);
    if (B == 0) return IQ_OK;
    IQ_REQUIRE(logits && v, "iq_reward: null pointer");
    hipLaunchKernelGGL(reward_kernel, dim3((B + 255) / 256), dim3(256), 0, iq::as_stream(stream),
                       logits, label, modified, v, B, C);
    return iq::check_launch("reward_kernel");
}

extern "C" int iq_shapley_accum(const float* v, const int32_t* orders, double* sv_rows, double* phi_sum,
                                const int32_t* snap_counts, int n_snap, double* snaps, int R, int S,
                                iq_stream_t stream) {
    IQ_REQUIRE(R >= 1 && R <= IQ_MAX_REGIONS && S >= 0, "iq_shapley_accum: R=%d S=%d", R, S);
    IQ_REQUIRE(phi_sum && sv_rows, "iq_shapley_accum: phi_sum and sv_rows are required");
    IQ_REQUIRE(n_snap == 0 || (snap_counts && snaps), "iq_shapley_accum: snapshots need counts and output");
    IQ_REQUIRE(S == 0 || (v && orders), "iq_shapley_accum: null input");
    hipStream_t st = iq::as_stream(stream);
    if (S > 0) {
        hipLaunchKernelGGL(shapley_scatter_kernel, dim3((S * R + 255) / 256), dim3(256), 0, st,
                           v, orders, sv_rows, R, S);
        int rc = iq::check_launch("shapley_scatter_kernel");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(shapley_sum_kernel, dim3(1), dim3(64), 0, st, sv_rows, phi_sum, snap_counts,
                       n_snap, snaps, R, S);
    return iq::check_launch("shapley_sum_kernel");
}

extern "C" int iq_interaction_reduce(const float* v, float* out, int n, iq_stream_t stream) {
    IQ_REQUIRE(n >= 0, "iq_interaction_reduce: n=%d", n);
    if (n == 0) return IQ_OK;
    IQ_REQUIRE(v && out, "iq_interaction_reduce: null pointer");
    hipLaunchKernelGGL(interaction_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0,
                       iq::as_stream(stream), v, out, n);
    return iq::check_launch("interaction_reduce_kernel");
}
