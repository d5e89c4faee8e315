// Stage 5 of exp_shapley.sh: the linearity / planarity / scattering enumeration of every region
// (reference final_smoothness_center_enum_all.py:23-242, 245-356).  The reference walks one region at a time with
// autograd on a few dozen points and three host syncs per gradient step; regions never touch each other's points,
// so here ONE launch runs the whole enumeration - all epochs, all gradient steps, all stop conditions - with one
// wavefront per region, the region's points resident in LDS and the closed-form gradient of the variance ratios.
// Built with -ffp-contract=off (build.py): the trajectory is compared against the reference step by step.
#include "iq_common.h"

namespace {

constexpr int kMaxRegionPoints = 1024;
constexpr int kWave = 64;

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
__device__ inline int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// Symmetric 3x3 eigen-decomposition, cyclic Jacobi in fp64 (replaces torch.symeig,
// final_smoothness_center_enum_all.py:41).  Eigenvalues ascending in w, eigenvectors in the columns of V.
__device__ void eigh3(const double a_in[3][3], double w[3], double V[3][3]) {
    double a[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { a[i][j] = a_in[i][j]; V[i][j] = (i == j) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 32; ++sweep) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        const double diag = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
        if (off <= 1e-300 || off <= 1e-17 * diag) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {  // A <- A J
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {  // A <- J^T A
                    const double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    // ascending order of the diagonal (insertion sort, stable); no array is indexed by a run-time value, so none lives in scratch
    const double d0 = a[0][0], d1 = a[1][1], d2 = a[2][2];
    auto diag = [&](int i) { return i == 0 ? d0 : (i == 1 ? d1 : d2); };
    int i0 = 0, i1 = 1, i2 = 2;
    if (diag(i1) < diag(i0)) { const int t = i1; i1 = i0; i0 = t; }
    if (diag(i2) < diag(i1)) {
        const int t = i2; i2 = i1; i1 = t;
        if (diag(i1) < diag(i0)) { const int u = i1; i1 = i0; i0 = u; }
    }
    auto col = [&](int r, int i) { return i == 0 ? V[r][0] : (i == 1 ? V[r][1] : V[r][2]); };
    w[0] = diag(i0); w[1] = diag(i1); w[2] = diag(i2);
    double Vs[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r) { Vs[r][0] = col(r, i0); Vs[r][1] = col(r, i1); Vs[r][2] = col(r, i2); }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) V[r][k] = Vs[r][k];
}

// np.argsort of three values (final_smoothness_center_enum_all.py:85-99): stable, NaN last.
// (no array is indexed by a run-time value: pick3 keeps everything in registers)
__device__ inline float pick3(const float v[3], int i) { return i == 0 ? v[0] : (i == 1 ? v[1] : v[2]); }
__device__ inline bool pick3b(const bool v[3], int i) { return i == 0 ? v[0] : (i == 1 ? v[1] : v[2]); }
__device__ inline bool less_nan_last(float a, float b) { return (a < b) || (b != b && a == a); }
__device__ inline void argsort3(const float v[3], int idx[3]) {
    int i0 = 0, i1 = 1, i2 = 2;                      // insertion sort, as before: element 1, then element 2
    if (less_nan_last(pick3(v, i1), pick3(v, i0))) { const int t = i1; i1 = i0; i0 = t; }
    if (less_nan_last(pick3(v, i2), pick3(v, i1))) {
        const int t = i2; i2 = i1; i1 = t;
        if (less_nan_last(pick3(v, i1), pick3(v, i0))) { const int u = i1; i1 = i0; i0 = u; }
    }
    idx[0] = i0; idx[1] = i1; idx[2] = i2;
}

struct Vars { float var[3]; float mean[3]; };

// torch.var (unbiased) of the three projections X.o_k  (cal_variance, :48-63)
__device__ inline Vars variances(const float* pts, int S, int lane, const float o[3][3]) {
    Vars r;
    float s[3] = {0.f, 0.f, 0.f};
    for (int i = lane; i < S; i += kWave) {
        const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
#pragma unroll
        for (int k = 0; k < 3; ++k) s[k] += x * o[k][0] + y * o[k][1] + z * o[k][2];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) r.mean[k] = wave_sum(s[k]) / (float)S;
    float q[3] = {0.f, 0.f, 0.f};
    for (int i = lane; i < S; i += kWave) {
        const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float d = (x * o[k][0] + y * o[k][1] + z * o[k][2]) - r.mean[k];
            q[k] += d * d;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) r.var[k] = wave_sum(q[k]) / (float)(S - 1);
    return r;
}

__device__ inline float smoothness_of(int mode, float smin, float smid, float smax) {
    if (mode == 0) return (smax - smid) / smax;   // linearity
    if (mode == 1) return (smid - smin) / smax;   // planarity
    return smin / smax;                           // scattering
}

__global__ __launch_bounds__(kWave) void smooth_enum_kernel(
    const float* __restrict__ cloud, const float* __restrict__ origin, const int32_t* __restrict__ region_id, int N, int R,
    int mode, int objective, iq_smoothness_params prm, float* __restrict__ data_out, float* __restrict__ smooth_out,
    float* __restrict__ var_out, float* __restrict__ orig_out, int32_t* __restrict__ stop_epoch) {
    // One wave per region, everything in LDS and registers (no scratch: the argsort permutation is applied with selects).
    // This file is built WITHOUT packed float32 instructions - see NO_PACKED_FP32 in build.py: up to epochs x max_iteration
    // gradient steps amplify a one-ulp difference, and beside one particular neighbour on a shared GPU packed float32 did differ.
    __shared__ float cur[kMaxRegionPoints * 3];
    __shared__ float org[kMaxRegionPoints * 3];
    __shared__ int32_t pidx[kMaxRegionPoints];
    __shared__ float o_sh[9];
    const int r = blockIdx.x;
    const int lane = threadIdx.x;
    const int E = prm.epochs;

    // the region's points in index order (data[:, region_id == r, :])
    int S = 0;
    for (int base = 0; base < N; base += kWave) {
        const int i = base + lane;
        const bool mine = i < N && region_id[i] == r;
        const unsigned long long m = __ballot(mine);
        if (mine) {
            const int pos = S + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < kMaxRegionPoints) pidx[pos] = i;
        }
        S += __popcll(m);
    }
    if (S > kMaxRegionPoints) S = kMaxRegionPoints;  // rejected on the host (N <= 1024 regions cannot exceed it)
    __syncthreads();
    for (int i = lane; i < S; i += kWave) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            cur[3 * i + c] = cloud[(size_t)pidx[i] * 3 + c];
            org[3 * i + c] = origin[(size_t)pidx[i] * 3 + c];
        }
    }
    __syncthreads();

    if (S < 2) {  // the reference cannot process a one-point region at all; leave it untouched
        for (int e = 0; e < E; ++e) {
            for (int i = lane; i < 3 * S; i += kWave) data_out[((size_t)e * N + pidx[i / 3]) * 3 + i % 3] = cur[i];
            if (lane == 0) {
                smooth_out[(size_t)e * R + r] = NAN;
                for (int k = 0; k < 3; ++k) var_out[((size_t)e * R + r) * 3 + k] = NAN;
            }
        }
        if (lane == 0) {
            for (int k = 0; k < 4; ++k) orig_out[r * 4 + k] = NAN;
            stop_epoch[r] = -1;
        }
        return;
    }

    // principal orientations (cal_principal_orientation, :23-45): covariance in fp32 as the reference builds it
    {
        float m[3] = {0.f, 0.f, 0.f};
        for (int i = lane; i < S; i += kWave)
            for (int c = 0; c < 3; ++c) m[c] += org[3 * i + c];
        for (int c = 0; c < 3; ++c) m[c] = wave_sum(m[c]) / (float)S;
        float cv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int i = lane; i < S; i += kWave) {
            const float dx = org[3 * i] - m[0], dy = org[3 * i + 1] - m[1], dz = org[3 * i + 2] - m[2];
            cv[0] += dx * dx; cv[1] += dx * dy; cv[2] += dx * dz; cv[3] += dy * dy; cv[4] += dy * dz; cv[5] += dz * dz;
        }
        for (int k = 0; k < 6; ++k) cv[k] = wave_sum(cv[k]) / (float)(S - 1);
        if (lane == 0) {
            const double A[3][3] = {{cv[0], cv[1], cv[2]}, {cv[1], cv[3], cv[4]}, {cv[2], cv[4], cv[5]}};
            double w[3], V[3][3];
            eigh3(A, w, V);
            for (int k = 0; k < 3; ++k)       // o1 <- largest eigenvalue (column 2), o3 <- smallest (column 0)
                for (int c = 0; c < 3; ++c) o_sh[3 * k + c] = (float)V[c][2 - k];
        }
    }
    __syncthreads();
    float o[3][3];
    for (int k = 0; k < 3; ++k)
        for (int c = 0; c < 3; ++c) o[k][c] = o_sh[3 * k + c];

    const Vars v0 = variances(org, S, lane, o);
    float ub[3], lb[3];
    const float vth = (float)prm.var_threshold;
    for (int k = 0; k < 3; ++k) { ub[k] = v0.var[k] + vth; lb[k] = v0.var[k] - vth; }
    int si[3];
    argsort3(v0.var, si);
    float smooth = smoothness_of(mode, pick3(v0.var, si[0]), pick3(v0.var, si[1]), pick3(v0.var, si[2]));
    if (lane == 0) {
        for (int k = 0; k < 3; ++k) orig_out[r * 4 + k] = v0.var[k];
        orig_out[r * 4 + 3] = smooth;
    }

    const float step = (float)prm.step, dth = (float)prm.dist_threshold;
    bool active = true;
    int stop = E;
    float lastvar[3] = {v0.var[0], v0.var[1], v0.var[2]};
    for (int e = 0; e < E; ++e) {
        if (active) {  // update_region (:183-242)
            const double target = (double)smooth + (objective > 0 ? prm.enum_step : -prm.enum_step);
            float sm = smooth;
            int iteration = 0;
            while (objective > 0 ? ((double)sm < target) : ((double)sm > target)) {
                const Vars v = variances(cur, S, lane, o);
                bool live[3];
                for (int k = 0; k < 3; ++k) {
                    live[k] = !(v.var[k] > ub[k] || v.var[k] < lb[k]);  // apply_var_bound (:66-74)
                    lastvar[k] = v.var[k];
                }
                argsort3(v.var, si);
                const float smin = pick3(v.var, si[0]), smid = pick3(v.var, si[1]), smax = pick3(v.var, si[2]);
                const bool lmin = pick3b(live, si[0]), lmid = pick3b(live, si[1]), lmax = pick3b(live, si[2]);
                sm = smoothness_of(mode, smin, smid, smax);
                // d smoothness / d var, zero through a detached variance
                float cmin, cmid, cmax;
                bool has_grad;
                if (mode == 0) {
                    cmin = 0.f; cmid = -1.f / smax; cmax = 1.f / smax - (smax - smid) / (smax * smax);
                    has_grad = lmax || lmid;
                } else if (mode == 1) {
                    cmin = -1.f / smax; cmid = 1.f / smax; cmax = -(smid - smin) / (smax * smax);
                    has_grad = lmax || lmid || lmin;
                } else {
                    cmin = 1.f / smax; cmid = 0.f; cmax = -smin / (smax * smax);
                    has_grad = lmax || lmin;
                }
                const bool grad_none = !has_grad;
                if (has_grad) {  // gradient_descent (:121-138)
                    float ck[3];
                    const float kmin = lmin ? cmin : 0.f, kmid = lmid ? cmid : 0.f, kmax = lmax ? cmax : 0.f;
#pragma unroll
                    for (int k = 0; k < 3; ++k) ck[k] = (k == si[0]) ? kmin : ((k == si[1]) ? kmid : kmax);
                    const float two_over = 2.f / (float)(S - 1);
                    float n2 = 0.f;
                    for (int i = lane; i < S; i += kWave) {
                        const float x = cur[3 * i], y = cur[3 * i + 1], z = cur[3 * i + 2];
                        float g[3] = {0.f, 0.f, 0.f};
                        for (int k = 0; k < 3; ++k) {
                            const float gp = ck[k] * (two_over * ((x * o[k][0] + y * o[k][1] + z * o[k][2]) - v.mean[k]));
                            for (int c = 0; c < 3; ++c) g[c] += gp * o[k][c];
                        }
                        n2 += g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
                    }
                    const float norm = sqrtf(wave_sum(n2));
                    for (int i = lane; i < S; i += kWave) {
                        const float x = cur[3 * i], y = cur[3 * i + 1], z = cur[3 * i + 2];
                        float g[3] = {0.f, 0.f, 0.f};
                        for (int k = 0; k < 3; ++k) {
                            const float gp = ck[k] * (two_over * ((x * o[k][0] + y * o[k][1] + z * o[k][2]) - v.mean[k]));
                            for (int c = 0; c < 3; ++c) g[c] += gp * o[k][c];
                        }
                        for (int c = 0; c < 3; ++c) {
                            const float delta = (norm != 0.f) ? (step * g[c]) / norm : 1e-8f;
                            cur[3 * i + c] = objective > 0 ? cur[3 * i + c] + delta : cur[3 * i + c] - delta;
                        }
                    }
                }
                // apply_distance_bound (:102-118).  The reference's write-back `data_region_i[i].data = ...` assigns to a
                // temporary view and never reaches the region, so points are only COUNTED there; projecting them back
                // onto the sphere (what the comment at :110 intends) is the opt-in prm.project_to_bound.
                int count = 0;
                for (int i = lane; i < S; i += kWave) {
                    const float dx = cur[3 * i] - org[3 * i], dy = cur[3 * i + 1] - org[3 * i + 1], dz = cur[3 * i + 2] - org[3 * i + 2];
                    const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
                    if (dist > dth) {
                        ++count;
                        if (prm.project_to_bound) {
                            cur[3 * i] = org[3 * i] + (dth * dx) / dist;
                            cur[3 * i + 1] = org[3 * i + 1] + (dth * dy) / dist;
                            cur[3 * i + 2] = org[3 * i + 2] + (dth * dz) / dist;
                        }
                    }
                }
                count = wave_sum_i(count);
                ++iteration;
                // check_stop_condition (:163-180)
                if ((double)count / (double)S > prm.stop_ratio || grad_none || iteration > prm.max_iteration) {
                    active = false;
                    stop = e;
                    break;
                }
            }
            smooth = sm;
        }
        for (int i = lane; i < 3 * S; i += kWave) data_out[((size_t)e * N + pidx[i / 3]) * 3 + i % 3] = cur[i];
        if (lane == 0) {
            smooth_out[(size_t)e * R + r] = smooth;
            for (int k = 0; k < 3; ++k) var_out[((size_t)e * R + r) * 3 + k] = lastvar[k];
        }
    }
    if (lane == 0) stop_epoch[r] = stop;
}

}  // namespace

extern "C" int iq_smoothness_enum(const float* cloud, const float* origin, const int32_t* region_id, int N, int R, int mode, int objective,
                                  const iq_smoothness_params* prm, float* data_out, float* smooth_out, float* var_out,
                                  float* orig_out, int32_t* stop_epoch, iq_stream_t stream) {
    IQ_REQUIRE(cloud && region_id && prm && data_out && smooth_out && var_out && orig_out && stop_epoch,
               "iq_smoothness_enum: null pointer");
    IQ_REQUIRE(N >= 1 && N <= kMaxRegionPoints, "iq_smoothness_enum: N=%d not in [1,%d]", N, kMaxRegionPoints);
    IQ_REQUIRE(R >= 1 && R <= IQ_MAX_REGIONS, "iq_smoothness_enum: R=%d not in [1,%d]", R, IQ_MAX_REGIONS);
    IQ_REQUIRE(mode >= 0 && mode <= 2, "iq_smoothness_enum: mode %d (0 linearity, 1 planarity, 2 scattering)", mode);
    IQ_REQUIRE(objective == 1 || objective == -1, "iq_smoothness_enum: objective %d (+1 inc, -1 dec)", objective);
    IQ_REQUIRE(prm->epochs >= 1 && prm->epochs <= 4096 && prm->max_iteration >= 0,
               "iq_smoothness_enum: epochs=%d max_iteration=%d", prm->epochs, prm->max_iteration);
    hipLaunchKernelGGL(smooth_enum_kernel, dim3(R), dim3(kWave), 0, iq::as_stream(stream), cloud,
                       origin ? origin : cloud, region_id, N, R, mode, objective, *prm, data_out, smooth_out, var_out, orig_out, stop_epoch);
    return iq::check_launch("smooth_enum_kernel");
}
