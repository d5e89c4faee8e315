// PointConv (density) forward (models/pointconv.py:103-424) on materialised (masked) clouds.
//
// Per set abstraction:  Gaussian KDE density -> FPS -> K nearest (expanded-form distance, index set) ->
// shared MLP on [x_p - c ; f_p] -> x DensityNet(inverse density / group max) -> contraction with
// WeightNet(x_p - c) over the K members: out[c][w] = sum_k h[k][c] s_k wt[k][w] -> Linear(16 C -> C)+BN+ReLU.
// Members are SUMMED here, so (unlike the max-pooled families) duplicates count with their multiplicity
// and nothing is skipped.  Layer 1 of sa2 / sa3 uses the exact linear split
// W.[x_p - c ; f_p] = W_x (x_p - c) + (W_f f_p + b); the per-point part is one batched GEMM.
// Grouped kernel: 64-row chunks -> LDS act1 -> MFMA C1->C2 -> LDS act2 -> MFMA C2->C3 -> the contraction runs
// on the accumulator tiles in registers (16 rows x 16 weights per lane and tile), so the (C3 x K) member
// features never leave the CU.
#include "iq_common.h"
#include "iq_bf3.h"
#include "iq_mfma.h"
#include "iq_profile.h"
#include "iq_srclist.h"
#include "iq_topk.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMC = 64;

__global__ void pc_gather_xyz_kernel(const float* __restrict__ xyz, int ldx, const int32_t* __restrict__ idx,
                                     float* __restrict__ out, int N, int S, int total) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int b = t / S;
    const float* src = xyz + ((size_t)b * N + idx[t]) * ldx;
    out[t * 3] = src[0]; out[t * 3 + 1] = src[1]; out[t * 3 + 2] = src[2];
}

__global__ void pc_compact_xyz_kernel(const float* __restrict__ xyz, int ldx, float* __restrict__ out, int total) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const float* src = xyz + (size_t)t * ldx;
    out[t * 3] = src[0]; out[t * 3 + 1] = src[1]; out[t * 3 + 2] = src[2];
}

// ---- inverse Gaussian KDE density (models/pointconv.py:199-209, :352) --------------------------------
// inv[i] = 1 / mean_j( exp(-d_ij / (2 bw^2)) / (2.5 bw) ),  d = -2 x_i.x_j + |x_i|^2 + |x_j|^2
__global__ __launch_bounds__(kThreads) void pc_density_kernel(const float* __restrict__ xyz, float bw,
                                                              float* __restrict__ inv_density, int N) {
    extern __shared__ float pts[];  // N x 4
    const int b = blockIdx.y;
    const float* src = xyz + (size_t)b * N * 3;
    for (int p = threadIdx.x; p < N; p += kThreads) {
        const float x = src[p * 3], y = src[p * 3 + 1], z = src[p * 3 + 2];
        pts[p * 4] = x; pts[p * 4 + 1] = y; pts[p * 4 + 2] = z; pts[p * 4 + 3] = (x * x + y * y) + z * z;
    }
    __syncthreads();
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= N) return;
    const f32x4 q = *reinterpret_cast<const f32x4*>(pts + i * 4);
    // exp(-d / c0) / c1 summed as exp2(d * k) with the division by c1 once at the end: the N^2 loop is 3 fma + 2 add + 1 mul +
    // v_exp_f32 per pair instead of a full-precision expf and two divisions (the density only feeds DensityNet, a smooth
    // function; agreement with the reference stays at the 1e-6 level)
    const float c0 = 2.0f * bw * bw, c1 = 2.5f * bw;
    const float k2 = -1.4426950408889634f / c0;
    float s0 = 0.f, s1 = 0.f;
    int j = 0;
    for (; j + 1 < N; j += 2) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(pts + j * 4);
        const f32x4 u = *reinterpret_cast<const f32x4*>(pts + j * 4 + 4);
        const float d0 = (-2.f * fmaf(q[2], v[2], fmaf(q[1], v[1], q[0] * v[0])) + q[3]) + v[3];
        const float d1 = (-2.f * fmaf(q[2], u[2], fmaf(q[1], u[1], q[0] * u[0])) + q[3]) + u[3];
        s0 += __builtin_amdgcn_exp2f(d0 * k2);
        s1 += __builtin_amdgcn_exp2f(d1 * k2);
    }
    if (j < N) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(pts + j * 4);
        s0 += __builtin_amdgcn_exp2f(((-2.f * fmaf(q[2], v[2], fmaf(q[1], v[1], q[0] * v[0])) + q[3]) + v[3]) * k2);
    }
    inv_density[(size_t)b * N + i] = 1.0f / (((s0 + s1) / c1) / (float)N);
}

// ---- K nearest points (models/pointconv.py:103-114): queries on the lanes, keys on the accumulator rows ----
template <int K>
__global__ __launch_bounds__(kThreads, 1) void pc_knn_kernel(const float* __restrict__ keys8 /*(B,N,8)*/,
                                                             const float* __restrict__ q8 /*(B,S,8)*/,
                                                             int16_t* __restrict__ idx /*(B,S,K)*/, int N, int S, int B,
                                                             int wgs_per_cloud, const int32_t* __restrict__ n_unique) {
    constexpr int LD = 12;
    __shared__ __attribute__((aligned(16))) float tile[2][32 * LD];
    __shared__ float kxx[2][32];
    __shared__ double queue[kThreads / 64][16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = blockIdx.x >> 3;   // XCD-aware: a cloud's workgroups share one L2
    const int b = iq::xcd_cloud(blockIdx.x, wgs_per_cloud, B);
    if (b >= B) return;
    if (n_unique && (slot % wgs_per_cloud) * 128 >= n_unique[b]) return;   // duplicate centroids (copies of centroid 0): filled later
    const float* kb = keys8 + (size_t)b * N * 8;
    const int q0 = (slot % wgs_per_cloud) * 128 + wave * 32;
    const int fl = lane & 31, fh = lane >> 5;
    const int qrow = min(q0 + fl, S - 1);
    const float* qp = q8 + ((size_t)b * S + qrow) * 8;
    const f32x4 qf = *reinterpret_cast<const f32x4*>(qp + 4 * fh);
    const float xxq = (qp[0] * qp[0] + qp[1] * qp[1]) + qp[2] * qp[2];

    auto stage = [&](int t, int buf) {
        if (tid < 64) {
            const int row = tid >> 1, half = tid & 1;
            *reinterpret_cast<f32x4*>(&tile[buf][row * LD + half * 4]) =
                *reinterpret_cast<const f32x4*>(kb + (size_t)(t * 32 + row) * 8 + half * 4);
        }
        if (tid >= 64 && tid < 96) {
            const float* p = kb + (size_t)(t * 32 + tid - 64) * 8;
            kxx[buf][tid - 64] = (p[0] * p[0] + p[1] * p[1]) + p[2] * p[2];
        }
    };
    QueuedTopK<K, 16> top;
    top.init(queue[wave]);
    const int ntiles = N / 32;
    stage(0, 0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) stage(t + 1, buf ^ 1);
        f32x16 acc = {0};
        acc = mfma4(lds_frag<LD>(tile[buf] + fl * LD + 4 * fh, 0, 0), qf, acc);
        float d[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)  // -(((-2 q.p) + |q|^2) + |p|^2): largest = nearest
            d[r] = -(((-2.f * acc[r]) + xxq) + kxx[buf][c_row(r, lane)]);
        top.offer_tile(d, t * 32, lane);
        __syncthreads();
    }
    top.flush(lane);
    top.merge_halves();
    if (fh == 0 && q0 + fl < S) {
        int16_t* o = idx + ((size_t)b * S + q0 + fl) * K;
#pragma unroll
        for (int q = 0; q < K; ++q) o[q] = (int16_t)top.index(q);
    }
}

// ---- the two tiny per-member nets (models/pointconv.py:212-265), BN folded: [w | b] rows ---------------------
struct TinyNets {
    const float* dn;  // densitynet: l0 16x(1+1), l1 8x(16+1), l2 1x(8+1)  = 32 + 136 + 9 floats
    const float* wn;  // weightnet : l0 8x(3+1),  l1 8x(8+1),  l2 16x(8+1) = 32 + 72 + 144 floats
};

__device__ __forceinline__ float density_net(const float* __restrict__ p, float rho) {
    float h0[16], h1[8];
#pragma unroll
    for (int o = 0; o < 16; ++o) h0[o] = fmaxf(fmaf(p[o * 2], rho, p[o * 2 + 1]), 0.f);
    p += 32;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        float a = p[o * 17 + 16];
#pragma unroll
        for (int i = 0; i < 16; ++i) a = fmaf(p[o * 17 + i], h0[i], a);
        h1[o] = fmaxf(a, 0.f);
    }
    p += 136;
    float a = p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a = fmaf(p[i], h1[i], a);
    return fmaxf(a, 0.f);
}

__device__ __forceinline__ void weight_net(const float* __restrict__ p, float dx, float dy, float dz, float (&out)[16]) {
    float h0[8], h1[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) h0[o] = fmaxf(fmaf(p[o * 4 + 2], dz, fmaf(p[o * 4 + 1], dy, fmaf(p[o * 4], dx, p[o * 4 + 3]))), 0.f);
    p += 32;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        float a = p[o * 9 + 8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a = fmaf(p[o * 9 + i], h0[i], a);
        h1[o] = fmaxf(a, 0.f);
    }
    p += 72;
#pragma unroll
    for (int o = 0; o < 16; ++o) {
        float a = p[o * 9 + 8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a = fmaf(p[o * 9 + i], h1[i], a);
        out[o] = fmaxf(a, 0.f);
    }
}

// ---- per-member pre-pass: relative coordinates and density scale x WeightNet ------------------------------------------
// One thread per member (b, s, k): rel = x_p - c (+ the member index), sw[16] = DensityNet(rho_p / max rho in the group) x
// WeightNet(rel).  ~450 flops per member on ALL lanes; inside the grouped kernel this ran on 64 of 256 threads in front
// of every chunk's first MFMA.  80 bytes per member through HBM (1.3 MB per coalition for sa1) buys a grouped kernel
// whose prologue is two coalesced loads.
template <int K>
__global__ __launch_bounds__(kThreads) void pc_member_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                             const int16_t* __restrict__ idx, const float* __restrict__ inv_density,
                                                             TinyNets nets, float* __restrict__ mrel /*(B,S,K,4)*/,
                                                             float* __restrict__ msw /*(B,S,K,16)*/, int N, int S, size_t total,
                                                             const int32_t* __restrict__ n_unique) {
    const size_t t = (size_t)blockIdx.x * kThreads + threadIdx.x;       // member index; K divides 64: groups never straddle waves
    const bool live = t < total;
    const size_t tt = live ? t : total - 1;
    const size_t g = tt / K;                                             // b * S + s
    const int bb = (int)(g / S);
    if (n_unique && (int)(g - (size_t)bb * S) >= n_unique[bb]) return;   // a duplicate centroid's group (whole K-lane group leaves)
    const int p = idx[tt];
    const float* x = xyz + ((size_t)bb * N + p) * 3;
    const float* c = new_xyz + g * 3;
    const float dx = x[0] - c[0], dy = x[1] - c[1], dz = x[2] - c[2];
    const float rho = inv_density[(size_t)bb * N + p];
    float mx = rho;
#pragma unroll
    for (int o = 1; o < K; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));   // max over the K members of the group
    const float sc = density_net(nets.dn, rho / mx);
    float wt[16];
    weight_net(nets.wn, dx, dy, dz, wt);
    if (!live) return;
    *reinterpret_cast<f32x4*>(mrel + t * 4) = (f32x4){dx, dy, dz, __int_as_float(p)};
#pragma unroll
    for (int w4 = 0; w4 < 4; ++w4)
        *reinterpret_cast<f32x4*>(msw + t * 16 + w4 * 4) = (f32x4){sc * wt[w4 * 4], sc * wt[w4 * 4 + 1], sc * wt[w4 * 4 + 2], sc * wt[w4 * 4 + 3]};
}

// ---- grouped MLP + density scale + WeightNet contraction ------------------------------------------------------
struct PcGroupArgs {
    const float* mrel;         // (B,S,K,4) relative coordinates + member index (pc_member_kernel)
    const float* msw;          // (B,S,K,16) density scale x WeightNet output
    const float* U;            // (B,N,ldu) per-point part of layer 1 (bias included) or null
    int ldu;
    const float* w1x;          // [C1][4] = (wx0, wx1, wx2, bias)
    const float* w2; const float* b2;
    const float* w3; const float* b3;
    const unsigned short* w2_bf3;  // the same two layers as three bf16 terms (iq_pack_weight_bf3) or null:
    const unsigned short* w3_bf3;  // the 128-128-256 stage then runs pc_group_bf3_kernel
    float* out;                // (B,S,C3*16): [c][w]
    int N, S, K;    int B, wgs_per_cloud, chunks_per_wg;
    const int32_t* n_unique;   // (B) or null: groups s >= n_unique[b] are copies of group 0 and are not computed
};

// One workgroup = `chunks_per_wg` consecutive 64-member chunks of one cloud (K = 32: two groups per chunk, K = 64: one).
//   stage 0a  members' rel / sw rows -> LDS (sw transposed: swT[w][member], row 16 = zeros)      [coalesced, one chunk ahead]
//   stage 0b  layer 1 (VALU, 4 channels per thread) + U[p] rows by 16-byte buffer loads -> act1
//   L2, L3    fp32 MFMA as in pn2_group_kernel
//   contraction out[c][w] = sum_k h[k][c] sw[k][w] ON THE MFMA (v_mfma_f32_16x16x4_f32, all 16 columns = the 16 WeightNet
//             outputs): two neighbouring accumulator registers of an L3 tile (bias + ReLU applied) hold, per 16-lane row,
//             h[member][c] for members m, m+1 (rows 0/1: channels 0-15 / 16-31) and m+4, m+5 (rows 2/3); ONE
//             v_permlane16_swap turns the pair into the two A operands (channels 0-15 / 16-31 x those 4 members) of a
//             16x16x4 MFMA whose B operand is sw[member][w], read from LDS as one dword per lane.  16 half-size MFMAs per
//             32x32 tile (+8 % / +17 % matrix work for sa2 / sa1) replace 256 FMAs + 64 LDS reads per lane and tile on the
//             VALU (during which the matrix pipe of that wave sat idle) and 32 live registers, which is what lets the next
//             chunk's gather travel behind the L3 MFMAs.
template <int C1, int C2, int C3>
__global__ __launch_bounds__(kThreads, 2) void pc_group_kernel(PcGroupArgs a) {
    constexpr int LD1 = C1 + 4, LD2 = C2 + 4, LDS_SW = kMC + 2;   // swT row stride 66: conflict-free dword reads
    constexpr int KB1 = C1 / 8, KB2 = C2 / 8, NT2 = C2 / 32, NT3 = C3 / 32;
    static_assert(NT3 >= 4, "C3 >= 128");
    constexpr int Q1 = C1 / 4, NR = kMC * Q1 / kThreads;
    __shared__ __attribute__((aligned(16))) float act1[kMC * LD1];
    __shared__ __attribute__((aligned(16))) float act2[kMC * LD2];
    __shared__ __attribute__((aligned(16))) float rel[2 * kMC * 4];          // dx,dy,dz, member index (bits); double-buffered
    __shared__ __attribute__((aligned(16))) float swT[2 * 16 * LDS_SW];      // [buf][w][member]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 1-D grid; workgroups go round-robin over the 8 XCDs: keep all workgroups of a cloud on one XCD (one L2 holds its
    // U rows and member records)
    const int slot = blockIdx.x >> 3;
    const int b = iq::xcd_cloud(blockIdx.x, a.wgs_per_cloud, a.B);
    if (b >= a.B) return;
    const int K = a.K;                                   // 32 or 64
    const int live_groups = a.n_unique ? min(a.S, a.n_unique[b]) : a.S;
    const int chunks_total = (live_groups * K + kMC - 1) / kMC;
    const int ch0 = (slot % a.wgs_per_cloud) * a.chunks_per_wg;
    if (ch0 >= chunks_total) return;
    const int nchunks = min(a.chunks_per_wg, chunks_total - ch0);
    const int fl = lane & 31, fh = lane >> 5;
    const float* a1base = act1 + fl * LD1 + 4 * fh;
    const float* a2base = act2 + fl * LD2 + 4 * fh;
    float* c2base = act2 + (4 * fh) * LD2 + fl;
    const int wave_s = uniform(wave);
    const WBuf w2b = wbuf_make(a.w2, lane), w3b = wbuf_make(a.w3, lane);
    const size_t member0 = (size_t)b * a.S * K;          // first member record of this cloud

    // B operand of the contraction: lane (w = lane & 15, k = lane >> 4) reads sw of member base + (k & 1) + 4 (k >> 1)
    const float* swlane = swT + (lane & 15) * LDS_SW + ((lane >> 4) & 1) + 4 * (lane >> 5);

    const int c4 = tid % Q1, rsub = tid / Q1;
    f32x4 w1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w1[e] = *reinterpret_cast<const f32x4*>(a.w1x + (c4 * 4 + e) * 4);
    const __amdgpu_buffer_rsrc_t ursrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.U ? a.U + (size_t)b * a.N * a.ldu : a.w1x), 0, 0x7fffffff, 0x00020000);
    f32x4 ureg[NR];
    auto stage0a = [&](int ch, int buf) {
        const size_t m = member0 + (size_t)(ch0 + ch) * kMC;
        if (tid < kMC) *reinterpret_cast<f32x4*>(rel + (buf * kMC + tid) * 4) = *reinterpret_cast<const f32x4*>(a.mrel + (m + tid) * 4);
        const int mem = tid >> 2, q4 = tid & 3;
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.msw + (m + mem) * 16 + q4 * 4);
        float* dst = swT + buf * 16 * LDS_SW + (q4 * 4) * LDS_SW + mem;
        dst[0] = v[0]; dst[LDS_SW] = v[1]; dst[2 * LDS_SW] = v[2]; dst[3 * LDS_SW] = v[3];
    };
    auto gather_u = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = rsub + i * (kThreads / Q1);
            const int p = __float_as_int(rel[(buf * kMC + r) * 4 + 3]);
            if (a.U) ureg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ursrc, (p * a.ldu + c4 * 4) * 4, 0, 0));
            else ureg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    stage0a(0, 0);
    __syncthreads();
    gather_u(0);

    WRing ring2, ring3;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int cur = ch & 1, nxt = cur ^ 1;
        if (NT2 >= 4) wring_prime(ring2, w2b, wave_s * KB1 * kFragBytes);   // in flight across stage 0b
        // ---- stage 0b: layer 1 -> act1 ----------------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = rsub + i * (kThreads / Q1);
            const f32x4 v = *reinterpret_cast<const f32x4*>(rel + (cur * kMC + r) * 4);
            f32x4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = fmaf(w1[e][2], v[2], fmaf(w1[e][1], v[1], w1[e][0] * v[0])) + w1[e][3];
                if (a.U) t += ureg[i][e];
                h[e] = fmaxf(t, 0.f);
            }
            *reinterpret_cast<f32x4*>(act1 + r * LD1 + c4 * 4) = h;
        }
        __syncthreads();  // act1 complete; every wave has finished L3 of the previous chunk (act2, swT[nxt], rel[nxt] are free)
        // ---- layer 2 -------------------------------------------------------------------------------------------------
        if (NT2 >= 4) {
#pragma unroll
            for (int q = 0; q < NT2 / 4; ++q) {
                const int nt = q * 4 + wave, nts = q * 4 + wave_s;
                f32x16 acc0 = {0}, acc1 = {0};
                const int wq = nts * KB1 * kFragBytes;
                const int wn = (q + 1 < NT2 / 4 ? nts + 4 : nts) * KB1 * kFragBytes;
                mfma_ntile<LD1, KB1, 2>(a1base, w2b, wq, wn, ring2, acc0, acc1);
                const float bias = a.b2[nt * 32 + fl];
                float* dst = c2base + nt * 32;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    dst[c_row_i(i) * LD2] = fmaxf(acc0[i] + bias, 0.f);
                    dst[(32 + c_row_i(i)) * LD2] = fmaxf(acc1[i] + bias, 0.f);
                }
            }
        } else {
            for (int t = wave; t < 2 * NT2; t += 4) {
                const int mt = t / NT2, nt = t - mt * NT2;
                f32x16 acc = {0};
                const int wq = uniform(nt) * KB1 * kFragBytes;
#pragma unroll 4
                for (int kb = 0; kb < KB1; ++kb)
                    acc = mfma4(lds_frag<LD1>(a1base + mt * 32 * LD1, 0, kb), wbuf_load(w2b, wq + kb * kFragBytes), acc);
                const float bias = a.b2[nt * 32 + fl];
                float* dst = c2base + mt * 32 * LD2 + nt * 32;
#pragma unroll
                for (int i = 0; i < 16; ++i) dst[c_row_i(i) * LD2] = fmaxf(acc[i] + bias, 0.f);
            }
        }
        wring_prime(ring3, w3b, wave_s * KB2 * kFragBytes);                 // in flight across the barrier
        if (ch + 1 < nchunks) stage0a(ch + 1, nxt);
        __syncthreads();  // act2 complete; rel[nxt] / swT[nxt] visible
        if (ch + 1 < nchunks) gather_u(nxt);                                // consumed after L3
        // ---- layer 3 + contraction over the members (MFMA) -------------------------------------------------------------
        const float* swc = swlane + cur * 16 * LDS_SW;
        const int g_first = (ch0 + ch) * (kMC / K);                         // first group of this chunk
#pragma unroll
        for (int q = 0; q < NT3 / 4; ++q) {
            const int nt = q * 4 + wave, nts = q * 4 + wave_s;
            f32x16 acc0 = {0}, acc1 = {0};
            const int wq = nts * KB2 * kFragBytes;
            const int wn = (q + 1 < NT3 / 4 ? nts + 4 : nts) * KB2 * kFragBytes;
            mfma_ntile<LD2, KB2, 2>(a2base, w3b, wq, wn, ring3, acc0, acc1);
            const float bias = a.b3[nt * 32 + fl];
            // d[mt][half]: 16x16 tiles (channels nt*32 + 16 half + 4 (lane >> 4) + j, w = lane & 15) of m-tile mt
            f32x4 d00 = {0, 0, 0, 0}, d01 = d00, d10 = d00, d11 = d00;
#pragma unroll
            for (int p = 0; p < 8; ++p) {      // accumulator registers 2 p, 2 p + 1: members c_row_i(2 p) + {0, 1, 4, 5}
                const int m = c_row_i(2 * p);
                const float s0 = swc[m], s1 = swc[32 + m];
                const auto a0 = __builtin_amdgcn_permlane16_swap(__float_as_uint(fmaxf(acc0[2 * p] + bias, 0.f)),
                                                                 __float_as_uint(fmaxf(acc0[2 * p + 1] + bias, 0.f)), false, false);
                const auto a1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(fmaxf(acc1[2 * p] + bias, 0.f)),
                                                                 __float_as_uint(fmaxf(acc1[2 * p + 1] + bias, 0.f)), false, false);
                d00 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a0[0]), s0, d00, 0, 0, 0);
                d01 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a0[1]), s0, d01, 0, 0, 0);
                d10 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a1[0]), s1, d10, 0, 0, 0);
                d11 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a1[1]), s1, d11, 0, 0, 0);
            }
            const int w = lane & 15, cq = 4 * (lane >> 4);
            float* dst = a.out + ((size_t)b * a.S + g_first) * (C3 * 16) + (size_t)(nt * 32 + cq) * 16 + w;
            if (K == 64) {        // one group: both m-tiles
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dst[j * 16] = d00[j] + d10[j];
                    dst[(16 + j) * 16] = d01[j] + d11[j];
                }
            } else {              // K == 32: m-tile 0 = group g_first, m-tile 1 = g_first + 1
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dst[j * 16] = d00[j];
                    dst[(16 + j) * 16] = d01[j];
                    dst[C3 * 16 + j * 16] = d10[j];
                    dst[C3 * 16 + (16 + j) * 16] = d11[j];
                }
            }
        }
    }
}

// ---- the 128-128-256 stage (sa2) on the bf16 matrix pipe: bf16x3, float32-exact (iq_bf3.h) ----------------------
// pc_group_kernel with layers 2 and 3 as six bf16 products per float32 product.  As in pn2_group_bf3_kernel (iq_pointnet2.hip):
// activations as three bf16 planes of 272-byte rows, act1 and act2 in ONE 52 KB image (layer 2's tiles wait in registers for
// the barrier), layer 3 as 2 x 2 tiles per wave, weights through small register rings.  The contraction over the members stays
// on v_mfma_f32_16x16x4_f32 (float32 operands straight out of the accumulators).
template <int MTS, bool TR>   // TR: transposed tiles (weights as the A operand), as gb_layer2 in iq_pointnet2.hip
__device__ __forceinline__ void pcb_layer2(const unsigned char* abase, const __amdgpu_buffer_rsrc_t& rs, int voff, int nt,
                                           B3 (&ring)[4], f32x16 (&acc)[MTS][1]) {
    constexpr int ROWB = 272, PLANEB = 64 * ROWB, TS = 4 * 8 * 1024;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        bf16x8 af[MTS][3];
#pragma unroll
        for (int i = 0; i < MTS; ++i) a3_load<PLANEB>(af[i], abase + i * 32 * ROWB, ks);
        const B3 b[1] = {ring[ks & 3]};
        if (ks + 4 < 8) ring[ks & 3] = b3_load_at(rs, voff, (nt * 8 + ks + 4) * 1024, TS);
        if (TR) mfma_bf3_block_tr<MTS>(af, b[0], acc);
        else mfma_bf3_block<MTS, 1>(af, b, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
}
struct PcB3x2 { B3 b[2]; };
__device__ __forceinline__ void pcb_layer3(const unsigned char* abase, const __amdgpu_buffer_rsrc_t& rs, int voff, int nt0,
                                           PcB3x2 (&ring)[2], f32x16 (&acc)[2][2]) {
    constexpr int ROWB = 272, PLANEB = 64 * ROWB, TS = 8 * 8 * 1024;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        bf16x8 af[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i) a3_load<PLANEB>(af[i], abase + i * 32 * ROWB, ks);
        const B3 b[2] = {ring[ks & 1].b[0], ring[ks & 1].b[1]};
        if (ks + 2 < 8) {
            ring[ks & 1].b[0] = b3_load_at(rs, voff, (nt0 * 8 + ks + 2) * 1024, TS);
            ring[ks & 1].b[1] = b3_load_at(rs, voff, ((nt0 + 4) * 8 + ks + 2) * 1024, TS);
        }
        mfma_bf3_block<2, 2>(af, b, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// TR (the default): layer 2's tiles transposed - act2 stored with whole 8-byte stores, no two-lane DPP trade (tuning key 7 = 1: the
// untransposed form; same products in the same order).
template <bool TR>
__global__ __launch_bounds__(kThreads, 2) void pc_group_bf3_kernel(PcGroupArgs a) {
    constexpr int C1 = 128, C3 = 256, ROWB = 272, PLANEB = kMC * ROWB, LDS_SW = kMC + 2;
    constexpr int Q1 = C1 / 4, NR = kMC * Q1 / kThreads;
    __shared__ __attribute__((aligned(16))) unsigned char planes[3 * PLANEB];   // act1, then act2: three bf16 planes [64][136]
    __shared__ __attribute__((aligned(16))) float rel[2 * kMC * 4];
    __shared__ __attribute__((aligned(16))) float swT[2 * 16 * LDS_SW];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = blockIdx.x >> 3;
    const int b = iq::xcd_cloud(blockIdx.x, a.wgs_per_cloud, a.B);
    if (b >= a.B) return;
    const int K = a.K;
    const int live_groups = a.n_unique ? min(a.S, a.n_unique[b]) : a.S;
    const int chunks_total = (live_groups * K + kMC - 1) / kMC;
    const int ch0 = (slot % a.wgs_per_cloud) * a.chunks_per_wg;
    if (ch0 >= chunks_total) return;
    const int nchunks = min(a.chunks_per_wg, chunks_total - ch0);
    const int fl = lane & 31, fh = lane >> 5;
    const unsigned char* abase = planes + fl * ROWB + 16 * fh;
    const int wave_s = uniform(wave), voff = lane * 16;
    const __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.w2_bf3), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t w3rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.w3_bf3), 0, 0x7fffffff, 0x00020000);
    const size_t member0 = (size_t)b * a.S * K;
    const float* swlane = swT + (lane & 15) * LDS_SW + ((lane >> 4) & 1) + 4 * (lane >> 5);

    const int c4 = tid % Q1, rsub = tid / Q1;
    f32x4 w1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w1[e] = *reinterpret_cast<const f32x4*>(a.w1x + (c4 * 4 + e) * 4);
    const __amdgpu_buffer_rsrc_t ursrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.U ? a.U + (size_t)b * a.N * a.ldu : a.w1x), 0, 0x7fffffff, 0x00020000);
    f32x4 ureg[NR];
    auto stage0a = [&](int ch, int buf) {   // as pc_group_kernel
        const size_t m = member0 + (size_t)(ch0 + ch) * kMC;
        if (tid < kMC) *reinterpret_cast<f32x4*>(rel + (buf * kMC + tid) * 4) = *reinterpret_cast<const f32x4*>(a.mrel + (m + tid) * 4);
        const int mem = tid >> 2, q4 = tid & 3;
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.msw + (m + mem) * 16 + q4 * 4);
        float* dst = swT + buf * 16 * LDS_SW + (q4 * 4) * LDS_SW + mem;
        dst[0] = v[0]; dst[LDS_SW] = v[1]; dst[2 * LDS_SW] = v[2]; dst[3 * LDS_SW] = v[3];
    };
    auto gather_u = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = rsub + i * (kThreads / Q1);
            const int p = __float_as_int(rel[(buf * kMC + r) * 4 + 3]);
            if (a.U) ureg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ursrc, (p * a.ldu + c4 * 4) * 4, 0, 0));
            else ureg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    stage0a(0, 0);
    __syncthreads();
    gather_u(0);

    for (int ch = 0; ch < nchunks; ++ch) {
        const int cur = ch & 1, nxt = cur ^ 1;
        B3 ring2[4];                                 // layer 2's weights (n-tile = wave), in flight across stage 0b
#pragma unroll
        for (int i = 0; i < 4; ++i) ring2[i] = b3_load_at(w2rs, voff, (wave_s * 8 + i) * 1024, 4 * 8 * 1024);
        // ---- stage 0b: layer 1 -> act1 (three planes) ------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = rsub + i * (kThreads / Q1);
            const f32x4 v = *reinterpret_cast<const f32x4*>(rel + (cur * kMC + r) * 4);
            f32x4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = fmaf(w1[e][2], v[2], fmaf(w1[e][1], v[1], w1[e][0] * v[0])) + w1[e][3];
                if (a.U) t += ureg[i][e];
                h[e] = fmaxf(t, 0.f);
            }
            row4_to_planes<PLANEB>(planes + r * ROWB + c4 * 8, h);
        }
        __syncthreads();  // act1 complete
        // ---- layer 2: tiles (m-tile 0..1, n-tile = wave) kept in registers ---------------------------------------------------
        f32x16 acc2[2][1] = {{{0}}, {{0}}};
        pcb_layer2<2, TR>(abase, w2rs, voff, wave_s, ring2, acc2);
        PcB3x2 ring3[2];                             // layer 3's weights (n-tiles wave, wave + 4), in flight across the epilogue
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ring3[i].b[0] = b3_load_at(w3rs, voff, (wave_s * 8 + i) * 1024, 8 * 8 * 1024);
            ring3[i].b[1] = b3_load_at(w3rs, voff, ((wave_s + 4) * 8 + i) * 1024, 8 * 8 * 1024);
        }
        if (ch + 1 < nchunks) stage0a(ch + 1, nxt);
        __syncthreads();  // every wave has read act1: the image is free
        {
            if (TR) {   // register r = channel c_row_i(r) + 4 fh of this wave's n-tile
                f32x4 bq[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const f32x4*>(a.b2 + wave * 32 + 8 * g + 4 * fh);
                ct_tile_to_planes<ROWB, PLANEB>(planes + wave * 64, lane, [&](int r) { return fmaxf(acc2[0][0][r] + bq[r >> 2][r & 3], 0.f); });
                ct_tile_to_planes<ROWB, PLANEB>(planes + 32 * ROWB + wave * 64, lane, [&](int r) { return fmaxf(acc2[1][0][r] + bq[r >> 2][r & 3], 0.f); });
            } else {
                const float bias = a.b2[wave * 32 + fl];
                c_tile_to_planes<ROWB, PLANEB>(planes + wave * 64, lane, [&](int i) { return fmaxf(acc2[0][0][i] + bias, 0.f); });
                c_tile_to_planes<ROWB, PLANEB>(planes + 32 * ROWB + wave * 64, lane, [&](int i) { return fmaxf(acc2[1][0][i] + bias, 0.f); });
            }
        }
        __syncthreads();  // act2 complete; rel[nxt] / swT[nxt] visible
        if (ch + 1 < nchunks) gather_u(nxt);                                // consumed after layer 3
        // ---- layer 3 (2 x 2 tiles per wave) + contraction over the members (fp32 MFMA, as pc_group_kernel) -------------------
        f32x16 acc3[2][2] = {{{0}, {0}}, {{0}, {0}}};
        pcb_layer3(abase, w3rs, voff, wave_s, ring3, acc3);
        const float* swc = swlane + cur * 16 * LDS_SW;
        const int g_first = (ch0 + ch) * (kMC / K);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int nt = q * 4 + wave;
            const float bias = a.b3[nt * 32 + fl];
            f32x4 d00 = {0, 0, 0, 0}, d01 = d00, d10 = d00, d11 = d00;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int m = c_row_i(2 * p);
                const float s0 = swc[m], s1 = swc[32 + m];
                const auto a0 = __builtin_amdgcn_permlane16_swap(__float_as_uint(fmaxf(acc3[0][q][2 * p] + bias, 0.f)),
                                                                 __float_as_uint(fmaxf(acc3[0][q][2 * p + 1] + bias, 0.f)), false, false);
                const auto a1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(fmaxf(acc3[1][q][2 * p] + bias, 0.f)),
                                                                 __float_as_uint(fmaxf(acc3[1][q][2 * p + 1] + bias, 0.f)), false, false);
                d00 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a0[0]), s0, d00, 0, 0, 0);
                d01 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a0[1]), s0, d01, 0, 0, 0);
                d10 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a1[0]), s1, d10, 0, 0, 0);
                d11 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a1[1]), s1, d11, 0, 0, 0);
            }
            const int w = lane & 15, cq = 4 * (lane >> 4);
            float* dst = a.out + ((size_t)b * a.S + g_first) * (C3 * 16) + (size_t)(nt * 32 + cq) * 16 + w;
            if (K == 64) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dst[j * 16] = d00[j] + d10[j];
                    dst[(16 + j) * 16] = d01[j] + d11[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dst[j * 16] = d00[j];
                    dst[(16 + j) * 16] = d01[j];
                    dst[C3 * 16 + j * 16] = d10[j];
                    dst[C3 * 16 + (16 + j) * 16] = d11[j];
                }
            }
        }
        __syncthreads();  // every wave has read act2 and swT[cur]: the next chunk's stage 0b may overwrite the image
    }
}

// ---- sa3 (group all, models/pointconv.py:147-167) helpers ----------------------------------------------------------
// mean centre, relative coordinates, density scale and WeightNet for the S points of each cloud; layer 1.
__global__ __launch_bounds__(128) void pc_all_prepare_kernel(const float* __restrict__ xyz /*(B,S,3)*/,
                                                             const float* __restrict__ inv_density, TinyNets nets,
                                                             const float* __restrict__ w1x /*[C1][4]*/,
                                                             const float* __restrict__ U /*(B,S,C1)*/, int C1,
                                                             float* __restrict__ h1 /*(B,S,C1)*/,
                                                             float* __restrict__ sw /*(B,S,16)*/, int S) {
    __shared__ float red[128 * 4];
    __shared__ float rels[128 * 3];
    const int b = blockIdx.x, t = threadIdx.x;
    const float* p = xyz + ((size_t)b * S + t) * 3;
    const float x = t < S ? p[0] : 0.f, y = t < S ? p[1] : 0.f, z = t < S ? p[2] : 0.f;
    const float rho = t < S ? inv_density[(size_t)b * S + t] : -INFINITY;
    red[t * 4] = x; red[t * 4 + 1] = y; red[t * 4 + 2] = z; red[t * 4 + 3] = rho;
    __syncthreads();
    float mx = 0.f, my = 0.f, mz = 0.f, mr = -INFINITY;
    for (int i = 0; i < S; ++i) { mx += red[i * 4]; my += red[i * 4 + 1]; mz += red[i * 4 + 2]; mr = fmaxf(mr, red[i * 4 + 3]); }
    mx /= (float)S; my /= (float)S; mz /= (float)S;   // xyz.mean(dim=1)
    const float dx = x - mx, dy = y - my, dz = z - mz;
    if (t < S) {
        const float s = density_net(nets.dn, rho / mr);
        float wt[16];
        weight_net(nets.wn, dx, dy, dz, wt);
        float* o = sw + ((size_t)b * S + t) * 16;
#pragma unroll
        for (int w = 0; w < 16; ++w) o[w] = s * wt[w];
        rels[t * 3] = dx; rels[t * 3 + 1] = dy; rels[t * 3 + 2] = dz;
    }
    __syncthreads();
    for (int e = t; e < S * C1; e += 128) {
        const int r = e / C1, c = e - r * C1;
        const f32x4 w = *reinterpret_cast<const f32x4*>(w1x + c * 4);
        const float h = fmaf(w[2], rels[r * 3 + 2], fmaf(w[1], rels[r * 3 + 1], w[0] * rels[r * 3])) + w[3] + U[((size_t)b * S + r) * C1 + c];
        h1[((size_t)b * S + r) * C1 + c] = fmaxf(h, 0.f);
    }
}

// out[b][c*16 + w] = sum_k h[b][k][c] * sw[b][k][w]
__global__ __launch_bounds__(kThreads) void pc_all_contract_kernel(const float* __restrict__ h, const float* __restrict__ sw,
                                                                   float* __restrict__ out, int S, int C) {
    __shared__ __attribute__((aligned(16))) float s[128 * 16];
    const int b = blockIdx.y;
    for (int e = threadIdx.x; e < S * 16; e += kThreads) s[e] = sw[(size_t)b * S * 16 + e];
    __syncthreads();
    const int c = blockIdx.x * kThreads + threadIdx.x;
    if (c >= C) return;
    float o[16];
#pragma unroll
    for (int w = 0; w < 16; ++w) o[w] = 0.f;
    for (int k = 0; k < S; ++k) {
        const float hv = h[((size_t)b * S + k) * C + c];
#pragma unroll
        for (int w = 0; w < 16; ++w) o[w] = fmaf(hv, s[k * 16 + w], o[w]);
    }
    float* dst = out + ((size_t)b * C + c) * 16;
#pragma unroll
    for (int w = 0; w < 16; ++w) dst[w] = o[w];
}

// (B,n,3) -> (B,np,8), np >= n a multiple of 32: padding rows sit at 1e18 in every coordinate, so as keys they are further
// than any real point (|p|^2 = 3e36 stays finite) and are never among the K nearest
__global__ void pc_pad8_kernel(const float* __restrict__ xyz, float* __restrict__ out, int B, int n, int np) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * np) return;
    const int b = t / np, i = t - b * np;
    f32x4 a = {1e18f, 1e18f, 1e18f, 0.f};
    if (i < n) {
        const float* src = xyz + ((size_t)b * n + i) * 3;
        a = (f32x4){src[0], src[1], src[2], 0.f};
    }
    reinterpret_cast<f32x4*>(out)[(size_t)t * 2] = a;
    reinterpret_cast<f32x4*>(out)[(size_t)t * 2 + 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// rows s >= n_unique[b] of (B,S,C) := row 0 of the cloud (duplicate centroids: FPS returns index 0 once the distinct
// locations are used up, so those groups ARE group 0)
__global__ void pc_fill_dup_rows_kernel(float* __restrict__ out, int S, int C, const int32_t* __restrict__ n_unique) {
    const int b = blockIdx.y, s = n_unique[b] + blockIdx.x;
    if (s >= S) return;
    const float* src = out + (size_t)b * S * C;
    float* dst = out + ((size_t)b * S + s) * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) dst[c] = src[c];
}

// ==== coalitions: the K-nearest groups of sa1 / sa2 from the source cloud's sorted neighbour lists (iq_srclist.h) ==========
// All coalitions of a call are masked copies of a few source clouds.  In xyz space the distance between two source points
// (or a point and the centre, where all masked points sit) does not depend on the coalition, so the K nearest points of a
// centroid are the first entries of its source point's sorted list that exist in the coalition:
//   sa1 (keys = the N points of the masked cloud): a kept point is itself; the centre entry stands for ALL masked points
//       (identical coordinates and densities, so which of them fill the group does not matter to the sum over members);
//   sa2 (keys = the 512 sa1 centroids): a source point exists if sa1's FPS picked it (position < n_unique); the location of
//       position 0 also owns the 512 - n_unique duplicate positions that FPS returns once the distinct locations are used up.
// The distances are pc_knn_kernel's own (sl_dist_kernel<1>), so the groups are the same point sets up to ties.

// per coalition: kept-point bitmap (32 words), the first 64 masked point indices and their number
__global__ __launch_bounds__(64) void pc_coal_prep_kernel(const int32_t* __restrict__ region_id, const uint64_t* __restrict__ keep,
                                                          const int32_t* __restrict__ cloud_of, uint32_t* __restrict__ kept,
                                                          int16_t* __restrict__ mfirst, int32_t* __restrict__ mcount, int N,
                                                          int nclouds) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int c = cloud_of ? cloud_of[b] : (nclouds == 1 ? 0 : b);
    const uint64_t k = keep[b];
    const int32_t* rid = region_id + (size_t)c * N;
    int nm = 0;
    for (int i0 = 0; i0 < kWalkMaxN; i0 += 64) {
        const int i = i0 + lane;
        const bool in = i < N;
        const bool kp = in && iq::keep_bit(k, rid[min(i, N - 1)]);
        const unsigned long long m = __ballot(kp), mm = __ballot(in && !kp);
        if (lane == 0) { kept[(size_t)b * 32 + (i0 >> 5)] = (uint32_t)m; kept[(size_t)b * 32 + (i0 >> 5) + 1] = (uint32_t)(m >> 32); }
        if (in && !kp) {
            const int pos = nm + __popcll(mm & ((1ull << lane) - 1ull));
            if (pos < 64) mfirst[(size_t)b * 64 + pos] = (int16_t)i;
        }
        nm += __popcll(mm);
    }
    if (lane == 0) mcount[b] = nm;
}

// sa1 groups: thread = centroid position s of coalition b; idx (B,S,K) indices into the masked cloud
template <int K>
__global__ __launch_bounds__(64) void pc_walk1_kernel(const int16_t* __restrict__ sorted, const uint32_t* __restrict__ kept,
                                                      const int16_t* __restrict__ mfirst, const int32_t* __restrict__ mcount,
                                                      const int32_t* __restrict__ fps, const int32_t* __restrict__ n_unique,
                                                      const int32_t* __restrict__ cloud_of, int16_t* __restrict__ idx, int N, int S,
                                                      int Nsl, int nclouds) {
    __shared__ uint32_t bits[32];
    const int b = blockIdx.y, lane = threadIdx.x, s = blockIdx.x * 64 + lane;
    if (blockIdx.x * 64 >= n_unique[b]) return;        // wave-uniform: duplicate centroids are filled afterwards
    if (lane < 32) bits[lane] = kept[(size_t)b * 32 + lane];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (s >= S || s >= n_unique[b]) return;
    const int c = cloud_of ? cloud_of[b] : (nclouds == 1 ? 0 : b);
    const int pi = fps[(size_t)b * S + s];
    const int ci = (bits[pi >> 5] >> (pi & 31)) & 1u ? pi : N;       // a masked centroid sits at the centre
    const int16_t* list = sorted + ((size_t)c * (N + 1) + ci) * Nsl;
    const int16_t* mf = mfirst + (size_t)b * 64;
    const int M = mcount[b];
    int16_t* o = idx + ((size_t)b * S + s) * K;
    int n = 0;
    for (int j0 = 0; j0 <= N && n < K; j0 += 8) {
        const uint4 chunk = *reinterpret_cast<const uint4*>(list + j0);
        const unsigned wds[4] = {chunk.x, chunk.y, chunk.z, chunk.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int p = (int)((wds[e >> 1] >> (16 * (e & 1))) & 0xffffu);
            if (j0 + e > N || n >= K) continue;
            if (p == N) {                                            // the centre: every masked point is here
                const int t = min(M, K - n);
                for (int u = 0; u < t; ++u) o[n++] = mf[u];
            } else if ((bits[p >> 5] >> (p & 31)) & 1u) {
                o[n++] = (int16_t)p;
            }
        }
    }
}

// per coalition: which source point each sa1 position stands for, and the position of each picked source point
__global__ __launch_bounds__(64) void pc_pos_kernel(const uint32_t* __restrict__ kept, const int32_t* __restrict__ fps1,
                                                    const int32_t* __restrict__ n_unique, int16_t* __restrict__ src1 /*(B,S1)*/,
                                                    int16_t* __restrict__ pos /*(B,Npos)*/, int N, int S1, int Npos) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const uint32_t* kb = kept + (size_t)b * 32;
    const int nu = n_unique[b];
    for (int i = lane; i < Npos; i += 64) pos[(size_t)b * Npos + i] = -1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int p0 = fps1[(size_t)b * S1];
    const int e0 = (kb[p0 >> 5] >> (p0 & 31)) & 1u ? p0 : N;
    for (int j = lane; j < S1; j += 64) {
        const int pi = fps1[(size_t)b * S1 + j];
        const int sp = j < nu ? ((kb[pi >> 5] >> (pi & 31)) & 1u ? pi : N) : e0;   // positions >= n_unique repeat position 0
        src1[(size_t)b * S1 + j] = (int16_t)sp;
        if (j < nu) pos[(size_t)b * Npos + sp] = (int16_t)j;
    }
}

// sa2 groups: thread = sa2 centroid s2 of coalition b; idx (B,S2,K) positions in the sa1 point list
template <int K>
__global__ __launch_bounds__(64) void pc_walk2_kernel(const int16_t* __restrict__ sorted, const int16_t* __restrict__ src1,
                                                      const int16_t* __restrict__ pos, const int32_t* __restrict__ fps2,
                                                      const int32_t* __restrict__ n_unique, const int32_t* __restrict__ cloud_of,
                                                      int16_t* __restrict__ idx, int N, int S1, int S2, int Nsl, int Npos,
                                                      int nclouds) {
    const int b = blockIdx.y, s = blockIdx.x * 64 + threadIdx.x;
    if (s >= S2) return;
    const int c = cloud_of ? cloud_of[b] : (nclouds == 1 ? 0 : b);
    const int nu = n_unique[b];
    const int16_t* ps = pos + (size_t)b * Npos;
    const int ci = src1[(size_t)b * S1 + fps2[(size_t)b * S2 + s]];
    const int e0 = src1[(size_t)b * S1];                             // the location that owns the duplicate positions nu .. S1-1
    const int16_t* list = sorted + ((size_t)c * (N + 1) + ci) * Nsl;
    int16_t* o = idx + ((size_t)b * S2 + s) * K;
    int n = 0;
    for (int j0 = 0; j0 <= N && n < K; j0 += 8) {
        const uint4 chunk = *reinterpret_cast<const uint4*>(list + j0);
        const unsigned wds[4] = {chunk.x, chunk.y, chunk.z, chunk.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int p = (int)((wds[e >> 1] >> (16 * (e & 1))) & 0xffffu);
            if (j0 + e > N || n >= K) continue;
            const int j = ps[p];
            if (j < 0) continue;
            o[n++] = (int16_t)j;
            if (p == e0) {
                const int t = min(S1 - nu, K - n);
                for (int u = 0; u < t; ++u) o[n++] = (int16_t)(nu + u);
            }
        }
    }
}

// ---- sa1 from a pair table -----------------------------------------------------------------------------------------------
// sa1 has no input features: a member's row after the 3 -> 64 -> 64 -> 128 MLP is a function of (member point, centroid
// point) only, and both are source points or the centre.  Per source cloud the rows of ALL (N+1)^2 pairs are computed once
// per call (layer 1 by the grouped kernel's own expression, layers 2 / 3 by the shared dense layer, whose MFMA order over
// k is the grouped kernel's: the same rows), and a group is 32 table rows contracted with the members' density x WeightNet
// weights (pc_member_kernel) - no MLP per coalition.  One wave per group: lane l holds channels 2l, 2l+1 against the 16
// WeightNet outputs (32 accumulators), a member costs one 8-byte load per lane (a coalesced 512-byte row) and 16 scalar
// weights.  (The sum over the members runs sequentially here and in 4-member MFMA steps in pc_group_kernel: equal up to
// rounding.)
__global__ void pc_tab_l1_kernel(const float* __restrict__ xs /*(n1p,8) source points + centre*/, const float* __restrict__ w1x,
                                 float* __restrict__ h1 /*(n1*n1, 64)*/, int n1) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // (row, channel quad)
    const size_t rows = (size_t)n1 * n1;
    if (t >= rows * 16) return;
    const size_t row = t >> 4;
    const int c4 = (int)(t & 15);
    const int q = (int)(row / n1), p = (int)(row - (size_t)q * n1);      // centroid q, member p
    const float* xp = xs + (size_t)p * 8;
    const float* xq = xs + (size_t)q * 8;
    const float v0 = xp[0] - xq[0], v1 = xp[1] - xq[1], v2 = xp[2] - xq[2];
    f32x4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(w1x + (c4 * 4 + e) * 4);
        const float tt = fmaf(w[2], v2, fmaf(w[1], v1, w[0] * v0)) + w[3];   // pc_group_kernel's stage 0b
        h[e] = fmaxf(tt, 0.f);
    }
    *reinterpret_cast<f32x4*>(h1 + row * 64 + c4 * 4) = h;
}

__global__ __launch_bounds__(kThreads) void pc_tab_group_kernel(const float* __restrict__ feat /*(nc, n1*n1, 128)*/,
                                                                const float* __restrict__ msw /*(B,S,32,16)*/,
                                                                const int16_t* __restrict__ idx /*(B,S,32)*/,
                                                                const int32_t* __restrict__ fps /*(B,S)*/,
                                                                const uint32_t* __restrict__ kept /*(B,32)*/,
                                                                const int32_t* __restrict__ n_unique, const int32_t* __restrict__ cloud_of,
                                                                float* __restrict__ out /*(B,S,2048)*/, int N, int S, int B, int nclouds) {
    const int lane = threadIdx.x & 63;
    const int g = uniform((int)(blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6)));   // group = b * S + s
    if (g >= B * S) return;
    const int b = g / S, s = g - b * S;
    if (s >= n_unique[b]) return;                          // duplicate centroids are filled afterwards
    const int c = cloud_of ? cloud_of[b] : (nclouds == 1 ? 0 : b);
    const uint32_t* kb = kept + (size_t)b * 32;
    const int n1 = N + 1;
    const int pi = fps[g];
    const int q = (kb[pi >> 5] >> (pi & 31)) & 1u ? pi : N;
    const float* tab = feat + ((size_t)c * n1 * n1 + (size_t)q * n1) * 128;
    const int16_t* mem = idx + (size_t)g * 32;
    const float* sw = msw + (size_t)g * 32 * 16;
    float acc[2][16];
#pragma unroll
    for (int w = 0; w < 16; ++w) { acc[0][w] = 0.f; acc[1][w] = 0.f; }
    // lane k resolves member k to its table row; the loop then reads the row numbers back as scalars (v_readlane), so the 32
    // row loads do not depend on one another and are all in flight
    int myrow = 0;
    if (lane < 32) {
        const int p = mem[lane];
        myrow = (kb[p >> 5] >> (p & 31)) & 1u ? p : N;
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const int ps = __builtin_amdgcn_readlane(myrow, k);
        const float2 f = *reinterpret_cast<const float2*>(tab + (size_t)ps * 128 + 2 * lane);
        const float* swk = sw + k * 16;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const float sv = swk[w];                       // wave-uniform: a scalar load
            acc[0][w] = fmaf(f.x, sv, acc[0][w]);
            acc[1][w] = fmaf(f.y, sv, acc[1][w]);
        }
    }
    float* o = out + (size_t)g * 2048 + (size_t)(2 * lane) * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4)
            *reinterpret_cast<f32x4*>(o + i * 16 + w4 * 4) = (f32x4){acc[i][w4 * 4], acc[i][w4 * 4 + 1], acc[i][w4 * 4 + 2], acc[i][w4 * 4 + 3]};
}


// ---- sa1 from the pair table, contraction and the 2048 -> 128 layer in ONE kernel ------------------------------------------
// pc_tab_group_kernel wrote the groups' (128 x 16) contractions as a (B, 512, 2048) tensor - 10.3 GB per 3300-coalition step
// with the duplicate centroids skipped - and the dense layer behind it read them straight back (measured, rocprofv3
// WRITE_SIZE / FETCH_SIZE: profiles/r04_pointconv_traffic.csv; 6.0 + 7.1 ms, the layer at 0.64 of the MFMA peak, waiting on
// that stream).  Here a workgroup owns 32 consecutive centroids of one coalition and walks the 2048 contraction columns in 8
// chunks of 16 channels x 16 WeightNet outputs = 256 columns:
//   contraction of chunk j, wave w, its 8 groups: D[c][o] = sum_k h[row_k][16 j + c] sw[k][o] as 8 v_mfma_f32_16x16x4_f32 per
//       group (A: one dword per lane straight from the table row - 16 lanes read 64 contiguous bytes of member 4 t + (lane >> 4);
//       B: sw, chunk-invariant, 64 registers per lane for the 8 groups), the 16 x 16 result -> LDS tile Ds[32 groups][256];
//   dense layer: acc[32 groups x 32 outputs of wave w] += Ds . W[256 j .. 256 j + 255] with the shared fp32 MFMA tile code
//       (weights from the packed image through the register ring, the same order over k as iq::launch_linear: given the same
//       contraction values the layer's result is bit-identical).
// Ds is double-buffered; the table loads of chunk j + 1 are issued before the layer's MFMAs of chunk j and land behind them.
// The contraction tensor never exists.  (The sum over the 32 members runs in 4-member MFMA steps here and sequentially in
// pc_tab_group_kernel: equal up to rounding, as pc_group_kernel's.)
struct PcFusedArgs {
    const float* feat;         // (nc, n1*n1, 128) sa1 pair table
    const float* msw;          // (B,S,32,16)
    const int16_t* idx;        // (B,S,32)
    const int32_t* fps;        // (B,S)
    const uint32_t* kept;      // (B,32)
    const int32_t* n_unique;   // (B)
    const int32_t* cloud_of;
    const float* w; const float* bias;   // the 2048 -> 128 layer, packed
    float* out;                // (B,S,128)
    int N, S, B, nclouds;
};

__global__ __launch_bounds__(kThreads, 2) void pc_tab_fused_kernel(PcFusedArgs a) {
    constexpr int GT = 32, CH = 256, LD = CH + 4, KBC = CH / 8, NCH = 2048 / CH, KBT = 2048 / 8;
    __shared__ __attribute__((aligned(16))) float Ds[2][GT * LD];
    __shared__ __attribute__((aligned(16))) int rowoff[GT * 32];     // [group][member & 3][member >> 2]: byte offset of the member's table row
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
    // Workgroups go round-robin over the 8 XCDs (blockIdx & 7).  XCD x owns the contiguous coalitions [x nb, (x + 1) nb),
    // nb = ceil(B / 8), and walks them tile index by tile index (tile 0 of all its coalitions, then tile 1, ...): every XCD
    // gets the same mix of small and large coalitions (a coalition's LIVE tiles are its first ones - with the plain order
    // b * 16 + t most live workgroups landed on the low XCDs: 58 against 44 of every 408), and the workgroups resident on one
    // L2 at a time are the same tile of neighbouring coalitions, whose centroids and members - table rows - largely coincide.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nb = (a.B + 7) >> 3;
    const int b = xcd * nb + slot % nb, s0 = (slot / nb) * GT;
    if (b >= a.B) return;
    const int c = a.cloud_of ? a.cloud_of[b] : (a.nclouds == 1 ? 0 : b);
    const int n1 = a.N + 1;
    // Everything the tile needs from memory before its first table load is requested in ONE round trip (the tile's 32 groups
    // are read as they lie - the entries of a group beyond the live count are never written: they are clamped into the table
    // below, the group's output is not stored - instead of centroid -> kept word -> members -> ... one after the other: five
    // dependent misses took a third of a tile's time, profiles/r04_pointconv_fused_probes.txt)
    const int nu_raw = a.n_unique[b];
    const uint32_t kword = a.kept[(size_t)b * 32 + (lane & 31)];      // lane l (and l + 32) holds kept word l
    const int gl0 = tid >> 3;
    const size_t g0 = (size_t)b * a.S + s0 + gl0;
    const int pi_raw = a.fps[g0];
    int p_raw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) p_raw[e] = a.idx[g0 * 32 + (tid & 7) * 4 + e];
    // B operand of the contraction for this wave's 8 groups: sw[member 4 t + (lane >> 4)][lane & 15]
    float swr[8][8];
#pragma unroll
    for (int gi = 0; gi < 8; ++gi) {
        const float* sw = a.msw + ((size_t)b * a.S + s0 + wave * 8 + gi) * 32 * 16 + (lane >> 4) * 16 + (lane & 15);
#pragma unroll
        for (int t = 0; t < 8; ++t) swr[gi][t] = sw[t * 64];
    }
    const int nu = min(a.S, nu_raw);
    if (s0 >= nu) return;                                             // duplicate centroids only: filled afterwards
    {   // member -> table row for the 32 groups of the tile
        auto kept_bit = [&](int p) { return (__shfl((int)kword, p >> 5, 64) >> (p & 31)) & 1; };
        const int pi = min(max(pi_raw, 0), a.N - 1);
        const int kq = kept_bit(pi);
        const int q = kq ? pi : a.N;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = (tid & 7) * 4 + e;
            const int p = min(max(p_raw[e], 0), a.N - 1);
            const int kp = kept_bit(p);               // (every lane takes part in the shuffle)
            const int row = ((unsigned)p_raw[e] < (unsigned)a.N && kp) ? p : a.N;
            rowoff[gl0 * 32 + (k & 3) * 8 + (k >> 2)] = (q * n1 + row) * 512;
        }
    }
    const __amdgpu_buffer_rsrc_t trsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.feat + (size_t)c * n1 * n1 * 128), 0, 0x7fffffff, 0x00020000);
    __syncthreads();
    float hreg[8][8];
    const int lane4 = (lane & 15) * 4;
    auto load_h = [&](int j) {       // chunk j: channels 16 j .. 16 j + 15 of the 32 member rows of each of the wave's groups
#pragma unroll
        for (int gi = 0; gi < 8; ++gi) {
            const int* ro = rowoff + (wave * 8 + gi) * 32 + (lane >> 4) * 8;
            const int4 r0 = *reinterpret_cast<const int4*>(ro), r1 = *reinterpret_cast<const int4*>(ro + 4);
            const int offs[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
            for (int t = 0; t < 8; ++t)
                hreg[gi][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(trsrc, offs[t] + lane4, j * 64, 0));
        }
    };
    auto contract = [&](int buf) {   // hreg x swr -> Ds[buf]
#pragma unroll
        for (int gi = 0; gi < 8; ++gi) {
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 8; ++t) d = __builtin_amdgcn_mfma_f32_16x16x4f32(hreg[gi][t], swr[gi][t], d, 0, 0, 0);
            // lane holds D[c = 4 (lane >> 4) + r][o = lane & 15] -> column c * 16 + o of the chunk
            float* dst = Ds[buf] + (wave * 8 + gi) * LD + (lane >> 4) * 64 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[r * 16] = d[r];
        }
    };
    const WBuf wb = wbuf_make(a.w, lane);
    const int wbase = wave * KBT * kFragBytes;        // this wave's output tile: columns 32 wave .. 32 wave + 31
    const float bias = a.bias[wave * 32 + (lane & 31)];
    load_h(0);
    contract(0);
    load_h(1);
    WRing ring;
    wring_prime(ring, wb, wbase);
    __syncthreads();
    f32x16 acc = {0}, unused = {0};
#pragma unroll 1
    for (int j = 0; j < NCH; ++j) {
        const float* abase = Ds[j & 1] + (lane & 31) * LD + 4 * (lane >> 5);
        const int scur = wbase + j * KBC * kFragBytes;
        mfma_ntile<LD, KBC, 1>(abase, wb, scur, j + 1 < NCH ? scur + KBC * kFragBytes : scur, ring, acc, unused);
        if (j + 1 < NCH) {
            contract((j + 1) & 1);
            if (j + 2 < NCH) load_h(j + 2);
        }
        __syncthreads();
    }
    const int col = wave * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int s = s0 + c_row(i, lane);
        if (s < nu) a.out[((size_t)b * a.S + s) * 128 + col] = fmaxf(acc[i] + bias, 0.f);
    }
}


struct WsC {
    float *inv1, *inv2, *inv3;
    int32_t *fps1, *fps2, *nu1;
    float *nx1, *nx2;          // (B,512,3), (B,128,3)
    float *k8, *q8;            // padded keys / queries for the kNN MFMA
    int16_t *idx1, *idx2;      // (B,512,32), (B,128,64)
    float *g1, *l1;            // (B,512,2048), (B,512,128)
    float *u2, *g2, *l2;       // (B,512,128), (B,128,4096), (B,128,256)
    float *u3, *h1, *h2, *h3, *sw3, *g3, *l3;
    float *f1, *f2;
    float *mrel, *msw;         // member records of the stage being processed: (B,16384,4), (B,16384,16)
    size_t bytes;
};

WsC carve_c(void* base, int B, int N) {
    WsC s{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = iq::align_up(off + bytes, 256);
        return reinterpret_cast<char*>(base) + o;
    };
    const size_t b = (size_t)B;
    s.inv1 = (float*)take(b * N * 4); s.inv2 = (float*)take(b * 512 * 4); s.inv3 = (float*)take(b * 128 * 4);
    s.fps1 = (int32_t*)take(b * 512 * 4); s.fps2 = (int32_t*)take(b * 128 * 4); s.nu1 = (int32_t*)take(b * 4);
    s.nx1 = (float*)take(b * 512 * 3 * 4); s.nx2 = (float*)take(b * 128 * 3 * 4);
    // k8 holds the keys of both kNN launches: the N points (sa1) and the 512 sa1 centroids (sa2) - the larger of the two
    s.k8 = (float*)take(b * (size_t)std::max((N + 31) / 32 * 32, 512) * 8 * 4); s.q8 = (float*)take(b * 512 * 8 * 4);
    s.idx1 = (int16_t*)take(b * 512 * 32 * 2); s.idx2 = (int16_t*)take(b * 128 * 64 * 2);
    s.g1 = (float*)take(b * 512 * 2048 * 4); s.l1 = (float*)take(b * 512 * 128 * 4);
    s.u2 = (float*)take(b * 512 * 128 * 4); s.g2 = (float*)take(b * 128 * 4096 * 4); s.l2 = (float*)take(b * 128 * 256 * 4);
    s.u3 = (float*)take(b * 128 * 256 * 4); s.h1 = (float*)take(b * 128 * 256 * 4); s.h2 = (float*)take(b * 128 * 512 * 4);
    s.h3 = (float*)take(b * 128 * 1024 * 4); s.sw3 = (float*)take(b * 128 * 16 * 4); s.g3 = (float*)take(b * 16384 * 4);
    s.l3 = (float*)take(b * 1024 * 4); s.f1 = (float*)take(b * 512 * 4); s.f2 = (float*)take(b * 256 * 4);
    s.mrel = (float*)take(b * 16384 * 4 * 4); s.msw = (float*)take(b * 16384 * 16 * 4);
    s.bytes = off;
    return s;
}

template <int K>
int launch_pc_knn(const float* keys, int nkeys, const float* queries, int S, WsC& s, int16_t* idx, int B, hipStream_t st,
                  const int32_t* n_unique = nullptr) {
    const int nkp = (nkeys + 31) / 32 * 32;
    hipLaunchKernelGGL(pc_pad8_kernel, dim3((B * nkp + 255) / 256), dim3(256), 0, st, keys, s.k8, B, nkeys, nkp);
    hipLaunchKernelGGL(pc_pad8_kernel, dim3((B * S + 255) / 256), dim3(256), 0, st, queries, s.q8, B, S, S);
    const int wpc = (S + 127) / 128;
    hipLaunchKernelGGL(pc_knn_kernel<K>, dim3((unsigned)((B + 7) / 8 * 8 * wpc)), dim3(kThreads), 0, st, s.k8, s.q8, idx, nkp, S, B,
                       wpc, n_unique);
    return iq::check_launch("pc_knn_kernel");
}

// members -> (rel, sw) records, then the grouped kernel.  xyz (B,N,3) member coordinates, new_xyz (B,S,3) centroids,
// idx (B,S,K), inv_density (B,N), U (B,N,ldu) or null.
int launch_pc_group(const iq_pointconv_sa& sa, const float* xyz, const float* new_xyz, const int16_t* idx, const float* inv_density,
                    const float* U, int ldu, float* out, WsC& s, int N, int S, int K, int B, hipStream_t st,
                    const int32_t* n_unique = nullptr, const void* l2_bf3 = nullptr, const void* l3_bf3 = nullptr) {
    IQ_REQUIRE((size_t)S * K <= 16384 && (S * K) % kMC == 0 && (K == 32 || K == 64), "pointconv group: S=%d K=%d", S, K);
    const size_t total = (size_t)B * S * K;
    const TinyNets nets{sa.densitynet, sa.weightnet};
    const unsigned mgrid = (unsigned)((total + kThreads - 1) / kThreads);
    if (K == 32) hipLaunchKernelGGL(pc_member_kernel<32>, dim3(mgrid), dim3(kThreads), 0, st, xyz, new_xyz, idx, inv_density, nets, s.mrel, s.msw, N, S, total, n_unique);
    else         hipLaunchKernelGGL(pc_member_kernel<64>, dim3(mgrid), dim3(kThreads), 0, st, xyz, new_xyz, idx, inv_density, nets, s.mrel, s.msw, N, S, total, n_unique);
    int rc = iq::check_launch("pc_member_kernel");
    if (rc) return rc;
    PcGroupArgs a{};
    a.mrel = s.mrel; a.msw = s.msw; a.U = U; a.ldu = ldu;
    a.w1x = sa.w1x;
    a.w2 = sa.l2.w; a.b2 = sa.l2.b; a.w3 = sa.l3.w; a.b3 = sa.l3.b;
    a.w2_bf3 = reinterpret_cast<const unsigned short*>(l2_bf3); a.w3_bf3 = reinterpret_cast<const unsigned short*>(l3_bf3);
    a.out = out; a.N = N; a.S = S; a.K = K; a.B = B; a.n_unique = n_unique;
    a.chunks_per_wg = 4;                                   // 256 members per workgroup: the prologue is amortised, the tail stays even
    const int chunks = S * K / kMC;
    a.wgs_per_cloud = (chunks + a.chunks_per_wg - 1) / a.chunks_per_wg;
    dim3 grid((unsigned)((B + 7) / 8 * 8 * a.wgs_per_cloud));
    const int c1 = sa.l2.cin, c2 = sa.l2.cout, c3 = sa.l3.cout;
    if (c1 == 64 && c2 == 64 && c3 == 128) hipLaunchKernelGGL((pc_group_kernel<64, 64, 128>), grid, dim3(kThreads), 0, st, a);
    else if (c1 == 128 && c2 == 128 && c3 == 256) {
        // every group runs all its K members (sums, not maxima: nothing is skipped); MFMA work = the two dense layers + the
        // contraction's 16x16x4 tiles
        iq::ProfileSpan dom(iq::kSlotDominant, st, 2.0 * (double)B * S * K * ((double)c1 * c2 + (double)c2 * c3 + 16.0 * c3));
        if (a.w2_bf3 && a.w3_bf3 && iq::tuning(iq::kTuneExperiment) != 56)   // 5 = 56: the fp32-MFMA kernel (A/B and tests)
        {
            if (!iq::tuning(iq::kTuneNoTranspose)) hipLaunchKernelGGL(pc_group_bf3_kernel<true>, grid, dim3(kThreads), 0, st, a);
            else hipLaunchKernelGGL(pc_group_bf3_kernel<false>, grid, dim3(kThreads), 0, st, a);
        }
        else
            hipLaunchKernelGGL((pc_group_kernel<128, 128, 256>), grid, dim3(kThreads), 0, st, a);
    }
    else return iq::fail(IQ_EUNSUPPORTED, "pointconv stage %d-%d-%d has no kernel instantiation", c1, c2, c3);
    return iq::check_launch("pc_group_kernel");
}

}  // namespace

extern "C" size_t iq_pointconv_workspace_bytes(int B, int N) {
    if (B < 0 || N < 0) return 0;
    return carve_c(nullptr, B, N).bytes;
}

namespace {

// groups from the source clouds' sorted neighbour lists instead of pc_knn_kernel (iq_pointconv_coalitions)
struct PcWalk {
    const int16_t* sorted;     // (nclouds, N+1, Nsl)
    const uint32_t* kept;      // (B, 32)
    const int16_t* mfirst;     // (B, 64)
    const int32_t* mcount;     // (B)
    int16_t* src1;             // (B, 512)
    int16_t* pos;              // (B, Npos)
    const int32_t* cloud_of;
    int nclouds, Nsl, Npos;
    const float* feat_tab;     // (nclouds, (N+1)^2, 128) sa1 pair table, or null
};

// the network on B materialised clouds xyz (B,N,3)
int run_pointconv(const iq_pointconv_weights* w, const float* xyz, float* logits, WsC& s, int B, int N, hipStream_t st,
                  const PcWalk* walk) {
    int rc;
    constexpr int S1 = 512, S2 = 128;
    bool fused_sa1 = false;

    // ---- sa1: 1024 -> 512 points, K = 32, 3 -> 64 -> 64 -> 128 ---------------------------------------------------
    {
        iq::ProfileSpan span(iq::kSlotPrepool, st);
        hipLaunchKernelGGL(pc_density_kernel, dim3((N + kThreads - 1) / kThreads, B), dim3(kThreads), (size_t)N * 16, st, xyz,
                           w->sa[0].bandwidth, s.inv1, N);
        if ((rc = iq::launch_fps(xyz, s.fps1, s.nu1, B, N, S1, st))) return rc;
        hipLaunchKernelGGL(pc_gather_xyz_kernel, dim3((B * S1 + 255) / 256), dim3(256), 0, st, xyz, 3, s.fps1, s.nx1, N, S1, B * S1);
        // a masked cloud has at most kept + 1 distinct locations: once FPS has used them up it returns index 0, so centroids
        // s >= nu1 are copies of centroid 0 and their groups (kNN, members, MLP, contraction, linear layer) are not computed
        if (walk) {
            hipLaunchKernelGGL(pc_walk1_kernel<32>, dim3(S1 / 64, B), dim3(64), 0, st, walk->sorted, walk->kept, walk->mfirst,
                               walk->mcount, s.fps1, s.nu1, walk->cloud_of, s.idx1, N, S1, walk->Nsl, walk->nclouds);
            if ((rc = iq::check_launch("pc_walk1_kernel"))) return rc;
        } else if ((rc = launch_pc_knn<32>(xyz, N, s.nx1, S1, s, s.idx1, B, st, s.nu1))) {
            return rc;
        }
        if (walk && walk->feat_tab) {   // members' weights as usual, the MLP rows from the pair table
            const size_t total = (size_t)B * S1 * 32;
            const TinyNets nets{w->sa[0].densitynet, w->sa[0].weightnet};
            hipLaunchKernelGGL(pc_member_kernel<32>, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, xyz, s.nx1,
                               s.idx1, s.inv1, nets, s.mrel, s.msw, N, S1, total, s.nu1);
            const iq_dense_layer& lin = w->sa[0].linear;
            if (lin.cin == 2048 && lin.cout == 128 && iq::tuning(iq::kTuneExperiment) != 31) {
                // contraction + the 2048 -> 128 layer in one kernel (5 = 31: the two-kernel form, A/B runs and tests)
                PcFusedArgs fa{walk->feat_tab, s.msw, s.idx1, s.fps1, walk->kept, s.nu1, walk->cloud_of, lin.w, lin.b, s.l1, N, S1, B,
                               walk->nclouds};
                hipLaunchKernelGGL(pc_tab_fused_kernel, dim3((unsigned)(8 * ((B + 7) / 8) * (S1 / 32))), dim3(kThreads), 0, st, fa);
                if ((rc = iq::check_launch("pc_tab_fused_kernel"))) return rc;
                fused_sa1 = true;
            } else {
                hipLaunchKernelGGL(pc_tab_group_kernel, dim3((unsigned)(((size_t)B * S1 + kThreads / 64 - 1) / (kThreads / 64))), dim3(kThreads),
                                   0, st, walk->feat_tab, s.msw, s.idx1, s.fps1, walk->kept, s.nu1, walk->cloud_of, s.g1, N, S1, B,
                                   walk->nclouds);
                if ((rc = iq::check_launch("pc_tab_group_kernel"))) return rc;
            }
        } else if ((rc = launch_pc_group(w->sa[0], xyz, s.nx1, s.idx1, s.inv1, nullptr, 0, s.g1, s, N, S1, 32, B, st, s.nu1))) {
            return rc;
        }
        if (!fused_sa1 && (rc = iq::launch_linear(s.g1, 2048, w->sa[0].linear, s.l1, 128, B * S1, 1, st, nullptr, s.nu1, S1))) return rc;
        hipLaunchKernelGGL(pc_fill_dup_rows_kernel, dim3(S1, B), dim3(64), 0, st, s.l1, S1, 128, s.nu1);
        if ((rc = iq::check_launch("pc_fill_dup_rows_kernel"))) return rc;
    }
    // ---- sa2: 512 -> 128 points, K = 64, 131 -> 128 -> 128 -> 256 --------------------------------------------------
    {
        iq::ProfileSpan span(iq::kSlotFstn, st);
        hipLaunchKernelGGL(pc_density_kernel, dim3((S1 + kThreads - 1) / kThreads, B), dim3(kThreads), (size_t)S1 * 16, st, s.nx1,
                           w->sa[1].bandwidth, s.inv2, S1);
        if ((rc = iq::launch_fps(s.nx1, s.fps2, nullptr, B, S1, S2, st))) return rc;
        hipLaunchKernelGGL(pc_gather_xyz_kernel, dim3((B * S2 + 255) / 256), dim3(256), 0, st, s.nx1, 3, s.fps2, s.nx2, S1, S2, B * S2);
        if (walk) {
            hipLaunchKernelGGL(pc_pos_kernel, dim3(B), dim3(64), 0, st, walk->kept, s.fps1, s.nu1, walk->src1, walk->pos, N, S1, walk->Npos);
            hipLaunchKernelGGL(pc_walk2_kernel<64>, dim3(S2 / 64, B), dim3(64), 0, st, walk->sorted, walk->src1, walk->pos, s.fps2, s.nu1,
                               walk->cloud_of, s.idx2, N, S1, S2, walk->Nsl, walk->Npos, walk->nclouds);
            if ((rc = iq::check_launch("pc_walk2_kernel"))) return rc;
        } else if ((rc = launch_pc_knn<64>(s.nx1, S1, s.nx2, S2, s, s.idx2, B, st))) {
            return rc;
        }
        if ((rc = iq::launch_linear(s.l1, 128, w->sa[1].u, s.u2, 128, B * S1, 0, st))) return rc;
        if ((rc = launch_pc_group(w->sa[1], s.nx1, s.nx2, s.idx2, s.inv2, s.u2, 128, s.g2, s, S1, S2, 64, B, st, nullptr, w->sa2_l2_bf3, w->sa2_l3_bf3))) return rc;
        if ((rc = iq::launch_linear(s.g2, 4096, w->sa[1].linear, s.l2, 256, B * S2, 1, st))) return rc;
    }
    // ---- sa3: group all 128 points, 259 -> 256 -> 512 -> 1024 ------------------------------------------------------
    {
        iq::ProfileSpan span(iq::kSlotTrunk, st);
        hipLaunchKernelGGL(pc_density_kernel, dim3(1, B), dim3(kThreads), (size_t)S2 * 16, st, s.nx2, w->sa[2].bandwidth, s.inv3, S2);
        if ((rc = iq::launch_linear(s.l2, 256, w->sa[2].u, s.u3, 256, B * S2, 0, st))) return rc;
        TinyNets nets{w->sa[2].densitynet, w->sa[2].weightnet};
        hipLaunchKernelGGL(pc_all_prepare_kernel, dim3(B), dim3(128), 0, st, s.nx2, s.inv3, nets, w->sa[2].w1x, s.u3, 256, s.h1,
                           s.sw3, S2);
        if ((rc = iq::check_launch("pc_all_prepare_kernel"))) return rc;
        if ((rc = iq::launch_linear(s.h1, 256, w->sa[2].l2, s.h2, 512, B * S2, 1, st))) return rc;
        if ((rc = iq::launch_linear(s.h2, 512, w->sa[2].l3, s.h3, 1024, B * S2, 1, st))) return rc;
        hipLaunchKernelGGL(pc_all_contract_kernel, dim3(1024 / kThreads, B), dim3(kThreads), 0, st, s.h3, s.sw3, s.g3, S2, 1024);
        if ((rc = iq::check_launch("pc_all_contract_kernel"))) return rc;
        // 16384 -> 1024 on only B rows: split K over workgroups (scratch: the h3 buffer, free by now, B*128*1024 floats)
        if ((rc = iq::launch_linear_splitk(s.g3, 16384, w->sa[2].linear, s.l3, 1024, B, 1, s.h3, (size_t)B * 128 * 1024, st))) return rc;
    }
    if ((rc = iq::launch_linear(s.l3, 1024, w->fc1, s.f1, 512, B, 1, st))) return rc;
    if ((rc = iq::launch_linear(s.f1, 512, w->fc2, s.f2, 256, B, 1, st))) return rc;
    if ((rc = iq::launch_linear(s.f2, 256, w->fc3, logits, w->fc3.cout, B, 0, st))) return rc;
    return IQ_OK;
}

// masked clouds: X[b][i] = kept ? clouds[c][i] : centers[c]   (models see what mask_data_batch would write)
__global__ void pc_mask_kernel(const float* __restrict__ clouds, const float* __restrict__ centers, const uint32_t* __restrict__ kept,
                               const int32_t* __restrict__ cloud_of, float* __restrict__ out, int N, int B, int nclouds) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * N) return;
    const int b = t / N, i = t - b * N;
    const int c = cloud_of ? cloud_of[b] : (nclouds == 1 ? 0 : b);
    const bool kp = (kept[(size_t)b * 32 + (i >> 5)] >> (i & 31)) & 1u;
    const float* src = kp ? clouds + ((size_t)c * N + i) * 3 : centers + (size_t)c * 3;
    out[(size_t)t * 3] = src[0]; out[(size_t)t * 3 + 1] = src[1]; out[(size_t)t * 3 + 2] = src[2];
}

struct WsW {   // coalition extras behind the forward's workspace
    float* X;            // (B,N,3)
    float *xs, *xxs, *dmat;
    int16_t* sorted;
    uint32_t* kept;
    int16_t* mfirst;
    int32_t* mcount;
    int16_t *src1, *pos;
    float *th1, *th2, *feat;   // pair table of sa1 (nc <= kTabClouds): layer outputs (n1^2, 64) x 2 of ONE cloud, rows (nc, n1^2, 128)
    size_t bytes;
};

// The pair table takes 0.54 GB per source cloud (+ 0.54 GB of layer outputs while ONE is built): 4.8 GB for the 8 poses a sweep
// launch carries - nothing on a 288 GB part, and the table path (fused kernel) runs sa1 at 11.1 ms per 3300 coalitions against
// 20.5 ms for the grouped MLP (round 3 stopped at 2 clouds).  Building a table costs ~0.4 ms per cloud and launch.
constexpr int kTabClouds = 8;

// Two parts.  `cloud` = what depends on the SOURCE clouds only (their padded rows, sorted neighbour lists, sa1 pair tables): it
// sits at the very start of the workspace, at offsets that depend on (nc, N) alone, so that a caller that evaluates chunk after
// chunk of coalitions of the same clouds can have it kept (iq_pointconv_coalitions_cached).  `rest` = per-coalition arrays and
// the scratch of the table build; it follows the forward's own workspace.
size_t carve_w_cloud(WsW& s, void* base, int nc, int N) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = iq::align_up(off + bytes, 256);
        return reinterpret_cast<char*>(base) + o;
    };
    const size_t Nsp = (size_t)(N + 1 + 31) / 32 * 32, Nsl = (size_t)(N + 1 + 7) / 8 * 8;
    s.xs = (float*)take((size_t)nc * Nsp * 8 * 4);
    s.xxs = (float*)take((size_t)nc * Nsp * 4);
    s.sorted = (int16_t*)take((size_t)nc * (N + 1) * Nsl * 2);
    if (nc <= kTabClouds) s.feat = (float*)take((size_t)nc * (size_t)(N + 1) * (N + 1) * 128 * 4);
    return off;
}

size_t carve_w_rest(WsW& s, void* base, int B, int nc, int N) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = iq::align_up(off + bytes, 256);
        return reinterpret_cast<char*>(base) + o;
    };
    const size_t Nsp = (size_t)(N + 1 + 31) / 32 * 32, Nsl = (size_t)(N + 1 + 7) / 8 * 8;
    s.X = (float*)take((size_t)B * N * 3 * 4);
    s.dmat = (float*)take((size_t)nc * Nsp * Nsp * 4);
    s.kept = (uint32_t*)take((size_t)B * 32 * 4);
    s.mfirst = (int16_t*)take((size_t)B * 64 * 2);
    s.mcount = (int32_t*)take((size_t)B * 4);
    s.src1 = (int16_t*)take((size_t)B * 512 * 2);
    s.pos = (int16_t*)take((size_t)B * Nsl * 2);
    if (nc <= kTabClouds) {
        const size_t rows = (size_t)(N + 1) * (N + 1);
        s.th1 = (float*)take(rows * 64 * 4);
        s.th2 = (float*)take(rows * 64 * 4);
    }
    return off;
}

}  // namespace

extern "C" int iq_pointconv_forward(const iq_pointconv_weights* w, const float* xyz, float* logits, void* workspace,
                                    size_t workspace_bytes, int B, int N, iq_stream_t stream) {
    IQ_REQUIRE(w && xyz && logits, "iq_pointconv_forward: null pointer");
    IQ_REQUIRE(B >= 0 && N >= 64 && N <= 4096, "iq_pointconv_forward: N=%d not in [64, 4096]", N);
    IQ_REQUIRE(w->sa[0].nsample == 32 && w->sa[1].nsample == 64, "iq_pointconv_forward: nsample must be 32 / 64");
    if (B == 0) return IQ_OK;
    const size_t need = carve_c(nullptr, B, N).bytes;
    if (!workspace || workspace_bytes < need)
        return iq::fail(IQ_EWORKSPACE, "iq_pointconv_forward: workspace %zu < %zu bytes", workspace_bytes, need);
    WsC s = carve_c(workspace, B, N);
    hipStream_t st = iq::as_stream(stream);
    iq::ProfileSpan call_span(iq::kSlotCall, st);
    return run_pointconv(w, xyz, logits, s, B, N, st, nullptr);
}

extern "C" size_t iq_pointconv_coalitions_workspace_bytes(int B, int nclouds, int N) {
    if (B < 0 || nclouds < 1 || N < 1) return 0;
    WsW t{};
    return carve_w_cloud(t, nullptr, nclouds, N) + iq::align_up(carve_c(nullptr, B, N).bytes, 256) + carve_w_rest(t, nullptr, B, nclouds, N);
}

extern "C" size_t iq_pointconv_tables_bytes(int nclouds, int N) {
    if (nclouds < 1 || N < 1) return 0;
    WsW t{};
    return carve_w_cloud(t, nullptr, nclouds, N);
}

// Logits of B coalitions given as region bit masks over nclouds source clouds (same call as iq_pointnet2_coalitions): the
// masked clouds are written into the workspace and run through the forward; when a few source clouds serve many coalitions
// (nclouds * 8 <= B, N <= 1024) the K-nearest groups of sa1 / sa2 come from the source clouds' sorted neighbour lists.
extern "C" int iq_pointconv_coalitions_cached(const iq_pointconv_weights* w, const float* clouds, const float* centers,
                                              const int32_t* region_id, const uint64_t* keep, const int32_t* cloud_of, float* logits,
                                              void* workspace, size_t workspace_bytes, int B, int nclouds, int N, int* tables_state,
                                              iq_stream_t stream);

extern "C" int iq_pointconv_coalitions(const iq_pointconv_weights* w, const float* clouds, const float* centers,
                                       const int32_t* region_id, const uint64_t* keep, const int32_t* cloud_of, float* logits,
                                       void* workspace, size_t workspace_bytes, int B, int nclouds, int N, iq_stream_t stream) {
    return iq_pointconv_coalitions_cached(w, clouds, centers, region_id, keep, cloud_of, logits, workspace, workspace_bytes, B, nclouds, N,
                                          nullptr, stream);
}

// The same call for a caller that evaluates several launches on the SAME source clouds (the chunks of an interaction setting, the
// batches of a Shapley pose): `tables_state` (host int, in / out) says which per-cloud structures the first
// iq_pointconv_tables_bytes(nclouds, N) bytes of `workspace` already hold - bit 0 the sorted neighbour lists, bit 1 the sa1 pair
// tables - for exactly these clouds, centres, nclouds and N (the caller's promise: same workspace base, contents untouched);
// what is missing is built and the bits are set.  Round 3 rebuilt both on every call (~1 GB of writes per source cloud).
extern "C" int iq_pointconv_coalitions_cached(const iq_pointconv_weights* w, const float* clouds, const float* centers,
                                              const int32_t* region_id, const uint64_t* keep, const int32_t* cloud_of, float* logits,
                                              void* workspace, size_t workspace_bytes, int B, int nclouds, int N, int* tables_state,
                                              iq_stream_t stream) {
    IQ_REQUIRE(B >= 0 && nclouds >= 1, "iq_pointconv_coalitions: B=%d nclouds=%d", B, nclouds);
    IQ_REQUIRE(w && clouds && centers && region_id && (B == 0 || (keep && logits)), "iq_pointconv_coalitions: null pointer");
    IQ_REQUIRE(N >= 64 && N <= kWalkMaxN, "iq_pointconv_coalitions: N=%d not in [64, %d]", N, kWalkMaxN);
    IQ_REQUIRE(cloud_of || nclouds == 1 || nclouds == B, "iq_pointconv_coalitions: cloud_of required when 1 < nclouds != B");
    IQ_REQUIRE(w->sa[0].nsample == 32 && w->sa[1].nsample == 64, "iq_pointconv_coalitions: nsample must be 32 / 64");
    if (B == 0) return IQ_OK;
    WsW t{};
    const size_t cloud_bytes = carve_w_cloud(t, workspace, nclouds, N);
    const size_t base_bytes = iq::align_up(carve_c(nullptr, B, N).bytes, 256);
    const size_t need = cloud_bytes + base_bytes + carve_w_rest(t, nullptr, B, nclouds, N);
    if (!workspace || workspace_bytes < need)
        return iq::fail(IQ_EWORKSPACE, "iq_pointconv_coalitions: workspace %zu < %zu bytes", workspace_bytes, need);
    WsC s = carve_c(reinterpret_cast<char*>(workspace) + cloud_bytes, B, N);
    carve_w_rest(t, reinterpret_cast<char*>(workspace) + cloud_bytes + base_bytes, B, nclouds, N);
    int have = tables_state ? (*tables_state & 3) : 0;    // bit 0: sorted lists, bit 1: pair tables (see above)
    const int force = tables_state ? ((*tables_state >> 2) & 3) : 0;
    hipStream_t st = iq::as_stream(stream);
    int rc;
    iq::ProfileSpan call_span(iq::kSlotCall, st);
    const int Nsp = (N + 1 + 31) / 32 * 32, Nsl = (N + 1 + 7) / 8 * 8;
    hipLaunchKernelGGL(pc_coal_prep_kernel, dim3(B), dim3(64), 0, st, region_id, keep, cloud_of, t.kept, t.mfirst, t.mcount, N, nclouds);
    hipLaunchKernelGGL(pc_mask_kernel, dim3((unsigned)(((size_t)B * N + 255) / 256)), dim3(256), 0, st, clouds, centers, t.kept, cloud_of,
                       t.X, N, B, nclouds);
    if ((rc = iq::check_launch("pc_mask_kernel"))) return rc;
    // the lists for up to 8 source clouds whatever B is: the two ways of forming a group differ in the order of the sum over its
    // members, and a coalition's logits must not depend on how many others share its launch (5 = 14: pc_knn_kernel)
    // tables_state bits 2-3 (in): 1 = always the list walk, 2 = never - a caller that splits one batch over several launches names
    // the path once, so that every launch (a short tail included) forms its groups the same way; 0 = decide from this launch
    const bool use_walk = (force == 1 || (force == 0 && (nclouds <= 8 || (long long)nclouds * 8 <= B))) && iq::tuning(iq::kTuneExperiment) != 14;
    PcWalk walk{};
    if (use_walk) {
        if (!(have & 1)) {
            hipLaunchKernelGGL(sl_rows_kernel, dim3((Nsp + 255) / 256, nclouds), dim3(256), 0, st, clouds, centers, t.xs, t.xxs, N, Nsp);
            hipLaunchKernelGGL(sl_dist_kernel<1>, dim3(Nsp / 32, nclouds), dim3(64), 0, st, t.xs, t.xxs, t.dmat, Nsp);
            hipLaunchKernelGGL(sl_sort_kernel, dim3(N + 1, nclouds), dim3(256), 0, st, t.dmat, t.sorted, N, Nsp, Nsl);
            if ((rc = iq::check_launch("sl_sort_kernel"))) return rc;
            have |= 1;
        }
        walk = PcWalk{t.sorted, t.kept, t.mfirst, t.mcount, t.src1, t.pos, cloud_of, nclouds, Nsl, Nsl, nullptr};
        const iq_pointconv_sa& sa = w->sa[0];
        // (whatever B is: a coalition's logits must not depend on how many others share its launch)
        if (nclouds <= kTabClouds && sa.l2.cin == 64 && sa.l2.cout == 64 && sa.l3.cout == 128 &&
            iq::tuning(iq::kTuneExperiment) != 15) {   // 5 = 15: the grouped MLP (A/B runs, tests)
            const int n1 = N + 1;
            const size_t rows = (size_t)n1 * n1;
            for (int c = 0; c < nclouds && !(have & 2); ++c) {
                hipLaunchKernelGGL(pc_tab_l1_kernel, dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0, st, t.xs + (size_t)c * Nsp * 8,
                                   sa.w1x, t.th1, n1);
                if ((rc = iq::check_launch("pc_tab_l1_kernel"))) return rc;
                if ((rc = iq::launch_linear(t.th1, 64, sa.l2, t.th2, 64, (int)rows, 1, st))) return rc;
                if ((rc = iq::launch_linear(t.th2, 64, sa.l3, t.feat + (size_t)c * rows * 128, 128, (int)rows, 1, st))) return rc;
            }
            have |= 2;
            walk.feat_tab = t.feat;
        }
    }
    if (tables_state) *tables_state = have | (force << 2);
    return run_pointconv(w, t.X, logits, s, B, N, st, use_walk ? &walk : nullptr);
}
