// PointNet++ MSG forward (models/pointnet2.py:244-276) on materialised (masked) clouds.
//
//   sa1: FPS 1024->512, ball query x3 (r .1/.2/.4, K 16/32/128), grouped MLP 3->C1->C2->C3, max over K
//   sa2: FPS 512->128, ball query x3 (r .2/.4/.8, K 32/64/128), grouped MLP 323->C1->C2->C3, max over K
//   sa3: group-all MLP 643->256->512->1024, max over the 128 points; FC 512/256/10
//
// Exact restructurings:
//   * layer 1 of sa2 splits linearly: W.[f_p ; x_p - c] = (W_f f_p + b)  +  W_x (x_p - c).  The first
//     term depends on the member point only and is one batched GEMM over the 512 points (U); the
//     grouped kernel gathers U rows and adds the 3-term xyz part.  16-128x fewer MACs for that layer.
//   * duplicate centroids: once FPS has exhausted the distinct locations it returns index 0 forever
//     (iq_geom.hip), so groups s >= n_unique are copies of group 0 and are filled, not recomputed.
// Grouped kernel: 64-row chunks -> LDS act1 -> MFMA C1->C2 -> LDS act2 -> MFMA C2->C3 -> group max in
// registers.  Index-valued steps (ball query) use explicitly rounded arithmetic in the reference's
// evaluation order.
#include "iq_common.h"
#include "iq_mfma.h"
#include "iq_profile.h"

// Index-valued results depend on individually rounded operations: forbid the compiler from fusing
// a*b+c into an fma anywhere in this file (explicit fmaf / MFMA calls are unaffected).
#pragma clang fp contract(off)

namespace {

constexpr int kThreads = 256;
constexpr int kMC = 64;

// ---- gather: out[b][s][:] = xyz[b][idx[b][s]][:] ------------------------------------------------
__global__ void gather_xyz_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ idx,
                                  float* __restrict__ out, int ldo, int N, int S, int total) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int b = t / S;
    const float* src = xyz + ((size_t)b * N + idx[t]) * 3;
    float* dst = out + (size_t)t * ldo;
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
}

// ---- ball query (models/pointnet2.py:70-91), up to 3 radii in one pass ----------------------------
// One lane per centroid, points (x,y,z,|p|^2) broadcast from LDS in index order; d = ((-2 c.p) + |c|^2)
// + |p|^2 with the matmul row as an fma chain; a point is a member iff d <= (float)(r*r).
struct BallArgs {
    const float* xyz;      // (B,N,3)
    const float* new_xyz;  // (B,S,ldc) centroid coordinates (first 3 floats of each row)
    int ldc;
    int N, S, nr;
    float r2[3];
    int K[3];
    void* idx[3];          // (B,S,K) each
};

template <typename IdxT>
__global__ __launch_bounds__(kThreads) void ball_query_kernel(BallArgs a) {
    extern __shared__ float pts[];  // N x 4
    const int b = blockIdx.y;
    const float* src = a.xyz + (size_t)b * a.N * 3;
    for (int p = threadIdx.x; p < a.N; p += kThreads) {
        const float x = src[p * 3], y = src[p * 3 + 1], z = src[p * 3 + 2];
        pts[p * 4] = x; pts[p * 4 + 1] = y; pts[p * 4 + 2] = z;
        pts[p * 4 + 3] = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
    }
    __syncthreads();
    const int s = blockIdx.x * kThreads + threadIdx.x;
    const bool live = s < a.S;
    const float* c = a.new_xyz + ((size_t)b * a.S + (live ? s : 0)) * a.ldc;
    const float cx = c[0], cy = c[1], cz = c[2];
    const float sc = __fadd_rn(__fadd_rn(__fmul_rn(cx, cx), __fmul_rn(cy, cy)), __fmul_rn(cz, cz));
    int cnt[3] = {0, 0, 0};
    IdxT first[3] = {0, 0, 0};
    IdxT* out[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)
        out[q] = q < a.nr ? reinterpret_cast<IdxT*>(a.idx[q]) + ((size_t)b * a.S + (live ? s : 0)) * a.K[q] : nullptr;
    for (int p = 0; p < a.N; ++p) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(pts + p * 4);
        float dot = __fmul_rn(cx, v[0]);
        dot = __fmaf_rn(cy, v[1], dot);
        dot = __fmaf_rn(cz, v[2], dot);
        const float d = __fadd_rn(__fadd_rn(__fmul_rn(-2.f, dot), sc), v[3]);
        bool open = false;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (q < a.nr && cnt[q] < a.K[q]) {
                if (live && !(d > a.r2[q])) {
                    if (cnt[q] == 0) first[q] = (IdxT)p;
                    out[q][cnt[q]++] = (IdxT)p;
                }
                open = open || (cnt[q] < a.K[q]);
            }
        }
        if (!__any(open && live)) break;
    }
    if (!live) return;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        if (q >= a.nr) continue;
        for (int j = cnt[q]; j < a.K[q]; ++j) out[q][j] = first[q];  // pad with the first hit (:88-90)
    }
}

// ---- grouped MLP + max ----------------------------------------------------------------------------
struct GroupArgs {
    const float* xyz;        // (B,N,ldx) member coordinates (first 3 floats of each row)
    int ldx;
    const float* new_xyz;    // (B,S,ldc) centroids
    int ldc;
    const int16_t* idx;      // (B,S,K)
    const float* U;          // (B,N,ldu) per-point part of layer 1 (bias included) or null
    int ldu;
    const float* w1x;        // [C1][4] = (wx0, wx1, wx2, bias)
    const float* w2; const float* b2;  // packed C1->C2
    const float* w3; const float* b3;  // packed C2->C3
    float* out;              // (B,S,ldo) at the scale's column offset
    int ldo;
    const int32_t* n_unique; // (B) or null
    int N, S, K, groups_per_wg;
};

template <int C1, int C2, int C3>
__global__ __launch_bounds__(kThreads, 2) void pn2_group_kernel(GroupArgs a) {
    constexpr int LD1 = C1 + 4, LD2 = C2 + 4;
    constexpr int KB1 = C1 / 8, KB2 = C2 / 8, NT2 = C2 / 32, NT3 = C3 / 32;
    constexpr int NQ3 = NT3 >= 4 ? NT3 / 4 : 1;
    __shared__ __attribute__((aligned(16))) float act1[kMC * LD1];
    __shared__ __attribute__((aligned(16))) float act2[kMC * LD2];
    __shared__ __attribute__((aligned(16))) float rel[kMC * 4];  // dx,dy,dz, member index (bits)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int K = a.K;
    const int g0 = blockIdx.x * a.groups_per_wg;
    const int gend_all = min(a.S, g0 + a.groups_per_wg);
    const int gend = a.n_unique ? min(gend_all, a.n_unique[b]) : gend_all;  // computed groups
    if (g0 >= gend) return;
    const int rows_total = (gend - g0) * K;
    const int nchunks = (rows_total + kMC - 1) / kMC;

    const int fl = lane & 31, fh = lane >> 5;
    const float* a1base = act1 + fl * LD1 + 4 * fh;
    const float* a2base = act2 + fl * LD2 + 4 * fh;
    float* c2base = act2 + (4 * fh) * LD2 + fl;
    float runmax[NQ3];
#pragma unroll
    for (int q = 0; q < NQ3; ++q) runmax[q] = -INFINITY;

    for (int ch = 0; ch < nchunks; ++ch) {
        const int row0 = ch * kMC;
        // ---- stage 0a: member -> relative coordinates (x_p - c, rounded like the reference's `-=`) --
        if (tid < kMC) {
            int rr = row0 + tid;
            if (rr >= rows_total) rr = rows_total - 1;  // padding rows replicate a valid row
            const int g = g0 + rr / K, k = rr - (rr / K) * K;
            const int p = a.idx[((size_t)b * a.S + g) * K + k];
            const float* x = a.xyz + ((size_t)b * a.N + p) * a.ldx;
            const float* c = a.new_xyz + ((size_t)b * a.S + g) * a.ldc;
            f32x4 v;
            v[0] = __fsub_rn(x[0], c[0]); v[1] = __fsub_rn(x[1], c[1]); v[2] = __fsub_rn(x[2], c[2]);
            v[3] = __int_as_float(p);
            *reinterpret_cast<f32x4*>(rel + tid * 4) = v;
        }
        __syncthreads();  // also: previous chunk's readers of act1/act2 are done
        // ---- stage 0b: layer 1 -> act1 -------------------------------------------------------------
        {
            const int chn = tid % C1;
            const f32x4 w = *reinterpret_cast<const f32x4*>(a.w1x + chn * 4);
            for (int r = tid / C1; r < kMC; r += kThreads / C1) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(rel + r * 4);
                float h = fmaf(w[2], v[2], fmaf(w[1], v[1], w[0] * v[0])) + w[3];
                if (a.U) h += a.U[((size_t)b * a.N + __float_as_int(v[3])) * a.ldu + chn];
                act1[r * LD1 + chn] = fmaxf(h, 0.f);
            }
        }
        __syncthreads();
        // ---- layer 2: C1 -> C2 (+bn, relu) -> act2 -------------------------------------------------
        if (NT2 >= 4) {
            for (int nt = wave; nt < NT2; nt += 4) {
                f32x16 acc0 = {0}, acc1 = {0};
                const float* wq = a.w2 + (size_t)nt * KB1 * 256;
#pragma unroll 4
                for (int kb = 0; kb < KB1; ++kb) {
                    const f32x4 bw = glb_b(wq + kb * 256, lane);
                    acc0 = mfma4(lds_frag<LD1>(a1base, 0, kb), bw, acc0);
                    acc1 = mfma4(lds_frag<LD1>(a1base, 1, kb), bw, acc1);
                }
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
                const float* wq = a.w2 + (size_t)nt * KB1 * 256;
#pragma unroll 4
                for (int kb = 0; kb < KB1; ++kb)
                    acc = mfma4(lds_frag<LD1>(a1base + mt * 32 * LD1, 0, kb), glb_b(wq + kb * 256, lane), acc);
                const float bias = a.b2[nt * 32 + fl];
                float* dst = c2base + mt * 32 * LD2 + nt * 32;
#pragma unroll
                for (int i = 0; i < 16; ++i) dst[c_row_i(i) * LD2] = fmaxf(acc[i] + bias, 0.f);
            }
        }
        __syncthreads();
        // ---- layer 3: C2 -> C3 (+bn, relu), max over each group's K rows ---------------------------
        if (NT3 >= 4) {
#pragma unroll
            for (int q = 0; q < NQ3; ++q) {
                const int nt = q * 4 + wave;
                f32x16 acc0 = {0}, acc1 = {0};
                const float* wq = a.w3 + (size_t)nt * KB2 * 256;
#pragma unroll 4
                for (int kb = 0; kb < KB2; ++kb) {
                    const f32x4 bw = glb_b(wq + kb * 256, lane);
                    acc0 = mfma4(lds_frag<LD2>(a2base, 0, kb), bw, acc0);
                    acc1 = mfma4(lds_frag<LD2>(a2base, 1, kb), bw, acc1);
                }
                const float bias = a.b3[nt * 32 + fl];
                float* orow = a.out + (size_t)b * a.S * a.ldo + nt * 32 + fl;
                if (K >= 64) {
                    // one group per chunk (K = 64) or per two chunks (K = 128)
                    float m = fmaxf(max16(acc0), max16(acc1));
                    m = fmaxf(m, __shfl_xor(m, 32));
                    runmax[q] = fmaxf(runmax[q], m);
                    const int rows_done = row0 + kMC;
                    if (rows_done % K == 0 || ch == nchunks - 1) {
                        const int g = g0 + (rows_done - 1) / K;
                        if (g < gend && fh == 0) orow[(size_t)g * a.ldo] = fmaxf(runmax[q] + bias, 0.f);
                        runmax[q] = -INFINITY;
                    }
                } else {  // K = 32: one group per m-tile
                    float m0 = max16(acc0), m1 = max16(acc1);
                    m0 = fmaxf(m0, __shfl_xor(m0, 32));
                    m1 = fmaxf(m1, __shfl_xor(m1, 32));
                    const int ga = g0 + row0 / 32;
                    if (fh == 0) {
                        if (ga < gend) orow[(size_t)ga * a.ldo] = fmaxf(m0 + bias, 0.f);
                        if (ga + 1 < gend) orow[(size_t)(ga + 1) * a.ldo] = fmaxf(m1 + bias, 0.f);
                    }
                }
            }
        } else {  // few n-tiles (C3 = 64): one (m-tile, n-tile) pair per wave; K is 16 or 32 here
            for (int t = wave; t < 2 * NT3; t += 4) {
                const int mt = t / NT3, nt = t - mt * NT3;
                f32x16 acc = {0};
                const float* wq = a.w3 + (size_t)nt * KB2 * 256;
#pragma unroll 4
                for (int kb = 0; kb < KB2; ++kb)
                    acc = mfma4(lds_frag<LD2>(a2base + mt * 32 * LD2, 0, kb), glb_b(wq + kb * 256, lane), acc);
                const float bias = a.b3[nt * 32 + fl];
                float* orow = a.out + (size_t)b * a.S * a.ldo + nt * 32 + fl;
                if (K == 16) {  // rows 0-15 live in registers 0-7, rows 16-31 in registers 8-15
                    float lo = fmaxf(fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3])),
                                     fmaxf(fmaxf(acc[4], acc[5]), fmaxf(acc[6], acc[7])));
                    float hi = fmaxf(fmaxf(fmaxf(acc[8], acc[9]), fmaxf(acc[10], acc[11])),
                                     fmaxf(fmaxf(acc[12], acc[13]), fmaxf(acc[14], acc[15])));
                    lo = fmaxf(lo, __shfl_xor(lo, 32));
                    hi = fmaxf(hi, __shfl_xor(hi, 32));
                    const int ga = g0 + (row0 + mt * 32) / 16;
                    if (fh == 0) {
                        if (ga < gend) orow[(size_t)ga * a.ldo] = fmaxf(lo + bias, 0.f);
                        if (ga + 1 < gend) orow[(size_t)(ga + 1) * a.ldo] = fmaxf(hi + bias, 0.f);
                    }
                } else {        // K = 32
                    float m = max16(acc);
                    m = fmaxf(m, __shfl_xor(m, 32));
                    const int ga = g0 + (row0 + mt * 32) / 32;
                    if (fh == 0 && ga < gend) orow[(size_t)ga * a.ldo] = fmaxf(m + bias, 0.f);
                }
            }
        }
        // next chunk's first barrier (after stage 0a) orders these LDS reads before act1/act2 are rewritten
    }
}

// rows s >= n_unique[b] := row 0 (duplicate centroids), columns [c0, c0+ncols)
__global__ void fill_dup_rows_kernel(float* __restrict__ out, int ldo, int S, int c0, int ncols,
                                     const int32_t* __restrict__ n_unique) {
    const int b = blockIdx.y;
    const int nu = n_unique[b];
    const int s = nu + blockIdx.x;
    if (s >= S) return;
    const float* src = out + (size_t)b * S * ldo + c0;
    float* dst = out + ((size_t)b * S + s) * ldo + c0;
    for (int c = threadIdx.x; c < ncols; c += blockDim.x) dst[c] = src[c];
}

// zero the padding columns [c0, ld) of every row
__global__ void zero_cols_kernel(float* __restrict__ buf, int ld, int c0, int rows) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    for (int c = c0; c < ld; ++c) buf[(size_t)r * ld + c] = 0.f;
}

// out[b][c] = max over the S rows of in[b][s][c]
__global__ void colmax_kernel(const float* __restrict__ in, float* __restrict__ out, int S, int C) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float* p = in + (size_t)b * S * C + c;
    float m = -INFINITY;
    for (int s = 0; s < S; ++s) m = fmaxf(m, p[(size_t)s * C]);
    out[(size_t)b * C + c] = m;
}

template <int C1, int C2, int C3>
int launch_group_t(const GroupArgs& a, int B, hipStream_t st) {
    dim3 grid((a.S + a.groups_per_wg - 1) / a.groups_per_wg, B);
    hipLaunchKernelGGL((pn2_group_kernel<C1, C2, C3>), grid, dim3(kThreads), 0, st, a);
    return iq::check_launch("pn2_group_kernel");
}

int launch_group(const iq_pn2_scale& sc, GroupArgs a, int B, hipStream_t st) {
    a.w1x = sc.w1x;
    a.w2 = sc.l2.w; a.b2 = sc.l2.b;
    a.w3 = sc.l3.w; a.b3 = sc.l3.b;
    a.K = sc.nsample;
    // >= 512 rows per workgroup, whole groups
    a.groups_per_wg = max(1, 512 / a.K);
    const int c1 = sc.l2.cin, c2 = sc.l2.cout, c3 = sc.l3.cout;
    IQ_REQUIRE(sc.l3.cin == c2, "pointnet2 scale: layer sizes do not chain");
    IQ_REQUIRE(a.K == 16 || a.K == 32 || a.K == 64 || a.K == 128, "pointnet2 scale: nsample %d unsupported", a.K);
    if (c1 == 32 && c2 == 32 && c3 == 64) return launch_group_t<32, 32, 64>(a, B, st);
    if (c1 == 64 && c2 == 64 && c3 == 128) return launch_group_t<64, 64, 128>(a, B, st);
    if (c1 == 64 && c2 == 96 && c3 == 128) return launch_group_t<64, 96, 128>(a, B, st);
    if (c1 == 128 && c2 == 128 && c3 == 256) return launch_group_t<128, 128, 256>(a, B, st);
    return iq::fail(IQ_EUNSUPPORTED, "pointnet2 scale %d-%d-%d has no kernel instantiation", c1, c2, c3);
}

int launch_ball(const float* xyz, const float* new_xyz, int ldc, const iq_pn2_scale* sc, int nr, int16_t* const* idx,
                int B, int N, int S, hipStream_t st) {
    BallArgs a{};
    a.xyz = xyz; a.new_xyz = new_xyz; a.ldc = ldc; a.N = N; a.S = S; a.nr = nr;
    for (int q = 0; q < nr; ++q) {
        a.r2[q] = (float)((double)sc[q].radius * (double)sc[q].radius);  // python float r**2, cast by the comparison
        a.K[q] = sc[q].nsample;
        a.idx[q] = idx[q];
    }
    hipLaunchKernelGGL(ball_query_kernel<int16_t>, dim3((S + kThreads - 1) / kThreads, B), dim3(kThreads),
                       (size_t)N * 4 * sizeof(float), st, a);
    return iq::check_launch("ball_query_kernel");
}

struct Ws2 {
    int32_t *fps1, *nu1, *fps2, *nu2;
    float *nx1;            // (B,512,3)
    int16_t* idx1[3];      // (B,512,K)
    float* l1;             // (B,512,320)
    float* U;              // (B,512,320)
    int16_t* idx2[3];      // (B,128,K)
    float* a3;             // (B,128,648): [xyz(3), feats(640), 0-pad]
    float *h1, *h2, *h3;   // (B*128, 256/512/1024)
    float *g, *f1, *f2;    // (B,1024/512/256)
    size_t bytes;
};

Ws2 carve2(void* base, int B, const iq_pointnet2_weights* w) {
    Ws2 s{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = iq::align_up(off + bytes, 256);
        return reinterpret_cast<char*>(base) + o;
    };
    const size_t b = (size_t)B;
    s.fps1 = (int32_t*)take(b * 512 * 4); s.nu1 = (int32_t*)take(b * 4);
    s.fps2 = (int32_t*)take(b * 128 * 4); s.nu2 = (int32_t*)take(b * 4);
    s.nx1 = (float*)take(b * 512 * 3 * 4);
    for (int q = 0; q < 3; ++q) s.idx1[q] = (int16_t*)take(b * 512 * (w ? w->sa1[q].nsample : 128) * 2);
    s.l1 = (float*)take(b * 512 * 320 * 4);
    s.U = (float*)take(b * 512 * 320 * 4);
    for (int q = 0; q < 3; ++q) s.idx2[q] = (int16_t*)take(b * 128 * (w ? w->sa2[q].nsample : 128) * 2);
    s.a3 = (float*)take(b * 128 * 648 * 4);
    s.h1 = (float*)take(b * 128 * 256 * 4);
    s.h2 = (float*)take(b * 128 * 512 * 4);
    s.h3 = (float*)take(b * 128 * 1024 * 4);
    s.g = (float*)take(b * 1024 * 4); s.f1 = (float*)take(b * 512 * 4); s.f2 = (float*)take(b * 256 * 4);
    s.bytes = off;
    return s;
}

}  // namespace

extern "C" size_t iq_pointnet2_workspace_bytes(int B) {
    if (B < 0) return 0;
    return carve2(nullptr, B, nullptr).bytes;
}

extern "C" int iq_ball_query(const float* xyz, const float* new_xyz, float radius, int K, int32_t* idx, int B, int N,
                             int S, iq_stream_t stream) {
    IQ_REQUIRE(B >= 0 && N >= 1 && N <= 4096 && S >= 1 && K >= 1, "iq_ball_query: B=%d N=%d S=%d K=%d", B, N, S, K);
    if (B == 0) return IQ_OK;
    IQ_REQUIRE(xyz && new_xyz && idx, "iq_ball_query: null pointer");
    BallArgs a{};
    a.xyz = xyz; a.new_xyz = new_xyz; a.ldc = 3; a.N = N; a.S = S; a.nr = 1;
    a.r2[0] = (float)((double)radius * (double)radius);
    a.K[0] = K;
    a.idx[0] = idx;
    hipLaunchKernelGGL(ball_query_kernel<int32_t>, dim3((S + kThreads - 1) / kThreads, B), dim3(kThreads),
                       (size_t)N * 4 * sizeof(float), iq::as_stream(stream), a);
    return iq::check_launch("ball_query_kernel");
}

extern "C" int iq_pointnet2_forward(const iq_pointnet2_weights* w, const float* xyz, float* logits, void* workspace,
                                    size_t workspace_bytes, int B, int N, iq_stream_t stream) {
    IQ_REQUIRE(w && xyz && logits, "iq_pointnet2_forward: null pointer");
    IQ_REQUIRE(B >= 0 && N >= 1 && N <= 4096, "iq_pointnet2_forward: B=%d N=%d", B, N);
    if (B == 0) return IQ_OK;
    const size_t need = carve2(nullptr, B, w).bytes;
    if (!workspace || workspace_bytes < need)
        return iq::fail(IQ_EWORKSPACE, "iq_pointnet2_forward: workspace %zu < %zu bytes", workspace_bytes, need);
    Ws2 s = carve2(workspace, B, w);
    hipStream_t st = iq::as_stream(stream);
    int rc;
    iq::ProfileSpan call_span(iq::kSlotCall, st);
    constexpr int S1 = 512, S2 = 128, F1 = 320, LD3 = 648;

    // ---- sa1 ---------------------------------------------------------------------------------------
    if ((rc = iq::launch_fps(xyz, s.fps1, s.nu1, B, N, S1, st))) return rc;
    hipLaunchKernelGGL(gather_xyz_kernel, dim3((B * S1 + 255) / 256), dim3(256), 0, st, xyz, s.fps1, s.nx1, 3, N, S1, B * S1);
    if ((rc = iq::check_launch("gather_xyz_kernel"))) return rc;
    if ((rc = launch_ball(xyz, s.nx1, 3, w->sa1, 3, s.idx1, B, N, S1, st))) return rc;
    int col = 0;
    for (int q = 0; q < 3; ++q) {
        GroupArgs a{};
        a.xyz = xyz; a.ldx = 3; a.new_xyz = s.nx1; a.ldc = 3; a.idx = s.idx1[q];
        a.U = nullptr; a.ldu = 0;
        a.out = s.l1 + col; a.ldo = F1; a.n_unique = s.nu1; a.N = N; a.S = S1;
        iq::ProfileSpan span(iq::kSlotPrepool, st);
        if ((rc = launch_group(w->sa1[q], a, B, st))) return rc;
        col += w->sa1[q].l3.cout;
    }
    IQ_REQUIRE(col == F1, "pointnet2: sa1 output channels %d != 320", col);
    hipLaunchKernelGGL(fill_dup_rows_kernel, dim3(S1, B), dim3(64), 0, st, s.l1, F1, S1, 0, F1, s.nu1);
    if ((rc = iq::check_launch("fill_dup_rows_kernel"))) return rc;

    // ---- sa2 ---------------------------------------------------------------------------------------
    if ((rc = iq::launch_fps(s.nx1, s.fps2, s.nu2, B, S1, S2, st))) return rc;
    hipLaunchKernelGGL(gather_xyz_kernel, dim3((B * S2 + 255) / 256), dim3(256), 0, st, s.nx1, s.fps2, s.a3, LD3, S1, S2, B * S2);
    hipLaunchKernelGGL(zero_cols_kernel, dim3((B * S2 + 255) / 256), dim3(256), 0, st, s.a3, LD3, 643, B * S2);
    if ((rc = iq::check_launch("gather/zero"))) return rc;
    if ((rc = launch_ball(s.nx1, s.a3, LD3, w->sa2, 3, s.idx2, B, S1, S2, st))) return rc;
    if ((rc = iq::launch_linear(s.l1, F1, w->sa2_u, s.U, F1, B * S1, 0, st))) return rc;  // U = W_f f_p + b (all scales)
    col = 0;
    int ucol = 0;
    for (int q = 0; q < 3; ++q) {
        GroupArgs a{};
        a.xyz = s.nx1; a.ldx = 3; a.new_xyz = s.a3; a.ldc = LD3; a.idx = s.idx2[q];
        a.U = s.U + ucol; a.ldu = F1;
        a.out = s.a3 + 3 + col; a.ldo = LD3; a.n_unique = s.nu2; a.N = S1; a.S = S2;
        iq::ProfileSpan span(iq::kSlotFstn, st);
        if ((rc = launch_group(w->sa2[q], a, B, st))) return rc;
        col += w->sa2[q].l3.cout;
        ucol += w->sa2[q].l2.cin;
    }
    IQ_REQUIRE(col == 640 && ucol == F1, "pointnet2: sa2 channels %d / %d", col, ucol);
    hipLaunchKernelGGL(fill_dup_rows_kernel, dim3(S2, B), dim3(64), 0, st, s.a3, LD3, S2, 3, 640, s.nu2);
    if ((rc = iq::check_launch("fill_dup_rows_kernel"))) return rc;

    // ---- sa3 (group all) + head --------------------------------------------------------------------
    {
        iq::ProfileSpan span(iq::kSlotTrunk, st);
        if ((rc = iq::launch_linear(s.a3, LD3, w->sa3_l1, s.h1, 256, B * S2, 1, st))) return rc;
        if ((rc = iq::launch_linear(s.h1, 256, w->sa3_l2, s.h2, 512, B * S2, 1, st))) return rc;
        if ((rc = iq::launch_linear(s.h2, 512, w->sa3_l3, s.h3, 1024, B * S2, 1, st))) return rc;
    }
    hipLaunchKernelGGL(colmax_kernel, dim3(1024 / 256, B), dim3(256), 0, st, s.h3, s.g, S2, 1024);
    if ((rc = iq::check_launch("colmax_kernel"))) return rc;
    if ((rc = iq::launch_linear(s.g, 1024, w->fc1, s.f1, 512, B, 1, st))) return rc;
    if ((rc = iq::launch_linear(s.f1, 512, w->fc2, s.f2, 256, B, 1, st))) return rc;
    if ((rc = iq::launch_linear(s.f2, 256, w->fc3, logits, w->fc3.cout, B, 0, st))) return rc;
    return IQ_OK;
}
