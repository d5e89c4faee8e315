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
//   * padded members: a ball usually holds far fewer than K points (K = 128 at r = 0.4 sees ~65 of 1024
//     points) and the reference pads the list with copies of the first hit.  Copies cannot change a
//     max, so every group is processed as ceil(count/16) blocks of 16 rows instead of K rows; the
//     blocks of a cloud are compacted (pn2_blocks_kernel) and each block's column max is merged into
//     the group's output with an integer atomicMax (post-ReLU values are >= 0, so float order = int order).
// Grouped kernel: 64-row chunks -> LDS act1 -> MFMA C1->C2 -> LDS act2 -> MFMA C2->C3 -> group max in
// registers.  Index-valued steps (ball query) use explicitly rounded arithmetic in the reference's
// evaluation order.
#include <algorithm>
#include <vector>

#include "iq_common.h"
#include "iq_bf3.h"
#include "iq_mfma.h"
#include "iq_profile.h"

// Index-valued results depend on individually rounded operations: forbid the compiler from fusing
// a*b+c into an fma anywhere in this file (explicit fmaf / MFMA calls are unaffected).
#pragma clang fp contract(off)

namespace {

constexpr int kThreads = 256;

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
    int32_t* cnt[3];       // (B,S) number of true hits per list (optional)
    const int32_t* n_eff;  // (B) optional: only points p < n_eff[b] are candidates.  Used for sa2, whose
                           // 512 input points end with copies of point 0 once FPS has run out of distinct
                           // locations: the copies come last in index order and equal a point that is
                           // already a member whenever they are in range, so they never add a new row.
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
    const int n_eff = a.n_eff ? min(a.N, a.n_eff[b]) : a.N;
    for (int p = 0; p < n_eff; ++p) {
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
        if (a.cnt[q]) a.cnt[q][(size_t)b * a.S + s] = cnt[q];
    }
}

// ---- block table: group g of cloud b owns ceil(cnt/8) blocks of 8 rows (0 if g >= n_unique) ---------
// 8-row blocks (a quarter of a 32-row MFMA tile): a group wastes 4 rows on average instead of 8 - 4 % fewer MFMA tiles on
// the BASELINE shape, where a ball holds ~70 members.
constexpr int kBlk = 8;

__global__ __launch_bounds__(kThreads) void pn2_blocks_kernel(const int32_t* __restrict__ cnt, const int32_t* __restrict__ n_unique,
                                                              int32_t* __restrict__ block_start /*(B,S+1)*/,
                                                              uint16_t* __restrict__ blockmap /*(B,maxblocks)*/, int S,
                                                              int maxblocks) {
    __shared__ int part[kThreads];
    const int b = blockIdx.x, t = threadIdx.x;
    const int nu = n_unique ? n_unique[b] : S;
    const int per = (S + kThreads - 1) / kThreads;  // groups per thread, contiguous
    int nb[4];
    int sum = 0;
    for (int i = 0; i < per; ++i) {
        const int g = t * per + i;
        nb[i] = (g < S && g < nu) ? (cnt[(size_t)b * S + g] + kBlk - 1) / kBlk : 0;
        sum += nb[i];
    }
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < kThreads; off <<= 1) {  // Hillis-Steele inclusive scan
        const int v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - sum;
    for (int i = 0; i < per; ++i) {
        const int g = t * per + i;
        if (g < S) {
            block_start[(size_t)b * (S + 1) + g] = run;
            for (int j = 0; j < nb[i]; ++j) blockmap[(size_t)b * maxblocks + run + j] = (uint16_t)g;
            run += nb[i];
        }
    }
    if (t == kThreads - 1) block_start[(size_t)b * (S + 1) + S] = part[t];
}

// ---- grouped MLP + max over ragged 8-row blocks -------------------------------------------------------
struct GroupArgs {
    const float* xyz;        // (B,N,ldx) member coordinates (first 3 floats of each row)
    int ldx;
    const float* new_xyz;    // (B,S,ldc) centroids
    int ldc;
    const int16_t* idx;      // (B,S,K)
    const int32_t* cnt;      // (B,S) true hits
    const int32_t* block_start;  // (B,S+1)
    const uint16_t* blockmap;    // (B,maxblocks)
    int maxblocks;
    const float* U;          // (B,N,ldu) per-point part of layer 1 (bias included) or null
    int ldu;
    const float* w1x;        // [C1][4] = (wx0, wx1, wx2, bias)
    const float* w2; const float* b2;  // packed C1->C2
    const float* w3; const float* b3;  // packed C2->C3
    const unsigned short* w2_bf3;      // the same two layers as three bf16 terms (iq_pack_weight_bf3) or null:
    const unsigned short* w3_bf3;      // 128-128-256 scales then run pn2_group_bf3_kernel
    float* out;              // (B,S,ldo) at the scale's column offset, zero-initialised
    int ldo;
    int N, S, K, blocks_per_wg;
    int B, wgs_per_cloud;
    int probe;               // tuning key 5 (79: per-phase cycle counts of pn2_group_bf3_kernel)
};

__device__ __forceinline__ void merge_max(float* addr, float v) {
    atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));  // v >= +0: float order == int order
}

// column max of the four 8-row quarters of a 32x32 accumulator tile (valid on the lower half-wave), bias + ReLU applied:
// rows 8 q .. 8 q + 7 live in registers 4 q .. 4 q + 3 of both half-waves
struct TileMax { float v[4]; };
__device__ __forceinline__ TileMax reduce_tile(f32x16 acc, float bias) {
    TileMax m;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float x = fmaxf(fmaxf(acc[4 * q], acc[4 * q + 1]), fmaxf(acc[4 * q + 2], acc[4 * q + 3]));
        x = fmaxf(x, __shfl_xor(x, 32));
        m.v[q] = fmaxf(x + bias, 0.f);
    }
    return m;
}
// ... merged into the groups that own the quarters (consecutive quarters of one group first combined)
__device__ __forceinline__ void emit_tile(f32x16 acc, float bias, const int* g, float* orow, int ldo, int fh) {
    const TileMax m = reduce_tile(acc, bias);
    if (fh != 0) return;
    int run = g[0];
    float v = m.v[0];
#pragma unroll
    for (int q = 1; q < 4; ++q) {
        if (g[q] == run) {
            v = fmaxf(v, m.v[q]);
        } else {
            if (run >= 0) merge_max(orow + (size_t)run * ldo, v);
            run = g[q];
            v = m.v[q];
        }
    }
    if (run >= 0) merge_max(orow + (size_t)run * ldo, v);
}

// MC = rows per chunk: 64 (two 32-row MFMA tiles per weight fragment) or 32 (one tile: half the LDS and far fewer
// registers, so twice the workgroups per CU hide each other's barriers and stage-0 phases; each weight fragment then
// feeds 4 MFMAs instead of 8).
template <int C1, int C2, int C3, int MC, int WPS = (MC == 64 ? 2 : 4)>
__global__ __launch_bounds__(kThreads, WPS) void pn2_group_kernel(GroupArgs a) {
    constexpr int kMC = MC;
    static_assert(MC == 64 || MC == 32, "chunk rows");
    constexpr int LD1 = C1 + 4, LD2 = C2 + 4;
    constexpr int KB1 = C1 / 8, KB2 = C2 / 8, NT2 = C2 / 32, NT3 = C3 / 32;
    __shared__ __attribute__((aligned(16))) float act1[kMC * LD1];
    __shared__ __attribute__((aligned(16))) float act2[kMC * LD2];
    __shared__ __attribute__((aligned(16))) float rel[2 * kMC * 4];  // dx,dy,dz, member index (bits); double-buffered
    __shared__ int blk_group[2 * (kMC / kBlk)];                      // owner group of a chunk's blocks (-1 = none)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 1-D grid, workgroups round-robin over the 8 XCDs: all workgroups of a cloud on one XCD (its U rows, coordinates and
    // indices are then fetched into one L2)
    const int slot = blockIdx.x >> 3;
    const int b = iq::xcd_cloud(blockIdx.x, a.wgs_per_cloud, a.B);
    if (b >= a.B) return;
    const int K = a.K;
    const int32_t* bstart = a.block_start + (size_t)b * (a.S + 1);
    const int nblocks = bstart[a.S];
    const int j0 = (slot % a.wgs_per_cloud) * a.blocks_per_wg;
    if (j0 >= nblocks) return;
    const int jend = min(nblocks, j0 + a.blocks_per_wg);
    const int nchunks = (jend - j0 + (kMC / kBlk) - 1) / (kMC / kBlk);
    const uint16_t* bmap = a.blockmap + (size_t)b * a.maxblocks;

    const int fl = lane & 31, fh = lane >> 5;
    const float* a1base = act1 + fl * LD1 + 4 * fh;
    const float* a2base = act2 + fl * LD2 + 4 * fh;
    float* c2base = act2 + (4 * fh) * LD2 + fl;

    // Stage 0a (member -> relative coordinates) of chunk c+1 runs before the barrier that precedes L3
    // of chunk c, and the per-point layer-1 rows U[p] of chunk c+1 are requested right after that
    // barrier, so the gather latency hides behind the L3 MFMAs.  rel / blk_group are double-buffered.
    // stage 0b: a thread owns 4 consecutive channels (c4) of NR rows; the per-point layer-1 rows U[p] arrive as 16-byte raw
    // buffer loads (resource on this cloud's U rows, 32-bit per-lane offsets): 4x fewer vector-memory instructions and
    // no 64-bit address arithmetic next to the MFMAs
    constexpr int Q1 = C1 / 4;                   // channel quads per row
    constexpr int NR = kMC * Q1 / kThreads;      // rows per thread in stage 0b
    const int c4 = tid % Q1, rsub = tid / Q1;
    f32x4 w1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w1[e] = *reinterpret_cast<const f32x4*>(a.w1x + (c4 * 4 + e) * 4);
    const __amdgpu_buffer_rsrc_t ursrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.U ? a.U + (size_t)b * a.N * a.ldu : a.w1x), 0, 0x7fffffff, 0x00020000);
    f32x4 ureg[NR];
    auto stage0a = [&](int ch, int buf) {
        if (tid < kMC) {
            int j = j0 + ch * (kMC / kBlk) + tid / kBlk;
            const bool live = j < jend;
            if (!live) j = jend - 1;                 // padding blocks replicate a valid one (never emitted)
            const int g = bmap[j];
            int m = (j - bstart[g]) * kBlk + (tid % kBlk);
            if (m >= a.cnt[(size_t)b * a.S + g]) m = 0;  // tail of the last block: copies of the first hit
            const int p = a.idx[((size_t)b * a.S + g) * K + m];
            const float* x = a.xyz + ((size_t)b * a.N + p) * a.ldx;
            const float* c = a.new_xyz + ((size_t)b * a.S + g) * a.ldc;
            f32x4 v;
            v[0] = x[0] - c[0]; v[1] = x[1] - c[1]; v[2] = x[2] - c[2];  // rounded like the reference's `-=`
            v[3] = __int_as_float(p);
            *reinterpret_cast<f32x4*>(rel + (buf * kMC + tid) * 4) = v;
            if (tid % kBlk == 0) blk_group[buf * (kMC / kBlk) + tid / kBlk] = live ? g : -1;
        }
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

    // Block maxima are not sent to memory one by one: the blocks of a group are consecutive, so each wave keeps, per n-tile,
    // the running maximum of the group it is in and writes it when the group changes - a plain store if all of the
    // group's blocks belong to this workgroup, an atomic max only for a group that straddles a workgroup boundary.  One
    // store per group instead of one atomic per 16-row block (8x fewer for K = 128), and no atomic is in flight when
    // stage 0b waits for the gathered U rows (memory operations retire in order).
    constexpr int NQ = NT3 >= 4 ? NT3 / 4 : 1;
    int run_g[NQ];
    float run_v[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { run_g[q] = -1; run_v[q] = 0.f; }
    auto flush_group = [&](int q) {
        const int g = run_g[q];
        if (g < 0 || fh != 0) return;
        float* dst = a.out + ((size_t)b * a.S + g) * a.ldo + (q * 4 + wave) * 32 + fl;
        if (bstart[g] >= j0 && bstart[g + 1] <= jend) *dst = run_v[q];   // sole owner (out is zero-initialised, values >= 0)
        else merge_max(dst, run_v[q]);
    };
    auto feed = [&](int q, int g, float v) {   // g is wave-uniform
        if (g < 0) return;
        if (g != run_g[q]) {
            flush_group(q);
            run_g[q] = g;
            run_v[q] = v;
        } else {
            run_v[q] = fmaxf(run_v[q], v);
        }
    };
    WRing ring2, ring3;
    const int wave_s = uniform(wave);
    const WBuf w2b = wbuf_make(a.w2, lane), w3b = wbuf_make(a.w3, lane);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int cur = ch & 1, nxt = cur ^ 1;
        constexpr int BPC = kMC / kBlk;            // 8-row blocks per chunk (4 per MFMA tile)
        const int mts = (MC == 64 && (jend - (j0 + ch * BPC)) > BPC / 2) ? 2 : 1;  // second m-tile holds live blocks?
        int gq[BPC];                               // owner groups of the chunk's blocks (wave-uniform)
#pragma unroll
        for (int i = 0; i < BPC; ++i) gq[i] = blk_group[cur * BPC + i];
        if (NT2 >= 4) wring_prime(ring2, w2b, wave_s * KB1 * kFragBytes);  // in flight across stage 0b
        // ---- stage 0b: layer 1 -> act1 -------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = rsub + i * (kThreads / Q1);
            const f32x4 v = *reinterpret_cast<const f32x4*>(rel + (cur * kMC + r) * 4);
            f32x4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                h[e] = fmaxf(fmaf(w1[e][2], v[2], fmaf(w1[e][1], v[1], w1[e][0] * v[0])) + w1[e][3] + ureg[i][e], 0.f);
            *reinterpret_cast<f32x4*>(act1 + r * LD1 + c4 * 4) = h;
        }
        __syncthreads();  // act1 complete; every wave has finished L3 of the previous chunk (act2 is free)
        // ---- layer 2: C1 -> C2 (+bn, relu) -> act2 -------------------------------------------------
        if (NT2 >= 4) {
#pragma unroll
            for (int q = 0; q < NT2 / 4; ++q) {
                const int nt = q * 4 + wave, nts = q * 4 + wave_s;
                f32x16 acc0 = {0}, acc1 = {0};
                const int wq = nts * KB1 * kFragBytes;
                const int wn = (q + 1 < NT2 / 4 ? nts + 4 : nts) * KB1 * kFragBytes;
                if (MC == 64 && mts == 2) mfma_ntile<LD1, KB1, 2>(a1base, w2b, wq, wn, ring2, acc0, acc1);
                else                      mfma_ntile<LD1, KB1, 1>(a1base, w2b, wq, wn, ring2, acc0, acc1);
                const float bias = a.b2[nt * 32 + fl];
                float* dst = c2base + nt * 32;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    dst[c_row_i(i) * LD2] = fmaxf(acc0[i] + bias, 0.f);
                    if (MC == 64) dst[(32 + c_row_i(i)) * LD2] = fmaxf(acc1[i] + bias, 0.f);
                }
            }
        } else {
            for (int t = wave; t < mts * NT2; t += 4) {
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
        if (NT3 >= 4) wring_prime(ring3, w3b, wave_s * KB2 * kFragBytes);  // in flight across the barrier
        if (ch + 1 < nchunks) stage0a(ch + 1, nxt);
        __syncthreads();  // act2 complete; rel[nxt] visible
        if (ch + 1 < nchunks) gather_u(nxt);  // consumed after L3
        // ---- layer 3: C2 -> C3 (+bn, relu), block maxima merged into the owning groups ---------------
        if (NT3 >= 4) {
#pragma unroll
            for (int q = 0; q < NT3 / 4; ++q) {
                const int nt = q * 4 + wave, nts = q * 4 + wave_s;
                f32x16 acc0 = {0}, acc1 = {0};
                const int wq = nts * KB2 * kFragBytes;
                const int wn = (q + 1 < NT3 / 4 ? nts + 4 : nts) * KB2 * kFragBytes;
                if (MC == 64 && mts == 2) mfma_ntile<LD2, KB2, 2>(a2base, w3b, wq, wn, ring3, acc0, acc1);
                else                      mfma_ntile<LD2, KB2, 1>(a2base, w3b, wq, wn, ring3, acc0, acc1);
                const float bias = a.b3[nt * 32 + fl];
                const TileMax m0 = reduce_tile(acc0, bias);
#pragma unroll
                for (int i = 0; i < 4; ++i) feed(q, gq[i], m0.v[i]);
                if (MC == 64 && mts == 2) {
                    const TileMax m1 = reduce_tile(acc1, bias);
#pragma unroll
                    for (int i = 0; i < 4; ++i) feed(q, gq[(BPC - 4) + i], m1.v[i]);
                }
            }
        } else {
            for (int t = wave; t < mts * NT3; t += 4) {
                const int mt = t / NT3, nt = t - mt * NT3;
                f32x16 acc = {0};
                const int wq = uniform(nt) * KB2 * kFragBytes;
#pragma unroll 4
                for (int kb = 0; kb < KB2; ++kb)
                    acc = mfma4(lds_frag<LD2>(a2base + mt * 32 * LD2, 0, kb), wbuf_load(w3b, wq + kb * kFragBytes), acc);
                const float bias = a.b3[nt * 32 + fl];
                float* orow = a.out + (size_t)b * a.S * a.ldo + nt * 32 + fl;
                emit_tile(acc, bias, gq + (mt == 0 ? 0 : BPC - 4), orow, a.ldo, fh);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) flush_group(q);
}

// ---- the 128-128-256 scale on the bf16 matrix pipe: bf16x3, float32-exact (iq_bf3.h) ---------------------------
// Same blocks, groups, stage 0 and pooling as pn2_group_kernel; layers 2 and 3 as six bf16 products per float32 product.
// The bf16 pipe is 2.67x faster per float32 MAC, so operand delivery decides the shape: 64-row chunks (a weight fragment - three
// 1 KiB terms from L2 - feeds two m-tiles; with 32-row chunks the four waves would ask the L1 path for 64 B / clk, all it has),
// layer 3 as 2 x 2 tiles per wave (the A terms of a k-step are read from LDS once for both n-tiles).  Activations live in LDS as
// three bf16 planes of 272-byte rows (conflict-free ds_read_b128), split where they are produced; act1 and act2 SHARE one
// 52 KB image - layer 2 keeps its two tiles in registers until every wave has read act1 - so that two workgroups fit a CU and
// fill each other's barriers and stage-0 phases (four barriers per chunk instead of two).
// Round 5, three closed experiments on the 0.55 MFMA-busy of this kernel and its PointConv twin (profiles/r05_grouped_schedule.txt;
// patches under tools/experiments/): (1) s_setprio 1 inside the MFMA loops, or inside the VALU phases: no change (81.3-82.1 k
// coalitions/s either way); (2) "ping-pong": one 512-thread workgroup running two block ranges, the second half one phase behind,
// every barrier shared, so that each SIMD always pairs a VALU phase of one wave with an MFMA loop of the other: bit-identical, 9 %
// SLOWER (80.3 -> 73.3 k; PointConv 84.5 -> 82.3 k), and still 8 % slower with the A terms of the MFMA loops prefetched one k-step
// ahead (83.2 -> 76.6 k) - one wave alone does not keep the matrix pipe fed, the two workgroups' waves overlapping in their MFMA
// loops is what saturates it; (3) the VALU phases (layer 1, the three-term splits) written stage by stage over eight independent
// values instead of value by value (the compiler's schedule is one dependent chain after the other on two or three temporaries):
// bit-identical, no change (81.5 / 81.9 k; PointConv 85.0 / 84.9 k; chain kernel 781.2 / 781.7 k).
__device__ unsigned long long g_gb_dbg[12];
// TR: the TRANSPOSED tiles (weights as the A operand, iq_bf3.h ct_tile_to_planes): same fragments, same products, swapped operands
template <int MTS, bool TR>
__device__ __forceinline__ void gb_layer2(const unsigned char* abase, const __amdgpu_buffer_rsrc_t& rs, int voff, int nt,
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
struct B3x2 { B3 b[2]; };
template <int MTS>
__device__ __forceinline__ void gb_layer3(const unsigned char* abase, const __amdgpu_buffer_rsrc_t& rs, int voff, int nt0,
                                          B3x2 (&ring)[2], f32x16 (&acc)[MTS][2]) {
    constexpr int ROWB = 272, PLANEB = 64 * ROWB, TS = 8 * 8 * 1024;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        bf16x8 af[MTS][3];
#pragma unroll
        for (int i = 0; i < MTS; ++i) a3_load<PLANEB>(af[i], abase + i * 32 * ROWB, ks);
        const B3 b[2] = {ring[ks & 1].b[0], ring[ks & 1].b[1]};
        if (ks + 2 < 8) {
            ring[ks & 1].b[0] = b3_load_at(rs, voff, (nt0 * 8 + ks + 2) * 1024, TS);
            ring[ks & 1].b[1] = b3_load_at(rs, voff, ((nt0 + 4) * 8 + ks + 2) * 1024, TS);
        }
        mfma_bf3_block<MTS, 2>(af, b, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// TR (the default): layer 2's tiles transposed, so that act2 is stored with whole 8-byte stores and without the two-lane DPP trade
// (tuning key 7 = 1: the untransposed form; same products in the same order).
template <bool STAMP, bool TR>   // STAMP: diagnostic build (tuning key 5 = 79)
__global__ __launch_bounds__(kThreads, 2) void pn2_group_bf3_kernel(GroupArgs a) {
    constexpr int C1 = 128, kMC = 64, ROWB = 272, PLANEB = kMC * ROWB, BPC = kMC / kBlk;
    __shared__ __attribute__((aligned(16))) unsigned char planes[3 * PLANEB];   // act1, then act2: three bf16 planes [64][136]
    __shared__ __attribute__((aligned(16))) float rel[2 * kMC * 4];             // dx,dy,dz, member index (bits); double-buffered
    __shared__ int blk_group[2 * BPC];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = blockIdx.x >> 3;
    const int b = iq::xcd_cloud(blockIdx.x, a.wgs_per_cloud, a.B);
    if (b >= a.B) return;
    const int K = a.K;
    const int32_t* bstart = a.block_start + (size_t)b * (a.S + 1);
    const int nblocks = bstart[a.S];
    const int j0 = (slot % a.wgs_per_cloud) * a.blocks_per_wg;
    if (j0 >= nblocks) return;
    const int jend = min(nblocks, j0 + a.blocks_per_wg);
    const int nchunks = (jend - j0 + BPC - 1) / BPC;
    const uint16_t* bmap = a.blockmap + (size_t)b * a.maxblocks;

    const int fl = lane & 31, fh = lane >> 5;
    const unsigned char* abase = planes + fl * ROWB + 16 * fh;
    // STAMP: shader cycles per phase of wave 0, summed over workgroups (g_gb_dbg, printed by launch_group_t).  Measured (r4,
    // cycles per 64-row chunk and wave, two workgroups per CU): stage 0b 4 400, layer 2 5 700, stage 0a 1 850, layer-2
    // epilogue 4 000, layer 3 + pooling 14 300, barrier waits 1 250, prologue + flush 2 000 per chunk = 33 400, of which the
    // 288 MFMAs need 9 200: the other wave of the SIMD does the same, and its VALU phases do not hide under this wave's MFMAs
    // (the two share the SIMD's issue and register ports) - 0.55 MFMA-busy.
    unsigned long long t_last = 0;
    unsigned t_sum[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&](int slot) {
        if (STAMP) {
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            if (slot >= 0) t_sum[slot] += (unsigned)(t - t_last);
            t_last = t;
        }
    };

    constexpr int Q1 = C1 / 4, NR = kMC * Q1 / kThreads;   // stage 0b: a thread owns 4 consecutive channels of NR rows
    const int c4 = tid % Q1, rsub = tid / Q1;
    f32x4 w1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w1[e] = *reinterpret_cast<const f32x4*>(a.w1x + (c4 * 4 + e) * 4);
    const __amdgpu_buffer_rsrc_t ursrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.U ? a.U + (size_t)b * a.N * a.ldu : a.w1x), 0, 0x7fffffff, 0x00020000);
    f32x4 ureg[NR];
    auto stage0a = [&](int ch, int buf) {   // as pn2_group_kernel
        if (tid < kMC) {
            int j = j0 + ch * BPC + tid / kBlk;
            const bool live = j < jend;
            if (!live) j = jend - 1;
            const int g = bmap[j];
            int m = (j - bstart[g]) * kBlk + (tid % kBlk);
            if (m >= a.cnt[(size_t)b * a.S + g]) m = 0;
            const int p = a.idx[((size_t)b * a.S + g) * K + m];
            const float* x = a.xyz + ((size_t)b * a.N + p) * a.ldx;
            const float* c = a.new_xyz + ((size_t)b * a.S + g) * a.ldc;
            f32x4 v;
            v[0] = x[0] - c[0]; v[1] = x[1] - c[1]; v[2] = x[2] - c[2];
            v[3] = __int_as_float(p);
            *reinterpret_cast<f32x4*>(rel + (buf * kMC + tid) * 4) = v;
            if (tid % kBlk == 0) blk_group[buf * BPC + tid / kBlk] = live ? g : -1;
        }
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
    stamp(-1);
    stage0a(0, 0);
    __syncthreads();
    gather_u(0);
    stamp(0);   // prologue

    int run_g[2] = {-1, -1};
    float run_v[2] = {0.f, 0.f};
    auto flush_group = [&](int q) {   // as pn2_group_kernel: one store per group, an atomic max only across workgroups
        const int g = run_g[q];
        if (g < 0 || fh != 0) return;
        float* dst = a.out + ((size_t)b * a.S + g) * a.ldo + (q * 4 + wave) * 32 + fl;
        if (bstart[g] >= j0 && bstart[g + 1] <= jend) *dst = run_v[q];
        else merge_max(dst, run_v[q]);
    };
    auto feed = [&](int q, int g, float v) {
        if (g < 0) return;
        if (g != run_g[q]) {
            flush_group(q);
            run_g[q] = g;
            run_v[q] = v;
        } else {
            run_v[q] = fmaxf(run_v[q], v);
        }
    };
    const int wave_s = uniform(wave), voff = lane * 16;
    const __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.w2_bf3), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t w3rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.w3_bf3), 0, 0x7fffffff, 0x00020000);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int cur = ch & 1, nxt = cur ^ 1;
        const int mts = (jend - (j0 + ch * BPC)) > BPC / 2 ? 2 : 1;
        int gq[BPC];
#pragma unroll
        for (int i = 0; i < BPC; ++i) gq[i] = blk_group[cur * BPC + i];
        B3 ring2[4];                                 // layer 2's weights (n-tile = wave), in flight across stage 0b
#pragma unroll
        for (int i = 0; i < 4; ++i) ring2[i] = b3_load_at(w2rs, voff, (wave_s * 8 + i) * 1024, 4 * 8 * 1024);
        // ---- stage 0b: layer 1 -> act1 (three planes) ----------------------------------------------------
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = rsub + i * (kThreads / Q1);
            const f32x4 v = *reinterpret_cast<const f32x4*>(rel + (cur * kMC + r) * 4);
            f32x4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                h[e] = fmaxf(fmaf(w1[e][2], v[2], fmaf(w1[e][1], v[1], w1[e][0] * v[0])) + w1[e][3] + ureg[i][e], 0.f);
            row4_to_planes<PLANEB>(planes + r * ROWB + c4 * 8, h);
        }
        stamp(1);   // ring prime + stage 0b
        __syncthreads();  // act1 complete
        stamp(2);   // wait
        // ---- layer 2: 128 -> 128, tiles (m-tile 0..1, n-tile = wave) kept in registers -------------------
        f32x16 acc2[2][1] = {{{0}}, {{0}}};
        if (mts == 2) {
            gb_layer2<2, TR>(abase, w2rs, voff, wave_s, ring2, acc2);
        } else {
            f32x16 one[1][1] = {{{0}}};
            gb_layer2<1, TR>(abase, w2rs, voff, wave_s, ring2, one);
            acc2[0][0] = one[0][0];
        }
        B3x2 ring3[2];                               // layer 3's weights (n-tiles wave, wave + 4), in flight across the epilogue
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ring3[i].b[0] = b3_load_at(w3rs, voff, (wave_s * 8 + i) * 1024, 8 * 8 * 1024);
            ring3[i].b[1] = b3_load_at(w3rs, voff, ((wave_s + 4) * 8 + i) * 1024, 8 * 8 * 1024);
        }
        stamp(3);   // layer 2 MFMAs + ring 3 prime
        if (ch + 1 < nchunks) stage0a(ch + 1, nxt);
        stamp(4);   // stage 0a of the next chunk
        __syncthreads();  // every wave has read act1: the image is free
        stamp(5);   // wait
        {
            if (TR) {   // register r = channel c_row_i(r) + 4 fh of this wave's n-tile: the biases of the lane's 16 channels
                f32x4 bq[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const f32x4*>(a.b2 + wave * 32 + 8 * g + 4 * fh);
                ct_tile_to_planes<ROWB, PLANEB>(planes + wave * 64, lane, [&](int r) { return fmaxf(acc2[0][0][r] + bq[r >> 2][r & 3], 0.f); });
                if (mts == 2)
                    ct_tile_to_planes<ROWB, PLANEB>(planes + 32 * ROWB + wave * 64, lane,
                                                    [&](int r) { return fmaxf(acc2[1][0][r] + bq[r >> 2][r & 3], 0.f); });
            } else {
                const float bias = a.b2[wave * 32 + fl];
                c_tile_to_planes<ROWB, PLANEB>(planes + wave * 64, lane, [&](int i) { return fmaxf(acc2[0][0][i] + bias, 0.f); });
                if (mts == 2)
                    c_tile_to_planes<ROWB, PLANEB>(planes + 32 * ROWB + wave * 64, lane, [&](int i) { return fmaxf(acc2[1][0][i] + bias, 0.f); });
            }
        }
        stamp(6);   // layer 2 epilogue
        __syncthreads();  // act2 complete; rel[nxt] visible
        stamp(7);   // wait
        if (ch + 1 < nchunks) gather_u(nxt);  // consumed after layer 3
        // ---- layer 3: 128 -> 256, 2 x 2 tiles per wave, block maxima merged into the owning groups -------
        if (mts == 2) {
            f32x16 acc3[2][2] = {{{0}, {0}}, {{0}, {0}}};
                gb_layer3<2>(abase, w3rs, voff, wave_s, ring3, acc3);
    #pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float bias = a.b3[(q * 4 + wave) * 32 + fl];
                const TileMax m0 = reduce_tile(acc3[0][q], bias);
#pragma unroll
                for (int i = 0; i < 4; ++i) feed(q, gq[i], m0.v[i]);
                const TileMax m1 = reduce_tile(acc3[1][q], bias);
#pragma unroll
                for (int i = 0; i < 4; ++i) feed(q, gq[(BPC - 4) + i], m1.v[i]);
            }
        } else {
            f32x16 acc3[1][2] = {{{0}, {0}}};
                gb_layer3<1>(abase, w3rs, voff, wave_s, ring3, acc3);
    #pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float bias = a.b3[(q * 4 + wave) * 32 + fl];
                const TileMax m0 = reduce_tile(acc3[0][q], bias);
#pragma unroll
                for (int i = 0; i < 4; ++i) feed(q, gq[i], m0.v[i]);
            }
        }
        stamp(8);   // gather + layer 3 + pooling
        __syncthreads();  // every wave has read act2: the next chunk's stage 0b may overwrite the image
        stamp(9);   // wait
    }
    flush_group(0);
    flush_group(1);
    stamp(10);
    if (STAMP && tid == 0) {
        for (int i = 0; i < 11; ++i) atomicAdd(&g_gb_dbg[i], (unsigned long long)t_sum[i]);
        atomicAdd(&g_gb_dbg[11], (unsigned long long)nchunks);
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
int launch_group_t(GroupArgs a, int B, hipStream_t st) {
    a.B = B;
    a.wgs_per_cloud = (a.maxblocks + a.blocks_per_wg - 1) / a.blocks_per_wg;  // workgroups past a cloud's block count exit
    dim3 grid((unsigned)((B + 7) / 8 * 8 * a.wgs_per_cloud));
    // the widest stage (128-128-256: 70 KB of LDS and 150 registers at 64 rows = 2 workgroups per CU) runs 32-row chunks,
    // 4 workgroups per CU: 44.5 k -> 46.3 k coalitions/s (tuning key 5 = 64 forces 64-row chunks for A/B runs)
    if (C1 == 128 && C2 == 128 && C3 == 256 && a.w2_bf3 && a.w3_bf3 && iq::tuning(iq::kTuneExperiment) != 56 &&
        iq::tuning(iq::kTuneExperiment) != 64)    // 5 = 56 / 64: the fp32-MFMA kernel with 32- / 64-row chunks (A/B and tests)
    {
        if (a.probe != 79 && !iq::tuning(iq::kTuneNoTranspose)) hipLaunchKernelGGL((pn2_group_bf3_kernel<false, true>), grid, dim3(kThreads), 0, st, a);
        else if (a.probe != 79) hipLaunchKernelGGL((pn2_group_bf3_kernel<false, false>), grid, dim3(kThreads), 0, st, a);
        else {   // diagnostic: synchronous
            hipLaunchKernelGGL((pn2_group_bf3_kernel<true, true>), grid, dim3(kThreads), 0, st, a);
            unsigned long long h[12];
            (void)hipStreamSynchronize(st);
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gb_dbg), sizeof(h));
            const double n = (double)h[11];
            fprintf(stderr, "pn2_group_bf3 (wave 0, cycles per chunk over %.0f chunks): prologue/chunk %.0f | ring2+stage0b %.0f wait %.0f | L2 %.0f stage0a %.0f "
                            "wait %.0f | epilogue %.0f wait %.0f | gather+L3+pool %.0f wait %.0f | flush/chunk %.0f\n", n, h[0] / n, h[1] / n, h[2] / n,
                    h[3] / n, h[4] / n, h[5] / n, h[6] / n, h[7] / n, h[8] / n, h[9] / n, h[10] / n);
            for (auto& v : h) v = 0;
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gb_dbg), h, sizeof(h));
        }
    }
    else if (C3 >= 256 && iq::tuning(iq::kTuneExperiment) != 64)
        hipLaunchKernelGGL((pn2_group_kernel<C1, C2, C3, 32>), grid, dim3(kThreads), 0, st, a);
    else
        hipLaunchKernelGGL((pn2_group_kernel<C1, C2, C3, 64>), grid, dim3(kThreads), 0, st, a);
    return iq::check_launch("pn2_group_kernel");
}

// FLOP the MFMA tiles of pn2_group_kernel execute for the block table just built (profiling only: reads the per-cloud block
// counts back, one sync).  A workgroup owns blocks_per_wg 8-row blocks; every four blocks (or a last partial set) are one
// 32-row MFMA tile, whatever the chunk size.
double group_work(const GroupArgs& a, int B, int c1, int c2, int c3, hipStream_t st) {
    std::vector<int32_t> nb((size_t)B);
    if (hipMemcpy2DAsync(nb.data(), sizeof(int32_t), a.block_start + a.S, (size_t)(a.S + 1) * sizeof(int32_t), sizeof(int32_t), (size_t)B,
                         hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return 0.0;
    double rows = 0.0;
    for (int b = 0; b < B; ++b) {
        for (int j0 = 0; j0 < nb[b]; j0 += a.blocks_per_wg) {
            const int jend = std::min(nb[b], j0 + a.blocks_per_wg);
            rows += 32.0 * ((jend - j0 + 3) / 4);
        }
    }
    return rows * 2.0 * ((double)c1 * c2 + (double)c2 * c3);
}

int launch_group(const iq_pn2_scale& sc, GroupArgs a, const int32_t* n_unique, int B, hipStream_t st, bool dominant = false,
                 const void* l2_bf3 = nullptr, const void* l3_bf3 = nullptr) {
    a.w1x = sc.w1x;
    a.probe = iq::tuning(iq::kTuneExperiment);
    a.w2_bf3 = reinterpret_cast<const unsigned short*>(l2_bf3);
    a.w3_bf3 = reinterpret_cast<const unsigned short*>(l3_bf3);
    a.w2 = sc.l2.w; a.b2 = sc.l2.b;
    a.w3 = sc.l3.w; a.b3 = sc.l3.b;
    a.K = sc.nsample;
    // 24 blocks = 192 rows = 3 chunks of 64 rows per workgroup: enough to amortise the prologue, small enough for an even
    // tail (512 rows: -4 %, 2048: -18 %); tuning key 6 overrides for experiments (a multiple of 8: whole chunks)
    a.blocks_per_wg = iq::tuning(iq::kTuneGroupBlocks) > 0 ? (iq::tuning(iq::kTuneGroupBlocks) + 7) / 8 * 8 : 24;
    a.maxblocks = a.S * ((a.K + kBlk - 1) / kBlk);
    const int c1 = sc.l2.cin, c2 = sc.l2.cout, c3 = sc.l3.cout;
    IQ_REQUIRE(sc.l3.cin == c2, "pointnet2 scale: layer sizes do not chain");
    IQ_REQUIRE(a.S <= 4 * kThreads, "pointnet2: S=%d too large for the block scan", a.S);
    IQ_REQUIRE(a.maxblocks <= 8192, "pointnet2: %d groups x %d samples exceed the block table", a.S, a.K);
    hipLaunchKernelGGL(pn2_blocks_kernel, dim3(B), dim3(kThreads), 0, st, a.cnt, n_unique, const_cast<int32_t*>(a.block_start),
                       const_cast<uint16_t*>(a.blockmap), a.S, a.maxblocks);
    int rc = iq::check_launch("pn2_blocks_kernel");
    if (rc) return rc;
    if (c1 == 32 && c2 == 32 && c3 == 64) return launch_group_t<32, 32, 64>(a, B, st);
    if (c1 == 64 && c2 == 64 && c3 == 128) return launch_group_t<64, 64, 128>(a, B, st);
    if (c1 == 64 && c2 == 96 && c3 == 128) return launch_group_t<64, 96, 128>(a, B, st);
    if (c1 == 128 && c2 == 128 && c3 == 256) {
        iq::ProfileSpan span(iq::kSlotDominant, st, dominant && iq::profile_enabled() ? group_work(a, B, c1, c2, c3, st) : 0.0);
        return launch_group_t<128, 128, 256>(a, B, st);
    }
    return iq::fail(IQ_EUNSUPPORTED, "pointnet2 scale %d-%d-%d has no kernel instantiation", c1, c2, c3);
}

int launch_ball(const float* xyz, const float* new_xyz, int ldc, const iq_pn2_scale* sc, int nr, int16_t* const* idx,
                int32_t* const* cnt, const int32_t* n_eff, int B, int N, int S, hipStream_t st) {
    BallArgs a{};
    a.n_eff = n_eff;
    a.xyz = xyz; a.new_xyz = new_xyz; a.ldc = ldc; a.N = N; a.S = S; a.nr = nr;
    for (int q = 0; q < nr; ++q) {
        a.r2[q] = (float)((double)sc[q].radius * (double)sc[q].radius);  // python float r**2, cast by the comparison
        a.K[q] = sc[q].nsample;
        a.idx[q] = idx[q];
        a.cnt[q] = cnt[q];
    }
    hipLaunchKernelGGL(ball_query_kernel<int16_t>, dim3((S + kThreads - 1) / kThreads, B), dim3(kThreads),
                       (size_t)N * 4 * sizeof(float), st, a);
    return iq::check_launch("ball_query_kernel");
}

// ==== sa1 from pair tables (iq_pointnet2_coalitions) ==================================================================
// sa1 has no input features: a member row is MLP_s(x_q - x_p), a function of the POINT PAIR only.  All coalitions of a
// call are masked copies of a few source clouds, so every pair that a ball query can ever return - both points kept,
// or one / both of them the centre that masked points collapse to (index N below) - is known up front.  Per source
// cloud and scale the rows MLP_s(P[q] - P[p]) of all pairs inside the radius are computed ONCE (a few coalitions' worth
// of work) and sa1 of every coalition becomes a gather-max over table rows.  The pair predicate is the ball query's
// own expression, and the rows are produced by the same layer arithmetic as pn2_group_kernel (same fma chain for
// layer 1, same MFMA order over k in the dense layers), so the features are bit-identical to the grouped MLP.

struct PairTabs {
    const int32_t* map[3];   // (nc, N+1, N+1) row of pair (p, q) in feat, or -1
    const float* feat[3];    // (rows, C3)
    int c3[3];
    bool use[3];
    // ball query of sa1 by bit operations (all three scales on tables, N <= 1024): per (cloud, centroid point, scale) the
    // members inside the radius as a bit mask over the points (bit N, word 32: the centre), and per coalition its kept points
    const uint32_t* ball_bits[3];   // (nc, N+1, 33) each, or null
    const uint32_t* kept_bits;      // (B, 32)
    const uint64_t* touch[3];       // (nc, N+1) region-reduced tables (pt_regtab_kernel), or null
    const float* regtab[3];
};

__device__ __forceinline__ float pt_dist(const float* c, const float* v) {  // ball_query_kernel's expression
    const float sc = __fadd_rn(__fadd_rn(__fmul_rn(c[0], c[0]), __fmul_rn(c[1], c[1])), __fmul_rn(c[2], c[2]));
    const float sv = __fadd_rn(__fadd_rn(__fmul_rn(v[0], v[0]), __fmul_rn(v[1], v[1])), __fmul_rn(v[2], v[2]));
    float dot = __fmul_rn(c[0], v[0]);
    dot = __fmaf_rn(c[1], v[1], dot);
    dot = __fmaf_rn(c[2], v[2], dot);
    return __fadd_rn(__fadd_rn(__fmul_rn(-2.f, dot), sc), sv);
}

// ---- sa1 ball query by bit operations ----------------------------------------------------------------------------------
// Which source points lie inside a radius of a source point (or of the centre) does not depend on the coalition - the pair
// table's row map already says so (row >= 0) - only which of them exist does.  ball_query_kernel walks the 1024 points of
// every masked cloud for each of its 512 centroids; here the members of a ball are the set bits, lowest index first, of
//     (inside[centroid] & kept) | (centre inside ? masked : 0)
// over 32 words: the K lowest indices inside the radius of the masked cloud, masked points (all at the centre) included -
// the lists ball_query_kernel writes, minus the padding that nothing on this path reads.
constexpr int kBallWords = 33;   // 1024 point bits + the centre's

// bits[c][centroid][w] from the row map of a scale: bit j of word w = pair (centroid, 32 w + j) has a row
__global__ __launch_bounds__(64) void pt_bits_kernel(const int32_t* __restrict__ map, uint32_t* __restrict__ bits, int n1) {
    const int c = blockIdx.y, ci = blockIdx.x, w = threadIdx.x;
    if (w >= kBallWords) return;
    const int32_t* mrow = map + ((size_t)c * n1 + ci) * n1;
    uint32_t v = 0;
    for (int j = 0; j < 32; ++j) {
        const int e = w * 32 + j;
        if (e < n1 && mrow[e] >= 0) v |= 1u << j;
    }
    bits[((size_t)c * n1 + ci) * kBallWords + w] = v;
}

// kept[b][w]: bit j = point 32 w + j of coalition b's cloud is kept (N <= 1024)
__global__ __launch_bounds__(64) void pn2_kept_bits_kernel(const int32_t* __restrict__ region_id, const uint64_t* __restrict__ keep,
                                                           const int32_t* __restrict__ cloud_of, uint32_t* __restrict__ kept, int N,
                                                           int nclouds) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int c = cloud_of ? cloud_of[b] : (nclouds == 1 ? 0 : b);
    const uint64_t k = keep[b];
    const int32_t* rid = region_id + (size_t)c * N;
    for (int i0 = 0; i0 < 1024; i0 += 64) {
        const int i = i0 + lane;
        const unsigned long long m = __ballot(i < N && iq::keep_bit(k, rid[min(i, N - 1)]));
        if (lane == 0) { kept[(size_t)b * 32 + (i0 >> 5)] = (uint32_t)m; kept[(size_t)b * 32 + (i0 >> 5) + 1] = (uint32_t)(m >> 32); }
    }
}

struct BitBallArgs {
    const uint32_t* bits[3];
    const uint32_t* kept;
    const int32_t* fps;        // (B,S) centroid indices into the masked cloud
    const int32_t* n_unique;   // (B)
    const int32_t* cloud_of;
    int16_t* idx[3];
    int32_t* cnt[3];
    int K[3];
    int N, S, B, nclouds;
    const uint64_t* touch[3];  // (nc,N+1) per scale, or null: a "simple" ball (pt_regtab_kernel) needs no member list
};

__global__ __launch_bounds__(kThreads) void pn2_bitball_kernel(BitBallArgs a) {
    const int b = blockIdx.y, s = blockIdx.x * kThreads + threadIdx.x;
    if (s >= a.S || s >= a.n_unique[b]) return;           // duplicate centroids are filled from group 0 afterwards
    const int c = a.cloud_of ? a.cloud_of[b] : (a.nclouds == 1 ? 0 : b);
    const uint32_t* kept = a.kept + (size_t)b * 32;
    const int pi = a.fps[(size_t)b * a.S + s];
    const int ci = (kept[pi >> 5] >> (pi & 31)) & 1u ? pi : a.N;   // a masked centroid sits at the centre
    const int n1 = a.N + 1;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        if (a.touch[q] && a.touch[q][(size_t)c * n1 + ci]) continue;   // served from the region-reduced table: no list, no count
        const uint32_t* wb = a.bits[q] + ((size_t)c * n1 + ci) * kBallWords;
        const bool centre_in = (wb[a.N >> 5] >> (a.N & 31)) & 1u;     // the centre as a member: bit N
        int16_t* out = a.idx[q] + ((size_t)b * a.S + s) * a.K[q];
        const int K = a.K[q];
        int cnt = 0;
        for (int w = 0; w < 32 && cnt < K; ++w) {
            const int lo = w * 32;
            if (lo >= a.N) break;
            const uint32_t valid = a.N - lo >= 32 ? 0xffffffffu : (1u << (a.N - lo)) - 1u;
            const uint32_t kw = kept[w];
            uint32_t cand = (wb[w] & kw) | (centre_in ? (~kw & valid) : 0u);
            if (w == (a.N >> 5)) cand &= valid;                        // bit N itself is not a point
            while (cand && cnt < K) {
                out[cnt++] = (int16_t)(lo + __builtin_ctz(cand));
                cand &= cand - 1;
            }
        }
        a.cnt[q][(size_t)b * a.S + s] = cnt;
    }
}

// masked clouds: X[b][i] = kept ? clouds[c][i] : centers[c]
__global__ void pn2_mask_kernel(const float* __restrict__ clouds, const float* __restrict__ centers,
                                const int32_t* __restrict__ region_id, const uint64_t* __restrict__ keep,
                                const int32_t* __restrict__ cloud_of, float* __restrict__ out, int N, int B, int nclouds) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * N) return;
    const int b = t / N, i = t - b * N;
    const int c = cloud_of ? cloud_of[b] : (nclouds == 1 ? 0 : b);
    const bool kept = iq::keep_bit(keep[b], region_id[(size_t)c * N + i]);
    const float* src = kept ? clouds + ((size_t)c * N + i) * 3 : centers + (size_t)c * 3;
    out[(size_t)t * 3] = src[0]; out[(size_t)t * 3 + 1] = src[1]; out[(size_t)t * 3 + 2] = src[2];
}

// points of source cloud c: P[0..N-1] = cloud, P[N] = centre  -> (nc, N+1, 4) padded
__global__ void pt_points_kernel(const float* __restrict__ clouds, const float* __restrict__ centers, float* __restrict__ P,
                                 int N, int nclouds) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nclouds * (N + 1)) return;
    const int c = t / (N + 1), i = t - c * (N + 1);
    const float* src = i < N ? clouds + ((size_t)c * N + i) * 3 : centers + (size_t)c * 3;
    P[(size_t)t * 4] = src[0]; P[(size_t)t * 4 + 1] = src[1]; P[(size_t)t * 4 + 2] = src[2]; P[(size_t)t * 4 + 3] = 0.f;
}

// one workgroup per (p, c): number of q inside each radius -> cnt[(c*3 + s)*(N+1) + p]
__global__ __launch_bounds__(kThreads) void pt_count_kernel(const float* __restrict__ P, int N, float r0, float r1, float r2,
                                                            int32_t* __restrict__ cnt) {
    __shared__ int part[3][kThreads / 64];
    const int p = blockIdx.x, c = blockIdx.y;
    const float* Pc = P + (size_t)c * (N + 1) * 4;
    const float ctr[3] = {Pc[p * 4], Pc[p * 4 + 1], Pc[p * 4 + 2]};
    int n[3] = {0, 0, 0};
    for (int q = threadIdx.x; q <= N; q += kThreads) {
        const float d = pt_dist(ctr, Pc + q * 4);
        n[0] += !(d > r0); n[1] += !(d > r1); n[2] += !(d > r2);
    }
    for (int s = 0; s < 3; ++s) {
        int v = n[s];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0) part[s][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        int v = 0;
        for (int w = 0; w < kThreads / 64; ++w) v += part[threadIdx.x][w];
        cnt[((size_t)c * 3 + threadIdx.x) * (N + 1) + p] = v;
    }
}

// exclusive scan over p for each (c, s); total[c*3+s]
__global__ __launch_bounds__(1024) void pt_scan_kernel(const int32_t* __restrict__ cnt, int32_t* __restrict__ off,
                                                       int32_t* __restrict__ total, int N) {
    __shared__ int32_t part[1024];
    const int cs = blockIdx.x, t = threadIdx.x, n = N + 1;
    const int per = (n + 1023) / 1024;
    const int lo = t * per, hi = min(lo + per, n);
    const int32_t* c = cnt + (size_t)cs * n;
    int s = 0;
    for (int i = lo; i < hi; ++i) s += c[i];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = (t >= o) ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int i = lo; i < hi; ++i) { off[(size_t)cs * n + i] = run; run += c[i]; }
    if (t == 1023) total[cs] = part[1023];
}

// one workgroup per (p, c), scale s: map row + pair list, q ascending.  row0 = first table row of this (c, s).
__global__ __launch_bounds__(64) void pt_fill_kernel(const float* __restrict__ P, int N, float r2, const int32_t* __restrict__ off,
                                                     int row0, int32_t* __restrict__ map, uint32_t* __restrict__ pairs) {
    const int p = blockIdx.x, c = blockIdx.y, lane = threadIdx.x;
    const float* Pc = P + (size_t)c * (N + 1) * 4;
    const float ctr[3] = {Pc[p * 4], Pc[p * 4 + 1], Pc[p * 4 + 2]};
    int32_t* mrow = map + ((size_t)c * (N + 1) + p) * (N + 1);
    int pos = row0 + off[p];
    for (int q0 = 0; q0 <= N; q0 += 64) {
        const int q = q0 + lane;
        const bool in = q <= N && !(pt_dist(ctr, Pc + q * 4) > r2);
        const unsigned long long m = __ballot(in);
        if (q <= N) {
            const int row = pos + __popcll(m & ((1ull << lane) - 1ull));
            mrow[q] = in ? row : -1;
            if (in) pairs[row] = (uint32_t)p | ((uint32_t)q << 16);
        }
        pos += __popcll(m);
    }
}

// layer 1 of the scale on the pair rows: h1[row][ch] = relu(w . (P[q] - P[p]) + b), pn2_group_kernel's expression
__global__ void pt_l1_kernel(const float* __restrict__ Pc, const uint32_t* __restrict__ pairs, const float* __restrict__ w1x,
                             float* __restrict__ h1, int C1, int rows) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * C1) return;
    const int row = t / C1, ch = t - row * C1;
    const uint32_t pq = pairs[row];
    const float* x = Pc + (size_t)(pq >> 16) * 4;
    const float* c = Pc + (size_t)(pq & 0xffff) * 4;
    const float v0 = x[0] - c[0], v1 = x[1] - c[1], v2 = x[2] - c[2];
    const f32x4 w1 = *reinterpret_cast<const f32x4*>(w1x + ch * 4);
    const float h = fmaf(w1[2], v2, fmaf(w1[1], v1, w1[0] * v0)) + w1[3] + 0.f;
    h1[t] = fmaxf(h, 0.f);
}

// sa1 of coalition b, group s, scale with table `feat`: max over the group's true members of the pair rows
struct GatherArgs {
    const int32_t* fps;        // (B,S) centroid indices into the masked cloud
    const int16_t* idx;        // (B,S,K)
    const int32_t* cnt;        // (B,S)
    const int32_t* n_unique;   // (B)
    const int32_t* region_id;  // (nc,N)
    const uint64_t* keep;      // (B)
    const int32_t* cloud_of;   // (B) or null
    const int32_t* map;        // (nc,N+1,N+1)
    const float* feat;         // (rows,C3)
    const uint64_t* touch;     // (nc,N+1) regions with a member inside the ball of centroid p; 0 = not a "simple" ball (or no region table)
    const float* regtab;       // (nc,N+1,64,C3) per centroid and region: max over the region's members of the pair rows
    float* out;                // (B,S,ldo) at the scale's column offset
    int ldo, N, S, K, C3, nclouds, B;
};

// ---- region-reduced pair tables ---------------------------------------------------------------------------------------
// A group's output is a max over the pair rows of its kept members, and "kept" is a property of a member's REGION.  For a
// centroid p whose ball (a) does not contain the cloud centre - so no masked point is ever a member - and (b) holds at most
// K points - so the K-lowest-indices cap never cuts - the members of ANY coalition are exactly the kept points inside, and
// the output is max over the kept regions r of  T[p][r] = max over the members of region r of row(p, member).  T depends
// on the source cloud only: it is built once per call (pt_regtab_kernel) and a group then reads one row per kept region that
// touches its ball (r = 0.4: ~4 rows instead of ~32 member rows; the same rows for every coalition, so they stay in L2).
// Max is exact and order-free: bit-identical to the member walk (tested with tuning key 5 = 21, which disables the tables).
// Balls around the centre (6 % at r = 0.4) and over-full balls keep the member walk.
constexpr int kRegSlots = 64;

__global__ __launch_bounds__(kThreads) void pt_regtab_kernel(const float* __restrict__ feat, const uint32_t* __restrict__ pairs,
                                                             const int32_t* __restrict__ off, const int32_t* __restrict__ cntp,
                                                             int row0, const int32_t* __restrict__ region_id, int N, int K, int C3,
                                                             float* __restrict__ regtab, uint64_t* __restrict__ touch) {
    extern __shared__ unsigned tab_s[];                 // [kRegSlots][C3] non-negative floats as unsigned (order-preserving)
    __shared__ unsigned long long touch_s;
    __shared__ int centre_in;
    const int p = blockIdx.x, t = threadIdx.x;
    const int n = cntp[p], rbase = row0 + off[p];
    const int per = C3 / 4, lanes_j = kThreads / per;
    if (t == 0) { touch_s = 0ull; centre_in = 0; }
    for (int e = t; e < kRegSlots * C3; e += kThreads) tab_s[e] = 0u;
    __syncthreads();
    const int c4 = t % per;
    for (int j = t / per; j < n; j += lanes_j) {
        const int row = rbase + j;
        const int q = (int)(pairs[row] >> 16);
        if (q >= N) { if (c4 == 0) centre_in = 1; continue; }
        const int r = region_id[q];
        if ((unsigned)r >= (unsigned)kRegSlots) continue;      // a point of no region is masked in every coalition
        if (c4 == 0) atomicOr(&touch_s, 1ull << r);
        const f32x4 v = reinterpret_cast<const f32x4*>(feat)[(size_t)row * per + c4];
        unsigned* dst = tab_s + r * C3 + c4 * 4;
        atomicMax(dst + 0, __float_as_uint(v[0])); atomicMax(dst + 1, __float_as_uint(v[1]));
        atomicMax(dst + 2, __float_as_uint(v[2])); atomicMax(dst + 3, __float_as_uint(v[3]));
    }
    __syncthreads();
    const bool simple = !centre_in && n <= K;
    const unsigned long long tm = simple ? touch_s : 0ull;
    if (t == 0) touch[p] = tm;
    if (!simple) return;
    float* dst = regtab + (size_t)p * kRegSlots * C3;
    for (int e = t; e < kRegSlots * C3; e += kThreads)
        if ((tm >> (e / C3)) & 1ull) dst[e] = __uint_as_float(tab_s[e]);
}

// Two passes per group (its C3 / 4 lanes sit in one wave).  Pass 1: the lanes split the member list, each resolves its
// members to table rows (member index -> region bit -> row map: three dependent loads, now side by side over the lanes
// instead of one after the other per member) and appends the rows of kept members to a compact list in LDS; masked
// members are all the same point (the centre), whose row is appended once.  Pass 2: every lane walks the compact list
// with eight row loads in flight.  (The single loop - index, bit, map, row per member, one member at a time - was bound by
// the latency of that chain: 8.2 ms per 3300-coalition step for the three scales.)
__global__ __launch_bounds__(kThreads) void pt_gather_kernel(GatherArgs a) {
    __shared__ int32_t rows[kThreads / 16][128 + 4];   // per group: compact row list (K <= 128), padded to a multiple of 4
    const int per = a.C3 / 4;                       // float4 lanes per group (16 or 32)
    const int gpb = kThreads / per;                 // groups per workgroup
    const int t = threadIdx.x, gl = t / per, c4 = t - gl * per;
    // XCD-aware: a coalition's groups on one XCD (its index lists and the cloud's table rows stay in one L2)
    const int wgs_per_cloud = (a.S + gpb - 1) / gpb;
    const int slot = blockIdx.x >> 3;
    const int b = iq::xcd_cloud(blockIdx.x, wgs_per_cloud, a.B);
    if (b >= a.B) return;
    const int s = (slot % wgs_per_cloud) * gpb + gl;
    if (s >= a.S || s >= a.n_unique[b]) return;     // duplicate centroids are filled from group 0 afterwards
    const int c = a.cloud_of ? a.cloud_of[b] : (a.nclouds == 1 ? 0 : b);
    const uint64_t k = a.keep[b];
    const int32_t* rid = a.region_id + (size_t)c * a.N;
    const int pi = a.fps[(size_t)b * a.S + s];
    const int p = iq::keep_bit(k, rid[pi]) ? pi : a.N;
    const f32x4* F = reinterpret_cast<const f32x4*>(a.feat);
    f32x4 m = {0.f, 0.f, 0.f, 0.f};                 // rows are post-ReLU (>= 0) and every group has >= 1 member
    constexpr int U = 8;
    if (a.touch) {                                  // a simple ball: one row per kept region that reaches into it
        unsigned long long regs = a.touch[(size_t)c * (a.N + 1) + p] & k;
        if (regs) {
            const f32x4* T = reinterpret_cast<const f32x4*>(a.regtab) + ((size_t)c * (a.N + 1) + p) * kRegSlots * per + c4;
            bool more = true;
            while (more) {                          // four rows in flight; a used-up mask repeats its last row (max is idempotent)
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    v[u] = T[(size_t)__builtin_ctzll(regs) * per];
                    const unsigned long long nx = regs & (regs - 1);
                    if (nx) regs = nx; else more = false;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    m[0] = fmaxf(m[0], v[u][0]); m[1] = fmaxf(m[1], v[u][1]); m[2] = fmaxf(m[2], v[u][2]); m[3] = fmaxf(m[3], v[u][3]);
                }
            }
            *reinterpret_cast<f32x4*>(a.out + ((size_t)b * a.S + s) * a.ldo + c4 * 4) = m;
            return;
        }
    }
    const int32_t* mrow = a.map + ((size_t)c * (a.N + 1) + p) * (a.N + 1);
    const int16_t* mem = a.idx + ((size_t)b * a.S + s) * a.K;
    const int n = a.cnt[(size_t)b * a.S + s];
    int32_t* list = rows[gl];
    // ---- pass 1 ----
    const int lane = t & 63, gbase = lane - c4;     // first lane of this group inside its wave
    const uint64_t gmask = (per == 64 ? ~0ull : ((1ull << per) - 1ull)) << gbase;
    int count = 0;
    bool masked = false;
    for (int j0 = 0; j0 < n; j0 += per) {           // n is the same for all lanes of the group
        const int j = j0 + c4;
        bool kept = false;
        int row = -1;
        if (j < n) {
            const int qi = mem[j];
            kept = iq::keep_bit(k, rid[qi]);
            masked = masked || !kept;
            if (kept) row = mrow[qi];
        }
        const bool ok = row >= 0;                   // a kept member's pair always has a row (same predicate as the ball query)
        const uint64_t m = __ballot(ok) & gmask;
        if (ok) list[count + __popcll(m & ((1ull << lane) - 1ull))] = row;
        count += __popcll(m);
    }
    const bool any_masked = (__ballot(masked) & gmask) != 0;
    if (any_masked) {                               // the centre, once
        const int row = mrow[a.N];
        if (row >= 0) {
            if (c4 == 0) list[count] = row;
            ++count;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- pass 2 ----
    for (int j = 0; j < count; j += U) {
        int r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = list[min(j + u, count - 1)];   // the tail repeats the last row (max is idempotent)
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = F[(size_t)r[u] * per + c4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            m[0] = fmaxf(m[0], v[u][0]); m[1] = fmaxf(m[1], v[u][1]); m[2] = fmaxf(m[2], v[u][2]); m[3] = fmaxf(m[3], v[u][3]);
        }
    }
    *reinterpret_cast<f32x4*>(a.out + ((size_t)b * a.S + s) * a.ldo + c4 * 4) = m;
}


struct Ws2 {
    int32_t *fps1, *nu1, *fps2, *nu2;
    float *nx1;            // (B,512,3)
    int16_t* idx1[3];      // (B,512,K)
    int32_t* cnt1[3];      // (B,512)
    int32_t* cnt2[3];      // (B,128)
    int32_t* bstart;       // (B,513) block table of the scale being processed
    uint16_t* bmap;        // (B,8192): 512 groups x 128 / 8 blocks at most
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
    for (int q = 0; q < 3; ++q) { s.cnt1[q] = (int32_t*)take(b * 512 * 4); s.cnt2[q] = (int32_t*)take(b * 128 * 4); }
    s.bstart = (int32_t*)take(b * 513 * 4);
    s.bmap = (uint16_t*)take(b * 8192 * 2);
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

namespace {

// The network on B materialised clouds xyz (B,N,3).  tab != null: sa1 scales with tab->use[q] come from pair tables.
int run_pn2(const iq_pointnet2_weights* w, const float* xyz, float* logits, const Ws2& s, int B, int N, hipStream_t st,
            const PairTabs* tab, const GatherArgs* gat) {
    int rc;
    constexpr int S1 = 512, S2 = 128, F1 = 320, LD3 = 648;

    // ---- sa1 ---------------------------------------------------------------------------------------
    if ((rc = iq::launch_fps(xyz, s.fps1, s.nu1, B, N, S1, st))) return rc;
    hipLaunchKernelGGL(gather_xyz_kernel, dim3((B * S1 + 255) / 256), dim3(256), 0, st, xyz, s.fps1, s.nx1, 3, N, S1, B * S1);
    if ((rc = iq::check_launch("gather_xyz_kernel"))) return rc;
    if (tab && tab->ball_bits[0]) {
        BitBallArgs ba{};
        for (int q = 0; q < 3; ++q) { ba.bits[q] = tab->ball_bits[q]; ba.idx[q] = s.idx1[q]; ba.cnt[q] = s.cnt1[q]; ba.K[q] = w->sa1[q].nsample; }
        ba.kept = tab->kept_bits; ba.fps = s.fps1; ba.n_unique = s.nu1; ba.cloud_of = gat->cloud_of;
        ba.N = N; ba.S = S1; ba.B = B; ba.nclouds = gat->nclouds;
        for (int q = 0; q < 3; ++q) ba.touch[q] = tab->use[q] ? tab->touch[q] : nullptr;
        hipLaunchKernelGGL(pn2_bitball_kernel, dim3((S1 + kThreads - 1) / kThreads, B), dim3(kThreads), 0, st, ba);
        if ((rc = iq::check_launch("pn2_bitball_kernel"))) return rc;
    } else if ((rc = launch_ball(xyz, s.nx1, 3, w->sa1, 3, s.idx1, s.cnt1, nullptr, B, N, S1, st))) {
        return rc;
    }
    // the grouped kernel merges straddling groups with an atomic max (zero-initialised output); the table gather stores
    const bool all_tables = tab && tab->use[0] && tab->use[1] && tab->use[2];
    if (!all_tables && hipMemsetAsync(s.l1, 0, (size_t)B * S1 * F1 * sizeof(float), st) != hipSuccess)
        return iq::fail(IQ_ELAUNCH, "iq_pointnet2_forward: memset failed");
    int col = 0;
    for (int q = 0; q < 3; ++q) {
        IQ_REQUIRE(w->sa1[q].nsample <= 128 && w->sa2[q].nsample <= 128, "pointnet2: nsample > 128");
        if (tab && tab->use[q]) {  // sa1 of this scale = gather-max over the pair table
            GatherArgs g = *gat;
            g.fps = s.fps1; g.idx = s.idx1[q]; g.cnt = s.cnt1[q]; g.n_unique = s.nu1;
            g.map = tab->map[q]; g.feat = tab->feat[q]; g.C3 = tab->c3[q];
            g.touch = tab->touch[q]; g.regtab = tab->regtab[q];
            g.out = s.l1 + col; g.ldo = F1; g.N = N; g.S = S1; g.K = w->sa1[q].nsample; g.B = B;
            const int gpb = kThreads / (g.C3 / 4);
            iq::ProfileSpan span(iq::kSlotPrepool, st);
            hipLaunchKernelGGL(pt_gather_kernel, dim3((unsigned)((B + 7) / 8 * 8 * ((S1 + gpb - 1) / gpb))), dim3(kThreads), 0, st, g);
            if ((rc = iq::check_launch("pt_gather_kernel"))) return rc;
            col += w->sa1[q].l3.cout;
            continue;
        }
        GroupArgs a{};
        a.xyz = xyz; a.ldx = 3; a.new_xyz = s.nx1; a.ldc = 3; a.idx = s.idx1[q]; a.cnt = s.cnt1[q];
        a.block_start = s.bstart; a.blockmap = s.bmap;
        a.U = nullptr; a.ldu = 0;
        a.out = s.l1 + col; a.ldo = F1; a.N = N; a.S = S1;
        iq::ProfileSpan span(iq::kSlotPrepool, st);
        if ((rc = launch_group(w->sa1[q], a, s.nu1, B, st))) return rc;
        col += w->sa1[q].l3.cout;
    }
    IQ_REQUIRE(col == F1, "pointnet2: sa1 output channels %d != 320", col);
    hipLaunchKernelGGL(fill_dup_rows_kernel, dim3(S1, B), dim3(64), 0, st, s.l1, F1, S1, 0, F1, s.nu1);
    if ((rc = iq::check_launch("fill_dup_rows_kernel"))) return rc;

    // ---- sa2 ---------------------------------------------------------------------------------------
    if ((rc = iq::launch_fps(s.nx1, s.fps2, s.nu2, B, S1, S2, st))) return rc;
    if (hipMemsetAsync(s.a3, 0, (size_t)B * S2 * LD3 * sizeof(float), st) != hipSuccess)
        return iq::fail(IQ_ELAUNCH, "iq_pointnet2_forward: memset failed");
    hipLaunchKernelGGL(gather_xyz_kernel, dim3((B * S2 + 255) / 256), dim3(256), 0, st, s.nx1, s.fps2, s.a3, LD3, S1, S2, B * S2);
    if ((rc = iq::check_launch("gather_xyz_kernel"))) return rc;
    if ((rc = launch_ball(s.nx1, s.a3, LD3, w->sa2, 3, s.idx2, s.cnt2, s.nu1, B, S1, S2, st))) return rc;
    if ((rc = iq::launch_linear(s.l1, F1, w->sa2_u, s.U, F1, B * S1, 0, st))) return rc;  // U = W_f f_p + b (all scales)
    col = 0;
    int ucol = 0;
    for (int q = 0; q < 3; ++q) {
        GroupArgs a{};
        a.xyz = s.nx1; a.ldx = 3; a.new_xyz = s.a3; a.ldc = LD3; a.idx = s.idx2[q]; a.cnt = s.cnt2[q];
        a.block_start = s.bstart; a.blockmap = s.bmap;
        a.U = s.U + ucol; a.ldu = F1;
        a.out = s.a3 + 3 + col; a.ldo = LD3; a.N = S1; a.S = S2;
        iq::ProfileSpan span(iq::kSlotFstn, st);
        if ((rc = launch_group(w->sa2[q], a, s.nu2, B, st, true, w->sa2_l2_bf3[q], w->sa2_l3_bf3[q]))) return rc;
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

}  // namespace

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
    iq::ProfileSpan call_span(iq::kSlotCall, st);
    return run_pn2(w, xyz, logits, s, B, N, st, nullptr, nullptr);
}

// ---- coalitions -------------------------------------------------------------------------------------------------------
namespace {

constexpr int kPairFan = 192;  // table capacity per source cloud and scale: kPairFan * (N + 1) pair rows

struct WsC {
    float* X;          // (B,N,3) masked clouds
    float* P;          // (nc,N+1,4)
    int32_t *cntp, *off, *total;   // (nc*3*(N+1)) x2, (nc*3)
    int32_t* map[3];   // (nc,(N+1)^2)
    uint32_t* pairs;   // (cap) of the scale being built
    float *h1, *h2;    // (cap, C1 / C2) of the scale being built
    float* feat[3];    // (cap, C3)
    uint32_t* ball_bits[3];   // (nc, N+1, 33)
    uint32_t* kept_bits;      // (B, 32)
    float* regtab[3];         // (nc, N+1, 64, C3) region-reduced pair rows (only the rows of touched regions are ever written or read)
    uint64_t* touch[3];       // (nc, N+1)
    size_t bytes;
};

WsC carve_c(void* base, int B, int nc, int N) {
    WsC s{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = iq::align_up(off + bytes, 256);
        return reinterpret_cast<char*>(base) + o;
    };
    const size_t n1 = (size_t)N + 1, cap = (size_t)nc * kPairFan * n1;
    s.X = (float*)take((size_t)B * N * 3 * 4);
    s.P = (float*)take((size_t)nc * n1 * 4 * 4);
    s.cntp = (int32_t*)take((size_t)nc * 3 * n1 * 4);
    s.off = (int32_t*)take((size_t)nc * 3 * n1 * 4);
    s.total = (int32_t*)take((size_t)nc * 3 * 4);
    for (int q = 0; q < 3; ++q) s.map[q] = (int32_t*)take((size_t)nc * n1 * n1 * 4);
    s.pairs = (uint32_t*)take(cap * 4);
    s.h1 = (float*)take(cap * 64 * 4);
    s.h2 = (float*)take(cap * 96 * 4);
    const int c3[3] = {64, 128, 128};
    for (int q = 0; q < 3; ++q) s.feat[q] = (float*)take(cap * c3[q] * 4);
    for (int q = 0; q < 3; ++q) s.ball_bits[q] = (uint32_t*)take((size_t)nc * n1 * kBallWords * 4);
    s.kept_bits = (uint32_t*)take((size_t)B * 32 * 4);
    for (int q = 0; q < 3; ++q) s.regtab[q] = (float*)take((size_t)nc * n1 * kRegSlots * c3[q] * 4);
    for (int q = 0; q < 3; ++q) s.touch[q] = (uint64_t*)take((size_t)nc * n1 * 8);
    s.bytes = off;
    return s;
}

}  // namespace

extern "C" size_t iq_pointnet2_coalitions_workspace_bytes(int B, int nclouds, int N) {
    if (B < 0 || nclouds < 1 || N < 1) return 0;
    return iq::align_up(carve2(nullptr, B, nullptr).bytes, 256) + carve_c(nullptr, B, nclouds, N).bytes;
}

extern "C" int iq_pointnet2_coalitions(const iq_pointnet2_weights* w, const float* clouds, const float* centers,
                                       const int32_t* region_id, const uint64_t* keep, const int32_t* cloud_of,
                                       float* logits, void* workspace, size_t workspace_bytes, int B, int nclouds, int N,
                                       iq_stream_t stream) {
    IQ_REQUIRE(B >= 0 && nclouds >= 1 && nclouds <= 64, "iq_pointnet2_coalitions: B=%d nclouds=%d", B, nclouds);
    IQ_REQUIRE(w && clouds && centers && region_id && (B == 0 || (keep && logits)), "iq_pointnet2_coalitions: null pointer");
    IQ_REQUIRE(N >= 1 && N <= 4096, "iq_pointnet2_coalitions: N=%d", N);
    IQ_REQUIRE(cloud_of || nclouds == 1 || nclouds == B, "iq_pointnet2_coalitions: cloud_of required when 1 < nclouds != B");
    if (B == 0) return IQ_OK;
    const size_t base_bytes = iq::align_up(carve2(nullptr, B, w).bytes, 256);
    const size_t need = base_bytes + carve_c(nullptr, B, nclouds, N).bytes;
    if (!workspace || workspace_bytes < need)
        return iq::fail(IQ_EWORKSPACE, "iq_pointnet2_coalitions: workspace %zu < %zu bytes", workspace_bytes, need);
    Ws2 s = carve2(workspace, B, w);
    WsC t = carve_c(reinterpret_cast<char*>(workspace) + base_bytes, B, nclouds, N);
    hipStream_t st = iq::as_stream(stream);
    int rc;
    iq::ProfileSpan call_span(iq::kSlotCall, st);
    const int n1 = N + 1;

    hipLaunchKernelGGL(pn2_mask_kernel, dim3((B * N + 255) / 256), dim3(256), 0, st, clouds, centers, region_id, keep, cloud_of,
                       t.X, N, B, nclouds);
    hipLaunchKernelGGL(pt_points_kernel, dim3((nclouds * n1 + 255) / 256), dim3(256), 0, st, clouds, centers, t.P, N, nclouds);
    float r2[3];
    for (int q = 0; q < 3; ++q) r2[q] = (float)((double)w->sa1[q].radius * (double)w->sa1[q].radius);
    IQ_REQUIRE(r2[0] <= r2[1] && r2[1] <= r2[2], "iq_pointnet2_coalitions: sa1 radii must ascend");
    hipLaunchKernelGGL(pt_count_kernel, dim3(n1, nclouds), dim3(kThreads), 0, st, t.P, N, r2[0], r2[1], r2[2], t.cntp);
    hipLaunchKernelGGL(pt_scan_kernel, dim3(nclouds * 3), dim3(1024), 0, st, t.cntp, t.off, t.total, N);
    if ((rc = iq::check_launch("pt_scan_kernel"))) return rc;
    // the one host round trip of the call: pair counts decide, per scale, between the table and the grouped MLP
    int32_t totals[64 * 3];
    if (hipMemcpyAsync(totals, t.total, (size_t)nclouds * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return iq::fail(IQ_ELAUNCH, "iq_pointnet2_coalitions: reading the pair counts failed");
    const size_t cap = (size_t)nclouds * kPairFan * n1;
    PairTabs tab{};
    for (int q = 0; q < 3; ++q) {
        const iq_pn2_scale& sc = w->sa1[q];
        size_t rows = 0;
        for (int c = 0; c < nclouds; ++c) rows += (size_t)totals[c * 3 + q];
        tab.use[q] = rows <= cap && sc.l2.cin <= 64 && sc.l2.cout <= 96 && sc.l3.cout % 4 == 0 && sc.l3.cout <= 128 &&
                     kThreads % (sc.l3.cout / 4) == 0 && sc.l3.cin == sc.l2.cout && N < 65535;
        tab.map[q] = t.map[q]; tab.feat[q] = t.feat[q]; tab.c3[q] = sc.l3.cout;
        if (!tab.use[q]) continue;
        int row0 = 0;
        for (int c = 0; c < nclouds; ++c) {  // rows of cloud c: [row0, row0 + totals)
            hipLaunchKernelGGL(pt_fill_kernel, dim3(n1, 1), dim3(64), 0, st, t.P + (size_t)c * n1 * 4, N, r2[q],
                               t.off + ((size_t)c * 3 + q) * n1, row0, t.map[q] + (size_t)c * n1 * n1, t.pairs);
            const int nr = totals[c * 3 + q];
            if (nr > 0) {
                hipLaunchKernelGGL(pt_l1_kernel, dim3((unsigned)(((size_t)nr * sc.l2.cin + 255) / 256)), dim3(256), 0, st,
                                   t.P + (size_t)c * n1 * 4, t.pairs + row0, sc.w1x, t.h1 + (size_t)row0 * sc.l2.cin, sc.l2.cin, nr);
            }
            row0 += nr;
        }
        if ((rc = iq::check_launch("pt_fill_kernel"))) return rc;
        if (row0 > 0) {
            if ((rc = iq::launch_linear(t.h1, sc.l2.cin, sc.l2, t.h2, sc.l2.cout, row0, 1, st))) return rc;
            if ((rc = iq::launch_linear(t.h2, sc.l2.cout, sc.l3, t.feat[q], sc.l3.cout, row0, 1, st))) return rc;
        }
        if (iq::tuning(iq::kTuneExperiment) != 21) {   // region-reduced rows of this scale (21: member walk only, A/B and tests)
            int r0c = 0;
            for (int c = 0; c < nclouds; ++c) {
                hipLaunchKernelGGL(pt_regtab_kernel, dim3(n1), dim3(kThreads), (size_t)kRegSlots * sc.l3.cout * 4, st, t.feat[q], t.pairs,
                                   t.off + ((size_t)c * 3 + q) * n1, t.cntp + ((size_t)c * 3 + q) * n1, r0c, region_id + (size_t)c * N, N,
                                   sc.nsample, sc.l3.cout, t.regtab[q] + (size_t)c * n1 * kRegSlots * sc.l3.cout, t.touch[q] + (size_t)c * n1);
                r0c += totals[c * 3 + q];
            }
            if ((rc = iq::check_launch("pt_regtab_kernel"))) return rc;
            tab.touch[q] = t.touch[q]; tab.regtab[q] = t.regtab[q];
        }
    }
    if (tab.use[0] && tab.use[1] && tab.use[2] && N <= 1024 && iq::tuning(iq::kTuneExperiment) != 13) {
        // sa1's ball query by bit operations on the row maps just built (tuning key 5 = 13: ball_query_kernel)
        for (int q = 0; q < 3; ++q) {
            hipLaunchKernelGGL(pt_bits_kernel, dim3(n1, nclouds), dim3(64), 0, st, t.map[q], t.ball_bits[q], n1);
            tab.ball_bits[q] = t.ball_bits[q];
        }
        hipLaunchKernelGGL(pn2_kept_bits_kernel, dim3(B), dim3(64), 0, st, region_id, keep, cloud_of, t.kept_bits, N, nclouds);
        if ((rc = iq::check_launch("pn2_kept_bits_kernel"))) return rc;
        tab.kept_bits = t.kept_bits;
    }
    GatherArgs g{};
    g.region_id = region_id; g.keep = keep; g.cloud_of = cloud_of; g.nclouds = nclouds;
    return run_pn2(w, t.X, logits, s, B, N, st, &tab, &g);
}
