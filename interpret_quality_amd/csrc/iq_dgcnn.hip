// DGCNN / GCNN forward (models/dgcnn.py:12-194) on materialised (masked) clouds.
//
// EdgeConv restructuring (exact algebra): with W = [W_a | W_b] and BN(eval) y = s.z + t,
//     max_j LeakyReLU(s.(W_a (x_j - x_i) + W_b x_i) + t) = LeakyReLU( max_j P_j + Q_i ),
//     P = (s.W_a) x,  Q = (s.(W_b - W_a)) x + t,
// because the x_i terms are constant over the neighbourhood and LeakyReLU is increasing.  One GEMM
// over the N points per layer (2*Cout outputs) and a gather-max replace the 20x larger edge tensor.
//
// kNN (index-valued): -|x_i|^2 - (-2 x_i.x_j) - |x_j|^2 evaluated in the reference's order, inner
// products on the fp32 MFMA.  Keys run along the accumulator registers and queries along the lanes,
// so every lane keeps the running top-20 of its query in registers; ties between exact duplicates
// (masked points) are harmless because duplicates carry identical features.
//
// Ragged batches.  A masked cloud holds its kept points plus M copies of the centre, and every copy sees the same
// neighbourhoods and produces the same features in every layer.  iq_dgcnn_coalitions therefore never materialises the
// N rows: cloud b becomes D_b = kept + 1 rows - the centre once, carrying the multiplicity min(M, 20) with which it can
// appear in a top-20 (applied when knn_kernel writes the neighbour lists) - padded to a multiple of 32 with dead rows
// whose |x|^2 is +inf so that no query selects them.  Max-pooling and the neighbourhood max are set operations, so they
// are unchanged; the mean pool weights the centre by M.  Exact, and the kNN work drops
// with the square of the kept fraction.  All kernels below run on the ragged layout (row offsets per cloud); the dense
// forward is the special case D_b = N.
#include "iq_bf3.h"
#include "iq_common.h"
#include "iq_mfma.h"
#include "iq_profile.h"
#include "iq_srclist.h"
#include "iq_topk.h"

#include <type_traits>

namespace {

constexpr int kThreads = 256;
constexpr int kK = 20;  // K_FOR_DGCNN, tools/final_util.py:19
constexpr float kNearTie = 1e-5f;  // a boundary gap below this fraction of the summed terms' magnitude is re-ranked exactly
constexpr int kRefineMaxRows = 1056;  // rows of one cloud whose exact distances fit the refinement kernel's LDS
constexpr int kRoundLanes = 32;  // an insert round of the kNN selection runs once this many lanes have a queued candidate

// ---- pad xyz (B,N,3) -> (B,N,8) -------------------------------------------------------------------
// (B,N,3) -> (B,Np,8), Np = N rounded up to 32; rows N..Np-1 of a cloud are dead (zero; rownorm gives them +inf)
__global__ void pad_xyz_kernel(const float* __restrict__ xyz, float* __restrict__ out, int B, int N, int Np) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * Np) return;
    const int b = t / Np, i = t - b * Np;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 a = z;
    if (i < N) {
        const float* src = xyz + ((size_t)b * N + i) * 3;
        a = (f32x4){src[0], src[1], src[2], 0.f};
    }
    reinterpret_cast<f32x4*>(out)[(size_t)t * 2] = a;
    reinterpret_cast<f32x4*>(out)[(size_t)t * 2 + 1] = z;
}

// ---- ragged layout ---------------------------------------------------------------------------------------------
struct Ragged {
    const int32_t* roff;       // (B+1) first row of cloud b; roff[B] = total rows (all multiples of 32)
    const int32_t* nkept;      // (B) kept points = rows [0, nkept)
    const int32_t* ncopy;      // (B) min(masked points, 20): multiplicity of the centre row (row nkept) in a top-20; 0 = no centre row
    const int32_t* row_cloud;  // (rows) cloud of a row
    float* row_w;              // (rows) pooling weight: kept 1, the centre row M, dead rows 0
};

// dense forward: D_b = N
__global__ void dg_dense_layout_kernel(int32_t* __restrict__ roff, int32_t* __restrict__ nkept, int32_t* __restrict__ ncopy,
                                       int32_t* __restrict__ row_cloud, float* __restrict__ row_w, int B, int N, int Np) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < B * Np) { row_cloud[t] = t / Np; if (row_w) row_w[t] = (t % Np) < N ? 1.f : 0.f; }
    if (t <= B) roff[t] = t * Np;
    if (t < B) { nkept[t] = N; ncopy[t] = 0; }
}

// coalitions: kept points of coalition b = points whose region bit is set in keep[b]
__global__ __launch_bounds__(64) void dg_count_kernel(const int32_t* __restrict__ region_id, const uint64_t* __restrict__ keep,
                                                      const int32_t* __restrict__ cloud_of, int32_t* __restrict__ nkept,
                                                      int32_t* __restrict__ ncopy, int32_t* __restrict__ dpad, int N, int nclouds) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int c = cloud_of ? cloud_of[b] : (nclouds == 1 ? 0 : b);
    const uint64_t k = keep[b];
    const int32_t* rid = region_id + (size_t)c * N;
    int n = 0;
    for (int i = lane; i < N; i += 64) n += (int)iq::keep_bit(k, rid[i]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) n += __shfl_xor(n, o, 64);
    if (lane == 0) {
        const int copies = min(N - n, kK);        // multiplicity of the centre row in a top-20 (0: nothing masked)
        nkept[b] = n;
        ncopy[b] = copies;
        dpad[b] = (n + (copies > 0 ? 1 : 0) + 31) & ~31;
    }
}

// exclusive scan of dpad -> roff (single workgroup; B is at most a few thousand per call)
__global__ __launch_bounds__(1024) void dg_scan_kernel(const int32_t* __restrict__ dpad, int32_t* __restrict__ roff, int B) {
    __shared__ int32_t part[1024];
    const int t = threadIdx.x;
    const int per = (B + 1023) / 1024;
    const int lo = t * per, hi = min(lo + per, B);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += dpad[i];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = (t >= o) ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int i = lo; i < hi; ++i) { roff[i] = run; run += dpad[i]; }
    if (t == 1023) roff[B] = part[1023];
}

// rows of coalition b: kept points in index order, then the centre (once), then dead rows; (.,8) padded xyz
__global__ __launch_bounds__(64) void dg_compact_kernel(const float* __restrict__ clouds, const float* __restrict__ centers,
                                                        const int32_t* __restrict__ region_id, const uint64_t* __restrict__ keep,
                                                        const int32_t* __restrict__ cloud_of, const int32_t* __restrict__ roff,
                                                        const int32_t* __restrict__ nkept, const int32_t* __restrict__ ncopy,
                                                        float* __restrict__ x0, int32_t* __restrict__ row_cloud,
                                                        float* __restrict__ row_w, int N, int nclouds,
                                                        int16_t* __restrict__ src /*(B,Np) source point of a row, or null*/, int Np) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int c = cloud_of ? cloud_of[b] : (nclouds == 1 ? 0 : b);
    const uint64_t k = keep[b];
    const int32_t* rid = region_id + (size_t)c * N;
    const float* xyz = clouds + (size_t)c * N * 3;
    const int base = roff[b], end = roff[b + 1];
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    int pos = 0;
    for (int i0 = 0; i0 < N; i0 += 64) {
        const int i = i0 + lane;
        const bool kept = i < N && iq::keep_bit(k, rid[i]);
        const unsigned long long m = __ballot(kept);
        if (kept) {
            const int row = base + pos + __popcll(m & ((1ull << lane) - 1ull));
            const f32x4 a = {xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2], 0.f};
            reinterpret_cast<f32x4*>(x0)[(size_t)row * 2] = a;
            reinterpret_cast<f32x4*>(x0)[(size_t)row * 2 + 1] = z;
            if (src) src[(size_t)b * Np + (row - base)] = (int16_t)i;
        }
        pos += __popcll(m);
    }
    const int live = nkept[b] + (ncopy[b] > 0 ? 1 : 0);   // ONE centre row; its multiplicity is applied by knn_kernel
    const f32x4 ctr = {centers[c * 3], centers[c * 3 + 1], centers[c * 3 + 2], 0.f};
    for (int row = base + nkept[b] + lane; row < end; row += 64) {
        reinterpret_cast<f32x4*>(x0)[(size_t)row * 2] = (row < base + live) ? ctr : z;
        reinterpret_cast<f32x4*>(x0)[(size_t)row * 2 + 1] = z;
    }
    const int nk = nkept[b];
    if (src && lane == 0 && ncopy[b] > 0) src[(size_t)b * Np + nk] = (int16_t)N;   // the centre row
    for (int row = base + lane; row < end; row += 64) {
        row_cloud[row] = b;
        const int rl = row - base;
        row_w[row] = rl < nk ? 1.f : (rl == nk && ncopy[b] > 0 ? (float)(N - nk) : 0.f);
    }
}

// ---- xx[i] = sum_c x[i][c]^2 in channel order (torch.sum(x**2, dim=1)) ------------------------------
// 64 rows per workgroup; 32-channel slices go through a small LDS tile (coalesced float4 reads, many workgroups per
// CU) and each row is still summed sequentially over its channels by one lane - the order the golden kNN sets were
// produced with.  Dead (padding) rows get +inf: as keys they then score -inf and are never selected.
// `planes` (C = 64 / 128, or null): the same rows as three bf16 terms (iq_bf3.h) in MFMA fragment order, for knn_kernel<.., BF3>:
// fragment (term, 32-row tile, k-step of 16) = 1 KiB at ((term * term_tiles + tile) * C / 16 + k-step) KiB, lane (row & 31) +
// 32 * (k-half) at 16 bytes - a key tile's operand is then ONE fully coalesced 1 KiB load per term and k-step, and the query tile's
// operand is the same image.  Written from the registers the norm's loads already hold (6 bytes per value, no second read).
__global__ __launch_bounds__(kThreads) void rownorm_kernel(const float* __restrict__ x, int ldx, int C, float* __restrict__ xx,
                                                           Ragged rg, int B, unsigned char* __restrict__ planes, int term_tiles) {
    __shared__ float tile[64 * 33];
    const int rows = rg.roff[B];
    const int r0 = blockIdx.x * 64;
    if (r0 >= rows) return;
    const int tid = threadIdx.x;
    const int KS = C >> 4;
    const size_t term_bytes = (size_t)term_tiles * KS * 1024;
    float s = 0.f;
    for (int c0 = 0; c0 < C; c0 += 32) {
        const int cw = min(32, C - c0);
        if (cw == 32) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = tid + kThreads * i, row = e >> 3, c4 = e & 7;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (r0 + row < rows) v = *reinterpret_cast<const f32x4*>(x + (size_t)(r0 + row) * ldx + c0 + c4 * 4);
                float* t = tile + row * 33 + c4 * 4;
                t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3];
                if (planes && r0 + row < rows) {
                    const int c = c0 + c4 * 4, r = r0 + row;
                    unsigned char* d = planes + ((size_t)(r >> 5) * KS + (c >> 4)) * 1024 + ((r & 31) + 32 * ((c >> 3) & 1)) * 16 + (c & 7) * 2;
                    u32x2 h, m, l;
                    split4(v, h, m, l);
                    *reinterpret_cast<u32x2*>(d) = h;
                    *reinterpret_cast<u32x2*>(d + term_bytes) = m;
                    *reinterpret_cast<u32x2*>(d + 2 * term_bytes) = l;
                }
            }
        } else {
            for (int e = tid; e < 64 * cw; e += kThreads) {
                const int row = e / cw, c = e - row * cw;
                tile[row * 33 + c] = (r0 + row < rows) ? x[(size_t)(r0 + row) * ldx + c0 + c] : 0.f;
            }
        }
        __syncthreads();
        if (tid < 64) {
            const float* p = tile + tid * 33;
            for (int c = 0; c < cw; ++c) s += p[c] * p[c];
        }
        __syncthreads();
    }
    const int r = r0 + tid;
    if (tid >= 64 || r >= rows) return;
    const int b = rg.row_cloud[r];
    xx[r] = (r - rg.roff[b] < rg.nkept[b] + (rg.ncopy[b] > 0 ? 1 : 0)) ? s : INFINITY;
}

// ---- layer-1 graph of a coalition from the SOURCE cloud's sorted neighbour lists ---------------------------------------
// In xyz space the distance between two kept points does not depend on the coalition, and neither does the centre (the mean
// of the whole cloud): only the SET of candidates does.  So per source cloud (a few per call, against thousands of
// coalitions) every point's complete neighbour list - all other points and the centre, nearest first, by the very
// distances knn_kernel<8> computes - is built once (iq_srclist.h: sl_rows -> sl_dist -> sl_sort), and the 20 nearest of a
// coalition's query are the first entries of its source point's list that the coalition keeps, the centre entry counting
// min(M, 20) times (dg_walk_kernel: a few dozen 2-byte entries per query instead of a D x D distance matrix and a top-k).
// The same neighbours as knn_kernel<8> on the compact rows (same arithmetic for every distance, ties by point index).
// neighbour lists of the rows of coalition b from the sorted lists of its source cloud.  One wave = 64 query rows of one
// coalition.  The wave first builds the coalition's kept-point bitmap (N bits) and the prefix counts of its 32-bit words in
// LDS, so that "is point p kept" and "which compact row is it" (= the number of kept points before p) are two LDS reads
// and a popcount; the only global reads of the walk are the list entries themselves, eight per 16-byte load.
__global__ __launch_bounds__(64) void dg_walk_kernel(const int16_t* __restrict__ sorted, const int16_t* __restrict__ src,
                                                     const int32_t* __restrict__ region_id, const uint64_t* __restrict__ keep,
                                                     const int32_t* __restrict__ cloud_of, int16_t* __restrict__ idx, Ragged rg,
                                                     int N, int Np, int Nsl, int nclouds, int as_addr) {
    __shared__ unsigned bits[kWalkMaxN / 32];
    __shared__ int pre[kWalkMaxN / 32];
    const int b = blockIdx.y, lane = threadIdx.x, r = blockIdx.x * 64 + lane;
    const int base = rg.roff[b], D = rg.roff[b + 1] - base;
    if (blockIdx.x * 64 >= D) return;                 // wave-uniform
    const int c = cloud_of ? cloud_of[b] : (nclouds == 1 ? 0 : b);
    const uint64_t k = keep[b];
    const int32_t* rid = region_id + (size_t)c * N;
    for (int i0 = 0; i0 < kWalkMaxN; i0 += 64) {      // kept bitmap, 64 points per step
        const int i = i0 + lane;
        const unsigned long long m = __ballot(i < N && iq::keep_bit(k, rid[min(i, N - 1)]));
        if (lane == 0) { bits[i0 >> 5] = (unsigned)m; bits[(i0 >> 5) + 1] = (unsigned)(m >> 32); }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < kWalkMaxN / 32) {                      // exclusive prefix of the word popcounts (32 words: a wave scan)
        const int cnt = __popc(bits[lane]);
        int run = cnt;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            const int v = __shfl_up(run, o, 64);
            if (lane >= o) run += v;
        }
        pre[lane] = run - cnt;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (r >= D) return;
    const int nk = rg.nkept[b], mult = rg.ncopy[b];
    const int live = nk + (mult > 0 ? 1 : 0);
    auto code = [&](int row) { return (int16_t)(as_addr ? (row << 6) | (((row >> 2) & 3) << 4) : row); };
    int16_t* o = idx + ((size_t)base + r) * kK;
    if (r >= live) {   // dead padding row: never read as a neighbour list that matters; keep it valid
#pragma unroll
        for (int q = 0; q < kK; ++q) o[q] = code(0);
        return;
    }
    const int16_t* list = sorted + ((size_t)c * (N + 1) + src[(size_t)b * Np + r]) * Nsl;
    int weight = 0, n = 0;
    for (int j0 = 0; j0 <= N && weight < kK; j0 += 8) {
        const uint4 chunk = *reinterpret_cast<const uint4*>(list + j0);   // Nsl is a multiple of 8: whole chunks
        const unsigned wds[4] = {chunk.x, chunk.y, chunk.z, chunk.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int p = (int)((wds[e >> 1] >> (16 * (e & 1))) & 0xffffu);
            if (j0 + e > N || weight >= kK) continue;
            if (p == N) {
                if (mult > 0) { o[n++] = code(nk); weight += mult; }     // the centre stands for min(M, 20) identical points
            } else {
                const unsigned wbits = bits[p >> 5];
                if ((wbits >> (p & 31)) & 1u) {
                    o[n++] = code(pre[p >> 5] + __popc(wbits & ((1u << (p & 31)) - 1u)));
                    ++weight;
                }
            }
        }
    }
    const int16_t fill = code(mult > 0 ? nk : r);   // (only a cloud with fewer than 20 distinct rows gets here: it has a centre)
    for (; n < kK; ++n) o[n] = fill;
}

// ---- kNN ------------------------------------------------------------------------------------------
// (A variant with two query tiles per wave and a v_permlane32_swap of the accumulator halves - one top-K list per
// query instead of two half lists - was measured and dropped: 8 % faster at C = 64, 7 % slower at C = 8, spills at 128.)
__device__ unsigned long long g_knn_dbg[8];  // tuning key 4 = 3: [0] rounds, [1] busy lanes summed over rounds, [2] waves | [3] flagged queries, [4] re-ranked queries, [6] of which zero-gap (all rows ranked); [5], [7] unused
__global__ void knn_dbg_fetch_kernel(unsigned long long* dst) {
    for (int i = 0; i < 3; ++i) { dst[i] = g_knn_dbg[i]; g_knn_dbg[i] = 0; }   // (the op-level probe reads the selection counters only)
}

// One wave = one workgroup = 32 queries.  Key fragments come straight from L1/L2 (a key row's C floats are contiguous, so
// the four k-blocks of a 128-B line are consumed back to back), one k-block ahead of the MFMAs; nothing is shared
// between waves, so there is no barrier in the loop and waves drift freely past each other's selection rounds.  (The
// first version staged key tiles through LDS for 4 waves: with a barrier per 32-key tile and 2 workgroups per CU the
// MFMA + staging skeleton alone took 53 of the 91 ms of DGCNN's kNN.)
// REFINE (the feature-space graphs, C = 64 / 128): the list holds kK + 1 entries.  The 21st nearest tells how well the fp32
// expanded-form distances separate the neighbourhood from the rest: a query whose boundary gap (or, in a coalition's compact
// layout, the gap between the weighted centre row and a row next to it) is below kNearTie of the magnitude of the summed terms
// is FLAGGED: near_tie[row] holds its 21st candidate (else -1), its neighbour list the other 20 as raw rows, and
// knn_refine_kernel re-ranks the 21 in exact arithmetic (bit 20 of the word: the gap is exactly zero - identical rows, as in
// a dense masked cloud, may hide further candidates - so all rows are ranked).
// BF3 (round 5, C = 64 / 128): the inner products on the bf16 matrix pipe, float32-accurate - keys and queries as three bf16 terms
// each (the fragment image rownorm_kernel writes), six exact products per k-step of 16 accumulated in float32 (iq_bf3.h): 192 matrix
// cycles per 16 k instead of 512, a key tile's operand three coalesced 1 KiB loads per k-step.  The scores differ from the fp32
// MFMA's in the last bits only (both within float32 rounding of the exact inner product, the bf16x3 ones closer), which is what
// REFINE's margin covers; the selection, the flags and the epilogue are untouched.
template <int C, bool REFINE, int PF = 1, int QCAP = 16, bool BF3 = false>
__global__ __launch_bounds__(64, BF3 ? (C == 128 ? 2 : 3) : C == 128 ? (QCAP < 16 ? 3 : 2) : (QCAP < 16 ? 4 : 3)) void knn_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ xx,
                                                 int16_t* __restrict__ idx, int32_t* __restrict__ near_tie, Ragged rg, int B,
                                                 int tiles_per_cloud, int dbg, const unsigned char* __restrict__ planes, int term_tiles) {
    static_assert(!BF3 || (REFINE && C % 16 == 0 && C >= 32), "bf16x3 distances: feature-space graphs only");
    constexpr int KB = C / 8;
    constexpr int KS = C / 16;                     // BF3: k-steps of 16
    constexpr int KL = REFINE ? kK + 1 : kK;   // list length
    // REFINE: the integer list with tagged indices (iq_topk.h: TaggedTopK) - its 32-ulp buckets lie far inside the band that
    // knn_refine_kernel re-ranks exactly; the exact-only kernel keeps the packed fp64 list
    using Sel = std::conditional_t<REFINE, TaggedTopK<KL, QCAP>, QueuedTopK<KL, 16>>;
    constexpr int kSelLds = REFINE ? TaggedTopK<KL, QCAP>::kLdsBytes : 17 * 64 * 8;   // queue: QCAP slots per lane + the overflow slot of push()
    __shared__ __attribute__((aligned(16))) unsigned char queue[kSelLds];
    __shared__ __attribute__((aligned(16))) float kxs[2 * 64];   // |key|^2 of two pairs of key tiles (double-buffered)
    const int lane = threadIdx.x;
    // workgroups go round-robin over the 8 XCDs: give each XCD whole clouds, so that the ~17 waves which stream the same
    // keys share one L2 (and often one L1) instead of pulling the cloud into all eight
    const int slot = blockIdx.x >> 3;
    const int b = iq::xcd_cloud(blockIdx.x, tiles_per_cloud, B);
    if (b >= B) return;
    const int base = rg.roff[b];
    const int N = rg.roff[b + 1] - base;          // padded rows of this cloud (multiple of 32)
    const int q0 = (slot % tiles_per_cloud) * 32; // this wave's 32 queries
    if (q0 >= N) return;
    const float* xb = x + (size_t)base * ldx;
    const float* xxb = xx + base;
    const int fl = lane & 31, fh = lane >> 5;

    // B operand: queries, stationary in registers
    f32x4 qf[BF3 ? 1 : KB];
    bf16x8 qx[BF3 ? KS : 1][3];
    __amdgpu_buffer_rsrc_t prs[3];                  // BF3: this cloud's tiles of the fragment image, one resource per term
    if constexpr (BF3) {
        const size_t term_bytes = (size_t)term_tiles * KS * 1024;
        const unsigned char* pb = planes + (size_t)(base >> 5) * KS * 1024;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            prs[e] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(pb + e * term_bytes), 0, (N >> 5) * KS * 1024, 0x00020000);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                qx[ks][e] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(prs[e], lane * 16, ((q0 >> 5) * KS + ks) * 1024, 0));
        }
    } else {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
            qf[kb] = *reinterpret_cast<const f32x4*>(xb + (size_t)(q0 + fl) * ldx + 8 * kb + 4 * fh);
    }
    const float xxq = xxb[q0 + fl];

    Sel sel;
    if constexpr (REFINE) sel.init(queue);
    else sel.init(reinterpret_cast<double*>(queue));
    const int ntiles = N / 32;
    // Key fragments by raw buffer loads (iq_mfma.h: WBuf): resource on this cloud's rows, ONE loop-invariant per-lane offset
    // (row fl of the tile, k offset 4 fh), the tile / k-block position is a scalar offset.  |key|^2 of two tiles (64 floats)
    // comes with ONE coalesced load per pair of tiles and goes through a wave-private LDS strip, from which each lane reads
    // the 16 values of its accumulator rows (4 ds_read_b128) - instead of 4 more vector-memory instructions per tile.  VMEM
    // instructions per tile: C/8 + 1/2 (was C/8 + 4); each costs the SIMD tens of issue cycles next to the MFMAs.
    const WBuf kb_buf = wbuf_make(xb, lane);
    const int kvoff = (fl * ldx + 4 * fh) * 4;                   // bytes
    const int row_bytes = ldx * 4;
    const WBuf kx_buf = wbuf_make(xxb, lane);
    auto key_frag = [&](int t, int kb) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(kb_buf.rsrc, kvoff, (32 * t * row_bytes + 32 * kb), 0));
    };
    auto kx_pair = [&](int tp) {                                   // |key|^2 of tiles 2 tp, 2 tp + 1 (clamped to the cloud's rows)
        const int row = min(64 * tp + lane, N - 1);
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(kx_buf.rsrc, row * 4, 0, 0));
    };
    // Key tiles are taken in PAIRS (one |key|^2 strip per pair), the pairs in zigzag order around the query tile's own pair:
    // p0, p0 + 1, p0 - 1, p0 + 2, ... and straight on once one side is used up.  Whatever locality the row order has (the
    // compact layout of a coalition keeps its points in the source cloud's Morton order) then brings a query's near keys
    // first: the filter threshold is tight after a few tiles and most later candidates never reach the queue.  On rows in
    // random order the statistics are those of any fixed order.
    const int npairs = (ntiles + 1) >> 1;
    const int p0 = (q0 >> 5) >> 1;
    const int side = min(p0, npairs - 1 - p0);                     // pairs available on both sides
    auto pair_at = [&](int k) {                                    // k-th pair of the order (k clamped to the last)
        k = min(k, npairs - 1);
        if (k <= 2 * side) return (k & 1) ? p0 + ((k + 1) >> 1) : p0 - (k >> 1);
        return (npairs - 1 - p0 > p0) ? p0 + (k - side) : p0 - (k - side);
    };
    int k = 0, pcur = p0, j = 0;                                    // pair number in the order, its index, tile within the pair
    // key fragments PF k-blocks ahead of the MFMAs (PF = 1: the next one only, 256 MFMA cycles - less than an L2 round trip)
    static_assert(PF >= 1 && PF <= KB && (PF & (PF - 1)) == 0 && KB % PF == 0, "prefetch depth");
    f32x4 ring[PF];
    // BF3: the three terms of a key tile's k-step, PF k-steps (192 matrix cycles each) ahead
    auto key_frag3 = [&](int t, int ks) {
        const int o = (t * KS + ks) * 1024;
        return B3{__builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(prs[0], lane * 16, o, 0)),
                  __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(prs[1], lane * 16, o, 0)),
                  __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(prs[2], lane * 16, o, 0))};
    };
    B3 ring3[BF3 ? PF : 1];
    if constexpr (BF3) {
#pragma unroll
        for (int i = 0; i < PF; ++i) ring3[i] = key_frag3(2 * p0, i);
    } else {
#pragma unroll
        for (int i = 0; i < PF; ++i) ring[i] = key_frag(2 * p0, i);
    }
    kxs[lane] = kx_pair(p0);                                       // wave-private: no barrier (one wave per workgroup)
    float kx_next = kx_pair(pair_at(1));
    // ONE loop (a small state machine) instead of rounds nested in the tile loop: the 20-entry list is then carried
    // by a single loop and stays in 40 registers; nested, the allocator kept copies per loop level (300 registers).
    float d[16];
    int dbg_rounds = 0, dbg_work = 0;
    int tcur = 0, half = 2;  // half == 2: the current tile is used up
    for (;;) {
        const unsigned long long busy = __ballot(sel.cnt > 0);
        const bool last = half == 2 && k == npairs;
        if (busy != 0 && (last || __popcll(busy) >= kRoundLanes || __any(sel.cnt > (REFINE ? QCAP : 16) - 8))) {
            if ((dbg & 3) == 3) { ++dbg_rounds; dbg_work += __popcll(busy); }
            sel.round(lane);
            continue;
        }
        if (last) break;
        if (half == 2) {  // distances of key tile t = 2 pcur + j
            const int t = 2 * pcur + j;
            f32x4 kx[4];
            const float* ks = kxs + 64 * (k & 1) + 32 * j + 4 * fh;   // rows 8 i + 4 fh + (0..3) of this tile
#pragma unroll
            for (int i = 0; i < 4; ++i) kx[i] = *reinterpret_cast<const f32x4*>(ks + 8 * i);
            const bool pair_done = j == 1 || t + 1 >= ntiles;
            const int pnext = pair_at(k + 1);
            const int tn = pair_done ? 2 * pnext : t + 1;              // (after the last tile: a valid tile, unused)
            f32x16 acc = {0};
            if constexpr (BF3) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const B3 a = ring3[ks & (PF - 1)];
                    ring3[ks & (PF - 1)] = ks + PF < KS ? key_frag3(t, ks + PF) : key_frag3(tn, ks + PF - KS);
                    acc = mfma_bf3_tr(a, qx[ks], acc);     // keys = A operand (accumulator rows), queries = B (lanes)
                }
            } else {
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    const f32x4 a = ring[kb & (PF - 1)];
                    ring[kb & (PF - 1)] = kb + PF < KB ? key_frag(t, kb + PF) : key_frag(tn, kb + PF - KB);
                    acc = mfma4(a, qf[kb], acc);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // -xx - inner - xx^T with inner = -2 * matmul (models/dgcnn.py:13-15).  -2 * acc is exact, so (-xx) - inner rounds
                // once, exactly like the fused form 2 * acc + (-xx): one instruction instead of two
                d[r] = __builtin_fmaf(2.f, acc[r], -kx[r >> 2][r & 3]) - xxq;
            }
            tcur = t;
            if (pair_done) {   // the pair is used up: publish the next pair's strip, request the one after it
                kxs[64 * ((k + 1) & 1) + lane] = kx_next;
                kx_next = kx_pair(pair_at(k + 2));
                ++k; pcur = pnext; j = 0;
            } else {
                j = 1;
            }
            half = 0;
            if ((dbg & 3) == 2) { half = 2; continue; }
        }
        const float thr_f = sel.union_threshold();   // bound on the 20th largest of both half-waves' keys (iq_topk.h)
        // queue the candidates of this half of the tile (accumulator registers 8 half .. 8 half + 7)
        const int ib = tcur * 32 + 4 * fh + 16 * half;
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const float v = half ? d[8 + rr] : d[rr];
            sel.push(v, ib + (rr & 3) + 8 * (rr >> 2), thr_f, lane);
        }
        ++half;
        if ((dbg & 3) == 1) sel.cnt = 0;
    }
    if ((dbg & 3) == 3 && lane == 0) {  // diagnostic counters (tools/knn_probe.py)
        atomicAdd(&g_knn_dbg[0], (unsigned long long)dbg_rounds);
        atomicAdd(&g_knn_dbg[1], (unsigned long long)dbg_work);
        atomicAdd(&g_knn_dbg[2], 1ull);
    }
    // the list in its packed fp64 form (value | key index) for the merge of the two half-waves and the epilogue
    QueuedTopK<KL, 16> top;
    if constexpr (REFINE) sel.export_packed(top.v, lane);
    else top = sel;
    top.merge_halves();  // each half-wave saw half of the keys of every tile
    const int centre = rg.nkept[b], mult = rg.ncopy[b];
    bool flagged = false;
    float margin = 0.f;
    if constexpr (REFINE) {
        // the smallest of the 21 (the 21st nearest) goes to slot kK; slots 0..kK-1 are the (unordered) approximate top-20
        double v21 = top.v[0];
#pragma unroll
        for (int q = 1; q <= kK; ++q) v21 = fmin(v21, top.v[q]);
        // (ONE slot trades places with the last: with fewer than 21 live rows several slots hold -inf, and replacing them all
        // copied the last slot's row into each of them - the ranks behind it, which decide the cut when fewer than 20 points are
        // masked, were then counted against the copies.  Found on 21- to 32-point clouds in round 5.)
        const double last = top.v[kK];
        bool moved = false;
#pragma unroll
        for (int q = 0; q < kK; ++q) {
            const bool hit = !moved && top.v[q] == v21;
            top.v[q] = hit ? last : top.v[q];
            moved = moved || hit;
        }
        top.v[kK] = v21;
        double t20 = top.v[0];
#pragma unroll
        for (int q = 1; q < kK; ++q) t20 = fmin(t20, top.v[q]);
        // magnitude of the terms the expanded form sums: |q|^2 + |k|^2 + 2 |q.k| <= 2 (|q|^2 + |k|^2), |k|^2 <= 2 (|q|^2 + |q - k|^2)
        // (+ two buckets of the tagged list: its values are the lower ends of 32-ulp buckets)
        margin = kNearTie * 2.f * (xxq + fmaxf(-(float)t20, 0.f)) + 7.7e-6f * fabsf((float)t20);
        const bool live_q = q0 + fl < centre + (mult > 0 ? 1 : 0);
        flagged = live_q && (float)(t20 - v21) < margin;   // fewer than 21 live rows: v21 = -inf, never flagged by this rule
    }
    // The masked points of a coalition are ONE row (index nkept) that stands for `mult` = min(M, 20) identical points.  The
    // list holds the 20 nearest DISTINCT rows; of those, a row farther than the centre with `rank` rows ahead of it sits
    // at position rank - 1 + mult of the reference's top-k, so it is a neighbour only if rank <= 20 - mult.  Rows that
    // fall out are replaced by the centre (a duplicate neighbour does not change a max).  mult == 20, the usual case:
    // every row behind the centre falls out.  Rows filled from the dead padding (fewer than 20 live rows) are behind the
    // centre and fall out the same way.
    bool drop[kK];
#pragma unroll
    for (int q = 0; q < kK; ++q) drop[q] = false;
    if (mult > 0) {
        double vc = 0.0;
        bool has = false;
#pragma unroll
        for (int q = 0; q < kK; ++q)
            if (top.index(q) == centre) { vc = top.v[q]; has = true; }
        if constexpr (REFINE) {
            // which side of the weighted centre row a row lies on decides whether it is a neighbour at all
            if (has) {
                bool close = false;
#pragma unroll
                for (int q = 0; q < kK; ++q) close = close || (top.index(q) != centre && fabsf((float)(top.v[q] - vc)) < margin);
                const bool live_q = q0 + fl < centre + 1;
                flagged = (mult >= kK ? false : flagged) || (live_q && close);
            }
        }
        if (mult >= kK) {
#pragma unroll
            for (int q = 0; q < kK; ++q) drop[q] = has && top.v[q] < vc;
        } else if (__any(has)) {   // fewer than 20 masked points: ranks among the 20 (400 comparisons, rare)
#pragma unroll
            for (int q = 0; q < kK; ++q) {
                int rank = 0;
#pragma unroll
                for (int j = 0; j < kK; ++j) rank += top.v[j] > top.v[q] ? 1 : 0;
                drop[q] = has && top.v[q] < vc && rank > kK - mult;
            }
        }
    }
    if (REFINE && (dbg & 3) == 3) {
        const int nf = __popcll(__ballot(flagged && fh == 0));
        if (lane == 0) atomicAdd(&g_knn_dbg[3], (unsigned long long)nf);
    }
    if (fh == 0) {
        // dbg & 8: the consumer is edge_fused_kernel, which wants the byte address of the row's (swizzled) first float4 in
        // its LDS slice instead of the row index: row * 64 + ((row >> 2) & 3) * 16 (< 65536 for rows < 1024)
        const bool as_addr = (dbg & 8) != 0;
        int16_t* o = idx + ((size_t)base + q0 + fl) * kK;
        if constexpr (REFINE) {
            // a flagged query hands its 21 candidates over as they are (raw rows, nothing dropped, -1 for an empty slot)
            int word = -1;
            if (flagged) {
                bool zero_gap = false;
                double t20 = top.v[0];
#pragma unroll
                for (int q = 1; q < kK; ++q) t20 = fmin(t20, top.v[q]);
                zero_gap = (float)t20 == (float)top.v[kK];
                word = (top.v[kK] > -INFINITY ? top.index(kK) : 0x7fff) | (zero_gap ? 1 << 20 : 0);
            }
            near_tie[(size_t)base + q0 + fl] = word;
#pragma unroll
            for (int q = 0; q < kK; ++q) {
                const int row = drop[q] ? centre : top.index(q);
                const int raw = top.v[q] > -INFINITY ? top.index(q) : -1;
                o[q] = (int16_t)(flagged ? raw : (as_addr ? (row << 6) | (((row >> 2) & 3) << 4) : row));
            }
        } else {
#pragma unroll
            for (int q = 0; q < kK; ++q) {
                const int row = drop[q] ? centre : top.index(q);
                o[q] = (int16_t)(as_addr ? (row << 6) | (((row >> 2) & 3) << 4) : row);
            }
        }
    }
}

// ---- exact re-ranking of the queries knn_kernel flagged ---------------------------------------------------------------------
// -|q|^2 + 2 q.k - |k|^2 in float32 carries a rounding error of ~1e-6 of |q|^2 + |k|^2, far above the distance itself for
// near neighbours; where the 20th and 21st nearest are closer than that, which of them is a neighbour is decided by
// rounding noise - for the reference's float32 path exactly as for this one, and a flipped neighbour moves a DGCNN logit
// at the 1e-3 level.  Flagged queries are therefore ranked again by -sum_c (q_c - k_c)^2 with the differences in float32
// (relative error 6e-8 of the DIFFERENCE) squared and summed in float64: the order the reference's float64 run finds,
// except where two distances agree to ~1e-7 of themselves.  One wave per group of 64 consecutive rows; the few flagged
// rows of the group are handled one after the other by the whole wave: lane l owns rows l, l + 64, ... of the cloud.
// Same weighted selection as knn_kernel's epilogue (the centre row of a compact coalition counts mult times).
// Which rows can belong to the exact top-20 of a flagged query?  Only rows the float32 scores place within the band of
// the 20th: the 20 of the list, the 21st, and - if THREE rows crowd into a band of 1e-5 - a 22nd, which is ignored (under 1 %
// of the flagged queries, themselves 0.7 % of all).  So the re-ranking reads 21 rows, not the cloud: one lane per candidate
// computes the float64 sum, every candidate's rank by (distance, row index) comes from one pass over the candidates with
// v_readlane, and the cut is knn_kernel's (the centre row of a compact coalition counts mult times).  A first version scored
// all rows of the cloud per flagged query: 25 GB of L2 reads per 12 000-coalition step, 5 ms.  A gap of exactly zero means
// identical rows (the masked points of a DENSE masked cloud fill the list with copies and hide the rows behind them): those
// queries rank all live rows by their float64 scores (LDS, best first) - the dense forward's price, not the coalition path's.
// One wave per 64 consecutive rows; its flagged queries one after the other.
template <int C>
__global__ __launch_bounds__(64) void knn_refine_kernel(const float* __restrict__ x, int ldx, const int32_t* __restrict__ near_tie,
                                                        int16_t* __restrict__ idx, Ragged rg, int B, int as_addr, int dbg) {
    __shared__ double dist[kRefineMaxRows];
    __shared__ __attribute__((aligned(16))) float qs[C];
    const int lane = threadIdx.x;
    const int rows = rg.roff[B];
    const int r0 = blockIdx.x * 64;
    if (r0 >= rows) return;
    const int my_word = r0 + lane < rows ? near_tie[r0 + lane] : -1;
    unsigned long long todo = __ballot(my_word >= 0);
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto shfl_xor_f64 = [](double v, int o) {
        const long long b = __double_as_longlong(v);
        const int lo = __shfl_xor((int)(b & 0xffffffffll), o), hi = __shfl_xor((int)(b >> 32), o);
        return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    };
    auto readlane_f64 = [](double v, int src) {   // src wave-uniform
        const long long b = __double_as_longlong(v);
        const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src), hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
        return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    };
    // exact score of row j against the query in qs[]: differences in float32, squares summed in float64
    auto exact_score = [&](const float* xb, int j) {
        const f32x4* kr = reinterpret_cast<const f32x4*>(xb + (size_t)j * ldx);
        double acc = 0.0;
#pragma unroll 8
        for (int c4 = 0; c4 < C / 4; ++c4) {
            const f32x4 kv = kr[c4], qv = reinterpret_cast<const f32x4*>(qs)[c4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { const double df = (double)(qv[e] - kv[e]); acc = __builtin_fma(df, df, acc); }
        }
        return -acc;
    };
    while (todo) {
        const int bit = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int row = r0 + bit;
        const int word = __shfl(my_word, bit);
        const int b = rg.row_cloud[row];
        const int base = rg.roff[b], centre = rg.nkept[b], mult = rg.ncopy[b];
        const int live = centre + (mult > 0 ? 1 : 0);
        const float* xb = x + (size_t)base * ldx;
        int16_t* o = idx + (size_t)row * kK;
        // the candidates as knn_kernel left them
        int jc = -1;
        if (lane < kK) jc = o[lane];
        else if (lane == kK) jc = (word & 0x7fff) == 0x7fff ? -1 : (word & 0x7fff);
        if (lane < C / 4) reinterpret_cast<f32x4*>(qs)[lane] = *reinterpret_cast<const f32x4*>(x + (size_t)row * ldx + 4 * lane);
        wave_sync();
        int mine = centre;   // lane n < kK ends with the n-th neighbour
        const bool full = (word >> 20) & 1;
        if (dbg && lane == 0) {
            atomicAdd(&g_knn_dbg[4], 1ull);
            atomicAdd(&g_knn_dbg[6], full ? 1ull : 0ull);
        }
        if (!full || live > kRefineMaxRows) {
            const bool valid = jc >= 0 && jc < live;
            const double dc = valid ? exact_score(xb, jc) : -INFINITY;
            const int jv = valid ? jc : 0x7fffffff;
            int rank = 0, rank_centre = 0x7fffffff;
#pragma unroll
            for (int c = 0; c <= kK; ++c) {       // candidates ahead of mine by (distance, then row index)
                const double d2 = readlane_f64(dc, c);
                const int j2 = __builtin_amdgcn_readlane(jv, c);
                rank += (d2 > dc || (d2 == dc && j2 < jv)) ? 1 : 0;
            }
            if (mult > 0) {   // where the weighted centre row stands
                const unsigned long long isc = __ballot(valid && jc == centre);
                if (isc) rank_centre = __builtin_amdgcn_readlane(rank, __builtin_ctzll(isc));
            }
            // position of my row in the multiset order: rows behind the centre are pushed back by its mult - 1 copies
            const int posn = rank + (rank > rank_centre ? mult - 1 : 0);
            const bool in = valid && posn < kK && rank < kK;
#pragma unroll
            for (int n = 0; n < kK; ++n) {        // neighbour n = the included candidate of rank n (the included ranks are a prefix)
                const unsigned long long who = __ballot(in && rank == n);
                if (who) { const int jn = __builtin_amdgcn_readlane(jc, __builtin_ctzll(who)); if (lane == n) mine = jn; }
            }
        } else {
            for (int j = lane; j < live; j += 64) dist[j] = exact_score(xb, j);
            wave_sync();
            int n_out = 0, pos = 0;
            while (pos < kK && n_out < kK) {
                double bv = -INFINITY;
                int bj = 0x7fffffff;
                for (int j = lane; j < live; j += 64) {
                    const double d = dist[j];
                    if (d > bv) { bv = d; bj = j; }   // ascending j: the first of equal values stays
                }
                double m = bv;
#pragma unroll
                for (int o2 = 32; o2 >= 1; o2 >>= 1) m = fmax(m, shfl_xor_f64(m, o2));
                if (!(m > -INFINITY)) break;          // fewer live rows than slots: the rest stays the centre
                int win = bv == m ? bj : 0x7fffffff;
#pragma unroll
                for (int o2 = 32; o2 >= 1; o2 >>= 1) win = min(win, __shfl_xor(win, o2));
                if ((win & 63) == lane) dist[win] = -INFINITY;
                if (lane == n_out) mine = win;
                ++n_out;
                pos += (win == centre && mult > 0) ? mult : 1;
                wave_sync();
            }
        }
        if (lane < kK) o[lane] = (int16_t)(as_addr ? (mine << 6) | (((mine >> 2) & 3) << 4) : mine);
        wave_sync();   // qs is rewritten for the next flagged query
    }
}

// ---- out[i][c] = LeakyReLU(max_j P[idx[i][j]][c] + Q[i][c]) ------------------------------------------
__global__ __launch_bounds__(kThreads) void gather_max_kernel(const float* __restrict__ pq, int Co,
                                                              const int16_t* __restrict__ idx, float* __restrict__ out,
                                                              int ldo, Ragged rg, int B, int wgs_per_cloud) {
    const int per = Co / 4;                               // float4 lanes per point
    // Workgroups go round-robin over the 8 XCDs: give each XCD whole clouds (cloud b -> XCD b % 8, its workgroups
    // consecutive there), so that the ~20 reads of every P row of a cloud hit ONE L2.  In plain block order the rows of a
    // cloud were pulled into all eight L2s: FETCH_SIZE showed 4-9x the bytes of the PQ matrix per launch at 57 % L2 hit
    // rate and the kernel ran at the fabric's 6.4 TB/s (profiles/r02_stream_kernels.csv).  Adjacent clouds run on
    // adjacent XCDs at the same time, so the eight row streams stay close together in memory.
    const int slot = blockIdx.x >> 3;
    const int b = iq::xcd_cloud(blockIdx.x, wgs_per_cloud, B);
    if (b >= B) return;
    const int base = rg.roff[b];
    const int pt = base + (slot % wgs_per_cloud) * (kThreads / per) + threadIdx.x / per;
    const int c4 = threadIdx.x % per;
    if (pt >= rg.roff[b + 1]) return;
    // the 20 neighbour indices of the point: 40 contiguous bytes, read as five 8-byte words
    const uint2* nbw = reinterpret_cast<const uint2*>(idx + (size_t)pt * kK);
    int nb[kK];
#pragma unroll
    for (int w = 0; w < kK / 4; ++w) {
        const uint2 v = nbw[w];
        nb[4 * w] = (int)(v.x & 0xffffu); nb[4 * w + 1] = (int)(v.x >> 16);
        nb[4 * w + 2] = (int)(v.y & 0xffffu); nb[4 * w + 3] = (int)(v.y >> 16);
    }
    // All 20 row loads of a thread are in flight together, as raw buffer loads with 32-bit offsets on a resource based at
    // this cloud's first row (one multiply-add of address arithmetic per load instead of a 64-bit chain).
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pq) + (size_t)base * (2 * Co), 0, 0x7fffffff, 0x00020000);
    const int row_bytes = 2 * Co * 4;
    f32x4 v[kK];
#pragma unroll
    for (int j = 0; j < kK; ++j)
        v[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, nb[j] * row_bytes + c4 * 16, 0, 0));
    const f32x4 q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (pt - base) * row_bytes + (per + c4) * 16, 0, 0));
    f32x4 m = v[0];
#pragma unroll
    for (int j = 1; j < kK; ++j) {
        m[0] = fmaxf(m[0], v[j][0]); m[1] = fmaxf(m[1], v[j][1]); m[2] = fmaxf(m[2], v[j][2]); m[3] = fmaxf(m[3], v[j][3]);
    }
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float y = m[e] + q[e];
        o[e] = y > 0.f ? y : 0.2f * y;
    }
    *reinterpret_cast<f32x4*>(out + (size_t)pt * ldo + c4 * 4) = o;
}

// ---- the same with the P rows of a cloud staged in LDS -----------------------------------------------------------------
// Every P row of a cloud is read ~20 times (once per query that has it among its neighbours), in no order the caches can
// use: gather_max_kernel moves 20 x the P matrix through the texture-address path and L2 (7 TB/s measured, the fabric's
// limit).  Here a workgroup owns (cloud, channel chunk): it streams its D x cw slice of P into LDS once (coalesced, every
// byte of P leaves L2 once), and the 20 reads per output come from LDS.  cw is chosen per cloud so that the slice fills the
// 64 KB budget (D <= 256: 64 channels, <= 512: 32, <= 1024: 16); the grid holds Co / 16 workgroups per cloud and the
// surplus ones of small clouds exit at once.  Two workgroups per CU: one streams while the other gathers.
// max() is order-independent on the finite values here, so the result is bit-identical to gather_max_kernel.
constexpr int kGlThreads = 512;
constexpr int kGlFloats = 16384;   // 64 KB of P slice per workgroup
constexpr int kGlMaxRows = 1024;   // D * 16 channels must fit

__global__ __launch_bounds__(kGlThreads, 4) void gather_lds_kernel(const float* __restrict__ pq, int Co,
                                                                const int16_t* __restrict__ idx, float* __restrict__ out,
                                                                int ldo, Ragged rg, int B, int wgs_per_cloud) {
    __shared__ f32x4 slice[kGlFloats / 4];
    const int slot = blockIdx.x >> 3;   // whole clouds per XCD, as in gather_max_kernel
    const int b = iq::xcd_cloud(blockIdx.x, wgs_per_cloud, B);
    if (b >= B) return;
    const int base = rg.roff[b];
    const int D = rg.roff[b + 1] - base;
    int shift = 2;                                            // log2(float4 lanes per row of the slice)
    if (D * 32 <= kGlFloats && Co >= 32) shift = 3;
    if (D * 64 <= kGlFloats && Co >= 64) shift = 4;
    const int per = 1 << shift, cw = 4 * per;
    const int chunk = slot % wgs_per_cloud;
    if (chunk * cw >= Co || D == 0) return;
    const int c0 = chunk * cw;                                // first channel of this workgroup
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pq) + (size_t)base * (2 * Co), 0, 0x7fffffff, 0x00020000);
    const int row_bytes = 2 * Co * 4;
    const int items = D << shift;                             // (row, float4 lane) pairs, <= 4096; D % 32 == 0
    const int tid = threadIdx.x;
    // the neighbour lists (40 bytes per row, five 8-byte words) and the Q values of an item; the first item's are
    // requested before the slice, every later item's one item ahead
    const __amdgpu_buffer_rsrc_t irsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(idx) + (size_t)base * kK, 0, 0x7fffffff, 0x00020000);
    struct Item { uint2 w[kK / 4]; f32x4 q; };
    auto fetch = [&](int t) {
        Item it;
        const int r = min(t, items - 1) >> shift, c4 = t & (per - 1);
#pragma unroll
        for (int w = 0; w < kK / 4; ++w) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b64(irsrc, r * (kK * 2) + 8 * w, 0, 0);
            it.w[w] = __builtin_bit_cast(uint2, v);
        }
        it.q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, r * row_bytes + (Co + c0 + 4 * c4) * 4, 0, 0));
        return it;
    };
    Item cur = fetch(tid);
    // phase 1: the slice, all (up to eight) loads of a thread in flight (items is a multiple of 128, at most 4096)
    {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = tid + u * kGlThreads;
            const int off = (t >> shift) * row_bytes + (c0 + 4 * (t & (per - 1))) * 4;
            if (t < items) v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = tid + u * kGlThreads;
            if (t < items) slice[t] = v[u];
        }
    }
    __syncthreads();
    // phase 2
    const unsigned last = (unsigned)D - 1u;
    for (int t = tid; t < items; t += kGlThreads) {
        const int r = t >> shift, c4 = t & (per - 1);
        const Item it = cur;
        if (t + kGlThreads < items) cur = fetch(t + kGlThreads);
        const f32x4 q = it.q;
        unsigned nb[kK];
#pragma unroll
        for (int w = 0; w < kK / 4; ++w) {
            const uint2 v = it.w[w];
            nb[4 * w] = v.x & 0xffffu; nb[4 * w + 1] = v.x >> 16;
            nb[4 * w + 2] = v.y & 0xffffu; nb[4 * w + 3] = v.y >> 16;
        }
        f32x4 m = slice[(min(nb[0], last) << shift) + c4];
#pragma unroll
        for (int j = 1; j < kK; ++j) {
            const f32x4 v = slice[(min(nb[j], last) << shift) + c4];   // min: rows outside the slice are never addressed
            m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
        }
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float y = m[e] + q[e];
            o[e] = y > 0.f ? y : 0.2f * y;
        }
        *reinterpret_cast<f32x4*>(out + (size_t)(base + r) * ldo + c0 + 4 * c4) = o;
    }
}

// ---- EdgeConv layer in one kernel: P/Q GEMM + neighbourhood max --------------------------------------------------------
// out[i][c] = LeakyReLU(max_j P[idx[i][j]][c] + Q[i][c]) with [P | Q] = x W^T + bias computed HERE: the (rows, 2 Co) matrix
// never goes to HBM (the separate GEMM wrote it and the gather read it back: 2 x 8 KB per row over the four layers, and
// both ran at 2-3 TB/s).  A workgroup owns (cloud, 16 output channels).  The MFMA runs TRANSPOSED: its 32 "rows" are 32
// columns of the layer - 16 of P and the 16 of Q for the same channels - and its 32 "columns" are 32 points, so one
// 32x32 tile gives a lane (point = lane & 31, half h) P and Q of 8 channels of its point:
//     accumulator i:  0-3  P[4h + i]    4-7  P[8 + 4h + (i-4)]    8-11  Q[4h + (i-8)]    12-15  Q[8 + 4h + (i-12)]
// (weights are the A operand: lane m = lane & 31 holds column m < 16 ? c0 + m : Co + c0 + m - 16 of the packed image;
// products and the order over k are those of the plain GEMM, so the results are bit-identical to launch_linear +
// gather_max_kernel).  P goes to LDS (D x 16 floats <= 64 KB), Q stays in the accumulators; after the barrier every lane
// takes the maxima of its point's 20 neighbours from LDS (two 16-byte reads per neighbour) and writes 2 x 16 bytes.
// Two workgroups per CU: the MFMA phase of one runs under the LDS phase of the other.
//
// Operands.  The 32 x Cin weight slice of the workgroup (<= 16 KB) is staged in LDS once and read from there as
// fragments.  A wave works through its point tiles one after the other, a stage = the four k-blocks of a 128-byte line
// of x; the x fragments (global) and weight fragments (LDS) of stage q + 1 are requested before the MFMAs of stage q.
// (Measured and of no effect: perfectly coalesced fake x addresses, x groups 2-4 stages ahead.)
__device__ unsigned long long g_edge_dbg[8];   // tuning key 5 = 10: shader cycles per phase summed over workgroups (wave 0), count
__global__ void edge_dbg_print_kernel(int layer) {
    const double n = (double)g_edge_dbg[5];
    printf("edge_fused layer %d: %llu workgroups; cycles per workgroup (wave 0): prologue+weights %.0f, MFMA phase %.0f, "
           "wait at slice barrier %.0f, gather %.0f, total %.0f; tiles per wave 0: %.2f\n", layer, g_edge_dbg[5],
           g_edge_dbg[0] / n, g_edge_dbg[1] / n, g_edge_dbg[2] / n, g_edge_dbg[3] / n, g_edge_dbg[4] / n, g_edge_dbg[6] / n);
    for (int i = 0; i < 8; ++i) g_edge_dbg[i] = 0;
}
__device__ __forceinline__ unsigned long long edge_clock() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

template <int KB, int NBR = kK>   // Cin / 8; NBR < kK: timing probe (half of the neighbours: the gather's instructions halved, results wrong)
__global__ __launch_bounds__(kGlThreads, 4) void edge_fused_kernel(const float* __restrict__ x, int ldx,
                                                                   const float* __restrict__ wp, const float* __restrict__ bias,
                                                                   int Co, const int16_t* __restrict__ idx,
                                                                   float* __restrict__ out, int ldo, Ragged rg, int B,
                                                                   int wgs_per_cloud, int stamps = 0) {
    constexpr int G = KB >= 4 ? 4 : KB;                       // k-blocks per group (one 128-byte line of x when KB >= 4)
    constexpr int NG = KB / G;
    __shared__ f32x4 slice[kGlFloats / 4];                    // [row][4 float4]
    __shared__ f32x4 wl[KB * 64];                             // weight fragments [kb][lane]
    unsigned long long ts[5];
    if (stamps) ts[0] = edge_clock();
    const int slot = blockIdx.x >> 3;   // whole clouds per XCD, as in gather_max_kernel
    const int b = iq::xcd_cloud(blockIdx.x, wgs_per_cloud, B);
    if (b >= B) return;
    const int base = rg.roff[b];
    const int D = rg.roff[b + 1] - base;                      // multiple of 32, <= kGlMaxRows (host)
    if (D == 0) return;
    const int c0 = (slot % wgs_per_cloud) * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
    const int p = lane & 31, h = lane >> 5;
    const int ntiles = D >> 5;
    const int nv = min(4, max(0, (ntiles - wave + 7) >> 3));  // this wave's point tiles: wave, wave + 8, ... (wave-uniform)

    // weight slice -> LDS: entry e = (kb, lane l) is the fragment element of MFMA row m = l & 31, half l >> 5
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, 0x7fffffff, 0x00020000);
    for (int e = tid; e < KB * 64; e += kGlThreads) {
        const int kb = e >> 6, l = e & 63, m = l & 31;
        const int n = m < 16 ? c0 + m : Co + c0 + m - 16;     // layer column in MFMA row m
        wl[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, ((((n >> 5) * KB + kb) * 64) + (n & 31) + (l & 32)) * 16, 0, 0));
    }
    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x) + (size_t)base * ldx, 0, 0x7fffffff, 0x00020000);
    auto xvoff = [&](int u) { return ((min(wave + 8 * u, ntiles - 1) * 32 + p) * ldx + 4 * h) * 4; };
    struct XG { f32x4 f[G]; };
    auto xgroup = [&](int voff, int g) {
        XG r;
#pragma unroll
        for (int j = 0; j < G; ++j) r.f[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, (g * G + j) * 32, 0));
        return r;
    };
    // Operand pipeline: the x group (global) and the weight group (LDS) of stage q + 1 are requested before the MFMAs of
    // stage q (a stage = G k-blocks of one tile = 4 G MFMAs); sched_group_barrier pins that order - left alone the
    // scheduler read two weight fragments, waited for them, issued 8 MFMAs, read the next two ... and the wave stood
    // still for an LDS round trip (behind the other workgroup's gather traffic) every 8 MFMAs: 50 % MFMA-busy.
    constexpr int TOTAL = 4 * NG;
    struct WGp { f32x4 f[G]; };
    auto wgroup = [&](int g) {
        WGp r;
#pragma unroll
        for (int j = 0; j < G; ++j) r.f[j] = wl[(g * G + j) * 64 + lane];
        return r;
    };
    XG xr[2];
    WGp wr[2];
    xr[0] = xgroup(xvoff(0), 0);
    const __amdgpu_buffer_rsrc_t irs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(idx) + (size_t)base * kK, 0, 0x7fffffff, 0x00020000);
    struct Nb { uint2 w[kK / 4]; };
    auto fetch_nb = [&](int u) {
        Nb v;
        const int r = min(wave + 8 * u, ntiles - 1) * 32 + p;
#pragma unroll
        for (int w = 0; w < kK / 4; ++w) v.w[w] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(irs, r * (kK * 2) + 8 * w, 0, 0));
        return v;
    };
    // bias of the 16 channels (P) and of their Q columns: wave-uniform, in SGPRs; a lane picks its 8 by h
    float bps[16], bqs[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        bps[e] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, bias[c0 + e])));
        bqs[e] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, bias[Co + c0 + e])));
    }
    __syncthreads();                                          // wl complete
    if (stamps) ts[1] = edge_clock();
    wr[0] = wgroup(0);

    // The slice is XOR-swizzled: float4 column c of row r sits at r * 4 + (c ^ ((r >> 2) & 3)).  Unswizzled, the 16 lanes
    // of a ds_read_b128 lane group (same h, random rows) all read column h: 16 of the 64 banks, a 4-way conflict at best.
    // With the swizzle a row's column lands in one of 16 slots (measured: 7 conflict cycles per LDS instruction).
    f32x4 q0[4], q1[4];                                       // Q of this lane's channels, per tile
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (u < nv) {
            f32x16 acc = {0};
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int q = u * NG + g;
                const XG xf = xr[q & 1];
                const WGp wf = wr[q & 1];
                if (q + 1 < TOTAL) {
                    const int un = (q + 1) / NG;              // compile-time after unrolling
                    xr[(q + 1) & 1] = xgroup(xvoff(min(un, nv - 1)), (q + 1) % NG);
                    wr[(q + 1) & 1] = wgroup((q + 1) % NG);
                }
#pragma unroll
                for (int j = 0; j < G; ++j) acc = mfma4(wf.f[j], xf.f[j], acc);
                if (q + 1 < TOTAL) {
                    __builtin_amdgcn_sched_group_barrier(0x020, G, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, G, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * G, 0);
            }
            const int r = (wave + 8 * u) * 32 + p;
            f32x4 p0, p1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                p0[e] = acc[e] + (h ? bps[4 + e] : bps[e]); p1[e] = acc[4 + e] + (h ? bps[12 + e] : bps[8 + e]);
                q0[u][e] = acc[8 + e] + (h ? bqs[4 + e] : bqs[e]); q1[u][e] = acc[12 + e] + (h ? bqs[12 + e] : bqs[8 + e]);
            }
            const int a = r * 4 + (h ^ ((r >> 2) & 3));
            slice[a] = p0;
            slice[a ^ 2] = p1;
        }
    }
    Nb cur = fetch_nb(0);                                     // neighbour lists of the first tile (in flight across the barrier)
    if (stamps) ts[2] = edge_clock();
    __syncthreads();                                          // slice complete
    if (stamps) ts[3] = edge_clock();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (u < nv) {
            const Nb it = cur;
            if (u + 1 < nv) cur = fetch_nb(u + 1);
            const int r = (wave + 8 * u) * 32 + p;
            // the neighbour lists hold LDS byte addresses (knn_kernel, dbg & 8): row * 64 + swizzle(row) * 16 for h = 0; this
            // lane's two float4 are at (address ^ 16 h) and 32 bytes across.  (Row indices cost 10 VALU instructions per
            // neighbour to unpack, clamp and swizzle - 200 per tile against 96 for the maxima themselves.)  Any 16-bit
            // address lies inside this workgroup's own LDS allocation.
            unsigned nb[kK];
#pragma unroll
            for (int w = 0; w < kK / 4; ++w) {
                const uint2 v = it.w[w];
                nb[4 * w] = v.x & 0xffffu; nb[4 * w + 1] = v.x >> 16;
                nb[4 * w + 2] = v.y & 0xffffu; nb[4 * w + 3] = v.y >> 16;
            }
            const char* sb = reinterpret_cast<const char*>(slice);
            const unsigned hx = (unsigned)h << 4;
            auto rd = [&](unsigned a) { return *reinterpret_cast<const f32x4*>(sb + a); };
            f32x4 m0 = rd(nb[0] ^ hx), m1 = rd(nb[0] ^ hx ^ 32u);
#pragma unroll
            for (int j = 1; j < NBR; ++j) {
                const unsigned a = nb[j] ^ hx;
                const f32x4 v0 = rd(a), v1 = rd(a ^ 32u);
#pragma unroll
                for (int e = 0; e < 4; ++e) { m0[e] = fmaxf(m0[e], v0[e]); m1[e] = fmaxf(m1[e], v1[e]); }
            }
            f32x4 o0, o1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y0 = m0[e] + q0[u][e], y1 = m1[e] + q1[u][e];
                o0[e] = y0 > 0.f ? y0 : 0.2f * y0;
                o1[e] = y1 > 0.f ? y1 : 0.2f * y1;
            }
            float* o = out + (size_t)(base + r) * ldo + c0 + 4 * h;
            *reinterpret_cast<f32x4*>(o) = o0;
            *reinterpret_cast<f32x4*>(o + 8) = o1;
        }
    }
    if (stamps && tid == 0) {
        ts[4] = edge_clock();
        atomicAdd(&g_edge_dbg[0], ts[1] - ts[0]); atomicAdd(&g_edge_dbg[1], ts[2] - ts[1]); atomicAdd(&g_edge_dbg[2], ts[3] - ts[2]);
        atomicAdd(&g_edge_dbg[3], ts[4] - ts[3]); atomicAdd(&g_edge_dbg[4], ts[4] - ts[0]); atomicAdd(&g_edge_dbg[5], 1ull);
        atomicAdd(&g_edge_dbg[6], (unsigned long long)nv);
    }
}

// ---- global max and mean pooling over the N points of each cloud: folded into conv5 (launch_linear_pool) -------------
// Kept rows count once, the centre (all copies identical) counts with its multiplicity M = N - kept (row_w).
// second stage of the pooling after launch_linear_pool: tiles of 32 rows never straddle clouds
__global__ __launch_bounds__(kThreads) void pool_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                               Ragged rg, int N, int C) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * kThreads + threadIdx.x;
    if (c >= C) return;
    const int t0 = rg.roff[b] >> 5, t1 = rg.roff[b + 1] >> 5;
    float m = -INFINITY, s = 0.f;
    for (int t = t0; t < t1; ++t) {
        const float* p = partial + (size_t)t * 2 * C;
        m = fmaxf(m, p[c]);
        s += p[C + c];
    }
    out[(size_t)b * 2 * C + c] = m;
    out[(size_t)b * 2 * C + C + c] = s / (float)N;
}

// slice_addr: write the neighbours as LDS addresses for edge_fused_kernel (rows < 1024) instead of row indices.
// near_tie: one int32 per row (rows = upper bound of the row count), scratch of the exact re-ranking (C = 64 / 128).
// planes / term_tiles: the rows as bf16x3 fragments (rownorm_kernel) - the distances then run on the bf16 matrix pipe (C = 64 / 128)
int launch_knn(const float* x, int ldx, int C, const float* xx, int16_t* idx, int32_t* near_tie, int B, int N, int rows, const Ragged& rg,
               hipStream_t st, bool slice_addr = false, const unsigned char* planes = nullptr, int term_tiles = 0) {
    const int tiles = (N + 31) / 32;
    dim3 grid((unsigned)((B + 7) / 8 * 8 * tiles));
    const int dbg = (iq::tuning(iq::kTuneKnnDebug) & 7) | (slice_addr ? 8 : 0);
    const bool refine = iq::tuning(iq::kTuneExperiment) != 20;   // 20: float32 ranking only (A/B and tests)
    const dim3 rgrid((unsigned)((rows + 63) / 64));
    if (C == 8) hipLaunchKernelGGL((knn_kernel<8, false>), grid, dim3(64), 0, st, x, ldx, xx, idx, near_tie, rg, B, tiles, dbg, planes, term_tiles);
    else if (C == 64 && !refine) hipLaunchKernelGGL((knn_kernel<64, false>), grid, dim3(64), 0, st, x, ldx, xx, idx, near_tie, rg, B, tiles, dbg, planes, term_tiles);
    else if (C == 128 && !refine) hipLaunchKernelGGL((knn_kernel<128, false>), grid, dim3(64), 0, st, x, ldx, xx, idx, near_tie, rg, B, tiles, dbg, planes, term_tiles);
    else if (planes && (C == 64 || C == 128)) {
        // two k-steps ahead, one accumulator.  Measured and not adopted (same-box A/B, 12 000-coalition step, kNN slot 20.6 ms): four
        // k-steps ahead 21.3 (fewer waves), even / odd k-steps on two accumulators 20.6, both 25.7 (spills); 16 queue slots 20.8;
        // four independent waves of one cloud per workgroup (one CU, so that one L2 read might serve several) 23.2 with their
        // own key orders and 23.2 with a common one - the distance skeleton alone 14.1 against 13.6: not the L2 either
        if (C == 64)
            hipLaunchKernelGGL((knn_kernel<64, true, 2, 12, true>), grid, dim3(64), 0, st, x, ldx, xx, idx, near_tie, rg, B, tiles, dbg, planes, term_tiles);
        else
            hipLaunchKernelGGL((knn_kernel<128, true, 2, 16, true>), grid, dim3(64), 0, st, x, ldx, xx, idx, near_tie, rg, B, tiles, dbg, planes, term_tiles);
        if (C == 64) hipLaunchKernelGGL(knn_refine_kernel<64>, rgrid, dim3(64), 0, st, x, ldx, near_tie, idx, rg, B, slice_addr ? 1 : 0, (dbg & 3) == 3);
        else hipLaunchKernelGGL(knn_refine_kernel<128>, rgrid, dim3(64), 0, st, x, ldx, near_tie, idx, rg, B, slice_addr ? 1 : 0, (dbg & 3) == 3);
    } else if (C == 64) {
        // 12 queue slots per lane instead of 16: 10 KB of LDS per wave, FOUR waves per SIMD (116 VGPRs) - the selection's VALU
        // work of one wave overlaps with the MFMAs of more neighbours: 28.0 -> 26.7 ms in a same-call A/B (5 = 44: 16 slots, three
        // waves).  Measured and not adopted: key fragments 2 / 4 k-blocks ahead instead of 1 (27.9 / 27.8 ms), 10 slots (27.0 ms),
        // the C = 128 kernel at three waves per SIMD (no change)
        if (iq::tuning(iq::kTuneExperiment) == 44)
            hipLaunchKernelGGL((knn_kernel<64, true>), grid, dim3(64), 0, st, x, ldx, xx, idx, near_tie, rg, B, tiles, dbg, planes, term_tiles);
        else
            hipLaunchKernelGGL((knn_kernel<64, true, 1, 12>), grid, dim3(64), 0, st, x, ldx, xx, idx, near_tie, rg, B, tiles, dbg, planes, term_tiles);
        hipLaunchKernelGGL(knn_refine_kernel<64>, rgrid, dim3(64), 0, st, x, ldx, near_tie, idx, rg, B, slice_addr ? 1 : 0, (dbg & 3) == 3);
    } else if (C == 128) {
        hipLaunchKernelGGL((knn_kernel<128, true>), grid, dim3(64), 0, st, x, ldx, xx, idx, near_tie, rg, B, tiles, dbg, planes, term_tiles);
        hipLaunchKernelGGL(knn_refine_kernel<128>, rgrid, dim3(64), 0, st, x, ldx, near_tie, idx, rg, B, slice_addr ? 1 : 0, (dbg & 3) == 3);
    } else return iq::fail(IQ_EUNSUPPORTED, "knn: C=%d has no kernel instantiation (8, 64, 128)", C);
    return iq::check_launch("knn_kernel");
}

struct WsD {
    float* x0;      // (B,N,8)
    float* xc;      // (B,N,512)
    float* pq;      // (B,N,512)
    float* xx;      // (B,N)
    int16_t* idx;   // (B,N,20)
    int32_t* near_tie;  // (B,N) 21st candidate of a query whose neighbourhood boundary the float32 distances cannot decide, else -1
    float* h;       // (B*N/32, 2, 1024) per-tile max / weighted sum of conv5
    float *g, *f1, *f2;
    int32_t *roff, *nkept, *ncopy, *dpad, *row_cloud;  // ragged layout
    float* row_w;
    size_t bytes;
};

WsD carve_d(void* base, int B, int N) {
    WsD s{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = iq::align_up(off + bytes, 256);
        return reinterpret_cast<char*>(base) + o;
    };
    const size_t r = (size_t)B * ((N + 31) / 32 * 32);
    s.x0 = (float*)take(r * 8 * 4);
    s.xc = (float*)take(r * 512 * 4);
    s.pq = (float*)take(r * 512 * 4);
    s.xx = (float*)take(r * 4);
    s.idx = (int16_t*)take(r * kK * 2);
    s.near_tie = (int32_t*)take(r * 4);
    s.h = (float*)take(r * 64 * 4);  // pooling partials of conv5: (rows/32, 2, 1024)
    s.g = (float*)take((size_t)B * 2048 * 4);
    s.f1 = (float*)take((size_t)B * 512 * 4);
    s.f2 = (float*)take((size_t)B * 256 * 4);
    s.roff = (int32_t*)take((size_t)(B + 1) * 4);
    s.nkept = (int32_t*)take((size_t)B * 4);
    s.ncopy = (int32_t*)take((size_t)B * 4);
    s.dpad = (int32_t*)take((size_t)B * 4);
    s.row_cloud = (int32_t*)take(r * 4);
    s.row_w = (float*)take(r * 4);
    s.bytes = off;
    return s;
}

__global__ void widen_idx_kernel(const int16_t* __restrict__ in, int32_t* __restrict__ out, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = in[t];
}

}  // namespace

extern "C" int iq_debug_knn_counters(unsigned long long* out_host) {
    IQ_REQUIRE(out_host, "iq_debug_knn_counters: null pointer");
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_knn_dbg), sizeof(zero)) != hipSuccess ||
        hipMemcpyToSymbol(HIP_SYMBOL(g_knn_dbg), zero, sizeof(zero)) != hipSuccess)
        return iq::fail(IQ_ELAUNCH, "iq_debug_knn_counters: copy failed");
    return IQ_OK;
}

extern "C" size_t iq_dgcnn_workspace_bytes(int B, int N) {
    if (B < 0 || N < 0) return 0;
    return carve_d(nullptr, B, N).bytes;
}

// Op-level kNN for tests: x (B,N,C) row-major, C in {3, 64, 128}; idx (B,N,20) int32; tmp >= B*N*84 + 16*B + 8192 bytes
// (+ B*N*C*6 for the bf16x3 operand image of C = 64 / 128: without that room the fp32-MFMA kernels run).
extern "C" int iq_knn(const float* x, int32_t* idx, void* tmp, size_t tmp_bytes, int B, int N, int C, int k,
                      iq_stream_t stream) {
    IQ_REQUIRE(x && idx && tmp, "iq_knn: null pointer");
    IQ_REQUIRE(k == kK, "iq_knn: k=%d (only k=20, tools/final_util.py:19)", k);
    IQ_REQUIRE(B >= 1 && N >= 32 && N % 32 == 0 && N <= 32767, "iq_knn: N=%d must be a multiple of 32", N);
    IQ_REQUIRE(C == 3 || C == 64 || C == 128, "iq_knn: C=%d unsupported", C);
    const size_t r = (size_t)B * N;
    IQ_REQUIRE(tmp_bytes >= r * 84 + (size_t)B * 16 + 8192, "iq_knn: tmp too small");
    hipStream_t st = iq::as_stream(stream);
    char* p = reinterpret_cast<char*>(tmp);
    size_t off = 0;
    auto take = [&](size_t bytes) { char* q = p + off; off = iq::align_up(off + bytes, 256); return q; };
    float* x0 = reinterpret_cast<float*>(take(r * 8 * 4));
    float* xx = reinterpret_cast<float*>(take(r * 4));
    int16_t* i16 = reinterpret_cast<int16_t*>(take(r * kK * 2));
    int32_t* near_tie = reinterpret_cast<int32_t*>(take(r * 4));
    int32_t* row_cloud = reinterpret_cast<int32_t*>(take(r * 4));
    int32_t* roff = reinterpret_cast<int32_t*>(take((size_t)(B + 1) * 4));
    int32_t* nkept = reinterpret_cast<int32_t*>(take((size_t)B * 4));
    int32_t* ncopy = reinterpret_cast<int32_t*>(take((size_t)B * 4));
    hipLaunchKernelGGL(dg_dense_layout_kernel, dim3((r + 256) / 256), dim3(256), 0, st, roff, nkept, ncopy, row_cloud,
                       (float*)nullptr, B, N, N);
    const Ragged rg{roff, nkept, ncopy, row_cloud, nullptr};
    const float* src = x;
    int ld = C, cpad = C;
    if (C == 3) {
        hipLaunchKernelGGL(pad_xyz_kernel, dim3((r + 255) / 256), dim3(256), 0, st, x, x0, B, N, N);
        src = x0; ld = 8; cpad = 8;
    }
    // room for the bf16x3 operand image behind the other scratch: the distances of the feature-space graphs run on the bf16 matrix
    // pipe, as in the model path (5 = 22 / 20: the fp32 MFMA kernels)
    const int knob = iq::tuning(iq::kTuneExperiment);
    unsigned char* planes = nullptr;
    if ((C == 64 || C == 128) && knob != 22 && knob != 20 && tmp_bytes >= iq::align_up(off, 256) + r * C * 6)
        planes = reinterpret_cast<unsigned char*>(take(r * C * 6));
    hipLaunchKernelGGL(rownorm_kernel, dim3((r + 63) / 64), dim3(kThreads), 0, st, src, ld, C, xx, rg, B, planes, (int)(r / 32));
    int rc = launch_knn(src, ld, cpad, xx, i16, near_tie, B, N, (int)r, rg, st, false, planes, (int)(r / 32));
    if (rc) return rc;
    hipLaunchKernelGGL(widen_idx_kernel, dim3((r * kK + 255) / 256), dim3(256), 0, st, i16, idx, r * kK);
    if (iq::tuning(iq::kTuneKnnDebug) == 3)  // diagnostic: selection statistics into the first 24 bytes of tmp
        hipLaunchKernelGGL(knn_dbg_fetch_kernel, dim3(1), dim3(1), 0, st, reinterpret_cast<unsigned long long*>(tmp));
    return iq::check_launch("iq_knn");
}

namespace {

// The network on the ragged rows described by s.roff / s.nkept / s.ncopy / s.row_cloud; s.x0 holds the padded xyz rows.
// `rows` = upper bound of the row count (grid sizes); the live count is roff[B], on the device.
// Layer-1 graph from the source clouds' sorted neighbour lists (dg_walk_kernel) instead of rownorm + knn_kernel<8>
struct WalkCtx {
    const int16_t* sorted;   // (nclouds, N+1, Nsl)
    const int16_t* src;      // (B, Np)
    const int32_t* region_id;
    const uint64_t* keep;
    const int32_t* cloud_of;
    int nclouds, Nsl;
};

// does run_network take the fused EdgeConv path?  (the walk tables live in the P/Q buffer that only the other path uses)
bool fused_path(const iq_dgcnn_weights* w, int N) {
    const int knob = iq::tuning(iq::kTuneExperiment);   // 7: GEMM + L2 gather, 8: GEMM + LDS gather (A/B and tests)
    bool fused = (N + 31) / 32 * 32 <= kGlMaxRows && knob != 7 && knob != 8;
    int ci = 8;
    for (int l = 0; l < 4; ++l) {
        const int co = w->pq[l].cout / 2;
        fused = fused && co % 16 == 0 && (ci == 8 || ci == 64 || ci == 128) && w->pq[l].cin == ci;
        ci = co;
    }
    return fused;
}

int run_network(const iq_dgcnn_weights* w, const WsD& s, float* logits, int B, int N, int rows, int fixed_graph,
                hipStream_t st, const WalkCtx* walk = nullptr) {
    const Ragged rg{s.roff, s.nkept, s.ncopy, s.row_cloud, s.row_w};
    const int32_t* live = s.roff + B;
    int rc;
    const float* src = s.x0;
    int ld = 8, cin = 8, creal = 3, col = 0;
    const int Np = (N + 31) / 32 * 32;
    const int knob = iq::tuning(iq::kTuneExperiment);   // 7: GEMM + L2 gather, 8: GEMM + LDS gather (A/B and tests)
    // edge_fused_kernel for every layer or for none: the kNN kernel writes the neighbour lists in the form the consumer reads
    // (LDS addresses for the fused kernel, row indices otherwise), and GCNN's one list serves all four layers
    const bool fused = fused_path(w, N);
    const bool refine_off = knob == 20;   // float32 ranking only: the fp32-MFMA kernels
    for (int l = 0; l < 4; ++l) {
        const int co = w->pq[l].cout / 2;
        IQ_REQUIRE(w->pq[l].cin == cin, "iq_dgcnn: layer %d expects %d inputs, got %d", l, cin, w->pq[l].cin);
        if (l == 0 && walk) {
            iq::ProfileSpan span(iq::kSlotPrepool, st);
            hipLaunchKernelGGL(dg_walk_kernel, dim3((Np + 63) / 64, B), dim3(64), 0, st, walk->sorted, walk->src, walk->region_id,
                               walk->keep, walk->cloud_of, s.idx, rg, N, Np, walk->Nsl, walk->nclouds, fused ? 1 : 0);
            if ((rc = iq::check_launch("dg_walk_kernel"))) return rc;
        } else if (l == 0 || !fixed_graph) {
            iq::ProfileSpan span(iq::kSlotPrepool, st);
            // feature-space graphs of the fused path: distances on the bf16 matrix pipe, the operand image in the upper half of the
            // P/Q buffer (which the fused path leaves alone; the walk tables keep to the lower half).  5 = 22: the fp32 MFMA (A/B, tests)
            const bool bf3 = fused && (cin == 64 || cin == 128) && knob != 22 && !(refine_off);
            unsigned char* planes = bf3 ? reinterpret_cast<unsigned char*>(s.pq) + (size_t)rows * 1024 : nullptr;
            hipLaunchKernelGGL(rownorm_kernel, dim3((rows + 63) / 64), dim3(kThreads), 0, st, src, ld, creal, s.xx, rg, B, planes, rows / 32);
            if ((rc = launch_knn(src, ld, cin, s.xx, s.idx, s.near_tie, B, N, rows, rg, st, fused, planes, rows / 32))) return rc;
        }
        {
            iq::ProfileSpan span(iq::kSlotFstn, st);
            const bool fits = Np <= kGlMaxRows && co % 16 == 0;
            if (fused) {
                // the whole layer in one kernel: co / 16 workgroups per cloud
                const int wgs_per_cloud = co / 16;
                const dim3 grid((unsigned)((B + 7) / 8 * 8 * wgs_per_cloud));
                const iq_dense_layer& L = w->pq[l];
                if (cin == 8)
                    hipLaunchKernelGGL(edge_fused_kernel<1>, grid, dim3(kGlThreads), 0, st, src, ld, L.w, L.b, co, s.idx, s.xc + col, 512, rg, B, wgs_per_cloud, knob == 10);
                else if (cin == 64 && knob == 47)
                    hipLaunchKernelGGL((edge_fused_kernel<8, 10>), grid, dim3(kGlThreads), 0, st, src, ld, L.w, L.b, co, s.idx, s.xc + col, 512, rg, B, wgs_per_cloud, 0);
                else if (cin == 128 && knob == 47)
                    hipLaunchKernelGGL((edge_fused_kernel<16, 10>), grid, dim3(kGlThreads), 0, st, src, ld, L.w, L.b, co, s.idx, s.xc + col, 512, rg, B, wgs_per_cloud, 0);
                else if (cin == 64)
                    hipLaunchKernelGGL(edge_fused_kernel<8>, grid, dim3(kGlThreads), 0, st, src, ld, L.w, L.b, co, s.idx, s.xc + col, 512, rg, B, wgs_per_cloud, knob == 10);
                else
                    hipLaunchKernelGGL(edge_fused_kernel<16>, grid, dim3(kGlThreads), 0, st, src, ld, L.w, L.b, co, s.idx, s.xc + col, 512, rg, B, wgs_per_cloud, knob == 10);
                if (knob == 10) hipLaunchKernelGGL(edge_dbg_print_kernel, dim3(1), dim3(1), 0, st, l + 1);
                if ((rc = iq::check_launch("edge_fused_kernel"))) return rc;
                src = s.xc + col; ld = 512; cin = co; creal = co; col += co;
                continue;
            }
            if ((rc = iq::launch_linear(src, ld, w->pq[l], s.pq, 2 * co, rows, 0, st, live))) return rc;
            if (fits && knob != 7) {
                // P slices through LDS: co / 16 workgroups per cloud (the surplus ones of small clouds exit at once)
                const int wgs_per_cloud = co / 16;
                hipLaunchKernelGGL(gather_lds_kernel, dim3((unsigned)((B + 7) / 8 * 8 * wgs_per_cloud)), dim3(kGlThreads), 0, st,
                                   s.pq, co, s.idx, s.xc + col, 512, rg, B, wgs_per_cloud);
            } else {
                // clouds of more than 1024 rows: straight from L2
                // grid sized for the largest cloud (Np rows); workgroups past a cloud's live row count exit at once
                const int pts_per_wg = kThreads / (co / 4);
                const int wgs_per_cloud = (Np + pts_per_wg - 1) / pts_per_wg;
                hipLaunchKernelGGL(gather_max_kernel, dim3((unsigned)((B + 7) / 8 * 8 * wgs_per_cloud)), dim3(kThreads), 0, st,
                                   s.pq, co, s.idx, s.xc + col, 512, rg, B, wgs_per_cloud);
            }
            if ((rc = iq::check_launch("gather_max_kernel"))) return rc;
        }
        src = s.xc + col; ld = 512; cin = co; creal = co; col += co;
    }
    IQ_REQUIRE(col == 512, "iq_dgcnn: concatenated width %d != 512", col);
    {
        iq::ProfileSpan span(iq::kSlotTrunk, st);
        // conv5 + LeakyReLU with the pooling folded into the GEMM epilogue (s.h holds the per-tile partials)
        IQ_REQUIRE(w->conv5.cout == 1024 && w->conv5.cin == 512, "iq_dgcnn: conv5 must be 512 -> 1024");
        double work = 0.0;
        if (iq::profile_enabled()) {  // profiling only: the live row count (one sync); the GEMM issues whole 128-row tiles
            int32_t n = 0;
            if (hipMemcpyAsync(&n, live, sizeof(n), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess)
                work = 2.0 * 512.0 * 1024.0 * (double)((n + 127) / 128 * 128);
        }
        iq::ProfileSpan dom(iq::kSlotDominant, st, work);
        if ((rc = iq::launch_linear_pool(s.xc, 512, w->conv5, s.h, rows, 2, s.row_w, st, live, w->conv5_bf3))) return rc;
        hipLaunchKernelGGL(pool_reduce_kernel, dim3(1024 / kThreads, B), dim3(kThreads), 0, st, s.h, s.g, rg, N, 1024);
        if ((rc = iq::check_launch("pool_reduce_kernel"))) return rc;
    }
    if ((rc = iq::launch_linear(s.g, 2048, w->fc1, s.f1, 512, B, 2, st))) return rc;
    if ((rc = iq::launch_linear(s.f1, 512, w->fc2, s.f2, 256, B, 2, st))) return rc;
    if ((rc = iq::launch_linear(s.f2, 256, w->fc3, logits, w->fc3.cout, B, 0, st))) return rc;
    return IQ_OK;
}

}  // namespace

extern "C" int iq_dgcnn_forward(const iq_dgcnn_weights* w, const float* xyz, float* logits, void* workspace,
                                size_t workspace_bytes, int B, int N, int fixed_graph, iq_stream_t stream) {
    IQ_REQUIRE(w && xyz && logits, "iq_dgcnn_forward: null pointer");
    IQ_REQUIRE(B >= 0 && N >= kK && N <= 32767, "iq_dgcnn_forward: N=%d not in [%d, 32767]", N, kK);
    IQ_REQUIRE(w->k == kK, "iq_dgcnn_forward: k=%d (only 20)", w->k);
    if (B == 0) return IQ_OK;
    const size_t need = carve_d(nullptr, B, N).bytes;
    if (!workspace || workspace_bytes < need)
        return iq::fail(IQ_EWORKSPACE, "iq_dgcnn_forward: workspace %zu < %zu bytes", workspace_bytes, need);
    WsD s = carve_d(workspace, B, N);
    hipStream_t st = iq::as_stream(stream);
    const int Np = (N + 31) / 32 * 32;  // rows N..Np-1 of every cloud are dead padding
    const int rows = B * Np;
    int rc;
    iq::ProfileSpan call_span(iq::kSlotCall, st);
    hipLaunchKernelGGL(dg_dense_layout_kernel, dim3((rows + 256) / 256), dim3(256), 0, st, s.roff, s.nkept, s.ncopy,
                       s.row_cloud, s.row_w, B, N, Np);
    hipLaunchKernelGGL(pad_xyz_kernel, dim3((rows + 255) / 256), dim3(256), 0, st, xyz, s.x0, B, N, Np);
    if ((rc = iq::check_launch("pad_xyz_kernel"))) return rc;
    return run_network(w, s, logits, B, N, rows, fixed_graph, st);
}

extern "C" int iq_dgcnn_coalitions(const iq_dgcnn_weights* w, const float* clouds, const float* centers,
                                   const int32_t* region_id, const uint64_t* keep, const int32_t* cloud_of, float* logits,
                                   void* workspace, size_t workspace_bytes, int B, int nclouds, int N, int fixed_graph,
                                   iq_stream_t stream) {
    IQ_REQUIRE(B >= 0 && nclouds >= 1, "iq_dgcnn_coalitions: B=%d nclouds=%d", B, nclouds);
    IQ_REQUIRE(w && clouds && centers && region_id && (B == 0 || (keep && logits)), "iq_dgcnn_coalitions: null pointer");
    IQ_REQUIRE(N >= kK && N <= 32767, "iq_dgcnn_coalitions: N=%d not in [%d, 32767]", N, kK);
    IQ_REQUIRE(cloud_of || nclouds == 1 || nclouds == B, "iq_dgcnn_coalitions: cloud_of required when 1 < nclouds != B");
    IQ_REQUIRE(w->k == kK, "iq_dgcnn_coalitions: k=%d (only 20)", w->k);
    if (B == 0) return IQ_OK;
    const size_t need = carve_d(nullptr, B, N).bytes;
    if (!workspace || workspace_bytes < need)
        return iq::fail(IQ_EWORKSPACE, "iq_dgcnn_coalitions: workspace %zu < %zu bytes", workspace_bytes, need);
    WsD s = carve_d(workspace, B, N);
    hipStream_t st = iq::as_stream(stream);
    int rc;
    iq::ProfileSpan call_span(iq::kSlotCall, st);
    // Layer-1 graph from the source clouds' sorted neighbour lists when a few source clouds serve many coalitions.  The
    // tables are carved from the P/Q buffer, which the fused EdgeConv path does not touch (tuning key 5 = 12: knn_kernel<8>).
    const int Np = (N + 31) / 32 * 32, Nsp = (N + 1 + 31) / 32 * 32, Nsl = (N + 1 + 7) / 8 * 8;
    WalkCtx walk{};
    bool use_walk = false;
    {
        size_t off = 0;
        auto take = [&](size_t bytes) { size_t o = off; off = iq::align_up(off + bytes, 256); return reinterpret_cast<char*>(s.pq) + o; };
        float* xs = (float*)take((size_t)nclouds * Nsp * 8 * 4);
        float* xxs = (float*)take((size_t)nclouds * Nsp * 4);
        float* dmat = (float*)take((size_t)nclouds * Nsp * Nsp * 4);
        int16_t* sorted = (int16_t*)take((size_t)nclouds * (N + 1) * Nsl * 2);
        int16_t* srcrow = (int16_t*)take((size_t)B * Np * 2);
        use_walk = N <= kWalkMaxN && (long long)nclouds * 8 <= B && fused_path(w, N) && iq::tuning(iq::kTuneExperiment) != 12 &&
                   off <= (size_t)B * Np * 256 * 4;     // the lower half: the upper one holds the kNN operand image (run_network)
        if (use_walk) {
            hipLaunchKernelGGL(sl_rows_kernel, dim3((Nsp + 255) / 256, nclouds), dim3(256), 0, st, clouds, centers, xs, xxs, N, Nsp);
            hipLaunchKernelGGL(sl_dist_kernel<0>, dim3(Nsp / 32, nclouds), dim3(64), 0, st, xs, xxs, dmat, Nsp);
            hipLaunchKernelGGL(sl_sort_kernel, dim3(N + 1, nclouds), dim3(256), 0, st, dmat, sorted, N, Nsp, Nsl);
            if ((rc = iq::check_launch("sl_sort_kernel"))) return rc;
            walk = WalkCtx{sorted, srcrow, region_id, keep, cloud_of, nclouds, Nsl};
        }
    }
    hipLaunchKernelGGL(dg_count_kernel, dim3(B), dim3(64), 0, st, region_id, keep, cloud_of, s.nkept, s.ncopy, s.dpad, N, nclouds);
    hipLaunchKernelGGL(dg_scan_kernel, dim3(1), dim3(1024), 0, st, s.dpad, s.roff, B);
    hipLaunchKernelGGL(dg_compact_kernel, dim3(B), dim3(64), 0, st, clouds, centers, region_id, keep, cloud_of, s.roff, s.nkept,
                       s.ncopy, s.x0, s.row_cloud, s.row_w, N, nclouds, use_walk ? const_cast<int16_t*>(walk.src) : (int16_t*)nullptr, Np);
    if ((rc = iq::check_launch("dg_compact_kernel"))) return rc;
    return run_network(w, s, logits, B, N, B * Np, fixed_graph, st, use_walk ? &walk : nullptr);
}
