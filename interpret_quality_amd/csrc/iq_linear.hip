// The dense layer every model shares: out = act(A W^T + b) on weights pre-packed in MFMA B-fragment order
// (iq_pack_weight).  Two kernels with bit-identical results (same MFMA order over k), a split-K form for few rows x very
// long K, and a form whose epilogue does the first stage of a pooling layer.  fp32 MFMA (v_mfma_f32_32x32x2_f32) only.
#include <algorithm>
#include <cstring>

#include "iq_common.h"
#include "iq_mfma.h"
#include "iq_profile.h"

namespace {

constexpr int kThreads = 256;

// ---- batched dense layer  out = act(A W^T + b)  --------------------------------------------
// No LDS: the fp32 MFMA is slow enough (64 cycles) that both operands stream straight from
// L1/L2 into registers, one K-block ahead.  Wave tile (MT*32) x (NT*32); 4 waves as WM x WN.
template <int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(kThreads) void pn_linear_kernel(const float* __restrict__ A, int lda,
                                                             const float* __restrict__ wp,
                                                             const float* __restrict__ bias,
                                                             float* __restrict__ out, int ldo, int M, int K,
                                                             int Nout, int relu, const int32_t* __restrict__ m_dev,
                                                             int kb_per_split) {
    if (m_dev) M = min(M, *m_dev);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = (blockIdx.x * WM + wm) * MT * 32;
    const int nt0 = (blockIdx.y * WN + wn) * NT;
    const int KBT = K >> 3;                               // k-blocks of the whole layer (weight image stride)
    // split-K (kb_per_split > 0): workgroup z accumulates k-blocks [z kb_per_split, ...) and stores the raw partial
    // sums to out + z M ldo; bias and activation are applied by splitk_reduce_kernel
    const bool split = kb_per_split > 0;
    const int kb0 = split ? blockIdx.z * kb_per_split : 0;
    const int KB = split ? min(kb_per_split, KBT - kb0) : KBT;
    const int ntiles = (Nout + 31) >> 5;
    if (m0 >= M || nt0 >= ntiles) return;
    if (split) out += (size_t)blockIdx.z * M * ldo;

    const float* ap[MT];
    int bs[NT];                                   // scalar byte offset of n-tile j's fragment kb0 in the weight image
    const WBuf wb = wbuf_make(wp, lane);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int row = min(m0 + i * 32 + (lane & 31), M - 1);
        ap[i] = A + (size_t)row * lda + 4 * (lane >> 5) + 8 * kb0;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int nt = min(nt0 + j, ntiles - 1);
        bs[j] = uniform((nt * KBT + kb0) * kFragBytes);
    }
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x16){0};
    // the epilogue's bias values, requested now: loaded after the K loop they cost every wave a full memory round trip
    // (behind whatever operand loads are still in flight) with nothing left to hide it
    float breg[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) breg[j] = bias[min((nt0 + j) * 32 + (lane & 31), Nout - 1)];

    f32x4 av[MT], bv[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) av[i] = *reinterpret_cast<const f32x4*>(ap[i]);
#pragma unroll
    for (int j = 0; j < NT; ++j) bv[j] = wbuf_load(wb, bs[j]);
    for (int kb = 0; kb < KB; ++kb) {
        f32x4 an[MT], bn[NT];
        const int kn = min(kb + 1, KB - 1);
#pragma unroll
        for (int i = 0; i < MT; ++i) an[i] = *reinterpret_cast<const f32x4*>(ap[i] + 8 * kn);
#pragma unroll
        for (int j = 0; j < NT; ++j) bn[j] = wbuf_load(wb, bs[j] + kn * kFragBytes);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = mfma4(av[i], bv[j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < MT; ++i) av[i] = an[i];
#pragma unroll
        for (int j = 0; j < NT; ++j) bv[j] = bn[j];
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = (nt0 + j) * 32 + (lane & 31);
        if (nt0 + j >= ntiles || col >= Nout) continue;
        const float b = split ? 0.f : breg[j];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + i * 32 + c_row(r, lane);
                if (row < M) {
                    float v = acc[i][j][r] + b;
                    if (split) { out[(size_t)row * ldo + col] = acc[i][j][r]; continue; }
                    if (relu == 1) v = fmaxf(v, 0.f);
                    else if (relu == 2) v = v > 0.f ? v : 0.2f * v;  // LeakyReLU(0.2), models/dgcnn.py:66-80
                    out[(size_t)row * ldo + col] = v;
                }
            }
        }
    }
}

// Large-M variant: the A operand goes through LDS.  In pn_linear_kernel every A-fragment load touches 32 cache lines
// (32 rows x 16 B) and is repeated by the wave beside it, which keeps the L1 address path - not the MFMA - busy
// (measured 58 % of the fp32 MFMA peak on 512 -> 1024).  Here the workgroup copies a 128-row x 32-k chunk with
// full-line coalesced loads (8 lanes per 128 B row segment), double-buffered, and the four waves (2 x 2, wave tile
// 64 x NT*32) read their fragments from LDS (row stride 36 floats: conflict-free ds_read_b128); B fragments still
// stream from the packed image, prefetched two k-blocks ahead.  Same MFMA order over k as pn_linear_kernel, so the
// results are bit-identical.  Needs K % 32 == 0.
// POOL: instead of the activations, every 32-row tile writes its column-wise maximum and row-weighted sum over the
// rows with weight > 0 (out = (ceil(M/32), 2, Nout)): the input of a pooling layer without the round trip of the
// (M, Nout) activations through HBM.
// WN = waves across the columns (2: workgroup tile 128 rows x 2 NT 32 columns; 1: 256 rows x NT 32 columns - the shape for
// 128 outputs, where WN = 2 leaves a wave only NT = 2 tiles, i.e. half the MFMAs per LDS fragment and per barrier:
// PointConv's 2048 -> 128 layer ran at 0.64 of the peak on <2, 2> against 0.82-0.92 for the NT = 4 layers).
// NT = 5 exists for 320 outputs (PointNet++ sa2's per-point projections, 64 + 128 + 128): on NT = 4 the second column
// block held 2 real tiles of 8 and the launch issued 1.6 x the useful MFMAs.
template <int NT, bool POOL, int WN = 2>
__global__ __launch_bounds__(kThreads, 2) void pn_gemm_lds_kernel(const float* __restrict__ A, int lda,
                                                                  const float* __restrict__ wp,
                                                                  const float* __restrict__ bias, float* __restrict__ out,
                                                                  int ldo, int M, int K, int Nout, int relu,
                                                                  const int32_t* __restrict__ m_dev,
                                                                  const float* __restrict__ row_w,
                                                                  const int32_t* __restrict__ tile_nu, int rows_per_cloud,
                                                                  int col_blocks) {
    constexpr int KC = 32, LDA = KC + 4;
    constexpr int TM = (4 / WN) * 64;            // rows of the workgroup tile
    static_assert(WN == 1 || WN == 2, "WN");
    static_assert(!POOL || WN == 2, "the pooling epilogue assumes 128-row tiles");
    __shared__ __attribute__((aligned(16))) float As[2][TM * LDA];
    __shared__ float wrow[POOL ? 128 : 1];       // pooling weights of the tile's rows (read in the epilogue)
    // col_blocks > 0: a 1-D grid in which the column blocks of one row tile are neighbours ON THE SAME XCD (workgroups go
    // round-robin over the 8 XCDs): id = ((tile / 8) * col_blocks + column block) * 8 + tile % 8.  In a (tiles, column blocks)
    // grid the workgroups that read the same 128 rows of A sit a whole grid row apart, on whichever XCDs - each pulls the rows
    // from HBM into its own L2 (conv5: 12.7 GB of activations read four times per step).
    const int row_tiles = (M + TM - 1) / TM;
    int bx = blockIdx.x, by = blockIdx.y;
    if (col_blocks > 0) {
        const int per = 8 * col_blocks, grp = blockIdx.x / per, r = blockIdx.x - grp * per;
        bx = grp * 8 + (r & 7);
        by = r >> 3;
        if (bx >= row_tiles) return;
    }
    if (m_dev) M = min(M, *m_dev);
    int m0 = bx * TM;
    if (tile_nu) {   // rows beyond a cloud's live count are not wanted: skip tiles that hold nothing else
        // A cloud's live tiles are its FIRST ones and workgroups go round-robin over the 8 XCDs: in plain order (4 tiles per
        // cloud) every first tile lands on XCDs 0 and 4 and the XCDs that hold the last tiles idle (live workgroups 33 : 21 per
        // 33 coalitions of a Shapley batch).  Walk the grid cloud-fastest instead: every XCD sees every tile index equally often.
        const int ntc = rows_per_cloud / TM, nc = row_tiles / ntc;
        if (nc * ntc == row_tiles) m0 = ((bx % nc) * ntc + bx / nc) * TM;
        if (m0 >= M) return;
        const int c = m0 / rows_per_cloud;
        if (m0 - c * rows_per_cloud >= tile_nu[c]) return;
    }
    if (m0 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;
    if (POOL && tid < 128) wrow[tid] = m0 + tid < M ? row_w[m0 + tid] : 0.f;   // visible after the K loop's barriers
    const int KB = K >> 3, nchunks = K / KC;
    const int ntiles = (Nout + 31) >> 5;
    const int nt0 = (by * WN + wn) * NT;

    // chunk copy: TM rows x 8 float4; thread t owns (row, c4) = (e >> 3, e & 7) for e = t + 256 i.  Buffer loads on a
    // resource based at this workgroup's first row: loop-invariant per-lane offsets, the K position is a scalar offset.
    constexpr int NLD = TM * 8 / kThreads;       // float4 loads per thread and chunk
    const WBuf ab = wbuf_make(A + (size_t)m0 * lda, lane);
    int aoffb[NLD];
    int soff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + kThreads * i, row = e >> 3, c4 = e & 7;
        aoffb[i] = (min(row, M - 1 - m0) * lda + c4 * 4) * 4;
        soff[i] = row * LDA + c4 * 4;
    }
    f32x4 stage[NLD];
    auto load_chunk = [&](int kc) {
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            stage[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ab.rsrc, aoffb[i], kc * KC * 4, 0));
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) *reinterpret_cast<f32x4*>(&As[buf][soff[i]]) = stage[i];
    };

    const WBuf wb = wbuf_make(wp, lane);
    int bs[NT];                                   // scalar byte offset of n-tile j's first fragment
#pragma unroll
    for (int j = 0; j < NT; ++j) bs[j] = uniform(min(nt0 + j, ntiles - 1) * KB * kFragBytes);
    // B fragments in a 2-deep register ring (k-blocks kb, kb + 1); the slot just consumed is refilled with kb + 2, i.e.
    // 2 x 8 NT MFMAs (>= 4096 cycles) ahead of its use.  sched_group_barrier pins that order: left alone, the
    // scheduler sinks the loads next to their uses and the MFMAs wait on L2.
    f32x4 ring[2][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        ring[0][j] = wbuf_load(wb, bs[j]);
        ring[1][j] = wbuf_load(wb, bs[j] + min(1, KB - 1) * kFragBytes);
    }
    f32x16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x16){0};
    float breg[NT];                               // bias of this wave's columns, for the epilogue (see pn_linear_kernel)
#pragma unroll
    for (int j = 0; j < NT; ++j) breg[j] = bias[min((nt0 + j) * 32 + (lane & 31), Nout - 1)];

    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    const int aoff = (wm * 64 + (lane & 31)) * LDA + 4 * (lane >> 5);
    for (int kc = 0; kc < nchunks; ++kc) {
        const float* as = As[kc & 1] + aoff;
        f32x4 a0n = *reinterpret_cast<const f32x4*>(as), a1n = *reinterpret_cast<const f32x4*>(as + 32 * LDA);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const int kb = kc * 4 + k4;
            const f32x4 a0 = a0n, a1 = a1n;
            if (k4 < 3) {
                a0n = *reinterpret_cast<const f32x4*>(as + 8 * (k4 + 1));
                a1n = *reinterpret_cast<const f32x4*>(as + 32 * LDA + 8 * (k4 + 1));
            }
            const int kn = min(kb + 2, KB - 1) * kFragBytes;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const f32x4 bk = ring[k4 & 1][j];
                acc[0][j] = mfma4(a0, bk, acc[0][j]);
                acc[1][j] = mfma4(a1, bk, acc[1][j]);
                ring[k4 & 1][j] = wbuf_load(wb, bs[j] + kn);
            }
            if (k4 < 3) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // next chunk's A rows: issued behind the first k-block so that no MFMA of this chunk has to wait for them
            // (memory returns in order: a wait on an older B fragment never covers these)
            if (k4 == 0 && kc + 1 < nchunks) {
                load_chunk(kc + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (kc + 1 < nchunks) store_chunk((kc + 1) & 1);
        __syncthreads();
    }
    if (POOL) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int trow = m0 + wm * 64 + i * 32;  // first row of this 32-row tile
            if (trow >= M) continue;
            float w[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) w[r] = wrow[wm * 64 + i * 32 + c_row(r, lane)];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = (nt0 + j) * 32 + (lane & 31);
                const bool ok = nt0 + j < ntiles && col < Nout;
                const float b = ok ? breg[j] : 0.f;
                float mx = -INFINITY, sm = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r] + b;
                    if (relu == 1) v = fmaxf(v, 0.f);
                    else if (relu == 2) v = v > 0.f ? v : 0.2f * v;
                    if (w[r] > 0.f) {
                        mx = fmaxf(mx, v);
                        sm += w[r] * v;
                    }
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                sm += __shfl_xor(sm, 32);
                if (ok && lane < 32) {
                    float* o = out + (size_t)(trow >> 5) * 2 * Nout;
                    o[col] = mx;
                    o[Nout + col] = sm;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = (nt0 + j) * 32 + (lane & 31);
        if (nt0 + j >= ntiles || col >= Nout) continue;
        const float b = breg[j];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + c_row(r, lane);
                if (row < M) {
                    float v = acc[i][j][r] + b;
                    if (relu == 1) v = fmaxf(v, 0.f);
                    else if (relu == 2) v = v > 0.f ? v : 0.2f * v;
                    out[(size_t)row * ldo + col] = v;
                }
            }
        }
    }
}


// ---- the pooled dense layer on the bf16 matrix pipe, float32-exact ------------------------------------------------------------
// v_mfma_f32_32x32x2_f32 runs at the vector rate on gfx950 (157 TFLOP/s); v_mfma_f32_32x32x16_bf16 is sixteen times faster.  A
// float32 value is the sum of three bf16 terms (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): 24 mantissa bits, the residuals
// exact), a product of two bf16 values is exact in float32, and of the nine products of two split operands the three smallest
// (m.l, l.m, l.l) lie below 2^-24 of |a||b|: a.b = h.h + h.m + m.h + m.m + h.l + l.h to float32 accuracy, accumulated in float32
// inside the MFMA.  Measured on random data against float64: max error 2.0e-7 of the largest output, the fp32 GEMM 4.4e-7.
// 16 k cost 6 x 32 = 192 matrix cycles instead of 8 x 64 = 512.
//   * weights: split and packed once on the host (iq_pack_weight_bf3): [term][n-tile][k-step of 16][lane][8 bf16], 1 KB per
//     fragment, streamed through a ring of the four n-tiles' fragments one k-step (1 536 matrix cycles) ahead;
//   * activations: read ONCE as float32 (128 rows x 32 k per chunk, full-line coalesced), split in registers while they are
//     staged, three bf16 planes in LDS (row stride 80 bytes: conflict-free ds_read_b128), double-buffered;
//   * tile, epilogue and output exactly those of pn_gemm_lds_kernel<4, true>: 128 rows x 256 columns per workgroup, per 32-row
//     tile the column maxima and weighted sums.
// Not bit-identical to the fp32-MFMA kernel (another summation order), equal to it within float32 rounding.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// POOL = false: the plain dense layer out (M, ldo) = act(A W^T + b) on the same tiles (launch_linear, for layers that carry
// iq_dense_layer.w_bf3); tile_nu / rows_per_cloud as in pn_gemm_lds_kernel.
template <bool POOL, int PROBE = 0, int NW = 4, bool RAGGED = false, bool SPLITK = false>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void pn_gemm_bf3_kernel(const float* __restrict__ A, int lda,
                                                                  const unsigned short* __restrict__ w3,
                                                                  const float* __restrict__ bias, float* __restrict__ out, int ldo,
                                                                  int M, int K, int Nout, int relu,
                                                                  const int32_t* __restrict__ m_dev,
                                                                  const float* __restrict__ row_w, int col_blocks,
                                                                  const int32_t* __restrict__ tile_nu, int rows_per_cloud, int Kreal,
                                                                  int chunks_per_split) {
    // SPLITK (few rows, very long K: launch_linear_splitk): workgroup row blockIdx.y takes chunks_per_split 32-k chunks and writes
    // its RAW partial sums to out + blockIdx.y * M * ldo; bias and activation belong to splitk_reduce_kernel.
    // K = the layer's inputs rounded up to a multiple of 32 (the weight image is zero there, iq_pack_weight_bf3), Kreal = the
    // columns A really has (a multiple of 8).  RAGGED (Kreal < K; its own instantiation - the few registers it needs would spill
    // in the others): the last chunk's columns beyond Kreal are taken as zero.
    // wave tile: ALL 128 rows (MT = 4 m-tiles) x 64 columns (NT = 2): a weight fragment feeds four m-tiles - with 64 x 128 wave
    // tiles (two m-tiles per fragment) the weight stream alone asked the L2 for 19 TB/s at full matrix rate
    constexpr int MT = 4, NT = 2, KC = 32, ROWB = 80, PLANE = 128 * ROWB;      // bytes
    constexpr int NTH = 64 * NW, NLD = 1024 / NTH;    // threads; (row, 4 k) items per thread and chunk
    __shared__ __attribute__((aligned(16))) unsigned char As[2][3 * PLANE];
    __shared__ float wrow[POOL ? 128 : 1];
    const int row_tiles = (M + 127) / 128;
    const int per = 8 * col_blocks, grp = blockIdx.x / per, rr = blockIdx.x - grp * per;
    const int bx = grp * 8 + (rr & 7), by = rr >> 3;                   // column blocks of a row tile side by side on one XCD
    if (bx >= row_tiles) return;
    if (m_dev) M = min(M, *m_dev);
    int m0 = bx * 128;
    if (!POOL && tile_nu) {   // as pn_gemm_lds_kernel: cloud-fastest walk, tiles beyond a cloud's live rows skipped
        const int ntc = rows_per_cloud / 128, nc = row_tiles / ntc;
        if (nc * ntc == row_tiles) m0 = ((bx % nc) * ntc + bx / nc) * 128;
        if (m0 >= M) return;
        const int c = m0 / rows_per_cloud;
        if (m0 - c * rows_per_cloud >= tile_nu[c]) return;
    }
    if (m0 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (POOL && tid < 128) wrow[tid] = m0 + tid < M ? row_w[m0 + tid] : 0.f;
    const int KS = K >> 4, nchunks = K / KC;
    int kc0 = 0, kc1 = nchunks;
    if constexpr (SPLITK) {
        kc0 = blockIdx.y * chunks_per_split;
        kc1 = min(nchunks, kc0 + chunks_per_split);
        if (kc0 >= kc1) return;
        out += (size_t)blockIdx.y * M * ldo;
    }
    const int NTT = (Nout + 31) >> 5;
    const int nt0 = (by * NW + wave) * NT;

    // activations: thread t owns (row, 4 k) = (e >> 3, (e & 7) * 4) for e = t + 256 i
    // A's resource ends with the last row's last real column, so that the partial last chunk of a layer whose inputs are no
    // multiple of 32 reads zeros there, not memory behind the matrix (rows are clamped to M - 1 below)
    WBuf ab = wbuf_make(A + (size_t)m0 * lda, lane);
    if constexpr (RAGGED) {
        const long long left = ((long long)(M - 1 - m0) * lda + Kreal) * 4;
        ab.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + (size_t)m0 * lda), 0, (int)(left < 0x7fffffffLL ? left : 0x7fffffffLL), 0x00020000);
    }
    int aoffb[NLD], soff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + NTH * i, row = e >> 3, c4 = e & 7;
        aoffb[i] = (min(row, M - 1 - m0) * lda + c4 * 4) * 4;
        soff[i] = row * ROWB + c4 * 8;
    }
    f32x4 stage[NLD];
    auto load_chunk = [&](int kc) {
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            stage[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ab.rsrc, aoffb[i], kc * KC * 4, 0));
    };
    auto store_chunk = [&](int buf, int kc) {     // split into the three planes
        // the layer's partial last chunk (Kreal < K): a lane's four columns lie wholly inside or wholly outside the row - outside,
        // whatever the load brought (the row's padding, the next row, zeros past the matrix' end: see `ab`) is replaced by zeros
        const bool outside = RAGGED && kc == nchunks - 1 && kc * KC + (tid & 7) * 4 >= Kreal;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            if (RAGGED && outside) stage[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (PROBE == 1) {   // timing probe: the planes without the split's arithmetic (results WRONG)
                unsigned char* d = As[buf] + soff[i];
                const f32x2 lo = {stage[i][0], stage[i][1]}, hi = {stage[i][2], stage[i][3]};
                *reinterpret_cast<f32x2*>(d) = lo;
                *reinterpret_cast<f32x2*>(d + PLANE) = hi;
                *reinterpret_cast<f32x2*>(d + 2 * PLANE) = lo;
                continue;
            }
            bf16x4 h, m, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = stage[i][e];
                const __bf16 a = (__bf16)v;
                const float r1 = v - (float)a;
                const __bf16 b = (__bf16)r1;
                h[e] = a; m[e] = b; l[e] = (__bf16)(r1 - (float)b);
            }
            unsigned char* d = As[buf] + soff[i];
            *reinterpret_cast<bf16x4*>(d) = h;
            *reinterpret_cast<bf16x4*>(d + PLANE) = m;
            *reinterpret_cast<bf16x4*>(d + 2 * PLANE) = l;
        }
    };
    // weights: fragment (term e, n-tile nt, k-step ks) at ((e NTT + nt) KS + ks) KB
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(w3), 0, 0x7fffffff, 0x00020000);
    const int wvoff = lane * 16;
    int wsoff[NT];                         // scalar byte offset of n-tile j's term-0 fragment of k-step 0
#pragma unroll
    for (int j = 0; j < NT; ++j) wsoff[j] = uniform(min(nt0 + j, NTT - 1) * KS * 1024);
    const int term_stride = NTT * KS * 1024;
    struct B3 { bf16x8 h, m, l; };
    auto wfrag = [&](int j, int ks) {
        const int o = wsoff[j] + min(ks, KS - 1) * 1024;
        return B3{__builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, o, 0)),
                  __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, o + term_stride, 0)),
                  __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, o + 2 * term_stride, 0))};
    };
    B3 ring[2][NT];                        // two k-steps (3 072 matrix cycles) ahead
#pragma unroll
    for (int j = 0; j < NT; ++j) { ring[0][j] = wfrag(j, 2 * kc0); ring[1][j] = wfrag(j, 2 * kc0 + 1); }
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x16){0};
    float breg[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) breg[j] = bias[min((nt0 + j) * 32 + (lane & 31), Nout - 1)];

    load_chunk(kc0);
    store_chunk(kc0 & 1, kc0);
    __syncthreads();
    const int aoff = (lane & 31) * ROWB + (lane >> 5) * 16;          // bytes: row of m-tile 0, this lane's 8 k
    for (int kc = kc0; kc < kc1; ++kc) {
        const unsigned char* as = As[kc & 1] + aoff;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ks = kc * 2 + s;
            bf16x8 a[MT][3];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int e = 0; e < 3; ++e) a[i][e] = *reinterpret_cast<const bf16x8*>(as + e * PLANE + i * 32 * ROWB + s * 32);
            if (s == 0 && kc + 1 < kc1) load_chunk(kc + 1);         // the next chunk's rows, behind this chunk's first operands
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const B3 b = ring[s][j];                            // (k-step parity = s: two steps per chunk)
                ring[s][j] = wfrag(j, ks + 2);
                // four accumulation chains interleaved (a dependent MFMA waits for its predecessor); small terms first
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b.h, acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b.l, acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b.m, acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b.h, acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b.m, acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b.h, acc[i][j], 0, 0, 0);
            }
        }
        if (kc + 1 < kc1) store_chunk((kc + 1) & 1, kc + 1);
        __syncthreads();
    }
    if (!POOL) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if (m0 + i * 32 >= M) continue;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = (nt0 + j) * 32 + (lane & 31);
                if (nt0 + j >= NTT || col >= Nout) continue;
                const float b = breg[j];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + i * 32 + c_row(r, lane);
                    if (row < M) {
                        float v = acc[i][j][r];
                        if constexpr (!SPLITK) {
                            v += b;
                            if (relu == 1) v = fmaxf(v, 0.f);
                            else if (relu == 2) v = v > 0.f ? v : 0.2f * v;
                        }
                        out[(size_t)row * ldo + col] = v;
                    }
                }
            }
        }
        return;
    }
    // (Round 5, measured: the staging split costs 9 % of conv5 (PROBE 1) and this epilogue 8 % (PROBE 2).  NW = 8 - one workgroup
    // of eight waves per CU on 128 rows x 512 columns, the split shared by twice as many waves: conv5 29.5 ms against 29.1, no.  Taking the column maximum
    // over the raw sums - bias and activation are increasing maps - with an additive -inf mask for the dead rows, 8 instead of 10
    // VALU instructions per value: conv5 30.0 ms against 30.0, not adopted.)
    if constexpr (PROBE == 2) {   // timing probe: no pooling epilogue (results WRONG)
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) sacc += acc[i][j][0] + acc[i][j][5];
        if (sacc == 12345.678f) out[0] = sacc;
        return;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int trow = m0 + i * 32;
        if (trow >= M) continue;
        float w[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) w[r] = wrow[i * 32 + c_row(r, lane)];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = (nt0 + j) * 32 + (lane & 31);
            const bool ok = nt0 + j < NTT && col < Nout;
            const float b = ok ? breg[j] : 0.f;
            float mx = -INFINITY, sm = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r] + b;
                if (relu == 1) v = fmaxf(v, 0.f);
                else if (relu == 2) v = v > 0.f ? v : 0.2f * v;
                if (w[r] > 0.f) {
                    mx = fmaxf(mx, v);
                    sm += w[r] * v;
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            sm += __shfl_xor(sm, 32);
            if (ok && lane < 32) {
                float* o = out + (size_t)(trow >> 5) * 2 * Nout;
                o[col] = mx;
                o[Nout + col] = sm;
            }
        }
    }
}

}  // namespace

int iq::launch_linear(const float* A, int lda, const iq_dense_layer& L, float* out, int ldo, int M, int relu,
                      hipStream_t st, const int32_t* m_dev, const int32_t* tile_nu, int rows_per_cloud) {
    if (M == 0) return IQ_OK;
    IQ_REQUIRE(L.w && L.b && L.cin % 8 == 0 && L.cout >= 1, "dense layer: bad descriptor (cin=%d cout=%d)", L.cin, L.cout);
    const int ntiles = (L.cout + 31) / 32;
    if (tile_nu && (rows_per_cloud <= 0 || rows_per_cloud % 128 != 0)) tile_nu = nullptr;   // tiles must not straddle clouds
    // bf16x3 on the bf16 matrix pipe, float32-exact - for EVERY M, so that a row's result does not depend on the launch it is in
    // (5 = 57: fp32 MFMA, A/B and tests)
    // (column blocks of 256: 320 outputs would leave the second block a quarter full, and lose to the NT = 5 fp32 tiling - so the
    // whole blocks of such a layer go to the bf16 pipe and its last 64 columns to the fp32 MFMA as a layer of their own; which
    // columns take which arithmetic depends on the layer only, never on M)
    const int Kp = (L.cin + 31) & ~31;     // the bf16x3 image's k range (iq_pack_weight_bf3 pads with zero columns)
    if (L.w_bf3 && L.cout > 256 && L.cout % 256 == 64 && L.cin >= 32 && iq::tuning(iq::kTuneExperiment) != 57 &&
        iq::tuning(iq::kTuneExperiment) != 59) {       // 5 = 59: these layers alone on the fp32 MFMA (A/B)
        const int gx = (M + 127) / 128, gy = L.cout / 256;
        if (Kp != L.cin)
            hipLaunchKernelGGL((pn_gemm_bf3_kernel<false, 0, 4, true>), dim3((unsigned)((gx + 7) / 8 * 8 * gy)), dim3(kThreads), 0, st, A, lda,
                               reinterpret_cast<const unsigned short*>(L.w_bf3), L.b, out, ldo, M, Kp, L.cout, relu, m_dev, nullptr, gy,
                               tile_nu, rows_per_cloud, L.cin, 0);
        else
            hipLaunchKernelGGL(pn_gemm_bf3_kernel<false>, dim3((unsigned)((gx + 7) / 8 * 8 * gy)), dim3(kThreads), 0, st, A, lda,
                               reinterpret_cast<const unsigned short*>(L.w_bf3), L.b, out, ldo, M, Kp, L.cout, relu, m_dev, nullptr, gy,
                               tile_nu, rows_per_cloud, L.cin, 0);
        int rc = iq::check_launch("pn_gemm_bf3_kernel");
        if (rc) return rc;
        iq_dense_layer rest = L;                       // n-tiles 8 gy, 8 gy + 1 of the fp32 image (n-tile-major, cin / 8 fragments each)
        rest.w = L.w + (size_t)(8 * gy) * (L.cin / 8) * (kFragBytes / 4);
        rest.b = L.b + 256 * gy;
        rest.cout = 64;
        rest.w_bf3 = nullptr;
        return launch_linear(A, lda, rest, out + 256 * gy, ldo, M, relu, st, m_dev, tile_nu, rows_per_cloud);
    }
    if (L.w_bf3 && L.cout % 256 == 0 && L.cin >= 32 && iq::tuning(iq::kTuneExperiment) != 57 &&
        !(Kp != L.cin && iq::tuning(iq::kTuneExperiment) == 59)) {      // (59 also: the layers whose inputs are no multiple of 32)
        const int gx = (M + 127) / 128, gy = (L.cout + 255) / 256;
        if (Kp != L.cin)
            hipLaunchKernelGGL((pn_gemm_bf3_kernel<false, 0, 4, true>), dim3((unsigned)((gx + 7) / 8 * 8 * gy)), dim3(kThreads), 0, st, A, lda,
                               reinterpret_cast<const unsigned short*>(L.w_bf3), L.b, out, ldo, M, Kp, L.cout, relu, m_dev, nullptr, gy,
                               tile_nu, rows_per_cloud, L.cin, 0);
        else
            hipLaunchKernelGGL(pn_gemm_bf3_kernel<false>, dim3((unsigned)((gx + 7) / 8 * 8 * gy)), dim3(kThreads), 0, st, A, lda,
                               reinterpret_cast<const unsigned short*>(L.w_bf3), L.b, out, ldo, M, Kp, L.cout, relu, m_dev, nullptr, gy,
                               tile_nu, rows_per_cloud, L.cin, 0);
        return iq::check_launch("pn_gemm_bf3_kernel");
    }
    if (M >= 2048 && (ntiles >= 4 || (ntiles == 2 && (M + 255) / 256 >= 2048)) && L.cin % 32 == 0 && iq::tuning(iq::kTuneNoLdsGemm) == 0) {
        const int shape = iq::tuning(iq::kTuneExperiment);   // 5 = 30: round 3's choice of shapes (A/B runs)
        if (ntiles % 10 == 0 && shape != 30 && (long long)((M + 127) / 128) * (ntiles / 10) >= 2048) {
            // 320 / 640 ... outputs: column blocks of exactly 10 tiles (NT = 5), nothing padded
            dim3 grid((M + 127) / 128, ntiles / 10);
            hipLaunchKernelGGL((pn_gemm_lds_kernel<5, false>), grid, dim3(kThreads), 0, st, A, lda, L.w, L.b, out, ldo, M, L.cin,
                               L.cout, relu, m_dev, nullptr, tile_nu, rows_per_cloud, 0);
        } else if (ntiles == 2 && (M + 255) / 256 >= 2048) {
            // 64 outputs (the fp32 rest of a 320-output layer): 256-row tiles, every wave both column tiles
            dim3 grid((M + 255) / 256, 1);
            hipLaunchKernelGGL((pn_gemm_lds_kernel<2, false, 1>), grid, dim3(kThreads), 0, st, A, lda, L.w, L.b, out, ldo, M, L.cin,
                               L.cout, relu, m_dev, nullptr, rows_per_cloud % 256 == 0 ? tile_nu : nullptr, rows_per_cloud, 0);
        } else if (ntiles == 4 && shape != 30 && (M + 255) / 256 >= 2048) {
            // 128 outputs: 256-row workgroup tiles, every wave all four column tiles
            dim3 grid((M + 255) / 256, 1);
            hipLaunchKernelGGL((pn_gemm_lds_kernel<4, false, 1>), grid, dim3(kThreads), 0, st, A, lda, L.w, L.b, out, ldo, M, L.cin,
                               L.cout, relu, m_dev, nullptr, rows_per_cloud % 256 == 0 ? tile_nu : nullptr, rows_per_cloud, 0);
        } else if (ntiles >= 8 && (long long)((M + 127) / 128) * ((ntiles + 7) / 8) >= 2048) {
            const int gx = (M + 127) / 128, gy = (ntiles + 7) / 8;
            const bool flat = gy > 1 && shape != 48;       // column blocks of a row tile side by side on one XCD (see the kernel)
            hipLaunchKernelGGL((pn_gemm_lds_kernel<4, false>), flat ? dim3((unsigned)((gx + 7) / 8 * 8 * gy)) : dim3(gx, gy), dim3(kThreads), 0,
                               st, A, lda, L.w, L.b, out, ldo, M, L.cin, L.cout, relu, m_dev, nullptr, tile_nu, rows_per_cloud, flat ? gy : 0);
        } else {
            dim3 grid((M + 127) / 128, (ntiles + 3) / 4);
            hipLaunchKernelGGL((pn_gemm_lds_kernel<2, false>), grid, dim3(kThreads), 0, st, A, lda, L.w, L.b, out, ldo, M, L.cin,
                               L.cout, relu, m_dev, nullptr, tile_nu, rows_per_cloud, 0);
        }
        return iq::check_launch("pn_gemm_lds_kernel");
    }
    if (ntiles >= 4) {
        dim3 grid((M + 127) / 128, (ntiles + 3) / 4);
        hipLaunchKernelGGL((pn_linear_kernel<2, 2, 2, 2>), grid, dim3(kThreads), 0, st, A, lda, L.w, L.b, out, ldo, M,
                           L.cin, L.cout, relu, m_dev, 0);
    } else {
        dim3 grid((M + 255) / 256, ntiles);
        hipLaunchKernelGGL((pn_linear_kernel<2, 1, 4, 1>), grid, dim3(kThreads), 0, st, A, lda, L.w, L.b, out, ldo, M,
                           L.cin, L.cout, relu, m_dev, 0);
    }
    return iq::check_launch("pn_linear_kernel");
}

namespace {
// out[row][col] = act(bias[col] + sum_z partial[z][row][col]), z ascending: fixed summation order
__global__ void splitk_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bias, float* __restrict__ out,
                                     int ldo, int M, int N, int splits, int relu) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * N) return;
    const int row = t / N, col = t - row * N;
    float v = 0.f;
    for (int z = 0; z < splits; ++z) v += partial[((size_t)z * M + row) * N + col];
    v += bias[col];
    if (relu == 1) v = fmaxf(v, 0.f);
    else if (relu == 2) v = v > 0.f ? v : 0.2f * v;
    out[(size_t)row * ldo + col] = v;
}
}  // namespace

// Few rows, very long K (PointConv's 16384 -> 1024 layer on B rows): a plain launch has ceil(M/128) x cout/128 workgroups -
// 48 for 660 rows - each walking all of K.  Split K over blockIdx.z until the grid fills the chip; partial sums go
// through `scratch` (splits x M x cout floats) and are added in a fixed order.
int iq::launch_linear_splitk(const float* A, int lda, const iq_dense_layer& L, float* out, int ldo, int M, int relu,
                             float* scratch, size_t scratch_floats, hipStream_t st) {
    if (M == 0) return IQ_OK;
    IQ_REQUIRE(L.w && L.b && L.cin % 8 == 0 && L.cout >= 1, "dense layer: bad descriptor (cin=%d cout=%d)", L.cin, L.cout);
    const int ntiles = (L.cout + 31) / 32, KB = L.cin / 8;
    // 512 k per split, whatever M is: the summation order must not depend on how many rows share the launch, or a
    // coalition's logits would change with the batch it travels in (found by the two-rank artefact comparison)
    const int kbs = 64;
    const int splits = (KB + kbs - 1) / kbs;
    if (splits <= 1 || ntiles < 4) return launch_linear(A, lda, L, out, ldo, M, relu, st);
    if (L.w_bf3 && L.cout % 256 == 0 && L.cin % 32 == 0 && iq::tuning(iq::kTuneExperiment) != 57 && iq::tuning(iq::kTuneExperiment) != 59) {
        // the same 512-k splits on the bf16 matrix pipe (three-term products): 16 chunks of 32 k per workgroup row
        IQ_REQUIRE(scratch && (size_t)splits * M * L.cout <= scratch_floats, "split-K dense layer: scratch %zu floats < %zu",
                   scratch_floats, (size_t)splits * M * L.cout);
        const int gx = (M + 127) / 128, gy = L.cout / 256;
        hipLaunchKernelGGL((pn_gemm_bf3_kernel<false, 0, 4, false, true>), dim3((unsigned)((gx + 7) / 8 * 8 * gy), (unsigned)splits), dim3(kThreads), 0,
                           st, A, lda, reinterpret_cast<const unsigned short*>(L.w_bf3), L.b, scratch, L.cout, M, L.cin, L.cout, 0, nullptr, nullptr,
                           gy, nullptr, 0, L.cin, kbs * 8 / 32);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(((size_t)M * L.cout + 255) / 256)), dim3(256), 0, st, scratch, L.b,
                           out, ldo, M, L.cout, splits, relu);
        return iq::check_launch("pn_gemm_bf3_kernel<split-K>");
    }
    IQ_REQUIRE(scratch && (size_t)splits * M * L.cout <= scratch_floats, "split-K dense layer: scratch %zu floats < %zu",
               scratch_floats, (size_t)splits * M * L.cout);
    dim3 grid((M + 127) / 128, (ntiles + 3) / 4, splits);
    hipLaunchKernelGGL((pn_linear_kernel<2, 2, 2, 2>), grid, dim3(kThreads), 0, st, A, lda, L.w, L.b, scratch, L.cout, M, L.cin,
                       L.cout, 0, nullptr, kbs);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(((size_t)M * L.cout + 255) / 256)), dim3(256), 0, st, scratch, L.b,
                       out, ldo, M, L.cout, splits, relu);
    return iq::check_launch("pn_linear_kernel<split-K>");
}

int iq::launch_linear_pool(const float* A, int lda, const iq_dense_layer& L, float* partial, int M, int relu,
                           const float* row_w, hipStream_t st, const int32_t* m_dev, const void* w_bf3) {
    if (M == 0) return IQ_OK;
    IQ_REQUIRE(L.w && L.b && row_w && partial, "dense layer + pool: null pointer");
    const int ntiles = (L.cout + 31) / 32;
    if (L.cin % 32 != 0 || ntiles < 8 || L.cout % 32 != 0)
        return iq::fail(IQ_EUNSUPPORTED, "dense layer + pool: cin=%d cout=%d", L.cin, L.cout);
    const int gy = (ntiles + 7) / 8, gx = (M + 127) / 128;
    if (w_bf3 && L.cout % 256 == 0 && iq::tuning(iq::kTuneExperiment) != 53) {   // 5 = 53: the fp32 MFMA (A/B and tests)
        const int probe = iq::tuning(iq::kTuneExperiment);     // 94 / 95: timing probes (no split arithmetic / no pooling; results WRONG)
        if (probe == 94 || probe == 95) {
            if (probe == 94)
                hipLaunchKernelGGL((pn_gemm_bf3_kernel<true, 1>), dim3((unsigned)((gx + 7) / 8 * 8 * gy)), dim3(kThreads), 0, st, A, lda,
                                   reinterpret_cast<const unsigned short*>(w_bf3), L.b, partial, 0, M, L.cin, L.cout, relu, m_dev, row_w, gy, nullptr, 0, L.cin, 0);
            else
                hipLaunchKernelGGL((pn_gemm_bf3_kernel<true, 2>), dim3((unsigned)((gx + 7) / 8 * 8 * gy)), dim3(kThreads), 0, st, A, lda,
                                   reinterpret_cast<const unsigned short*>(w_bf3), L.b, partial, 0, M, L.cin, L.cout, relu, m_dev, row_w, gy, nullptr, 0, L.cin, 0);
            return iq::check_launch("pn_gemm_bf3_kernel<pool, probe>");
        }
        hipLaunchKernelGGL(pn_gemm_bf3_kernel<true>, dim3((unsigned)((gx + 7) / 8 * 8 * gy)), dim3(kThreads), 0, st, A, lda,
                           reinterpret_cast<const unsigned short*>(w_bf3), L.b, partial, 0, M, L.cin, L.cout, relu, m_dev, row_w, gy,
                           nullptr, 0, L.cin, 0);
        return iq::check_launch("pn_gemm_bf3_kernel<pool>");
    }
    if (iq::tuning(iq::kTuneExperiment) == 48) {   // 5 = 48: the (tiles, column blocks) grid of rounds 1-3 (A/B)
        hipLaunchKernelGGL((pn_gemm_lds_kernel<4, true>), dim3(gx, gy), dim3(kThreads), 0, st, A, lda, L.w, L.b, partial, 0, M, L.cin, L.cout,
                           relu, m_dev, row_w, nullptr, 0, 0);
    } else {
        hipLaunchKernelGGL((pn_gemm_lds_kernel<4, true>), dim3((unsigned)((gx + 7) / 8 * 8 * gy)), dim3(kThreads), 0, st, A, lda, L.w, L.b,
                           partial, 0, M, L.cin, L.cout, relu, m_dev, row_w, nullptr, 0, gy);
    }
    return iq::check_launch("pn_gemm_lds_kernel<pool>");
}

extern "C" int iq_linear(const float* A, int lda, const iq_dense_layer* L, float* out, int ldo, int M, int act,
                         iq_stream_t stream) {
    IQ_REQUIRE(A && L && out, "iq_linear: null pointer");
    IQ_REQUIRE(M >= 0 && act >= 0 && act <= 2 && lda >= L->cin && ldo >= L->cout, "iq_linear: M=%d act=%d lda=%d ldo=%d", M,
               act, lda, ldo);
    return iq::launch_linear(A, lda, *L, out, ldo, M, act, iq::as_stream(stream));
}

// ---- host-side weight packing ---------------------------------------------------------------
extern "C" int iq_padded_cout(int cout) { return (cout + 31) / 32 * 32; }

extern "C" size_t iq_packed_floats(int cout, int cin) { return (size_t)iq_padded_cout(cout) * cin; }

// bf16 (round to nearest even) of a finite float, as v_cvt_pk_bf16_f32 rounds
static inline unsigned short iq_bf16_of(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static inline float iq_float_of_bf16(unsigned short h) {
    const unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

extern "C" size_t iq_packed_bf3_elems(int cout, int cin) { return (size_t)3 * iq_padded_cout(cout) * ((cin + 31) & ~31); }

// [term][n-tile][k-step of 16][lane][8]: lane (n & 31) + 32 ((k >> 3) & 1) holds k-aligned-8 elements of column n
extern "C" int iq_pack_weight_bf3(const float* w, unsigned short* out, int cout, int cin) {
    IQ_REQUIRE(w && out && cout >= 1 && cin >= 8 && cin % 8 == 0, "iq_pack_weight_bf3: cout=%d cin=%d", cout, cin);
    const int KS = ((cin + 31) & ~31) / 16, ntiles = iq_padded_cout(cout) / 32;   // k padded to a multiple of 32 with zero columns
    const size_t term = (size_t)ntiles * KS * 512;
    for (int nt = 0; nt < ntiles; ++nt)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int n = nt * 32 + (lane & 31), k = 16 * ks + 8 * (lane >> 5) + j;
                    const float v = n < cout && k < cin ? w[(size_t)n * cin + k] : 0.f;
                    const unsigned short h = iq_bf16_of(v);
                    const float r1 = v - iq_float_of_bf16(h);
                    const unsigned short m = iq_bf16_of(r1);
                    const unsigned short l = iq_bf16_of(r1 - iq_float_of_bf16(m));
                    const size_t o = (((size_t)nt * KS + ks) * 64 + lane) * 8 + j;
                    out[o] = h; out[term + o] = m; out[2 * term + o] = l;
                }
    return IQ_OK;
}

extern "C" int iq_pack_weight(const float* w, float* out, int cout, int cin) {
    IQ_REQUIRE(w && out && cout >= 1 && cin >= 8 && cin % 8 == 0, "iq_pack_weight: cout=%d cin=%d", cout, cin);
    const int KB = cin / 8, ntiles = iq_padded_cout(cout) / 32;
    for (int nt = 0; nt < ntiles; ++nt)
        for (int kb = 0; kb < KB; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    const int n = nt * 32 + (lane & 31), k = 8 * kb + 4 * (lane >> 5) + j;
                    out[(((size_t)nt * KB + kb) * 64 + lane) * 4 + j] = n < cout ? w[(size_t)n * cin + k] : 0.f;
                }
    return IQ_OK;
}

