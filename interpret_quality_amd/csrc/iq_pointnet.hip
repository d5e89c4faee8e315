// Fused fp32-MFMA PointNet forward over coalitions (models/pointnet.py:11-115).
//
// A masked cloud is {kept points} + {the centre, if anything was masked}.  Every per-point layer
// is a pointwise function and every pooling is a max, so
//   * the input-STN's 3->64->128->1024 chain is evaluated ONCE per cloud and pre-pooled per region
//     (chain<kPrepool>); a coalition's pooled feature is a max over its kept regions (+ centre);
//   * the feature-STN and trunk chains run only over a coalition's DISTINCT points.
// Both are exact with respect to evaluating the same kernels on the materialised cloud.
//
// chain kernel: one workgroup (4 waves) per coalition, 64-row chunks:
//   stage0 (VALU)  x -> x.trans -> conv1(3->64)+bn+relu                      -> LDS act0
//   L1  (MFMA)     64->64   (fstn.conv1 | per-coalition trans_feat product)  -> LDS act1
//   L2  (MFMA)     64->128  conv2+bn+relu                                    -> LDS act2
//   L3  (MFMA)     128->1024 conv3+bn(+relu), column max over rows           -> registers
// v_mfma_f32_32x32x2_f32 throughout (exact fp32 fma chains).  A operands come from LDS
// (row stride padded by 4 floats: conflict-free ds_read_b128), B operands (weights) stream from L2 in a
// pre-packed fragment order (1 KiB contiguous per wave-load).  LDS = 52 KB -> 3 workgroups/CU.
#include <algorithm>

#include "iq_common.h"
#include "iq_profile.h"

#include "iq_bf3.h"
#include "iq_mfma.h"

namespace {

constexpr int kFeat = IQ_NUM_FEAT;
constexpr int kMC = 64;        // rows per chunk
constexpr int kMaxN = 4096;    // points per cloud supported by the chain kernel (= IQ_MAX_POINTS; row lists are uint16)
constexpr int kThreads = 256;
constexpr int kRowCap = kMaxN + kMC;  // row-list stride per item (uint16)

enum ChainMode { kPrepool = 0, kFstn = 1, kTrunk = 2 };

// LDS activation images are row-major [row][k] with the row stride padded by 4 floats (68 / 132):
// a 16-lane ds_read_b128 group reads 16 different rows at one k, i.e. bank offsets 4*row mod 64 ->
// conflict-free, and every address in the MFMA loops is ONE per-lane base plus an immediate.
constexpr int kLd1 = 64 + 4;    // act0 / act1 (64 channels)
constexpr int kLd2 = 128 + 4;   // act2 (128 channels)

struct ChainArgs {
    const float* clouds;       // strides below, in floats
    int ps, cs, cl;            // point, channel, cloud stride
    const float* centers;      // (nclouds,3)
    const uint16_t* rows;        // (items,kRowCap) compacted point list of every item (pn_rows_kernel)
    const int32_t* nrows;        // (items) row count | centre flag << 16
    const int32_t* item_order;   // (items) launch order (largest coalitions first) or null
    const int32_t* cloud_of;   // (items) or null                          [kFstn/kTrunk]
    const float* trans;        // (items,9) input transform                [kFstn/kTrunk]
    const float* w_in;         // [64][4] folded 3->64 layer
    const float* w1;           // kFstn: packed 64x64; kTrunk: per item packed image (items,4096)
    const float* b1;
    const float* w2;
    const float* b2;
    const float* w3;
    const float* b3;
    const unsigned short* w3_bf3;   // layer 3's weights as three bf16 terms (iq_pack_weight_bf3) or null: L3 on the bf16 matrix pipe (L3V = 3)
    const unsigned short* w2_bf3;   // layer 2's, likewise (L3V = 3 needs both)
    float* out;                // (items,1024)
    int32_t* argrow;           // (items,1024) point index of the row that attains each column maximum, or null [ARGMAX trunk only]
    int N, R, items, nclouds, with_centre;
    int tail16;                  // last m-tile of an item on 16x16x4 MFMAs when it holds at most 16 rows (l3_tail16)
    int l3_single;               // bf16x3 layer 3 one n-tile per pass (tuning key 5 = 58: A/B against two per pass)
    unsigned long long* stamps;  // diagnostic build only
};

// ---- L3: 128 -> 1024 for one 64-row chunk; wave `wave` owns the n-tiles q*4 + wave ------------
// Variant 0: B fragments two K-blocks ahead in a small register ring, compiler-scheduled.
template <int MTS>
__device__ __forceinline__ void l3_pass_v0(const WBuf& w3, const float* abase, int wave_s, float (&runmax)[8]) {
    int wq = wave_s * 16 * kFragBytes;
    f32x4 b0 = wbuf_load(w3, wq);
    f32x4 b1 = wbuf_load(w3, wq + kFragBytes);
#pragma unroll 1
    for (int q = 0; q < 8; ++q) {
        const int wnext = (min(q + 1, 7) * 4 + wave_s) * 16 * kFragBytes;
        f32x16 acc0 = {0}, acc1 = {0};
#pragma unroll 4
        for (int kb = 0; kb < 16; ++kb) {
            const f32x4 bn = wbuf_load(w3, (kb + 2 < 16) ? wq + (kb + 2) * kFragBytes : wnext + (kb + 2 - 16) * kFragBytes);
            acc0 = mfma4(lds_frag<kLd2>(abase, 0, kb), b0, acc0);
            if (MTS == 2) acc1 = mfma4(lds_frag<kLd2>(abase, 1, kb), b0, acc1);
            b0 = b1;
            b1 = bn;
        }
        float m = max16(acc0);
        if (MTS == 2) m = fmaxf(m, max16(acc1));
#pragma unroll
        for (int i = 0; i < 8; ++i) runmax[i] = (i == q) ? fmaxf(runmax[i], m) : runmax[i];
        wq = wnext;
    }
}

// Variant 2: an 8-deep register ring of B fragments (32 VGPRs) that runs 8 K-blocks (4 096 MFMA
// cycles) ahead and never drains: it rolls over n-tile boundaries and, through `ring`, over chunk
// boundaries (the last n-tile of a chunk prefetches the first K-blocks of n-tile 0).  A fragments are
// double-buffered one K-block ahead.  sched_group_barrier pins the issue order inside each K-block:
// LDS reads of the next block, the MFMAs of this one, one ring refill.
struct BRing {
    f32x4 r[8];
};

__device__ __forceinline__ void bring_init(BRing& ring, const WBuf& w3, int wave_s) {
#pragma unroll
    for (int i = 0; i < 8; ++i) ring.r[i] = wbuf_load(w3, (wave_s * 16 + i) * kFragBytes);
}

// wave_s: the wave index as a scalar; the fragment offsets are then scalar too (iq_mfma.h: WBuf)
// ARGMAX (dense forward with crt_points only): besides the column maximum, the position in the item's row list of the row
// that attains it (largest value, then lowest row).  `rowbase` = first row of this chunk, fh = lane >> 5.
template <int MTS, bool ARGMAX = false>
__device__ __forceinline__ void l3_pass_v2(const WBuf& w3, const float* abase, int wave_s, float (&runmax)[8], BRing& ring,
                                           int (&runarg)[8], int rowbase = 0, int fh = 0) {
    const int w0 = wave_s * 16 * kFragBytes;
    int wq = w0;
#pragma unroll 1
    for (int q = 0; q < 8; ++q) {
        const int wn = (q < 7) ? wq + 4 * 16 * kFragBytes : w0;
        f32x16 acc0 = {0}, acc1 = {0};
        f32x4 a0n = lds_frag<kLd2>(abase, 0, 0), a1n = a0n;
        if (MTS == 2) a1n = lds_frag<kLd2>(abase, 1, 0);
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            const f32x4 a0 = a0n, a1 = a1n;
            if (kb + 1 < 16) {
                a0n = lds_frag<kLd2>(abase, 0, kb + 1);
                if (MTS == 2) a1n = lds_frag<kLd2>(abase, 1, kb + 1);
            }
            const f32x4 bk = ring.r[kb & 7];
            acc0 = mfma4(a0, bk, acc0);
            if (MTS == 2) acc1 = mfma4(a1, bk, acc1);
            ring.r[kb & 7] = wbuf_load(w3, kb < 8 ? wq + (kb + 8) * kFragBytes : wn + (kb - 8) * kFragBytes);
            if (kb + 1 < 16) __builtin_amdgcn_sched_group_barrier(0x100, MTS, 0);  // DS reads of K-block kb+1
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * MTS, 0);               // MFMAs of K-block kb
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                     // ring refill
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ARGMAX) {
            float m = -INFINITY;
            int r = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {          // rows ascend with i inside a tile: strict > keeps the lowest row
                if (acc0[i] > m) { m = acc0[i]; r = rowbase + c_row_i(i) + 4 * fh; }
            }
            if (MTS == 2) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (acc1[i] > m) { m = acc1[i]; r = rowbase + 32 + c_row_i(i) + 4 * fh; }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {           // chunks come in row order: strict > again
                const bool better = (i == q) && m > runmax[i];
                runarg[i] = better ? r : runarg[i];
                runmax[i] = better ? m : runmax[i];
            }
        } else {
            float m = max16(acc0);
            if (MTS == 2) m = fmaxf(m, max16(acc1));
#pragma unroll
            for (int i = 0; i < 8; ++i) runmax[i] = (i == q) ? fmaxf(runmax[i], m) : runmax[i];
        }
        wq = wn;
    }
}

// ---- Variant 3: L3 on the bf16 matrix pipe, float32-exact -----------------------------------------------------------------------
// (csrc/iq_linear.hip, pn_gemm_bf3_kernel<pool>, has the arithmetic: a float32 is three bf16 terms, a product the six largest of the
// nine term products, each exact, accumulated in float32; 16 k cost 192 matrix cycles instead of 512.)  act2 lives in LDS as three
// bf16 planes [64][136] (row stride 272 bytes = 68 dwords: the same conflict-free ds_read_b128 pattern as the float image), written
// split by layer 2's epilogue; the weights come split and packed from the host (iq_pack_weight_bf3: fragment (term, n-tile, k-step)
// at ((term 32 + n-tile) 8 + k-step) KB), through a ring four k-steps (1 536 matrix cycles) ahead that rolls over n-tile and chunk
// boundaries like BRing.  A row's result does not depend on the tile or chunk it sits in, so the pooled maxima are those of the
// same rows in any arrangement (fused = materialised, bitwise, as before).  16-row tail tiles are not used here (a 32-row tile).
constexpr int kLdB = 272;                 // bytes per act2 row of one bf16 plane
constexpr int kPlaneB = kMC * kLdB;       // bytes per plane
struct B3Ring { B3 r[4]; };
constexpr int kLd1B = 144;                // bytes per act1 row of one bf16 plane (64 k; 36 dwords: conflict-free ds_read_b128)
constexpr int kPlane1B = kMC * kLd1B;

// layer 2's weight fragment (n-tile nt of 4, k-step ks of 4) as three terms: iq_pack_weight_bf3 of a (128,64) matrix
__device__ __forceinline__ B3 b3_load_l2(const __amdgpu_buffer_rsrc_t& rs, int voff, int nt, int ks) {
    const int o = (nt * 4 + ks) * 1024;
    return B3{__builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, o, 0)),
              __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, o + 16 * 1024, 0)),
              __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, o + 2 * 16 * 1024, 0))};
}

__device__ __forceinline__ B3 b3_load(const __amdgpu_buffer_rsrc_t& rs, int voff, int step, int wave_s) {
    // step = q * 8 + ks (mod 64): n-tile q * 4 + wave, k-step ks
    const int o = ((((step >> 3) & 7) * 4 + wave_s) * 8 + (step & 7)) * 1024;
    return B3{__builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, o, 0)),
              __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, o + 32 * 8 * 1024, 0)),
              __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, o + 2 * 32 * 8 * 1024, 0))};
}

template <int MTS, bool ARGMAX = false>
__device__ __forceinline__ void l3_pass_bf3(const __amdgpu_buffer_rsrc_t& rs, int voff, const unsigned char* abase, int wave_s,
                                            float (&runmax)[8], B3Ring& ring, int (&runarg)[8], int rowbase = 0, int fh = 0) {
#pragma unroll 1
    for (int q = 0; q < 8; ++q) {
        f32x16 acc0 = {0}, acc1 = {0};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            bf16x8 a0[3], a1[3];
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                a0[e] = *reinterpret_cast<const bf16x8*>(abase + e * kPlaneB + s * 32);
                if (MTS == 2) a1[e] = *reinterpret_cast<const bf16x8*>(abase + e * kPlaneB + 32 * kLdB + s * 32);
            }
            const B3 b = ring.r[s & 3];
            ring.r[s & 3] = b3_load(rs, voff, q * 8 + s + 4, wave_s);
            // small terms first; with two m-tiles the two accumulation chains alternate
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[2], b.h, acc0, 0, 0, 0);
            if (MTS == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[2], b.h, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], b.l, acc0, 0, 0, 0);
            if (MTS == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], b.l, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[1], b.m, acc0, 0, 0, 0);
            if (MTS == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[1], b.m, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[1], b.h, acc0, 0, 0, 0);
            if (MTS == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[1], b.h, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], b.m, acc0, 0, 0, 0);
            if (MTS == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], b.m, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], b.h, acc0, 0, 0, 0);
            if (MTS == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], b.h, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);     // one k-step at a time: left alone the scheduler hoists every step's operand
                                                   // reads to the top of the n-tile (256 registers, one wave per SIMD)
        }
        if (ARGMAX) {   // as l3_pass_v2
            float m = -INFINITY;
            int r = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (acc0[i] > m) { m = acc0[i]; r = rowbase + c_row_i(i) + 4 * fh; }
            }
            if (MTS == 2) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (acc1[i] > m) { m = acc1[i]; r = rowbase + 32 + c_row_i(i) + 4 * fh; }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool better = (i == q) && m > runmax[i];
                runarg[i] = better ? r : runarg[i];
                runmax[i] = better ? m : runmax[i];
            }
        } else {
            float m = max16(acc0);
            if (MTS == 2) m = fmaxf(m, max16(acc1));
#pragma unroll
            for (int i = 0; i < 8; ++i) runmax[i] = (i == q) ? fmaxf(runmax[i], m) : runmax[i];
        }
    }
}

// The same layer with TWO n-tiles per pass (n-tiles (2 qp) 4 + wave and (2 qp + 1) 4 + wave): the A terms of a k-step are read from
// LDS once for both, half the LDS traffic of l3_pass_bf3 (the kernel is power-bound: operand bytes cost clock).  Every tile sees
// the same products in the same order, so the maxima are bit-identical to l3_pass_bf3's.  ring.r[2 i + j] = fragment (k-step
// parity i, n-tile j of the pair), two k-steps (1 536 matrix cycles) ahead, rolling over pair and chunk boundaries.
// PROBE (diagnostic instantiations, tuning key 5 = 91 / 92 / 93, results WRONG, timing only): bit 0 = the weight fragments are
// never refilled (no L2 / L1 traffic in the loop), bit 1 = the A terms are read from LDS for the first k-step of a pass only.
template <int MTS, int PROBE = 0>
__device__ __forceinline__ void l3_pass_bf3_2x2(const __amdgpu_buffer_rsrc_t& rs, int voff, const unsigned char* abase, int wave_s,
                                                float (&runmax)[8], B3Ring& ring) {
#pragma unroll 1
    for (int qp = 0; qp < 4; ++qp) {
        f32x16 acc[MTS][2];
#pragma unroll
        for (int i = 0; i < MTS; ++i) { acc[i][0] = (f32x16){0}; acc[i][1] = (f32x16){0}; }
        bf16x8 af[MTS][3];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (!(PROBE & 2) || s == 0) {
#pragma unroll
                for (int i = 0; i < MTS; ++i) a3_load<kPlaneB>(af[i], abase + i * 32 * kLdB, s);
            }
            const B3 b[2] = {ring.r[2 * (s & 1)], ring.r[2 * (s & 1) + 1]};
            // k-step s + 2 of this pair, or k-step s - 6 of the next one (b3_load takes the n-tile index mod 8)
            const int qn = s + 2 < 8 ? 2 * qp : 2 * qp + 2, sn = (s + 2) & 7;
            if (!(PROBE & 1)) {
                ring.r[2 * (s & 1)] = b3_load(rs, voff, qn * 8 + sn, wave_s);
                ring.r[2 * (s & 1) + 1] = b3_load(rs, voff, (qn + 1) * 8 + sn, wave_s);
            }
            mfma_bf3_block<MTS, 2>(af, b, acc);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float m = max16(acc[0][j]);
            if (MTS == 2) m = fmaxf(m, max16(acc[MTS - 1][j]));
#pragma unroll
            for (int i = 0; i < 8; ++i) runmax[i] = (i == 2 * qp + j) ? fmaxf(runmax[i], m) : runmax[i];
        }
    }
}

// L3 for a LAST m-tile of at most 16 rows (rows row0 .. row0 + 15 of act2): v_mfma_f32_16x16x4_f32 instead of a 32-row tile
// of which half would be padding (a coalition's row count is uniform mod 32, so this saves a quarter tile per coalition and
// chain on average: 1.5 % of the kernel).  BIT-IDENTICAL to the 32x32x2 path: that instruction accumulates its two k values
// (8 kb + j for half h = 0, 8 kb + 4 + j for h = 1) and the 16x16x4 instruction its four as ONE sequential fma chain in operand
// order, so feeding the four lane groups kq = 0..3 with k = (j, 4 + j, j + 1, 5 + j), j = 0 then 2, repeats the chain of the
// steps (j, j + 1) of a k-block exactly (tools/micro/mfma_tail.hip: 0 of 102 400 outputs differ; the plain k order differs in
// 77 %).  Both operands come out of the SAME images: lane (r16 = lane & 15, kq) reads the float4 of act2 row row0 + r16 at
// k = 8 kb + 4 (kq & 1) and the float4 of weight-fragment lane (16 half + r16) + 32 (kq & 1), and takes elements (kq >> 1) and
// (kq >> 1) + 2.  C layout of 16x16x4: element i of lane l = row 4 (l >> 4) + i, column l & 15.
__device__ __forceinline__ void l3_tail16(const WBuf& w3, const float* act2, int row0, int wave_s, int lane, float (&runmax)[8],
                                          BRing& ring) {
    const int r16 = lane & 15, kq = lane >> 4, hsel = kq & 1;
    const bool odd = (kq >> 1) != 0;
    float a0[16], a1[16];
    const float* arow = act2 + (row0 + r16) * kLd2 + 4 * hsel;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(arow + 8 * kb);
        a0[kb] = odd ? v[1] : v[0];
        a1[kb] = odd ? v[3] : v[2];
    }
    const int voff0 = (r16 + 32 * hsel) * 16, voff1 = (16 + r16 + 32 * hsel) * 16;   // bytes inside a fragment, column halves 0 / 1
    // the B ring of the 32-row passes (dead by now: this is the item's last tile) holds the weight float4s of 4 k-blocks x 2
    // column halves, 4 k-blocks (16 MFMAs = 512 cycles) ahead, rolling over the n-tile boundaries
    const int w0 = wave_s * 16 * kFragBytes;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        ring.r[2 * d] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w3.rsrc, voff0, w0 + d * kFragBytes, 0));
        ring.r[2 * d + 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w3.rsrc, voff1, w0 + d * kFragBytes, 0));
    }
    int wq = w0;
#pragma unroll 1
    for (int q = 0; q < 8; ++q) {
        const int wn = (q < 7) ? wq + 4 * 16 * kFragBytes : w0;
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            const f32x4 b0 = ring.r[2 * (kb & 3)], b1 = ring.r[2 * (kb & 3) + 1];
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[kb], odd ? b0[1] : b0[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[kb], odd ? b1[1] : b1[0], c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[kb], odd ? b0[3] : b0[2], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[kb], odd ? b1[3] : b1[2], c1, 0, 0, 0);
            const int soff = kb < 12 ? wq + (kb + 4) * kFragBytes : wn + (kb - 12) * kFragBytes;
            ring.r[2 * (kb & 3)] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w3.rsrc, voff0, soff, 0));
            ring.r[2 * (kb & 3) + 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w3.rsrc, voff1, soff, 0));
        }
        float m0 = fmaxf(fmaxf(c0[0], c0[1]), fmaxf(c0[2], c0[3])), m1 = fmaxf(fmaxf(c1[0], c1[1]), fmaxf(c1[2], c1[3]));
        m0 = fmaxf(m0, __shfl_xor(m0, 16)); m1 = fmaxf(m1, __shfl_xor(m1, 16));
        m0 = fmaxf(m0, __shfl_xor(m0, 32)); m1 = fmaxf(m1, __shfl_xor(m1, 32));
        const float m = (lane & 16) ? m1 : m0;            // lane l keeps column l & 31 of the n-tile, as the 32x32 path does
#pragma unroll
        for (int i = 0; i < 8; ++i) runmax[i] = (i == q) ? fmaxf(runmax[i], m) : runmax[i];
        wq = wn;
    }
}

// Diagnostic build only (STAMP): per-phase shader-clock sums, added to a.stamps by lane 0 of every wave.
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define IQ_STAMP(slot)                                              \
    do {                                                            \
        if (STAMP) {                                                \
            const unsigned long long t_ = stamp_now();              \
            tsum[slot] += (unsigned)(t_ - tlast);                   \
            tlast = t_;                                             \
        }                                                           \
    } while (0)

template <int MODE, int L3V, bool STAMP = false, bool ARGMAX = false, int PROBE = 0>
__global__ __launch_bounds__(kThreads, L3V == 3 ? 2 : 3) void pn_chain_kernel(ChainArgs a) {
    // act0 (ld 68) then act2: float image (ld 132), or - L3V = 3 - three bf16 planes of 272-byte rows
    __shared__ __attribute__((aligned(16))) float bufA[L3V == 3 ? 3 * kPlaneB / 4 : kMC * kLd2];
    __shared__ __attribute__((aligned(16))) float bufB[L3V == 3 ? 3 * kPlane1B / 4 : kMC * kLd1];  // act1 (L3V = 3: three bf16 planes)
    __shared__ __attribute__((aligned(16))) float xs[kMC * 4];       // transformed inputs of the chunk (x,y,z,-)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int item = a.item_order ? a.item_order[blockIdx.x] : blockIdx.x;
    const int N = a.N;
    int cloud;
    if (MODE == kPrepool) cloud = item / (a.R + a.with_centre);
    else cloud = a.cloud_of ? a.cloud_of[item] : (a.nclouds == 1 ? 0 : item);

    const int nrows = a.nrows[item] & 0xffff;
    float* outp = a.out + (size_t)item * kFeat;
    if (nrows == 0) {  // empty region in the pre-pool: identity of max
        for (int c = tid; c < kFeat; c += kThreads) outp[c] = -INFINITY;
        return;
    }
    const int nchunks = (nrows + kMC - 1) / kMC;

    const int c0 = tid & 63, rg = tid >> 6;
    const f32x4 win = *reinterpret_cast<const f32x4*>(a.w_in + c0 * 4);
    const int wave_s = uniform(wave);
    // weight images as buffer resources (kTrunk: layer 1 is this item's packed 64x64 transform)
    const WBuf w1b = wbuf_make((MODE == kTrunk) ? a.w1 + (size_t)item * 4096 : a.w1, lane);
    const WBuf w2b = wbuf_make(a.w2, lane), w3b = wbuf_make(a.w3, lane);
    // per-lane bases: A-fragment reads and C-tile writes are base + immediate everywhere below
    const int frag_lane = (lane & 31), frag_h = lane >> 5;
    const float* a1base_A = bufA + frag_lane * kLd1 + 4 * frag_h;   // act0 in bufA
    const float* a1base_B = bufB + frag_lane * kLd1 + 4 * frag_h;   // act1 in bufB
    const float* a2base = bufA + frag_lane * kLd2 + 4 * frag_h;     // act2 in bufA
    float* c1base = bufB + (4 * frag_h) * kLd1 + frag_lane;         // C tiles of L1 -> act1
    float* c2base = bufA + (4 * frag_h) * kLd2 + frag_lane;         // C tiles of L2 -> act2

    float runmax[8];
    int runarg[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { runmax[q] = -INFINITY; runarg[q] = 0; }

    // Input points travel ahead in registers: lanes 0..15 of each wave own 16 rows of a chunk.  The
    // row index of chunk c+2 and the coordinates of chunk c+1 are requested while chunk c computes, so
    // neither of the two dependent global loads is ever waited for.
    const bool fetcher = lane < 16;
    const int frow = wave * 16 + lane;
    const uint16_t* rowp = a.rows + (size_t)item * kRowCap + frow;
    float px = 0.f, py = 0.f, pz = 0.f;
    auto load_point = [&](int p) {
        if (p == N) {
            px = a.centers[cloud * 3]; py = a.centers[cloud * 3 + 1]; pz = a.centers[cloud * 3 + 2];
        } else {
            const float* src = a.clouds + (size_t)cloud * a.cl + (size_t)p * a.ps;
            px = src[0]; py = src[a.cs]; pz = src[2 * a.cs];
        }
    };
    int pnext = 0;
    if (fetcher) {
        load_point(rowp[0]);
        if (nchunks > 1) pnext = rowp[kMC];
    }

    BRing ring;
    if (L3V == 2) bring_init(ring, w3b, wave_s);
    [[maybe_unused]] B3Ring ring3;
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t w3rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(L3V == 3 ? a.w3_bf3 : nullptr), 0, 0x7fffffff, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t w2rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(L3V == 3 ? a.w2_bf3 : nullptr), 0, 0x7fffffff, 0x00020000);
    if (L3V == 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i)   // l3_pass_bf3: k-steps 0..3 of n-tile 0; l3_pass_bf3_2x2: k-steps 0..1 of n-tiles 0 and 1
            ring3.r[i] = b3_load(w3rs, lane * 16, ARGMAX || a.l3_single ? i : (i & 1) * 8 + (i >> 1), wave_s);
    }
    unsigned tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = STAMP ? stamp_now() : 0ull;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int rows_here = min(kMC, nrows - ch * kMC);
        const int mts = rows_here > 32 ? 2 : 1;

        // ---- stage 0a: input transform (models/pointnet.py:67-69) --------------------------
        if (fetcher) {
            float x = px, y = py, z = pz;
            if (MODE != kPrepool) {
                const float* t9 = a.trans + (size_t)item * 9;  // uniform -> scalar loads
                const float x2 = fmaf(z, t9[6], fmaf(y, t9[3], x * t9[0]));
                const float y2 = fmaf(z, t9[7], fmaf(y, t9[4], x * t9[1]));
                const float z2 = fmaf(z, t9[8], fmaf(y, t9[5], x * t9[2]));
                x = x2; y = y2; z = z2;
            }
            *reinterpret_cast<f32x4*>(xs + frow * 4) = (f32x4){x, y, z, 0.f};
        }
        IQ_STAMP(0);
        __syncthreads();  // also orders the previous chunk's L3 reads of bufA before stage 0b rewrites it
        IQ_STAMP(1);
        if (fetcher && ch + 1 < nchunks) {
            load_point(pnext);
            if (ch + 2 < nchunks) pnext = rowp[(ch + 2) * kMC];
        }
        // ---- stage 0b: 3 -> 64 (+bn, relu), thread = (channel c0, 16 rows) -----------------
        {
            float* dst = ((MODE == kPrepool) ? bufB : bufA) + rg * 16 * kLd1 + c0;
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + (rg * 16 + i) * 4);
                const float f = fmaf(win[2], xv[2], fmaf(win[1], xv[1], win[0] * xv[0])) + win[3];
                dst[i * kLd1] = fmaxf(f, 0.f);
            }
        }
        // ---- L1: 64 -> 64 ------------------------------------------------------------------
        if (MODE != kPrepool) {
            const int mt = wave & 1, nt = wave >> 1;
            const int wq = (wave_s >> 1) * 8 * kFragBytes;
            f32x4 bw[8];  // weight fragments are requested before the barrier, consumed after it
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) bw[kb] = wbuf_load(w1b, wq + kb * kFragBytes);
            IQ_STAMP(2);
            __syncthreads();
            IQ_STAMP(3);
            if (mt < mts) {
                f32x16 acc = {0};
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) {
                    // L3V = 3: TRANSPOSED tile (the weight fragment as the A operand: iq_bf3.h, ct_tile_to_planes) - the same products in
                    // the same order, the lane then holds its row's channels in register quads and act1 needs no two-lane trade
                    if (L3V == 3) acc = mfma4(bw[kb], lds_frag<kLd1>(a1base_A + mt * 32 * kLd1, 0, kb), acc);
                    else acc = mfma4(lds_frag<kLd1>(a1base_A + mt * 32 * kLd1, 0, kb), bw[kb], acc);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const float bias = (MODE == kFstn && L3V != 3) ? a.b1[nt * 32 + frag_lane] : 0.f;
                if (L3V == 3) {   // act1 as three bf16 planes for layer 2; register r = channel c_row_i(r) + 4 frag_h of the n-tile
                    f32x4 bq[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                    if (MODE == kFstn) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const f32x4*>(a.b1 + nt * 32 + 8 * g + 4 * frag_h);
                    }
                    ct_tile_to_planes<kLd1B, kPlane1B>(reinterpret_cast<unsigned char*>(bufB) + mt * 32 * kLd1B + nt * 64, lane, [&](int r) {
                        const float v = acc[r] + bq[r >> 2][r & 3];
                        return (MODE == kFstn) ? fmaxf(v, 0.f) : v;
                    });
                } else {
                    float* dst = c1base + mt * 32 * kLd1 + nt * 32;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float v = acc[i] + bias;
                        if (MODE == kFstn) v = fmaxf(v, 0.f);
                        dst[c_row_i(i) * kLd1] = v;
                    }
                }
            }
        }
        // ---- L2: 64 -> 128 (+bn, relu): two passes (n-tiles nt0, nt0+2); the second pass's weights
        //      are requested while the first pass computes ------------------------------------
        if (L3V == 3) {   // on the bf16 matrix pipe: act1 and the weights as three bf16 terms, six products each (float32-exact)
            const int mt = wave & 1, nt0 = wave >> 1, nts = wave_s >> 1;
            constexpr int PF = ARGMAX ? 2 : 4;   // weight fragments in flight (the arg-max variant has no registers to spare)
            B3 bw[PF];
#pragma unroll
            for (int ks = 0; ks < PF; ++ks) bw[ks] = b3_load_l2(w2rs, lane * 16, nts, ks);
            IQ_STAMP(4);
            __syncthreads();
            IQ_STAMP(3);
            if (mt < mts) {
                const unsigned char* arow = reinterpret_cast<const unsigned char*>(bufB) + (mt * 32 + frag_lane) * kLd1B + frag_h * 16;
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    f32x16 acc = {0};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        bf16x8 af[3];
#pragma unroll
                        for (int e = 0; e < 3; ++e) af[e] = *reinterpret_cast<const bf16x8*>(arow + e * kPlane1B + ks * 32);
                        const int t = pass * 4 + ks, nx = t + PF;
                        const B3 b = bw[t % PF];
                        if (nx < 8) bw[t % PF] = b3_load_l2(w2rs, lane * 16, nts + 2 * (nx >> 2), nx & 3);
                        acc = mfma_bf3_tr(b, af, acc);      // transposed tile, as layer 1
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    f32x4 bq[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const f32x4*>(a.b2 + nt0 * 32 + pass * 64 + 8 * g + 4 * frag_h);
                    ct_tile_to_planes<kLdB, kPlaneB>(reinterpret_cast<unsigned char*>(bufA) + mt * 32 * kLdB + (nt0 * 32 + pass * 64) * 2, lane,
                                                     [&](int r) { return fmaxf(acc[r] + bq[r >> 2][r & 3], 0.f); });
                }
            }
        } else {
            const int mt = wave & 1, nt0 = wave >> 1;
            const int wq0 = (wave_s >> 1) * 8 * kFragBytes;
            f32x4 bw[8];
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) bw[kb] = wbuf_load(w2b, wq0 + kb * kFragBytes);
            IQ_STAMP(4);
            __syncthreads();
            IQ_STAMP(3);
            if (mt < mts) {
                const float* arow = a1base_B + mt * 32 * kLd1;
                float* dst = c2base + mt * 32 * kLd2 + nt0 * 32;
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    f32x16 acc = {0};
#pragma unroll
                    for (int kb = 0; kb < 8; ++kb) {
                        acc = mfma4(lds_frag<kLd1>(arow, 0, kb), bw[kb], acc);
                        if (pass == 0) bw[kb] = wbuf_load(w2b, wq0 + (2 * 8 + kb) * kFragBytes);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const float bias = a.b2[nt0 * 32 + pass * 64 + frag_lane];
#pragma unroll
                    for (int i = 0; i < 16; ++i) dst[c_row_i(i) * kLd2 + pass * 64] = fmaxf(acc[i] + bias, 0.f);
                }
            }
        }
        IQ_STAMP(5);
        __syncthreads();
        IQ_STAMP(3);
        // ---- L3: 128 -> 1024, running column max -------------------------------------------
        if (L3V == 3) {
            const unsigned char* ab3 = reinterpret_cast<const unsigned char*>(bufA) + frag_lane * kLdB + frag_h * 16;
            if (ARGMAX || a.l3_single) {
                if (mts == 2) l3_pass_bf3<2, ARGMAX>(w3rs, lane * 16, ab3, wave_s, runmax, ring3, runarg, ch * kMC, frag_h);
                else          l3_pass_bf3<1, ARGMAX>(w3rs, lane * 16, ab3, wave_s, runmax, ring3, runarg, ch * kMC, frag_h);
            } else {
                if (mts == 2) l3_pass_bf3_2x2<2, PROBE>(w3rs, lane * 16, ab3, wave_s, runmax, ring3);
                else          l3_pass_bf3_2x2<1, PROBE>(w3rs, lane * 16, ab3, wave_s, runmax, ring3);
            }
        } else if (L3V == 0) {
            if (mts == 2) l3_pass_v0<2>(w3b, a2base, wave_s, runmax);
            else          l3_pass_v0<1>(w3b, a2base, wave_s, runmax);
        } else {
            const bool tail = !ARGMAX && a.tail16 && rows_here - 32 * (mts - 1) <= 16;   // (uniform) the last m-tile holds <= 16 rows
            if (mts == 2 && !tail) l3_pass_v2<2, ARGMAX>(w3b, a2base, wave_s, runmax, ring, runarg, ch * kMC, frag_h);
            else if (mts == 2 || !tail) l3_pass_v2<1, ARGMAX>(w3b, a2base, wave_s, runmax, ring, runarg, ch * kMC, frag_h);
            if (tail) l3_tail16(w3b, bufA, 32 * (mts - 1), wave_s, lane, runmax, ring);
        }
        IQ_STAMP(6);
    }
    if (STAMP && lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(&a.stamps[i], (unsigned long long)tsum[i]);
    }

    // max commutes with the (monotone) per-column bias add and relu
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        float v = runmax[q];
        const int n = (q * 4 + wave) * 32 + (lane & 31);
        if (ARGMAX) {   // the other half-wave saw rows 4..7 (+8 k) of every tile: larger value wins, then the lower row
            const float v2 = __shfl_xor(v, 32);
            const int r2 = __shfl_xor(runarg[q], 32);
            const int r = (v2 > v || (v2 == v && r2 < runarg[q])) ? r2 : runarg[q];
            if (lane < 32 && a.argrow) a.argrow[(size_t)item * kFeat + n] = (int)a.rows[(size_t)item * kRowCap + r];   // row -> point
        }
        v = fmaxf(v, __shfl_xor(v, 32));
        v += a.b3[n];
        if (MODE != kTrunk) v = fmaxf(v, 0.f);
        if (lane < 32) outp[n] = v;
    }
}

// ---- per-cloud region tables: points grouped by region (counting sort) ------------------------
__global__ __launch_bounds__(kThreads) void pn_prepare_kernel(const int32_t* __restrict__ region_id,
                                                              uint16_t* __restrict__ sorted_pts,
                                                              int32_t* __restrict__ roff, int N, int R) {
    __shared__ int16_t rid[kMaxN];
    __shared__ int cnt[IQ_MAX_REGIONS + 1];
    const int cloud = blockIdx.x;
    if (threadIdx.x <= IQ_MAX_REGIONS) cnt[threadIdx.x] = 0;
    for (int p = threadIdx.x; p < N; p += kThreads) {
        // an id outside [0,R) (rejected by iq_check_index_range; never produced by iq_region_assign) goes to bucket R,
        // which no coalition keeps: the point counts as masked instead of indexing LDS / the workspace out of bounds
        const int r = region_id[(size_t)cloud * N + p];
        rid[p] = (int16_t)((unsigned)r < (unsigned)R ? r : R);
    }
    __syncthreads();
    for (int p = threadIdx.x; p < N; p += kThreads) atomicAdd(&cnt[rid[p]], 1);
    __syncthreads();
    if (threadIdx.x == 0) {  // exclusive scan, R <= 64
        int run = 0;
        for (int r = 0; r <= R; ++r) { const int c = r < R ? cnt[r] : 0; cnt[r] = run; run += c; }
    }
    __syncthreads();
    if (threadIdx.x <= R) roff[(size_t)cloud * (R + 1) + threadIdx.x] = cnt[threadIdx.x];
    for (int p = threadIdx.x; p < N; p += kThreads) {
        const int r = rid[p];
        if (r >= R) continue;
        int k = 0;
        for (int q = 0; q < p; ++q) k += (rid[q] == r);   // rank inside the region: ascending point index
        sorted_pts[(size_t)cloud * N + cnt[r] + k] = (uint16_t)p;
    }
}

// ---- compacted row list of every work item (one wave per item) --------------------------------
// Coalition items: keep[item] (null = everything), centre appended iff a point is masked.
// Pre-pool items (prepool != 0): item = cloud*(R+with_centre) + r; r < R keeps region r only (no
// centre), r == R is the centre alone.  The list is padded to a multiple of kMC with replicas of a
// valid row (duplicates never change a max).
__global__ __launch_bounds__(64) void pn_rows_kernel(const uint16_t* __restrict__ sorted_all,
                                                     const int32_t* __restrict__ roff_all,
                                                     const uint64_t* __restrict__ keep_all,
                                                     const int32_t* __restrict__ cloud_of, uint16_t* __restrict__ rows_all,
                                                     int32_t* __restrict__ nrows_all, int N, int R, int nclouds,
                                                     int with_centre, int prepool) {
    const int item = blockIdx.x, lane = threadIdx.x;
    int cloud;
    uint64_t keep;
    bool add_centre;
    if (prepool) {
        const int per = R + with_centre;
        cloud = item / per;
        const int r = item - cloud * per;
        keep = r < R ? (1ull << r) : 0ull;
        add_centre = (r == R);
    } else {
        cloud = cloud_of ? cloud_of[item] : (nclouds == 1 ? 0 : item);
        const uint64_t full = R >= 64 ? ~0ull : ((1ull << R) - 1);
        keep = (keep_all ? keep_all[item] : ~0ull) & full;
        add_centre = false;
    }
    const int32_t* roff = roff_all + (size_t)cloud * (R + 1);
    const uint16_t* sorted_pts = sorted_all + (size_t)cloud * N;
    uint16_t* rows = rows_all + (size_t)item * kRowCap;

    const int sz = (lane < R && ((keep >> lane) & 1)) ? roff[lane + 1] - roff[lane] : 0;
    int inc = sz;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    const int kst = inc - sz;
    const int nkept = __shfl(inc, 63);
    if (!prepool) add_centre = with_centre && nkept < N;
    const int nrows = nkept + (add_centre ? 1 : 0);
    for (int r = 0; r < R; ++r) {
        if (!((keep >> r) & 1)) continue;
        const int ks = __shfl(kst, r), off = roff[r], n = roff[r + 1] - off;
        for (int j = lane; j < n; j += 64) rows[ks + j] = sorted_pts[off + j];
    }
    // padding value: the centre if present, else the last kept point
    int padval = N;
    if (!add_centre && nkept > 0) {
        const unsigned long long last = __ballot(sz > 0 && kst + sz == nkept);
        const int rl = __ffsll((long long)last) - 1;
        padval = sorted_pts[roff[rl + 1] - 1];
    }
    const int npad = (nrows + kMC - 1) / kMC * kMC;
    for (int i = nkept + lane; i < npad; i += 64) rows[i] = (uint16_t)padval;
    if (lane == 0) nrows_all[item] = nrows | (add_centre ? (1 << 16) : 0);
}

// ---- launch order: largest coalitions first (LPT), a counting sort on the chunk count ---------
constexpr int kBins = kMaxN / kMC + 2;

__global__ __launch_bounds__(kThreads) void pn_order_count_kernel(const int32_t* __restrict__ nrows_all,
                                                                  int32_t* __restrict__ bin_of, int32_t* __restrict__ hist,
                                                                  int B) {
    __shared__ int lh[kBins];   // per-workgroup histogram: one global atomic per bin and workgroup instead of one per item
    if (threadIdx.x < kBins) lh[threadIdx.x] = 0;
    __syncthreads();
    const int item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item < B) {
        const int rows = nrows_all[item] & 0xffff;
        const int bin = kBins - 1 - min(kBins - 1, (rows + kMC - 1) / kMC);  // bin 0 = most chunks
        bin_of[item] = bin;
        atomicAdd(&lh[bin], 1);
    }
    __syncthreads();
    if (threadIdx.x < kBins && lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
}

__global__ void pn_order_scan_kernel(int32_t* __restrict__ hist) {
    if (threadIdx.x == 0) {
        int run = 0;
        for (int b = 0; b < kBins; ++b) { const int c = hist[b]; hist[b] = run; run += c; }
    }
}

// position inside a bin: the workgroup reserves a range per bin with one global atomic, its items take consecutive slots
// (the order inside a bin is irrelevant: equal chunk counts)
__global__ __launch_bounds__(kThreads) void pn_order_scatter_kernel(const int32_t* __restrict__ bin_of,
                                                                    int32_t* __restrict__ cursor,
                                                                    int32_t* __restrict__ order, int B) {
    __shared__ int lh[kBins], base[kBins];
    if (threadIdx.x < kBins) lh[threadIdx.x] = 0;
    __syncthreads();
    const int item = blockIdx.x * blockDim.x + threadIdx.x;
    int bin = 0, pos = 0;
    if (item < B) {
        bin = bin_of[item];
        pos = atomicAdd(&lh[bin], 1);
    }
    __syncthreads();
    if (threadIdx.x < kBins && lh[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], lh[threadIdx.x]);
    __syncthreads();
    if (item < B) order[base[bin] + pos] = item;
}

// ---- pooled input-STN feature of a coalition: max over kept regions (+ centre) -------------
__global__ __launch_bounds__(kThreads) void pn_stn_gather_kernel(const float* __restrict__ G,
                                                                 const int32_t* __restrict__ nrows_all,
                                                                 const uint64_t* __restrict__ keep,
                                                                 const int32_t* __restrict__ cloud_of,
                                                                 float* __restrict__ out, int R, int nclouds,
                                                                 int with_centre) {
    const int item = blockIdx.x;
    const int cloud = cloud_of ? cloud_of[item] : (nclouds == 1 ? 0 : item);
    const uint64_t full = R >= 64 ? ~0ull : ((1ull << R) - 1);
    const uint64_t k = (keep ? keep[item] : ~0ull) & full;
    const int per = R + with_centre;
    const f32x4* g4 = reinterpret_cast<const f32x4*>(G) + (size_t)cloud * per * (kFeat / 4) + threadIdx.x;
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int r = 0; r < R; ++r) {
        if ((k >> r) & 1) {
            const f32x4 v = g4[(size_t)r * (kFeat / 4)];
            m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
        }
    }
    if (nrows_all[item] >> 16) {  // a centre row exists (something was masked)
        const f32x4 v = g4[(size_t)R * (kFeat / 4)];
        m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
    }
    reinterpret_cast<f32x4*>(out)[(size_t)item * (kFeat / 4) + threadIdx.x] = m;
}

// every item's 64 x 64 transform := one packed image (4096 floats), float4 per thread
__global__ __launch_bounds__(kThreads) void pn_fill_rows_kernel(float* __restrict__ out, const float* __restrict__ image, int B) {
    const size_t t = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (t >= (size_t)B * 1024) return;
    reinterpret_cast<f32x4*>(out)[t] = reinterpret_cast<const f32x4*>(image)[t & 1023];
}

template <int MODE>
void launch_chain(const ChainArgs& a, hipStream_t st) {
    const size_t extra_lds = (size_t)iq::tuning(iq::kTuneExtraLds);  // experiment: lower the occupancy
    const int knob = iq::tuning(iq::kTuneExperiment);
    const bool fp32_l3 = knob == 54 || knob == 55;   // layer 3 on the fp32 MFMA (round 3's kernel; A/B and tests), 55: without tail16
    if (a.stamps && MODE == kFstn)
        hipLaunchKernelGGL((pn_chain_kernel<kFstn, 2, true>), dim3(a.items), dim3(kThreads), extra_lds, st, a);
    else if (iq::tuning(iq::kTuneL3Variant) == 0)
        hipLaunchKernelGGL((pn_chain_kernel<MODE, 0>), dim3(a.items), dim3(kThreads), extra_lds, st, a);
    else if (MODE == kTrunk && a.argrow && a.w3_bf3 && a.w2_bf3 && !fp32_l3)
        hipLaunchKernelGGL((pn_chain_kernel<kTrunk, 3, false, true>), dim3(a.items), dim3(kThreads), extra_lds, st, a);
    else if (MODE == kTrunk && a.argrow)
        hipLaunchKernelGGL((pn_chain_kernel<kTrunk, 2, false, true>), dim3(a.items), dim3(kThreads), extra_lds, st, a);
    else if (MODE != kPrepool && a.w3_bf3 && a.w2_bf3 && !fp32_l3 && knob >= 91 && knob <= 93) {   // timing probes, results wrong
        if (knob == 91) hipLaunchKernelGGL((pn_chain_kernel<MODE == kPrepool ? kFstn : MODE, 3, false, false, 1>), dim3(a.items), dim3(kThreads), extra_lds, st, a);
        else if (knob == 92) hipLaunchKernelGGL((pn_chain_kernel<MODE == kPrepool ? kFstn : MODE, 3, false, false, 2>), dim3(a.items), dim3(kThreads), extra_lds, st, a);
        else hipLaunchKernelGGL((pn_chain_kernel<MODE == kPrepool ? kFstn : MODE, 3, false, false, 3>), dim3(a.items), dim3(kThreads), extra_lds, st, a);
    }
    else if (MODE != kPrepool && a.w3_bf3 && a.w2_bf3 && !fp32_l3)
        hipLaunchKernelGGL((pn_chain_kernel<MODE, 3>), dim3(a.items), dim3(kThreads), extra_lds, st, a);
    else
        hipLaunchKernelGGL((pn_chain_kernel<MODE, 2>), dim3(a.items), dim3(kThreads), extra_lds, st, a);
}

}  // namespace

namespace {
using iq::launch_linear;

struct Workspace {
    uint16_t* sorted_pts;
    int32_t* roff;
    uint16_t* rows_pre;  // (nclouds*(R+1), kRowCap)
    int32_t* nrows_pre;
    uint16_t* rows;      // (B, kRowCap)
    int32_t* nrows;      // (B)
    int32_t* order;    // (B) launch order
    int32_t* bin_of;   // (B)
    int32_t* hist;     // (kBins)
    float* G;
    float* gbuf;
    float* h1;
    float* h2;
    float* trans;
    float* tfp;
    size_t bytes;
};

Workspace carve(void* base, int B, int nclouds, int N, int R) {
    Workspace w;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = iq::align_up(off + bytes, 256);
        return reinterpret_cast<char*>(base) + o;
    };
    w.sorted_pts = reinterpret_cast<uint16_t*>(take((size_t)nclouds * N * sizeof(uint16_t)));
    w.roff = reinterpret_cast<int32_t*>(take((size_t)nclouds * (IQ_MAX_REGIONS + 1) * sizeof(int32_t)));
    w.rows_pre = reinterpret_cast<uint16_t*>(take((size_t)nclouds * (R + 1) * kRowCap * sizeof(uint16_t)));
    w.nrows_pre = reinterpret_cast<int32_t*>(take((size_t)nclouds * (R + 1) * sizeof(int32_t)));
    w.rows = reinterpret_cast<uint16_t*>(take((size_t)B * kRowCap * sizeof(uint16_t)));
    w.nrows = reinterpret_cast<int32_t*>(take((size_t)B * sizeof(int32_t)));
    w.order = reinterpret_cast<int32_t*>(take((size_t)B * sizeof(int32_t)));
    w.bin_of = reinterpret_cast<int32_t*>(take((size_t)B * sizeof(int32_t)));
    w.hist = reinterpret_cast<int32_t*>(take((size_t)kBins * sizeof(int32_t)));
    w.G = reinterpret_cast<float*>(take((size_t)nclouds * (R + 1) * kFeat * sizeof(float)));
    w.gbuf = reinterpret_cast<float*>(take((size_t)B * kFeat * sizeof(float)));
    w.h1 = reinterpret_cast<float*>(take((size_t)B * 512 * sizeof(float)));
    w.h2 = reinterpret_cast<float*>(take((size_t)B * 256 * sizeof(float)));
    w.trans = reinterpret_cast<float*>(take((size_t)B * 9 * sizeof(float)));
    w.tfp = reinterpret_cast<float*>(take((size_t)B * 4096 * sizeof(float)));
    w.bytes = off;
    return w;
}

}  // namespace

namespace { unsigned long long* g_stamps = nullptr; }

// Diagnostic: phase stamps of the feature-STN chain (STAMP build).  enable allocates/zeroes a
// device buffer of 8 counters; read copies them to the host (synchronises the device).
extern "C" int iq_debug_stamps(int enable, unsigned long long* out_host /*8 or null*/) {
    if (out_host && g_stamps) {
        if (hipDeviceSynchronize() != hipSuccess) return IQ_ELAUNCH;
        if (hipMemcpy(out_host, g_stamps, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return IQ_ELAUNCH;
    }
    if (enable) {
        if (!g_stamps && hipMalloc(&g_stamps, 8 * sizeof(unsigned long long)) != hipSuccess) return IQ_ELAUNCH;
        if (hipMemset(g_stamps, 0, 8 * sizeof(unsigned long long)) != hipSuccess) return IQ_ELAUNCH;
    } else if (g_stamps) {
        (void)hipFree(g_stamps);
        g_stamps = nullptr;
    }
    return IQ_OK;
}

extern "C" int iq_debug_chain_occupancy(void) {
    int n0 = -1, n2 = -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n0, pn_chain_kernel<kFstn, 0>, kThreads, 0) != hipSuccess) return -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n2, pn_chain_kernel<kFstn, 2>, kThreads, 0) != hipSuccess) return -1;
    return n0 * 100 + n2;
}

extern "C" size_t iq_pointnet_workspace_bytes(int B, int nclouds, int N, int R) {
    if (B < 0 || nclouds < 0 || N < 0 || R < 0) return 0;
    return carve(nullptr, B, nclouds, N, R).bytes;
}

extern "C" double iq_pointnet_flops_per_coalition(int N) {
    // MACs per point: stn 3*64+64*128+128*1024, bmm 9, conv1 192, fstn 64*64+64*128+128*1024,
    // bmm 4096, conv2 8192, conv3 131072 = 426377; FC heads: stn 1024*512+512*256+256*9,
    // fstn ...+256*4096, cls ...+256*10
    const double per_point = 426377.0;
    const double fc = 3.0 * (1024.0 * 512 + 512.0 * 256) + 256.0 * (9 + 4096 + 10);
    return 2.0 * (per_point * N + fc);
}

extern "C" int iq_pointnet_coalitions(const iq_pointnet_weights* w, const float* clouds, const float* centers,
                                      const int32_t* region_id, const uint64_t* keep, const int32_t* cloud_of,
                                      float* logits, float* trans_feat_packed, void* workspace,
                                      size_t workspace_bytes, int B, int nclouds, int N, int R,
                                      int channel_first, iq_stream_t stream) {
    return iq_pointnet_coalitions_crt(w, clouds, centers, region_id, keep, cloud_of, logits, trans_feat_packed, nullptr, workspace,
                                      workspace_bytes, B, nclouds, N, R, channel_first, stream);
}

extern "C" int iq_pointnet_coalitions_crt(const iq_pointnet_weights* w, const float* clouds, const float* centers,
                                          const int32_t* region_id, const uint64_t* keep, const int32_t* cloud_of,
                                          float* logits, float* trans_feat_packed, int32_t* crt_points, void* workspace,
                                          size_t workspace_bytes, int B, int nclouds, int N, int R,
                                          int channel_first, iq_stream_t stream) {
    IQ_REQUIRE(B >= 0 && nclouds >= 1, "iq_pointnet_coalitions: B=%d nclouds=%d", B, nclouds);
    IQ_REQUIRE(w && clouds && region_id && (logits || B == 0), "iq_pointnet_coalitions: null pointer");
    IQ_REQUIRE(N >= 1 && N <= kMaxN, "iq_pointnet_coalitions: N=%d not in [1,%d]", N, kMaxN);
    IQ_REQUIRE(R >= 1 && R <= IQ_MAX_REGIONS, "iq_pointnet_coalitions: R=%d not in [1,%d]", R, IQ_MAX_REGIONS);
    IQ_REQUIRE(cloud_of || nclouds == 1 || nclouds == B, "iq_pointnet_coalitions: cloud_of required when 1 < nclouds != B");
    const int with_centre = centers ? 1 : 0;  // no centre = dense mode (nothing is ever masked)
    IQ_REQUIRE(centers || !keep, "iq_pointnet_coalitions: keep masks need centers");
    IQ_REQUIRE(!crt_points || iq::tuning(iq::kTuneL3Variant) == 2, "iq_pointnet_coalitions_crt: crt_points need the default L3 variant");
    if (B == 0) return IQ_OK;
    const size_t need = iq_pointnet_workspace_bytes(B, nclouds, N, R);
    if (!workspace || workspace_bytes < need)
        return iq::fail(IQ_EWORKSPACE, "iq_pointnet_coalitions: workspace %zu < %zu bytes", workspace_bytes, need);
    Workspace ws = carve(workspace, B, nclouds, N, R);
    hipStream_t st = iq::as_stream(stream);
    int rc;
    iq::ProfileSpan call_span(iq::kSlotCall, st);

    hipLaunchKernelGGL(pn_prepare_kernel, dim3(nclouds), dim3(kThreads), 0, st, region_id, ws.sorted_pts, ws.roff, N, R);
    if ((rc = iq::check_launch("pn_prepare_kernel"))) return rc;
    const int pre_items = nclouds * (R + with_centre);
    hipLaunchKernelGGL(pn_rows_kernel, dim3(pre_items), dim3(64), 0, st, ws.sorted_pts, ws.roff, nullptr, nullptr,
                       ws.rows_pre, ws.nrows_pre, N, R, nclouds, with_centre, 1);
    hipLaunchKernelGGL(pn_rows_kernel, dim3(B), dim3(64), 0, st, ws.sorted_pts, ws.roff, keep, cloud_of, ws.rows,
                       ws.nrows, N, R, nclouds, with_centre, 0);
    if ((rc = iq::check_launch("pn_rows_kernel"))) return rc;
    // launch order of the coalition chains: most rows first, so the grid drains evenly
    const bool lpt = iq::tuning(iq::kTuneNoLpt) == 0;
    if (lpt) {
        if (hipMemsetAsync(ws.hist, 0, kBins * sizeof(int32_t), st) != hipSuccess)
            return iq::fail(IQ_ELAUNCH, "iq_pointnet_coalitions: memset failed");
        hipLaunchKernelGGL(pn_order_count_kernel, dim3((B + kThreads - 1) / kThreads), dim3(kThreads), 0, st, ws.nrows,
                           ws.bin_of, ws.hist, B);
        hipLaunchKernelGGL(pn_order_scan_kernel, dim3(1), dim3(64), 0, st, ws.hist);
        hipLaunchKernelGGL(pn_order_scatter_kernel, dim3((B + kThreads - 1) / kThreads), dim3(kThreads), 0, st,
                           ws.bin_of, ws.hist, ws.order, B);
        if ((rc = iq::check_launch("pn_order kernels"))) return rc;
    }

    ChainArgs a{};
    a.clouds = clouds;
    if (channel_first) { a.ps = 1; a.cs = N; } else { a.ps = 3; a.cs = 1; }
    a.cl = 3 * N;
    a.centers = centers;
    a.rows = ws.rows_pre;
    a.nrows = ws.nrows_pre;
    a.item_order = nullptr;
    a.N = N; a.R = R; a.nclouds = nclouds; a.with_centre = with_centre;
    a.stamps = g_stamps;
    a.tail16 = iq::tuning(iq::kTuneExperiment) != 16 && iq::tuning(iq::kTuneExperiment) != 55;   // 16, 55: 32-row tiles only
    a.l3_single = iq::tuning(iq::kTuneExperiment) == 58;


    // 1. input-STN chain, pre-pooled per (cloud, region) [+ centre]
    a.cloud_of = nullptr; a.trans = nullptr;
    a.w_in = w->stn_in;
    a.w1 = nullptr; a.b1 = nullptr;
    a.w2 = w->stn_c2.w; a.b2 = w->stn_c2.b;
    a.w3 = w->stn_c3.w; a.b3 = w->stn_c3.b;
    a.out = ws.G;
    a.items = pre_items;
    {
        iq::ProfileSpan span(iq::kSlotPrepool, st);
        launch_chain<kPrepool>(a, st);
    }
    if ((rc = iq::check_launch("pn_chain_kernel<prepool>"))) return rc;

    hipLaunchKernelGGL(pn_stn_gather_kernel, dim3(B), dim3(kThreads), 0, st, ws.G, ws.nrows, keep, cloud_of, ws.gbuf,
                       R, nclouds, with_centre);
    if ((rc = iq::check_launch("pn_stn_gather_kernel"))) return rc;
    if ((rc = launch_linear(ws.gbuf, kFeat, w->stn_fc1, ws.h1, 512, B, 1, st))) return rc;
    if ((rc = launch_linear(ws.h1, 512, w->stn_fc2, ws.h2, 256, B, 1, st))) return rc;
    if ((rc = launch_linear(ws.h2, 256, w->stn_fc3, ws.trans, 9, B, 0, st))) return rc;

    // 2. feature-STN chain over each coalition's distinct points
    a.cloud_of = cloud_of; a.trans = ws.trans;
    a.rows = ws.rows; a.nrows = ws.nrows;
    a.item_order = lpt ? ws.order : nullptr;
    a.w_in = w->feat_in;
    a.items = B;
    a.out = ws.gbuf;
    float* tfp = trans_feat_packed ? trans_feat_packed : ws.tfp;
    if (w->fstn_c1.w) {
        a.w1 = w->fstn_c1.w; a.b1 = w->fstn_c1.b;
        a.w2 = w->fstn_c2.w; a.b2 = w->fstn_c2.b;
        a.w3 = w->fstn_c3.w; a.b3 = w->fstn_c3.b; a.w3_bf3 = reinterpret_cast<const unsigned short*>(w->fstn_c3_bf3);
        a.w2_bf3 = reinterpret_cast<const unsigned short*>(w->fstn_c2_bf3);
        {
            iq::ProfileSpan span(iq::kSlotFstn, st);
            launch_chain<kFstn>(a, st);
        }
        if ((rc = iq::check_launch("pn_chain_kernel<fstn>"))) return rc;
        if ((rc = launch_linear(ws.gbuf, kFeat, w->fstn_fc1, ws.h1, 512, B, 1, st))) return rc;
        if ((rc = launch_linear(ws.h1, 512, w->fstn_fc2, ws.h2, 256, B, 1, st))) return rc;
        if ((rc = launch_linear(ws.h2, 256, w->fstn_fc3, tfp, 4096, B, 0, st))) return rc;
    } else {
        // feature_transform = False (models/pointnet.py:62-63,72-78): no feature STN.  The trunk multiplies by the packed
        // IDENTITY instead (fstn_fc3.b = iq_pack_fstn_fc3 of a zero layer): sum_k f[k] I[k][n] = f[n] + zeros, exact.
        IQ_REQUIRE(w->fstn_fc3.b, "iq_pointnet_coalitions: without a feature STN, fstn_fc3.b must hold the packed identity");
        hipLaunchKernelGGL(pn_fill_rows_kernel, dim3((unsigned)(((size_t)B * 1024 + kThreads - 1) / kThreads)), dim3(kThreads), 0, st,
                           tfp, w->fstn_fc3.b, B);
        if ((rc = iq::check_launch("pn_fill_rows_kernel"))) return rc;
    }

    // 3. trunk chain
    a.argrow = crt_points;
    a.w1 = tfp; a.b1 = nullptr;
    a.w2 = w->feat_c2.w; a.b2 = w->feat_c2.b;
    a.w3 = w->feat_c3.w; a.b3 = w->feat_c3.b; a.w3_bf3 = reinterpret_cast<const unsigned short*>(w->feat_c3_bf3);
    a.w2_bf3 = reinterpret_cast<const unsigned short*>(w->feat_c2_bf3);
    {
        iq::ProfileSpan span(iq::kSlotTrunk, st);
        launch_chain<kTrunk>(a, st);
    }
    if ((rc = iq::check_launch("pn_chain_kernel<trunk>"))) return rc;
    if ((rc = launch_linear(ws.gbuf, kFeat, w->cls_fc1, ws.h1, 512, B, 1, st))) return rc;
    if ((rc = launch_linear(ws.h1, 512, w->cls_fc2, ws.h2, 256, B, 1, st))) return rc;
    if ((rc = launch_linear(ws.h2, 256, w->cls_fc3, logits, w->cls_fc3.cout, B, 0, st))) return rc;
    return IQ_OK;
}

// ---- host-side packing of the feature-STN output layer (generic packing: iq_linear.hip) ------------
extern "C" int iq_pack_fstn_fc3(const float* w, const float* b, float* out_w, float* out_b, int32_t* perm) {
    IQ_REQUIRE(w && b && out_w && out_b, "iq_pack_fstn_fc3: null pointer");
    // Output element e of the layer must be element e of the packed B image of trans_feat for the
    // product f1 @ trans_feat (models/pointnet.py:74-76): B[k][n] = trans_feat[k][n], i.e. source
    // row k*64 + n of fc3, at e = ((nt*8 + kb)*64 + lane)*4 + j.
    float* tmp = new float[4096 * 256];
    for (int nt = 0; nt < 2; ++nt)
        for (int kb = 0; kb < 8; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    const int n = nt * 32 + (lane & 31), k = 8 * kb + 4 * (lane >> 5) + j;
                    const int e = ((nt * 8 + kb) * 64 + lane) * 4 + j;
                    const int src = k * 64 + n;
                    for (int c = 0; c < 256; ++c) tmp[(size_t)e * 256 + c] = w[(size_t)src * 256 + c];
                    out_b[e] = b[src] + (k == n ? 1.f : 0.f);
                    if (perm) perm[e] = src;
                }
    const int rc = iq_pack_weight(tmp, out_w, 4096, 256);
    delete[] tmp;
    return rc;
}
