// bf16x3 building blocks (gfx950, v_mfma_f32_32x32x16_bf16): float32 GEMMs on the bf16 matrix pipe, float32-exact.
// A float32 is three bf16 terms, x = h + m + l (round to nearest, residuals exact); a product is the six largest of the nine
// term products, each exact, accumulated in float32 (small terms first).  See DESIGN.md 5a and iq_pack_weight_bf3 (iq_linear.hip)
// for the weight image: fragment (term, n-tile, k-step of 16) = 1 KiB at ((term * NT + n-tile) * KS + k-step) KiB, lane l at l * 16.
#pragma once
#include "iq_mfma.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct B3 { bf16x8 h, m, l; };

// Two float32 -> two bf16 (round to nearest even) in one dword, and back.
__device__ __forceinline__ unsigned bf16_pair(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf16_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// fragment at byte offset `off` of the image's first term; the other two terms lie term_stride bytes apart
__device__ __forceinline__ B3 b3_load_at(const __amdgpu_buffer_rsrc_t& rs, int voff, int off, int term_stride) {
    return B3{__builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, off, 0)),
              __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, off + term_stride, 0)),
              __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, off + 2 * term_stride, 0))};
}

// A fragment of k-step ks from three bf16 planes in LDS: abase = plane 0 + (first row of the m-tile + (lane & 31)) * ROWB +
// 16 * (lane >> 5); rows of ROWB bytes, ROWB / 4 = 4 (mod 8) dwords for conflict-free ds_read_b128
template <int PLANEB>
__device__ __forceinline__ void a3_load(bf16x8 (&a)[3], const unsigned char* abase, int ks) {
#pragma unroll
    for (int e = 0; e < 3; ++e) a[e] = *reinterpret_cast<const bf16x8*>(abase + e * PLANEB + ks * 32);
}

__device__ __forceinline__ f32x16 mfma_bf3(const bf16x8 (&a)[3], const B3& b, f32x16 acc) {   // six products, small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b.l, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b.m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b.m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b.h, acc, 0, 0, 0);
    return acc;
}

// One 32x32 C tile (lane: column lane & 31, rows c_row_i(i) + 4 (lane >> 5)) -> three bf16 planes in LDS.  Two lanes (columns c,
// c + 1) trade one value of each pair of rows (DPP quad_perm [1,0,3,2]) so that every store is a whole dword: 24 ds_write_b32
// instead of 48 ds_write_b16.  `tile`: plane 0, first row of the m-tile, first column of the n-tile; value(i): element i after
// bias / activation.
template <int ROWB, int PLANEB, typename F>
__device__ __forceinline__ void c_tile_to_planes(unsigned char* tile, int lane, F value) {
    const int odd = lane & 1;
    unsigned char* d = tile + (4 * (lane >> 5) + odd) * ROWB + ((lane & 31) & ~1) * 2;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        const float v0 = value(i), v1 = value(i + 1);                                         // rows r, r + 1 of column c
        const float got = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, odd ? v0 : v1), 0xB1, 0xF, 0xF, false));
        const float lo = odd ? got : v0, hi = odd ? v1 : got;                                 // columns c & ~1, c | 1 of row r + odd
        const unsigned h = bf16_pair(lo, hi);
        const float rl = lo - bf16_lo(h), rh = hi - bf16_hi(h);
        const unsigned m = bf16_pair(rl, rh);
        unsigned char* o = d + c_row_i(i) * ROWB;
        *reinterpret_cast<unsigned*>(o) = h;
        *reinterpret_cast<unsigned*>(o + PLANEB) = m;
        *reinterpret_cast<unsigned*>(o + 2 * PLANEB) = bf16_pair(rl - bf16_lo(m), rh - bf16_hi(m));
    }
}

// four consecutive channels of one row -> the three planes (8 bytes each); dst = plane 0 + row * ROWB + channel * 2
template <int PLANEB>
__device__ __forceinline__ void row4_to_planes(unsigned char* dst, f32x4 v) {
    const unsigned h0 = bf16_pair(v[0], v[1]), h1 = bf16_pair(v[2], v[3]);
    const float r0 = v[0] - bf16_lo(h0), r1 = v[1] - bf16_hi(h0), r2 = v[2] - bf16_lo(h1), r3 = v[3] - bf16_hi(h1);
    const unsigned m0 = bf16_pair(r0, r1), m1 = bf16_pair(r2, r3);
    *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(dst + PLANEB) = u32x2{m0, m1};
    *reinterpret_cast<u32x2*>(dst + 2 * PLANEB) =
        u32x2{bf16_pair(r0 - bf16_lo(m0), r1 - bf16_hi(m0)), bf16_pair(r2 - bf16_lo(m1), r3 - bf16_hi(m1))};
}

// four float32 values -> their three bf16 terms, packed two per dword (element 2 p in the low half)
__device__ __forceinline__ void split4(f32x4 v, u32x2& h, u32x2& m, u32x2& l) {
    const unsigned h0 = bf16_pair(v[0], v[1]), h1 = bf16_pair(v[2], v[3]);
    const float r0 = v[0] - bf16_lo(h0), r1 = v[1] - bf16_hi(h0), r2 = v[2] - bf16_lo(h1), r3 = v[3] - bf16_hi(h1);
    const unsigned m0 = bf16_pair(r0, r1), m1 = bf16_pair(r2, r3);
    h = u32x2{h0, h1};
    m = u32x2{m0, m1};
    l = u32x2{bf16_pair(r0 - bf16_lo(m0), r1 - bf16_hi(m0)), bf16_pair(r2 - bf16_lo(m1), r3 - bf16_hi(m1))};
}

// The TRANSPOSED C tile of a layer whose output goes back to LDS as an activation image (round 5): with the weight fragment as the
// A operand and the activation fragment as B - the same two fragments, swapped - lane (row = lane & 31 of the m-tile, half) holds its
// row's channels 8 g + 4 half + 0..3 in registers 4 g .. 4 g + 3: four consecutive channels per register quad, so the three bf16
// planes take whole 8-byte stores and the two-lane DPP trade of c_tile_to_planes (4 VALU per value pair) is not needed.
// `tile`: plane 0, first row of the m-tile, first channel of the n-tile; value(r): register r after bias / activation, where
// register r is channel c_row_i(r) + 4 * (lane >> 5) of the n-tile.
template <int ROWB, int PLANEB, typename F>
__device__ __forceinline__ void ct_tile_to_planes(unsigned char* tile, int lane, F value) {
    unsigned char* d = tile + (lane & 31) * ROWB + 8 * (lane >> 5);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        u32x2 h, m, l;
        split4(f32x4{value(4 * g), value(4 * g + 1), value(4 * g + 2), value(4 * g + 3)}, h, m, l);
        unsigned char* o = d + g * 16;
        *reinterpret_cast<u32x2*>(o) = h;
        *reinterpret_cast<u32x2*>(o + PLANEB) = m;
        *reinterpret_cast<u32x2*>(o + 2 * PLANEB) = l;
    }
}

// one tile, transposed: the six products of mfma_bf3 with the operands swapped (weights = A operand)
__device__ __forceinline__ f32x16 mfma_bf3_tr(const B3& w, const bf16x8 (&x)[3], f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.h, x[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.l, x[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.m, x[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.h, x[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.m, x[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.h, x[0], acc, 0, 0, 0);
    return acc;
}

// transposed six-product k-step over MT m-tiles and ONE n-tile: the weight fragment as the A operand, the activation terms as B;
// term-major over the tiles (a dependent MFMA is MT instructions away), small terms first, the same products in the same order
// as mfma_bf3_block<MT, 1>
template <int TW, int TX, int MT>
__device__ __forceinline__ void mfma_term_block_tr(const B3& w, const bf16x8 (&x)[MT][3], f32x16 (&acc)[MT][1]) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TW == 0 ? w.h : (TW == 1 ? w.m : w.l), x[i][TX], acc[i][0], 0, 0, 0);
}
template <int MT>
__device__ __forceinline__ void mfma_bf3_block_tr(const bf16x8 (&x)[MT][3], const B3& w, f32x16 (&acc)[MT][1]) {
    mfma_term_block_tr<0, 2, MT>(w, x, acc);   // (activation l) x (weight h)
    mfma_term_block_tr<2, 0, MT>(w, x, acc);   // (activation h) x (weight l)
    mfma_term_block_tr<1, 1, MT>(w, x, acc);
    mfma_term_block_tr<0, 1, MT>(w, x, acc);   // (activation m) x (weight h)
    mfma_term_block_tr<1, 0, MT>(w, x, acc);   // (activation h) x (weight m)
    mfma_term_block_tr<0, 0, MT>(w, x, acc);
}

// MT x NT tiles of one k-step: six products per tile, the tiles' accumulation chains interleaved (a dependent MFMA is MT * NT
// instructions away), small terms first
template <int TA, int TB>
__device__ __forceinline__ f32x16 mfma_term(const bf16x8 (&a)[3], const B3& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA], TB == 0 ? b.h : (TB == 1 ? b.m : b.l), c, 0, 0, 0);
}
template <int TA, int TB, int MT, int NT>
__device__ __forceinline__ void mfma_term_block(const bf16x8 (&a)[MT][3], const B3 (&b)[NT], f32x16 (&acc)[MT][NT]) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mfma_term<TA, TB>(a[i], b[j], acc[i][j]);
}
template <int MT, int NT>
__device__ __forceinline__ void mfma_bf3_block(const bf16x8 (&a)[MT][3], const B3 (&b)[NT], f32x16 (&acc)[MT][NT]) {
    mfma_term_block<2, 0, MT, NT>(a, b, acc);
    mfma_term_block<0, 2, MT, NT>(a, b, acc);
    mfma_term_block<1, 1, MT, NT>(a, b, acc);
    mfma_term_block<1, 0, MT, NT>(a, b, acc);
    mfma_term_block<0, 1, MT, NT>(a, b, acc);
    mfma_term_block<0, 0, MT, NT>(a, b, acc);
}
