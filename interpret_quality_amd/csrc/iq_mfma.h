// fp32 MFMA building blocks shared by the model kernels (gfx950, v_mfma_f32_32x32x2_f32).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/iq.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A fragment for K-block kb (8 k values): lane (row = l&31, h = l>>5) holds k = 8kb + 4h + j in
// element j; MFMA step j consumes element j of A and B.  abase = act + (lane&31)*LD + 4*(lane>>5)
// for a row-major activation image whose row stride LD is padded by 4 floats (conflict-free
// ds_read_b128: a 16-lane group reads 16 rows at bank offsets 4*row mod 64).
template <int LD>
__device__ __forceinline__ f32x4 lds_frag(const float* abase, int mt, int kb) {
    return *reinterpret_cast<const f32x4*>(abase + mt * 32 * LD + 8 * kb);
}

// B fragments come from the packed weight image (iq_pack_weight): fragment (nt, kb) = 1 KiB at ((nt*KB + kb)*64 + lane)*4
// floats.  They are fetched with RAW BUFFER LOADS: an SGPR resource on the image, ONE loop-invariant VGPR offset (lane * 16 B)
// and a scalar byte offset per fragment.  With flat `global_load_dwordx4` on per-lane 64-bit pointers every fragment cost a
// VMEM issue with two address VGPRs plus VALU pointer arithmetic between the MFMAs: measured 8 % of the chain kernel's L3
// loop (24.2 -> 22.6 ms, against 22.2 ms with the loads removed altogether; profiles/r02_chain_l3_ab.txt).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct WBuf {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;   // lane * 16 bytes
};

// `base` must be wave-uniform.  Raw buffer (stride 0): offsets are plain byte offsets, reads past num_records return 0.
__device__ __forceinline__ WBuf wbuf_make(const float* base, int lane) {
    WBuf w;
    w.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
    w.voff = lane * 16;
    return w;
}

// fragment at scalar byte offset `soff` (wave-uniform) of the image
__device__ __forceinline__ f32x4 wbuf_load(const WBuf& w, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w.rsrc, w.voff, soff, 0));
}

// wave-uniform copy of a value the compiler cannot prove uniform (e.g. threadIdx.x >> 6)
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

constexpr int kFragBytes = 1024;   // one (n-tile, k-block) fragment

__device__ __forceinline__ f32x16 mfma4(f32x4 a, f32x4 b, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], c, 0, 0, 0);
    return c;
}

// Column maximum of a tile's 16 accumulator registers.  Written as chains of fmaxf(fmaxf(a, b), c) so that the compiler forms
// v_max3_f32 (8 instructions for 16 values).  iq_pointnet.hip is compiled with -fno-honor-nans (build.py): hipcc otherwise puts a
// canonicalising `v_max_f32 x, x, x` in front of every fmaxf operand it cannot prove quiet - an MFMA result - and the chain kernel
// carried 144 of them, a third of layer 3's pooling instructions (round 5: +1 % on the headline).  (Hand-written v_max3 through
// inline asm is NOT an option: the hazard recogniser does not see an asm's reads of a just-written accumulator - the one-tile and
// two-tile passes of layer 3 stopped agreeing bit for bit.)
__device__ __forceinline__ float max16(f32x16 c) {
    float a = fmaxf(fmaxf(c[0], c[1]), c[2]), b = fmaxf(fmaxf(c[8], c[9]), c[10]);
    a = fmaxf(fmaxf(a, c[3]), c[4]);  b = fmaxf(fmaxf(b, c[11]), c[12]);
    a = fmaxf(fmaxf(a, c[5]), c[6]);  b = fmaxf(fmaxf(b, c[13]), c[14]);
    return fmaxf(fmaxf(a, c[7]), fmaxf(b, c[15]));
}

// C/D layout of the 32x32 MFMA: col = lane & 31, row = (i&3) + 8*(i>>2) + 4*(lane>>5).
__device__ __forceinline__ constexpr int c_row_i(int i) { return (i & 3) + 8 * (i >> 2); }
__device__ __forceinline__ int c_row(int i, int lane) { return c_row_i(i) + 4 * (lane >> 5); }

// ---- software-pipelined n-tile: B fragments in a 4-deep register ring, A fragments one K-block ahead ----
// The ring runs 4 K-blocks ahead of the MFMAs and rolls into the next tile's weights (wnext), so only
// the very first tile of a sequence pays a load latency (prime it before the barrier that precedes
// its use).  KB must be a multiple of 4.  sched_group_barrier pins the issue order inside a K-block.
struct WRing {
    f32x4 r[4];
};

// scur / snext: scalar byte offsets of the current / next n-tile's first fragment in the weight image
__device__ __forceinline__ void wring_prime(WRing& w, const WBuf& wb, int scur) {
#pragma unroll
    for (int i = 0; i < 4; ++i) w.r[i] = wbuf_load(wb, scur + i * kFragBytes);
}

template <int LD, int KB, int MTS>
__device__ __forceinline__ void mfma_ntile(const float* abase, const WBuf& wb, int scur, int snext,
                                           WRing& ring, f32x16& acc0, f32x16& acc1) {
    static_assert(KB % 4 == 0, "KB must be a multiple of 4");
    f32x4 a0n = lds_frag<LD>(abase, 0, 0), a1n = a0n;
    if (MTS == 2) a1n = lds_frag<LD>(abase, 1, 0);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const f32x4 a0 = a0n, a1 = a1n;
        if (kb + 1 < KB) {
            a0n = lds_frag<LD>(abase, 0, kb + 1);
            if (MTS == 2) a1n = lds_frag<LD>(abase, 1, kb + 1);
        }
        const f32x4 bk = ring.r[kb & 3];
        acc0 = mfma4(a0, bk, acc0);
        if (MTS == 2) acc1 = mfma4(a1, bk, acc1);
        ring.r[kb & 3] = wbuf_load(wb, kb + 4 < KB ? scur + (kb + 4) * kFragBytes : snext + (kb + 4 - KB) * kFragBytes);
        if (kb + 1 < KB) __builtin_amdgcn_sched_group_barrier(0x100, MTS, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * MTS, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

namespace iq {
// Batched dense layer out = act(A W^T + b) on packed weights (iq_linear.hip).
// relu: 0 = none, 1 = ReLU, 2 = LeakyReLU(0.2).  m_dev (optional, device): the live row count when it is only
// known on the device (ragged batches); M is then the upper bound the grid is sized for.
// tile_nu / rows_per_cloud (optional): rows are rows_per_cloud consecutive rows per cloud, of which only the first
// tile_nu[cloud] are wanted - 128-row tiles that lie entirely beyond are skipped (their outputs are left untouched).
int launch_linear(const float* A, int lda, const iq_dense_layer& L, float* out, int ldo, int M, int relu,
                  hipStream_t st, const int32_t* m_dev = nullptr, const int32_t* tile_nu = nullptr, int rows_per_cloud = 0);
// Same layer for few rows and a very long K: K is split over workgroups in slices of 512 (independent of M), partial
// sums pass through `scratch` (cin/512 x M x cout floats) and are added in a fixed order.
int launch_linear_splitk(const float* A, int lda, const iq_dense_layer& L, float* out, int ldo, int M, int relu, float* scratch,
                         size_t scratch_floats, hipStream_t st);
// Dense layer fused with the first stage of a pooling layer: partial (ceil(M/32), 2, cout) receives, per 32-row tile,
// the column maxima and the row_w-weighted column sums over the rows with row_w > 0 (the activations are never stored).
// cin % 32 == 0, cout % 32 == 0, cout >= 256.
// w_bf3 (optional): the layer's weights split into three bf16 terms (iq_pack_weight_bf3): the products then run on the bf16
// matrix pipe, float32-exact (pn_gemm_bf3_kernel<pool>).
int launch_linear_pool(const float* A, int lda, const iq_dense_layer& L, float* partial, int M, int relu,
                       const float* row_w, hipStream_t st, const int32_t* m_dev = nullptr, const void* w_bf3 = nullptr);
// Farthest point sampling (iq_geom.hip); n_unique may be null.
int launch_fps(const float* xyz, int32_t* idx, int32_t* n_unique, int B, int N, int S, hipStream_t st);
}  // namespace iq
