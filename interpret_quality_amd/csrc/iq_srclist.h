// Sorted neighbour lists of a SOURCE cloud (its N points and the centre), shared by the model kernels that replace a per-coalition
// K-nearest search in xyz space by a walk over such a list (iq_dgcnn.hip: layer-1 graph; iq_pointconv.hip: sa1 / sa2 groups).
// In xyz space neither the distance between two points nor the centre depends on the coalition - only the candidate set does.
#pragma once
#include "iq_common.h"
#include "iq_mfma.h"

namespace {

constexpr int kWalkMaxN = 1024;   // source cloud points (the per-row sort holds 2048 entries)

// rows of source cloud c: its N points, the centre (row N), zero padding with |x|^2 = +inf
__global__ void sl_rows_kernel(const float* __restrict__ clouds, const float* __restrict__ centers, float* __restrict__ xs,
                                   float* __restrict__ xxs, int N, int Nsp) {
    const int c = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Nsp) return;
    const float* src = i < N ? clouds + ((size_t)c * N + i) * 3 : centers + (size_t)c * 3;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    float s = INFINITY;
    if (i <= N) {
        a = (f32x4){src[0], src[1], src[2], 0.f};
        s = 0.f;                      // rownorm_kernel's order: ((0 + x^2) + y^2) + z^2
        s += a[0] * a[0];
        s += a[1] * a[1];
        s += a[2] * a[2];
    }
    float* o = xs + ((size_t)c * Nsp + i) * 8;
    *reinterpret_cast<f32x4*>(o) = a;
    *reinterpret_cast<f32x4*>(o + 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
    xxs[(size_t)c * Nsp + i] = s;
}

// dmat[c][q][k] = the consumer's distance value (larger = nearer) of query row q and key row k, by its own expression on the
// same MFMA inner products (one wave = 32 queries, all key tiles).  FORM 0: knn_kernel<8> of iq_dgcnn.hip, (2 q.k - |k|^2) - |q|^2;
// FORM 1: pc_knn_kernel of iq_pointconv.hip, -(((-2 q.k) + |q|^2) + |k|^2).
template <int FORM>
__global__ __launch_bounds__(64) void sl_dist_kernel(const float* __restrict__ xs, const float* __restrict__ xxs,
                                                         float* __restrict__ dmat, int Nsp) {
    const int c = blockIdx.y, q0 = blockIdx.x * 32, lane = threadIdx.x;
    const int fl = lane & 31, fh = lane >> 5;
    const float* xb = xs + (size_t)c * Nsp * 8;
    const float* xxb = xxs + (size_t)c * Nsp;
    const f32x4 qf = *reinterpret_cast<const f32x4*>(xb + (size_t)(q0 + fl) * 8 + 4 * fh);
    const float xxq = xxb[q0 + fl];
    float* drow = dmat + ((size_t)c * Nsp + q0 + fl) * Nsp;
    for (int t = 0; t < Nsp / 32; ++t) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(xb + (size_t)(t * 32 + fl) * 8 + 4 * fh);
        f32x16 acc = {0};
        acc = mfma4(a, qf, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = t * 32 + c_row(r, lane);
            drow[key] = FORM == 0 ? __builtin_fmaf(2.f, acc[r], -xxb[key]) - xxq      // knn_kernel<8>'s expression
                                  : -(((-2.f * acc[r]) + xxq) + xxb[key]);               // pc_knn_kernel's
        }
    }
}

// sorted[c][q][.] = the rows 0..N of source cloud c, nearest to row q first (larger distance value first; ties: lower index)
__global__ __launch_bounds__(256) void sl_sort_kernel(const float* __restrict__ dmat, int16_t* __restrict__ sorted, int N,
                                                          int Nsp, int Nsl) {
    __shared__ unsigned long long e[2048];
    const int c = blockIdx.y, q = blockIdx.x, t = threadIdx.x;
    const float* drow = dmat + ((size_t)c * Nsp + q) * Nsp;
    for (int j = t; j < 2048; j += 256) {
        unsigned long long key = ~0ull;
        if (j <= N) {
            const unsigned o = __float_as_uint(-drow[j]);                       // ascending in -d = descending in d
            const unsigned u = (o & 0x80000000u) ? ~o : (o | 0x80000000u);      // order-preserving map of a float to an unsigned
            key = ((unsigned long long)u << 16) | (unsigned)j;
        }
        e[j] = key;
    }
    __syncthreads();
    for (int k = 2; k <= 2048; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < 2048; i += 256) {
                const int p = i ^ j;
                if (p > i) {
                    const unsigned long long a = e[i], b = e[p];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { e[i] = b; e[p] = a; }
                }
            }
            __syncthreads();
        }
    int16_t* o = sorted + ((size_t)c * (N + 1) + q) * Nsl;
    for (int j = t; j <= N; j += 256) o[j] = (int16_t)(e[j] & 0xffffu);
}


}  // namespace
