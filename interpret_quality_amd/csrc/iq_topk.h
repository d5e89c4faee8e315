// Per-lane running top-K (largest values) kept in registers: replace-the-minimum insertion.
#pragma once
#include <hip/hip_runtime.h>

template <int K>
struct TopK {
    float v[K];
    int i[K];
    float minv;
    int minp;
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int q = 0; q < K; ++q) { v[q] = -INFINITY; i[q] = 0; }
        minv = -INFINITY; minp = 0;
    }
    __device__ __forceinline__ void offer(float val, int idx) {
        if (val > minv) {
#pragma unroll
            for (int q = 0; q < K; ++q)
                if (q == minp) { v[q] = val; i[q] = idx; }
            minv = v[0]; minp = 0;
#pragma unroll
            for (int q = 1; q < K; ++q)
                if (v[q] < minv) { minv = v[q]; minp = q; }
        }
    }
    // insert the 16 candidates of one accumulator tile, best first; leaves as soon as no lane has one left
    __device__ __forceinline__ void offer_tile(float (&d)[16], int idx_base, int fh) {
        for (;;) {
            float best = minv;
            int br = -1;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (d[r] > best) { best = d[r]; br = r; }
            if (!__any(br >= 0)) break;
            if (br >= 0) {
                offer(best, idx_base + (br & 3) + 8 * (br >> 2) + 4 * fh);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (r == br) d[r] = -INFINITY;
            }
        }
    }
    // merge with the partner half-wave (lane ^ 32)
    __device__ __forceinline__ void merge_halves() {
        float pv[K];
        int pi[K];
#pragma unroll
        for (int q = 0; q < K; ++q) { pv[q] = __shfl_xor(v[q], 32); pi[q] = __shfl_xor(i[q], 32); }
#pragma unroll
        for (int q = 0; q < K; ++q) offer(pv[q], pi[q]);
    }
};
