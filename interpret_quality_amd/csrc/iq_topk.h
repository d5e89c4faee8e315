// Per-lane running top-K (largest values) for the MFMA distance tiles of the kNN kernels.
#pragma once
#include <hip/hip_runtime.h>

// ------------------------------------------------------------------------------------------------------------------
// Queued top-K for the MFMA distance tiles (queries on lanes, 16 keys per lane and tile).
//
// Selection, not distance arithmetic, bounds kNN: a lane inserts ~K(1 + ln(n/K)) of its n keys, but with 64 queries in
// lock-step nearly every key position has SOME lane inserting, so a wave that inserts "when any lane must" pays for
// the maximum over lanes at every tile.  Two changes remove that:
//  * the list is ONE sorted array of doubles: (double)distance with the key index OR-ed into mantissa bits far below
//    fp32 precision.  Order by value (ties: by index), conversion back to fp32 is exact, and a sorted insert is
//    v[q] = max(min(c, v[q-1]), v[q]) - two full-rate fp64 ops per slot, no index array, no minimum search;
//  * candidates that beat the lane's threshold are appended to a small per-lane LDS queue and inserted in rounds that
//    only run while at least half of the lanes have work (or a queue could overflow), so the rounds per wave
//    approach the busiest lane's total instead of the sum over tiles of the per-tile maximum.
// A candidate may wait while the threshold rises; inserting it late is then a no-op, never an error.
template <int K, int CAP = 16>
struct QueuedTopK {
    static constexpr int kIndexBits = 15;  // key index < 32768
    double v[K];                           // descending
    float thr;                             // (float)v[K-1]
    int cnt;                               // queued candidates of this lane
    double* q;                             // this wave's queue, slot-major: q[slot * 64 + lane]

    __device__ __forceinline__ void init(double* wave_queue) {
#pragma unroll
        for (int i = 0; i < K; ++i) v[i] = -INFINITY;
        thr = -INFINITY;
        cnt = 0;
        q = wave_queue;
    }
    static __device__ __forceinline__ double pack(float d, int idx) {
        return __longlong_as_double(__double_as_longlong((double)d) | (long long)idx);
    }
    // v_min_f64 / v_max_f64 written out: through fmin() / fmax() the compiler quiets every operand it cannot prove
    // canonical first (v_max_f64 x, x, x) - the whole list, since it is carried by a loop - and an insert cost 60 fp64
    // instructions instead of 39 (8.8 cycles each per wave: tools/micro/valu_rate.hip).  No NaN ever enters the list.
    static __device__ __forceinline__ double min_f64(double a, double b) {
        double r;
        asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    static __device__ __forceinline__ double max_f64(double a, double b) {
        double r;
        asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    __device__ __forceinline__ void insert(double c) {
        if (K <= 32) {
#pragma unroll
            for (int i = K - 1; i >= 1; --i) v[i] = max_f64(min_f64(c, v[i - 1]), v[i]);
            v[0] = max_f64(c, v[0]);
        } else {   // K = 64: the list spills into AGPRs, which the asm operands cannot name (10 more spills, 40 % slower)
#pragma unroll
            for (int i = K - 1; i >= 1; --i) v[i] = fmax(fmin(c, v[i - 1]), v[i]);
            v[0] = fmax(c, v[0]);
        }
        thr = (float)v[K - 1];
    }
    // Filter threshold for candidates: the two half-waves of a query (lane, lane ^ 32) each keep the top K of THEIR half of the
    // keys, sorted.  a_i of one list and b_j of the other with i + j + 2 >= K give K elements >= min(a_i, b_j), so the K-th
    // largest of the union - below which no key can reach the final K - is at least that; the best of five such pairs is
    // used (own and partner's K-th, the two medians, the two quartile crossings).  Against each lane's own K-th value this
    // cuts the sorted inserts by a quarter (68 -> 50 per lane at K = 20, n = 544).
    __device__ __forceinline__ float union_threshold() const {
        // a_i of one list and b_j of the other with (i + 1) + (j + 1) >= K, i.e. i + j = K - 2 (0-based)
        constexpr int i1 = K / 4 - 1, j1 = K - 2 - i1, i2 = K / 2 - 1, j2 = K - 2 - i2;
        const float a1 = (float)v[i1], a3 = (float)v[j1], a2 = (float)v[i2], a4 = thr;
        const float b1 = __shfl_xor(a1, 32), b2 = __shfl_xor(a2, 32), b3 = __shfl_xor(a3, 32), b4 = __shfl_xor(a4, 32);
        float mid;
        if (i2 == j2) {
            mid = fminf(a2, b2);
        } else {   // odd K: the two median crossings differ
            const float a2b = (float)v[j2], b2b = __shfl_xor(a2b, 32);
            mid = fmaxf(fminf(a2, b2b), fminf(a2b, b2));
        }
        return fmaxf(fmaxf(a4, b4), fmaxf(mid, fmaxf(fminf(a1, b3), fminf(a3, b1))));
    }
    // A queued candidate is the raw pair {key index, fp32 distance} (8 bytes); it becomes the packed double when it is
    // popped (one conversion per round instead of one per candidate).  push() is branch-free: the pair is always written
    // to the lane's next free slot and the slot is kept only if the candidate beats the threshold (a rejected one is
    // overwritten by the next push) - five instructions per candidate where the branchy form (compare, save exec,
    // branch, convert, two ORs, address, store, count, restore) issued twelve plus two branches, 16 times per key tile.
    // The queue must hold CAP + 1 slots per lane (the slot behind a full queue takes the rejected writes).
    __device__ __forceinline__ void push(float d, int idx, float thr_f, int lane) {
        int2* slot = reinterpret_cast<int2*>(q) + cnt * 64 + lane;
        *slot = make_int2(idx, __float_as_int(d));
        cnt += d > thr_f ? 1 : 0;
    }
    __device__ __forceinline__ void round(int lane) {
        double c = -INFINITY;
        if (cnt > 0) {
            const int2 e = reinterpret_cast<const int2*>(q)[(--cnt) * 64 + lane];
            c = pack(__int_as_float(e.y), e.x);
        }
        insert(c);
    }
    // make room for `need` more candidates per lane, and use rounds that are well filled anyway
    __device__ __forceinline__ void drain(int lane, int need) {
        for (;;) {
            const unsigned long long busy = __ballot(cnt > 0);
            if (busy == 0) break;
            if (__popcll(busy) < 32 && !__any(cnt > CAP - need)) break;
            round(lane);
        }
    }
    __device__ __forceinline__ void flush(int lane) {
        while (__any(cnt > 0)) round(lane);
    }
    // the 16 distances of one accumulator tile (rows c_row(r, lane) of the key tile starting at idx_base)
    __device__ __forceinline__ void offer_tile(const float (&d)[16], int idx_base, int lane) {
        const int fh = lane >> 5;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            drain(lane, 8);
            const float thr_f = union_threshold();
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int r = half * 8 + rr;
                push(d[r], idx_base + (r & 3) + 8 * (r >> 2) + 4 * fh, thr_f, lane);
            }
        }
    }
    // union with the partner half-wave (lane ^ 32), which saw the other half of every key tile: the K largest of two
    // descending lists are max(a[i], b[K-1-i]) (first step of a bitonic merge; the result is an unordered set)
    __device__ __forceinline__ void merge_halves() {
        auto partner = [](double x) {
            const long long pb = __double_as_longlong(x);
            const int lo = __shfl_xor((int)(pb & 0xffffffffll), 32), hi = __shfl_xor((int)(pb >> 32), 32);
            return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
        };
#pragma unroll
        for (int i = 0; i < K / 2; ++i) {  // positions i and K-1-i only depend on each other: in place
            const double bi = partner(v[i]), bj = partner(v[K - 1 - i]);
            v[i] = fmax(v[i], bj);
            v[K - 1 - i] = fmax(v[K - 1 - i], bi);
        }
        if (K & 1) v[K / 2] = fmax(v[K / 2], partner(v[K / 2]));
    }
    __device__ __forceinline__ int index(int i) const { return (int)(__double_as_longlong(v[i]) & ((1ll << kIndexBits) - 1)); }
};

// ------------------------------------------------------------------------------------------------------------------
// Tagged top-K [r3] - the selection of the kNN kernels whose near-ties are re-ranked exactly afterwards (knn_refine_kernel).
//
// QueuedTopK pays 45 fp64 min / max (8.8 cycles each) per sorted insert because a list entry must carry its key index.  Here an
// entry is ONE 32-bit integer: the float32 distance as an order-preserving integer key (negative floats with their magnitude
// bits flipped) whose low 5 bits are replaced by a TAG - the slot of the entry's key index in a small per-lane LDS table.
// A sorted insert is v[i] = med3(c, v[i-1], v[i]) on integers: 21 full-rate instructions.  K + 1 tags circulate: the K list
// entries hold K of them, the next candidate takes the free one, and the entry that drops out of the list - min(candidate,
// old last entry) - hands its tag back.  The price is resolution: keys are compared in buckets of 32 float32 ulps (3.8e-6
// relative), candidates in the bucket of the threshold are rejected like exact ties, and two entries of one bucket rank by
// tag.  That is harmless exactly where knn_refine_kernel stands behind the selection: a boundary gap that small is far inside
// the band (>= 2e-5 of the value) that is re-ranked in exact arithmetic anyway.  The exact-only kernel (C = 8) keeps QueuedTopK.
template <int K, int CAP = 16>
struct TaggedTopK {
    static constexpr int kTagBits = 5, kTagMask = (1 << kTagBits) - 1;
    static_assert(K + 1 <= (1 << kTagBits), "K + 1 tags must fit the tag bits");
    static constexpr int kKeyNegInf = (int)0x807fffffu;   // key of -inf: every real distance is above it, every empty slot below
    static constexpr int kLdsBytes = (CAP + 1) * 64 * 8 + (K + 1) * 64 * 2;
    int v[K];             // descending keys; low 5 bits = tag
    int cnt;              // queued candidates of this lane
    int free_tag;
    int2* q;              // this wave's queue, slot-major: q[slot * 64 + lane] = {key index, float bits}
    unsigned short* tab;  // tab[tag * 64 + lane] = key index of the entry that carries `tag`

    static __device__ __forceinline__ int key_of(float d) {   // order-preserving, its own inverse on the bit pattern
        const int b = __float_as_int(d);
        return b ^ ((b >> 31) & 0x7fffffff);
    }
    static __device__ __forceinline__ float float_of(int key) { return __int_as_float(key ^ ((key >> 31) & 0x7fffffff)); }
    static __device__ __forceinline__ int med3(int a, int b, int c) {
        int r;
        asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
        return r;
    }
    __device__ __forceinline__ void init(void* wave_lds) {
#pragma unroll
        for (int i = 0; i < K; ++i) v[i] = (int)0x80000000u + (K - 1 - i);   // empty slots: below every key, tags K-1 .. 0
        cnt = 0;
        free_tag = K;
        q = reinterpret_cast<int2*>(wave_lds);
        tab = reinterpret_cast<unsigned short*>(q + (CAP + 1) * 64);
    }
    // float32 threshold of a key: the upper end of its bucket (a candidate inside the bucket is rejected, like a tie)
    static __device__ __forceinline__ float threshold_of(int key) { return float_of(max(key | kTagMask, kKeyNegInf)); }
    __device__ __forceinline__ float union_threshold() const {   // see QueuedTopK::union_threshold
        constexpr int i1 = K / 4 - 1, j1 = K - 2 - i1, i2 = K / 2 - 1, j2 = K - 2 - i2;
        const int a1 = v[i1], a3 = v[j1], a2 = v[i2], a4 = v[K - 1];
        const int b1 = __shfl_xor(a1, 32), b2 = __shfl_xor(a2, 32), b3 = __shfl_xor(a3, 32), b4 = __shfl_xor(a4, 32);
        int mid;
        if (i2 == j2) {
            mid = min(a2, b2);
        } else {
            const int a2b = v[j2], b2b = __shfl_xor(a2b, 32);
            mid = max(min(a2, b2b), min(a2b, b2));
        }
        return threshold_of(max(max(a4, b4), max(mid, max(min(a1, b3), min(a3, b1)))));
    }
    __device__ __forceinline__ void push(float d, int idx, float thr_f, int lane) {   // as QueuedTopK::push
        q[cnt * 64 + lane] = make_int2(idx, __float_as_int(d));
        cnt += d > thr_f ? 1 : 0;
    }
    __device__ __forceinline__ void round(int lane) {   // branch-free: a lane without a candidate inserts an empty-slot key
        const bool has = cnt > 0;
        cnt -= has ? 1 : 0;
        const int2 e = q[cnt * 64 + lane];
        const int key = has ? ((key_of(__int_as_float(e.y)) & ~kTagMask) | free_tag) : ((int)0x80000000u | free_tag);
        tab[free_tag * 64 + lane] = (unsigned short)e.x;
        const int dropped = min(key, v[K - 1]);
#pragma unroll
        for (int i = K - 1; i >= 1; --i) v[i] = med3(key, v[i - 1], v[i]);
        v[0] = max(key, v[0]);
        free_tag = dropped & kTagMask;
    }
    // the list as QueuedTopK holds it: (double)distance - the lower end of the entry's bucket - with the key index in the low
    // mantissa bits; -inf for an empty slot
    __device__ __forceinline__ void export_packed(double (&out)[K], int lane) const {
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int idx = tab[(v[i] & kTagMask) * 64 + lane];
            const double p = __longlong_as_double(__double_as_longlong((double)float_of(v[i] & ~kTagMask)) | (long long)idx);
            out[i] = v[i] > kKeyNegInf ? p : -INFINITY;
        }
    }
};
