// Shared host-side helpers for libiq_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/iq.h"

namespace iq {

inline char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(IQ_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return IQ_OK;
}

inline hipStream_t as_stream(iq_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Bit `rid` of a coalition's keep mask; a region id outside [0, 64) (which iq_check_index_range rejects) is never kept,
// so an unvalidated id gives a defined result instead of an undefined shift.
__host__ __device__ inline bool keep_bit(unsigned long long keep, int rid) {
    return (unsigned)rid < 64u && ((keep >> rid) & 1ull);
}

// The cloud a workgroup of an XCD-aware grid works on.  Workgroups go round-robin over the 8 XCDs (blockIdx & 7); all
// workgroups of a cloud sit on one XCD (they share its rows through one L2), clouds c = 8 k + xcd.  A batch whose work has
// period 4 in the cloud index - the four masked clouds S+{i,j}, S+{i}, S+{j}, S of every interaction context, largest first
// (final_point_binary_interaction_logits.py:48-52) - would put every largest cloud on XCDs 0 and 4 and every smallest on 3 and
// 7 (kNN work ~ rows^2: 9.6 % above the mean over the 13 ratios, 2x for the low orders): the low two bits of the cloud index
// are rotated by the group-of-eight number, so each XCD sees all four kinds equally often.  A bijection on every full group of 4.
__device__ inline int xcd_cloud(int block, int wgs_per_cloud, int B) {
    const int c = ((block >> 3) / wgs_per_cloud) * 8 + (block & 7);
    return (c | 3) < B ? (c & ~3) | ((c + (c >> 3)) & 3) : c;
}

}  // namespace iq

#define IQ_REQUIRE(cond, ...) \
    do {                      \
        if (!(cond)) return iq::fail(IQ_EINVAL, __VA_ARGS__); \
    } while (0)
