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

}  // namespace iq

#define IQ_REQUIRE(cond, ...) \
    do {                      \
        if (!(cond)) return iq::fail(IQ_EINVAL, __VA_ARGS__); \
    } while (0)
