// Version, error reporting and the optional HIP-event profiler of the C ABI (include/iq.h).
#include <algorithm>
#include <vector>

#include "../../include/iq_debug.h"
#include "iq_common.h"
#include "iq_profile.h"

extern "C" int iq_version(void) { return IQ_ABI_VERSION; }

extern "C" const char* iq_last_error(void) { return iq::err_buf(); }

// Profiler and experiment knobs are state of the CALLING THREAD (like iq_last_error): a thread that enables the profiler
// or flips a knob affects only the launches it issues itself, so the library stays re-entrant across threads.
namespace iq {
namespace {
struct Span { hipEvent_t start, stop; int which; double work; };
struct ThreadState {
    std::vector<Span> spans;
    bool enabled = false;
    int tune[kTuneCount] = {2, 0, 0, 0, 0, 0, 0, 0};  // default: L3 variant 2 (B-fragment ring)
};
ThreadState& ts() {
    static thread_local ThreadState s;
    return s;
}
}  // namespace

bool profile_enabled() { return ts().enabled; }

int tuning(int key) { return (key >= 0 && key < kTuneCount) ? ts().tune[key] : 0; }
void set_tuning(int key, int value) { if (key >= 0 && key < kTuneCount) ts().tune[key] = value; }

ProfileSpan::ProfileSpan(int which, hipStream_t st, double work) : which_(which), st_(st), on_(ts().enabled), work_(work) {
    if (!on_) return;
    (void)hipEventCreate(&start_);
    (void)hipEventCreate(&stop_);
    (void)hipEventRecord(start_, st_);
}

ProfileSpan::~ProfileSpan() {
    if (!on_) return;
    (void)hipEventRecord(stop_, st_);
    ts().spans.push_back(Span{start_, stop_, which_, work_});
}
}  // namespace iq

extern "C" int iq_set_tuning(int key, int value) {
    IQ_REQUIRE(key >= 0 && key < iq::kTuneCount, "iq_set_tuning: key %d", key);
    iq::set_tuning(key, value);
    return IQ_OK;
}

extern "C" int iq_profile_enable(int on) {
    iq::ts().enabled = on != 0;
    return IQ_OK;
}

extern "C" int iq_profile_read_work(int which, double* total_ms, int* launches, double* total_work) {
    IQ_REQUIRE(total_ms && launches, "iq_profile_read: null output");
    double tot = 0.0, work = 0.0;
    int n = 0;
    std::vector<iq::Span> rest;
    for (const iq::Span& s : iq::ts().spans) {
        if (s.which != which) { rest.push_back(s); continue; }
        float ms = 0.f;
        if (hipEventSynchronize(s.stop) != hipSuccess || hipEventElapsedTime(&ms, s.start, s.stop) != hipSuccess)
            return iq::fail(IQ_ELAUNCH, "iq_profile_read: event query failed");
        tot += ms;
        work += s.work;
        ++n;
        (void)hipEventDestroy(s.start);
        (void)hipEventDestroy(s.stop);
    }
    iq::ts().spans.swap(rest);
    *total_ms = tot;
    *launches = n;
    if (total_work) *total_work = work;
    return IQ_OK;
}

extern "C" int iq_profile_read(int which, double* total_ms, int* launches) {
    return iq_profile_read_work(which, total_ms, launches, nullptr);
}

// ---- diagnostic: what the bf16 matrix pipe SUSTAINS on this board, now ------------------------------------------------------
// A register-only loop of v_mfma_f32_32x32x16_bf16 (four independent accumulators per wave, operands with random mantissas and
// signs so that the datapath toggles), one wave per SIMD on every CU, for about `seconds`.  MFMA-dense kernels on MI355X are
// bounded by the power cap, not by issue slots (MI355X_MICROARCH.md, DVFS give-back): the clock the governor holds differs from
// board to board, so bench.py measures this ceiling on the board it runs on instead of quoting a constant.
namespace {
typedef float dbg_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 dbg_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned dbg_u32x4 __attribute__((ext_vector_type(4)));
__device__ unsigned long long g_sustained_clk[2];   // shader-clock ticks (s_memtime) and 100 MHz ticks (s_memrealtime) of block 0's loop

__device__ inline dbg_bf16x8 dbg_operand(unsigned seed) {   // eight bf16 in [1, 2) with random mantissas, random signs
    dbg_u32x4 v;
    for (int i = 0; i < 4; ++i) {
        seed = seed * 1664525u + 1013904223u;
        const unsigned lo = 0x3f80u | ((seed >> 9) & 0x7fu) | ((seed >> 3) & 0x8000u);
        seed = seed * 1664525u + 1013904223u;
        const unsigned hi = 0x3f80u | ((seed >> 9) & 0x7fu) | ((seed >> 3) & 0x8000u);
        v[i] = lo | (hi << 16);
    }
    return __builtin_bit_cast(dbg_bf16x8, v);
}

__global__ __launch_bounds__(256) void mfma_sustained_kernel(float* out, int iters, int seed0) {
    const dbg_bf16x8 a0 = dbg_operand(seed0 + threadIdx.x * 7 + blockIdx.x), a1 = dbg_operand(seed0 * 3 + threadIdx.x * 11 + blockIdx.x);
    const dbg_bf16x8 b0 = dbg_operand(seed0 * 5 + threadIdx.x * 13), b1 = dbg_operand(seed0 * 9 + threadIdx.x * 17);
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    dbg_f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c3, 0, 0, 0);
        }
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned long long t1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        g_sustained_clk[0] = t1 - t0;
        g_sustained_clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[5] + c2[9] + c3[15];
}
}  // namespace

extern "C" int iq_debug_mfma_sustained(double seconds, float* scratch, size_t scratch_floats, double* tflops, double* clock_ghz,
                                       iq_stream_t stream) {
    IQ_REQUIRE(scratch && tflops && seconds > 0.0 && seconds <= 10.0, "iq_debug_mfma_sustained: seconds in (0, 10], scratch required");
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return iq::fail(IQ_ELAUNCH, "iq_debug_mfma_sustained: device query failed");
    const int grid = cus;                                        // 4 waves per workgroup, one workgroup per CU: one wave per SIMD, every
                                                                 // workgroup resident from the first cycle to the last (r4: two waves per
                                                                 // SIMD sustain the same rate, profiles/r04_power_probe.txt)
    IQ_REQUIRE(scratch_floats >= (size_t)grid * 256, "iq_debug_mfma_sustained: scratch holds %zu floats, %zu needed", scratch_floats,
               (size_t)grid * 256);
    hipStream_t st = iq::as_stream(stream);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return iq::fail(IQ_ELAUNCH, "hipEventCreate failed");
    int iters = 20000;
    float ms = 0.f;
    int rc = IQ_OK;
    for (int pass = 0; pass < 2 && rc == IQ_OK; ++pass) {        // pass 0 calibrates the iteration count for ~`seconds`
        (void)hipEventRecord(e0, st);
        hipLaunchKernelGGL(mfma_sustained_kernel, dim3(grid), dim3(256), 0, st, scratch, iters, 12345 + pass);
        (void)hipEventRecord(e1, st);
        if ((rc = iq::check_launch("mfma_sustained_kernel"))) break;
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = iq::fail(IQ_ELAUNCH, "event timing failed");
        if (pass == 0) iters = (int)std::min(2.0e9, std::max(1000.0, iters * seconds * 1e3 / std::max(ms, 1e-3f)));
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != IQ_OK) return rc;
    unsigned long long clk[2] = {0, 0};
    if (hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_sustained_clk), sizeof(clk)) != hipSuccess) return iq::fail(IQ_ELAUNCH, "hipMemcpyFromSymbol failed");
    const double mfma = (double)grid * 4.0 * (double)iters * 16.0;
    *tflops = mfma * (2.0 * 32 * 32 * 16) / (ms * 1e-3) / 1e12;
    // in-kernel clock: block 0's s_memtime ticks over the launch's wall time (block 0 is resident for the whole launch; this is the
    // form round 4 validated against the counters).  The s_memtime / s_memrealtime ratio read 9.7 where the PMC clock was 1.94 GHz
    // (two waves per SIMD, round 5), so it is NOT used.
    (void)clk[1];
    if (clock_ghz) *clock_ghz = (double)clk[0] / (ms * 1e-3) / 1e9;
    return IQ_OK;
}
