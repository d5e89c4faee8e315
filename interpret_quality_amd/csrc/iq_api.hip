// Version, error reporting and the optional HIP-event profiler of the C ABI (include/iq.h).
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
