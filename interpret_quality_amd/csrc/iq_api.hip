// Version, error reporting and the optional HIP-event profiler of the C ABI (include/iq.h).
#include <mutex>
#include <vector>

#include "iq_common.h"
#include "iq_profile.h"

extern "C" int iq_version(void) { return 100; }  // 0.1.0

extern "C" const char* iq_last_error(void) { return iq::err_buf(); }

namespace iq {
namespace {
struct Span { hipEvent_t start, stop; int which; };
std::mutex g_mu;
std::vector<Span> g_spans;
bool g_enabled = false;
}  // namespace

bool profile_enabled() { return g_enabled; }

namespace { int g_tune[kTuneCount] = {2, 0, 0, 0, 0, 0, 0, 0}; }  // default: L3 variant 2 (B-fragment ring)
int tuning(int key) { return (key >= 0 && key < kTuneCount) ? g_tune[key] : 0; }
void set_tuning(int key, int value) { if (key >= 0 && key < kTuneCount) g_tune[key] = value; }

ProfileSpan::ProfileSpan(int which, hipStream_t st) : which_(which), st_(st), on_(g_enabled) {
    if (!on_) return;
    (void)hipEventCreate(&start_);
    (void)hipEventCreate(&stop_);
    (void)hipEventRecord(start_, st_);
}

ProfileSpan::~ProfileSpan() {
    if (!on_) return;
    (void)hipEventRecord(stop_, st_);
    std::lock_guard<std::mutex> lk(g_mu);
    g_spans.push_back(Span{start_, stop_, which_});
}
}  // namespace iq

extern "C" int iq_set_tuning(int key, int value) {
    IQ_REQUIRE(key >= 0 && key < iq::kTuneCount, "iq_set_tuning: key %d", key);
    iq::set_tuning(key, value);
    return IQ_OK;
}

extern "C" int iq_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(iq::g_mu);
    iq::g_enabled = on != 0;
    return IQ_OK;
}

extern "C" int iq_profile_read(int which, double* total_ms, int* launches) {
    IQ_REQUIRE(total_ms && launches, "iq_profile_read: null output");
    std::lock_guard<std::mutex> lk(iq::g_mu);
    double tot = 0.0;
    int n = 0;
    std::vector<iq::Span> rest;
    for (const iq::Span& s : iq::g_spans) {
        if (s.which != which) { rest.push_back(s); continue; }
        float ms = 0.f;
        if (hipEventSynchronize(s.stop) != hipSuccess || hipEventElapsedTime(&ms, s.start, s.stop) != hipSuccess)
            return iq::fail(IQ_ELAUNCH, "iq_profile_read: event query failed");
        tot += ms;
        ++n;
        (void)hipEventDestroy(s.start);
        (void)hipEventDestroy(s.stop);
    }
    iq::g_spans.swap(rest);
    *total_ms = tot;
    *launches = n;
    return IQ_OK;
}
