// RAII HIP-event bracket around a kernel launch, recorded on the launch stream (bench.py's
// roofline leg reads the totals through iq_profile_read).  Costs nothing when disabled.
#pragma once
#include <hip/hip_runtime.h>

namespace iq {

// kSlotDominant: the ONE kernel that dominates a model's step (bench.py's per-model roofline), with the FLOP its MFMA tiles
// execute attached as `work`
enum ProfileSlot { kSlotPrepool = 0, kSlotFstn = 1, kSlotTrunk = 2, kSlotCall = 3, kSlotMask = 4, kSlotDominant = 5 };

bool profile_enabled();

// Experiment knobs (iq_set_tuning): A/B kernel variants inside one process (guide rule 24).
enum TuneKey { kTuneL3Variant = 0, kTuneExtraLds = 1, kTuneNoLpt = 2, kTuneNoLdsGemm = 3, kTuneKnnDebug = 4, kTuneExperiment = 5, kTuneGroupBlocks = 6, kTuneNoTranspose = 7, kTuneCount = 8 };
int tuning(int key);

class ProfileSpan {
  public:
    ProfileSpan(int which, hipStream_t st, double work = 0.0);
    ~ProfileSpan();
    ProfileSpan(const ProfileSpan&) = delete;
    ProfileSpan& operator=(const ProfileSpan&) = delete;

  private:
    int which_;
    hipStream_t st_;
    bool on_;
    double work_;
    hipEvent_t start_{}, stop_{};
};

}  // namespace iq
