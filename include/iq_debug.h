/*
 * iq_debug.h - diagnostic entry points of libiq_hip.so: the HIP-event profiler bench.py's roofline leg reads, the experiment
 * knobs the A/B tools under tools/ flip, and debug counters.  They are exported by the same library but are NOT part of the
 * drop-in surface (include/iq.h): nothing in the reference's interface corresponds to them and no product path calls them.
 *
 * State: everything here is state of the CALLING THREAD (thread_local in iq_api.hip, like iq_last_error): a thread that enables
 * the profiler or flips a knob affects only the launches it issues itself; other threads and other processes are untouched.
 * WITHIN that thread a knob applies to every engine: iq_set_tuning(5, v) in particular switches the arithmetic path (bf16x3 <->
 * float32 MFMA twins, fused <-> two-kernel forms) of all later launches of the thread until it is set back - tests and tools
 * restore it in try/finally.
 */
#ifndef IQ_DEBUG_H_
#define IQ_DEBUG_H_

#include "iq.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Optional HIP-event profiler (bench.py's roofline leg).  While enabled, iq_pointnet_coalitions
 * brackets its chain-kernel launches with hipEvents recorded on the launch stream.
 * iq_profile_read(slot) synchronises on the recorded events of that slot, returns their summed
 * duration and count, and forgets them.  Slots: 0 = input-STN pre-pool chain, 1 = feature-STN
 * chain, 2 = trunk chain, 3 = whole iq_pointnet_coalitions call. */
int iq_profile_enable(int on);
/* As iq_profile_read, plus the summed `work` (executed FLOP of the MFMA tiles issued) the library attached to the spans
 * of that slot.  Slot 5 = the dominant kernel of a model's step: PointNet++ pn2_group_kernel<128,128,256> (sa2, third
 * scale), DGCNN / GCNN conv5 + pooling GEMM, PointConv pc_group_kernel<128,128,256> (sa2).  Profiler state and the
 * iq_set_tuning knobs belong to the calling thread (like iq_last_error). */
int iq_profile_read_work(int slot, double* total_ms, int* launches, double* total_work);
/* Experiment knob: selects between co-compiled kernel variants so that they can be timed
 * interleaved in ONE process.  key 0 = L3 weight-streaming variant of the chain kernel; 1 = extra dynamic LDS
 * of the chain kernel (occupancy experiment); 2 = 1: no LPT launch order; 3 = 1: dense layers never use the
 * LDS-staged GEMM (pn_gemm_lds_kernel), only the register-streaming one; 4 = kNN diagnostics (results are NOT valid
 * except for 3): 1 queue appends without insertion rounds, 2 no selection at all (MFMA + load skeleton), 3 normal
 * selection + round / busy-lane counters returned in the first 24 bytes of iq_knn's tmp (tools/knn_probe.py);
 * 6 = 16-row member blocks per workgroup of the PointNet++ grouped kernel (0 = default 12); 7 = 1: layer 2 of the grouped bf16x3
 * kernels with untransposed tiles (round 4's epilogue; bit-identical results, tools/r05_tr_ab.sh).  Key 5 also selects the fp32-MFMA
 * twins of the bf16x3 kernels (53 / 54 / 56 / 57; 22: DGCNN's feature-space kNN distances; 59: only round 5's additions, the
 * dense layers with 256 n + 64 outputs, with inputs that are no multiple of 32, and the split-K layer)
 * and timing probes whose results are WRONG: 91 / 92 / 93 in the chain kernel's layer 3, 94 / 95 in conv5's pooled GEMM (no split
 * arithmetic when the activations are staged / no pooling epilogue; tools/r05_conv5_probe.sh). */
int iq_set_tuning(int key, int value);
/* Debug: workgroups per CU the runtime admits for the chain kernel variants (100*v0 + v2). */
int iq_debug_chain_occupancy(void);
/* Debug: per-phase shader-clock sums of the feature-STN chain (diagnostic STAMP instantiation, never
 * used unless enabled).  enable != 0 arms and zeroes 8 counters; out_host (8 x u64 or NULL) receives
 * the counters accumulated so far. */
int iq_debug_stamps(int enable, unsigned long long* out_host);
/* Debug: counters of the kNN kernels while tuning key 4 = 3 (synchronises the device, reads and clears them): [0] selection rounds,
 * [1] busy lanes summed over rounds, [2] waves, [3] queries flagged as near-ties, [4] queries re-ranked (their 21 candidates each),
 * [6] of those, zero-gap queries whose re-ranking covers ALL rows; [5] and [7] are not written (always 0). */
int iq_debug_knn_counters(unsigned long long* out_host /*8, host*/);
int iq_profile_read(int slot, double* total_ms, int* launches);

/* Diagnostic: the rate the bf16 matrix pipe SUSTAINS on this board now - a register-only loop of v_mfma_f32_32x32x16_bf16 on
 * random operands, one wave per SIMD on every CU, for about `seconds` (two launches: a calibration, then the measurement; both
 * synchronise the stream).  tflops: dense bf16 TFLOP/s of the measured launch; clock_ghz (optional): shader clock held by its first
 * wave (s_memtime ticks / wall time of the launch).  scratch: device buffer of at least 256 * compute-unit-count floats.  MFMA-dense kernels
 * on MI355X are power-bound: bench.py divides by THIS figure for `frac_of_sustained_bf16_ceiling` instead of a constant. */
int iq_debug_mfma_sustained(double seconds, float* scratch /*device*/, size_t scratch_floats, double* tflops /*host*/,
                            double* clock_ghz /*host or NULL*/, iq_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* IQ_DEBUG_H_ */
