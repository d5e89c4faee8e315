#!/usr/bin/env python3
"""Headline benchmark: coalitions/sec (masked forward passes/sec), PointNet, 1024-point clouds,
32 regions x 1000 sampled permutations per cloud (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W          # N > 1: starts its own N ranks (interpret_quality_amd/launch.py)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    ... bench.py --scaling strong [--model dgcnn]          # the path's OWN shard axes at N ranks (poses, pairs)
    ... bench.py --scaling sweep [--sweep-clouds 1]        # BASELINE configs[4]: all models x datasets, units pulled by the ranks

One step (default, `"scaling": "weak"`) = the whole Shapley hot path for one synthetic cloud, ALL of it on the device and all
of it inside the timed region: 1000 permutations sampled from NumPy's legacy generator stream (iq_sample_permutations; the state
stays in HBM from step to step), their 33 000 prefix coalitions as region bit masks (iq_prefix_keep_masks), 33 000 masked
forward passes, reward, per-region float64 accumulation.  Resident in HBM before the timed region: the cloud, its region ids,
the generator state, the weights.  With N > 1 every rank works on its own cloud (independent objects) and one RCCL all-gather
per step brings the per-coalition logits to every rank, as the artefact writer on rank 0 needs them.
IQ_FORCE_DIST=1 creates the process group (and runs the barrier / all-gather) at N = 1 too.  Rank 0 prints ONE JSON line.

Timing: W warm-up steps, then `--repeats` (5) timed regions of at least `--min-region-s` (2 s) each, whatever K
is: a region is ceil(2 s / block time) back-to-back BLOCKS of EXACTLY K steps, every block bracketed by barrier + synchronize on
both sides and reduced with MAX over ranks; `value` / `ms_per_step` are the MEDIAN block's (`timing.region_s` states the region).  The library's HIP-event profiler is OFF in the timed regions; kernel durations come from a separate
profiled pass afterwards.

`roofline` (dominant kernel = pn_chain_kernel, matrix-pipe bound): `achieved` = the float32 FLOP of the MFMA tiles it issues over
its HIP-event launch time; `peak` = the matrix pipe's float32-equivalent rate for the kernel's instruction mix (layers 1-2 on the
fp32 MFMA at 157.3 TF, layer 3 as six exact bf16 products per float32 product on the bf16 MFMA at 16 x 157.3 TF: `peak_basis`);
`frac` = achieved / peak (`frac_basis: "executed"`); `frac_useful` counts the coalitions' distinct rows only (no tile padding);
`frac_algorithmic` is SURVEY 8d's figure, the dense reference layers over all 1024 rows of every coalition (the kernel skips
duplicate points, exactly: DESIGN.md 3).

Every default line also carries `strong_scaling` (`--strong-steps`, default 1 step after the headline measurement): the same
workload as `--scaling strong` at this rank count, so the driver's N = 1, 2, 4, 8 runs hold a strong-scaling curve too.
`--scaling strong`: ONE cloud's work sharded over the N ranks with the drivers' own code - the 216-pose rotation sweep
(pose_sweep.sharded_shapley: poses sharded, tools/final_common.py:158) and one 300-pair x 13-ratio interaction setting
(interaction.compute_order_interaction_logits: pairs sharded, final_point_binary_interaction_logits.py:37), every coalition a
forward pass (no driver-level de-duplication), plus the share of the step spent in the all-gathers.

After the timed regions (N = 1 only): the other model families on their BASELINE shapes with a roofline each
(`other_models`), the stock PyTorch-ROCm eager restatement on the same GPU (`gpu_eager_baseline`), the CPU oracle on the host
cores (`cpu_baseline`), and the HBM traffic of the dominant kernel from two rocprofv3 counter passes of this script
(`roofline.traffic`; null when the profiler is unavailable).
"""
import argparse
import contextlib
import csv
import ctypes
import glob
import io
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

NUM_POINTS, NUM_REGIONS, NUM_PERMS = 1024, 32, 1000
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 MFMA peak (no xf32/TF32 on gfx950)
# What a pure register loop of v_mfma_f32_32x32x16_bf16 SUSTAINS under the board's power cap, as a fraction of the dense peak, is
# MEASURED on the board the bench runs on (iq_debug_mfma_sustained, ~1.5 s right after the timed regions, outside them): boards
# differ by 10 % and more in the clock they hold (MI355X_MICROARCH.md, DVFS give-back item 5; round 4 quoted a constant 0.776 from
# the builder's board).  Reported beside `frac`, never instead.  None until measured / when --sustained-s 0.
SUSTAINED = {"frac": None}
PEAK_BF16_MFMA_TFLOPS = 16 * 157.3   # same table: the bf16 MFMA runs 16 x the fp32 one (~2.5 PF dense)
BF3_PRODUCTS = 6              # bf16 products per float32 product when both operands are split in three bf16 terms
CHAIN_MAC_PER_ROW = 143360.0  # 64*64 + 64*128 + 128*1024: the MFMA layers of one chain-kernel row
CHAIN_MAC_L12, CHAIN_MAC_L3 = 64 * 64 + 64 * 128, 128 * 1024   # layers 1-2 run in 32-row tiles, layer 3 ends on a 16-row tile
CHAIN_MAC_L1, CHAIN_MAC_L2 = 64 * 64, 64 * 128                  # (fp32 kernel only); with bf16x3 layer 1 alone stays on the fp32 MFMA
SLOT_DOMINANT = 5


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30, help="steps per timed region (30 x 70 ms = 2.1 s)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions; each is as many blocks of --steps steps as --min-region-s needs")
    ap.add_argument("--min-region-s", type=float, default=2.0, help="seconds of work per timed region (SURVEY 8d: >= 2 s)")
    ap.add_argument("--scaling", choices=["weak", "strong", "sweep"], default="weak",
                    help="weak: one cloud per rank (headline); strong: one cloud's pose sweep + interaction setting sharded over the ranks; "
                         "sweep: BASELINE configs[4], all models x datasets through both pipelines (tools/sweep.py), --sweep-clouds each")
    ap.add_argument("--sweep-clouds", type=int, default=1, help="--scaling sweep: clouds per (model, dataset); the reference's is 30")
    ap.add_argument("--sweep-models", default="", help="--scaling sweep: comma list (default: all six)")
    ap.add_argument("--sweep-datasets", default="", help="--scaling sweep: comma list (default: both)")
    ap.add_argument("--sweep-reduced", type=int, default=0, help="--scaling sweep: 1 = rehearsal sizes (100 saved permutations, 5 pairs, "
                    "3 contexts) instead of the reference's")
    ap.add_argument("--model", default="pointnet", help="model of the strong-scaling mode")
    ap.add_argument("--profile-steps", type=int, default=3, help="steps of the separate profiled pass (kernel durations)")
    ap.add_argument("--strong-steps", type=int, default=1, help="weak mode: also time this many strong-scaling steps (one cloud sharded over the "
                    "ranks) after the headline measurement and report them under `strong_scaling` (0: skip)")
    ap.add_argument("--perms", type=int, default=NUM_PERMS, help="permutations per cloud per step")
    ap.add_argument("--regions", type=int, default=NUM_REGIONS)
    ap.add_argument("--cpu-baseline", type=int, default=1, help="time the CPU oracle on rank 0 at N=1")
    ap.add_argument("--cpu-perms", type=int, default=40, help="permutations of the bounded CPU sample")
    ap.add_argument("--eager-baseline", type=int, default=1, help="time the stock PyTorch-ROCm eager restatement (N=1)")
    ap.add_argument("--other-models", type=int, default=1, help="after the timed region (N=1 only): coalitions/s and the "
                    "dominant kernel's roofline of the other model families on BASELINE configs[2..3] shapes")
    ap.add_argument("--sustained-s", type=float, default=1.5, help="seconds of the register-only bf16 MFMA loop that measures the "
                    "board's sustained matrix rate after the timed regions (0 = skip; frac_of_sustained_bf16_ceiling is then null)")
    ap.add_argument("--traffic", type=int, default=1, help="measure the chain kernel's HBM bytes with two rocprofv3 "
                    "counter passes of this script (N=1 only; null if rocprofv3 is unavailable)")
    return ap.parse_args()


def host_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, 64)


def cpu_baseline(num_regions, perms, bs=10):
    """The CPU oracle (port of the reference's PyTorch path) on the host cores, bounded sample."""
    from interpret_quality_amd import synth
    from oracle import ref_cpu
    torch.set_num_threads(host_cores())
    model = ref_cpu.PointNetOracle(synth.to_torch(synth.pointnet_state_dict(0)))
    pts, label = synth.make_cloud(0)
    data = torch.from_numpy(pts).unsqueeze(0)
    lbl = torch.tensor([label])
    fps = ref_cpu.farthest_point_sample(data, num_regions)[0]
    region_id = ref_cpu.cal_region_id(data, fps)
    orders = synth.make_orders(perms, num_regions, seed=1)
    perms = perms // bs * bs
    ref_cpu.shap_sampling_all_regions_batch(model, data, lbl, region_id, orders[:bs], bs, bs, num_regions)  # warm-up
    t0 = time.time()
    ref_cpu.shap_sampling_all_regions_batch(model, data, lbl, region_id, orders, perms, bs, num_regions)
    dt = time.time() - t0
    n = perms * (num_regions + 1)
    return {"value": n / dt, "unit": "coalitions/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle/ref_cpu.py shap_sampling_all_regions_batch, PointNet, 1 synthetic cloud, R=%d, "
                      "%d permutations (bs=%d, %d clouds per forward) = %d coalitions in %.1f s"
                      % (num_regions, perms, bs, bs * (num_regions + 1), n, dt)}


def gpu_eager_baseline(num_regions):
    """SURVEY §8d baseline (ii): the reference's Shapley loop restated with stock PyTorch-ROCm eager ops on this GPU
    (index-put masking storm, one batched forward, one host sync per permutation; config.py's bs = 50 for PointNet)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("iq_eager_gpu_baseline", os.path.join(REPO, "oracle", "eager_gpu_baseline.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.run_eager(perms=100, bs=50, regions=num_regions)


def _read_slot(lib, slot):
    from interpret_quality_amd import _lib
    ms, n, work = ctypes.c_double(0), ctypes.c_int(0), ctypes.c_double(0)
    _lib.check(lib.iq_profile_read_work(slot, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(work)), "iq_profile_read_work")
    return ms.value, n.value, work.value


# dense MAC count of each reference network per coalition (the algorithmic figure) and the kernel that
# dominates its step here
OTHER_MODELS = {
    "pointnet2": {"config": "BASELINE configs[2]", "dense_gflop": 7.83, "kernel": "pn2_group_bf3_kernel (sa2, the 128-128-256 scales)", "bf3": True},
    "dgcnn": {"config": "BASELINE configs[3]", "dense_gflop": 5.33, "kernel": "pn_gemm_bf3_kernel<pool> (conv5 + max/mean pool)", "bf3": True},
    "gcnn": {"config": "BASELINE configs[3] (gcnn)", "dense_gflop": 4.79, "kernel": "pn_gemm_bf3_kernel<pool> (conv5 + max/mean pool)", "bf3": True},
    # layers 2-3 as bf16x3, the contraction over the members (16 c3 of c1 c2 + c2 c3 + 16 c3 MACs per member) on the fp32 MFMA
    "pointconv": {"config": "Shapley shape of configs[2]", "dense_gflop": None, "kernel": "pc_group_bf3_kernel (sa2)", "bf3": True,
                  "bf3_share": (128 * 128 + 128 * 256) / (128 * 128 + 128 * 256 + 16 * 256)},
}


def pointconv_dense_mac(n=1024):
    """Dense MAC count of models/pointconv.py:324-424 per cloud (the recount SURVEY 8d asks for; its estimate was 1.2 G):
    per set abstraction the kNN / density distance products (3 per pair), the grouped MLP, DensityNet 1-16-8-1 and WeightNet
    3-8-8-16 per member, the (C x K).(K x 16) contraction per group and the 16C -> C linear layer; then the classifier."""
    total = 0
    for n_in, groups, k, cin, mlp in ((n, 512, 32, 3, (64, 64, 128)), (512, 128, 64, 131, (128, 128, 256)), (128, 1, 128, 259, (256, 512, 1024))):
        rows = groups * k
        chain = sum(a * b for a, b in zip((cin,) + mlp[:-1], mlp))
        total += 3 * n_in * n_in                       # compute_density: square_distance(xyz, xyz), :199-209
        total += 3 * groups * n_in if groups > 1 else 0  # knn_point: square_distance(new_xyz, xyz), :103-114
        total += rows * (chain + (16 + 16 * 8 + 8) + (3 * 8 + 8 * 8 + 8 * 16))
        total += groups * mlp[-1] * k * 16               # new_points x weights, :376-381
        total += groups * 16 * mlp[-1] * mlp[-1]         # self.linear, :382
    return total + 1024 * 512 + 512 * 256 + 256 * 10


OTHER_MODELS["pointconv"]["dense_gflop"] = 2.0 * pointconv_dense_mac() / 1e9   # 1 206.9 M MAC = 2.414 GFLOP


def interaction_workload(regions, num_pairs=300, max_ctx=100, seed=1):
    """The reference's interaction inputs for one cloud (final_gen_pair.py:288-300 and :18-43): 300 random region pairs and,
    for each of the 13 ratios, up to 100 sampled contexts per pair (all C(R-2, m) when there are fewer).  Returns
    [(pairs (P,2), contexts (P,C,m))] per ratio; 1032 contexts per pair at R = 32."""
    from argparse import Namespace
    from interpret_quality_amd import gen_pair
    from interpret_quality_amd.interaction import DEFAULT_RATIOS
    a = Namespace(num_regions=regions, num_pairs_random=num_pairs, num_save_context_max=max_ctx, ratio=DEFAULT_RATIOS)
    np.random.seed(seed)
    pairs = gen_pair.gen_pair_random(a)
    out = []
    tmp = tempfile.mkdtemp(prefix="iq_bench_ctx_")
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            gen_pair.gen_context(pairs, tmp + "/", a)
        for ratio in DEFAULT_RATIOS:
            out.append((pairs, np.load(tmp + "/ratio%d_context_list.npy" % int(ratio * 100))))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def other_models(dev, lib, regions=32):
    """Not the headline: coalitions/s of the other model families through the same C ABI, each on the shape BASELINE.json
    names for it - PointNet++ / PointConv: Shapley, 100 permutations x 33 = 3300 coalitions per step (configs[2]);
    DGCNN / GCNN: the interaction stage of ONE cloud setting at full size, 300 pairs x 13 ratios (1032 contexts per pair) x 4
    = 1 238 400 coalitions per step (configs[3]).  Every coalition row is a forward pass (no driver-level de-duplication).
    A second, profiled pass gives the dominant kernel's HIP-event time and the FLOP its MFMA tiles execute -> `roofline`.
    Failures are reported, never raised: the headline line must still be printed."""
    import argparse as ap
    from interpret_quality_amd import final_common, hip_ops, interaction, synth
    from interpret_quality_amd.dgcnn import DGCNN_cls, GCNN_cls
    from interpret_quality_amd.pointconv import PointConvDensityClsSsg
    from interpret_quality_amd.pointnet2 import PointNet2ClsMsg
    saved = final_common.distinct_coalitions
    final_common.distinct_coalitions = lambda k: (np.asarray(k, dtype=np.uint64), np.arange(len(k)))
    out = {}
    try:
        pts, label = synth.make_cloud(0)
        data = torch.from_numpy(pts).unsqueeze(0).to(dev)
        lbl = torch.tensor([label], device=dev)
        region_id = hip_ops.region_assign(data[0].contiguous(), hip_ops.fps(data, regions)[0].contiguous()).cpu().numpy()
        orders = synth.make_orders(100, regions, seed=1)
        inter = interaction_workload(regions)
        n_inter = sum(4 * p.shape[0] * c.shape[1] for p, c in inter)
        specs = (("pointnet2", PointNet2ClsMsg, synth.pointnet2_state_dict, "shapley"),
                 ("dgcnn", DGCNN_cls, synth.dgcnn_state_dict, "interaction"),
                 ("gcnn", GCNN_cls, synth.dgcnn_state_dict, "interaction"),
                 ("pointconv", PointConvDensityClsSsg, synth.pointconv_state_dict, "shapley"))
        for name, cls, sd, mode in specs:
            try:
                m = cls(ap.Namespace(dataset="modelnet10", k=20) if "cnn" in name else None)
                m.load_state_dict(synth.to_torch(sd(0)))
                m = m.to(dev).eval()
                a = ap.Namespace(model=name, softmax_type="modified", num_points=NUM_POINTS, num_regions=regions, num_samples=100,
                                 shapley_batch_size=20, interaction_batch_size=100, verbose=False)
                if mode == "shapley":
                    def run(frac=1.0):
                        final_common.shap_sampling_all_regions_batch(m, data, lbl, region_id, orders, a)
                        return 100 * (regions + 1)
                    steps, workload = 5, "Shapley, %d regions x 100 permutations = %d coalitions per step" % (regions, 100 * (regions + 1))
                else:
                    def run(frac=1.0):
                        n = 0
                        with contextlib.redirect_stdout(io.StringIO()):
                            for pairs, ctx in inter:
                                k = max(1, int(round(pairs.shape[0] * frac)))
                                interaction.compute_order_interaction_logits(m, data, region_id, pairs[:k], ctx[:k], a)
                                n += 4 * k * ctx.shape[1]
                        return n
                    steps, workload = 1, ("interaction, one cloud setting: 300 pairs x 13 ratios (%d contexts per pair) x 4 = %d "
                                          "coalitions per step" % (n_inter // 1200, n_inter))
                run(0.1)                                    # warm-up (a tenth of the pairs for the interaction shape)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = sum(run() for _ in range(steps))
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                # profiled pass (its syncs for the work counts stay out of the throughput above)
                lib.iq_profile_enable(1)
                run(0.1 if mode == "interaction" else 1.0)
                torch.cuda.synchronize()
                lib.iq_profile_enable(0)
                ms, launches, work = _read_slot(lib, SLOT_DOMINANT)
                for s in range(5):
                    _read_slot(lib, s)                      # forget the other spans
                spec = OTHER_MODELS[name]
                achieved = work / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
                # bf16x3 kernels run on the bf16 matrix pipe, six exact bf16 products per float32 product; a kernel
                # that keeps part of its FLOP on the fp32 MFMA (bf3_share < 1) is priced against the blend of the two rates
                peak = PEAK_F32_MFMA_TFLOPS
                if spec.get("bf3"):
                    share = spec.get("bf3_share", 1.0)
                    peak = 1.0 / (share / (PEAK_BF16_MFMA_TFLOPS / BF3_PRODUCTS) + (1.0 - share) / PEAK_F32_MFMA_TFLOPS)
                out[name] = {"value": n / dt, "unit": "coalitions/s", "steps": steps, "workload": workload, "config": spec["config"],
                             "roofline": {"bound": "mfma", "kernel": spec["kernel"], "achieved": achieved, "peak": peak,
                                          "unit": "TFLOP/s", "frac": achieved / peak, "frac_of_fp32_mfma_peak": achieved / PEAK_F32_MFMA_TFLOPS,
                                          "frac_of_sustained_bf16_ceiling": (achieved / (peak * SUSTAINED["frac"])
                                                                             if spec.get("bf3") and SUSTAINED["frac"] else None),
                                          "traffic": None,
                                          "avg_launch_ms": ms / max(launches, 1), "launches": launches,
                                          "executed_flop_per_launch": work / max(launches, 1),
                                          "algorithmic_tflops_whole_step": spec["dense_gflop"] * 1e9 * (n / dt) / 1e12,
                                          "note": "achieved = FLOP of the MFMA tiles the dominant kernel issues / its HIP-event time; "
                                                  "algorithmic_tflops_whole_step = SURVEY 8d dense FLOP per coalition x coalitions/s "
                                                  "(exceeds the peak where exact restructurings skip work)"}}
                del m
            except Exception as e:  # noqa: BLE001
                out[name] = {"error": repr(e)[:300]}
    finally:
        final_common.distinct_coalitions = saved
        lib.iq_profile_enable(0)
    return out


def measure_sustained(lib, dev, seconds):
    """The bf16 matrix pipe's sustained rate on THIS board, right now (warm from the timed regions): a register-only MFMA loop of
    about `seconds` inside the library (include/iq_debug.h).  Returns the record for the JSON line and sets SUSTAINED["frac"]."""
    if seconds <= 0:
        return None
    from interpret_quality_amd import _lib
    try:
        scratch = torch.empty(512 * 1024, dtype=torch.float32, device=dev)
        tf, clk = ctypes.c_double(0), ctypes.c_double(0)
        _lib.check(lib.iq_debug_mfma_sustained(float(seconds), ctypes.c_void_p(scratch.data_ptr()), scratch.numel(), ctypes.byref(tf),
                                               ctypes.byref(clk), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   "iq_debug_mfma_sustained")
        SUSTAINED["frac"] = tf.value / PEAK_BF16_MFMA_TFLOPS
        return {"tflops": tf.value, "frac_of_dense_peak": SUSTAINED["frac"], "shader_clock_ghz": clk.value, "seconds": seconds,
                "what": "register-only loop of v_mfma_f32_32x32x16_bf16 on random operands, one wave per SIMD on every CU, measured on "
                        "this board after the timed regions (outside them)"}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:300]}


def measure_traffic():
    """HBM bytes per chain-kernel launch, measured NOW: two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE: they do not fit
    one pass) of this script as child processes, kernel trace only.  Units and the gfx950 correction per
    MI355X_MICROARCH.md §HBM: the counters are in KB, FETCH_SIZE reports half the bytes of wide coalesced reads.  Returns
    (bytes per launch averaged over the feature-STN and trunk instantiations, detail dict) or (None, reason)."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    per = {}
    tmp = tempfile.mkdtemp(prefix="iq_bench_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "IQ_FORCE_DIST"):
        env.pop(k, None)
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--repeats", "1", "--min-region-s", "0", "--profile-steps", "1",
                   "--cpu-baseline", "0", "--other-models", "0", "--eager-baseline", "0", "--traffic", "0", "--strong-steps", "0", "--sustained-s", "0"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s failed (rc %d): %s" % (counter, r.returncode, r.stderr.decode(errors="replace")[-200:])
            for row in csv.DictReader(open(files[0])):
                name = row["Kernel_Name"]
                if "pn_chain_kernel" in name and row["Counter_Name"] == counter:
                    key = "fstn" if "pn_chain_kernel<1" in name else "trunk" if "pn_chain_kernel<2" in name else "prepool"
                    per.setdefault(key, {}).setdefault(counter, []).append(float(row["Counter_Value"]) * 1024.0)
    except (OSError, subprocess.SubprocessError, KeyError, ValueError) as e:
        return None, "traffic pass failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    detail, tot, n = {}, 0.0, 0
    for key in ("fstn", "trunk"):
        if key not in per or "FETCH_SIZE" not in per[key] or "WRITE_SIZE" not in per[key]:
            return None, "no %s chain launches in the counter output" % key
        fetch = 2.0 * float(np.mean(per[key]["FETCH_SIZE"]))    # gfx950: FETCH_SIZE counts half of a wide coalesced read
        write = float(np.mean(per[key]["WRITE_SIZE"]))
        detail[key] = {"fetch_bytes_corrected_x2": fetch, "write_bytes": write}
        tot += fetch + write
        n += 1
    return tot / n, detail


def rows_of(keep_np, sizes, regions):
    """Distinct rows of each coalition (kept points + one centre row when anything is masked)."""
    bits = ((keep_np[:, None] >> np.arange(regions, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.int64)
    kept = bits @ sizes
    return kept + (kept < NUM_POINTS)


def strong_scaling(args, rank, world, dev, collectives, rehearsal, steps=None, warmup=None, repeats=None):
    """`--scaling strong`: ONE cloud's work through the drivers' own sharded code paths; see the module docstring.
    Returns the JSON record on rank 0 (None elsewhere)."""
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    repeats = args.repeats if repeats is None else repeats
    import argparse as ap
    import torch.distributed as dist
    from interpret_quality_amd import dist as iqdist
    from interpret_quality_amd import final_common, hip_ops, interaction, pose_sweep, synth
    from interpret_quality_amd.final_util import load_model, set_interaction_batch_size, set_model_args, set_shapley_batch_size
    R = args.regions
    a = ap.Namespace(model=args.model, dataset="modelnet10", synthetic=True, device=dev, num_points=NUM_POINTS, num_regions=R,
                     num_samples=100, softmax_type="modified", verbose=False, angle_threshold=pose_sweep.ANGLE_THRESHOLD,
                     num_grid_enum_rotate=pose_sweep.NUM_GRID_ENUM_ROTATE)
    set_model_args(a)
    set_shapley_batch_size(a)
    set_interaction_batch_size(a)
    model = load_model(a)
    pts, label = synth.make_cloud(0)
    data = torch.from_numpy(pts).unsqueeze(0).to(dev)
    lbl = torch.tensor([label], device=dev)
    region_id = hip_ops.region_assign(data[0].contiguous(), hip_ops.fps(data, R)[0].contiguous()).cpu().numpy().astype(np.int64)
    orders = synth.make_orders(a.num_samples, R, seed=1)
    params = pose_sweep.generate_rotate_angle(a, dev)
    poses = torch.cat([pose_sweep.rotate_xyz(data, params[i]) for i in range(params.shape[0])], dim=0)
    inter = interaction_workload(R)
    n_sweep = (poses.shape[0] + 1) * a.num_samples * (R + 1)
    n_inter = sum(4 * p.shape[0] * c.shape[1] for p, c in inter)
    saved = final_common.distinct_coalitions
    final_common.distinct_coalitions = lambda k: (np.asarray(k, dtype=np.uint64), np.arange(len(k)))   # every coalition is a forward pass
    iqdist.GATHER_EVENTS = []

    def step():
        with contextlib.redirect_stdout(io.StringIO()):
            pose_sweep.sharded_shapley(model, data, poses, lbl, region_id, orders, a)
            for pairs, ctx in inter:
                interaction.compute_order_interaction_logits(model, data, region_id, pairs, ctx, a)

    def fence():
        if collectives:
            iqdist.group_barrier()
        torch.cuda.synchronize()

    try:
        for _ in range(warmup):
            step()
        fence()
        regions_s, gather_s = [], []
        for _ in range(max(repeats, 1)):
            iqdist.GATHER_EVENTS.clear()
            fence()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            fence()
            el = time.perf_counter() - t0
            g = sum(s_.elapsed_time(e_) for s_, e_ in iqdist.GATHER_EVENTS) * 1e-3
            if collectives:
                t = torch.tensor([el, g], dtype=torch.float64, device="cpu" if rehearsal else dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el, g = float(t[0].item()), float(t[1].item())
            regions_s.append(el)
            gather_s.append(g)
    finally:
        final_common.distinct_coalitions = saved
        iqdist.GATHER_EVENTS = None
    if rank != 0:
        return None
    k = int(np.argsort(regions_s)[len(regions_s) // 2])
    elapsed = regions_s[k]
    total = (n_sweep + n_inter) * steps
    return {
        "metric": "coalitions/sec (masked forward passes/sec), %s 1024-pt ModelNet10" % ("PointNet" if args.model == "pointnet" else args.model),
        "value": total / elapsed, "unit": "coalitions/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "ONE cloud per step, sharded over the ranks by the drivers' own code: rotation sweep (%d poses + the original) "
                               "x 100 permutations x %d = %d coalitions, poses sharded; interaction setting 300 pairs x 13 ratios x 4 = %d "
                               "coalitions, pairs sharded; every coalition evaluated" % (poses.shape[0], R + 1, n_sweep, n_inter),
                   "model": args.model, "num_points": NUM_POINTS, "num_regions": R,
                   "parallelism": "%d rank(s); one padded all_gather_into_tensor per cloud (Shapley values, logits) and per ratio (logits)" % world},
        "timing": {"protocol": "median of %d regions of %d steps, barrier + synchronize around each, MAX over ranks" % (len(regions_s), steps),
                   "regions_s": regions_s},
        "gather": {"seconds_per_step": gather_s[k] / steps, "share_of_step": gather_s[k] / elapsed,
                   "note": "HIP-event time of the all-gathers on the slowest rank (includes waiting for the last rank to arrive, "
                           "i.e. the shard imbalance: 217 poses and 300 pairs do not divide evenly)"},
    }


def sweep_scaling(args, rank, world, collectives):
    """`--scaling sweep`: BASELINE configs[4] through tools/sweep.py's own code in this process group - every (model, dataset,
    cloud) unit through exp_shapley.sh's and exp_interaction.sh's stages, units pulled by the ranks from a shared queue.  ONE
    step = the whole sweep (no warm-up: model loading, artefact writing and the phase barriers are part of the job), in a
    scratch directory that is removed afterwards.  Returns the record on rank 0."""
    import importlib.util
    import torch.distributed as dist
    spec = importlib.util.spec_from_file_location("iq_sweep", os.path.join(REPO, "tools", "sweep.py"))
    sweep = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sweep)
    box = [tempfile.mkdtemp(prefix="iq_bench_sweep_", dir="/tmp") if rank == 0 else None]
    if collectives:
        dist.broadcast_object_list(box, src=0)
    argv = ["--synthetic", "--num_clouds", str(args.sweep_clouds)]
    if args.sweep_models:
        argv += ["--models", args.sweep_models]
    if args.sweep_datasets:
        argv += ["--datasets", args.sweep_datasets]
    if args.sweep_reduced:
        argv += ["--num_samples_save", "100", "--num_pairs_random", "5", "--num_save_context_max", "3"]
    cwd = os.getcwd()
    os.chdir(box[0])
    try:
        rec = sweep.run(sweep.parse(argv), emit=False)
    finally:
        os.chdir(cwd)
        if collectives:
            from interpret_quality_amd import dist as iqdist
            iqdist.group_barrier()
        if rank == 0:
            shutil.rmtree(box[0], ignore_errors=True)
    if rec is not None:
        rec["steps"], rec["warmup"] = 1, 0
        rec["timing"] = {"protocol": "one whole sweep, wall clock on rank 0 from the first phase to the last phase barrier; no warm-up"}
    return rec


def main():
    args = parse()
    from interpret_quality_amd import launch
    if args.gpus > 1 and not launch.under_launcher():
        # plain `python bench.py --gpus N`: this process has not touched the GPU; it becomes the parent of N fresh ranks
        # (interpret_quality_amd/launch.py), relays rank 0's JSON line and the worst exit code
        raise SystemExit(launch.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world    # the launcher's world size wins over the flag
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # IQ_BENCH_REHEARSAL=1 (= IQ_REHEARSAL=1): rehearse the N > 1 code path on a one-GPU box (all ranks on cuda:0, gloo
    # collectives).  Never set by the driver; the numbers of a rehearsal mean nothing.
    rehearsal = os.environ.get("IQ_BENCH_REHEARSAL") == "1" or os.environ.get("IQ_REHEARSAL") == "1"
    force_dist = os.environ.get("IQ_FORCE_DIST") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    from interpret_quality_amd import dist as iqdist
    collectives = world > 1 or force_dist
    if collectives:
        launch.ensure_rendezvous()    # the launcher's MASTER_PORT; a free one for a forced single-rank group
        import datetime
        dist.init_process_group("gloo" if rehearsal else "nccl", rank=rank, world_size=world,   # nccl = RCCL on ROCm
                                timeout=datetime.timedelta(seconds=int(os.environ.get("IQ_DIST_TIMEOUT_S", "1800"))))
    try:
        if args.scaling == "sweep":
            out = sweep_scaling(args, rank, world, collectives)
        elif args.scaling == "strong":
            out = strong_scaling(args, rank, world, dev, collectives, rehearsal)
        else:
            out = weak_scaling(args, rank, world, dev, collectives, rehearsal, force_dist)
            if args.strong_steps > 0:
                # the SAME line also carries the path's own shard axes at this rank count (one cloud's pose sweep + interaction
                # setting, poses / pairs sharded): the driver's N = 1, 2, 4, 8 runs then hold a strong-scaling curve as well
                strong = strong_scaling(args, rank, world, dev, collectives, rehearsal, steps=args.strong_steps, warmup=1, repeats=1)
                if rank == 0:
                    out["strong_scaling"] = {k: strong[k] for k in ("value", "unit", "ms_per_step", "steps", "config", "gather")}
        if rank == 0:
            print(json.dumps(out), flush=True)
    finally:
        if collectives:   # orderly exit: no rank's communicator goes away under a peer (interpret_quality_amd/dist.py: shutdown)
            try:
                torch.cuda.synchronize()
                iqdist.group_barrier()
            finally:
                dist.destroy_process_group()


def weak_scaling(args, rank, world, dev, collectives, rehearsal, force_dist):
    import torch.distributed as dist
    from interpret_quality_amd import _lib, hip_ops, synth
    from interpret_quality_amd import dist as iqdist
    from interpret_quality_amd.pointnet import PointNetCls

    lib = _lib.load()
    if os.environ.get("IQ_BENCH_FP32_L3") == "1":
        lib.iq_set_tuning(5, 54)      # layer 3 of the chains on the fp32 MFMA (round 3's kernel), for A/B runs
    R, S = args.regions, args.perms
    model = PointNetCls(None)
    model.load_state_dict(synth.to_torch(synth.pointnet_state_dict(0)))
    model = model.to(dev).eval()

    # ---- inputs, resident in HBM before timing (cloud index = rank: independent objects) ------
    pts, label = synth.make_cloud(rank)
    data = torch.from_numpy(pts).unsqueeze(0).to(dev)
    fps_idx = hip_ops.fps(data, R)[0].contiguous()
    region_id = hip_ops.region_assign(data[0].contiguous(), fps_idx).reshape(1, -1)
    hip_ops.check_index_range(region_id, 0, R, "region_id")   # once, before timing; the timed calls skip the check
    np.random.seed(1)                                            # set_random(1), tools/final_util.py:113-120
    mt_state = hip_ops.mt_state_to_device(dev)                   # the generator the reference samples from, continued on the device
    center = torch.mean(data, dim=1).contiguous()
    n_coal = S * (R + 1)
    # ONE preallocated receive buffer for the per-step all-gather of the logits (132 KB per rank)
    gathered = torch.empty((world * n_coal, 10), dtype=torch.float32, device=dev) if collectives else None

    def step():
        orders = hip_ops.sample_permutations(mt_state, S, R)      # coalition sampling (final_shapley_value.py:59-72)
        keep = hip_ops.prefix_keep_masks(orders)                   # region masking, as bit masks (tools/final_common.py:56-60)
        logits = model.coalition_logits(data, center, region_id, keep, None, num_regions=R, validate=False)
        v = hip_ops.reward(logits, label, True)
        phi_sum, _, _ = hip_ops.shapley_accum(v, orders)
        if collectives:
            dist.all_gather_into_tensor(gathered, logits)
        return phi_sum, logits, orders, keep

    def fence():
        if collectives:
            iqdist.group_barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()

    def block():
        """EXACTLY args.steps steps, barrier + synchronize on both sides, MAX over ranks -> seconds."""
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        fence()
        el = time.perf_counter() - t0
        if collectives:
            t = torch.tensor([el], dtype=torch.float64, device="cpu" if rehearsal else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, out

    # SURVEY 8d asks for >= 2 s of work per timed region whatever --steps is: a region is as many back-to-back blocks of
    # exactly --steps steps (each one bracketed as the contract says) as it takes to fill --min-region-s; the reported time is
    # the MEDIAN BLOCK's.  Every rank sees the same block times (MAX over ranks), so all of them stop a region at the same block.
    regions_s, blocks_s, blocks_per_region = [], [], []
    for _ in range(max(args.repeats, 1)):
        this = []
        while not this or sum(this) < args.min_region_s:
            el, (phi_sum, logits, orders, keep) = block()
            this.append(el)
        blocks_s.extend(this)
        blocks_per_region.append(len(this))
        regions_s.append(float(np.sum(this)))
    blocks = int(np.median(blocks_per_region))
    elapsed = float(np.sort(blocks_s)[len(blocks_s) // 2])        # the median block of args.steps steps
    if collectives:
        # every rank's chunk of the gather must hold that rank's logits (rank r's own chunk is checkable locally)
        assert torch.equal(gathered[rank * n_coal:(rank + 1) * n_coal], logits), "all-gather returned a wrong chunk"

    # sanity on the last timed step: efficiency axiom (sum phi = v(N) - v(empty) for every permutation), and the device sampler
    # left NumPy's stream where NumPy itself would be after the same number of draws
    v = hip_ops.reward(logits, label, True)
    eff = abs(float(phi_sum.sum().item()) / S - float((v[R] - v[0]).item()))
    assert eff < 1e-3, "efficiency check failed: %g" % eff
    assert sorted(orders[0].tolist()) == list(range(R)) and int(keep[R].item()) == (-1 if R == 64 else (1 << R) - 1)

    # ---- separate profiled pass: HIP-event durations of the dominant kernel and the rows it executed ----------------------
    sizes = np.bincount(region_id.cpu().numpy().reshape(-1), minlength=R).astype(np.int64)
    keeps = []
    lib.iq_profile_enable(1)
    for _ in range(max(args.profile_steps, 1)):
        keeps.append(step()[3])
    torch.cuda.synchronize()
    lib.iq_profile_enable(0)
    sustained = measure_sustained(lib, dev, args.sustained_s) if rank == 0 else None
    pre_ms, pre_n, _ = _read_slot(lib, 0)
    f_ms, f_n, _ = _read_slot(lib, 1)
    t_ms, t_n, _ = _read_slot(lib, 2)
    call_ms, call_n, _ = _read_slot(lib, 3)

    if rank == 0:
        total = n_coal * args.steps * world
        launches = f_n + t_n
        launch_s = (f_ms + t_ms) * 1e-3                            # all chain launches of the profiled pass (2 per step)
        # what the chain kernel (feature-STN and trunk instantiations: same shape) executes per launch: each coalition's
        # DISTINCT rows (kept points + the centre), in 32-row MFMA tiles, 143 360 MAC per row
        rows = np.concatenate([rows_of(k.cpu().numpy().view(np.uint64), sizes, R) for k in keeps])
        rows32 = (rows + 31) // 32 * 32
        # layer 3 (91 % of a row's MACs) runs on the bf16 matrix pipe, float32-exact (six bf16 products per float32 product), in
        # 32-row tiles; layer 1 on the fp32 MFMA, layer 2 as bf16x3 too.  (IQ_BENCH_FP32_L3=1 / tuning key 5 = 54: layer 3 on the fp32 MFMA too, with
        # 16-row tail tiles - round 3's kernel.)
        bf3 = os.environ.get("IQ_BENCH_FP32_L3") != "1"
        rows_l3 = rows32 if bf3 else np.where((rows - 1) % 32 < 16, (rows + 15) // 16 * 16, rows32)
        per_launch = 2.0 / len(keeps)                              # FLOP per MAC, averaged over the profiled steps
        l12_flop = per_launch * CHAIN_MAC_L12 * float(rows32.sum())
        l3_flop = per_launch * CHAIN_MAC_L3 * float(rows_l3.sum())
        executed_flop = l12_flop + l3_flop
        useful_flop = per_launch * CHAIN_MAC_PER_ROW * float(rows.sum())
        algorithmic_flop = 2.0 * CHAIN_MAC_PER_ROW * NUM_POINTS * n_coal          # the dense reference layers: all 1024 rows
        avg_launch_s = launch_s / max(launches, 1)
        tf = lambda flop: flop / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0  # noqa: E731
        achieved = tf(executed_flop)
        # the matrix pipe's minimum time for what the kernel issues: fp32 MFMAs at 157.3 TF, bf16 MFMAs at 16 x that, six per product
        if bf3:   # layer 1 (the 64 x 64 transform) on the fp32 MFMA, layers 2-3 as six bf16 products per float32 product
            f32_flop = l12_flop * CHAIN_MAC_L1 / CHAIN_MAC_L12
            t_min = f32_flop / (PEAK_F32_MFMA_TFLOPS * 1e12) + BF3_PRODUCTS * (executed_flop - f32_flop) / (PEAK_BF16_MFMA_TFLOPS * 1e12)
        else:
            t_min = executed_flop / (PEAK_F32_MFMA_TFLOPS * 1e12)
        peak_mix = executed_flop / t_min / 1e12
        step_flops = lib.iq_pointnet_flops_per_coalition(NUM_POINTS) * n_coal
        traffic, traffic_detail = (None, "not measured (N > 1 or --traffic 0)")
        if world == 1 and args.traffic and not force_dist:
            traffic, traffic_detail = measure_traffic()
        out = {
            "metric": "coalitions/sec (masked forward passes/sec), PointNet 1024-pt ModelNet10",
            "value": total / elapsed, "unit": "coalitions/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "PointNet ModelNet10-style Shapley, %d regions x %d permutations per cloud "
                                   "(%d coalitions per step per GPU), BASELINE configs[1]; permutations sampled and masked on the "
                                   "device inside the timed step" % (R, S, n_coal),
                       "num_points": NUM_POINTS, "num_regions": R, "permutations": S,
                       "parallelism": "clouds sharded over %d GPU(s), one all_gather_into_tensor of the logits per step%s"
                                      % (world, " (forced single-rank RCCL group)" if force_dist and world == 1 else "")},
            "timing": {"protocol": "%d regions of >= %.1f s, each %d block(s) of exactly %d steps; barrier + synchronize around every block, "
                                   "MAX over ranks; value / ms_per_step = the median block; HIP-event profiler off"
                                   % (len(regions_s), args.min_region_s, blocks, args.steps),
                       "blocks_per_region": blocks, "region_s": float(np.median(regions_s)), "regions_s": regions_s,
                       "blocks_s": blocks_s, "block_spread": float((max(blocks_s) - min(blocks_s)) / elapsed)},
            "roofline": {"bound": "mfma", "kernel": "pn_chain_kernel<fstn|trunk>", "achieved": achieved,
                         "peak": peak_mix, "unit": "TFLOP/s", "frac": achieved / peak_mix,
                         "frac_basis": "executed",
                         "peak_basis": ("float32-equivalent TFLOP/s of the matrix pipe for this kernel's instruction mix: layer 1 (3 % of the "
                                        "MACs) on v_mfma_f32_32x32x2_f32 at 157.3 TF, layers 2-3 as six exact bf16 products per float32 product "
                                        "on v_mfma_f32_32x32x16_bf16 at 16 x 157.3 / 6 = 419 TF") if bf3 else "dense fp32 MFMA peak",
                         "layers_2_3_on": "bf16 matrix pipe, three-term split, float32-exact" if bf3 else "fp32 MFMA",
                         "frac_of_fp32_mfma_peak": achieved / PEAK_F32_MFMA_TFLOPS,
                         "frac_of_sustained_bf16_ceiling": (achieved / (peak_mix * SUSTAINED["frac"])) if bf3 and SUSTAINED["frac"] else None,
                         "sustained_bf16_mfma": sustained,
                         "frac_useful": tf(useful_flop) / peak_mix,
                         "frac_algorithmic": tf(algorithmic_flop) / peak_mix,
                         "traffic": traffic, "traffic_detail": traffic_detail,
                         "avg_launch_ms": avg_launch_s * 1e3, "launches": launches,
                         "executed_flop_per_launch": executed_flop,
                         "useful_flop_per_launch": useful_flop,
                         "algorithmic_flop_per_launch": algorithmic_flop,
                         "mean_rows_per_coalition": {"distinct": float(rows.mean()), "in_32_row_tiles": float(rows32.mean()),
                                                     "layer3_tiles_16_row_tail": float(rows_l3.mean()), "dense": NUM_POINTS},
                         "note": "frac_basis executed: achieved = float32 FLOP of the MFMA tiles the kernel issues (32-row tiles) / HIP-event "
                                 "launch time (separate profiled pass); frac = achieved / peak, i.e. the matrix pipe's minimum time for "
                                 "the issued instructions (see peak_basis) / the launch time.  frac_of_fp32_mfma_peak: achieved / 157.3 - "
                                 "above 1 because layers 2-3 no longer run on the fp32 MFMA.  frac_of_sustained_bf16_ceiling: against what a pure "
                                 "register loop of bf16 MFMAs holds under the board's power cap (0.776 of the dense peak at 1.86 GHz, measured: "
                                 "profiles/r04_power_probe.txt) - the kernel draws 1.3 of 1.4 kW.  frac_useful: the coalitions' distinct rows "
                                 "only (no tile padding).  frac_algorithmic: SURVEY 8d's figure, the dense reference layers this kernel "
                                 "implements (every coalition, all 1024 rows): the kernel evaluates each coalition's distinct "
                                 "points only, an exact skip (DESIGN.md 3).  traffic: HBM bytes per launch from rocprofv3 FETCH_SIZE x2 "
                                 "(MI355X_MICROARCH.md: the counter reports half of wide coalesced reads on gfx950) + WRITE_SIZE of this "
                                 "same script, measured now",
                         "step_tflops_algorithmic": step_flops / (elapsed / args.steps) / 1e12,
                         "prepool_ms_per_launch": pre_ms / max(pre_n, 1), "call_ms": call_ms / max(call_n, 1)},
        }
        if world == 1 and args.other_models:
            out["other_models"] = other_models(dev, lib, R)
        if world == 1 and args.eager_baseline:
            try:
                out["gpu_eager_baseline"] = gpu_eager_baseline(R)
            except Exception as e:  # noqa: BLE001
                out["gpu_eager_baseline"] = {"error": repr(e)[:300]}
        if world == 1 and args.cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(R, args.cpu_perms)
        return out
    return None


if __name__ == "__main__":
    main()
