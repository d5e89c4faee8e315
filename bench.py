#!/usr/bin/env python3
"""Headline benchmark: coalitions/sec (masked forward passes/sec), PointNet, 1024-point clouds,
32 regions x 1000 sampled permutations per cloud (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = the whole Shapley hot path for one synthetic cloud: 1000 permutations x 33 prefix
coalitions = 33 000 masked forward passes, reward, per-region float64 accumulation.  Inputs
(cloud, region ids, permutations) are resident in HBM before the timed region.  With N > 1 every
rank works on its own cloud (weak scaling; independent objects) and one RCCL all-gather per step
brings the per-coalition logits to every rank, as the artefact writer on rank 0 needs them
(SURVEY.md §8e).  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

NUM_POINTS, NUM_REGIONS, NUM_PERMS = 1024, 32, 1000
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 MFMA peak (no xf32/TF32 on gfx950)
SUSTAINED_F32_MFMA_TFLOPS = 138.7  # bare MFMA loop on random operands, this chip: profiles/r01_mfma_shape_microbench.txt


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--perms", type=int, default=NUM_PERMS, help="permutations per cloud per step")
    ap.add_argument("--regions", type=int, default=NUM_REGIONS)
    ap.add_argument("--cpu-baseline", type=int, default=1, help="time the CPU oracle on rank 0 at N=1")
    ap.add_argument("--cpu-perms", type=int, default=40, help="permutations of the bounded CPU sample")
    ap.add_argument("--other-models", type=int, default=1, help="after the timed region (N=1 only): coalitions/s of the other "
                    "model families on BASELINE configs[2..3] shapes, reported under 'other_models'")
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


def other_models(dev, regions=32):
    """Not the headline: coalitions/s of the other model families through the same C ABI, each on the shape BASELINE.json
    names for it (PointNet++ / PointConv: Shapley, 100 permutations = 3300 coalitions per step; DGCNN / GCNN: interaction,
    30 pairs x 100 contexts x 4 = 12 000 coalitions per step).  Every coalition row is a forward pass (no driver-level
    de-duplication).  Failures are reported, never raised: the headline line must still be printed."""
    import argparse as ap
    from interpret_quality_amd import final_common, hip_ops, interaction, synth
    from interpret_quality_amd.dgcnn import DGCNN_cls, GCNN_cls
    from interpret_quality_amd.pointconv import PointConvDensityClsSsg
    from interpret_quality_amd.pointnet2 import PointNet2ClsMsg
    import contextlib
    import io
    saved = final_common.distinct_coalitions
    final_common.distinct_coalitions = lambda k: (np.asarray(k, dtype=np.uint64), np.arange(len(k)))
    out = {}
    try:
        pts, label = synth.make_cloud(0)
        data = torch.from_numpy(pts).unsqueeze(0).to(dev)
        lbl = torch.tensor([label], device=dev)
        region_id = hip_ops.region_assign(data[0].contiguous(), hip_ops.fps(data, regions)[0].contiguous()).cpu().numpy()
        orders = synth.make_orders(100, regions, seed=1)
        rng = np.random.default_rng(0)
        all_pairs = np.array([[i, j] for i in range(regions) for j in range(regions) if j > i])
        pairs = all_pairs[rng.choice(len(all_pairs), size=30, replace=False)]
        ctx = np.stack([np.stack([rng.choice([r for r in range(regions) if r not in pr], 15, replace=False) for _ in range(100)])
                        for pr in pairs])
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
                    run = lambda: final_common.shap_sampling_all_regions_batch(m, data, lbl, region_id, orders, a)
                    n = 100 * (regions + 1)
                else:
                    def run():
                        with contextlib.redirect_stdout(io.StringIO()):
                            return interaction.compute_order_interaction_logits(m, data, region_id, pairs, ctx, a)
                    n = 30 * 100 * 4
                run()
                torch.cuda.synchronize()
                steps = 3
                t0 = time.perf_counter()
                for _ in range(steps):
                    run()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                out[name] = {"value": n * steps / dt, "unit": "coalitions/s", "steps": steps,
                             "workload": "%s, %d regions, %d coalitions per step" % (mode, regions, n)}
                del m
            except Exception as e:  # noqa: BLE001
                out[name] = {"error": repr(e)[:200]}
    finally:
        final_common.distinct_coalitions = saved
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # IQ_BENCH_REHEARSAL=1: rehearse the N > 1 code path on a one-GPU box (all ranks on cuda:0, gloo
    # collectives).  Never set by the driver; the numbers of a rehearsal mean nothing.
    rehearsal = os.environ.get("IQ_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo" if rehearsal else "nccl", rank=rank, world_size=world)  # nccl = RCCL on ROCm

    from interpret_quality_amd import _lib, final_common, hip_ops, synth
    from interpret_quality_amd.pointnet import PointNetCls

    lib = _lib.load()
    R, S = args.regions, args.perms
    model = PointNetCls(None)
    model.load_state_dict(synth.to_torch(synth.pointnet_state_dict(0)))
    model = model.to(dev).eval()

    # ---- inputs, resident in HBM before timing (cloud index = rank: independent objects) ------
    pts, label = synth.make_cloud(rank)
    data = torch.from_numpy(pts).unsqueeze(0).to(dev)
    fps_idx = hip_ops.fps(data, R)[0].contiguous()
    region_id = hip_ops.region_assign(data[0].contiguous(), fps_idx).reshape(1, -1)
    orders_np = synth.make_orders(S, R, seed=1)
    orders = hip_ops.as_i32(orders_np, dev)
    keep = hip_ops.masks_to_tensor(final_common.prefix_keep_masks(orders_np, R), dev)
    center = torch.mean(data, dim=1).contiguous()
    n_coal = S * (R + 1)
    gathered = [torch.empty((n_coal, 10), dtype=torch.float32, device=dev) for _ in range(world)] if world > 1 else None

    def step():
        logits = model.coalition_logits(data, center, region_id, keep, None, num_regions=R)
        v = hip_ops.reward(logits, label, True)
        phi_sum, _, _ = hip_ops.shapley_accum(v, orders)
        if world > 1:
            dist.all_gather(gathered, logits)
        return phi_sum, logits

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    lib.iq_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        phi_sum, logits = step()
    fence()
    elapsed = time.perf_counter() - t0
    lib.iq_profile_enable(0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- HIP-event durations of the dominant kernel over the timed region ----------------------
    def slot(i):
        ms, n = ctypes.c_double(0), ctypes.c_int(0)
        _lib.check(lib.iq_profile_read(i, ctypes.byref(ms), ctypes.byref(n)), "iq_profile_read")
        return ms.value, n.value
    pre_ms, pre_n = slot(0)
    f_ms, f_n = slot(1)
    t_ms, t_n = slot(2)
    call_ms, call_n = slot(3)

    # sanity: efficiency axiom on the last step (sum phi = v(N) - v(empty) for every permutation)
    v = hip_ops.reward(logits, label, True)
    eff = abs(float(phi_sum.sum().item()) / S - float((v[R] - v[0]).item()))
    assert eff < 1e-3, "efficiency check failed: %g" % eff

    if rank == 0:
        total = n_coal * args.steps * world
        # the chain kernel (feature-STN and trunk instantiations: same shape, 143 360 MAC per point)
        flop_per_launch = 2.0 * 143360.0 * NUM_POINTS * n_coal
        avg_launch_s = (f_ms + t_ms) / max(f_n + t_n, 1) * 1e-3
        achieved = flop_per_launch / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0
        # what the kernel actually executes: each coalition's DISTINCT rows (kept points + the centre), in 32-row MFMA tiles
        sizes = np.bincount(region_id.cpu().numpy().reshape(-1), minlength=R).astype(np.int64)
        bits = ((final_common.prefix_keep_masks(orders_np, R)[:, None] >> np.arange(R, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.int64)
        kept = bits @ sizes
        rows = kept + (kept < NUM_POINTS)
        rows32 = (rows + 31) // 32 * 32
        executed = 2.0 * 143360.0 * float(rows32.sum()) / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0
        traffic = None
        tpath = os.path.join(REPO, "profiles", "traffic_r01.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        step_flops = lib.iq_pointnet_flops_per_coalition(NUM_POINTS) * n_coal
        out = {
            "metric": "coalitions/sec (masked forward passes/sec), PointNet 1024-pt ModelNet10",
            "value": total / elapsed, "unit": "coalitions/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "PointNet ModelNet10-style Shapley, %d regions x %d permutations per cloud "
                                   "(%d coalitions per step per GPU), BASELINE configs[1]" % (R, S, n_coal),
                       "num_points": NUM_POINTS, "num_regions": R, "permutations": S,
                       "parallelism": "clouds sharded over %d GPU(s), all-gather of logits" % world},
            "roofline": {"bound": "mfma", "kernel": "pn_chain_kernel<fstn|trunk>", "achieved": achieved,
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                         "traffic": traffic, "avg_launch_ms": avg_launch_s * 1e3, "launches": f_n + t_n,
                         "algorithmic_flop_per_launch": flop_per_launch,
                         "executed": executed, "executed_frac_of_peak": executed / PEAK_F32_MFMA_TFLOPS,
                         "sustained_mfma_random_operands": SUSTAINED_F32_MFMA_TFLOPS,
                         "executed_frac_of_sustained": executed / SUSTAINED_F32_MFMA_TFLOPS,
                         "note": "achieved/frac: algorithmic FLOP of the dense reference layers this kernel implements; the "
                                 "kernel evaluates each coalition's distinct points only (exact), so frac can exceed 1. "
                                 "executed: FLOP of the 32-row MFMA tiles it really issues",
                         "step_tflops_algorithmic": step_flops / (elapsed / args.steps) / 1e12,
                         "prepool_ms_per_launch": pre_ms / max(pre_n, 1), "call_ms": call_ms / max(call_n, 1)},
        }
        if world == 1 and args.other_models:
            out["other_models"] = other_models(dev, R)
        if world == 1 and args.cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(R, args.cpu_perms)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
