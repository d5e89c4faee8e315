#!/usr/bin/env python3
"""Coalitions/s of the Shapley path for the models that consume materialised masked clouds
(BASELINE.json configs[2..]: PointNet++ ...).  Not the headline bench (bench.py); used for tuning and
for the numbers quoted in DESIGN.md.

    python tools/bench_models.py --model pointnet2 [--perms 100] [--batch 10] [--steps 5]
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interpret_quality_amd import _lib, final_common, hip_ops, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="pointnet2")
ap.add_argument("--perms", type=int, default=100)
ap.add_argument("--batch", type=int, default=20, help="permutations per forward batch (config.py knob)")
ap.add_argument("--regions", type=int, default=32)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--mode", default="shapley", choices=["shapley", "interaction"])
ap.add_argument("--pairs", type=int, default=30)
ap.add_argument("--contexts", type=int, default=100)
ap.add_argument("--dedup", action="store_true", help="let the drivers evaluate each distinct coalition once (default: every row "
                "is a forward pass, so the figure is kernel throughput)")
ap.add_argument("--tune", default="", help="iq_set_tuning pairs, e.g. 3=1 (no LDS GEMM)")
ap.add_argument("--dense", action="store_true", help="materialise the masked clouds even if the model has a coalition path")
ap.add_argument("--morton", action="store_true", help="experiment: hand the model the cloud with its points sorted along a Morton curve")
a = ap.parse_args()

dev = torch.device("cuda:0")
lib = _lib.load()
if not a.dedup:   # identity "dedup": every coalition row is evaluated
    final_common.distinct_coalitions = lambda k: (np.asarray(k, dtype=np.uint64), np.arange(len(k)))
for kv in filter(None, a.tune.split(",")):
    k, v = kv.split("=")
    lib.iq_set_tuning(int(k), int(v))
if a.model == "pointnet2":
    from interpret_quality_amd.pointnet2 import PointNet2ClsMsg
    model = PointNet2ClsMsg(None)
    model.load_state_dict(synth.to_torch(synth.pointnet2_state_dict(0)))
elif a.model in ("dgcnn", "gcnn"):
    from interpret_quality_amd.dgcnn import DGCNN_cls, GCNN_cls
    model = (DGCNN_cls if a.model == "dgcnn" else GCNN_cls)(argparse.Namespace(dataset="modelnet10", k=20))
    model.load_state_dict(synth.to_torch(synth.dgcnn_state_dict(0)))
elif a.model == "pointconv":
    from interpret_quality_amd.pointconv import PointConvDensityClsSsg
    model = PointConvDensityClsSsg(None)
    model.load_state_dict(synth.to_torch(synth.pointconv_state_dict(0)))
else:
    raise SystemExit("unknown model")
model = model.to(dev).eval()
if a.dense and hasattr(model, "coalition_logits"):
    class DenseOnly:                       # hides coalition_logits: callers fall back to mask kernel + forward_points
        def __init__(self, m):
            self.forward_points, self.m = m.forward_points, m

        def __call__(self, x):
            return self.m(x)
    model = DenseOnly(model)
pts, label = synth.make_cloud(0)
if a.morton:
    q = ((pts - pts.min(0)) / (pts.max(0) - pts.min(0) + 1e-9) * 1023).astype(np.int64)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        return (v | (v << 2)) & 0x09249249
    pts = np.ascontiguousarray(pts[np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2), kind="stable")])
data = torch.from_numpy(pts).unsqueeze(0).to(dev)
lbl = torch.tensor([label], device=dev)
R, S = a.regions, a.perms
region_id = hip_ops.region_assign(data[0].contiguous(), hip_ops.fps(data, R)[0].contiguous()).cpu().numpy()
orders = synth.make_orders(S, R, seed=1)
args = argparse.Namespace(model=a.model, softmax_type="modified", num_points=1024, num_regions=R, num_samples=S,
                          shapley_batch_size=a.batch, verbose=False)
if a.mode == "interaction":
    # BASELINE configs[3] shape: pairs x contexts x 4 masked clouds (ratio 0.5 -> m = 15 of the 30 other regions)
    from interpret_quality_amd import interaction
    rng = np.random.default_rng(0)
    all_pairs = np.array([[i, j] for i in range(R) for j in range(R) if j > i])
    pairs = all_pairs[rng.choice(len(all_pairs), size=a.pairs, replace=False)]
    ctx = np.stack([np.stack([rng.choice([r for r in range(R) if r not in pr], 15, replace=False) for _ in range(a.contexts)])
                    for pr in pairs])
    args.interaction_batch_size = a.batch
    import contextlib, io

    def run():
        with contextlib.redirect_stdout(io.StringIO()):
            return interaction.compute_order_interaction_logits(model, data, region_id, pairs, ctx, args)
    n_per_step = a.pairs * a.contexts * 4
else:
    def run():
        return final_common.shap_sampling_all_regions_batch(model, data, lbl, region_id, orders, args)
    n_per_step = S * (R + 1)


def read(slot):
    ms, n = ctypes.c_double(0), ctypes.c_int(0)
    lib.iq_profile_read(slot, ctypes.byref(ms), ctypes.byref(n))
    return ms.value


run()
torch.cuda.synchronize()
lib.iq_profile_enable(1)
t0 = time.perf_counter()
for _ in range(a.steps):
    run()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
lib.iq_profile_enable(0)
n = n_per_step * a.steps
sa1, sa2, sa3, call = read(0), read(1), read(2), read(3)
print("%s: %d coalitions in %.3f s = %.0f coalitions/s | per step: slot0 (sa1 | kNN) %.1f ms, slot1 (sa2 | EdgeConv) %.1f ms, slot2 (sa3 | conv5+pool) %.1f ms, "
      "whole forward calls %.1f ms of %.1f ms" % (a.model, n, dt, n / dt, sa1 / a.steps, sa2 / a.steps, sa3 / a.steps,
                                                   call / a.steps, dt / a.steps * 1e3))
if "4=3" in a.tune:
    cnt = (ctypes.c_ulonglong * 8)()
    _lib.check(lib.iq_debug_knn_counters(cnt), "iq_debug_knn_counters")
    print("kNN counters: rounds %d, busy lanes %d, waves %d (32 queries each) | flagged %d, re-ranked %d (their 21 candidates), of which zero-gap "
          "rankings of all rows %d" % (cnt[0], cnt[1], cnt[2], cnt[3], cnt[4], cnt[6]))
