#!/usr/bin/env python3
"""The HBM-/L2-bound streaming kernels north_star names (coalition masking, PointNet++ pair-table gather, DGCNN EdgeConv
gather) on bench-size inputs, for rocprofv3 counter passes (tools/pmc_stream.sh -> profiles/r02_stream_kernels.csv).
Prints the algorithmic bytes each launch moves.

    python tools/stream_kernels.py [--reps 5]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interpret_quality_amd import final_common, hip_ops, interaction, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
R, S = 32, 1000
pts, label = synth.make_cloud(0)
data = torch.from_numpy(pts).unsqueeze(0).to(dev)
rid = hip_ops.region_assign(data[0].contiguous(), hip_ops.fps(data, R)[0].contiguous())
orders = synth.make_orders(S, R, seed=1)
center = torch.mean(data, dim=1).reshape(3).contiguous()
od = hip_ops.as_i32(orders, dev)
# 1. mask_rows_kernel: 33 000 coalitions x 12 288 B written = 405.5 MB per launch (the source cloud is L2-resident)
for layout in (False, True):
    for _ in range(a.reps):
        out = hip_ops.mask_shapley(data[0].contiguous(), rid, od, center, channel_first=layout)
torch.cuda.synchronize()
print("mask_rows_kernel: %d coalitions x 12288 B = %.1f MB written per launch" % (S * (R + 1), S * (R + 1) * 12288 / 1e6))
del out
# 2. the models' gather kernels on the BASELINE shapes (PointNet++ Shapley 3300; DGCNN interaction 30 x 100 x 4)
import argparse as _ap  # noqa: E402
from interpret_quality_amd.dgcnn import DGCNN_cls  # noqa: E402
from interpret_quality_amd.pointnet2 import PointNet2ClsMsg  # noqa: E402
final_common.distinct_coalitions = lambda k: (np.asarray(k, dtype=np.uint64), np.arange(len(k)))
lbl = torch.tensor([label], device=dev)
region_id = rid.cpu().numpy()
m = PointNet2ClsMsg(None)
m.load_state_dict(synth.to_torch(synth.pointnet2_state_dict(0)))
m = m.to(dev).eval()
args = _ap.Namespace(model="pointnet2", softmax_type="modified", num_points=1024, num_regions=R, num_samples=100, shapley_batch_size=20,
                     verbose=False)
for _ in range(2):
    final_common.shap_sampling_all_regions_batch(m, data, lbl, region_id, orders[:100], args)
torch.cuda.synchronize()
print("pt_gather_kernel: PointNet++ sa1 from pair tables, 3300 coalitions per step (3 scales)")
del m
m = DGCNN_cls(_ap.Namespace(dataset="modelnet10", k=20))
m.load_state_dict(synth.to_torch(synth.dgcnn_state_dict(0)))
m = m.to(dev).eval()
rng = np.random.default_rng(0)
allp = np.array([[i, j] for i in range(R) for j in range(R) if j > i])
pairs = allp[rng.choice(len(allp), size=30, replace=False)]
ctx = np.stack([np.stack([rng.choice([r for r in range(R) if r not in pr], 15, replace=False) for _ in range(100)]) for pr in pairs])
args = _ap.Namespace(model="dgcnn", softmax_type="modified", num_regions=R, interaction_batch_size=100)
import contextlib, io  # noqa: E402
for _ in range(2):
    with contextlib.redirect_stdout(io.StringIO()):
        interaction.compute_order_interaction_logits(m, data, region_id, pairs, ctx, args)
torch.cuda.synchronize()
print("gather_max_kernel: DGCNN EdgeConv neighbourhood max, 12 000 coalitions per step (4 layers); per output row 20 rows of Co floats "
      "gathered from L2 / MALL + 1 row of Q read, Co floats written")
