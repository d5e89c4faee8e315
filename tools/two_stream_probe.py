#!/usr/bin/env python3
"""Experiment: do two coalition batches of one model overlap when they run on two HIP streams (each with its own engine /
workspace)?  The MFMA-bound kernels of one batch (conv5, grouped MLPs) could run next to the VALU- / LDS-bound kernels of the
other (kNN selection, EdgeConv gather).  Prints coalitions/s for: one stream, two streams.

    python tools/two_stream_probe.py --model dgcnn [--batch 4096] [--reps 4]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interpret_quality_amd import hip_ops, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="dgcnn")
ap.add_argument("--batch", type=int, default=4096, help="coalitions per call")
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--regions", type=int, default=32)
a = ap.parse_args()
dev = torch.device("cuda:0")


def build():
    if a.model == "pointnet2":
        from interpret_quality_amd.pointnet2 import PointNet2ClsMsg
        m, sd = PointNet2ClsMsg(None), synth.pointnet2_state_dict(0)
    elif a.model in ("dgcnn", "gcnn"):
        from interpret_quality_amd.dgcnn import DGCNN_cls, GCNN_cls
        m, sd = (DGCNN_cls if a.model == "dgcnn" else GCNN_cls)(argparse.Namespace(dataset="modelnet10", k=20)), synth.dgcnn_state_dict(0)
    elif a.model == "pointconv":
        from interpret_quality_amd.pointconv import PointConvDensityClsSsg
        m, sd = PointConvDensityClsSsg(None), synth.pointconv_state_dict(0)
    else:
        from interpret_quality_amd.pointnet import PointNetCls
        m, sd = PointNetCls(None), synth.pointnet_state_dict(0)
    m.load_state_dict(synth.to_torch(sd))
    return m.to(dev).eval()


models = [build(), build()]
pts, _ = synth.make_cloud(0)
data = torch.from_numpy(pts).unsqueeze(0).to(dev)
R = a.regions
region_id = hip_ops.region_assign(data[0].contiguous(), hip_ops.fps(data, R)[0].contiguous()).reshape(1, -1)
center = torch.mean(data, dim=1).contiguous()
rng = np.random.default_rng(0)
keeps = []
for _ in range(2):   # interaction-like coalitions: 17 of 32 regions kept
    k = np.zeros(a.batch, dtype=np.uint64)
    for i in range(a.batch):
        for r in rng.choice(R, 17, replace=False):
            k[i] |= np.uint64(1) << np.uint64(r)
    keeps.append(torch.from_numpy(k.view(np.int64)).to(dev))


def call(m, keep):
    return m.coalition_logits(data, center, region_id, keep, None, num_regions=R, validate=False)


ref = [call(models[0], keeps[0]), call(models[1], keeps[1])]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    call(models[0], keeps[0])
    call(models[1], keeps[1])
torch.cuda.synchronize()
t_seq = (time.perf_counter() - t0) / a.reps
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
out = [None, None]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            out[i] = call(models[i], keeps[i])
torch.cuda.synchronize()
t_two = (time.perf_counter() - t0) / a.reps
same = all(torch.equal(out[i], ref[i]) for i in range(2))
n = 2 * a.batch
print("%s: 2 x %d coalitions | one stream %.1f ms (%.0f coalitions/s) | two streams %.1f ms (%.0f coalitions/s) | logits identical: %s"
      % (a.model, a.batch, t_seq * 1e3, n / t_seq, t_two * 1e3, n / t_two, same))
