#!/usr/bin/env python3
"""Does the headline launch give the same bits every time?  33 000 PointNet coalitions (1000 permutations x 33 prefixes, 32 regions)
evaluated over and over for --seconds; every launch's logits are compared bit by bit, on the device, with the first launch's.

    [IQ_LIBPATH=interpret_quality_amd/lib_packed_ab/libiq_hip.so] python tools/chain_repro.py [--seconds 15] [--model pointnet|pointnet2|dgcnn|pointconv]

Why: round 4 blamed packed float32 instructions beside bf16 MFMAs for a few-ulp nondeterminism of ANOTHER
process's kernel; the chain kernel itself held 51 of them beside its own 480 bf16 MFMAs.  If that mechanism were real inside one
kernel, this loop would see it within seconds (it compares 1.3 M floats per launch).  Product code only, nothing from oracle/.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interpret_quality_amd import _lib, final_common, hip_ops, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=15.0)
ap.add_argument("--model", default="pointnet")
ap.add_argument("--perms", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda:0")
_lib.load()
R = 32
if a.model == "pointnet":
    from interpret_quality_amd.pointnet import PointNetCls
    model, sd, S = PointNetCls(None), synth.pointnet_state_dict(0), a.perms or 1000
elif a.model == "pointnet2":
    from interpret_quality_amd.pointnet2 import PointNet2ClsMsg
    model, sd, S = PointNet2ClsMsg(None), synth.pointnet2_state_dict(0), a.perms or 100
elif a.model == "pointconv":
    from interpret_quality_amd.pointconv import PointConvDensityClsSsg
    model, sd, S = PointConvDensityClsSsg(None), synth.pointconv_state_dict(0), a.perms or 100
else:
    from interpret_quality_amd.dgcnn import DGCNN_cls
    model, sd, S = DGCNN_cls(argparse.Namespace(dataset="modelnet10", k=20)), synth.dgcnn_state_dict(0), a.perms or 100
model.load_state_dict(synth.to_torch(sd))
model = model.to(dev).eval()
pts, _ = synth.make_cloud(0)
data = torch.from_numpy(pts).unsqueeze(0).to(dev)
rid = hip_ops.region_assign(data[0].contiguous(), hip_ops.fps(data, R)[0].contiguous()).reshape(1, -1)
keep = hip_ops.masks_to_tensor(final_common.prefix_keep_masks(synth.make_orders(S, R, seed=1), R), dev)
center = data.mean(dim=1).contiguous()
first = model.coalition_logits(data, center, rid, keep, None, num_regions=R).clone()
torch.cuda.synchronize()
launches, bad, worst = 0, 0, 0.0
t0 = time.time()
while time.time() - t0 < a.seconds:
    got = model.coalition_logits(data, center, rid, keep, None, num_regions=R)
    if not torch.equal(got, first):
        bad += 1
        worst = max(worst, (got - first).abs().max().item())
    launches += 1
print("%s (%s): %d launches of %d coalitions, %d differ from the first (max |d| %.3g)" % (
    a.model, os.path.relpath(_lib.lib_path()), launches, keep.numel(), bad, worst))
sys.exit(1 if bad else 0)
