#!/usr/bin/env python3
"""A/B the co-compiled chain-kernel variants in ONE process, interleaved rounds (guide rule 24).
usage: python tools/ab_chain.py [--key 0] [--values 0,1] [--rounds 7] [--perms 1000]"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interpret_quality_amd import _lib, final_common, hip_ops, synth  # noqa: E402
from interpret_quality_amd.pointnet import PointNetCls  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--key", type=int, default=0)
ap.add_argument("--values", default="0,1")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--stamps", type=int, default=0, help="run the diagnostic STAMP build once and print phase shares")
ap.add_argument("--fixed", default="", help="key=value,... set once before the runs")
ap.add_argument("--perms", type=int, default=1000)
ap.add_argument("--regions", type=int, default=32)
ap.add_argument("--dense", type=int, default=0, help="B coalitions that keep every region (uniform 1024-row items)")
args = ap.parse_args()
values = [int(v) for v in args.values.split(",")]

lib = _lib.load()
dev = torch.device("cuda:0")
model = PointNetCls(None)
model.load_state_dict(synth.to_torch(synth.pointnet_state_dict(0)))
model = model.to(dev).eval()
pts, label = synth.make_cloud(0)
data = torch.from_numpy(pts).unsqueeze(0).to(dev)
R, S = args.regions, args.perms
region_id = hip_ops.region_assign(data[0].contiguous(), hip_ops.fps(data, R)[0].contiguous()).reshape(1, -1)
orders = synth.make_orders(S, R, seed=1)
keep = hip_ops.masks_to_tensor(final_common.prefix_keep_masks(orders, R), dev)
center = torch.mean(data, dim=1).contiguous()
if args.dense:
    keep = hip_ops.masks_to_tensor(np.full(args.dense, (1 << R) - 1, dtype=np.uint64), dev)


def read(slot):
    ms, n = ctypes.c_double(0), ctypes.c_int(0)
    lib.iq_profile_read(slot, ctypes.byref(ms), ctypes.byref(n))
    return ms.value / max(n.value, 1)


for kv in [x for x in args.fixed.split(",") if x]:
    k, v = kv.split("=")
    lib.iq_set_tuning(int(k), int(v))
if args.stamps:
    lib.iq_set_tuning(0, 2)
    model.coalition_logits(data, center, region_id, keep, None, num_regions=R)
    torch.cuda.synchronize()
    lib.iq_debug_stamps(1, None)
    model.coalition_logits(data, center, region_id, keep, None, num_regions=R)
    buf = (ctypes.c_ulonglong * 8)()
    lib.iq_debug_stamps(0, buf)
    names = ["stage0a (transform)", "wait barrier 1", "stage0b + L1 weight issue", "wait barriers (L1/L2/L3 entry)",
             "L1 compute + L2 weight issue", "L2 compute", "L3", "-"]
    tot = float(sum(buf))
    for n, v in zip(names, buf):
        print("  %-34s %6.2f %%" % (n, 100.0 * v / tot))
    sys.exit(0)
ref = None
times = {v: [] for v in values}
for rnd in range(args.rounds + 1):
    for v in values:
        lib.iq_set_tuning(args.key, v)
        lib.iq_profile_enable(1)
        logits = model.coalition_logits(data, center, region_id, keep, None, num_regions=R)
        torch.cuda.synchronize()
        lib.iq_profile_enable(0)
        f, t, c = read(1), read(2), read(3)
        read(0)
        if ref is None:
            ref = logits.clone()
        same = torch.equal(ref, logits)
        if rnd > 0:
            times[v].append((f, t, c))
        print("round %d variant %d: fstn %.3f ms trunk %.3f ms call %.3f ms bitwise_equal_to_first=%s" % (rnd, v, f, t, c, same), flush=True)
for v in values:
    a = np.array(times[v])
    nb = keep.numel()
    print("variant %d: median fstn %.3f trunk %.3f call %.3f | min call %.3f -> %.0f coalitions/s" % (
        v, np.median(a[:, 0]), np.median(a[:, 1]), np.median(a[:, 2]), a[:, 2].min(), nb / np.median(a[:, 2]) * 1e3))
    if args.dense:
        fl = 2.0 * 143360 * 1024 * nb
        print("   dense: executed = algorithmic FLOP; fstn %.1f TF/s trunk %.1f TF/s" % (
            fl / np.median(a[:, 0]) / 1e9, fl / np.median(a[:, 1]) / 1e9))
