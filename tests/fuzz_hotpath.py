#!/usr/bin/env python3
"""Randomised probe of the hot path's boundary (GPU): shap_sampling_all_regions_batch, compute_order_interaction_logits and the
region assignment behind them on random cloud sizes (8-4096), region counts (1-64), permutation counts / batch sizes, both softmax
types - the HIP path (interpret_quality_amd.final_common / interaction) against the CPU oracle's restatement of the reference loop.

    python tests/fuzz_hotpath.py [seed] [seconds]

PointNet (the oracle's restatement of the loop is PointNet's; the other families' coalition paths: tests/fuzz_sizes.py).  Prints every case that
raises or disagrees (> 1e-4 of the largest logit / Shapley value) and a final count.  Test infrastructure (imports oracle/); not
collected by pytest."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from interpret_quality_amd import final_common, hip_ops, synth
from interpret_quality_amd.pointnet import PointNetCls
from interpret_quality_amd.pointnet2 import PointNet2ClsMsg
from oracle import ref_cpu as O

d = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
models = {}


def get(name):
    if name not in models:
        cls, sdf, orc = {"pointnet": (PointNetCls, synth.pointnet_state_dict, O.PointNetOracle),
                         "pointnet2": (PointNet2ClsMsg, synth.pointnet2_state_dict, O.PointNet2Oracle)}[name]
        sd = synth.to_torch(sdf(0))
        m = cls(None)
        m.load_state_dict(sd)
        models[name] = (m.to(d).eval(), orc(sd))
    return models[name]


t0, ncase, nbad = time.time(), 0, 0
while time.time() - t0 < budget:
    name = "pointnet"       # (the oracle's loop unpacks PointNet's output tuple: tools/final_common.py:36)
    n = int(rng.integers(8 if name == "pointnet" else 128, 4097 if name == "pointnet" else 400))
    r = int(rng.choice([1, 2, 3, 8, 17, 32, 64]))
    bs = int(rng.choice([1, 2, 3]))
    s = bs * int(rng.integers(1, 4))
    sm = str(rng.choice(["modified", "normal"]))
    desc = "%s N=%d R=%d S=%d bs=%d %s" % (name, n, r, s, bs, sm)
    try:
        model, om = get(name)
        pts, label = synth.make_cloud(int(rng.integers(0, 1000)), num_points=n)
        data_c = torch.from_numpy(pts).unsqueeze(0)
        data = data_c.to(d)
        lbl = torch.tensor([label], device=d)
        fps = hip_ops.fps(data, r)
        want_fps = O.farthest_point_sample(data_c, r)
        region_id = hip_ops.region_assign(data[0].contiguous(), fps[0].contiguous()).cpu().numpy()
        want_rid = np.asarray(O.cal_region_id(data_c, want_fps[0]))
        bad_geom = not np.array_equal(fps.cpu().numpy(), want_fps.numpy()) or (region_id != want_rid).mean() > 0.01   # near-ties only
        orders = np.stack([rng.permutation(r) for _ in range(s)]).astype(np.int64)
        args = argparse.Namespace(model=name, softmax_type=sm, num_points=n, verbose=False, num_regions=r, num_samples=s,
                                  shapley_batch_size=bs)
        phi, logits = final_common.shap_sampling_all_regions_batch(model, data, lbl, want_rid, orders, args)
        o_phi, o_logits = O.shap_sampling_all_regions_batch(om, data_c, torch.tensor([label]), want_rid, orders, s, bs, r, sm)
        e_l = np.abs(logits.cpu().numpy() - o_logits.numpy()).max() / np.abs(o_logits.numpy()).max()
        e_p = np.abs(phi - o_phi).max() / max(np.abs(o_phi).max(), 1e-6)
        # loop C on the same cloud: random pairs, contexts of a random order m (final_point_binary_interaction_logits.py:15-70)
        e_i = 0.0
        if r >= 2:
            from interpret_quality_amd import interaction
            npair, nctx = int(rng.integers(1, 4)), int(rng.integers(1, 5))
            m_order = int(rng.integers(0, r - 1))
            pairs = np.stack([rng.choice(r, size=2, replace=False) for _ in range(npair)]).astype(np.int64)
            ctxs = np.stack([np.stack([rng.choice([x for x in range(r) if x not in pr], size=m_order, replace=False) for _ in range(nctx)])
                             for pr in pairs]).astype(np.int64).reshape(npair, nctx, m_order)
            args.interaction_batch_size = int(rng.integers(1, 4))
            got_i = interaction.compute_order_interaction_logits(model, data, want_rid, pairs, ctxs, args).cpu()
            want_i = O.compute_order_interaction_logits(om, data_c, want_rid, pairs, ctxs, args.interaction_batch_size)
            e_i = float(np.abs(got_i.numpy() - want_i.numpy()).max() / np.abs(want_i.numpy()).max())
            # the reduction ((v0 + v3) - v1) - v2 (final_cal_interactions.py:28-36) on the SAME logits: a difference of nearly equal
            # rewards, each within 2e-6 of the reference's (exp / log differ in the last bits; the reduction itself is bit-exact given v:
            # tests/test_hip_parity.py) - so an absolute bar, which still catches a wrong row order
            iv = interaction.compute_order_interaction(want_i.to(d), lbl, args)
            ow = O.compute_order_interaction(want_i, torch.tensor([label]), sm)
            if np.abs(np.asarray(iv) - ow).max() > 2e-4:
                e_i = max(e_i, 1.0)
        ncase += 1
        if bad_geom or e_l > 1e-4 or e_p > 2e-3 or e_i > 1e-4 or not np.isfinite(phi).all():
            nbad += 1
            print("MISMATCH %s: geometry %s, logits %.2g, phi %.2g (max |phi| %.3g), interaction logits %.2g" % (desc, "BAD" if bad_geom else "ok", e_l, e_p, np.abs(o_phi).max(), e_i), flush=True)
    except Exception as e:   # noqa: BLE001 - a probe: report and go on
        ncase += 1
        print("%s: %s: %s" % (desc, type(e).__name__, str(e)[:160]), flush=True)
print("cases %d, mismatches %d" % (ncase, nbad))
