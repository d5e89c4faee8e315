#!/usr/bin/env python3
"""Time iq_sample_permutations / iq_prefix_keep_masks stand-alone (HIP events, back-to-back launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from interpret_quality_amd import hip_ops
dev = torch.device("cuda:0")
np.random.seed(1)
st = hip_ops.mt_state_to_device(dev)
for s, r in ((1000, 32), (100, 32), (30000, 30), (1000, 64), (1000, 8)):
    for _ in range(3):
        o = hip_ops.sample_permutations(st, s, r)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        o = hip_ops.sample_permutations(st, s, r)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / n
    e0.record()
    for _ in range(n):
        k = hip_ops.prefix_keep_masks(o)
    e1.record(); torch.cuda.synchronize()
    print("S=%d R=%d: sample %.3f ms, prefix masks %.3f ms" % (s, r, t, e0.elapsed_time(e1) / n))
