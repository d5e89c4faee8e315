"""Baseline leg of bench.py - checker-side code like the rest of oracle/, never on the product path: the reference's Shapley loop restated with stock
PyTorch-ROCm eager ops on the MI355X - the stand-in for the unpublished "reference on a GPU"
figure (SURVEY.md §8d baseline (ii)).  Same structure as tools/final_common.py:64-103: bs*R
boolean-index assignments per batch, one batched forward, one host sync per permutation.
bench.py reports it as `gpu_eager_baseline`.

    python oracle/eager_gpu_baseline.py [--perms 100] [--bs 50]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run_eager(perms=100, bs=50, regions=32):
    from interpret_quality_amd import synth
    from oracle import ref_cpu as O
    dev = torch.device("cuda:0")
    sd = {k: v.to(dev) for k, v in synth.to_torch(synth.pointnet_state_dict(0)).items()}
    pts, label = synth.make_cloud(0)
    data = torch.from_numpy(pts).unsqueeze(0).to(dev)
    lbl = torch.tensor([label], device=dev)
    cpu = torch.from_numpy(pts).unsqueeze(0)
    region_id = O.cal_region_id(cpu, O.farthest_point_sample(cpu, regions)[0])
    orders = synth.make_orders(perms, regions, seed=1)
    R = regions

    def run(n_perm):
        center = torch.mean(data, dim=1).squeeze()
        phi = np.zeros((R,))
        with torch.no_grad():
            for it in range(n_perm // bs):
                chunk = orders[it * bs:(it + 1) * bs]
                masked = data.expand((R + 1) * bs, 1024, 3).clone()
                for o, order in enumerate(chunk):                      # the reference's index-put storm
                    for j in range(1, R + 1):
                        masked[(R + 1) * o:(R + 1) * o + j, torch.from_numpy(region_id == order[j - 1]).to(dev), :] = center
                logits, _, _ = O.pointnet_forward(sd, masked.permute(0, 2, 1).contiguous())
                v = O.get_reward(logits, lbl)
                for o, order in enumerate(chunk):
                    vo = v[(R + 1) * o:(R + 1) * (o + 1)]
                    phi[order] += (vo[1:] - vo[:-1]).cpu().numpy()      # one sync per permutation
        torch.cuda.synchronize()
        return phi

    run(bs)
    t0 = time.time()
    run(perms)
    dt = time.time() - t0
    n = perms // bs * bs * (R + 1)
    return {"value": n / dt, "unit": "coalitions/s", "kind": "stock PyTorch-ROCm eager ops on the same GPU (no custom kernels)",
            "device": torch.cuda.get_device_name(0),
            "sample": "tools/final_common.py:64-103 restated with eager torch ops, PointNet, R=%d, %d permutations, bs=%d "
                      "(%d clouds per forward) = %d coalitions in %.2f s" % (R, perms // bs * bs, bs, bs * (R + 1), n, dt)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--perms", type=int, default=100)
    ap.add_argument("--bs", type=int, default=50)  # config.py: pointnet shapley batch
    ap.add_argument("--regions", type=int, default=32)
    a = ap.parse_args()
    print(run_eager(a.perms, a.bs, a.regions))
