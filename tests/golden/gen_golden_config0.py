"""Golden vectors for BASELINE.json configs[0] AT SPEC, from the REFERENCE (imported from /root/reference, never copied):
PointNet on 30 clouds, 8 FPS regions, 64 sampled permutations per cloud, shapley_batch_size 8, CPU PyTorch
(30 x 64 x 9 = 17 280 masked forward passes).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/gen_golden_config0.py      (about 4 minutes, 8 cores)

Per cloud: FPS index, region ids, the permutations (the reference's global NumPy RNG stream continues from cloud to cloud,
as in final_shapley_value.py:110-156), the region Shapley values, every coalition's reward; the logits of the first two
clouds.  Output tests/golden/pointnet_config0.npz (data only)."""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refenv  # noqa: E402

synth = _refenv.setup()   # the reference first on sys.path, the repository root (its `tools/` shims) off it

from tools import final_common as ref_common  # noqa: E402
from tools import final_util as ref_util  # noqa: E402
import final_shapley_value as ref_stage1  # noqa: E402
import final_save_fps as ref_fps  # noqa: E402
from models.pointnet import PointNetCls  # noqa: E402

NUM_CLOUDS, R, S, BS = 30, 8, 64, 8


def main():
    torch.set_num_threads(8)
    model = PointNetCls(argparse.Namespace(dataset="modelnet10", feature_transform=True, model="pointnet"))
    model.load_state_dict(synth.to_torch(synth.pointnet_state_dict(0)))
    model.eval()
    args = argparse.Namespace(model="pointnet", softmax_type="modified", num_points=1024, num_regions=R, num_samples=S,
                              shapley_batch_size=BS, num_samples_save=S)
    ref_util.set_random(1)                       # once: the permutation stream runs on across the clouds
    fps, rid, orders, phi, v, logits01, norm = [], [], [], [], [], [], []
    for ci in range(NUM_CLOUDS):
        pts, label = synth.make_cloud(ci)
        data = torch.from_numpy(pts).unsqueeze(0)
        lbl = torch.tensor([label], dtype=torch.long)
        fps_index = ref_fps.farthest_point_sample(data, R)[0]
        region_id = ref_stage1.cal_region_id(data, fps_index, None, save=False)
        o = ref_stage1.generate_all_orders(None, args, save=False)
        with torch.no_grad():
            center = torch.mean(data, dim=1).squeeze()
            norm.append(ref_stage1.cal_norm_factor(model, data, lbl, center, None, args, save=False))
            p, lg = ref_common.shap_sampling_all_regions_batch(model, data, lbl, region_id, o, args)
            v.append(ref_common.get_reward(lg, lbl, args).numpy())
        fps.append(fps_index.numpy()); rid.append(region_id); orders.append(o); phi.append(p)
        if ci < 2:
            logits01.append(lg.numpy())
        print("cloud %2d: sum(phi) = %.6f  norm_factor = %.6f" % (ci, p.sum(), norm[-1]), flush=True)
    np.savez_compressed(os.path.join(HERE, "pointnet_config0.npz"), num_regions=R, num_samples=S, bs=BS,
                        fps_index=np.stack(fps).astype(np.int16), region_id=np.stack(rid).astype(np.int8),
                        orders=np.stack(orders).astype(np.int8), phi=np.stack(phi), v=np.stack(v).astype(np.float32),
                        norm_factor=np.array(norm, dtype=np.float64), logits_first2=np.stack(logits01))


if __name__ == "__main__":
    main()
