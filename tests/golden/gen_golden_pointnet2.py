"""Golden vectors for PointNet++ MSG from the REFERENCE (imported from /root/reference, never copied).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/gen_golden_pointnet2.py

Outputs tests/golden/pointnet2.npz: ball-query indices for the 6 (radius, K) pairs, FPS indices of
both levels, sa1/sa2 features, logits of raw and masked clouds, Shapley values for a small case."""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refenv  # noqa: E402

synth = _refenv.setup()   # the reference first on sys.path, the repository root (its `tools/` shims) off it

from models import pointnet2 as ref_pn2  # noqa: E402
from tools import final_common as ref_common  # noqa: E402
import final_shapley_value as ref_stage1  # noqa: E402
import final_save_fps as ref_fps  # noqa: E402


SEL = [0, 5, 17]  # all-centre, partially masked, untouched


def main():
    torch.set_num_threads(8)
    model = ref_pn2.PointNet2ClsMsg(argparse.Namespace(dataset="modelnet10"))
    model.load_state_dict(synth.to_torch(synth.pointnet2_state_dict(0)))
    model.eval()
    out = {}
    pts, label = synth.make_cloud(0)
    data = torch.from_numpy(pts).unsqueeze(0)
    lbl = torch.tensor([label])
    num_regions = 8
    fps_index = ref_fps.farthest_point_sample(data, num_regions)[0]
    region_id = ref_stage1.cal_region_id(data, fps_index, None, save=False)
    np.random.seed(1)
    args = argparse.Namespace(model="pointnet2", softmax_type="modified", num_points=1024, num_regions=num_regions,
                              num_samples=2, shapley_batch_size=2, num_samples_save=2)
    orders = ref_stage1.generate_all_orders(None, args, save=False)
    center = torch.mean(data, dim=1).squeeze()
    masked = data.expand((num_regions + 1) * 2, 1024, 3).clone()
    masked = ref_common.mask_data_batch(masked, center, orders, region_id, args)
    x = masked.permute(0, 2, 1).contiguous()  # 18 masked clouds incl. the all-centre one
    with torch.no_grad():
        xyz = x.permute(0, 2, 1)
        fps1 = ref_pn2.farthest_point_sample(xyz, 512)
        new_xyz = ref_pn2.index_points(xyz, fps1)
        for r, k in ((0.1, 16), (0.2, 32), (0.4, 128)):
            out["sa1_ball_r%g" % r] = ref_pn2.query_ball_point(r, k, xyz, new_xyz)[SEL].numpy().astype(np.int16)
        l1_xyz, l1_points = model.sa1(x, None)
        xyz2 = l1_xyz.permute(0, 2, 1)
        fps2 = ref_pn2.farthest_point_sample(xyz2, 128)
        new_xyz2 = ref_pn2.index_points(xyz2, fps2)
        for r, k in ((0.2, 32), (0.4, 64), (0.8, 128)):
            out["sa2_ball_r%g" % r] = ref_pn2.query_ball_point(r, k, xyz2, new_xyz2)[SEL].numpy().astype(np.int16)
        l2_xyz, l2_points = model.sa2(l1_xyz, l1_points)
        logits = model(x)
        phi, logits2 = ref_common.shap_sampling_all_regions_batch(model, data, lbl, region_id, orders, args)
    out.update(region_id=region_id, orders=orders, fps1=fps1.numpy().astype(np.int16), fps2=fps2.numpy().astype(np.int16),
               sel=np.array(SEL), l1_points_rows=l1_points[SEL][:, :, :8].numpy(), l2_points_rows=l2_points[SEL][:, :, :8].numpy(),
               l1_points_absmax=l1_points.abs().amax(dim=(1, 2)).numpy(), l2_points_absmax=l2_points.abs().amax(dim=(1, 2)).numpy(),
               logits=logits.numpy(), phi=phi, shap_logits=logits2.numpy())
    np.savez_compressed(os.path.join(HERE, "pointnet2.npz"), **out)
    print("logits spread", logits.std().item(), "phi", phi)
    print(logits.numpy()[[0, 4, 8, 17]].round(3))


if __name__ == "__main__":
    main()
