"""Golden vectors for PointConv from the REFERENCE (imported from /root/reference, never copied).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/gen_golden_pointconv.py

Outputs tests/golden/pointconv.npz: densities, kNN-point index sets, stage features and logits of raw and
masked clouds, Shapley logits / values for a small case."""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refenv  # noqa: E402

synth = _refenv.setup()   # the reference first on sys.path, the repository root (its `tools/` shims) off it

from models import pointconv as ref_pc  # noqa: E402
from tools import final_common as ref_common  # noqa: E402
import final_shapley_value as ref_stage1  # noqa: E402
import final_save_fps as ref_fps  # noqa: E402

SEL = [0, 4, 8]  # of the 9 prefix coalitions of permutation 0: all-centre, half, untouched


def main():
    torch.set_num_threads(8)
    model = ref_pc.PointConvDensityClsSsg(argparse.Namespace(dataset="modelnet10"))
    model.load_state_dict(synth.to_torch(synth.pointconv_state_dict(0)))
    model.eval()
    pts, label = synth.make_cloud(0)
    data = torch.from_numpy(pts).unsqueeze(0)
    lbl = torch.tensor([label])
    num_regions = 8
    fps_index = ref_fps.farthest_point_sample(data, num_regions)[0]
    region_id = ref_stage1.cal_region_id(data, fps_index, None, save=False)
    np.random.seed(1)
    args = argparse.Namespace(model="pointconv", softmax_type="modified", num_points=1024, num_regions=num_regions,
                              num_samples=2, shapley_batch_size=2, num_samples_save=2)
    orders = ref_stage1.generate_all_orders(None, args, save=False)
    center = torch.mean(data, dim=1).squeeze()
    masked = data.expand((num_regions + 1) * 2, 1024, 3).clone()
    masked = ref_common.mask_data_batch(masked, center, orders, region_id, args)
    x = masked.permute(0, 2, 1).contiguous()
    out = dict(region_id=region_id, orders=orders, sel=np.array(SEL))
    with torch.no_grad():
        xs = x[SEL]
        xyz = xs.permute(0, 2, 1)
        out["density_sa1"] = ref_pc.compute_density(xyz, 0.1).numpy()
        fps1 = ref_pc.farthest_point_sample(xyz, 512)
        new_xyz = ref_pc.index_points(xyz, fps1)
        out["knn_sa1"] = ref_pc.knn_point(32, xyz, new_xyz).numpy().astype(np.int16)
        l1_xyz, l1_points = model.sa1(xs, None)
        l2_xyz, l2_points = model.sa2(l1_xyz, l1_points)
        out["l1_points_first8"] = l1_points[:, :8, :].numpy()
        out["l2_points_first8"] = l2_points[:, :8, :].numpy()
        out["logits"] = model(x).numpy()
        phi, logits2 = ref_common.shap_sampling_all_regions_batch(model, data, lbl, region_id, orders, args)
        out["phi"], out["shap_logits"] = phi, logits2.numpy()
    np.savez_compressed(os.path.join(HERE, "pointconv.npz"), **out)
    print("phi", phi)
    print(out["logits"][[0, 4, 8]].round(3))


if __name__ == "__main__":
    main()
