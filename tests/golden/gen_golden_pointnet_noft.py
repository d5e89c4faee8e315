"""PointNet WITHOUT the feature STN (models/pointnet.py:62-63,72-78: `feature_transform=False`, a constructor option the
stage scripts never take - tools/final_util.py:173,191 set True), from the REFERENCE imported from /root/reference: dense
logits and crt_points of two raw clouds, and the Shapley path (8 regions, 4 permutations, bs 2) of one.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/gen_golden_pointnet_noft.py

Output tests/golden/pointnet_noft.npz (data only)."""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refenv  # noqa: E402

synth = _refenv.setup()

from tools import final_common as ref_common  # noqa: E402
from tools import final_util as ref_util  # noqa: E402
import final_shapley_value as ref_stage1  # noqa: E402
import final_save_fps as ref_fps  # noqa: E402
from models.pointnet import PointNetCls  # noqa: E402


def main():
    model = PointNetCls(argparse.Namespace(dataset="modelnet10", feature_transform=False, model="pointnet"))
    model.load_state_dict(synth.to_torch(synth.pointnet_state_dict(0, feature_transform=False)))   # strict: the keys are the reference's
    model.eval()
    out = {}
    x = torch.stack([torch.from_numpy(synth.make_cloud(i)[0]) for i in (2, 9)]).permute(0, 2, 1).contiguous()
    with torch.no_grad():
        logits, trans_feat, crt = model(x)
    assert trans_feat is None
    out["dense_cloud_ids"], out["dense_logits"], out["dense_crt"] = np.array([2, 9]), logits.numpy(), crt.numpy().astype(np.int16)
    ci, num_regions, num_samples, bs = 4, 8, 4, 2
    args = argparse.Namespace(model="pointnet", softmax_type="modified", num_points=1024, num_regions=num_regions, num_samples=num_samples,
                              shapley_batch_size=bs, num_samples_save=num_samples)
    pts, label = synth.make_cloud(ci)
    data, lbl = torch.from_numpy(pts).unsqueeze(0), torch.tensor([label], dtype=torch.long)
    region_id = ref_stage1.cal_region_id(data, ref_fps.farthest_point_sample(data, num_regions)[0], None, save=False)
    ref_util.set_random(1)
    orders = ref_stage1.generate_all_orders(None, args, save=False)
    with torch.no_grad():
        phi, shap_logits = ref_common.shap_sampling_all_regions_batch(model, data, lbl, region_id, orders, args)
    out.update(shap_cloud_id=ci, region_id=region_id.astype(np.int8), orders=orders.astype(np.int8), phi=phi, shap_logits=shap_logits.numpy())
    np.savez_compressed(os.path.join(HERE, "pointnet_noft.npz"), **out)
    print({k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
