"""Wider fixtures for PointNet++ (MSG) and PointConv at the reference's region count (32), from the REFERENCE (imported
from /root/reference, never copied): round 1 held 18 masked clouds of ONE cloud at R = 8 for each of them.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/gen_golden_families_r32.py     (about 4 minutes, 8 cores)

Per model, cloud 7 (ShapeNet-style synthetic), R = 32:
  * Shapley: 4 sampled permutations x 33 prefix coalitions through shap_sampling_all_regions_batch (phi + 132 logits rows)
  * interaction: 5 random pairs x 3 ratios x 4 contexts x 4 masked clouds = 240 logits rows + the interactions
Output tests/golden/families_r32.npz (data only)."""
import argparse
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refenv  # noqa: E402

synth = _refenv.setup()   # the reference first on sys.path, the repository root (its `tools/` shims) off it

from models import pointconv as ref_pc  # noqa: E402
from models import pointnet2 as ref_pn2  # noqa: E402
from tools import final_common as ref_common  # noqa: E402
from tools import final_util as ref_util  # noqa: E402
import final_cal_interactions as ref_cal  # noqa: E402
import final_gen_pair as ref_pair  # noqa: E402
import final_point_binary_interaction_logits as ref_inter  # noqa: E402
import final_save_fps as ref_fps  # noqa: E402
import final_shapley_value as ref_stage1  # noqa: E402

CLOUD, R = 7, 32


def main():
    torch.set_num_threads(8)
    pts, label = synth.make_cloud(CLOUD)
    data = torch.from_numpy(pts).unsqueeze(0)
    lbl = torch.tensor([label])
    region_id = ref_stage1.cal_region_id(data, ref_fps.farthest_point_sample(data, R)[0], None, save=False)
    out = {"cloud_id": CLOUD, "label": label, "region_id": region_id.astype(np.int8)}
    ratios = [0.1, 0.5, 0.9]
    models = {"pointnet2": (ref_pn2.PointNet2ClsMsg(argparse.Namespace(dataset="shapenet")), synth.pointnet2_state_dict),
              "pointconv": (ref_pc.PointConvDensityClsSsg(argparse.Namespace(dataset="shapenet")), synth.pointconv_state_dict)}
    for name, (model, sd) in models.items():
        model.load_state_dict(synth.to_torch(sd(0)))
        model.eval()
        args = argparse.Namespace(model=name, softmax_type="modified", num_points=1024, num_regions=R, num_samples=4, shapley_batch_size=2,
                                  num_samples_save=4, num_pairs_random=5, num_save_context_max=4, ratio=ratios, interaction_batch_size=4)
        ref_util.set_random(11)
        orders = ref_stage1.generate_all_orders(None, args, save=False)
        pairs = ref_pair.gen_pair_random(args)
        with torch.no_grad(), tempfile.TemporaryDirectory() as td:
            phi, logits = ref_common.shap_sampling_all_regions_batch(model, data, lbl, region_id, orders, args)
            out[name + "_orders"] = orders.astype(np.int8)
            out[name + "_phi"] = phi
            out[name + "_shap_logits"] = logits.numpy()
            ref_pair.gen_context(pairs, td + "/", args)
            out[name + "_pairs"] = pairs.astype(np.int8)
            for ratio in ratios:
                tag = "%s_ratio%d" % (name, int(ratio * 100))
                ctx = np.load(td + "/ratio%d_context_list.npy" % int(ratio * 100))
                lg = ref_inter.compute_order_interaction_logits(model, data, region_id, pairs, ctx, args)
                out[tag + "_contexts"] = ctx.astype(np.int8)
                out[tag + "_logits"] = lg.numpy()
                out[tag + "_interaction"] = ref_cal.compute_order_interaction(lg, lbl, args)
        print(name, "phi sum %.5f" % phi.sum(), "logit spread %.3f" % logits.std().item(), flush=True)
    out["ratios"] = np.array(ratios)
    np.savez_compressed(os.path.join(HERE, "families_r32.npz"), **out)


if __name__ == "__main__":
    main()
