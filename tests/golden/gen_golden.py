"""Generate golden vectors by running the REFERENCE ITSELF (imported from /root/reference, never
copied) on the seeded synthetic inputs of interpret_quality_amd/synth.py.

Run in the build container only (the reference does not exist on the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/gen_golden.py

Outputs: tests/golden/pointnet_shapley_R{8,32}.npz, pointnet_interaction_R32.npz, geometry.npz.
The fixtures hold inputs' seeds and expected outputs only (data, not source).
"""
import argparse
import hashlib
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, HERE)
import _refenv  # noqa: E402

synth = _refenv.setup()   # the reference first on sys.path, the repository root (its `tools/` shims) off it


# ---- the reference, imported as-is --------------------------------------------------------
from tools import final_common as ref_common  # noqa: E402
from tools import final_util as ref_util  # noqa: E402
import final_shapley_value as ref_stage1  # noqa: E402
import final_save_fps as ref_fps  # noqa: E402
import final_point_binary_interaction_logits as ref_inter  # noqa: E402
import final_cal_interactions as ref_cal  # noqa: E402
import final_gen_pair as ref_pair  # noqa: E402
import final_rotate_center_enum_all as ref_rot  # noqa: E402
import final_trans_center_enum_all as ref_trans  # noqa: E402
import final_scale_center_enum_all as ref_scale  # noqa: E402
from models.pointnet import PointNetCls  # noqa: E402


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t.numpy() if isinstance(t, torch.Tensor) else t).tobytes()).hexdigest()


def ref_pointnet(seed=0):
    args = argparse.Namespace(dataset="modelnet10", feature_transform=True, model="pointnet")
    model = PointNetCls(args)
    model.load_state_dict(synth.to_torch(synth.pointnet_state_dict(seed)))
    return model.eval()


def shapley_golden(model, num_regions, num_samples, bs, cloud_ids):
    out = {"num_regions": num_regions, "num_samples": num_samples, "bs": bs, "cloud_ids": np.array(cloud_ids)}
    args = argparse.Namespace(model="pointnet", softmax_type="modified", num_points=1024,
                              num_regions=num_regions, num_samples=num_samples, shapley_batch_size=bs,
                              num_samples_save=num_samples)
    for ci in cloud_ids:
        pts, label = synth.make_cloud(ci)
        data = torch.from_numpy(pts).unsqueeze(0)
        lbl = torch.tensor([label], dtype=torch.long)
        fps_index = ref_fps.farthest_point_sample(data, num_regions)[0]
        region_id = ref_stage1.cal_region_id(data, fps_index, None, save=False)
        ref_util.set_random(1)
        orders = ref_stage1.generate_all_orders(None, args, save=False)
        center = torch.mean(data, dim=1).squeeze()
        with torch.no_grad():
            norm_factor = ref_stage1.cal_norm_factor(model, data, lbl, center, None, args, save=False)
            phi, logits = ref_common.shap_sampling_all_regions_batch(model, data, lbl, region_id, orders, args)
            # one batch of masked clouds, checksummed
            masked = data.expand((num_regions + 1) * bs, 1024, 3).clone()
            masked = ref_common.mask_data_batch(masked, center, orders[:bs], region_id, args)
            v, _ = ref_common.cal_reward(model, masked, lbl, args)
            # stage-1 style (one order per forward, final_shapley_value.py:138-150)
            m1 = data.expand(num_regions + 1, 1024, 3).clone()
            m1 = ref_stage1.mask_data(m1, center, orders[0], region_id)
            v1, _ = ref_common.cal_reward(model, m1, lbl, args)
            v_normal = ref_common.get_reward(logits[:16], lbl, argparse.Namespace(softmax_type="normal"))
        p = "c%d_" % ci
        out[p + "fps_index"] = fps_index.numpy()
        out[p + "region_id"] = region_id
        out[p + "orders"] = orders
        out[p + "norm_factor"] = np.float64(norm_factor)
        out[p + "phi"] = phi
        out[p + "logits"] = logits.numpy()
        out[p + "masked_sha256"] = np.array(sha(masked))
        out[p + "masked_row1"] = masked[1].numpy()
        out[p + "masked_last_block_row3"] = masked[(num_regions + 1) * (bs - 1) + 3].numpy()
        out[p + "v_batch0"] = v.numpy()
        out[p + "stage1_masked_sha256"] = np.array(sha(m1))
        out[p + "stage1_v"] = v1.numpy()
        out[p + "v_normal16"] = v_normal.numpy()
        print("R=%d cloud %d: norm_factor=%.6f sum(phi)=%.6f logits spread=%.3f" % (
            num_regions, ci, norm_factor, phi.sum(), logits.std().item()))
    return out


def interaction_golden(model, cloud_id=0, num_regions=32, num_pairs=4, ctx_max=6, bs=4):
    pts, label = synth.make_cloud(cloud_id)
    data = torch.from_numpy(pts).unsqueeze(0)
    lbl = torch.tensor([label], dtype=torch.long)
    fps_index = ref_fps.farthest_point_sample(data, num_regions)[0]
    region_id = ref_stage1.cal_region_id(data, fps_index, None, save=False)
    ratios = [0.0, 0.04, 0.5, 1.0]
    args = argparse.Namespace(model="pointnet", softmax_type="modified", num_regions=num_regions,
                              num_pairs_random=num_pairs, num_save_context_max=ctx_max, ratio=ratios,
                              interaction_batch_size=bs)
    ref_util.set_random(1)
    pairs = ref_pair.gen_pair_random(args)
    out = {"cloud_id": cloud_id, "num_regions": num_regions, "bs": bs, "pairs": pairs,
           "region_id": region_id, "ratios": np.array(ratios)}
    with tempfile.TemporaryDirectory() as td:
        ref_pair.gen_context(pairs, td + "/", args)
        for ratio in ratios:
            tag = "ratio%d" % int(ratio * 100)
            ctx = np.load(td + "/%s_context_list.npy" % tag)
            logits = ref_inter.compute_order_interaction_logits(model, data, region_id, pairs, ctx, args)
            inter = ref_cal.compute_order_interaction(logits, lbl, args)
            out[tag + "_contexts"] = ctx
            out[tag + "_logits"] = logits.numpy()
            out[tag + "_interaction"] = inter
            print(tag, ctx.shape, logits.shape, "max|I|=%.3e" % np.abs(inter).max())
    return out


def geometry_golden():
    out = {}
    pts, _ = synth.make_cloud(3)
    data = torch.from_numpy(pts).unsqueeze(0)
    # FPS incl. a half-collapsed cloud (ties -> lowest index on CPU)
    collapsed = data.clone()
    collapsed[0, 300:, :] = collapsed[0, :300].mean(dim=0)
    both = torch.cat([data, collapsed], dim=0)
    for s in (32, 128, 512):
        out["fps_%d" % s] = ref_fps.farthest_point_sample(both, s).numpy()
    out["square_distance_8x5"] = ref_util.square_distance(data[:, :8], data[:, 100:105]).numpy()
    angle = torch.tensor([0.3, -0.7, 0.5])
    out["rotate_in_angle"] = angle.numpy()
    out["rotate_out_sha256"] = np.array(sha(ref_rot.rotate_xyz(data, angle)))
    out["rotate_out_first4"] = ref_rot.rotate_xyz(data, angle)[0, :4].numpy()
    a = argparse.Namespace(angle_threshold=ref_rot.ANGLE_THRESHOLD, num_grid_enum_rotate=6,
                           trans_dist_threshold=0.5, num_grid_enum_trans=6,
                           scale_lower=0.5, scale_upper=2.0, num_grid_enum_scale=30)
    out["rotate_grid"] = ref_rot.generate_rotate_angle(a, "cpu").numpy()
    out["trans_grid"] = ref_trans.generate_trans_vector(a, "cpu").numpy()
    out["scale_grid"] = ref_scale.generate_scale(a, "cpu").numpy()
    out["translate_out_first4"] = ref_trans.translate_pc(data, torch.tensor([0.1, -0.2, 0.3]))[0, :4].numpy()
    out["scale_out_first4"] = ref_scale.scale_pc(data, torch.tensor(1.7))[0, :4].numpy()
    return out


def main():
    torch.set_num_threads(8)
    model = ref_pointnet(0)
    np.savez_compressed(os.path.join(HERE, "pointnet_shapley_R8.npz"),
                        **shapley_golden(model, 8, 8, 4, [0, 1]))
    np.savez_compressed(os.path.join(HERE, "pointnet_shapley_R32.npz"),
                        **shapley_golden(model, 32, 8, 4, [0, 1]))
    np.savez_compressed(os.path.join(HERE, "pointnet_interaction_R32.npz"), **interaction_golden(model))
    np.savez_compressed(os.path.join(HERE, "geometry.npz"), **geometry_golden())
    # dense forward on raw clouds (what final_gen_pair.py:267-270 and cal_norm_factor issue)
    x = torch.stack([torch.from_numpy(synth.make_cloud(i)[0]) for i in range(4)]).permute(0, 2, 1).contiguous()
    with torch.no_grad():
        logits, trans_feat, crt = model(x)
    np.savez_compressed(os.path.join(HERE, "pointnet_dense.npz"), logits=logits.numpy(),
                        trans_feat_sha256=np.array(sha(trans_feat)), trans_feat_b0_row0=trans_feat[0, 0].numpy(),
                        crt_points=crt.numpy())
    print("dense logits:\n", logits.numpy())


if __name__ == "__main__":
    main()
