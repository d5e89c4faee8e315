"""Golden vectors for the DEVICE logic of final_gen_pair.py, from the REFERENCE (imported from /root/reference, never
copied): check_adv_success (:221-286: dense forward over the 216 rotations -> argmin of the reward -> pose_idx.npy,
transform_params.npy), save_pair_single_region (:145-218: range ranks, max / min poses, ball-query neighbour pairs and the
folder names they are stored under) and save_pred_label (:90-123).  The reference's functions read their inputs from
files and from module globals (folder_name_list, DataLoader / dataset classes, load_model): those are plain module
attributes and are set here before the calls - the functions themselves run unmodified.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/gen_golden_gen_pair.py

Inputs: synthetic cloud 3, PointNet with the synthetic weights, the reference's own 6^3 rotation grid, and a seeded
(216, 32) table standing in for rotate_all/region_shapley_value.npy (the real one costs 713 k reference forward passes;
these functions only rank it).  Output tests/golden/gen_pair.npz (data only)."""
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

import final_gen_pair as ref_pair  # noqa: E402
import final_rotate_center_enum_all as ref_rot  # noqa: E402
import final_save_fps as ref_fps  # noqa: E402
import final_shapley_value as ref_stage1  # noqa: E402
from models.pointnet import PointNetCls  # noqa: E402

CLOUD, R, NAME = 3, 32, "synthetic_03"


def main():
    torch.set_num_threads(8)
    pts, label = synth.make_cloud(CLOUD)
    data = torch.from_numpy(pts).unsqueeze(0)
    model = PointNetCls(argparse.Namespace(dataset="modelnet10", feature_transform=True, model="pointnet"))
    model.load_state_dict(synth.to_torch(synth.pointnet_state_dict(0)))
    model.eval()
    region_id = ref_stage1.cal_region_id(data, ref_fps.farthest_point_sample(data, R)[0], None, save=False)
    grid = ref_rot.generate_rotate_angle(argparse.Namespace(angle_threshold=ref_rot.ANGLE_THRESHOLD, num_grid_enum_rotate=6),
                                         "cpu").numpy()
    table = np.random.default_rng(42).standard_normal((grid.shape[0], R))          # stands in for region_shapley_value.npy
    out = {"cloud_id": CLOUD, "region_id": region_id.astype(np.int8), "angle_tuple": grid, "region_shapley_value": table}
    with tempfile.TemporaryDirectory() as td:
        exp = td + "/exp/"
        base = exp + NAME + "/"
        os.makedirs(base + "rotate_all")
        os.makedirs(base + "interaction_seed1/rotate_adv")
        os.makedirs(base + "interaction_seed1/normal")
        np.save(base + "region_id.npy", region_id)
        np.save(base + "rotate_all/angle_tuple.npy", grid)
        np.save(base + "rotate_all/region_shapley_value.npy", table)
        args = argparse.Namespace(model="pointnet", dataset="modelnet10", mode="rotate", seed=1, num_points=1024, num_regions=R,
                                  test_batch_size=1, device=torch.device("cpu"), exp_folder=exp, softmax_type="modified")
        # module-level names the reference's functions read (they only exist under __main__ there)
        ref_pair.folder_name_list = [NAME]
        ref_pair.ModelNet_Loader_Shapley_test = lambda *a, **k: [(pts, np.int64(label))]
        ref_pair.DataLoader = lambda ds, **k: [(torch.from_numpy(ds[0][0]).unsqueeze(0), torch.tensor([int(ds[0][1])]))]
        ref_pair.load_model = lambda a: model
        ref_pair.check_adv_success(args, ref_rot.rotate_xyz)
        ref_pair.save_pair_single_region(args)
        ref_pair.save_pred_label(args, ref_rot.rotate_xyz)
        adv = base + "interaction_seed1/rotate_adv/"
        out["pose_idx"] = np.load(adv + "pose_idx.npy")
        out["transform_params"] = np.load(adv + "transform_params.npy")
        out["pred_labels"] = np.load(adv + "pred_labels.npy")
        single = base + "interaction_seed1/rotate_adv_single_region/"
        names = sorted(os.listdir(single))
        assert len(names) == R
        out["range_rank"] = np.array([int(n[10:12]) for n in sorted(names, key=lambda n: int(n[-2:]))])   # by region
        pairs, counts, maxp, minp, pl = [], [], [], [], []
        for r in range(R):
            f = single + "range_rank%02d_region%02d/" % (out["range_rank"][r], r)
            p = np.load(f + "region_pair_list.npy").reshape(-1, 2)
            pairs.append(p); counts.append(len(p))
            maxp.append(np.load(f + "max_pose/pose_idx.npy")); minp.append(np.load(f + "min_pose/pose_idx.npy"))
            pl.append(np.concatenate([np.load(f + "max_pose/pred_labels.npy"), np.load(f + "min_pose/pred_labels.npy")]))
        out["single_pairs"] = np.concatenate(pairs).astype(np.int16)
        out["single_pair_counts"] = np.array(counts)
        out["max_pose_idx"], out["min_pose_idx"] = np.array(maxp), np.array(minp)
        out["single_pred_labels"] = np.stack(pl)
    print("pose_idx", out["pose_idx"], "pred", out["pred_labels"], "pairs", out["single_pair_counts"].sum(),
          "ranks", out["range_rank"][:8])
    np.savez_compressed(os.path.join(HERE, "gen_pair.npz"), **out)


if __name__ == "__main__":
    main()
