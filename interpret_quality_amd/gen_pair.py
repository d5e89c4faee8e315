"""Stage 1 of scripts/exp_interaction.sh - host-side mirror of final_gen_pair.py: which region pairs
and which contexts the interaction stages evaluate, and which pose is "adversarial".

Everything random here is HOST NumPy on the global RNG, called in the reference's order, so for the
same seed the artefacts (region_pair_list.npy, ratio*_context_list.npy) are the reference's.  The
only device work is two dense forwards: all poses of a cloud in one batch (check_adv_success,
final_gen_pair.py:221-286) and the prediction at a chosen pose (gen_pred_label, :74-88).
"""
import argparse
import itertools
import os

import numpy as np
import torch
from scipy.special import comb

from . import final_common, work
from .final_util import (BALL_QUERY_COEF, ball_query, cal_rank, get_folder_name_list, load_model, mkdir,
                         square_distance_np)  # noqa: F401
from .interaction import DEFAULT_RATIOS
from .pose_sweep import rotate_xyz, translate_pc
from . import dist as iqdist
from .shapley_stage import data_loader, finish_args


def gen_context(region_pair_list, save_path, args):
    """final_gen_pair.py:18-43.  For every ratio: m = int((R-2)*ratio) context regions per sample; at
    most num_save_context_max sampled contexts per pair, all C(R-2, m) of them when there are fewer.

    The reference draws every sampled context with np.random.choice(rest, m, replace=False) - the first m entries of a fresh
    np.random.permutation(R - 2) - on the global generator: 330 000 draws of 29 numbers for 300 pairs, 2.5 s of host time per
    cloud.  With ``args.device`` set the SAME stream is continued on the device (iq_sample_permutations, bit-identical, one
    launch per ratio) and handed back to NumPy; without it (host-only callers) the reference's loop runs as it is."""
    dev = getattr(args, "device", None)
    if dev is not None and getattr(dev, "type", None) == "cuda" and len(region_pair_list) > 0:
        return _gen_context_device(np.asarray(region_pair_list).reshape(-1, 2), save_path, args, dev)
    for ratio in args.ratio:
        m = int((args.num_regions - 2) * ratio)
        per_pair = []
        for region_i, region_j in region_pair_list:
            rest = [r for r in range(args.num_regions) if r != region_i and r != region_j]
            if comb(len(rest), m) > args.num_save_context_max:
                per_pair.append([np.random.choice(rest, m, replace=False) for _ in range(args.num_save_context_max)])
            else:
                per_pair.append(list(itertools.combinations(rest, m)))
        context_list = np.array(per_pair)  # (num_pairs, num_context, m)
        print(context_list.shape)
        np.save(save_path + "ratio%d_context_list.npy" % int(ratio * 100), context_list)


def _gen_context_device(pairs, save_path, args, dev):
    from . import hip_ops
    n_rest, cmax, p = args.num_regions - 2, args.num_save_context_max, pairs.shape[0]
    regions = np.arange(args.num_regions)
    rest = np.stack([regions[(regions != i) & (regions != j)] for i, j in pairs])          # (P, R-2) ascending, as the reference's list
    state = hip_ops.mt_state_to_device(dev)
    drawn = {}
    for ratio in args.ratio:            # the generator runs through the sampled ratios in the reference's order
        m = int(n_rest * ratio)
        if comb(n_rest, m) > cmax:
            drawn[ratio] = hip_ops.sample_permutations(state, p * cmax, n_rest)
    hip_ops.mt_state_to_host(state, set_global=True)                                        # one sync; NumPy goes on from here
    for ratio in args.ratio:
        m = int(n_rest * ratio)
        if ratio in drawn:
            first = drawn[ratio].cpu().numpy().astype(np.int64).reshape(p, cmax, n_rest)[:, :, :m]
            context_list = np.take_along_axis(rest[:, None, :], first, axis=2)             # rest[perm[:m]]
        else:
            context_list = np.array([list(itertools.combinations(row.tolist(), m)) for row in rest])
        print(context_list.shape)
        np.save(save_path + "ratio%d_context_list.npy" % int(ratio * 100), context_list)


def gen_pair_random(args):
    """final_gen_pair.py:288-300: num_pairs_random distinct unordered pairs (i < j)."""
    all_pairs = np.array([[i, j] for i in range(args.num_regions) for j in range(args.num_regions) if j > i])
    return all_pairs[np.random.choice(all_pairs.shape[0], size=args.num_pairs_random, replace=False)]


def gen_pair_single_region(region, neighbor_idx, args):
    """final_gen_pair.py:127-142: (region, neighbour) for every ball-query neighbour except itself."""
    neighbors = np.arange(args.num_regions)[neighbor_idx[region]]
    return np.array([[region, n] for n in neighbors if n != region])


def _interaction_folder(args, name):
    return args.exp_folder + "%s/" % name + "interaction_seed%d/" % args.seed


def save_pair_random(args, folder_name_list):
    """final_gen_pair.py:302-320."""
    print("gen pair random...")
    for name in folder_name_list:
        folder = _interaction_folder(args, name)
        mkdir(folder + "normal/")
        mkdir(folder + "%s_adv/" % args.mode)
        np.save(folder + "region_pair_list.npy", gen_pair_random(args))


def _dense_logits(model, clouds_cf, args):
    work.add(clouds_cf.shape[0])
    out = model(clouds_cf)
    return out[0] if args.model == "pointnet" else out


def check_adv_success(args, disturb_fn, folder_name_list):
    """final_gen_pair.py:221-286: one dense forward over all poses of each cloud; saves the pose with the
    lowest reward on the true class (max attacking utility)."""
    model = load_model(args)
    with torch.no_grad():
        for pc_idx, (data, lbl) in enumerate(data_loader(args)):
            name = folder_name_list[pc_idx]
            data, lbl = data.to(args.device), lbl.to(args.device)
            base_folder = args.exp_folder + "%s/" % name
            mode_folder = base_folder + "%s_all/" % args.mode
            params = np.load(mode_folder + ("trans_vector.npy" if args.mode == "trans" else "angle_tuple.npy"))
            poses = torch.cat([disturb_fn(data, torch.from_numpy(params[i]).to(args.device)) for i in range(params.shape[0])], dim=0)
            logits = _dense_logits(model, poses.permute(0, 2, 1).contiguous(), args)
            pred = torch.argmax(logits, dim=1)
            print("%d poses are misclassified" % int((pred != lbl[0].item()).sum()))
            v = final_common.get_reward(logits, lbl, args)
            pose_idx = torch.argmin(v).item()
            folder = _interaction_folder(args, name) + "%s_adv/" % args.mode
            np.save(folder + "pose_idx.npy", pose_idx)
            np.save(folder + "transform_params.npy", params[pose_idx])
            print("Pose idx with max attacking utility: %d" % pose_idx)


def save_pair_single_region(args, folder_name_list):
    """final_gen_pair.py:145-218: per region, the poses of its largest / smallest Shapley value and the
    pairs (region, neighbour) inside a ball of BALL_QUERY_COEF x cloud diameter around the region centre."""
    print("gen pair single region...")
    assert args.mode == "trans" or args.mode == "rotate"
    for pc_idx, (data, _) in enumerate(data_loader(args)):
        name = folder_name_list[pc_idx]
        pts = data.cpu().numpy().squeeze()
        base_folder = args.exp_folder + "%s/" % name
        mode_folder = base_folder + "%s_all/" % args.mode
        single = _interaction_folder(args, name) + "%s_adv_single_region/" % args.mode
        mkdir(single)
        region_id = np.load(base_folder + "region_id.npy")
        phi = np.load(mode_folder + "region_shapley_value.npy")                 # (num_poses, R)
        params = np.load(mode_folder + ("trans_vector.npy" if args.mode == "trans" else "angle_tuple.npy"))
        max_pose, min_pose = np.argmax(phi, axis=0), np.argmin(phi, axis=0)
        range_rank = args.num_regions - cal_rank(np.max(phi, axis=0) - np.min(phi, axis=0))  # 1 = largest range
        diameter = np.sqrt(np.maximum(square_distance_np(pts), 0)).max()
        centers = np.stack([pts[region_id == r].mean(axis=0) for r in range(args.num_regions)])
        neighbor_idx = ball_query(centers, r=BALL_QUERY_COEF * diameter)
        for region in range(args.num_regions):
            folder = single + "range_rank%02d_region%02d/" % (range_rank[region], region)
            for sub in ("normal/", "max_pose/", "min_pose/"):
                mkdir(folder + sub)
            np.save(folder + "max_pose/transform_params.npy", params[max_pose[region]])
            np.save(folder + "max_pose/pose_idx.npy", max_pose[region])
            np.save(folder + "min_pose/transform_params.npy", params[min_pose[region]])
            np.save(folder + "min_pose/pose_idx.npy", min_pose[region])
            pairs = gen_pair_single_region(region, neighbor_idx, args)
            print(pairs.shape)
            if len(pairs) == 0:
                print("NO NEIGHBORS!!!")
            np.save(folder + "region_pair_list.npy", pairs)


def _region_folders(single):
    if not os.path.isdir(single):
        return []
    return [single + d + "/" for d in sorted(os.listdir(single)) if os.path.isdir(single + d)]


def save_context(args, folder_name_list):
    """final_gen_pair.py:44-70."""
    print("gen context start...")
    for name in folder_name_list:
        folder = _interaction_folder(args, name)
        gen_context(np.load(folder + "region_pair_list.npy"), folder, args)
        for region_folder in _region_folders(folder + "%s_adv_single_region/" % args.mode):
            gen_context(np.load(region_folder + "region_pair_list.npy"), region_folder, args)


def gen_pred_label(model, data, lbl, disturb_fn, save_path, args):
    """final_gen_pair.py:74-88."""
    params = torch.from_numpy(np.load(save_path + "transform_params.npy").astype(np.float32)).to(data.device)
    logits = _dense_logits(model, disturb_fn(data, params).permute(0, 2, 1).contiguous(), args)
    pred = torch.argmax(logits, dim=1)
    with open(save_path + "pred_labels.txt", "w") as f:
        f.write("lbl: %d\npred_lbl: %d\n" % (lbl[0].item(), pred[0].item()))
    np.save(save_path + "pred_labels.npy", np.array([lbl[0].item(), pred[0].item()]))


def save_pred_label(args, disturb_fn, folder_name_list):
    """final_gen_pair.py:90-123."""
    print("saving pred labels...")
    model = load_model(args)
    with torch.no_grad():
        for pc_idx, (data, lbl) in enumerate(data_loader(args)):
            data, lbl = data.to(args.device), lbl.to(args.device)
            folder = _interaction_folder(args, folder_name_list[pc_idx])
            gen_pred_label(model, data, lbl, disturb_fn, folder + "%s_adv/" % args.mode, args)
            for region_folder in _region_folders(folder + "%s_adv_single_region/" % args.mode):
                gen_pred_label(model, data, lbl, disturb_fn, region_folder + "max_pose/", args)
                gen_pred_label(model, data, lbl, disturb_fn, region_folder + "min_pose/", args)


def make_args(argv=None):
    """final_gen_pair.py:323-374, same flags."""
    p = argparse.ArgumentParser(description="Point Cloud Recognition")
    p.add_argument("--model", type=str, default="pointnet", metavar="N",
                   choices=["pointnet", "pointnet2", "pointconv", "dgcnn", "gcnn", "gcnn_adv"])
    p.add_argument("--test_batch_size", type=int, default=1, metavar="batch_size", help="Size of batch)")
    p.add_argument("--dataset", type=str, default="shapenet", metavar="N", choices=["modelnet10", "shapenet"])
    p.add_argument("--no_cuda", type=bool, default=False, help="enables CUDA training")
    p.add_argument("--seed", type=int, default=1, metavar="S", help="random seed (default: 1)")
    p.add_argument("--device_id", type=int, default=0)
    p.add_argument("--mode", default="rotate", type=str)
    p.add_argument("--ratio", default=DEFAULT_RATIOS, type=int)
    p.add_argument("--num_pairs_random", default=300, type=int)
    p.add_argument("--num_save_context_max", default=100, type=int)
    p.add_argument("--softmax_type", default="modified", type=str, choices=["normal", "modified"])
    p.add_argument("--synthetic", action="store_true")
    p.add_argument("--num_clouds", type=int, default=30)
    return p.parse_args(argv)


def run(args):
    """Same stage order as final_gen_pair.py:323-374.  Host RNG streams that run on from cloud to cloud + one small search
    per cloud: never sharded by cloud (the draws of cloud k depend on what clouds < k consumed)."""
    names = get_folder_name_list(args)
    disturb_fn = translate_pc if args.mode == "trans" else rotate_xyz
    save_pair_random(args, names)
    check_adv_success(args, disturb_fn, names)
    save_pair_single_region(args, names)
    save_context(args, names)
    save_pred_label(args, disturb_fn, names)


@iqdist.record
def main(argv=None):
    args = make_args(argv)
    finish_args(args)
    # under a multi-rank launch rank 0 does the work and the others wait, so that no two ranks write the same files
    if iqdist.rank() == 0:
        run(args)
    iqdist.barrier()
