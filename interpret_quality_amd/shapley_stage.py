"""Stage 1 of scripts/exp_shapley.sh - host-side mirror of final_shapley_value.py (and of the
pre-stage final_save_fps.py): region ids, norm factor, 1000 sampled permutations and the region
Shapley values of every cloud at its original pose.

Same artefacts as the reference: region_id.npy, norm_factor.npy, all_orders.npy,
shapley/<i>_<count>.npy, region_shapley/<i>_<count>.npy, region_sv_all.npy.  The reference issues
one 33-cloud forward per permutation with a host sync each (final_shapley_value.py:138-150); here
all permutations of a cloud go through ONE fused launch per rank and the float64 accumulation runs
on the device in permutation order (bit-identical order of adds).
"""
import argparse
import os

import numpy as np
import torch

from . import dist as iqdist
from . import final_common, hip_ops
from .final_util import (NUM_POINTS, NUM_REGIONS, NUM_SAMPLES_SAVE, IOStream, exp_folder, get_folder_name_list,
                         load_model, mkdir, set_model_args, set_random, synthetic_loader)

SAMPLE_NUMS = [100, 200, 300, 400, 500, 600, 700, 800, 900, 1000, 2000, 3000, 4000, 5000]  # final_shapley_value.py:111


def farthest_point_sample(xyz, npoint):
    """final_save_fps.py:10-31.  xyz (B,N,3) GPU tensor -> (B,npoint) int64."""
    return hip_ops.fps(xyz.contiguous(), npoint).long()


def fps_index_path(args):
    return "fps_%s_%d_%d_index_final30.npy" % (args.dataset, args.num_points, args.num_regions)


def save_fps(args, loader=None):
    """final_save_fps.py:34-54 -> fps_<dataset>_<N>_<R>_index_final30.npy (30,R) int64."""
    loader = loader if loader is not None else data_loader(args)
    rows = [farthest_point_sample(data.to(args.device), args.num_regions).cpu().numpy() for data, _ in loader]
    out = np.concatenate(rows)
    tmp = fps_index_path(args) + ".tmp.npy"
    np.save(tmp, out)
    os.replace(tmp, fps_index_path(args))  # never a half-written file under its final name
    return out


def data_loader(args):
    """The 30-cloud loaders of final_data_shapley.py (data_shapley.py here) when the dataset trees exist;
    ``--synthetic`` (additive flag) substitutes the seeded synthetic clouds when they do not."""
    cache = getattr(args, "data_cache", None)   # additive: the sweep driver parses a dataset once per process
    key = (args.dataset, bool(getattr(args, "synthetic", False)), getattr(args, "num_clouds", 30), args.num_points)
    if cache is not None and key in cache:
        return iter(cache[key])
    if getattr(args, "synthetic", False):
        items = synthetic_loader(args)
    else:
        from .data_shapley import shapley_test_loader
        items = shapley_test_loader(args)
    if cache is None:
        return items
    cache[key] = list(items)
    return iter(cache[key])


def cal_region_id(data, fps_index, result_path, save=True):
    """final_shapley_value.py:20-35.  data (1,N,3), fps_index (R,) -> (N,) int64 ndarray."""
    idx = hip_ops.as_i32(fps_index, data.device)
    region_id = hip_ops.region_assign(data[0].contiguous(), idx).cpu().numpy().astype(np.int64)
    if save:
        np.save(result_path + "region_id.npy", region_id)
    return region_id


def cal_norm_factor(model, data, lbl, center, result_path, args, save=True):
    """final_shapley_value.py:39-56: v(N) - v(empty)."""
    empty = center.view(1, 1, 3).expand(data.shape[0], args.num_points, 3).clone()
    v_n, _ = final_common.cal_reward(model, data, lbl, args)
    v_0, _ = final_common.cal_reward(model, empty, lbl, args)
    norm_factor = (v_n - v_0).cpu().item()
    if save:
        np.save(result_path + "norm_factor.npy", norm_factor)
    return norm_factor


def generate_all_orders(result_path, args, save=True):
    """final_shapley_value.py:59-72.  The reference draws ``num_samples_save`` permutations from NumPy's global generator;
    here the SAME stream is continued on the device (iq_sample_permutations: MT19937 + the legacy shuffle, bit-identical),
    and the advanced state is handed back to NumPy, so all_orders.npy and every later host draw are the reference's for the
    same seed.  Returns the (S,R) int64 ndarray the reference returns."""
    state = hip_ops.mt_state_to_device(args.device)
    orders = hip_ops.sample_permutations(state, args.num_samples_save, args.num_regions)
    hip_ops.mt_state_to_host(state, set_global=True)
    all_orders = orders.cpu().numpy().astype(np.int64)
    if save:
        np.save(result_path + "all_orders.npy", all_orders)
    return all_orders


def mask_data(masked_data, center, order, region_id):
    """final_shapley_value.py:74-88 (one order)."""
    ns = argparse.Namespace(num_regions=len(order))
    return final_common.mask_data_batch(masked_data, center, np.asarray(order)[None, :], region_id, ns)


def save_shapley(region_shap_value, pc_idx, count, result_path, region_id, args):
    """final_shapley_value.py:91-106."""
    shap_value = np.zeros((args.num_points,))
    mkdir(result_path + "shapley/")
    mkdir(result_path + "region_shapley/")
    for k in range(args.num_regions):
        shap_value[region_id == k] = region_shap_value[k] / count
    np.save(result_path + "shapley/%s.npy" % (str(pc_idx) + "_" + str(count)), shap_value)
    np.save(result_path + "region_shapley/%s.npy" % (str(pc_idx) + "_" + str(count)), region_shap_value / count)


def shapley_all_orders(model, data, lbl, region_id, all_orders, args):
    """Loop A for one cloud (final_shapley_value.py:138-156), sharded over ranks by permutation.
    Returns (running sums at SAMPLE_NUMS {count: (R,)}, region_sv_all (S,R) float64, total (R,))."""
    s = len(all_orders)

    def rewards(lo, hi):  # this rank's permutations -> (hi-lo, R+1) rewards
        if hi == lo:
            return torch.zeros((0, args.num_regions + 1), dtype=torch.float32, device=data.device)
        logits = final_common.shapley_logits(model, data, lbl, region_id, all_orders[lo:hi], args)
        return final_common.get_reward(logits, lbl, args).reshape(hi - lo, args.num_regions + 1)

    v = iqdist.sharded_rows(s, rewards).reshape(-1)  # one gather per cloud
    counts = [c for c in SAMPLE_NUMS if c <= s]
    total, rows, snaps = hip_ops.shapley_accum(v.contiguous(), hip_ops.as_i32(all_orders, v.device), snap_counts=counts)
    snaps = snaps.cpu().numpy() if snaps is not None else np.zeros((0, args.num_regions))
    return {c: snaps[k] for k, c in enumerate(counts)}, rows.cpu().numpy(), total.cpu().numpy()


def shap_sampling(model, dataloader, args, folder_name_list):
    """final_shapley_value.py:110-156."""
    with torch.no_grad():
        fps_indices = np.load(fps_index_path(args))
        subset = getattr(args, "cloud_subset", None)
        for i, (data, lbl) in enumerate(dataloader):
            if subset is not None and i > max(subset):
                break      # nothing later in this call reads the stream: no need to draw the remaining clouds' permutations
            result_path = args.exp_folder + "%s/" % folder_name_list[i]
            if not iqdist.cloud_selected(args, i):
                generate_all_orders(result_path, args, save=False)  # the stream runs on from cloud to cloud: draw, do not compute
                continue
            mkdir(result_path)
            data, lbl = data.to(args.device), lbl.to(args.device)
            write = iqdist.rank() == 0
            region_id = cal_region_id(data, fps_indices[i], result_path, save=write)
            center = torch.mean(data, dim=1).squeeze()
            cal_norm_factor(model, data, lbl, center, result_path, args, save=write)
            all_orders = generate_all_orders(result_path, args, save=write)  # same RNG stream on every rank
            print("pointcloud:%s, index:%d, samples:%d" % (folder_name_list[i], i, len(all_orders)))
            snaps, region_sv_all, _ = shapley_all_orders(model, data, lbl, region_id, all_orders, args)
            if write:
                for count, running in snaps.items():
                    save_shapley(running, i, count, result_path, region_id, args)
                np.save(result_path + "region_sv_all.npy", region_sv_all)


def test(args):
    """final_shapley_value.py:159-174."""
    model = load_model(args)
    folder_name_list = get_folder_name_list(args)
    if iqdist.rank() == 0 and not os.path.exists(fps_index_path(args)):
        save_fps(args)  # the reference runs final_save_fps.py by hand before exp_shapley.sh
    iqdist.barrier()    # the other ranks read the file below
    shap_sampling(model, data_loader(args), args, folder_name_list)


def build_parser(default_model="pointconv", default_dataset="shapenet"):
    """Flags of final_shapley_value.py:178-187, verbatim, plus additive ones for offline use."""
    p = argparse.ArgumentParser(description="Point Cloud Recognition")
    p.add_argument("--model", type=str, default=default_model, metavar="N",
                   choices=["pointnet", "dgcnn", "gcnn", "pointnet2", "pointconv", "gcnn_adv"])
    p.add_argument("--dataset", type=str, default=default_dataset, metavar="N", choices=["modelnet10", "shapenet"])
    p.add_argument("--test_batch_size", type=int, default=1, metavar="batch_size", help="Size of batch)")
    p.add_argument("--no_cuda", type=bool, default=False, help="enables CUDA training")
    p.add_argument("--seed", type=int, default=1, metavar="S", help="random seed (default: 1)")
    p.add_argument("--device_id", type=int, default=0, help="gpu id to use")
    p.add_argument("--softmax_type", type=str, default="modified", choices=["normal", "modified"])
    # additive (not in the reference)
    p.add_argument("--synthetic", action="store_true", help="synthetic clouds/weights (no datasets or checkpoints offline)")
    p.add_argument("--num_clouds", type=int, default=30)
    return p


def prepare_args(args, device):
    """The device-independent tail of every stage's main() (final_shapley_value.py:189-203): constants, folders, seeds,
    model arguments.  ``device``: the torch device this process already selected."""
    args.num_points = NUM_POINTS
    args.num_regions = getattr(args, "num_regions", None) or NUM_REGIONS
    args.exp_folder = exp_folder(args)
    args.cuda = True
    args.device = device
    mkdir("checkpoints")
    mkdir(args.exp_folder)
    set_random(args.seed)
    set_model_args(args)
    return args


def finish_args(args):
    """Common tail of every stage's main(): process group, device selection, then prepare_args."""
    _, world, local_rank = iqdist.init_from_env("cuda")
    if world == 1:
        # The reference pins the process with CUDA_VISIBLE_DEVICES = device_id (final_shapley_value.py:196-198).  Here the
        # device is SELECTED (torch.cuda.set_device below), never hidden: writing HIP_VISIBLE_DEVICES only works before the
        # first HIP call of the process, which no code path of torch guarantees.  device_id counts within whatever the
        # launcher left visible; one beyond that (the reference's final_cal_interactions.py defaults to 1) wraps around.
        n_dev = torch.cuda.device_count()
        local_rank = args.device_id % max(n_dev, 1)
        if local_rank != args.device_id:
            print("warning: --device_id %d is not among the %d visible device(s); using device %d" % (args.device_id, n_dev, local_rank))
    if args.no_cuda or not torch.cuda.is_available():
        raise SystemExit("this build has no CPU path: a GPU is required (the reference's --no_cuda is not supported)")
    torch.cuda.set_device(local_rank)
    prepare_args(args, torch.device("cuda", local_rank))
    print("Using GPU : %d from %d devices" % (torch.cuda.current_device(), torch.cuda.device_count()))
    return args


def make_args(argv=None):
    parser = build_parser()
    parser.add_argument("--num_samples_save", type=int, default=NUM_SAMPLES_SAVE)  # additive
    parser.add_argument("--num_regions", type=int, default=NUM_REGIONS)            # additive
    return parser.parse_args(argv)


@iqdist.record
def main(argv=None):
    args = make_args(argv)
    finish_args(args)
    test(args)


if __name__ == "__main__":
    main()
