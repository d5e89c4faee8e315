"""Stage 5 of scripts/exp_shapley.sh - host-side mirror of final_smoothness_center_enum_all.py: region Shapley
values while the linearity / planarity / scattering of every region is pushed up ("inc") or down ("dec") in steps
of ENUM_STEP.

The reference runs the enumeration region by region with autograd on a few dozen points and several host syncs per
gradient step, then evaluates the Shapley values of the deformed cloud after every epoch.  Regions only ever move
their own points, so here the whole enumeration (all epochs, all regions, all stop conditions) is ONE launch of
``iq_smoothness_enum`` (one wavefront per region, closed-form gradient) with a single D2H of 32 integers to learn
how many epochs ran; the deformed clouds of all epochs then go through the fused coalition path as a batch of poses,
sharded over the ranks like stages 2-4.  Artefacts as the reference writes them under
``<cloud>/<mode>_all/allregion_<inc|dec>/``: orig_shapley_value.npy (R,), region_shapley_value.npy (P,R),
all_logits.pt (P,S*(R+1),C), <mode>.npy (P,R) f64, data_smoothness.npy (P,1,N,3) f32, log.txt.

Reproduced on purpose: the reference's distance bound never moves a point - ``data_region_i[i].data = ...`` (:117)
assigns to a temporary view - it only counts the points beyond DIST_THRESHOLD for the stop condition; the kernel
does the same unless ``project_to_bound`` is requested (hip_ops.smoothness_enum).

Deviations, on purpose: (1) torch.symeig is gone from current torch; the 3x3 eigen-decomposition runs inside the
kernel (fp64 Jacobi).  (2) On a CPU tensor the reference's ``data_list`` aliases the working cloud, so every saved
epoch shows the final state; on a GPU (how the experiment is run) each epoch is a snapshot - that is what is written
here.  (3) A region with fewer than two points (which makes the reference fail) is left untouched, smoothness NaN.
"""
import time

import numpy as np
import torch

from . import dist as iqdist
from . import final_common, hip_ops
from .final_util import NUM_SAMPLES, IOStream, get_folder_name_list, load_model, mkdir, set_shapley_batch_size
from .pose_sweep import sharded_shapley
from .shapley_stage import build_parser, data_loader, finish_args

STEP = 1e-3            # final_smoothness_center_enum_all.py:13-19
ENUM_STEP = 0.05
EPOCH = 50
VAR_THRESHOLD = 0.003
DIST_THRESHOLD = 0.03
STOP_RATIO = 0.5
MAX_ITERATION = 100


def enumerate_smoothness(data, region_id, args, objective):
    """The epoch loop of test_all_region (:303-335) without its Shapley calls.  data (1,N,3) GPU tensor ->
    (poses (P,N,3) f32 GPU tensor, smoothness (P,R) f64 ndarray, raw kernel outputs)."""
    dev = data.device
    res = hip_ops.smoothness_enum(data[0].contiguous(), hip_ops.region_ids(region_id, dev, args.num_regions), args.num_regions, args.mode, objective,
                                  step=args.step, enum_step=args.enum_step, var_threshold=args.var_threshold,
                                  dist_threshold=args.dist_threshold, stop_ratio=args.stop_ratio, epochs=args.epoch,
                                  max_iteration=args.max_iteration)
    stop = res["stop_epoch"].cpu().numpy()                       # the one sync of the enumeration
    n_epochs = max(1, min(int(args.epoch), int(stop.max()) + 1))  # break once every indicator is False (:333-334)
    res["stop_epoch_host"] = stop
    return res["data"][:n_epochs], res["smoothness"][:n_epochs].double().cpu().numpy(), res


def _log_enumeration(io, res, n_epochs, args, objective):
    """The per-region lines of get_original_region_info / update_region (:204,239-240,256-257)."""
    orig = res["orig"].cpu().numpy()
    sm = res["smoothness"].cpu().numpy()
    var = res["var"].cpu().numpy()
    stop = res["stop_epoch_host"]
    sign = 1.0 if objective == "inc" else -1.0
    for r in range(args.num_regions):
        io.cprint("var1 orig: %.8f, var2 orig: %.8f, var3 orig: %.8f" % tuple(orig[r, :3]))
        io.cprint("orig %s: %.8f" % (args.mode, orig[r, 3]))
    for e in range(n_epochs):
        io.cprint("\n************ epoch %d ***********" % e)
        for r in range(args.num_regions):
            if 0 <= e <= stop[r]:
                before = orig[r, 3] if e == 0 else sm[e - 1, r]
                io.cprint("\tregion%d orig %s: %.8f, target %s: %.8f" % (r, args.mode, before, args.mode,
                                                                         before + sign * args.enum_step))
                io.cprint("var1: %.8f, var2: %.8f, var3: %.8f" % tuple(var[e, r]))
                io.cprint("curr smoothness: %.8f" % sm[e, r])


def test_all_region(model, data, lbl, load_order_list, region_id, mode_folder, args, objective):
    """final_smoothness_center_enum_all.py:280-356."""
    assert objective in ["inc", "dec"]
    t_start = time.time()
    write = iqdist.rank() == 0
    result_path = mode_folder + "allregion_%s/" % objective
    io = None
    if write:
        mkdir(result_path)
        io = IOStream(result_path + "log.txt")
        io.cprint(str(args))
    with torch.no_grad():
        poses, smoothness_list, res = enumerate_smoothness(data, region_id, args, objective)
        n_pose = poses.shape[0]
        orig_shap_value, phi, logits = sharded_shapley(model, data, poses, lbl, region_id, load_order_list, args)
    if write:
        io.cprint("origin shapley of this region: %s" % str(orig_shap_value))
        np.save(result_path + "orig_shapley_value.npy", orig_shap_value)
        _log_enumeration(io, res, n_pose, args, objective)
        phi_np = phi.cpu().numpy()
        for e in range(n_pose):
            io.cprint("epoch %d region shapley value: %s" % (e, str(phi_np[e])))
        np.save(result_path + "region_shapley_value.npy", phi_np)             # (num_poses, num_regions)
        torch.save(logits, result_path + "all_logits.pt")                      # (num_poses, S*(R+1), C)
        np.save(result_path + "%s.npy" % args.mode, smoothness_list)           # (num_poses, num_regions)
        np.save(result_path + "data_smoothness.npy", poses.unsqueeze(1).cpu().numpy())  # (num_poses,1,N,3)
        io.cprint("time: %f" % (time.time() - t_start))
        io.close()


def test_smoothness(args):
    """final_smoothness_center_enum_all.py:360-390."""
    model = load_model(args)
    folder_name_list = get_folder_name_list(args)
    for pc_index, (data, lbl) in enumerate(data_loader(args)):
        if not iqdist.cloud_selected(args, pc_index):
            continue
        data, lbl = data.to(args.device), lbl.to(args.device)
        base_folder = args.exp_folder + "%s/" % folder_name_list[pc_index]
        mode_folder = base_folder + "%s_all/" % args.mode
        region_id = np.load(base_folder + "region_id.npy")
        load_order_list = np.load(base_folder + "all_orders.npy")
        test_all_region(model, data, lbl, load_order_list, region_id, mode_folder, args, objective="inc")
        test_all_region(model, data, lbl, load_order_list, region_id, mode_folder, args, objective="dec")


def make_args(argv=None):
    args = build_parser("pointnet").parse_args(argv)
    args.num_samples = NUM_SAMPLES
    args.step, args.enum_step, args.epoch = STEP, ENUM_STEP, EPOCH
    args.var_threshold, args.dist_threshold = VAR_THRESHOLD, DIST_THRESHOLD
    args.stop_ratio, args.max_iteration = STOP_RATIO, MAX_ITERATION
    return args


def run(args):
    set_shapley_batch_size(args)
    for mode in ("linearity", "planarity", "scattering"):   # :413-418
        args.mode = mode
        test_smoothness(args)


@iqdist.record
def main(argv=None):
    args = make_args(argv)
    finish_args(args)
    run(args)
