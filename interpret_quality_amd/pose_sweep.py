"""Stages 2-4 of scripts/exp_shapley.sh - host-side mirror of tools.final_common.test
(tools/final_common.py:107-174) and of final_{trans,rotate,scale}_center_enum_all.py: region
Shapley values of every cloud under 216 translations / 216 rotations / 30 scales.

The reference runs one pose at a time (3300 forwards).  Here the poses of a cloud are sharded over
the ranks and each rank evaluates ``pose_batch`` poses per fused launch (coalitions of several
perturbed clouds in one grid via ``cloud_of``); rank 0 gathers once per cloud and writes the same
artefacts: orig_shapley_value.npy, region_shapley_value.npy (P,R), all_logits.pt (P,S*(R+1),C),
trans_vector.npy | angle_tuple.npy | scale.npy, log.txt.
"""
import math
import time

import numpy as np
import torch

from . import dist as iqdist
from . import final_common, hip_ops, work
from .final_util import NUM_SAMPLES, IOStream, get_folder_name_list, load_model, mkdir, set_shapley_batch_size
from .shapley_stage import build_parser, data_loader, finish_args

ANGLE_THRESHOLD = math.pi / 4   # final_rotate_center_enum_all.py:11-12
NUM_GRID_ENUM_ROTATE = 6
TRANS_DIST_THRESHOLD = 0.5      # final_trans_center_enum_all.py:9-10
NUM_GRID_ENUM_TRANS = 6
SCALE_UPPER, SCALE_LOWER, NUM_GRID_ENUM_SCALE = 2.0, 0.5, 30  # final_scale_center_enum_all.py:10-12


# ---- perturbations (a13) ------------------------------------------------------------------------
def rotate_xyz(x, angle_tuple):
    """final_rotate_center_enum_all.py:15-38: R = Rx.Ry.Rz, returns x.R^T.  x (B,N,3)."""
    tx, ty, tz = angle_tuple[0], angle_tuple[1], angle_tuple[2]
    cx, cy, cz = torch.cos(tx), torch.cos(ty), torch.cos(tz)
    sx, sy, sz = torch.sin(tx), torch.sin(ty), torch.sin(tz)
    rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], device=x.device)
    ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], device=x.device)
    rz = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], device=x.device)
    r = torch.matmul(torch.matmul(rx, ry), rz)
    return torch.matmul(x, r.expand(x.shape[0], 3, 3).permute(0, 2, 1))


def translate_pc(data, trans):
    """final_trans_center_enum_all.py:13-21."""
    return torch.add(data, trans)


def scale_pc(data, scale):
    """final_scale_center_enum_all.py:14-22."""
    return data * scale


def generate_rotate_angle(args, device):
    """final_rotate_center_enum_all.py:41-58 -> (6^3, 3) float32, 'ij' order."""
    t = np.linspace(-args.angle_threshold, args.angle_threshold, num=args.num_grid_enum_rotate)
    gx, gy, gz = np.meshgrid(t, t, t, indexing="ij")
    g = np.stack([gx.reshape(-1), gy.reshape(-1), gz.reshape(-1)], axis=1)
    return torch.tensor(g, dtype=torch.float32).to(device)


def generate_trans_vector(args, device):
    """final_trans_center_enum_all.py:24-43 -> (6^3, 3), vectors clipped to the threshold norm."""
    t = np.linspace(-args.trans_dist_threshold, args.trans_dist_threshold, num=args.num_grid_enum_trans)
    gx, gy, gz = np.meshgrid(t, t, t, indexing="ij")
    rows = []
    for v in zip(gx.reshape(-1), gy.reshape(-1), gz.reshape(-1)):
        tv = torch.tensor(list(v), dtype=torch.float32)
        if torch.norm(tv) > args.trans_dist_threshold:
            tv = tv / torch.norm(tv) * args.trans_dist_threshold
        rows.append(tv)
    return torch.stack(rows, dim=0).to(device)


def generate_scale(args, device):
    """final_scale_center_enum_all.py:25-31."""
    return torch.from_numpy(np.linspace(start=args.scale_lower, stop=args.scale_upper,
                                        num=args.num_grid_enum_scale)).float().to(device)


# ---- log / save helpers (same text as the reference) --------------------------------------------
def print_rotate_info(io, angle_tuple, region_shapley_value, epoch):
    io.cprint("rotation angle: [%f pi, %f pi, %f pi]" % (
        angle_tuple[0].item() / np.pi, angle_tuple[1].item() / np.pi, angle_tuple[2].item() / np.pi))
    io.cprint("shapley value after %d epoch:\n%s" % (epoch, str(region_shapley_value)))


def save_rotate_info(all_rotate_angle, result_path):
    np.save(result_path + "angle_tuple.npy", all_rotate_angle.cpu().numpy())


def print_trans_info(io, trans, region_shapley_value, epoch):
    io.cprint("translation vector: [%f, %f, %f]" % (trans[0].item(), trans[1].item(), trans[2].item()))
    io.cprint("translation distance: %f" % torch.norm(trans).item())
    io.cprint("shapley value after %d epoch:\n%s" % (epoch, str(region_shapley_value)))


def save_trans_info(all_trans_vector, result_path):
    np.save(result_path + "trans_vector.npy", all_trans_vector.cpu().numpy())
    np.save(result_path + "trans_distance.npy", torch.norm(all_trans_vector, dim=1).cpu().numpy())


def print_scale_info(io, scale, region_shapley_value, epoch):
    io.cprint("scale: %f" % scale)
    io.cprint("shapley value after %d epoch:\n%s" % (epoch, str(region_shapley_value)))


def save_scale_info(all_scale, result_path):
    np.save(result_path + "scale.npy", all_scale.cpu().numpy())


# ---- the sweep ---------------------------------------------------------------------------------
def shapley_over_poses(model, poses, lbl, region_id, orders, args, pose_batch=8):
    """Region Shapley values of several perturbed copies of one cloud.  poses (P,N,3) on the GPU.
    Returns (phi (P,R) float64 tensor, logits (P, S*(R+1), C))."""
    dev = poses.device
    r, s = args.num_regions, args.num_samples
    if s % args.shapley_batch_size != 0:
        raise final_common.IqError("shapley_batch_size=%d does not divide num_samples=%d" % (args.shapley_batch_size, s))
    orders = np.asarray(orders[:s])
    orders_dev = hip_ops.as_i32(orders, dev)
    rid = hip_ops.region_ids(region_id, dev, r)
    p = poses.shape[0]
    per = s * (r + 1)
    phis, all_logits = [], []
    if hasattr(model, "coalition_logits") and not getattr(args, "strict_batch_cap", False):
        # every pose uses the same permutations, hence the same sets: evaluate the distinct ones (final_common.distinct_coalitions)
        uniq, inv = final_common.distinct_coalitions(final_common.prefix_keep_masks(orders, r))
        nu = len(uniq)
        inv_t = torch.from_numpy(inv.astype(np.int64)).to(dev)
        for lo in range(0, p, pose_batch):
            clouds = poses[lo:lo + pose_batch].contiguous()
            nb = clouds.shape[0]
            work.add(nb * per, nb * nu)
            # centre of the PERTURBED cloud (tools/final_common.py:80), reduced one (1,N,3) cloud at a time as the reference
            # does: torch.mean over a (nb,N,3) batch picks a different reduction order for some nb, which moved the centre by
            # an ulp and made the result depend on how the poses were sharded (found by the two-rank test)
            centers = torch.cat([torch.mean(clouds[k:k + 1], dim=1) for k in range(nb)], dim=0).contiguous()
            keep = hip_ops.masks_to_tensor(np.tile(uniq, nb), dev)
            cloud_of = torch.arange(nb, dtype=torch.int32, device=dev).repeat_interleave(nu).contiguous()
            logits = model.coalition_logits(clouds, centers, rid.reshape(1, -1).expand(nb, -1).contiguous(), keep,
                                            cloud_of, num_regions=r, validate=False)
            logits = logits.reshape(nb, nu, -1).index_select(1, inv_t).reshape(nb * per, -1)
            v = final_common.get_reward(logits, lbl, args)
            for k in range(nb):
                phi_sum, _, _ = hip_ops.shapley_accum(v[k * per:(k + 1) * per].contiguous(), orders_dev)
                phis.append(phi_sum / s)
            all_logits.append(logits.reshape(nb, per, -1))
    else:
        for k in range(p):
            logits = final_common.shapley_logits(model, poses[k:k + 1], lbl, region_id, orders, args)
            v = final_common.get_reward(logits, lbl, args)
            phi_sum, _, _ = hip_ops.shapley_accum(v, orders_dev)
            phis.append(phi_sum / s)
            all_logits.append(logits.unsqueeze(0))
    if p == 0:  # an empty shard still has to agree with the others on the trailing shape of the gather
        return (torch.zeros((0, r), dtype=torch.float64, device=dev),
                torch.zeros((0, per, iqdist.num_classes_of(model)), dtype=torch.float32, device=dev))
    return torch.stack(phis, dim=0), torch.cat(all_logits, dim=0)


def sharded_shapley(model, data, poses, lbl, region_id, orders, args):
    """Region Shapley values of the original cloud ``data`` (1,N,3) and of its perturbed copies ``poses`` (P,N,3), the
    P + 1 clouds sharded over the ranks (the original pose travels as pose 0 of the batch, so no rank repeats it) and
    gathered once.  Returns (orig (R,) float64 ndarray, phi (P,R) float64 tensor, logits (P, S*(R+1), C)), the same on
    every rank.  A cloud's values do not depend on the batch or shard it travels in (tested bitwise)."""
    allp = torch.cat([data.reshape(1, -1, 3), poses.reshape(-1, data.shape[1], 3)], dim=0)
    n = allp.shape[0]
    lo, hi = iqdist.shard_range(n)
    phi, logits = shapley_over_poses(model, allp[lo:hi].contiguous(), lbl, region_id, orders, args)
    phi = iqdist.all_gather_rows(phi, n)          # one gather per cloud
    logits = iqdist.all_gather_rows(logits, n)
    return phi[0].cpu().numpy(), phi[1:], logits[1:]


def test(args, get_transform_params_fn, disturb_fn, print_info_fn, save_info_fn):
    """tools/final_common.py:107-174."""
    model = load_model(args)
    folder_name_list = get_folder_name_list(args)
    write = iqdist.rank() == 0
    for pc_index, (data, lbl) in enumerate(data_loader(args)):
        if not iqdist.cloud_selected(args, pc_index):
            continue
        data, lbl = data.to(args.device), lbl.to(args.device)
        base_folder = args.exp_folder + "%s/" % folder_name_list[pc_index]
        mode_folder = base_folder + "%s_all/" % args.mode
        io = None
        if write:
            mkdir(mode_folder)
            io = IOStream(mode_folder + "log.txt")
            io.cprint(str(args))
            io.cprint("norm factor: %f" % np.load(base_folder + "norm_factor.npy"))
        region_id = np.load(base_folder + "region_id.npy")
        load_order_list = np.load(base_folder + "all_orders.npy")

        t_start = time.time()
        with torch.no_grad():
            all_params = get_transform_params_fn(args, data.device)
            n_pose = all_params.size()[0]
            poses = torch.cat([disturb_fn(data, all_params[i]) for i in range(n_pose)], dim=0)
            orig, phi, logits = sharded_shapley(model, data, poses, lbl, region_id, load_order_list, args)
        if write:
            io.cprint("origin region shapley: %s" % str(orig))
            np.save(mode_folder + "orig_shapley_value.npy", orig)
            phi_np = phi.cpu().numpy()
            for i in range(n_pose):
                print_info_fn(io, all_params[i], phi_np[i], i)
            np.save(mode_folder + "region_shapley_value.npy", phi_np)
            torch.save(logits, mode_folder + "all_logits.pt")
            save_info_fn(all_params, mode_folder)
            io.cprint("time: %f" % (time.time() - t_start))
            io.close()


def make_args(mode, argv=None):
    default_model = {"trans": "gcnn_adv", "rotate": "pointconv", "scale": "pointconv"}[mode]
    args = build_parser(default_model).parse_args(argv)
    args.num_samples = NUM_SAMPLES
    args.mode = mode
    args.angle_threshold, args.num_grid_enum_rotate = ANGLE_THRESHOLD, NUM_GRID_ENUM_ROTATE
    args.trans_dist_threshold, args.num_grid_enum_trans = TRANS_DIST_THRESHOLD, NUM_GRID_ENUM_TRANS
    args.scale_upper, args.scale_lower, args.num_grid_enum_scale = SCALE_UPPER, SCALE_LOWER, NUM_GRID_ENUM_SCALE
    return args


def run(args):
    """The body of the three sweep scripts after argument handling (final_*_center_enum_all.py mains)."""
    set_shapley_batch_size(args)
    fns = {"trans": (generate_trans_vector, translate_pc, print_trans_info, save_trans_info),
           "rotate": (generate_rotate_angle, rotate_xyz, print_rotate_info, save_rotate_info),
           "scale": (generate_scale, scale_pc, print_scale_info, save_scale_info)}[args.mode]
    test(args, *fns)


@iqdist.record
def _main(mode, argv=None):
    args = make_args(mode, argv)
    finish_args(args)
    run(args)


def main_trans(argv=None):
    _main("trans", argv)


def main_rotate(argv=None):
    _main("rotate", argv)


def main_scale(argv=None):
    _main("scale", argv)
