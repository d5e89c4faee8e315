"""Stages 2-3 of scripts/exp_interaction.sh - host-side mirror of
final_point_binary_interaction_logits.py (logits of the 4 masked clouds per context) and
final_cal_interactions.py (I_ij = v(S+ij) + v(S) - v(S+i) - v(S+j)).

The reference evaluates one region pair at a time in batches of ``interaction_batch_size`` contexts
and pulls every interaction to the host with ``.item()`` (P*C syncs).  Here all (pair, context)
coalitions of a setting go through one fused launch per rank (pairs sharded over ranks), one
all-gather, one reduction kernel and one device->host copy.
"""
import argparse
import math
import os
import time

import numpy as np
import torch

from . import dist as iqdist
from . import final_common, hip_ops, work
from .final_util import (MODELNET_INTER_SELECTED_SAMPLE, NUM_POINTS, NUM_REGIONS, SHAPENET_INTER_SELECTED_SAMPLE,
                         get_folder_name_list, load_model, set_interaction_batch_size)
from .pose_sweep import rotate_xyz, translate_pc
from .shapley_stage import data_loader, finish_args

DEFAULT_RATIOS = [0., 0.04, 0.07, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.]  # :153


def context_keep_masks(region_pair_list, context_list, num_regions=64):
    """(P,2) pairs and (P,C,m) contexts -> (P*C*4,) uint64 keep masks in the reference's row order
    4k: S+{i,j}, 4k+1: S+{i}, 4k+2: S+{j}, 4k+3: S (final_point_binary_interaction_logits.py:48-52).
    m = 0 gives the empty context (np.in1d(region_id, []) is all False)."""
    pairs = np.asarray(region_pair_list, dtype=np.int64).reshape(-1, 2)
    ctx = np.asarray(context_list, dtype=np.int64)
    p, c = ctx.shape[0], ctx.shape[1]
    hip_ops.check_host_indices(pairs, 0, num_regions, "region_pair_list")
    hip_ops.check_host_indices(ctx, 0, num_regions, "context_list")
    one = np.uint64(1)
    if ctx.shape[2] > 0:
        s = np.bitwise_or.reduce(np.left_shift(one, ctx.astype(np.uint64)), axis=2)  # (P,C)
    else:
        s = np.zeros((p, c), dtype=np.uint64)
    bi = np.left_shift(one, pairs[:, 0].astype(np.uint64))[:, None]
    bj = np.left_shift(one, pairs[:, 1].astype(np.uint64))[:, None]
    out = np.stack([s | bi | bj, s | bi, s | bj, s], axis=2)  # (P,C,4)
    return out.reshape(-1)


def compute_order_interaction_logits(model, data_disturb, region_id, region_pair_list, context_list, args):
    """final_point_binary_interaction_logits.py:15-70.  data_disturb (1,N,3); returns
    (num_pairs, 4*num_context, num_class).  An empty pair list returns an empty tensor (the
    reference's torch.cat([]) raises there; documented deviation)."""
    t0 = time.time()
    dev = data_disturb.device
    r = args.num_regions
    pairs = np.asarray(region_pair_list).reshape(-1, 2)
    num_pairs = pairs.shape[0]
    ctx = np.asarray(context_list)
    num_context = ctx.shape[1] if ctx.ndim >= 2 else 0
    lo, hi = iqdist.shard_range(num_pairs)
    center = torch.mean(data_disturb, dim=1)  # (1,3), :32
    rid = hip_ops.region_ids(region_id, dev, r)
    with torch.no_grad():
        if hi > lo:
            # equal sets are equal clouds (few-region contexts repeat a lot): evaluate the distinct ones once
            keep_np, inv = final_common.distinct_coalitions(context_keep_masks(pairs[lo:hi], ctx[lo:hi], r))
            work.add(inv.size, keep_np.size)
            inv_t = torch.from_numpy(inv.astype(np.int64)).to(dev)
            strict = getattr(args, "strict_batch_cap", False)
            if hasattr(model, "coalition_logits"):
                logits = final_common.coalition_logits_capped(model, data_disturb.contiguous(), center.contiguous(), rid.reshape(1, -1),
                                                              hip_ops.masks_to_tensor(keep_np, dev), r,
                                                              4 * args.interaction_batch_size if strict else None)
            else:
                # config.py's knob is a floor (a cap with strict_batch_cap): rows are independent in eval mode, larger
                # launches give the same logits
                bs = 4 * args.interaction_batch_size if strict else max(4 * args.interaction_batch_size,
                                                                       getattr(model, "preferred_clouds_per_call", 0))
                keep = hip_ops.masks_to_tensor(keep_np, dev)
                chunks = []
                points_api = hasattr(model, "forward_points")  # consumes (B,N,3) directly: no transpose
                for i in range(0, keep.numel(), bs):
                    x = hip_ops.mask_coalitions(data_disturb[0].contiguous(), rid, keep[i:i + bs].contiguous(),
                                                center.reshape(3).contiguous(), channel_first=not points_api)
                    chunks.append(model.forward_points(x) if points_api else model(x))
                logits = torch.cat(chunks, dim=0)
            logits = logits.index_select(0, inv_t).reshape(hi - lo, 4 * num_context, -1)
        else:
            logits = torch.zeros((0, 4 * num_context, iqdist.num_classes_of(model)), dtype=torch.float32, device=dev)
        all_logits = iqdist.all_gather_rows(logits, num_pairs)
    print("shape of all_logits: ", all_logits.size())
    print("done time: ", time.time() - t0)
    return all_logits


def save_logits_all_orders(model, data, region_id, save_path, args):
    """final_point_binary_interaction_logits.py:73-80."""
    region_pair_list = np.load(save_path + "../region_pair_list.npy")
    for ratio in args.ratio:
        print("\tratio: %f" % ratio)
        context_list = np.load(save_path + "../ratio%d_context_list.npy" % int(ratio * 100))
        all_logits = compute_order_interaction_logits(model, data, region_id, region_pair_list, context_list, args)
        if iqdist.rank() == 0:
            torch.save(all_logits, save_path + "ratio%d_all_logits.pt" % int(ratio * 100))


def _selected(args):
    return MODELNET_INTER_SELECTED_SAMPLE if args.dataset == "modelnet10" else SHAPENET_INTER_SELECTED_SAMPLE


def save_logits(args, disturb_fn):
    """final_point_binary_interaction_logits.py:83-135."""
    model = load_model(args)
    folder_name_list = get_folder_name_list(args)
    with torch.no_grad():
        for pc_idx, (data, lbl) in enumerate(data_loader(args)):
            if pc_idx not in _selected(args) or not iqdist.cloud_selected(args, pc_idx):
                continue
            name = folder_name_list[pc_idx]
            print("======= sample %s =========" % name)
            data = data.to(args.device)
            base_folder = args.exp_folder + "%s/" % name
            interaction_folder = base_folder + "interaction_seed%d/" % args.gen_pair_seed
            single_region_folder = interaction_folder + "%s_adv_single_region/" % args.mode
            region_id = np.load(base_folder + "region_id.npy")
            save_logits_all_orders(model, data, region_id, interaction_folder + "normal/", args)
            params = np.load(interaction_folder + "%s_adv/transform_params.npy" % args.mode).astype(np.float32)
            data_disturb = disturb_fn(data, torch.from_numpy(params).to(args.device))
            save_logits_all_orders(model, data_disturb, region_id, interaction_folder + "%s_adv/" % args.mode, args)
            if not os.path.isdir(single_region_folder):
                continue
            for region_folder_name in sorted(os.listdir(single_region_folder)):
                if not os.path.isdir(single_region_folder + region_folder_name):
                    continue
                if int(region_folder_name[10:12]) != 1:  # only the most rotation-sensitive region, :129-131
                    continue
                save_logits_all_orders(model, data, region_id, single_region_folder + region_folder_name + "/normal/", args)


def compute_order_interaction(all_logits, lbl, args):
    """final_cal_interactions.py:14-37.  (P,4C,K) logits -> (P,C) float64 ndarray."""
    p, c4, k = all_logits.shape
    if p == 0 or c4 == 0:
        return np.zeros((p, c4 // 4))
    with torch.no_grad():
        v = final_common.get_reward(all_logits.reshape(p * c4, k).contiguous(), lbl, args)
        inter = hip_ops.interaction_reduce(v)
    return inter.cpu().numpy().astype(np.float64).reshape(p, c4 // 4)  # one device->host copy


def cal_interaction_all_orders(lbl, save_path, args):
    """final_cal_interactions.py:40-46."""
    for ratio in args.ratio:
        print("\tratio: %f" % ratio)
        all_logits = torch.load(save_path + "ratio%d_all_logits.pt" % int(ratio * 100), map_location=args.device)
        all_interaction = compute_order_interaction(all_logits, lbl, args)
        print(all_interaction.shape)
        if iqdist.rank() == 0:
            np.save(save_path + "ratio%d_%s_interaction.npy" % (int(ratio * 100), args.output_type), all_interaction)


def cal_interaction(args):
    """final_cal_interactions.py:49-99."""
    folder_name_list = get_folder_name_list(args)
    with torch.no_grad():
        for pc_idx, (data, lbl) in enumerate(data_loader(args)):
            if pc_idx not in _selected(args) or not iqdist.cloud_selected(args, pc_idx):
                continue
            name = folder_name_list[pc_idx]
            print("======= sample %s =========" % name)
            lbl = lbl.to(args.device)
            base_folder = args.exp_folder + "%s/" % name
            interaction_folder = base_folder + "interaction_seed%d/" % args.gen_pair_seed
            single_region_folder = interaction_folder + "%s_adv_single_region/" % args.mode
            print("##### normal pose")
            cal_interaction_all_orders(lbl, interaction_folder + "normal/", args)
            print("##### max attacking utility pose")
            pred_class = np.load(interaction_folder + "%s_adv/pred_labels.npy" % args.mode)[1]
            pred = torch.tensor([pred_class], dtype=torch.long, device=args.device)
            use = lbl if args.output_type == "gt" else pred
            cal_interaction_all_orders(use, interaction_folder + "%s_adv/" % args.mode, args)
            if not os.path.isdir(single_region_folder):
                continue
            for region_folder_name in sorted(os.listdir(single_region_folder)):
                if not os.path.isdir(single_region_folder + region_folder_name):
                    continue
                if int(region_folder_name[10:12]) != 1:
                    continue
                cal_interaction_all_orders(lbl, single_region_folder + region_folder_name + "/normal/", args)


def build_parser(with_cal_flags):
    """Flags of final_point_binary_interaction_logits.py:141-157 / final_cal_interactions.py:104-123."""
    p = argparse.ArgumentParser(description="Point Cloud Recognition")
    p.add_argument("--model", type=str, default="pointnet", metavar="N",
                   choices=["pointnet", "pointnet2", "pointconv", "dgcnn", "gcnn", "gcnn_adv"])
    p.add_argument("--dataset", type=str, default="shapenet", metavar="N", choices=["modelnet10", "shapenet"])
    p.add_argument("--test_batch_size", type=int, default=1, metavar="batch_size", help="Size of batch)")
    p.add_argument("--no_cuda", type=bool, default=False, help="enables CUDA training")
    p.add_argument("--seed", type=int, default=1, metavar="S", help="random seed (default: 1)")
    p.add_argument("--gen_pair_seed", type=int, default=1)
    p.add_argument("--device_id", type=int, default=1 if with_cal_flags else 0)
    p.add_argument("--mode", default="rotate", type=str)
    p.add_argument("--ratio", default=DEFAULT_RATIOS, type=int)  # as in the reference: not settable from the CLI
    p.add_argument("--num_pairs_random", default=300, type=int)
    p.add_argument("--num_save_context_max", default=100, type=int)
    if with_cal_flags:
        p.add_argument("--softmax_type", default="modified", type=str, choices=["normal", "modified", "yi", "minuslog"])
        p.add_argument("--output_type", default="pred", type=str, choices=["gt", "pred"])
    p.add_argument("--synthetic", action="store_true")
    p.add_argument("--num_clouds", type=int, default=30)
    return p


def make_args(with_cal_flags, argv=None):
    args = build_parser(with_cal_flags).parse_args(argv)
    if not with_cal_flags:
        args.softmax_type = "modified"
    return args


def run_logits(args):
    set_interaction_batch_size(args)
    save_logits(args, disturb_fn=translate_pc if args.mode == "trans" else rotate_xyz)


@iqdist.record
def main_logits(argv=None):
    args = make_args(False, argv)
    finish_args(args)
    run_logits(args)


@iqdist.record
def main_cal(argv=None):
    args = make_args(True, argv)
    finish_args(args)
    cal_interaction(args)
