"""Host-side mirror of the reference's shared Shapley kernel module (tools/final_common.py).

Same function names, argument meaning and return types as the reference, so the stage scripts
read the same; the arithmetic runs in libiq_hip.so.  For PointNet the masked clouds are never
materialised (``model.coalition_logits``); for any other model the masking kernel writes the
batch and the model consumes it, as in the reference.
"""
import time

import numpy as np
import torch

from . import hip_ops, work
from ._lib import IqError


def _is_pointnet(args):
    return args.model == "pointnet" or args.model == "pointnet_roty_da"  # tools/final_common.py:36


def get_reward(logits, lbl, args):
    """tools/final_common.py:11-24.  logits (B',C) -> v (B',)."""
    modified = getattr(args, "softmax_type", "modified") != "normal"  # 'yi'/'minuslog' behave as 'modified'
    return hip_ops.reward(logits.contiguous(), int(lbl[0]), modified)


def cal_reward(model, data, lbl, args):
    """tools/final_common.py:26-43.  data (B',N,3) -> (v (B',), logits (B',C))."""
    x = data.permute(0, 2, 1).contiguous()
    work.add(x.shape[0])
    out = model(x)
    logits = out[0] if _is_pointnet(args) else out
    return get_reward(logits, lbl, args), logits


def mask_data_batch(masked_data, center, orders, region_id, args):
    """tools/final_common.py:46-61.  In-place on ``masked_data`` ((R+1)*bs, N, 3), which holds
    bs*(R+1) copies of one cloud on entry (tools/final_common.py:88); one launch instead of
    R*bs index assignments."""
    dev = masked_data.device
    cloud = masked_data[-1].clone()  # the last row of a block is never masked
    hip_ops.check_host_indices(np.asarray(orders), 0, args.num_regions, "orders")
    out = hip_ops.mask_shapley(cloud, hip_ops.region_ids(region_id, dev, args.num_regions), hip_ops.as_i32(np.asarray(orders), dev),
                               center.contiguous(), channel_first=False)
    masked_data.copy_(out)
    return masked_data


def prefix_keep_masks(orders, num_regions):
    """(S,R) permutations -> (S*(R+1),) uint64 keep masks: row i of order o keeps orders[o][:i]."""
    hip_ops.check_host_indices(orders, 0, num_regions, "orders")
    orders = np.asarray(orders, dtype=np.uint64)
    bits = np.left_shift(np.uint64(1), orders)                       # (S,R)
    pref = np.concatenate([np.zeros((orders.shape[0], 1), dtype=np.uint64),
                           np.bitwise_or.accumulate(bits, axis=1)], axis=1)
    return pref.reshape(-1)


def distinct_coalitions(keep_masks):
    """Prefix coalitions of different permutations coincide as SETS (the empty and the full set once per permutation,
    the 32 single-region sets, ...): of the 3300 coalitions of 100 permutations over 32 regions about 2950 are distinct,
    of 33 000 about 27 700.  Equal sets are equal clouds, hence equal logits (a coalition's logits do not depend on
    what else is in a launch - tested bitwise), so each distinct set is evaluated once and its row is replicated.
    Returns (unique masks, inverse index) with ``unique[inverse] == keep_masks``."""
    k = np.asarray(keep_masks, dtype=np.uint64).reshape(-1)
    if k.size == 0:
        return k, np.zeros((0,), dtype=np.int64)
    # one stable sort + a flag scan (np.unique(..., return_inverse=True) takes 40-180 ms on 33 000 masks here, more than the
    # forward passes it saves; this takes 2 ms)
    idx = np.argsort(k, kind="stable")
    ks = k[idx]
    first = np.empty(k.size, dtype=bool)
    first[0] = True
    first[1:] = ks[1:] != ks[:-1]
    inv = np.empty(k.size, dtype=np.int64)
    inv[idx] = np.cumsum(first) - 1
    return ks[first], inv


def coalition_logits_capped(model, clouds, centers, rid, keep, num_regions, cap):
    """model.coalition_logits on one source cloud, at most ``cap`` coalitions per launch (cap None / 0: one launch)."""
    if not cap or keep.numel() <= cap:
        return model.coalition_logits(clouds, centers, rid, keep, None, num_regions=num_regions, validate=False)
    return torch.cat([model.coalition_logits(clouds, centers, rid, keep[i:i + cap].contiguous(), None, num_regions=num_regions,
                                             validate=False) for i in range(0, keep.numel(), cap)], dim=0)


def shapley_logits(model, data, lbl, region_id, orders, args, center=None):
    """Logits of all prefix coalitions of ``orders`` ((S,R) ndarray) for one cloud (1,N,3), in the reference's row order
    (row o*(R+1)+i keeps orders[o][:i]).  Models with a coalition entry point take the region bit masks directly;
    for the others the mask kernel writes the distinct coalitions' clouds in batches of at least
    ``args.shapley_batch_size`` permutations' worth."""
    dev = data.device
    r = args.num_regions
    if center is None:
        center = torch.mean(data, dim=1)  # (1,3), tools/final_common.py:80
    rid = hip_ops.region_ids(region_id, dev, r)   # validated here: the model calls below skip their own check
    uniq, inv = distinct_coalitions(prefix_keep_masks(orders, r))
    work.add(inv.size, uniq.size)
    inv_t = torch.from_numpy(inv.astype(np.int64)).to(dev)
    keep = hip_ops.masks_to_tensor(uniq, dev)
    strict = getattr(args, "strict_batch_cap", False)
    knob = getattr(args, "shapley_batch_size", 1) * (r + 1)
    if hasattr(model, "coalition_logits"):
        logits = coalition_logits_capped(model, data.contiguous(), center.reshape(1, 3).contiguous(), rid.reshape(1, -1), keep, r,
                                         knob if strict else None)
        return logits.index_select(0, inv_t)
    # config.py's knob is a floor unless strict_batch_cap is set: rows are independent in eval mode, so larger launches give
    # the same logits (stage 1 sets no batch size: the reference evaluates one permutation per forward there,
    # final_shapley_value.py:138-144)
    bs = knob if strict else max(knob, getattr(model, "preferred_clouds_per_call", 0))
    chunks = []
    points_api = hasattr(model, "forward_points")  # consumes (B,N,3) directly: no transpose
    for i in range(0, keep.numel(), bs):
        x = hip_ops.mask_coalitions(data[0].contiguous(), rid, keep[i:i + bs].contiguous(), center.reshape(3).contiguous(),
                                    channel_first=not points_api)
        chunks.append(model.forward_points(x) if points_api else model(x))
    return torch.cat(chunks, dim=0).index_select(0, inv_t)


def shap_sampling_all_regions_batch(model, data_disturb, lbl, region_id, load_order_list, args):
    """tools/final_common.py:64-103.  Returns (region_shap_value (R,) float64 ndarray,
    all_logits_this_pose (num_samples*(R+1), C) tensor).  As in the reference only
    ``(num_samples // bs) * bs`` permutations are evaluated while the sum is divided by
    ``num_samples`` (:78,:97); a batch size that does not divide num_samples is rejected up front
    instead of tripping the assert at :99."""
    bs = args.shapley_batch_size
    if args.num_samples % bs != 0:
        raise IqError("shapley_batch_size=%d does not divide num_samples=%d (tools/final_common.py:78,99)"
                      % (bs, args.num_samples))
    t_start = time.time()
    orders = np.asarray(load_order_list[:args.num_samples])
    with torch.no_grad():
        logits = shapley_logits(model, data_disturb, lbl, region_id, orders, args)
        v = get_reward(logits, lbl, args)
        phi_sum, _, _ = hip_ops.shapley_accum(v, hip_ops.as_i32(orders, data_disturb.device))
        region_shap_value = phi_sum.cpu().numpy()  # the only device->host sync of the pose
    region_shap_value /= args.num_samples
    assert logits.size()[0] == args.num_samples * (args.num_regions + 1)
    if getattr(args, "verbose", True):
        print("done time: ", time.time() - t_start)
    return region_shap_value, logits
