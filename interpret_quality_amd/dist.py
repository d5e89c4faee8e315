"""Multi-GPU sharding of the embarrassingly parallel outer loops.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo" in the CPU
tests).  Units of work (permutations, poses, region pairs) are split into contiguous, nearly equal
ranges; there is NO collective on the data path except one all-gather of per-coalition logits (or
rewards) per (cloud, pose | setting) so that rank 0 can write the artefacts and reduce in the
reference's order.  Payloads are small (<= a few MB), so the gather is latency-bound: one padded
all_gather, never one per batch.
"""
import os

import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def init_from_env(device_type="cuda"):
    """Initialise from torchrun's environment (RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*).  Returns
    (rank, world, local_rank).  A single process needs no process group."""
    w = int(os.environ.get("WORLD_SIZE", "1"))
    r = int(os.environ.get("RANK", "0"))
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    # IQ_REHEARSAL=1: rehearse the N > 1 code path on a one-GPU box - every rank on device 0, gloo collectives
    # (RCCL needs one device per rank).  For tests only; never set by the drivers' users.
    rehearsal = os.environ.get("IQ_REHEARSAL") == "1"
    if rehearsal:
        lr = 0
    if w > 1 and not dist.is_initialized():
        dist.init_process_group("nccl" if device_type == "cuda" and not rehearsal else "gloo", rank=r, world_size=w)
    return r, w, lr


def shard_range(n, r=None, w=None):
    """Contiguous range of rank r out of n units: the first n % w ranks get one extra unit
    (300 pairs over 8 ranks -> 38,38,38,38,37,37,37,37)."""
    r = rank() if r is None else r
    w = world() if w is None else w
    base, extra = divmod(n, w)
    lo = r * base + min(r, extra)
    return lo, lo + base + (1 if r < extra else 0)


def shard_counts(n, w=None):
    w = world() if w is None else w
    return [shard_range(n, r, w)[1] - shard_range(n, r, w)[0] for r in range(w)]


def all_gather_rows(t, n_total):
    """Each rank holds rows shard_range(n_total) of a (n_total, ...) tensor; returns the full tensor
    on every rank.  Pads to equal chunks for a single all_gather."""
    w = world()
    if w == 1:
        return t
    counts = shard_counts(n_total, w)
    cap = max(counts)
    pad = torch.zeros((cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    out = [torch.empty_like(pad) for _ in range(w)]
    dist.all_gather(out, pad)
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)


def sharded_rows(n_total, compute_fn):
    """Run ``compute_fn(lo, hi) -> (hi-lo, ...) tensor`` on this rank's contiguous share of n_total
    units and return the full (n_total, ...) tensor on every rank (one all-gather)."""
    lo, hi = shard_range(n_total)
    return all_gather_rows(compute_fn(lo, hi), n_total)


def barrier():
    if world() > 1:
        dist.barrier()
