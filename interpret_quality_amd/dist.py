"""Multi-GPU sharding of the embarrassingly parallel outer loops.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo" in the CPU
tests).  Units of work (permutations, poses, region pairs) are split into contiguous, nearly equal
ranges; there is NO collective on the data path except one all-gather of per-coalition logits (or
rewards) per (cloud, pose | setting) so that rank 0 can write the artefacts and reduce in the
reference's order.  Payloads are small (<= a few MB), so the gather is latency-bound: one padded
all_gather, never one per batch.
"""
import contextlib
import os

import torch
import torch.distributed as dist

_LOCAL_ONLY = [False]
GATHER_EVENTS = None   # a list: all_gather_rows appends (start, end) HIP events around its collective (bench.py --scaling strong)


@contextlib.contextmanager
def local_only():
    """Inside this context the stage code sees a world of ONE rank (it computes every unit it is given and writes its
    artefacts itself, no collective) although a process group exists: the mode of the outer sweep (tools/sweep.py), which
    hands whole (model, dataset, cloud) units to ranks and needs the group only for its barriers (SURVEY.md 8e bullet 3)."""
    prev = _LOCAL_ONLY[0]
    _LOCAL_ONLY[0] = True
    try:
        yield
    finally:
        _LOCAL_ONLY[0] = prev


def group_rank():
    """Rank / size of the real process group, whatever local_only() says."""
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def group_world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def world():
    return 1 if _LOCAL_ONLY[0] else group_world()


def rank():
    return 0 if _LOCAL_ONLY[0] else group_rank()


def cloud_selected(args, index):
    """Stage loops run over the 30 clouds of a dataset; ``args.cloud_subset`` (a set of indices, additive) restricts a call
    to some of them - the unit of the outer sweep.  None / absent: all clouds, as in the reference."""
    subset = getattr(args, "cloud_subset", None)
    return subset is None or index in subset


DEFAULT_TIMEOUT_S = 1800   # the per-stage scripts meet once per cloud, seconds to minutes apart


def init_from_env(device_type="cuda", timeout_s=None):
    """Initialise from the launcher's environment (RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*; torchrun or launch.self_launch).
    Returns (rank, world, local_rank).  A single process needs no process group.  ``timeout_s``: collective timeout of the
    group (default IQ_DIST_TIMEOUT_S or 30 minutes, so a hung or dead peer is noticed; tools/sweep.py, whose ranks meet only
    at phase barriers hours apart, asks for its own)."""
    w = int(os.environ.get("WORLD_SIZE", "1"))
    r = int(os.environ.get("RANK", "0"))
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    # IQ_REHEARSAL=1: rehearse the N > 1 code path on a one-GPU box - every rank on device 0, gloo collectives
    # (RCCL needs one device per rank).  For tests only; never set by the drivers' users.
    rehearsal = os.environ.get("IQ_REHEARSAL") == "1"
    if rehearsal:
        lr = 0
    # IQ_FORCE_DIST=1: create the process group even for a single rank, so that the RCCL communicator, the barrier and
    # the all-gather of the N > 1 path are exercised on a one-GPU box (a world of 1 is otherwise collective-free).
    if (w > 1 or force_dist()) and not dist.is_initialized():
        from . import launch
        launch.ensure_rendezvous()   # the launcher's MASTER_PORT; a free port for a forced single-rank group
        # ranks meet at a gather per cloud, up to minutes apart: the backend's default collective timeout (10 minutes) is too
        # close to ordinary load imbalance, 12 hours (round 3) hides a dead peer
        import datetime
        if timeout_s is None:
            timeout_s = int(os.environ.get("IQ_DIST_TIMEOUT_S", str(DEFAULT_TIMEOUT_S)))
        timeout = datetime.timedelta(seconds=int(timeout_s))
        dist.init_process_group("nccl" if device_type == "cuda" and not rehearsal else "gloo", rank=r, world_size=w, timeout=timeout)
    return r, w, lr


def force_dist():
    return os.environ.get("IQ_FORCE_DIST") == "1"


def collectives_on():
    """True when results travel through the collectives: more than one rank, or a forced single-rank group."""
    return (not _LOCAL_ONLY[0]) and dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_dist())


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
    if not collectives_on():
        return t
    counts = shard_counts(n_total, w)
    cap = max(counts)
    tail = tuple(t.shape[1:])
    if t.shape[0] == cap:
        pad = t.contiguous()
    else:
        pad = torch.zeros((cap,) + tail, dtype=t.dtype, device=t.device)
        pad[:t.shape[0]] = t
    # ONE preallocated (w * cap, ...) receive buffer: all_gather_into_tensor writes every rank's chunk in place (the
    # list form of all_gather costs an extra copy per rank on RCCL, and these payloads are latency-bound)
    out = torch.empty((w * cap,) + tail, dtype=t.dtype, device=t.device)
    if GATHER_EVENTS is not None and t.is_cuda:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        dist.all_gather_into_tensor(out, pad)
        ev[1].record()
        GATHER_EVENTS.append(ev)
    else:
        dist.all_gather_into_tensor(out, pad)
    if all(c == cap for c in counts):
        return out
    return torch.cat([out[r * cap:r * cap + c] for r, c in enumerate(counts)], dim=0)


def num_classes_of(model, default=10):
    """Class count of a model's logits (the trailing dimension an empty shard must still agree on)."""
    for obj in (model, getattr(model, "weights", None)):
        n = getattr(obj, "num_classes", None) or getattr(obj, "output_channels", None)
        if n:
            return int(n)
    sd = getattr(model, "state_dict", None)
    if sd is not None:
        last = [v for k, v in sd().items() if k.endswith("weight") and v.dim() == 2]
        if last:
            return int(last[-1].shape[0])
    return default


def sharded_rows(n_total, compute_fn):
    """Run ``compute_fn(lo, hi) -> (hi-lo, ...) tensor`` on this rank's contiguous share of n_total
    units and return the full (n_total, ...) tensor on every rank (one all-gather)."""
    lo, hi = shard_range(n_total)
    return all_gather_rows(compute_fn(lo, hi), n_total)


def _group_barrier():
    """``dist.barrier()`` of the default group.  RCCL runs a barrier as a collective on a device: name this rank's device
    (the one the stage selected) instead of leaving the backend to guess it from the rank number."""
    if dist.get_backend() == "nccl":
        dist.barrier(device_ids=[torch.cuda.current_device()])
    else:
        dist.barrier()


def barrier():
    if collectives_on():
        _group_barrier()


def group_barrier():
    """Barrier of the real group (the sweep's phase boundaries), whatever local_only() says."""
    if dist.is_available() and dist.is_initialized():
        _group_barrier()


def shutdown(ok=True):
    """Leave the process group in an orderly way: after a successful stage every rank waits for the others (so that no
    rank's sockets / communicator disappear under a peer that is still inside its last gather or still writing the
    artefacts) and then destroys the group.  A process that exits with a live group leaves the teardown of the backend's
    worker threads to interpreter exit, where an exception in one of them is a C++ terminate = SIGABRT after the work is
    done; every stage main goes through here (``record`` below).  After a failure there is no barrier (the peers may
    never reach it): the group is dropped and the exception travels on."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    try:
        if ok:
            if torch.cuda.is_available() and torch.cuda.is_initialized():
                torch.cuda.synchronize()
            _group_barrier()
    finally:
        dist.destroy_process_group()


def record(fn):
    """Decorator for the stage scripts' main(): (1) the process group the stage created is shut down on every exit path
    (``shutdown``); (2) under a multi-rank launch a failing rank's exception (with its traceback) goes to torchrun's
    error file and to stderr BEFORE the launcher tears the other ranks down, so the cause of a child death is on record
    (torch.distributed.elastic's ``record``)."""
    import functools
    import sys
    import traceback

    inner = fn
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        try:
            from torch.distributed.elastic.multiprocessing.errors import record as _record
            inner = _record(fn)
        except ImportError:
            pass

    @functools.wraps(fn)
    def wrapper(*a, **k):
        owns = not (dist.is_available() and dist.is_initialized())  # a caller's group (sweep driver, tests) is the caller's
        try:
            out = inner(*a, **k)
        except BaseException:
            if int(os.environ.get("WORLD_SIZE", "1")) > 1:
                sys.stderr.write("[rank %s] stage failed:\n%s\n" % (os.environ.get("RANK", "?"), traceback.format_exc()))
                sys.stderr.flush()
            if owns:
                shutdown(ok=False)
            raise
        if owns:
            shutdown(ok=True)
        return out
    return wrapper
