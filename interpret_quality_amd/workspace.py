"""Sizing the engines' device workspaces from the memory that is actually free.

Every model family's engine evaluates coalitions in launches of up to ``max_clouds_per_call`` clouds and keeps ONE growable
workspace for them (3-10 MB per cloud: 13-39 GB at 4096).  On an empty 288 GB MI355X that is nothing; with a second stage
process on the same GPU (the reference's documented way of using one card for two scripts, README.md:87), another family's
engine still alive, or a smaller part it is an out-of-memory error (round 3: `Tried to allocate 17.39 GiB ... 99.80 GiB
allocated`, tools/sweep.py's two-rank rehearsal).  So: the launch size is the largest one whose workspace fits
``free + the allocator's cached blocks + the engine's current workspace - reserve``, re-evaluated per call, and a launch that
still fails to allocate is retried at half the size.  A coalition's logits do not depend on what else is in the launch
(tested bitwise), so the results are the same whatever the step.
"""
import gc
import os

import torch

from ._lib import IqError

RESERVE_BYTES = int(os.environ.get("IQ_WS_RESERVE_MB", "1024")) << 20   # left for logits, index tensors, the next stage's inputs
MAX_HALVINGS = 4
STATS = {"fitted_below_cap": 0, "oom_retries": 0, "last_step": 0}       # for the tests and the sweep's log


def available_bytes(device, held=0):
    """Bytes a new workspace could get: free device memory + what torch's caching allocator holds without using (it is handed
    back on an allocation failure) + the workspace this engine would release first - the reserve."""
    free, _ = torch.cuda.mem_get_info(device)
    cached = torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
    return int(free) + int(cached) + int(held) - RESERVE_BYTES


def fit_step(want, need_fn, device, held=0):
    """Largest step <= want with need_fn(step) <= available_bytes (need_fn monotone in step)."""
    if want <= 0:
        return 0
    if held >= need_fn(want):       # the workspace the engine holds already covers it
        return want
    room = available_bytes(device, held)
    if need_fn(want) <= room:
        return want
    if need_fn(1) > room:
        raise IqError("not enough free device memory for one cloud's workspace: need %d bytes, %d available after a reserve of %d"
                      % (need_fn(1), room, RESERVE_BYTES))
    lo, hi = 1, want                # invariant: need_fn(lo) fits, need_fn(hi) does not
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if need_fn(mid) <= room:
            lo = mid
        else:
            hi = mid
    STATS["fitted_below_cap"] += 1
    return lo


def ensure(eng, nbytes):
    """The engine's workspace, at least nbytes; the old one is dropped BEFORE the new one is allocated (never both alive)."""
    if eng._ws is None or eng._ws.numel() < nbytes:
        eng._ws = None
        if hasattr(eng, "_tab"):
            eng._tab = {}           # whatever the old workspace cached (PointConv's per-cloud tables) is gone
        eng._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=eng.device)
    return eng._ws


def run_in_steps(eng, total, cap, need_fn, call):
    """``call(lo, hi) -> (hi-lo, ...) tensor`` over [0, total) in launches of the fitted step; see the module docstring."""
    if total <= 0:
        return call(0, 0)
    held = eng._ws.numel() if eng._ws is not None else 0
    step = fit_step(min(cap, total), need_fn, eng.device, held)
    STATS["last_step"] = step
    out, lo, halvings = [], 0, 0
    while lo < total:
        hi = min(lo + step, total)
        try:
            out.append(call(lo, hi))
        except torch.OutOfMemoryError:
            if step <= 1 or halvings >= MAX_HALVINGS:
                raise
            # someone else took the memory between the look and the allocation (another process on this GPU): give everything
            # back, halve, try this chunk again
            eng._ws = None
            gc.collect()
            torch.cuda.empty_cache()
            step = max(1, step // 2)
            halvings += 1
            STATS["oom_retries"] += 1
            STATS["last_step"] = step
            continue
        lo = hi
    if not out:
        return None
    return out[0] if len(out) == 1 else torch.cat(out, dim=0)
