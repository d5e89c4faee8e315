#!/usr/bin/env python3
"""BASELINE.json configs[4]: ALL models x datasets through both pipelines (scripts/exp_shapley.sh and
scripts/exp_interaction.sh), spread over the GPUs of one node.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/sweep.py [--synthetic]
    python tools/sweep.py --models pointnet,dgcnn --datasets modelnet10 --synthetic          # one GPU

The reference leaves this to the user: one (model, dataset) per shell script, clouds looped serially, `device_id` edited by
hand to use several GPUs (README.md:87, scripts/exp_shapley.sh:2-4, tools/final_common.py:134).  Every (model, dataset,
cloud) is an independent object (third bullet), so whole clouds are handed to ranks and each rank runs the
complete per-cloud pipeline IN ONE PROCESS - no per-stage start-up (1.4 s x 8 scripts, which is half of PointNet's per-cloud
time), no collective on the data path, one barrier per phase:

  phase 0  the FPS region centres of each dataset (final_save_fps.py), one rank per dataset
  phase A  per (model, dataset, cloud): final_shapley_value, final_{trans,rotate,scale}_center_enum_all,
           final_smoothness_center_enum_all
  phase B  per (model, dataset): final_gen_pair - its pair / context draws run on from cloud to cloud on ONE host generator
           and depend on the rotation sweeps of all earlier clouds, so it cannot be split by cloud
  phase C  per (model, dataset, selected cloud): final_point_binary_interaction_logits, final_cal_interactions

Units go to ranks longest-first onto the least loaded rank (static per-model cost estimates below; a PointConv cloud costs
six PointNet clouds), which needs no communication and is deterministic.  The stage code is the stage scripts' own
(interpret_quality_amd.*: run / test with args.cloud_subset = {cloud}), inside dist.local_only(), so each rank computes and
writes its clouds itself; every script's set_random(seed) and the draws of the clouds before the unit's own are replayed, so the
artefacts are bit-identical to running the per-stage scripts (tests/test_sweep_gpu.py).
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed  # noqa: E402

from interpret_quality_amd import dist as iqdist  # noqa: E402
from interpret_quality_amd import gen_pair, interaction, pose_sweep, shapley_stage, smoothness  # noqa: E402
from interpret_quality_amd.final_util import DATASETS, MODELNET_INTER_SELECTED_SAMPLE, MODELS, SHAPENET_INTER_SELECTED_SAMPLE  # noqa: E402

# seconds per cloud on one MI355X at the reference's sizes (profiles/r03_e2e_times.txt minus the per-script start-up):
# only the RATIOS matter - they balance the assignment
COST_A = {"pointnet": 6.0, "pointnet2": 29.3, "pointconv": 35.0, "dgcnn": 14.8, "gcnn": 10.1, "gcnn_adv": 10.1}
COST_C = {"pointnet": 5.3, "pointnet2": 36.2, "pointconv": 33.6, "dgcnn": 16.4, "gcnn": 11.6, "gcnn_adv": 11.6}
STAGES_A = ("shapley_value", "trans", "rotate", "scale", "smoothness")
STAGES_C = ("logits", "cal")
ALL_STAGES = STAGES_A + ("gen_pair",) + STAGES_C


def assign(units, costs, world):
    """Longest processing time first onto the least loaded rank; ties by position, so every rank computes the same table.
    -> [rank of unit k]."""
    load = [0.0] * world
    owner = [0] * len(units)
    for k in sorted(range(len(units)), key=lambda k: (-costs[k], k)):
        r = min(range(world), key=lambda r: (load[r], r))
        owner[k] = r
        load[r] += costs[k]
    return owner


def parse(argv=None):
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    p.add_argument("--models", default=",".join(MODELS))
    p.add_argument("--datasets", default=",".join(DATASETS))
    p.add_argument("--stages", default=",".join(ALL_STAGES), help="subset of " + ",".join(ALL_STAGES))
    p.add_argument("--mode", default="rotate", help="interaction pipeline's pose family (final_gen_pair.py --mode)")
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--synthetic", action="store_true", help="synthetic clouds / weights (no datasets or checkpoints offline)")
    p.add_argument("--num_clouds", type=int, default=30)
    # the stage scripts' additive size flags, for rehearsals
    p.add_argument("--num_samples_save", type=int, default=None)
    p.add_argument("--num_pairs_random", type=int, default=None)
    p.add_argument("--num_save_context_max", type=int, default=None)
    p.add_argument("--quiet", type=int, default=1, help="swallow the stage scripts' per-pose prints")
    p.add_argument("--resume", action="store_true", help="skip the units a previous run with the same settings completed "
                   "(markers under checkpoints/.sweep/, written after a unit's last stage)")
    return p.parse_args(argv)


def marker_path(phase, unit):
    """checkpoints/.sweep/<phase>/<unit>.json: written (atomically) once every stage of the unit has finished."""
    name = "-".join(str(x) for x in (unit if isinstance(unit, tuple) else (unit,)))
    return os.path.join("checkpoints", ".sweep", phase, name + ".json")


def unit_done(phase, unit, settings):
    """True if a previous run completed this unit under the same settings (the inter-stage artefacts are the checkpoint the
    reference's pipelines have; a unit starts from set_random(seed) and replays the draws of the clouds before
    its own, so skipping one changes nothing for the others)."""
    try:
        with open(marker_path(phase, unit)) as f:
            return json.load(f) == settings
    except (OSError, ValueError):
        return False


def mark_done(phase, unit, settings):
    path = marker_path(phase, unit)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    tmp = "%s.tmp%d" % (path, os.getpid())
    with open(tmp, "w") as f:
        json.dump(settings, f)
    os.replace(tmp, path)


def stage_argv(a, model, dataset, extra=()):
    v = ["--model", model, "--dataset", dataset, "--seed", str(a.seed), "--num_clouds", str(a.num_clouds)] + list(extra)
    return v + (["--synthetic"] if a.synthetic else [])


class Runner:
    """One process = one GPU: the caches every unit of this rank shares (models, parsed datasets)."""

    def __init__(self, a, device):
        self.a, self.device = a, device
        self.model_cache, self.data_cache = {}, {}

    def release_models(self):
        """Drop the cached models (and with them their engines' workspaces) and hand the memory back to the device."""
        import gc
        self.model_cache.clear()
        gc.collect()
        torch.cuda.empty_cache()

    def prepare(self, args, cloud=None):
        args.model_cache, args.data_cache = self.model_cache, self.data_cache
        args.cloud_subset = None if cloud is None else {cloud}
        shapley_stage.prepare_args(args, self.device)   # folders, set_random(seed), model arguments - as each script's main does
        return args

    def run(self, stage, model, dataset, cloud=None):
        a = self.a
        sv = lambda *extra: stage_argv(a, model, dataset, extra)  # noqa: E731
        inter = ["--mode", a.mode, "--gen_pair_seed", str(a.seed)]
        sizes = []
        if a.num_pairs_random is not None:
            sizes += ["--num_pairs_random", str(a.num_pairs_random)]
        if a.num_save_context_max is not None:
            sizes += ["--num_save_context_max", str(a.num_save_context_max)]
        if stage == "fps":
            args = self.prepare(shapley_stage.make_args(sv()))
            if not os.path.exists(shapley_stage.fps_index_path(args)):
                shapley_stage.save_fps(args)
        elif stage == "shapley_value":
            extra = ["--num_samples_save", str(a.num_samples_save)] if a.num_samples_save is not None else []
            shapley_stage.test(self.prepare(shapley_stage.make_args(sv(*extra)), cloud))
        elif stage in ("trans", "rotate", "scale"):
            pose_sweep.run(self.prepare(pose_sweep.make_args(stage, sv()), cloud))
        elif stage == "smoothness":
            smoothness.run(self.prepare(smoothness.make_args(sv()), cloud))
        elif stage == "gen_pair":
            gen_pair.run(self.prepare(gen_pair.make_args(sv("--mode", a.mode, *sizes))))
        elif stage == "logits":
            interaction.run_logits(self.prepare(interaction.make_args(False, sv(*inter, *sizes)), cloud))
        elif stage == "cal":
            interaction.cal_interaction(self.prepare(interaction.make_args(True, sv(*inter, *sizes)), cloud))
        else:
            raise ValueError(stage)


@iqdist.record   # the process group is shut down (barrier, destroy) on every exit path; a failing rank's traceback is kept
def main(argv=None):
    a = parse(argv)
    models = [m for m in a.models.split(",") if m]
    datasets = [d for d in a.datasets.split(",") if d]
    stages = [s for s in a.stages.split(",") if s]
    for name, pool in (("model", MODELS), ("dataset", DATASETS)):
        bad = [x for x in (models if name == "model" else datasets) if x not in pool]
        if bad:
            raise SystemExit("unknown %s(s): %s" % (name, ", ".join(bad)))
    if [s for s in stages if s not in ALL_STAGES]:
        raise SystemExit("unknown stage(s): %s" % ", ".join(s for s in stages if s not in ALL_STAGES))
    rank, world, local_rank = iqdist.init_from_env("cuda")
    if not torch.cuda.is_available():
        raise SystemExit("tools/sweep.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    runner = Runner(a, torch.device("cuda", local_rank))
    sink = io.StringIO() if a.quiet else None
    log = {"rank": rank, "world": world, "phases": {}}
    t_all = time.time()

    def phase(name, units, costs, fn, phase_stages=()):
        """units of this phase -> the ones this rank owns, run in unit order; then the phase barrier."""
        t0 = time.time()
        settings = {"seed": a.seed, "mode": a.mode, "num_clouds": a.num_clouds, "synthetic": bool(a.synthetic),
                    "num_samples_save": a.num_samples_save, "num_pairs_random": a.num_pairs_random,
                    "num_save_context_max": a.num_save_context_max, "stages": [s for s in phase_stages if s in stages]}
        skipped = 0
        if a.resume:   # before the assignment: the units that are left are balanced over the ranks
            # rank 0 looks (a fast rank may finish a unit of this phase before a slow one has looked) and tells the others
            done = [[unit_done(name, u, settings) for u in units] if rank == 0 else None]
            if world > 1:
                torch.distributed.broadcast_object_list(done, src=0)
            left = [(u, c) for u, c, dn in zip(units, costs, done[0]) if not dn]
            skipped = len(units) - len(left)
            units, costs = [u for u, _ in left], [c for _, c in left]
        owner = assign(units, costs, world)
        mine = [u for u, r in zip(units, owner) if r == rank]
        # a rank's units one model after the other: each family's engine holds a workspace of tens of GB (sized for thousands of
        # coalitions per launch), so only ONE is alive at a time - the order inside a phase is free, every unit writes its own files
        mine.sort(key=lambda u: (u[0] if isinstance(u, tuple) else "",) + (tuple(u[1:]) if isinstance(u, tuple) else (u,)))
        current = None
        for u in mine:
            model = u[0] if isinstance(u, tuple) else None
            if model != current:
                runner.release_models()
                current = model
            with iqdist.local_only(), (contextlib.redirect_stdout(sink) if sink is not None else contextlib.nullcontext()):
                fn(u)
            mark_done(name, u, settings)
            if sink is not None:
                sink.seek(0)
                sink.truncate()
        busy = time.time() - t0
        runner.release_models()
        iqdist.group_barrier()
        log["phases"][name] = {"units": len(units), "skipped": skipped, "mine": len(mine), "busy_s": round(busy, 3),
                               "wall_s": round(time.time() - t0, 3)}
        if rank == 0:
            print("[sweep] phase %s: %d units over %d rank(s)%s, rank 0 ran %d in %.1f s (phase wall %.1f s)"
                  % (name, len(units), world, " (%d done before, skipped)" % skipped if skipped else "", len(mine), busy,
                     time.time() - t0), flush=True)

    selected = {"modelnet10": MODELNET_INTER_SELECTED_SAMPLE, "shapenet": SHAPENET_INTER_SELECTED_SAMPLE}
    md = [(m, d) for d in datasets for m in models]
    if any(s in stages for s in STAGES_A):
        phase("0_fps", datasets, [1.0] * len(datasets), lambda d: runner.run("fps", models[0], d), ("fps",))
        units = [(m, d, c) for m, d in md for c in range(a.num_clouds)]
        phase("A_shapley", units, [COST_A[m] for m, _, _ in units],
              lambda u: [runner.run(s, u[0], u[1], u[2]) for s in STAGES_A if s in stages], STAGES_A)
    if "gen_pair" in stages:
        phase("B_gen_pair", md, [1.0] * len(md), lambda u: runner.run("gen_pair", u[0], u[1]), ("gen_pair",))
    if any(s in stages for s in STAGES_C):
        units = [(m, d, c) for m, d in md for c in selected[d] if c < a.num_clouds]
        phase("C_interaction", units, [COST_C[m] for m, _, _ in units],
              lambda u: [runner.run(s, u[0], u[1], u[2]) for s in STAGES_C if s in stages], STAGES_C)
    log["total_s"] = round(time.time() - t_all, 3)
    if rank == 0:
        print("[sweep] done: %d (model, dataset) pairs x %d clouds on %d GPU(s) in %.1f s" % (len(md), a.num_clouds, world, log["total_s"]))
        print(json.dumps({"sweep": log}), flush=True)


if __name__ == "__main__":
    main()
