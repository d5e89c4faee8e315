#!/usr/bin/env python3
"""BASELINE.json configs[4]: ALL models x datasets through both pipelines (scripts/exp_shapley.sh and
scripts/exp_interaction.sh), spread over the GPUs of one node.

    python tools/sweep.py --gpus 8 [--synthetic]            # starts its own 8 ranks (interpret_quality_amd/launch.py)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/sweep.py [--synthetic]
    python tools/sweep.py --models pointnet,dgcnn --datasets modelnet10 --synthetic          # one GPU

The reference leaves this to the user: one (model, dataset) per shell script, clouds looped serially, `device_id` edited by
hand to use several GPUs (README.md:87, scripts/exp_shapley.sh:2-9, scripts/exp_interaction.sh:2-7,
tools/final_common.py:134).  Every (model, dataset, cloud) is an independent object (third bullet), so whole
clouds are handed to ranks and each rank runs the complete per-cloud pipeline IN ONE PROCESS - no per-stage start-up (1.4 s x 8
scripts, which is half of PointNet's per-cloud time), no collective on the data path, one barrier per phase:

  phase 0  the FPS region centres of each dataset (final_save_fps.py), one rank per dataset
  phase A  per (model, dataset, cloud): final_shapley_value, final_{trans,rotate,scale}_center_enum_all,
           final_smoothness_center_enum_all
  phase B  per (model, dataset): final_gen_pair - its pair / context draws run on from cloud to cloud on ONE host generator
           and depend on the rotation sweeps of all earlier clouds, so it cannot be split by cloud
  phase C  per (model, dataset, selected cloud): final_point_binary_interaction_logits, final_cal_interactions

Units are PULLED: the units of a phase stand in one queue, heavy model families first and one family after the other (so that a
rank keeps one family's engine alive at a time), and every rank takes the next index from a counter in the process group's
c10d store when it has finished its unit - a slow rank or an expensive real cloud delays nobody (round 3 assigned units up
front from a static cost table).  Which rank runs a unit changes nothing in its files: the stage code is the stage scripts' own
(interpret_quality_amd.*: run / test with args.cloud_subset = {cloud}), inside dist.local_only(), each unit starting from the
script's set_random(seed) and replaying the draws of the clouds before its own, so the artefacts are bit-identical to running
the per-stage scripts (tests/test_sweep_gpu.py).

The last line rank 0 prints is ONE JSON object in bench.py's schema: `value` = coalitions of the whole sweep / its wall time,
with the coalitions counted as the stages run (interpret_quality_amd/work.py), per-phase coalitions/s, per-rank busy / wall
and the imbalance.
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
from interpret_quality_amd import gen_pair, interaction, launch, pose_sweep, shapley_stage, smoothness, work  # noqa: E402
from interpret_quality_amd.final_util import DATASETS, get_folder_name_list, MODELNET_INTER_SELECTED_SAMPLE, MODELS, SHAPENET_INTER_SELECTED_SAMPLE  # noqa: E402

# rough seconds per cloud on one MI355X at the reference's sizes: ONLY the order of the queue comes from them (heavy families
# first, the light ones fill the tail); the balance comes from pulling
HINT_A = {"pointnet": 6.0, "pointnet2": 29.3, "pointconv": 35.0, "dgcnn": 14.8, "gcnn": 10.1, "gcnn_adv": 10.1}
HINT_C = {"pointnet": 5.3, "pointnet2": 36.2, "pointconv": 33.6, "dgcnn": 16.4, "gcnn": 11.6, "gcnn_adv": 11.6}
STAGES_A = ("shapley_value", "trans", "rotate", "scale", "smoothness")
STAGES_C = ("logits", "cal")
ALL_STAGES = STAGES_A + ("gen_pair",) + STAGES_C
SWEEP_TIMEOUT_S = 43200    # ranks meet only at the phase barriers, up to hours apart at real sizes
RUNS = [0]                 # run() calls of this process: a second sweep on the same process group starts its counters afresh


def queue_order(units, hint):
    """Order in which a phase's units are pulled: heavy model families first, ONE family after the other (ranks then work on the
    same family at the same time and each keeps one engine alive), inside a family by dataset and cloud."""
    fam = lambda u: u[0] if isinstance(u, tuple) else ""   # noqa: E731
    return sorted(units, key=lambda u: (-hint.get(fam(u), 1.0), fam(u)) + (tuple(u[1:]) if isinstance(u, tuple) else (u,)))


class PullQueue:
    """Shared position in a phase's queue: a counter in the process group's c10d store (TCPStore.add is atomic), so every index
    is handed out exactly once; a single process counts for itself."""

    def __init__(self, name, rank, world):
        self.key, self.rank, self.world, self.local = "next/" + name, rank, world, 0
        self.store = None
        if world > 1:
            from torch.distributed.distributed_c10d import _get_default_store
            self.store = torch.distributed.PrefixStore("iq_sweep", _get_default_store())

    def next(self):
        if self.store is None:
            self.local += 1
            return self.local - 1
        return int(self.store.add(self.key, 1)) - 1

    def publish(self, name, obj):
        if self.store is not None:
            self.store.set("stats/%s/%d" % (name, self.rank), json.dumps(obj))

    def collect(self, name, mine):
        """Every rank's stats record, on rank 0 after the phase barrier."""
        if self.store is None:
            return [mine]
        return [json.loads(self.store.get("stats/%s/%d" % (name, r)).decode()) for r in range(self.world)]


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
    p.add_argument("--gpus", type=int, default=1, help="N > 1 outside a launcher: start N ranks of this script, one per GPU")
    p.add_argument("--timeout_s", type=int, default=SWEEP_TIMEOUT_S, help="collective timeout of the process group (the phase barriers)")
    p.add_argument("--resume", action="store_true", help="skip the units a previous run with the same settings completed "
                   "(markers under checkpoints/.sweep/, written after a unit's last stage)")
    return p.parse_args(argv)


def marker_path(phase, unit):
    """checkpoints/.sweep/<phase>/<unit>.json: written (atomically) once every stage of the unit has finished."""
    name = "-".join(str(x) for x in (unit if isinstance(unit, tuple) else (unit,)))
    return os.path.join("checkpoints", ".sweep", phase, name + ".json")


def unit_done(phase, unit, settings):
    """True if a previous run completed this unit under the same settings AND the files the marker names as the unit's
    witnesses are still there (a cleaned checkpoints/ with a kept checkpoints/.sweep/ must not skip anything).  The
    inter-stage artefacts are the checkpoint the reference's pipelines have; a unit starts from
    set_random(seed) and replays the draws of the clouds before its own, so skipping one changes nothing for the others."""
    try:
        with open(marker_path(phase, unit)) as f:
            m = json.load(f)
    except (OSError, ValueError):
        return False
    witness = m.pop("witness", [])
    return m == settings and all(os.path.exists(w) for w in witness)


def mark_done(phase, unit, settings, witness=()):
    path = marker_path(phase, unit)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    tmp = "%s.tmp%d" % (path, os.getpid())
    with open(tmp, "w") as f:
        json.dump(dict(settings, witness=sorted(set(witness))), f)
    os.replace(tmp, path)


def stage_argv(a, model, dataset, extra=()):
    v = ["--model", model, "--dataset", dataset, "--seed", str(a.seed), "--num_clouds", str(a.num_clouds)] + list(extra)
    return v + (["--synthetic"] if a.synthetic else [])


class Runner:
    """One process = one GPU: the caches every unit of this rank shares (models, parsed datasets)."""

    def __init__(self, a, device):
        self.a, self.device = a, device
        self.model_cache, self.data_cache = {}, {}
        self.witness = []     # key output files of the unit being run (the resume markers name them)

    def release_models(self):
        """Drop the cached models (and with them their engines' workspaces) and hand the memory back to the device."""
        import gc
        if not self.model_cache:
            return                  # nothing cached: no collection (55 ms each, seven times per sweep)
        self.model_cache.clear()
        gc.collect()
        torch.cuda.empty_cache()

    def prepare(self, args, cloud=None):
        args.model_cache, args.data_cache = self.model_cache, self.data_cache
        args.cloud_subset = None if cloud is None else {cloud}
        shapley_stage.prepare_args(args, self.device)   # folders, set_random(seed), model arguments - as each script's main does
        return args

    def _cloud_folder(self, args, cloud):
        return args.exp_folder + "%s/" % get_folder_name_list(args)[cloud]

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
            self.witness.append(shapley_stage.fps_index_path(args))
        elif stage == "shapley_value":
            extra = ["--num_samples_save", str(a.num_samples_save)] if a.num_samples_save is not None else []
            args = self.prepare(shapley_stage.make_args(sv(*extra)), cloud)
            shapley_stage.test(args)
            self.witness.append(self._cloud_folder(args, cloud) + "region_sv_all.npy")
        elif stage in ("trans", "rotate", "scale"):
            args = self.prepare(pose_sweep.make_args(stage, sv()), cloud)
            pose_sweep.run(args)
            self.witness.append(self._cloud_folder(args, cloud) + "%s_all/region_shapley_value.npy" % stage)
        elif stage == "smoothness":
            smoothness.run(self.prepare(smoothness.make_args(sv()), cloud))
        elif stage == "gen_pair":
            args = self.prepare(gen_pair.make_args(sv("--mode", a.mode, *sizes)))
            gen_pair.run(args)
            self.witness.append(args.exp_folder)
        elif stage == "logits":
            args = self.prepare(interaction.make_args(False, sv(*inter, *sizes)), cloud)
            interaction.run_logits(args)
            self.witness.append(self._cloud_folder(args, cloud) + "interaction_seed%d/normal/ratio0_all_logits.pt" % args.gen_pair_seed)
        elif stage == "cal":
            interaction.cal_interaction(self.prepare(interaction.make_args(True, sv(*inter, *sizes)), cloud))
        else:
            raise ValueError(stage)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    a = parse(argv)
    if a.gpus > 1 and not launch.under_launcher():
        # plain `python tools/sweep.py --gpus N`: this process has not touched the GPU; it becomes the parent of N fresh ranks
        raise SystemExit(launch.self_launch(os.path.abspath(__file__), argv, a.gpus))
    return run(a)


@iqdist.record   # the process group is shut down (barrier, destroy) on every exit path; a failing rank's traceback is kept
def run(a, emit=True):
    """The sweep on this rank (the process group exists already or is created from the launcher's environment).  Returns the
    bench-schema record on rank 0 (None elsewhere); ``emit``: rank 0 prints it as the last line."""
    models = [m for m in a.models.split(",") if m]
    datasets = [d for d in a.datasets.split(",") if d]
    stages = [s for s in a.stages.split(",") if s]
    for name, pool in (("model", MODELS), ("dataset", DATASETS)):
        bad = [x for x in (models if name == "model" else datasets) if x not in pool]
        if bad:
            raise SystemExit("unknown %s(s): %s" % (name, ", ".join(bad)))
    if [s for s in stages if s not in ALL_STAGES]:
        raise SystemExit("unknown stage(s): %s" % ", ".join(s for s in stages if s not in ALL_STAGES))
    rank, world, local_rank = iqdist.init_from_env("cuda", timeout_s=a.timeout_s)
    if not torch.cuda.is_available():
        raise SystemExit("tools/sweep.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    runner = Runner(a, torch.device("cuda", local_rank))
    sink = io.StringIO() if a.quiet else None
    phases = {}
    RUNS[0] += 1
    run_tag = "run%d/" % RUNS[0]
    t_all = time.time()

    def phase(name, units, hint, fn, phase_stages=()):
        """The units of this phase, pulled one at a time by whichever rank is free; then the phase barrier."""
        t0 = time.time()
        settings = {"seed": a.seed, "mode": a.mode, "num_clouds": a.num_clouds, "synthetic": bool(a.synthetic),
                    "num_samples_save": a.num_samples_save, "num_pairs_random": a.num_pairs_random,
                    "num_save_context_max": a.num_save_context_max, "stages": [s for s in phase_stages if s in stages]}
        skipped = 0
        if a.resume:
            # rank 0 looks (a fast rank may finish a unit of this phase before a slow one has looked) and tells the others
            done = [[unit_done(name, u, settings) for u in units] if rank == 0 else None]
            if world > 1:
                torch.distributed.broadcast_object_list(done, src=0)
            left = [u for u, dn in zip(units, done[0]) if not dn]
            skipped = len(units) - len(left)
            units = left
        queue = queue_order(units, hint)          # the same list on every rank
        pull = PullQueue(run_tag + name, rank, world)
        mine = {"rank": rank, "units": 0, "busy_s": 0.0, "coalitions": 0, "evaluated": 0, "by_model": {}}
        current = None
        while True:
            k = pull.next()
            if k >= len(queue):
                break
            u = queue[k]
            model = u[0] if isinstance(u, tuple) else None
            if model != current:
                # each family's engine holds a workspace of tens of GB: only ONE is alive at a time
                runner.release_models()
                current = model
            runner.witness = []
            before, tu = work.snapshot(), time.time()
            try:
                with iqdist.local_only(), (contextlib.redirect_stdout(sink) if sink is not None else contextlib.nullcontext()):
                    fn(u)
            except BaseException:
                if sink is not None:   # which cloud, which pose: the stage's own prints up to the failure
                    sys.stderr.write("[sweep] rank %d failed in phase %s unit %r; the stage printed:\n%s\n" % (rank, name, u, sink.getvalue()[-4000:]))
                raise
            dt, did = time.time() - tu, work.since(before)
            mark_done(name, u, settings, runner.witness)
            if sink is not None:
                sink.seek(0)
                sink.truncate()
            mine["units"] += 1
            mine["busy_s"] += dt
            for key in ("coalitions", "evaluated"):
                mine[key] += did[key]
            bm = mine["by_model"].setdefault(model or "-", {"units": 0, "busy_s": 0.0, "coalitions": 0})
            bm["units"] += 1
            bm["busy_s"] += dt
            bm["coalitions"] += did["coalitions"]
        runner.release_models()
        pull.publish(name, mine)
        iqdist.group_barrier()
        wall = time.time() - t0
        if rank == 0:
            per_rank = pull.collect(name, mine)
            busy = [r["busy_s"] for r in per_rank]
            coal = sum(r["coalitions"] for r in per_rank)
            by_model = {}
            for r in per_rank:
                for m, v in r["by_model"].items():
                    t = by_model.setdefault(m, {"units": 0, "busy_s": 0.0, "coalitions": 0})
                    for key in t:
                        t[key] += v[key]
            for v in by_model.values():
                v["coalitions_per_busy_s"] = v["coalitions"] / v["busy_s"] if v["busy_s"] > 0 else 0.0
                v["busy_s"] = round(v["busy_s"], 3)
            mean_busy = sum(busy) / len(busy)
            phases[name] = {"units": len(queue), "skipped": skipped, "wall_s": round(wall, 3), "coalitions": coal,
                            "evaluated": sum(r["evaluated"] for r in per_rank),
                            "coalitions_per_s": coal / wall if wall > 0 else 0.0,
                            "per_rank": [{"rank": r["rank"], "units": r["units"], "busy_s": round(r["busy_s"], 3),
                                          "busy_over_wall": round(r["busy_s"] / wall, 4) if wall > 0 else 0.0,
                                          "coalitions": r["coalitions"]} for r in per_rank],
                            "imbalance_max_over_mean_busy": round(max(busy) / mean_busy, 4) if mean_busy > 0 else 1.0,
                            "idle_share": round(1.0 - sum(busy) / (len(busy) * wall), 4) if wall > 0 else 0.0,
                            "by_model": by_model}
            print("[sweep] phase %s: %d units over %d rank(s)%s, rank 0 ran %d in %.1f s (phase wall %.1f s, %d coalitions, "
                  "busy max/mean %.2f)" % (name, len(queue), world, " (%d done before, skipped)" % skipped if skipped else "",
                                            mine["units"], mine["busy_s"], wall, coal, phases[name]["imbalance_max_over_mean_busy"]),
                  flush=True)

    selected = {"modelnet10": MODELNET_INTER_SELECTED_SAMPLE, "shapenet": SHAPENET_INTER_SELECTED_SAMPLE}
    md = [(m, d) for d in datasets for m in models]
    if any(s in stages for s in STAGES_A):
        phase("0_fps", datasets, {}, lambda d: runner.run("fps", models[0], d), ("fps",))
        units = [(m, d, c) for m, d in md for c in range(a.num_clouds)]
        phase("A_shapley", units, HINT_A, lambda u: [runner.run(s, u[0], u[1], u[2]) for s in STAGES_A if s in stages], STAGES_A)
    if "gen_pair" in stages:
        phase("B_gen_pair", md, HINT_A, lambda u: runner.run("gen_pair", u[0], u[1]), ("gen_pair",))
    if any(s in stages for s in STAGES_C):
        units = [(m, d, c) for m, d in md for c in selected[d] if c < a.num_clouds]
        phase("C_interaction", units, HINT_C, lambda u: [runner.run(s, u[0], u[1], u[2]) for s in STAGES_C if s in stages], STAGES_C)
    total_s = time.time() - t_all
    if rank != 0:
        return None
    coal = sum(p["coalitions"] for p in phases.values())
    sized = {k: getattr(a, k) for k in ("num_samples_save", "num_pairs_random", "num_save_context_max") if getattr(a, k) is not None}
    rec = {
        "metric": "coalitions/sec (rows the reference pushes through the network per second; value_evaluated = the distinct masked "
                  "forward passes the device ran), all models x datasets sweep",
        "value": coal / total_s if total_s > 0 else 0.0,
        "value_evaluated": sum(p["evaluated"] for p in phases.values()) / total_s if total_s > 0 else 0.0, "unit": "coalitions/s", "n_gpus": world, "steps": 1, "warmup": 0,
        "ms_per_step": total_s * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic" if a.synthetic else "files",
        "config": {"workload": "BASELINE configs[4]: %d model(s) x %d dataset(s) x %d cloud(s) through exp_shapley.sh + exp_interaction.sh "
                               "(stages %s)%s" % (len(models), len(datasets), a.num_clouds, ",".join(stages),
                                                  "; reduced sizes %s" % sized if sized else "; the reference's sizes"),
                   "models": models, "datasets": datasets, "num_clouds": a.num_clouds, "stages": stages,
                   "parallelism": "%d rank(s), whole (model, dataset, cloud) units pulled from a shared queue, one barrier per phase, "
                                  "no data-path collective" % world},
        "coalitions": coal, "evaluated": sum(p["evaluated"] for p in phases.values()),
        "note": "coalitions = rows the reference would push through the network (tools/final_common.py:88-93, "
                "final_point_binary_interaction_logits.py:45-56), counted as the stages run; evaluated = the distinct clouds the "
                "device ran; wall time includes model loading, artefact writing and the phase barriers",
        "phases": phases,
    }
    print("[sweep] done: %d (model, dataset) pairs x %d clouds on %d GPU(s) in %.1f s, %d coalitions, %.0f coalitions/s"
          % (len(md), a.num_clouds, world, total_s, coal, rec["value"]))
    if emit:
        print(json.dumps(rec), flush=True)
    return rec


if __name__ == "__main__":
    main()
