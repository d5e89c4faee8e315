"""GPU: tools/sweep.py (BASELINE configs[4]: whole (model, dataset, cloud) units handed to ranks, each rank running the
complete per-cloud pipeline in one process) against the per-stage scripts run one after the other as scripts/exp_shapley.sh and
scripts/exp_interaction.sh do: every artefact file bitwise identical - including the sampled permutations, pairs and contexts,
whose host / device generator streams run on from cloud to cloud and are replayed for the clouds a rank does not own."""
import os
import sys

import pytest

from test_dist_gpu import REPO, _artefacts, _assert_same, _env, _run, _run_chains, _torchrun

pytestmark = pytest.mark.gpu

STAGES = "shapley_value,rotate,scale,smoothness,gen_pair,logits,cal"
SIZES = ["--num_samples_save", "100"]
PAIRS = ["--num_pairs_random", "5", "--num_save_context_max", "3"]


def _per_stage_scripts(work, model, dataset, clouds):
    """The chain of launches of one model: one process per stage, in the order of the two shell scripts."""
    common = ["--model", model, "--dataset", dataset, "--synthetic", "--num_clouds", str(clouds)]
    plan = [("final_shapley_value.py", SIZES), ("final_rotate_center_enum_all.py", []), ("final_scale_center_enum_all.py", []),
            ("final_smoothness_center_enum_all.py", []), ("final_gen_pair.py", PAIRS),
            ("final_point_binary_interaction_logits.py", PAIRS), ("final_cal_interactions.py", PAIRS + ["--device_id", "0"])]
    return [([sys.executable, os.path.join(REPO, script)] + common + extra, work, _env()) for script, extra in plan]


def _record(stdout):
    """The ONE bench-schema JSON line the sweep prints last."""
    import json
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


MODELS, DATASET, CLOUDS = ["pointnet", "gcnn"], "modelnet10", 4      # clouds 0 and 3 are interaction samples (final_util.py:26)


ALL_SIX_FLAGS = ["--synthetic", "--num_clouds", "1", "--stages", "shapley_value,scale", "--num_samples_save", "100", "--resume"]


@pytest.fixture(scope="module")
def background(tmp_path_factory):
    """Three independent chains of launches, side by side (four GPU processes beside this one): the per-stage scripts of the two
    models (the reference side of the first test below) and the all-pairs sweep on two ranks (the second test's run).  Most of a
    launch is process start-up, so this is what keeps the module's wall time down; what each test asserts is unchanged."""
    one = tmp_path_factory.mktemp("scripts")
    six = tmp_path_factory.mktemp("all_six")
    import threading
    chains = [_per_stage_scripts(one, m, DATASET, CLOUDS) for m in MODELS]
    # (the two models share only the FPS index file of the dataset; whichever first stage finds it missing writes it under a
    # temporary name and renames it - shapley_stage.save_fps - so two writers of the same bytes cannot hurt each other)
    # Stage order inside a model as in the shell scripts, except that the scale sweep and the smoothness stage - which read stage
    # 1's files only - run in the third chain behind the all-pairs sweep, once both models' stage 1 is done: three chains of
    # about equal length instead of two of seven launches.
    sv_done = [threading.Event(), threading.Event()]
    main = [[c[0], sv_done[i].set, c[1], c[4], c[5], c[6]] for i, c in enumerate(chains)]            # sv | rotate, gen_pair, logits, cal
    sweep = os.path.join(REPO, "tools", "sweep.py")
    third = [(_torchrun(2, 29735) + [sweep] + ALL_SIX_FLAGS, six, _env(IQ_REHEARSAL="1")), lambda: [e.wait(600) for e in sv_done],
             chains[0][2], chains[0][3], chains[1][2], chains[1][3]]                                      # scale, smoothness of both
    res = _run_chains([main[0], main[1], third])
    return {"scripts": _artefacts(one), "all_six": (res[2][0], six)}


@pytest.fixture(scope="module")
def script_artefacts(background):
    """The per-stage scripts, one process each, once for the parametrisations below."""
    return background["scripts"]


@pytest.mark.parametrize("ranks", [2])   # (one process: tools/r03_e2e.sh runs the sweep that way; the RCCL test below too)
def test_sweep_writes_the_same_artefacts_as_the_per_stage_scripts(tmp_path, script_artefacts, ranks):
    models, dataset, clouds = MODELS, DATASET, CLOUDS
    two = tmp_path / "sweep"
    two.mkdir()
    flags = ["--models", ",".join(models), "--datasets", dataset, "--synthetic", "--num_clouds", str(clouds), "--stages", STAGES] + SIZES + PAIRS
    sweep = os.path.join(REPO, "tools", "sweep.py")
    if ranks == 1:
        r = _run([sys.executable, sweep] + flags, two, _env())
    else:   # two ranks on the one GPU of this box (gloo group: the phase barriers and the pull queue's store use it), started by
        # the script itself: plain `python tools/sweep.py --gpus 2`, no torchrun
        r = _run([sys.executable, sweep, "--gpus", "2"] + flags, two, _env(IQ_REHEARSAL="1"))
    assert "[sweep] done" in r.stdout
    rec = _record(r.stdout)
    assert rec["n_gpus"] == ranks and rec["unit"] == "coalitions/s" and rec["value"] > 0 and rec["scaling"] == "strong"
    pa, pc = rec["phases"]["A_shapley"], rec["phases"]["C_interaction"]
    assert pa["units"] == len(models) * clouds and sum(x["units"] for x in pa["per_rank"]) == pa["units"]
    assert len(pa["per_rank"]) == ranks and all(0.0 <= x["busy_over_wall"] <= 1.0 for x in pa["per_rank"])
    assert pa["imbalance_max_over_mean_busy"] >= 1.0 and 0.0 <= pa["idle_share"] < 1.0
    # the coalition count is the reference's row count: per cloud, stage 1 = 100 permutations x 33 + the norm factor's 2 clouds;
    # rotate = 216 poses + the original, each x 100 x 33; the scale grid and smoothness (>= 2 x (1 pose + the original)) on top
    per_cloud_min = 100 * 33 + 2 + (217 + 2 + 4) * 3300
    assert pa["coalitions"] >= len(models) * clouds * per_cloud_min and pa["evaluated"] <= pa["coalitions"]
    assert pc["coalitions"] > 0 and rec["coalitions"] == sum(p["coalitions"] for p in rec["phases"].values())
    assert set(pa["by_model"]) == set(models)
    a, b = script_artefacts, _artefacts(two)
    _assert_same(a, b)
    assert any("interaction_seed1" in k and k.endswith("_pred_interaction.npy") for k in a)
    assert any(k.endswith("region_sv_all.npy") for k in a) and any("allregion_inc" in k for k in a)


def test_sweep_covers_all_six_models_and_both_datasets_on_two_ranks(background):
    """BASELINE configs[4] plumbing: every (model, dataset) pair through the sweep driver, two ranks (rehearsal: both on cuda:0),
    one cloud each, stage 1 + the scale sweep (the 216-pose sweeps and the interaction stages of every family are covered by the
    tests above and in test_dist_gpu.py; here the point is that all 12 pairs are assigned, run and written exactly once)."""
    import glob
    # (--resume on a fresh directory skips nothing; it makes rank 0 broadcast its view of the finished units to the other rank;
    # the run itself was started by the module's `background` fixture, beside the per-stage scripts)
    r, tmp_path = background["all_six"]
    rec = _record(r.stdout)
    pa = rec["phases"]["A_shapley"]
    assert rec["n_gpus"] == 2 and pa["units"] == 12 and sorted(x["rank"] for x in pa["per_rank"]) == [0, 1]
    assert all(0 < x["units"] < 12 for x in pa["per_rank"]) and sum(x["units"] for x in pa["per_rank"]) == 12
    assert set(pa["by_model"]) == {"pointnet", "pointnet2", "pointconv", "dgcnn", "gcnn", "gcnn_adv"}
    for model in ("pointnet", "pointnet2", "pointconv", "dgcnn", "gcnn", "gcnn_adv"):
        for dataset in ("modelnet10", "shapenet"):
            root = tmp_path / "checkpoints" / ("exp_MODEL_%s_DATA_%s_POINTNUM_1024_REGIONNUM_32_shapley_test" % (model, dataset)) / "synthetic_00"
            assert (root / "region_sv_all.npy").exists() and (root / "scale_all" / "region_shapley_value.npy").exists(), (model, dataset)
    assert len(glob.glob(str(tmp_path / "fps_*_index_final30.npy"))) == 2


def test_sweep_phase_barriers_and_teardown_on_rccl_with_a_forced_single_rank_group(tmp_path):
    """The sweep's only collectives - the phase barriers and the orderly teardown (dist.shutdown) - on the real library:
    IQ_FORCE_DIST=1 creates the "nccl" (= RCCL) process group at world size 1."""
    sweep = os.path.join(REPO, "tools", "sweep.py")
    flags = ["--models", "pointnet", "--datasets", "modelnet10", "--synthetic", "--num_clouds", "2", "--stages", "shapley_value,scale",
             "--num_samples_save", "100"]
    r = _run(_torchrun(1, 29737) + [sweep] + flags, tmp_path, _env(IQ_FORCE_DIST="1"))
    assert "[sweep] done" in r.stdout
    root = tmp_path / "checkpoints" / "exp_MODEL_pointnet_DATA_modelnet10_POINTNUM_1024_REGIONNUM_32_shapley_test"
    assert (root / "synthetic_01" / "scale_all" / "region_shapley_value.npy").exists()
    # --resume: the units a run with the same settings completed are skipped (markers under checkpoints/.sweep/); a unit whose
    # marker is gone is run again and writes the same files; other settings do not match the markers
    before = _artefacts(tmp_path)
    stamp = os.path.getmtime(root / "synthetic_00" / "scale_all" / "region_shapley_value.npy")
    os.remove(tmp_path / "checkpoints" / ".sweep" / "A_shapley" / "pointnet-modelnet10-1.json")
    r = _run(_torchrun(1, 29737) + [sweep] + flags + ["--resume"], tmp_path, _env(IQ_FORCE_DIST="1"))
    assert "phase A_shapley: 1 units over 1 rank(s) (1 done before, skipped), rank 0 ran 1" in r.stdout, r.stdout
    assert "phase 0_fps: 0 units over 1 rank(s) (1 done before, skipped)" in r.stdout
    assert os.path.getmtime(root / "synthetic_00" / "scale_all" / "region_shapley_value.npy") == stamp
    _assert_same(before, _artefacts(tmp_path))
    r = _run([sys.executable, sweep] + flags[:-1] + ["50", "--resume"], tmp_path, _env())
    assert "phase A_shapley: 2 units over 1 rank(s), rank 0 ran 2" in r.stdout, r.stdout
    # a marker whose unit's files are gone (checkpoints/ cleaned, checkpoints/.sweep/ kept) does not skip the unit
    os.remove(root / "synthetic_01" / "region_sv_all.npy")
    r = _run([sys.executable, sweep] + flags[:-1] + ["50", "--resume"], tmp_path, _env())
    assert "phase A_shapley: 1 units over 1 rank(s) (1 done before, skipped), rank 0 ran 1" in r.stdout, r.stdout
    assert (root / "synthetic_01" / "region_sv_all.npy").exists()
