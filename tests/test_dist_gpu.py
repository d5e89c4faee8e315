"""GPU: the multi-rank path of the drivers and of bench.py.

* RCCL itself (backend "nccl") on the one GPU this box has: IQ_FORCE_DIST=1 creates a single-rank process group, so the
  communicator, the barrier and the all_gather_into_tensor of the N > 1 path run on the real collective library.
* Two ranks (IQ_REHEARSAL=1: both on cuda:0, gloo in place of RCCL, which needs one device per rank) against a single
  process for every model family: all artefact files bitwise identical.
* Regression tests for the two file races fixed in round 1 (every rank creating `checkpoints/`; the FPS index file
  written while another rank reads it), with the directories / files present and absent at start.
"""
import glob
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = dict(os.environ, PYTHONPATH=REPO, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "IQ_FORCE_DIST", "IQ_REHEARSAL", "IQ_BENCH_REHEARSAL"):
        env.pop(k, None)
    env.update(kw)
    return env


def _torchrun(nproc, port):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
            "--master-port", str(port)]


def _run(cmd, cwd, env, timeout=900):
    # the child processes share the ONE GPU of the box with this pytest process: hand back what its caching allocator holds
    # (freed workspaces of earlier tests' models, tens of GB each) before they start
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        import gc
        gc.collect()
        torch.cuda.empty_cache()
    r = subprocess.run(cmd, cwd=str(cwd), env=env, capture_output=True, text=True, timeout=timeout)
    if r.returncode != 0:
        # keep the WHOLE output of a failed launch where it survives the run (gpurun_out/ is merged back): the assertion text below
        # is shortened by pytest, and a child's own traceback comes before torchrun's wrapper trace
        keep = os.path.join(REPO, "gpurun_out", "test_failures")
        os.makedirs(keep, exist_ok=True)
        tag = "%s_%d" % (os.path.basename([c for c in cmd if c.endswith(".py")][-1])[:-3], os.getpid())
        with open(os.path.join(keep, tag + ".txt"), "w") as f:
            f.write("cmd: %r\ncwd: %s\nrc: %d\n---- stdout ----\n%s\n---- stderr ----\n%s\n" % (cmd, cwd, r.returncode, r.stdout, r.stderr))
    assert r.returncode == 0, (cmd[-6:], r.stderr[:3000], r.stderr[-3000:])
    return r


def _run_chains(chains):
    """Several chains of launches side by side (each chain in order, the chains concurrently): the one-process and the two-rank run
    of a pipeline, or the per-stage scripts of two models, do not depend on each other, and most of a launch is process start-up.
    At most three chains = four GPU processes beside this one (the pool allows six).  chains: [[(cmd, cwd, env), ...], ...]."""
    import concurrent.futures as cf
    assert len(chains) <= 3

    def one(chain):   # an element is a launch (cmd, cwd, env) or a callable (e.g. an Event's set / wait between chains)
        return [step() if callable(step) else _run(*step) for step in chain]
    with cf.ThreadPoolExecutor(max_workers=len(chains)) as ex:
        futs = [ex.submit(one, c) for c in chains]
        return [f.result() for f in futs]


def _artefacts(root):
    out = {}
    for f in sorted(glob.glob(os.path.join(str(root), "checkpoints", "**", "*"), recursive=True)):
        if os.path.isdir(f) or f.endswith(".txt"):
            continue
        rel = os.path.relpath(f, str(root))
        out[rel] = np.load(f) if f.endswith(".npy") else torch.load(f, map_location="cpu").numpy()
    return out


def _assert_same(a, b):
    assert set(a) == set(b) and len(a) > 0, sorted(set(a) ^ set(b))[:10]
    for k in a:
        assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k], equal_nan=True), k


def test_bench_runs_its_collectives_on_rccl_with_a_forced_single_rank_group(tmp_path):
    """bench.py under `torchrun --nproc-per-node 1` with IQ_FORCE_DIST=1: process group "nccl" (= RCCL), barrier on both sides
    of the timed region, all_gather_into_tensor of the logits every step, max-reduce of the elapsed time."""
    bench = os.path.join(REPO, "bench.py")
    flags = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--repeats", "1", "--perms", "100", "--other-models", "0", "--cpu-baseline", "0",
             "--eager-baseline", "0", "--traffic", "0", "--strong-steps", "0", "--sustained-s", "0.3"]
    # under torchrun, and (second chain, side by side) without it - RANK / WORLD_SIZE absent: defaults, a free rendezvous port
    (r,), (r2,) = _run_chains([[(_torchrun(1, 29611) + [bench] + flags, tmp_path, _env(IQ_FORCE_DIST="1"))],
                               [([sys.executable, bench] + flags, tmp_path, _env(IQ_FORCE_DIST="1"))]])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and "forced single-rank RCCL group" in d["config"]["parallelism"]
    assert 0 < d["roofline"]["frac"] <= 1.0 and d["roofline"]["traffic"] is None
    # the power-capped ceiling is measured on this board by the library's register-only MFMA loop, not quoted
    sus = d["roofline"]["sustained_bf16_mfma"]
    assert 0.3 < sus["frac_of_dense_peak"] <= 1.0 and 0.8 < sus["shader_clock_ghz"] < 2.6
    assert abs(d["roofline"]["frac_of_sustained_bf16_ceiling"] - d["roofline"]["frac"] / sus["frac_of_dense_peak"]) < 1e-9
    assert len([ln for ln in r2.stdout.splitlines() if ln.startswith("{")]) == 1


@pytest.fixture(scope="module")
def bench_runs(tmp_path_factory):
    """Five bench.py launches the tests below look at, started together as two chains (most of a launch is process start-up; at most
    four GPU processes beside this one): [self-launched 2-rank weak run, 2-rank sweep mode] and [strong scaling on 1 rank, on 2, the
    driver's torchrun form at N = 2]."""
    bench = os.path.join(REPO, "bench.py")
    root = tmp_path_factory.mktemp("bench_runs")
    dirs = {k: root / k for k in ("self", "sweep", "strong1", "strong2", "n2")}
    for v in dirs.values():
        v.mkdir()
    weak = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--repeats", "1", "--perms", "100", "--strong-steps", "0", "--profile-steps", "1"]
    sweep = ["--gpus", "2", "--scaling", "sweep", "--sweep-models", "pointnet,gcnn", "--sweep-datasets", "modelnet10", "--sweep-clouds", "1",
             "--sweep-reduced", "1"]
    strong = lambda n: ["--gpus", str(n), "--scaling", "strong", "--steps", "1", "--warmup", "0", "--repeats", "1"]
    # the driver's own launch form for N = 2 (torchrun, every default leg of the line on)
    n2 = ["--gpus", "2", "--steps", "1", "--warmup", "1", "--repeats", "1", "--perms", "100", "--sustained-s", "0.3"]
    (r_self, r_sweep), (r_s1, r_s2, r_n2) = _run_chains([
        [([sys.executable, bench] + weak, dirs["self"], _env(IQ_REHEARSAL="1")),
         ([sys.executable, bench] + sweep, dirs["sweep"], _env(IQ_REHEARSAL="1"))],
        [([sys.executable, bench] + strong(1), dirs["strong1"], _env()),
         (_torchrun(2, 29615) + [bench] + strong(2), dirs["strong2"], _env(IQ_REHEARSAL="1")),
         (_torchrun(2, 29591) + [bench] + n2, dirs["n2"], _env(IQ_BENCH_REHEARSAL="1"))]])
    return {"self": r_self, "sweep": r_sweep, 1: r_s1, 2: r_s2, "n2": r_n2, "dirs": dirs}


def test_bench_starts_its_own_ranks_without_a_launcher(bench_runs, tmp_path):
    """Plain `python bench.py --gpus 2 --steps 2` - no torchrun: the script becomes the parent of two fresh ranks before it
    touches the GPU (interpret_quality_amd/launch.py), picks a free rendezvous port, and rank 0's ONE JSON line comes back
    through it (rehearsal: both ranks on cuda:0, gloo collectives)."""
    bench = os.path.join(REPO, "bench.py")
    r = bench_runs["self"]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    # >= 2 s per timed region whatever --steps is: several blocks of exactly --steps steps, the median block reported
    t = d["timing"]
    assert t["blocks_per_region"] > 1 and t["region_s"] >= 2.0 and len(t["blocks_s"]) >= t["blocks_per_region"]
    assert abs(d["ms_per_step"] * d["steps"] * 1e-3 - sorted(t["blocks_s"])[len(t["blocks_s"]) // 2]) < 1e-9
    assert "clouds sharded over 2 GPU(s)" in d["config"]["parallelism"]
    # a node with fewer GPUs than ranks is refused up front (no rehearsal switch), before anything is started
    r = subprocess.run([sys.executable, bench, "--gpus", "64"], cwd=str(tmp_path), env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "this node shows" in r.stderr


def test_bench_n2_rehearsal_prints_one_valid_json_line(bench_runs):
    """bench.py's N > 1 path as the driver starts it (torchrun, --gpus 2): barrier + max over ranks + all-gather of the logits with 2
    ranks on the one GPU (IQ_BENCH_REHEARSAL=1: gloo instead of RCCL; the number itself means nothing)."""
    r = bench_runs["n2"]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["unit"] == "coalitions/s" and d["value"] > 0 and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])
    # the same line carries the path's own shard axes at this rank count (one cloud's pose sweep + interaction setting sharded)
    st = d["strong_scaling"]
    assert st["value"] > 0 and st["steps"] == 1 and "poses sharded" in st["config"]["workload"] and 0.0 <= st["gather"]["share_of_step"] < 1.0


def test_bench_sweep_mode_emits_a_configs4_line_on_two_self_launched_ranks(bench_runs):
    """`python bench.py --gpus 2 --scaling sweep`: BASELINE configs[4] in the bench schema (two families x one dataset x one cloud
    at rehearsal sizes here), units pulled from the shared queue."""
    r = bench_runs["sweep"]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["unit"] == "coalitions/s" and d["value"] > 0
    assert d["coalitions"] == sum(p["coalitions"] for p in d["phases"].values()) > 0
    assert set(d["phases"]) == {"0_fps", "A_shapley", "B_gen_pair", "C_interaction"}
    assert len(d["phases"]["A_shapley"]["per_rank"]) == 2 and "configs[4]" in d["config"]["workload"]
    assert not (bench_runs["dirs"]["sweep"] / "checkpoints").exists()      # the sweep ran in a scratch directory


@pytest.mark.parametrize("ranks", [1, 2])
def test_bench_strong_scaling_mode_shards_one_cloud_over_the_ranks(bench_runs, ranks):
    """`bench.py --scaling strong`: the rotation sweep (poses sharded) and one interaction setting (pairs sharded) of ONE cloud
    through the drivers' own sharded code, 1 process and 2 ranks (rehearsal: both on cuda:0, gloo; under torchrun)."""
    r = bench_runs[ranks]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["scaling"] == "strong" and d["n_gpus"] == ranks and d["value"] > 0 and d["unit"] == "coalitions/s"
    assert "716100" in d["config"]["workload"] and "1238400" in d["config"]["workload"]
    assert 0.0 <= d["gather"]["share_of_step"] < 1.0
    if ranks == 1:
        assert d["gather"]["share_of_step"] == 0.0   # a single process has no collective on its path


def test_stage_scripts_give_the_same_artefacts_through_a_forced_rccl_group(tmp_path):
    """Stage 1 and the scale sweep with the gather / barrier path on RCCL (single rank) against the collective-free run."""
    common = ["--model", "pointnet", "--dataset", "modelnet10", "--synthetic", "--num_clouds", "1"]
    chains = []
    for tag, launcher, env in (("plain", [sys.executable], _env()),
                               ("rccl", _torchrun(1, 29621), _env(IQ_FORCE_DIST="1"))):
        work = tmp_path / tag
        work.mkdir()
        chain = []
        for k, (script, extra) in enumerate((("final_shapley_value.py", ["--num_samples_save", "100"]),
                                             ("final_scale_center_enum_all.py", []))):
            cmd = list(launcher)
            if tag == "rccl":
                cmd[cmd.index("29621")] = str(29621 + k)
            chain.append((cmd + [os.path.join(REPO, script)] + common + extra, work, env))
        chains.append(chain)
    _run_chains(chains)
    _assert_same(_artefacts(tmp_path / "plain"), _artefacts(tmp_path / "rccl"))


@pytest.mark.parametrize("model,dataset,stages", [
    ("pointnet2", "shapenet", "shapley"),
    ("dgcnn", "modelnet10", "shapley"),
    ("pointconv", "shapenet", "shapley"),
    ("gcnn_adv", "shapenet", "interaction"),
])
def test_two_ranks_write_the_same_artefacts_as_one_process_for_every_family(tmp_path, model, dataset, stages):
    """tools/two_rank_check.sh as a test: the drivers of every model family (and `gcnn_adv`, `--dataset shapenet`, which no other
    -m gpu test sends through a driver) with 1 process and with 2 ranks; ALL artefact files must be bitwise identical."""
    common = ["--model", model, "--dataset", dataset, "--synthetic", "--num_clouds", "1"]
    if stages == "shapley":      # stage 1 (permutations sharded), scale sweep (poses sharded), smoothness (epochs sharded)
        plan = [("final_shapley_value.py", ["--num_samples_save", "100"]), ("final_scale_center_enum_all.py", [])]
    else:                        # exp_interaction.sh: needs the rotation sweep's artefacts, then pairs sharded
        plan = [("final_shapley_value.py", ["--num_samples_save", "100"]), ("final_rotate_center_enum_all.py", []),
                ("final_gen_pair.py", ["--num_pairs_random", "5", "--num_save_context_max", "3"]),
                ("final_point_binary_interaction_logits.py", []), ("final_cal_interactions.py", [])]
    base = 29640 + 10 * ["pointnet2", "dgcnn", "pointconv", "gcnn_adv"].index(model)
    chains = []
    for tag in ("one", "two"):     # side by side: three processes on the card
        work = tmp_path / tag
        work.mkdir()
        chains.append([(([sys.executable] if tag == "one" else _torchrun(2, base + k)) + [os.path.join(REPO, script)] + common + extra, work,
                        _env(IQ_REHEARSAL="1") if tag == "two" else _env()) for k, (script, extra) in enumerate(plan)])
    _run_chains(chains)
    _assert_same(_artefacts(tmp_path / "one"), _artefacts(tmp_path / "two"))


@pytest.mark.parametrize("preexisting", [False, True])
def test_two_ranks_do_not_race_on_checkpoints_dir_and_fps_index(tmp_path, preexisting):
    """The two races fixed in round 1: (a) every rank calls mkdir('checkpoints') - must not fail whether the directory exists
    or not; (b) the FPS index file is written by rank 0 behind a barrier, never while another rank reads it, and a
    pre-existing file is used as it is."""
    common = ["--model", "pointnet", "--dataset", "modelnet10", "--synthetic", "--num_clouds", "2", "--num_samples_save", "100"]
    fps_file = tmp_path / "fps_modelnet10_1024_32_index_final30.npy"
    if preexisting:
        (tmp_path / "checkpoints").mkdir()
        _run([sys.executable, os.path.join(REPO, "final_shapley_value.py")] + common, tmp_path, _env())
        want = np.load(str(fps_file))
        stamp = os.path.getmtime(str(fps_file))
    _run(_torchrun(2, 29691 + int(preexisting)) + [os.path.join(REPO, "final_shapley_value.py")] + common, tmp_path,
         _env(IQ_REHEARSAL="1"))
    got = np.load(str(fps_file))
    assert got.shape == (2, 32) and got.dtype == np.int64
    if preexisting:
        assert np.array_equal(got, want) and os.path.getmtime(str(fps_file)) == stamp   # reused, not rewritten
    assert not glob.glob(str(tmp_path / "*.tmp.npy"))


def test_a_failing_rank_leaves_its_own_traceback_on_stderr(tmp_path):
    """dist.record on the stage mains: when a rank dies, its exception text reaches stderr (round 1 lost the cause of an
    intermittent 2-rank failure because only torchrun's wrapper trace was kept).  Provoked with a region count the model
    rejects."""
    cmd = _torchrun(2, 29699) + [os.path.join(REPO, "final_shapley_value.py"), "--model", "pointnet", "--dataset", "modelnet10",
                                 "--synthetic", "--num_clouds", "1", "--num_samples_save", "10", "--num_regions", "65"]
    r = subprocess.run(cmd, cwd=str(tmp_path), env=_env(IQ_REHEARSAL="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "stage failed" in r.stderr and "Traceback" in r.stderr
