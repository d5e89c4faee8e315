"""GPU: the host-side mirror of the reference interface (same function names / argument meaning)
and the drop-in stage scripts, end to end, against golden vectors and the CPU oracle."""
import argparse
import os

import numpy as np
import pytest
import torch

from conftest import assert_close_elementwise, load_golden
from interpret_quality_amd import final_common, hip_ops, interaction, pose_sweep, shapley_stage, synth
from interpret_quality_amd.pointnet import PointNetCls

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def dev():
    return torch.device("cuda:0")


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.fixture(scope="module")
def oracle():
    from oracle import ref_cpu
    return ref_cpu


@pytest.fixture(scope="module")
def model(pointnet_sd):
    m = PointNetCls(None)
    m.load_state_dict(pointnet_sd)
    return m.to(dev()).eval()


def ns(**kw):
    base = dict(model="pointnet", softmax_type="modified", num_points=1024, verbose=False)
    base.update(kw)
    return argparse.Namespace(**base)


@pytest.mark.parametrize("num_regions", [8, 32])
def test_shap_sampling_all_regions_batch_matches_reference(model, num_regions):
    g = load_golden("pointnet_shapley_R%d.npz" % num_regions)
    args = ns(num_regions=num_regions, num_samples=int(g["num_samples"]), shapley_batch_size=int(g["bs"]))
    for ci in g["cloud_ids"]:
        p = "c%d_" % ci
        pts, label = synth.make_cloud(int(ci))
        data = torch.from_numpy(pts).unsqueeze(0).to(dev())
        lbl = torch.tensor([label], device=dev())
        phi, logits = final_common.shap_sampling_all_regions_batch(model, data, lbl, g[p + "region_id"], g[p + "orders"], args)
        assert isinstance(phi, np.ndarray) and phi.dtype == np.float64 and phi.shape == (num_regions,)
        assert rel_err(logits.cpu().numpy(), g[p + "logits"]) < RTOL
        assert_close_elementwise(logits.cpu().numpy(), g[p + "logits"])   # and element-wise, with an absolute floor (conftest.py)
        assert np.abs(phi - g[p + "phi"]).max() < RTOL * np.abs(g[p + "phi"]).max()
    bad = ns(num_regions=num_regions, num_samples=8, shapley_batch_size=3)
    with pytest.raises(Exception):
        final_common.shap_sampling_all_regions_batch(model, data, lbl, g[p + "region_id"], g[p + "orders"], bad)


def test_mask_data_batch_inplace_and_cal_reward(model, oracle):
    g = load_golden("pointnet_shapley_R32.npz")
    pts, label = synth.make_cloud(0)
    data = torch.from_numpy(pts).unsqueeze(0)
    center = torch.mean(data, dim=1).squeeze()
    orders = g["c0_orders"][:4]
    want = oracle.shapley_masked_batch(data, center, orders, g["c0_region_id"])
    masked = data.to(dev()).expand(33 * 4, 1024, 3).clone()
    out = final_common.mask_data_batch(masked, center.to(dev()), orders, g["c0_region_id"], ns(num_regions=32))
    assert out.data_ptr() == masked.data_ptr() and torch.equal(masked.cpu(), want)
    m1 = data.to(dev()).expand(33, 1024, 3).clone()
    shapley_stage.mask_data(m1, center.to(dev()), orders[0], g["c0_region_id"])
    assert torch.equal(m1.cpu(), want[:33])
    v, logits = final_common.cal_reward(model, masked, torch.tensor([label], device=dev()), ns())
    assert rel_err(v.cpu().numpy(), g["c0_v_batch0"]) < RTOL
    assert_close_elementwise(v.cpu().numpy(), g["c0_v_batch0"])   # and element-wise, with an absolute floor (conftest.py)


def test_shapley_over_poses_matches_oracle(model, oracle, pointnet_sd):
    num_regions, s = 8, 4
    pts, label = synth.make_cloud(1)
    data = torch.from_numpy(pts).unsqueeze(0)
    lbl = torch.tensor([label])
    region_id = oracle.cal_region_id(data, oracle.farthest_point_sample(data, num_regions)[0])
    orders = synth.make_orders(s, num_regions, seed=7)
    poses = [oracle.rotate_xyz(data, torch.tensor([0.3, -0.2, 0.6])), oracle.translate_pc(data, torch.tensor([0.2, 0.1, -0.3])),
             oracle.scale_pc(data, torch.tensor(1.5))]
    args = ns(num_regions=num_regions, num_samples=s, shapley_batch_size=2)
    phi, logits = pose_sweep.shapley_over_poses(model, torch.cat(poses).to(dev()), lbl.to(dev()), region_id, orders, args,
                                                pose_batch=2)
    om = oracle.PointNetOracle(pointnet_sd)
    for k, pose in enumerate(poses):
        o_phi, o_logits = oracle.shap_sampling_all_regions_batch(om, pose, lbl, region_id, orders, s, 2, num_regions)
        assert rel_err(logits[k].cpu().numpy(), o_logits.numpy()) < RTOL
        assert_close_elementwise(logits[k].cpu().numpy(), o_logits.numpy())   # and element-wise, with an absolute floor (conftest.py)
        assert np.abs(phi[k].cpu().numpy() - o_phi).max() < RTOL * np.abs(o_phi).max()
    # device-side perturbations agree with the oracle's
    assert rel_err(pose_sweep.rotate_xyz(data.to(dev()), torch.tensor([0.3, -0.2, 0.6], device=dev())).cpu().numpy(),
                   poses[0].numpy()) < 1e-6


def test_interaction_functions_match_reference(model):
    g = load_golden("pointnet_interaction_R32.npz")
    pts, label = synth.make_cloud(int(g["cloud_id"]))
    data = torch.from_numpy(pts).unsqueeze(0).to(dev())
    lbl = torch.tensor([label], device=dev())
    args = ns(num_regions=32, interaction_batch_size=int(g["bs"]))
    for tag in ("ratio0", "ratio4", "ratio50", "ratio100"):
        logits = interaction.compute_order_interaction_logits(model, data, g["region_id"], g["pairs"], g[tag + "_contexts"], args)
        assert logits.shape == g[tag + "_logits"].shape
        assert rel_err(logits.cpu().numpy(), g[tag + "_logits"]) < RTOL
        assert_close_elementwise(logits.cpu().numpy(), g[tag + "_logits"])   # and element-wise, with an absolute floor (conftest.py)
        inter = interaction.compute_order_interaction(torch.from_numpy(g[tag + "_logits"]).to(dev()), lbl, args)
        assert inter.dtype == np.float64
        np.testing.assert_allclose(inter, g[tag + "_interaction"], rtol=0, atol=2e-6)
        # |dI| <= 1e-4 max|v| when the logits come from the HIP path (differences of 4 near-equal rewards)
        mine = interaction.compute_order_interaction(logits, lbl, args)
        vmax = np.abs(hip_ops.reward(torch.from_numpy(g[tag + "_logits"]).reshape(-1, 10).to(dev()), label).cpu().numpy()).max()
        assert np.abs(mine - g[tag + "_interaction"]).max() < RTOL * vmax
    empty = interaction.compute_order_interaction_logits(model, data, g["region_id"], np.zeros((0, 2), dtype=np.int64),
                                                         np.zeros((0, 6, 15), dtype=np.int64), args)
    assert tuple(empty.shape) == (0, 24, 10)


def test_stage_scripts_end_to_end(tmp_path, monkeypatch, oracle, pointnet_sd):
    """exp_shapley.sh stage 1 + the scale sweep + exp_interaction.sh stages 2-3 on synthetic data:
    artefact names/shapes of SURVEY.md §8b, efficiency axiom, values against the oracle."""
    monkeypatch.chdir(tmp_path)
    common = ["--model", "pointnet", "--dataset", "modelnet10", "--synthetic", "--num_clouds", "1"]
    shapley_stage.main(common + ["--num_samples_save", "100", "--num_regions", "8"])
    root = "checkpoints/exp_MODEL_pointnet_DATA_modelnet10_POINTNUM_1024_REGIONNUM_8_shapley_test/synthetic_00/"
    region_id = np.load(root + "region_id.npy")
    orders = np.load(root + "all_orders.npy")
    nf = float(np.load(root + "norm_factor.npy"))
    sv_all = np.load(root + "region_sv_all.npy")
    assert region_id.shape == (1024,) and region_id.dtype == np.int64 and orders.shape == (100, 8) and sv_all.shape == (100, 8)
    assert np.array_equal(orders, synth.make_orders(100, 8, seed=1))          # the reference's RNG stream
    np.testing.assert_allclose(sv_all.sum(1), nf, rtol=0, atol=2e-4)          # efficiency, every permutation
    phi100 = np.load(root + "region_shapley/0_100.npy")
    pts, label = synth.make_cloud(0)
    data, lbl = torch.from_numpy(pts).unsqueeze(0), torch.tensor([label])
    o_total, _ = oracle.shap_sampling_stage1(oracle.PointNetOracle(pointnet_sd), data, lbl, region_id, orders[:20], 8)
    assert np.abs(sv_all[:20].sum(0) - o_total).max() < RTOL * np.abs(o_total).max()
    np.testing.assert_allclose(phi100, sv_all.sum(0) / 100, rtol=1e-12)
    per_point = np.load(root + "shapley/0_100.npy")
    assert per_point.shape == (1024,) and np.allclose(per_point, phi100[region_id])
    assert os.path.exists("fps_modelnet10_1024_8_index_final30.npy")


def test_pose_sweep_and_interaction_scripts(tmp_path, monkeypatch, oracle):
    monkeypatch.chdir(tmp_path)
    common = ["--model", "pointnet", "--dataset", "modelnet10", "--synthetic", "--num_clouds", "1"]
    shapley_stage.main(common + ["--num_samples_save", "100"])
    root = "checkpoints/exp_MODEL_pointnet_DATA_modelnet10_POINTNUM_1024_REGIONNUM_32_shapley_test/synthetic_00/"
    pose_sweep.main_scale(common)
    phi = np.load(root + "scale_all/region_shapley_value.npy")
    logits = torch.load(root + "scale_all/all_logits.pt")
    assert phi.shape == (30, 32) and tuple(logits.shape) == (30, 3300, 10)
    assert np.load(root + "scale_all/scale.npy").shape == (30,)
    orig = np.load(root + "scale_all/orig_shapley_value.npy")
    k = int(np.argmin(np.abs(np.load(root + "scale_all/scale.npy") - 1.0)))
    assert orig.shape == (32,) and os.path.exists(root + "scale_all/log.txt")
    # efficiency per pose: sum phi = mean over permutations of v(N) - v(empty)
    v = hip_ops.reward(logits.reshape(-1, 10).cuda().contiguous(), 0, True).reshape(30, 100, 33)
    np.testing.assert_allclose(phi.sum(1), (v[:, :, 32] - v[:, :, 0]).double().mean(1).cpu().numpy(), atol=2e-4)
    assert k >= 0
    # interaction: hand-made stage-1 artefacts (final_gen_pair.py is a "next" row)
    inter = root + "interaction_seed1/"
    os.makedirs(inter + "normal")
    os.makedirs(inter + "rotate_adv")
    rng = np.random.default_rng(0)
    pairs = np.array([[1, 5], [7, 30], [0, 31]])
    np.save(inter + "region_pair_list.npy", pairs)
    ratios = interaction.DEFAULT_RATIOS
    for ratio in ratios:
        m = int(30 * ratio)
        c = 1 if m in (0, 30) else 3
        ctx = np.zeros((3, c, m), dtype=np.int64)
        for p in range(3):
            rest = [r for r in range(32) if r not in pairs[p]]
            for j in range(c):
                ctx[p, j] = rng.choice(rest, m, replace=False)
        np.save(inter + "ratio%d_context_list.npy" % int(ratio * 100), ctx)
    np.save(inter + "rotate_adv/transform_params.npy", np.array([0.4, -0.3, 0.2]))
    np.save(inter + "rotate_adv/pred_labels.npy", np.array([0, 3]))
    interaction.main_logits(common)
    interaction.main_cal(common)
    for sub, lab in (("normal/", 0), ("rotate_adv/", 3)):
        for ratio in (0.0, 0.5, 1.0):
            tag = "ratio%d" % int(ratio * 100)
            lg = torch.load(inter + sub + tag + "_all_logits.pt")
            got = np.load(inter + sub + tag + "_pred_interaction.npy")
            want = oracle.compute_order_interaction(lg.cpu(), torch.tensor([lab]))
            assert got.shape == want.shape == (3, lg.shape[1] // 4)
            np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)


def test_exp_interaction_full_pipeline_incl_gen_pair(tmp_path, monkeypatch):
    """exp_shapley.sh (stage 1 + rotate sweep) then the complete exp_interaction.sh, synthetic cloud."""
    from interpret_quality_amd import gen_pair
    monkeypatch.chdir(tmp_path)
    common = ["--model", "pointnet", "--dataset", "modelnet10", "--synthetic", "--num_clouds", "1"]
    shapley_stage.main(common + ["--num_samples_save", "100"])
    pose_sweep.main_rotate(common)
    root = "checkpoints/exp_MODEL_pointnet_DATA_modelnet10_POINTNUM_1024_REGIONNUM_32_shapley_test/synthetic_00/"
    assert np.load(root + "rotate_all/region_shapley_value.npy").shape == (216, 32)
    gen_pair.main(common + ["--num_pairs_random", "4", "--num_save_context_max", "3"])
    inter = root + "interaction_seed1/"
    assert np.load(inter + "region_pair_list.npy").shape == (4, 2)
    assert np.load(inter + "ratio50_context_list.npy").shape == (4, 3, 15)
    assert np.load(inter + "ratio0_context_list.npy").shape == (4, 1, 0)
    lab = np.load(inter + "rotate_adv/pred_labels.npy")
    assert lab.shape == (2,) and lab[0] == 0
    pose_idx = int(np.load(inter + "rotate_adv/pose_idx.npy"))
    np.testing.assert_array_equal(np.load(inter + "rotate_adv/transform_params.npy"),
                                  np.load(root + "rotate_all/angle_tuple.npy")[pose_idx])
    singles = sorted(os.listdir(inter + "rotate_adv_single_region"))
    assert len(singles) == 32 and singles[0].startswith("range_rank01_region")
    interaction.main_logits(common)
    interaction.main_cal(common)
    for sub in ("normal/", "rotate_adv/", "rotate_adv_single_region/" + singles[0] + "/normal/"):
        lg = torch.load(inter + sub + "ratio50_all_logits.pt")
        it = np.load(inter + sub + "ratio50_pred_interaction.npy")
        if lg.shape[0] == 0:   # a region without ball-query neighbours ("NO NEIGHBORS!!!"): empty, not a crash
            assert it.shape[0] == 0
        else:
            assert lg.shape[1] == 12 and it.shape == (lg.shape[0], 3)


@pytest.mark.parametrize("model_name", ["pointnet2", "dgcnn", "pointconv"])
def test_stage1_and_a_sweep_for_models_without_the_fused_pointnet_path(tmp_path, monkeypatch, model_name):
    """exp_shapley.sh stage 1 + the scale sweep for the other model families (materialised clouds / compact clouds):
    artefacts and the efficiency axiom."""
    monkeypatch.chdir(tmp_path)
    common = ["--model", model_name, "--dataset", "modelnet10", "--synthetic", "--num_clouds", "1"]
    shapley_stage.main(common + ["--num_samples_save", "100", "--num_regions", "8"])
    root = "checkpoints/exp_MODEL_%s_DATA_modelnet10_POINTNUM_1024_REGIONNUM_8_shapley_test/synthetic_00/" % model_name
    nf = float(np.load(root + "norm_factor.npy"))
    sv_all = np.load(root + "region_sv_all.npy")
    assert sv_all.shape == (100, 8)
    np.testing.assert_allclose(sv_all.sum(1), nf, rtol=0, atol=1e-3 * max(1.0, abs(nf)))


def test_full_size_shapley_properties(model, oracle, pointnet_sd):
    """BASELINE configs[1] at full size (32 regions x 1000 permutations = 33 000 coalitions of one cloud), through
    size-independent properties: efficiency of every permutation, exact float64 additivity over blocks of permutations,
    permutation-order equivariance of the logits, and a spot check of 3 random permutations against the CPU oracle."""
    pts, label = synth.make_cloud(3)
    data = torch.from_numpy(pts).unsqueeze(0).to(dev())
    lbl = torch.tensor([label], device=dev())
    region_id = hip_ops.region_assign(data[0].contiguous(), hip_ops.fps(data, 32)[0].contiguous()).cpu().numpy().astype(np.int64)
    orders = synth.make_orders(1000, 32, seed=1)
    args = ns(num_regions=32, num_samples=1000, shapley_batch_size=50)
    phi, logits = final_common.shap_sampling_all_regions_batch(model, data, lbl, region_id, orders, args)
    assert tuple(logits.shape) == (33000, 10)
    v = hip_ops.reward(logits.contiguous(), label, True).reshape(1000, 33).double().cpu().numpy()
    # efficiency: the telescoping sum of a permutation is v(all) - v(none); rows 0 / 32 are the same two clouds everywhere
    assert np.abs(v[:, 0] - v[0, 0]).max() == 0 and np.abs(v[:, 32] - v[0, 32]).max() == 0
    np.testing.assert_allclose(phi.sum(), v[0, 32] - v[0, 0], rtol=0, atol=5e-5 * max(1.0, abs(v[0, 32] - v[0, 0])))
    # additivity: phi over 1000 permutations = mean of the 10 block results (float64 sums of float32 differences)
    blocks = []
    for k in range(10):
        a = ns(num_regions=32, num_samples=100, shapley_batch_size=50)
        pk, lk = final_common.shap_sampling_all_regions_batch(model, data, lbl, region_id, orders[100 * k:100 * (k + 1)], a)
        assert torch.equal(lk, logits[3300 * k:3300 * (k + 1)])         # a coalition's logits do not depend on the batch
        blocks.append(pk)
    np.testing.assert_allclose(np.mean(blocks, axis=0), phi, rtol=0, atol=1e-12)
    # spot check against the oracle
    rng = np.random.default_rng(0)
    pick = rng.choice(1000, size=3, replace=False)
    want, _ = oracle.shap_sampling_all_regions_batch(oracle.PointNetOracle(pointnet_sd), data.cpu(), lbl.cpu(), region_id,
                                                     orders[pick], 3, 3, 32)
    got = np.zeros(32)
    for o in pick:
        dv = v[o, 1:] - v[o, :-1]
        got[orders[o]] += dv
    assert np.abs(got / 3 - want).max() < RTOL * max(np.abs(want).max(), 1e-3) + 1e-6


def test_two_rank_pipelines_write_the_same_artefacts_as_one_process(tmp_path, monkeypatch):
    """The N > 1 driver path end to end on the HIP kernels: stage 1, the rotation sweep and the whole interaction pipeline of
    one cloud with 2 ranks (torch.distributed.run, IQ_REHEARSAL=1: both ranks on cuda:0, gloo instead of RCCL) against a
    single process.  Shards: 100 permutations -> 50 + 50, 216 poses -> 108 + 108, 4 pairs -> 2 + 2; rank 0 writes.
    (An earlier version of this test caught two races: every rank creating `checkpoints/`, and every rank writing the
    FPS index file while another was already reading it.)"""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--model", "pointnet", "--dataset", "modelnet10", "--synthetic", "--num_clouds", "1"]
    env = dict(os.environ, PYTHONPATH=repo, IQ_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    root = "checkpoints/exp_MODEL_pointnet_DATA_modelnet10_POINTNUM_1024_REGIONNUM_32_shapley_test/synthetic_00/"
    stages = (("final_shapley_value.py", ["--num_samples_save", "100"]), ("final_rotate_center_enum_all.py", []),
              ("final_gen_pair.py", ["--num_pairs_random", "4", "--num_save_context_max", "3"]),
              ("final_point_binary_interaction_logits.py", []), ("final_cal_interactions.py", []))
    results = {}
    launchers = (("one", [sys.executable]),
                 ("two", [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29573"]))

    def pipeline(tag, launcher):
        work = tmp_path / tag
        work.mkdir()
        for k, (script, extra) in enumerate(stages):
            cmd = list(launcher)
            if tag == "two":
                cmd[cmd.index("29573")] = str(29573 + k)       # a fresh rendezvous port per launch
            r = subprocess.run(cmd + [os.path.join(repo, script)] + common + extra, cwd=str(work), env=env,
                               capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, (script, r.stderr[-3000:])

    import concurrent.futures as cf
    with cf.ThreadPoolExecutor(max_workers=2) as ex:           # the one-process and the two-rank pipeline side by side
        for f in [ex.submit(pipeline, t, l) for t, l in launchers]:
            f.result()
    for tag, _ in launchers:
        work = tmp_path / tag
        inter = work / root / "interaction_seed1"
        results[tag] = {
            "sv": np.load(str(work / root / "region_sv_all.npy")),
            "phi": np.load(str(work / root / "rotate_all" / "region_shapley_value.npy")),
            "logits": torch.load(str(work / root / "rotate_all" / "all_logits.pt"), map_location="cpu"),
            "pairs": np.load(str(inter / "region_pair_list.npy")),
            "inter_logits": torch.load(str(inter / "normal" / "ratio50_all_logits.pt"), map_location="cpu"),
            "inter": np.load(str(inter / "rotate_adv" / "ratio50_pred_interaction.npy")),
        }
    one, two = results["one"], results["two"]
    assert np.array_equal(one["sv"], two["sv"])          # float64 accumulation in permutation order
    assert np.array_equal(one["phi"], two["phi"]) and torch.equal(one["logits"], two["logits"])
    assert np.array_equal(one["pairs"], two["pairs"]) and torch.equal(one["inter_logits"], two["inter_logits"])
    assert np.array_equal(one["inter"], two["inter"]) and one["inter"].shape[0] == 4


def test_checkpoint_file_in_the_references_layout_is_loaded(tmp_path, monkeypatch):
    """tools/final_util.py:236-262: a `.t7` saved from an nn.DataParallel model (keys prefixed with `module.`) at the
    reference's path is what the drivers load; --synthetic only stands in when the file is absent."""
    from interpret_quality_amd import final_util
    monkeypatch.chdir(tmp_path)
    args = argparse.Namespace(model="pointnet", dataset="modelnet10", synthetic=True, device=dev())
    final_util.set_model_args(args)
    os.makedirs(os.path.dirname(args.model_path))
    sd = synth.to_torch(synth.pointnet_state_dict(1))                      # NOT the seed-0 stand-in
    torch.save({"module." + k: v for k, v in sd.items()}, args.model_path)
    from_file = final_util.load_model(args)
    x = torch.from_numpy(synth.make_cloud(0)[0]).unsqueeze(0).permute(0, 2, 1).contiguous().to(dev())
    want = PointNetCls(None)
    want.load_state_dict(sd)
    assert torch.equal(from_file(x)[0], want.to(dev()).eval()(x)[0])
    os.remove(args.model_path)
    stand_in = final_util.load_model(args)                                  # file gone: the seed-0 stand-in
    assert not torch.equal(stand_in(x)[0], from_file(x)[0])
    args.synthetic = False
    with pytest.raises(FileNotFoundError):
        final_util.load_model(args)


def test_stage1_on_a_real_format_dataset_tree_without_the_synthetic_flag(tmp_path, monkeypatch):
    """exp_shapley.sh stage 1 exactly as a user of the reference would run it: ShapeNet-format scans under data/,
    misc/ lists, a .t7 checkpoint - no --synthetic.  Folder names, FPS index file and artefacts per cloud."""
    from interpret_quality_amd import final_util
    synth.write_dataset_tree(str(tmp_path))
    monkeypatch.chdir(tmp_path)
    args = argparse.Namespace(model="pointnet", dataset="shapenet")
    final_util.set_model_args(args)
    os.makedirs(os.path.dirname(args.model_path))
    torch.save({"module." + k: v for k, v in synth.to_torch(synth.pointnet_state_dict(0)).items()}, args.model_path)
    shapley_stage.main(["--model", "pointnet", "--dataset", "shapenet", "--num_samples_save", "20", "--num_regions", "8"])
    fps = np.load("fps_shapenet_1024_8_index_final30.npy")
    assert fps.shape == (3, 8) and fps.dtype == np.int64 and (fps[:, 0] == 0).all()
    root = "checkpoints/exp_MODEL_pointnet_DATA_shapenet_POINTNUM_1024_REGIONNUM_8_shapley_test/"
    for name in ("Bag_aaaa0001", "Mug_bbbb0002", "Rocket_cccc0003"):          # classname_uuid, tools/final_util.py:275-276
        sv = np.load(root + name + "/region_sv_all.npy")
        nf = float(np.load(root + name + "/norm_factor.npy"))
        assert sv.shape == (20, 8) and np.abs(sv.sum(1) - nf).max() < 2e-4 * max(1.0, abs(nf))
        assert np.load(root + name + "/region_id.npy").shape == (1024,)


@pytest.mark.parametrize("model_name", ["pointnet", "dgcnn", "pointconv"])
def test_strict_batch_cap_bounds_every_launch_and_changes_no_result(model_name, monkeypatch):
    """config.py's knobs as a CAP (CONFIG['strict_batch_cap'] / IQ_STRICT_BATCH=1, the reference's meaning of config.py:2-17)
    against the default floor semantics: no launch exceeds knob x (R+1) (x 4) coalitions, logits bitwise identical."""
    from interpret_quality_amd import final_util
    a = ns(model=model_name, dataset="modelnet10", device=dev(), synthetic=True, model_path="checkpoints/none.t7", k=20,
           num_regions=8, num_samples=8, shapley_batch_size=2, interaction_batch_size=3)
    m = final_util.load_model(a)
    pts, label = synth.make_cloud(4)
    data = torch.from_numpy(pts).unsqueeze(0).to(dev())
    lbl = torch.tensor([label], device=dev())
    rid = hip_ops.region_assign(data[0].contiguous(), hip_ops.fps(data, 8)[0].contiguous()).cpu().numpy().astype(np.int64)
    orders = synth.make_orders(8, 8, seed=3)
    pairs = np.array([[0, 5], [2, 3]])
    ctx = np.array([[[1, 4], [6, 7], [1, 7], [4, 6], [2, 6]], [[0, 1], [5, 6], [4, 7], [0, 7], [1, 5]]])
    sizes = []
    target = m
    for name in ("coalition_logits", "forward_points"):
        if hasattr(target, name):
            orig = getattr(target, name)

            def spy(*args, _orig=orig, _name=name, **kw):
                sizes.append(args[3].shape[0] if _name == "coalition_logits" else args[0].shape[0])
                return _orig(*args, **kw)
            monkeypatch.setattr(target, name, spy)
    out = {}
    for strict in (False, True):
        a.strict_batch_cap = strict
        sizes.clear()
        phi, logits = final_common.shap_sampling_all_regions_batch(m, data, lbl, rid, orders, a)
        il = interaction.compute_order_interaction_logits(m, data, rid, pairs, ctx, a)
        out[strict] = (phi, logits, il)
        if strict:
            assert sizes and max(sizes) <= max(2 * 9, 4 * 3), sizes
    assert np.array_equal(out[False][0], out[True][0])
    assert torch.equal(out[False][1], out[True][1]) and torch.equal(out[False][2], out[True][2])
