"""GPU: parity against the reference-generated fixtures added in round 2.

* gen_pair.npz      - the device logic of final_gen_pair.py (check_adv_success, save_pair_single_region, save_pred_label)
* pointnet_config0.npz - BASELINE configs[0] at spec: 30 clouds, 8 regions, 64 permutations, bs 8 (17 280 coalitions)
* dgcnn_scale.npz / gcnn_scale.npz - 2016 interaction coalitions each, the reference in float32 and float64: conditioning
  against exact arithmetic, near-tie attribution, context-averaged interactions
"""
import argparse
import os

import numpy as np
import pytest
import torch

from conftest import assert_close_elementwise, load_golden
from interpret_quality_amd import final_common, gen_pair, hip_ops, interaction, pose_sweep, shapley_stage, synth
from interpret_quality_amd.dgcnn import DGCNN_cls
from interpret_quality_amd.pointnet import PointNetCls

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def pointnet(pointnet_sd):
    m = PointNetCls(None)
    m.load_state_dict(pointnet_sd)
    return m.to(dev()).eval()


def test_gen_pair_device_logic_matches_the_reference(tmp_path, monkeypatch, pointnet):
    """final_gen_pair.py:221-286 (216-pose dense forward -> argmin of the reward -> pose_idx / transform_params), :145-218
    (range ranks, max / min poses, neighbour pairs, folder names) and :90-123 (predicted labels) on the inputs the reference
    was run on (tests/golden/gen_golden_gen_pair.py)."""
    g = load_golden("gen_pair.npz")
    ci, r, name = int(g["cloud_id"]), 32, "synthetic_03"
    exp = str(tmp_path) + "/exp/"
    base = exp + name + "/"
    os.makedirs(base + "rotate_all")
    os.makedirs(base + "interaction_seed1/rotate_adv")
    os.makedirs(base + "interaction_seed1/normal")
    np.save(base + "region_id.npy", g["region_id"].astype(np.int64))
    np.save(base + "rotate_all/angle_tuple.npy", g["angle_tuple"])
    np.save(base + "rotate_all/region_shapley_value.npy", g["region_shapley_value"])
    args = argparse.Namespace(model="pointnet", dataset="modelnet10", mode="rotate", seed=1, num_points=1024, num_regions=r,
                              device=dev(), exp_folder=exp, softmax_type="modified")
    pts, label = synth.make_cloud(ci)
    monkeypatch.setattr(gen_pair, "data_loader", lambda a: [(torch.from_numpy(pts).unsqueeze(0), torch.tensor([label]))])
    monkeypatch.setattr(gen_pair, "load_model", lambda a: pointnet)
    gen_pair.check_adv_success(args, pose_sweep.rotate_xyz, [name])
    gen_pair.save_pair_single_region(args, [name])
    gen_pair.save_pred_label(args, pose_sweep.rotate_xyz, [name])
    adv = base + "interaction_seed1/rotate_adv/"
    assert int(np.load(adv + "pose_idx.npy")) == int(g["pose_idx"])
    assert np.array_equal(np.load(adv + "transform_params.npy"), g["transform_params"])
    assert np.array_equal(np.load(adv + "pred_labels.npy"), g["pred_labels"])
    single = base + "interaction_seed1/rotate_adv_single_region/"
    want_names = sorted("range_rank%02d_region%02d" % (g["range_rank"][k], k) for k in range(r))
    assert sorted(os.listdir(single)) == want_names
    off = np.concatenate([[0], np.cumsum(g["single_pair_counts"])])
    for k in range(r):
        f = single + "range_rank%02d_region%02d/" % (g["range_rank"][k], k)
        got = np.load(f + "region_pair_list.npy").reshape(-1, 2)
        assert np.array_equal(got, g["single_pairs"][off[k]:off[k + 1]].astype(np.int64).reshape(-1, 2)), k
        assert int(np.load(f + "max_pose/pose_idx.npy")) == g["max_pose_idx"][k]
        assert int(np.load(f + "min_pose/pose_idx.npy")) == g["min_pose_idx"][k]
        assert np.array_equal(np.load(f + "max_pose/transform_params.npy"), g["angle_tuple"][g["max_pose_idx"][k]])
        labels = np.concatenate([np.load(f + "max_pose/pred_labels.npy"), np.load(f + "min_pose/pred_labels.npy")])
        assert np.array_equal(labels, g["single_pred_labels"][k])


def test_config0_at_spec_30_clouds_8_regions_64_permutations(pointnet):
    """BASELINE.json configs[0], every one of its 17 280 coalitions: region ids, FPS picks, rewards and Shapley values of
    all 30 clouds against the reference's CPU run (permutation stream continuing from cloud to cloud)."""
    g = load_golden("pointnet_config0.npz")
    r, s, bs = int(g["num_regions"]), int(g["num_samples"]), int(g["bs"])
    args = argparse.Namespace(model="pointnet", softmax_type="modified", num_points=1024, num_regions=r, num_samples=s,
                              shapley_batch_size=bs, verbose=False)
    n_clouds = g["phi"].shape[0]
    assert n_clouds == 30 and g["v"].shape == (30, s * (r + 1))
    np.random.seed(1)                                                   # set_random(1), tools/final_util.py:113-120
    worst = 0.0
    for ci in range(n_clouds):
        pts, label = synth.make_cloud(ci)
        data = torch.from_numpy(pts).unsqueeze(0).to(dev())
        lbl = torch.tensor([label], device=dev())
        fps = hip_ops.fps(data, r)[0]
        assert np.array_equal(fps.cpu().numpy(), g["fps_index"][ci])
        rid = hip_ops.region_assign(data[0].contiguous(), fps.contiguous()).cpu().numpy()
        assert (rid != g["region_id"][ci]).sum() <= 2                   # true near-ties only (test_hip_parity.py::test_region_assign)
        # final_shapley_value.py:59-72 through the product's device sampler (iq_sample_permutations continues NumPy's stream)
        orders = shapley_stage.generate_all_orders(None, argparse.Namespace(device=dev(), num_samples_save=s, num_regions=r), save=False)
        assert orders.dtype == np.int64 and np.array_equal(orders, g["orders"][ci])
        phi, logits = final_common.shap_sampling_all_regions_batch(pointnet, data, lbl, g["region_id"][ci].astype(np.int64), orders, args)
        v = final_common.get_reward(logits, lbl, args).cpu().numpy()
        vmax = np.abs(g["v"][ci]).max()
        worst = max(worst, np.abs(v - g["v"][ci]).max() / vmax)
        assert np.abs(v - g["v"][ci]).max() < RTOL * vmax
        assert np.abs(phi - g["phi"][ci]).max() < RTOL * np.abs(g["phi"][ci]).max()
        assert abs(phi.sum() - g["norm_factor"][ci]) < 1e-4 * max(abs(g["norm_factor"][ci]), 1.0)   # efficiency
        if ci < 2:
            assert_close_elementwise(logits.cpu().numpy(), g["logits_first2"][ci])
    print("configs[0]: worst reward error %.2e of max |v| over 17 280 coalitions" % worst)


def _knn_min_margin(sd, x):
    """Smallest relative gap between the 20th and a later kNN candidate over the three feature-space graphs of ONE cloud
    x (1,3,N) (models/dgcnn.py:12-18 distances through the oracle): below float32 resolution a neighbour may flip."""
    from oracle import ref_cpu as O
    with torch.no_grad():
        _, aux = O.dgcnn_forward(sd, x, 20, False, return_aux=True)
        best = 1.0
        for t in (aux["x1"], aux["x2"], aux["x3"]):
            inner = torch.matmul(t.transpose(2, 1), t) * -2
            xx = torch.sum(t ** 2, dim=1, keepdim=True)
            d = (-xx - inner - xx.transpose(2, 1))[0]
            top = d.topk(40, dim=-1)[0]
            gap = top[:, 19:20] - top[:, 20:]
            gap = torch.where(gap > 0, gap, torch.full_like(gap, float("inf"))).min(dim=1)[0]
            best = min(best, float((gap / xx.max()).min()))
    return best


_SCALE_CACHE = {}


def _scale_logits(name):
    """HIP logits of the 2016 interaction coalitions of <name>_scale.npz (once per test session), beside the reference's
    float32 and float64 logits: {tag: (contexts, got, ref32, ref64)}, plus the fixture, the CPU cloud and the state dict."""
    if name in _SCALE_CACHE:
        return _SCALE_CACHE[name]
    from interpret_quality_amd.dgcnn import GCNN_cls
    g = load_golden("%s_scale.npz" % name)
    sd = synth.to_torch(synth.dgcnn_state_dict(0))
    model = {"dgcnn": DGCNN_cls, "gcnn": GCNN_cls}[name](argparse.Namespace(dataset="modelnet10", k=20))
    model.load_state_dict(sd)
    model = model.to(dev()).eval()
    pts, _ = synth.make_cloud(int(g["cloud_id"]))
    data_cpu = torch.from_numpy(pts).unsqueeze(0)
    args = argparse.Namespace(model=name, softmax_type="modified", num_regions=32, interaction_batch_size=6)
    rows = {}
    for ratio in g["ratios"]:
        tag = "ratio%d" % int(ratio * 100)
        ctx = g[tag + "_contexts"].astype(np.int64)
        got = interaction.compute_order_interaction_logits(model, data_cpu.to(dev()), g["region_id"].astype(np.int64),
                                                           g["pairs"].astype(np.int64), ctx, args).cpu().numpy()
        rows[tag] = (ctx, got, g[tag + "_logits"], g[tag + "_logits_fp64"])
    _SCALE_CACHE[name] = (g, sd, data_cpu, rows)
    return _SCALE_CACHE[name]


def test_dgcnn_parity_rate_at_scale():
    """2016 DGCNN interaction coalitions.  DGCNN rebuilds its kNN graph in feature space at every layer (models/dgcnn.py:12-18);
    a coalition on a kNN near-tie flips a neighbour under ANY change of rounding, after which its logits move at the 1e-3
    level - for the reference too: the fixture holds the reference's float32 AND float64 logits of the same inputs, and they
    disagree beyond 1e-4 on a few per cent of the coalitions.  The bars, none of them fitted to a measurement of this path:

    (a) against exact arithmetic the HIP path is AT LEAST AS WELL CONDITIONED AS THE REFERENCE: the number of coalitions whose
        logits differ from the reference's float64 logits by more than 1e-4 (of max |logit|) is no larger than the number on
        which the reference's own float32 run does;
    (b) every coalition on which the HIP path differs from the reference's float32 run by more than 1e-4 is explained: the
        reference's float32 run is itself off its float64 run there, or one of the coalition's OWN three feature-space graphs
        has a kNN margin (gap between the 20th and a later candidate) below 1e-5 of the squared feature norm - no
        well-separated coalition deviates;
    (c) the median error against the reference's float32 logits is at float32 level (< 1e-5)."""
    from oracle import ref_cpu as O
    g, sd, data_cpu, rows = _scale_logits("dgcnn")
    region_id, pairs = g["region_id"].astype(np.int64), g["pairs"].astype(np.int64)
    center = torch.mean(data_cpu, dim=1).squeeze()
    e32, e64, r64, unexplained = [], [], [], []
    for tag, (ctx, got, ref32, ref64) in rows.items():
        scale = np.abs(ref32).max()
        err32 = np.abs(got - ref32).max(axis=-1) / scale          # (P, 4C): HIP vs the reference's float32 run
        err64 = np.abs(got - ref64).max(axis=-1) / scale          # HIP vs the reference's float64 run
        ref_err = np.abs(ref32 - ref64).max(axis=-1) / scale      # the reference against itself across precisions
        e32.append(err32.reshape(-1)), e64.append(err64.reshape(-1)), r64.append(ref_err.reshape(-1))
        for p, c in zip(*np.nonzero((err32 >= RTOL) & (ref_err < RTOL))):
            masked = O.interaction_masked_batch(data_cpu.permute(0, 2, 1), center, region_id, pairs[p][0], pairs[p][1], ctx[p])
            margin = _knn_min_margin(sd, masked[c:c + 1].contiguous())
            if margin >= 1e-5:
                unexplained.append((tag, int(p), int(c), float(err32[p, c]), margin))
    e32, e64, r64 = np.concatenate(e32), np.concatenate(e64), np.concatenate(r64)
    n = e32.size
    n_hip64, n_ref64, n_hip32 = int((e64 >= RTOL).sum()), int((r64 >= RTOL).sum()), int((e32 >= RTOL).sum())
    print("DGCNN at scale, %d coalitions; above 1e-4 of max |logit|: HIP vs reference-fp64 %d (%.2f %%), reference-fp32 vs reference-fp64 "
          "%d (%.2f %%), HIP vs reference-fp32 %d (%.2f %%); median HIP vs fp32 %.2e, vs fp64 %.2e; worst vs fp64: HIP %.2e, reference-fp32 %.2e"
          % (n, n_hip64, 100.0 * n_hip64 / n, n_ref64, 100.0 * n_ref64 / n, n_hip32, 100.0 * n_hip32 / n, np.median(e32), np.median(e64),
             e64.max(), r64.max()))
    assert n >= 2000
    assert n_hip64 <= n_ref64, "HIP is off exact arithmetic on %d coalitions, the reference's own float32 run on %d" % (n_hip64, n_ref64)
    assert not unexplained, "deviation from the reference without a near-tie: %s" % (unexplained[:5],)
    assert np.median(e32) < 1e-5


@pytest.mark.parametrize("name", ["dgcnn", "gcnn"])
def test_context_averaged_interactions_at_scale(name):
    """What the paper publishes from these logits is not a single I_ij(S) but its average over the sampled contexts
    (final_cal_interactions.py:28-36 then plot_interaction.py:40-41: `I.mean()` and `|I.mean(axis=1)|.mean()` per order), on the
    2016-coalition fixtures.

    GCNN (one fixed xyz graph): every logit row, every per-(pair, order) context mean of I_ij and both published per-order
    figures are within 1e-4 (of max |logit| / max |v|) of the reference's float32 run.
    DGCNN: the reference's float32 run is itself off its float64 run by more than 1e-4 * max|v| on some (pair, order) means
    (a coalition on a kNN near-tie, see test_dgcnn_parity_rate_at_scale), so the yardstick is the reference's FLOAT64 result:
    the two published per-order figures and the worst per-(pair, order) context mean are within 1e-4 * max|v| of it (measured:
    9.4e-6; the reference's own float32 run against the same yardstick is printed beside it)."""
    g, _, _, rows = _scale_logits(name)
    label = int(g["label"])
    lbl = torch.tensor([label], device=dev())
    args = argparse.Namespace(model=name, softmax_type="modified", num_regions=32)

    def inter(logits):
        return interaction.compute_order_interaction(torch.from_numpy(np.ascontiguousarray(logits)).to(dev()), lbl, args)      # (P, C)

    worst_pair, worst_pair_ref, worst_pub = 0.0, 0.0, 0.0
    for tag, (ctx, got, ref32, ref64) in rows.items():
        truth = ref32 if name == "gcnn" else ref64
        if name == "gcnn":
            assert np.abs(got - ref32).max() < RTOL * np.abs(ref32).max()
        i_got, i_true, i_ref32 = inter(got), inter(truth), inter(ref32)
        vmax = np.abs(hip_ops.reward(torch.from_numpy(ref32).reshape(-1, ref32.shape[-1]).to(dev()), label).cpu().numpy()).max()
        worst_pair = max(worst_pair, np.abs(i_got.mean(axis=1) - i_true.mean(axis=1)).max() / vmax)
        worst_pair_ref = max(worst_pair_ref, np.abs(i_ref32.mean(axis=1) - i_true.mean(axis=1)).max() / vmax)
        d_mean = abs(i_got.mean() - i_true.mean()) / vmax
        d_abs = abs(np.abs(i_got.mean(axis=1)).mean() - np.abs(i_true.mean(axis=1)).mean()) / vmax
        worst_pub = max(worst_pub, d_mean, d_abs)
    print("%s: context-averaged interactions, error in units of max |v|: published per-order figures %.2e, worst (pair, order) mean %.2e "
          "(the reference's float32 run against the same yardstick: %.2e)" % (name, worst_pub, worst_pair, worst_pair_ref))
    assert worst_pub < RTOL
    assert worst_pair < RTOL


@pytest.mark.parametrize("name", ["pointnet2", "pointconv"])
def test_pointnet2_and_pointconv_at_32_regions(name):
    """families_r32.npz: Shapley (4 permutations x 33 prefix coalitions) and interaction (5 pairs x 3 ratios x 4 contexts x 4)
    rows of cloud 7 at R = 32 from the reference's CPU run - 372 coalitions per model beside round 1's 18 at R = 8."""
    from interpret_quality_amd.pointconv import PointConvDensityClsSsg
    from interpret_quality_amd.pointnet2 import PointNet2ClsMsg
    g = load_golden("families_r32.npz")
    cls, sd = {"pointnet2": (PointNet2ClsMsg, synth.pointnet2_state_dict), "pointconv": (PointConvDensityClsSsg, synth.pointconv_state_dict)}[name]
    model = cls(None)
    model.load_state_dict(synth.to_torch(sd(0)))
    model = model.to(dev()).eval()
    pts, label = synth.make_cloud(int(g["cloud_id"]))
    data = torch.from_numpy(pts).unsqueeze(0).to(dev())
    lbl = torch.tensor([label], device=dev())
    region_id = g["region_id"].astype(np.int64)
    args = argparse.Namespace(model=name, softmax_type="modified", num_points=1024, num_regions=32, num_samples=4, shapley_batch_size=2,
                              interaction_batch_size=4, verbose=False)
    phi, logits = final_common.shap_sampling_all_regions_batch(model, data, lbl, region_id, g[name + "_orders"].astype(np.int64), args)
    want = g[name + "_shap_logits"]
    assert np.abs(logits.cpu().numpy() - want).max() < RTOL * np.abs(want).max()
    assert_close_elementwise(logits.cpu().numpy(), want)
    assert np.abs(phi - g[name + "_phi"]).max() < RTOL * max(np.abs(g[name + "_phi"]).max(), np.abs(want).max() * 0.1)
    pairs = g[name + "_pairs"].astype(np.int64)
    for ratio in g["ratios"]:
        tag = "%s_ratio%d" % (name, int(ratio * 100))
        lg = interaction.compute_order_interaction_logits(model, data, region_id, pairs, g[tag + "_contexts"].astype(np.int64), args)
        want = g[tag + "_logits"]
        assert np.abs(lg.cpu().numpy() - want).max() < RTOL * np.abs(want).max()
        assert_close_elementwise(lg.cpu().numpy(), want)
        inter = interaction.compute_order_interaction(lg, lbl, args)
        vmax = np.abs(hip_ops.reward(torch.from_numpy(want).reshape(-1, 10).to(dev()), label).cpu().numpy()).max()
        assert np.abs(inter - g[tag + "_interaction"]).max() < RTOL * vmax
