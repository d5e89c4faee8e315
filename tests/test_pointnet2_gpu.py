"""GPU parity: PointNet++ MSG (FPS, ball query, grouped MLPs, full model) against golden vectors from
the reference and the CPU oracle."""
import argparse

import numpy as np
import pytest
import torch

from conftest import assert_close_elementwise, load_golden
from interpret_quality_amd import final_common, hip_ops, synth
from interpret_quality_amd.pointnet2 import PointNet2ClsMsg

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
def masked(oracle):
    g = load_golden("pointnet2.npz")
    pts, _ = synth.make_cloud(0)
    data = torch.from_numpy(pts).unsqueeze(0)
    center = torch.mean(data, dim=1).squeeze()
    return oracle.shapley_masked_batch(data, center, g["orders"], g["region_id"])  # (18,1024,3)


@pytest.fixture(scope="module")
def model():
    m = PointNet2ClsMsg(None)
    m.load_state_dict(synth.to_torch(synth.pointnet2_state_dict(0)))
    return m.to(dev()).eval()


def test_fps_two_levels_incl_exhaustion(masked):
    g = load_golden("pointnet2.npz")
    x = masked.to(dev())
    fps1 = hip_ops.fps(x, 512)
    assert np.array_equal(fps1.cpu().numpy(), g["fps1"])
    nx = torch.gather(x, 1, fps1.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    fps2 = hip_ops.fps(nx, 128)
    assert np.array_equal(fps2.cpu().numpy(), g["fps2"])
    assert (g["fps1"][0] == 0).all()  # the all-centre cloud: one distinct location


def _ball_mismatch(got, want, xyz, new_xyz, radius, oracle):
    """Index-valued: rows may differ only where a point sits within rounding of the sphere."""
    bad = np.nonzero((got != want).any(axis=-1))
    if len(bad[0]) == 0:
        return 0
    d = oracle.square_distance(new_xyz, xyz).numpy()
    r2 = np.float32(radius ** 2)
    for b, s in zip(*bad):
        near = np.abs(d[b, s] - r2) < 4e-7
        assert near.any(), "ball query row (%d,%d) differs without a boundary point" % (b, s)
    return len(bad[0])


def test_ball_query_matches_reference(masked, oracle):
    g = load_golden("pointnet2.npz")
    sel = list(g["sel"])
    xyz = masked[sel].contiguous()
    new_xyz = oracle.index_points(xyz, torch.from_numpy(g["fps1"][sel].astype(np.int64)))
    nbad = 0
    for r, k in ((0.1, 16), (0.2, 32), (0.4, 128)):
        got = hip_ops.ball_query(xyz.to(dev()), new_xyz.to(dev()), r, k).cpu().numpy()
        nbad += _ball_mismatch(got, g["sa1_ball_r%g" % r].astype(np.int32), xyz, new_xyz, r, oracle)
    new_xyz2 = oracle.index_points(new_xyz, torch.from_numpy(g["fps2"][sel].astype(np.int64)))
    for r, k in ((0.2, 32), (0.4, 64), (0.8, 128)):
        got = hip_ops.ball_query(new_xyz.contiguous().to(dev()), new_xyz2.contiguous().to(dev()), r, k).cpu().numpy()
        nbad += _ball_mismatch(got, g["sa2_ball_r%g" % r].astype(np.int32), new_xyz, new_xyz2, r, oracle)
    assert nbad <= 8  # of 3*(512+128)*3 rows


def test_pointnet2_forward_matches_reference(model, masked):
    g = load_golden("pointnet2.npz")
    logits = model(masked.permute(0, 2, 1).contiguous().to(dev()))
    assert rel_err(logits.cpu().numpy(), g["logits"]) < RTOL
    assert_close_elementwise(logits.cpu().numpy(), g["logits"])   # and element-wise, with an absolute floor (conftest.py)
    # channel-last entry point, split into several calls
    model.max_clouds_per_call = 7
    l2 = model.forward_points(masked.to(dev()))
    model.max_clouds_per_call = 2048
    assert torch.equal(l2, logits)


def test_pointnet2_shapley_matches_reference(model):
    g = load_golden("pointnet2.npz")
    pts, label = synth.make_cloud(0)
    data = torch.from_numpy(pts).unsqueeze(0).to(dev())
    lbl = torch.tensor([label], device=dev())
    args = argparse.Namespace(model="pointnet2", softmax_type="modified", num_points=1024, num_regions=8, num_samples=2,
                              shapley_batch_size=2, verbose=False)
    phi, logits = final_common.shap_sampling_all_regions_batch(model, data, lbl, g["region_id"], g["orders"], args)
    assert rel_err(logits.cpu().numpy(), g["shap_logits"]) < RTOL
    assert_close_elementwise(logits.cpu().numpy(), g["shap_logits"])   # and element-wise, with an absolute floor (conftest.py)
    assert np.abs(phi - g["phi"]).max() < RTOL * np.abs(g["phi"]).max()


def test_pointnet2_raw_clouds_vs_oracle(model, oracle):
    x = torch.stack([torch.from_numpy(synth.make_cloud(i)[0]) for i in (3, 4)]).permute(0, 2, 1).contiguous()
    want = oracle.PointNet2Oracle(synth.to_torch(synth.pointnet2_state_dict(0)))(x)
    got = model(x.to(dev()))
    assert rel_err(got.cpu().numpy(), want.numpy()) < RTOL
    assert_close_elementwise(got.cpu().numpy(), want.numpy())   # and element-wise, with an absolute floor (conftest.py)


def test_coalitions_with_pair_tables_equal_the_forward_on_masked_clouds(model):
    """iq_pointnet2_coalitions (sa1 = gather-max over per-cloud pair tables) against iq_pointnet2_forward on the
    materialised masked clouds: nothing / few / many / everything masked, two source clouds (cloud_of)."""
    d = dev()
    rng = np.random.default_rng(5)
    clouds = torch.stack([torch.from_numpy(synth.make_cloud(i)[0]) for i in (0, 1)]).to(d)           # (2,1024,3)
    rid = torch.from_numpy(rng.integers(0, 32, size=(2, 1024)).astype(np.int32))
    for c in range(2):   # spatially coherent regions as in the real pipeline
        data = clouds[c:c + 1]
        rid[c] = hip_ops.region_assign(data[0].contiguous(), hip_ops.fps(data, 32)[0].contiguous()).cpu()
    centers = clouds.mean(dim=1)
    full = (1 << 32) - 1
    keep = [full, 0, full ^ (1 << 7), 1 << 3, 0x0f0f0f0f, 0xffff, full, 1 << 5, full ^ 1, 0xaaaaaaaa, 7, 0]
    cloud_of = [0] * 6 + [1] * 6
    keep_t = hip_ops.masks_to_tensor(keep, d)
    co_t = torch.tensor(cloud_of, dtype=torch.int32, device=d)
    got = model.coalition_logits(clouds, centers, rid.to(d), keep_t, co_t, num_regions=32)
    dense = [hip_ops.mask_coalitions(clouds[c], rid[c].to(d), hip_ops.masks_to_tensor([k], d), centers[c].contiguous())[0]
             for k, c in zip(keep, cloud_of)]
    want = model.forward_points(torch.stack(dense))
    assert torch.equal(got, want)   # same arithmetic per member row, max is order-independent


def test_region_reduced_tables_equal_the_member_walk(model):
    """sa1 from the region-reduced pair tables (csrc/iq_pointnet2.hip: pt_regtab_kernel - one row per kept region that reaches
    into a ball) against the member walk over the pair rows (tuning key 5 = 21): bit-identical logits on 400 random coalitions
    of two source clouds, among them the empty, the full and single-region ones."""
    from interpret_quality_amd import _lib
    d = dev()
    rng = np.random.default_rng(11)
    clouds = torch.stack([torch.from_numpy(synth.make_cloud(i)[0]) for i in (2, 6)]).to(d)
    rid = torch.stack([hip_ops.region_assign(clouds[c].contiguous(), hip_ops.fps(clouds[c:c + 1], 32)[0].contiguous()) for c in range(2)])
    centers = clouds.mean(dim=1)
    keep = [0, (1 << 32) - 1] + [1 << r for r in range(0, 32, 5)] + [int(x) for x in rng.integers(0, 1 << 32, size=390, dtype=np.uint64)]
    keep_t = hip_ops.masks_to_tensor(keep, d)
    co_t = torch.tensor([i % 2 for i in range(len(keep))], dtype=torch.int32, device=d)
    got = model.coalition_logits(clouds, centers, rid, keep_t, co_t, num_regions=32)
    lib = _lib.load()
    lib.iq_set_tuning(5, 21)
    try:
        walk = model.coalition_logits(clouds, centers, rid, keep_t, co_t, num_regions=32)
    finally:
        lib.iq_set_tuning(5, 0)
    assert torch.equal(got, walk)


def test_sa2_on_the_bf16_matrix_pipe_equals_the_fp32_mfma_kernel_to_rounding(model):
    """The 128-128-256 scales of sa2 run their grouped MLP as six bf16 products per float32 product, float32 accumulation
    (pn2_group_bf3_kernel, 64-row chunks); tuning key 5 = 56 / 64 selects the fp32-MFMA kernel (32- / 64-row chunks), which
    gives the same bits for either chunk size.  Same logits to float32 rounding (the two sum k in different orders), on dense
    clouds and on coalitions with few and many kept points."""
    from interpret_quality_amd import _lib
    d = dev()
    rng = np.random.default_rng(17)
    clouds = torch.stack([torch.from_numpy(synth.make_cloud(i)[0]) for i in (1, 5)]).to(d)
    rid = torch.stack([hip_ops.region_assign(clouds[c].contiguous(), hip_ops.fps(clouds[c:c + 1], 32)[0].contiguous()) for c in range(2)])
    centers = clouds.mean(dim=1)
    keep = [(1 << 32) - 1, 1 << 9, 0x00ff00ff] + [int(x) for x in rng.integers(0, 1 << 32, size=61, dtype=np.uint64)]
    keep_t = hip_ops.masks_to_tensor(keep, d)
    co_t = torch.tensor([i % 2 for i in range(len(keep))], dtype=torch.int32, device=d)
    got = model.coalition_logits(clouds, centers, rid, keep_t, co_t, num_regions=32)
    dense = model.forward_points(clouds)
    lib = _lib.load()
    ref = {}
    try:
        for knob in (56, 64):
            lib.iq_set_tuning(5, knob)
            ref[knob] = (model.coalition_logits(clouds, centers, rid, keep_t, co_t, num_regions=32), model.forward_points(clouds))
    finally:
        lib.iq_set_tuning(5, 0)
    assert torch.equal(ref[56][0], ref[64][0]) and torch.equal(ref[56][1], ref[64][1])
    assert not torch.equal(got, ref[56][0])                               # a different kernel did run
    for a, b in ((got, ref[56][0]), (dense, ref[56][1])):
        assert (a - b).abs().max().item() < 3e-6 * b.abs().max().item()
