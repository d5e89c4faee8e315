"""GPU parity: PointConv (density, kNN grouping, weighted aggregation, full model) against golden vectors
from the reference."""
import argparse

import numpy as np
import pytest
import torch

from conftest import assert_close_elementwise, load_golden
from interpret_quality_amd import final_common, hip_ops, synth
from interpret_quality_amd.pointconv import PointConvDensityClsSsg

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def dev():
    return torch.device("cuda:0")


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.fixture(scope="module")
def model():
    m = PointConvDensityClsSsg(None)
    m.load_state_dict(synth.to_torch(synth.pointconv_state_dict(0)))
    return m.to(dev()).eval()


def masked_clouds():
    from oracle import ref_cpu as O
    g = load_golden("pointconv.npz")
    pts, _ = synth.make_cloud(0)
    data = torch.from_numpy(pts).unsqueeze(0)
    center = torch.mean(data, dim=1).squeeze()
    return O.shapley_masked_batch(data, center, g["orders"], g["region_id"])  # (18,1024,3)


def test_pointconv_forward_matches_reference(model):
    g = load_golden("pointconv.npz")
    logits = model(masked_clouds().permute(0, 2, 1).contiguous().to(dev()))
    err = np.abs(logits.cpu().numpy() - g["logits"]).max(axis=1) / np.abs(g["logits"]).max()
    assert err.max() < RTOL, err


def test_pointconv_shapley_matches_reference(model):
    g = load_golden("pointconv.npz")
    pts, label = synth.make_cloud(0)
    data = torch.from_numpy(pts).unsqueeze(0).to(dev())
    lbl = torch.tensor([label], device=dev())
    args = argparse.Namespace(model="pointconv", softmax_type="modified", num_points=1024, num_regions=8, num_samples=2,
                              shapley_batch_size=2, verbose=False)
    phi, logits = final_common.shap_sampling_all_regions_batch(model, data, lbl, g["region_id"], g["orders"], args)
    assert rel_err(logits.cpu().numpy(), g["shap_logits"]) < RTOL
    assert_close_elementwise(logits.cpu().numpy(), g["shap_logits"])   # and element-wise, with an absolute floor (conftest.py)
    assert np.abs(phi - g["phi"]).max() < RTOL * np.abs(g["phi"]).max()


def test_pointconv_raw_clouds_vs_oracle(model):
    from oracle import ref_cpu as O
    x = torch.stack([torch.from_numpy(synth.make_cloud(i)[0]) for i in (3, 4)]).permute(0, 2, 1).contiguous()
    want = O.PointConvOracle(synth.to_torch(synth.pointconv_state_dict(0)))(x)
    got = model(x.to(dev()))
    assert rel_err(got.cpu().numpy(), want.numpy()) < RTOL
    assert_close_elementwise(got.cpu().numpy(), want.numpy())   # and element-wise, with an absolute floor (conftest.py)


@pytest.mark.parametrize("nclouds", [2, 8])
def test_coalitions_from_source_lists_equal_the_forward_on_masked_clouds(model, nclouds):
    """iq_pointconv_coalitions (masked clouds written inside; sa1 / sa2 groups from the source clouds' sorted neighbour
    lists, sa1's MLP rows from the per-cloud pair table) against iq_pointconv_forward on the materialised masked clouds
    (pc_knn_kernel, grouped MLP): the same groups up to ties
    and the choice among interchangeable masked points, so the logits agree to summation-order rounding - and the
    kNN-kernel path of the same entry point (tuning key 5 = 14) does too.  Nothing / few / many / everything masked, two
    source clouds (and [r4] eight - the poses of one sweep launch, each with its own pair table); 40 / 72 coalitions so that the
    lists are used (clouds * 8 <= coalitions)."""
    from interpret_quality_amd import _lib
    lib = _lib.load()
    d = dev()
    rng = np.random.default_rng(8)
    clouds = torch.stack([torch.from_numpy(synth.make_cloud(i)[0]) for i in range(nclouds)]).to(d)
    rid = torch.empty((nclouds, 1024), dtype=torch.int32)
    for c in range(nclouds):
        data = clouds[c:c + 1]
        rid[c] = hip_ops.region_assign(data[0].contiguous(), hip_ops.fps(data, 32)[0].contiguous()).cpu()
    centers = clouds.mean(dim=1)
    full = (1 << 32) - 1
    keep = [full, 0, full ^ (1 << 7), 1 << 3, 0x0f0f0f0f, 0xffff, 7, full ^ 1] + [int(x) for x in rng.integers(0, 1 << 32, size=8 * nclouds + 16)]
    cloud_of = [i % nclouds for i in range(len(keep))]
    keep_t = hip_ops.masks_to_tensor(keep, d)
    co_t = torch.tensor(cloud_of, dtype=torch.int32, device=d)
    got = model.coalition_logits(clouds, centers, rid.to(d), keep_t, co_t, num_regions=32)
    dense = [hip_ops.mask_coalitions(clouds[c], rid[c].to(d), hip_ops.masks_to_tensor([k], d), centers[c].contiguous())[0]
             for k, c in zip(keep, cloud_of)]
    want = model.forward_points(torch.stack(dense))
    assert rel_err(got.cpu().numpy(), want.cpu().numpy()) < 2e-5
    lib.iq_set_tuning(5, 15)       # lists, but sa1's rows from the grouped MLP instead of the pair table
    try:
        grouped = model.coalition_logits(clouds, centers, rid.to(d), keep_t, co_t, num_regions=32)
    finally:
        lib.iq_set_tuning(5, 0)
    assert rel_err(got.cpu().numpy(), grouped.cpu().numpy()) < 2e-5 and rel_err(grouped.cpu().numpy(), want.cpu().numpy()) < 2e-5
    lib.iq_set_tuning(5, 14)
    try:
        knn = model.coalition_logits(clouds, centers, rid.to(d), keep_t, co_t, num_regions=32)
    finally:
        lib.iq_set_tuning(5, 0)
    assert torch.equal(knn, want)      # the same kernels on the same masked clouds
    # [r4] sa1's contraction and its 2048 -> 128 layer in one kernel (pc_tab_fused_kernel, the default) against the two-kernel
    # form that writes the (B, 512, 2048) contractions (5 = 31): same table rows, same weights, the sum over a group's 32
    # members in 4-member MFMA steps instead of sequentially
    lib.iq_set_tuning(5, 31)
    try:
        two = model.coalition_logits(clouds, centers, rid.to(d), keep_t, co_t, num_regions=32)
    finally:
        lib.iq_set_tuning(5, 0)
    assert rel_err(got.cpu().numpy(), two.cpu().numpy()) < 2e-6 and not torch.equal(two, want)
    # a coalition's logits do not depend on what else is in the launch (the fused kernel's tiles are per coalition)
    solo = model.coalition_logits(clouds, centers, rid.to(d), keep_t[5:6].contiguous(), co_t[5:6].contiguous(), num_regions=32)
    assert torch.equal(solo[0], got[5])


def test_per_cloud_tables_are_kept_across_launches_and_never_go_stale(model):
    """iq_pointconv_coalitions_cached: the sorted lists and sa1 pair tables of the source clouds stay at the head of the workspace
    between launches that pass the same cloud / centre tensors (chunks of an interaction ratio, batches of a pose).  Same logits,
    bit for bit, as an engine that builds them for every launch; a dense forward in between (it overwrites the head), other
    tensors, an in-place write into the same tensor, a different number of coalitions (every B-dependent offset moves) and a
    larger workspace (re-allocation) are all noticed."""
    d = dev()
    rng = np.random.default_rng(3)
    eng = model.engine()

    def setup(ids):
        clouds = torch.stack([torch.from_numpy(synth.make_cloud(i)[0]) for i in ids]).to(d)
        rid = torch.stack([hip_ops.region_assign(clouds[c].contiguous(), hip_ops.fps(clouds[c:c + 1], 32)[0].contiguous()) for c in range(len(ids))])
        return clouds, clouds.mean(dim=1).contiguous(), rid.contiguous()

    def fresh(clouds, centers, rid, keep, co):
        eng._tab = {}
        return model.coalition_logits(clouds, centers, rid, keep, co, num_regions=32)

    a = setup((2, 3))
    keep = hip_ops.masks_to_tensor([int(x) for x in rng.integers(0, 1 << 32, size=48)], d)
    co = torch.tensor([i % 2 for i in range(48)], dtype=torch.int32, device=d)
    first = fresh(*a, keep, co)
    assert eng._tab.get("state") == 3                                   # lists and pair tables built and remembered
    again = model.coalition_logits(*a, keep, co, num_regions=32)        # re-used
    assert eng._tab.get("state") == 3 and torch.equal(first, again)
    part = model.coalition_logits(*a, keep[:17].contiguous(), co[:17].contiguous(), num_regions=32)   # another B: re-used too
    assert torch.equal(part, first[:17])
    model.forward_points(a[0])                                          # the dense forward takes the head of the workspace
    assert eng._tab == {}
    assert torch.equal(model.coalition_logits(*a, keep, co, num_regions=32), first)
    b = setup((4, 5))                                                   # other clouds in new tensors
    got_b = model.coalition_logits(*b, keep, co, num_regions=32)
    assert torch.equal(got_b, fresh(*b, keep, co)) and not torch.equal(got_b, first)
    a[0][1, 7, 0] += 0.25                                               # the same tensor, one coordinate changed in place
    rid1 = hip_ops.region_assign(a[0][1].contiguous(), hip_ops.fps(a[0][1:2], 32)[0].contiguous())
    a2 = (a[0], a[0].mean(dim=1).contiguous(), torch.stack([a[2][0], rid1]).contiguous())
    moved = model.coalition_logits(*a2, keep, co, num_regions=32)
    assert torch.equal(moved, fresh(*a2, keep, co)) and not torch.equal(moved, first)
    big = hip_ops.masks_to_tensor([int(x) for x in rng.integers(0, 1 << 32, size=4000)], d)   # a larger workspace: re-allocated
    cob = torch.tensor([i % 2 for i in range(4000)], dtype=torch.int32, device=d)
    model.coalition_logits(*a2, big, cob, num_regions=32)
    assert torch.equal(model.coalition_logits(*a2, keep, co, num_regions=32), moved)


def test_sa2_on_the_bf16_matrix_pipe_equals_the_fp32_mfma_kernel_to_rounding(model):
    """sa2's grouped MLP (128 -> 128 -> 256) runs as six bf16 products per float32 product with float32 accumulation
    (pc_group_bf3_kernel); tuning key 5 = 56 selects the fp32-MFMA kernel.  Same logits to float32 rounding, on dense clouds and
    on coalitions."""
    from interpret_quality_amd import _lib
    d = dev()
    rng = np.random.default_rng(23)
    clouds = torch.stack([torch.from_numpy(synth.make_cloud(i)[0]) for i in (2, 7)]).to(d)
    rid = torch.stack([hip_ops.region_assign(clouds[c].contiguous(), hip_ops.fps(clouds[c:c + 1], 32)[0].contiguous()) for c in range(2)])
    centers = clouds.mean(dim=1)
    keep = [(1 << 32) - 1, 0xffff0000] + [int(x) | 0xff for x in rng.integers(0, 1 << 32, size=30, dtype=np.uint64)]
    keep_t = hip_ops.masks_to_tensor(keep, d)
    co_t = torch.tensor([i % 2 for i in range(len(keep))], dtype=torch.int32, device=d)
    got = model.coalition_logits(clouds, centers, rid, keep_t, co_t, num_regions=32)
    dense = model.forward_points(clouds)
    lib = _lib.load()
    lib.iq_set_tuning(5, 56)
    try:
        ref = model.coalition_logits(clouds, centers, rid, keep_t, co_t, num_regions=32)
        dense_ref = model.forward_points(clouds)
    finally:
        lib.iq_set_tuning(5, 0)
    assert not torch.equal(got, ref)                                      # a different kernel did run
    for a, b in ((got, ref), (dense, dense_ref)):
        assert (a - b).abs().max().item() < 3e-6 * b.abs().max().item()


@pytest.mark.parametrize("n", [300, 64])
def test_clouds_of_fewer_than_512_points(model, n):
    """sa1 samples 512 centroids: on a smaller cloud the reference's farthest point sampling returns index 0 once every point is
    taken (models/pointconv.py:54-77) and the 512 - N duplicate centroids take part in sa2 like any other point.  Dense forward
    against the CPU oracle; the coalition path (groups from the source cloud's lists) against the oracle on the masked clouds and
    against the dense forward on them."""
    from oracle import ref_cpu as O
    d = dev()
    sd = synth.to_torch(synth.pointconv_state_dict(0))
    orc = O.PointConvOracle(sd)

    def oracle_logits(x):                                        # x (B,N,3) on the CPU
        out = orc(x.permute(0, 2, 1).contiguous())
        return (out[0] if isinstance(out, tuple) else out).numpy()
    pts = torch.from_numpy(np.stack([synth.make_cloud(20 + i, num_points=n)[0] for i in range(3)]))
    want = oracle_logits(pts)
    got = model.forward_points(pts.to(d)).cpu().numpy()
    assert np.abs(got - want).max() / np.abs(want).max() < 1e-4
    rng = np.random.default_rng(n)
    clouds = pts[:2].to(d)
    rid = torch.from_numpy(rng.integers(0, 8, size=(2, n)).astype(np.int32)).to(d)
    centers = clouds.mean(dim=1)
    keep = [255, 0, 1, 3, 0x0f, 0xf0, 0x55, 0xaa, 254, 127]
    cloud_of = [i % 2 for i in range(len(keep))]
    got = model.coalition_logits(clouds, centers, rid, hip_ops.masks_to_tensor(keep, d),
                                 torch.tensor(cloud_of, dtype=torch.int32, device=d), num_regions=8).cpu().numpy()
    masked = torch.cat([hip_ops.mask_coalitions(clouds[c].contiguous(), rid[c].contiguous(), hip_ops.masks_to_tensor([k], d),
                                                centers[c].contiguous()) for k, c in zip(keep, cloud_of)])
    dense = model.forward_points(masked).cpu().numpy()
    assert np.abs(got - dense).max() / np.abs(dense).max() < 2e-5
    want = oracle_logits(masked.cpu())
    assert np.abs(got - want).max() / np.abs(want).max() < 1e-4
