"""GPU parity tests: the HIP path, called through the C ABI (ctypes), against the CPU oracle and
the golden vectors generated from the reference.  Bit-exact for masks / indices / reductions over
given v; 1e-4 relative (the north star's tolerance) for anything that runs the network."""
import argparse

import numpy as np
import pytest
import torch

from conftest import assert_close_elementwise, load_golden
from interpret_quality_amd import final_common, hip_ops, synth
from interpret_quality_amd.pointnet import PointNetCls

pytestmark = pytest.mark.gpu

RTOL = 1e-4  # north star: "within 1e-4 relative fp32"


def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
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


def cloud_setup(oracle, ci, num_regions):
    pts, label = synth.make_cloud(ci)
    data = torch.from_numpy(pts).unsqueeze(0)
    fps = oracle.farthest_point_sample(data, num_regions)[0]
    region_id = oracle.cal_region_id(data, fps)
    return data, label, fps.numpy(), region_id


# ------------------------------------------------------------------------------------------------
# masking
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("num_regions,bs", [(8, 4), (32, 50), (32, 1)])
@pytest.mark.parametrize("channel_first", [False, True])
def test_mask_shapley_bitwise(oracle, num_regions, bs, channel_first):
    data, _, _, region_id = cloud_setup(oracle, 0, num_regions)
    orders = synth.make_orders(bs, num_regions, seed=3)
    center = torch.mean(data, dim=1).squeeze()
    want = oracle.shapley_masked_batch(data, center, orders, region_id)
    if channel_first:
        want = want.permute(0, 2, 1).contiguous()
    d = dev()
    got = hip_ops.mask_shapley(data[0].to(d), hip_ops.as_i32(region_id, d), hip_ops.as_i32(orders, d),
                               center.to(d), channel_first=channel_first)
    assert torch.equal(got.cpu(), want)


@pytest.mark.parametrize("n", [1022, 130, 21])
@pytest.mark.parametrize("channel_first", [False, True])
def test_masks_on_cloud_sizes_that_are_no_multiple_of_four(oracle, n, channel_first):
    """mask_data_batch (tools/final_common.py:37-60) takes any cloud; so do the mask kernels (their float4 stores need N % 4 == 0,
    other sizes go element by element): Shapley prefixes and explicit coalitions, both layouts, bit for bit."""
    rng = np.random.default_rng(n)
    pts, _ = synth.make_cloud(3, num_points=n)
    data = torch.from_numpy(pts).unsqueeze(0)
    region_id = torch.from_numpy(rng.integers(0, 8, size=n).astype(np.int64))
    orders = synth.make_orders(3, 8, seed=n)
    center = torch.mean(data, dim=1).squeeze()
    want = oracle.shapley_masked_batch(data, center, orders, region_id)                 # (3 * 9, n, 3)
    d = dev()
    got = hip_ops.mask_shapley(data[0].to(d), hip_ops.as_i32(region_id, d), hip_ops.as_i32(orders, d), center.to(d),
                               channel_first=channel_first)
    assert torch.equal(got.cpu(), want.permute(0, 2, 1).contiguous() if channel_first else want)
    keep = [0, 255, 1, 0x5a, 0x81]
    got = hip_ops.mask_coalitions(data[0].to(d), hip_ops.as_i32(region_id, d), hip_ops.masks_to_tensor(keep, d), center.to(d),
                                  channel_first=channel_first).cpu()
    for row, k in zip(got, keep):
        kept = torch.tensor([(k >> int(r)) & 1 for r in region_id], dtype=torch.bool)
        ref = torch.where(kept[:, None], data[0], center[None, :])
        assert torch.equal(row, ref.t().contiguous() if channel_first else ref)


def test_mask_shapley_golden_checksum(oracle):
    import hashlib
    g = load_golden("pointnet_shapley_R32.npz")
    pts, _ = synth.make_cloud(0)
    data = torch.from_numpy(pts).unsqueeze(0)
    center = torch.mean(data, dim=1).squeeze()
    d = dev()
    got = hip_ops.mask_shapley(data[0].to(d), hip_ops.as_i32(g["c0_region_id"], d),
                               hip_ops.as_i32(g["c0_orders"][:int(g["bs"])], d), center.to(d))
    assert hashlib.sha256(got.cpu().numpy().tobytes()).hexdigest() == str(g["c0_masked_sha256"])


def test_mask_interaction_bitwise(oracle):
    g = load_golden("pointnet_interaction_R32.npz")
    pts, _ = synth.make_cloud(int(g["cloud_id"]))
    data = torch.from_numpy(pts).unsqueeze(0)
    center = torch.mean(data, dim=1).squeeze()
    d = dev()
    for tag in ("ratio0", "ratio4", "ratio50", "ratio100"):
        ctx = g[tag + "_contexts"]
        for p, (ri, rj) in enumerate(g["pairs"]):
            want = oracle.interaction_masked_batch(data.permute(0, 2, 1), center, g["region_id"], ri, rj, ctx[p])
            nb = ctx.shape[1]
            pairs = np.tile(np.array([[ri, rj]]), (nb, 1))
            masks = [hip_ops.region_bitmask(c) for c in ctx[p]]
            got = hip_ops.mask_interaction(data[0].to(d), hip_ops.as_i32(g["region_id"], d), hip_ops.as_i32(pairs, d),
                                           hip_ops.masks_to_tensor(masks, d), center.to(d), 32)
            assert torch.equal(got.cpu(), want.contiguous())


def test_mask_empty_batch():
    d = dev()
    cloud = torch.zeros((1024, 3), device=d)
    rid = torch.zeros((1024,), dtype=torch.int32, device=d)
    out = hip_ops.mask_shapley(cloud, rid, torch.zeros((0, 32), dtype=torch.int32, device=d), torch.zeros(3, device=d))
    assert out.shape == (0, 1024, 3)


# ------------------------------------------------------------------------------------------------
# reward / reductions
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("modified", [True, False])
def test_reward(oracle, modified):
    g = load_golden("pointnet_shapley_R32.npz")
    logits = torch.from_numpy(g["c1_logits"])
    for label in (0, 1, 9):
        want = oracle.get_reward(logits, torch.tensor([label]), "modified" if modified else "normal")
        got = hip_ops.reward(logits.to(dev()), label, modified)
        np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=2e-6, atol=2e-6)
    if not modified:
        np.testing.assert_allclose(hip_ops.reward(logits[:16].to(dev()), 1, False).cpu().numpy(), g["c1_v_normal16"],
                                   rtol=2e-6, atol=2e-6)


def test_shapley_accum_bitwise():
    rng = np.random.default_rng(0)
    s, r = 1000, 32
    orders = synth.make_orders(s, r, seed=2)
    v = rng.standard_normal(s * (r + 1)).astype(np.float32)
    # the reference's host loop (tools/final_common.py:92-96)
    phi = np.zeros((r,))
    rows = np.zeros((s, r))
    snaps_want = {}
    for o in range(s):
        vo = v[o * (r + 1):(o + 1) * (r + 1)]
        dv = vo[1:] - vo[:-1]
        phi[orders[o]] += dv
        rows[o, orders[o]] += dv
        if o + 1 in (100, 500, 1000):
            snaps_want[o + 1] = phi.copy()
    d = dev()
    got_phi, got_rows, snaps = hip_ops.shapley_accum(torch.from_numpy(v).to(d), hip_ops.as_i32(orders, d),
                                                     snap_counts=[100, 500, 1000])
    assert np.array_equal(got_phi.cpu().numpy(), phi)
    assert np.array_equal(got_rows.cpu().numpy(), rows)
    for k, c in enumerate((100, 500, 1000)):
        assert np.array_equal(snaps[k].cpu().numpy(), snaps_want[c])
    # efficiency: every permutation telescopes to v(N) - v(empty)
    np.testing.assert_allclose(got_rows.cpu().numpy().sum(1),
                               (v.reshape(s, r + 1)[:, -1].astype(np.float64) - v.reshape(s, r + 1)[:, 0]), atol=1e-5)


def test_interaction_reduce_bitwise(oracle):
    g = load_golden("pointnet_interaction_R32.npz")
    for tag in ("ratio4", "ratio50"):
        logits = torch.from_numpy(g[tag + "_logits"])
        lbl = torch.tensor([0])
        p, c4, k = logits.shape
        v = oracle.get_reward(logits.reshape(p * c4, k), lbl)
        got = hip_ops.interaction_reduce(v.to(dev())).cpu().numpy().astype(np.float64).reshape(p, c4 // 4)
        want = oracle.compute_order_interaction(logits, lbl)
        assert np.array_equal(got, want)


# ------------------------------------------------------------------------------------------------
# geometry
# ------------------------------------------------------------------------------------------------
def test_fps_golden():
    g = load_golden("geometry.npz")
    pts, _ = synth.make_cloud(3)
    data = torch.from_numpy(pts).unsqueeze(0)
    collapsed = data.clone()
    collapsed[0, 300:, :] = collapsed[0, :300].mean(dim=0)
    both = torch.cat([data, collapsed], dim=0).to(dev())
    for s in (32, 128, 512):
        got = hip_ops.fps(both, s).cpu().numpy()
        assert np.array_equal(got, g["fps_%d" % s]), "FPS S=%d differs" % s


@pytest.mark.parametrize("num_regions", [8, 32])
def test_region_assign(oracle, num_regions):
    g = load_golden("pointnet_shapley_R%d.npz" % num_regions)
    for ci in (0, 1):
        pts, _ = synth.make_cloud(ci)
        d = dev()
        got = hip_ops.region_assign(torch.from_numpy(pts).to(d), hip_ops.as_i32(g["c%d_fps_index" % ci], d)).cpu().numpy()
        want = g["c%d_region_id" % ci]
        diff = np.nonzero(got != want)[0]
        # index-valued: allow only genuine near-ties of the cancellation-prone expanded distance
        if len(diff):
            data = torch.from_numpy(pts).unsqueeze(0)
            dist = oracle.square_distance(data, data[:, torch.from_numpy(g["c%d_fps_index" % ci]), :])[0].numpy()
            for p in diff:
                assert abs(dist[p, got[p]] - dist[p, want[p]]) < 1e-6, "point %d assigned to a non-tied centre" % p
        assert len(diff) <= 2


# ------------------------------------------------------------------------------------------------
# PointNet
# ------------------------------------------------------------------------------------------------
def test_pointnet_dense_forward(model, oracle, pointnet_sd):
    g = load_golden("pointnet_dense.npz")
    x = torch.stack([torch.from_numpy(synth.make_cloud(i)[0]) for i in range(4)]).permute(0, 2, 1).contiguous()
    logits, trans_feat, crt = model(x.to(dev()))
    # crt_points (models/pointnet.py:83), the third element of the reference's tuple: the point that attains each pooled
    # channel's maximum, from the arg-max variant of the trunk kernel; the reference's own 4 x 1024 indices (a near-tie between
    # two points may resolve differently under a different fp32 summation order: at most 4 of 4096 may differ)
    assert crt.dtype == torch.int64 and tuple(crt.shape) == (4, 1024)
    assert int((crt.cpu().numpy() != g["crt_points"]).sum()) <= 4
    assert rel_err(logits.cpu().numpy(), g["logits"]) < RTOL
    assert_close_elementwise(logits.cpu().numpy(), g["logits"])   # and element-wise, with an absolute floor (conftest.py)
    o_logits, o_tf, _ = oracle.PointNetOracle(pointnet_sd)(x)
    assert rel_err(trans_feat.cpu().numpy(), o_tf.numpy()) < RTOL
    assert_close_elementwise(trans_feat.cpu().numpy(), o_tf.numpy())   # and element-wise, with an absolute floor (conftest.py)
    assert rel_err(logits.cpu().numpy(), o_logits.numpy()) < RTOL
    assert_close_elementwise(logits.cpu().numpy(), o_logits.numpy())   # and element-wise, with an absolute floor (conftest.py)


@pytest.mark.parametrize("num_regions", [8, 32])
def test_pointnet_coalitions_vs_golden_and_oracle(model, oracle, num_regions):
    """Shapley rows: fused coalition path == reference logits (golden) within 1e-4."""
    g = load_golden("pointnet_shapley_R%d.npz" % num_regions)
    d = dev()
    ns = int(g["num_samples"])
    for ci in g["cloud_ids"]:
        p = "c%d_" % ci
        pts, label = synth.make_cloud(int(ci))
        orders = g[p + "orders"]
        data = torch.from_numpy(pts).unsqueeze(0)
        center = torch.mean(data, dim=1)  # same centre bits as the reference run
        keep = []
        for o in range(ns):
            for i in range(num_regions + 1):
                keep.append(hip_ops.region_bitmask(orders[o][:i]))
        logits = model.coalition_logits(data.to(d), center.to(d), hip_ops.as_i32(g[p + "region_id"], d).reshape(1, -1),
                                        hip_ops.masks_to_tensor(keep, d), None, num_regions=num_regions)
        assert rel_err(logits.cpu().numpy(), g[p + "logits"]) < RTOL
        assert_close_elementwise(logits.cpu().numpy(), g[p + "logits"])   # and element-wise, with an absolute floor (conftest.py)
        v = hip_ops.reward(logits, label, True)
        phi, _, _ = hip_ops.shapley_accum(v, hip_ops.as_i32(orders, d))
        phi = phi.cpu().numpy() / ns
        assert np.abs(phi - g[p + "phi"]).max() < RTOL * np.abs(g[p + "phi"]).max()
        assert abs(phi.sum() - float(g[p + "norm_factor"])) < 1e-4 * abs(float(g[p + "norm_factor"])) + 1e-5


def test_pointnet_fused_equals_dense_on_materialised_clouds(model, oracle):
    """Exactness of the dedup: coalition path == the same kernels on the masked clouds, bitwise."""
    d = dev()
    data, _, _, region_id = cloud_setup(oracle, 2, 32)
    orders = synth.make_orders(3, 32, seed=5)
    center = torch.mean(data, dim=1)
    keep = [hip_ops.region_bitmask(orders[o][:i]) for o in range(3) for i in range(33)]
    rid = hip_ops.as_i32(region_id, d)
    fused = model.coalition_logits(data.to(d), center.to(d), rid.reshape(1, -1), hip_ops.masks_to_tensor(keep, d),
                                   None, num_regions=32)
    masked = hip_ops.mask_shapley(data[0].to(d), rid, hip_ops.as_i32(orders, d), center[0].to(d), channel_first=True)
    dense, _, _ = model(masked)
    assert torch.equal(fused, dense)


def test_pointnet_interaction_vs_golden(model):
    g = load_golden("pointnet_interaction_R32.npz")
    d = dev()
    pts, _ = synth.make_cloud(int(g["cloud_id"]))
    data = torch.from_numpy(pts).unsqueeze(0)
    center = torch.mean(data, dim=1)
    rid = hip_ops.as_i32(g["region_id"], d).reshape(1, -1)
    for tag in ("ratio0", "ratio4", "ratio50", "ratio100"):
        ctx = g[tag + "_contexts"]
        keep = []
        for p, (ri, rj) in enumerate(g["pairs"]):
            for c in ctx[p]:
                s = hip_ops.region_bitmask(c)
                bi, bj = 1 << int(ri), 1 << int(rj)
                keep += [s | bi | bj, s | bi, s | bj, s]
        logits = model.coalition_logits(data.to(d), center.to(d), rid, hip_ops.masks_to_tensor(keep, d), None,
                                        num_regions=32)
        want = g[tag + "_logits"].reshape(-1, 10)
        assert rel_err(logits.cpu().numpy(), want) < RTOL
        assert_close_elementwise(logits.cpu().numpy(), want)   # and element-wise, with an absolute floor (conftest.py)


def test_pointnet_multi_cloud_batch(model, oracle):
    """Coalitions of several clouds (poses) in one launch via cloud_of."""
    d = dev()
    datas, cents, rids = [], [], []
    for ci in (0, 1, 2):
        data, _, _, region_id = cloud_setup(oracle, ci, 8)
        datas.append(data[0]); cents.append(torch.mean(data, dim=1)[0]); rids.append(torch.from_numpy(region_id))
    clouds = torch.stack(datas).to(d)
    centers = torch.stack(cents).to(d)
    region_id = torch.stack(rids).to(device=d, dtype=torch.int32)
    keep = [0, 1, 3, 0xff, 0x55, 0xaa, 0x80, 0x7f, 0xff]
    cloud_of = [0, 0, 1, 1, 2, 2, 0, 1, 2]
    got = model.coalition_logits(clouds, centers, region_id, hip_ops.masks_to_tensor(keep, d),
                                 torch.tensor(cloud_of, dtype=torch.int32, device=d), num_regions=8)
    for k, (m, c) in enumerate(zip(keep, cloud_of)):
        one = model.coalition_logits(clouds[c:c + 1].contiguous(), centers[c:c + 1].contiguous(),
                                     region_id[c:c + 1].contiguous(), hip_ops.masks_to_tensor([m], d), None,
                                     num_regions=8)
        assert torch.equal(got[k], one[0])


# ------------------------------------------------------------------------------------------------
# the shared dense layer (register-streaming and LDS-staged GEMM variants)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,cin,cout,act", [(5000, 512, 1024, 2), (4133, 64, 128, 0), (2048, 128, 512, 1), (3000, 1024, 160, 1),
                                             (300, 256, 40, 0), (2500, 8, 128, 0),
                                             # [r4] the two large-M shapes with their own tilings: 320 outputs as column blocks of
                                             # exactly 10 tiles (NT = 5), 128 outputs on 256-row workgroup tiles (WN = 1)
                                             (262200, 320, 320, 1), (524400, 128, 128, 1), (525000, 64, 128, 0)])
def test_linear_variants_agree_and_match_fp32_reference(m, cin, cout, act):
    from interpret_quality_amd import _lib
    rng = np.random.default_rng(m + cin)
    w = (rng.standard_normal((cout, cin)) / np.sqrt(cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    x = rng.standard_normal((m, cin)).astype(np.float32)
    d = dev()
    layer = hip_ops.PackedLinear(w, b, d)
    xt = torch.from_numpy(x).to(d)
    lib = _lib.load()
    got = hip_ops.linear(xt, layer, act)                       # LDS-staged GEMM where the shape allows it
    lib.iq_set_tuning(3, 1)
    try:
        streaming = hip_ops.linear(xt, layer, act)             # register-streaming kernel only
    finally:
        lib.iq_set_tuning(3, 0)
    assert torch.equal(got, streaming)                         # same MFMA order over k: bit-identical
    ref = torch.from_numpy(x).double() @ torch.from_numpy(w).double().T + torch.from_numpy(b).double()
    if act == 1:
        ref = torch.relu(ref)
    elif act == 2:
        ref = torch.nn.functional.leaky_relu(ref, 0.2)
    assert rel_err(got.cpu().numpy(), ref.numpy()) < 1e-5


@pytest.mark.parametrize("m,cin,cout,act", [(5000, 512, 1024, 2), (1, 1024, 512, 1), (131, 4096, 256, 1), (40000, 128, 256, 0),
                                             (129, 32, 768, 1), (1025, 96, 256, 2),    # one chunk of k; three column blocks
                                             # [r5] 256 n + 64 outputs: whole column blocks on the bf16 pipe, the last 64 columns on
                                             # the fp32 MFMA (enough rows for the 256-row tiling of the 64-column rest; small; two blocks)
                                             (525000, 64, 320, 1), (700, 64, 320, 0), (2100, 128, 576, 2),
                                             # [r5] inputs that are no multiple of 32 (PointNet++ / PointConv sa3: 643 -> 648, 259 -> 264 columns):
                                             # the image's k range is padded with zeros, A's columns beyond cin are never read
                                             (3001, 648, 256, 1), (1500, 264, 256, 1), (130, 40, 256, 0)])
def test_dense_layer_on_the_bf16_matrix_pipe_is_float32_exact(m, cin, cout, act):
    """A layer that carries its weights as three bf16 terms (iq_dense_layer.w_bf3, cout % 256 == 0 or 64) takes six exact bf16 products
    per float32 product, accumulated in float32 (pn_gemm_bf3_kernel<false>), for every row count: at least as close to the
    float64 result as the fp32-MFMA kernel (tuning key 5 = 57), equal to it within float32 rounding, and a row's result does
    not depend on the rows around it."""
    from interpret_quality_amd import _lib
    rng = np.random.default_rng(m + cin)
    w = (rng.standard_normal((cout, cin)) / np.sqrt(cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    x = rng.standard_normal((m, cin)).astype(np.float32)
    d = dev()
    layer = hip_ops.PackedLinear(w, b, d, bf3=True)
    xt = torch.from_numpy(x).to(d)
    lib = _lib.load()
    got = hip_ops.linear(xt, layer, act)
    lib.iq_set_tuning(5, 57)
    try:
        f32 = hip_ops.linear(xt, layer, act)
    finally:
        lib.iq_set_tuning(5, 0)
    ref = torch.from_numpy(x).double() @ torch.from_numpy(w).double().T + torch.from_numpy(b).double()
    if act == 1:
        ref = torch.relu(ref)
    elif act == 2:
        ref = torch.nn.functional.leaky_relu(ref, 0.2)
    e_bf3, e_f32 = rel_err(got.cpu().numpy(), ref.numpy()), rel_err(f32.cpu().numpy(), ref.numpy())
    assert e_bf3 < 2e-6 and e_bf3 <= 1.5 * e_f32 + 1e-7, (e_bf3, e_f32)      # (float32 accumulation over cin terms in both)
    assert not torch.equal(got, f32) or m == 1
    assert (got - f32).abs().max().item() < 3e-6 * f32.abs().max().item()
    if cout % 256 == 64:                                                      # the fp32 rest of a 256 n + 64 layer
        assert torch.equal(got[:, -64:], f32[:, -64:]) and not torch.equal(got[:, :-64], f32[:, :-64])
    k = min(m, 77)
    assert torch.equal(hip_ops.linear(xt[:k].contiguous(), layer, act), got[:k])      # launch-size independent
    if cin % 32:                                    # nothing beyond a row's cin columns is read: NaNs behind the last row change nothing
        buf = torch.full((m * cin + 64,), float("nan"), dtype=torch.float32, device=d)
        buf[:m * cin] = xt.reshape(-1)
        assert torch.equal(hip_ops.linear(buf[:m * cin].view(m, cin), layer, act), got)


def test_bf16x3_split_loses_no_bit_of_a_float32():
    """The three-term split itself, through the kernel: with a signed power-of-two diagonal as the weight, a bf16x3 layer must
    return every input bit for bit (x = h + m + l exactly, each term times 2^k exactly, float32 accumulation of one nonzero
    product per term) - for ordinary values, for values whose low mantissa bits are all set, and across 60 binades."""
    rng = np.random.default_rng(9)
    d = dev()
    n = 256
    scale = np.ldexp(1.0, rng.integers(-6, 7, size=n)).astype(np.float32) * rng.choice([-1.0, 1.0], size=n).astype(np.float32)
    layer = hip_ops.PackedLinear(np.diag(scale).astype(np.float32), np.zeros(n, np.float32), d, bf3=True)
    x = rng.standard_normal((4096, n)).astype(np.float32)
    x[1000:2000] = np.ldexp(x[1000:2000], rng.integers(-30, 31, size=(1000, n))).astype(np.float32)      # 2^-30 .. 2^30
    x[2000:3000] = (x[2000:3000].view(np.uint32) | np.uint32(0xffff)).view(np.float32)                  # low 16 mantissa bits set
    x[3000:3100] = 0.0
    got = hip_ops.linear(torch.from_numpy(x).to(d), layer, 0).cpu().numpy()
    want = x * scale[None, :] + np.float32(0.0)          # the layer adds its (zero) bias: -0 becomes +0
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_pointnet_without_feature_transform():
    """`feature_transform=False` (models/pointnet.py:62-63,72-78; a constructor option no stage script takes): no feature STN -
    iq_pointnet_coalitions skips that chain and the trunk multiplies by the packed identity (exact).  Dense forward (logits,
    trans_feat None, crt_points) and the Shapley path against the reference's own run (pointnet_noft.npz)."""
    g = load_golden("pointnet_noft.npz")
    sd = synth.to_torch(synth.pointnet_state_dict(0, feature_transform=False))
    m = PointNetCls(argparse.Namespace(dataset="modelnet10", feature_transform=False))
    m.load_state_dict(sd)                                            # strict: the module has no feat.fstn.* parameters
    m = m.to(dev()).eval()
    x = torch.stack([torch.from_numpy(synth.make_cloud(int(i))[0]) for i in g["dense_cloud_ids"]]).permute(0, 2, 1).contiguous()
    logits, trans_feat, crt = m(x.to(dev()))
    assert trans_feat is None
    assert np.abs(logits.cpu().numpy() - g["dense_logits"]).max() < RTOL * np.abs(g["dense_logits"]).max()
    assert (crt.cpu().numpy() != g["dense_crt"]).sum() <= 4             # near-tie flips of an arg-max, as with the feature STN
    pts, label = synth.make_cloud(int(g["shap_cloud_id"]))
    args = argparse.Namespace(model="pointnet", softmax_type="modified", num_points=1024, num_regions=8, num_samples=4, shapley_batch_size=2,
                              verbose=False)
    phi, shap_logits = final_common.shap_sampling_all_regions_batch(m, torch.from_numpy(pts).unsqueeze(0).to(dev()),
                                                                    torch.tensor([label], device=dev()), g["region_id"].astype(np.int64),
                                                                    g["orders"].astype(np.int64), args)
    assert np.abs(shap_logits.cpu().numpy() - g["shap_logits"]).max() < RTOL * np.abs(g["shap_logits"]).max()
    assert np.abs(phi - g["phi"]).max() < RTOL * np.abs(g["phi"]).max()


def test_chain_tail_tiles_are_bit_identical_for_every_row_count(model):
    """Layer 3 of the chain kernel, for coalitions whose distinct-row counts cover every residue mod 32 (64 regions of 1 to 31
    points, 400 random coalitions):
    - the fp32-MFMA kernel (tuning key 5 = 54) with its 16-row tail tiles (l3_tail16: v_mfma_f32_16x16x4_f32 fed in the k order
      of the 32x32x2 fragments) against 32-row tiles only (5 = 55): bit-identical logits and packed feature transforms;
    - the default kernel (layers 2-3 as six bf16 products per float32 product, float32 accumulation) against the fp32-MFMA one:
      the same logits to float32 rounding (measured 3e-7 of the largest logit; the two sum k in different orders); two n-tiles
      per pass (default) against one (5 = 58): bit-identical;
    - and the dense forward (1024 rows) on the materialised clouds of a few coalitions, bitwise against the coalition path."""
    from interpret_quality_amd import _lib
    d = dev()
    rng = np.random.default_rng(3)
    pts, _ = synth.make_cloud(11)
    sizes = rng.integers(1, 32, size=64)
    sizes[-1] += 1024 - sizes.sum() if sizes.sum() <= 1024 else 0
    while sizes.sum() > 1024:
        sizes[rng.integers(0, 64)] = max(1, sizes[rng.integers(0, 64)] - 1)
        sizes = np.maximum(sizes, 1)
    sizes[0] += 1024 - sizes.sum()
    rid = np.repeat(np.arange(64), sizes)[:1024].astype(np.int32)
    rng.shuffle(rid)
    data = torch.from_numpy(pts).unsqueeze(0).to(d)
    center = torch.mean(data, dim=1).contiguous()
    keep = [int(x) for x in rng.integers(0, 1 << 63, size=398, dtype=np.uint64)] + [1, 3]
    keep_t = hip_ops.masks_to_tensor(keep, d)
    rid_t = torch.from_numpy(rid).to(d).reshape(1, -1)
    eng = model.engine()
    kept_rows = np.array([int(np.isin(rid, [r for r in range(64) if (k >> r) & 1]).sum()) for k in keep])
    rows = kept_rows + (kept_rows < 1024)
    assert len(set((rows % 32).tolist())) == 32                      # every residue occurs
    got, tf = eng.coalition_logits(data, center, rid_t, keep_t, None, num_regions=64, return_trans_feat=True)
    lib = _lib.load()
    try:
        lib.iq_set_tuning(5, 54)
        f32, tf_f32 = eng.coalition_logits(data, center, rid_t, keep_t, None, num_regions=64, return_trans_feat=True)
        lib.iq_set_tuning(5, 55)
        ref, tf_ref = eng.coalition_logits(data, center, rid_t, keep_t, None, num_regions=64, return_trans_feat=True)
        lib.iq_set_tuning(5, 58)     # bf16x3 layer 3 one n-tile per pass instead of two: the same products in the same order
        one, tf_one = eng.coalition_logits(data, center, rid_t, keep_t, None, num_regions=64, return_trans_feat=True)
    finally:
        lib.iq_set_tuning(5, 0)
    assert torch.equal(f32, ref) and torch.equal(tf_f32, tf_ref)
    assert torch.equal(got, one) and torch.equal(tf, tf_one)
    assert not torch.equal(got, f32)                                  # two different kernels did run
    assert (got - f32).abs().max().item() < 2e-6 * f32.abs().max().item()
    assert (tf - tf_f32).abs().max().item() < 2e-6 * tf_f32.abs().max().item()
    sel = [0, 57, 211, 398, 399]
    dense = hip_ops.mask_coalitions(data[0].contiguous(), rid_t[0].contiguous(), keep_t[sel].contiguous(), center.reshape(3).contiguous(),
                                    channel_first=True)
    assert torch.equal(model(dense)[0], got[sel])
