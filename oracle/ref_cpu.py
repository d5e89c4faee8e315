"""CPU ORACLE (test infrastructure, NOT product code).

A plain-PyTorch/NumPy restatement of the reference's Shapley / multi-order-interaction hot path
(SURVEY.md §8a).  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module; nothing under ``interpret_quality_amd/`` does.

Parity status: PINNED.  ``tests/golden/gen_golden.py`` imports the reference itself from
``/root/reference`` (never copied), runs it on the seeded inputs of
``interpret_quality_amd/synth.py`` and commits the outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks every function below against those vectors (bit-exact for
indices and masks, <=1e-6 relative for floating point; in practice the CPU results are identical
because the same ATen kernels run in the same order).

Each function cites the reference lines (relative to /root/reference) it restates.  The code is
deliberately written differently from the reference (vectorised masks, functional network with an
explicit state dict) - it restates behaviour, it does not copy text.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm1d default, models/pointnet.py:22-26


# --------------------------------------------------------------------------------------------
# geometry
# --------------------------------------------------------------------------------------------

def square_distance(src, dst):
    """tools/final_util.py:134-147 - expanded form, cancellation included:
    ``-2 src.dst^T`` then ``+= |src|^2`` then ``+= |dst|^2`` (that order)."""
    b, n, _ = src.shape
    m = dst.shape[1]
    d = torch.matmul(src, dst.transpose(1, 2)) * -2
    d = d + (src * src).sum(-1).reshape(b, n, 1)
    d = d + (dst * dst).sum(-1).reshape(b, 1, m)
    return d


def farthest_point_sample(xyz, npoint):
    """final_save_fps.py:10-31 (= models/pointnet2.py:45-68): start at index 0, running
    min-distance, argmax with first-index ties.  xyz (B,N,3) -> (B,npoint) int64."""
    b, n, _ = xyz.shape
    out = torch.zeros(b, npoint, dtype=torch.long)
    mind = torch.full((b, n), 1e10, dtype=xyz.dtype)
    cur = torch.zeros(b, dtype=torch.long)
    ar = torch.arange(b)
    for i in range(npoint):
        out[:, i] = cur
        c = xyz[ar, cur, :].reshape(b, 1, 3)
        d = ((xyz - c) ** 2).sum(-1)
        mind = torch.where(d < mind, d, mind)
        cur = torch.max(mind, -1)[1]
    return out


def farthest_point_sample_np(point, npoint):
    """final_data_shapley.py:71-92 - the NumPy sampler of the ShapeNet loader: start at index 0, squared
    distances in the dtype of the points (fp32 there), running minimum kept in fp64, first-index argmax."""
    xyz = np.asarray(point)[:, :3]
    out = np.zeros((npoint,), dtype=np.int64)
    mind = np.full((xyz.shape[0],), 1e10)
    cur = 0
    for i in range(npoint):
        out[i] = cur
        d = ((xyz - xyz[cur]) ** 2).sum(-1)
        closer = d < mind
        mind[closer] = d[closer]
        cur = int(np.argmax(mind))
    return out


def cal_region_id(data, fps_index):
    """final_shapley_value.py:20-35 - nearest FPS centre per point.  data (1,N,3) tensor,
    fps_index (R,) -> (N,) int64 ndarray."""
    idx = torch.as_tensor(fps_index, dtype=torch.long)
    centres = data[:, idx, :]
    return torch.argmin(square_distance(data, centres), dim=2).reshape(-1).numpy()


def rotate_xyz(x, angle_tuple):
    """final_rotate_center_enum_all.py:15-38 - R = Rx.Ry.Rz, returns x.R^T."""
    tx, ty, tz = angle_tuple[0], angle_tuple[1], angle_tuple[2]
    cx, cy, cz = torch.cos(tx), torch.cos(ty), torch.cos(tz)
    sx, sy, sz = torch.sin(tx), torch.sin(ty), torch.sin(tz)
    rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    r = torch.matmul(torch.matmul(rx, ry), rz)
    return torch.matmul(x, r.expand(x.shape[0], 3, 3).permute(0, 2, 1))


def translate_pc(data, trans):
    """final_trans_center_enum_all.py:13-21."""
    return data + trans


def scale_pc(data, scale):
    """final_scale_center_enum_all.py:14-22."""
    return data * scale


def generate_rotate_angle(angle_threshold=math.pi / 4, num=6):
    """final_rotate_center_enum_all.py:41-58 - 6^3 grid, 'ij' order, float32."""
    t = np.linspace(-angle_threshold, angle_threshold, num=num)
    gx, gy, gz = np.meshgrid(t, t, t, indexing="ij")
    g = np.stack([gx.reshape(-1), gy.reshape(-1), gz.reshape(-1)], axis=1)
    return torch.tensor(g, dtype=torch.float32)


def generate_trans_vector(threshold=0.5, num=6):
    """final_trans_center_enum_all.py:24-43 - grid clipped to norm <= threshold (in float32)."""
    t = np.linspace(-threshold, threshold, num=num)
    gx, gy, gz = np.meshgrid(t, t, t, indexing="ij")
    rows = []
    for v in zip(gx.reshape(-1), gy.reshape(-1), gz.reshape(-1)):
        tv = torch.tensor(list(v), dtype=torch.float32)
        if torch.norm(tv) > threshold:
            tv = tv / torch.norm(tv) * threshold
        rows.append(tv)
    return torch.stack(rows, dim=0)


def generate_scale(lower=0.5, upper=2.0, num=30):
    """final_scale_center_enum_all.py:25-31."""
    return torch.from_numpy(np.linspace(start=lower, stop=upper, num=num)).float()


# --------------------------------------------------------------------------------------------
# PointNet forward (models/pointnet.py)
# --------------------------------------------------------------------------------------------

def _bn(x, sd, name):
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"],
                        sd[name + ".weight"], sd[name + ".bias"], False, 0.0, BN_EPS)


def _conv(x, sd, name):
    return F.conv1d(x, sd[name + ".weight"], sd[name + ".bias"])


def _fc(x, sd, name):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def _stn(x, sd, p, k):
    """models/pointnet.py:29-47 - STNkd.forward."""
    h = F.relu(_bn(_conv(x, sd, p + ".conv1"), sd, p + ".bn1"))
    h = F.relu(_bn(_conv(h, sd, p + ".conv2"), sd, p + ".bn2"))
    h = F.relu(_bn(_conv(h, sd, p + ".conv3"), sd, p + ".bn3"))
    h = torch.max(h, 2, keepdim=True)[0].reshape(-1, 1024)
    h = F.relu(_bn(_fc(h, sd, p + ".fc1"), sd, p + ".bn4"))
    h = F.relu(_bn(_fc(h, sd, p + ".fc2"), sd, p + ".bn5"))
    h = _fc(h, sd, p + ".fc3")
    h = h + torch.eye(k, dtype=h.dtype, device=h.device).reshape(1, k * k)
    return h.reshape(-1, k, k)


def pointnet_forward(sd, x):
    """models/pointnet.py:64-89,109-115 - eval-mode PointNetCls.  x (B,3,N) float32 ->
    (logits (B,10), trans_feat (B,64,64), crt_points (B,1024) int64)."""
    trans = _stn(x, sd, "feat.stn", 3)
    h = torch.bmm(x.transpose(2, 1), trans).transpose(2, 1)
    h = F.relu(_bn(_conv(h, sd, "feat.conv1"), sd, "feat.bn1"))
    trans_feat = None
    if "feat.fstn.conv1.weight" in sd:   # feature_transform (models/pointnet.py:72-78); None without it
        trans_feat = _stn(h, sd, "feat.fstn", 64)
        h = torch.bmm(h.transpose(2, 1), trans_feat).transpose(2, 1)
    h = F.relu(_bn(_conv(h, sd, "feat.conv2"), sd, "feat.bn2"))
    h = _bn(_conv(h, sd, "feat.conv3"), sd, "feat.bn3")
    g, crt = torch.max(h, 2)
    g = g.reshape(-1, 1024)
    g = F.relu(_bn(_fc(g, sd, "fc1"), sd, "bn1"))
    g = F.relu(_bn(_fc(g, sd, "fc2"), sd, "bn2"))  # dropout is identity in eval mode
    return _fc(g, sd, "fc3"), trans_feat, crt


class PointNetOracle:
    """Callable with the reference module's call signature (returns the 3-tuple)."""

    def __init__(self, state_dict):
        self.sd = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v)))
                   for k, v in state_dict.items()}

    def __call__(self, x):
        with torch.no_grad():
            return pointnet_forward(self.sd, x)


# --------------------------------------------------------------------------------------------
# reward, masking, Shapley sampling (tools/final_common.py, final_shapley_value.py)
# --------------------------------------------------------------------------------------------

def get_reward(logits, lbl, softmax_type="modified"):
    """tools/final_common.py:11-24.  'modified': z_y - logsumexp(z_{!=y}); 'normal':
    log_softmax[y]."""
    y = int(lbl[0])
    if softmax_type == "normal":
        return F.log_softmax(logits, dim=1)[:, y]
    others = np.arange(logits.shape[1]) != y
    return logits[:, y] - torch.logsumexp(logits[:, others], dim=1)


def cal_reward(model, data, lbl, softmax_type="modified", is_pointnet=True):
    """tools/final_common.py:26-43.  data (B',N,3) -> (v (B',), logits (B',C))."""
    x = data.permute(0, 2, 1).contiguous()
    out = model(x)
    logits = out[0] if is_pointnet else out
    return get_reward(logits, lbl, softmax_type), logits


def shapley_masked_batch(data, center, orders, region_id):
    """tools/final_common.py:46-61 and :88-89 (= final_shapley_value.py:74-88,141-142 for one
    order).  Row ``o*(R+1)+i`` keeps the regions ``orders[o][:i]`` and collapses every other point
    onto ``center``.  Vectorised; the reference does it with R*bs index assignments."""
    orders = np.asarray(orders)
    bs, r = orders.shape
    n = data.shape[1]
    pos = np.empty((bs, r), dtype=np.int64)
    np.put_along_axis(pos, orders, np.broadcast_to(np.arange(r), (bs, r)), axis=1)
    pos_pt = pos[:, np.asarray(region_id)]                      # (bs, N) position of each point's region
    keep = pos_pt[:, None, :] < np.arange(r + 1)[None, :, None]  # (bs, R+1, N)
    keep = torch.from_numpy(keep.reshape(bs * (r + 1), n, 1))
    full = data.expand(bs * (r + 1), n, 3)
    return torch.where(keep, full, center.reshape(1, 1, 3).expand_as(full)).clone()


def cal_norm_factor(model, data, lbl, center, softmax_type="modified"):
    """final_shapley_value.py:39-56 - v(N) - v(empty) as a python float."""
    empty = center.reshape(1, 1, 3).expand(data.shape[0], data.shape[1], 3).clone()
    v_n, _ = cal_reward(model, data, lbl, softmax_type)
    v_0, _ = cal_reward(model, empty, lbl, softmax_type)
    return (v_n - v_0).item()


def generate_all_orders(num_samples_save, num_regions):
    """final_shapley_value.py:59-72 - consumes the GLOBAL NumPy RNG (seed it with np.random.seed
    first, as tools/final_util.py:113-120 does)."""
    rows = [np.random.permutation(np.arange(0, num_regions, 1)).reshape((1, -1))
            for _ in range(num_samples_save)]
    return np.concatenate(rows, axis=0)


def shap_sampling_all_regions_batch(model, data, lbl, region_id, orders, num_samples, bs,
                                    num_regions, softmax_type="modified"):
    """tools/final_common.py:64-103 - loop B body.  Returns (phi (R,) float64, logits
    (num_samples*(R+1), C) float32).  ``num_samples // bs`` batches (remainder dropped, :78), fp32
    ``dv`` added into a float64 accumulator in permutation order (:93-96), divided by num_samples
    (:97)."""
    center = torch.mean(data, dim=1).squeeze()
    phi = np.zeros((num_regions,))
    all_logits = []
    with torch.no_grad():
        for it in range(num_samples // bs):
            chunk = orders[it * bs:(it + 1) * bs]
            masked = shapley_masked_batch(data, center, chunk, region_id)
            v, logits = cal_reward(model, masked, lbl, softmax_type)
            all_logits.append(logits)
            v = v.reshape(bs, num_regions + 1)
            for o in range(bs):
                dv = v[o, 1:] - v[o, :-1]
                phi[chunk[o]] += dv.numpy()
    phi /= num_samples
    all_logits = torch.cat(all_logits, dim=0)
    assert all_logits.shape[0] == num_samples * (num_regions + 1)  # tools/final_common.py:99
    return phi, all_logits


def shap_sampling_stage1(model, data, lbl, region_id, orders, num_regions, softmax_type="modified"):
    """final_shapley_value.py:138-156 - loop A for one cloud: one order per forward; returns
    (running sum (R,) float64, region_sv_all (S,R) float64)."""
    center = torch.mean(data, dim=1).squeeze()
    total = np.zeros((num_regions,))
    rows = []
    with torch.no_grad():
        for order in orders:
            masked = shapley_masked_batch(data, center, order[None, :], region_id)
            v, _ = cal_reward(model, masked, lbl, softmax_type)
            dv = (v[1:] - v[:-1]).numpy()
            total[order] += dv
            row = np.zeros((num_regions,))
            row[order] += dv
            rows.append(row)
    return total, np.stack(rows, axis=0)


# --------------------------------------------------------------------------------------------
# multi-order interaction (final_point_binary_interaction_logits.py, final_cal_interactions.py)
# --------------------------------------------------------------------------------------------

def interaction_masked_batch(data_cf, center, region_id, region_i, region_j, contexts):
    """final_point_binary_interaction_logits.py:45-56.  data_cf (1,3,N); contexts (bs,m) int.
    Rows 4k..4k+3 keep S+{i,j}, S+{i}, S+{j}, S; masked entries are exactly ``x*0 + c``."""
    region_id = np.asarray(region_id)
    bs = contexts.shape[0]
    n = data_cf.shape[2]
    keep = np.zeros((4 * bs, n), dtype=bool)
    is_i = region_id == region_i
    is_j = region_id == region_j
    for k in range(bs):
        s = np.isin(region_id, contexts[k])
        keep[4 * k + 0] = s | is_i | is_j
        keep[4 * k + 1] = s | is_i
        keep[4 * k + 2] = s | is_j
        keep[4 * k + 3] = s
    mask = torch.from_numpy(keep).to(data_cf.dtype).reshape(4 * bs, 1, n).expand(4 * bs, 3, n)
    fill = center.reshape(1, 3, 1).expand(4 * bs, 3, n) * (1 - mask)
    return data_cf.expand(4 * bs, -1, -1) * mask + fill


def compute_order_interaction_logits(model, data, region_id, pairs, contexts, bs, is_pointnet=True):
    """final_point_binary_interaction_logits.py:15-70 - loop C.  data (1,N,3); pairs (P,2);
    contexts (P,C,m) -> logits (P,4C,num_class)."""
    num_context = contexts.shape[1]
    center = torch.mean(data, dim=1).squeeze()
    data_cf = data.permute(0, 2, 1)
    per_pair = []
    with torch.no_grad():
        for p, (ri, rj) in enumerate(pairs):
            chunks = []
            for it in range(math.ceil(num_context / bs)):
                ctx = contexts[p][it * bs:min(num_context, (it + 1) * bs)]
                masked = interaction_masked_batch(data_cf, center, region_id, ri, rj, ctx)
                out = model(masked)
                chunks.append(out[0] if is_pointnet else out)
            per_pair.append(torch.cat(chunks, dim=0).unsqueeze(0))
    return torch.cat(per_pair, dim=0)


def compute_order_interaction(all_logits, lbl, softmax_type="modified"):
    """final_cal_interactions.py:14-37 - ((v0 + v3) - v1) - v2 in float32, widened to float64 by
    ``.item()``.  (P,4C,K) -> (P,C) float64 ndarray."""
    p = all_logits.shape[0]
    c = all_logits.shape[1] // 4
    out = np.zeros((p, c))
    for i in range(p):
        v = get_reward(all_logits[i], lbl, softmax_type).reshape(c, 4)
        out[i] = (v[:, 0] + v[:, 3] - v[:, 1] - v[:, 2]).double().numpy()
    return out


# --------------------------------------------------------------------------------------------
# PointNet++ MSG (models/pointnet2.py)
# --------------------------------------------------------------------------------------------

def index_points(points, idx):
    """models/pointnet2.py:27-43 - batched gather.  points (B,N,C), idx (B,...) -> (B,...,C)."""
    b = points.shape[0]
    batch = torch.arange(b, device=points.device).reshape([b] + [1] * (idx.dim() - 1)).expand_as(idx)
    return points[batch, idx, :]


def query_ball_point(radius, nsample, xyz, new_xyz):
    """models/pointnet2.py:70-91 - the K LOWEST-INDEX points with d^2 <= r^2 (expanded-form
    distance), padded with the first hit.  -> (B,S,K) int64."""
    b, n, _ = xyz.shape
    s = new_xyz.shape[1]
    idx = torch.arange(n, dtype=torch.long).reshape(1, 1, n).repeat(b, s, 1)
    d = square_distance(new_xyz, xyz)
    idx[d > radius ** 2] = n
    idx = idx.sort(dim=-1)[0][:, :, :nsample]
    first = idx[:, :, 0].reshape(b, s, 1).repeat(1, 1, nsample)
    pad = idx == n
    idx[pad] = first[pad]
    return idx


def _conv2d_bn_relu(x, sd, conv, bn):
    y = F.conv2d(x, sd[conv + ".weight"], sd[conv + ".bias"])
    y = F.batch_norm(y, sd[bn + ".running_mean"], sd[bn + ".running_var"], sd[bn + ".weight"], sd[bn + ".bias"],
                     False, 0.0, BN_EPS)
    return F.relu(y)


def set_abstraction_msg(sd, name, npoint, radius_list, nsample_list, n_layers, xyz_cf, points_cf, return_aux=False):
    """models/pointnet2.py:201-240.  xyz_cf (B,3,N), points_cf (B,D,N) or None ->
    (new_xyz (B,3,S), new_points (B,D',S)).  Features first, relative xyz last (:226)."""
    xyz = xyz_cf.permute(0, 2, 1)
    points = points_cf.permute(0, 2, 1) if points_cf is not None else None
    b, n, c = xyz.shape
    fps = farthest_point_sample(xyz, npoint)
    new_xyz = index_points(xyz, fps)
    outs, aux = [], {"fps": fps, "group_idx": []}
    for i, radius in enumerate(radius_list):
        k = nsample_list[i]
        gidx = query_ball_point(radius, k, xyz, new_xyz)
        aux["group_idx"].append(gidx)
        g_xyz = index_points(xyz, gidx) - new_xyz.reshape(b, npoint, 1, c)
        g = torch.cat([index_points(points, gidx), g_xyz], dim=-1) if points is not None else g_xyz
        g = g.permute(0, 3, 2, 1)  # (B, D, K, S)
        for j in range(n_layers[i]):
            g = _conv2d_bn_relu(g, sd, "%s.conv_blocks.%d.%d" % (name, i, j), "%s.bn_blocks.%d.%d" % (name, i, j))
        outs.append(torch.max(g, 2)[0])
    res = (new_xyz.permute(0, 2, 1), torch.cat(outs, dim=1))
    return res + (aux,) if return_aux else res


def set_abstraction_all(sd, name, n_layers, xyz_cf, points_cf):
    """models/pointnet2.py:153-178 with group_all: [xyz, features] (xyz FIRST, absolute, :132-135)."""
    xyz = xyz_cf.permute(0, 2, 1)
    b, n, c = xyz.shape
    g = torch.cat([xyz.reshape(b, 1, n, c), points_cf.permute(0, 2, 1).reshape(b, 1, n, -1)], dim=-1)
    g = g.permute(0, 3, 2, 1)
    for j in range(n_layers):
        g = _conv2d_bn_relu(g, sd, "%s.mlp_convs.%d" % (name, j), "%s.mlp_bns.%d" % (name, j))
    return torch.max(g, 2)[0]


def pointnet2_forward(sd, xyz_cf, return_aux=False):
    """models/pointnet2.py:264-276 - eval-mode PointNet2ClsMsg.  (B,3,N) -> logits (B,10)."""
    b = xyz_cf.shape[0]
    r1 = set_abstraction_msg(sd, "sa1", 512, [0.1, 0.2, 0.4], [16, 32, 128], [3, 3, 3], xyz_cf, None, return_aux)
    r2 = set_abstraction_msg(sd, "sa2", 128, [0.2, 0.4, 0.8], [32, 64, 128], [3, 3, 3], r1[0], r1[1], return_aux)
    l3 = set_abstraction_all(sd, "sa3", 3, r2[0], r2[1])
    x = l3.reshape(b, 1024)
    x = F.relu(_bn(_fc(x, sd, "fc1"), sd, "bn1"))
    x = F.relu(_bn(_fc(x, sd, "fc2"), sd, "bn2"))
    logits = _fc(x, sd, "fc3")
    if return_aux:
        return logits, {"sa1": r1[2], "sa2": r2[2], "l1_xyz": r1[0], "l1_points": r1[1], "l2_xyz": r2[0], "l2_points": r2[1]}
    return logits


class PointNet2Oracle:
    def __init__(self, state_dict):
        self.sd = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v)))
                   for k, v in state_dict.items()}

    def __call__(self, x):
        with torch.no_grad():
            return pointnet2_forward(self.sd, x)


# --------------------------------------------------------------------------------------------
# DGCNN / GCNN (models/dgcnn.py)
# --------------------------------------------------------------------------------------------

def knn(x, k):
    """models/dgcnn.py:12-18.  x (B,C,N) -> (B,N,k) indices of the k largest of
    -|x_i|^2 - (-2 x_i.x_j) - |x_j|^2 (self included; expanded form, d(i,i) is not exactly 0)."""
    inner = torch.matmul(x.transpose(2, 1), x) * -2
    xx = torch.sum(x ** 2, dim=1, keepdim=True)
    pairwise = -xx - inner - xx.transpose(2, 1)
    return pairwise.topk(k=k, dim=-1)[1]


def get_graph_feature(x, k=20, idx=None):
    """models/dgcnn.py:21-47.  x (B,C,N) -> [x_j - x_i ; x_i] (B,2C,N,k)."""
    b, c, n = x.shape
    if idx is None:
        idx = knn(x, k)
    xt = x.transpose(2, 1).contiguous()                                   # (B,N,C)
    flat = (idx + torch.arange(b).reshape(-1, 1, 1) * n).reshape(-1)
    nbr = xt.reshape(b * n, c)[flat].reshape(b, n, k, c)
    ctr = xt.reshape(b, n, 1, c).expand(b, n, k, c)
    return torch.cat((nbr - ctr, ctr), dim=3).permute(0, 3, 1, 2)


def _edge_conv(x, sd, j, k, idx=None):
    f = get_graph_feature(x, k, idx)
    y = F.conv2d(f, sd["conv%d.0.weight" % j])
    y = F.batch_norm(y, sd["bn%d.running_mean" % j], sd["bn%d.running_var" % j], sd["bn%d.weight" % j],
                     sd["bn%d.bias" % j], False, 0.0, BN_EPS)
    return F.leaky_relu(y, 0.2).max(dim=-1)[0]


def dgcnn_forward(sd, x, k=20, fixed_graph=False, return_aux=False):
    """models/dgcnn.py:83-120 (DGCNN_cls) / :156-194 (GCNN_cls, fixed_graph=True: one xyz graph, :163)."""
    b = x.shape[0]
    idx = knn(x, k) if fixed_graph else None
    x1 = _edge_conv(x, sd, 1, k, idx)
    x2 = _edge_conv(x1, sd, 2, k, idx)
    x3 = _edge_conv(x2, sd, 3, k, idx)
    x4 = _edge_conv(x3, sd, 4, k, idx)
    h = torch.cat((x1, x2, x3, x4), dim=1)
    h = F.conv1d(h, sd["conv5.0.weight"])
    h = F.leaky_relu(F.batch_norm(h, sd["bn5.running_mean"], sd["bn5.running_var"], sd["bn5.weight"], sd["bn5.bias"],
                                  False, 0.0, BN_EPS), 0.2)
    g = torch.cat((F.adaptive_max_pool1d(h, 1).reshape(b, -1), F.adaptive_avg_pool1d(h, 1).reshape(b, -1)), 1)
    g = F.leaky_relu(_bn(F.linear(g, sd["linear1.weight"]), sd, "bn6"), 0.2)
    g = F.leaky_relu(_bn(_fc(g, sd, "linear2"), sd, "bn7"), 0.2)
    logits = _fc(g, sd, "linear3")
    if return_aux:
        return logits, {"x1": x1, "x2": x2, "x3": x3, "x4": x4}
    return logits


class DgcnnOracle:
    def __init__(self, state_dict, fixed_graph=False, k=20):
        self.sd = {kk: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))) for kk, v in state_dict.items()}
        self.fixed_graph, self.k = fixed_graph, k

    def __call__(self, x):
        with torch.no_grad():
            return dgcnn_forward(self.sd, x, self.k, self.fixed_graph)


# --------------------------------------------------------------------------------------------
# PointConv (models/pointconv.py)
# --------------------------------------------------------------------------------------------

def knn_point(nsample, xyz, new_xyz):
    """models/pointconv.py:103-114 - the nsample smallest expanded-form distances, unsorted."""
    return torch.topk(square_distance(new_xyz, xyz), nsample, dim=-1, largest=False, sorted=False)[1]


def compute_density(xyz, bandwidth):
    """models/pointconv.py:199-209 - Gaussian KDE over all points of the cloud."""
    d = square_distance(xyz, xyz)
    return (torch.exp(-d / (2.0 * bandwidth * bandwidth)) / (2.5 * bandwidth)).mean(dim=-1)


def _scalar_net(x, sd, prefix):
    """DensityNet / WeightNet (models/pointconv.py:212-265): three 1x1 conv + BN + ReLU layers (the sigmoid branch
    of DensityNet is unreachable: `i == len(self.mlp_convs)` is never true, :228-231)."""
    for j in range(3):
        x = _conv2d_bn_relu(x, sd, "%s.mlp_convs.%d" % (prefix, j), "%s.mlp_bns.%d" % (prefix, j))
    return x


def pointconv_sa(sd, name, npoint, nsample, bandwidth, group_all, xyz_cf, points_cf, return_aux=False):
    """models/pointconv.py:324-391 - PointConvDensitySetAbstraction.forward."""
    xyz = xyz_cf.permute(0, 2, 1)
    points = points_cf.permute(0, 2, 1) if points_cf is not None else None
    b, n, _ = xyz.shape
    inv_density = 1.0 / compute_density(xyz, bandwidth)
    aux = {}
    if group_all:
        new_xyz = xyz.mean(dim=1, keepdim=True)
        g_xyz = xyz.reshape(b, 1, n, 3) - new_xyz.reshape(b, 1, 1, 3)
        new_points = torch.cat([g_xyz, points.reshape(b, 1, n, -1)], dim=-1) if points is not None else g_xyz
        g_density = inv_density.reshape(b, 1, n, 1)
        s = 1
    else:
        fps = farthest_point_sample(xyz, npoint)
        new_xyz = index_points(xyz, fps)
        idx = knn_point(nsample, xyz, new_xyz)
        aux = {"fps": fps, "knn": idx}
        g_xyz = index_points(xyz, idx) - new_xyz.reshape(b, npoint, 1, 3)
        new_points = torch.cat([g_xyz, index_points(points, idx)], dim=-1) if points is not None else g_xyz  # xyz FIRST
        g_density = index_points(inv_density.reshape(b, n, 1), idx)
        s = npoint
    h = new_points.permute(0, 3, 2, 1)  # (B, C, K, S)
    for j in range(3):
        h = _conv2d_bn_relu(h, sd, "%s.mlp_convs.%d" % (name, j), "%s.mlp_bns.%d" % (name, j))
    scale = g_density / g_density.max(dim=2, keepdim=True)[0]
    h = h * _scalar_net(scale.permute(0, 3, 2, 1), sd, name + ".densitynet")
    w = _scalar_net(g_xyz.permute(0, 3, 2, 1), sd, name + ".weightnet")
    out = torch.matmul(h.permute(0, 3, 1, 2), w.permute(0, 3, 2, 1)).reshape(b, s, -1)
    out = F.linear(out, sd[name + ".linear.weight"], sd[name + ".linear.bias"])
    out = F.relu(_bn(out.permute(0, 2, 1), sd, name + ".bn_linear"))
    res = (new_xyz.permute(0, 2, 1), out)
    return res + (aux,) if return_aux else res


def pointconv_forward(sd, xyz_cf, return_aux=False):
    """models/pointconv.py:414-424 - eval-mode PointConvDensityClsSsg.  (B,3,N) -> logits (B,10)."""
    b = xyz_cf.shape[0]
    r1 = pointconv_sa(sd, "sa1", 512, 32, 0.1, False, xyz_cf, None, return_aux)
    r2 = pointconv_sa(sd, "sa2", 128, 64, 0.2, False, r1[0], r1[1], return_aux)
    r3 = pointconv_sa(sd, "sa3", 1, None, 0.4, True, r2[0], r2[1], return_aux)
    x = r3[1].reshape(b, 1024)
    x = F.relu(_bn(_fc(x, sd, "fc1"), sd, "bn1"))
    x = F.relu(_bn(_fc(x, sd, "fc2"), sd, "bn2"))
    logits = _fc(x, sd, "fc3")
    if return_aux:
        return logits, {"sa1": r1[2], "sa2": r2[2], "l1_points": r1[1], "l2_points": r2[1], "l1_xyz": r1[0]}
    return logits


class PointConvOracle:
    def __init__(self, state_dict):
        self.sd = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))) for k, v in state_dict.items()}

    def __call__(self, x):
        with torch.no_grad():
            return pointconv_forward(self.sd, x)


# --------------------------------------------------------------------------------------------
# Smoothness enumeration (final_smoothness_center_enum_all.py) - stage 5 of exp_shapley.sh
# --------------------------------------------------------------------------------------------
SMOOTH_DEFAULTS = dict(step=1e-3, enum_step=0.05, epoch=50, var_threshold=0.003, dist_threshold=0.03,
                       stop_ratio=0.5, max_iteration=100)  # :13-19


def principal_orientations(pts):
    """:23-45 - eigenvectors of the unbiased covariance, largest eigenvalue first.  torch.symeig (removed from
    current torch) returned ascending eigenvalues exactly like torch.linalg.eigh."""
    s = pts.shape[0]
    d = (pts - pts.mean(dim=0)).unsqueeze(2)
    cov = torch.bmm(d, d.transpose(1, 2)).sum(dim=0) / (s - 1)
    _, vec = torch.linalg.eigh(cov)
    return vec[:, 2].clone(), vec[:, 1].clone(), vec[:, 0].clone()


def projected_variances(pts, orient):
    """:48-63 - unbiased variance of the projections on o1, o2, o3."""
    return [torch.var(torch.matmul(pts, o)) for o in orient]


def _smoothness_value(mode, var):
    """:85-99 + :211-228 - variances sorted by value (np.argsort), then the ratio of the mode."""
    order = np.argsort(np.array([v.item() for v in var])).tolist()
    s_min, s_mid, s_max = var[order[0]], var[order[1]], var[order[2]]
    if mode == "linearity":
        return (s_max - s_mid) / s_max, (s_max, s_mid)
    if mode == "planarity":
        return (s_mid - s_min) / s_max, (s_max, s_mid, s_min)
    return s_min / s_max, (s_max, s_min)


def smoothness_enumerate(data, region_id, num_regions, mode, objective, start=None, **overrides):
    """test_all_region's epoch loop (:303-335) without the Shapley evaluations: returns
    (data_list (P,1,N,3), smoothness_list (P,R), orig (R,4) = var1..3 and smoothness of the untouched regions)."""
    prm = dict(SMOOTH_DEFAULTS, **overrides)
    region_id = np.asarray(region_id)
    cur = (data if start is None else start).clone().detach()  # start: resume from a deformed cloud (test hook)
    info = []
    for r in range(num_regions):  # get_original_region_info (:245-267)
        org = data[:, region_id == r, :].squeeze().clone().detach()
        orient = principal_orientations(org)
        var0 = projected_variances(org, orient)
        with torch.no_grad():
            sm0 = _smoothness_value(mode, var0)[0].item()
        info.append(dict(org=org, orient=orient, ub=[v + prm["var_threshold"] for v in var0],
                         lb=[v - prm["var_threshold"] for v in var0], smooth=sm0, var0=[v.item() for v in var0], sm0=sm0))
    live = [True] * num_regions
    data_list, smooth_list = [], []
    for _ in range(prm["epoch"]):
        row = []
        for r in range(num_regions):
            reg = info[r]
            if live[r]:  # update_region (:183-242)
                sel = region_id == r
                smooth = reg["smooth"]
                target = smooth + prm["enum_step"] if objective == "inc" else smooth - prm["enum_step"]
                it = 0
                while (smooth < target) if objective == "inc" else (smooth > target):
                    x = cur[:, sel, :].squeeze().clone().detach().requires_grad_(True)
                    var = projected_variances(x, reg["orient"])
                    var = [v.detach() if (v > ub or v < lb) else v for v, ub, lb in zip(var, reg["ub"], reg["lb"])]
                    val, deps = _smoothness_value(mode, var)
                    smooth = val.item()
                    if any(d.requires_grad for d in deps):
                        val.backward()
                    grad_none = x.grad is None
                    if not grad_none:  # gradient_descent (:121-138)
                        g = x.grad.data
                        norm = torch.norm(g)
                        delta = prm["step"] * g / norm if norm != 0 else 1e-8
                        x.data = x.data + delta if objective == "inc" else x.data - delta
                    # apply_distance_bound (:102-118).  NB the assignment below is the reference's: `.data =` on the
                    # temporary view x[i] does not reach x, so a point beyond the bound is counted but not moved.
                    with torch.no_grad():
                        diff = x - reg["org"]
                        dist = torch.norm(diff, dim=1)
                        count = 0
                        for i in range(dist.shape[0]):
                            if dist[i] > prm["dist_threshold"]:
                                count += 1
                                x[i].data = reg["org"][i].data + prm["dist_threshold"] * diff[i] / dist[i]
                    cur[:, sel, :] = x.unsqueeze(0).data
                    it += 1
                    if count / x.shape[0] > prm["stop_ratio"] or grad_none or it > prm["max_iteration"]:
                        live[r] = False
                        break
                reg["smooth"] = smooth
            row.append(reg["smooth"])
        smooth_list.append(row)
        data_list.append(cur.numpy().copy())
        if not any(live):
            break
    orig = np.array([reg["var0"] + [reg["sm0"]] for reg in info])
    return np.array(data_list), np.array(smooth_list), orig


# --------------------------------------------------------------------------------------------
# Consumers of the artefacts (final_result.py) - only what the format-compatibility test needs
# --------------------------------------------------------------------------------------------

def consumer_sensitivity(base_folder, mode):
    """final_result.py:83-104 (cal_sensitivity): per-region range of the Shapley values over the enumeration, normalised
    by the mean L1 norm; reads ``<mode>_all/region_shapley_value.npy`` or, for the smoothness modes, the inc / dec pair."""
    if mode in ("linearity", "planarity", "scattering"):
        vals = np.concatenate((np.load(base_folder + "%s_all/allregion_inc/region_shapley_value.npy" % mode),
                               np.load(base_folder + "%s_all/allregion_dec/region_shapley_value.npy" % mode)), axis=0)
    else:
        vals = np.load(base_folder + "%s_all/region_shapley_value.npy" % mode)
    return (vals.max(axis=0) - vals.min(axis=0)) / np.mean(np.sum(np.abs(vals), axis=1))


def consumer_mean_sv_intensity(base_folder, mode):
    """final_result.py:62-80 for one cloud: E_pose |phi_region|."""
    return np.mean(np.abs(np.load(base_folder + "%s_all/region_shapley_value.npy" % mode)), axis=0)
