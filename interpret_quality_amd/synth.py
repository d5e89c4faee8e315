"""Deterministic synthetic inputs: point clouds, FPS-free helpers and reference-format weights.

No datasets or checkpoints exist offline (SURVEY.md §8c limits), so every test, the golden
generator and bench.py draw their inputs from here.  Recipes follow SURVEY.md §8(d):

* cloud ``i``: ``np.random.default_rng(1000 + i)``, 1024 points uniform in the unit ball, then
  centred and scaled to max-norm 1 the way the reference's ShapeNet loader does
  (final_data_shapley.py:155-157), float32; label ``i % 10``.
* weights: one PCG64 stream per parameter, keyed by ``crc32(parameter name)`` so that a tensor's
  values never depend on which other tensors exist.  Keys and shapes are those of the reference's
  ``state_dict()`` (models/pointnet.py:11-115), i.e. what tools/final_util.py:236-262 loads.
"""
import zlib

import numpy as np

NUM_CLASSES = 10


def make_cloud(i, num_points=1024):
    """Synthetic cloud ``i`` -> ((num_points, 3) float32, int label)."""
    rng = np.random.default_rng(1000 + i)
    direction = rng.standard_normal((num_points, 3))
    direction /= np.linalg.norm(direction, axis=1, keepdims=True)
    radius = rng.random((num_points, 1)) ** (1.0 / 3.0)
    pts = (direction * radius).astype(np.float32)
    pts = pts - np.expand_dims(np.mean(pts, axis=0), 0)
    dist = np.max(np.sqrt(np.sum(pts ** 2, axis=1)), 0)
    pts = pts / dist
    return pts.astype(np.float32), i % NUM_CLASSES


def make_orders(num_samples, num_regions, seed=1):
    """Permutations exactly as the reference draws them (final_shapley_value.py:59-72):
    global NumPy RNG seeded by ``set_random`` (tools/final_util.py:113-120), one
    ``np.random.permutation`` per sample."""
    np.random.seed(seed)
    rows = [np.random.permutation(np.arange(0, num_regions, 1)) for _ in range(num_samples)]
    return np.stack(rows, axis=0).astype(np.int64)


# --------------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------------

def _rng_for(name, seed):
    return np.random.Generator(np.random.PCG64([zlib.crc32(name.encode()), seed]))


def _linear(sd, name, cout, cin, seed, conv, gain, bias_std=0.05):
    w = _rng_for(name + ".weight", seed).standard_normal((cout, cin)) * (gain / np.sqrt(cin))
    b = _rng_for(name + ".bias", seed).standard_normal((cout,)) * bias_std
    sd[name + ".weight"] = w.astype(np.float32).reshape((cout, cin, 1) if conv else (cout, cin))
    sd[name + ".bias"] = b.astype(np.float32)


def _bn(sd, name, c, seed):
    sd[name + ".weight"] = _rng_for(name + ".weight", seed).uniform(0.8, 1.2, (c,)).astype(np.float32)
    sd[name + ".bias"] = (_rng_for(name + ".bias", seed).standard_normal((c,)) * 0.1).astype(np.float32)
    sd[name + ".running_mean"] = (_rng_for(name + ".running_mean", seed).standard_normal((c,)) * 0.1).astype(np.float32)
    sd[name + ".running_var"] = _rng_for(name + ".running_var", seed).uniform(0.5, 1.5, (c,)).astype(np.float32)
    sd[name + ".num_batches_tracked"] = np.array(100, dtype=np.int64)


def _stn(sd, prefix, k, seed):
    g = np.sqrt(2.0)
    _linear(sd, prefix + ".conv1", 64, k, seed, True, g)
    _linear(sd, prefix + ".conv2", 128, 64, seed, True, g)
    _linear(sd, prefix + ".conv3", 1024, 128, seed, True, g)
    _linear(sd, prefix + ".fc1", 512, 1024, seed, False, g)
    _linear(sd, prefix + ".fc2", 256, 512, seed, False, g)
    # the regressed transform is identity + a moderate perturbation, as in a trained net
    _linear(sd, prefix + ".fc3", k * k, 256, seed, False, 0.3 / np.sqrt(k), bias_std=0.02)
    for j, c in enumerate((64, 128, 1024, 512, 256), start=1):
        _bn(sd, "%s.bn%d" % (prefix, j), c, seed)


def pointnet_state_dict(seed=0, feature_transform=True):
    """Reference-keyed PointNetCls state dict (111 tensors; 74 without the feature STN, models/pointnet.py:62-63) as numpy arrays."""
    if not feature_transform:
        return {k: v for k, v in pointnet_state_dict(seed).items() if not k.startswith("feat.fstn.")}
    sd = {}
    g = np.sqrt(2.0)
    _stn(sd, "feat.stn", 3, seed)
    _linear(sd, "feat.conv1", 64, 3, seed, True, g)
    _linear(sd, "feat.conv2", 128, 64, seed, True, g)
    _linear(sd, "feat.conv3", 1024, 128, seed, True, g)
    for j, c in enumerate((64, 128, 1024), start=1):
        _bn(sd, "feat.bn%d" % j, c, seed)
    _stn(sd, "feat.fstn", 64, seed)
    _linear(sd, "fc1", 512, 1024, seed, False, g)
    _linear(sd, "fc2", 256, 512, seed, False, g)
    _linear(sd, "fc3", NUM_CLASSES, 256, seed, False, 2.0)
    _bn(sd, "bn1", 512, seed)
    _bn(sd, "bn2", 256, seed)
    return sd


def to_torch(sd):
    import torch
    return {k: (torch.tensor(v) if np.ndim(v) == 0 else torch.from_numpy(np.ascontiguousarray(v))) for k, v in sd.items()}


# --------------------------------------------------------------------------------------------
# PointNet++ MSG (models/pointnet2.py:244-276): 163 tensors
# --------------------------------------------------------------------------------------------
PN2_SA1 = dict(npoint=512, radius=[0.1, 0.2, 0.4], nsample=[16, 32, 128], in_channel=0,
               mlp=[[32, 32, 64], [64, 64, 128], [64, 96, 128]])
PN2_SA2 = dict(npoint=128, radius=[0.2, 0.4, 0.8], nsample=[32, 64, 128], in_channel=320,
               mlp=[[64, 64, 128], [128, 128, 256], [128, 128, 256]])
PN2_SA3_MLP = [256, 512, 1024]


def _conv2d(sd, name, cout, cin, seed, gain):
    w = _rng_for(name + ".weight", seed).standard_normal((cout, cin)) * (gain / np.sqrt(cin))
    b = _rng_for(name + ".bias", seed).standard_normal((cout,)) * 0.05
    sd[name + ".weight"] = w.astype(np.float32).reshape(cout, cin, 1, 1)
    sd[name + ".bias"] = b.astype(np.float32)


def pointnet2_state_dict(seed=0):
    """Reference-keyed PointNet2ClsMsg state dict as numpy arrays."""
    sd = {}
    g = np.sqrt(2.0)
    for sa, cfg in (("sa1", PN2_SA1), ("sa2", PN2_SA2)):
        for i, mlp in enumerate(cfg["mlp"]):
            last = cfg["in_channel"] + 3
            for j, c in enumerate(mlp):
                # relative coordinates are small (radius-scaled): larger first-layer gain keeps signal alive
                _conv2d(sd, "%s.conv_blocks.%d.%d" % (sa, i, j), c, last, seed, g * (4.0 if (sa == "sa1" and j == 0) else 1.0))
                _bn(sd, "%s.bn_blocks.%d.%d" % (sa, i, j), c, seed)
                last = c
    last = 640 + 3
    for j, c in enumerate(PN2_SA3_MLP):
        _conv2d(sd, "sa3.mlp_convs.%d" % j, c, last, seed, g)
        _bn(sd, "sa3.mlp_bns.%d" % j, c, seed)
        last = c
    _linear(sd, "fc1", 512, 1024, seed, False, g)
    _bn(sd, "bn1", 512, seed)
    _linear(sd, "fc2", 256, 512, seed, False, g)
    _bn(sd, "bn2", 256, seed)
    _linear(sd, "fc3", NUM_CLASSES, 256, seed, False, 0.7)
    return sd


# --------------------------------------------------------------------------------------------
# DGCNN / GCNN (models/dgcnn.py:51-194): 70 tensors, BatchNorms registered twice (bnK.* and convK.1.*)
# --------------------------------------------------------------------------------------------
DGCNN_CONVS = [(6, 64), (128, 64), (128, 128), (256, 256)]


def dgcnn_state_dict(seed=0):
    """Reference-keyed DGCNN_cls / GCNN_cls state dict (same layout for both) as numpy arrays."""
    sd = {}
    g = np.sqrt(2.0)
    chans = [c for _, c in DGCNN_CONVS] + [1024, 512, 256]
    for j, c in enumerate(chans, start=1):
        _bn(sd, "bn%d" % j, c, seed)
    for j, (cin, cout) in enumerate(DGCNN_CONVS, start=1):
        w = _rng_for("conv%d.0.weight" % j, seed).standard_normal((cout, cin)) * (g / np.sqrt(cin))
        sd["conv%d.0.weight" % j] = w.astype(np.float32).reshape(cout, cin, 1, 1)
    w = _rng_for("conv5.0.weight", seed).standard_normal((1024, 512)) * (g / np.sqrt(512))
    sd["conv5.0.weight"] = w.astype(np.float32).reshape(1024, 512, 1)
    for j in range(1, 6):  # the Sequential's BN is the same module object as self.bnJ
        for f in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            sd["conv%d.1.%s" % (j, f)] = sd["bn%d.%s" % (j, f)]
    sd["linear1.weight"] = (_rng_for("linear1.weight", seed).standard_normal((512, 2048)) * (g / np.sqrt(2048))).astype(np.float32)
    _linear(sd, "linear2", 256, 512, seed, False, g)
    _linear(sd, "linear3", NUM_CLASSES, 256, seed, False, 1.0)
    return sd


# --------------------------------------------------------------------------------------------
# PointConv (models/pointconv.py:394-424): 226 tensors
# --------------------------------------------------------------------------------------------
PC_SA = [dict(npoint=512, nsample=32, in_channel=3, mlp=[64, 64, 128], bandwidth=0.1),
         dict(npoint=128, nsample=64, in_channel=128 + 3, mlp=[128, 128, 256], bandwidth=0.2),
         dict(npoint=1, nsample=None, in_channel=256 + 3, mlp=[256, 512, 1024], bandwidth=0.4)]


PC_LINEAR_SCALE = (0.0244, 0.00617, 0.00377)


def _bn_pos(sd, name, c, seed):
    """BatchNorm with a positive shift, for the tiny scalar nets (keeps ReLU outputs alive)."""
    _bn(sd, name, c, seed)
    sd[name + ".bias"] = (np.abs(sd[name + ".bias"]) + 0.3).astype(np.float32)


def pointconv_state_dict(seed=0):
    """Reference-keyed PointConvDensityClsSsg state dict as numpy arrays."""
    sd = {}
    g = np.sqrt(2.0)
    for k, cfg in enumerate(PC_SA, start=1):
        p = "sa%d" % k
        last = cfg["in_channel"]
        for j, c in enumerate(cfg["mlp"]):
            _conv2d(sd, "%s.mlp_convs.%d" % (p, j), c, last, seed, g * (3.0 if (k == 1 and j == 0) else 1.0))
            _bn(sd, "%s.mlp_bns.%d" % (p, j), c, seed)
            last = c
        for net, dims in (("weightnet", [3, 8, 8, 16]), ("densitynet", [1, 16, 8, 1])):
            for j in range(3):
                name = "%s.%s.mlp_convs.%d" % (p, net, j)
                _conv2d(sd, name, dims[j + 1], dims[j], seed, g * (3.0 if (net == "weightnet" and j == 0) else 1.0))
                if net == "densitynet" or j > 0:  # non-negative weights on non-negative inputs: the scalar nets stay alive
                    sd[name + ".weight"] = np.abs(sd[name + ".weight"])
                _bn_pos(sd, "%s.%s.mlp_bns.%d" % (p, net, j), dims[j + 1], seed)
        # the weighted sum over the K members is not normalised: keep the following linear layer's output O(1)
        _linear(sd, p + ".linear", cfg["mlp"][-1], 16 * cfg["mlp"][-1], seed, False, g / (cfg["nsample"] or 128))
        for f in ("weight", "bias"):  # empirical: brings the stage output to O(1) for the synthetic clouds
            sd["%s.linear.%s" % (p, f)] = (sd["%s.linear.%s" % (p, f)] * PC_LINEAR_SCALE[k - 1]).astype(np.float32)
        _bn(sd, p + ".bn_linear", cfg["mlp"][-1], seed)
    _linear(sd, "fc1", 512, 1024, seed, False, g)
    _bn(sd, "bn1", 512, seed)
    _linear(sd, "fc2", 256, 512, seed, False, g)
    _bn(sd, "bn2", 256, seed)
    _linear(sd, "fc3", NUM_CLASSES, 256, seed, False, 1.0)
    return sd


# ------------------------------------------------------------------------------------------------
# a miniature dataset tree in the reference's on-disk formats (final_data_shapley.py), for loader tests
# ------------------------------------------------------------------------------------------------
DATASET_TREE = {
    "modelnet10": [("bathtub", "bathtub_0003", 1100), ("chair", "chair_0007", 2048)],
    "shapenet": [("02773838", "aaaa0001", 2607), ("03797390", "bbbb0002", 1024), ("04099429", "cccc0003", 2890)],
}


def raw_scan(seed, n):
    """An un-normalised, off-centre 'scan': anisotropic blob + a denser lump (so FPS has something to do)."""
    rng = np.random.default_rng(7000 + seed)
    body = rng.normal(size=(n - n // 4, 3)) * np.array([0.31, 0.12, 0.22])
    lump = rng.normal(size=(n // 4, 3)) * 0.03 + np.array([0.2, -0.05, 0.1])
    return (np.concatenate([body, lump]) + np.array([0.4, -0.2, 0.1])).astype(np.float64)


def write_dataset_tree(root):
    """Creates <root>/data/modelnet10_numpy, <root>/data/shapenetcore_partanno_segmentation_benchmark_v0 and
    <root>/misc with the tiny lists above; returns the raw arrays keyed by sample name."""
    import json
    import os
    raw = {}
    mn = os.path.join(root, "data", "modelnet10_numpy")
    os.makedirs(mn, exist_ok=True)
    names = ["bathtub", "bed", "chair", "desk", "dresser", "monitor", "night_stand", "sofa", "table", "toilet"]
    with open(os.path.join(mn, "modelnet10_shape_names.txt"), "w") as f:
        f.write("\n".join(names) + "\n")
    os.makedirs(os.path.join(root, "misc"), exist_ok=True)
    with open(os.path.join(root, "misc", "modelnet10_train_final30.txt"), "w") as f:
        for i, (folder, name, n) in enumerate(DATASET_TREE["modelnet10"]):
            os.makedirs(os.path.join(mn, folder), exist_ok=True)
            arr = np.concatenate([raw_scan(i, n), np.ones((n, 3))], axis=1).astype(np.float32)  # xyz + normals
            np.save(os.path.join(mn, folder, name + ".npy"), arr)
            raw[name] = arr
            f.write(name + "\n")
    sn = os.path.join(root, "data", "shapenetcore_partanno_segmentation_benchmark_v0")
    os.makedirs(sn, exist_ok=True)
    cats = {"Airplane": "02691156", "Bag": "02773838", "Cap": "02954340", "Car": "02958343", "Chair": "03001627",
            "Earphone": "03261776", "Guitar": "03467517", "Knife": "03624134", "Lamp": "03636649", "Laptop": "03642806",
            "Motorbike": "03790512", "Mug": "03797390", "Pistol": "03948459", "Rocket": "04099429",
            "Skateboard": "04225987", "Table": "04379243"}
    with open(os.path.join(sn, "synsetoffset2category.txt"), "w") as f:
        for k, v in cats.items():
            f.write("%s\t%s\n" % (k, v))
    files = []
    for i, (cat, uuid, n) in enumerate(DATASET_TREE["shapenet"]):
        os.makedirs(os.path.join(sn, cat, "points"), exist_ok=True)
        os.makedirs(os.path.join(sn, cat, "points_label"), exist_ok=True)
        arr = raw_scan(10 + i, n)
        np.savetxt(os.path.join(sn, cat, "points", uuid + ".pts"), arr, fmt="%.5f")
        np.savetxt(os.path.join(sn, cat, "points_label", uuid + ".seg"), np.ones(n, dtype=np.int64), fmt="%d")
        raw[uuid] = arr
        files.append("shape_data/%s/%s" % (cat, uuid))
    with open(os.path.join(root, "misc", "shapenet_train_selected.json"), "w") as f:
        json.dump(files, f)
    return raw
