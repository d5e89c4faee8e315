"""The 30-cloud Shapley test loaders (reference final_data_shapley.py), same class names, constructor
arguments and item format, reading the same on-disk trees (``data/modelnet10_numpy``,
``data/shapenetcore_partanno_segmentation_benchmark_v0``, ``misc/`` lists) relative to the working directory.

The one piece of device work is the ShapeNet down-sampling: the reference picks ``npoints`` of each raw scan
with a NumPy farthest-point sampler (final_data_shapley.py:71-92, start index 0, fp32 squared distances,
first-index arg-max).  Here it is the same ``iq_fps`` kernel the rest of the path uses -
no CPU sampler is kept, so the ShapeNet loader needs the GPU like everything else.
"""
import json
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from . import _lib, hip_ops
from .final_util import DATA_MODELNET_SHAPLEY_TEST, DATA_SHAPENET_SHAPLEY_TEST

MAX_SCAN_POINTS = 8192  # iq_fps keeps the running distances of one cloud in registers/LDS


def _read_lines(path):
    with open(path) as f:
        return [line.rstrip() for line in f.readlines()]


def make_dataset_modelnet10(mode, opt):
    """final_data_shapley.py:10-42: [(path to <folder>/<name>.npy, label)], label = row of the folder in
    modelnet10_shape_names.txt; sample names are ``<folder>_<4 digits>``."""
    data_dir = os.path.join(getattr(opt, "data_root", os.getcwd()), "data", "modelnet10_numpy")
    shape_list = _read_lines(os.path.join(data_dir, "modelnet10_shape_names.txt"))
    if mode != "train":
        raise Exception("Network mode error.")
    dataset = []
    for name in _read_lines(os.path.join("misc", DATA_MODELNET_SHAPLEY_TEST)):
        folder = name[0:-5]
        dataset.append((os.path.join(data_dir, folder, name + ".npy"), shape_list.index(folder)))
    return dataset


class ModelNet_Loader_Shapley_test(Dataset):
    """final_data_shapley.py:47-69: first ``num_points`` rows, xyz columns, float32, no augmentation."""

    def __init__(self, opt, num_points, partition="train"):
        super().__init__()
        self.opt, self.partition, self.num_points = opt, partition, num_points
        self.dataset = make_dataset_modelnet10(self.partition, opt)

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, index):
        pc_np_file, class_id = self.dataset[index]
        data = np.load(pc_np_file)
        return data[0:self.num_points, 0:3].astype(np.float32), class_id


def farthest_point_sample_np(point, npoint, device=None):
    """final_data_shapley.py:71-92 on the HIP path: point (N,D) ndarray -> (npoint,) int64 indices."""
    xyz = np.ascontiguousarray(np.asarray(point)[:, :3], dtype=np.float32)
    if xyz.shape[0] > MAX_SCAN_POINTS:
        raise _lib.IqError("scan of %d points exceeds the %d supported by iq_fps" % (xyz.shape[0], MAX_SCAN_POINTS))
    device = torch.device(device if device is not None else "cuda:0")
    idx = hip_ops.fps(torch.from_numpy(xyz).unsqueeze(0).to(device), int(npoint))
    return idx[0].cpu().numpy().astype(np.int64)


def _synset_table(root, class_choice=None):
    """synsetoffset2category.txt -> [(category name, synset id)] in file order, restricted to ``class_choice``."""
    rows = [line.split()[:2] for line in _read_lines(os.path.join(root, "synsetoffset2category.txt")) if line.strip()]
    return [(name, synset) for name, synset in rows if class_choice is None or name in class_choice]


def _selected_scans(root, table):
    """The scans of misc/shapenet_train_selected.json as (category name, .pts path, .seg path), ordered by category in
    the order of ``table`` and, inside a category, in list order - the item order of final_data_shapley.py:130-139."""
    rank = {synset: k for k, (_, synset) in enumerate(table)}
    with open(os.path.join("misc", DATA_SHAPENET_SHAPLEY_TEST)) as f:
        entries = [e.split("/")[1:3] for e in json.load(f)]
    entries = [(synset, uuid) for synset, uuid in entries if synset in rank]
    entries.sort(key=lambda e: rank[e[0]])                                   # stable: list order inside a category
    return [(table[rank[synset]][0], os.path.join(root, synset, "points", uuid + ".pts"),
             os.path.join(root, synset, "points_label", uuid + ".seg")) for synset, uuid in entries]


def normalise_scan(points):
    """Centre on the mean and scale to max-norm 1 in float32 (final_data_shapley.py:155-157)."""
    points = points - points.mean(axis=0, keepdims=True)
    return points / np.sqrt((points ** 2).sum(axis=1)).max()


class ShapeNetDataset_Shapley_test(Dataset):
    """final_data_shapley.py:95-179.  Items: (point_set (npoints,3) f32 tensor, cls 0-d int64 tensor) or, with
    classification=False, (point_set, seg (npoints,) int64).  Public attributes as in the reference: ``cat``
    (name -> synset), ``id2cat``, ``datapath`` [(name, pts, seg)], ``classes`` (name -> label by sorted name),
    ``seg_classes`` / ``num_seg_classes`` (when misc/num_seg_classes.txt exists)."""

    def __init__(self, opt, root="./data/shapenetcore_partanno_segmentation_benchmark_v0", npoints=2500,
                 classification=True, class_choice=None, split="train"):
        self.npoints, self.opt, self.root, self.split, self.classification = npoints, opt, root, split, classification
        table = _synset_table(root, class_choice)
        self.cat = dict(table)
        self.id2cat = {synset: name for name, synset in table}
        self.datapath = _selected_scans(root, table)
        self.meta = {name: [(p, g) for n, p, g in self.datapath if n == name] for name, _ in table}
        self.classes = {name: label for label, name in enumerate(sorted(self.cat))}
        seg_file = os.path.join("misc", "num_seg_classes.txt")
        self.seg_classes = ({ls[0]: int(ls[1]) for ls in (line.split() for line in _read_lines(seg_file)) if len(ls) >= 2}
                            if os.path.exists(seg_file) else {})
        self.num_seg_classes = self.seg_classes.get(table[0][0]) if table else None

    def __len__(self):
        return len(self.datapath)

    def __getitem__(self, index):
        name, pts_file, seg_file = self.datapath[index]
        scan = normalise_scan(np.loadtxt(pts_file).astype(np.float32))
        choice = farthest_point_sample_np(scan, self.npoints, getattr(self.opt, "device", None))
        point_set = torch.from_numpy(scan[choice].astype(np.float32))
        if self.classification:
            return point_set, torch.from_numpy(np.array(self.classes[name]).astype(np.int64))
        return point_set, torch.from_numpy(np.loadtxt(seg_file).astype(np.int64)[choice])


def batches_of_one(dataset):
    """``DataLoader(dataset, batch_size=1, shuffle=False)`` without worker processes (the ShapeNet items touch
    the GPU, which must not be shared with forked workers): yields (data (1,N,3) f32, lbl (1,) int64)."""
    for i in range(len(dataset)):
        pts, cls = dataset[i]
        pts = pts if isinstance(pts, torch.Tensor) else torch.from_numpy(np.asarray(pts))
        yield pts.unsqueeze(0), torch.as_tensor(cls, dtype=torch.long).reshape(1)


def shapley_test_loader(args):
    """The loader construction shared by every driver (final_shapley_value.py:160-170 and clones)."""
    from .final_util import SHAPENET_CLASS
    if args.dataset == "modelnet10":
        return batches_of_one(ModelNet_Loader_Shapley_test(args, partition="train", num_points=args.num_points))
    if args.dataset == "shapenet":
        return batches_of_one(ShapeNetDataset_Shapley_test(args, split="train", npoints=args.num_points,
                                                           class_choice=SHAPENET_CLASS, classification=True))
    raise Exception("Dataset does not exist")
