"""Host-side mirror of tools/final_util.py: constants, argument helpers, model loading.

Only what the Shapley / interaction path reads is mirrored: training helpers
(cal_loss, rot_angle_axis) are out of scope.
"""
import importlib
import json
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

from ._lib import IqError

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# tools/final_util.py:15-19
NUM_POINTS = 1024
NUM_REGIONS = 32
NUM_SAMPLES_SAVE = 1000
NUM_SAMPLES = 100
K_FOR_DGCNN = 20
# tools/final_util.py:22-27
DATA_MODELNET_SHAPLEY_TEST = "modelnet10_train_final30.txt"
DATA_SHAPENET_SHAPLEY_TEST = "shapenet_train_selected.json"
MODELNET_INTER_SELECTED_SAMPLE = [0, 3, 6, 9, 12, 15, 18, 21, 24, 27]
SHAPENET_INTER_SELECTED_SAMPLE = [0, 3, 6, 9, 12, 15, 19, 21, 24, 27]
BALL_QUERY_COEF = 0.25

MODELS = ("pointnet", "pointnet2", "pointconv", "dgcnn", "gcnn", "gcnn_adv")
DATASETS = ("modelnet10", "shapenet")


def model_path(dataset, model):
    """Checkpoint locations of tools/final_util.py:52-66."""
    if model == "gcnn_adv":
        return "checkpoints/exp_MODEL_gcnn_adv_DATA_%s_POINTNUM_1024_clean_with_all_rot_da/models/model_399.t7" % dataset
    return "checkpoints/exp_MODEL_%s_DATA_%s_POINTNUM_1024_clean/models/model_best.t7" % (model, dataset)


# the reference's module-level names (tools/final_util.py:52-66)
for _ds in ("modelnet10", "shapenet"):
    for _m in ("pointnet", "pointnet2", "pointconv", "dgcnn", "gcnn", "gcnn_adv"):
        globals()["MODEL_PATH_%s_%s" % ("MODELNET" if _ds == "modelnet10" else "SHAPENET", _m.upper())] = model_path(_ds, _m)
del _ds, _m


def square_distance_np(x):
    """tools/final_util.py:122-132 (host, NumPy; used on the 32 region centres)."""
    xx = np.sum(x ** 2, axis=1, keepdims=True)
    return xx + xx.T - 2 * np.matmul(x, x.T)


def square_distance(src, dst):
    """tools/final_util.py:134-147: (B,N,3), (B,M,3) -> (B,N,M) expanded-form squared distances (torch; the region
    assignment itself runs in iq_region_assign with the same expression)."""
    b, n, _ = src.shape
    m = dst.shape[1]
    dist = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    dist += torch.sum(src ** 2, -1).view(b, n, 1)
    dist += torch.sum(dst ** 2, -1).view(b, 1, m)
    return dist


def ball_query(x, r):
    """tools/final_util.py:150-160: boolean neighbour matrix of the region centres."""
    return square_distance_np(x) < r ** 2


def exp_folder(args):
    """Artefact root shared by every stage (final_shapley_value.py:194-195)."""
    return "./checkpoints/exp_MODEL_%s_DATA_%s_POINTNUM_%d_REGIONNUM_%d_shapley_test/" % (
        args.model, args.dataset, args.num_points, args.num_regions)


def _config():
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    return importlib.import_module("config").CONFIG


def mkdir(path):
    """tools/final_util.py:85-87; race-free when every rank of a multi-GPU run creates the same folder."""
    os.makedirs(path, exist_ok=True)


class IOStream:
    """tools/final_util.py:90-100 - print and append to a log file."""

    def __init__(self, path):
        self.f = open(path, "a")

    def cprint(self, text):
        print(text)
        self.f.write(text + "\n")
        self.f.flush()

    def close(self):
        self.f.close()


def cal_rank(values):
    return np.argsort(np.argsort(values))


def set_random(seed):
    """tools/final_util.py:113-120."""
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)


def set_model_args(args):
    """tools/final_util.py:162-204."""
    if args.dataset not in DATASETS:
        raise Exception("Dataset does not exist")
    if args.model not in MODELS:
        raise Exception("Model not implemented")
    if args.model in ("dgcnn", "gcnn", "gcnn_adv"):
        args.k = K_FOR_DGCNN
    if args.model == "pointnet":
        args.feature_transform = True
    args.model_path = model_path(args.dataset, args.model)


def _batch_key(model):
    return "gcnn" if model == "gcnn_adv" else model  # tools/final_util.py:214,228


def strict_batch_cap():
    """config.py's additive "strict_batch_cap" key or IQ_STRICT_BATCH=1: the batch-size knobs cap every launch (reference
    semantics, config.py:2-17) instead of being a floor."""
    return bool(_config().get("strict_batch_cap", False)) or os.environ.get("IQ_STRICT_BATCH") == "1"


def set_shapley_batch_size(args):
    """tools/final_util.py:207-219."""
    table = _config()["shapley_batch_size"]
    if _batch_key(args.model) not in table:
        raise Exception("Not implemented")
    args.shapley_batch_size = table[_batch_key(args.model)]
    args.strict_batch_cap = strict_batch_cap()


def set_interaction_batch_size(args):
    """tools/final_util.py:221-233."""
    table = _config()["interaction_batch_size"]
    if _batch_key(args.model) not in table:
        raise Exception("Not implemented")
    args.interaction_batch_size = table[_batch_key(args.model)]
    args.strict_batch_cap = strict_batch_cap()


def load_model(args):
    """tools/final_util.py:236-262: build the module, load the `.t7` state dict (stripping the
    DataParallel ``module.`` prefix), eval mode.  ``args.synthetic`` (additive flag of the drop-in
    scripts) substitutes the deterministic synthetic weights when no checkpoint exists offline."""
    from . import synth
    cache = getattr(args, "model_cache", None)   # additive: the sweep driver builds each (model, dataset) once per process
    key = (args.model, args.dataset, args.model_path, bool(getattr(args, "synthetic", False)))
    if cache is not None and key in cache:
        return cache[key]
    if args.model == "pointnet":
        from .pointnet import PointNetCls
        model, synth_sd = PointNetCls(args).to(args.device), synth.pointnet_state_dict
    elif args.model == "pointnet2":
        from .pointnet2 import PointNet2ClsMsg
        model, synth_sd = PointNet2ClsMsg(args).to(args.device), synth.pointnet2_state_dict
    elif args.model in ("dgcnn", "gcnn", "gcnn_adv"):
        from .dgcnn import DGCNN_cls, GCNN_cls
        model = (DGCNN_cls if args.model == "dgcnn" else GCNN_cls)(args).to(args.device)
        synth_sd = synth.dgcnn_state_dict
    elif args.model == "pointconv":
        from .pointconv import PointConvDensityClsSsg
        model, synth_sd = PointConvDensityClsSsg(args).to(args.device), synth.pointconv_state_dict
    else:
        raise IqError("model %r is unknown" % args.model)
    if getattr(args, "synthetic", False) and not os.path.exists(args.model_path):
        state_dict = synth.to_torch(synth_sd(0))
    else:
        state_dict = torch.load(args.model_path, map_location=args.device)
    new_state_dict = OrderedDict()
    for k, v in state_dict.items():
        new_state_dict[k[len("module."):] if "module." in k else k] = v
    model.load_state_dict(new_state_dict)
    model = model.eval()
    if cache is not None:
        cache[key] = model
    return model


def get_folder_name_list(args):
    """tools/final_util.py:265-283: names of the 30 clouds.  Falls back to synthetic names when the
    misc/ lists (dataset metadata, not shipped) are absent and args.synthetic is set."""
    misc = os.path.join("misc", DATA_MODELNET_SHAPLEY_TEST if args.dataset == "modelnet10" else DATA_SHAPENET_SHAPLEY_TEST)
    if not os.path.exists(misc):
        if getattr(args, "synthetic", False):
            return ["synthetic_%02d" % i for i in range(getattr(args, "num_clouds", 30))]
        raise FileNotFoundError(misc)
    if args.dataset == "modelnet10":
        with open(misc) as f:
            return [line.rstrip() for line in f.readlines()]
    names = []
    for file in json.load(open(misc)):
        _, category, uuid = file.split("/")
        names.append(SHAPENET_ID2CAT[category] + "_" + uuid)
    return names


SHAPENET_CLASS = ["Bag", "Cap", "Earphone", "Knife", "Laptop", "Motorbike", "Mug", "Pistol", "Rocket", "Skateboard"]
SHAPENET_ID2CAT = {
    "02691156": "Airplane", "02773838": "Bag", "02954340": "Cap", "02958343": "Car", "03001627": "Chair",
    "03261776": "Earphone", "03467517": "Guitar", "03624134": "Knife", "03636649": "Lamp", "03642806": "Laptop",
    "03790512": "Motorbike", "03797390": "Mug", "03948459": "Pistol", "04099429": "Rocket",
    "04225987": "Skateboard", "04379243": "Table",
}
SHAPENET_CAT2ID = {v: k for k, v in SHAPENET_ID2CAT.items()}


def synthetic_loader(args):
    """Stand-in for final_data_shapley.py's 30-cloud loaders (datasets are not available offline):
    yields (data (1,N,3) float32, lbl (1,) int64) exactly like ``DataLoader(batch_size=1)``."""
    from . import synth
    for i in range(getattr(args, "num_clouds", 30)):
        pts, label = synth.make_cloud(i, args.num_points)
        yield torch.from_numpy(pts).unsqueeze(0), torch.tensor([label], dtype=torch.long)
