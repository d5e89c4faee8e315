"""Golden vectors for the dataset loaders from the REFERENCE (imported from /root/reference, never copied).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/gen_golden_loaders.py

Builds the miniature dataset tree of interpret_quality_amd.synth.write_dataset_tree in a temp directory, runs the
reference's ModelNet_Loader_Shapley_test / ShapeNetDataset_Shapley_test / get_folder_name_list on it (CPU) and
stores the items they return -> tests/golden/loaders.npz."""
import argparse
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refenv  # noqa: E402

synth = _refenv.setup()   # the reference first on sys.path, the repository root (its `tools/` shims) off it

import final_data_shapley as ref_data  # noqa: E402
from tools import final_util as ref_util  # noqa: E402


def main():
    out = {}
    with tempfile.TemporaryDirectory() as root:
        synth.write_dataset_tree(root)
        os.chdir(root)
        # the reference resolves data/modelnet10_numpy next to its own file: point that at the temp tree
        ref_file = ref_data.__file__
        ref_data.__file__ = os.path.join(root, "final_data_shapley.py")
        args = argparse.Namespace(dataset="modelnet10")
        ds = ref_data.ModelNet_Loader_Shapley_test(args, partition="train", num_points=1024)
        out["modelnet_names"] = np.array(ref_util.get_folder_name_list(args))
        for i in range(len(ds)):
            pts, cls = ds[i]
            out["modelnet_%d_points" % i] = pts
            out["modelnet_%d_label" % i] = np.int64(cls)
        ref_data.__file__ = ref_file     # misc/num_seg_classes.txt is read next to the reference's file
        args = argparse.Namespace(dataset="shapenet")
        ds = ref_data.ShapeNetDataset_Shapley_test(args, split="train", npoints=1024,
                                                   class_choice=ref_util.SHAPENET_CLASS, classification=True)
        out["shapenet_names"] = np.array(ref_util.get_folder_name_list(args))
        out["shapenet_len"] = np.int64(len(ds))
        for i in range(len(ds)):
            pts, cls = ds[i]
            out["shapenet_%d_points" % i] = pts.numpy()
            out["shapenet_%d_label" % i] = cls.numpy()
        raw = synth.raw_scan(10, 2607).astype(np.float32)
        out["fps_np_2607_to_64"] = ref_data.farthest_point_sample_np(raw, 64)
    np.savez_compressed(os.path.join(HERE, "loaders.npz"), **out)
    print({k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main()
