"""Golden vectors for the smoothness enumeration from the REFERENCE (imported from /root/reference, never copied).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/gen_golden_smoothness.py

The reference calls torch.symeig (final_smoothness_center_enum_all.py:41), which current torch no longer has; the
call is routed to torch.linalg.eigh, its documented successor with the same ascending-eigenvalue convention.  The
Shapley evaluations inside test_all_region are not what this fixture pins (tests/golden/pointnet_shapley_*.npz do),
so the model is a stub returning constant logits; the fixture holds the clouds and smoothness values of every epoch
-> tests/golden/smoothness.npz."""
import argparse
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refenv  # noqa: E402

synth = _refenv.setup()   # the reference first on sys.path, the repository root (its `tools/` shims) off it

# torch 2.x keeps the name only as a stub that raises "removed"
torch.symeig = lambda a, eigenvectors=True: torch.linalg.eigh(a)  # noqa: E731

import final_smoothness_center_enum_all as ref  # noqa: E402
import final_shapley_value as ref_stage1  # noqa: E402
import final_save_fps as ref_fps  # noqa: E402


class ConstantLogits(torch.nn.Module):
    def forward(self, x):
        return torch.zeros(x.shape[0], 10), None, None


def main():
    out = {}
    pts, label = synth.make_cloud(2)
    data = torch.from_numpy(pts).unsqueeze(0)
    lbl = torch.tensor([label])
    num_regions = 32
    fps_index = ref_fps.farthest_point_sample(data, num_regions)[0]
    region_id = ref_stage1.cal_region_id(data, fps_index, None, save=False)
    out["region_id"] = np.asarray(region_id)
    orders = np.stack([np.arange(num_regions)])
    for mode in ("linearity", "planarity", "scattering"):
        for objective in ("inc", "dec"):
            args = argparse.Namespace(model="pointnet", softmax_type="modified", num_points=1024, num_regions=num_regions,
                                      num_samples=1, shapley_batch_size=1, step=ref.STEP, enum_step=ref.ENUM_STEP,
                                      epoch=ref.EPOCH, var_threshold=ref.VAR_THRESHOLD, dist_threshold=ref.DIST_THRESHOLD,
                                      stop_ratio=ref.STOP_RATIO, max_iteration=ref.MAX_ITERATION, mode=mode)
            key = "%s_%s" % (mode, objective)
            # On the CPU the reference's data_list aliases data_copy (ndarray views of one tensor, :326), so every saved
            # epoch shows the final cloud; the state after epoch k is taken from a run limited to k epochs instead.
            for epochs in (1, 2, ref.EPOCH):
                args.epoch = epochs
                with tempfile.TemporaryDirectory() as tmp:
                    ref.test_all_region(ConstantLogits(), data, lbl, orders, region_id, tmp + "/", args, objective)
                    res = tmp + "/allregion_%s/" % objective
                    d = np.load(res + "data_smoothness.npy")
                    s = np.load(res + "%s.npy" % mode)
                tag = "full" if epochs == ref.EPOCH else "after%d" % epochs
                out["%s_%s_data" % (key, tag)] = d[-1, 0]
                if epochs == ref.EPOCH:
                    out[key + "_smoothness"] = s
                    out[key + "_epochs"] = np.int64(d.shape[0])
            print(key, d.shape, s.shape)
    np.savez_compressed(os.path.join(HERE, "smoothness.npz"), **out)


if __name__ == "__main__":
    main()
