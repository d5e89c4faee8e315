"""DGCNN parity AT SCALE, from the REFERENCE (imported from /root/reference, never copied): logits of 2016 interaction
coalitions of one cloud (21 random region pairs x 4 context sizes x 6 contexts x 4 masked clouds), the reference's CPU
float32 path.  DGCNN rebuilds its kNN graph in feature space at every layer, so a coalition that sits on a near-tie can
flip a neighbour under any rounding change; this fixture is what the rate of such coalitions is measured on
(tests/test_dgcnn_gpu.py::test_dgcnn_parity_rate_at_scale).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/gen_golden_dgcnn_scale.py [dgcnn|gcnn]   (about 10 minutes, 8 cores)

Output tests/golden/dgcnn_scale.npz (default) or gcnn_scale.npz: the same inputs through GCNN_cls (one fixed xyz graph; the
control for the near-tie attribution and the second model of the context-averaged interaction check).  Data only."""
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

from models import dgcnn as ref_dg  # noqa: E402
from tools import final_util as ref_util  # noqa: E402
import final_shapley_value as ref_stage1  # noqa: E402
import final_save_fps as ref_fps  # noqa: E402
import final_gen_pair as ref_pair  # noqa: E402
import final_point_binary_interaction_logits as ref_inter  # noqa: E402


def main():
    torch.set_num_threads(8)
    name = sys.argv[1] if len(sys.argv) > 1 else "dgcnn"
    cls = {"dgcnn": ref_dg.DGCNN_cls, "gcnn": ref_dg.GCNN_cls}[name]
    model = cls(argparse.Namespace(dataset="modelnet10", k=20))
    model.load_state_dict(synth.to_torch(synth.dgcnn_state_dict(0)))
    model.eval()
    cloud_id, num_regions = 5, 32
    pts, label = synth.make_cloud(cloud_id)
    data = torch.from_numpy(pts).unsqueeze(0)
    fps_index = ref_fps.farthest_point_sample(data, num_regions)[0]
    region_id = ref_stage1.cal_region_id(data, fps_index, None, save=False)
    ratios = [0.07, 0.3, 0.6, 0.9]            # m = 2, 9, 18, 27 context regions
    args = argparse.Namespace(model=name, softmax_type="modified", num_regions=num_regions, num_pairs_random=21,
                              num_save_context_max=6, ratio=ratios, interaction_batch_size=6)
    ref_util.set_random(7)
    pairs = ref_pair.gen_pair_random(args)
    out = {"cloud_id": cloud_id, "label": label, "region_id": region_id.astype(np.int8), "pairs": pairs.astype(np.int8),
           "ratios": np.array(ratios)}
    # the reference itself in float64 on the same inputs: how far ITS float32 result is from exact arithmetic shows which
    # coalitions sit on a kNN near-tie (their graphs flip under any change of rounding), i.e. the reference's own conditioning
    m64 = cls(argparse.Namespace(dataset="modelnet10", k=20)).double()
    m64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in synth.to_torch(synth.dgcnn_state_dict(0)).items()})
    m64.eval()
    n = 0
    with tempfile.TemporaryDirectory() as td, torch.no_grad():
        ref_pair.gen_context(pairs, td + "/", args)
        for ratio in ratios:
            tag = "ratio%d" % int(ratio * 100)
            ctx = np.load(td + "/%s_context_list.npy" % tag)
            logits = ref_inter.compute_order_interaction_logits(model, data, region_id, pairs, ctx, args)
            out[tag + "_contexts"] = ctx.astype(np.int8)
            out[tag + "_logits"] = logits.numpy()
            l64 = ref_inter.compute_order_interaction_logits(m64, data.double(), region_id, pairs, ctx, args)
            out[tag + "_logits_fp64"] = l64.numpy().astype(np.float32)      # its deviation from the float32 run is >= 1e-5 where it matters
            n += logits.shape[0] * logits.shape[1]
            print(tag, ctx.shape, tuple(logits.shape), flush=True)
    print("coalitions:", n)
    np.savez_compressed(os.path.join(HERE, "%s_scale.npz" % name), **out)


if __name__ == "__main__":
    main()
