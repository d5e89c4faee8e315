"""Golden vectors for DGCNN / GCNN from the REFERENCE (imported from /root/reference, never copied).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/gen_golden_dgcnn.py

Outputs tests/golden/dgcnn.npz: kNN indices (xyz and 64-d feature space), layer-1 features, logits of
raw and interaction-masked clouds for both models, interaction logits/values for a small case."""
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
import final_cal_interactions as ref_cal  # noqa: E402


def main():
    torch.set_num_threads(8)
    a = argparse.Namespace(dataset="modelnet10", k=20)
    sd = synth.to_torch(synth.dgcnn_state_dict(0))
    models = {}
    for name, cls in (("dgcnn", ref_dg.DGCNN_cls), ("gcnn", ref_dg.GCNN_cls)):
        m = cls(a)
        m.load_state_dict(sd)
        models[name] = m.eval()
    out = {}
    pts, label = synth.make_cloud(0)
    data = torch.from_numpy(pts).unsqueeze(0)
    lbl = torch.tensor([label])
    num_regions = 32
    fps_index = ref_fps.farthest_point_sample(data, num_regions)[0]
    region_id = ref_stage1.cal_region_id(data, fps_index, None, save=False)
    ratios = [0.0, 0.5, 1.0]
    args = argparse.Namespace(model="dgcnn", softmax_type="modified", num_regions=num_regions, num_pairs_random=2,
                              num_save_context_max=3, ratio=ratios, interaction_batch_size=2)
    ref_util.set_random(1)
    pairs = ref_pair.gen_pair_random(args)
    out.update(region_id=region_id, pairs=pairs, ratios=np.array(ratios))
    with tempfile.TemporaryDirectory() as td, torch.no_grad():
        ref_pair.gen_context(pairs, td + "/", args)
        for ratio in ratios:
            tag = "ratio%d" % int(ratio * 100)
            ctx = np.load(td + "/%s_context_list.npy" % tag)
            out[tag + "_contexts"] = ctx
            for name in ("dgcnn", "gcnn"):
                args.model = name
                logits = ref_inter.compute_order_interaction_logits(models[name], data, region_id, pairs, ctx, args)
                out["%s_%s_logits" % (tag, name)] = logits.numpy()
                if name == "dgcnn":
                    # the reference in float64 on the same inputs: how far its own float32 result is from
                    # exact arithmetic tells which clouds sit on a kNN near-tie (dynamic graphs flip there)
                    m64 = ref_dg.DGCNN_cls(a).double()
                    m64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()})
                    l64 = ref_inter.compute_order_interaction_logits(m64.eval(), data.double(), region_id, pairs, ctx, args)
                    out["%s_dgcnn_logits_fp64" % tag] = l64.numpy()
                out["%s_%s_interaction" % (tag, name)] = ref_cal.compute_order_interaction(logits, lbl, args)
        # op level on a half-masked cloud and on a raw cloud
        center = torch.mean(data, dim=1).squeeze()
        half = data.clone()
        half[0, region_id >= 16, :] = center
        x = torch.cat([data, half], dim=0).permute(0, 2, 1).contiguous()
        out["knn_xyz"] = ref_dg.knn(x, 20).numpy().astype(np.int16)
        f = ref_dg.get_graph_feature(x, 20)
        x1 = models["dgcnn"].conv1(f).max(dim=-1)[0]
        out["x1_first8"] = x1[:, :8, :].numpy()
        out["knn_feat64"] = ref_dg.knn(x1, 20).numpy().astype(np.int16)
        for name in ("dgcnn", "gcnn"):
            out["raw_logits_" + name] = models[name](x).numpy()
    np.savez_compressed(os.path.join(HERE, "dgcnn.npz"), **out)
    for k, v in out.items():
        if "interaction" in k:
            print(k, np.abs(v).max())
    print(out["raw_logits_dgcnn"].round(3))


if __name__ == "__main__":
    main()
