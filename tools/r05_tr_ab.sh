# Round 5: layer 2 of the grouped bf16x3 kernels with TRANSPOSED tiles (default) against the untransposed form (tuning key 7 = 1)
R=$GRAFT_REPO_ROOT; cd $R
python3 - <<'PY'
import numpy as np, torch
from interpret_quality_amd import _lib, hip_ops, synth
from interpret_quality_amd.pointnet2 import PointNet2ClsMsg
from interpret_quality_amd.pointconv import PointConvDensityClsSsg
lib = _lib.load(); d = torch.device("cuda:0")
rng = np.random.default_rng(29)
clouds = torch.stack([torch.from_numpy(synth.make_cloud(i)[0]) for i in (0, 3, 6)]).to(d)
rid = torch.stack([hip_ops.region_assign(clouds[c].contiguous(), hip_ops.fps(clouds[c:c + 1], 32)[0].contiguous()) for c in range(3)])
keep = [(1 << 32) - 1, 1, 1 << 31, 3, 0xffff] + [int(x) for x in rng.integers(0, 1 << 32, size=85, dtype=np.uint64)]
keep += [int(x) & int(y) & int(z) | 1 for x, y, z in rng.integers(0, 1 << 32, size=(30, 3), dtype=np.uint64)]
kt = hip_ops.masks_to_tensor(keep, d); co = torch.tensor([i % 3 for i in range(len(keep))], dtype=torch.int32, device=d)
for name, cls, sd in (("pointnet2", PointNet2ClsMsg, synth.pointnet2_state_dict), ("pointconv", PointConvDensityClsSsg, synth.pointconv_state_dict)):
    m = cls(None); m.load_state_dict(synth.to_torch(sd(0))); m = m.to(d).eval()
    res = {}
    for t in (0, 1):
        lib.iq_set_tuning(7, t)
        res[t] = (m.coalition_logits(clouds, clouds.mean(dim=1), rid, kt, co, num_regions=32).clone(), m.forward_points(clouds).clone())
    lib.iq_set_tuning(7, 0)
    e = [(res[0][i] - res[1][i]).abs().max().item() / res[1][i].abs().max().item() for i in (0, 1)]
    print(name, "transposed == untransposed bit for bit:", torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), "max rel diff", e)
PY
for m in pointnet2 pointconv; do
  for rep in 1 2 3; do
    for t in 1 0; do
      echo "$m 7=$t: $(python3 tools/bench_models.py --model $m --mode shapley --steps 8 --tune 7=$t 2>&1 | tail -1 | cut -c1-190)"
    done
  done
done
