# (A/B of a CLOSED experiment: apply tools/experiments/r05_grouped_pingpong.patch first; see tools/experiments/README.md)
# Round 5: ping-pong halves (+ A-term prefetch in the MFMA loops) against two independent 256-thread workgroups per CU
# (tuning key 7 = 1), grouped bf16x3 kernels.  Bitwise check first (coalition logits of 200 random coalitions, both schedules).
R=$GRAFT_REPO_ROOT; cd $R
python3 - <<'PY'
import numpy as np, torch
from interpret_quality_amd import _lib, hip_ops, synth
from interpret_quality_amd.pointnet2 import PointNet2ClsMsg
lib = _lib.load(); d = torch.device("cuda:0")
m = PointNet2ClsMsg(None); m.load_state_dict(synth.to_torch(synth.pointnet2_state_dict(0))); m = m.to(d).eval()
rng = np.random.default_rng(29)
clouds = torch.stack([torch.from_numpy(synth.make_cloud(i)[0]) for i in (0, 3, 6)]).to(d)
rid = torch.stack([hip_ops.region_assign(clouds[c].contiguous(), hip_ops.fps(clouds[c:c + 1], 32)[0].contiguous()) for c in range(3)])
keep = [(1 << 32) - 1, 0, 1, 1 << 31, 3, 0xffff] + [int(x) for x in rng.integers(0, 1 << 32, size=150, dtype=np.uint64)]
keep += [int(x) & int(y) & int(z) for x, y, z in rng.integers(0, 1 << 32, size=(44, 3), dtype=np.uint64)]
kt = hip_ops.masks_to_tensor(keep, d); co = torch.tensor([i % 3 for i in range(len(keep))], dtype=torch.int32, device=d)
res = {}
for blocks in (0, 8, 40):
    for t in (0, 1):
        lib.iq_set_tuning(6, blocks); lib.iq_set_tuning(7, t)
        res[(blocks, t)] = (m.coalition_logits(clouds, clouds.mean(dim=1), rid, kt, co, num_regions=32).clone(), m.forward_points(clouds).clone())
lib.iq_set_tuning(6, 0); lib.iq_set_tuning(7, 0)
ok = all(torch.equal(res[(b, 0)][i], res[(0, 1)][i]) and torch.equal(res[(b, 1)][i], res[(0, 1)][i]) for b in (0, 8, 40) for i in (0, 1))
print("ping-pong == two workgroups, bit for bit (3 range sizes, coalitions + dense):", ok)
PY
for rep in 1 2 3; do
  for t in 1 0; do
    echo "pointnet2 7=$t: $(python3 tools/bench_models.py --model pointnet2 --mode shapley --steps 8 --tune 7=$t 2>&1 | tail -1 | cut -c1-200)"
  done
done
