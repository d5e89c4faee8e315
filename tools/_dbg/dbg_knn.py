import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from conftest import load_golden
from interpret_quality_amd import hip_ops, synth
from oracle import ref_cpu as O
g = load_golden("dgcnn.npz")
pts,_ = synth.make_cloud(0)
data = torch.from_numpy(pts).unsqueeze(0)
center = torch.mean(data, dim=1).squeeze()
half = data.clone(); half[0, g["region_id"] >= 16, :] = center
clouds = torch.cat([data, half], dim=0)
x_cf = clouds.permute(0,2,1).contiguous()
sd = synth.to_torch(synth.dgcnn_state_dict(0))
with torch.no_grad():
    _, aux = O.dgcnn_forward(sd, x_cf, 20, False, return_aux=True)
def cmp(name, x_rows, want):
    got = hip_ops.knn(x_rows.cuda().contiguous(), 20).cpu().numpy()
    for b in range(2):
        bad = sum(set(got[b,i].tolist()) != set(want[b,i].tolist()) for i in range(1024))
        print(name, 'cloud', b, 'rows with different neighbour sets:', bad)
cmp('xyz', clouds, g["knn_xyz"].astype(np.int32))
cmp('feat64', aux["x1"].permute(0,2,1), g["knn_feat64"].astype(np.int32))
with torch.no_grad():
    for nm, t in (("x2", aux["x2"]), ("x3", aux["x3"])):
        cmp(nm, t.permute(0,2,1), O.knn(t, 20).numpy())
