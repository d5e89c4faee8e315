import sys, argparse; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from conftest import load_golden
from interpret_quality_amd import hip_ops, synth, interaction
from interpret_quality_amd.dgcnn import DGCNN_cls
from oracle import ref_cpu as O
g = load_golden("dgcnn.npz")
m = DGCNN_cls(argparse.Namespace(dataset="modelnet10", k=20)); m.load_state_dict(synth.to_torch(synth.dgcnn_state_dict(0))); m = m.cuda().eval()
pts, label = synth.make_cloud(0)
data = torch.from_numpy(pts).unsqueeze(0)
center = torch.mean(data, dim=1).squeeze()
sd = synth.to_torch(synth.dgcnn_state_dict(0))
for tag in ("ratio0","ratio50","ratio100"):
    ctx = g[tag+"_contexts"]
    for p,(ri,rj) in enumerate(g["pairs"]):
        masked = O.interaction_masked_batch(data.permute(0,2,1), center, g["region_id"], ri, rj, ctx[p]).contiguous()
        got = m(masked.cuda()).cpu()
        want = torch.from_numpy(g["%s_dgcnn_logits"%tag][p])
        e = (got-want).abs().max(dim=1)[0]/want.abs().max()
        print(tag, p, ['%.1e'%v for v in e.tolist()])
        bad = int(e.argmax())
        if e.max() > 5e-5:
            x = masked[bad:bad+1]
            with torch.no_grad():
                _, aux = O.dgcnn_forward(sd, x, 20, False, return_aux=True)
            for nm,t in (("xyz", x), ("x1",aux["x1"]),("x2",aux["x2"]),("x3",aux["x3"])):
                rows = t.permute(0,2,1).contiguous()
                gi = hip_ops.knn(rows.cuda(), 20).cpu().numpy()[0]
                wi = O.knn(t, 20).numpy()[0]
                inner = torch.matmul(t.transpose(2,1), t) * -2; xx = torch.sum(t**2, dim=1, keepdim=True)
                dist = (-xx - inner - xx.transpose(2,1))[0].numpy()
                nb=0; worst=0
                for i in range(1024):
                    sg, sw = set(gi[i].tolist()), set(wi[i].tolist())
                    if sg != sw:
                        dg = np.sort(dist[i, list(sg-sw)]); dw = np.sort(dist[i, list(sw-sg)])
                        gap = np.abs(dg-dw).max()
                        if gap > 0: nb += 1; worst = max(worst, gap)
                print('   ', nm, 'rows with non-tied set differences:', nb, 'worst gap', worst)
