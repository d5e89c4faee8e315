"""How tight is a kNN filter threshold taken from the PREVIOUS layer graph?  CPU study on the oracle: for masked
clouds of 8 / 16 / 24 / 32 kept regions, per feature-space layer: the number of keys per query that pass tau = the farthest of the
query's 20 previous neighbours, measured in this layer's features (20 must pass), and how many of the previous neighbours are
neighbours again.  Result (profiles/r04_knn_seeded.txt): 27-35 pass on average, p99 42-77; 16.3-18.0 of 20 stay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # (under tests/: the oracle is test infrastructure)
import numpy as np, torch
from oracle import ref_cpu as O
from interpret_quality_amd import synth
torch.set_num_threads(8)
sd = synth.to_torch(synth.dgcnn_state_dict(0))
pts,_ = synth.make_cloud(5)
rng = np.random.default_rng(0)
# compact coalition clouds: kept points + 1 centre (distinct rows), sizes vary
def compact(m_regions):
    data = torch.from_numpy(pts).unsqueeze(0)
    fps = O.farthest_point_sample(data, 32)[0]
    rid = O.cal_region_id(data, fps)
    keep = rng.choice(32, m_regions, replace=False)
    mask = np.isin(rid, keep)
    x = pts[mask]
    c = pts.mean(0, keepdims=True)
    return np.concatenate([x, c]).astype(np.float32) if m_regions < 32 else pts
def pair(x):  # (C,N) -> neg sq dist (N,N)
    inner = x.t() @ x * -2
    xx = (x**2).sum(0, keepdim=True)
    return -xx.t() - inner - xx
for m in (8, 16, 24, 32):
    xyz = torch.from_numpy(compact(m)).t().unsqueeze(0)   # (1,3,N)  (distinct rows only: an approximation of the compact layout, multiplicities ignored)
    n = xyz.shape[2]
    x = xyz; idx_prev=None
    feats=[]
    for j in range(1,5):
        idx = O.knn(x, 20)
        if idx_prev is not None:
            d = pair(x[0])                                   # (N,N), larger = nearer
            seed = torch.gather(d, 1, idx_prev[0])           # distances to the previous layer's neighbours in THIS layer's features
            tau = seed.min(dim=1)[0]                         # the farthest of the 20 seeds
            surv = (d >= tau[:,None]).sum(1).float()
            true20 = torch.gather(d,1,idx[0]).min(1)[0]
            print("m=%2d N=%4d layer %d C=%3d: survivors per query mean %.1f median %.0f p90 %.0f p99 %.0f max %.0f | seeds that are true neighbours %.1f of 20"
                  % (m, n, j, x.shape[1], surv.mean(), surv.median(), surv.quantile(0.9), surv.quantile(0.99), surv.max(),
                     (idx[0].unsqueeze(2)==idx_prev[0].unsqueeze(1)).any(1).sum(1).float().mean()))
        x = O._edge_conv(x, sd, j, 20, idx)
        idx_prev = idx
