"""Diagnostic: rounds / busy-lane statistics of the kNN selection (tuning key 4 = 3) on DGCNN-like inputs."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from interpret_quality_amd import _lib, hip_ops, synth
lib = _lib.load()
d = torch.device("cuda:0")
pts = torch.stack([torch.from_numpy(synth.make_cloud(i)[0]) for i in range(64)]).to(d)   # (64,1024,3)
def morton(p):
    """(B,N,3) -> the same points, each cloud sorted along a 30-bit Morton curve (spatially close points get close indices)"""
    q = ((p - p.amin(dim=1, keepdim=True)) / (p.amax(dim=1, keepdim=True) - p.amin(dim=1, keepdim=True) + 1e-9) * 1023).long()
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        return (v | (v << 2)) & 0x09249249
    code = spread(q[..., 0]) | (spread(q[..., 1]) << 1) | (spread(q[..., 2]) << 2)
    order = code.argsort(dim=1)
    return torch.gather(p, 1, order.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
sub = pts[:, torch.randperm(1024, device=d)[:544]].contiguous()
for name, x in (("xyz 1024 rows", pts), ("xyz 1024 rows, Morton order", morton(pts)), ("xyz 544 rows", sub), ("xyz 544 rows, Morton order", morton(sub)),
                ("gauss64 544 rows", torch.randn(64, 544, 64, device=d))):
    b, n, c = x.shape
    outp = torch.empty((b, n, 20), dtype=torch.int32, device=d)
    tmp = torch.zeros((b * n * 80 + 16 * b + 8192,), dtype=torch.uint8, device=d)
    lib.iq_set_tuning(4, 3)
    lib.iq_knn(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(outp.data_ptr()), ctypes.c_void_p(tmp.data_ptr()), tmp.numel(), b, n, c, 20, None)
    torch.cuda.synchronize()   # first call: clears counters left by earlier launches
    lib.iq_knn(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(outp.data_ptr()), ctypes.c_void_p(tmp.data_ptr()), tmp.numel(), b, n, c, 20, None)
    torch.cuda.synchronize()
    lib.iq_set_tuning(4, 0)
    # xx lives at tmp + align256(b*n*32)
    cnt = tmp[:24].cpu().numpy().view(np.uint64)
    rounds, work, waves = [int(v) for v in cnt]
    print("%s: waves %d, rounds/wave %.1f, busy lanes/round %.1f, inserts per lane %.1f" % (name, waves, rounds / waves, work / max(rounds, 1), work / waves / 64))
