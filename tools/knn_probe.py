"""Diagnostic: rounds / busy-lane statistics of the kNN selection (tuning key 4 = 3) on DGCNN-like inputs."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from interpret_quality_amd import _lib, hip_ops, synth
lib = _lib.load()
d = torch.device("cuda:0")
pts = torch.stack([torch.from_numpy(synth.make_cloud(i)[0]) for i in range(64)]).to(d)   # (64,1024,3)
for name, x in (("xyz 1024 rows", pts), ("xyz 544 rows", pts[:, :544].contiguous()), ("gauss64 544 rows", torch.randn(64, 544, 64, device=d))):
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
