"""DGCNN and GCNN classifiers on the HIP path.

Host-side mirror of models/dgcnn.py:51-194 (DGCNN_cls: dynamic feature-space kNN graph per layer;
GCNN_cls: one fixed xyz graph): same constructor argument, same ``state_dict`` keys (70 tensors, the
BatchNorms appear twice as ``bnK.*`` and ``convK.1.*``), same call ``model(x: (B,3,N)) -> logits``.
kNN, the EdgeConv GEMMs, the neighbour max, conv5, pooling and the head run in libiq_hip.so
(csrc/iq_dgcnn.hip); the torch modules only hold parameters.  ``gcnn_adv`` is GCNN_cls with another
checkpoint (tools/final_util.py:243-244).
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import _lib, hip_ops, workspace
from .pointnet import BN_EPS, _np

CONVS = [(6, 64), (128, 64), (128, 128), (256, 256)]  # models/dgcnn.py:66-77


def _bn_affine(sd, bn):
    s = _np(sd[bn + ".weight"]) / np.sqrt(_np(sd[bn + ".running_var"]) + BN_EPS)
    t = _np(sd[bn + ".bias"]) - _np(sd[bn + ".running_mean"]) * s
    return s, t


class PackedWeightsD:
    def __init__(self, sd, device, k):
        lib = _lib.load()
        self._keep = []
        self.struct = _lib.DgcnnWeights()

        def dense(w, b):
            cout, cin = w.shape
            w32 = np.ascontiguousarray(w, dtype=np.float32)
            out = np.empty(lib.iq_packed_floats(cout, cin), dtype=np.float32)
            _lib.check(lib.iq_pack_weight(w32.ctypes.data, out.ctypes.data, cout, cin), "iq_pack_weight")
            bp = np.zeros(lib.iq_padded_cout(cout), dtype=np.float32)
            bp[:cout] = b
            wt = torch.from_numpy(out).to(device)
            bt = torch.from_numpy(bp).to(device)
            self._keep += [wt, bt]
            return _lib.DenseLayer(wt.data_ptr(), bt.data_ptr(), cin, cout)

        for j, (cin2, cout) in enumerate(CONVS, start=1):
            c = cin2 // 2
            w = _np(sd["conv%d.0.weight" % j]).reshape(cout, cin2)
            s, t = _bn_affine(sd, "bn%d" % j)
            wa, wb = w[:, :c], w[:, c:]                       # [x_j - x_i ; x_i]  (models/dgcnn.py:45)
            cpad = 8 if c == 3 else c
            pq = np.zeros((2 * cout, cpad))
            pq[:cout, :c] = wa * s[:, None]                   # P = (s.W_a) x
            pq[cout:, :c] = (wb - wa) * s[:, None]            # Q = (s.(W_b - W_a)) x + t
            self.struct.pq[j - 1] = dense(pq, np.concatenate([np.zeros(cout), t]))
        s, t = _bn_affine(sd, "bn5")
        w5 = np.ascontiguousarray(_np(sd["conv5.0.weight"]).reshape(1024, 512) * s[:, None], dtype=np.float32)
        self.struct.conv5 = dense(w5, t)
        # conv5 (half of a DGCNN step, three quarters of GCNN's) runs on the bf16 matrix pipe: the same folded float32 weights as
        # three bf16 terms (csrc/iq_linear.hip: pn_gemm_bf3_kernel<pool>)
        w3 = np.empty(lib.iq_packed_bf3_elems(1024, 512), dtype=np.uint16)
        _lib.check(lib.iq_pack_weight_bf3(w5.ctypes.data, w3.ctypes.data, 1024, 512), "iq_pack_weight_bf3")
        w3t = torch.from_numpy(w3.view(np.int16)).to(device)
        self._keep.append(w3t)
        self.struct.reserved = 0
        self.struct.conv5_bf3 = w3t.data_ptr()
        s, t = _bn_affine(sd, "bn6")
        self.struct.fc1 = dense(_np(sd["linear1.weight"]) * s[:, None], t)      # linear1 has no bias (:79)
        s, t = _bn_affine(sd, "bn7")
        self.struct.fc2 = dense(_np(sd["linear2.weight"]) * s[:, None], _np(sd["linear2.bias"]) * s + t)
        self.struct.fc3 = dense(_np(sd["linear3.weight"]), _np(sd["linear3.bias"]))
        self.struct.k = k
        self.num_classes = int(sd["linear3.weight"].shape[0])


class DgcnnEngine:
    def __init__(self, state_dict, device, k, fixed_graph):
        if torch.device(device).type != "cuda":
            raise _lib.IqError("DgcnnEngine needs a GPU device (no CPU fallback)")
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.fixed_graph = int(fixed_graph)
        self.weights = PackedWeightsD(state_dict, self.device, k)
        self._ws = None

    def forward_points(self, xyz):
        if not xyz.is_cuda or xyz.dtype != torch.float32 or not xyz.is_contiguous():
            raise _lib.IqError("xyz must be a contiguous float32 GPU tensor (B,N,3)")
        b, n, _ = xyz.shape
        workspace.ensure(self, self.lib.iq_dgcnn_workspace_bytes(b, n))
        logits = torch.empty((b, self.weights.num_classes), dtype=torch.float32, device=self.device)
        rc = self.lib.iq_dgcnn_forward(ctypes.byref(self.weights.struct), ctypes.c_void_p(xyz.data_ptr()),
                                       ctypes.c_void_p(logits.data_ptr()), ctypes.c_void_p(self._ws.data_ptr()),
                                       self._ws.numel(), b, n, self.fixed_graph,
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "iq_dgcnn_forward")
        return logits


def _coalition_logits(eng, clouds, centers, region_id, keep, cloud_of):
    """iq_dgcnn_coalitions: clouds (nc,N,3), centers (nc,3), region_id (nc,N) i32, keep (B,) i64 bit masks,
    cloud_of (B,) i32 or None -> logits (B,C).  The masked clouds are never written."""
    for t, dt, nm in ((clouds, torch.float32, "clouds"), (centers, torch.float32, "centers"), (region_id, torch.int32, "region_id"),
                      (keep, torch.int64, "keep"), (cloud_of, torch.int32, "cloud_of")):
        if t is None and nm == "cloud_of":
            continue
        if t is None or not t.is_cuda or t.dtype != dt or not t.is_contiguous():
            raise _lib.IqError("%s must be a contiguous %s GPU tensor" % (nm, dt))
    nc, n, _ = clouds.shape
    b = keep.shape[0]
    workspace.ensure(eng, eng.lib.iq_dgcnn_workspace_bytes(b, n))
    logits = torch.empty((b, eng.weights.num_classes), dtype=torch.float32, device=eng.device)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
    rc = eng.lib.iq_dgcnn_coalitions(ctypes.byref(eng.weights.struct), p(clouds), p(centers), p(region_id), p(keep), p(cloud_of),
                                     p(logits), p(eng._ws), eng._ws.numel(), b, nc, n, eng.fixed_graph,
                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "iq_dgcnn_coalitions")
    return logits


class _GraphCnn(nn.Module):
    fixed_graph = False
    max_clouds_per_call = 4096  # bounds the workspace (4.4 MB per cloud)

    def __init__(self, args=None):
        super().__init__()
        self.args = args
        self.k = getattr(args, "k", 20)
        self.output_channels = 40 if getattr(args, "dataset", "modelnet10") == "modelnet40" else 10
        self.bn1, self.bn2, self.bn3, self.bn4 = nn.BatchNorm2d(64), nn.BatchNorm2d(64), nn.BatchNorm2d(128), nn.BatchNorm2d(256)
        self.bn5 = nn.BatchNorm1d(1024)
        for j, (cin, cout) in enumerate(CONVS, start=1):
            setattr(self, "conv%d" % j, nn.Sequential(nn.Conv2d(cin, cout, kernel_size=1, bias=False),
                                                     getattr(self, "bn%d" % j), nn.LeakyReLU(negative_slope=0.2)))
        self.conv5 = nn.Sequential(nn.Conv1d(512, 1024, kernel_size=1, bias=False), self.bn5, nn.LeakyReLU(negative_slope=0.2))
        self.linear1 = nn.Linear(2048, 512, bias=False)
        self.bn6 = nn.BatchNorm1d(512)
        self.linear2 = nn.Linear(512, 256)
        self.bn7 = nn.BatchNorm1d(256)
        self.linear3 = nn.Linear(256, self.output_channels)
        self._engine = None

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    def engine(self):
        if self.training:
            raise _lib.IqError("the HIP DGCNN path implements eval mode only")
        if self._engine is None:
            self._engine = DgcnnEngine(self.state_dict(), self.linear3.weight.device, self.k, self.fixed_graph)
        return self._engine

    def forward_points(self, xyz):
        """(B,N,3) channel-last clouds -> logits."""
        eng = self.engine()
        n = xyz.shape[1]
        return workspace.run_in_steps(eng, xyz.shape[0], self.max_clouds_per_call, lambda b: eng.lib.iq_dgcnn_workspace_bytes(b, n),
                                      lambda lo, hi: eng.forward_points(xyz if (lo, hi) == (0, xyz.shape[0]) else xyz[lo:hi].contiguous()))

    def forward(self, x):
        """x (B,3,N) as in the reference -> logits (B,10)."""
        return self.forward_points(x.permute(0, 2, 1).contiguous())

    def coalition_logits(self, clouds, centers, region_id, keep, cloud_of=None, num_regions=None, validate=True):
        """Same call as PointNetCls.coalition_logits: logits of B coalitions given as region bit masks."""
        if validate:
            hip_ops.check_index_range(region_id, 0, int(num_regions) if num_regions else 64, "region_id")
        eng = self.engine()
        nc, b, n = clouds.shape[0], keep.shape[0], clouds.shape[1]
        if cloud_of is None and nc not in (1, b):
            raise _lib.IqError("cloud_of is required when 1 < number of clouds != number of coalitions")
        own = [cloud_of]

        def call(lo, hi):
            if (lo, hi) == (0, b):
                return _coalition_logits(eng, clouds, centers, region_id, keep, cloud_of)
            if own[0] is None and nc == b:     # one cloud per coalition, split over launches: name each launch's clouds
                own[0] = torch.arange(b, dtype=torch.int32, device=keep.device)
            return _coalition_logits(eng, clouds, centers, region_id, keep[lo:hi].contiguous(),
                                     own[0][lo:hi].contiguous() if own[0] is not None else None)
        # the launch size comes from the memory that is free now (workspace.py), at most max_clouds_per_call
        return workspace.run_in_steps(eng, b, self.max_clouds_per_call, lambda k: eng.lib.iq_dgcnn_workspace_bytes(k, n), call)


class DGCNN_cls(_GraphCnn):
    """models/dgcnn.py:51-120."""
    fixed_graph = False


class GCNN_cls(_GraphCnn):
    """models/dgcnn.py:123-194."""
    fixed_graph = True
