"""PointConv (density) classifier on the HIP path.

Host-side mirror of models/pointconv.py:394-424 (PointConvDensityClsSsg): same constructor argument,
same ``state_dict`` keys (226 tensors), same call ``model(xyz: (B,3,N)) -> logits (B,10)``.  Density,
FPS, kNN grouping, the shared MLPs, DensityNet / WeightNet and the weighted aggregation run in
libiq_hip.so (csrc/iq_pointconv.hip); the torch modules only hold parameters.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import _lib, hip_ops, workspace
from .pointnet2 import fold_conv_bn

SA = [dict(npoint=512, nsample=32, in_channel=3, mlp=[64, 64, 128], bandwidth=0.1),          # models/pointconv.py:403
      dict(npoint=128, nsample=64, in_channel=128 + 3, mlp=[128, 128, 256], bandwidth=0.2),  # :404
      dict(npoint=1, nsample=None, in_channel=256 + 3, mlp=[256, 512, 1024], bandwidth=0.4)]  # :405


class PackedWeightsC:
    def __init__(self, sd, device):
        lib = _lib.load()
        self._keep = []
        self.struct = _lib.PointConvWeights()

        def dev(arr):
            t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to(device)
            self._keep.append(t)
            return t

        def dense(w, b):
            cout, cin = w.shape
            w32 = np.ascontiguousarray(w, dtype=np.float32)
            out = np.empty(lib.iq_packed_floats(cout, cin), dtype=np.float32)
            _lib.check(lib.iq_pack_weight(w32.ctypes.data, out.ctypes.data, cout, cin), "iq_pack_weight")
            bp = np.zeros(lib.iq_padded_cout(cout), dtype=np.float32)
            bp[:cout] = b
            wt, bt = dev(out), dev(bp)
            wide = cout % 256 == 0 and cin >= 32                          # wide layers: also as three bf16 terms (include/iq.h)
            return _lib.DenseLayer(wt.data_ptr(), bt.data_ptr(), cin, cout, bf3(w32) if wide else None)

        def bf3(w):
            """the same folded weights as three bf16 terms (iq_pack_weight_bf3) on the device"""
            cout, cin = w.shape
            w32 = np.ascontiguousarray(w, dtype=np.float32)
            out = np.empty(lib.iq_packed_bf3_elems(cout, cin), dtype=np.uint16)
            _lib.check(lib.iq_pack_weight_bf3(w32.ctypes.data, out.ctypes.data, cout, cin), "iq_pack_weight_bf3")
            t = torch.from_numpy(out.view(np.int16)).to(device)
            self._keep.append(t)
            return t.data_ptr()

        def tiny(prefix):
            rows = []
            for j in range(3):
                w, b = fold_conv_bn(sd, "%s.mlp_convs.%d" % (prefix, j), "%s.mlp_bns.%d" % (prefix, j))
                rows.append(np.concatenate([w, b[:, None]], axis=1).reshape(-1))
            return dev(np.concatenate(rows)).data_ptr()

        for k, cfg in enumerate(SA):
            p = "sa%d" % (k + 1)
            dst = self.struct.sa[k]
            w0, b0 = fold_conv_bn(sd, p + ".mlp_convs.0", p + ".mlp_bns.0")
            feat = cfg["in_channel"] - 3                               # input = [x_p - c (3) ; features]  (:133-135)
            bias = b0 if feat == 0 else np.zeros_like(b0)
            dst.w1x = dev(np.concatenate([w0[:, :3], bias[:, None]], axis=1)).data_ptr()
            if feat:
                dst.u = dense(w0[:, 3:], b0)
            w2, b2 = fold_conv_bn(sd, p + ".mlp_convs.1", p + ".mlp_bns.1")
            w3, b3 = fold_conv_bn(sd, p + ".mlp_convs.2", p + ".mlp_bns.2")
            dst.l2, dst.l3 = dense(w2, b2), dense(w3, b3)
            if k == 1 and w2.shape == (128, 128) and w3.shape == (256, 128):   # sa2's grouped MLP on the bf16 matrix pipe
                self.struct.sa2_l2_bf3, self.struct.sa2_l3_bf3 = bf3(w2), bf3(w3)
            dst.densitynet = tiny(p + ".densitynet")
            dst.weightnet = tiny(p + ".weightnet")
            dst.linear = dense(*fold_conv_bn(sd, p + ".linear", p + ".bn_linear"))
            dst.bandwidth = cfg["bandwidth"]
            dst.nsample = cfg["nsample"] or 0
        self.struct.fc1 = dense(*fold_conv_bn(sd, "fc1", "bn1"))
        self.struct.fc2 = dense(*fold_conv_bn(sd, "fc2", "bn2"))
        self.struct.fc3 = dense(*fold_conv_bn(sd, "fc3", None))
        self.num_classes = int(sd["fc3.weight"].shape[0])


class PointConvEngine:
    def __init__(self, state_dict, device):
        if torch.device(device).type != "cuda":
            raise _lib.IqError("PointConvEngine needs a GPU device (no CPU fallback)")
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.weights = PackedWeightsC(state_dict, self.device)
        self._ws = None
        self._tab = {}      # which source clouds the tables at the head of the workspace belong to (coalition_logits)

    def forward_points(self, xyz):
        if not xyz.is_cuda or xyz.dtype != torch.float32 or not xyz.is_contiguous():
            raise _lib.IqError("xyz must be a contiguous float32 GPU tensor (B,N,3)")
        b, n, _ = xyz.shape
        workspace.ensure(self, self.lib.iq_pointconv_workspace_bytes(b, n))
        self._tab = {}              # the dense forward's arrays start at the head of the workspace, where the coalition path keeps its tables
        logits = torch.empty((b, self.weights.num_classes), dtype=torch.float32, device=self.device)
        rc = self.lib.iq_pointconv_forward(ctypes.byref(self.weights.struct), ctypes.c_void_p(xyz.data_ptr()),
                                           ctypes.c_void_p(logits.data_ptr()), ctypes.c_void_p(self._ws.data_ptr()),
                                           self._ws.numel(), b, n, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "iq_pointconv_forward")
        return logits

    def coalition_logits(self, clouds, centers, region_id, keep, cloud_of=None, walk=None):
        """iq_pointconv_coalitions: clouds (nc,N,3), centers (nc,3), region_id (nc,N) i32, keep (B,) i64 bit masks,
        cloud_of (B,) i32 or None -> logits (B,C).  walk: True / False names how groups are formed (sorted-list walk or a kNN per
        coalition) for a batch that is split over several launches; None lets the library decide from this launch alone."""
        for t, dt, nm in ((clouds, torch.float32, "clouds"), (centers, torch.float32, "centers"), (region_id, torch.int32, "region_id"),
                          (keep, torch.int64, "keep"), (cloud_of, torch.int32, "cloud_of")):
            if t is None and nm == "cloud_of":
                continue
            if t is None or not t.is_cuda or t.dtype != dt or not t.is_contiguous():
                raise _lib.IqError("%s must be a contiguous %s GPU tensor" % (nm, dt))
        nc, n, _ = clouds.shape
        b = keep.shape[0]
        workspace.ensure(self, self.lib.iq_pointconv_coalitions_workspace_bytes(b, nc, n))
        logits = torch.empty((b, self.weights.num_classes), dtype=torch.float32, device=self.device)
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
        state = ctypes.c_int(self._tables_state(clouds, centers, nc, n) | (0 if walk is None else (4 if walk else 8)))
        rc = self.lib.iq_pointconv_coalitions_cached(ctypes.byref(self.weights.struct), p(clouds), p(centers), p(region_id), p(keep),
                                                     p(cloud_of), p(logits), p(self._ws), self._ws.numel(), b, nc, n,
                                                     ctypes.byref(state), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "iq_pointconv_coalitions")
        self._tab["state"] = state.value & 3
        return logits

    def _tables_state(self, clouds, centers, nc, n):
        """Which per-cloud structures (sorted lists, sa1 pair tables) the head of the workspace still holds for THESE clouds:
        the same workspace allocation, the same (nc, N), and the same `clouds` / `centers` TENSORS at the same version (torch
        counts every in-place write) as the call that built them - an identity test, no read-back, no synchronisation (comparing
        the coordinates on the device cost a host sync per launch: 4.8 ms of exposed launch latency per 3300-coalition step).
        The two tensors are held, so their memory cannot be handed to other data meanwhile.  The launches of one driver call
        (the chunks of an interaction ratio, the batches of a pose) pass the same tensors and re-use the tables; round 3 rebuilt
        ~1 GB per source cloud on every launch.  (Writing into a held tensor behind torch's back - a raw pointer - is not seen.)"""
        t = self._tab
        try:
            versions = (clouds._version, centers._version)
        except RuntimeError:        # inference tensors carry no version counter: nothing to tell a rewrite by, so rebuild
            self._tab = {}
            return 0
        # the stream is part of the key: tables another stream is still building must not be read from this one
        key = (self._ws.data_ptr(), nc, n, clouds.data_ptr(), centers.data_ptr(), versions, torch.cuda.current_stream().cuda_stream)
        if t.get("key") == key and t.get("state", 0) and t["clouds"] is clouds and t["centers"] is centers:
            return int(t["state"])
        self._tab = {"key": key, "clouds": clouds, "centers": centers, "state": 0}
        return 0


def _tiny_holder(dims):
    m = nn.Module()
    m.mlp_convs, m.mlp_bns = nn.ModuleList(), nn.ModuleList()
    for j in range(3):
        m.mlp_convs.append(nn.Conv2d(dims[j], dims[j + 1], 1))
        m.mlp_bns.append(nn.BatchNorm2d(dims[j + 1]))
    return m


def _sa_holder(cfg):
    m = nn.Module()
    m.mlp_convs, m.mlp_bns = nn.ModuleList(), nn.ModuleList()
    last = cfg["in_channel"]
    for c in cfg["mlp"]:
        m.mlp_convs.append(nn.Conv2d(last, c, 1))
        m.mlp_bns.append(nn.BatchNorm2d(c))
        last = c
    m.weightnet = _tiny_holder([3, 8, 8, 16])
    m.linear = nn.Linear(16 * last, last)
    m.bn_linear = nn.BatchNorm1d(last)
    m.densitynet = _tiny_holder([1, 16, 8, 1])
    return m


class PointConvDensityClsSsg(nn.Module):
    """Parameter container with the reference's state-dict layout; forward runs on the HIP path."""

    max_clouds_per_call = 4096  # bounds the workspace (9.5 MB per cloud: 39 GB of the 288 GB; one launch covers a 3300-coalition pose)
    preferred_clouds_per_call = 4096  # drivers batch at least this many materialised clouds per launch

    def __init__(self, args=None):
        super().__init__()
        self.args = args
        self.output_channels = 40 if getattr(args, "dataset", "modelnet10") == "modelnet40" else 10
        self.sa1, self.sa2, self.sa3 = _sa_holder(SA[0]), _sa_holder(SA[1]), _sa_holder(SA[2])
        self.fc1, self.bn1 = nn.Linear(1024, 512), nn.BatchNorm1d(512)
        self.fc2, self.bn2 = nn.Linear(512, 256), nn.BatchNorm1d(256)
        self.fc3 = nn.Linear(256, self.output_channels)
        self._engine = None

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    def engine(self):
        if self.training:
            raise _lib.IqError("the HIP PointConv path implements eval mode only")
        if self._engine is None:
            self._engine = PointConvEngine(self.state_dict(), self.fc3.weight.device)
        return self._engine

    def forward_points(self, xyz):
        eng = self.engine()
        n = xyz.shape[1]
        return workspace.run_in_steps(eng, xyz.shape[0], self.max_clouds_per_call, lambda b: eng.lib.iq_pointconv_workspace_bytes(b, n),
                                      lambda lo, hi: eng.forward_points(xyz if (lo, hi) == (0, xyz.shape[0]) else xyz[lo:hi].contiguous()))

    def forward(self, xyz):
        """xyz (B,3,N) as in the reference -> logits (B,10)."""
        return self.forward_points(xyz.permute(0, 2, 1).contiguous())

    def coalition_logits(self, clouds, centers, region_id, keep, cloud_of=None, num_regions=None, validate=True):
        """Same call as PointNetCls.coalition_logits: logits of B coalitions given as region bit masks (the masked clouds are
        written inside the library; sa1 / sa2 groups from the source clouds' sorted neighbour lists, csrc/iq_pointconv.hip).
        Clouds of more than 1024 points go through mask kernel + forward_points, source cloud by source cloud; fewer than 64
        points are rejected here as in the dense forward (sa2 groups 64 neighbours; below 512 points sa1's sampling repeats
        index 0, as models/pointconv.py:54-77 does)."""
        if validate:
            hip_ops.check_index_range(region_id, 0, int(num_regions) if num_regions else 64, "region_id")
        eng = self.engine()
        nc, b = clouds.shape[0], keep.shape[0]
        if cloud_of is None and nc not in (1, b):
            raise _lib.IqError("cloud_of is required when 1 < number of clouds != number of coalitions")
        if clouds.shape[1] < 64:
            raise _lib.IqError("PointConv needs at least 64 points per cloud (sa2 groups 64 neighbours), got %d" % clouds.shape[1])
        if clouds.shape[1] > 1024:   # beyond the library's coalition entry: mask kernel + forward, source cloud by source cloud
            which = cloud_of if cloud_of is not None else (torch.zeros(b, dtype=torch.int32, device=keep.device) if nc == 1
                                                          else torch.arange(b, dtype=torch.int32, device=keep.device))
            out = torch.empty((b, self.output_channels), dtype=torch.float32, device=keep.device)
            for c in range(nc):
                sel = torch.nonzero(which == c).flatten()
                if sel.numel():
                    x = hip_ops.mask_coalitions(clouds[c].contiguous(), region_id[c].contiguous(), keep[sel].contiguous(),
                                                centers[c].contiguous())
                    out[sel] = self.forward_points(x)
            return out
        own = [cloud_of]
        n = clouds.shape[1]
        # how groups are formed is decided ONCE, from the whole batch (the library's own rule, iq.h): a memory-tight run that splits
        # the batch - or its short last launch - must not switch to the other summation order
        walk = nc <= 8 or nc * 8 <= b

        def call(lo, hi):
            if (lo, hi) == (0, b):
                return eng.coalition_logits(clouds, centers, region_id, keep, cloud_of, walk)
            if own[0] is None and nc == b:     # one cloud per coalition, split over launches: name each launch's clouds
                own[0] = torch.arange(b, dtype=torch.int32, device=keep.device)
            return eng.coalition_logits(clouds, centers, region_id, keep[lo:hi].contiguous(),
                                        own[0][lo:hi].contiguous() if own[0] is not None else None, walk)
        # the launch size comes from the memory that is free now (workspace.py), at most max_clouds_per_call
        return workspace.run_in_steps(eng, b, self.max_clouds_per_call,
                                      lambda k: eng.lib.iq_pointconv_coalitions_workspace_bytes(k, nc, n), call)
