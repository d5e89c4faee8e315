"""PointNet++ MSG classifier on the HIP path.

Host-side mirror of models/pointnet2.py:244-276 (PointNet2ClsMsg): same constructor argument, same
``state_dict`` keys (163 tensors), same call ``model(xyz: (B,3,N)) -> logits (B,10)``.  The torch
modules only hold parameters; FPS, ball query, grouping, the shared MLPs and the pooling run in
libiq_hip.so (csrc/iq_pointnet2.hip, iq_geom.hip).
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import _lib, hip_ops, workspace
from .pointnet import BN_EPS, _np

SA1 = dict(npoint=512, radius=[0.1, 0.2, 0.4], nsample=[16, 32, 128], in_channel=0,
           mlp=[[32, 32, 64], [64, 64, 128], [64, 96, 128]])                         # models/pointnet2.py:253
SA2 = dict(npoint=128, radius=[0.2, 0.4, 0.8], nsample=[32, 64, 128], in_channel=320,
           mlp=[[64, 64, 128], [128, 128, 256], [128, 128, 256]])                    # :254
SA3_MLP = [256, 512, 1024]                                                           # :255


def fold_conv_bn(sd, conv, bn):
    w = _np(sd[conv + ".weight"])
    w = w.reshape(w.shape[0], -1)
    b = _np(sd[conv + ".bias"])
    if bn is not None:
        s = _np(sd[bn + ".weight"]) / np.sqrt(_np(sd[bn + ".running_var"]) + BN_EPS)
        w = w * s[:, None]
        b = (b - _np(sd[bn + ".running_mean"])) * s + _np(sd[bn + ".bias"])
    return w, b  # float64


class PackedWeights2:
    def __init__(self, sd, device):
        lib = _lib.load()
        self._keep = []
        self.struct = _lib.PointNet2Weights()

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
            # wide layers: also as three bf16 terms (include/iq.h); 320 outputs: the first 256 columns take them
            wide = (cout % 256 == 0 or (cout > 256 and cout % 256 == 64)) and cin >= 32
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

        def scale(dst, sa, i, cfg, feat_in):
            w0, b0 = fold_conv_bn(sd, "%s.conv_blocks.%d.0" % (sa, i), "%s.bn_blocks.%d.0" % (sa, i))
            wx = w0[:, feat_in:feat_in + 3]                       # relative xyz comes LAST (models/pointnet2.py:226)
            bias = b0 if feat_in == 0 else np.zeros_like(b0)      # with features the bias travels in U
            dst.w1x = dev(np.concatenate([wx, bias[:, None]], axis=1)).data_ptr()
            w2, b2 = fold_conv_bn(sd, "%s.conv_blocks.%d.1" % (sa, i), "%s.bn_blocks.%d.1" % (sa, i))
            w3, b3 = fold_conv_bn(sd, "%s.conv_blocks.%d.2" % (sa, i), "%s.bn_blocks.%d.2" % (sa, i))
            dst.l2, dst.l3 = dense(w2, b2), dense(w3, b3)
            if sa == "sa2" and w2.shape == (128, 128) and w3.shape == (256, 128):   # the widest scales: bf16 matrix pipe
                self.struct.sa2_l2_bf3[i], self.struct.sa2_l3_bf3[i] = bf3(w2), bf3(w3)
            dst.radius = cfg["radius"][i]
            dst.nsample = cfg["nsample"][i]
            return w0[:, :feat_in], b0

        for i in range(3):
            scale(self.struct.sa1[i], "sa1", i, SA1, 0)
        uw, ub = [], []
        for i in range(3):
            wf, b0 = scale(self.struct.sa2[i], "sa2", i, SA2, 320)
            uw.append(wf)
            ub.append(b0)
        self.struct.sa2_u = dense(np.concatenate(uw, axis=0), np.concatenate(ub, axis=0))
        w, b = fold_conv_bn(sd, "sa3.mlp_convs.0", "sa3.mlp_bns.0")   # input = [xyz, features] (xyz FIRST, :132-135)
        wpad = np.zeros((w.shape[0], 648))
        wpad[:, :643] = w
        self.struct.sa3_l1 = dense(wpad, b)
        self.struct.sa3_l2 = dense(*fold_conv_bn(sd, "sa3.mlp_convs.1", "sa3.mlp_bns.1"))
        self.struct.sa3_l3 = dense(*fold_conv_bn(sd, "sa3.mlp_convs.2", "sa3.mlp_bns.2"))
        self.struct.fc1 = dense(*fold_conv_bn(sd, "fc1", "bn1"))
        self.struct.fc2 = dense(*fold_conv_bn(sd, "fc2", "bn2"))
        self.struct.fc3 = dense(*fold_conv_bn(sd, "fc3", None))
        self.num_classes = int(sd["fc3.weight"].shape[0])


class PointNet2Engine:
    def __init__(self, state_dict, device):
        if torch.device(device).type != "cuda":
            raise _lib.IqError("PointNet2Engine needs a GPU device (no CPU fallback)")
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.weights = PackedWeights2(state_dict, self.device)
        self._ws = None

    def forward_points(self, xyz):
        """xyz (B,N,3) contiguous float32 on the GPU -> logits (B,10)."""
        if not xyz.is_cuda or xyz.dtype != torch.float32 or not xyz.is_contiguous():
            raise _lib.IqError("xyz must be a contiguous float32 GPU tensor (B,N,3)")
        b, n, _ = xyz.shape
        workspace.ensure(self, self.lib.iq_pointnet2_workspace_bytes(b))
        logits = torch.empty((b, self.weights.num_classes), dtype=torch.float32, device=self.device)
        rc = self.lib.iq_pointnet2_forward(ctypes.byref(self.weights.struct), ctypes.c_void_p(xyz.data_ptr()),
                                           ctypes.c_void_p(logits.data_ptr()), ctypes.c_void_p(self._ws.data_ptr()),
                                           self._ws.numel(), b, n, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "iq_pointnet2_forward")
        return logits


    def coalition_logits(self, clouds, centers, region_id, keep, cloud_of=None):
        """iq_pointnet2_coalitions: clouds (nc,N,3), centers (nc,3), region_id (nc,N) i32, keep (B,) i64 bit masks,
        cloud_of (B,) i32 or None -> logits (B,C)."""
        for t, dt, nm in ((clouds, torch.float32, "clouds"), (centers, torch.float32, "centers"), (region_id, torch.int32, "region_id"),
                          (keep, torch.int64, "keep"), (cloud_of, torch.int32, "cloud_of")):
            if t is None and nm == "cloud_of":
                continue
            if t is None or not t.is_cuda or t.dtype != dt or not t.is_contiguous():
                raise _lib.IqError("%s must be a contiguous %s GPU tensor" % (nm, dt))
        nc, n, _ = clouds.shape
        b = keep.shape[0]
        workspace.ensure(self, self.lib.iq_pointnet2_coalitions_workspace_bytes(b, nc, n))
        logits = torch.empty((b, self.weights.num_classes), dtype=torch.float32, device=self.device)
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
        rc = self.lib.iq_pointnet2_coalitions(ctypes.byref(self.weights.struct), p(clouds), p(centers), p(region_id), p(keep),
                                              p(cloud_of), p(logits), p(self._ws), self._ws.numel(), b, nc, n,
                                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "iq_pointnet2_coalitions")
        return logits


def _holder_msg(cfg):
    m = nn.Module()
    m.conv_blocks, m.bn_blocks = nn.ModuleList(), nn.ModuleList()
    for mlp in cfg["mlp"]:
        convs, bns = nn.ModuleList(), nn.ModuleList()
        last = cfg["in_channel"] + 3
        for c in mlp:
            convs.append(nn.Conv2d(last, c, 1))
            bns.append(nn.BatchNorm2d(c))
            last = c
        m.conv_blocks.append(convs)
        m.bn_blocks.append(bns)
    return m


class PointNet2ClsMsg(nn.Module):
    """Parameter container with the reference's state-dict layout; forward runs on the HIP path."""

    max_clouds_per_call = 4096  # bounds the workspace (3.2 MB per cloud: 13 GB; one launch covers a 3300-coalition pose)
    preferred_clouds_per_call = 1024  # drivers batch at least this many materialised clouds per launch

    def __init__(self, args=None):
        super().__init__()
        self.args = args
        self.output_channels = 40 if getattr(args, "dataset", "modelnet10") == "modelnet40" else 10
        self.sa1, self.sa2 = _holder_msg(SA1), _holder_msg(SA2)
        sa3 = nn.Module()
        sa3.mlp_convs, sa3.mlp_bns = nn.ModuleList(), nn.ModuleList()
        last = 640 + 3
        for c in SA3_MLP:
            sa3.mlp_convs.append(nn.Conv2d(last, c, 1))
            sa3.mlp_bns.append(nn.BatchNorm2d(c))
            last = c
        self.sa3 = sa3
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
            raise _lib.IqError("the HIP PointNet++ path implements eval mode only")
        if self._engine is None:
            self._engine = PointNet2Engine(self.state_dict(), self.fc3.weight.device)
        return self._engine

    def forward_points(self, xyz):
        """(B,N,3) channel-last clouds (what the mask kernel writes) -> logits."""
        eng = self.engine()
        return workspace.run_in_steps(eng, xyz.shape[0], self.max_clouds_per_call, lambda b: eng.lib.iq_pointnet2_workspace_bytes(b),
                                      lambda lo, hi: eng.forward_points(xyz if (lo, hi) == (0, xyz.shape[0]) else xyz[lo:hi].contiguous()))

    def forward(self, xyz):
        """xyz (B,3,N) as in the reference -> logits (B,10)."""
        return self.forward_points(xyz.permute(0, 2, 1).contiguous())

    def coalition_logits(self, clouds, centers, region_id, keep, cloud_of=None, num_regions=None, validate=True):
        """Same call as PointNetCls.coalition_logits: logits of B coalitions given as region bit masks (sa1 from the
        per-cloud pair tables, csrc/iq_pointnet2.hip)."""
        if validate:
            hip_ops.check_index_range(region_id, 0, int(num_regions) if num_regions else 64, "region_id")
        eng = self.engine()
        nc, b, n = clouds.shape[0], keep.shape[0], clouds.shape[1]
        if cloud_of is None and nc not in (1, b):
            raise _lib.IqError("cloud_of is required when 1 < number of clouds != number of coalitions")
        own = [cloud_of]

        def call(lo, hi):
            if (lo, hi) == (0, b):
                return eng.coalition_logits(clouds, centers, region_id, keep, cloud_of)
            if own[0] is None and nc == b:     # one cloud per coalition, split over launches: name each launch's clouds
                own[0] = torch.arange(b, dtype=torch.int32, device=keep.device)
            return eng.coalition_logits(clouds, centers, region_id, keep[lo:hi].contiguous(),
                                        own[0][lo:hi].contiguous() if own[0] is not None else None)
        # the launch size comes from the memory that is free now (workspace.py), at most max_clouds_per_call
        return workspace.run_in_steps(eng, b, self.max_clouds_per_call,
                                      lambda k: eng.lib.iq_pointnet2_coalitions_workspace_bytes(k, nc, n), call)
