"""PointNet classifier on the HIP path.

Host-side mirror of models/pointnet.py:91-115 (PointNetCls) of the reference: same constructor
argument, same ``state_dict`` keys (so tools/final_util.py:236-262-style checkpoints load), same
call signature ``model(x: (B,3,N)) -> (logits, trans_feat, crt_points)``.  The arithmetic runs in
libiq_hip.so (csrc/iq_pointnet.hip); the torch modules below only hold parameters.

In addition to the reference's dense ``forward`` the module exposes ``coalition_logits``: the
logits of a batch of region coalitions of one or more clouds WITHOUT materialising the masked
clouds - this is what the Shapley / interaction drivers call.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import _lib, hip_ops

BN_EPS = 1e-5


def _np(t):
    return t.detach().cpu().double().numpy()


def fold_bn(sd, layer, bn):
    """(W, b) of ``bn(layer(x))`` in eval mode, folded in float64 and rounded once to float32."""
    w = _np(sd[layer + ".weight"])
    w = w.reshape(w.shape[0], -1)
    b = _np(sd[layer + ".bias"])
    if bn is not None:
        s = _np(sd[bn + ".weight"]) / np.sqrt(_np(sd[bn + ".running_var"]) + BN_EPS)
        w = w * s[:, None]
        b = (b - _np(sd[bn + ".running_mean"])) * s + _np(sd[bn + ".bias"])
    return w.astype(np.float32), b.astype(np.float32)


class PackedWeights:
    """Device-resident, BN-folded, fragment-packed weights + the ctypes struct pointing at them."""

    def __init__(self, state_dict, device):
        lib = _lib.load()
        self.device = device
        self._keep = []
        self.struct = _lib.PointNetWeights()
        sd = state_dict

        def dev(arr):
            t = torch.from_numpy(np.ascontiguousarray(arr)).to(device)
            self._keep.append(t)
            return t

        def pack(w):
            cout, cin = w.shape
            out = np.empty(lib.iq_packed_floats(cout, cin), dtype=np.float32)
            w = np.ascontiguousarray(w, dtype=np.float32)
            _lib.check(lib.iq_pack_weight(w.ctypes.data, out.ctypes.data, cout, cin), "iq_pack_weight")
            return out

        def dense(name, layer, bn, extra_bias=None):
            w, b = fold_bn(sd, layer, bn)
            if extra_bias is not None:
                b = (b.astype(np.float64) + extra_bias).astype(np.float32)
            cout, cin = w.shape
            bp = np.zeros(lib.iq_padded_cout(cout), dtype=np.float32)
            bp[:cout] = b
            wt, bt = dev(pack(w)), dev(bp)

            def bf3():   # the same weights as three bf16 terms (iq_pack_weight_bf3): products on the bf16 matrix pipe, float32-exact
                w32 = np.ascontiguousarray(w, dtype=np.float32)
                w3 = np.empty(lib.iq_packed_bf3_elems(cout, cin), dtype=np.uint16)
                _lib.check(lib.iq_pack_weight_bf3(w32.ctypes.data, w3.ctypes.data, cout, cin), "iq_pack_weight_bf3")
                return dev(w3.view(np.int16)).data_ptr()
            wide = cout % 256 == 0 and cin % 32 == 0       # the 1024 -> 512 -> 256 heads (include/iq.h: iq_dense_layer.w_bf3)
            setattr(self.struct, name, _lib.DenseLayer(wt.data_ptr(), bt.data_ptr(), cin, cout, bf3() if wide else None))
            if name in ("fstn_c2", "feat_c2", "fstn_c3", "feat_c3"):   # layers 2-3 of the coalition chains
                setattr(self.struct, name + "_bf3", bf3())

        def in_layer(name, layer, bn):
            w, b = fold_bn(sd, layer, bn)  # (64,3), (64,)
            t = dev(np.concatenate([w, b[:, None]], axis=1).astype(np.float32))
            setattr(self.struct, name, t.data_ptr())

        in_layer("stn_in", "feat.stn.conv1", "feat.stn.bn1")
        dense("stn_c2", "feat.stn.conv2", "feat.stn.bn2")
        dense("stn_c3", "feat.stn.conv3", "feat.stn.bn3")
        dense("stn_fc1", "feat.stn.fc1", "feat.stn.bn4")
        dense("stn_fc2", "feat.stn.fc2", "feat.stn.bn5")
        dense("stn_fc3", "feat.stn.fc3", None, extra_bias=np.eye(3).reshape(-1))  # + iden, models/pointnet.py:42-45
        in_layer("feat_in", "feat.conv1", "feat.bn1")
        self.feature_transform = "feat.fstn.conv1.weight" in sd
        if self.feature_transform:
            dense("fstn_c1", "feat.fstn.conv1", "feat.fstn.bn1")
            dense("fstn_c2", "feat.fstn.conv2", "feat.fstn.bn2")
            dense("fstn_c3", "feat.fstn.conv3", "feat.fstn.bn3")
            dense("fstn_fc1", "feat.fstn.fc1", "feat.fstn.bn4")
            dense("fstn_fc2", "feat.fstn.fc2", "feat.fstn.bn5")
            # fc3 of the feature STN: output = packed B image of trans_feat (+ identity)
            w3, b3 = fold_bn(sd, "feat.fstn.fc3", None)
        else:   # feature_transform = False: the trunk gets the packed identity (the struct's fstn_c1.w stays NULL)
            w3, b3 = np.zeros((4096, 256), dtype=np.float32), np.zeros(4096, dtype=np.float32)
        w3 = np.ascontiguousarray(w3)
        b3 = np.ascontiguousarray(b3)
        ow = np.empty(lib.iq_packed_floats(4096, 256), dtype=np.float32)
        ob = np.empty(4096, dtype=np.float32)
        perm = np.empty(4096, dtype=np.int32)
        _lib.check(lib.iq_pack_fstn_fc3(w3.ctypes.data, b3.ctypes.data, ow.ctypes.data, ob.ctypes.data,
                                        perm.ctypes.data), "iq_pack_fstn_fc3")
        wt, bt = dev(ow), dev(ob)
        self.struct.fstn_fc3 = _lib.DenseLayer(wt.data_ptr(), bt.data_ptr(), 256, 4096)
        inv = np.empty(4096, dtype=np.int64)
        inv[perm] = np.arange(4096)
        self.unpack_index = torch.from_numpy(inv).to(device)  # trans_feat.flatten() = packed[unpack_index]
        dense("feat_c2", "feat.conv2", "feat.bn2")
        dense("feat_c3", "feat.conv3", "feat.bn3")
        dense("cls_fc1", "fc1", "bn1")
        dense("cls_fc2", "fc2", "bn2")
        dense("cls_fc3", "fc3", None)
        self.num_classes = int(sd["fc3.weight"].shape[0])


class PointNetEngine:
    """Owns packed weights and a growable workspace; issues iq_pointnet_coalitions."""

    def __init__(self, state_dict, device):
        if torch.device(device).type != "cuda":
            raise _lib.IqError("PointNetEngine needs a GPU device (no CPU fallback)")
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.weights = PackedWeights(state_dict, self.device)
        self._ws = None

    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self._ws

    def coalition_logits(self, clouds, centers, region_id, keep, cloud_of=None, num_regions=None,
                         channel_first=False, return_trans_feat=False, return_crt=False):
        """clouds (nc,N,3) [or (nc,3,N)], centers (nc,3) or None (dense: nothing masked),
        region_id (nc,N) int32, keep (B,) int64 bit masks or None, cloud_of (B,) int32 or None.
        -> logits (B, num_classes) [, packed trans_feat (B,4096)] [, crt_points (B,1024) int32: the point that attains each
        pooled channel's maximum, index N = the centre (iq_pointnet_coalitions_crt)]."""
        nc = clouds.shape[0]
        n = clouds.shape[2] if channel_first else clouds.shape[1]
        b = keep.shape[0] if keep is not None else (cloud_of.shape[0] if cloud_of is not None else nc)
        r = int(num_regions)
        for t, dt, nm in ((clouds, torch.float32, "clouds"), (region_id, torch.int32, "region_id")):
            if not t.is_cuda or t.dtype != dt or not t.is_contiguous():
                raise _lib.IqError("%s must be a contiguous %s GPU tensor" % (nm, dt))
        for t, dt, nm in ((centers, torch.float32, "centers"), (keep, torch.int64, "keep"),
                          (cloud_of, torch.int32, "cloud_of")):
            if t is not None and (not t.is_cuda or t.dtype != dt or not t.is_contiguous()):
                raise _lib.IqError("%s must be a contiguous %s GPU tensor" % (nm, dt))
        logits = torch.empty((b, self.weights.num_classes), dtype=torch.float32, device=self.device)
        tfp = torch.empty((b, 4096), dtype=torch.float32, device=self.device) if return_trans_feat else None
        crt = torch.empty((b, 1024), dtype=torch.int32, device=self.device) if return_crt else None
        need = self.lib.iq_pointnet_workspace_bytes(b, nc, n, r)
        ws = self._workspace(need)
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
        rc = self.lib.iq_pointnet_coalitions_crt(ctypes.byref(self.weights.struct), p(clouds), p(centers), p(region_id),
                                                 p(keep), p(cloud_of), p(logits), p(tfp), p(crt), p(ws), ws.numel(),
                                                 b, nc, n, r, int(channel_first),
                                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "iq_pointnet_coalitions")
        out = (logits,) + ((tfp,) if return_trans_feat else ()) + ((crt,) if return_crt else ())
        return out if len(out) > 1 else logits

    def forward(self, x):
        """Dense forward, x (B,3,N) -> (logits, trans_feat (B,64,64), crt_points (B,1024) int64)."""
        b, _, n = x.shape
        rid = torch.zeros((b, n), dtype=torch.int32, device=self.device)
        logits, tfp, crt = self.coalition_logits(x.contiguous(), None, rid, None, None, num_regions=1,
                                                 channel_first=True, return_trans_feat=True, return_crt=True)
        trans_feat = tfp.index_select(1, self.weights.unpack_index).reshape(b, 64, 64) if self.weights.feature_transform else None
        return logits, trans_feat, crt.long()   # trans_feat is None without a feature STN, as in models/pointnet.py:78


def _param_holder_stn(k):
    m = nn.Module()
    m.conv1, m.conv2, m.conv3 = nn.Conv1d(k, 64, 1), nn.Conv1d(64, 128, 1), nn.Conv1d(128, 1024, 1)
    m.fc1, m.fc2, m.fc3 = nn.Linear(1024, 512), nn.Linear(512, 256), nn.Linear(256, k * k)
    for j, c in enumerate((64, 128, 1024, 512, 256), start=1):
        setattr(m, "bn%d" % j, nn.BatchNorm1d(c))
    return m


class PointNetCls(nn.Module):
    """Parameter container with the reference's state-dict layout; forward runs on the HIP path."""

    def __init__(self, args=None):
        super().__init__()
        self.args = args
        dataset = getattr(args, "dataset", "modelnet10")
        self.output_channels = 40 if dataset == "modelnet40" else 10  # models/pointnet.py:95-98
        self.feature_transform = bool(getattr(args, "feature_transform", True))   # models/pointnet.py:99 (the scripts set True)
        feat = nn.Module()
        feat.stn = _param_holder_stn(3)
        feat.conv1, feat.conv2, feat.conv3 = nn.Conv1d(3, 64, 1), nn.Conv1d(64, 128, 1), nn.Conv1d(128, 1024, 1)
        feat.bn1, feat.bn2, feat.bn3 = nn.BatchNorm1d(64), nn.BatchNorm1d(128), nn.BatchNorm1d(1024)
        if self.feature_transform:
            feat.fstn = _param_holder_stn(64)
        self.feat = feat
        self.fc1, self.fc2, self.fc3 = nn.Linear(1024, 512), nn.Linear(512, 256), nn.Linear(256, self.output_channels)
        self.bn1, self.bn2 = nn.BatchNorm1d(512), nn.BatchNorm1d(256)
        self._engine = None

    # any parameter change invalidates the packed image
    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    def engine(self):
        if self.training:
            raise _lib.IqError("the HIP PointNet path implements eval mode only (BN running stats, no dropout)")
        if self._engine is None:
            dev = self.fc3.weight.device
            self._engine = PointNetEngine(self.state_dict(), dev)
        return self._engine

    def forward(self, x):
        """x (B,3,N) -> (logits, trans_feat, crt_points), the reference's tuple (models/pointnet.py:109-115): crt_points (B,1024)
        int64 = the point that attains each pooled channel's maximum (:83), from the arg-max variant of the trunk kernel.
        The coalition path (every hot-path caller discards crt_points, tools/final_common.py:36-37) does not compute it."""
        return self.engine().forward(x)

    def coalition_logits(self, clouds, centers, region_id, keep, cloud_of=None, num_regions=None, validate=True):
        """``validate``: check region_id against [0, num_regions) first (one stream sync); the drivers validate the ids
        once per cloud on the host and pass False."""
        if validate:
            hip_ops.check_index_range(region_id, 0, int(num_regions), "region_id")
        return self.engine().coalition_logits(clouds, centers, region_id, keep, cloud_of, num_regions=num_regions)
