"""torch-tensor front end of the C ABI (include/iq.h).  PyTorch only owns the device memory and
the stream; all arithmetic happens in libiq_hip.so.  Every function raises if a tensor is not on
a GPU - there is no CPU fallback."""
import ctypes

import numpy as np
import torch

from . import _lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.IqError("%s must be a CUDA/HIP tensor (the HIP path has no CPU fallback)" % name)
    if t.dtype != dtype:
        raise _lib.IqError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise _lib.IqError("%s must be contiguous" % name)
    return ctypes.c_void_p(t.data_ptr())


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def as_i32(x, device):
    """int64 ndarray/tensor -> int32 device tensor (indices are int32 at the ABI)."""
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    return x.to(device=device, dtype=torch.int32).contiguous()


def check_index_range(t, lo, hi, what):
    """Raise IqError unless every entry of the int32 device tensor ``t`` lies in [lo, hi) (iq_check_index_range; the one
    call that synchronises the stream).  The kernels never fault on a bad id, but their results are then meaningless."""
    lib = _lib.load()
    if t.numel() == 0:
        return
    scratch = torch.empty((1,), dtype=torch.int32, device=t.device)
    rc = lib.iq_check_index_range(_dev(t, torch.int32, what), t.numel(), int(lo), int(hi), _p(scratch), _stream())
    if rc != 0:
        msg = lib.iq_last_error()
        raise _lib.IqError("%s: %s" % (what, msg.decode() if msg else "index out of range"))


def check_host_indices(arr, lo, hi, what):
    """The same check for indices that are still on the host (region_id.npy, all_orders.npy, pair lists): free."""
    a = np.asarray(arr)
    if a.size and (a.min() < lo or a.max() >= hi):
        bad = int(np.flatnonzero((a.reshape(-1) < lo) | (a.reshape(-1) >= hi))[0])
        raise _lib.IqError("%s: index at position %d is outside [%d, %d)" % (what, bad, lo, hi))


def region_ids(region_id, device, num_regions):
    """Region ids of one cloud (ndarray from region_id.npy, or a tensor) -> validated int32 device tensor."""
    if isinstance(region_id, np.ndarray):
        check_host_indices(region_id, 0, num_regions, "region_id")
        return as_i32(region_id, device)
    t = as_i32(region_id, device)
    check_index_range(t, 0, num_regions, "region_id")
    return t


def region_bitmask(regions):
    """Iterable of region ids -> python int bit mask."""
    m = 0
    for r in regions:
        m |= 1 << int(r)
    return m


def masks_to_tensor(masks, device):
    """uint64 bit masks as an int64-typed device tensor (same bits)."""
    arr = np.asarray(masks, dtype=np.uint64).view(np.int64)
    return torch.from_numpy(np.ascontiguousarray(arr)).to(device)


def mt_state_to_device(device, state=None):
    """NumPy legacy generator state (np.random.get_state(); default: the GLOBAL generator's) -> (625,) int32 device tensor
    holding the 624 key words and the position, the form iq_sample_permutations advances."""
    st = np.random.get_state() if state is None else state
    if st[0] != "MT19937":
        raise _lib.IqError("the reference's sampling stream is NumPy's legacy MT19937 generator, got %r" % (st[0],))
    if not 0 <= int(st[2]) <= 624:     # np.random.set_state accepts any position; the generator itself never leaves 0..624
        raise _lib.IqError("MT19937 state with position %d outside 0..624" % int(st[2]))
    words = np.empty(625, dtype=np.uint32)
    words[:624], words[624] = st[1], st[2]
    return torch.from_numpy(words.view(np.int32)).to(device)


def mt_state_to_host(mt_state, set_global=True):
    """The advanced state back on the host (one device->host copy); ``set_global`` installs it as NumPy's global generator
    so that whatever the host draws next continues the reference's stream."""
    words = mt_state.cpu().numpy().view(np.uint32)
    if int(words[624]) > 624:          # iq_sample_permutations marks a state it could not draw all permutations from
        raise _lib.IqError("the device sampler did not complete (state position word %#x): the permutations of the last "
                           "sample_permutations call on this state are not valid" % int(words[624]))
    cur = np.random.get_state()          # the device drew 32-bit words only: a cached Gaussian of the host generator stays as it is
    st = ("MT19937", words[:624].copy(), int(words[624]), cur[3], cur[4])
    if set_global:
        np.random.set_state(st)
    return st


def sample_permutations(mt_state, num_samples, num_regions):
    """iq_sample_permutations: (S,R) int32 permutations continuing the generator ``mt_state`` (advanced in place, no sync)."""
    lib = _lib.load()
    orders = torch.empty((int(num_samples), int(num_regions)), dtype=torch.int32, device=mt_state.device)
    _lib.check(lib.iq_sample_permutations(_dev(mt_state, torch.int32, "mt_state"), _p(orders), int(num_samples), int(num_regions),
                                          _stream()), "iq_sample_permutations")
    return orders


def prefix_keep_masks(orders):
    """orders (S,R) i32 -> (S*(R+1),) int64-typed keep masks of the prefix coalitions (tools/final_common.py:56-60)."""
    lib = _lib.load()
    s, r = orders.shape
    keep = torch.empty((s * (r + 1),), dtype=torch.int64, device=orders.device)
    _lib.check(lib.iq_prefix_keep_masks(_dev(orders, torch.int32, "orders"), _p(keep), s, r, _stream()), "iq_prefix_keep_masks")
    return keep


def context_keep_masks(pairs, contexts):
    """pairs (P,2) i32, contexts (P,C,m) i32 -> (4*P*C,) int64-typed keep masks, rows S+{i,j}, S+{i}, S+{j}, S per context
    (final_point_binary_interaction_logits.py:45-52)."""
    lib = _lib.load()
    p, c, m = contexts.shape
    keep = torch.empty((4 * p * c,), dtype=torch.int64, device=pairs.device)
    _lib.check(lib.iq_context_keep_masks(_dev(pairs, torch.int32, "pairs"), _dev(contexts, torch.int32, "contexts") if m else ctypes.c_void_p(0),
                                         _p(keep), p, c, m, _stream()), "iq_context_keep_masks")
    return keep


def mask_shapley(cloud, region_id, orders, center, channel_first=False):
    """cloud (N,3) f32, region_id (N,) i32, orders (bs,R) i32, center (3,) f32 ->
    (bs*(R+1), N, 3) or (bs*(R+1), 3, N)."""
    lib = _lib.load()
    n = cloud.shape[0]
    bs, r = orders.shape
    shape = (bs * (r + 1), 3, n) if channel_first else (bs * (r + 1), n, 3)
    out = torch.empty(shape, dtype=torch.float32, device=cloud.device)
    _lib.check(lib.iq_mask_shapley(_dev(cloud, torch.float32, "cloud"), _dev(region_id, torch.int32, "region_id"),
                                   _dev(orders, torch.int32, "orders"), _dev(center, torch.float32, "center"),
                                   _p(out), n, r, bs, int(channel_first), _stream()), "iq_mask_shapley")
    return out


def mask_interaction(cloud, region_id, pairs, ctx_mask, center, num_regions):
    """pairs (nb,2) i32, ctx_mask (nb,) i64 (bit masks) -> (4*nb, 3, N)."""
    lib = _lib.load()
    n = cloud.shape[0]
    nb = pairs.shape[0]
    out = torch.empty((4 * nb, 3, n), dtype=torch.float32, device=cloud.device)
    _lib.check(lib.iq_mask_interaction(_dev(cloud, torch.float32, "cloud"), _dev(region_id, torch.int32, "region_id"),
                                       _dev(pairs, torch.int32, "pairs"), _dev(ctx_mask, torch.int64, "ctx_mask"),
                                       _dev(center, torch.float32, "center"), _p(out), n, num_regions, nb,
                                       _stream()), "iq_mask_interaction")
    return out


def mask_coalitions(cloud, region_id, keep, center, channel_first=False):
    lib = _lib.load()
    n = cloud.shape[0]
    b = keep.shape[0]
    shape = (b, 3, n) if channel_first else (b, n, 3)
    out = torch.empty(shape, dtype=torch.float32, device=cloud.device)
    _lib.check(lib.iq_mask_coalitions(_dev(cloud, torch.float32, "cloud"), _dev(region_id, torch.int32, "region_id"),
                                      _dev(keep, torch.int64, "keep"), _dev(center, torch.float32, "center"),
                                      _p(out), n, b, int(channel_first), _stream()), "iq_mask_coalitions")
    return out


def reward(logits, label, modified=True):
    lib = _lib.load()
    b, c = logits.shape
    v = torch.empty((b,), dtype=torch.float32, device=logits.device)
    _lib.check(lib.iq_reward(_dev(logits, torch.float32, "logits"), int(label), int(modified), _p(v), b, c,
                             _stream()), "iq_reward")
    return v


def shapley_accum(v, orders, snap_counts=None):
    """v (S*(R+1),) f32, orders (S,R) i32 -> (phi_sum (R,) f64, sv_rows (S,R) f64, snaps or None)."""
    lib = _lib.load()
    s, r = orders.shape
    dev = v.device
    sv_rows = torch.zeros((max(s, 1), r), dtype=torch.float64, device=dev)
    phi = torch.empty((r,), dtype=torch.float64, device=dev)
    snaps = None
    counts = None
    n_snap = 0
    if snap_counts is not None and len(snap_counts) > 0:
        counts = torch.tensor(list(snap_counts), dtype=torch.int32, device=dev)
        n_snap = counts.numel()
        snaps = torch.zeros((n_snap, r), dtype=torch.float64, device=dev)
    _lib.check(lib.iq_shapley_accum(_dev(v, torch.float32, "v"), _dev(orders, torch.int32, "orders"), _p(sv_rows),
                                    _p(phi), _p(counts), n_snap, _p(snaps), r, s, _stream()), "iq_shapley_accum")
    return phi, sv_rows[:s], snaps


def interaction_reduce(v):
    lib = _lib.load()
    n = v.numel() // 4
    out = torch.empty((n,), dtype=torch.float32, device=v.device)
    _lib.check(lib.iq_interaction_reduce(_dev(v, torch.float32, "v"), _p(out), n, _stream()), "iq_interaction_reduce")
    return out


def region_assign(cloud, fps_idx):
    lib = _lib.load()
    n = cloud.shape[0]
    r = fps_idx.shape[0]
    out = torch.empty((n,), dtype=torch.int32, device=cloud.device)
    _lib.check(lib.iq_region_assign(_dev(cloud, torch.float32, "cloud"), _dev(fps_idx, torch.int32, "fps_idx"),
                                    _p(out), n, r, _stream()), "iq_region_assign")
    return out


def fps(xyz, npoint):
    """xyz (B,N,3) f32 -> (B,npoint) i32."""
    lib = _lib.load()
    b, n, _ = xyz.shape
    out = torch.empty((b, npoint), dtype=torch.int32, device=xyz.device)
    _lib.check(lib.iq_fps(_dev(xyz, torch.float32, "xyz"), _p(out), b, n, npoint, _stream()), "iq_fps")
    return out


def ball_query(xyz, new_xyz, radius, nsample):
    """xyz (B,N,3), new_xyz (B,S,3) f32 -> (B,S,nsample) i32 (models/pointnet2.py:70-91)."""
    lib = _lib.load()
    b, n, _ = xyz.shape
    s = new_xyz.shape[1]
    out = torch.empty((b, s, nsample), dtype=torch.int32, device=xyz.device)
    _lib.check(lib.iq_ball_query(_dev(xyz, torch.float32, "xyz"), _dev(new_xyz, torch.float32, "new_xyz"),
                                 ctypes.c_float(radius), nsample, _p(out), b, n, s, _stream()), "iq_ball_query")
    return out


def knn(x, k=20):
    """x (B,N,C) row-major f32, C in {3,64,128} -> (B,N,k) i32 neighbour sets (models/dgcnn.py:12-18)."""
    lib = _lib.load()
    b, n, c = x.shape
    out = torch.empty((b, n, k), dtype=torch.int32, device=x.device)
    tmp = torch.empty((b * n * (84 + (6 * c if c in (64, 128) else 0)) + 16 * b + 16384,), dtype=torch.uint8, device=x.device)
    _lib.check(lib.iq_knn(_dev(x, torch.float32, "x"), _p(out), _p(tmp), tmp.numel(), b, n, c, k, _stream()), "iq_knn")
    return out


SMOOTHNESS_MODES = ("linearity", "planarity", "scattering")


def smoothness_enum(cloud, region_id, num_regions, mode, objective, step=1e-3, enum_step=0.05, var_threshold=0.003,
                    dist_threshold=0.03, stop_ratio=0.5, epochs=50, max_iteration=100, origin=None, project_to_bound=False):
    """final_smoothness_center_enum_all.py:183-242,303-335 for all regions and epochs in one launch.
    cloud (N,3) f32, region_id (N,) i32 -> dict(data (E,N,3) f32, smoothness (E,R) f32, var (E,R,3) f32,
    orig (R,4) f32, stop_epoch (R,) i32); see include/iq.h.  ``origin`` (N,3): restart from a deformed ``cloud``."""
    lib = _lib.load()
    n, r, e = cloud.shape[0], int(num_regions), int(epochs)
    dev = cloud.device
    out = {"data": torch.empty((e, n, 3), dtype=torch.float32, device=dev),
           "smoothness": torch.empty((e, r), dtype=torch.float32, device=dev),
           "var": torch.empty((e, r, 3), dtype=torch.float32, device=dev),
           "orig": torch.empty((r, 4), dtype=torch.float32, device=dev),
           "stop_epoch": torch.empty((r,), dtype=torch.int32, device=dev)}
    prm = _lib.SmoothnessParams(step, enum_step, var_threshold, dist_threshold, stop_ratio, e, int(max_iteration), int(bool(project_to_bound)), 0)
    if mode not in SMOOTHNESS_MODES or objective not in ("inc", "dec"):
        raise _lib.IqError("smoothness_enum: mode %r / objective %r" % (mode, objective))
    org = _dev(origin, torch.float32, "origin") if origin is not None else ctypes.c_void_p(0)
    _lib.check(lib.iq_smoothness_enum(_dev(cloud, torch.float32, "cloud"), org, _dev(region_id, torch.int32, "region_id"), n, r,
                                      SMOOTHNESS_MODES.index(mode), 1 if objective == "inc" else -1, ctypes.byref(prm),
                                      _p(out["data"]), _p(out["smoothness"]), _p(out["var"]), _p(out["orig"]),
                                      _p(out["stop_epoch"]), _stream()), "iq_smoothness_enum")
    return out


class PackedLinear:
    """A (cout,cin) weight + bias in the library's MFMA fragment order (iq_pack_weight) on the device.  bf3: also as three bf16
    terms (iq_pack_weight_bf3) - a layer with cout % 256 == 0 (or 64: all but the last 64 columns) and cin >= 32 then takes its products on the bf16 matrix pipe,
    float32-exact (include/iq.h: iq_dense_layer.w_bf3)."""

    def __init__(self, weight, bias, device, bf3=False):
        lib = _lib.load()
        w = np.ascontiguousarray(weight, dtype=np.float32)
        self.cout, self.cin = w.shape
        packed = np.empty(lib.iq_packed_floats(self.cout, self.cin), dtype=np.float32)
        _lib.check(lib.iq_pack_weight(w.ctypes.data, packed.ctypes.data, self.cout, self.cin), "iq_pack_weight")
        bp = np.zeros(lib.iq_padded_cout(self.cout), dtype=np.float32)
        bp[:self.cout] = np.asarray(bias, dtype=np.float32)
        self.w, self.b = torch.from_numpy(packed).to(device), torch.from_numpy(bp).to(device)
        self.w3 = None
        if bf3:
            terms = np.empty(lib.iq_packed_bf3_elems(self.cout, self.cin), dtype=np.uint16)
            _lib.check(lib.iq_pack_weight_bf3(w.ctypes.data, terms.ctypes.data, self.cout, self.cin), "iq_pack_weight_bf3")
            self.w3 = torch.from_numpy(terms.view(np.int16)).to(device)
        self.struct = _lib.DenseLayer(self.w.data_ptr(), self.b.data_ptr(), self.cin, self.cout,
                                      self.w3.data_ptr() if self.w3 is not None else None)


def linear(x, layer, act=0):
    """x (M,cin) f32 -> act(x W^T + b) (M,cout); act 0 none, 1 ReLU, 2 LeakyReLU(0.2)."""
    lib = _lib.load()
    m = x.shape[0]
    out = torch.empty((m, layer.cout), dtype=torch.float32, device=x.device)
    _lib.check(lib.iq_linear(_dev(x, torch.float32, "x"), x.shape[1], ctypes.byref(layer.struct), _p(out), layer.cout, m, int(act),
                             _stream()), "iq_linear")
    return out
