"""ctypes binding of libiq_hip.so (include/iq.h).  cffi is not installed in the target image
(SURVEY.md §0), so the thin C-ABI layer the north star asks for is bound with ctypes.

There is NO fallback: if the shared library is missing or a call fails, an exception is raised.
"""
import ctypes
import os

from . import build as _build

_HERE = os.path.dirname(os.path.abspath(__file__))

c_f32p = ctypes.c_void_p
c_ptr = ctypes.c_void_p


class IqError(RuntimeError):
    pass


class DenseLayer(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p), ("b", ctypes.c_void_p), ("cin", ctypes.c_int32), ("cout", ctypes.c_int32),
                ("w_bf3", ctypes.c_void_p)]


class PointNetWeights(ctypes.Structure):
    _fields_ = [
        ("stn_in", ctypes.c_void_p),
        ("stn_c2", DenseLayer), ("stn_c3", DenseLayer), ("stn_fc1", DenseLayer), ("stn_fc2", DenseLayer),
        ("stn_fc3", DenseLayer),
        ("feat_in", ctypes.c_void_p),
        ("fstn_c1", DenseLayer), ("fstn_c2", DenseLayer), ("fstn_c3", DenseLayer), ("fstn_fc1", DenseLayer),
        ("fstn_fc2", DenseLayer), ("fstn_fc3", DenseLayer),
        ("feat_c2", DenseLayer), ("feat_c3", DenseLayer), ("cls_fc1", DenseLayer), ("cls_fc2", DenseLayer),
        ("cls_fc3", DenseLayer),
        ("fstn_c3_bf3", ctypes.c_void_p), ("feat_c3_bf3", ctypes.c_void_p),
        ("fstn_c2_bf3", ctypes.c_void_p), ("feat_c2_bf3", ctypes.c_void_p),
    ]


class Pn2Scale(ctypes.Structure):
    _fields_ = [("w1x", ctypes.c_void_p), ("l2", DenseLayer), ("l3", DenseLayer), ("radius", ctypes.c_float),
                ("nsample", ctypes.c_int32)]


class PointNet2Weights(ctypes.Structure):
    _fields_ = [("sa1", Pn2Scale * 3), ("sa2_u", DenseLayer), ("sa2", Pn2Scale * 3),
                ("sa3_l1", DenseLayer), ("sa3_l2", DenseLayer), ("sa3_l3", DenseLayer),
                ("fc1", DenseLayer), ("fc2", DenseLayer), ("fc3", DenseLayer),
                ("sa2_l2_bf3", ctypes.c_void_p * 3), ("sa2_l3_bf3", ctypes.c_void_p * 3)]


class PointConvSa(ctypes.Structure):
    _fields_ = [("w1x", ctypes.c_void_p), ("u", DenseLayer), ("l2", DenseLayer), ("l3", DenseLayer),
                ("densitynet", ctypes.c_void_p), ("weightnet", ctypes.c_void_p), ("linear", DenseLayer),
                ("bandwidth", ctypes.c_float), ("nsample", ctypes.c_int32)]


class PointConvWeights(ctypes.Structure):
    _fields_ = [("sa", PointConvSa * 3), ("fc1", DenseLayer), ("fc2", DenseLayer), ("fc3", DenseLayer),
                ("sa2_l2_bf3", ctypes.c_void_p), ("sa2_l3_bf3", ctypes.c_void_p)]


class DgcnnWeights(ctypes.Structure):
    _fields_ = [("pq", DenseLayer * 4), ("conv5", DenseLayer), ("fc1", DenseLayer), ("fc2", DenseLayer),
                ("fc3", DenseLayer), ("k", ctypes.c_int32), ("reserved", ctypes.c_int32), ("conv5_bf3", ctypes.c_void_p)]


class SmoothnessParams(ctypes.Structure):
    _fields_ = [("step", ctypes.c_double), ("enum_step", ctypes.c_double), ("var_threshold", ctypes.c_double),
                ("dist_threshold", ctypes.c_double), ("stop_ratio", ctypes.c_double), ("epochs", ctypes.c_int32),
                ("max_iteration", ctypes.c_int32), ("project_to_bound", ctypes.c_int32), ("reserved", ctypes.c_int32)]


_I = ctypes.c_int
_P = ctypes.c_void_p
_SZ = ctypes.c_size_t

# name -> (restype, argtypes); every symbol include/iq.h declares
SIGNATURES = {
    "iq_version": (_I, []),
    "iq_last_error": (ctypes.c_char_p, []),
    "iq_mask_shapley": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "iq_mask_interaction": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "iq_mask_coalitions": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "iq_check_index_range": (_I, [_P, _SZ, _I, _I, _P, _P]),
    "iq_sample_permutations": (_I, [_P, _P, _I, _I, _P]),
    "iq_prefix_keep_masks": (_I, [_P, _P, _I, _I, _P]),
    "iq_context_keep_masks": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "iq_reward": (_I, [_P, _I, _I, _P, _I, _I, _P]),
    "iq_shapley_accum": (_I, [_P, _P, _P, _P, _P, _I, _P, _I, _I, _P]),
    "iq_interaction_reduce": (_I, [_P, _P, _I, _P]),
    "iq_region_assign": (_I, [_P, _P, _P, _I, _I, _P]),
    "iq_fps": (_I, [_P, _P, _I, _I, _I, _P]),
    "iq_smoothness_enum": (_I, [_P, _P, _P, _I, _I, _I, _I, ctypes.POINTER(SmoothnessParams), _P, _P, _P, _P, _P, _P]),
    "iq_linear": (_I, [_P, _I, ctypes.POINTER(DenseLayer), _P, _I, _I, _I, _P]),
    "iq_packed_floats": (_SZ, [_I, _I]),
    "iq_padded_cout": (_I, [_I]),
    "iq_pack_weight": (_I, [_P, _P, _I, _I]),
    "iq_packed_bf3_elems": (_SZ, [_I, _I]),
    "iq_pack_weight_bf3": (_I, [_P, _P, _I, _I]),
    "iq_pack_fstn_fc3": (_I, [_P, _P, _P, _P, _P]),
    "iq_pointnet_workspace_bytes": (_SZ, [_I, _I, _I, _I]),
    "iq_pointnet_coalitions": (_I, [ctypes.POINTER(PointNetWeights), _P, _P, _P, _P, _P, _P, _P, _P, _SZ,
                                    _I, _I, _I, _I, _I, _P]),
    "iq_pointnet_coalitions_crt": (_I, [ctypes.POINTER(PointNetWeights), _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ,
                                        _I, _I, _I, _I, _I, _P]),
    "iq_pointnet_flops_per_coalition": (ctypes.c_double, [_I]),
    "iq_ball_query": (_I, [_P, _P, ctypes.c_float, _I, _P, _I, _I, _I, _P]),
    "iq_pointnet2_workspace_bytes": (_SZ, [_I]),
    "iq_pointnet2_forward": (_I, [ctypes.POINTER(PointNet2Weights), _P, _P, _P, _SZ, _I, _I, _P]),
    "iq_pointnet2_coalitions_workspace_bytes": (_SZ, [_I, _I, _I]),
    "iq_pointnet2_coalitions": (_I, [ctypes.POINTER(PointNet2Weights), _P, _P, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _P]),
    "iq_knn": (_I, [_P, _P, _P, _SZ, _I, _I, _I, _I, _P]),
    "iq_dgcnn_workspace_bytes": (_SZ, [_I, _I]),
    "iq_dgcnn_forward": (_I, [ctypes.POINTER(DgcnnWeights), _P, _P, _P, _SZ, _I, _I, _I, _P]),
    "iq_dgcnn_coalitions": (_I, [ctypes.POINTER(DgcnnWeights), _P, _P, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _P]),
    "iq_pointconv_workspace_bytes": (_SZ, [_I, _I]),
    "iq_pointconv_forward": (_I, [ctypes.POINTER(PointConvWeights), _P, _P, _P, _SZ, _I, _I, _P]),
    "iq_pointconv_coalitions_workspace_bytes": (_SZ, [_I, _I, _I]),
    "iq_pointconv_coalitions": (_I, [ctypes.POINTER(PointConvWeights), _P, _P, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _P]),
    "iq_pointconv_tables_bytes": (_SZ, [_I, _I]),
    "iq_pointconv_coalitions_cached": (_I, [ctypes.POINTER(PointConvWeights), _P, _P, _P, _P, _P, _P, _P, _SZ, _I, _I, _I,
                                          ctypes.POINTER(ctypes.c_int), _P]),
    "iq_profile_enable": (_I, [_I]),
    "iq_set_tuning": (_I, [_I, _I]),
    "iq_debug_chain_occupancy": (_I, []),
    "iq_debug_stamps": (_I, [_I, ctypes.POINTER(ctypes.c_ulonglong)]),
    "iq_debug_knn_counters": (_I, [ctypes.POINTER(ctypes.c_ulonglong)]),
    "iq_profile_read_work": (_I, [_I, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)]),
    "iq_profile_read": (_I, [_I, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]),
    "iq_debug_mfma_sustained": (_I, [ctypes.c_double, _P, _SZ, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), _P]),
}

_lib = None
ABI_VERSION = 102   # IQ_ABI_VERSION of include/iq.h these struct layouts were written for


def lib_path():
    """The product library; IQ_LIBPATH names another BUILD of the same sources (A/B runs of tools/, e.g. lib_packed_ab/)."""
    return os.environ.get("IQ_LIBPATH") or _build.LIBPATH


def load():
    """Load libiq_hip.so (never builds implicitly on the GPU box: the .so ships with the repo)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch bundles its own libamdhip64.so.7; load it FIRST so that libiq_hip.so binds to the same
    # HIP runtime instance (two runtimes in one process do not share devices, streams or memory).
    import torch  # noqa: F401
    path = lib_path()
    if not os.path.exists(path):
        raise IqError("libiq_hip.so not found at %s - run `python -m interpret_quality_amd.build` "
                      "(there is no CPU or PyTorch fallback for the HIP path)" % path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.iq_version() != ABI_VERSION:
        raise IqError("%s reports ABI version %d, these bindings are written for %d - rebuild it (python -m interpret_quality_amd.build "
                      "--force)" % (path, lib.iq_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().iq_last_error()
        raise IqError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))
