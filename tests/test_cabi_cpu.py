"""CPU: the C-ABI library loads, exports every symbol include/iq.h declares, and its host-side
helpers (weight packing, sizes) behave.  No compute call is made - there is no GPU here."""
import os
import re

import numpy as np
import pytest

from interpret_quality_amd import _lib, build

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build(verbose=False)  # hipcc cross-compiles gfx950 without a GPU
    return _lib.load()


def test_header_symbols_exported_and_bound(lib):
    declared = {}
    for h in ("iq.h", "iq_debug.h"):      # the drop-in surface | diagnostics (profiler, experiment knobs, debug counters)
        header = open(os.path.join(REPO, "include", h)).read()
        header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)  # drop comments
        declared[h] = set(re.findall(r"\b(iq_[a-z0-9_]+)\s*\(", header))
        assert declared[h], "no declarations parsed in %s" % h
    both = declared["iq.h"] | declared["iq_debug.h"]
    assert not (declared["iq.h"] & declared["iq_debug.h"])
    assert both == set(_lib.SIGNATURES), (both ^ set(_lib.SIGNATURES))
    for name in both:
        assert hasattr(lib, name), name
    # nothing diagnostic is left on the drop-in surface
    assert not [n for n in declared["iq.h"] if n.startswith(("iq_debug_", "iq_profile_", "iq_set_tuning"))]
    assert declared["iq_debug.h"] == {n for n in _lib.SIGNATURES if n.startswith(("iq_debug_", "iq_profile_", "iq_set_tuning"))}
    m = re.search(r"#define IQ_ABI_VERSION (\d+)", open(os.path.join(REPO, "include", "iq.h")).read())
    assert lib.iq_version() == int(m.group(1)) >= 101


def test_pack_weight_layout(lib):
    rng = np.random.default_rng(0)
    cout, cin = 40, 16
    w = rng.standard_normal((cout, cin)).astype(np.float32)
    assert lib.iq_padded_cout(cout) == 64
    out = np.empty(lib.iq_packed_floats(cout, cin), dtype=np.float32)
    assert lib.iq_pack_weight(w.ctypes.data, out.ctypes.data, cout, cin) == 0
    kb_n = cin // 8
    for nt in range(2):
        for kb in range(kb_n):
            for lane in range(64):
                for j in range(4):
                    n, k = nt * 32 + (lane & 31), 8 * kb + 4 * (lane >> 5) + j
                    want = w[n, k] if n < cout else 0.0
                    assert out[((nt * kb_n + kb) * 64 + lane) * 4 + j] == want
    assert lib.iq_pack_weight(w.ctypes.data, out.ctypes.data, cout, 12) != 0  # cin % 8
    assert b"cin" in lib.iq_last_error()


def test_pack_fstn_fc3_is_a_row_permutation_plus_identity(lib):
    rng = np.random.default_rng(1)
    w = rng.standard_normal((4096, 256)).astype(np.float32)
    b = rng.standard_normal(4096).astype(np.float32)
    ow = np.empty(lib.iq_packed_floats(4096, 256), dtype=np.float32)
    ob = np.empty(4096, dtype=np.float32)
    perm = np.empty(4096, dtype=np.int32)
    assert lib.iq_pack_fstn_fc3(w.ctypes.data, b.ctypes.data, ow.ctypes.data, ob.ctypes.data, perm.ctypes.data) == 0
    assert sorted(perm.tolist()) == list(range(4096))
    k, n = perm // 64, perm % 64
    np.testing.assert_array_equal(ob, b[perm] + (k == n).astype(np.float32))
    # packed rows = permuted rows run through iq_pack_weight
    ref = np.empty_like(ow)
    wp = np.ascontiguousarray(w[perm])
    assert lib.iq_pack_weight(wp.ctypes.data, ref.ctypes.data, 4096, 256) == 0
    np.testing.assert_array_equal(ow, ref)


def test_workspace_and_flops(lib):
    assert lib.iq_pointnet_workspace_bytes(0, 1, 1024, 32) > 0
    small = lib.iq_pointnet_workspace_bytes(100, 1, 1024, 32)
    big = lib.iq_pointnet_workspace_bytes(3300, 1, 1024, 32)
    assert big > small > 100 * 4096 * 4
    assert abs(lib.iq_pointnet_flops_per_coalition(1024) / 0.879e9 - 1) < 0.01


def test_hip_path_refuses_cpu_tensors():
    import torch
    from interpret_quality_amd import hip_ops
    with pytest.raises(_lib.IqError):
        hip_ops.reward(torch.zeros(4, 10), 0)


@pytest.mark.parametrize("cin", [48, 40, 64])
def test_pack_weight_bf3_terms_add_up_to_the_float32_weight_exactly(cin):
    """iq_pack_weight_bf3 (host): three bf16 terms in the fragment order of v_mfma_f32_32x32x16_bf16.  h + m + l reproduces every
    float32 weight exactly (24 mantissa bits in three terms), the image's k range is cin rounded up to a multiple of 32, padded rows
    and padded k columns are zero, and the layout is the documented one."""
    import numpy as np
    from interpret_quality_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(0)
    cout, kp = 40, (cin + 31) // 32 * 32
    w = (rng.standard_normal((cout, cin)) * np.exp(rng.uniform(-8, 8, size=(cout, cin)))).astype(np.float32)
    n = lib.iq_packed_bf3_elems(cout, cin)
    assert n == 3 * 64 * kp
    out = np.empty(n, dtype=np.uint16)
    assert lib.iq_pack_weight_bf3(w.ctypes.data, out.ctypes.data, cout, cin) == 0
    terms = (out.astype(np.uint32) << 16).view(np.float32).reshape(3, 2, kp // 16, 64, 8)     # [term][n-tile][k-step][lane][j]
    lane = np.arange(64)
    got = np.zeros((64, kp), dtype=np.float64)
    for nt in range(2):
        for ks in range(kp // 16):
            for j in range(8):
                got[nt * 32 + (lane & 31), 16 * ks + 8 * (lane >> 5) + j] = terms[:, nt, ks, :, j].astype(np.float64).sum(axis=0)
    assert np.array_equal(got[:cout, :cin], w.astype(np.float64)) and not got[cout:].any() and not got[:, cin:].any()
    assert np.abs(terms[1]).max() <= np.abs(terms[0]).max() * 2.0 ** -8 and np.abs(terms[2]).max() <= np.abs(terms[0]).max() * 2.0 ** -16


def test_ctypes_structures_have_the_layout_of_the_header(tmp_path):
    """Every struct of include/iq.h that crosses the boundary, field by field: sizeof and offsetof as gcc sees the header against
    the ctypes mirror in _lib.py (a field added on one side only would shift every pointer behind it, silently)."""
    import ctypes
    import subprocess
    from interpret_quality_amd import _lib
    pairs = {"iq_dense_layer": _lib.DenseLayer, "iq_pointnet_weights": _lib.PointNetWeights, "iq_pn2_scale": _lib.Pn2Scale,
             "iq_pointnet2_weights": _lib.PointNet2Weights, "iq_pointconv_sa": _lib.PointConvSa,
             "iq_pointconv_weights": _lib.PointConvWeights, "iq_dgcnn_weights": _lib.DgcnnWeights,
             "iq_smoothness_params": _lib.SmoothnessParams}
    # (iq_debug.h - the diagnostics - must be plain C too and pulls iq.h in; IQ_ABI_VERSION is what iq_version() returns)
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "iq_debug.h"', '#if IQ_ABI_VERSION < 101', '#error "ABI version"', '#endif',
             'int main(void) {']
    for cname, cls in pairs.items():
        lines.append('printf("%s sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ['return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.run(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe)], check=True, capture_output=True)
    seen = 0
    for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines():
        cname, what, val = ln.split()
        cls = pairs[cname]
        want = ctypes.sizeof(cls) if what == "sizeof" else getattr(cls, what).offset
        assert int(val) == want, (cname, what, val, want)
        seen += 1
    assert seen == sum(1 + len(c._fields_) for c in pairs.values())
