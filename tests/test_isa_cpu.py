"""CPU: the gfx950 code INSIDE the libiq_hip.so that ships holds no packed float32 VALU instruction.

Rounds 4-5 (profiles/r05_packed_fp32_victim.txt): v_pk_mul_f32 / v_pk_add_f32 with op_sel:[0,1] - the form hipcc's SLP
vectoriser uses to broadcast one float of a register pair - return a wrong low result in lanes 48-63 while a second process runs a
bf16-MFMA-dense kernel on the same GPU (a register-only loop shows it in every launch); round 4 fenced ONE file with
-fno-slp-vectorize, a vectoriser heuristic.  The library is now built with the `packed-fp32-ops` subtarget feature switched off,
which removes the instructions at instruction selection; beside MFMAs they are also a measured anti-lever (MI355X_MICROARCH.md,
cycle constants).  This test disassembles the shared library itself (tools/isa_audit.py), so a compiler bump, a new flag or an
explicit float2 expression cannot bring them back silently."""
import os
import sys

import pytest

from interpret_quality_amd import build

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import isa_audit  # noqa: E402


@pytest.fixture(scope="module")
def audit():
    if not os.path.exists(os.path.join(isa_audit.LLVM_BIN, "llvm-objdump")):
        pytest.skip("llvm-objdump not found")
    return isa_audit.audit(build.build(verbose=False))


def test_no_packed_float32_beside_matrix_instructions(audit):
    mfma_kernels = {k: r for k, r in audit.items() if r["mfma"]}
    assert len(mfma_kernels) >= 40, "the audit did not find the MFMA kernels (%d)" % len(mfma_kernels)
    # the hot loops really are where the audit looks: the headline chain kernel and the grouped bf16x3 kernels
    for needle in ("pn_chain_kernel", "pn2_group_bf3_kernel", "pc_group_bf3_kernel", "pn_gemm_bf3_kernel", "edge_fused_kernel",
                   "knn_kernel"):
        assert any(needle in k for k in mfma_kernels), needle
    assert any(r["mfma_kinds"].get("v_mfma_f32_32x32x16_bf16", 0) >= 400 for r in mfma_kernels.values())
    offenders = {k: r["packed_kinds"] for k, r in mfma_kernels.items() if r["packed"]}
    assert not offenders, offenders


def test_no_packed_float32_anywhere_in_the_library(audit):
    """Index-producing kernels (FPS, ball query, kNN, region assignment) and the smoothness iteration included."""
    assert any("smooth_enum_kernel" in k for k in audit)
    offenders = {k: r["packed_kinds"] for k, r in audit.items() if r["packed"]}
    assert not offenders, offenders
