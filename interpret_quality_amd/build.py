"""Build recipe for libiq_hip.so (gfx950 only, hipcc cross-compiles without a GPU).

    python -m interpret_quality_amd.build [--force]

The shared library is written in-tree (interpret_quality_amd/lib/) so that it travels to the GPU
box with the repository snapshot; it is git-ignored.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBPATH = os.path.join(LIBDIR, "libiq_hip.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "iq.h")
ARCH = "gfx950"
# Index-valued kernels (FPS, ball query, region assignment) need individually rounded operations:
# hipcc's default -ffp-contract=fast fuses a*b+c into fma even through the __f*_rn intrinsics.
NO_CONTRACT = ("iq_geom.hip", "iq_pointnet2.hip", "iq_dgcnn.hip", "iq_pointconv.hip", "iq_smooth.hip")
# The smoothness enumeration is built without the SLP vectoriser, i.e. without packed float32 instructions (v_pk_mul_f32,
# v_pk_add_f32, v_pk_fma_f32).  Measured on MI355X (tools/dbg_smooth_det.py, profiles/r04_shared_gpu_determinism.txt): while a
# SECOND process ran the PointNet chain kernel with layer 3 on the bf16 matrix pipe on the same GPU, 40 % of the smoothness
# launches differed from the launch before (first by a few ulp in a few lanes, then amplified by the iteration up to 2e-3);
# with scalar float32 instructions none did, and no other neighbour (copies, rocBLAS float32, hipBLASLt bf16, DGCNN with the
# bf16x3 conv5, the chain kernel with layer 3 on the fp32 MFMA) had any effect.  No wave was preempted or moved (HW_ID, step
# gaps), the kernel uses no scratch and no global memory inside its loop.  One process per GPU - the product's layout - never
# meets this; the two-ranks-on-one-GPU tests do.
NO_PACKED_FP32 = ("iq_smooth.hip",)


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    return sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [HEADER]


def up_to_date():
    if not os.path.exists(LIBPATH):
        return False
    t = os.path.getmtime(LIBPATH)
    return all(os.path.getmtime(p) <= t for p in _deps())


def build(force=False, verbose=True, extra_flags=()):
    if not force and up_to_date():
        return LIBPATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ".o")
        newer = os.path.exists(obj) and all(os.path.getmtime(p) <= os.path.getmtime(obj)
                                            for p in [src, HEADER] + [d for d in _deps() if d.endswith(".h")])
        if force or not newer:
            cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj,
                   "-Wall", "-Wno-unused-function"] + list(extra_flags)
            if os.path.basename(src) in NO_CONTRACT:
                cmd.append("-ffp-contract=off")
            if os.path.basename(src) in NO_PACKED_FP32:
                cmd.append("-fno-slp-vectorize")
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIBPATH] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIBPATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIBPATH)
