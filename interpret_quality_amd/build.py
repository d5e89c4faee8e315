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
DEBUG_HEADER = os.path.join(os.path.dirname(HERE), "include", "iq_debug.h")
ARCH = "gfx950"
# Index-valued kernels (FPS, ball query, region assignment) need individually rounded operations:
# hipcc's default -ffp-contract=fast fuses a*b+c into fma even through the __f*_rn intrinsics.
NO_CONTRACT = ("iq_geom.hip", "iq_pointnet2.hip", "iq_dgcnn.hip", "iq_pointconv.hip", "iq_smooth.hip")
# The PointNet chain kernel pools with fmaxf over MFMA results; without -fno-honor-nans every such operand gets a canonicalising
# v_max_f32 x, x, x first (iq_mfma.h, max16).  The file holds no index-valued kernel and no NaN test.
NO_NANS = ("iq_pointnet.hip", "iq_linear.hip", "iq_dgcnn.hip")
# Packed float32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) are switched OFF for the whole library: the
# gfx950 subtarget feature `packed-fp32-ops` is removed for the device compile, so instruction selection cannot emit them
# whatever the SLP vectoriser or an explicit float2 expression asks for (tests/test_isa_cpu.py disassembles the .so that ships
# and asserts it).  Two reasons, both measured (profiles/r05_packed_fp32_*.txt):
#   1. beside MFMAs they are an anti-lever (MI355X_MICROARCH.md cycle constants: 2 v_pk_add_f32 per MFMA gap +26 cycles against
#      two scalar adds) and every bf16x3 hot loop carried 48-66 of them from the split residuals;
#   2. rounds 4-5: `v_pk_mul_f32` / `v_pk_add_f32` with `op_sel:[0,1]` (the low result taking the high dword of src1 - the form the
#      vectoriser uses to broadcast one float of a pair) return a wrong low result in lanes 48-63 while ANOTHER process runs a
#      bf16-MFMA-dense kernel on the same GPU: a register-only loop shows it in every launch (tools/micro/pk_victim.hip variant 3,
#      profiles/r05_packed_fp32_victim.txt).  Round 4's smoothness kernel held five such instructions, DGCNN's kNN kernel one.
# The host pass of hipcc does not know the feature and says so once per pass; that one line is filtered from the output.
NO_PACKED_FP32_FLAGS = ("-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops")
_HOST_NOISE = "'-packed-fp32-ops' is not a recognized feature for this target"


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    return sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [HEADER, DEBUG_HEADER]


def up_to_date():
    if not os.path.exists(LIBPATH):
        return False
    t = os.path.getmtime(LIBPATH)
    return all(os.path.getmtime(p) <= t for p in _deps())


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), flush=True)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    out = "\n".join(l for l in p.stdout.split("\n") if _HOST_NOISE not in l).strip()
    if out:
        print(out, flush=True)
    if p.returncode != 0:
        raise subprocess.CalledProcessError(p.returncode, cmd)


def build(force=False, verbose=True, extra_flags=(), packed_fp32=False, libdir=None):
    """Compile csrc/*.hip into <libdir>/libiq_hip.so.  packed_fp32=True / another libdir: A/B builds only (tools/r05_pk_ab.sh)."""
    libdir = libdir or LIBDIR
    libpath = os.path.join(libdir, "libiq_hip.so")
    if not force and libdir == LIBDIR and up_to_date():
        return LIBPATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(libdir, exist_ok=True)
    objs = []
    for src in sources():
        obj = os.path.join(libdir, os.path.basename(src)[:-4] + ".o")
        newer = os.path.exists(obj) and all(os.path.getmtime(p) <= os.path.getmtime(obj)
                                            for p in [src, HEADER, DEBUG_HEADER] + [d for d in _deps() if d.endswith(".h")])
        if force or not newer:
            cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj,
                   "-Wall", "-Wno-unused-function"] + list(extra_flags)
            if os.path.basename(src) in NO_CONTRACT:
                cmd.append("-ffp-contract=off")
            if os.path.basename(src) in NO_NANS:
                cmd.append("-fno-honor-nans")
            if not packed_fp32:
                cmd += NO_PACKED_FP32_FLAGS
            _run(cmd, verbose)
        objs.append(obj)
    _run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", libpath] + objs, verbose)
    return libpath


if __name__ == "__main__":
    if "--packed-ab" in sys.argv:      # the round-4 code generation (packed float32 allowed), beside the product build, for A/B runs
        print(build(force=True, packed_fp32=True, libdir=os.path.join(HERE, "lib_packed_ab")))
    else:
        print(build(force="--force" in sys.argv))
