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
