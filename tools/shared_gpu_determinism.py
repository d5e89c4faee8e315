"""Is a kernel of the library reproducible while a SECOND process works on the same GPU?

    python tools/shared_gpu_determinism.py [--load pointnet|pointnet_fp32_l3|dgcnn|linear|matmul|bf16mm|copy|idle] [--seconds 20]

The parent process repeats FPS, the region assignment and the smoothness enumeration (all three modes, both objectives, three
synthetic clouds) on fixed inputs and compares every result bit by bit with the first one; the child process keeps the GPU
busy with the chosen kind of work.  Prints `rounds N mismatches {...}`; exit code 1 if anything differed.

Why it exists: in round 4 the two-ranks-on-one-GPU sweep test failed on the smoothness artefacts only.  This loop showed the
smoothness kernel deterministic alone and beside every neighbour except ONE - the PointNet chain kernel with layer 3 on the
bf16 matrix pipe - and only while the smoothness kernel used packed float32 instructions.  Round 5 found the ingredient with
register-only victims (tools/micro/pk_victim.hip, smooth_victim.hip; profiles/r05_packed_fp32_victim.txt): v_pk_mul_f32 /
v_pk_add_f32 with op_sel:[0,1] return a wrong low result in lanes 48-63 beside such a neighbour; the library is built without
packed float32 now (interpret_quality_amd/build.py).  IQ_LIBPATH=interpret_quality_amd/lib_packed_ab/libiq_hip.so (python -m
interpret_quality_amd.build --packed-ab) runs this tool on round 4's code generation.  tests/test_smoothness_gpu.py runs it for a few
seconds.  (Product code only: nothing under oracle/ is imported.)
"""
import argparse
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from interpret_quality_amd import _lib, hip_ops, synth

LOADS = ("pointnet", "pointnet_fp32_l3", "dgcnn", "linear", "matmul", "bf16mm", "copy", "idle")


def load(kind, seconds):
    """The neighbour: `seconds` of one kind of GPU work."""
    dev = torch.device("cuda:0")
    if kind == "idle":
        torch.zeros(1, device=dev)
        time.sleep(seconds)
        return
    if kind in ("pointnet", "pointnet_fp32_l3"):
        if kind == "pointnet_fp32_l3":
            _lib.load().iq_set_tuning(5, 54)       # layer 3 of the chains on the fp32 MFMA
        from interpret_quality_amd.pointnet import PointNetCls
        m = PointNetCls(None)
        m.load_state_dict(synth.to_torch(synth.pointnet_state_dict(0)))
        m = m.to(dev).eval()
        pts, _ = synth.make_cloud(1)
        data = torch.from_numpy(pts).unsqueeze(0).to(dev)
        rid = hip_ops.region_assign(data[0].contiguous(), hip_ops.fps(data, 32)[0].contiguous()).reshape(1, -1)
        keep = hip_ops.masks_to_tensor([int(x) for x in np.random.default_rng(0).integers(0, 1 << 32, size=3000)], dev)
        step = lambda: m.coalition_logits(data, data.mean(dim=1), rid, keep, None, num_regions=32, validate=False)
    elif kind == "dgcnn":
        from interpret_quality_amd.dgcnn import DGCNN_cls
        m = DGCNN_cls(argparse.Namespace(dataset="modelnet10", k=20))
        m.load_state_dict(synth.to_torch(synth.dgcnn_state_dict(0)))
        m = m.to(dev).eval()
        x = torch.randn(16, 3, 1024, device=dev)
        step = lambda: m(x)
    elif kind == "linear":                          # the library's dense layer (fp32 MFMA, accumulators in AGPRs)
        rng = np.random.default_rng(0)
        layer = hip_ops.PackedLinear(rng.standard_normal((256, 256), dtype=np.float32), np.zeros(256, np.float32), dev)
        x = torch.randn(1 << 18, 256, device=dev)
        step = lambda: hip_ops.linear(x, layer, 1)
    elif kind in ("matmul", "bf16mm"):
        dt = torch.bfloat16 if kind == "bf16mm" else torch.float32
        n = 8192 if kind == "bf16mm" else 4096
        a, b = torch.randn(n, n, device=dev, dtype=dt), torch.randn(n, n, device=dev, dtype=dt)
        step = lambda: a @ b
    elif kind == "copy":
        a = torch.randn(1 << 28, device=dev)
        b = torch.empty_like(a)
        step = lambda: b.copy_(a)
    else:
        raise SystemExit("unknown load %r" % kind)
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(4):
            step()
        torch.cuda.synchronize()


def check(seconds, show=4):
    dev = torch.device("cuda:0")
    fixed = {}
    for ci in (0, 1, 2):
        pts, _ = synth.make_cloud(ci)
        data = torch.from_numpy(pts).to(dev)
        f = hip_ops.fps(data.unsqueeze(0), 32)
        rid = hip_ops.region_assign(data.contiguous(), f[0].contiguous())
        torch.cuda.synchronize()
        fixed[ci] = (data, f.cpu().numpy(), rid, rid.cpu().numpy())
    first, bad, rounds, shown = {}, {"fps": 0, "region_assign": 0, "smoothness": 0}, 0, 0
    t0 = time.time()
    while time.time() - t0 < seconds:
        for ci, (data, f0, rid, rid0) in fixed.items():
            f = hip_ops.fps(data.unsqueeze(0), 32)
            bad["fps"] += not np.array_equal(f.cpu().numpy(), f0)
            bad["region_assign"] += not np.array_equal(hip_ops.region_assign(data.contiguous(), f[0].contiguous()).cpu().numpy(), rid0)
            for mode in hip_ops.SMOOTHNESS_MODES:
                for obj in ("inc", "dec"):
                    got = {k: v.cpu().numpy() for k, v in hip_ops.smoothness_enum(data, rid, 32, mode, obj).items()}
                    ref = first.setdefault((ci, mode, obj), got)
                    if all(np.array_equal(ref[k], got[k], equal_nan=True) for k in got):
                        continue
                    bad["smoothness"] += 1
                    if shown < show:
                        shown += 1
                        d = np.abs(ref["data"] - got["data"])                     # (E, N, 3)
                        e0 = int(np.nonzero(d.max(axis=(1, 2)))[0].min()) if d.max() > 0 else -1
                        pts_bad = np.nonzero(d[e0].max(axis=1))[0] if e0 >= 0 else []
                        print("MISMATCH cloud %d %s %s round %d: max |d| %.3g, first at epoch %d in %d points of regions %s"
                              % (ci, mode, obj, rounds, d.max(), e0, len(pts_bad), np.unique(rid0[pts_bad]).tolist()), flush=True)
        rounds += 1
    return rounds, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--load", default="pointnet", choices=LOADS)
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--role", default="main", choices=("main", "load"), help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.role == "load":
        load(a.load, a.seconds)
        return 0
    child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--role", "load", "--load", a.load,
                              "--seconds", str(a.seconds + 7)])
    try:
        time.sleep(5)                                  # the neighbour imports torch and builds its model first
        rounds, bad = check(a.seconds)
    finally:
        rc = child.wait()
    if rc != 0:
        print("the neighbour process failed (exit code %d): nothing was tested" % rc)
        return 2
    print("load %s: rounds %d mismatches %s" % (a.load, rounds, bad))
    return 1 if any(bad.values()) else 0


if __name__ == "__main__":
    sys.exit(main())
