#!/usr/bin/env python3
"""Instruction audit of the SHIPPED library: per kernel of libiq_hip.so, how many matrix instructions (v_mfma_*) and how many
packed float32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) its gfx950 code holds.

    python tools/isa_audit.py [path/to/libiq_hip.so] [--all]

Why: (1) MI355X_MICROARCH.md prices packed float32 beside MFMAs as an anti-lever (2 v_pk_add_f32 per MFMA gap +26 cycles against
two scalar ones); (2) rounds 4-5 found one operand form of v_pk_mul_f32 / v_pk_add_f32 (op_sel:[0,1]: the low result taking the
high dword of src1) returning wrong results in lanes 48-63 while another process runs a bf16-MFMA-dense kernel on the same GPU
(profiles/r05_packed_fp32_victim.txt).  The library is built with the packed-fp32 subtarget feature switched off
(interpret_quality_amd/build.py); tests/test_isa_cpu.py runs this audit on the .so that ships and fails on any packed float32
instruction, in a kernel that issues MFMAs or anywhere else.

The device code is taken out of the .so itself (llvm-objdump --offloading, in a scratch directory), so what is audited is what
the GPU box loads - not a side compile.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM_BIN = os.environ.get("IQ_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
PACKED = re.compile(r"\bv_pk_(mul|add|fma)_f32\b")
MFMA = re.compile(r"\bv_mfma_\w+")
LABEL = re.compile(r"^[0-9a-f]+ <([^>]+)>:\s*$")


def device_objects(so_path, scratch):
    """Extract every gfx950 code object bundled in the shared library into `scratch`; returns their paths."""
    local = os.path.join(scratch, os.path.basename(so_path))
    shutil.copy(so_path, local)
    subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", local], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    return sorted(os.path.join(scratch, f) for f in os.listdir(scratch) if "amdgcn-amd-amdhsa--gfx950" in f)


def audit(so_path):
    """{kernel or device function symbol: {"mfma": n, "packed": n, "mfma_kinds": {...}, "packed_kinds": {...}}}"""
    out = {}
    with tempfile.TemporaryDirectory() as scratch:
        objs = device_objects(so_path, scratch)
        if not objs:
            raise RuntimeError("no gfx950 code object found in %s" % so_path)
        for obj in objs:
            text = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--no-show-raw-insn", obj], check=True,
                                  capture_output=True, text=True).stdout
            cur = None
            for line in text.split("\n"):
                m = LABEL.match(line)
                if m:
                    cur = out.setdefault(m.group(1), {"mfma": 0, "packed": 0, "mfma_kinds": {}, "packed_kinds": {}})
                    continue
                if cur is None:
                    continue
                m = MFMA.search(line)
                if m:
                    cur["mfma"] += 1
                    cur["mfma_kinds"][m.group(0)] = cur["mfma_kinds"].get(m.group(0), 0) + 1
                m = PACKED.search(line)
                if m:
                    cur["packed"] += 1
                    cur["packed_kinds"][m.group(0)] = cur["packed_kinds"].get(m.group(0), 0) + 1
    return out


def demangle(names):
    try:
        res = subprocess.run([os.path.join(LLVM_BIN, "llvm-cxxfilt")], input="\n".join(names), capture_output=True, text=True,
                             check=True).stdout.split("\n")
        return dict(zip(names, res))
    except Exception:
        return {n: n for n in names}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = args[0] if args else os.path.join(here, "interpret_quality_amd", "lib", "libiq_hip.so")
    res = audit(so)
    names = demangle(sorted(res))
    show_all = "--all" in sys.argv
    print("%-110s %6s %6s" % ("kernel", "mfma", "packed"))
    bad = 0
    for sym in sorted(res, key=lambda s: -res[s]["packed"]):
        r = res[sym]
        if not (show_all or r["packed"] or r["mfma"]):
            continue
        nm = re.sub(r"\(anonymous namespace\)::", "", names[sym])
        nm = re.sub(r"\(.*", "", nm)[:110]
        print("%-110s %6d %6d %s" % (nm, r["mfma"], r["packed"], r["packed_kinds"] or ""))
        bad += bool(r["packed"] and r["mfma"])
    print("kernels: %d, with MFMAs: %d, with packed float32: %d, with both: %d" % (
        len(res), sum(1 for r in res.values() if r["mfma"]), sum(1 for r in res.values() if r["packed"]), bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
