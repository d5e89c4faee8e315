#!/usr/bin/env python3
"""Where the HOST time of one cloud's pipeline goes (tools/sweep.py, one model, one cloud, one process): cProfile over the sweep's own
code, wall time per stage, and the GPU-busy share from the library's HIP-event profiler (slot 3 = whole forward calls).

    python tools/profile_sweep_host.py --model pointnet [--clouds 1]     (in a scratch directory)
"""
import argparse
import cProfile
import ctypes
import io
import os
import pstats
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="pointnet")
ap.add_argument("--clouds", type=int, default=1)
ap.add_argument("--top", type=int, default=45)
a = ap.parse_args()

import torch  # noqa: E402
import sweep  # noqa: E402
from interpret_quality_amd import _lib  # noqa: E402

lib = _lib.load()
argv = ["--models", a.model, "--datasets", "modelnet10", "--synthetic", "--num_clouds", str(a.clouds)]
sweep.run(sweep.parse(argv + ["--stages", "shapley_value"]), emit=False)          # warm-up: build caches, first-call costs
import shutil  # noqa: E402
shutil.rmtree("checkpoints", ignore_errors=True)
torch.cuda.synchronize()
lib.iq_profile_enable(1)
prof = cProfile.Profile()
t0 = time.time()
prof.enable()
rec = sweep.run(sweep.parse(argv), emit=False)
prof.disable()
torch.cuda.synchronize()
wall = time.time() - t0
lib.iq_profile_enable(0)
ms, n = ctypes.c_double(0), ctypes.c_int(0)
lib.iq_profile_read(3, ctypes.byref(ms), ctypes.byref(n))
print("model %s, %d cloud(s): wall %.2f s, %d coalitions (%d evaluated), forward calls on the GPU %.2f s in %d calls (%.0f %% of the wall)"
      % (a.model, a.clouds, wall, rec["coalitions"], rec["evaluated"], ms.value * 1e-3, n.value, 100.0 * ms.value * 1e-3 / wall))
for name, p in rec["phases"].items():
    print("  phase %-14s wall %7.2f s  coalitions %9d" % (name, p["wall_s"], p["coalitions"]))
s = io.StringIO()
pstats.Stats(prof, stream=s).sort_stats("cumulative").print_stats(a.top)
print(s.getvalue())
s = io.StringIO()
pstats.Stats(prof, stream=s).sort_stats("tottime").print_stats(25)
print(s.getvalue())
