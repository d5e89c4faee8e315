"""Start the ranks of a one-node job from a plain ``python script.py --gpus N`` (one process per GPU).

The reference has no launcher at all (README.md:87: "edit ``device_id``" and start another shell).  Here a script asked for
N > 1 GPUs outside ``torch.distributed.run`` becomes the PARENT of N fresh child processes - the same script, with the
rendezvous environment torchrun would give them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) - and does
nothing else: it must not have touched the GPU (no HIP call, no ``torch.cuda.is_available()``) before it calls
``self_launch``, and it never replaces itself (no exec).  Rank 0's stdout is the parent's stdout (one JSON line stays one JSON
line); the other ranks' stdout goes to stderr.  The parent returns 0 when every rank did, else the exit code of the rank that failed first; when a rank dies the survivors get
``grace_s`` to finish on their own (a peer blocked in a collective never would) and are then stopped by PID.
"""
import os
import signal
import socket
import subprocess
import sys
import time


def free_port():
    """A TCP port nobody is listening on right now, chosen by the kernel (instead of a fixed 295xx every job on the box shares)."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def under_launcher():
    """True inside a rank some launcher (torchrun, or self_launch below) started."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def ensure_rendezvous():
    """MASTER_ADDR / MASTER_PORT for a process group this process creates on its own (a forced single-rank group): a free
    port.  Several ranks cannot each pick one - they need the launcher's."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise SystemExit("WORLD_SIZE > 1 but no MASTER_PORT: start the ranks with `--gpus N` or torch.distributed.run")
        os.environ["MASTER_PORT"] = str(free_port())


KFD_NODES = "/sys/class/kfd/kfd/topology/nodes"
_VISIBLE_ENV = ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")


def kfd_gpus(nodes_dir=KFD_NODES, dri_dir="/dev/dri"):
    """GPUs the kernel driver shows to THIS process, from the KFD topology in sysfs - no HIP runtime, no torch: the nodes with
    SIMDs (CPU nodes have simd_count 0) whose render node this process may open (a container is handed its GPUs as
    /dev/dri/renderD<minor> devices; sysfs itself is not filtered).  None when there is no KFD tree to read."""
    try:
        names = sorted(os.listdir(nodes_dir), key=lambda n: (len(n), n))
    except OSError:
        return None
    try:
        have_dri = any(f.startswith("renderD") for f in os.listdir(dri_dir))
    except OSError:
        have_dri = False
    count = 0
    for n in names:
        try:
            props = dict(l.split(None, 1) for l in open(os.path.join(nodes_dir, n, "properties")).read().splitlines() if " " in l)
        except OSError:
            continue                                  # a node of another container: not ours
        if int(props.get("simd_count", "0")) <= 0:
            continue
        minor = int(props.get("drm_render_minor", "-1"))
        dev = os.path.join(dri_dir, "renderD%d" % minor)
        if minor < 0 or not have_dri:
            count += 1                                # no render-node bookkeeping (or no /dev/dri view at all): the node itself
        elif os.path.exists(dev) and os.access(dev, os.R_OK | os.W_OK):
            count += 1
    return count


def visible_gpus(nodes_dir=KFD_NODES, dri_dir="/dev/dri"):
    """Devices a rank of this job could use, WITHOUT initialising any GPU runtime in the parent: the KFD topology (kfd_gpus),
    cut down by the *_VISIBLE_DEVICES lists the ranks will inherit.  Only where there is no KFD tree at all does it fall back to
    torch.cuda.device_count() (which does not create a HIP context on this image)."""
    n = kfd_gpus(nodes_dir, dri_dir)
    if n is None:
        import torch
        return torch.cuda.device_count()
    for var in _VISIBLE_ENV:
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(script, argv, nproc, grace_s=30.0, poll_s=0.2):
    """Run ``python script argv`` as ``nproc`` ranks; returns the worst exit code.  See the module docstring."""
    if under_launcher():
        raise RuntimeError("self_launch called from inside a rank")
    rehearsal = os.environ.get("IQ_REHEARSAL") == "1" or os.environ.get("IQ_BENCH_REHEARSAL") == "1"
    have = visible_gpus()
    if not rehearsal and have < nproc:
        raise SystemExit("%s --gpus %d: this node shows %d GPU(s) (IQ_REHEARSAL=1 puts every rank on device 0 with gloo, "
                         "for tests)" % (os.path.basename(script), nproc, have))
    port = free_port()
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(nproc), LOCAL_WORLD_SIZE=str(nproc),
                IQ_SELF_LAUNCHED="1")
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: the only mode the host driver supports
    procs = []
    for r in range(nproc):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0")
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env, stdout=None if r == 0 else sys.stderr))

    def stop(sig):
        for p in procs:
            if p.poll() is None:
                try:
                    p.send_signal(sig)
                except OSError:
                    pass

    def relay(signum, _frame):      # Ctrl-C / a job manager's TERM reaches the ranks, by their PIDs
        stop(signum)

    old = {s: signal.signal(s, relay) for s in (signal.SIGINT, signal.SIGTERM)}
    first_bad = None
    try:
        failed_at = None
        while any(p.poll() is None for p in procs):
            codes = [p.poll() for p in procs]
            if failed_at is None and any(c not in (None, 0) for c in codes):
                failed_at = time.time()
                bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
                first_bad = bad[0][1]      # the rank that died FIRST: its peers fail after it, with less telling codes
                sys.stderr.write("[launch] rank(s) %s exited with %s; the others get %.0f s\n"
                                 % ([r for r, _ in bad], [c for _, c in bad], grace_s))
            if failed_at is not None and time.time() - failed_at > grace_s:
                stop(signal.SIGTERM)
                t0 = time.time()
                while any(p.poll() is None for p in procs) and time.time() - t0 < 10.0:
                    time.sleep(poll_s)
                stop(signal.SIGKILL)
            time.sleep(poll_s)
    finally:
        for s, h in old.items():
            signal.signal(s, h)
    codes = [p.wait() for p in procs]
    if first_bad is None:
        first_bad = next((c for c in codes if c != 0), 0)
    return first_bad if first_bad >= 0 else 128 - first_bad      # a signal's negative code -> the shell's 128 + signum
