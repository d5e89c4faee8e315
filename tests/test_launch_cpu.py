"""CPU: the self-launcher (interpret_quality_amd/launch.py) and the sweep's pull queue, two gloo ranks.

`python bench.py --gpus N` / `python tools/sweep.py --gpus N` outside torchrun become the parent of N fresh ranks; here the same
launcher starts a small CPU script (IQ_REHEARSAL=1: no device count check), so the rendezvous environment, the relay of rank
0's stdout, the exit-code propagation and the stop of a survivor blocked in a collective are covered without a GPU."""
import json
import os
import subprocess
import sys
import textwrap
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    sys.path.insert(0, os.path.join(%r, "tools"))
    import torch, torch.distributed as dist
    from interpret_quality_amd import dist as iqdist
    mode = sys.argv[1]
    rank, world, _ = iqdist.init_from_env("cpu", timeout_s=120)
    assert dist.get_backend() == "gloo" and world == 2
    if mode == "queue":
        import sweep
        got = []
        for name in ("p0", "p1"):
            q = sweep.PullQueue("run1/" + name, rank, world)
            mine = []
            iqdist.group_barrier()                           # both start pulling together
            while True:
                k = q.next()
                if k >= 37:
                    break
                mine.append(k)
                time.sleep(0.001 + 0.03 * rank)         # a slow rank takes fewer
            q.publish(name, {"rank": rank, "got": mine})
            iqdist.group_barrier()
            if rank == 0:
                got.append(q.collect(name, None))
        if rank == 0:
            print(json.dumps({"got": got, "port": os.environ["MASTER_PORT"], "self": os.environ.get("IQ_SELF_LAUNCHED")}))
        else:
            print("rank 1 says hello")     # must not reach the parent's stdout
        iqdist.shutdown(ok=True)
    elif mode == "die":
        if rank == 1:
            sys.exit(7)
        iqdist.group_barrier()               # rank 0 would wait here for the timeout
""") % (REPO, REPO)


def _launch(tmp_path, mode, grace):
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    drv = ("import sys; sys.path.insert(0, %r)\nfrom interpret_quality_amd import launch\n"
           "sys.exit(launch.self_launch(%r, [%r], 2, grace_s=%r))\n" % (REPO, str(script), mode, grace))
    env = dict(os.environ, IQ_REHEARSAL="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    return subprocess.run([sys.executable, "-c", drv], env=env, capture_output=True, text=True, timeout=300)


def test_self_launch_starts_two_ranks_and_the_pull_queue_hands_every_index_out_once(tmp_path):
    r = _launch(tmp_path, "queue", 30.0)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip() and not ln.startswith("[Gloo]")]   # (gloo's own connection notice)
    assert len(lines) == 1, r.stdout                      # rank 1's stdout went to stderr
    assert "rank 1 says hello" in r.stderr
    d = json.loads(lines[0])
    assert d["self"] == "1" and int(d["port"]) not in (29531, 29533)
    for per_rank in d["got"]:
        a, b = per_rank[0]["got"], per_rank[1]["got"]
        assert sorted(a + b) == list(range(37)) and a and b
        assert len(a) > len(b)                            # pulled, not dealt: the slow rank took fewer


def test_self_launch_returns_a_dead_ranks_exit_code_and_stops_the_survivor(tmp_path):
    t0 = time.time()
    r = _launch(tmp_path, "die", 3.0)
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])
    assert time.time() - t0 < 100                         # not the 120 s collective timeout
    assert "rank(s) [1] exited with [7]" in r.stderr


def test_free_port_and_rendezvous_defaults(monkeypatch):
    sys.path.insert(0, REPO)
    from interpret_quality_amd import launch
    p = launch.free_port()
    assert 1024 < p < 65536
    monkeypatch.delenv("MASTER_PORT", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    launch.ensure_rendezvous()
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 1024
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.delenv("MASTER_PORT", raising=False)
    try:
        launch.ensure_rendezvous()
        raise AssertionError("several ranks without a launcher's port must be refused")
    except SystemExit:
        pass
