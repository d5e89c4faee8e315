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
    assert dist.get_backend() == "gloo" and world == int(os.environ["IQ_TEST_WORLD"])
    if mode == "eight":
        # the 8-way splits of the reference's sizes (SURVEY 8e): 300 region pairs, 217 poses, 1000 permutations, and the ONE padded
        # all_gather_into_tensor that reassembles them - every rank holds the full tensor in unit order afterwards
        import sweep
        rec = {}
        for n in (300, 217, 1000, 5, 8):
            lo, hi = iqdist.shard_range(n)
            part = torch.arange(lo, hi, dtype=torch.float32).reshape(-1, 1) * torch.ones(1, 3)
            full = iqdist.all_gather_rows(part, n)
            assert full.shape == (n, 3) and torch.equal(full[:, 0], torch.arange(n, dtype=torch.float32)), (n, rank)
            rec[n] = [lo, hi]
        q = sweep.PullQueue("run8/A", rank, world)
        mine = []
        iqdist.group_barrier()
        while True:
            k = q.next()
            if k >= 360:                 # 6 models x 2 datasets x 30 clouds: the units of phase A of configs[4]
                break
            mine.append(k)
        q.publish("A", {"rank": rank, "got": mine, "shards": rec})
        iqdist.group_barrier()
        if rank == 0:
            print(json.dumps({"all": q.collect("A", None), "counts300": iqdist.shard_counts(300), "counts217": iqdist.shard_counts(217)}))
        iqdist.shutdown(ok=True)
    elif mode == "queue":
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


def _launch(tmp_path, mode, grace, world=2):
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    drv = ("import sys; sys.path.insert(0, %r)\nfrom interpret_quality_amd import launch\n"
           "sys.exit(launch.self_launch(%r, [%r], %d, grace_s=%r))\n" % (REPO, str(script), mode, world, grace))
    env = dict(os.environ, IQ_REHEARSAL="1", IQ_TEST_WORLD=str(world), OMP_NUM_THREADS="1")
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


def test_eight_ranks_shard_the_references_sizes_and_pull_every_unit_once(tmp_path):
    """The N = 8 layout the driver's scaling run uses, rehearsed with gloo on the CPU (the GPU pool allows six GPU processes per
    box, so eight ranks cannot be rehearsed on its one card): self-launch of 8 ranks, 300 pairs -> 38,38,38,38,37,37,37,37,
    217 poses -> 28 x 1 + 27 x 7, the padded all-gather back into unit order, and the sweep's pull queue over 360 units."""
    r = _launch(tmp_path, "eight", 30.0, world=8)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip() and not ln.startswith("[Gloo]")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["counts300"] == [38, 38, 38, 38, 37, 37, 37, 37]
    assert d["counts217"] == [28, 27, 27, 27, 27, 27, 27, 27] and sum(d["counts217"]) == 217
    ranks = sorted(d["all"], key=lambda x: x["rank"])
    assert [x["rank"] for x in ranks] == list(range(8))
    assert sorted(k for x in ranks for k in x["got"]) == list(range(360))          # every unit exactly once
    for n, w in (("300", 300), ("217", 217), ("1000", 1000), ("5", 5), ("8", 8)):
        spans = [x["shards"][n] for x in ranks]
        assert spans[0][0] == 0 and spans[-1][1] == w and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


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


def test_visible_gpus_counts_kfd_nodes_with_an_accessible_render_node(tmp_path, monkeypatch):
    """launch.visible_gpus() reads the KFD topology (no HIP runtime, no torch call in the parent): GPU nodes have SIMDs, CPU nodes
    have none; a container sees every node in sysfs but is handed only its own /dev/dri/renderD*; *_VISIBLE_DEVICES cut further."""
    sys.path.insert(0, REPO)
    from interpret_quality_amd import launch
    nodes, dri = tmp_path / "nodes", tmp_path / "dri"
    dri.mkdir()
    for i, (simd, minor) in enumerate([(0, -1), (0, -1), (1024, 128), (1024, 129), (1024, 130)]):
        (nodes / str(i)).mkdir(parents=True)
        (nodes / str(i) / "properties").write_text("cpu_cores_count %d\nsimd_count %d\ndrm_render_minor %d\n" % (0 if simd else 64, simd, minor))
    for k in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    assert launch.kfd_gpus(str(nodes), str(dri)) == 3            # no /dev/dri view at all: the nodes themselves
    (dri / "renderD128").write_text("")
    (dri / "renderD129").write_text("")
    (dri / "renderD130").write_text("")
    os.chmod(str(dri / "renderD130"), 0)
    if os.geteuid() != 0:                                       # root may open anything
        assert launch.kfd_gpus(str(nodes), str(dri)) == 2        # a render node of another container is not ours
    os.chmod(str(dri / "renderD130"), 0o600)
    assert launch.visible_gpus(str(nodes), str(dri)) == 3
    os.remove(str(dri / "renderD129"))                          # the GPU box: ten nodes in sysfs, one render node in /dev/dri
    assert launch.kfd_gpus(str(nodes), str(dri)) == 2
    (dri / "renderD129").write_text("")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    assert launch.visible_gpus(str(nodes), str(dri)) == 1
    assert launch.kfd_gpus(str(tmp_path / "absent")) is None    # no KFD tree: visible_gpus() falls back to torch's count
