"""N>1 path on CPU: two ranks over gloo (torch.distributed.run, 127.0.0.1 rendezvous)."""
import json
import os
import subprocess
import sys

import pytest

from cova_amd.multigpu import Group, cpus_for_rank, parse_cpulist, streams_of_rank

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stream_sharding_is_a_partition():
    for world in (1, 2, 4, 8):
        for n in (1, 7, 8, 19):
            parts = [streams_of_rank(n, r, world) for r in range(world)]
            flat = sorted(s for p in parts for s in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
            for r, p in enumerate(parts):
                assert all(s % world == r for s in p)       # stream s -> GPU s mod world


def test_cpulist_and_core_shares():
    """Rank r pins itself to the cores of GPU r's NUMA node that its cgroup grants; ranks whose GPUs share a node split them;
    without node information the granted set is split -- never an empty set, never a core outside the grant."""
    assert parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert parse_cpulist("") == [] and parse_cpulist("x-y") == []
    allowed = list(range(16, 48))
    node = list(range(0, 32))                                     # node 0 of a 2 x 32-core host; the cgroup grants 16..47
    assert cpus_for_rank(allowed, node, 1, 0) == list(range(16, 32))
    a, b = cpus_for_rank(allowed, node, 2, 0), cpus_for_rank(allowed, node, 2, 1)
    assert a == list(range(16, 24)) and b == list(range(24, 32))
    assert cpus_for_rank(allowed, [], 4, 3) == list(range(40, 48))          # unknown node: the grant in four parts
    assert cpus_for_rank(allowed, list(range(100, 110)), 2, 1) == list(range(32, 48))   # a node outside the grant: same
    assert cpus_for_rank([5], [5], 8, 7) == [5]                   # fewer cores than ranks: shared
    parts = [cpus_for_rank(list(range(10)), [], 3, k) for k in range(3)]
    assert sorted(c for p_ in parts for c in p_) == list(range(10)) and all(parts)
    # ranks on DIFFERENT nodes whose nodes are all outside the grant: each one is the only sharer of its node, and without the
    # world / rank arguments every one of them would take the whole grant -- with them the grant is split among all ranks
    grant = list(range(8))
    outside = [list(range(100 + 10 * r, 110 + 10 * r)) for r in range(4)]
    shares = [cpus_for_rank(grant, outside[r], 1, 0, world=4, rank=r) for r in range(4)]
    assert shares == [[0, 1], [2, 3], [4, 5], [6, 7]]
    assert cpus_for_rank(grant, outside[0], 1, 0) == grant                      # (the old call: no world given)
    assert cpus_for_rank(allowed, node, 2, 1, world=8, rank=5) == list(range(24, 32))   # node known: world / rank play no part


def test_pin_binds_the_threads_that_already_exist():
    """sched_setaffinity(0) binds the calling thread only: a process that pins itself after a runtime has started threads must
    walk /proc/self/task (multigpu.pin_threads).  Run in a child so that the test runner keeps its own mask."""
    code = (
        "import os, sys, threading, time\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from cova_amd.multigpu import pin_threads\n"
        "allowed = sorted(os.sched_getaffinity(0))\n"
        "ev = threading.Event(); tids = []\n"
        "def w():\n"
        "    tids.append(threading.get_native_id()); ev.wait()\n"
        "ts = [threading.Thread(target=w) for _ in range(3)]\n"
        "[t.start() for t in ts]\n"
        "while len(tids) < 3: time.sleep(0.01)\n"
        "target = allowed[:1]\n"
        "n = pin_threads(target)\n"
        "masks = [sorted(os.sched_getaffinity(t)) for t in tids] + [sorted(os.sched_getaffinity(0))]\n"
        "ev.set(); [t.join() for t in ts]\n"
        "assert n >= 4, n\n"
        "assert all(m == target for m in masks), masks\n"
        "print('PINNED', n)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "PINNED" in r.stdout, r.stderr[-2000:]


def test_group_is_gloo_only():
    """The job's control plane never opens an RCCL communicator (north_star: no RCCL collectives): asking for one is an error."""
    with pytest.raises(ValueError):
        Group("nccl")
    g = Group()                                                   # world 1: a no-op object
    assert g.world == 1 and g.max(3.5) == 3.5 and g.gather({"a": 1}) == [{"a": 1}]
    g.barrier()
    g.close()


def test_two_rank_gloo_job(tmp_path):
    env = dict(os.environ)
    env.pop("RANK", None)
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29731",
           os.path.join(ROOT, "tests", "helpers", "gloo_worker.py"), str(tmp_path), "5"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    ranks = [json.load(open(tmp_path / f"rank{i}.json")) for i in range(2)]
    assert ranks[0]["streams"] == [0, 2, 4] and ranks[1]["streams"] == [1, 3]
    assert ranks[0]["world"] == ranks[1]["world"] == 2
    # MAX over ranks agrees on both ranks and is >= each local time; SUM sees every stream once
    assert ranks[0]["tmax"] == ranks[1]["tmax"] >= max(r_["elapsed"] for r_ in ranks) - 1e-9
    assert ranks[0]["total_streams"] == ranks[1]["total_streams"] == 5.0
    # the gather every rank sees: both ranks' stream sets, in rank order
    assert ranks[0]["gathered"] == ranks[1]["gathered"] == [[0, 2, 4], [1, 3]]
    assert ranks[0]["backend"] == "gloo"
    # per-stream host state is independent: every stream emits n-3 stacked frames and one track
    for r_ in ranks:
        for res in r_["results"]:
            assert res["emitted"] == 117 and res["dead_tracks"] == 1


def test_eight_rank_control_path_of_bench():
    """the exact 8-way control path of `bench.py --gpus 8` -- the ranks started as a child before anything
    touches a GPU, rendezvous on 127.0.0.1, stream partition, NUMA / core split with its documented sharing fallback, barrier,
    MAX over ranks, gather, ONE aggregate line -- runs once before an 8-GPU node ever sees it.  No kernel runs
    (--control-plane-only): the pool's process guard admits six processes on a card, so eight GPU ranks are the driver's to
    launch, not a test's.  Partition to match: gst-plugins/gst-gopsplit/gstgopsplit.cpp:556-603."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--control-plane-only"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE aggregate line on stdout"
    line = lines[0]
    assert line["n_gpus"] == 8 and line["value"] is None and line["control_plane_only"] and "NOT a measurement" in line["rehearsal"]
    assert "no RCCL" in line["control_plane"] and line["scaling"] == "weak"
    ranks = line["ranks"]
    assert [x["rank"] for x in ranks] == list(range(8))
    sets = [set(x["streams"]) for x in ranks]
    assert all(sets) and set().union(*sets) == set(range(64)) and sum(len(s) for s in sets) == 64   # eight disjoint stream sets
    for r_, x in enumerate(ranks):
        assert all(s % 8 == r_ for s in x["streams"])
    assert len({x["input_seed"] for x in ranks}) == 8
    assert abs(line["slowest_rank_region_s"] - 8e-3) < 1e-9                                            # MAX over the ranks' clocks
    allowed = sorted(os.sched_getaffinity(0))
    cpus = [x["cpus"] for x in ranks]
    assert all(c and set(c) <= set(allowed) for c in cpus)
    if len(allowed) >= 8 and all(x["pinned"] for x in ranks):
        flat = [c for cs in cpus for c in cs]
        assert len(flat) == len(set(flat)), "eight ranks split the granted cores"                      # eight disjoint core sets
    else:
        assert all(c == cpus[0] for c in cpus) or all(x["pinned"] for x in ranks)                      # the documented sharing fallback
    per_rank = [json.loads(l)["bench_rank"] for l in r.stderr.splitlines() if l.startswith('{"bench_rank"')]   # (one write per line: bench.rank_line)
    assert sorted(x["rank"] for x in per_rank) == list(range(8))


def test_kfd_topology_enumeration(tmp_path, monkeypatch):
    """multigpu.kfd_gpu_bus_ids on a made-up KFD topology: CPU nodes (no SIMDs) are skipped, GPUs whose render node the process
    cannot open are skipped (what the runtime does in a container that is granted one GPU of eight), location_id / domain become a
    PCI address, and a *_VISIBLE_DEVICES list of indices selects from that order; anything else than indices gives up ([])."""
    from cova_amd.multigpu import kfd_gpu_bus_ids
    nodes, dri = tmp_path / "nodes", tmp_path / "dri"
    dri.mkdir()
    spec = [  # (simd_count, drm_render_minor, location_id, domain, render node present)
        (0, -1, 0, 0, False),                 # the CPU
        (1024, 128, 0x0500, 0, True),         # 0000:05:00.0
        (1024, 129, 0x1508, 0, False),        # not granted to this process
        (1024, 130, 0x2600, 1, True),         # 0001:26:00.0
        (1024, 131, (0x65 << 8) | (0x1f << 3) | 0x7, 0, True),   # 0000:65:1f.7
    ]
    for n, (simd, minor, loc, dom, present) in enumerate(spec):
        d = nodes / str(n)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count 0\nsimd_count {simd}\ndrm_render_minor {minor}\nlocation_id {loc}\ndomain {dom}\n")
        if present:
            (dri / f"renderD{minor}").write_text("")
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    ids = kfd_gpu_bus_ids(str(nodes), str(dri))
    assert ids == ["0000:05:00.0", "0001:26:00.0", "0000:65:1f.7"]
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "2,0")
    assert kfd_gpu_bus_ids(str(nodes), str(dri)) == ["0000:65:1f.7", "0000:05:00.0"]
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1,2")           # applied first: the runtime below HIP sees two GPUs
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1")
    assert kfd_gpu_bus_ids(str(nodes), str(dri)) == ["0000:65:1f.7"]
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "GPU-0123456789abcdef")
    assert kfd_gpu_bus_ids(str(nodes), str(dri)) == []
    assert kfd_gpu_bus_ids(str(tmp_path / "absent"), str(dri)) == []
