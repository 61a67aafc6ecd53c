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
