"""N>1 path on CPU: two ranks over gloo (torch.distributed.run, 127.0.0.1 rendezvous)."""
import json
import os
import subprocess
import sys

from cova_amd.multigpu import streams_of_rank

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
    # per-stream host state is independent: every stream emits n-3 stacked frames and one track
    for r_ in ranks:
        for res in r_["results"]:
            assert res["emitted"] == 117 and res["dead_tracks"] == 1
