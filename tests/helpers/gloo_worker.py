"""Worker for tests/test_multigpu_gloo.py: one rank of a world_size-2 gloo job (CPU only).

Exercises the N>1 plumbing bench.py uses -- rendezvous, barrier, MAX-over-ranks, and the
stream -> rank sharding -- together with the per-stream host objects (metapreprocess ring, SORT,
cova GoP filter) that each rank owns for its streams.  No GPU and no data-path collective.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from cova_amd import _lib as L  # noqa: E402
from cova_amd import elements as E  # noqa: E402
from cova_amd.multigpu import Group, streams_of_rank  # noqa: E402

CLK = 1_000_000_000 // 30


def run_stream(sid: int, n_frames: int = 120):
    """Host side of one stream: stacking ring + tracker fed with a scripted detection sequence."""
    mp = E.MetaPreprocess(timestep=4, gamma=1)
    mp.set_caps(1280, 720)
    rng = np.random.default_rng(1000 + sid)
    emitted = 0
    sort = E._SortHandle(10, 5, 0.1)
    dead_total = 0
    for i in range(n_frames):
        flow, out = mp.transform(rng.integers(0, 255, 1280 * 720 * 3 // 2, dtype=np.uint8))
        emitted += flow == E.FLOW_OK
        dets = np.zeros(1, dtype=L.BBOX_DTYPE)
        dets[0] = E.make_bbox(5 + 0.5 * i + sid, 10, 6, 6)[0]
        dead, lens = sort.update(dets if i < 60 else dets[:0], i * CLK)
        dead_total += len(lens)
    fin, lens = sort.finalize()
    return {"stream": sid, "emitted": int(emitted), "dead_tracks": int(dead_total + len(lens))}


def main():
    out_dir = sys.argv[1]
    n_streams = int(sys.argv[2])
    grp = Group(backend="gloo")
    mine = streams_of_rank(n_streams, grp.rank, grp.world)
    grp.barrier()
    t0 = time.perf_counter()
    results = [run_stream(s) for s in mine]
    time.sleep(0.05 * (grp.rank + 1))          # make the ranks finish at different times
    grp.barrier()
    elapsed = time.perf_counter() - t0
    tmax = grp.max(elapsed)
    total_streams = grp.sum(float(len(mine)))
    gathered = grp.gather(mine)
    with open(os.path.join(out_dir, f"rank{grp.rank}.json"), "w") as f:
        json.dump({"rank": grp.rank, "world": grp.world, "streams": mine, "results": results,
                   "elapsed": elapsed, "tmax": tmax, "total_streams": total_streams, "gathered": gathered,
                   "backend": grp.backend}, f)
    grp.close()


if __name__ == "__main__":
    main()
