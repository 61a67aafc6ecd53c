import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from cova_amd import synth, weights as W
from cova_amd.elements import BlobNetInfer, Context, FilterPipe, pack_frames
B, H, Wd = 256, 68, 120
ctx = Context(0)
net = BlobNetInfer(ctx, W.blob_like(7) if os.environ.get("PS_BLOB") else W.random_init(1234), H, Wd, max_batch=B)
frames, index = synth.carrier_batch(B, H, Wd, seed=1, streams=8)
nf = frames.shape[0]
src = pack_frames(frames)
ONLY = os.environ.get('PS_ONLY')
for lanes in ((int(ONLY.split(",")[0]),) if ONLY else (1, 2, 3, 4)):
  ctx.set_lanes(lanes)
  for ns in ((int(ONLY.split(',')[1]),) if ONLY else (2, 3, 4, 5, 6, 8)):
    pipe = FilterPipe(net, max_batch=B, max_frames=nf, max_boxes=256, n_slots=ns, packed=True)
    slots = []
    for _ in range(ns):
        slot, pf, pi = pipe.acquire(); pf[:nf] = src; pi[:B] = index; pipe.submit(slot, nf, B, 1); slots.append(slot)
    for s in slots: pipe.collect(s)
    steps = int(os.environ.get('PS_STEPS', 400))
    for rep in range(2):
        t0 = time.perf_counter(); inflight = []
        for _ in range(steps):
            acq = pipe.acquire()
            while acq is None:
                pipe.collect(inflight.pop(0)); acq = pipe.acquire()
            slot, pf, pi = acq
            pipe.submit(slot, nf, B, 1); inflight.append(slot)
        for s in inflight: pipe.collect(s)
        dt = time.perf_counter() - t0
    print(f"lanes {lanes} slots {ns}: {steps*B/dt/1e6:.3f} M frames/s  {dt/steps*1e6:.1f} us per batch", flush=True)
    pipe.acquire(); pipe.close()
