"""PCIe-inclusive rate of the hot path through covahip_pipe_* (pinned slots, three streams) at b = 256, 68x120:
carrier frames of 8 streams in, packed boxes out.  Two figures: slots pre-filled (what the link and the GPU sustain
when the decoders write into the slots themselves) and with the copy of every batch into its slot on this thread."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cova_amd import synth, weights as W  # noqa: E402
from cova_amd.elements import BlobNetInfer, Context, FilterPipe  # noqa: E402


def run(pipe, frames, index, b, steps, fill):
    nf = frames.shape[0]
    slots = []
    for _ in range(3):                         # pre-fill every slot once
        slot, pf, pi = pipe.acquire()
        pf[:nf] = frames
        pi[:b] = index
        slots.append((slot, pf, pi))
        pipe.submit(slot, nf, b, 1)
    for slot, _, _ in slots:
        pipe.collect(slot)
    t0 = time.perf_counter()
    inflight = []
    boxes = 0
    for k in range(steps):
        acq = pipe.acquire()
        while acq is None:
            c, o, bx, _ = pipe.collect(inflight.pop(0))
            boxes += int(o[-1])
            acq = pipe.acquire()
        slot, pf, pi = acq
        if fill:
            pf[:nf] = frames
            pi[:b] = index
        pipe.submit(slot, nf, b, 1)
        inflight.append(slot)
    for slot in inflight:
        c, o, bx, _ = pipe.collect(slot)
        boxes += int(o[-1])
    dt = time.perf_counter() - t0
    return steps * b / dt, boxes / (steps * b)


def main():
    H, Wd, B = 68, 120, 256
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    ctx = Context(0)
    net = BlobNetInfer(ctx, W.random_init(1234), H, Wd, max_batch=B)
    frames, index = synth.carrier_batch(B, H, Wd, seed=0xC07A, streams=8)
    pipe = FilterPipe(net, max_batch=B, max_frames=frames.shape[0], max_boxes=2048, n_slots=3)
    out = {"batch": B, "carrier_frames_per_batch": int(frames.shape[0]), "h2d_bytes_per_batch": int(frames.nbytes)}
    out["frames_per_s_slots_prefilled"], out["boxes_per_frame"] = run(pipe, frames, index, B, steps, fill=False)
    out["frames_per_s_with_host_fill_one_thread"], _ = run(pipe, frames, index, B, steps, fill=True)
    print(json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in out.items()}))
    pipe.close()
    ctx.close()


if __name__ == "__main__":
    main()
