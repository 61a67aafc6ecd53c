"""bboxcc batch sweep: device-resident masks, HIP-event timing, GB/s of mask bytes
against the 8 TB/s HBM peak.  Mask kinds: thresholded-noise ("noise", ~500 components/frame, what random
BlobNet weights produce), sparse blobs ("blobs", a few ellipses per frame, what a trained BlobNet produces),
Bernoulli(p)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cova_amd.elements import BboxCc, Context  # noqa: E402

H, W = 68, 120


def make_masks(kind, n, seed=0):
    rng = np.random.default_rng(seed)
    if kind.startswith("obj"):            # "obj50": exactly 50 ellipses of 1 .. 5 macroblocks radius per frame
        k = int(kind[3:])
        yy, xx = np.mgrid[0:H, 0:W]
        m = np.zeros((n, H, W), np.uint8)
        for i in range(n):
            for _ in range(k):
                cy, cx, ry, rx = rng.uniform(0, H), rng.uniform(0, W), rng.uniform(1, 5), rng.uniform(1, 5)
                m[i] |= (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1).astype(np.uint8)
        return m
    if kind == "blobs":
        yy, xx = np.mgrid[0:H, 0:W]
        m = np.zeros((n, H, W), np.uint8)
        for i in range(n):
            for _ in range(int(rng.integers(0, 9))):
                cy, cx, ry, rx = rng.uniform(0, H), rng.uniform(0, W), rng.uniform(1, 8), rng.uniform(1, 8)
                m[i] |= (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1).astype(np.uint8)
        return m
    p = {"noise": 0.3, "p05": 0.05, "p50": 0.5}[kind]
    return (rng.random((n, H, W)) < p).astype(np.uint8)


def main():
    ctx = Context(0)
    out = []
    cap = int(os.environ.get("SWEEP_CAP", "0"))        # developer switch: wave-kernel run capacity (0 auto, < 0 off)
    kinds = os.environ.get("SWEEP_KINDS", "blobs,noise,p05,p50").split(",")
    batches = [int(x) for x in os.environ.get("SWEEP_BATCHES", "256,1024,4096,16384,65536").split(",")]
    for kind in kinds:
        base = make_masks(kind, 256)
        for B in batches:
            masks = np.tile(base, (B // 256, 1, 1))
            d_m = ctx.malloc(masks.nbytes)
            ctx.h2d(d_m, masks)
            maxb = 64 if kind == "blobs" else 2048
            d_b = ctx.malloc(B * maxb * 20)
            d_c = ctx.malloc(B * 4)
            cc = BboxCc(ctx, cc_threshold=1, max_boxes=maxb)
            cc.set_wave_cap(cap)
            for _ in range(3):
                cc.regionprops_device(d_m, B, H, W, d_b, d_c)
            ctx.sync()
            reps = 20 if B <= 4096 else 5
            ctx.timer_start(1)
            for _ in range(reps):
                cc.regionprops_device(d_m, B, H, W, d_b, d_c)
            ctx.timer_stop(1)
            us = ctx.timer_ms(1) / reps * 1e3
            gbs = B * H * W / (us * 1e-6) / 1e9
            st = cc.overflow_stats()
            cnt = np.zeros(B, dtype=np.int32)
            ctx.d2h(cnt, d_c)
            out.append({"kind": kind, "batch": B, "wave_cap": cap, "us": round(us, 1), "ns_per_frame": round(us * 1e3 / B, 1),
                        "GBps": round(gbs, 1), "frac_hbm_peak": round(gbs / 8000, 4), "boxes_per_frame": round(float(cnt.mean()), 1),
                        "cap_pass1": st["cap_pass1"] if st["batch"] else None,
                        "overflow_frac_pass1": round(st["overflow_pass1"] / B, 4) if st["batch"] else None,
                        "overflow_frac_pass2": round(st["overflow_pass2"] / B, 4) if st["batch"] else None})
            print(out[-1], flush=True)
            for p in (d_m, d_b, d_c):
                ctx.free(p)
    path = os.environ.get("SWEEP_OUT", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out",
                                                    "bboxcc_sweep.json"))
    json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
