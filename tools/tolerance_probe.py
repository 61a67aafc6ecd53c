"""Measures |logit_hip - logit_oracle| over weight / input seeds at 68x120 (what the BlobNet tolerance in
tests/test_gpu_blobnet.py is set from).  Prints one JSON line per (weight seed, input seed)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cova_amd import synth, weights as W  # noqa: E402
from cova_amd.elements import BlobNetInfer, Context  # noqa: E402
from oracle import ref  # noqa: E402

ctx = Context(0)
h, w = 68, 120
for wseed in (1234, 7, 2025, 99, 31337):
    flat = W.random_init(wseed)
    for iseed in (42, 4242):
        stack = synth.stacked_batch(32, h, w, seed=iseed, streams=4)
        net = BlobNetInfer(ctx, flat, h, w, max_batch=32)
        logits, mask = net.infer(stack)
        r, _ = ref.blobnet_forward(flat, stack, h, w)
        err = np.abs(logits - r)
        k = int(err.argmax())
        out = {"wseed": wseed, "iseed": iseed, "max_err": float(err.max()), "at_logit": float(r.reshape(-1)[k]),
               "logit_absmax": float(np.abs(r).max()), "logit_rms": float(np.sqrt((r ** 2).mean())),
               "p999_err": float(np.quantile(err, 0.999)), "mean_err": float(err.mean())}
        for a, rt in ((1e-2, 5e-3), (2e-2, 5e-3), (2e-2, 1e-2), (3e-2, 1e-2)):
            out[f"worst_ratio_{a}_{rt}"] = float((err / (a + rt * np.abs(r))).max())
        out["mask_flips"] = int((mask != (r > 0)).sum())
        out["max_abs_ref_at_flips"] = float(np.abs(r[mask != (r > 0)]).max()) if out["mask_flips"] else 0.0
        print(json.dumps(out), flush=True)
ctx.close()
