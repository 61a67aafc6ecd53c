"""Generates tests/golden/blobnet_golden.npz: seeded inputs + expected BlobNet logits.

Expected logits come from tests/torch_blobnet.py -- a torch.nn.functional composition of the
reference's Keras graph in float64, independent of oracle/blobnet_ref.c and of the HIP kernels.
The reference model itself (TensorFlow/Keras) cannot be imported in this image and ships no
weights, so these are spec-derived vectors.  Weights are regenerated from
the seed (cova_amd.weights.random_init(1234)); their SHA-256 is stored to detect drift.
Run from the repo root:   python tests/golden/gen_blobnet_golden.py
"""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cova_amd import synth, weights as W  # noqa: E402
from tests import torch_blobnet as tb  # noqa: E402


def main():
    flat = W.random_init(1234)
    out = {"weights_sha256": np.frombuffer(hashlib.sha256(flat.tobytes()).digest(), dtype=np.uint8)}
    for (h, w, b, seed) in [(45, 80, 2, 101), (67, 120, 1, 102), (68, 120, 1, 103)]:
        stack = synth.stacked_batch(b, h, w, seed=seed)
        logits = tb.forward(flat, stack, h, w, dtype=torch.float64).astype(np.float32)
        out[f"stack_{h}x{w}"] = stack
        out[f"logits_{h}x{w}"] = logits
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "blobnet_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
