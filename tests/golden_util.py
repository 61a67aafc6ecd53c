import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def ccl_golden():
    d = json.load(open(os.path.join(GOLDEN, "ccl_golden.json")))
    out = {}
    for name, c in d["cases"].items():
        m = np.array([[1 if ch == "#" else 0 for ch in row] for row in c["rows"]], dtype=np.uint8)
        assert m.shape == (c["h"], c["w"])
        out[name] = (m, np.array(c["boxes"], dtype=np.int64).reshape(-1, 5))
    return out


def blobnet_golden(weights_flat):
    z = np.load(os.path.join(GOLDEN, "blobnet_golden.npz"))
    sha = np.frombuffer(hashlib.sha256(weights_flat.tobytes()).digest(), dtype=np.uint8)
    assert (sha == z["weights_sha256"]).all(), "seeded weights drifted from the ones the golden logits were made with"
    return {hw: (z[f"stack_{hw}"], z[f"logits_{hw}"]) for hw in ("45x80", "67x120", "68x120")}
