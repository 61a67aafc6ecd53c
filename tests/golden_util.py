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


# BlobNet tolerance (fp16 weights / activations with fp32 accumulation on the GPU vs the all-fp32 oracle).
# The error of such a network scales with the size of its outputs, so the absolute term is stated relative to
# the RMS of the oracle's logits:   |logit_hip - logit_oracle| <= 6e-3 * rms(logit_oracle) + 5e-3 * |logit_oracle|.
# Measured on MI355X over 5 weight seeds x 2 input seeds x 32 frames at 68x120 (tools/tolerance_probe.py,
# profiles/r2/tolerance_probe.jsonl): max |dlogit| / rms between 1.5e-3 and 6.5e-3 (absolute 5.8e-3 .. 5.4e-2 for
# logit RMS 2.6 .. 9.2); worst |dlogit| / tolerance 0.79.  For the bench weights (seed 1234, RMS 2.65) this is
# 1.6e-2 + 5e-3*|x| -- SURVEY.md section 8(c) started from 2e-2 + 1e-2*|x|.
BLOBNET_ATOL_PER_RMS, BLOBNET_RTOL = 6e-3, 5e-3


def blobnet_tolerance(ref_logits):
    import numpy as np
    rms = float(np.sqrt(np.mean(np.square(ref_logits, dtype=np.float64))))
    return BLOBNET_ATOL_PER_RMS * rms, BLOBNET_RTOL
