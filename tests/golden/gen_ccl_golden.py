"""Generates tests/golden/ccl_golden.json: masks + expected regionprops output.

Expected values come from an implementation INDEPENDENT of oracle/ccl_ref.c and of the HIP
kernel: scipy.ndimage.label (pixel-based, 8-connectivity) for the components, numpy for the
statistics, and the ordering rule of OpenCV's block-based labelling (components sorted by
the raster index of the first 2x2 block they touch).  Run from the repo root:
    python tests/golden/gen_ccl_golden.py
"""
import json
import os
import sys

import numpy as np
from scipy import ndimage

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.ccl_cases import hand_cases  # noqa: E402


def expected_boxes(mask):
    h, w = mask.shape
    lab, n = ndimage.label(mask != 0, structure=np.ones((3, 3), int))
    bw = (w + 1) // 2
    comps = []
    for l in range(1, n + 1):
        ys, xs = np.nonzero(lab == l)
        key = int(((ys >> 1) * bw + (xs >> 1)).min())
        comps.append((key, [int(xs.min()), int(ys.min()), int(xs.max() - xs.min() + 1), int(ys.max() - ys.min() + 1),
                            int(len(ys))]))
    comps.sort(key=lambda c: c[0])
    return [c[1] for c in comps]


def main():
    out = {"_comment": "boxes = [left, top, width, height, area_px] in OpenCV label order, unfiltered", "cases": {}}
    cases = dict(hand_cases())
    rng = np.random.default_rng(2024)
    for (h, w, p) in [(12, 16, 0.3), (11, 13, 0.5), (45, 80, 0.1), (68, 120, 0.15), (67, 120, 0.4)]:
        cases[f"random_{h}x{w}_p{int(p * 100)}"] = (rng.random((h, w)) < p).astype(np.uint8)
    for name, m in cases.items():
        out["cases"][name] = {"h": int(m.shape[0]), "w": int(m.shape[1]),
                              "rows": ["".join("#" if v else "." for v in row) for row in m],
                              "boxes": expected_boxes(m)}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ccl_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
