"""Seeded synthetic compressed-domain inputs.

The reference's real input is the patched-FFmpeg entropy decoder's metadata carrier
(4 bytes per macroblock: mb_type, mv_x, mv_y, unused --
cova-rs/gst-plugins/src/tfrecordsink/imp.rs:105-112).  That decoder is an
un-vendored submodule, so benchmarks and tests use this generator instead.
"""
from __future__ import annotations

import numpy as np


def carrier_frames(n_frames: int, h_mb: int, w_mb: int, seed: int = 0xC07A, n_objects: int | None = None
                   ) -> np.ndarray:
    """u8 [n_frames][h_mb][w_mb][4] macroblock-metadata frames of one stream.

    Background: mb_type mostly 0 (skip) with sparse other types, motion vectors
    mostly 0.  A handful of axis-aligned ellipses move at constant velocity;
    inside them mv bytes are 1..12 (so the clip at 6 is exercised) and mb_type 1..7.
    Byte 3 is random and must be ignored by the network.
    """
    rng = np.random.default_rng(seed)
    f = np.zeros((n_frames, h_mb, w_mb, 4), dtype=np.uint8)
    bg_type = rng.integers(0, 8, size=f.shape[:3], dtype=np.uint8)
    f[..., 0] = np.where(rng.random(f.shape[:3]) < 0.8, 0, bg_type)
    for c in (1, 2):
        mv = rng.integers(1, 4, size=f.shape[:3], dtype=np.uint8)
        f[..., c] = np.where(rng.random(f.shape[:3]) < 0.9, 0, mv)
    f[..., 3] = rng.integers(0, 256, size=f.shape[:3], dtype=np.uint8)

    k = int(rng.integers(0, 13)) if n_objects is None else n_objects
    yy, xx = np.mgrid[0:h_mb, 0:w_mb]
    for _ in range(k):
        cy, cx = rng.uniform(0, h_mb), rng.uniform(0, w_mb)
        ry, rx = rng.uniform(1, 10), rng.uniform(1, 10)
        vy, vx = rng.uniform(-2, 2), rng.uniform(-2, 2)
        for i in range(n_frames):
            inside = ((yy - (cy + vy * i)) / ry) ** 2 + ((xx - (cx + vx * i)) / rx) ** 2 <= 1.0
            n = int(inside.sum())
            if n == 0:
                continue
            f[i, inside, 0] = rng.integers(1, 8, size=n, dtype=np.uint8)
            f[i, inside, 1] = rng.integers(1, 13, size=n, dtype=np.uint8)
            f[i, inside, 2] = rng.integers(1, 13, size=n, dtype=np.uint8)
    return f


def stacked_batch(batch: int, h_mb: int, w_mb: int, t: int = 4, seed: int = 0xC07A, streams: int = 1
                  ) -> np.ndarray:
    """u8 [batch][t*h_mb][w_mb][4]: what `metapreprocess timestep=t` would emit for
    `batch` consecutive frames (row block k = frame i-k; metapreprocess/imp.rs:307-320),
    drawn round-robin from `streams` independent synthetic streams."""
    per = -(-batch // streams)
    out = np.empty((batch, t * h_mb, w_mb, 4), dtype=np.uint8)
    for s in range(streams):
        fr = carrier_frames(per + t - 1, h_mb, w_mb, seed=seed + s)
        for j in range(per):
            b = j * streams + s
            if b >= batch:
                break
            i = j + t - 1
            for k in range(t):
                out[b, k * h_mb:(k + 1) * h_mb] = fr[i - k]
    return out


def random_masks(batch: int, h: int, w: int, density: float, seed: int = 7) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return (rng.random((batch, h, w)) < density).astype(np.uint8)


def carrier_batch(batch: int, h_mb: int, w_mb: int, t: int = 4, seed: int = 0xC07A, streams: int = 1):
    """The carrier-frame form of stacked_batch(batch, ..., seed, streams): (frames u8 [F][h][w][4], index i32 [batch][t])
    such that stacking frames[index[b, k]] for k = 0..t-1 along the row axis gives stacked_batch(...)[b]."""
    per = -(-batch // streams)
    frames = np.concatenate([carrier_frames(per + t - 1, h_mb, w_mb, seed=seed + s) for s in range(streams)])
    index = np.empty((batch, t), dtype=np.int32)
    for s in range(streams):
        for j in range(per):
            b = j * streams + s
            if b >= batch:
                break
            for k in range(t):
                index[b, k] = s * (per + t - 1) + j + t - 1 - k
    return frames, index
