"""metapreprocess stacking ring (C-ABI) vs the oracle restatement and hand-derived expectations."""
import numpy as np
import pytest

from cova_amd.elements import FLOW_DROPPED, FLOW_OK, MetaPreprocess
from oracle import ref


def _frames(n, size, seed=0):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (n, size), dtype=np.uint8)


def test_caps_math():
    mp = MetaPreprocess(timestep=4)
    assert mp.transform_caps(1280, 720) == (80, 180)       # 45 * 4
    assert mp.transform_caps(1920, 1080) == (120, 268)     # 1080/16 = 67 (integer division), * 4
    assert mp.transform_caps(1920, 1088) == (120, 272)


@pytest.mark.parametrize("t,gamma", [(4, 1), (4, 2), (1, 1), (2, 3), (3, 1)])
def test_stack_sequence(t, gamma):
    w, h = 1280, 720
    mp = MetaPreprocess(timestep=t, gamma=gamma)
    ow, oh = mp.set_caps(w, h)
    spb = mp.size_per_buf
    assert spb == 80 * 45 * 4
    carrier = w * h * 3 // 2                 # I420 frame; only the first spb bytes matter
    frames = _frames(9, carrier, seed=t * 10 + gamma)
    outs, idx = [], []
    for i in range(9):
        flow, out = mp.transform(frames[i])
        if flow == FLOW_OK:
            outs.append(out)
            idx.append(i)
        else:
            assert flow == FLOW_DROPPED and out is None
    r_out, r_idx = ref.metapreprocess(frames, spb, t, gamma)
    assert idx == list(r_idx)
    assert idx == [i for i in range(t - 1, 9) if (i - (t - 1)) % gamma == 0]
    for k, i in enumerate(idx):
        np.testing.assert_array_equal(outs[k], r_out[k])
        for j in range(t):                   # row block j = frame i-j (imp.rs:307-320)
            np.testing.assert_array_equal(outs[k][j * spb:(j + 1) * spb], frames[i - j][:spb])


def test_first_outputs_dropped_for_seven_frames():
    mp = MetaPreprocess(timestep=4, gamma=1)
    mp.set_caps(1280, 720)
    flows = [mp.transform(np.full(1280 * 720 * 3 // 2, i, np.uint8))[0] for i in range(7)]
    assert flows == [FLOW_DROPPED] * 3 + [FLOW_OK] * 4
