"""HIP bboxcc vs the CPU oracle: bit-exact boxes, counts and order."""
import numpy as np
import pytest

from cova_amd import synth
from cova_amd.elements import BboxCc
from oracle import ref
from tests.ccl_cases import hand_cases

pytestmark = pytest.mark.gpu


# kernel choice (include/covahip_dev.h): automatic, wave-per-frame kernel with a run capacity small enough that dense
# frames take the overflow pass, workgroup-per-frame kernel only
WAVE_CAPS = (0, 24, -1)


def _compare(ctx, masks, thresh, max_boxes=None):
    for cap in WAVE_CAPS:
        _compare_one(ctx, masks, thresh, max_boxes, cap)


def _compare_one(ctx, masks, thresh, max_boxes, cap):
    b, h, w = masks.shape
    max_boxes = max_boxes or ((h + 1) // 2) * ((w + 1) // 2)
    cc = BboxCc(ctx, cc_threshold=thresh, max_boxes=max_boxes)
    cc.set_wave_cap(cap)
    try:
        boxes, counts = cc.regionprops(masks)
    finally:
        cc.set_wave_cap(0)
    rboxes, rcounts = ref.regionprops_batch(masks, thresh, max_boxes)
    np.testing.assert_array_equal(counts, rcounts)
    for i in range(b):
        n = min(int(counts[i]), max_boxes)
        got = boxes[i, :n]
        exp = rboxes[i, :n]
        for f, g in (("left", "left"), ("top", "top"), ("width", "width"), ("height", "height"), ("area_px", "area")):
            np.testing.assert_array_equal(got[f], exp[g], err_msg=f"frame {i} field {f}")


@pytest.mark.parametrize("name", sorted(hand_cases().keys()))
@pytest.mark.parametrize("thresh", [1, 3])
def test_hand_cases(ctx, name, thresh):
    m = hand_cases()[name]
    # the kernel needs >= 1 block; embed small cases as they are
    _compare(ctx, m[None], thresh)


@pytest.mark.parametrize("hw", [(68, 120), (67, 120), (45, 80), (17, 23), (9, 15)])
@pytest.mark.parametrize("density", [0.01, 0.05, 0.15, 0.3, 0.5, 0.6])
def test_random_masks(ctx, hw, density):
    masks = synth.random_masks(16, hw[0], hw[1], density, seed=int(density * 1000) + hw[0])
    _compare(ctx, masks, 1)
    _compare(ctx, masks, 30)


def test_adversarial_full_size(ctx):
    h, w = 68, 120
    yy, xx = np.mgrid[0:h, 0:w]
    checker = ((yy + xx) % 2).astype(np.uint8)
    serp = np.zeros((h, w), np.uint8)
    serp[::2, :] = 1
    for k, r in enumerate(range(1, h, 2)):
        serp[r, w - 1 if k % 2 == 0 else 0] = 1
    spiral_free = np.zeros((h, w), np.uint8)
    spiral_free[::3, ::3] = 1
    masks = np.stack([checker, serp, np.ones((h, w), np.uint8), np.zeros((h, w), np.uint8), spiral_free,
                      (yy % 2).astype(np.uint8), (xx % 2).astype(np.uint8)])
    _compare(ctx, masks, 1)
    _compare(ctx, masks, 30)


def test_nonbinary_values_are_foreground(ctx):
    rng = np.random.default_rng(3)
    m = (rng.random((4, 68, 120)) < 0.2).astype(np.uint8) * rng.integers(1, 256, (4, 68, 120)).astype(np.uint8)
    _compare(ctx, m, 1)


def test_max_boxes_truncation(ctx):
    m = np.zeros((1, 68, 120), np.uint8)
    m[0, ::2, ::2] = 1          # 34*60 isolated pixels
    cc = BboxCc(ctx, cc_threshold=1, max_boxes=10)
    boxes, counts = cc.regionprops(m)
    assert counts[0] == 34 * 60
    rb, rc = ref.regionprops_batch(m, 1, 10)
    assert rc[0] == counts[0]
    np.testing.assert_array_equal(boxes[0]["left"], rb[0]["left"])
    np.testing.assert_array_equal(boxes[0]["top"], rb[0]["top"])


def test_large_batch_checksum(ctx):
    """BASELINE-size batch (b=256, 68x120): counts and a checksum of all boxes agree."""
    masks = synth.random_masks(256, 68, 120, 0.12, seed=99)
    _compare(ctx, masks, 1)


def test_transform_ip_bincode(ctx):
    from cova_amd.elements import deserialize_vec
    m = hand_cases()["area_edge"]
    cc = BboxCc(ctx, cc_threshold=3)
    data = cc.transform_ip(m.tobytes(), m.shape[1], m.shape[0])
    bb = deserialize_vec(data)
    assert len(data) == 8 + 24 * 2
    assert [tuple(map(float, (b["left"], b["top"], b["width"], b["height"], b["area"]))) for b in bb] == \
        [(4.0, 2.0, 3.0, 1.0, 3.0), (8.0, 4.0, 4.0, 1.0, 4.0)]


@pytest.mark.parametrize("hw", [(135, 240), (136, 240), (200, 250), (301, 97)])
@pytest.mark.parametrize("density", [0.02, 0.3, 0.55])
def test_frames_larger_than_lds(ctx, hw, density):
    """4K-class macroblock grids (2160p = 135 x 240): the per-frame state does not fit in 160 KB of LDS, the launch takes
    the block-based body with its arrays in a global-memory slab (bboxcc_big_kernel).  Same boxes, counts and order."""
    masks = synth.random_masks(5, hw[0], hw[1], density, seed=hw[0] + int(density * 100))
    _compare_one(ctx, masks, 1, None, 0)
    _compare_one(ctx, masks, 30, 300, 0)
    yy, xx = np.mgrid[0:hw[0], 0:hw[1]]
    adversarial = np.stack([((yy + xx) % 2).astype(np.uint8), np.ones(hw, np.uint8), np.zeros(hw, np.uint8),
                            ((yy % 4 < 2) & (xx % 6 < 5)).astype(np.uint8)])
    _compare_one(ctx, adversarial, 1, None, 0)


def test_frames_wider_than_256_pixels_are_refused(ctx):
    from cova_amd import _lib as L
    cc = BboxCc(ctx, cc_threshold=1, max_boxes=64)
    with pytest.raises(L.CovahipError):
        cc.regionprops(np.zeros((1, 300, 264), np.uint8))
