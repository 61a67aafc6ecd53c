"""The CPU oracle itself: pinned against the committed golden vectors and against independent
implementations (scipy.ndimage.label, torch.nn.functional)."""
import numpy as np
import pytest
from scipy import ndimage

from cova_amd import synth
from oracle import ref
from tests.ccl_cases import hand_cases
from tests.golden_util import blobnet_golden, ccl_golden


def _as_rows(boxes):
    return np.stack([boxes[k] for k in ("left", "top", "width", "height", "area")], axis=1).astype(np.int64) \
        if len(boxes) else np.zeros((0, 5), np.int64)


@pytest.mark.parametrize("name", sorted(ccl_golden().keys()))
def test_ccl_oracle_matches_golden(name):
    m, exp = ccl_golden()[name]
    boxes, n = ref.regionprops(m, 1)
    assert n == len(exp)
    np.testing.assert_array_equal(_as_rows(boxes), exp)
    # area filter is >= on the pixel count (process.rs:39)
    for thr in (2, 3, 4, 30):
        boxes, n = ref.regionprops(m, thr)
        np.testing.assert_array_equal(_as_rows(boxes), exp[exp[:, 4] >= thr])


def test_ccl_specific_expectations():
    c = hand_cases()
    assert ref.regionprops(c["empty"], 1)[1] == 0
    b, n = ref.regionprops(c["full"], 1)
    assert n == 1 and tuple(b[0]) == (0, 0, 8, 6, 48)
    b, n = ref.regionprops(c["diagonal_chain"], 1)
    assert n == 1 and tuple(b[0]) == (0, 0, 8, 8, 8)
    b, n = ref.regionprops(c["checkerboard"], 1)
    assert n == 1
    b, n = ref.regionprops(c["isolated_grid"], 1)
    assert n == 4 * 5
    # block-raster order differs from pixel-raster order here: (1,1) before (0,6)
    b, n = ref.regionprops(c["order_block_vs_pixel"], 1)
    assert [(int(x["left"]), int(x["top"])) for x in b] == [(1, 1), (6, 0)]
    b, n = ref.regionprops(c["area_edge"], 3)
    assert [int(x["area"]) for x in b] == [3, 4]


@pytest.mark.parametrize("density", [0.02, 0.1, 0.3, 0.55])
def test_ccl_oracle_labels_vs_scipy(density):
    for m in synth.random_masks(6, 68, 120, density, seed=int(density * 100)):
        boxes, n, labels, n_labels = ref.regionprops(m, 1, want_labels=True)
        lab, k = ndimage.label(m, structure=np.ones((3, 3), int))
        assert k == n_labels == n
        # same partition: the label maps are a bijection of each other
        pairs = set(zip(lab[m > 0].tolist(), labels[m > 0].tolist()))
        assert len(pairs) == k
        assert (labels[m == 0] == 0).all()


def test_blobnet_oracle_matches_golden(weights_flat):
    for hw, (stack, logits) in blobnet_golden(weights_flat).items():
        h, w = map(int, hw.split("x"))
        got, mask = ref.blobnet_forward(weights_flat, stack, h, w)
        np.testing.assert_allclose(got, logits, rtol=0, atol=5e-5)
        np.testing.assert_array_equal(mask, (got > 0).astype(np.uint8))


def test_blobnet_oracle_vs_torch_live(weights_flat):
    from tests import torch_blobnet as tb
    h, w = 45, 80
    stack = synth.stacked_batch(1, h, w, seed=77)
    got, _ = ref.blobnet_forward(weights_flat, stack, h, w)
    exp, levels = tb.forward(weights_flat, stack, h, w, return_levels=True)
    np.testing.assert_allclose(got, exp, rtol=0, atol=5e-5)
    for lvl in range(4):
        lv = ref.blobnet_encoder_level(weights_flat, stack[0], h, w, lvl)
        np.testing.assert_allclose(lv, levels[lvl][0], rtol=1e-5, atol=5e-5)


def test_weights_roundtrip_and_count():
    from cova_amd import weights as W
    flat = W.random_init(5)
    assert flat.size == W.N_PARAMS == ref.lib().cova_ref_blobnet_num_params() == 320305
    back = W.from_bytes(W.to_bytes(flat))
    np.testing.assert_array_equal(flat, back)
    np.testing.assert_array_equal(W.flatten(W.unflatten(flat)), flat)
    with pytest.raises(ValueError):
        W.from_bytes(b"\0" * 64 + flat.tobytes())


def test_blob_like_weights_give_connected_blobs_over_the_moving_objects():
    """cova_amd.weights.blob_like: the representative weight set of the element / pipe / chain legs of bench.py -- a few
    connected components over the synthetic objects (whatever they cover: 1 .. 10 % here), nothing in the background."""
    from cova_amd import weights as W
    flat = W.blob_like(7)
    h, w = 45, 80
    fr = synth.carrier_frames(10, h, w, seed=5, n_objects=4)
    stack = np.stack([np.concatenate([fr[i - k] for k in range(4)], axis=0) for i in range(3, 10)])
    _, mask = ref.blobnet_forward(flat, stack, h, w)
    for i in range(len(stack)):
        cur = fr[i + 3]
        moving = ndimage.binary_dilation((cur[..., 1] > 3) | (cur[..., 2] > 3), iterations=3)
        assert not (mask[i].astype(bool) & ~moving).any()           # nothing outside the (dilated) objects
        _, k = ndimage.label(mask[i], structure=np.ones((3, 3), int))
        assert 1 <= k <= 6 and 0.005 < mask[i].mean() < 0.12   # (as much as the objects cover)
