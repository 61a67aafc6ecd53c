"""HIP BlobNet vs the fp32 CPU oracle (SURVEY.md §8c tolerances)."""
import numpy as np
import pytest

from cova_amd import synth
from cova_amd.elements import BlobNetInfer
from oracle import ref

pytestmark = pytest.mark.gpu

from tests.golden_util import blobnet_tolerance  # noqa: E402


def _check(logits, mask, ref_logits):
    err = np.abs(logits - ref_logits)
    atol, rtol = blobnet_tolerance(ref_logits)
    tol = atol + rtol * np.abs(ref_logits)
    assert (err <= tol).all(), f"max err {err.max():.4g}, worst excess {(err - tol).max():.4g}"
    # the mask is the sign of the build's own logits ...
    np.testing.assert_array_equal(mask, (logits > 0).astype(np.uint8))
    # ... and may differ from the oracle's mask only where the oracle logit is within tolerance of 0
    diff = mask != (ref_logits > 0)
    assert (np.abs(ref_logits[diff]) <= atol).all()
    return float(err.max())


@pytest.mark.parametrize("impl", ["mfma", "dec_separate"])
@pytest.mark.parametrize("hw", [(68, 120), (67, 120), (45, 80), (35, 60)])
def test_logits_match_oracle(ctx, weights_flat, hw, impl):
    h, w = hw
    stack = synth.stacked_batch(3, h, w, seed=11, streams=3)
    net = BlobNetInfer(ctx, weights_flat, h, w, max_batch=4)
    net.set_impl(impl)
    logits, mask = net.infer(stack)
    ref_logits, _ = ref.blobnet_forward(weights_flat, stack, h, w)
    _check(logits, mask, ref_logits)


def test_alpha_channel_ignored_and_clip(ctx, weights_flat):
    h, w = 45, 80
    stack = synth.stacked_batch(2, h, w, seed=5)
    net = BlobNetInfer(ctx, weights_flat, h, w, max_batch=2)
    l0, _ = net.infer(stack)
    s2 = stack.copy()
    s2[..., 3] = 255 - s2[..., 3]
    l1, _ = net.infer(s2)
    np.testing.assert_array_equal(l0, l1)
    s3 = stack.copy()
    s3[..., :3] = np.where(s3[..., :3] >= 6, 200, s3[..., :3])   # everything >= 6 clips to 6
    l2, _ = net.infer(s3)
    np.testing.assert_array_equal(l0, l2)


def _mixed_gamma_weights(seed):
    """Every second BN gamma of every encoder level negative: the ALLPOS=false kernel variants."""
    from cova_amd import weights as W
    wts = W.unflatten(W.random_init(seed))
    for i in range(4):
        g = wts[f"enc{i}.bn.gamma"]
        g[::2] *= -1.0
    return W.flatten(wts)


@pytest.mark.parametrize("impl", ["mfma", "dec_separate"])
@pytest.mark.parametrize("hw,b", [((68, 120), 16), ((67, 120), 16), ((45, 80), 2)])
def test_negative_bn_gamma(ctx, impl, hw, b):
    """BN runs after ReLU and before max-pool; a negative gamma must not commute with the max
    (encoder.py:61-66).  A trained model with one negative gamma per level takes these kernels."""
    flat = _mixed_gamma_weights(77)
    h, w = hw
    stack = synth.stacked_batch(b, h, w, seed=21, streams=4)
    net = BlobNetInfer(ctx, flat, h, w, max_batch=b)
    net.set_impl(impl)
    logits, mask = net.infer(stack)
    ref_logits, _ = ref.blobnet_forward(flat, stack, h, w)
    _check(logits, mask, ref_logits)


@pytest.mark.parametrize("wseed", [1234, 7, 2025])
@pytest.mark.parametrize("iseed", [42, 4242])
def test_many_seeds_at_baseline_size(ctx, wseed, iseed):
    """68x120 (BASELINE size): 32 frames x 3 weight seeds x 2 input seeds against the oracle."""
    from cova_amd import weights as W
    flat = W.random_init(wseed)
    h, w = 68, 120
    stack = synth.stacked_batch(32, h, w, seed=iseed, streams=4)
    net = BlobNetInfer(ctx, flat, h, w, max_batch=32)
    logits, mask = net.infer(stack)
    ref_logits, _ = ref.blobnet_forward(flat, stack, h, w)
    err = _check(logits, mask, ref_logits)
    print(f"weights {wseed} inputs {iseed}: max |dlogit| = {err:.4g}")


def test_batch_independence_and_full_batch(ctx, weights_flat):
    """b=256 at 68x120 (BASELINE config): frames are independent -> a frame's logits do not
    depend on its batch position; spot-check a sample of frames against the oracle."""
    h, w = 68, 120
    stack = synth.stacked_batch(256, h, w, seed=42, streams=8)
    net = BlobNetInfer(ctx, weights_flat, h, w, max_batch=256)
    logits, mask = net.infer(stack)
    idx = [0, 1, 100, 255]
    l_small, _ = net.infer(stack[idx])
    np.testing.assert_array_equal(logits[idx], l_small)
    ref_logits, _ = ref.blobnet_forward(weights_flat, stack[idx], h, w)
    _check(logits[idx], mask[idx], ref_logits)


def test_fused_filter_matches_separate(ctx, weights_flat):
    from cova_amd.elements import BboxCc
    h, w = 68, 120
    stack = synth.stacked_batch(8, h, w, seed=7, streams=2)
    net = BlobNetInfer(ctx, weights_flat, h, w, max_batch=8)
    boxes, counts, mask = net.filter(stack, cc_threshold=1, max_boxes=2048, want_mask=True)
    rb, rc = ref.regionprops_batch(mask, 1, 2048)
    np.testing.assert_array_equal(counts, rc)
    for i in range(8):
        n = int(counts[i])
        for f, g in (("left", "left"), ("top", "top"), ("width", "width"), ("height", "height"), ("area_px", "area")):
            np.testing.assert_array_equal(boxes[i, :n][f], rb[i, :n][g])
    _, mask2 = net.infer(stack)
    np.testing.assert_array_equal(mask, mask2)


@pytest.mark.parametrize("hw", [(68, 120), (67, 120), (45, 80), (35, 60)])
@pytest.mark.parametrize("want_mask", [False, True])
def test_fused_tail_matches_separate_kernels(ctx, weights_flat, hw, want_mask):
    """covahip_filter_forward runs the last decoder block and bboxcc in ONE launch (mask in LDS); its boxes, counts
    and (when asked for) mask equal covahip_blobnet_forward followed by covahip_bboxcc on the mask."""
    from cova_amd.elements import BboxCc
    h, w = hw
    stack = synth.stacked_batch(5, h, w, seed=17, streams=2)
    net = BlobNetInfer(ctx, weights_flat, h, w, max_batch=8)
    boxes, counts, mask = net.filter(stack, cc_threshold=2, max_boxes=2048, want_mask=want_mask)
    _, mask_sep = net.infer(stack)
    boxes_sep, counts_sep = BboxCc(ctx, 2, 2048).regionprops(mask_sep)
    np.testing.assert_array_equal(counts, counts_sep)
    for i in range(len(counts)):
        np.testing.assert_array_equal(boxes[i, :counts[i]], boxes_sep[i, :counts[i]])
    if want_mask:
        np.testing.assert_array_equal(mask, mask_sep)
    assert counts.sum() > 0


def test_batch_size_sequence_on_one_context(ctx, weights_flat):
    """The band planners, swizzle choice and item order depend on the batch size; one model must serve any
    sequence of batch sizes with identical per-frame results (frames are independent)."""
    h, w = 68, 120
    stack = synth.stacked_batch(40, h, w, seed=23, streams=4)
    net = BlobNetInfer(ctx, weights_flat, h, w, max_batch=64)
    ref_l, ref_m = net.infer(stack)
    ref_boxes, ref_counts, _ = net.filter(stack, cc_threshold=1, max_boxes=2048)
    for n in (1, 3, 17, 40, 2, 33, 8):
        l, m = net.infer(stack[:n])
        np.testing.assert_array_equal(l, ref_l[:n])
        np.testing.assert_array_equal(m, ref_m[:n])
        boxes, counts, _ = net.filter(stack[:n], cc_threshold=1, max_boxes=2048)
        np.testing.assert_array_equal(counts, ref_counts[:n])
        for i in range(n):
            np.testing.assert_array_equal(boxes[i, :counts[i]], ref_boxes[i, :counts[i]])


def test_host_buffer_call_equals_device_pointer_call(ctx, weights_flat):
    """covahip_filter_forward on host buffers (staged H2D, results copied back) gives the boxes, counts and masks of
    the device-pointer call on the same batch."""
    h, w, b = 68, 120, 200
    stack = synth.stacked_batch(b, h, w, seed=29, streams=8)
    net = BlobNetInfer(ctx, weights_flat, h, w, max_batch=256)
    boxes, counts, mask = net.filter(stack, cc_threshold=1, max_boxes=1024, want_mask=True)
    d_stack = ctx.malloc(stack.nbytes)
    ctx.h2d(d_stack, stack)
    d_boxes, d_counts, d_mask = ctx.malloc(b * 1024 * 20), ctx.malloc(b * 4), ctx.malloc(b * h * w)
    net.filter_device(d_stack, b, 1, d_boxes, d_counts, 1024, d_mask)
    from cova_amd import _lib as L
    boxes2 = np.zeros((b, 1024), dtype=L.BOX_DTYPE)
    counts2 = np.zeros(b, dtype=np.int32)
    mask2 = np.zeros((b, h, w), dtype=np.uint8)
    ctx.d2h(boxes2, d_boxes); ctx.d2h(counts2, d_counts); ctx.d2h(mask2, d_mask)
    np.testing.assert_array_equal(counts, counts2)
    np.testing.assert_array_equal(mask, mask2)
    for i in range(b):
        np.testing.assert_array_equal(boxes[i, :counts[i]], boxes2[i, :counts[i]])


@pytest.mark.parametrize("hw", [(68, 120), (67, 120), (45, 80), (35, 60)])
@pytest.mark.parametrize("mixed_gamma", [False, True])
def test_carrier_frame_entry_equals_stacked_entry_bitwise(ctx, weights_flat, hw, mixed_gamma):
    """covahip_filter_forward_frames (stacking as an index gather on the GPU, level 0 computed once per carrier frame)
    gives the logits, masks and boxes of covahip_filter_forward on the stacks the index table describes, bit for bit."""
    h, w = hw
    b, streams = 40, 3
    flat = _mixed_gamma_weights(77) if mixed_gamma else weights_flat
    frames, index = synth.carrier_batch(b, h, w, seed=61, streams=streams)
    stack = synth.stacked_batch(b, h, w, seed=61, streams=streams)
    np.testing.assert_array_equal(np.concatenate([frames[index[:, k]] for k in range(4)], axis=1), stack)
    net = BlobNetInfer(ctx, flat, h, w, max_batch=64)
    boxes, counts, mask = net.filter(stack, cc_threshold=2, max_boxes=2048, want_mask=True)
    logits, _ = net.infer(stack)
    for impl in ("mfma", "dec_separate"):
        net.set_impl(impl)
        fboxes, fcounts, fmask, flogits = net.filter_frames(frames, index, 2, max_boxes=2048, want_mask=True, want_logits=True)
        np.testing.assert_array_equal(flogits, logits, err_msg=impl)
        np.testing.assert_array_equal(fmask, mask)
        np.testing.assert_array_equal(fcounts, counts)
        for i in range(b):
            np.testing.assert_array_equal(fboxes[i, :counts[i]], boxes[i, :counts[i]])
        # a shuffled batch: any order of the table's rows
        perm = np.random.default_rng(3).permutation(b)
        _, pcounts, pmask, plogits = net.filter_frames(frames, index[perm], 2, max_boxes=2048, want_mask=True, want_logits=True)
        np.testing.assert_array_equal(plogits, logits[perm], err_msg=impl + " shuffled")
        np.testing.assert_array_equal(pcounts, counts[perm])
    net.set_impl("mfma")
    # one stream in order, no table: output k = frames k+3 .. k
    one = synth.carrier_frames(20, h, w, seed=5)
    st1 = np.stack([np.concatenate([one[i - k] for k in range(4)], axis=0) for i in range(3, 20)])
    b1, c1, m1 = net.filter(st1, cc_threshold=2, max_boxes=2048, want_mask=True)
    b2, c2, m2, _ = net.filter_frames(one, None, 2, max_boxes=2048, want_mask=True)
    np.testing.assert_array_equal(m2, m1)
    np.testing.assert_array_equal(c2, c1)


def test_carrier_frame_entry_rejects_bad_tables(ctx, weights_flat):
    from cova_amd import _lib as L
    h, w = 45, 80
    net = BlobNetInfer(ctx, weights_flat, h, w, max_batch=8)
    frames = synth.carrier_frames(6, h, w, seed=3)
    idx = np.array([[3, 2, 1, 0], [4, 3, 2, 6]], dtype=np.int32)          # 6 is outside the 6 frames
    with pytest.raises(L.CovahipError):
        net.filter_frames(frames, idx, 1)
    with pytest.raises(L.CovahipError):
        net.filter_frames(frames, np.array([[3, 2, 1, -1]], dtype=np.int32), 1)
    boxes, counts, _, _ = net.filter_frames(frames, np.array([[5, 4, 3, 2], [5, 5, 5, 5]], dtype=np.int32), 1)
    assert counts.shape == (2,)


@pytest.mark.parametrize("plans", [[(1, 9, 2)], [(2, 8, 2), (3, 2, 2)], [(1, 7, 1), (2, 3, 1), (3, 4, 1)]])
def test_encoder_band_plans_give_identical_bits(ctx, weights_flat, plans):
    """covahip_blobnet_set_enc_plan (developer header): other band counts and the double-buffered item loop (the next
    band requested while this one is computed) change the schedule, not one bit of the result -- both entry points."""
    h, w, b = 68, 120, 24
    frames, index = synth.carrier_batch(b, h, w, seed=9, streams=2)
    stack = synth.stacked_batch(b, h, w, seed=9, streams=2)
    net = BlobNetInfer(ctx, weights_flat, h, w, max_batch=b)
    # (a level-1 plan selects the round-1..3 level-1 kernel -- enc_mfma<16,32> -- whose band count it is: the reference
    # run uses that kernel too; the round-4 kernel rounds differently and is compared with the oracle elsewhere)
    net.set_impl("enc1_legacy")
    try:
        ref_logits, _ = net.infer(stack)
        ref = net.filter_frames(frames, index, 2, max_boxes=1024, want_mask=True, want_logits=True)
        for level, nbands, nbuf in plans:
            net.set_enc_plan(level, nbands, nbuf)
        logits, _ = net.infer(stack)
        got = net.filter_frames(frames, index, 2, max_boxes=1024, want_mask=True, want_logits=True)
    finally:
        net.set_impl("mfma")
        for level, _, _ in plans:
            net.set_enc_plan(level, 0)
    np.testing.assert_array_equal(logits, ref_logits)
    np.testing.assert_array_equal(got[3], ref[3])
    np.testing.assert_array_equal(got[2], ref[2])
    np.testing.assert_array_equal(got[1], ref[1])


@pytest.mark.parametrize("hw", [(68, 120), (67, 120)])
@pytest.mark.parametrize("mixed_gamma", [False, True])
def test_row_aligned_tiles_match_general_tiles_bitwise(ctx, weights_flat, hw, mixed_gamma):
    """Encoder levels 2 and 3 on row-aligned tiles (eight windows of one window row, fragment addresses from per-kernel lane
    constants, periodic swizzle; the default at 1080p) vs the general tiles (developer switch "enc_general_tiles"): another
    tile enumeration and LDS layout, the same products in the same order -- logits, masks and boxes bit for bit, both gamma
    sign classes, both entries."""
    h, w = hw
    b = 20
    flat = _mixed_gamma_weights(78) if mixed_gamma else weights_flat
    stack = synth.stacked_batch(b, h, w, seed=23, streams=2)
    frames, index = synth.carrier_batch(b, h, w, seed=23, streams=2)
    net = BlobNetInfer(ctx, flat, h, w, max_batch=b)
    logits, mask = net.infer(stack)
    got = net.filter_frames(frames, index, 2, max_boxes=1024, want_mask=True, want_logits=True)
    net.set_impl("enc_general_tiles")
    try:
        logits2, mask2 = net.infer(stack)
        got2 = net.filter_frames(frames, index, 2, max_boxes=1024, want_mask=True, want_logits=True)
    finally:
        net.set_impl("mfma")
    np.testing.assert_array_equal(logits, logits2)
    np.testing.assert_array_equal(mask, mask2)
    for a, c in zip(got, got2):
        np.testing.assert_array_equal(a, c)


@pytest.mark.parametrize("hw", [(68, 120), (67, 120), (66, 116), (61, 100), (36, 64)])
@pytest.mark.parametrize("mixed_gamma", [False, True])
def test_fused_encoder_levels_2_3_match_separate_launches_bitwise(ctx, weights_flat, hw, mixed_gamma):
    """Encoder levels 2 + 3 as ONE launch (enc23_mfma, round 5: level 2's input through a ring of rows, its output written
    into level 3's LDS band, T halves of a tile on two waves) vs the two launches of enc_mfma (developer switch
    "enc23_separate"): the same products in the same order per accumulator -- logits, masks and boxes bit for bit, both
    gamma sign classes, both entries, grids with odd / even level-2 and level-3 sizes (pad rows and columns)."""
    h, w = hw
    b = 20
    flat = _mixed_gamma_weights(91) if mixed_gamma else weights_flat
    stack = synth.stacked_batch(b, h, w, seed=29, streams=2)
    frames, index = synth.carrier_batch(b, h, w, seed=29, streams=2)
    net = BlobNetInfer(ctx, flat, h, w, max_batch=b)
    net.set_impl("enc23_force")    # (a batch of 20 would take the two launches by default)
    logits, mask = net.infer(stack)
    got = net.filter_frames(frames, index, 2, max_boxes=1024, want_mask=True, want_logits=True)
    net.set_impl("enc23_separate")
    try:
        logits2, mask2 = net.infer(stack)
        got2 = net.filter_frames(frames, index, 2, max_boxes=1024, want_mask=True, want_logits=True)
    finally:
        net.set_impl("mfma")
    np.testing.assert_array_equal(logits, logits2)
    np.testing.assert_array_equal(mask, mask2)
    for a, c in zip(got, got2):
        np.testing.assert_array_equal(a, c)
    # twice in a row on the same workspace (the persistent zero borders of the LDS band, the ring's reuse between frames)
    net.set_impl("enc23_force")
    try:
        logits3, _ = net.infer(stack)
    finally:
        net.set_impl("mfma")
    np.testing.assert_array_equal(logits, logits3)


@pytest.mark.parametrize("hw", [(68, 120), (67, 120), (45, 80), (35, 60)])
def test_fused_decoder_blocks_match_separate_launches_bitwise(ctx, weights_flat, hw):
    """Decoder blocks 0..2 as one launch (one workgroup per frame, intermediates in LDS) vs the three launches of
    dec_mfma (developer switch "dec_separate"): same logits, masks and boxes, bit for bit."""
    h, w = hw
    b = 20
    stack = synth.stacked_batch(b, h, w, seed=17, streams=2)
    net = BlobNetInfer(ctx, weights_flat, h, w, max_batch=b)
    logits, mask = net.infer(stack)
    boxes, counts, _ = net.filter(stack, cc_threshold=2, max_boxes=1024)
    net.set_impl("dec_separate")
    try:
        logits2, mask2 = net.infer(stack)
        boxes2, counts2, _ = net.filter(stack, cc_threshold=2, max_boxes=1024)
    finally:
        net.set_impl("mfma")
    np.testing.assert_array_equal(logits, logits2)
    np.testing.assert_array_equal(mask, mask2)
    np.testing.assert_array_equal(counts, counts2)
    for i in range(b):
        np.testing.assert_array_equal(boxes[i, :counts[i]], boxes2[i, :counts[i]])


@pytest.mark.parametrize("hw", [(68, 120), (67, 120), (45, 80), (35, 60), (61, 100)])
@pytest.mark.parametrize("mixed_gamma", [False, True])
def test_tail_partial_logits_match_the_skip_tensor_form(ctx, weights_flat, hw, mixed_gamma):
    """Round 5: the last decoder block is linear in concat(up, skip) (decoder.py:122-134 -> final 1x1 conv, blobnet.py:44-48, no
    non-linearity between them), so the level-1 kernel computes the skip half's share of every logit from the T = 0 slice it
    holds in LDS and the fused tail adds those fp32 partial logits to its "up" half -- the level-0 skip TENSOR never crosses
    HBM.  Against the round-1..4 form (developer switch "tail_skip_tensor": the tail reads the tensor): the same fp16 operands
    and fp32 products, only the order of the fp32 additions differs -> logits equal to a few ulp of the largest term, masks
    equal wherever the logit is not within that distance of zero; both entries, odd / even grids (crop offsets, the grid row
    behind an odd image), infer() and filter() identical to each other in both forms."""
    h, w = hw
    b = 12
    flat = _mixed_gamma_weights(57) if mixed_gamma else weights_flat
    stack = synth.stacked_batch(b, h, w, seed=41, streams=2)
    frames, index = synth.carrier_batch(b, h, w, seed=41, streams=2)
    net = BlobNetInfer(ctx, flat, h, w, max_batch=b)
    logits, mask = net.infer(stack)
    got = net.filter_frames(frames, index, 1, max_boxes=2048, want_mask=True, want_logits=True)
    np.testing.assert_array_equal(got[3], logits)          # carrier-frame entry + fused tail == stacked entry + stand-alone last block
    np.testing.assert_array_equal(got[2], mask)
    net.set_impl("tail_skip_tensor")
    try:
        logits2, mask2 = net.infer(stack)
        got2 = net.filter_frames(frames, index, 1, max_boxes=2048, want_mask=True, want_logits=True)
    finally:
        net.set_impl("mfma")
    np.testing.assert_array_equal(got2[3], logits2)
    scale = float(np.abs(logits2).max())
    err = np.abs(logits - logits2)
    assert err.max() <= 4e-6 * scale + 1e-6, (err.max(), scale)
    differ = mask != mask2
    assert (np.abs(logits2)[differ] <= 4e-6 * scale + 1e-6).all()
    assert differ.mean() < 1e-3


@pytest.mark.parametrize("hw", [(68, 120), (67, 120), (45, 80), (35, 60), (61, 104), (24, 32), (128, 120)])
@pytest.mark.parametrize("mixed_gamma", [False, True])
def test_tail_row_tiles_and_ballot_planes_match_the_band_tiles(ctx, weights_flat, hw, mixed_gamma):
    """Round 6: the fused tail on ROW tiles (dec3cc_rows_mfma): a tile = 32 positions of one grid row, the four logits of a position
    are balloted straight into bboxcc's parity planes (no mask bytes in LDS, bboxcc starts at its phase B), the mask the caller asked
    for is expanded from the planes.  Against the band-tile form (developer switch "tail_band_tiles", round 5's dec3cc_mfma): the
    same products in the same order -> logits, masks, counts, boxes AND their order are bit-identical; odd / even grids (crop
    offsets, rows of one and two tiles, a partial second tile), both gamma sign classes, with and without the logits / mask outputs,
    more frames than workgroups (600 frames: the tile's border and the planes are rewritten per frame)."""
    h, w = hw
    b = 12
    flat = _mixed_gamma_weights(57) if mixed_gamma else weights_flat
    frames, index = synth.carrier_batch(b, h, w, seed=43, streams=2)
    net = BlobNetInfer(ctx, flat, h, w, max_batch=b)
    got = net.filter_frames(frames, index, 1, max_boxes=2048, want_mask=True, want_logits=True)
    lean = net.filter_frames(frames, index, 3, max_boxes=64)            # no mask, no logits, another threshold, truncation
    net.set_impl("tail_band_tiles")
    try:
        got2 = net.filter_frames(frames, index, 1, max_boxes=2048, want_mask=True, want_logits=True)
        lean2 = net.filter_frames(frames, index, 3, max_boxes=64)
    finally:
        net.set_impl("mfma")
    for a, c in zip(got[1:], got2[1:]):
        np.testing.assert_array_equal(a, c)
    np.testing.assert_array_equal(lean[1], lean2[1])
    for i in range(b):
        n = min(int(got[1][i]), 2048)                                   # (entries behind a frame's count are unspecified)
        np.testing.assert_array_equal(got[0][i, :n], got2[0][i, :n])
        n = min(int(lean[1][i]), 64)
        np.testing.assert_array_equal(lean[0][i, :n], lean2[0][i, :n])
    assert got[1].sum() > 0 and (got[2] == (got[3] > 0)).all()


def test_tail_row_tiles_more_frames_than_workgroups(ctx, weights_flat):
    h, w, b = 68, 120, 600
    frames, index = synth.carrier_batch(b, h, w, seed=47, streams=5)
    net = BlobNetInfer(ctx, weights_flat, h, w, max_batch=b)
    got = net.filter_frames(frames, index, 1, max_boxes=1024, want_mask=True)
    net.set_impl("tail_band_tiles")
    try:
        got2 = net.filter_frames(frames, index, 1, max_boxes=1024, want_mask=True)
    finally:
        net.set_impl("mfma")
    np.testing.assert_array_equal(got[1], got2[1])
    np.testing.assert_array_equal(got[2], got2[2])
    for i in range(b):
        n = min(int(got[1][i]), 1024)
        np.testing.assert_array_equal(got[0][i, :n], got2[0][i, :n])


@pytest.mark.parametrize("b", [1, 7, 33, 191, 192, 193, 255, 257])
def test_default_chain_across_batch_sizes(ctx, weights_flat, b):
    """The default kernel chain picks its forms by batch size (levels 2 + 3 as one launch from three quarters of a frame per CU
    on, the stack table by value up to 256 stacks, band plans by rounds): around every threshold the result is the one the
    two-launch chain gives, bit for bit, the two entries agree, and frame k of a batch does not depend on the batch around it."""
    h, w = 68, 120
    frames, index = synth.carrier_batch(257, h, w, seed=53, streams=3)
    net = BlobNetInfer(ctx, weights_flat, h, w, max_batch=257)
    got = net.filter_frames(frames, index[:b], 1, max_boxes=1024, want_mask=True, want_logits=True)
    net.set_impl("enc23_separate")
    try:
        got2 = net.filter_frames(frames, index[:b], 1, max_boxes=1024, want_mask=True, want_logits=True)
    finally:
        net.set_impl("mfma")
    for a, c in zip(got, got2):
        np.testing.assert_array_equal(a, c)
    # batch independence: the last frame of this batch, alone
    one = net.filter_frames(frames, index[b - 1:b], 1, max_boxes=1024, want_mask=True, want_logits=True)
    np.testing.assert_array_equal(one[3][0], got[3][b - 1])
    np.testing.assert_array_equal(one[2][0], got[2][b - 1])
    assert one[1][0] == got[1][b - 1]


def test_fused_encoder_levels_2_3_more_frames_than_workgroups(ctx, weights_flat):
    """enc23_mfma is persistent over frames (grid = min(batch, CUs)): 600 frames on 256 CUs take two to three frames per
    workgroup -- the band's borders stay zero, the ring and the store scratch are reused -- and give the bits of the two
    launches; 600 >= 3/4 of the CUs, so this is also the default chain at 68 x 120."""
    h, w, b = 68, 120, 600
    frames, index = synth.carrier_batch(b, h, w, seed=31, streams=6)
    net = BlobNetInfer(ctx, weights_flat, h, w, max_batch=b)
    got = net.filter_frames(frames, index, 1, max_boxes=1024, want_mask=True, want_logits=True)
    net.set_impl("enc23_separate")
    try:
        got2 = net.filter_frames(frames, index, 1, max_boxes=1024, want_mask=True, want_logits=True)
    finally:
        net.set_impl("mfma")
    for a, c in zip(got, got2):
        np.testing.assert_array_equal(a, c)


def test_whole_path_on_a_4k_grid(ctx, weights_flat):
    """2160p = 135 x 240 macroblocks: the fused decoder and the fused tail do not fit in LDS there, the launch plan falls
    back to one kernel per block and to bboxcc with its state in global memory -- logits within the tolerance, frames
    entry == stacked entry bit for bit, boxes == the oracle's regionprops of the mask."""
    h, w, b = 135, 240, 2
    stack = synth.stacked_batch(b, h, w, seed=3, streams=1)
    frames, index = synth.carrier_batch(b, h, w, seed=3, streams=1)
    net = BlobNetInfer(ctx, weights_flat, h, w, max_batch=b)
    logits, mask = net.infer(stack)
    ref_logits, _ = ref.blobnet_forward(weights_flat, stack, h, w)
    _check(logits, mask, ref_logits)
    boxes, counts, fmask = net.filter(stack, cc_threshold=2, max_boxes=8192, want_mask=True)
    fboxes, fcounts, ffmask, flogits = net.filter_frames(frames, index, 2, max_boxes=8192, want_mask=True, want_logits=True)
    np.testing.assert_array_equal(fmask, mask)
    np.testing.assert_array_equal(flogits, logits)
    np.testing.assert_array_equal(ffmask, mask)
    np.testing.assert_array_equal(fcounts, counts)
    rboxes, rcounts = ref.regionprops_batch(mask, 2, 8192)
    np.testing.assert_array_equal(counts, rcounts)
    for i in range(b):
        for f, g in (("left", "left"), ("top", "top"), ("width", "width"), ("height", "height"), ("area_px", "area")):
            np.testing.assert_array_equal(boxes[i, :counts[i]][f], rboxes[i, :counts[i]][g])
            np.testing.assert_array_equal(fboxes[i, :counts[i]][f], rboxes[i, :counts[i]][g])
