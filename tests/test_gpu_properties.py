"""Size-independent properties of the HIP path at BASELINE sizes (b=256, 68x120), where running the
oracle on everything would take too long for a test: conservation laws and invariances."""
import os

import numpy as np
import pytest

from cova_amd import _lib as L
from cova_amd import synth
from cova_amd.elements import BboxCc, BlobNetInfer

pytestmark = pytest.mark.gpu
H, W, B = 68, 120, 256


def _boxes(ctx, masks, thr=1, cap=2048):
    cc = BboxCc(ctx, cc_threshold=thr, max_boxes=cap)
    return cc.regionprops(masks)


@pytest.mark.parametrize("density", [0.03, 0.2, 0.45])
def test_ccl_conservation_and_bounds(ctx, density):
    masks = synth.random_masks(B, H, W, density, seed=int(density * 100))
    boxes, counts = _boxes(ctx, masks)
    for i in range(B):
        b = boxes[i, :counts[i]]
        # every foreground pixel belongs to exactly one component
        assert int(b["area_px"].sum()) == int(masks[i].sum())
        assert (b["left"] >= 0).all() and (b["top"] >= 0).all()
        assert (b["left"] + b["width"] <= W).all() and (b["top"] + b["height"] <= H).all()
        assert (b["area_px"] <= b["width"] * b["height"]).all() and (b["area_px"] >= 1).all()
        # a component's box corners rows/cols really contain foreground
        for bx in b[:5]:
            sub = masks[i, bx["top"]:bx["top"] + bx["height"], bx["left"]:bx["left"] + bx["width"]]
            assert sub[0].any() and sub[-1].any() and sub[:, 0].any() and sub[:, -1].any()


def test_ccl_threshold_is_a_filter_of_the_unfiltered_list(ctx):
    masks = synth.random_masks(B, H, W, 0.25, seed=77)
    b1, c1 = _boxes(ctx, masks, 1)
    b30, c30 = _boxes(ctx, masks, 30)
    for i in range(0, B, 17):
        full = b1[i, :c1[i]]
        keep = full[full["area_px"] >= 30]
        np.testing.assert_array_equal(keep, b30[i, :c30[i]])      # same boxes, same order


def test_ccl_block_translation_invariance(ctx):
    """Shifting a mask by one 2x2 block (2 px) right/down shifts every box and keeps the order."""
    rng = np.random.default_rng(5)
    m = np.zeros((64, H, W), np.uint8)
    m[:, 2:H - 4, 2:W - 4] = (rng.random((64, H - 6, W - 6)) < 0.2)
    shifted = np.zeros_like(m)
    shifted[:, 2:, 2:] = m[:, :-2, :-2]
    b0, c0 = _boxes(ctx, m)
    b1, c1 = _boxes(ctx, shifted)
    np.testing.assert_array_equal(c0, c1)
    for i in range(64):
        a, b = b0[i, :c0[i]], b1[i, :c1[i]]
        np.testing.assert_array_equal(a["left"] + 2, b["left"])
        np.testing.assert_array_equal(a["top"] + 2, b["top"])
        np.testing.assert_array_equal(a["area_px"], b["area_px"])


def test_ccl_is_deterministic_over_repeated_launches(ctx):
    masks = synth.random_masks(B, H, W, 0.3, seed=9)
    ref_b, ref_c = _boxes(ctx, masks)
    for _ in range(5):
        b, c = _boxes(ctx, masks)
        np.testing.assert_array_equal(c, ref_c)
        np.testing.assert_array_equal(b, ref_b)


def test_blobnet_full_batch_determinism_and_permutation(ctx, weights_flat):
    """Frames are independent: permuting the batch permutes the logits bit for bit, and two runs of the
    same batch agree bit for bit (no atomics / no order dependence in the BlobNet kernels)."""
    stack = synth.stacked_batch(B, H, W, seed=123, streams=8)
    net = BlobNetInfer(ctx, weights_flat, H, W, max_batch=B)
    l0, m0 = net.infer(stack)
    l1, m1 = net.infer(stack)
    np.testing.assert_array_equal(l0, l1)
    perm = np.random.default_rng(0).permutation(B)
    l2, m2 = net.infer(stack[perm])
    np.testing.assert_array_equal(l0[perm], l2)
    np.testing.assert_array_equal(m0[perm], m2)
    assert 0.0 < m0.mean() < 1.0


def test_fused_counts_match_separate_path_full_batch(ctx, weights_flat):
    stack = synth.stacked_batch(B, H, W, seed=321, streams=8)
    net = BlobNetInfer(ctx, weights_flat, H, W, max_batch=B)
    boxes, counts, mask = net.filter(stack, cc_threshold=1, max_boxes=2048, want_mask=True)
    b2, c2 = _boxes(ctx, mask)
    np.testing.assert_array_equal(counts, c2)
    # checksum of checksums (over the counts[i] boxes of every frame: entries behind a frame's count are unspecified -- the check
    # used to sum the whole array and passed only while the staging buffer happened to be fresh, zeroed memory)
    assert sum(int(boxes[i, :counts[i]]["area_px"].sum()) for i in range(B)) == int(mask.sum())
    for i in range(0, B, 31):
        np.testing.assert_array_equal(boxes[i, :counts[i]], b2[i, :c2[i]])


def test_kernel_chains_agree_over_many_geometries(ctx):
    """Odd and even heights, widths from 16 to 160 macroblocks: the fused decoder vs one launch per block, the stacked vs
    the carrier-frame entry, infer vs filter -- the same logits, masks, counts and boxes bit for bit, all finite (crop
    parities, band plans and LDS fits differ per geometry; a sweep of 192 geometries up to 135 x 240 was clean)."""
    import numpy as np
    from cova_amd import synth, weights as W
    from cova_amd.elements import BlobNetInfer
    flat = W.random_init(7)
    for h, w in [(16, 16), (17, 20), (21, 36), (26, 44), (31, 28), (36, 60), (41, 100), (46, 132), (51, 80), (56, 120),
                 (61, 16), (66, 160), (71, 60), (76, 44), (81, 120), (96, 100)]:
        net = BlobNetInfer(ctx, flat, h, w, max_batch=3)
        stack = synth.stacked_batch(3, h, w, seed=h * 1000 + w, streams=1)
        frames, index = synth.carrier_batch(3, h, w, seed=h * 1000 + w, streams=1)
        lg, mk = net.infer(stack)
        b1, c1, m1 = net.filter(stack, cc_threshold=1, max_boxes=4096, want_mask=True)
        fb, fc, fm, fl = net.filter_frames(frames, index, 1, max_boxes=4096, want_mask=True, want_logits=True)
        net.set_impl("dec_separate")
        try:
            lg2, mk2 = net.infer(stack)
            b2, c2, _ = net.filter(stack, cc_threshold=1, max_boxes=4096, want_mask=True)
        finally:
            net.set_impl("mfma")
        assert np.isfinite(lg).all(), (h, w)
        for a, b in ((lg, lg2), (mk, mk2), (m1, mk), (c1, c2), (fl, lg), (fm, mk), (fc, c1)):
            np.testing.assert_array_equal(a, b, err_msg=f"{h}x{w}")
        for i in range(3):
            np.testing.assert_array_equal(b1[i, :c1[i]], b2[i, :c1[i]], err_msg=f"{h}x{w}")
            np.testing.assert_array_equal(fb[i, :c1[i]], b1[i, :c1[i]], err_msg=f"{h}x{w}")


@pytest.mark.gpu
def test_bench_two_ranks_rehearsed_on_this_gpu():
    """The N > 1 path of bench.py on real hardware, every round: two ranks started by bench.py itself as a
    child process (before anything in it touches the GPU), gloo control plane, both on this box's GPU.  Checks what an 8-GPU
    node will rely on -- world size, disjoint stream sets, distinct inputs, per-rank pinning report, a finite aggregate -- and
    that the line says it is NOT a scaling result."""
    import json
    import math
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "5",
                        "--warmup", "2", "--min-warmup-s", "0.05", "--no-extra-legs", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE aggregate line on stdout"
    line = lines[0]
    assert line["n_gpus"] == 2 and "NOT a scaling result" in line["rehearsal"]
    assert math.isfinite(line["value"]) and line["value"] > 0 and line["scaling"] == "weak"
    assert "no RCCL" in line["control_plane"]
    ranks = line["ranks"]
    assert [x["rank"] for x in ranks] == [0, 1]
    s0, s1 = set(ranks[0]["streams"]), set(ranks[1]["streams"])
    assert s0 and s1 and not (s0 & s1) and s0 | s1 == set(range(16))          # stream s -> rank s mod 2
    assert ranks[0]["input_seed"] != ranks[1]["input_seed"]
    for x in ranks:
        assert x["frames_per_s_own_clock"] > 0 and x["cpus"]
    if all(x["pinned"] for x in ranks) and len(ranks[0]["cpus"]) + len(ranks[1]["cpus"]) > 2:
        assert not (set(ranks[0]["cpus"]) & set(ranks[1]["cpus"])), "two ranks on one NUMA node split its cores"
    # the aggregate is both ranks' frames over the slowest rank's region
    assert abs(line["value"] - 2 * line["config"]["batch_per_gpu"] * line["steps"] / (line["ms_per_step"] * 1e-3 * line["steps"])) / line["value"] < 1e-3
    per_rank = [json.loads(l)["bench_rank"] for l in r.stderr.splitlines() if l.startswith('{"bench_rank"')]   # (one write per line: bench.rank_line)
    assert sorted(x["rank"] for x in per_rank) == [0, 1]


@pytest.mark.gpu
def test_sysfs_gpu_enumeration_agrees_with_the_runtime(ctx):
    """A rank pins itself to its GPU's NUMA node BEFORE it touches the GPU: the PCI address comes from the KFD
    topology in sysfs (cova_amd.multigpu.kfd_gpu_bus_ids), not from a HIP call.  Where the topology is readable its device order
    must be the runtime's: device d of the list = hipDeviceGetPCIBusId(d), for every device the runtime shows."""
    import ctypes as C
    from cova_amd import _lib as L
    from cova_amd.multigpu import gpu_numa, gpu_numa_sysfs, kfd_gpu_bus_ids
    ids = kfd_gpu_bus_ids()
    n = C.c_int(0)
    assert L.lib().covahip_device_count(C.byref(n)) == 0 and n.value >= 1
    if not ids:
        pytest.skip("KFD topology not readable on this box: bench.py falls back to asking a child process")
    assert len(ids) == n.value, (ids, n.value)
    for d in range(n.value):
        buf = C.create_string_buffer(32)
        assert L.lib().covahip_device_pci_bus_id(d, buf, len(buf)) == 0
        assert buf.value.decode().lower() == ids[d], (d, buf.value, ids)
        node, cpus, dev = gpu_numa_sysfs(d)
        assert dev == d and (node, cpus) == gpu_numa(d)
