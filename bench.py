#!/usr/bin/env python3
"""bench.py -- compressed-domain frames/sec (BlobNet + bboxcc) at 1080p, b=256, N GPUs.

One "step" = one pass of the hot path (temporal stacking -> BlobNet forward -> threshold -> bboxcc) that
produces the boxes of 256 frames of a synthetic 68x120 macroblock grid, inputs already resident in HBM
(BASELINE.json configs[2]).  Default entry point: covahip_filter_forward_frames -- the carrier frames of 8 streams
(8 x 35 frames of [mb_type, mv_x, mv_y] records) plus the table of which four frames make up each of the 256
stacks; metapreprocess' stacking is an index gather on the GPU and the first encoder level runs once per carrier
frame.  `--entry stack` times covahip_filter_forward on the 256 pre-stacked tensors instead (round 1's workload;
the same 256 stacks, bit-identical results); whichever is not timed is reported as an extra key.  With --gpus N
every rank runs the same batch size on its own GPU (frames/streams are independent: no collective on the data
path), the timed region is bracketed by barrier + synchronize on both sides, the MAX over ranks is taken and
rank 0 prints ONE JSON line.

The K timed steps are enqueued back to back on a ctx with `--lanes` (default 2) batches in flight: step k+1 runs on
the other lane (own HIP stream and activation workspace) and its launches fill the ramps and tails of step k's --
the way the reference keeps sixteen BlobNet engines busy on one GPU (experiment/cova/config.yaml:33-34).  `value` =
frames of all K steps / wall time; `ms_per_step_one_lane` is a step on its own (--lanes 1 times that).

    python bench.py                       # N=1, defaults finish in a few minutes
    python bench.py --gpus N              # starts the N ranks itself (torch.distributed.run as a child process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Beside `value` the line carries, as extra keys measured outside the timed region on rank 0: the other entry point,
the PCIe-inclusive rates (pageable host call, pinned pipelined covahip_pipe, through the GStreamer batching
element), BASELINE configs[1] (BlobNet only, b=32) and the reference's default geometry (45x80, b=512), a leg with
mixed-sign BN gammas (the ALLPOS=false kernel variants), and the CPU baselines of SURVEY.md section 8(d).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H_MB, W_MB, T = 68, 120, 4        # 1080p macroblock grid (BASELINE.json), timestep 4
BATCH = 256
CC_THRESHOLD = 1                   # experiment/cova/config.yaml:59
MAX_BOXES = 2048                   # >= ceil(H/2)*ceil(W/2) = 2040: no truncation possible
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0          # dense fp16/bf16 MFMA peak


def level_dims(h=H_MB, w=W_MB):
    hs, ws = [h], [w]
    for _ in range(4):
        hs.append((hs[-1] + 1) // 2); ws.append((ws[-1] + 1) // 2)
    return hs, ws


def kernel_macs_per_frame():
    """Algorithmic MACs per frame of each BlobNet kernel at 68x120."""
    hs, ws = level_dims()
    enc_c = [3, 16, 32, 64, 128]
    out = {}
    for i in range(4):
        conv = T * hs[i] * ws[i] * 9 * enc_c[i] * enc_c[i + 1]
        tmix = hs[i + 1] * ws[i + 1] * enc_c[i + 1] * 32
        out[f"enc{i}_mfma"] = conv + tmix
    dec_ci, dec_co = [128, 128, 64, 32], [64, 32, 16, 16]
    for j in range(3):
        out[f"dec{j}_mfma"] = hs[4 - j] * ws[4 - j] * 16 * dec_ci[j] * dec_co[j]
    out["dec3_final_mfma"] = hs[1] * ws[1] * 16 * dec_ci[3] * dec_co[3] + H_MB * W_MB * 16
    out["dec012_mfma"] = out["dec0_mfma"] + out["dec1_mfma"] + out["dec2_mfma"]      # the three blocks in one launch
    out["enc23_mfma"] = out["enc2_mfma"] + out["enc3_mfma"]                          # levels 2 + 3 in one launch (round 5)
    out["dec3_bboxcc_fused"] = out["dec3_final_mfma"]
    return out


def dec3_fold_macs_per_frame():
    """(algorithmic, executed) MACs per frame of the last decoder block + final 1x1 conv: the kernels run them FOLDED into one
    32 -> 1 transposed conv (no non-linearity between them): 4 parities x 4 taps x 32 channels per grid position."""
    hs, ws = level_dims()
    alg = hs[1] * ws[1] * 16 * 32 * 16 + H_MB * W_MB * 16
    return alg, (hs[1] + 1) * (ws[1] + 1) * 4 * 4 * 32


def kernel_bytes_per_frame():
    """Compulsory HBM bytes per frame of each BlobNet kernel at 68x120 when every level is its own kernel:
    input tensor(s) read once + output tensor written once (fp16 activations, u8 input / mask; weights excluded)."""
    hs, ws = level_dims()
    enc_c = [3, 16, 32, 64, 128]
    out = {"enc0_mfma": T * hs[0] * ws[0] * 4 + T * hs[1] * ws[1] * enc_c[1] * 2}
    for i in range(1, 4):
        to = 1 if i == 3 else T
        out[f"enc{i}_mfma"] = T * hs[i] * ws[i] * enc_c[i] * 2 + to * hs[i + 1] * ws[i + 1] * enc_c[i + 1] * 2
    dec_c1, dec_c2, dec_co = [0, 64, 32, 16], [128, 64, 32, 16], [64, 32, 16, 16]
    for j in range(3):
        out[f"dec{j}_mfma"] = hs[4 - j] * ws[4 - j] * (dec_c1[j] + dec_c2[j]) * 2 + hs[3 - j] * ws[3 - j] * dec_co[j] * 2
    out["dec3_final_mfma"] = hs[1] * ws[1] * (dec_c1[3] + dec_c2[3]) * 2 + H_MB * W_MB
    # one launch for blocks 0..2: the skip inputs of the three blocks in, block 2's output out (the intermediates stay in LDS)
    out["dec012_mfma"] = sum(hs[4 - j] * ws[4 - j] * dec_c2[j] * 2 for j in range(3)) + hs[1] * ws[1] * dec_co[2] * 2
    # levels 2 + 3 in one launch: level 2's input in; the T = 0 slice of level 2's output (decoder skip) and level 3's T = 0 output out
    out["enc23_mfma"] = T * hs[2] * ws[2] * enc_c[2] * 2 + hs[3] * ws[3] * enc_c[3] * 2 + hs[4] * ws[4] * enc_c[4] * 2
    out["dec3_bboxcc_fused"] = out["dec3_final_mfma"]
    return out


PMC_KERNEL_KEYS = {"enc0p_mfma": "enc0p_mfma", "enc1_mfma": "enc1_mfma<", "enc2_mfma": "enc_mfma<32, 64",
                   "enc3_mfma": "enc_mfma<64, 128", "enc23_mfma": "enc23_mfma", "dec0_mfma": "dec_mfma<0, 128", "dec1_mfma": "dec_mfma<64, 64",
                   "dec2_mfma": "dec_mfma<32, 32", "dec012_mfma": "dec012_mfma", "dec3_final_mfma": "dec_mfma<16, 16", "dec3_bboxcc_fused": "dec3cc_",
                   "bboxcc_kernel": "bboxcc_kernel"}


def committed_traffic(kernel, entry="stack"):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary (profiles/r*/traffic.json for
    the stacked entry, traffic_frames.json for the carrier-frame entry: FETCH_SIZE/WRITE_SIZE passes of
    tools/collect_profiles.sh, b=256)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic_frames.json" if entry == "frames" else "traffic.json")))
    if not files or kernel not in PMC_KERNEL_KEYS:
        return None, None
    data = json.load(open(files[-1]))
    for name, v in data.items():
        if name.startswith(PMC_KERNEL_KEYS[kernel]):
            return v["hbm_bytes_per_launch"], os.path.relpath(files[-1], ROOT)
    return None, None


# ------------------------------------------------------------------------------------------------ CPU baselines
def cpu_quota(text: str):
    """CPUs granted by a cgroup v2 `cpu.max` line ("<quota> <period>" or "max <period>"): None when unlimited or unreadable."""
    try:
        quota, period = text.split()[:2]
        if quota == "max" or int(quota) <= 0 or int(period) <= 0:
            return None
        return max(1, int(quota) // int(period))
    except (ValueError, IndexError):
        return None


def effective_cores() -> int:
    """CPUs this process may actually use: the smallest of os.cpu_count(), the affinity mask and the cgroup's CPU quota (a GPU
    box shows all 256 hardware threads to os.cpu_count() but grants a share of them -- /sys/fs/cgroup/cpu.max "1600000 100000"
    = 16; OpenMP on 256 threads under that quota spends its time being throttled)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    texts = []
    try:
        texts.append(open("/sys/fs/cgroup/cpu.max").read())                                   # cgroup v2
    except OSError:
        pass
    try:
        texts.append(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read().strip() + " " +
                     open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip())            # cgroup v1
    except OSError:
        pass
    for t in texts:
        q = cpu_quota(t)
        if q:
            n = min(n, q)
    return max(1, n)


def cpu_baseline(flat, stack_sample, target_seconds=8.0):
    """CPU baseline on this host, on a bounded sample of the same workload.  Two ports of the reference
    path are timed and the FASTER one is reported as `value` (the reference itself -- Rust + OpenCV +
    TensorRT -- cannot be built in this image, and onnxruntime is not installed):
      * the C oracle (oracle/blobnet_ref.c, OpenMP over frames, fp32) + oracle regionprops (1 thread);
      * the same graph in PyTorch-CPU (oneDNN convs, all cores) + oracle regionprops.
    Both are straightforward, untuned ports: a reported baseline, not a target."""
    cores = effective_cores()
    from oracle import ref
    n0 = min(len(stack_sample), max(2, cores))
    t0 = time.perf_counter()
    ref.blobnet_forward(flat, stack_sample[:n0], H_MB, W_MB)
    dt = time.perf_counter() - t0
    n = int(min(len(stack_sample), max(n0, target_seconds / max(dt / n0, 1e-6))))
    n = max(n0, (n // n0) * n0)
    passes = max(1, int(target_seconds / max(dt / n0 * n, 1e-6)))
    t0 = time.perf_counter()
    for _ in range(passes):
        _, mask = ref.blobnet_forward(flat, stack_sample[:n], H_MB, W_MB)
    t_net = (time.perf_counter() - t0) / passes
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        ref.regionprops_batch(mask, CC_THRESHOLD, MAX_BOXES)
    t_cc = (time.perf_counter() - t0) / reps           # single thread, as the reference element runs
    oracle_fps = n / (t_net + t_cc)
    out = {"unit": "frames/s", "cores": cores, "kind": "port", "tuning": "untuned ports of the reference path",
           "oracle_c_blobnet_frames_per_s": round(n / t_net, 2), "oracle_bboxcc_frames_per_s": round(n / t_cc, 1),
           "oracle_c_combined_frames_per_s": round(oracle_fps, 2)}
    best, which = oracle_fps, "C oracle (OpenMP)"
    try:
        import torch
        from tests import torch_blobnet as tb
        tthreads = min(cores, 32)        # more threads than that slow these small convolutions down
        torch.set_num_threads(tthreads)
        nt = min(len(stack_sample), 64)
        with torch.no_grad():
            tb.forward(flat, stack_sample[:8], H_MB, W_MB)                      # warm-up
            t0 = time.perf_counter()
            reps_t = 0
            while time.perf_counter() - t0 < target_seconds / 2:
                tb.forward(flat, stack_sample[:nt], H_MB, W_MB)
                reps_t += 1
            t_torch = (time.perf_counter() - t0) / reps_t
        torch_fps = nt / (t_torch + t_cc * nt / n)
        out["torch_cpu_threads"] = tthreads
        out["torch_cpu_blobnet_frames_per_s"] = round(nt / t_torch, 2)
        out["torch_cpu_combined_frames_per_s"] = round(torch_fps, 2)
        if torch_fps > best:
            best, which = torch_fps, "PyTorch-CPU fp32"
    except Exception as e:  # torch missing on the box: keep the C port
        out["torch_cpu_error"] = repr(e)[:200]
    out["value"] = round(best, 2)
    out["sample"] = (f"{n} frames of the b=256 68x120 workload for the C oracle, 64 for PyTorch-CPU; BlobNet fp32 on {cores} "
                     f"host threads + oracle bboxcc on 1 thread (serial sum); reported value = {which}")
    return out


def _blob_masks(n, seed):
    """Sparse-blob masks with temporally coherent objects (what a trained BlobNet emits): a few ellipses that move."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H_MB, 0:W_MB]
    objs = [(rng.uniform(0, H_MB), rng.uniform(0, W_MB), rng.uniform(1.5, 7), rng.uniform(1.5, 7),
             rng.uniform(-0.4, 0.4), rng.uniform(-0.6, 0.6), int(rng.integers(0, n // 2)), int(rng.integers(n // 2, n)))
            for _ in range(6)]
    m = np.zeros((n, H_MB, W_MB), np.uint8)
    for i in range(n):
        for cy, cx, ry, rx, vy, vx, a, b in objs:
            if a <= i <= b:
                m[i] |= ((((yy - cy - vy * (i - a)) / ry) ** 2 + ((xx - cx - vx * (i - a)) / rx) ** 2) <= 1).astype(np.uint8)
    return m


def _track_stream(args):
    """bboxcc (C oracle regionprops, what the Rust element computes through OpenCV) + SORT (the C++ port behind the
    C-ABI, sort/src/lib.rs:134-187) over one stream on ONE thread, as the reference's per-branch elements run."""
    seed, n = args
    from cova_amd import elements as E
    from cova_amd import _lib as L
    from oracle import ref
    masks = _blob_masks(n, seed)
    sort = E._SortHandle(60, 30, 0.1)          # experiment/cova/launch.py:43-44, config.yaml:67
    t0 = time.perf_counter()
    t_sort = 0.0
    for i in range(n):
        boxes, cnt = ref.regionprops(masks[i], CC_THRESHOLD, 256)
        bx = np.zeros(cnt, dtype=L.BOX_DTYPE)
        for f, g in (("left", "left"), ("top", "top"), ("width", "width"), ("height", "height"), ("area_px", "area")):
            bx[f] = boxes[:cnt][g]
        dets = E.boxes_to_bbox(bx)
        t1 = time.perf_counter()
        sort.update(dets, i * (1_000_000_000 // 30))
        t_sort += time.perf_counter() - t1
    return time.perf_counter() - t0, t_sort


def cpu_tracking_baseline(frames_per_stream=600):
    """SURVEY.md section 8(d) CPU leg (1): bboxcc + sorttracker on the host, one thread per stream -- at one thread and
    at N = min(cores, 32) independent streams in N processes."""
    import multiprocessing as mp
    cores = effective_cores()
    out = {}
    try:
        # worker processes, forked before this process has loaded the HIP library or touched the GPU
        n = max(1, min(cores, 32))
        with mp.get_context("fork").Pool(n) as pool:
            t_all, t_sort = pool.map(_track_stream, [(11, frames_per_stream)])[0]
            out["bboxcc_sort_1thread_frames_per_s"] = round(frames_per_stream / t_all, 1)
            out["sort_update_frames_per_s"] = round(frames_per_stream / max(t_sort, 1e-9), 1)
            t0 = time.perf_counter()
            pool.map(_track_stream, [(100 + s, frames_per_stream) for s in range(n)])
            out["bboxcc_sort_nstreams_frames_per_s"] = round(n * frames_per_stream / (time.perf_counter() - t0), 1)
        out["bboxcc_sort_nstreams"] = n
        out["tracking_sample"] = (f"{frames_per_stream} sparse-blob 68x120 masks per stream; regionprops = C oracle, SORT = the C++ port "
                                  "behind the C-ABI (maxage 60, minhits 30, iou 0.1), Python glue included")
    except Exception as e:
        out["tracking_error"] = repr(e)[:200]
    return out


# ------------------------------------------------------------------------------------------------ extra legs (rank 0)
def board_power_file():
    """sysfs power sensor (microwatts) and cap of THIS process's GPU: hipDeviceGetPCIBusId(0) -> /sys/bus/pci/devices/<id>/hwmon."""
    import ctypes
    import glob
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, 0) != 0:
            return None, None
        base = "/sys/bus/pci/devices/" + buf.value.decode().lower() + "/hwmon/hwmon*/"
        f = (glob.glob(base + "power1_input") or glob.glob(base + "power1_average") or [None])[0]
        cap = (glob.glob(base + "power1_cap") or [None])[0]
        return f, (int(open(cap).read()) / 1e6 if cap else None)
    except (OSError, ValueError):
        return None, None


def board_power_under(ctx, step, seconds=2.0, settle=0.8):
    """Mean / max board power (W) while `step` loops for `seconds` (samples of the first `settle` seconds dropped), and the
    step time of that loop.  None when the sensor is not readable."""
    import threading
    path, cap = board_power_file()
    if not path:
        return None
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            try:
                samples.append((time.perf_counter(), int(open(path).read()) / 1e6))
            except (OSError, ValueError):
                pass
            time.sleep(0.01)
    th = threading.Thread(target=sampler, daemon=True)
    ctx.sync()
    t0 = time.perf_counter()
    th.start()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(64):
            step()
        n += 64
        ctx.sync()
    t1 = time.perf_counter()
    stop.set()
    th.join()
    w = [p for (t, p) in samples if t - t0 >= settle]
    if not w:
        return None
    ms = (t1 - t0) / n * 1e3
    return {"mean_w": round(sum(w) / len(w), 1), "max_w": round(max(w), 1), "cap_w": cap, "samples": len(w),
            # at the cap = the mean within 2 % of it (the firmware holds the board there by lowering the clock)
            "at_cap": bool(cap and sum(w) / len(w) >= 0.98 * cap),
            "ms_per_step_in_this_loop": round(ms, 4), "mj_per_step": round(sum(w) / len(w) * ms, 1)}



def timed_steps(ctx, fn, steps, warmup=5, warm_s=0.15):
    """ms per call of `fn` over `steps` calls, after `warmup` calls and at least `warm_s` seconds of them (the clocks of an idle
    chip take that long to come up: 50 steps right after start-up measured a step on its own 12 % slow)."""
    t_w = time.perf_counter()
    n = 0
    while n < warmup or time.perf_counter() - t_w < warm_s:
        fn()
        n += 1
        if n % 32 == 0:
            ctx.sync()
    ctx.sync()
    ctx.timer_start(2)
    for _ in range(steps):
        fn()
    ctx.timer_stop(2)
    return ctx.timer_ms(2) / steps


def extra_legs(ctx, flat, steps):
    """Other configurations, each on a model of its own on this rank's GPU.  Run BEFORE the main model is loaded (a ctx
    holds one model)."""
    from cova_amd import synth, weights as W
    from cova_amd.elements import BlobNetInfer
    out = {}
    # BASELINE configs[1]: BlobNet forward only, b = 32, 120x68x4, fp16
    net = BlobNetInfer(ctx, flat, H_MB, W_MB, max_batch=32)
    st = synth.stacked_batch(32, H_MB, W_MB, seed=5, streams=4)
    d_st, d_mask = ctx.malloc(st.nbytes), ctx.malloc(32 * H_MB * W_MB)
    ctx.h2d(d_st, st)
    ms = timed_steps(ctx, lambda: net.infer_device(d_st, 32, None, d_mask), steps)
    out["config_blobnet_only_b32_68x120"] = {"frames_per_s": round(32 / ms * 1e3, 1), "ms_per_step": round(ms, 4)}
    # ... and with batches in flight (filter entry on device pointers is what lanes apply to; bboxcc rides in the last launch)
    d_bx = [ctx.malloc(32 * 512 * 20) for _ in range(4)]
    d_ct = [ctx.malloc(32 * 4) for _ in range(4)]
    tn = [0]

    def step32():
        tn[0] = (tn[0] + 1) % 4
        net.filter_device(d_st, 32, CC_THRESHOLD, d_bx[tn[0]], d_ct[tn[0]], 512)
    for nl in (1, 2, 4):
        ctx.set_lanes(nl)
        ms = timed_steps(ctx, step32, 2 * steps * nl, 8)
        out["config_blobnet_only_b32_68x120"][f"filter_ms_per_step_{nl}_lanes"] = round(ms, 4)
        out["config_blobnet_only_b32_68x120"][f"filter_frames_per_s_{nl}_lanes"] = round(32 / ms * 1e3, 1)
    ctx.set_lanes(1)
    for p in d_bx + d_ct:
        ctx.free(p)
    ctx.free(d_st); ctx.free(d_mask)
    # the reference's own default: 720p grid 45x80, batch 512 (model/tasks.py:44-49)
    net = BlobNetInfer(ctx, flat, 45, 80, max_batch=512)
    st = synth.stacked_batch(512, 45, 80, seed=6, streams=8)
    d_st, d_b, d_c = ctx.malloc(st.nbytes), ctx.malloc(512 * 920 * 20), ctx.malloc(512 * 4)
    ctx.h2d(d_st, st)
    ms = timed_steps(ctx, lambda: net.filter_device(d_st, 512, CC_THRESHOLD, d_b, d_c, 920), steps)
    out["config_filter_b512_45x80"] = {"frames_per_s": round(512 / ms * 1e3, 1), "ms_per_step": round(ms, 4)}
    for p in (d_st, d_b, d_c):
        ctx.free(p)
    # mixed-sign BN gammas: the ALLPOS=false kernel variants a trained model with a negative gamma takes
    wts = W.unflatten(flat.copy())
    for i in range(4):
        wts[f"enc{i}.bn.gamma"][::2] *= -1.0
    net = BlobNetInfer(ctx, W.flatten(wts), H_MB, W_MB, max_batch=BATCH)
    st = synth.stacked_batch(BATCH, H_MB, W_MB, seed=7, streams=8)
    frg, idg = synth.carrier_batch(BATCH, H_MB, W_MB, seed=7, streams=8)
    d_fg = ctx.malloc(frg.nbytes)
    ctx.h2d(d_fg, frg)
    og = [(ctx.malloc(BATCH * MAX_BOXES * 20), ctx.malloc(BATCH * 4)) for _ in range(2)]
    tg = [0]

    def stepg():
        tg[0] = 1 - tg[0]
        net.filter_frames_device(d_fg, frg.shape[0], idg, BATCH, CC_THRESHOLD, og[tg[0]][0], og[tg[0]][1], MAX_BOXES)
    ms1 = timed_steps(ctx, stepg, steps)
    ctx.set_lanes(2)
    ms2 = timed_steps(ctx, stepg, 2 * steps, 6)
    ctx.set_lanes(1)
    out["gamma_sign_mixed_b256_68x120"] = {"frames_per_s": round(BATCH / ms2 * 1e3, 1), "ms_per_step": round(ms2, 4),
                                           "ms_per_step_one_lane": round(ms1, 4), "entry": "frames",
                                           "note": "every other BN gamma of every encoder level negative (the med3 pooling epilogue)"}
    ctx.free(d_fg)
    for o in og:
        ctx.free(o[0]); ctx.free(o[1])
    # the timed workload with a representative weight set (cova_amd.weights.blob_like: a few connected blobs per frame instead of
    # the ~500 one-macroblock components the seed-1234 random weights of `value` emit): bboxcc inside the fused tail has less to do
    net = BlobNetInfer(ctx, W.blob_like(7), H_MB, W_MB, max_batch=BATCH)
    frames, index = synth.carrier_batch(BATCH, H_MB, W_MB, seed=8, streams=8)
    d_fr = ctx.malloc(frames.nbytes)
    ctx.h2d(d_fr, frames)
    outs = [(ctx.malloc(BATCH * 256 * 20), ctx.malloc(BATCH * 4)) for _ in range(2)]
    turn = [0]

    def step():
        k = turn[0] = 1 - turn[0]
        net.filter_frames_device(d_fr, frames.shape[0], index, BATCH, CC_THRESHOLD, outs[k][0], outs[k][1], 256)
    ms1 = timed_steps(ctx, step, steps)
    ctx.set_lanes(2)
    ms2 = timed_steps(ctx, step, 2 * steps)
    ctx.set_lanes(1)
    cnt = np.zeros(BATCH, dtype=np.int32)
    ctx.d2h(cnt, outs[0][1])
    out["blob_like_weights_b256_68x120"] = {"frames_per_s": round(BATCH / ms2 * 1e3, 1), "ms_per_step": round(ms2, 4),
                                            "ms_per_step_one_lane": round(ms1, 4), "boxes_per_frame_mean": round(float(cnt.mean()), 2)}
    ctx.free(d_fr)
    for o in outs:
        ctx.free(o[0]); ctx.free(o[1])
    return out


def pipelined_host_rate(net, frames, index, steps):
    """PCIe-inclusive, carrier frames in / packed boxes out through covahip_pipe (pinned slots, three HIP streams).  The frames
    cross PCIe as two-byte records (covahip_carrier_pack, what the batching element does too); "four_byte_frames" is the same with
    the decoder's four bytes per macroblock."""
    from cova_amd.elements import FilterPipe, pack_frames
    B, nf = index.shape[0], frames.shape[0]
    NSLOT = 4      # one more slot than lanes (round 6; rounds 2-5: three)
    res = {"slots": NSLOT}
    for packed, fill in ((True, False), (True, True), (False, False)):
      pipe = FilterPipe(net, max_batch=B, max_frames=nf, max_boxes=MAX_BOXES, n_slots=NSLOT, packed=packed)
      src = pack_frames(frames) if packed else frames
      for fill in (fill,):
        slots = []
        for _ in range(NSLOT):                 # every slot holds the batch once; warms the plan and the speculative copy size
            slot, pf, pi = pipe.acquire()
            pf[:nf] = src; pi[:B] = index
            pipe.submit(slot, nf, B, CC_THRESHOLD)
            slots.append(slot)
        for slot in slots:
            pipe.collect(slot)
        t0 = time.perf_counter()
        inflight = []
        for _ in range(steps):
            acq = pipe.acquire()
            while acq is None:
                pipe.collect(inflight.pop(0))
                acq = pipe.acquire()
            slot, pf, pi = acq
            if fill:      # the element's per-frame work on ONE thread: pack every carrier frame into the slot
                pf[:nf] = pack_frames(frames); pi[:B] = index
            pipe.submit(slot, nf, B, CC_THRESHOLD)
            inflight.append(slot)
        for slot in inflight:
            pipe.collect(slot)
        key = "four_byte_frames_slots_prefilled" if not packed else "with_host_fill_one_thread" if fill else "slots_prefilled"
        res[key] = round(steps * B / (time.perf_counter() - t0), 1)
      pipe.acquire()   # releases the held result views
      pipe.close()
    return res


def element_rate(script="element_bench.sh", args=("20000", "8", str(CC_THRESHOLD)), env=None):
    """frames/s through the GStreamer elements (tools/element_bench.sh: the batching element alone; tools/chain_bench.sh: the
    batching element + one cova element per stream = BASELINE config 4); a child process with its own ctx."""
    try:
        r = subprocess.run(["bash", os.path.join(ROOT, "tools", script), *args],
                           capture_output=True, text=True, timeout=180, env=dict(os.environ, **(env or {})))
        for line in r.stdout.splitlines():
            if line.startswith("{"):
                return json.loads(line)
        return {"error": (r.stderr or r.stdout)[-200:]}
    except Exception as e:
        return {"error": repr(e)[:200]}


def rank_line(mine):
    """One rank's object on stderr as ONE write (line + newline together: eight ranks share the pipe, and print() sends the newline
    separately -- two ranks' lines then run into each other)."""
    sys.stderr.write(json.dumps({"bench_rank": mine}) + "\n")
    sys.stderr.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--entry", choices=["frames", "stack"], default="frames",
                    help="entry point of the timed step: carrier frames + stack table (default) or pre-stacked tensors")
    ap.add_argument("--lanes", type=int, default=3,
                    help="batches in flight per GPU (covahip_ctx_set_lanes): 1 = one step after the other on one stream")
    ap.add_argument("--min-warmup-s", type=float, default=0.3, help="warm up for at least this long on top of --warmup steps")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="developer rehearsal of the multi-rank path on a box with fewer GPUs than ranks: rank r uses GPU r %% (GPUs present); "
                         "the line then says so and its value is NOT a scaling result")
    ap.add_argument("--control-plane-only", action="store_true",
                    help="rehearsal of the N-rank CONTROL path alone, without a GPU: rendezvous, stream partition, NUMA / core split, "
                         "barrier, MAX over ranks, gather, the aggregate line's shape -- no kernel runs and the line carries no rate")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="only the timed workload (profiling runs)")
    args = ap.parse_args()
    os.environ.setdefault("OMP_NUM_THREADS", str(effective_cores()))   # before anything loads libgomp (the oracle, torch)

    # ---- --gpus N without a launcher: start the N ranks ourselves, as a CHILD (this process has not touched the GPU and
    # never will), relay its output and exit with its code
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, env=dict(os.environ)).returncode)
    if env_world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: launch one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {args.gpus} ...) or let --gpus start them")

    # ---- host-only legs first, before this process loads the HIP library: the GStreamer element bench (a child
    # process with a GPU context of its own) and the CPU tracking baseline (forked workers)
    pre = {}
    if env_world == 1:
        if not args.no_extra_legs:
            pre["through_gstreamer_elements"] = element_rate()
            # BASELINE config 4: metapreprocess -> BlobNet -> bboxcc -> cova (embedded SORT + GoP filter) per stream, the
            # experiment's tracker parameters, blob-like weights (a few boxes per frame, as a trained BlobNet gives)
            # (round 5: the streams hand `blobnetfilter` packed two-byte records -- caps application/x-cova-records, what
            # `h264entropydec records=true` emits -- instead of I420-sized carrier frames; the I420 form is measured beside it)
            chain = element_rate("chain_bench.sh", ("150000", "16"), env={"CHAINBENCH_RECORDS": "1"})   # (round 6: 150,000 frames per stream; 60,000 moved +- 20 % from run to run)
            chain_i420 = element_rate("chain_bench.sh", ("150000", "16"), env={"CHAINBENCH_RECORDS": "0"})
            if "frames_per_s_full_chain" in chain:
                chain["input"] = "application/x-cova-records (packed two-byte records per macroblock)"
                chain["frames_per_s_full_chain_i420_carrier_frames"] = chain_i420.get("frames_per_s_full_chain")
            if "frames_per_s_full_chain" in chain:
                t_cova = chain["seconds"] * min(16, effective_cores()) / max(1, chain["frames_in"])
                chain["limiter"] = ("host: every core of the box's share is busy (the per-stream cova elements -- SORT with the "
                                    "experiment's minhits 30 / maxage 60 keeps ~130 young trackers per stream alive -- on the pusher "
                                    "threads, the feeders' packing into the slots, GStreamer's per-buffer costs); 8 or 16 pusher threads "
                                    "give the same rate")
                chain["host_threads"] = ("16 carrier-frame feeders + 16 access-unit feeders + 8 pusher threads + collector + submitter on "
                                         "this box's CPU share")
                chain["cova_us_per_frame_upper_bound"] = round(t_cova * 1e6, 1)
            pre["full_filter_chain"] = chain
            # the same chain through the C-ABI alone (covahip_pipe + per-stream covahip_gopfilter, no GStreamer): the library's own
            # share of the host cost -- what an element written against include/covahip.h pays before its framework's per-buffer work
            native = element_rate("native_chain.sh", ("4000", "16", "8"))
            if "frames_per_s_native_chain" in native:
                native["host_cores_one_gpu_would_need"] = None   # filled in once `value` is known (below)
            pre["native_chain_c_abi_only"] = native
            # the pinned pipeline driven from C (no interpreter in the loop), three lanes x six slots (what `blobnetfilter` runs), on a
            # seeded random batch, beside the SAME batch resident in HBM through covahip_filter_forward_frames_packed: the PCIe-inclusive
            # share of the resident rate on one input (round 6)
            pre["pcie_inclusive_c_driver"] = element_rate("pipe_host_cost.sh", ("--bench",))
        if not args.no_cpu_baseline:
            pre["cpu_tracking"] = cpu_tracking_baseline()

    from cova_amd.multigpu import Group, pin_to_gpu, streams_of_rank
    # torch.distributed only when WORLD_SIZE > 1, and only as the job's control plane: rendezvous, barrier, MAX over ranks, the
    # gather of the per-rank lines -- gloo on CPU tensors.  RCCL is never initialised: the path has no exchange step.
    grp = Group()
    rank, local_rank, world = grp.rank, grp.local_rank, grp.world
    if world != args.gpus:
        sys.exit(f"bench.py: world size {world} != --gpus {args.gpus}")

    from cova_amd import synth, weights as W
    from cova_amd.elements import BlobNetInfer, Context

    # rank r = GPU r + the cores of that GPU's NUMA node: the per-stream host work (decoder threads, cova elements) of the
    # streams a rank owns runs next to the GPU it feeds.  Ranks whose GPUs share a node split its cores.
    # The GPU's PCI address is asked for in a short-lived CHILD (cova_amd.multigpu.gpu_numa_in_child) and every thread this
    # process has by now (gloo's) is bound as well: the pin is in place BEFORE the HIP runtime starts its threads, and
    # OMP_NUM_THREADS follows the narrowed mask.
    dev = local_rank
    pin = {"pinned": False}
    if world > 1 or args.rehearse_on_one_gpu:
        from cova_amd.multigpu import gpu_numa_in_child, gpu_numa_sysfs
        # sysfs first (KFD topology: no HIP call, no process on the card); the child only where the topology does not say
        info = gpu_numa_sysfs(local_rank, modulo_present=args.rehearse_on_one_gpu)
        # (the child opens the device for a moment: in a rehearsal of more than three ranks on one card that would double the processes
        # on it -- the pool admits six --, so there the node simply stays unknown and the granted cores are split among the ranks)
        if info is None and not args.control_plane_only and not (args.rehearse_on_one_gpu and world > 3):
            info = gpu_numa_in_child(local_rank, modulo_present=args.rehearse_on_one_gpu)
        node, node_cpus, dev = info if info is not None else (-1, [], local_rank)
        if world > 1:
            nodes = grp.gather(node)
            same = [r for r in range(world) if nodes[r] == nodes[rank]]
            pin = pin_to_gpu(dev, len(same), same.index(rank), node_info=(node, node_cpus), world=world, rank=rank)
            if pin["pinned"]:
                os.environ["OMP_NUM_THREADS"] = str(len(pin["cpus"]))
                if grp.torch is not None:
                    grp.torch.set_num_threads(len(pin["cpus"]))
    if args.control_plane_only:
        # what an 8-GPU node's launch exercises besides the kernels: every rank has joined, owns its streams, has its cores; the
        # barrier + MAX + gather that bracket the timed region run on made-up clocks
        grp.barrier()
        fake_s = 1e-3 * (1 + rank)
        slowest = grp.max(fake_s)
        mine = {"rank": rank, "local_rank": local_rank, "device": dev, "streams": streams_of_rank(8 * world, rank, world),
                "frames_per_step": args.batch, "numa_node": pin.get("numa_node"), "cpus": pin.get("cpus") or sorted(os.sched_getaffinity(0)),
                "pinned": pin.get("pinned"), "threads_bound": pin.get("threads_bound"), "input_seed": 0xC07A + 1000 * rank,
                "omp_num_threads": os.environ.get("OMP_NUM_THREADS")}
        rank_line(mine)
        all_ranks = grp.gather(mine)
        if rank == 0:
            print(json.dumps({"metric": "compressed-domain frames/sec (BlobNet+bboxcc) at 1080p b=256", "value": None, "unit": "frames/s",
                              "n_gpus": world, "scaling": "weak", "control_plane_only": True,
                              "rehearsal": f"control plane of {world} ranks only, no GPU touched: NOT a measurement",
                              "slowest_rank_region_s": slowest, "ranks": all_ranks,
                              "control_plane": "gloo on CPU tensors (rendezvous, barrier, MAX over ranks, gather of these objects); "
                                               "no RCCL communicator exists"}), flush=True)
        grp.barrier()
        grp.close()
        return
    ctx = Context(dev)
    ctx.set_lanes(1)         # the extra legs and the per-kernel pass run one step after the other
    B = args.batch
    NL = max(1, min(4, args.lanes))
    flat = W.random_init(1234)
    extras = {}
    if rank == 0 and not args.no_extra_legs:
        extras = extra_legs(ctx, flat, max(20, args.steps // 4))
    net = BlobNetInfer(ctx, flat, H_MB, W_MB, max_batch=B)
    # synthetic metapreprocess output: 8 independent streams interleaved, distinct per rank
    seed = 0xC07A + 1000 * rank
    stack = synth.stacked_batch(B, H_MB, W_MB, seed=seed, streams=8)
    d_stack = ctx.malloc(stack.nbytes)
    ctx.h2d(d_stack, stack)
    # one set of output buffers per lane: steps in flight together must not share them (include/covahip.h, lanes)
    d_boxes = [ctx.malloc(B * MAX_BOXES * 20) for _ in range(NL)]
    d_counts = [ctx.malloc(B * 4) for _ in range(NL)]
    d_mask = [ctx.malloc(B * H_MB * W_MB) for _ in range(NL)]

    # the same 256 stacks as carrier frames + table
    frames, index = synth.carrier_batch(B, H_MB, W_MB, seed=seed, streams=8)
    assert np.array_equal(np.concatenate([frames[index[:, k]] for k in range(T)], axis=1), stack)
    d_frames = ctx.malloc(frames.nbytes)
    ctx.h2d(d_frames, frames)
    # a SECOND batch (other streams, another stream interleaving -> a different stack table and different carrier frames): the
    # timed steps alternate between the two, so that every step validates and uploads its table (a streaming caller's does change;
    # an unchanged table is not uploaded again -- that cached figure is reported beside `value`)
    frames_b, index_b = synth.carrier_batch(B, H_MB, W_MB, seed=seed + 77, streams=4)
    stack_b = synth.stacked_batch(B, H_MB, W_MB, seed=seed + 77, streams=4)
    d_frames_b, d_stack_b = ctx.malloc(frames_b.nbytes), ctx.malloc(stack_b.nbytes)
    ctx.h2d(d_frames_b, frames_b)
    ctx.h2d(d_stack_b, stack_b)
    turn = [0]
    alternate = [True]

    def step_stack():
        turn[0] += 1
        k = turn[0] % NL
        second = alternate[0] and (turn[0] // NL) % 2 == 1
        net.filter_device(d_stack_b if second else d_stack, B, CC_THRESHOLD, d_boxes[k], d_counts[k], MAX_BOXES, d_mask[k])

    def step_frames():
        turn[0] += 1
        k = turn[0] % NL
        if alternate[0] and (turn[0] // NL) % 2 == 1:      # every lane sees the two batches in turn
            net.filter_frames_device(d_frames_b, frames_b.shape[0], index_b, B, CC_THRESHOLD, d_boxes[k], d_counts[k], MAX_BOXES, d_mask[k])
        else:
            net.filter_frames_device(d_frames, frames.shape[0], index, B, CC_THRESHOLD, d_boxes[k], d_counts[k], MAX_BOXES, d_mask[k])

    step = step_frames if args.entry == "frames" else step_stack
    other = step_stack if args.entry == "frames" else step_frames

    def barrier():
        ctx.sync()           # every lane of the ctx
        grp.barrier()        # torch.cuda.synchronize() + dist.barrier() when world > 1

    # ---- one lane: per-kernel times of a step on its own (profiled pass, not timed), the dominant kernel, the step latency
    for _ in range(3):
        step()
    ctx.profile(True)
    for _ in range(5):
        step()
    ctx.sync()
    prof = ctx.profile_read()
    ctx.profile(False)
    per_kernel_us = {k: t / n * 1e3 for k, (t, n) in prof.items()}
    blob_kernels = {k: v for k, v in per_kernel_us.items() if k not in ("bboxcc_kernel", "bboxcc_wave_kernel", "dec3_bboxcc_fused")}
    dominant = max(blob_kernels, key=blob_kernels.get)

    # ---- warm-up with the lanes of the timed region: --warmup steps and at least --min-warmup-s of them (clocks and caches settle)
    ctx.set_lanes(NL)
    t_w = time.perf_counter()
    n_w = 0
    while n_w < args.warmup or time.perf_counter() - t_w < args.min_warmup_s:
        step()
        n_w += 1
        if n_w % 64 == 0:
            ctx.sync()
    # ---- timed region: K steps between barriers; only the dominant kernel carries HIP events (2 per step).  A K-step region
    # of a few tens of milliseconds is repeated (at least five times while it is shorter than 50 ms) and the MEDIAN region
    # counts: `steps` stays K, every repeat is a complete barrier-to-barrier measurement of K steps.
    ctx.profile(True, only=dominant)
    regions, ev_regions = [], []
    while True:
        barrier()
        t0 = time.perf_counter()
        ctx.timer_start(0)
        for _ in range(args.steps):
            step()
        ctx.timer_stop(0)
        barrier()
        regions.append(grp.max(time.perf_counter() - t0))     # the slowest rank's region (every rank appends the same number)
        ev_regions.append(ctx.timer_ms(0))
        if len(regions) >= 5 or (len(regions) >= 1 and regions[0] >= 0.05):
            break
    order = sorted(range(len(regions)), key=lambda i: regions[i])
    mid = order[len(order) // 2]
    elapsed = regions[mid]
    ev_ms = ev_regions[mid]
    dom_ms, dom_n = ctx.profile_read()[dominant]
    ctx.profile(False)
    # the same K steps with ONE batch (unchanged stack table: no validation upload after the first step), for comparison
    alternate[0] = False
    for _ in range(2 * NL):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed_cached = grp.max(time.perf_counter() - t0)
    # ---- one lane: a step on its own, measured like the timed region (warm clocks, alternating batches, median of five
    # barrier-to-barrier regions of K steps) -- rounds 1-3 timed 50 steps right after start-up, before the clocks had settled
    ctx.set_lanes(1)
    alternate[0] = True
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < args.min_warmup_s:
        for _ in range(32):
            step()
        ctx.sync()
    regions1 = []
    for _ in range(5):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        regions1.append(grp.max(time.perf_counter() - t0))
    serial_ms = sorted(regions1)[len(regions1) // 2] / args.steps * 1e3
    # per-kernel times of a step on its own, warm (the pass before the timed region only picked the dominant kernel)
    ctx.profile(True)
    for _ in range(20):
        step()
    ctx.sync()
    per_kernel_us = {k: t / n * 1e3 for k, (t, n) in ctx.profile_read().items()}
    ctx.profile(False)
    alternate[0] = False
    ctx.set_lanes(NL)
    for _ in range(2 * NL):
        step()
    barrier()
    # (the legs and cross-checks below stay on the first batch)
    # shader clock under this load: a probe wave beside 60 more steps (untimed)
    for _ in range(60):
        step()
    clock_mhz = ctx.clock_mhz(300)
    ctx.sync()

    # ---- bboxcc roofline (outside the timed region).  Inside the hot path bboxcc runs in the same launch as the last
    # decoder block; standalone it is timed (a) on the masks the last step left in HBM (b = 256: one frame per CU,
    # latency bound) and (b) at B = 65,536 sparse-blob masks, where the byte rate of the wave-per-frame kernel shows.
    from cova_amd.elements import BboxCc
    ctx.set_lanes(1)
    cc = BboxCc(ctx, CC_THRESHOLD, MAX_BOXES)
    d_boxes2 = ctx.malloc(B * MAX_BOXES * 20)
    d_counts2 = ctx.malloc(B * 4)
    cc_ms = timed_steps(ctx, lambda: cc.regionprops_device(d_mask[0], B, H_MB, W_MB, d_boxes2, d_counts2), args.steps, 3)
    counts2 = np.zeros(B, dtype=np.int32)
    ctx.d2h(counts2, d_counts2)
    counts = np.zeros(B, dtype=np.int32)
    ctx.d2h(counts, d_counts[0])
    assert (counts == counts2).all(), "fused decoder tail and standalone bboxcc disagree"
    for k in range(1, NL):
        ctx.d2h(counts2, d_counts[k])
        assert (counts == counts2).all(), "the lanes disagree"
    ctx.free(d_boxes2); ctx.free(d_counts2)

    rank0 = {}
    if rank == 0 and not args.no_extra_legs:
        # bboxcc at large batch (SURVEY.md section 8d asks for the sweep; tools/bboxcc_sweep.py has all of it)
        from tools.bboxcc_sweep import make_masks
        BB = 65536
        masks = np.tile(make_masks("blobs", 256), (BB // 256, 1, 1))
        d_m, d_b, d_c = ctx.malloc(masks.nbytes), ctx.malloc(BB * 64 * 20), ctx.malloc(BB * 4)
        ctx.h2d(d_m, masks)
        cc64 = BboxCc(ctx, CC_THRESHOLD, 64)
        ms = timed_steps(ctx, lambda: cc64.regionprops_device(d_m, BB, H_MB, W_MB, d_b, d_c), 5, 2)
        rank0["bboxcc_B65536_sparse_blobs"] = {"ns_per_frame": round(ms * 1e6 / BB, 2), "GBps": round(BB * H_MB * W_MB / ms / 1e6, 1),
                                               "frac_of_8TBs": round(BB * H_MB * W_MB / ms / 1e6 / HBM_PEAK_GBS, 4)}
        for p in (d_m, d_b, d_c):
            ctx.free(p)
        # the other entry point on the SAME 256 stacks: one lane (the latency of a step on its own) and NL lanes
        ms1 = timed_steps(ctx, other, max(20, args.steps // 2))
        ctx.set_lanes(NL)
        msn = timed_steps(ctx, other, args.steps, 2 * NL)
        ctx.set_lanes(1)
        ctx.d2h(counts2, d_counts[0])
        assert (counts == counts2).all(), "carrier-frame entry and stacked entry disagree"
        rank0["stacked_entry" if args.entry == "frames" else "carrier_frame_entry"] = {
            "frames_per_s": round(B / msn * 1e3, 1), "ms_per_step": round(msn, 4), "ms_per_step_one_lane": round(ms1, 4),
            "carrier_frames_per_step": int(frames.shape[0]), "carrier_input_bytes_per_step": int(frames.nbytes),
            "stacked_input_bytes_per_step": int(stack.nbytes)}
        # board power while the timed workload loops ("The roof over the step is the board's power"): with NL lanes
        # the board sits at or near its cap, and power x step time is the same with one lane
        ctx.set_lanes(NL)
        pw_n = board_power_under(ctx, step)
        ctx.set_lanes(1)
        pw_1 = board_power_under(ctx, step)
        if pw_n and pw_1:
            rank0["board_power"] = {f"{NL}_lanes": pw_n, "one_lane": pw_1,
                                    "note": "hwmon power sensor of this GPU sampled every 10 ms over a 2 s loop of the timed step "
                                            "(untimed leg; the first 0.8 s dropped)"}
        # PCIe-inclusive rates (never `value`)
        net.filter(stack, CC_THRESHOLD, max_boxes=MAX_BOXES)        # warm the staging buffers
        t1 = time.perf_counter()
        for _ in range(5):
            net.filter(stack, CC_THRESHOLD, max_boxes=MAX_BOXES)
        rank0["frames_per_s_pcie_inclusive_host_buffers"] = round(5 * B / (time.perf_counter() - t1), 1)
        ctx.set_lanes(NL)
        rank0["frames_per_s_pcie_inclusive_pipelined_carrier_frames"] = pipelined_host_rate(net, frames, index, args.steps)
        ctx.set_lanes(1)

    # ---- one object per rank (stderr, and gathered into the aggregate line): what it owns and what it measured on its own clock
    n_streams = 8 * world
    mine = {"rank": rank, "local_rank": local_rank, "device": dev, "streams": streams_of_rank(n_streams, rank, world),
            "frames_per_step": B, "ms_per_step_own_clock": round(ev_ms / args.steps, 4),
            "frames_per_s_own_clock": round(B * args.steps / (ev_ms * 1e-3), 1),
            "ms_per_step_one_lane": round(serial_ms, 4), "numa_node": pin.get("numa_node"), "cpus": pin.get("cpus"),
            "pinned": pin.get("pinned"), "input_seed": seed}
    if world > 1:
        rank_line(mine)
    all_ranks = grp.gather(mine)

    if rank == 0:
        macs = kernel_macs_per_frame()
        kbytes = kernel_bytes_per_frame()
        hs, ws = level_dims()
        # carrier path: level 0 up to the pool once per carrier frame; level 1 + level 0's temporal MLP + the gather per stack
        nfr = frames.shape[0] if args.entry == "frames" else T * B     # (the stacked tensor is T * B carrier frames)
        macs["enc0p_mfma"] = hs[0] * ws[0] * 9 * 3 * 16 * nfr / B
        kbytes["enc0p_mfma"] = (hs[0] * ws[0] * 4 + hs[1] * ws[1] * 16 * 2) * nfr / B
        macs["enc1_mfma"] += hs[1] * ws[1] * 16 * 32            # + level 0's temporal MLP (applied while staging)
        # + the level-0 skip connection: as fp32 partial logits, four per grid position of the last block (round 5; the level-1 kernel
        # runs the skip half of the folded last block, (H1 + 1)(W1 + 1) x 4 taps x 16 channels x 4 parities MAC, counted as executed
        # work of this kernel, not as algorithmic FLOP -- those stay with the last block)
        part_bytes = (hs[1] + 1) * (ws[1] + 1) * 16
        kbytes["enc1_mfma"] += part_bytes
        kbytes["dec3_bboxcc_fused"] = kbytes["dec3_final_mfma"] = hs[1] * ws[1] * 16 * 2 + part_bytes + H_MB * W_MB
        dom_s = dom_ms / dom_n * 1e-3
        cc_s = cc_ms * 1e-3
        cc_gbs = B * H_MB * W_MB / cc_s / 1e9
        total_flop = 2.0 * net.macs_per_frame * B
        # what the timed entry executes: the carrier-frame entry runs level 0's convolution once per carrier frame
        # (nfr slices) instead of once per (stack, T slice) (4 B slices)
        executed_flop = total_flop - (2.0 * hs[0] * ws[0] * 9 * 3 * 16 * (T * B - nfr) if args.entry == "frames" else 0.0)
        # ... and the last decoder block + final conv as ONE folded 32 -> 1 transposed conv (16.8 M -> 1.1 M MAC per frame)
        fold_alg, fold_exec = dec3_fold_macs_per_frame()
        executed_flop -= 2.0 * (fold_alg - fold_exec) * B
        # every kernel of a step against both roofs, from the warm one-lane per-kernel pass (HIP events: ~2 us of event
        # overhead per launch are inside these times -- the rocprofv3 figures in profiles/ are the ones without)
        kernels = {}
        for k, us in sorted(per_kernel_us.items()):
            if k not in macs:
                continue
            fl, by = 2.0 * macs[k] * B, kbytes[k] * B
            # frac_hbm is priced on the bytes the fabric counters saw for this launch (the committed FETCH_SIZE x 2 + WRITE_SIZE
            # summary, b = 256): the structural count below it includes re-reads that L2 absorbs (level 1 gathers every P slice
            # four times: 109 MB structural, 71 MB on the fabric) and is no HBM figure
            cnt, cnt_src = committed_traffic(k, args.entry) if B == BATCH else (None, None)
            kernels[k] = {"us_alone": round(us, 2), "alg_flop": fl, "frac_mfma": round(fl / (us * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                          "hbm_bytes_counter": cnt, "hbm_bytes_counter_source": cnt_src,
                          "frac_hbm": round(cnt / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if cnt else None,
                          "bytes_structural": by, "frac_hbm_structural": round(by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
        if "dec3_bboxcc_fused" in kernels:
            kernels["dec3_bboxcc_fused"]["executed_flop"] = 2.0 * fold_exec * B
            kernels["dec3_bboxcc_fused"]["note"] = "alg_flop = convT 32->16 + 1x1 as written; executed folded; bboxcc runs in the same launch"
        step_s = elapsed / args.steps
        other_entry = rank0.get("stacked_entry" if args.entry == "frames" else "carrier_frame_entry", {})
        dom_flop = 2.0 * macs[dominant] * B
        ach_tflops = dom_flop / dom_s / 1e12
        dom_bytes = kbytes[dominant] * B
        dom_traffic, dom_traffic_src = committed_traffic(dominant, args.entry) if B == BATCH else (None, None)
        cc_traffic, _ = committed_traffic("bboxcc_kernel", "stack") if B == BATCH else (None, None)
        serial_dom_s = per_kernel_us[dominant] * 1e-6
        line = {
            "metric": "compressed-domain frames/sec (BlobNet+bboxcc) at 1080p b=256",
            "value": round(world * B * args.steps / elapsed, 1),
            "unit": "frames/s",
            # the literal BASELINE config-3 input (pre-stacked tensors) and a step on its own, right beside `value`
            "value_stacked_entry": other_entry.get("frames_per_s") if args.entry == "frames" else round(world * B * args.steps / elapsed, 1),
            "ms_per_step_stacked_entry": other_entry.get("ms_per_step") if args.entry == "frames" else round(elapsed / args.steps * 1e3, 4),
            "ms_per_step_one_lane": round(serial_ms, 4),
            "value_one_lane": round(world * B / serial_ms * 1e3, 1),
            "schema": 6,
            "schema_note": ("6 (round 6): roofline.alone / roofline.in_situ are both explicit; roofline.achieved / frac / avg_launch_us repeat "
                            "roofline.alone (the dominant launch alone on the chip, as in round 5; rounds 1-4 printed the in-situ figure "
                            "under those names); kernels{}.frac_hbm is priced on counter bytes (hbm_bytes_counter), the structural count "
                            "moved to bytes_structural"),
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(step_s * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "lanes": NL,
            **({"rehearsal": f"{world} ranks shared the GPUs present on this box: NOT a scaling result"} if args.rehearse_on_one_gpu else {}),
            **({"ranks": all_ranks, "control_plane": "gloo on CPU tensors (rendezvous, barrier, MAX over ranks, gather of these objects); "
                                                      "no RCCL communicator exists"} if world > 1 else {}),
            "timed_regions_s": [round(r, 5) for r in regions],
            "timed_region": f"median of {len(regions)} barrier-to-barrier regions of {args.steps} steps each",
            "ms_per_step_one_batch": round(elapsed_cached / args.steps * 1e3, 4),
            "shader_clock_mhz_under_load": round(clock_mhz, 1),
            "blobnet_mfma_util_whole_net": round(total_flop / step_s / 1e12 / MFMA_PEAK_TFLOPS, 4),
            "blobnet_mfma_util_whole_net_executed": round(executed_flop / step_s / 1e12 / MFMA_PEAK_TFLOPS, 4),
            "blobnet_mfma_util_whole_net_one_lane": round(total_flop / (serial_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
            "blobnet_mfma_util_whole_net_one_lane_executed": round(executed_flop / (serial_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
            "algorithmic_flop_per_step": total_flop,
            "executed_flop_per_step": executed_flop,
            "per_kernel_us": {k: round(v, 2) for k, v in sorted(per_kernel_us.items())},
            "per_kernel_us_note": ("HIP events around every launch of 20 one-lane steps, warm: the event pair costs ~2 us per launch, so "
                                   "the sum exceeds ms_per_step_one_lane; kernel times without it: profiles/r*/kernel_stats_frames_lanes1.csv (rocprofv3)"),
            "launches_per_step": len(per_kernel_us),
            "kernels": kernels,
        }
        line.update({k: rank0.pop(k) for k in ("stacked_entry", "carrier_frame_entry") if k in rank0})
        line.update({
            # SURVEY.md section 8(d) prices the BlobNet kernels against the MFMA roof; the same launch against
            # the HBM roof (compulsory bytes of the kernel / time) is given beside it, with the tighter one named.
            # The dominant kernel's roofline is a property of the kernel: the headline figures are the launch ALONE on the chip
            # (HIP events on its own stream in the warm one-lane timed pass; profiles/r*/kernel_stats_frames_lanes1.csv is the
            # rocprofv3 view of the same).  Inside the multi-lane timed region the launch shares the CUs with other steps' launches
            # and its bracket also holds queueing: that figure is given beside it, not instead of it.
            "roofline": {"kernel": dominant, "bound": "mfma", "achieved": round(dom_flop / serial_dom_s / 1e12, 2),
                         "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(dom_flop / serial_dom_s / 1e12 / MFMA_PEAK_TFLOPS, 4),
                         "traffic": dom_traffic, "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE)",
                         "traffic_source": dom_traffic_src,
                         "algorithmic_flop_per_launch": dom_flop, "avg_launch_us": round(serial_dom_s * 1e6, 2),
                         "measured": "HIP events on the launch's own stream, the launch alone on the chip (20 one-lane steps after the timed regions, warm)",
                         "alone": {"avg_launch_us": round(serial_dom_s * 1e6, 2), "achieved": round(dom_flop / serial_dom_s / 1e12, 2),
                                   "frac": round(dom_flop / serial_dom_s / 1e12 / MFMA_PEAK_TFLOPS, 4)},
                         "in_situ": {"lanes": NL, "avg_launch_us": round(dom_s * 1e6, 2), "achieved": round(ach_tflops, 2),
                                     "frac": round(ach_tflops / MFMA_PEAK_TFLOPS, 4)},
                         f"in_situ_{NL}_lanes": {"avg_launch_us": round(dom_s * 1e6, 2), "launches_timed": dom_n,
                                                 "achieved": round(ach_tflops, 2), "frac": round(ach_tflops / MFMA_PEAK_TFLOPS, 4),
                                                 "measured": (f"HIP events inside the timed region; with {NL} lanes the launch shares the chip with "
                                                              "the other lanes' launches, so its duration is longer than on its own")},
                         "hbm_view": {"algorithmic_bytes_per_launch": dom_bytes,
                                      "achieved_GBs": round(dom_bytes / serial_dom_s / 1e9, 1),
                                      "frac_of_8TBs": round(dom_bytes / serial_dom_s / 1e9 / HBM_PEAK_GBS, 4)},
                         "tighter_roof": "hbm" if dom_bytes / (HBM_PEAK_GBS * 1e9) > dom_flop / (MFMA_PEAK_TFLOPS * 1e12) else "mfma"},
            "roofline_bboxcc": {"kernel": "bboxcc_kernel", "bound": "hbm", "achieved": round(cc_gbs, 2),
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(cc_gbs / HBM_PEAK_GBS, 5),
                                "traffic": cc_traffic, "algorithmic_bytes_per_launch": B * H_MB * W_MB,
                                "avg_launch_us": round(cc_s * 1e6, 2),
                                "note": "standalone kernel on the step's masks (in the hot path bboxcc runs inside the last "
                                        "decoder block's launch); b=256 masks are 2.09 MB: latency bound -- the byte rate is "
                                        "bboxcc_B65536_sparse_blobs"},
            "blobnet_mfma_util_note": "BlobNet FLOP (algorithmic: SURVEY.md 8d; executed: level 0 once per carrier frame, last decoder block + final conv folded) over the "
                                      "WHOLE step time (bboxcc included: it shares a launch)",
            "hip_event_ms_per_step_rank0": round(ev_ms / args.steps, 4),
            "boxes_per_frame_mean": float(counts.mean()),
            "device": ctx.info(),
        })
        line.update(rank0)
        line.update(extras)
        for k in ("through_gstreamer_elements", "full_filter_chain", "native_chain_c_abi_only", "pcie_inclusive_c_driver"):
            if k in pre:
                line[k] = pre[k]
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(flat, stack)
            line["cpu_baseline"].update(pre.get("cpu_tracking", {}))
            cb = line["cpu_baseline"]
            if "bboxcc_sort_nstreams_frames_per_s" in cb:
                # the whole chain on the host's cores: BlobNet and the per-stream bboxcc + SORT share them
                cb["full_chain_frames_per_s"] = round(1.0 / (1.0 / cb["value"] + 1.0 / cb["bboxcc_sort_nstreams_frames_per_s"]), 2)
            if "frames_per_s_full_chain" in pre.get("full_filter_chain", {}):
                fc = pre["full_filter_chain"]
                fc["vs_cpu_full_chain"] = round(fc["frames_per_s_full_chain"] / cb.get("full_chain_frames_per_s", cb["value"]), 1)
            line["gpu_over_cpu"] = round(line["value"] / world / line["cpu_baseline"]["value"], 1)
        nc = pre.get("native_chain_c_abi_only", {})
        if "cpu_us_per_frame" in nc and world == 1:   # process CPU time per frame (eight worker threads, OpenMP waits and the HIP runtime's threads included)
            nc["host_cores_one_gpu_would_need"] = round(line["value"] * nc["cpu_us_per_frame"] * 1e-6, 1)
        fc = pre.get("full_filter_chain", {})
        if "frames_per_s_full_chain" in fc and world == 1:
            fc["host_cores_one_gpu_would_need"] = int(line["value"] / max(1.0, fc["frames_per_s_full_chain"]) * min(16, effective_cores()))
        # the long strings last, so that the numbers survive a truncated log
        line["config"] = {"workload": ("temporal stacking as a GPU gather + BlobNet + bboxcc fused (covahip_filter_forward_frames): carrier "
                                       "frames of 8 streams + stack table, 256 output frames, 1080p macroblock grid 68x120, T=4, inputs "
                                       "resident in HBM" if args.entry == "frames" else
                                       "BlobNet + bboxcc fused (covahip_filter_forward) on 256 pre-stacked tensors, 1080p macroblock grid "
                                       "68x120, T=4, inputs resident in HBM"),
                          "entry": args.entry, "lanes": NL,
                          "lanes_note": (f"{NL} batches of {B} in flight per GPU on {NL} HIP streams with a workspace each, as the reference keeps "
                                         "16 BlobNet engines busy on one GPU (experiment/cova/config.yaml:33-34); ms_per_step = wall time / steps; "
                                         "ms_per_step_one_lane = one step after the other"),
                          "stack_table": ("two batches (different streams and interleaving: different carrier frames, different stack tables) alternate "
                                          "across the timed steps of every lane; every step validates its table on the host and hands it to the "
                                          "level-1 kernel by value, in its kernel arguments (blobnet.hip prepare_frames; no copy in front of the "
                                          "kernels); ms_per_step_one_batch = the same steps on ONE batch"),
                          "batch_per_gpu": B, "grid_mb": [H_MB, W_MB], "timestep": T, "cc_threshold": CC_THRESHOLD,
                          "parallelism": f"{world} x independent per-GPU batches, no collective"}
        print(json.dumps(line), flush=True)

    barrier()
    ctx.close()
    grp.close()


if __name__ == "__main__":
    main()
