"""Developer helper, ON THE GPU BOX with a -DPHASE_TIMING build copied over cova_amd/libcovahip.so (tools/build_phase_lib.sh):
life span of every workgroup of the last launch of the encoder kernels -- start / end relative to the first start, grouped by
the CU they ran on (hardware id) -- to see ramps, tails and the older / younger workgroup of a CU."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cova_amd import synth, weights as W  # noqa: E402
from cova_amd.elements import BlobNetInfer, Context  # noqa: E402
from cova_amd import _lib as L  # noqa: E402

B, H, Wd = 256, 68, 120
ctx = Context(0)
net = BlobNetInfer(ctx, W.random_init(1234), H, Wd, max_batch=B)
if os.environ.get("QB_IMPL"):
    net.set_impl(os.environ["QB_IMPL"])
for spec in filter(None, os.environ.get("QB_PLAN", "").split(",")):      # level:nbands:nbuf (covahip_blobnet_set_enc_plan)
    net.set_enc_plan(*[int(x) for x in spec.split(":")])
frames, index = synth.carrier_batch(B, H, Wd, seed=1, streams=8)
d_frames = ctx.malloc(frames.nbytes)
ctx.h2d(d_frames, frames)
d_boxes, d_counts, d_mask = ctx.malloc(B * 256 * 20), ctx.malloc(B * 4), ctx.malloc(B * H * Wd)
lib = ctypes.CDLL(L.LIB_PATH)
for _ in range(20):
    net.filter_frames_device(d_frames, frames.shape[0], index, B, 1, d_boxes, d_counts, 256, d_mask)
ctx.sync()
for kid, name, nwg in ((5, "level 0 (enc0p)", 768), (0, "level 1", 512), (1, "level 2", 512), (2, "level 3", 256), (3, "dec012", 256), (4, "dec3cc", 256)):
    out = (ctypes.c_ulonglong * (1024 * 4))()
    assert lib.covahip_dev_wgspan_read(out, kid) == 0
    v = np.array(list(out), dtype=np.int64).reshape(1024, 4)[:nwg]
    v = v[v[:, 0] > 0]
    nwg = len(v)
    if nwg == 0:
        continue
    t0 = v[:, 0].min()
    st, en = (v[:, 0] - t0) / 100.0, (v[:, 1] - t0) / 100.0      # us
    print(f"{name}: {nwg} workgroups; starts {st.min():.2f} .. {st.max():.2f} us (median {np.median(st):.2f}); ends {en.min():.2f} .. {en.max():.2f} us "
          f"(median {np.median(en):.2f}, 10 % {np.percentile(en, 10):.2f}, 90 % {np.percentile(en, 90):.2f}); life {np.median(en - st):.2f} us median")
    clk = v[:, 3] / np.maximum(v[:, 1] - v[:, 0], 1) * 100.0     # MHz: shader cycles per 10 ns tick
    print(f"   shader clock over the workgroups' lives: median {np.median(clk):.0f} MHz, 10 % {np.percentile(clk, 10):.0f}, 90 % {np.percentile(clk, 90):.0f}")
    # by CU: hardware id bits (wave, simd, pipe, cu, sh, se, ...): group by the id without the wave / simd fields
    cu = v[:, 2] >> 8
    first, second = [], []
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]
        if len(idx) == 2:
            a, b = sorted(idx, key=lambda i: st[i])
            first.append(en[a] - st[a]); second.append(en[b] - st[b])
    if first:
        print(f"   CUs with two workgroups: {len(first)}; first-started life {np.median(first):.2f} us, second-started {np.median(second):.2f} us; "
              f"role by block index: blocks < {nwg // 2} end at {np.median(en[:nwg // 2]):.2f}, blocks >= {nwg // 2} at {np.median(en[nwg // 2:]):.2f}")
    if os.environ.get("WGSPAN_DETAIL") == str(kid):
        life = en - st
        print("   life histogram (us):", np.histogram(life, bins=12)[1].round(1).tolist(), np.histogram(life, bins=12)[0].tolist())
        print("   end histogram (us): ", np.histogram(en, bins=12)[1].round(1).tolist(), np.histogram(en, bins=12)[0].tolist())
        for c in np.unique(cu)[:12]:
            idx = sorted(np.where(cu == c)[0], key=lambda i: st[i])
            print("   cu", hex(int(c)), " ".join(f"[wg {i}: {st[i]:.2f}-{en[i]:.2f} clk {clk[i]:.0f}]" for i in idx))
        if kid in (1, 2) and hasattr(lib, "covahip_dev_itemspan_read"):
            io = (ctypes.c_ulonglong * (1024 * 16))()
            assert lib.covahip_dev_itemspan_read(io, kid) == 0
            t = (np.array(list(io), dtype=np.int64).reshape(1024, 4, 4)[:nwg] - t0) / 100.0
            for name2, sel in (("workgroups < half", slice(0, nwg // 2)), ("workgroups >= half", slice(nwg // 2, nwg))):
                for i in range(4):
                    if (t[sel, i, 0] > -1e6).all() and (t[sel, i, 0] > 0).any():
                        print(f"   {name2}, item {i}: start {np.median(t[sel, i, 0]):.2f}, requested {np.median(t[sel, i, 1]):.2f}, landed+barrier {np.median(t[sel, i, 2]):.2f}, tiles done {np.median(t[sel, i, 3]):.2f}")
