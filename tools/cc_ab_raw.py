"""Developer helper, ON THE GPU BOX: bboxcc at B = 65,536 through the raw C-ABI of ANY library build (only entry points that
exist since round 1), for A/B runs across rounds on one box.  usage: cc_ab_raw.py <lib.so> [<lib.so> ...]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bboxcc_sweep import make_masks, H, W  # noqa: E402

B = 65536
kinds = os.environ.get("SWEEP_KINDS", "blobs,obj6,obj20").split(",")
masks = {k: np.ascontiguousarray(np.tile(make_masks(k, 256), (B // 256, 1, 1))) for k in kinds}
for rep in range(int(os.environ.get("CC_REPS", "2"))):
    for path in sys.argv[1:]:
        lib = C.CDLL(os.path.abspath(path))
        ctx = C.c_void_p()
        assert lib.covahip_ctx_create(0, C.byref(ctx)) == 0
        line = [os.path.basename(path)]
        for k in kinds:
            d_m, d_b, d_c = C.c_void_p(), C.c_void_p(), C.c_void_p()
            lib.covahip_malloc(ctx, C.c_size_t(masks[k].nbytes), C.byref(d_m))
            lib.covahip_malloc(ctx, C.c_size_t(B * 64 * 20), C.byref(d_b))
            lib.covahip_malloc(ctx, C.c_size_t(B * 4), C.byref(d_c))
            lib.covahip_memcpy_h2d(ctx, d_m, masks[k].ctypes.data_as(C.c_void_p), C.c_size_t(masks[k].nbytes))
            call = lambda: lib.covahip_bboxcc(ctx, d_m, B, H, W, 1, d_b, d_c, 64, 1)   # noqa: E731  (1 = COVAHIP_MEM_DEVICE)
            for _ in range(3):
                assert call() == 0
            lib.covahip_ctx_sync(ctx)
            lib.covahip_timer_start(ctx, 1)
            for _ in range(8):
                call()
            lib.covahip_timer_stop(ctx, 1)
            ms = C.c_float()
            lib.covahip_timer_elapsed_ms(ctx, 1, C.byref(ms))
            ns = ms.value / 8 * 1e6 / B
            line.append(f"{k}: {ns:.2f} ns/frame = {H * W / ns / 8000:.3f} of 8 TB/s")
            for p in (d_m, d_b, d_c):
                lib.covahip_free(ctx, p)
        lib.covahip_ctx_destroy(ctx)
        print("   ".join(line), flush=True)
