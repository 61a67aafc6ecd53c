"""The C-ABI library loads on a CPU-only host and exports every symbol include/covahip.h declares."""
import ctypes as C
import os
import re

from cova_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="covahip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(covahip_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported_and_bound():
    syms = _declared_symbols()
    assert len(syms) >= 45
    lib = C.CDLL(L.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in covahip.h but not exported"
    # and the Python binding covers exactly the declared set
    assert sorted(L.PROTOTYPES.keys()) == syms
    # developer switches live in their own header, outside the drop-in boundary
    dev = _declared_symbols("covahip_dev.h")
    assert sorted(L.DEV_PROTOTYPES.keys()) == dev and not set(dev) & set(syms)
    for s in dev:
        assert hasattr(lib, s)


def test_strerror_and_version():
    lib = L.lib()
    assert lib.covahip_strerror(0) == b"ok"
    assert b"unknown" in lib.covahip_strerror(1234)
    assert b"gfx950" in lib.covahip_version()


def test_no_gpu_calls_fail_loudly_not_silently():
    """On a host without a HIP device ctx creation must fail (no CPU fallback)."""
    lib = L.lib()
    n = C.c_int(-1)
    rc = lib.covahip_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        return  # GPU box: nothing to check here
    h = C.c_void_p()
    assert lib.covahip_ctx_create(0, C.byref(h)) != 0
    assert not h.value


def test_struct_layouts_match_header():
    assert C.sizeof(L.Box) == 20 and L.BOX_DTYPE.itemsize == 20
    assert L.BBOX_DTYPE.itemsize == 56
    assert L.AU_OUT_DTYPE.itemsize == 24
    assert L.KERNEL_TIME_DTYPE.itemsize == 64


def test_product_never_imports_oracle():
    """cova_amd/ (product) must not reference oracle/ in any form."""
    pkg = os.path.join(ROOT, "cova_amd")
    for dirpath, _, files in os.walk(pkg):
        if "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", "Makefile")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in src.replace("# oracle-free", ""), f"{f} mentions the oracle"


def test_null_and_invalid_arguments_do_not_need_a_gpu():
    lib = L.lib()
    assert lib.covahip_blobnet_load(None, b"x" * 64, 64, 68, 120, 4, 1) == 1
    assert lib.covahip_ctx_sync(None) == 1
    assert lib.covahip_sort_new(3, 3, 0.2, None) == 1
    assert lib.covahip_stack_new(0, 4, 1, C.byref(C.c_void_p())) == 1
    assert lib.covahip_stack_new(16, 0, 1, C.byref(C.c_void_p())) == 1
    n = C.c_size_t()
    assert lib.covahip_bbox_deserialize_vec(None, 0, None, 0, C.byref(n)) == 1
