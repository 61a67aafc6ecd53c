"""BlobNet weight container for the MI355X filter stage.

The reference ships no weights (model/{tf,onnx,trt}_model hold only placeholders)
and its Keras model cannot be imported here, so this build owns its weight file
format.  The logical tensors keep the Keras layouts of the reference model
(utils/model/encoder.py:35-46, utils/model/pointwise.py:8-12,
utils/model/decoder.py:9-24,106-119; hyper-parameters
utils/train-blobnet.py:57-69):

  enc{i}.conv.kernel [3,3,Cin,Cout]   (Conv3D kernel, depth-1 axis dropped)
  enc{i}.conv.bias   [Cout]
  enc{i}.bn.{gamma,beta,mean,var} [Cout]
  enc{i}.tmix.w1/w2  [4,4]            (Conv1D kernel [Tin,Tout], no bias)
  dec{j}.up.kernel   [4,4,Cout,Cin]   (Conv3DTranspose kernel)
  dec{j}.up.bias     [Cout]
  dec{j}.bn.*        [Cout]           (j = 0..2)
  final.kernel [16], final.bias [1]

File = 64-byte header (16 x u32) + flat little-endian fp32 payload in the order
above (320,305 floats).
"""
from __future__ import annotations

import struct
from collections import OrderedDict

import numpy as np

MAGIC = 0x57485643  # "CVHW"
VERSION = 1
T = 4
ENC_C = (3, 16, 32, 64, 128)
DEC_CO = (64, 32, 16, 16)
DEC_CI = (128, 128, 64, 32)
N_PARAMS = 320_305


def tensor_specs() -> "OrderedDict[str, tuple]":
    specs: "OrderedDict[str, tuple]" = OrderedDict()
    for i in range(4):
        ci, co = ENC_C[i], ENC_C[i + 1]
        specs[f"enc{i}.conv.kernel"] = (3, 3, ci, co)
        specs[f"enc{i}.conv.bias"] = (co,)
        for n in ("gamma", "beta", "mean", "var"):
            specs[f"enc{i}.bn.{n}"] = (co,)
        specs[f"enc{i}.tmix.w1"] = (T, T)
        specs[f"enc{i}.tmix.w2"] = (T, T)
    for j in range(4):
        ci, co = DEC_CI[j], DEC_CO[j]
        specs[f"dec{j}.up.kernel"] = (4, 4, co, ci)
        specs[f"dec{j}.up.bias"] = (co,)
        if j < 3:
            for n in ("gamma", "beta", "mean", "var"):
                specs[f"dec{j}.bn.{n}"] = (co,)
    specs["final.kernel"] = (16,)
    specs["final.bias"] = (1,)
    return specs


def flatten(tensors: dict) -> np.ndarray:
    parts = []
    for name, shape in tensor_specs().items():
        a = np.asarray(tensors[name], dtype=np.float32)
        if a.shape != shape:
            raise ValueError(f"{name}: expected {shape}, got {a.shape}")
        parts.append(a.reshape(-1))
    flat = np.concatenate(parts)
    assert flat.size == N_PARAMS
    return flat


def unflatten(flat: np.ndarray) -> "OrderedDict[str, np.ndarray]":
    flat = np.asarray(flat, dtype=np.float32).reshape(-1)
    if flat.size != N_PARAMS:
        raise ValueError(f"expected {N_PARAMS} floats, got {flat.size}")
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    off = 0
    for name, shape in tensor_specs().items():
        n = int(np.prod(shape))
        out[name] = flat[off:off + n].reshape(shape)
        off += n
    return out


def to_bytes(flat: np.ndarray) -> bytes:
    flat = np.ascontiguousarray(flat, dtype="<f4").reshape(-1)
    hdr = struct.pack("<16I", MAGIC, VERSION, T, ENC_C[0], *ENC_C[1:], *DEC_CO, flat.size, 0, 0, 0)
    return hdr + flat.tobytes()


def from_bytes(blob: bytes) -> np.ndarray:
    hdr = struct.unpack("<16I", blob[:64])
    if hdr[0] != MAGIC or hdr[1] != VERSION:
        raise ValueError("not a covahip BlobNet weight blob")
    if hdr[2] != T or tuple(hdr[3:8]) != ENC_C or tuple(hdr[8:12]) != DEC_CO:
        raise ValueError("unsupported BlobNet hyper-parameters")
    n = hdr[12]
    flat = np.frombuffer(blob, dtype="<f4", count=n, offset=64).copy()
    if flat.size != N_PARAMS:
        raise ValueError("truncated weight blob")
    return flat


def random_init(seed: int = 1234, fg_bias: float = -3.9) -> np.ndarray:
    """Seeded random weights of the reference architecture.

    He-normal kernels as in the reference (kernel_initializer="he_normal",
    encoder.py:42, decoder.py:18); BN statistics drawn so that the affine is
    non-trivial (gamma may be small but stays positive here; negative gamma is
    exercised separately in the tests).  `fg_bias` shifts the final logit so a
    modest fraction of macroblocks comes out foreground.
    """
    rng = np.random.default_rng(seed)
    t = {}
    for name, shape in tensor_specs().items():
        kind = name.split(".", 1)[1]
        if kind == "conv.kernel":
            fan_in = shape[0] * shape[1] * shape[2]
            t[name] = rng.normal(0.0, np.sqrt(2.0 / fan_in), shape)
        elif kind == "up.kernel":
            # Keras fan_in for Conv3DTranspose kernel [kh,kw,Cout,Cin] = kh*kw*Cout
            fan_in = shape[0] * shape[1] * shape[2]
            t[name] = rng.normal(0.0, np.sqrt(2.0 / fan_in), shape)
        elif kind.endswith("bias"):
            t[name] = rng.normal(0.0, 0.05, shape)
        elif kind == "bn.gamma":
            t[name] = rng.uniform(0.5, 1.5, shape)
        elif kind in ("bn.beta", "bn.mean"):
            t[name] = rng.normal(0.0, 0.1, shape)
        elif kind == "bn.var":
            t[name] = rng.uniform(0.5, 1.5, shape)
        elif kind in ("tmix.w1", "tmix.w2"):
            t[name] = rng.normal(0.0, np.sqrt(2.0 / T), shape)
        elif name == "final.kernel":
            t[name] = rng.normal(0.0, np.sqrt(2.0 / 16), shape)
        else:
            raise AssertionError(name)
    t["final.bias"] = np.array([fg_bias])
    return flatten(t)


def blob_like(seed: int = 7, noise: float = 0.04) -> np.ndarray:
    """A seeded weight set whose masks look like a trained BlobNet's: connected blobs over the moving objects of the
    synthetic inputs, background empty (random_init gives salt-and-pepper masks with ~500 one-macroblock components per
    1080p frame -- pessimistic for bboxcc and meaningless for the trackers behind it; the reference ships no weights).

    random_init(seed) scaled down to `noise` on every kernel, plus one hand-made signal path through the reference
    architecture: encoder level 0, channel 0 = relu(3x3 mean of (mv_x + mv_y) / 2 - 0.35) (motion-vector energy; the
    inputs arrive as clip(x, 0, 6) / 6, utils/model/preprocessing.py:6-7), which reaches the last decoder block through
    the level-0 skip connection (utils/model/blobnet.py:32, decoder.py:128), is spread by the 4x4 stride-2 transposed
    convolution and read by the final 1x1 convolution.  The kernels do the same work whatever the weights are."""
    t = unflatten(random_init(seed))
    for name in t:
        kind = name.split(".", 1)[1]
        if kind in ("conv.kernel", "up.kernel") or name == "final.kernel" or kind.startswith("tmix"):
            t[name] = t[name] * noise
        elif kind.endswith("bias"):
            t[name] = t[name] * noise
    for i in range(4):      # BN of the signal channel: identity
        for n, v in (("gamma", 1.0), ("beta", 0.0), ("mean", 0.0), ("var", 1.0)):
            t[f"enc{i}.bn.{n}"][0] = v
    k0 = t["enc0.conv.kernel"]            # [3][3][Cin=3][Cout=16]
    k0[:, :, :, 0] = 0.0
    k0[:, :, 1, 0] = 1.0 / 18.0
    k0[:, :, 2, 0] = 1.0 / 18.0
    t["enc0.conv.bias"][0] = -0.35
    for w in ("enc0.tmix.w1", "enc0.tmix.w2"):
        t[w] = t[w] * 0.0 + t[w] * 0.25   # the temporal MLP adds little: out = relu(relu(..) + p) ~ p
    up = t["dec3.up.kernel"]              # [4][4][Cout=16][Cin=32]: input channels 16.. are the level-0 skip
    up[:, :, 0, :] = 0.0
    up[:, :, 0, 16] = 0.25
    t["dec3.up.bias"][0] = 0.0
    t["final.kernel"][0] = 22.0
    t["final.bias"] = np.array([-3.0])
    return flatten(t)
