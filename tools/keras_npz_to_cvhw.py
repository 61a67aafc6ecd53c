#!/usr/bin/env python3
"""Trained BlobNet weights -> the weight blob covahip_blobnet_load takes (.cvhw, cova_amd/weights.py).

The reference exports its trained Keras model as SavedModel -> ONNX -> TensorRT engine (model/tasks.py:17-60); this
library takes the weights themselves.  On the machine that has TensorFlow and the trained model:

    model = tf.keras.models.load_model("model/tf_model/<dataset>")        # or the BlobNet(...) object after training
    np.savez("blobnet.npz", **{w.name: w.numpy() for w in model.weights})

and here:   python tools/keras_npz_to_cvhw.py blobnet.npz blobnet.cvhw

Variable names are Keras' automatic layer names: `<layer>[_<k>]/<variable>[:0]`, possibly behind name-scope prefixes
(`blob_net/encoder/conv3d_1/kernel:0` ...).  The counter k depends on what else the exporting process had built, so the
layers are matched by TYPE, ORDER OF CREATION and SHAPE, never by the number itself:

  layer type (creation order in utils/model)               variables (Keras layout)                     -> tensor here
  conv3d          x4 encoder (encoder.py:35-43), then       kernel [1,3,3,Cin,Cout], bias [Cout]          enc{i}.conv.kernel [3,3,Cin,Cout], .bias
                  the final 1x1x1 (decoder.py:118)          kernel [1,1,1,16,1], bias [1]                 final.kernel [16], final.bias [1]
  batch_normalization x4 encoder (encoder.py:46),           gamma, beta, moving_mean, moving_variance     enc{i}.bn.{gamma,beta,mean,var}
                  then x3 decoder (decoder.py:106)                                                        dec{j}.bn.*
  conv1d          x8: two per encoder level                 kernel [1,4,4] = [1,Tin,Tout], no bias        enc{i}.tmix.w1 / w2 [Tin,Tout]
                  (pointwise.py:8-12)
  conv3d_transpose x4 (decoder.py:9-24; the shape-probe     kernel [1,4,4,Cout,Cin], bias [Cout]          dec{j}.up.kernel [4,4,Cout,Cin], .bias
                  layers of :27-40 own no variables in the
                  model and only consume name indices)

Hyper-parameters are the reference's (utils/train-blobnet.py:57-69); anything else is refused.  Optimizer slots and
other variables that belong to no layer above are ignored with a note.
"""
from __future__ import annotations

import os
import re
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cova_amd import weights as W  # noqa: E402

_VAR = re.compile(r"(?:^|/)(conv3d_transpose|conv3d|conv1d|batch_normalization)(?:_(\d+))?/([a-z_]+)(?::0)?$")


def group_layers(npz_keys):
    """{layer type: [(creation index, {variable: key}), ...] sorted by creation index}; unknown keys are returned too."""
    layers, unknown = {}, []
    for key in npz_keys:
        m = _VAR.search(key)
        if not m:
            unknown.append(key)
            continue
        kind, idx, var = m.group(1), int(m.group(2) or 0), m.group(3)
        layers.setdefault(kind, {}).setdefault(idx, {})[var] = key
    return {k: sorted(v.items()) for k, v in layers.items()}, unknown


def convert(arrays: dict) -> np.ndarray:
    """Keras-named arrays -> flat fp32 parameter vector in the order of cova_amd.weights.tensor_specs()."""
    layers, unknown = group_layers(arrays.keys())

    def take(kind, n):
        got = layers.get(kind, [])
        if len(got) != n:
            raise ValueError(f"expected {n} {kind} layers with variables, found {len(got)}: {[i for i, _ in got]}")
        return [v for _, v in got]

    def arr(key, shape):
        a = np.asarray(arrays[key], dtype=np.float32)
        if a.shape != tuple(shape):
            raise ValueError(f"{key}: shape {a.shape}, expected {tuple(shape)} (hyper-parameters of utils/train-blobnet.py:57-69)")
        return a

    t = {}
    conv3d = take("conv3d", 5)
    bns = take("batch_normalization", 7)
    conv1d = take("conv1d", 8)
    convt = take("conv3d_transpose", 4)
    for i in range(4):
        ci, co = W.ENC_C[i], W.ENC_C[i + 1]
        t[f"enc{i}.conv.kernel"] = arr(conv3d[i]["kernel"], (1, 3, 3, ci, co))[0]
        t[f"enc{i}.conv.bias"] = arr(conv3d[i]["bias"], (co,))
        for ours, keras_name in (("gamma", "gamma"), ("beta", "beta"), ("mean", "moving_mean"), ("var", "moving_variance")):
            t[f"enc{i}.bn.{ours}"] = arr(bns[i][keras_name], (co,))
        t[f"enc{i}.tmix.w1"] = arr(conv1d[2 * i]["kernel"], (1, W.T, W.T))[0]
        t[f"enc{i}.tmix.w2"] = arr(conv1d[2 * i + 1]["kernel"], (1, W.T, W.T))[0]
    for j in range(4):
        ci, co = W.DEC_CI[j], W.DEC_CO[j]
        t[f"dec{j}.up.kernel"] = arr(convt[j]["kernel"], (1, 4, 4, co, ci))[0]
        t[f"dec{j}.up.bias"] = arr(convt[j]["bias"], (co,))
        if j < 3:
            for ours, keras_name in (("gamma", "gamma"), ("beta", "beta"), ("mean", "moving_mean"), ("var", "moving_variance")):
                t[f"dec{j}.bn.{ours}"] = arr(bns[4 + j][keras_name], (co,))
    t["final.kernel"] = arr(conv3d[4]["kernel"], (1, 1, 1, 16, 1)).reshape(16)
    t["final.bias"] = arr(conv3d[4]["bias"], (1,))
    if unknown:
        print(f"note: {len(unknown)} variables belong to no BlobNet layer and were ignored, e.g. {unknown[:3]}", file=sys.stderr)
    return W.flatten(t)


def to_keras_arrays(flat: np.ndarray, prefix: str = "", first_index: int = 0) -> dict:
    """The inverse (what `{w.name: w.numpy() for w in model.weights}` of a model with these weights holds): used by the
    tests, and handy for moving weights the other way.  first_index shifts every layer counter, as a process that had
    built other layers before would."""
    t = W.unflatten(np.asarray(flat, dtype=np.float32))
    out = {}

    def name(kind, k, var):
        k += first_index
        return f"{prefix}{kind}{'_' + str(k) if k else ''}/{var}:0"

    for i in range(4):
        out[name("conv3d", i, "kernel")] = t[f"enc{i}.conv.kernel"][None]
        out[name("conv3d", i, "bias")] = t[f"enc{i}.conv.bias"]
        for ours, keras_name in (("gamma", "gamma"), ("beta", "beta"), ("mean", "moving_mean"), ("var", "moving_variance")):
            out[name("batch_normalization", i, keras_name)] = t[f"enc{i}.bn.{ours}"]
        out[name("conv1d", 2 * i, "kernel")] = t[f"enc{i}.tmix.w1"][None]
        out[name("conv1d", 2 * i + 1, "kernel")] = t[f"enc{i}.tmix.w2"][None]
    for j in range(4):
        # every block's shape-probe layer takes the name index behind the block's own layer (decoder.py:27-40)
        out[name("conv3d_transpose", 2 * j, "kernel")] = t[f"dec{j}.up.kernel"][None]
        out[name("conv3d_transpose", 2 * j, "bias")] = t[f"dec{j}.up.bias"]
        if j < 3:
            for ours, keras_name in (("gamma", "gamma"), ("beta", "beta"), ("mean", "moving_mean"), ("var", "moving_variance")):
                out[name("batch_normalization", 4 + j, keras_name)] = t[f"dec{j}.bn.{ours}"]
    out[name("conv3d", 4, "kernel")] = t["final.kernel"].reshape(1, 1, 1, 16, 1)
    out[name("conv3d", 4, "bias")] = t["final.bias"]
    return out


def main(argv):
    if len(argv) != 3:
        print(__doc__.split("\n\n")[1], file=sys.stderr)
        print(f"usage: {argv[0]} weights.npz out.cvhw", file=sys.stderr)
        return 2
    with np.load(argv[1]) as z:
        flat = convert({k: z[k] for k in z.files})
    with open(argv[2], "wb") as f:
        f.write(W.to_bytes(flat))
    print(f"{argv[2]}: {flat.size} parameters, {os.path.getsize(argv[2])} bytes")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
