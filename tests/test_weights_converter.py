"""tools/keras_npz_to_cvhw.py: Keras-named variables of the reference model (utils/model/*.py) -> the weight blob of
covahip_blobnet_load.  The reference's own export chain is SavedModel -> ONNX -> TensorRT (model/tasks.py:17-60)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from cova_amd import synth
from cova_amd import weights as W
from tests import torch_blobnet as tb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import keras_npz_to_cvhw as K  # noqa: E402


@pytest.mark.parametrize("prefix,first", [("", 0), ("blob_net/encoder/", 5), ("model/", 17)])
def test_round_trip_through_keras_names(prefix, first):
    flat = W.random_init(99)
    arrays = K.to_keras_arrays(flat, prefix=prefix, first_index=first)
    assert len(arrays) == 2 * 5 + 4 * 7 + 8 + 2 * 4           # kernels + biases, BN quadruples, Conv1D kernels, convT pairs
    # a process that trained the model also holds optimizer slots: ignored
    arrays["Adam/conv3d/kernel/m:0"] = np.zeros((1, 3, 3, 3, 16), np.float32)
    arrays["iter:0"] = np.zeros((), np.int64)
    back = K.convert(arrays)
    np.testing.assert_array_equal(back, flat)


def test_wrong_hyper_parameters_and_missing_layers_are_refused():
    arrays = K.to_keras_arrays(W.random_init(3))
    bad = dict(arrays)
    bad["conv3d_1/kernel:0"] = np.zeros((1, 3, 3, 16, 48), np.float32)
    with pytest.raises(ValueError, match="shape"):
        K.convert(bad)
    bad = {k: v for k, v in arrays.items() if not k.startswith("conv1d_7/")}
    with pytest.raises(ValueError, match="conv1d"):
        K.convert(bad)


def test_command_line_writes_a_loadable_blob(tmp_path):
    flat = W.random_init(4)
    np.savez(tmp_path / "w.npz", **K.to_keras_arrays(flat, prefix="blob_net/"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "keras_npz_to_cvhw.py"), str(tmp_path / "w.npz"),
                        str(tmp_path / "w.cvhw")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    np.testing.assert_array_equal(W.from_bytes((tmp_path / "w.cvhw").read_bytes()), flat)


def test_keras_layouts_mean_what_the_forward_pass_assumes():
    """One layer of every kind computed straight from the KERAS-layout arrays (Conv3D kernel [kd,kh,kw,Cin,Cout], Conv3DTranspose
    kernel [kd,kh,kw,Cout,Cin], Conv1D kernel [1,Tin,Tout], BN gamma/beta/moving_*) against the same layer of tests/torch_blobnet.py
    fed with the converted blob: the axis order the converter documents is the one the network code consumes."""
    flat = W.random_init(21)
    ka = {k: torch.from_numpy(np.asarray(v)) for k, v in K.to_keras_arrays(flat).items()}
    h, w = 20, 24
    stack = synth.stacked_batch(1, h, w, seed=3)
    _, levels = tb.forward(flat, stack, h, w, return_levels=True)
    # encoder level 0 from the Keras arrays
    x = torch.from_numpy(np.ascontiguousarray(stack[..., :3])).float().permute(0, 3, 1, 2).reshape(1, 3, 4, h, w)
    x = torch.clamp(x, 0.0, 6.0) / 6.0
    k = ka["conv3d/kernel:0"].permute(4, 3, 0, 1, 2)                           # [Cout, Cin, kd, kh, kw]
    x = F.relu(F.conv3d(x, k, ka["conv3d/bias:0"], padding=(0, 1, 1)))
    x = F.batch_norm(x, ka["batch_normalization/moving_mean:0"], ka["batch_normalization/moving_variance:0"],
                     ka["batch_normalization/gamma:0"], ka["batch_normalization/beta:0"], training=False, eps=1e-3)
    x = F.max_pool3d(x, (1, 2, 2))
    y = x.permute(0, 1, 3, 4, 2)                                               # [N,C,H,W,T]: Conv1D over the last axis
    y = F.relu(F.conv1d(y.reshape(-1, 4, 1), ka["conv1d/kernel:0"].permute(2, 1, 0)))      # Keras [k,Tin,Tout] -> torch [Tout,Tin,k]
    y = F.relu(F.conv1d(y, ka["conv1d_1/kernel:0"].permute(2, 1, 0))).reshape(x.permute(0, 1, 3, 4, 2).shape)
    lvl0 = F.relu(y.permute(0, 1, 4, 2, 3) + x)
    np.testing.assert_allclose(lvl0.numpy(), levels[0], rtol=1e-5, atol=1e-5)
    # first decoder block from the Keras arrays (conv3d_transpose = block 0; its shape probe took index 1)
    skip = torch.from_numpy(levels[3])[:, :, :1]
    kt = ka["conv3d_transpose/kernel:0"].permute(4, 3, 0, 1, 2)                # [Cin, Cout, kd, kh, kw]
    up_k = F.conv_transpose3d(F.relu(skip), kt, ka["conv3d_transpose/bias:0"], stride=(1, 2, 2))
    wt = W.unflatten(flat)
    kk = torch.from_numpy(wt["dec0.up.kernel"]).permute(3, 2, 0, 1).unsqueeze(2)
    up_t = F.conv_transpose3d(F.relu(skip), kk, torch.from_numpy(wt["dec0.up.bias"]), stride=(1, 2, 2))
    np.testing.assert_allclose(up_k.numpy(), up_t.numpy(), rtol=1e-6, atol=1e-6)
