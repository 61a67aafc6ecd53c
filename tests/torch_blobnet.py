"""Independent torch.nn.functional composition of the BlobNet graph (test helper).

Written from the reference's Keras model definition (utils/model/blobnet.py:8-48,
encoder.py:30-80, pointwise.py:5-26, decoder.py:5-134, preprocessing.py:6-7), using
torch's conv3d / conv_transpose3d / max_pool3d / batch_norm primitives rather than the
hand loops of oracle/blobnet_ref.c, so the two restatements check each other.
CPU fp32 (or fp64 when `dtype=torch.float64`).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from cova_amd import weights as W

BN_EPS = 1e-3  # Keras BatchNormalization default


def forward(flat_weights: np.ndarray, stack: np.ndarray, h: int, w: int, dtype=torch.float32,
            return_levels: bool = False):
    wt = {k: torch.from_numpy(np.array(v)).to(dtype) for k, v in W.unflatten(flat_weights).items()}
    b = stack.shape[0]
    x = torch.from_numpy(np.ascontiguousarray(stack[..., :3])).to(dtype)       # [B, T*H, W, 3]
    x = x.permute(0, 3, 1, 2).reshape(b, 3, W.T, h, w)                         # Reshape((3,4,H,W))
    x = torch.clamp(x, 0.0, 6.0) / 6.0
    levels = []
    for i in range(4):
        k = wt[f"enc{i}.conv.kernel"].permute(3, 2, 0, 1).unsqueeze(2)        # [Cout,Cin,1,3,3]
        x = F.relu(F.conv3d(x, k, wt[f"enc{i}.conv.bias"], padding=(0, 1, 1)))
        hh, ww = x.shape[-2], x.shape[-1]
        x = F.batch_norm(x, wt[f"enc{i}.bn.mean"], wt[f"enc{i}.bn.var"], wt[f"enc{i}.bn.gamma"],
                         wt[f"enc{i}.bn.beta"], training=False, eps=BN_EPS)
        x = F.max_pool3d(x, (1, 2, 2))
        if hh % 2:
            x = F.pad(x, (0, 0, 1, 0))                                         # zero row on top
        if ww % 2:
            x = F.pad(x, (1, 0, 0, 0))                                         # zero column on the left
        y = x.permute(0, 1, 3, 4, 2)                                           # [N,C,H,W,T]
        y = F.relu(y @ wt[f"enc{i}.tmix.w1"])
        y = F.relu(y @ wt[f"enc{i}.tmix.w2"])
        x = F.relu(y.permute(0, 1, 4, 2, 3) + x)
        levels.append(x)
    skips = [lv[:, :, :1] for lv in reversed(levels)]
    shapes = [s.shape for s in skips] + [(b, 3, W.T, h, w)]
    x = skips[0]
    for j in range(4):
        k = wt[f"dec{j}.up.kernel"].permute(3, 2, 0, 1).unsqueeze(2)          # [Cin,Cout,1,4,4]
        x = F.conv_transpose3d(F.relu(x), k, wt[f"dec{j}.up.bias"], stride=(1, 2, 2))
        ph = x.shape[-2] - shapes[j + 1][-2]
        pw = x.shape[-1] - shapes[j + 1][-1]
        assert ph >= 0 and pw >= 0
        x = x[..., ph // 2 + ph % 2: x.shape[-2] - ph // 2, pw // 2 + pw % 2: x.shape[-1] - pw // 2]
        if j < 3:
            x = F.batch_norm(x, wt[f"dec{j}.bn.mean"], wt[f"dec{j}.bn.var"], wt[f"dec{j}.bn.gamma"],
                             wt[f"dec{j}.bn.beta"], training=False, eps=BN_EPS)
            x = torch.cat([x, skips[j + 1]], dim=1)
    logit = (x * wt["final.kernel"].view(1, -1, 1, 1, 1)).sum(1, keepdim=True) + wt["final.bias"]
    logit = logit[:, 0, 0]                                                     # [B,H,W]
    if return_levels:
        return logit.numpy(), [lv.numpy() for lv in levels]
    return logit.numpy()
