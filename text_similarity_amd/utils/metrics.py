"""``cos_sim`` (/root/reference/src/utils/metrics.py:81-101): dense cosine matrix, computed by the HIP kernel
behind ``tsim_cos_sim``.  Same promotion rules: non-tensors are converted, 1-D inputs become one row; no eps, so a
zero row yields NaN exactly like the reference."""
from __future__ import annotations

import torch

from .. import ops


def cos_sim(a, b, device=None):
    if not isinstance(a, torch.Tensor):
        a = torch.tensor(a)
    if not isinstance(b, torch.Tensor):
        b = torch.tensor(b)
    if len(a.shape) == 1:
        a = a.unsqueeze(0)
    if len(b.shape) == 1:
        b = b.unsqueeze(0)
    dev = device or (a.device if a.is_cuda else (b.device if b.is_cuda else torch.device("cuda")))
    return ops.cos_sim_dense(a.to(dev), b.to(dev))
