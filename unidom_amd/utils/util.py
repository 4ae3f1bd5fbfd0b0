"""Reward distances (the only pieces of the reference's utils/util.py that feed the gradient).

calc_chamfer / calc_l2 mirror /root/reference/DaXBench/daxbench/core/utils/util.py:138-159 on torch
device tensors (note the per-point distance is an RMS over xyz, `sqrt(mean((a-b)^2))`, not Euclidean).
"""
from __future__ import annotations

import torch


def calc_chamfer(x, y, metric="l2", direction="bi"):   # util.py:138-153
    """x [B,P,3], y [Q,3] -> [B]"""
    d = torch.sqrt(((x[:, :, None, :] - y[None, None, :, :]) ** 2).mean(-1))   # [B,P,Q]
    x2y_min = d.min(-1).values.mean(1)
    y2x_min = d.min(-2).values.mean(1)
    return y2x_min + x2y_min


def calc_l2(x, y):                                      # util.py:156-159
    return torch.sqrt(((x - y[None]) ** 2).mean(-1)).mean(-1)
