"""Oracle (numpy / torch-CPU) for the normalisation layers of the reference's ANN baselines  --  test infrastructure only.

Restates what nn.BatchNorm2d / nn.InstanceNorm2d(track_running_stats=True) do inside reference models/submodules.py:46-56
(ConvLayer), :122-132 (TransposedConvLayer), :169-180 (UpsampleConvLayer), :273-301 (ResidualBlock): training mode
normalises with the statistics of the input (batch norm: per channel over B,H,W; instance norm: per sample and channel
over H,W; biased variance) and moves the running statistics with the UNBIASED variance (instance norm: averaged over the
batch); eval mode normalises with the running statistics -- for the instance norm too, because the reference constructs it
with track_running_stats=True."""

import numpy as np


def norm2d(x, weight, bias, running_mean, running_var, *, instance, training, momentum=0.1, eps=1e-5):
    """x [B,C,H,W] float32 -> (y, new_running_mean, new_running_var)."""
    x = np.asarray(x, np.float64)
    B, C, H, W = x.shape
    axes = (2, 3) if instance else (0, 2, 3)
    n = H * W if instance else B * H * W
    rm, rv = np.asarray(running_mean, np.float64), np.asarray(running_var, np.float64)
    if training:
        mean = x.mean(axes, keepdims=True)
        var = x.var(axes, keepdims=True)  # biased
        unb = var * n / max(n - 1, 1)
        m_c = mean.reshape(-1, C).mean(0) if instance else mean.reshape(C)
        v_c = unb.reshape(-1, C).mean(0) if instance else unb.reshape(C)
        rm = (1 - momentum) * rm + momentum * m_c
        rv = (1 - momentum) * rv + momentum * v_c
    else:
        mean, var = rm.reshape(1, C, 1, 1), rv.reshape(1, C, 1, 1)
    y = (x - mean) / np.sqrt(var + eps)
    if weight is not None:
        y = y * np.asarray(weight, np.float64).reshape(1, C, 1, 1)
    if bias is not None:
        y = y + np.asarray(bias, np.float64).reshape(1, C, 1, 1)
    return y.astype(np.float32), rm.astype(np.float32), rv.astype(np.float32)
