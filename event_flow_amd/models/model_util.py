"""Model helpers -- mirror of reference models/model_util.py (host-side shape
arithmetic and state copying; nothing here is on the GPU hot path)."""

import copy
from math import ceil, floor

import torch
from torch.nn import ZeroPad2d


def skip_concat(x1, x2):
    """Zero-pad x1 to x2's spatial size and concatenate on channels.
    Reference: models/model_util.py:14-19."""
    dh, dw = x2.shape[2] - x1.shape[2], x2.shape[3] - x1.shape[3]
    x1 = torch.nn.functional.pad(x1, (dw // 2, dw - dw // 2, dh // 2, dh - dh // 2))
    return torch.cat([x1, x2], dim=1)


def skip_sum(x1, x2):
    """Reference: models/model_util.py:22-27."""
    dh, dw = x2.shape[2] - x1.shape[2], x2.shape[3] - x1.shape[3]
    x1 = torch.nn.functional.pad(x1, (dw // 2, dw - dw // 2, dh // 2, dh - dh // 2))
    from . import hip_ops  # the sum runs in libevflow_hip.so

    return hip_ops.add(x1, x2)


def optimal_crop_size(max_size, max_subsample_factor, safety_margin=0):
    """Smallest multiple of 2^factor that is >= max_size + margin.
    Reference: models/model_util.py:30-38."""
    k = 2 ** max_subsample_factor
    return int(k * ceil(max_size / k)) + safety_margin * k


class CropParameters:
    """Pad/crop bookkeeping so every encoder sees even sizes.
    Reference: models/model_util.py:41-79."""

    def __init__(self, width, height, num_encoders, safety_margin=0):
        self.height, self.width, self.num_encoders = height, width, num_encoders
        self.width_crop_size = optimal_crop_size(width, num_encoders, safety_margin)
        self.height_crop_size = optimal_crop_size(height, num_encoders, safety_margin)
        self.padding_top = ceil(0.5 * (self.height_crop_size - height))
        self.padding_bottom = floor(0.5 * (self.height_crop_size - height))
        self.padding_left = ceil(0.5 * (self.width_crop_size - width))
        self.padding_right = floor(0.5 * (self.width_crop_size - width))
        self.pad = ZeroPad2d((self.padding_left, self.padding_right, self.padding_top, self.padding_bottom))
        self.cx, self.cy = floor(self.width_crop_size / 2), floor(self.height_crop_size / 2)
        self.ix0, self.ix1 = self.cx - floor(width / 2), self.cx + ceil(width / 2)
        self.iy0, self.iy1 = self.cy - floor(height / 2), self.cy + ceil(height / 2)

    def crop(self, img):
        return img[..., self.iy0 : self.iy1, self.ix0 : self.ix1]


def recursive_clone(tensor):
    """Deep clone of a tensor or a (nested) tuple/list of tensors.
    Reference: models/model_util.py:82-93."""
    if hasattr(tensor, "clone"):
        return tensor.clone()
    try:
        return type(tensor)(recursive_clone(t) for t in tensor)
    except TypeError:
        return copy.deepcopy(tensor)


def copy_states(states):
    """Reference: models/model_util.py:96-102 (a list whose first entry is None
    is passed through unchanged)."""
    if states[0] is None:
        return copy.deepcopy(states)
    return recursive_clone(states)
