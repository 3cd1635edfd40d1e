"""Model helpers -- same names and results as reference models/model_util.py (host-side shape arithmetic and state
copying; nothing here is on the GPU hot path except the `skip_sum` addition, which runs in libevflow_hip.so)."""

import copy

import torch
from torch.nn import ZeroPad2d


def _centred(x, like):
    """x zero-padded to the spatial size of `like`, the odd pixel going to the right / bottom."""
    gap_h, gap_w = like.shape[2] - x.shape[2], like.shape[3] - x.shape[3]
    if gap_h == 0 and gap_w == 0:
        return x
    y = torch.nn.functional.pad(x, (gap_w // 2, gap_w - gap_w // 2, gap_h // 2, gap_h - gap_h // 2))
    for attr in ("_evf_spike", "_evf_spike_int"):  # zero padding keeps the provenance of a spike-valued tensor (hip_ops.spike_tag)
        if hasattr(x, attr):
            setattr(y, attr, getattr(x, attr))
    return y


def skip_concat(x1, x2):
    """cat([pad(x1), x2]) on channels.  Reference: models/model_util.py:14-19."""
    return torch.cat([_centred(x1, x2), x2], dim=1)


def skip_sum(x1, x2):
    """pad(x1) + x2.  Reference: models/model_util.py:22-27."""
    from . import hip_ops

    return hip_ops.add(_centred(x1, x2), x2)


def optimal_crop_size(max_size, max_subsample_factor, safety_margin=0):
    """Smallest multiple of 2^factor that is >= max_size, plus `safety_margin` such multiples.
    Reference: models/model_util.py:30-38."""
    step = 1 << int(max_subsample_factor)
    blocks = -(-int(max_size) // step) if float(max_size).is_integer() else int(-(-max_size // step))
    return (blocks + safety_margin) * step


class CropParameters:
    """Pad an image so that every encoder halves an even size, and find the original again afterwards:
    `.pad(x)` (ZeroPad2d), `.crop(x)`, `.ix0/.ix1/.iy0/.iy1`, `.padding_{top,bottom,left,right}`,
    `.{width,height}_crop_size`.  Reference: models/model_util.py:41-79."""

    def __init__(self, width, height, num_encoders, safety_margin=0):
        self.width, self.height, self.num_encoders = width, height, num_encoders
        spans = {}
        for axis, size in (("width", width), ("height", height)):
            full = optimal_crop_size(size, num_encoders, safety_margin)
            slack = full - size
            lead, trail = (slack + 1) // 2, slack // 2  # ceil / floor of half the slack
            start = full // 2 - size // 2
            spans[axis] = (full, lead, trail, start, start + size)
        self.width_crop_size, self.padding_left, self.padding_right, self.ix0, self.ix1 = spans["width"]
        self.height_crop_size, self.padding_top, self.padding_bottom, self.iy0, self.iy1 = spans["height"]
        self.cx, self.cy = self.width_crop_size // 2, self.height_crop_size // 2
        self.pad = ZeroPad2d((self.padding_left, self.padding_right, self.padding_top, self.padding_bottom))

    def crop(self, img):
        return img[..., self.iy0 : self.iy1, self.ix0 : self.ix1]


def recursive_clone(tensor):
    """Clone of a tensor, or of every tensor inside nested tuples / lists (containers keep their type).
    Reference: models/model_util.py:82-93."""
    if torch.is_tensor(tensor) or hasattr(tensor, "clone"):
        return tensor.clone()
    if isinstance(tensor, (list, tuple)):
        return type(tensor)(recursive_clone(item) for item in tensor)
    return copy.deepcopy(tensor)


def copy_states(states):
    """States of a model: cloned, unless the list still starts with None (then a plain deep copy).
    Reference: models/model_util.py:96-102."""
    return recursive_clone(states) if states[0] is not None else copy.deepcopy(states)
