"""Oracle (numpy fp32, CPU) for the IWE primitives  --  test infrastructure only.

Restates reference utils/iwe.py:4-153.  The float32 operation ORDER of the
reference is kept literally (no fused multiply-add, same association), since
the rounded-index IWE must be bit-exact.
"""

import numpy as np

F32 = np.float32


def warp_events(events, flow, tref, flow_scaling):
    """warped(y,x) = (y,x) + ((tref - t) * f) * S   -- utils/iwe.py:37
    events [B,N,4] (t,y,x,p), flow [B,N,2] (fy,fx) -> [B,N,2]."""
    events = np.asarray(events, dtype=F32)
    flow = np.asarray(flow, dtype=F32)
    dt = (F32(tref) - events[:, :, 0:1]).astype(F32)
    return (events[:, :, 1:3] + ((dt * flow).astype(F32) * F32(flow_scaling)).astype(F32)).astype(F32)


def purge_unfeasible(idx, res):
    """Out-of-image corners: index -> 0, weight mask -> 0.  utils/iwe.py:4-17."""
    bad = (idx[:, :, 0:1] < 0) | (idx[:, :, 0:1] >= res[0]) | (idx[:, :, 1:2] < 0) | (idx[:, :, 1:2] >= res[1])
    mask = np.where(bad, F32(0), F32(1)).astype(F32)
    return (idx * mask).astype(F32), mask


def get_interpolation(events, flow, tref, res, flow_scaling, round_idx=False):
    """Linear pixel index + weight per (event, corner).  utils/iwe.py:20-74.
    Returns idx [B,M,1] (float32, integer valued), weights [B,M,1];
    M = N (round_idx) or 4N ordered (top-left, top-right, bottom-left,
    bottom-right) blocks of N."""
    w = warp_events(events, flow, tref, flow_scaling)
    if round_idx:
        idx = np.rint(w).astype(F32)  # torch.round: half to even
        weights = np.ones_like(idx)
    else:
        top = np.floor(w[:, :, 0:1])
        bot = np.floor(w[:, :, 0:1] + F32(1))
        left = np.floor(w[:, :, 1:2])
        right = np.floor(w[:, :, 1:2] + F32(1))
        idx = np.concatenate(
            [
                np.concatenate([top, left], 2),
                np.concatenate([top, right], 2),
                np.concatenate([bot, left], 2),
                np.concatenate([bot, right], 2),
            ],
            1,
        ).astype(F32)
        w4 = np.concatenate([w, w, w, w], 1)
        weights = np.maximum(F32(0), F32(1) - np.abs(w4 - idx)).astype(F32)
    idx, mask = purge_unfeasible(idx, res)
    weights = (weights[:, :, 0:1] * weights[:, :, 1:2]).astype(F32) * mask
    lin = (idx[:, :, 0:1] * F32(res[1]) + idx[:, :, 1:2]).astype(F32)
    return lin, weights.astype(F32)


def interpolate(idx, weights, res, polarity_mask=None):
    """Scatter-add weights into a [B,1,H,W] image.  utils/iwe.py:77-92."""
    if polarity_mask is not None:
        weights = (weights * polarity_mask).astype(F32)
    B = idx.shape[0]
    iwe = np.zeros((B, res[0] * res[1]), dtype=F32)
    ii = idx[:, :, 0].astype(np.int64)
    for b in range(B):
        np.add.at(iwe[b], ii[b], weights[b, :, 0])
    return iwe.reshape(B, 1, res[0], res[1])


def gather_event_flow(flow, event_list, res):
    """Per-event (fy, fx) looked up at the event's pixel.
    utils/iwe.py:108-119 / loss/flow.py:65-84.  flow [B,2,H,W] = (x, y) maps."""
    lin = (event_list[:, :, 1] * F32(res[1]) + event_list[:, :, 2]).astype(F32).astype(np.int64)
    f = np.asarray(flow, dtype=F32).reshape(flow.shape[0], 2, -1)
    fy = np.take_along_axis(f[:, 1, :], lin, 1)
    fx = np.take_along_axis(f[:, 0, :], lin, 1)
    return np.stack([fy, fx], 2).astype(F32)


def deblur_events(flow, event_list, res, flow_scaling=128, round_idx=True, polarity_mask=None):
    """utils/iwe.py:95-129 (tref = 1)."""
    ef = gather_event_flow(flow, event_list, res)
    idx, w = get_interpolation(event_list, ef, 1, res, flow_scaling, round_idx=round_idx)
    if not round_idx and polarity_mask is not None:
        polarity_mask = np.concatenate([polarity_mask] * 4, 1)
    return interpolate(idx, w, res, polarity_mask)


def compute_pol_iwe(flow, event_list, res, pos_mask, neg_mask, flow_scaling=128, round_idx=True):
    """Per-polarity image of warped events [B,2,H,W].  utils/iwe.py:132-153."""
    pos = deblur_events(flow, event_list, res, flow_scaling, round_idx, pos_mask)
    neg = deblur_events(flow, event_list, res, flow_scaling, round_idx, neg_mask)
    return np.concatenate([pos, neg], 1)
