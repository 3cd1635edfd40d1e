"""Oracle (numpy, CPU) for the event encodings  --  test infrastructure only.

Restates reference dataloader/encodings.py:30-85 and the layout-defining
helpers of dataloader/base.py:66-86,159-222,248-265 on numpy arrays.
All arithmetic is float32 like the reference's torch tensors.
"""

import numpy as np

F32 = np.float32


def events_to_image(xs, ys, vals, sensor_size, accumulate=True):
    """img[y, x] (+)= val.  Reference: dataloader/encodings.py:30-45
    (index_put_ with accumulate=True/False; indices are truncated to int64)."""
    H, W = int(sensor_size[0]), int(sensor_size[1])
    img = np.zeros((H, W), dtype=F32)
    xi = np.asarray(xs).astype(np.int64)
    yi = np.asarray(ys).astype(np.int64)
    v = np.asarray(vals, dtype=F32)
    if accumulate:
        # np.add.at applies the additions one by one in event order, in fp32,
        # exactly as the sequential CPU index_put_ does.
        np.add.at(img, (yi, xi), v)
    else:
        img[yi, xi] = v  # last write wins
    return img


def events_to_channels(xs, ys, ps, sensor_size):
    """Two-channel per-polarity event count.  Reference:
    dataloader/encodings.py:70-85 (value added is ps*mask = p^2 for the
    event's own polarity, 0 otherwise)."""
    ps = np.asarray(ps, dtype=F32)
    assert len(xs) == len(ys) == len(ps)
    mask_pos = np.where(ps < 0, F32(0), ps)
    mask_neg = np.where(ps > 0, F32(0), ps)
    pos = events_to_image(xs, ys, ps * mask_pos, sensor_size)
    neg = events_to_image(xs, ys, ps * mask_neg, sensor_size)
    return np.stack([pos, neg])


def events_to_voxel(xs, ys, ts, ps, num_bins, sensor_size, round_ts=False):
    """Temporal-bilinear voxel grid.  Reference: dataloader/encodings.py:48-67."""
    ts = np.asarray(ts, dtype=F32)
    ps = np.asarray(ps, dtype=F32)
    assert len(xs) == len(ys) == len(ts) == len(ps)
    t = ts * F32(num_bins - 1)
    if round_ts:
        t = np.rint(t).astype(F32)  # torch.round = half-to-even
    out = []
    for b in range(num_bins):
        w = np.maximum(F32(0), F32(1.0) - np.abs(t - F32(b))).astype(F32)
        out.append(events_to_image(xs, ys, ps * w, sensor_size))
    return np.stack(out)


def event_formatting(xs, ys, ts, ps):
    """ts -> [0,1], p in {0,1} -> {-1,+1}.  Reference: dataloader/base.py:66-86."""
    xs = np.asarray(xs).astype(F32)
    ys = np.asarray(ys).astype(F32)
    ts = np.asarray(ts).astype(F32)
    ps = np.asarray(ps).astype(F32) * F32(2) - F32(1)
    if ts.shape[0] > 0:
        ts = (ts - ts[0]) / (ts[-1] - ts[0])
    return xs, ys, ts.astype(F32), ps


def create_mask_encoding(xs, ys, ps, sensor_size):
    """Binary event mask [1,H,W].  Reference: dataloader/base.py:159-171."""
    m = events_to_image(xs, ys, np.abs(np.asarray(ps, dtype=F32)), sensor_size, accumulate=False)
    return m[None]


def create_list_encoding(xs, ys, ts, ps):
    """[4,N] rows (ts, ys, xs, ps).  Reference: dataloader/base.py:197-208."""
    return np.stack([np.asarray(a, dtype=F32) for a in (ts, ys, xs, ps)])


def create_polarity_mask(ps):
    """[2,N] rows (pos, neg) in {0,1}.  Reference: dataloader/base.py:210-222."""
    ps = np.asarray(ps, dtype=F32)
    pos = np.where(ps < 0, F32(0), ps)
    neg = np.where(ps > 0, F32(0), ps) * F32(-1)
    return np.stack([pos, neg]).astype(F32)


def collate(samples):
    """Stack per-sample dicts; 3-D stacked items are transposed (2,1) so the
    event list becomes [B,N,4] and the polarity mask [B,N,2].
    Reference: dataloader/base.py:248-265."""
    out = {}
    for key in samples[0]:
        item = np.stack([s[key] for s in samples])
        if item.ndim == 3:
            item = item.transpose(0, 2, 1)
        out[key] = np.ascontiguousarray(item)
    return out


def encode_window(xs, ys, ts, ps, num_bins, sensor_size, round_ts=False):
    """All five tensors the reference loader emits for one window
    (dataloader/h5.py:282-286), from already formatted events."""
    return {
        "event_cnt": events_to_channels(xs, ys, ps, sensor_size),
        "event_mask": create_mask_encoding(xs, ys, ps, sensor_size),
        "event_voxel": events_to_voxel(xs, ys, ts, ps, num_bins, sensor_size, round_ts),
        "event_list": create_list_encoding(xs, ys, ts, ps),
        "event_list_pol_mask": create_polarity_mask(ps),
    }
