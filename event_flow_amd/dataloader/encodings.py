"""Event encodings on the MI355X -- host-side mirror of reference
dataloader/encodings.py (same function names, arguments and error behaviour).

The arithmetic runs in libevflow_hip.so (`evf_events_to_image`,
`evf_encode_events`); tensors must live on the GPU -- there is no CPU path.
`encode_event_list` is the batched entry point the training loop uses instead
of the reference's per-sample CPU loop (dataloader/h5.py:282-286)."""

import torch

from .. import _lib


def _f32(t):
    return t.to(torch.float32).contiguous()


def events_to_image(xs, ys, ps, sensor_size=(180, 240), accumulate=True):
    """Accumulate events into an image.  Reference: dataloader/encodings.py:30-45."""
    _lib.require_gpu(xs, "events_to_image")
    xs, ys, ps = _f32(xs), _f32(ys), _f32(ps)
    H, W = int(sensor_size[0]), int(sensor_size[1])
    img = torch.empty((H, W), dtype=torch.float32, device=xs.device)
    _lib.call("evf_events_to_image", _lib.ptr(xs), _lib.ptr(ys), _lib.ptr(ps), xs.numel(), H, W, 1 if accumulate else 0,
              _lib.ptr(img))
    return img


def _event_rows(xs, ys, ts, ps):
    return torch.stack([_f32(ts), _f32(ys), _f32(xs), _f32(ps)], dim=1).unsqueeze(0).contiguous()  # [1,N,4]


def events_to_voxel(xs, ys, ts, ps, num_bins, sensor_size=(180, 240), round_ts=False):
    """Voxel grid with temporal bilinear interpolation.  Reference:
    dataloader/encodings.py:48-67."""
    assert len(xs) == len(ys) and len(ys) == len(ts) and len(ts) == len(ps)
    _lib.require_gpu(xs, "events_to_voxel")
    H, W = int(sensor_size[0]), int(sensor_size[1])
    ev = _event_rows(xs, ys, ts, ps)
    voxel = torch.empty((1, num_bins, H, W), dtype=torch.float32, device=xs.device)
    _lib.call("evf_encode_events", _lib.ptr(ev), 1, ev.shape[1], H, W, int(num_bins), 1 if round_ts else 0, None, None,
              _lib.ptr(voxel), None)
    return voxel[0]


def events_to_channels(xs, ys, ps, sensor_size=(180, 240)):
    """Two-channel per-polarity event count.  Reference: dataloader/encodings.py:70-85."""
    assert len(xs) == len(ys) and len(ys) == len(ps)
    _lib.require_gpu(xs, "events_to_channels")
    H, W = int(sensor_size[0]), int(sensor_size[1])
    ev = _event_rows(xs, ys, torch.zeros_like(ps), ps)
    cnt = torch.empty((1, 2, H, W), dtype=torch.float32, device=xs.device)
    _lib.call("evf_encode_events", _lib.ptr(ev), 1, ev.shape[1], H, W, 2, 0, _lib.ptr(cnt), None, None, None)
    return cnt[0]


def encode_event_list(event_list, num_bins, sensor_size, round_ts=False, want=("cnt", "mask", "voxel", "pol")):
    """Batched encodings from `event_list` [B,N,4] rows (t,y,x,p) -> dict with
    event_cnt [B,2,H,W], event_mask [B,1,H,W], event_voxel [B,nb,H,W],
    event_list_pol_mask [B,N,2] -- the tensors reference custom_collate
    (dataloader/base.py:248-265) would have produced."""
    _lib.require_gpu(event_list, "encode_event_list")
    ev = _f32(event_list)
    B, N, _ = ev.shape
    H, W = int(sensor_size[0]), int(sensor_size[1])
    dev = ev.device
    out = {}
    cnt = torch.empty((B, 2, H, W), dtype=torch.float32, device=dev) if "cnt" in want else None
    mask = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev) if "mask" in want else None
    voxel = torch.empty((B, num_bins, H, W), dtype=torch.float32, device=dev) if "voxel" in want else None
    pol = torch.empty((B, N, 2), dtype=torch.float32, device=dev) if "pol" in want else None
    _lib.call("evf_encode_events", _lib.ptr(ev), B, N, H, W, int(num_bins), 1 if round_ts else 0, _lib.ptr(cnt),
              _lib.ptr(mask), _lib.ptr(voxel), _lib.ptr(pol))
    if cnt is not None:
        out["event_cnt"] = cnt
    if mask is not None:
        out["event_mask"] = mask
    if voxel is not None:
        out["event_voxel"] = voxel
    if pol is not None:
        out["event_list_pol_mask"] = pol
    out["event_list"] = ev
    return out


def encode_event_lists(event_lists, num_bins, sensor_size, round_ts=False, want=("cnt", "mask", "voxel", "pol")):
    """encode_event_list for all passes of a window at once: the P lists [B,N,4] are binned as ONE batch of P*B
    samples (one zero-fill + one kernel instead of P of each) and handed back as P dicts of views."""
    lists = [_f32(e) for e in event_lists]
    if len(lists) < 2 or any(e.shape != lists[0].shape for e in lists):
        return [encode_event_list(e, num_bins, sensor_size, round_ts, want) for e in lists]
    P = len(lists)
    B, N, _ = lists[0].shape
    full = encode_event_list(torch.stack(lists).view(P * B, N, 4), num_bins, sensor_size, round_ts, want)
    return [{k: v.view(P, B, *v.shape[1:])[i] for k, v in full.items()} for i in range(P)]


def binary_search_array(array, x, left=None, right=None, side="left"):
    """Index of x in a sorted array (host-side loader helper).  Reference:
    dataloader/encodings.py:9-27 -- including its quirk that `side` only
    matters when the first call already has left > right (q17)."""
    lo = 0 if left is None else left
    hi = len(array) - 1 if right is None else right
    if lo > hi:
        return lo if side == "left" else hi
    while lo <= hi:
        mid = lo + (hi - lo) // 2
        if array[mid] == x:
            return mid
        if x < array[mid]:
            hi = mid - 1
        else:
            lo = mid + 1
    return lo  # recursion in the reference drops `side` -> always the "left" answer


def get_hot_event_mask(event_rate, idx, max_px=100, min_obvs=5, max_rate=0.8):
    """Binary mask removing up to max_px pixels whose event rate exceeds
    max_rate (host-side loader helper, mutates event_rate like the reference).
    Reference: dataloader/encodings.py:88-103."""
    mask = torch.ones_like(event_rate)
    if idx > min_obvs:
        flat_rate, flat_mask = event_rate.view(-1), mask.view(-1)
        for _ in range(max_px):
            k = int(torch.argmax(flat_rate))
            if flat_rate[k] > max_rate:
                flat_rate[k] = 0
                flat_mask[k] = 0
            else:
                break
    return mask
