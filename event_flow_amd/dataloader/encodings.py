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


def window_base(tensors):
    """If the P tensors [B, ...] are the slices base[:, p] of ONE contiguous fp32 tensor base [B, P, ...] (in order,
    nothing in between), return that base, else None.  Lets the window's consumers read [B, P * ...] in place instead
    of torch.cat-ing the passes back together."""
    t0 = tensors[0]
    P = len(tensors)
    if t0.dtype != torch.float32 or t0.dim() < 2:
        return None
    inner = 1
    for d in t0.shape[1:]:
        inner *= d
    want = (P * inner,) + tuple(torch.empty(t0.shape[1:]).stride())  # a row of P contiguous slices per batch element
    base_ptr, store = t0.data_ptr(), t0.untyped_storage().data_ptr()
    for p, t in enumerate(tensors):
        if (t.dtype != t0.dtype or t.device != t0.device or t.shape != t0.shape or tuple(t.stride()) != want
                or t.untyped_storage().data_ptr() != store or t.data_ptr() != base_ptr + 4 * p * inner):
            return None
    if t0.shape[0] > 1 and t0.stride(0) != P * inner:
        return None
    return torch.as_strided(t0, (t0.shape[0], P) + tuple(t0.shape[1:]), (P * inner, inner) + want[1:])


def encode_window(ev, num_bins, sensor_size, round_ts=False, want=("cnt", "mask", "voxel", "pol")):
    """All P passes of a BPTT window binned by ONE launch (+ one zero-fill).  ev [B,P,N,4] float32, batch-major.
    Returns P dicts like encode_event_list's whose tensors are views of four window buffers: the network inputs
    (event_cnt / event_voxel) are one contiguous tensor per pass, the loss inputs (event_list, event_list_pol_mask,
    event_mask) are the slices [:, p] of batch-major buffers -- the window's [B,P*N,4] event list, [B,P*N,2] polarity
    mask and [B,P,H,W] mask stack exist without a torch.cat (loss/flow.py's record reads them in place)."""
    _lib.require_gpu(ev, "encode_window")
    if ev.dtype != torch.float32 or ev.dim() != 4 or ev.shape[3] != 4 or not ev.is_contiguous():
        raise _lib.EvflowError("encode_window needs a contiguous float32 [B,P,N,4] tensor")
    B, P, N, _ = ev.shape
    H, W = int(sensor_size[0]), int(sensor_size[1])
    dev = ev.device
    HW = H * W
    n_cnt = P * B * 2 * HW if "cnt" in want else 0
    n_vox = P * B * num_bins * HW if "voxel" in want else 0
    n_mask = B * P * HW if "mask" in want else 0
    dense = torch.empty(max(n_cnt + n_vox + n_mask, 1), dtype=torch.float32, device=dev)
    pol = torch.empty((B, P, N, 2), dtype=torch.float32, device=dev) if "pol" in want else None
    flags = (1 if n_cnt else 0) | (2 if n_vox else 0) | (4 if n_mask else 0)
    _lib.call("evf_encode_window", _lib.ptr(ev), B, P, N, H, W, int(num_bins), 1 if round_ts else 0, flags,
              _lib.ptr(dense) if flags else None, _lib.ptr(pol))
    cnt = dense[:n_cnt].view(P, B, 2, H, W) if n_cnt else None
    vox = dense[n_cnt:n_cnt + n_vox].view(P, B, num_bins, H, W) if n_vox else None
    mask = dense[n_cnt + n_vox:n_cnt + n_vox + n_mask].view(B, P, H, W) if n_mask else None
    out = []
    for p in range(P):
        d = {"event_list": ev[:, p]}
        if cnt is not None:
            d["event_cnt"] = cnt[p]
        if vox is not None:
            d["event_voxel"] = vox[p]
        if mask is not None:
            d["event_mask"] = mask[:, p:p + 1]
        if pol is not None:
            d["event_list_pol_mask"] = pol[:, p]
        out.append(d)
    return out


def encode_event_lists(event_lists, num_bins, sensor_size, round_ts=False, want=("cnt", "mask", "voxel", "pol")):
    """encode_event_list for all passes of a window at once: the P lists [B,N,4] are binned as ONE batch of P*B
    samples (one zero-fill + one kernel instead of P of each) and handed back as P dicts of views.  Lists that are
    the slices [:, p] of one [B,P,N,4] buffer go through encode_window (no torch.stack, loss inputs in place)."""
    if len(event_lists) >= 2:
        base = window_base(list(event_lists))
        if base is not None and base.dim() == 4 and base.shape[3] == 4:
            return encode_window(base, num_bins, sensor_size, round_ts, want)
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
