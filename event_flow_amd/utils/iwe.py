"""Image-of-warped-events primitives on the MI355X -- host-side mirror of
reference utils/iwe.py (same names / arguments).  The arithmetic runs in
libevflow_hip.so; there is no CPU path.

`deblur_events` / `compute_pol_iwe` use the fused warp+splat kernel
(`evf_iwe_splat`): coalesced event reads, flow gather, rounding / bilinear
weights and the atomic scatter in one launch, nothing materialised.
`get_interpolation` / `interpolate` / `purge_unfeasible` keep the reference's
materialising API for callers that want the index / weight tensors.
"""

import torch

from .. import _lib


def _f32c(t):
    return t.to(torch.float32).contiguous()


def _strided_mask(m, B, N):
    """[B,N,1] (possibly a slice of a [B,N,2] mask) -> (tensor, element stride)."""
    if m is None:
        return None, 1
    if m.dtype != torch.float32:
        m = m.float()
    if m.numel() != B * N:
        raise _lib.EvflowError(f"mask has {m.numel()} elements, expected {B}x{N}")
    if m.dim() >= 2 and m.shape[0] == B and m.shape[1] == N:
        sb, sn = m.stride(0), m.stride(1)
        if sn > 0 and sb == N * sn:  # e.g. pol_mask[:, :, 0:1] of a contiguous [B,N,2]
            return m, sn
    return m.reshape(B, N).contiguous(), 1


def purge_unfeasible(x, res):
    """Zero the indices / weights of out-of-image locations.  x [B,M,2] (y,x).
    Reference: utils/iwe.py:4-17.  (Tiny elementwise helper kept for API
    parity; the fused kernels do this in-register.)"""
    ok = ((x[:, :, 0:1] >= 0) & (x[:, :, 0:1] < res[0]) & (x[:, :, 1:2] >= 0) & (x[:, :, 1:2] < res[1])).to(x.dtype)
    return x * ok, ok


def get_interpolation(events, flow, tref, res, flow_scaling, round_idx=False):
    """events [B,N,4] (ts,y,x,p), flow [B,N,2] (y,x) -> idx [B,M,1], weights [B,M,1].
    Reference: utils/iwe.py:20-74."""
    _lib.require_gpu(events, "get_interpolation")
    ev, fl = _f32c(events), _f32c(flow)
    B, N, _ = ev.shape
    M = N if round_idx else 4 * N
    idx = torch.empty((B, M, 1), dtype=torch.float32, device=ev.device)
    wgt = torch.empty((B, M, 1), dtype=torch.float32, device=ev.device)
    _lib.call("evf_get_interpolation", _lib.ptr(ev), _lib.ptr(fl), B, N, int(res[0]), int(res[1]), float(flow_scaling),
              float(tref), 1 if round_idx else 0, _lib.ptr(idx), _lib.ptr(wgt))
    return idx, wgt


def interpolate(idx, weights, res, polarity_mask=None):
    """Scatter-add into a [B,1,H,W] image.  Reference: utils/iwe.py:77-92."""
    _lib.require_gpu(weights, "interpolate")
    B, M = idx.shape[0], idx.shape[1]
    idx_f = _f32c(idx.to(torch.float32))
    w = _f32c(weights)
    pm, ps = _strided_mask(polarity_mask, B, M)
    out = torch.empty((B, 1, int(res[0]), int(res[1])), dtype=torch.float32, device=w.device)
    _lib.call("evf_interpolate", _lib.ptr(idx_f), _lib.ptr(w), pm.data_ptr() if pm is not None else None, ps, B, M,
              int(res[0]), int(res[1]), _lib.ptr(out))
    return out


def iwe_splat(flow_maps, event_list, res, flow_scaling, tref, *, round_idx, w0=None, w1=None, nch=1, zero_flow=False,
              with_ts=False, ts_from_tref=None, map_of_event=None, ts_shift=None):
    """Thin wrapper over evf_iwe_splat (see include/evflow.h).  flow_maps is
    [B,2,H,W] or [n_maps,B,2,H,W]."""
    _lib.require_gpu(event_list, "iwe_splat")
    ev = _f32c(event_list)
    fl = _f32c(flow_maps)
    B, M, _ = ev.shape
    H, W = int(res[0]), int(res[1])
    a0, s0 = _strided_mask(w0, B, M)
    a1, s1 = _strided_mask(w1, B, M)
    if a0 is not None and a1 is not None and s0 != s1:
        a0, s0 = a0.reshape(B, M).contiguous(), 1
        a1, s1 = a1.reshape(B, M).contiguous(), 1
    mode = (1 if round_idx else 0) | (2 if zero_flow else 0) | (4 if with_ts else 0) | (8 if ts_from_tref is not None else 0)
    out = torch.empty((B, nch, H, W), dtype=torch.float32, device=ev.device)
    _lib.call(
        "evf_iwe_splat", _lib.ptr(fl), _lib.ptr(ev), _lib.ptr(map_of_event), _lib.ptr(ts_shift),
        a0.data_ptr() if a0 is not None else None, a1.data_ptr() if a1 is not None else None, s0 if a0 is not None else s1,
        B, M, H, W, float(flow_scaling), float(tref), float(ts_from_tref if ts_from_tref is not None else 0.0), mode, nch,
        _lib.ptr(out),
    )
    return out


def deblur_events(flow, event_list, res, flow_scaling=128, round_idx=True, polarity_mask=None):
    """IWE [B,1,H,W] of the events warped to t_ref = 1.  Reference: utils/iwe.py:95-129."""
    return iwe_splat(flow, event_list, res, flow_scaling, 1.0, round_idx=round_idx, w0=polarity_mask, nch=1)


def compute_pol_iwe(flow, event_list, res, pos_mask, neg_mask, flow_scaling=128, round_idx=True):
    """Per-polarity IWE [B,2,H,W].  Reference: utils/iwe.py:132-153.  With
    round_idx=True the result is an integer histogram, bit-exact with the
    reference."""
    return iwe_splat(flow, event_list, res, flow_scaling, 1.0, round_idx=round_idx, w0=pos_mask, w1=neg_mask, nch=2)
