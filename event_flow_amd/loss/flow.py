"""Contrast-maximisation loss and validation metrics on the MI355X -- host-side
mirror of reference loss/flow.py (same class names, constructor arguments,
methods and properties, so train_flow.py / eval_flow.py call it unchanged).

Differences in mechanism, not in results:
  * `event_flow_association` only records references (no per-pass gather /
    O(P^2) re-concatenation, loss/flow.py:65-116): the per-event flow lookup
    happens inside the fused warp kernels from the stacked flow maps.
  * `forward()` is one autograd.Function: 4 launches forward
    (`evf_cm_loss_fwd`), 3 backward (`evf_cm_loss_bwd`) for all scales.
  * the caller's `event_list` is never mutated (reference quirk q1,
    loss/flow.py:90): the pass offset is added in-register.
"""

import os

import torch

from .. import _lib
from ..utils.iwe import iwe_splat


def spatial_variance(x):
    """Unbiased variance over the spatial dims of [B,C,H,W] -> [B,C,1,1].
    Reference: loss/flow.py:13-23."""
    _lib.require_gpu(x, "spatial_variance")
    x = x.to(torch.float32).contiguous()
    B, C = x.shape[0], x.shape[1]
    out = torch.empty(B * C, dtype=torch.float32, device=x.device)
    _lib.call("evf_image_variance", _lib.ptr(x), B * C, x[0, 0].numel(), _lib.ptr(out))
    return out.view(B, C, 1, 1)


class _WindowRecord:
    """Events / masks / flow maps of the passes accumulated since reset()."""

    def __init__(self):
        self.passes = 0
        self.events = []  # per pass [B,N,4] as handed in (NOT shifted, NOT mutated)
        self.pol = []  # per pass [B,N,2]
        self.masks = []  # per pass [B,1,H,W]
        self.flows = []  # per pass: list over scales of [B,2,H,W]
        self.overwritten = False
        self._cache = None

    def add(self, flow_list, event_list, pol_mask, event_mask):
        self.events.append(event_list)
        self.pol.append(pol_mask)
        self.masks.append(event_mask)
        self.flows.append(list(flow_list))
        self.passes += 1
        self._cache = None

    @property
    def num_events(self):
        return sum(e.shape[1] for e in self.events)

    def packed(self):
        """(ev [B,M,4], pol [B,M,2], ev_pass int32 [M]) concatenated over passes -- in place when the passes are
        the slices of one window buffer (dataloader.encodings.encode_window), else by torch.cat."""
        if self._cache is None:
            ev = _in_place(self.events, 1)
            if ev is None:
                ev = torch.cat([e.to(torch.float32) for e in self.events], 1).contiguous()
            pol = _in_place(self.pol, 1)
            if pol is None:
                pol = torch.cat([p.to(torch.float32) for p in self.pol], 1).contiguous()
            ev_pass = _pass_index(tuple(e.shape[1] for e in self.events), ev.device)
            self._cache = (ev, pol, ev_pass)
        return self._cache

    def mask_stack(self):
        m = _in_place(self.masks, 1)
        if m is not None:
            return m
        return torch.cat([m.to(torch.float32) for m in self.masks], 1).contiguous()  # [B,P,H,W]


def _in_place(tensors, dim):
    """The concatenation of `tensors` along dim 1 WITHOUT a copy, if they are consecutive slices of one buffer."""
    from ..dataloader.encodings import window_base

    if len(tensors) < 2:
        return None
    base = window_base(tensors)  # [B, P, n, ...]
    if base is None:
        return None
    return base.reshape((base.shape[0], base.shape[1] * base.shape[2]) + tuple(base.shape[3:]))


def _stacked_in_place(tensors):
    """torch.stack(tensors) WITHOUT a copy, if they are consecutive contiguous slices of one buffer."""
    t0 = tensors[0]
    if t0.dtype != torch.float32 or not t0.is_contiguous():
        return None
    n = t0.numel()
    store = t0.untyped_storage().data_ptr()  # (neighbours in memory that are separate allocations do not count)
    for k, t in enumerate(tensors):
        if (t.dtype != t0.dtype or t.device != t0.device or t.shape != t0.shape or not t.is_contiguous()
                or t.untyped_storage().data_ptr() != store or t.data_ptr() != t0.data_ptr() + 4 * k * n):
            return None
    return torch.as_strided(t0, (len(tensors),) + tuple(t0.shape), (n,) + tuple(t0.stride()))


def _mask_union(masks):
    """[B,P,H,W] binary masks of the passes -> [B,1,H,W] mask of the window, min(sum, 1)."""
    masks = masks.contiguous()
    B, P, H, W = masks.shape
    out = torch.empty((B, 1, H, W), dtype=torch.float32, device=masks.device)
    _lib.call("evf_mask_union", _lib.ptr(masks), B, P, H, W, _lib.ptr(out))
    return out


_PASS_INDEX = {}


def _pass_index(lengths, device):
    """int32 [sum(lengths)]: pass number of every event of the concatenated window (built once per shape)."""
    key = (lengths, str(device))
    if key not in _PASS_INDEX:
        import numpy as np

        idx = np.repeat(np.arange(len(lengths), dtype=np.int32), lengths)
        _PASS_INDEX[key] = torch.from_numpy(idx).to(device)
    return _PASS_INDEX[key]


# event-scale pairs from which evf_cm_loss_fwd gets the workspace of its LDS-striped splat (MI355X: 59 -> 26 us at
# 8 x 15 k events, 1164 -> 210 us at 4 scales x 8 x 50 k; below, the pre-warp pass does not pay)
CM_LDS_MIN_EVENTS = int(os.environ.get("EVF_CM_LDS_MIN_EVENTS", 32768))


class _CMLoss(torch.autograd.Function):
    """EventWarping.forward (loss/flow.py:176-301) as one fused op.
    Inputs: the flow maps, flat over (scale, pass); output: 0-d loss."""

    @staticmethod
    def forward(ctx, meta, *flows):
        S, Pm, P = meta["S"], meta["Pm"], meta["P"]
        ev, pol, ev_pass, mask = meta["ev"], meta["pol"], meta["ev_pass"], meta["mask"]
        B, M = ev.shape[0], ev.shape[1]
        H, W = meta["res"]
        dev = ev.device
        fl = _stacked_in_place(flows)  # (the engine writes a window's flow maps into one buffer)
        if fl is None:
            fl = torch.stack([f.to(torch.float32) for f in flows])
        fl = fl.view(S, Pm, B, 2, H, W).contiguous()
        images = torch.empty((S, B, 8, H, W), dtype=torch.float32, device=dev)
        stats = torch.empty((S, B, 2, 2), dtype=torch.float32, device=dev)
        nblk = _lib.load().evf_cm_smooth_blocks(B, Pm, H, W)
        part = torch.empty((S, nblk), dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        # enough event-scale pairs to amortise the pre-warp pass: LDS-striped accumulation instead of global atomics
        nws = _lib.load().evf_cm_loss_ws(S, B, M, H, W) if S * B * M >= CM_LDS_MIN_EVENTS else 0
        ws = torch.empty(nws, dtype=torch.float32, device=dev) if nws > 0 else None
        _lib.call(
            "evf_cm_loss_fwd", _lib.ptr(fl), _lib.ptr(ev), _lib.ptr(pol), _lib.ptr(ev_pass), _lib.ptr(mask), S, P, B, M, H,
            W, float(meta["flow_scaling"]), float(meta["weight"]), meta["flags"], _lib.ptr(images), _lib.ptr(stats),
            _lib.ptr(part), _lib.ptr(loss), _lib.ptr(ws),
        )
        ctx.meta = meta
        ctx.save_for_backward(fl, images, stats)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        meta = ctx.meta
        fl, images, stats = ctx.saved_tensors
        S, Pm, P = meta["S"], meta["Pm"], meta["P"]
        ev, pol, ev_pass, mask = meta["ev"], meta["pol"], meta["ev_pass"], meta["mask"]
        B, M = ev.shape[0], ev.shape[1]
        H, W = meta["res"]
        g = g.to(torch.float32).reshape(1).contiguous()
        gim = torch.empty_like(images)
        dflow = torch.empty_like(fl)
        _lib.call(
            "evf_cm_loss_bwd", _lib.ptr(fl), _lib.ptr(ev), _lib.ptr(pol), _lib.ptr(ev_pass), _lib.ptr(mask), S, P, B, M, H,
            W, float(meta["flow_scaling"]), float(meta["weight"]), meta["flags"], _lib.ptr(images), _lib.ptr(stats),
            _lib.ptr(g), _lib.ptr(gim), _lib.ptr(dflow),
        )
        d = dflow.view(S * Pm, B, 2, H, W)
        return (None,) + tuple(d[i] for i in range(S * Pm))


class EventWarping(torch.nn.Module):
    """Contrast maximisation loss (Zhu et al., CVPR'19, Section 3.2): per-pixel,
    per-polarity image of averaged timestamps of the motion-compensated events,
    minimised forward and backward in time, plus Charbonnier flow smoothing.
    Reference: loss/flow.py:26-301."""

    def __init__(self, config, device, flow_scaling=None, loss_scaling=True):
        super().__init__()
        self.loss_scaling = loss_scaling
        self.res = config["loader"]["resolution"]
        self.flow_scaling = flow_scaling if flow_scaling is not None else max(config["loader"]["resolution"])
        self.weight = config["loss"]["flow_regul_weight"]
        self.smoothing_mask = False if "mask_output" not in config["model"].keys() else config["model"]["mask_output"]
        self.overwrite_intermediate = (
            False if "overwrite_intermediate" not in config["loss"].keys() else config["loss"]["overwrite_intermediate"]
        )
        self.device = device
        self._win = _WindowRecord()
        self._final_flow = None

    @property
    def _passes(self):
        return self._win.passes

    def event_flow_association(self, flow_list, event_list, pol_mask, event_mask):
        """flow_list: list of [B,2,H,W] (x,y) maps; event_list [B,N,4]
        (ts,y,x,p); pol_mask [B,N,2]; event_mask [B,1,H,W].
        Reference: loss/flow.py:56-119."""
        _lib.require_gpu(event_list, "EventWarping.event_flow_association")
        self._win.add(flow_list, event_list, pol_mask, event_mask)

    def overwrite_intermediate_flow(self, flow_list):
        """Use the final flow maps for every event of the window.
        Reference: loss/flow.py:121-150."""
        self._final_flow = list(flow_list)
        self._win.overwritten = True

    def reset(self):
        self._win = _WindowRecord()
        self._final_flow = None

    @property
    def num_events(self):
        return self._win.num_events

    @property
    def event_mask(self):
        """Mask of the window (overwrite) or of the last pass.  loss/flow.py:168-174."""
        if self._win.overwritten:
            return _mask_union(self._win.mask_stack())  # loss/flow.py:149-150
        if self.overwrite_intermediate:
            return self._win.mask_stack()
        return self._win.masks[-1].to(torch.float32).contiguous()

    def forward(self):
        win = self._win
        P = win.passes
        ev, pol, ev_pass = win.packed()
        overwritten = win.overwritten
        if overwritten:
            flows_by_pass = [self._final_flow]
            mask = _mask_union(win.mask_stack())
        else:
            flows_by_pass = win.flows
            mask = win.mask_stack()
        S = len(flows_by_pass[0])
        Pm = len(flows_by_pass)
        flat = [flows_by_pass[p][s] for s in range(S) for p in range(Pm)]
        # NOTE: like the reference, the "overwrite" behaviour of the smoothness
        # term (no temporal component) follows the *config* flag (loss/flow.py:189,285,290)
        ow_cfg = bool(self.overwrite_intermediate)
        if ow_cfg != overwritten:
            raise _lib.EvflowError(
                "overwrite_intermediate config and overwrite_intermediate_flow() call disagree; "
                "the reference's loss is only defined when they match (train_flow.py:144-145)"
            )
        flags = (1 if self.smoothing_mask else 0) | (2 if overwritten else 0) | (4 if self.loss_scaling else 0)
        meta = dict(S=S, Pm=Pm, P=P, ev=ev, pol=pol, ev_pass=ev_pass, mask=mask if self.smoothing_mask else None,
                    res=(int(self.res[0]), int(self.res[1])), flow_scaling=self.flow_scaling, weight=self.weight, flags=flags)
        return _CMLoss.apply(meta, *flat)


class BaseValidationLoss(torch.nn.Module):
    """Base class for the validation metrics.  Reference: loss/flow.py:304-465."""

    def __init__(self, config, device, flow_scaling=128):
        super().__init__()
        self.res = config["loader"]["resolution"]
        self.flow_scaling = flow_scaling  # should be specified by the user
        self.overwrite_intermediate = (
            False if "overwrite_intermediate" not in config["loss"].keys() else config["loss"]["overwrite_intermediate"]
        )
        self.device = device
        self._win = _WindowRecord()
        self._final_flow = None
        self._gtflow = None
        self._dt_input = None
        self._dt_gt = None

    @property
    def _passes(self):
        return self._win.passes

    @property
    def num_events(self):
        return self._win.num_events

    def event_flow_association(self, flow_list, inputs):
        """Only the highest-resolution flow (flow_list[-1]) is used.
        Reference: loss/flow.py:332-396."""
        dev = self.device
        self._win.add(
            [flow_list[-1].detach()], inputs["event_list"].to(dev), inputs["event_list_pol_mask"].to(dev),
            inputs["event_mask"].to(dev),
        )
        self._gtflow = inputs["gtflow"].to(dev) if "gtflow" in inputs.keys() else None
        self._dt_input = inputs["dt_input"]
        self._dt_gt = inputs["dt_gt"]

    def overwrite_intermediate_flow(self, flow_list):
        """Reference: loss/flow.py:398-422."""
        self._final_flow = flow_list[-1].detach()
        self._win.overwritten = True

    def reset(self):
        self._win = _WindowRecord()
        self._final_flow = None

    # -- helpers -----------------------------------------------------------
    def _flow_maps(self):
        """-> (maps [n,B,2,H,W], map_of_event int32 [M] or None)."""
        ev, pol, ev_pass = self._win.packed()
        if self._win.overwritten:
            return self._final_flow.to(torch.float32).contiguous().unsqueeze(0), None
        maps = torch.stack([f[0].to(torch.float32) for f in self._win.flows]).contiguous()
        return maps, ev_pass

    def _event_mask_stack(self):
        if self._win.overwritten:
            return _mask_union(self._win.mask_stack())
        return self._win.mask_stack()

    def _splat(self, *, round_idx, nch, zero_flow=False, with_ts=False, pol=True):
        ev, polm, ev_pass = self._win.packed()
        maps, moe = self._flow_maps()
        w0 = polm[:, :, 0:1] if pol else None
        w1 = polm[:, :, 1:2] if pol else None
        return iwe_splat(maps, ev, self.res, self.flow_scaling, float(self._win.passes), round_idx=round_idx, w0=w0, w1=w1,
                         nch=nch, zero_flow=zero_flow, with_ts=with_ts, map_of_event=moe, ts_shift=ev_pass)

    def compute_window_events(self):
        """Per-polarity event count image of the window [B,2,H,W].  loss/flow.py:432-441."""
        return self._splat(round_idx=True, nch=2, zero_flow=True)

    def compute_masked_window_flow(self):
        """loss/flow.py:443-452."""
        mask = self._event_mask_stack()
        if self.overwrite_intermediate:
            ff = self._final_flow.to(torch.float32).contiguous()
            B, _, H, W = ff.shape
            out = torch.empty_like(ff)
            _lib.call("evf_masked_flow_mean", _lib.ptr(ff), _lib.ptr(mask.contiguous()), B, 1, H, W, _lib.ptr(out))
            return out  # flow * mask for a binary mask
        maps = torch.stack([f[0].to(torch.float32) for f in self._win.flows], 1).contiguous()  # [B,P,2,H,W]
        B, P, _, H, W = maps.shape
        out = torch.empty((B, 2, H, W), dtype=torch.float32, device=maps.device)
        _lib.call("evf_masked_flow_mean", _lib.ptr(maps), _lib.ptr(mask.contiguous()), B, P, H, W, _lib.ptr(out))
        return out

    def compute_window_iwe(self, round_idx=True):
        """Per-polarity IWE of the window at t_ref = P [B,2,H,W].  loss/flow.py:454-465
        (without the reference's permanent x4 growth of the polarity list, q12)."""
        return self._splat(round_idx=round_idx, nch=2)


class FWL(BaseValidationLoss):
    """Flow Warp Loss (Stoffregen, Scheerlinck et al., ECCV'20): variance of the
    IWE over the variance of the image of un-warped events; larger is better.
    Reference: loss/flow.py:468-500."""

    def __init__(self, config, device, flow_scaling=128):
        super().__init__(config, device, flow_scaling)

    def forward(self):
        fw_iwe = self._splat(round_idx=True, nch=1, pol=False)
        ie = self._splat(round_idx=True, nch=1, pol=False, zero_flow=True)
        fwl = spatial_variance(fw_iwe) / spatial_variance(ie)
        return fwl.view(fw_iwe.shape[0])


class RSAT(BaseValidationLoss):
    """Ratio of the squared averaged timestamps of the IWE and of the image of
    un-warped events; lower is better.  Reference: loss/flow.py:503-579."""

    def __init__(self, config, device, flow_scaling=128):
        super().__init__(config, device, flow_scaling)

    def _ts_sum(self, zero_flow):
        im = self._splat(round_idx=True, nch=4, with_ts=True, zero_flow=zero_flow)
        B = im.shape[0]
        out = torch.empty(B, dtype=torch.float32, device=im.device)
        _lib.call("evf_avg_ts_ratio", _lib.ptr(im), B, im[0, 0].numel(), float(self._win.passes), _lib.ptr(out))
        return out

    def forward(self):
        return self._ts_sum(False) / self._ts_sum(True)


class AEE(BaseValidationLoss):
    """Average endpoint error and outlier percentage.  Reference: loss/flow.py:582-628.
    dt_gt/dt_input is applied per sample (identical to the reference for its
    only supported case B = 1, quirk q11); the outlier count is summed over
    the whole batch like loss/flow.py:626."""

    def __init__(self, config, device, flow_scaling=128):
        super().__init__(config, device, flow_scaling)

    @property
    def num_events(self):
        return float("inf")

    def forward(self):
        flow = (self._final_flow if self._win.overwritten else self._win.flows[-1][0]).to(torch.float32).contiguous()
        B, _, H, W = flow.shape
        dev = flow.device
        ratio = (torch.as_tensor(self._dt_gt, dtype=torch.float32) / torch.as_tensor(self._dt_input, dtype=torch.float32))
        ratio = ratio.reshape(-1).to(dev).expand(B).contiguous()
        gt = self._gtflow.to(torch.float32).contiguous()
        mask = self._event_mask_stack()[:, -1].contiguous()
        out = torch.empty((B, 3), dtype=torch.float32, device=dev)
        _lib.call("evf_aee", _lib.ptr(flow), _lib.ptr(gt), _lib.ptr(mask), _lib.ptr(ratio), B, H, W,
                  float(self.flow_scaling), _lib.ptr(out))
        aee = out[:, 0] / (out[:, 1] + 1e-9)
        percent = out[:, 2].sum() / (out[:, 1] + 1e-9)
        return aee, percent
