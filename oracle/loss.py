"""Oracle (PyTorch-CPU fp32) for the contrast-maximisation loss and the
FWL / RSAT / AEE metrics  --  test infrastructure only.

Restates reference loss/flow.py:26-301 (EventWarping), :304-465
(BaseValidationLoss), :468-500 (FWL), :503-579 (RSAT), :582-628 (AEE) as
closed-form functions over an explicit window record.  Gradients come from
torch autograd over the same op sequence as the reference (so the
`max(0, 1-|d|)` tie sub-gradients and the `#nonzero px` denominator path of
SURVEY.md section 9 q7/q8 are inherited, not re-derived).
"""

import torch


# ----------------------------------------------------------------------------
# window record
# ----------------------------------------------------------------------------
def gather_event_flow(flow, event_list, res):
    """(fy, fx) per event from a [B,2,H,W] (x,y) flow map.  loss/flow.py:65-84."""
    lin = (event_list[:, :, 1] * res[1] + event_list[:, :, 2]).long()
    f = flow.reshape(flow.shape[0], 2, -1)
    fy = torch.gather(f[:, 1, :], 1, lin)
    fx = torch.gather(f[:, 0, :], 1, lin)
    return torch.stack([fy, fx], dim=2)


class Window:
    """Accumulates P passes of (flow maps, events, polarity mask, event mask).
    Mirrors the bookkeeping of EventWarping.event_flow_association
    (loss/flow.py:56-119) / BaseValidationLoss.event_flow_association
    (:332-396) without mutating the caller's tensors."""

    def __init__(self, res):
        self.res = tuple(res)
        self.passes = 0
        self.events = []  # per pass [B,N,4], ts already shifted by pass index
        self.pol = []  # per pass [B,N,2]
        self.mask = []  # per pass [B,1,H,W]
        self.flow_maps = []  # per pass: list over scales of [B,2,H,W]
        self.ev_flow = None  # list over scales of list over passes [B,N,2]

    def add(self, flow_list, event_list, pol_mask, event_mask):
        ev = event_list.clone()
        ev[:, :, 0:1] = ev[:, :, 0:1] + self.passes  # loss/flow.py:90
        if self.ev_flow is None:
            self.ev_flow = [[] for _ in flow_list]
        for i, flow in enumerate(flow_list):
            self.ev_flow[i].append(gather_event_flow(flow, event_list, self.res))
        self.events.append(ev)
        self.pol.append(pol_mask)
        self.mask.append(event_mask)
        self.flow_maps.append(list(flow_list))
        self.passes += 1

    def overwrite(self, flow_list):
        """loss/flow.py:121-150: re-gather every event with the final flow,
        collapse the masks to min(sum, 1)."""
        allev = torch.cat(self.events, 1)
        self.ev_flow = [[gather_event_flow(f, allev, self.res)] for f in flow_list]
        self.flow_maps = [list(flow_list)]
        m = torch.cat(self.mask, 1).sum(1, keepdim=True)
        self.mask = [torch.where(m > 1, torch.ones_like(m), m)]
        self._overwritten = True

    @property
    def num_events(self):
        return sum(e.shape[1] for e in self.events)


# ----------------------------------------------------------------------------
# IWE building blocks (torch, differentiable)
# ----------------------------------------------------------------------------
def _warp(events, ev_flow, tref, S):
    return events[:, :, 1:3] + (tref - events[:, :, 0:1]) * ev_flow * S  # utils/iwe.py:37


def _corners(warped, res, round_idx):
    """-> lin idx [B,M] (long), weight [B,M] ; M = N or 4N.  utils/iwe.py:39-72."""
    if round_idx:
        idx = torch.round(warped)
        wgt = torch.ones_like(idx)
    else:
        y, x = warped[:, :, 0:1], warped[:, :, 1:2]
        ty, by = torch.floor(y), torch.floor(y + 1)
        lx, rx = torch.floor(x), torch.floor(x + 1)
        idx = torch.cat(
            [torch.cat([ty, lx], 2), torch.cat([ty, rx], 2), torch.cat([by, lx], 2), torch.cat([by, rx], 2)], 1
        )
        w4 = torch.cat([warped] * 4, 1)
        wgt = torch.max(torch.zeros_like(w4), 1 - torch.abs(w4 - idx))
    ok = ((idx[:, :, 0] >= 0) & (idx[:, :, 0] < res[0]) & (idx[:, :, 1] >= 0) & (idx[:, :, 1] < res[1])).to(
        warped.dtype
    )
    wgt = torch.prod(wgt, dim=-1) * ok
    idx = idx * ok[:, :, None]
    lin = (idx[:, :, 0] * res[1] + idx[:, :, 1]).long()
    return lin, wgt


def _splat(lin, wgt, res):
    img = torch.zeros(lin.shape[0], res[0] * res[1], dtype=wgt.dtype)
    return img.scatter_add(1, lin, wgt)  # utils/iwe.py:87-91


def _avg_ts_term(lin, wgt, ts, pol, res, P, loss_scaling=True):
    """One warping direction: sum_px (A_pos^2 + A_neg^2) / #nonzero px, [B].
    loss/flow.py:201-226 (forward) / :234-259 (backward)."""
    ipos = _splat(lin, wgt * pol[:, :, 0], res)
    ineg = _splat(lin, wgt * pol[:, :, 1], res)
    tpos = _splat(lin, wgt * ts * pol[:, :, 0], res)
    tneg = _splat(lin, wgt * ts * pol[:, :, 1], res)
    apos = tpos / (ipos + 1e-9) / P
    aneg = tneg / (ineg + 1e-9) / P
    val = (apos**2).sum(1) + (aneg**2).sum(1)
    if loss_scaling:
        s = ipos + ineg
        nz = torch.where(s > 0, torch.ones_like(s), s)  # masked assign :222-225 keeps grad where s == 0
        val = val / nz.sum(1)
    return val


def _charbonnier(u):
    return torch.sqrt(u**2 + 1e-6)


def smoothness(fx, fy, mask, use_mask, overwrite):
    """fx, fy [B,P,H,W]; mask [B,Pm,H,W].  loss/flow.py:183-190,261-294.
    Charbonnier is applied to (dfx + dfy): components are summed BEFORE the
    square (quirk q5)."""
    terms = [
        ((slice(None), slice(None, -1)), (slice(None), slice(1, None))),  # dx
        ((slice(None, -1), slice(None)), (slice(1, None), slice(None))),  # dy
        ((slice(None, -1), slice(None, -1)), (slice(1, None), slice(1, None))),  # diag down-right
        ((slice(1, None), slice(None, -1)), (slice(None, -1), slice(1, None))),  # diag up-right
    ]
    total = 0
    for a, b in terms:
        ia = (slice(None), slice(None)) + a
        ib = (slice(None), slice(None)) + b
        c = _charbonnier((fx[ia] - fx[ib]) + (fy[ia] - fy[ib]))
        if use_mask:
            c = mask[ia] * mask[ib] * c
        total = total + c.sum()
    comps = 4
    if not overwrite:
        c = _charbonnier((fx[:, :-1] - fx[:, 1:]) + (fy[:, :-1] - fy[:, 1:]))
        if use_mask:
            c = mask[:, :-1] * mask[:, 1:] * c
        total = total + c.sum()
        comps = 5
    return total / comps / fx.shape[1]


def event_warping_loss(win, flow_scaling, weight, smoothing_mask=True, overwrite=False, loss_scaling=True):
    """EventWarping.forward, loss/flow.py:176-301, on a Window record."""
    P = win.passes
    res = win.res
    events = torch.cat(win.events, 1)
    pol = torch.cat(win.pol, 1)
    mask = torch.cat(win.mask, 1)
    ts = events[:, :, 0]
    nscales = len(win.ev_flow)
    loss = 0
    for i in range(nscales):
        ef = torch.cat(win.ev_flow[i], 1)
        # forward warp, t_ref = P, timestamp image of t
        lin, wgt = _corners(_warp(events, ef, P, flow_scaling), res, False)
        rep = lin.shape[1] // events.shape[1]
        fw = _avg_ts_term(lin, wgt, ts.repeat(1, rep), pol.repeat(1, rep, 1), res, P, loss_scaling).sum()
        # backward warp, t_ref = 0, timestamp image of (P - t)
        lin, wgt = _corners(_warp(events, ef, 0, flow_scaling), res, False)
        bw = _avg_ts_term(lin, wgt, (P - ts).repeat(1, rep), pol.repeat(1, rep, 1), res, P, loss_scaling).sum()
        fx = torch.cat([fm[i][:, 0:1] for fm in win.flow_maps], 1)
        fy = torch.cat([fm[i][:, 1:2] for fm in win.flow_maps], 1)
        sm = smoothness(fx, fy, mask, smoothing_mask, overwrite)
        loss = loss + fw + bw + weight * sm
    return loss / nscales


# ----------------------------------------------------------------------------
# validation metrics (only the last flow scale is used: loss/flow.py:350)
# ----------------------------------------------------------------------------
def _round_images(events, ef, pol, tref, res, S):
    lin, wgt = _corners(_warp(events, ef, tref, S), res, True)
    return lin, wgt


def fwl(win, flow_scaling):
    """FWL.forward, loss/flow.py:481-500: var(IWE(flow)) / var(IWE(0)), [B]."""
    P, res = win.passes, win.res
    events = torch.cat(win.events, 1)
    ef = torch.cat(win.ev_flow[-1], 1)
    lin, wgt = _round_images(events, ef, None, P, res, flow_scaling)
    iwe = _splat(lin, wgt, res)
    lin0, wgt0 = _round_images(events, ef * 0, None, P, res, flow_scaling)
    ie = _splat(lin0, wgt0, res)
    return torch.var(iwe, dim=1) / torch.var(ie, dim=1)  # unbiased, loss/flow.py:13-23


def rsat(win, flow_scaling):
    """RSAT.forward, loss/flow.py:514-579, [B]."""
    P, res = win.passes, win.res
    events = torch.cat(win.events, 1)
    pol = torch.cat(win.pol, 1)
    ef = torch.cat(win.ev_flow[-1], 1)
    ts = events[:, :, 0]
    lin, wgt = _round_images(events, ef, pol, P, res, flow_scaling)
    num = _avg_ts_term(lin, wgt, ts, pol, res, P)
    lin0, wgt0 = _round_images(events, ef * 0, pol, P, res, flow_scaling)
    den = _avg_ts_term(lin0, wgt0, ts, pol, res, P)
    return num / den


def aee(flow_last, gtflow, event_mask_last, flow_scaling, dt_gt, dt_input):
    """AEE.forward, loss/flow.py:594-628.  dt ratio applied per sample
    ([B,1,1,1]); identical to the reference for B = 1 (quirk q11).
    Returns (AEE [B], percent_outliers [B]) with the reference's batch-wide
    outlier sum (:626)."""
    B = flow_last.shape[0]
    ratio = (torch.as_tensor(dt_gt, dtype=flow_last.dtype) / torch.as_tensor(dt_input, dtype=flow_last.dtype)).reshape(
        -1, 1, 1, 1
    )
    flow = flow_last * flow_scaling * ratio
    mag = flow.pow(2).sum(1).sqrt()
    err = (flow - gtflow).pow(2).sum(1).sqrt()
    valid = event_mask_last.bool() & ~((gtflow[:, 0] == 0.0) & (gtflow[:, 1] == 0.0))
    valid = valid.reshape(B, -1)
    err = err.reshape(B, -1) * valid
    mag = mag.reshape(B, -1) * valid
    n = valid.sum(1)
    a = err.sum(1) / (n + 1e-9)
    outl = (err > 3.0) & (err > 0.05 * mag)
    return a, outl.sum() / (n + 1e-9)


def window_events(win):
    """compute_window_events, loss/flow.py:432-441: per-polarity count image."""
    events = torch.cat(win.events, 1)
    pol = torch.cat(win.pol, 1)
    lin = (events[:, :, 1] * win.res[1] + events[:, :, 2]).long()
    pos = _splat(lin, pol[:, :, 0], win.res)
    neg = _splat(lin, pol[:, :, 1], win.res)
    return torch.stack([pos, neg], 1).reshape(-1, 2, *win.res)


def window_iwe(win, flow_scaling, round_idx=True):
    """compute_window_iwe, loss/flow.py:454-465 -> [B,2,H,W]."""
    P, res = win.passes, win.res
    events = torch.cat(win.events, 1)
    pol = torch.cat(win.pol, 1)
    ef = torch.cat(win.ev_flow[-1], 1)
    lin, wgt = _corners(_warp(events, ef, P, flow_scaling), res, round_idx)
    rep = lin.shape[1] // events.shape[1]
    polr = pol.repeat(1, rep, 1)
    pos = _splat(lin, wgt * polr[:, :, 0], res)
    neg = _splat(lin, wgt * polr[:, :, 1], res)
    return torch.stack([pos, neg], 1).reshape(-1, 2, *res)


def masked_window_flow(win, overwrite):
    """compute_masked_window_flow, loss/flow.py:443-452 (last scale)."""
    mask = torch.cat(win.mask, 1)
    if overwrite:
        return win.flow_maps[-1][-1] * mask
    acc = 0
    for k, fm in enumerate(win.flow_maps):
        acc = acc + fm[-1] * mask[:, k : k + 1]
    return acc / (mask.sum(1, keepdim=True) + 1e-9)
