#!/usr/bin/env python
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (tudelft/event_flow,
mounted read-only at /root/reference) in the build container.

The reference has no tests or golden vectors of its own, so
these fixtures are what pins the oracle (oracle/) and, through it, the HIP
path.  Each fixture holds inputs + the reference's outputs (data only).
This script is the only place the reference is imported; it cannot run on the
GPU box (no /root/reference there) and nothing at test/bench time needs it.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py [generator ...]

Recorded skew: the reference pins torch==1.7.0 (requirements.txt:1); fixtures
are produced with the torch in this image (see meta.json), whose
`torch.max(zeros, x)` tie sub-gradient is 0.5/0.5.
"""

import json
import os
import sys

sys.dont_write_bytecode = True
REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)

import numpy as np
import torch

torch.set_num_threads(4)

from dataloader import encodings as r_enc  # noqa: E402  (reference)
from dataloader.base import BaseDataLoader as r_Base  # noqa: E402
from loss import flow as r_loss  # noqa: E402
from models import model as r_model  # noqa: E402
from models import spiking_submodules as r_cells  # noqa: E402
from utils import iwe as r_iwe  # noqa: E402

from event_flow_amd import synthetic  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def save(name, **arrays):
    conv = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(conv)} arrays")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def window_tensors(n, H, W, seed, num_bins=2, round_ts=False, kind="uniform"):
    """Run the REFERENCE encodings on one synthetic window -> dict of tensors."""
    if kind == "uniform":
        xs, ys, ts, ps = synthetic.uniform_events(n, H, W, seed)
    else:
        xs, ys, ts, ps, _ = synthetic.moving_dots_events(n, H, W, seed, max_disp=6.0, k=40)
    xs, ys, ts, ps = T(xs), T(ys), T(ts), T(ps)
    d = {
        "event_cnt": r_enc.events_to_channels(xs, ys, ps, sensor_size=(H, W)),
        "event_voxel": r_enc.events_to_voxel(xs, ys, ts, ps, num_bins, sensor_size=(H, W), round_ts=round_ts),
        "event_mask": r_enc.events_to_image(xs, ys, ps.abs(), sensor_size=(H, W), accumulate=False).view(1, H, W),
        "event_list": r_Base.create_list_encoding(xs, ys, ts, ps),
        "event_list_pol_mask": r_Base.create_polarity_mask(ps),
    }
    return d


def batch_windows(B, n, H, W, seed0, **kw):
    return r_Base.custom_collate([window_tensors(n, H, W, seed0 + b, **kw) for b in range(B)])


# ----------------------------------------------------------------------------
def g1_encodings():
    H, W, n = 40, 48, 600
    xs, ys, ts, ps = synthetic.uniform_events(n, H, W, 101)
    # duplicate some pixels so accumulation matters
    xs[:50], ys[:50] = xs[50:100], ys[50:100]
    a = dict(xs=xs, ys=ys, ts=ts, ps=ps, sensor=np.array([H, W]))
    txs, tys, tts, tps = T(xs), T(ys), T(ts), T(ps)
    a["cnt"] = r_enc.events_to_channels(txs, tys, tps, sensor_size=(H, W))
    a["mask"] = r_enc.events_to_image(txs, tys, tps.abs(), sensor_size=(H, W), accumulate=False)
    a["image_acc"] = r_enc.events_to_image(txs, tys, tps, sensor_size=(H, W), accumulate=True)
    for nb in (2, 5):
        for rnd in (0, 1):
            a[f"voxel_nb{nb}_r{rnd}"] = r_enc.events_to_voxel(txs, tys, tts, tps, nb, sensor_size=(H, W), round_ts=bool(rnd))
    a["list"] = r_Base.create_list_encoding(txs, tys, tts, tps)
    a["polmask"] = r_Base.create_polarity_mask(tps)
    # event_formatting on raw integer-ish inputs
    raw_t = np.sort(np.random.default_rng(5).uniform(10.0, 10.5, 64))
    raw_p = np.random.default_rng(6).integers(0, 2, 64)
    fx, fy, ft, fp = r_Base.event_formatting(xs[:64].copy(), ys[:64].copy(), raw_t.copy(), raw_p.copy())
    a.update(raw_t=raw_t, raw_p=raw_p, fmt_t=ft, fmt_p=fp)
    # collate layout
    col = batch_windows(2, 50, H, W, 200)
    for k, v in col.items():
        a["collate_" + k] = v
    save("g1_encodings", **a)


def special_events(B, n, H, W, seed):
    """Random events + flows with planted edge cases: out of image on each
    side, partially out, exactly-integer warped coordinates, zero flow."""
    g = np.random.default_rng(seed)
    ev = synthetic.event_list_batch(B, n, H, W, seed)
    fl = g.uniform(-0.4, 0.4, size=(B, n, 2)).astype(np.float32)
    fl[:, :10] = 0.0  # exact-integer warped coords (tie weights)
    fl[:, 10:14] = 3.0  # far out of the image
    fl[:, 14:18] = -3.0
    ev[:, 18, 1:3] = (0, 0)
    fl[:, 18] = (-0.001, -0.001)  # straddles the top/left border
    ev[:, 19, 1:3] = (H - 1, W - 1)
    fl[:, 19] = (0.001, 0.001)  # straddles the bottom/right border
    ev[:, 20, 0] = 0.5
    fl[:, 20] = (2.0 / 16, -4.0 / 16)  # lands exactly on an integer for S=16, tref=1
    return ev, fl


def g2_interpolation():
    B, n, H, W = 2, 200, 24, 32
    ev, fl = special_events(B, n, H, W, 7)
    a = dict(events=ev, flow=fl, res=np.array([H, W]))
    for tref in (1, 3, 0):
        for rnd in (0, 1):
            for S in (16, 128):
                idx, w = r_iwe.get_interpolation(T(ev), T(fl), tref, (H, W), S, round_idx=bool(rnd))
                a[f"idx_t{tref}_r{rnd}_s{S}"] = idx
                a[f"w_t{tref}_r{rnd}_s{S}"] = w
    save("g2_interpolation", **a)


def g3_pol_iwe():
    a = {}
    for tag, (B, n, H, W, amp) in {"c1": (1, 1000, 64, 64, 0.05), "b2": (2, 700, 48, 40, 0.2)}.items():
        g = np.random.default_rng(11)
        ev = synthetic.event_list_batch(B, n, H, W, 300)
        flow = g.uniform(-amp, amp, size=(B, 2, H, W)).astype(np.float32)
        pol = np.stack([(ev[:, :, 3] > 0), (ev[:, :, 3] < 0)], 2).astype(np.float32)
        for S in (128, 32):
            for rnd in (1, 0):
                iwe = r_iwe.compute_pol_iwe(T(flow), T(ev), (H, W), T(pol[:, :, 0:1]), T(pol[:, :, 1:2]), flow_scaling=S, round_idx=bool(rnd))
                a[f"{tag}_iwe_s{S}_r{rnd}"] = iwe
        a[f"{tag}_events"], a[f"{tag}_flow"], a[f"{tag}_pol"], a[f"{tag}_res"] = ev, flow, pol, np.array([H, W])
    save("g3_pol_iwe", **a)


def loss_config(H, W, mask, overwrite, weight=0.001):
    return {"loader": {"resolution": [H, W]}, "loss": {"flow_regul_weight": weight, "overwrite_intermediate": overwrite}, "model": {"mask_output": mask}}


def g4_event_warping():
    B, n, H, W = 2, 150, 32, 32
    a = dict(res=np.array([H, W]))
    case = 0
    cases = []
    for P in (1, 3):
        for mask in (True, False):
            for overwrite in (False, True):
                for scales in (1, 3):
                    for zero_flow in (False, True):
                        if zero_flow and not (mask and scales == 1):
                            continue
                        g = np.random.default_rng(1000 + case)
                        lossf = r_loss.EventWarping(loss_config(H, W, mask, overwrite), "cpu")
                        flows_all = []
                        batches = batch_windows(B, n, H, W, 400 + 10 * case)
                        per_pass = []
                        for k in range(P):
                            d = batch_windows(B, n, H, W, 400 + 10 * case + 100 * k, kind="dots" if k % 2 else "uniform")
                            fl = []
                            for s in range(scales):
                                f = g.uniform(-0.08, 0.08, size=(B, 2, H, W)).astype(np.float32)
                                if zero_flow:
                                    f[:] = 0
                                fl.append(T(f).requires_grad_(True))
                            flows_all.append(fl)
                            per_pass.append(d)
                            lossf.event_flow_association(fl, d["event_list"].clone(), d["event_list_pol_mask"], d["event_mask"])
                        if overwrite:
                            lossf.overwrite_intermediate_flow(flows_all[-1])
                        val = lossf()
                        grads = torch.autograd.grad(val, [f for fl in flows_all for f in fl], allow_unused=True)
                        tag = f"c{case}"
                        cases.append(dict(tag=tag, P=P, mask=mask, overwrite=overwrite, scales=scales, zero_flow=zero_flow))
                        a[tag + "_loss"] = val
                        gi = 0
                        for k in range(P):
                            for key in ("event_list", "event_list_pol_mask", "event_mask"):
                                a[f"{tag}_p{k}_{key}"] = per_pass[k][key]
                            for s in range(scales):
                                a[f"{tag}_p{k}_flow{s}"] = flows_all[k][s]
                                gr = grads[gi]
                                a[f"{tag}_p{k}_gflow{s}"] = gr if gr is not None else torch.zeros(B, 2, H, W)
                                gi += 1
                        case += 1
    a["cases_json"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    save("g4_event_warping", **a)


def g5_metrics():
    B, n, H, W, P = 2, 400, 32, 40, 2
    g = np.random.default_rng(77)
    cfg = loss_config(H, W, True, False)
    a = dict(res=np.array([H, W]), P=np.array(P))
    for overwrite in (False, True):
        cfg["loss"]["overwrite_intermediate"] = overwrite
        ms = [r_loss.FWL(cfg, "cpu", flow_scaling=32), r_loss.RSAT(cfg, "cpu", flow_scaling=32), r_loss.AEE(cfg, "cpu", flow_scaling=32)]
        tag = f"ow{int(overwrite)}"
        last = None
        for k in range(P):
            d = batch_windows(B, n, H, W, 900 + 50 * k, kind="dots")
            flow = T(g.uniform(-0.2, 0.2, size=(B, 2, H, W)).astype(np.float32))
            gt = g.uniform(-4, 4, size=(B, 2, H, W)).astype(np.float32)
            gt[:, :, :4] = 0  # invalid ground truth rows
            d["gtflow"] = T(gt)
            d["dt_input"] = torch.tensor([1.0])
            d["dt_gt"] = torch.tensor([1.0])
            for m in ms:
                m.event_flow_association([flow], d)
            for key in ("event_list", "event_list_pol_mask", "event_mask", "gtflow"):
                a[f"{tag}_p{k}_{key}"] = d[key]
            a[f"{tag}_p{k}_flow"] = flow
            last = flow
        if overwrite:
            for m in ms:
                m.overwrite_intermediate_flow([last])
        a[tag + "_fwl"] = ms[0]()
        a[tag + "_rsat"] = ms[1]()
        a[tag + "_window_events"] = ms[0].compute_window_events()
        a[tag + "_window_iwe"] = ms[0].compute_window_iwe()
        a[tag + "_masked_flow"] = ms[0].compute_masked_window_flow()
    # AEE is only meaningful for B = 1 in the reference (quirk q11)
    cfg["loss"]["overwrite_intermediate"] = False
    m = r_loss.AEE(cfg, "cpu", flow_scaling=32)
    d = batch_windows(1, n, H, W, 950, kind="dots")
    flow = T(g.uniform(-0.2, 0.2, size=(1, 2, H, W)).astype(np.float32))
    gt = g.uniform(-4, 4, size=(1, 2, H, W)).astype(np.float32)
    gt[:, :, :4] = 0
    d["gtflow"], d["dt_input"], d["dt_gt"] = T(gt), torch.tensor([0.5]), torch.tensor([1.25])
    m.event_flow_association([flow], d)
    ae, pe = m()
    a.update(aee_event_mask=d["event_mask"], aee_flow=flow, aee_gt=gt, aee_dt=np.array([1.25, 0.5]), aee_val=ae, aee_outl=pe)
    save("g5_metrics", **a)


CELLS = {
    "lif": (r_cells.ConvLIF, r_cells.ConvLIFRecurrent),
    "plif": (r_cells.ConvPLIF, r_cells.ConvPLIFRecurrent),
    "alif": (r_cells.ConvALIF, r_cells.ConvALIFRecurrent),
    "xlif": (r_cells.ConvXLIF, r_cells.ConvXLIFRecurrent),
}


def g6_cells():
    B, Cin, C, H, W = 2, 4, 8, 12, 10
    a = {}
    cases = []
    ci = 0
    for kind, (FF, REC) in CELLS.items():
        for recurrent in (False, True):
            for hard in (True, False):
                acts = ["arctanspike", "superspike", "trianglespike", "mgspike"] if kind == "lif" else ["arctanspike"]
                for act in acts:
                    torch.manual_seed(50 + ci)
                    kw = dict(activation=act, hard_reset=hard)
                    if kind in ("lif", "plif"):
                        kw["thresh"] = (0.3, 0.1)
                    else:
                        kw["t0"] = (0.2, 0.05)
                        kw["t1"] = (0.5, 0.1)
                        kw["learn_thresh"] = True
                    if act == "mgspike":
                        kw["act_width"] = 0.5
                    if act == "trianglespike":
                        kw["act_width"] = 1.0
                    cell = (REC if recurrent else FF)(Cin if not recurrent else C, C, 3, **kw)
                    cin = Cin if not recurrent else C
                    x = (torch.rand(B, cin, H, W) < 0.3).float() * torch.randint(1, 3, (B, cin, H, W)).float()
                    x.requires_grad_(True)
                    nstate = 2 if kind == "lif" else 3
                    st = torch.rand(nstate, B, C, H, W)
                    st[1] = (st[1] < 0.3).float()
                    st.requires_grad_(True)
                    out, new = cell(x, st)
                    g_out = torch.randn_like(out)
                    g_new = torch.randn_like(new) * 0.5
                    params = dict(cell.named_parameters())
                    grads = torch.autograd.grad([out, new], [x, st] + list(params.values()), [g_out, g_new], allow_unused=True)
                    tag = f"k{ci}"
                    cases.append(dict(tag=tag, kind=kind, recurrent=recurrent, hard_reset=hard, act=act))
                    a.update({tag + "_x": x, tag + "_state": st, tag + "_out": out, tag + "_new": new, tag + "_g_out": g_out, tag + "_g_new": g_new})
                    a[tag + "_gx"] = grads[0]
                    a[tag + "_gstate"] = grads[1]
                    for (pn, _), gr in zip(params.items(), grads[2:]):
                        a[f"{tag}_grad_{pn}"] = gr if gr is not None else torch.zeros_like(params[pn])
                    for pn, v in cell.state_dict().items():
                        a[f"{tag}_param_{pn}"] = v
                    ci += 1
    a["cases_json"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    save("g6_cells", **a)


def g15_cells_k5():
    """Single steps of the reference's cells with the 5x5 kernels models/unet.py:51 defaults to (and a 7x7 one): feed-forward
    stride 1 and 2, recurrent, PLIF (its trace pools with the same kernel size, spiking_submodules.py:212)."""
    B, Cin, C, H, W = 2, 4, 8, 13, 11
    a, cases = {}, []
    todo = [("lif", False, 5, 1), ("lif", False, 5, 2), ("lif", True, 5, 1), ("plif", False, 5, 1), ("plif", True, 5, 1),
            ("alif", False, 5, 2), ("xlif", True, 5, 1), ("lif", False, 7, 1)]
    for ci, (kind, recurrent, k, stride) in enumerate(todo):
        FF, REC = CELLS[kind]
        torch.manual_seed(150 + ci)
        kw = dict(activation="arctanspike", hard_reset=True)
        if kind in ("lif", "plif"):
            kw["thresh"] = (0.3, 0.1)
        else:
            kw.update(t0=(0.2, 0.05), t1=(0.5, 0.1), learn_thresh=True)
        cin = C if recurrent else Cin
        cell = REC(cin, C, k, **kw) if recurrent else FF(cin, C, k, stride=stride, **kw)
        x = (torch.rand(B, cin, H, W) < 0.3).float() * torch.randint(1, 3, (B, cin, H, W)).float()
        x.requires_grad_(True)
        Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
        nstate = 2 if kind == "lif" else 3
        st = torch.rand(nstate, B, C, Ho, Wo)
        st[1] = (st[1] < 0.3).float()
        st.requires_grad_(True)
        out, new = cell(x, st)
        g_out, g_new = torch.randn_like(out), torch.randn_like(new) * 0.5
        params = dict(cell.named_parameters())
        grads = torch.autograd.grad([out, new], [x, st] + list(params.values()), [g_out, g_new], allow_unused=True)
        tag = f"k{ci}"
        cases.append(dict(tag=tag, kind=kind, recurrent=recurrent, hard_reset=True, act="arctanspike", ksz=k, stride=stride))
        a.update({tag + "_x": x, tag + "_state": st, tag + "_out": out, tag + "_new": new, tag + "_g_out": g_out, tag + "_g_new": g_new,
                  tag + "_gx": grads[0], tag + "_gstate": grads[1]})
        for (pn, _), gr in zip(params.items(), grads[2:]):
            a[f"{tag}_grad_{pn}"] = gr if gr is not None else torch.zeros_like(params[pn])
        for pn, v in cell.state_dict().items():
            a[f"{tag}_param_{pn}"] = v
    a["cases_json"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    save("g15_cells_k5", **a)


def g18_cells_weightnorm():
    """Single steps of the reference's LIF cells with norm="weight" (spiking_submodules.py:87-88, :502-504: nn.utils.weight_norm
    on ff / rec): feed-forward stride 1 and 2, recurrent.  weight_g is moved away from ||weight_v|| (at construction w == v)."""
    B, Cin, C, H, W = 2, 4, 8, 12, 10
    a, cases = {}, []
    for ci, (recurrent, stride) in enumerate([(False, 1), (False, 2), (True, 1)]):
        torch.manual_seed(180 + ci)
        kw = dict(activation="arctanspike", hard_reset=True, thresh=(0.3, 0.1), norm="weight")
        cin = C if recurrent else Cin
        cell = r_cells.ConvLIFRecurrent(cin, C, 3, **kw) if recurrent else r_cells.ConvLIF(cin, C, 3, stride=stride, **kw)
        with torch.no_grad():
            cell.ff.weight_g.mul_(torch.rand_like(cell.ff.weight_g) + 0.5)
            if recurrent:
                cell.rec.weight_g.mul_(torch.rand_like(cell.rec.weight_g) + 0.5)
        x = (torch.rand(B, cin, H, W) < 0.3).float() * torch.randint(1, 3, (B, cin, H, W)).float()
        x.requires_grad_(True)
        Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
        st = torch.rand(2, B, C, Ho, Wo)
        st[1] = (st[1] < 0.3).float()
        st.requires_grad_(True)
        out, new = cell(x, st)
        g_out, g_new = torch.randn_like(out), torch.randn_like(new) * 0.5
        params = dict(cell.named_parameters())
        grads = torch.autograd.grad([out, new], [x, st] + list(params.values()), [g_out, g_new], allow_unused=True)
        tag = f"k{ci}"
        cases.append(dict(tag=tag, kind="lif", recurrent=recurrent, hard_reset=True, act="arctanspike", ksz=3, stride=stride, norm="weight"))
        a.update({tag + "_x": x, tag + "_state": st, tag + "_out": out, tag + "_new": new, tag + "_g_out": g_out, tag + "_g_new": g_new,
                  tag + "_gx": grads[0], tag + "_gstate": grads[1]})
        for (pn, _), gr in zip(params.items(), grads[2:]):
            a[f"{tag}_grad_{pn}"] = gr if gr is not None else torch.zeros_like(params[pn])
        for pn, v in cell.state_dict().items():
            a[f"{tag}_param_{pn}"] = v
    a["cases_json"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    save("g18_cells_weightnorm", **a)


def g19_cells_groupnorm():
    """Single steps of the reference's LIF cells with norm="group" (spiking_submodules.py:90-99, :507-529: nn.GroupNorm(1, C) on the
    input; recurrent cell: also on the previous spikes, whose normalised values enter the reset too).  The affine parameters are
    moved away from (1, 0)."""
    B, Cin, C, H, W = 2, 4, 8, 12, 10
    a, cases = {}, []
    for ci, (recurrent, stride) in enumerate([(False, 1), (False, 2), (True, 1)]):
        torch.manual_seed(190 + ci)
        kw = dict(activation="arctanspike", hard_reset=True, thresh=(0.3, 0.1), norm="group")
        cin = C if recurrent else Cin
        cell = r_cells.ConvLIFRecurrent(cin, C, 3, **kw) if recurrent else r_cells.ConvLIF(cin, C, 3, stride=stride, **kw)
        with torch.no_grad():
            for n, p in cell.named_parameters():
                if n.startswith("norm"):
                    p.add_(torch.randn_like(p) * 0.3)
        x = (torch.rand(B, cin, H, W) < 0.3).float() * torch.randint(1, 3, (B, cin, H, W)).float()
        x.requires_grad_(True)
        Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
        st = torch.rand(2, B, C, Ho, Wo)
        st[1] = (st[1] < 0.3).float()
        st.requires_grad_(True)
        out, new = cell(x, st)
        g_out, g_new = torch.randn_like(out), torch.randn_like(new) * 0.5
        params = dict(cell.named_parameters())
        grads = torch.autograd.grad([out, new], [x, st] + list(params.values()), [g_out, g_new], allow_unused=True)
        tag = f"k{ci}"
        cases.append(dict(tag=tag, kind="lif", recurrent=recurrent, hard_reset=True, act="arctanspike", ksz=3, stride=stride, norm="group"))
        a.update({tag + "_x": x, tag + "_state": st, tag + "_out": out, tag + "_new": new, tag + "_g_out": g_out, tag + "_g_new": g_new,
                  tag + "_gx": grads[0], tag + "_gstate": grads[1]})
        for (pn, _), gr in zip(params.items(), grads[2:]):
            a[f"{tag}_grad_{pn}"] = gr if gr is not None else torch.zeros_like(params[pn])
        for pn, v in cell.state_dict().items():
            a[f"{tag}_param_{pn}"] = v
    a["cases_json"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    save("g19_cells_groupnorm", **a)


def g16_norm_layers():
    """The reference's ANN layers with norm = "BN" / "IN" and its transposed-conv decoder layer (models/submodules.py:12-137,
    140-185, 238-311), and a MultiResUNet built with norm="BN", use_upsample_conv=False (models/unet.py:196-311): two
    forward calls in train mode (the running statistics move), gradients of sum(y^2) + sum(y) over both, one forward in eval
    mode (running statistics normalise)."""
    from models import submodules as r_sub
    from models import unet as r_unet

    a, cases = {}, []
    B, H, W = 2, 12, 10
    todo = [
        ("ConvLayer", dict(in_channels=4, out_channels=8, kernel_size=3, activation="relu", norm="BN", BN_momentum=0.3)),
        ("ConvLayer", dict(in_channels=4, out_channels=8, kernel_size=3, stride=2, activation="tanh", norm="IN")),
        ("ConvLayer_", dict(in_channels=8, out_channels=8, kernel_size=3, activation="relu", norm="BN")),
        ("TransposedConvLayer", dict(in_channels=8, out_channels=4, kernel_size=3, activation="relu", norm=None)),
        ("TransposedConvLayer", dict(in_channels=8, out_channels=4, kernel_size=5, activation="tanh", norm="BN")),
        ("TransposedConvLayer", dict(in_channels=6, out_channels=8, kernel_size=3, activation=None, norm="IN")),
        ("UpsampleConvLayer", dict(in_channels=4, out_channels=8, kernel_size=3, activation="relu", norm="IN")),
        ("ResidualBlock", dict(in_channels=8, out_channels=8, activation="relu", norm="BN")),
        ("ResidualBlock", dict(in_channels=8, out_channels=8, activation="relu", norm="IN")),
        ("MultiResUNet", dict(base_num_channels=4, num_encoders=2, num_residual_blocks=1, num_output_channels=2, skip_type="concat",
                              norm="BN", use_upsample_conv=False, num_bins=2, kernel_size=3, channel_multiplier=2,
                              activations=["relu", None], final_activation="tanh")),
    ]
    for ci, (cls, kw) in enumerate(todo):
        torch.manual_seed(300 + ci)
        tag = f"n{ci}"
        if cls == "MultiResUNet":
            m = r_unet.MultiResUNet(dict(kw))
            cin, hh, ww = 2, 16, 16
        else:
            m = getattr(r_sub, cls)(**kw)
            cin, hh, ww = kw["in_channels"], H, W
        m.train()
        for pn, v in m.state_dict().items():
            a[f"{tag}_param0_{pn}"] = v.clone()
        xs = [torch.randn(B, cin, hh, ww).requires_grad_(True) for _ in range(2)]
        res = [torch.randn(B, cin, hh, ww) for _ in range(2)]  # (ConvLayer_: residual of the output's shape, Cin = Cout here)
        tot = 0
        for k, x in enumerate(xs):
            if cls == "ConvLayer_":
                y = m(x, None, res[k])[0]
                a[f"{tag}_res{k}"] = res[k]
            else:
                y = m(x)
            ys = y if isinstance(y, (list, tuple)) else [y]
            for j, yy in enumerate(ys):
                a[f"{tag}_y{k}_{j}"] = yy
                tot = tot + yy.pow(2).sum() + yy.sum()
            a[f"{tag}_x{k}"] = x
        grads = torch.autograd.grad(tot, xs + list(m.parameters()), allow_unused=True)
        for k in range(2):
            a[f"{tag}_gx{k}"] = grads[k]
        for (pn, prm), gr in zip(m.named_parameters(), grads[2:]):
            a[f"{tag}_grad_{pn}"] = gr if gr is not None else torch.zeros_like(prm)
        for pn, v in m.state_dict().items():
            a[f"{tag}_param1_{pn}"] = v.clone()  # running statistics after the two training calls
        m.eval()
        with torch.no_grad():
            ye = m(xs[0], None, res[0])[0] if cls == "ConvLayer_" else m(xs[0])
        for j, yy in enumerate(ye if isinstance(ye, (list, tuple)) else [ye]):
            a[f"{tag}_yeval_{j}"] = yy
        cases.append(dict(tag=tag, cls=cls, kwargs=kw))
    a["cases_json"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    save("g16_norm_layers", **a)


def model_cfg(name, C=32, neuron=None, num_bins=2, encoding="cnt", acts=("arctanspike", "arctanspike")):
    return {
        "name": name, "encoding": encoding, "round_encoding": False, "norm_input": False, "num_bins": num_bins,
        "base_num_channels": C, "kernel_size": 3, "activations": list(acts), "mask_output": True,
        "spiking_neuron": neuron,
    }


LIF_NEURON = {"leak": [-4.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True, "learn_thresh": True, "hard_reset": True}
PLIF_NEURON = {"leak_v": [-4.0, 0.1], "leak_pt": [-4.0, 0.1], "add_pt": [-2.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True, "learn_thresh": True, "hard_reset": True}


def build(name, cfg):
    r_model.FireNet.kwargs[0].clear()  # shared class-level dict (quirk q2)
    return getattr(r_model, name)(dict(cfg))


def g7_firenet_train(name="LIFFireNet", neuron=LIF_NEURON, fname="g7_liffirenet_train", thresh_scale=None):
    """state_dict + 3 passes -> per-layer v', z', flow, loss, grads, params after one clip+Adam step."""
    torch.manual_seed(0)
    B, n, H, W, P = 2, 220, 24, 24, 3
    model = build(name, model_cfg(name, neuron=neuron))
    if thresh_scale is not None:  # lower thresholds so the small fixture actually spikes in deep layers
        with torch.no_grad():
            for pn, prm in model.named_parameters():
                if pn.endswith("thresh"):
                    prm.mul_(thresh_scale)
    model.train()
    a = {}
    for pn, v in model.state_dict().items():
        a["param0_" + pn] = v.clone()
    lossf = r_loss.EventWarping(loss_config(H, W, True, False), "cpu")
    opt = torch.optim.Adam(model.parameters(), lr=2e-4)
    opt.zero_grad()
    layer_names = ["head", "G1", "R1a", "R1b", "G2", "R2a", "R2b"]
    for k in range(P):
        d = batch_windows(B, n, H, W, 2000 + 10 * k, kind="dots" if k else "uniform")
        out = model(d["event_voxel"], d["event_cnt"], log=True)
        for key in ("event_cnt", "event_voxel", "event_list", "event_list_pol_mask", "event_mask"):
            a[f"p{k}_{key}"] = d[key]
        a[f"p{k}_flow"] = out["flow"][0]
        for li, ln in enumerate(layer_names):
            st = model._states[li].detach()
            a[f"p{k}_v_{ln}"] = st[0]
            a[f"p{k}_z_{ln}"] = st[1].to(torch.uint8)  # spikes are exactly 0/1
            if st.shape[0] == 3:
                a[f"p{k}_aux_{ln}"] = st[2]
        print(name, "pass", k, "activity", {kk: round(vv, 4) for kk, vv in out["activity"].items()})
        a[f"p{k}_activity"] = np.array([out["activity"][kk] for kk in sorted(out["activity"])])
        lossf.event_flow_association(out["flow"], d["event_list"].clone(), d["event_list_pol_mask"], d["event_mask"])
    loss = lossf()
    loss.backward()
    a["loss"] = loss
    for pn, prm in model.named_parameters():
        a["grad_" + pn] = prm.grad.clone() if prm.grad is not None else torch.zeros_like(prm)
    a["grad_norm"] = torch.nn.utils.clip_grad.clip_grad_norm_(model.parameters(), 100.0)
    opt.step()
    for pn, v in model.state_dict().items():
        a["param1_" + pn] = v.clone()
    a["meta_P"] = np.array(P)
    save(fname, **a)


def g8_firenet_ann():
    torch.manual_seed(1)
    B, n, H, W = 1, 1000, 64, 64
    model = build("FireNet", model_cfg("FireNet", neuron=None, encoding="voxel", acts=("relu", None)))
    model.eval()
    a = {}
    for pn, v in model.state_dict().items():
        a["param_" + pn] = v.clone()
    with torch.no_grad():
        for k in range(2):
            d = batch_windows(B, n, H, W, 3000 + k)
            out = model(d["event_voxel"], d["event_cnt"])
            a[f"p{k}_event_voxel"], a[f"p{k}_event_cnt"] = d["event_voxel"], d["event_cnt"]
            a[f"p{k}_flow"] = out["flow"][0]
            a[f"p{k}_state_G1"], a[f"p{k}_state_G2"] = model._states[1], model._states[4]
    save("g8_firenet_ann", **a)


def g9_spiking_unet():
    torch.manual_seed(2)
    B, n, H, W = 1, 2000, 64, 64
    neuron = dict(LIF_NEURON)
    neuron["thresh"] = [0.2, 0.05]
    model = build("SpikingRecEVFlowNet", model_cfg("SpikingRecEVFlowNet", C=4, neuron=neuron))
    model.train()
    a = {}
    for pn, v in model.state_dict().items():
        a["param_" + pn] = v.clone()
    for k in range(2):
        d = batch_windows(B, n, H, W, 4000 + k)
        out = model(d["event_voxel"], d["event_cnt"])
        a[f"p{k}_event_cnt"] = d["event_cnt"]
        for s, f in enumerate(out["flow"]):
            a[f"p{k}_flow{s}"] = f
        for si, st in enumerate(model.multires_unetrec.states):
            a[f"p{k}_state{si}"] = st
    tot = sum(f.pow(2).sum() for f in out["flow"])
    tot.backward()
    for pn, prm in model.named_parameters():
        a["grad_" + pn] = prm.grad.clone() if prm.grad is not None else torch.zeros_like(prm)
    save("g9_spiking_unet", **a)


ANN_FIRENETS = {
    # name: (activations, neuron kwargs)
    "FireFlowNet": (("relu", "relu"), None),
    "RNNFireNet": (("relu", None), None),
    "LeakyFireNet": (("relu", None), {"leak": [-1.0, 0.5], "learn_leak": True}),
    "LeakyFireFlowNet": (("relu", "tanh"), {"leak": [-1.0, 0.5], "learn_leak": True}),
}


def g10_ann_firenets():
    """ANN comparison FireNets (models/model.py:398-409,614-633,696-704): 3 passes, loss = sum flow^2 + sum flow,
    per-pass flows, final states, parameter gradients."""
    B, n, H, W, P = 2, 400, 32, 32, 3
    a = {}
    batches = [batch_windows(B, n, H, W, 5000 + 10 * k) for k in range(P)]
    for k, d in enumerate(batches):
        a[f"p{k}_event_cnt"] = d["event_cnt"]
    for name, (acts, neuron) in ANN_FIRENETS.items():
        torch.manual_seed(3)
        model = build(name, model_cfg(name, C=8, neuron=neuron, acts=acts))
        model.train()
        for pn, v in model.state_dict().items():
            a[f"{name}.param_{pn}"] = v.clone()
        tot = 0
        for k, d in enumerate(batches):
            out = model(d["event_voxel"], d["event_cnt"])
            a[f"{name}.p{k}_flow"] = out["flow"][0]
            tot = tot + out["flow"][0].pow(2).sum() + out["flow"][0].sum()
        for li, st in enumerate(model._states):
            if torch.is_tensor(st) and st.dim() == 4:
                a[f"{name}.state{li}"] = st
        tot.backward()
        a[f"{name}.loss"] = tot
        for pn, prm in model.named_parameters():
            a[f"{name}.grad_{pn}"] = prm.grad.clone() if prm.grad is not None else torch.zeros_like(prm)
        print(name, "loss", float(tot), "params", sum(p.numel() for p in model.parameters()))
    save("g10_ann_firenets", **a)


ANN_UNETS = {
    "EVFlowNet": (("relu", None), None),
    "RecEVFlowNet": (("relu", None), None),
    "RNNRecEVFlowNet": (("relu", None), None),
    "LeakyRecEVFlowNet": (("relu", None), {"leak": [-1.0, 0.5], "learn_leak": True}),
}


def g11_ann_unets():
    """Non-spiking EV-FlowNets (models/model.py:289-395 EVFlowNet, :412-547 RecEVFlowNet/ConvGRU, :594-601 ConvRNN,
    :604-611 leaky): 2 passes at 32x32, loss = sum flow^2 + sum flow over the 4 scales of both passes,
    per-scale flows, final states, parameter gradients."""
    B, n, H, W, P = 1, 600, 32, 32, 2
    a = {}
    batches = [batch_windows(B, n, H, W, 6000 + 10 * k) for k in range(P)]
    for k, d in enumerate(batches):
        a[f"p{k}_event_cnt"] = d["event_cnt"]
    for name, (acts, neuron) in ANN_UNETS.items():
        torch.manual_seed(4)
        model = build(name, model_cfg(name, C=4, neuron=neuron, acts=acts))
        model.train()
        for pn, v in model.state_dict().items():
            a[f"{name}.param_{pn}"] = v.clone()
        tot = 0
        for k, d in enumerate(batches):
            out = model(d["event_voxel"], d["event_cnt"])
            assert len(out["flow"]) == 4
            for si, f in enumerate(out["flow"]):
                a[f"{name}.p{k}_flow{si}"] = f
                tot = tot + f.pow(2).sum() + f.sum()
        if hasattr(model, "multires_unetrec"):
            for si, st in enumerate(model.multires_unetrec.states):
                a[f"{name}.state{si}"] = st
        tot.backward()
        a[f"{name}.loss"] = tot.detach()
        for pn, prm in model.named_parameters():
            a[f"{name}.grad_{pn}"] = prm.grad.clone() if prm.grad is not None else torch.zeros_like(prm)
        print(name, "loss", float(tot.detach()), "params", sum(p.numel() for p in model.parameters()))
    save("g11_ann_unets", **a)


def g12_e2vid():
    """E2VID (models/model.py:29-145; ConvLSTM encoders, skip 'sum'): 3 passes at 32x32 with a state detach after
    the first, loss = sum flow^2 + sum flow, flows, final (hidden, cell) states, parameter gradients."""
    B, n, H, W, P = 2, 500, 32, 32, 3
    a = {}
    torch.manual_seed(5)
    model = build("E2VID", model_cfg("E2VID", C=4, neuron=None, acts=("relu", None)))
    model.train()
    for pn, v in model.state_dict().items():
        a[f"param_{pn}"] = v.clone()
    tot = 0
    for k in range(P):
        d = batch_windows(B, n, H, W, 7000 + 10 * k)
        a[f"p{k}_event_cnt"] = d["event_cnt"]
        out = model(d["event_voxel"], d["event_cnt"])
        a[f"p{k}_flow"] = out["flow"][0]
        tot = tot + out["flow"][0].pow(2).sum() + out["flow"][0].sum()
    for si, (h, c) in enumerate(model.unetrecurrent.states):
        a[f"state{si}_hidden"], a[f"state{si}_cell"] = h, c
    tot.backward()
    a["loss"] = tot.detach()
    for pn, prm in model.named_parameters():
        a[f"grad_{pn}"] = prm.grad.clone() if prm.grad is not None else torch.zeros_like(prm)
    print("E2VID loss", float(tot.detach()), "params", sum(p.numel() for p in model.parameters()))
    save("g12_e2vid", **a)


def g13_loader():
    """Loader batches as the reference makes them (dataloader/base.py: event_formatting :67-86, augment_events :88-116,
    create_*_encoding :150-222, create_hot_mask :224-243, custom_collate :248-265; per-sample order of
    dataloader/h5.py:276-295) for two raw sequences cut into count windows (mode "events", h5.py:148-150) with fixed
    augmentation flags and the hot-pixel filter on.  The reference's own H5Loader needs h5py and is not importable."""
    H, W, win, nwin, B, nb = 32, 40, 300, 5, 2, 5
    cfg = {"loader": {"batch_size": B, "resolution": [H, W], "augment": ["Horizontal", "Vertical", "Polarity"],
                      "augment_prob": [0.5, 0.5, 0.5]},
           "hot_filter": {"enabled": True, "max_px": 100, "min_obvs": 2, "max_rate": 0.8}}

    class L(r_Base):
        def __getitem__(self, index):
            raise NotImplementedError

    loader = L(cfg, nb, round_encoding=False)
    flags = {"Horizontal": [True, False], "Vertical": [False, True], "Polarity": [True, True]}
    loader.batch_augmentation = {k: list(v) for k, v in flags.items()}
    a = {"meta_HW_win_nwin_nb": np.array([H, W, win, nwin, nb])}
    for k, v in flags.items():
        a["aug_" + k] = np.array(v)
    seqs = []
    for b in range(B):
        rng = np.random.Generator(np.random.PCG64(900 + b))
        n = win * nwin + 37  # the tail is shorter than a window: the loader must move on to the next file there
        xs, ys = rng.integers(0, W, n), rng.integers(0, H, n)
        hot = [(3 + b, 5), (20, 7 + b), (31, 39)]  # pixels that fire in every window
        for w in range(nwin + 1):
            for j, (hy, hx) in enumerate(hot):
                k = w * win + 11 * (j + 1)
                if k < n:
                    ys[k], xs[k] = hy, hx
        ts = np.sort(rng.random(n)) * 0.5 + 12.25 + b  # seconds, t0 > 0
        ps = rng.integers(0, 2, n)
        seqs.append((xs, ys, ts, ps))
        for nm, v in zip(("xs", "ys", "ts", "ps"), (xs, ys, ts, ps)):
            a[f"seq{b}_{nm}"] = v
    for w in range(nwin):
        samples = []
        for b in range(B):
            xs, ys, ts, ps = (v[w * win:(w + 1) * win] for v in seqs[b])
            ts = ts - seqs[b][2][0]
            dt_input = np.asarray(ts[-1] - ts[0])
            txs, tys, tts, tps = loader.event_formatting(xs, ys, ts, ps)
            txs, tys, tps = loader.augment_events(txs, tys, tps, b)
            cnt = loader.create_cnt_encoding(txs, tys, tps)
            mask = loader.create_mask_encoding(txs, tys, tps)
            voxel = loader.create_voxel_encoding(txs, tys, tts, tps)
            lst = loader.create_list_encoding(txs, tys, tts, tps)
            pol = loader.create_polarity_mask(tps)
            hm = loader.create_hot_mask(cnt, b)
            voxel = voxel * torch.stack([hm] * nb, axis=2).permute(2, 0, 1)
            cnt = cnt * torch.stack([hm] * 2, axis=2).permute(2, 0, 1)
            mask *= hm.view((1, H, W))
            samples.append({"event_cnt": cnt, "event_voxel": voxel, "event_mask": mask, "event_list": lst,
                            "event_list_pol_mask": pol, "dt_gt": torch.from_numpy(np.asarray(0.0)),
                            "dt_input": torch.from_numpy(dt_input)})
        batch = loader.custom_collate(samples)
        for k, v in batch.items():
            a[f"w{w}_{k}"] = v
        print("window", w, "hot pixels removed:", [int((1 - (s["event_mask"] >= 0).float()).sum()) for s in samples],
              "mask zeros", int((batch["event_mask"] == 0).sum()))
    save("g13_loader", **a)


def g14_checkpoint():
    """A checkpoint as the reference stores it (utils/utils.py:36-37 -> mlflow.pytorch.log_model: the pickled model
    OBJECT in <run>/artifacts/model/data/model.pth) for a small LIF-FireNet, plus its flow on one input."""
    torch.manual_seed(7)
    model = build("LIFFireNet", model_cfg("LIFFireNet", C=8, neuron=LIF_NEURON))
    d = batch_windows(1, 300, 16, 16, 8000)
    with torch.no_grad():
        flow = model(d["event_voxel"], d["event_cnt"])["flow"][0]
    model.reset_states()
    run = os.path.join(OUT, "mlruns", "0", "0123456789abcdef0123456789abcdef", "artifacts", "model", "data")
    os.makedirs(run, exist_ok=True)
    torch.save(model, os.path.join(run, "model.pth"))
    save("g14_checkpoint", event_cnt=d["event_cnt"], flow=flow, **{"param_" + k: v for k, v in model.state_dict().items()})


if __name__ == "__main__":
    if len(sys.argv) > 1:  # only the named generators, e.g. `tools/gen_golden.py g10_ann_firenets`
        for fn in sys.argv[1:]:
            globals()[fn]()
        sys.exit(0)
    g1_encodings()
    g2_interpolation()
    g3_pol_iwe()
    g4_event_warping()
    g5_metrics()
    g6_cells()
    g7_firenet_train()
    g7_firenet_train("PLIFFireNet", PLIF_NEURON, "g7_pliffirenet_train")
    g7_firenet_train("LIFFireNet", LIF_NEURON, "g7_liffirenet_lowthresh", thresh_scale=0.15)
    g8_firenet_ann()
    g9_spiking_unet()
    g10_ann_firenets()
    g11_ann_unets()
    g12_e2vid()
    g13_loader()
    g14_checkpoint()
    meta = {"torch": torch.__version__, "numpy": np.__version__, "reference": "tudelft/event_flow @ /root/reference (v1)",
            "note": "outputs of the reference run in the build container; reference pins torch==1.7.0"}
    with open(os.path.join(OUT, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    g15_cells_k5()
    g16_norm_layers()
    g18_cells_weightnorm()
    g19_cells_groupnorm()
