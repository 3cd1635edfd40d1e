"""Teacher-forced parity of the recorded (diagonal) kernels on an ALIVE network at the benched sizes (VERDICT r04, weak #1;
SURVEY.md section 7's layered protocol (a)).

A free-running comparison of two fp32 implementations of a spiking network stops being tight as soon as ONE neuron sits on its
threshold: the Heaviside flips, the spike trains diverge, and gradients of two different spike trains are compared.  Teacher
forcing removes the chaos and keeps the arithmetic: every kernel is fed the ORACLE's inputs.

  forward   every (pass, layer) cell gets the oracle's input spikes and previous state; the hidden cells of a pass are RECORDED
            (evf_fwd_defer_*) and run through the diagonal launch k_fwd_diag_t like the benched step's.  Asserted per layer:
            v' within 1e-5 (rel-L2 and max-abs relative to max|v'|), z' equal wherever |v' - thresh| > eps, flow within 1e-5.
  backward  the HIP window runs free (recorded forward), then the oracle's potentials, spike words / bit planes, traces and
            flow maps are loaded INTO THE HIP TAPE, and the HIP backward -- recorded fused-backward / input-gradient / head
            window kernels -- runs from the oracle's dL/dflow.  Asserted: whole-vector gradient rel-L2 <= 1e-5 (per tensor 1e-4),
            clip + Adam update <= 1e-6 on the weights with signal (measured 5e-7 / 4e-8).

Reference: models/spiking_submodules.py:516-551 (ConvLIFRecurrent), :554-657 (ConvPLIFRecurrent), models/spiking_util.py:82-93
(arctan surrogate), models/model.py:255-265, loss/flow.py:176-301, train_flow.py:141-171."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from event_flow_amd import _lib  # noqa: E402
from event_flow_amd.models.model import LIFFireNet, PLIFFireNet  # noqa: E402
from event_flow_amd.train import FlatAdam, encode_passes  # noqa: E402
from oracle import loss as oloss  # noqa: E402
from oracle import snn as osnn  # noqa: E402
from oracle import train as otrain  # noqa: E402

DEV = "cuda:0"
LAYERS = ["head", "G1", "R1a", "R1b", "G2", "R2a", "R2b"]
EPS_THRESH = 1e-5  # spikes must agree wherever |v' - thresh| exceeds this


def N(t):
    return t.detach().cpu().numpy()


def _oracle_window(name, params, keys, passes, res, lcfg):
    """The oracle's window with everything kept: states[t][l] = (v', z'[, pt']), flows[t], loss, dL/dparams, dL/dflow[t]."""
    leaves = {k: t.detach().clone().requires_grad_(k in keys) for k, t in params.items()}
    win = oloss.Window(res)
    states, flows, per_pass = [None] * 7, [], []
    for d in passes:
        flow, states = osnn.firenet_forward(name, leaves, d["event_cnt"], states)
        win.add([flow], d["event_list"], d["event_list_pol_mask"], d["event_mask"])
        flows.append(flow)
        per_pass.append(states)
    loss = oloss.event_warping_loss(win, max(res), lcfg["flow_regul_weight"], smoothing_mask=lcfg.get("mask_output", True), overwrite=False)
    g = torch.autograd.grad(loss, [leaves[k] for k in keys] + flows, allow_unused=True)
    grads = {k: (gi if gi is not None else torch.zeros_like(leaves[k])) for k, gi in zip(keys, g[:len(keys)])}
    gflows = [gi if gi is not None else torch.zeros_like(f) for gi, f in zip(g[len(keys):], flows)]
    return {"loss": float(loss.detach()), "grads": grads, "gflows": [x.detach() for x in gflows], "flows": [f.detach() for f in flows],
            "states": [[tuple(x.detach() for x in st) for st in sts] for sts in per_pass]}


def _hip_state(eng, sts):
    """Oracle states of one pass [(v, z[, pt])] * 7 -> the engine's tensors [(v NHWC, z words, zT planes[, pt NHWC])] * 7."""
    eng.set_states([torch.stack([x.to(DEV) for x in st]) for st in sts])
    out = list(eng._states)
    eng._states = [None] * 7
    return out


def _read_state(eng, hip):
    """engine tensors -> list of numpy [S,B,C,H,W] (v', z'[, pt'])."""
    keep = eng._states
    eng._states = hip
    out = [N(s) for s in eng.get_states()]
    eng._states = keep
    return out


def _teacher_forced(cls, name, cfg, H, W, B, P, n_ev, thresh_scale, kind, seed):
    from event_flow_amd import synthetic

    plif = name == "PLIFFireNet"
    torch.manual_seed(seed)
    model = cls(dict(cfg)).to(DEV)
    model.precision = "bf16x3"
    with torch.no_grad():
        for k, p in model.named_parameters():
            if k.endswith("thresh"):
                p.mul_(thresh_scale)
    model.train()
    opt = FlatAdam(model, lr=2e-4, clip=100.0)
    opt.zero_grad()
    params = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    keys = osnn.trainable_keys(params)
    lists = []
    for k in range(P):
        ev = synthetic.event_list_batch(B, n_ev, H, W, 8100 + 100 * k, kind=kind)
        lists.append(torch.from_numpy(ev[0] if isinstance(ev, tuple) else ev).to(DEV))
    passes = encode_passes(lists, 2, (H, W), want=("cnt", "mask", "pol"))
    for d in passes:
        d["event_voxel"] = None
    opasses = [{k: v.detach().cpu() for k, v in d.items() if v is not None} for d in passes]
    lcfg = {"flow_regul_weight": 0.001, "mask_output": True}
    torch.set_num_threads(32)
    ora = _oracle_window(name, params, keys, opasses, (H, W), lcfg)
    rates = [float(st[1].mean()) for st in ora["states"][-1]]
    print(f"[teacher-forced {name} x{thresh_scale} {kind}] oracle spike rate per layer (last pass): {[f'{r:.4f}' for r in rates]}, "
          f"loss {ora['loss']:.6f}")
    assert min(rates[:5]) > 1e-3, rates  # the network is alive (the point of this test)

    eng = model._eng()
    eng._prepare(torch.device(DEV))
    F, PK = eng._flat, eng._packed
    hard = 1
    L = _lib.load()

    # ---------------- forward, cell by cell on the oracle's inputs; the hidden cells of a pass as ONE recorded diagonal launch
    hip_prev = [None] * 7
    worst_v, worst_flow, band_mismatch, flips_outside = 0.0, 0.0, 0, 0
    for t in range(P):
        ost = ora["states"][t]
        inp = _hip_state(eng, ost)  # (z words of layer l - 1 = the input of layer l)
        x_in = passes[t]["event_cnt"].detach().float().contiguous()
        outs = []
        for i in range(7):
            v_out = torch.empty((B, H, W, 32), dtype=torch.float32, device=DEV)
            z_out = torch.empty((B, H, W), dtype=torch.int32, device=DEV)
            zT_out = torch.empty((B, H, 32, (W + 31) // 32), dtype=torch.int32, device=DEV)
            pt_out = torch.empty((B, H, W, 32), dtype=torch.float32, device=DEV) if plif else None
            P_out = torch.empty((B, H, W), dtype=torch.float32, device=DEV) if plif else None
            outs.append((v_out, z_out, zT_out, pt_out, P_out))
        flow = torch.empty((B, 2, H, W), dtype=torch.float32, device=DEV)
        pv = hip_prev
        st0 = pv[0]
        # head layer: launches at once (it reads the network input and its own state)
        if plif:
            _lib.call("evf_head_plif_fwd", _lib.ptr(x_in), _lib.ptr(F["0.ff"]), _lib.ptr(F["0.leak"]), _lib.ptr(F["0.leak_pt"]),
                      _lib.ptr(F["0.add_pt"]), _lib.ptr(F["0.thresh"]), _lib.ptr(st0[0]) if st0 else None, _lib.ptr(st0[1]) if st0 else None,
                      _lib.ptr(st0[3]) if st0 else None, B, 2, H, W, hard, _lib.ptr(outs[0][0]), _lib.ptr(outs[0][1]), _lib.ptr(outs[0][2]),
                      _lib.ptr(outs[0][3]), _lib.ptr(outs[0][4]))
        else:
            _lib.call("evf_head_lif_fwd", _lib.ptr(x_in), _lib.ptr(F["0.ff"]), _lib.ptr(F["0.leak"]), _lib.ptr(F["0.thresh"]),
                      _lib.ptr(st0[0]) if st0 else None, _lib.ptr(st0[1]) if st0 else None, B, 2, H, W, hard, _lib.ptr(outs[0][0]),
                      _lib.ptr(outs[0][1]), _lib.ptr(outs[0][2]))
        assert _lib.raw("evf_fwd_defer_begin") == 0
        try:
            for i in range(1, 7):
                assert _lib.raw("evf_fwd_defer_slot", 0) == 0  # all six cells of the pass are independent here: ONE index
                c = eng.cells[i]
                sp = pv[i]
                wrec = PK[(i, "rec", "b3")] if c.recurrent else None
                in_bits = inp[i - 1][1]  # the ORACLE's spikes of the layer below
                v_out, z_out, zT_out, pt_out, P_out = outs[i]
                if plif:
                    args = (_lib.ptr(in_bits), _lib.ptr(PK[(i, "ff", "b3")]), _lib.ptr(wrec), _lib.ptr(F[f"{i}.leak"]),
                            _lib.ptr(F[f"{i}.leak_pt"]), _lib.ptr(F[f"{i}.add_pt"]), _lib.ptr(F[f"{i}.thresh"]),
                            _lib.ptr(sp[0]) if sp else None, _lib.ptr(sp[1]) if sp else None, _lib.ptr(sp[3]) if sp else None, B, H, W,
                            hard, _lib.ptr(v_out), _lib.ptr(z_out), _lib.ptr(zT_out), _lib.ptr(pt_out), _lib.ptr(P_out))
                    if i == 6:
                        _lib.call("evf_conv_plif_fwd_b3_pred", *args, _lib.ptr(F["pred.w"]), _lib.ptr(F["pred.b"]), _lib.ptr(flow))
                    else:
                        _lib.call("evf_conv_plif_fwd_b3", *args)
                else:
                    args = (_lib.ptr(in_bits), _lib.ptr(PK[(i, "ff", "b3")]), _lib.ptr(wrec), _lib.ptr(F[f"{i}.leak"]),
                            _lib.ptr(F[f"{i}.thresh"]), _lib.ptr(sp[0]) if sp else None, _lib.ptr(sp[1]) if sp else None, B, H, W, hard,
                            _lib.ptr(v_out), _lib.ptr(z_out), _lib.ptr(zT_out))
                    if i == 6:
                        _lib.call("evf_conv_lif_fwd_b3_pred", *args, _lib.ptr(F["pred.w"]), _lib.ptr(F["pred.b"]), _lib.ptr(flow))
                    else:
                        _lib.call("evf_conv_lif_fwd_b3", *args)
            assert _lib.raw("evf_fwd_defer_pending") > 0  # (the cells were RECORDED, not launched)
        finally:
            _lib.call("evf_fwd_defer_flush")
        got = _read_state(eng, [(o[0], o[1], o[2], o[3]) if plif else (o[0], o[1], o[2]) for o in outs])
        for i in range(7):
            vo, zo = N(ost[i][0]), N(ost[i][1])
            vh, zh = got[i][0], got[i][1]
            th = np.maximum(N(params[f"{LAYERS[i]}.thresh"]).reshape(1, -1, 1, 1), 0.01)
            scale = max(float(np.abs(vo).max()), 1e-20)
            e_l2 = float(np.linalg.norm(vh - vo) / max(np.linalg.norm(vo), 1e-20))
            e_max = float(np.abs(vh - vo).max() / scale)
            worst_v = max(worst_v, e_l2, e_max)
            assert e_l2 <= 1e-5 and e_max <= 1e-5, (t, i, e_l2, e_max)
            away = np.abs(vo - th) > EPS_THRESH * np.maximum(1.0, np.abs(th))
            diff = zh != zo
            flips_outside += int((diff & away).sum())
            band_mismatch += int((diff & ~away).sum())
            assert not (diff & away).any(), (t, i, int((diff & away).sum()))
            if plif:
                po = N(ost[i][2])
                e_pt = float(np.abs(got[i][2] - po).max() / max(float(np.abs(po).max()), 1e-20))
                assert e_pt <= 1e-5, (t, i, e_pt)
        # the flow of a pass is tanh(1x1 conv) of the TOP layer's spikes of that pass: a pixel whose top-layer spike vector differs
        # inside the tolerated band (counted above) carries another flow by construction -- compared everywhere else
        fo = N(ora["flows"][t])
        same_top = ~(got[6][1] != N(ost[6][1])).any(axis=1)  # [B,H,W]
        keep = np.broadcast_to(same_top[:, None], fo.shape)
        assert keep.mean() > 0.999, keep.mean()
        e_f = float(np.linalg.norm((N(flow) - fo)[keep]) / max(np.linalg.norm(fo[keep]), 1e-20))
        worst_flow = max(worst_flow, e_f)
        assert e_f <= 1e-5, (t, e_f)
        hip_prev = inp  # the next pass starts from the ORACLE's state of this one
    print(f"[teacher-forced forward] {P} passes x 7 layers: worst v' error {worst_v:.2e}, worst flow rel-L2 {worst_flow:.2e}, "
          f"spike mismatches inside the |v'-thresh| <= {EPS_THRESH:g} band {band_mismatch}, outside {flips_outside}")

    # ---------------- backward: the HIP window free-running, the ORACLE's tape loaded into it, backward from the oracle's dL/dflow
    tapes = []
    orig = eng._forward_pass

    def wrapped(x, st, record):
        out = orig(x, st, record)
        if record:
            tapes.append(out[1])
        return out

    eng._forward_pass = wrapped
    model.reset_states()
    flows = []
    model.defer_forward(True)
    try:
        for k, d in enumerate(passes):
            flows.append(model(d["event_voxel"], d["event_cnt"])["flow"][0])
    finally:
        model.defer_forward(False)  # (launches what was recorded)
        eng._forward_pass = orig
    assert len(tapes) == P
    for t in range(P):
        src = _hip_state(eng, ora["states"][t])
        for i in range(7):
            lay = tapes[t]["layers"][i]  # (in_bits, v_prev, z_prev, v_out, z_out, in_bitsT, zT_prev, pt_prev, pt_out, P_out)
            lay[3].copy_(src[i][0])
            lay[4].copy_(src[i][1])
            # the bit planes of this layer's output: the NEXT layer's in_bitsT and this layer's zT_prev of the next pass alias them
            zT = tapes[t]["layers"][i + 1][5] if i + 1 < 7 else None
            if zT is not None:
                zT.copy_(src[i][2])
            if t + 1 < P and tapes[t + 1]["layers"][i][6] is not None:
                tapes[t + 1]["layers"][i][6].copy_(src[i][2])
            if plif:
                lay[8].copy_(src[i][3])
                xin = opasses[t]["event_cnt"] if i == 0 else ora["states"][t][i - 1][1]
                lay[9].copy_(osnn._pretrace(xin, 3, 1)[:, 0].to(DEV))  # pooled pre-synaptic activity of the ORACLE's input
        tapes[t]["flow"].data.copy_(ora["flows"][t].to(DEV))  # (the tape's flow map IS the node's output: .data has its own version counter)
        # aliasing the engine relies on (a pass's previous state IS the previous pass's output): checked, not assumed
        if t > 0:
            for i in range(7):
                assert tapes[t]["layers"][i][1].data_ptr() == tapes[t - 1]["layers"][i][3].data_ptr()
                assert tapes[t]["layers"][i][2].data_ptr() == tapes[t - 1]["layers"][i][4].data_ptr()
                if i > 0:
                    assert tapes[t]["layers"][i][0].data_ptr() == tapes[t]["layers"][i - 1][4].data_ptr()
    opt.mark_grad_dirty()
    model.defer_backward(True)
    try:
        torch.autograd.backward(flows, [g.to(DEV) for g in ora["gflows"]])
    finally:
        model.defer_backward(False)
    torch.cuda.synchronize()
    hip_grads = {k: N(p.grad).copy() for k, p in model.named_parameters() if p.requires_grad}
    num = den = 0.0
    worst = ("", 0.0)
    gn_all = np.sqrt(sum(float((ora["grads"][k].numpy() ** 2).sum()) for k in keys))
    for k in keys:
        ref, got = ora["grads"][k].numpy(), hip_grads[k]
        e, d = float(((got - ref) ** 2).sum()), float((ref ** 2).sum())
        num, den = num + e, den + d
        r = np.sqrt(e) / max(np.sqrt(d), 1e-20)
        if r > worst[1]:
            worst = (k, r)
        # every tensor: 1e-4 of its own norm (measured worst 1.4e-5; + 1e-6 of the whole gradient for tensors that are round-off
        # of cancelling sums)
        assert np.sqrt(e) <= 1e-4 * np.sqrt(d) + 1e-6 * gn_all, (k, r)
    grel = np.sqrt(num / den)
    print(f"[teacher-forced backward] gradient rel-L2 {grel:.3e} (|g| = {np.sqrt(den):.4e}); worst tensor {worst[0]} {worst[1]:.3e}")
    assert grel <= 1e-5, grel  # (measured 4.9e-7 .. 5.9e-7: two fp32 summation orders of the same spike trains)
    # clip + Adam (train_flow.py:157-163) on both sides from the same parameters
    old = N(opt.flat_param).copy()
    opt.step()
    torch.cuda.synchronize()
    upd = N(opt.flat_param) - old
    cl, _ = otrain.clip_grad_norm([ora["grads"][k] for k in keys], 100.0)
    names = [k for k, p in model.named_parameters() if p.requires_grad]
    ref_upd, gref = [], []
    for k in names:
        g = cl[keys.index(k)]
        newp, _, _ = otrain.adam_step(params[k], g, torch.zeros_like(g), torch.zeros_like(g), 1, 2e-4)
        ref_upd.append((newp - params[k]).reshape(-1).numpy())
        gref.append(ora["grads"][k].reshape(-1).numpy())
    ref_upd, gref = np.concatenate(ref_upd), np.concatenate(gref)
    sig = np.abs(gref) > 1e-3 * np.abs(gref).max()
    rel_sig = float(np.linalg.norm((upd - ref_upd)[sig]) / np.linalg.norm(ref_upd[sig]))
    print(f"[teacher-forced backward] clip+Adam update on the {int(sig.sum())} of {sig.size} weights with signal: rel-L2 {rel_sig:.3e}")
    assert sig.sum() > 0.05 * sig.size and rel_sig <= 1e-6, (int(sig.sum()), rel_sig)  # (measured 4e-8)
    return {"v": worst_v, "flow": worst_flow, "band": band_mismatch, "grad_rel": grel, "worst_tensor": worst, "update_rel_sig": rel_sig,
            "nsig": int(sig.sum()), "rates": rates}


LIF_CFG = {"name": "LIFFireNet", "encoding": "cnt", "round_encoding": False, "norm_input": False, "num_bins": 2, "base_num_channels": 32,
           "kernel_size": 3, "activations": ["arctanspike", "arctanspike"], "mask_output": True,
           "spiking_neuron": {"leak": [-4.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True, "learn_thresh": True, "hard_reset": True}}
PLIF_CFG = dict(LIF_CFG, name="PLIFFireNet",
                spiking_neuron={"leak_v": [-4.0, 0.1], "leak_pt": [-4.0, 0.1], "add_pt": [-2.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True,
                                "learn_thresh": True, "hard_reset": True})


@pytest.mark.parametrize("thresh_scale,kind", [(0.25, "uniform"), (0.5, "moving_dots")])
def test_teacher_forced_c3_lif_firenet_alive_at_benched_size(thresh_scale, kind):
    """BASELINE configs[2] per-GPU shard: B = 8, 128 x 128, 10 passes x 1500 events, thresholds x 0.25 (uniform events) and x 0.5
    (moving dots): every layer spikes, most weights carry gradient signal."""
    _teacher_forced(LIFFireNet, "LIFFireNet", LIF_CFG, 128, 128, 8, 10, 1500, thresh_scale, kind, seed=0)


def test_teacher_forced_c5_plif_firenet_alive_at_per_gpu_batch():
    """BASELINE configs[4] per-GPU shard: PLIF-FireNet, 260 x 346, B = 4, 10 passes x 1500 events, thresholds x 0.25."""
    _teacher_forced(PLIFFireNet, "PLIFFireNet", PLIF_CFG, 260, 346, 4, 10, 1500, 0.25, "uniform", seed=1)
