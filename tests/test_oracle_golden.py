"""Pins the CPU oracle (oracle/) against fixtures produced by RUNNING THE
REFERENCE (tools/gen_golden.py).  Integer / index outputs must be bit-exact;
floating point within the stated tolerance."""

import numpy as np
import pytest
import torch

from conftest import golden_cases, load_golden
from oracle import encodings as oenc
from oracle import iwe as oiwe
from oracle import loss as oloss
from oracle import snn as osnn
from oracle import train as otrain

torch.set_num_threads(4)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


# --------------------------------------------------------------------- G1
def test_g1_encodings_bit_exact():
    g = load_golden("g1_encodings")
    xs, ys, ts, ps = g["xs"], g["ys"], g["ts"], g["ps"]
    res = tuple(g["sensor"])
    assert np.array_equal(oenc.events_to_channels(xs, ys, ps, res), g["cnt"])
    assert np.array_equal(oenc.create_mask_encoding(xs, ys, ps, res)[0], g["mask"])
    assert np.array_equal(oenc.events_to_image(xs, ys, ps, res), g["image_acc"])
    for nb in (2, 5):
        for rnd in (0, 1):
            got = oenc.events_to_voxel(xs, ys, ts, ps, nb, res, round_ts=bool(rnd))
            assert np.array_equal(got, g[f"voxel_nb{nb}_r{rnd}"]), (nb, rnd)
    assert np.array_equal(oenc.create_list_encoding(xs, ys, ts, ps), g["list"])
    assert np.array_equal(oenc.create_polarity_mask(ps), g["polmask"])
    _, _, ft, fp = oenc.event_formatting(xs[:64], ys[:64], g["raw_t"], g["raw_p"])
    assert np.array_equal(ft, g["fmt_t"]) and np.array_equal(fp, g["fmt_p"])


def test_g1_collate_layout():
    g = load_golden("g1_encodings")
    assert g["collate_event_list"].shape == (2, 50, 4)
    assert g["collate_event_list_pol_mask"].shape == (2, 50, 2)
    assert g["collate_event_cnt"].shape == (2, 2, 40, 48)
    assert g["collate_event_mask"].shape == (2, 1, 40, 48)
    samples = []
    for b in range(2):
        ev = g["collate_event_list"][b]
        samples.append(oenc.encode_window(ev[:, 2], ev[:, 1], ev[:, 0], ev[:, 3], 2, (40, 48)))
    col = oenc.collate(samples)
    for k in col:
        assert np.array_equal(col[k], g["collate_" + k]), k


# --------------------------------------------------------------------- G2
@pytest.mark.parametrize("tref", [1, 3, 0])
@pytest.mark.parametrize("rnd", [0, 1])
@pytest.mark.parametrize("S", [16, 128])
def test_g2_get_interpolation_bit_exact(tref, rnd, S):
    g = load_golden("g2_interpolation")
    idx, w = oiwe.get_interpolation(g["events"], g["flow"], tref, tuple(g["res"]), S, round_idx=bool(rnd))
    assert np.array_equal(idx, g[f"idx_t{tref}_r{rnd}_s{S}"])
    assert np.array_equal(w, g[f"w_t{tref}_r{rnd}_s{S}"])


# --------------------------------------------------------------------- G3
@pytest.mark.parametrize("tag", ["c1", "b2"])
@pytest.mark.parametrize("S", [128, 32])
def test_g3_pol_iwe(tag, S):
    g = load_golden("g3_pol_iwe")
    ev, flow, pol, res = g[tag + "_events"], g[tag + "_flow"], g[tag + "_pol"], tuple(g[tag + "_res"])
    got = oiwe.compute_pol_iwe(flow, ev, res, pol[:, :, 0:1], pol[:, :, 1:2], flow_scaling=S, round_idx=True)
    ref = g[f"{tag}_iwe_s{S}_r1"]
    assert np.array_equal(got, ref)  # integer histogram: bit exact
    assert np.array_equal(ref, np.rint(ref))
    got = oiwe.compute_pol_iwe(flow, ev, res, pol[:, :, 0:1], pol[:, :, 1:2], flow_scaling=S, round_idx=False)
    np.testing.assert_allclose(got, g[f"{tag}_iwe_s{S}_r0"], rtol=0, atol=2e-6)


# --------------------------------------------------------------------- G4
def _g4_window(g, c, requires_grad=True):
    tag = c["tag"]
    res = tuple(g["res"])
    win = oloss.Window(res)
    flows = []
    for k in range(c["P"]):
        fl = [T(g[f"{tag}_p{k}_flow{s}"]).requires_grad_(requires_grad) for s in range(c["scales"])]
        flows.append(fl)
        win.add(fl, T(g[f"{tag}_p{k}_event_list"]), T(g[f"{tag}_p{k}_event_list_pol_mask"]), T(g[f"{tag}_p{k}_event_mask"]))
    if c["overwrite"]:
        win.overwrite(flows[-1])
    return win, flows


def test_g4_event_warping_loss_and_grad():
    g = load_golden("g4_event_warping")
    cases = golden_cases(g)
    assert len(cases) >= 20
    for c in cases:
        win, flows = _g4_window(g, c)
        val = oloss.event_warping_loss(win, max(win.res), 0.001, smoothing_mask=c["mask"], overwrite=c["overwrite"])
        np.testing.assert_allclose(float(val.detach()), float(g[c["tag"] + "_loss"]), rtol=2e-6, err_msg=str(c))
        leaves = [f for fl in flows for f in fl]
        grads = torch.autograd.grad(val, leaves, allow_unused=True)
        gi = 0
        for k in range(c["P"]):
            for s in range(c["scales"]):
                ref = g[f"{c['tag']}_p{k}_gflow{s}"]
                got = grads[gi].numpy() if grads[gi] is not None else np.zeros_like(ref)
                scale = max(np.abs(ref).max(), 1e-12)
                assert np.abs(got - ref).max() <= 2e-5 * scale + 1e-9, (c, k, s)
                gi += 1


# --------------------------------------------------------------------- G5
@pytest.mark.parametrize("ow", [0, 1])
def test_g5_metrics(ow):
    g = load_golden("g5_metrics")
    res, P = tuple(g["res"]), int(g["P"])
    tag = f"ow{ow}"
    win = oloss.Window(res)
    last = None
    for k in range(P):
        last = T(g[f"{tag}_p{k}_flow"])
        win.add([last], T(g[f"{tag}_p{k}_event_list"]), T(g[f"{tag}_p{k}_event_list_pol_mask"]), T(g[f"{tag}_p{k}_event_mask"]))
    we = oloss.window_events(win)
    if ow:
        win.overwrite([last])
    np.testing.assert_allclose(oloss.fwl(win, 32).numpy(), g[tag + "_fwl"], rtol=1e-5)
    np.testing.assert_allclose(oloss.rsat(win, 32).numpy(), g[tag + "_rsat"], rtol=1e-5)
    assert np.array_equal(we.numpy(), g[tag + "_window_events"])
    assert np.array_equal(oloss.window_iwe(win, 32).numpy(), g[tag + "_window_iwe"])
    np.testing.assert_allclose(oloss.masked_window_flow(win, bool(ow)).numpy(), g[tag + "_masked_flow"], rtol=1e-6, atol=1e-7)


def test_g5_aee():
    g = load_golden("g5_metrics")
    a, p = oloss.aee(T(g["aee_flow"]), T(g["aee_gt"]), T(g["aee_event_mask"])[:, -1], 32, g["aee_dt"][0:1], g["aee_dt"][1:2])
    np.testing.assert_allclose(a.numpy(), g["aee_val"], rtol=1e-5)
    np.testing.assert_allclose(p.numpy(), g["aee_outl"], rtol=1e-5)


# --------------------------------------------------------------------- G6
@pytest.mark.parametrize("fix,n", [("g6_cells", 28), ("g15_cells_k5", 8), ("g18_cells_weightnorm", 3), ("g19_cells_groupnorm", 3)])
def test_g6_cells_forward_backward(fix, n):
    """G6: 3x3 cells, all kinds / resets / surrogates; G15: 5x5 and 7x7 kernels (models/unet.py:51 defaults to 5), stride 1 / 2;
    G18 / G19: LIF cells with norm="weight" (nn.utils.weight_norm on ff / rec, spiking_submodules.py:87-88, :502-504) and
    norm="group" (nn.GroupNorm(1, C) on the input / the previous spikes, :90-99, :507-529)."""
    g = load_golden(fix)
    cases = golden_cases(g)
    assert len(cases) == n
    for c in cases:
        tag = c["tag"]
        pre = "c."
        p = {}
        for k in g.files:
            if k.startswith(tag + "_param_"):
                p[pre + k[len(tag + "_param_"):]] = T(g[k]).clone()
        names = [k for k in p if not k.endswith("act_width")]
        for k in names:
            p[k].requires_grad_(True)
        x = T(g[tag + "_x"]).requires_grad_(True)
        st = T(g[tag + "_state"]).requires_grad_(True)
        out, new = osnn.cell_step(c["kind"], p, pre, x, tuple(st.unbind(0)), recurrent=c["recurrent"], act=c["act"], hard_reset=c["hard_reset"],
                                  stride=c.get("stride", 1))
        new = torch.stack(new)
        assert np.array_equal(out.detach().numpy(), g[tag + "_out"]), c  # spikes: exact
        np.testing.assert_allclose(new.detach().numpy(), g[tag + "_new"], rtol=1e-6, atol=1e-7, err_msg=str(c))
        grads = torch.autograd.grad([out, new], [x, st] + [p[k] for k in names], [T(g[tag + "_g_out"]), T(g[tag + "_g_new"])], allow_unused=True)
        np.testing.assert_allclose(grads[0].numpy(), g[tag + "_gx"], rtol=1e-4, atol=1e-6, err_msg=str(c))
        np.testing.assert_allclose(grads[1].numpy(), g[tag + "_gstate"], rtol=1e-4, atol=1e-6, err_msg=str(c))
        for k, gr in zip(names, grads[2:]):
            ref = g[f"{tag}_grad_{k[len(pre):]}"]
            got = gr.numpy() if gr is not None else np.zeros_like(ref)
            np.testing.assert_allclose(got, ref, rtol=2e-4, atol=1e-5 * max(1.0, np.abs(ref).max()), err_msg=f"{c} {k}")


# --------------------------------------------------------------------- G7
def _g7_passes(g):
    P = int(g["meta_P"])
    return [
        {k: T(g[f"p{i}_{k}"]) for k in ("event_cnt", "event_voxel", "event_list", "event_list_pol_mask", "event_mask")}
        for i in range(P)
    ]


@pytest.mark.parametrize("fix,name", [("g7_liffirenet_train", "LIFFireNet"), ("g7_liffirenet_lowthresh", "LIFFireNet"), ("g7_pliffirenet_train", "PLIFFireNet")])
def test_g7_firenet_train_step(fix, name):
    g = load_golden(fix)
    params = {k[len("param0_"):]: T(g[k]).clone() for k in g.files if k.startswith("param0_")}
    keys = osnn.trainable_keys(params)
    passes = _g7_passes(g)
    res = passes[0]["event_cnt"].shape[2:]
    # forward per-layer parity (teacher-forced by construction: same inputs, same state)
    states = [None] * 7
    with torch.no_grad():
        for i, d in enumerate(passes):
            col = {}
            flow, states = osnn.firenet_forward(name, params, d["event_cnt"], states, collect=col)
            for ln in osnn.FIRENET_LAYERS:
                v_ref, z_ref = g[f"p{i}_v_{ln}"], g[f"p{i}_z_{ln}"]
                v, z = col[ln][1][0].numpy(), col[ln][1][1].numpy()
                margin = np.abs(v_ref - params[ln + ".thresh"].clamp_min(0.01).numpy()[None]) > 1e-5
                assert np.array_equal(z[margin], z_ref[margin].astype(np.float32)), (i, ln)
                np.testing.assert_allclose(v, v_ref, rtol=1e-5, atol=1e-6, err_msg=f"{i} {ln}")
            np.testing.assert_allclose(flow.numpy(), g[f"p{i}_flow"], rtol=1e-5, atol=1e-7)
    opt = {"step": 0, "m": {}, "v": {}}
    loss, grads, newp, _ = otrain.train_step(
        name, params, keys, passes, [None] * 7, tuple(res), opt,
        loss_cfg={"flow_regul_weight": 0.001, "mask_output": True}, lr=2e-4, clip=100.0,
    )
    np.testing.assert_allclose(loss, float(g["loss"]), rtol=1e-5)
    gn = np.sqrt(sum(float((grads[k].double() ** 2).sum()) for k in keys))
    np.testing.assert_allclose(gn, float(g["grad_norm"]), rtol=1e-4)
    for k in keys:
        ref = g["grad_" + k]
        tol = 1e-4 * max(np.abs(ref).max(), 1e-8)
        assert np.abs(grads[k].numpy() - ref).max() <= tol + 1e-9, k
        np.testing.assert_allclose(newp[k].numpy(), g["param1_" + k], rtol=1e-5, atol=1e-7, err_msg=k)


# --------------------------------------------------------------------- G8
def test_g8_firenet_ann_forward():
    g = load_golden("g8_firenet_ann")
    params = {k[len("param_"):]: T(g[k]) for k in g.files if k.startswith("param_")}
    states = [None] * 7
    with torch.no_grad():
        for i in range(2):
            flow, states = osnn.firenet_forward("FireNet", params, T(g[f"p{i}_event_voxel"]), states)
            np.testing.assert_allclose(flow.numpy(), g[f"p{i}_flow"], rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(states[1].numpy(), g[f"p{i}_state_G1"], rtol=1e-4, atol=1e-6)


# --------------------------------------------------------------------- G9
def test_g9_spiking_unet():
    g = load_golden("g9_spiking_unet")
    params = {k[len("param_"):]: T(g[k]).clone() for k in g.files if k.startswith("param_")}
    keys = osnn.trainable_keys(params)
    for k in keys:
        params[k].requires_grad_(True)
    states = [None] * 10
    for i in range(2):
        flows, states = osnn.spiking_unet_forward("lif", params, T(g[f"p{i}_event_cnt"]), states)
        assert len(flows) == 4
        for s, f in enumerate(flows):
            np.testing.assert_allclose(f.detach().numpy(), g[f"p{i}_flow{s}"], rtol=1e-4, atol=1e-6)
    tot = sum(f.pow(2).sum() for f in flows)
    grads = torch.autograd.grad(tot, [params[k] for k in keys], allow_unused=True)
    for k, gr in zip(keys, grads):
        ref = g["grad_" + k]
        got = gr.numpy() if gr is not None else np.zeros_like(ref)
        assert np.abs(got - ref).max() <= 2e-4 * max(np.abs(ref).max(), 1e-10) + 1e-12, k


# --------------------------------------------------------------------- G10
ANN_ACTS = {"FireFlowNet": ("relu", "relu"), "RNNFireNet": ("relu", None), "LeakyFireNet": ("relu", None),
            "LeakyFireFlowNet": ("relu", "tanh")}


@pytest.mark.parametrize("name", sorted(ANN_ACTS))
def test_g10_ann_firenets(name):
    """FireFlowNet / RNNFireNet / LeakyFireNet / LeakyFireFlowNet (reference models/model.py:398-409,614-633,696-704):
    flows of three passes, final states, BPTT parameter gradients."""
    g = load_golden("g10_ann_firenets")
    pre = name + ".param_"
    params = {k[len(pre):]: T(g[k]).clone().requires_grad_(True) for k in g.files if k.startswith(pre)}
    states = [None] * 7
    tot = 0
    for i in range(3):
        flow, states = osnn.firenet_forward(name, params, T(g[f"p{i}_event_cnt"]), states, acts=ANN_ACTS[name])
        np.testing.assert_allclose(flow.detach().numpy(), g[f"{name}.p{i}_flow"], rtol=1e-4, atol=1e-6)
        tot = tot + flow.pow(2).sum() + flow.sum()
    for li, st in enumerate(states):
        key = f"{name}.state{li}"
        assert (key in g.files) == (st is not None)
        if st is not None:
            np.testing.assert_allclose(st.detach().numpy(), g[key], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(float(tot.detach()), float(g[f"{name}.loss"]), rtol=1e-5)
    keys = sorted(params)
    grads = torch.autograd.grad(tot, [params[k] for k in keys], allow_unused=True)
    for k, gr in zip(keys, grads):
        ref = g[f"{name}.grad_{k}"]
        got = gr.numpy() if gr is not None else np.zeros_like(ref)
        assert np.abs(got - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-10) + 1e-9, k


# --------------------------------------------------------------------- G11
ANN_UNET_STATES = {"EVFlowNet": 0, "RecEVFlowNet": 4, "RNNRecEVFlowNet": 4, "LeakyRecEVFlowNet": 10}


@pytest.mark.parametrize("name", sorted(ANN_UNET_STATES))
def test_g11_ann_unets(name):
    """EVFlowNet / RecEVFlowNet (ConvGRU) / RNNRecEVFlowNet / LeakyRecEVFlowNet (reference models/model.py:289-395,
    412-547, 594-611): 4 flow scales of two passes, final states, BPTT parameter gradients."""
    g = load_golden("g11_ann_unets")
    pre = name + ".param_"
    params = {k[len(pre):]: T(g[k]).clone().requires_grad_(True) for k in g.files if k.startswith(pre)}
    states = [None] * ANN_UNET_STATES[name]
    tot = 0
    for i in range(2):
        flows, states = osnn.ann_unet_forward(name, params, T(g[f"p{i}_event_cnt"]), states)
        assert len(flows) == 4
        for s_, f in enumerate(flows):
            np.testing.assert_allclose(f.detach().numpy(), g[f"{name}.p{i}_flow{s_}"], rtol=1e-4, atol=1e-6)
            tot = tot + f.pow(2).sum() + f.sum()
    for si, st in enumerate(states):
        st = torch.stack(st) if isinstance(st, tuple) else st
        np.testing.assert_allclose(st.detach().numpy(), g[f"{name}.state{si}"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(float(tot.detach()), float(g[f"{name}.loss"]), rtol=1e-5)
    keys = sorted(params)
    grads = torch.autograd.grad(tot, [params[k] for k in keys], allow_unused=True)
    for k, gr in zip(keys, grads):
        ref = g[f"{name}.grad_{k}"]
        got = gr.numpy() if gr is not None else np.zeros_like(ref)
        assert np.abs(got - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-10) + 1e-9, k


# --------------------------------------------------------------------- G12
def test_g12_e2vid():
    """E2VID (reference models/model.py:29-145): flows of three passes, final (hidden, cell) states, BPTT gradients."""
    g = load_golden("g12_e2vid")
    params = {k[len("param_"):]: T(g[k]).clone().requires_grad_(True) for k in g.files if k.startswith("param_")}
    states = [None] * 3
    tot = 0
    for i in range(3):
        flow, states = osnn.e2vid_forward(params, T(g[f"p{i}_event_cnt"]), states)
        np.testing.assert_allclose(flow.detach().numpy(), g[f"p{i}_flow"], rtol=1e-4, atol=1e-6)
        tot = tot + flow.pow(2).sum() + flow.sum()
    for si, (h, c) in enumerate(states):
        np.testing.assert_allclose(h.detach().numpy(), g[f"state{si}_hidden"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(c.detach().numpy(), g[f"state{si}_cell"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(float(tot.detach()), float(g["loss"]), rtol=1e-5)
    keys = sorted(params)
    grads = torch.autograd.grad(tot, [params[k] for k in keys], allow_unused=True)
    for k, gr in zip(keys, grads):
        ref = g["grad_" + k]
        got = gr.numpy() if gr is not None else np.zeros_like(ref)
        assert np.abs(got - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-10) + 1e-9, k


# --------------------------------------------------------------------- G16 (BN / IN layers)
def test_g16_norm_oracle_vs_reference():
    """oracle.norm.norm2d against the reference's ConvLayer(norm='BN'/'IN') outputs (conv by torch-CPU): both training calls,
    the running statistics they leave, and the eval-mode call."""
    import torch.nn.functional as F

    from oracle import norm as onorm

    g = load_golden("g16_norm_layers")
    for c in golden_cases(g):
        if c["cls"] != "ConvLayer":
            continue
        tag, kw = c["tag"], c["kwargs"]
        inst = kw["norm"] == "IN"
        w = T(g[f"{tag}_param0_conv2d.weight"])
        b = T(g[f"{tag}_param0_conv2d.bias"]) if f"{tag}_param0_conv2d.bias" in g.files else None
        nw = g[f"{tag}_param0_norm_layer.weight"] if f"{tag}_param0_norm_layer.weight" in g.files else None
        nb = g[f"{tag}_param0_norm_layer.bias"] if f"{tag}_param0_norm_layer.bias" in g.files else None
        rm, rv = g[f"{tag}_param0_norm_layer.running_mean"], g[f"{tag}_param0_norm_layer.running_var"]
        act = {"relu": lambda v: np.maximum(v, 0), "tanh": np.tanh}[kw["activation"]]
        for k in range(2):
            pre = F.conv2d(T(g[f"{tag}_x{k}"]), w, b, stride=kw.get("stride", 1), padding=kw["kernel_size"] // 2).detach().numpy()
            y, rm, rv = onorm.norm2d(pre, nw, nb, rm, rv, instance=inst, training=True, momentum=kw.get("BN_momentum", 0.1))
            np.testing.assert_allclose(act(y), g[f"{tag}_y{k}_0"], rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(rm, g[f"{tag}_param1_norm_layer.running_mean"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(rv, g[f"{tag}_param1_norm_layer.running_var"], rtol=1e-5, atol=1e-7)
        pre = F.conv2d(T(g[f"{tag}_x0"]), w, b, stride=kw.get("stride", 1), padding=kw["kernel_size"] // 2).detach().numpy()
        y, _, _ = onorm.norm2d(pre, nw, nb, rm, rv, instance=inst, training=False)
        np.testing.assert_allclose(act(y), g[f"{tag}_yeval_0"], rtol=1e-4, atol=2e-6)
