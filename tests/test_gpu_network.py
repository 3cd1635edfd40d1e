"""GPU parity of the spiking network path (fused conv+LIF forward, BPTT
backward, clip+Adam) against the reference-generated golden fixtures (G7) and
the CPU oracle.  Layered protocol of SURVEY.md section 7:
  (a) per-layer: v' within 1e-5, spikes exact wherever |v'-thresh| > margin;
  (b) end-to-end: flow / loss / gradient aggregates;
  (c) optimizer step."""

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

from event_flow_amd import _lib  # noqa: E402
from event_flow_amd.loss import flow as hloss  # noqa: E402
from event_flow_amd.models.model import LIFFireFlowNet, LIFFireNet, PLIFFireNet  # noqa: E402
from event_flow_amd.train import FlatAdam, train_window  # noqa: E402
from oracle import snn as osnn  # noqa: E402
from oracle import train as otrain  # noqa: E402

DEV = "cuda:0"
LAYERS = ["head", "G1", "R1a", "R1b", "G2", "R2a", "R2b"]
NEURON = {"leak": [-4.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True, "learn_thresh": True, "hard_reset": True}
PLIF_NEURON = {"leak_v": [-4.0, 0.1], "leak_pt": [-4.0, 0.1], "add_pt": [-2.0, 0.1], "thresh": [0.8, 0.1],
               "learn_leak": True, "learn_thresh": True, "hard_reset": True}
FIXTURES = {"g7_liffirenet_train": (LIFFireNet, NEURON), "g7_liffirenet_lowthresh": (LIFFireNet, NEURON),
            "g7_pliffirenet_train": (PLIFFireNet, PLIF_NEURON)}


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def N(t):
    return t.detach().cpu().numpy()


def model_cfg(neuron=NEURON):
    return {"num_bins": 2, "base_num_channels": 32, "kernel_size": 3, "encoding": "cnt", "norm_input": False,
            "mask_output": True, "activations": ["arctanspike", "arctanspike"], "spiking_neuron": dict(neuron)}


def loss_cfg(H, W):
    return {"loader": {"resolution": [H, W]}, "loss": {"flow_regul_weight": 0.001, "overwrite_intermediate": False},
            "model": {"mask_output": True}}


def build_from_golden(g, prefix="param0_", fix="g7_liffirenet_train"):
    cls, neuron = FIXTURES[fix]
    model = cls(model_cfg(neuron)).to(DEV)
    sd = {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}
    model.load_state_dict(sd)  # reference state_dict keys load unchanged
    return model


def passes_from_golden(g):
    P = int(g["meta_P"])
    keys = ("event_cnt", "event_voxel", "event_list", "event_list_pol_mask", "event_mask")
    return [{k: G(g[f"p{i}_{k}"]) for k in keys} for i in range(P)]


@pytest.mark.parametrize("fix", list(FIXTURES))
def test_forward_per_layer_and_flow(fix):
    g = load_golden(fix)
    model = build_from_golden(g, fix=fix)
    model.eval()
    passes = passes_from_golden(g)
    nflip = ntot = 0
    with torch.no_grad():
        for i, d in enumerate(passes):
            out = model(d["event_voxel"], d["event_cnt"], log=True)
            states = model.states
            for li, ln in enumerate(LAYERS):
                v_ref, z_ref = g[f"p{i}_v_{ln}"], g[f"p{i}_z_{ln}"].astype(np.float32)
                v, z = N(states[li][0]), N(states[li][1])
                th = np.maximum(g[f"param0_{ln}.thresh"], 0.01)[None]
                safe = np.abs(v_ref - th) > 1e-4
                nflip += int((z != z_ref).sum())
                ntot += z.size
                assert np.array_equal(z[safe], z_ref[safe]), (i, ln)
                # membrane potential: compare where the inputs were identical (a flipped
                # borderline spike upstream legitimately changes downstream v)
                if nflip == 0:
                    np.testing.assert_allclose(v, v_ref, rtol=1e-5, atol=2e-6, err_msg=f"{i} {ln}")
                    if f"p{i}_aux_{ln}" in g.files:  # PLIF pre-synaptic trace (third state)
                        np.testing.assert_allclose(N(states[li][2]), g[f"p{i}_aux_{ln}"], rtol=1e-5, atol=1e-7,
                                                   err_msg=f"{i} {ln} trace")
            if nflip == 0:
                np.testing.assert_allclose(N(out["flow"][0]), g[f"p{i}_flow"], rtol=1e-4, atol=1e-7)
            assert set(out["activity"].keys()) == {"0:input", "1:head", "2:G1", "3:R1a", "4:R1b", "5:G2", "6:R2a", "7:R2b", "8:pred"}
    assert nflip <= 1e-5 * ntot, (nflip, ntot)


def _train_once(g, use_flat_adam, fix="g7_liffirenet_train", precision=None):
    model = build_from_golden(g, fix=fix)
    if precision:
        model.precision = precision
    model.train()
    passes = passes_from_golden(g)
    H, W = passes[0]["event_cnt"].shape[2:]
    lossf = hloss.EventWarping(loss_cfg(H, W), DEV)
    if use_flat_adam:
        opt = FlatAdam(model, lr=2e-4, clip=100.0)
        opt.zero_grad()
    for d in passes:
        out = model(d["event_voxel"], d["event_cnt"])
        lossf.event_flow_association(out["flow"], d["event_list"], d["event_list_pol_mask"], d["event_mask"])
    # spikes of the last pass against the reference's (the free-running bars below are conditioned on this count)
    nflip = sum(int((N(model.states[li][1]) != g[f"p{len(passes) - 1}_z_{ln}"].astype(np.float32)).sum()) for li, ln in enumerate(LAYERS))
    _train_once.last_flips = (nflip, sum(int(model.states[li][1].numel()) for li in range(len(LAYERS))))
    loss = lossf()
    loss.backward()
    grads = {k: N(p.grad).copy() for k, p in model.named_parameters()}
    if use_flat_adam:
        opt.step()
        gn = opt.grad_norm()
    else:
        gn = float(torch.nn.utils.clip_grad_norm_(model.parameters(), 100.0))
        torch.optim.Adam(model.parameters(), lr=2e-4).step()
    newp = {k: N(v).copy() for k, v in model.state_dict().items()}
    return float(loss.detach()), grads, gn, newp


@pytest.mark.parametrize("fix", list(FIXTURES))
@pytest.mark.parametrize("flat", [True, False])
def test_train_step_vs_golden(fix, flat):
    g = load_golden(fix)
    loss, grads, gn, newp = _train_once(g, flat, fix)
    # the bars of a FREE-RUNNING comparison depend on whether a borderline neuron fired differently: the flip
    # count of the last pass is asserted, and with no flip the bars are the round-off ones (10x tighter)
    nflip, ntot = _train_once.last_flips
    assert nflip <= 1e-5 * ntot, (nflip, ntot)
    tight = nflip == 0
    np.testing.assert_allclose(loss, float(g["loss"]), rtol=2e-5 if tight else 2e-4)
    np.testing.assert_allclose(gn, float(g["grad_norm"]), rtol=2e-4 if tight else 2e-3)
    worst = 0.0
    gall = np.sqrt(sum(float((g["grad_" + k].astype(np.float64) ** 2).sum()) for k in grads))
    for k, got in grads.items():
        ref = g["grad_" + k]
        # aggregate parity (spike-flip chaos forbids element-wise 1e-4 on every weight)
        denom = max(np.linalg.norm(ref), 1e-12)
        worst = max(worst, float(np.linalg.norm(got - ref) / max(denom, 1e-6 * gall)))
        assert np.linalg.norm(got - ref) <= (2e-4 if tight else 2e-3) * denom + 1e-6 * gall, (k, np.linalg.norm(got - ref) / denom)
    print(f"[{fix} flat={flat}] spike flips in the last pass {nflip} of {ntot}; worst gradient tensor rel-L2 {worst:.2e}")
    for k, ref in ((k[len("param1_"):], g[k]) for k in g.files if k.startswith("param1_")):
        # the first Adam step moves every weight by ~lr*sign(g): weights whose gradient is at
        # the fp32 noise floor may move the other way (<= 2*lr apart); the bulk must agree
        d = np.abs(newp[k] - ref)
        assert d.max() <= 2 * 2e-4 + 1e-6, k
        assert np.mean(d > 2e-5) <= 0.02, (k, np.mean(d > 2e-5))


@pytest.mark.parametrize("fix", ["g7_liffirenet_train", "g7_liffirenet_lowthresh"])
def test_train_step_vs_golden_on_the_fp32_matrix_path(fix):
    """model.precision = "fp32" (v_mfma_f32_32x32x2_f32 kernels: k_conv_lif_fwd, k_conv_dgrad, k_conv_wgrad_bits, k_lif_bwd)
    end to end against the same reference-generated train step as the default bf16x3 path."""
    g = load_golden(fix)
    loss, grads, gn, _ = _train_once(g, True, fix, precision="fp32")
    np.testing.assert_allclose(loss, float(g["loss"]), rtol=2e-4)
    np.testing.assert_allclose(gn, float(g["grad_norm"]), rtol=2e-3)
    for k, got in grads.items():
        ref = g["grad_" + k]
        denom = max(np.linalg.norm(ref), 1e-12)
        assert np.linalg.norm(got - ref) <= 2e-3 * denom + 1e-10, (k, np.linalg.norm(got - ref) / denom)


def test_train_step_vs_oracle_at_config3_shape():
    """B=8, 128x128, P=3 passes x 1500 events against the CPU oracle (same weights, same inputs)."""
    from event_flow_amd import synthetic
    from event_flow_amd.dataloader.encodings import encode_event_list

    B, n, H, W, P = 8, 1500, 128, 128, 3
    torch.manual_seed(0)
    model = LIFFireNet(model_cfg()).to(DEV)
    with torch.no_grad():  # lower thresholds so every layer is active at this input rate
        for k, p in model.named_parameters():
            if k.endswith("thresh"):
                p.mul_(0.25)
    params = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    keys = osnn.trainable_keys(params)
    evs = [synthetic.event_list_batch(B, n, H, W, 3000 + 17 * 0 + 100 * k) for k in range(P)]
    passes = [encode_event_list(G(ev), 2, (H, W)) for ev in evs]
    model.train()
    lossf = hloss.EventWarping(loss_cfg(H, W), DEV)
    for d in passes:
        out = model(d["event_voxel"], d["event_cnt"])
        lossf.event_flow_association(out["flow"], d["event_list"], d["event_list_pol_mask"], d["event_mask"])
    loss = lossf()
    loss.backward()
    torch.set_num_threads(16)
    opasses = [{k: v.detach().cpu() for k, v in d.items()} for d in passes]
    oloss_v, ograds, _, _ = otrain.train_step(
        "LIFFireNet", params, keys, opasses, [None] * 7, (H, W), {"step": 0, "m": {}, "v": {}},
        loss_cfg={"flow_regul_weight": 0.001, "mask_output": True},
    )
    np.testing.assert_allclose(float(loss.detach()), oloss_v, rtol=1e-3)
    # borderline spike flips (see the config-5 test below) would legitimately move the gradient by a few percent:
    # count them and widen the tolerance only then
    states = [None] * 7
    with torch.no_grad():
        for d in opasses:
            _, states = osnn.firenet_forward("LIFFireNet", params, d["event_cnt"], states)
    got_states = model.states
    nflip = sum(int((N(got_states[li][1]) != states[li][1].numpy()).sum()) for li in range(7))
    assert nflip <= 1e-4 * sum(states[li][1].numel() for li in range(7)), nflip
    tol = 1e-2 if nflip == 0 else 1e-1
    for k, p in model.named_parameters():
        ref = ograds[k].numpy()
        denom = max(np.linalg.norm(ref), 1e-12)
        assert np.linalg.norm(N(p.grad) - ref) <= tol * denom + 1e-9, (k, np.linalg.norm(N(p.grad) - ref) / denom, nflip)


def test_plif_train_step_vs_oracle_at_config5_shape():
    """BASELINE config 5: PLIF-FireNet on MVSEC-shaped 260x346 windows (ragged 32-pixel tiles: 346 = 10*32 + 26,
    260 = 32*8 + 4), B=2, 2 passes x 6000 events, against the CPU oracle; AEE of the predicted flow on the same
    inputs through both paths."""
    from event_flow_amd import synthetic
    from event_flow_amd.dataloader.encodings import encode_event_list
    from oracle import loss as oloss

    B, n, H, W, P = 2, 6000, 260, 346, 2
    torch.manual_seed(1)
    model = PLIFFireNet(model_cfg(PLIF_NEURON)).to(DEV)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if k.endswith("thresh"):
                p.mul_(0.25)
    params = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    keys = osnn.trainable_keys(params)
    evs = [synthetic.event_list_batch(B, n, H, W, 5000 + 100 * k) for k in range(P)]
    passes = [encode_event_list(G(ev), 2, (H, W)) for ev in evs]
    model.train()
    lossf = hloss.EventWarping(loss_cfg(H, W), DEV)
    for d in passes:
        out = model(d["event_voxel"], d["event_cnt"])
        lossf.event_flow_association(out["flow"], d["event_list"], d["event_list_pol_mask"], d["event_mask"])
    loss = lossf()
    loss.backward()
    torch.set_num_threads(16)
    opasses = [{k: v.detach().cpu() for k, v in d.items()} for d in passes]
    oloss_v, ograds, _, _ = otrain.train_step(
        "PLIFFireNet", params, keys, opasses, [None] * 7, (H, W), {"step": 0, "m": {}, "v": {}},
        loss_cfg={"flow_regul_weight": 0.001, "mask_output": True},
    )
    np.testing.assert_allclose(float(loss.detach()), oloss_v, rtol=1e-3)
    # forward again through the oracle to count borderline spike flips: the Heaviside is discontinuous, a neuron
    # whose v' sits within fp32 round-off of the threshold may fire in one summation order and not in the other,
    # and the difference spreads through the layers above (measured here: 1 flip in R1a -> 222 in R2b, all in one
    # 10x10 neighbourhood).  No flips: every tensor within 1e-2; with flips (< 1e-4 of the neurons): every tensor
    # within 1e-1 and the whole gradient within 5e-2.
    states = [None] * 7
    with torch.no_grad():
        for d in opasses:
            f_ref, states = osnn.firenet_forward("PLIFFireNet", params, d["event_cnt"], states)
    got_states = model.states
    nflip = sum(int((N(got_states[li][1]) != states[li][1].numpy()).sum()) for li in range(7))
    ntot = sum(states[li][1].numel() for li in range(7))
    assert nflip <= 1e-4 * ntot, (nflip, ntot)
    tol = 1e-2 if nflip == 0 else 1e-1
    gn = float(np.sqrt(sum(float((g.numpy() ** 2).sum()) for g in ograds.values())))
    for k, p in model.named_parameters():
        ref = ograds[k].numpy()
        denom = max(np.linalg.norm(ref), 1e-12)
        assert np.linalg.norm(N(p.grad) - ref) <= tol * denom + 1e-4 * gn, (k, np.linalg.norm(N(p.grad) - ref) / denom)
    err_all = float(np.sqrt(sum(float(((N(p.grad) - ograds[k].numpy()) ** 2).sum()) for k, p in model.named_parameters())))
    assert err_all <= (1e-2 if nflip == 0 else 5e-2) * gn, err_all / gn
    # AEE of the last flow through both paths ("matching the reference within 1e-4" when no neuron flipped)
    flow = out["flow"][0].detach().cpu()
    gt = torch.zeros(B, 2, H, W)
    gt[:, 0], gt[:, 1] = 3.0, -2.0
    mask = opasses[-1]["event_mask"][:, 0]
    aee_ref, _ = oloss.aee(f_ref, gt, mask, 128.0, 1.0, 1.0)
    # ... the HIP flow through the HIP metric (loss.flow.AEE, reference loss/flow.py:560-628), per sample (B = 1 is the only
    # batch size the reference's own broadcast is right for)
    metric = hloss.AEE(loss_cfg(H, W), DEV, flow_scaling=128)
    last = passes[-1]
    metric.event_flow_association([out["flow"][0].detach()], {
        "event_list": last["event_list"], "event_list_pol_mask": last["event_list_pol_mask"], "event_mask": last["event_mask"],
        "gtflow": gt.to(DEV), "dt_input": torch.ones(B), "dt_gt": torch.ones(B)})
    aee_got = N(metric()[0]).reshape(-1)
    np.testing.assert_allclose(np.array(aee_got), aee_ref.numpy().reshape(-1), rtol=1e-4 if nflip == 0 else 1e-3)
    # where no neuron of the top layer flipped, the flow agrees to 1e-4
    same = (N(got_states[6][1]) == states[6][1].numpy()).all(axis=1)
    d = np.abs(flow.numpy() - f_ref.numpy()).max(axis=1)
    assert d[same].max() <= 1e-4 * max(float(np.abs(f_ref.numpy()).max()), 1e-6) + 1e-7

def test_window_semantics_detach_and_reset():
    g = load_golden("g7_liffirenet_lowthresh")
    model = build_from_golden(g)
    model.train()
    passes = passes_from_golden(g)
    H, W = passes[0]["event_cnt"].shape[2:]
    lossf = hloss.EventWarping(loss_cfg(H, W), DEV)
    opt = FlatAdam(model)
    l1 = train_window(model, lossf, opt, passes)
    # states survive detach_states (truncated BPTT), are dropped by reset_states
    assert model.states[0] is not None
    l2 = train_window(model, lossf, opt, passes)
    assert np.isfinite(float(l1)) and np.isfinite(float(l2))
    model.reset_states()
    assert model.states[0] is None
    # states round-trip through the [2,B,C,H,W] API
    with torch.no_grad():
        model(passes[0]["event_voxel"], passes[0]["event_cnt"])
    st = model.states
    assert st[1].shape == (2, 2, 32, H, W)
    model.states = st
    st2 = model.states
    for a, b in zip(st, st2):
        assert torch.equal(a, b)


def test_fireflownet_runs_and_unsupported_fail_loudly():
    from event_flow_amd.models.model import RecEVFlowNet

    model = LIFFireFlowNet(model_cfg()).to(DEV)
    x = torch.rand(1, 2, 16, 40, device=DEV).round()
    out = model(x, x)
    assert out["flow"][0].shape == (1, 2, 16, 40)
    ann = dict(model_cfg(), spiking_neuron=None, activations=["relu", None])
    # batch-norm layers and transposed-conv decoders run on the general path (layer-level parity: fixture G16)
    net = RecEVFlowNet(dict(ann, norm="BN", use_upsample_conv=False)).to(DEV)
    assert any(k.endswith("running_mean") for k in net.state_dict()) and any("transposed_conv2d" in k for k in net.state_dict())
    xx = torch.rand(1, 2, 32, 32, device=DEV).round()
    fl = net(xx, xx)["flow"]
    assert len(fl) == 4 and fl[-1].shape == (1, 2, 32, 32)
    sum(f.sum() for f in fl).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
    with pytest.raises(TypeError):
        RecEVFlowNet(model_cfg())  # LIF neuron kwargs on the ConvGRU net: TypeError, as in the reference
    with pytest.raises(_lib.EvflowError):
        LIFFireNet(model_cfg())(x.cpu(), x.cpu())  # CPU tensors: no fallback
    with pytest.raises(AttributeError):
        bad = model_cfg()
        bad["encoding"] = "nope"
        LIFFireNet(bad).to(DEV)(x, x)


@pytest.mark.parametrize("rec", [False, True])
def test_bf16x3_forward_is_fp32_equivalent(rec):
    """The bf16x3 forward (3-way exact weight split on the bf16 matrix cores) and the
    fp32-MFMA forward against a float64 convolution: both must sit at fp32 round-off."""
    B, H, W, C = 2, 24, 40, 32
    g = torch.Generator(device="cpu").manual_seed(5)
    w_ff = (torch.rand(C, C, 3, 3, generator=g) * 2 - 1) * 0.18
    w_rec = (torch.rand(C, C, 3, 3, generator=g) * 2 - 1) * 0.18
    xb = (torch.rand(B, C, H, W, generator=g) < 0.3).float()
    zb = (torch.rand(B, C, H, W, generator=g) < 0.2).float()
    v = torch.randn(B, C, H, W, generator=g) * 0.3
    leak, thresh = torch.randn(C, generator=g) * 0.1 - 4, torch.randn(C, generator=g) * 0.1 + 0.8
    lam, th = torch.sigmoid(leak.double()), thresh.double().clamp_min(0.01)
    cur = torch.nn.functional.conv2d(xb.double(), w_ff.double(), padding=1)
    if rec:
        cur = cur + torch.nn.functional.conv2d(zb.double(), w_rec.double(), padding=1)
    ref = v.double() * lam.view(1, C, 1, 1) * (1 - zb.double()) + (1 - lam.view(1, C, 1, 1)) * cur
    scale = float((w_ff.abs().sum((1, 2, 3)).max() + (w_rec.abs().sum((1, 2, 3)).max() if rec else 0)))

    keep = []  # device copies must outlive the asynchronous kernels that read them

    def to_dev(t):
        keep.append(t.to(DEV).contiguous())
        return keep[-1]

    xbits, zbits = torch.empty(B, H, W, dtype=torch.int32, device=DEV), torch.empty(B, H, W, dtype=torch.int32, device=DEV)
    _lib.call("evf_nchw_to_bits", to_dev(xb).data_ptr(), B, H, W, xbits.data_ptr())
    _lib.call("evf_nchw_to_bits", to_dev(zb).data_ptr(), B, H, W, zbits.data_ptr())
    vin = torch.empty(B, H, W, C, device=DEV)
    _lib.call("evf_nchw_to_nhwc", to_dev(v).data_ptr(), B, C, H, W, vin.data_ptr())
    dleak, dthresh = to_dev(leak), to_dev(thresh)
    outs = {}
    for mode in ("fp32", "b3"):
        packs = []
        for wt in (w_ff, w_rec):
            if mode == "fp32":
                p = torch.empty(9216, device=DEV)
                _lib.call("evf_pack_conv_weight", to_dev(wt).data_ptr(), C, C, 0, p.data_ptr())
            else:
                p = torch.empty(54 * 1024, dtype=torch.uint8, device=DEV)
                _lib.call("evf_pack_conv_weight_b3", to_dev(wt).data_ptr(), C, C, p.data_ptr())
            packs.append(p)
        vo, zo = torch.empty(B, H, W, C, device=DEV), torch.empty(B, H, W, dtype=torch.int32, device=DEV)
        _lib.call("evf_conv_lif_fwd_b3" if mode == "b3" else "evf_conv_lif_fwd", xbits.data_ptr(), packs[0].data_ptr(),
                  packs[1].data_ptr() if rec else None, dleak.data_ptr(), dthresh.data_ptr(), vin.data_ptr(),
                  zbits.data_ptr(), B, H, W, 1, vo.data_ptr(), zo.data_ptr(), None)
        vn = torch.empty(B, C, H, W, device=DEV)
        _lib.call("evf_nhwc_to_nchw", vo.data_ptr(), B, C, H, W, vn.data_ptr())
        zn = torch.empty(B, C, H, W, device=DEV)
        _lib.call("evf_bits_to_nchw", zo.data_ptr(), B, H, W, zn.data_ptr())
        err = float((vn.cpu().double() - ref).abs().max())
        outs[mode] = err
        assert err <= 4e-7 * max(scale, 1.0), (mode, err, scale)  # fp32 accumulation round-off
        spikes_ref = ((ref - th.view(1, C, 1, 1)) > 0)
        safe = (ref - th.view(1, C, 1, 1)).abs() > 1e-5
        assert torch.equal(zn.cpu().bool()[safe], spikes_ref[safe])
    assert outs["b3"] <= 3 * outs["fp32"] + 1e-7, outs  # same error class as the exact-fp32 chain


@pytest.mark.parametrize("rec", [False, True])
@pytest.mark.parametrize("shape", [(2, 20, 128), (1, 9, 200)])
def test_fused_lif_bwd_wgrad_matches_separate_kernels(rec, shape):
    """evf_lif_bwd_wgrad (bf16x3 matrix part) == evf_lif_bwd + evf_conv_wgrad_bits (fp32 MFMA)."""
    B, H, W = shape
    C = 32
    g = torch.Generator(device="cpu").manual_seed(11)
    R = lambda *s: torch.randn(*s, generator=g).to(DEV)
    gz, gv, vo, vp = R(B, H, W, C), R(B, H, W, C) * 0.5, R(B, H, W, C) * 0.5 + 0.6, R(B, H, W, C) * 0.5
    xb = (torch.rand(B, C, H, W, generator=g) < 0.3).float().to(DEV)
    zb = (torch.rand(B, C, H, W, generator=g) < 0.2).float().to(DEV)
    leak, thresh = R(C) * 0.1 - 4, R(C) * 0.1 + 0.8
    xbits, zbits = (torch.empty(B, H, W, dtype=torch.int32, device=DEV) for _ in range(2))
    _lib.call("evf_nchw_to_bits", xb.data_ptr(), B, H, W, xbits.data_ptr())
    _lib.call("evf_nchw_to_bits", zb.data_ptr(), B, H, W, zbits.data_ptr())
    nW = (W + 31) // 32
    xT, zT = (torch.empty(B, H, C, nW, dtype=torch.int32, device=DEV) for _ in range(2))
    _lib.call("evf_bits_transpose", xbits.data_ptr(), B, H, W, xT.data_ptr())
    _lib.call("evf_bits_transpose", zbits.data_ptr(), B, H, W, zT.data_ptr())
    lib = _lib.load()
    # reference: separate kernels
    gc0, gp0 = torch.empty_like(gz), torch.empty_like(gz)
    gl0, gt0 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    _lib.call("evf_lif_bwd", gz.data_ptr(), gv.data_ptr(), vo.data_ptr(), vp.data_ptr(), zbits.data_ptr(), leak.data_ptr(),
              thresh.data_ptr(), B, H, W, 1, 0, 10.0, gc0.data_ptr(), gp0.data_ptr(), gl0.data_ptr(), gt0.data_ptr())
    ns0 = lib.evf_conv_wgrad_slabs(B, H, W)
    dw0 = {}
    for nm, bits in (("ff", xbits), ("rec", zbits)):
        slab = torch.empty(ns0, 9216, device=DEV)
        _lib.call("evf_conv_wgrad_bits", bits.data_ptr(), gc0.data_ptr(), B, H, W, slab.data_ptr(), 0)
        dw0[nm] = torch.zeros(C, C, 3, 3, device=DEV)
        _lib.call("evf_reduce_slabs", slab.data_ptr(), ns0, 9216, 0, dw0[nm].data_ptr())
    # fused
    gc1, gp1 = torch.empty_like(gz), torch.empty_like(gz)
    gs1 = torch.empty(3, B, H, W, C, dtype=torch.bfloat16, device=DEV)
    gl1, gt1 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    ns1 = lib.evf_lif_bwd_wgrad_slabs(B, H, W)
    s_ff, s_rec = torch.empty(ns1, 9216, device=DEV), torch.empty(ns1, 9216, device=DEV)
    for acc in (0, 1):  # second call accumulates: result must double
        _lib.call("evf_lif_bwd_wgrad", gz.data_ptr(), gv.data_ptr(), vo.data_ptr(), vp.data_ptr(), zbits.data_ptr(), xT.data_ptr(),
                  zT.data_ptr() if rec else None, leak.data_ptr(), thresh.data_ptr(), B, H, W, 1, 0, 10.0, gc1.data_ptr(), gs1.data_ptr(),
                  gp1.data_ptr(), gl1.data_ptr(), gt1.data_ptr(), s_ff.data_ptr(), s_rec.data_ptr() if rec else None, acc)
    assert torch.equal(gc1, gc0) and torch.equal(gp1, gp0)  # elementwise part: identical arithmetic
    # exact 3-way bf16 split of g_cur (error <= 2^-24 relative)
    rec3 = gs1[0].float() + gs1[1].float() + gs1[2].float()
    assert float((rec3 - gc0).abs().max()) <= 1.2e-7 * float(gc0.abs().max())
    # input-gradient conv on the split vs the fp32-MFMA kernel
    wt = (torch.rand(C, C, 3, 3, generator=g) * 2 - 1).mul(0.2).to(DEV)
    p32, pb3 = torch.empty(9216, device=DEV), torch.empty(54 * 1024, dtype=torch.uint8, device=DEV)
    _lib.call("evf_pack_conv_weight", wt.data_ptr(), C, C, 1, p32.data_ptr())
    _lib.call("evf_pack_conv_weight_b3t", wt.data_ptr(), C, C, pb3.data_ptr())
    gx0, gx1 = torch.empty_like(gz), torch.empty_like(gz)
    _lib.call("evf_conv_dgrad", gc0.data_ptr(), p32.data_ptr(), gx0.data_ptr(), 0, None, None, 0, B, H, W)
    _lib.call("evf_conv_dgrad_b3", gs1.data_ptr(), pb3.data_ptr(), gx1.data_ptr(), 0, B, H, W, None, None)
    assert float((gx1 - gx0).abs().max()) <= 3e-6 * float(gx0.abs().max())
    np.testing.assert_allclose(N(gl1), 2 * N(gl0), rtol=2e-4, atol=1e-4)
    np.testing.assert_allclose(N(gt1), 2 * N(gt0), rtol=2e-4, atol=1e-4)
    for nm, slab in (("ff", s_ff),) + ((("rec", s_rec),) if rec else ()):
        dw = torch.zeros(C, C, 3, 3, device=DEV)
        _lib.call("evf_reduce_slabs", slab.data_ptr(), ns1, 9216, 0, dw.data_ptr())
        ref = 2 * N(dw0[nm])
        assert np.abs(N(dw) - ref).max() <= 2e-6 * np.abs(ref).max() + 1e-5, nm


@pytest.mark.parametrize("shape", [(2, 24, 40), (1, 13, 70)])
def test_exact_split_input_gradient_is_fp32_equivalent(shape):
    """evf_conv_dgrad_b3_f32 (six exact bf16 products of two real operands, three cross terms below 2^-24 dropped; both
    kernels behind it and the recorded-cell launch) against conv_transpose2d in float64 -- the fp32 `conv2d` input
    gradient of models/spiking_submodules.py:520,530.  Bound per output element: 4 * 2^-24 * sum |g| |w| (the fp32
    accumulation round-off of a 288-long sum plus the dropped terms), and the same error class as the exact fp32-MFMA
    chain (evf_conv_dgrad).  The gradient spans three decades, like dL/d(current) in a trained network."""
    B, H, W = shape
    C = 32
    gen = torch.Generator(device="cpu").manual_seed(21)
    g = torch.randn(B, C, H, W, generator=gen) * torch.pow(10.0, -3 * torch.rand(B, C, H, W, generator=gen))
    w = (torch.rand(C, C, 3, 3, generator=gen) * 2 - 1) * 0.2
    ref = torch.nn.functional.conv_transpose2d(g.double(), w.double(), padding=1)
    mag = torch.nn.functional.conv_transpose2d(g.double().abs(), w.double().abs(), padding=1)
    bound = 4 * 2.0 ** -24 * mag
    gd, wd = g.to(DEV).contiguous(), w.to(DEV).contiguous()
    g_nhwc = torch.empty(B, H, W, C, device=DEV)
    _lib.call("evf_nchw_to_nhwc", gd.data_ptr(), B, C, H, W, g_nhwc.data_ptr())
    p32, pb3 = torch.empty(9216, device=DEV), torch.empty(54 * 1024, dtype=torch.uint8, device=DEV)
    _lib.call("evf_pack_conv_weight", wd.data_ptr(), C, C, 1, p32.data_ptr())
    _lib.call("evf_pack_conv_weight_b3t", wd.data_ptr(), C, C, pb3.data_ptr())

    def nchw(t):
        o = torch.empty(B, C, H, W, device=DEV)
        _lib.call("evf_nhwc_to_nchw", t.data_ptr(), B, C, H, W, o.data_ptr())
        return o.cpu().double()

    gx32 = torch.empty(B, H, W, C, device=DEV)
    _lib.call("evf_conv_dgrad", g_nhwc.data_ptr(), p32.data_ptr(), gx32.data_ptr(), 0, None, None, 0, B, H, W)
    err32 = (nchw(gx32) - ref).abs()
    L = _lib.load()
    outs = []
    try:
        for which in (0, 1):
            assert L.evf_conv_dgrad_select(which) == 0
            gx = torch.empty(B, H, W, C, device=DEV)
            _lib.call("evf_conv_dgrad_b3_f32", g_nhwc.data_ptr(), pb3.data_ptr(), gx.data_ptr(), 0, B, H, W, None, None)
            outs.append(gx)
    finally:
        L.evf_conv_dgrad_select(-1)
    gx = torch.empty(B, H, W, C, device=DEV)  # the recorded-cell launch (k_dgrad_diag_ws)
    assert _lib.raw("evf_bwd_defer_begin") == 0
    try:
        assert _lib.raw("evf_bwd_defer_slot", 0) == 0
        _lib.call("evf_conv_dgrad_b3_f32", g_nhwc.data_ptr(), pb3.data_ptr(), gx.data_ptr(), 0, B, H, W, None, None)
    finally:
        _lib.call("evf_bwd_defer_flush")
    outs.append(gx)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    # ... and from the pre-split planes (what the recorded backward feeds k_dgrad_diag_dma)
    hi = g_nhwc.to(torch.bfloat16)
    r1 = g_nhwc - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    planes = torch.stack([hi, mid, lo]).contiguous()
    assert torch.equal(planes[0].float() + planes[1].float() + planes[2].float(), g_nhwc)  # exact split
    gx = torch.empty(B, H, W, C, device=DEV)
    assert _lib.raw("evf_bwd_defer_begin") == 0
    try:
        assert _lib.raw("evf_bwd_defer_slot", 0) == 0
        _lib.call("evf_conv_dgrad_b3", planes.data_ptr(), pb3.data_ptr(), gx.data_ptr(), 0, B, H, W, None, None)
    finally:
        _lib.call("evf_bwd_defer_flush")
    assert torch.equal(gx, outs[0])
    err = (nchw(outs[0]) - ref).abs()
    worst = float((err / bound.clamp_min(1e-300)).max())
    print(f"input gradient vs float64: max err / (4 * 2^-24 * sum|g||w|) = {worst:.3f} (exact split), "
          f"{float((err32 / bound.clamp_min(1e-300)).max()):.3f} (fp32 MFMA); max abs err {float(err.max()):.3e} / {float(err32.max()):.3e}")
    assert bool((err <= bound + 1e-30).all()), worst
    assert float(err.max()) <= 3 * float(err32.max()) + 1e-9, (float(err.max()), float(err32.max()))


@pytest.mark.parametrize("rec", [False, True])
def test_exact_split_weight_gradient_is_fp32_equivalent(rec):
    """The weight-gradient slabs of evf_lif_bwd_wgrad (binary spikes x the exact 3-way bf16 split of g_cur on the bf16 matrix
    cores) against the float64 correlation of the same operands -- the fp32 `conv2d` weight gradients of
    models/spiking_submodules.py:520,530.  Bound per weight: 4 * 2^-24 * sum_pix |x| |g_cur| (the products are exact; what is
    left is the fp32 accumulation over pixels, blocks and slabs), and the same error class as the fp32-MFMA kernel
    (evf_conv_wgrad_bits) on the same operands."""
    B, H, W, C = 2, 40, 128, 32
    gen = torch.Generator(device="cpu").manual_seed(23)
    R = lambda *s: torch.randn(*s, generator=gen)
    gz = (R(B, H, W, C) * torch.pow(10.0, -2 * torch.rand(B, H, W, C, generator=gen))).to(DEV)
    gv, vo, vp = (R(B, H, W, C) * 0.5).to(DEV), (R(B, H, W, C) * 0.5 + 0.6).to(DEV), (R(B, H, W, C) * 0.5).to(DEV)
    xb = (torch.rand(B, C, H, W, generator=gen) < 0.3).float()
    zb = (torch.rand(B, C, H, W, generator=gen) < 0.2).float()
    leak, thresh = (R(C) * 0.1 - 4).to(DEV), (R(C) * 0.1 + 0.8).to(DEV)
    xbits, zbits = (torch.empty(B, H, W, dtype=torch.int32, device=DEV) for _ in range(2))
    xbd, zbd = xb.to(DEV), zb.to(DEV)
    _lib.call("evf_nchw_to_bits", xbd.data_ptr(), B, H, W, xbits.data_ptr())
    _lib.call("evf_nchw_to_bits", zbd.data_ptr(), B, H, W, zbits.data_ptr())
    nW = (W + 31) // 32
    xT, zT = (torch.empty(B, H, C, nW, dtype=torch.int32, device=DEV) for _ in range(2))
    _lib.call("evf_bits_transpose", xbits.data_ptr(), B, H, W, xT.data_ptr())
    _lib.call("evf_bits_transpose", zbits.data_ptr(), B, H, W, zT.data_ptr())
    lib = _lib.load()
    gc, gp = torch.empty(B, H, W, C, device=DEV), torch.empty(B, H, W, C, device=DEV)
    gl, gt = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    ns = lib.evf_lif_bwd_wgrad_slabs(B, H, W)
    s_ff, s_rec = torch.empty(ns, 9216, device=DEV), torch.empty(ns, 9216, device=DEV)
    _lib.call("evf_lif_bwd_wgrad", gz.data_ptr(), gv.data_ptr(), vo.data_ptr(), vp.data_ptr(), zbits.data_ptr(), xT.data_ptr(),
              zT.data_ptr() if rec else None, leak.data_ptr(), thresh.data_ptr(), B, H, W, 1, 0, 10.0, gc.data_ptr(), None,
              gp.data_ptr(), gl.data_ptr(), gt.data_ptr(), s_ff.data_ptr(), s_rec.data_ptr() if rec else None, 0)
    g_nchw = torch.empty(B, C, H, W, device=DEV)
    _lib.call("evf_nhwc_to_nchw", gc.data_ptr(), B, C, H, W, g_nchw.data_ptr())
    gd = g_nchw.cpu().double()
    ns0 = lib.evf_conv_wgrad_slabs(B, H, W)
    for nm, slab, bits, x in (("ff", s_ff, xbits, xb),) + ((("rec", s_rec, zbits, zb),) if rec else ()):
        ref = torch.nn.grad.conv2d_weight(x.double(), (C, C, 3, 3), gd, padding=1)
        mag = torch.nn.grad.conv2d_weight(x.double(), (C, C, 3, 3), gd.abs(), padding=1)
        dw = torch.zeros(C, C, 3, 3, device=DEV)
        _lib.call("evf_reduce_slabs", slab.data_ptr(), ns, 9216, 0, dw.data_ptr())
        s32 = torch.empty(ns0, 9216, device=DEV)
        _lib.call("evf_conv_wgrad_bits", bits.data_ptr(), gc.data_ptr(), B, H, W, s32.data_ptr(), 0)
        dw32 = torch.zeros(C, C, 3, 3, device=DEV)
        _lib.call("evf_reduce_slabs", s32.data_ptr(), ns0, 9216, 0, dw32.data_ptr())
        err, err32 = (dw.cpu().double() - ref).abs(), (dw32.cpu().double() - ref).abs()
        bound = 4 * 2.0 ** -24 * mag
        worst = float((err / bound.clamp_min(1e-300)).max())
        print(f"dW_{nm} vs float64: max err / (4 * 2^-24 * sum|x||g|) = {worst:.3f} (exact split), "
              f"{float((err32 / bound.clamp_min(1e-300)).max()):.3f} (fp32 MFMA); rel-L2 {float(err.norm() / ref.norm()):.2e} / "
              f"{float(err32.norm() / ref.norm()):.2e}")
        assert bool((err <= bound + 1e-30).all()), (nm, worst)
        assert float(err.norm()) <= 3 * float(err32.norm()) + 1e-12, nm


def test_hipgraph_replay_equals_eager_steps():
    """bench.py replays the whole train step (binning -> passes -> loss -> backward -> clip+Adam) from a
    hipGraph: the Adam step counter lives on the device and the recurrent state in static buffers.  Four steps
    over two alternating windows must leave the parameters where four eager steps leave them.

    The contrast loss sums its images with float atomics, so this comparison is STATISTICAL (see the comment at its bars; the exact
    statement is the deterministic-loss test below): a run in which a neuron at its threshold flips early can exceed the bars --
    seen once in ~10 runs of the suite -- and is repeated (up to three attempts) before it counts as a failure."""
    for attempt in range(3):
        try:
            _hipgraph_replay_equals_eager_steps_once()
            return
        except AssertionError:
            if attempt == 2:
                raise


def _hipgraph_replay_equals_eager_steps_once():
    from event_flow_amd import synthetic
    from event_flow_amd.dataloader.encodings import encode_event_list

    B, n, H, W, P = 2, 600, 32, 64, 3
    pool = [[G(synthetic.event_list_batch(B, n, H, W, 7000 + 100 * w + k)) for k in range(P)] for w in range(2)]

    def make():
        torch.manual_seed(3)
        m = LIFFireNet(model_cfg()).to(DEV)
        with torch.no_grad():
            for k, p in m.named_parameters():
                if k.endswith("thresh"):
                    p.mul_(0.25)
        m.train()
        return m

    def step(model, lossf, opt, lists):
        passes = [encode_event_list(ev, 2, (H, W), want=("cnt", "mask", "pol")) for ev in lists]
        for d in passes:
            d["event_voxel"] = None
        return train_window(model, lossf, opt, passes)

    # graph mode: 2 eager warm-up steps, then capture one graph per window and replay
    m1 = make()
    opt1 = FlatAdam(m1, lr=2e-4, clip=100.0, device_step=True)
    opt1.zero_grad()
    m1.use_static_states(True)
    l1 = hloss.EventWarping(loss_cfg(H, W), DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    losses1 = []
    with torch.cuda.stream(side):
        for i in range(2):
            losses1.append(step(m1, l1, opt1, pool[i % 2]))
        torch.cuda.synchronize()
        graphs = []
        for lists in pool:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                out = step(m1, l1, opt1, lists)
            graphs.append((g, out))
            losses1.append(out.clone())
        for i in range(2):
            g, out = graphs[i % 2]
            g.replay()
            losses1.append(out.clone())
        torch.cuda.synchronize()
    losses1 = [float(x) for x in losses1]
    # capture executes nothing, so the graph model did 2 eager + 2 replayed = 4 steps (device-side step counter,
    # static state buffers); the reference does 4 eager steps with the host-side counter
    m2 = make()
    opt2 = FlatAdam(m2, lr=2e-4, clip=100.0)
    opt2.zero_grad()
    l2 = hloss.EventWarping(loss_cfg(H, W), DEV)
    ref_losses = [float(step(m2, l2, opt2, pool[i % 2])) for i in range(4)]
    got = [losses1[0], losses1[1], losses1[4], losses1[5]]
    np.testing.assert_allclose(got[:2], ref_losses[:2], rtol=2e-4)
    # later steps start from weights that differ in the last bits (the contrast loss sums its images with float atomics: two
    # runs of the same EAGER step differ as well): now and then a neuron at its threshold flips and moves the loss by ~1e-3.
    # The exact statement -- replay == eager, bit for bit, once the loss is deterministic -- is
    # test_hipgraph_replay_is_bitwise_the_eager_step_under_a_deterministic_loss below.
    np.testing.assert_allclose(got[2:], ref_losses[2:], rtol=5e-3)
    for (k, p), (_, q) in zip(m1.named_parameters(), m2.named_parameters()):
        d = np.abs(N(p) - N(q))
        # Adam's first steps move every weight by ~lr; atomics reorder the gradient sums: bulk agreement
        assert d.max() <= 4 * 2e-4 + 1e-6, k
        assert np.mean(d > 4e-5) <= 0.05, (k, float(np.mean(d > 4e-5)))


class _LinearWindowLoss:
    """A window loss with a deterministic backward: sum over the passes of <flow_t, w_t> through torch ops (no float atomics),
    one upstream-gradient tensor PER PASS.  Same duck type as loss.flow.EventWarping for train.window_backward."""
    overwrite_intermediate = False

    def __init__(self, weights):
        self.w, self.flows = weights, []

    def event_flow_association(self, flow_list, event_list, pol_mask, event_mask):
        self.flows.append(flow_list[0])

    def __call__(self):
        return sum((f * self.w[k % len(self.w)]).sum() for k, f in enumerate(self.flows))

    def reset(self):
        self.flows = []


def test_hipgraph_replay_is_bitwise_the_eager_step_under_a_deterministic_loss():
    """The contrast loss sums its images with float atomics, so two runs of the SAME eager step differ in the last bits and a
    comparison of graph replay against eager launches through it can only be approximate (test_hipgraph_replay_equals_eager_steps).
    Everything else in the step is order-deterministic (per-block slabs and rows, no atomics): with a loss whose backward is
    deterministic -- and one upstream gradient tensor per pass, the case the recorded backward has to keep alive until its
    flush -- two eager + two replayed steps must leave EXACTLY the parameters, Adam moments and recurrent states of four eager
    steps, with the diagonal (recorded) launches on.  Gradient norm below the clip threshold (the norm itself is an atomic sum)."""
    from event_flow_amd import synthetic
    from event_flow_amd.dataloader.encodings import encode_event_list

    B, n, H, W, P = 2, 600, 32, 64, 3
    pool = [[G(synthetic.event_list_batch(B, n, H, W, 7100 + 100 * w + k)) for k in range(P)] for w in range(2)]
    gw = torch.Generator(device="cpu").manual_seed(9)
    wts = [(torch.randn(B, 2, H, W, generator=gw) * 0.02).to(DEV) for _ in range(P)]

    def make():
        torch.manual_seed(3)
        m = LIFFireNet(model_cfg()).to(DEV)
        with torch.no_grad():
            for k, p in m.named_parameters():
                if k.endswith("thresh"):
                    p.mul_(0.25)
        m.train()
        return m

    def step(model, lossf, opt, lists):
        passes = [encode_event_list(ev, 2, (H, W), want=("cnt", "mask", "pol")) for ev in lists]
        for d in passes:
            d["event_voxel"] = None
        return train_window(model, lossf, opt, passes)

    m1 = make()
    opt1 = FlatAdam(m1, lr=2e-4, clip=100.0, device_step=True)
    opt1.zero_grad()
    m1.use_static_states(True)
    l1 = _LinearWindowLoss(wts)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(2):
            step(m1, l1, opt1, pool[i % 2])
        torch.cuda.synchronize()
        graphs = []
        for lists in pool:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                step(m1, l1, opt1, lists)
            graphs.append(g)
        for i in range(2):
            graphs[i % 2].replay()
        torch.cuda.synchronize()
    m2 = make()
    opt2 = FlatAdam(m2, lr=2e-4, clip=100.0, device_step=True)
    opt2.zero_grad()
    m2.use_static_states(True)
    l2 = _LinearWindowLoss(wts)
    for i in range(4):
        step(m2, l2, opt2, pool[i % 2])
    torch.cuda.synchronize()
    assert float(opt2.norm_ws[0].sqrt()) < 100.0  # no clipping: the (atomically summed) norm does not enter the update
    assert float(opt1.norm_ws[1]) == 4.0 and float(opt2.norm_ws[1]) == 4.0
    assert torch.equal(opt1.flat_param, opt2.flat_param)
    assert torch.equal(opt1.m, opt2.m) and torch.equal(opt1.v, opt2.v)
    for a, b in zip(m1.states, m2.states):
        assert torch.equal(a, b)
    assert float((opt1.flat_param - make().to(DEV).state_dict()["head.ff.weight"].new_zeros(1)).abs().sum()) > 0  # (it trained)


def test_recorders_are_per_thread_and_poison_marks_stale_reads():
    """The diagonal-launch recorders of the library are thread_local: two host threads, each driving its own model through a
    window with recorded forward and backward cells at the same time, must get the gradients they get alone (a process-global
    recorder would mix their cells or refuse the second evf_*_defer_begin).  evf_defer_poison(1): between recording and
    flush the recorded cells' outputs hold NaN / all-ones words instead of plausible stale values."""
    import threading

    from event_flow_amd import synthetic
    from event_flow_amd.dataloader.encodings import encode_event_list

    B, n, H, W, P = 2, 500, 32, 64, 3
    lists = [[G(synthetic.event_list_batch(B, n, H, W, 8100 + 100 * w + k)) for k in range(P)] for w in range(2)]

    def make(seed):
        torch.manual_seed(seed)
        m = LIFFireNet(model_cfg()).to(DEV)
        with torch.no_grad():
            for k, p in m.named_parameters():
                if k.endswith("thresh"):
                    p.mul_(0.25)
        m.train()
        return m

    def window(model, evs, out, key, barrier=None):
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            lossf = hloss.EventWarping(loss_cfg(H, W), DEV)
            opt = FlatAdam(model, lr=2e-4, clip=100.0)
            opt.zero_grad()
            passes = [encode_event_list(ev, 2, (H, W), want=("cnt", "mask", "pol")) for ev in evs]
            for d in passes:
                d["event_voxel"] = None
            if barrier is not None:
                barrier.wait()  # both threads enter their windows together
            from event_flow_amd.train import window_backward

            loss = window_backward(model, lossf, opt, passes)
            stream.synchronize()
            out[key] = (float(loss), opt.flat_grad.detach().clone())

    alone = {}
    for w in range(2):
        window(make(10 + w), lists[w], alone, w)
    both = {}
    bar = threading.Barrier(2)
    ths = [threading.Thread(target=window, args=(make(10 + w), lists[w], both, w, bar)) for w in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert set(both) == {0, 1}
    for w in range(2):
        assert abs(both[w][0] - alone[w][0]) <= 1e-5 * abs(alone[w][0]), (w, both[w][0], alone[w][0])
        ga, gb = alone[w][1], both[w][1]
        assert float((ga - gb).norm()) <= 2e-4 * float(ga.norm()), w  # (the loss backward sums with float atomics)
    L = _lib.load()
    assert _lib.raw("evf_fwd_defer_pending") == 0 and _lib.raw("evf_bwd_defer_pending") == 0
    # poison: record one pass, look at a hidden layer's state tensor through torch BEFORE the flush
    m = make(3)
    m.defer_forward(True)
    assert L.evf_defer_poison(1) == 0
    try:
        d = encode_event_list(lists[0][0], 2, (H, W), want=("cnt", "mask", "pol"))
        m(None, d["event_cnt"])
        eng = m._eng()
        v_hidden = eng._states[1][0]  # raw engine tensor of a recorded cell (the public accessors flush first)
        assert _lib.raw("evf_fwd_defer_pending") > 0
        assert bool(torch.isnan(v_hidden).all())
        m.defer_forward(False)  # launches what was recorded
        assert _lib.raw("evf_fwd_defer_pending") == 0 and bool(torch.isfinite(v_hidden).all())
    finally:
        L.evf_defer_poison(0)
        m.defer_forward(False)


def test_flat_adam_zero_grad_clears_a_backward_issued_after_step():
    """FlatAdam.step() clears the flat gradient buffer as it consumes it and zero_grad() skips its fill kernel while the buffer
    is known to be clean.  A backward issued between step() and zero_grad() -- gradient accumulation, a custom loop: not
    announced by train.window_backward -- must still be cleared: the writers announce themselves (AccumulateGrad hooks, the
    engine's direct accumulation into the bound .grad)."""
    from event_flow_amd import synthetic
    from event_flow_amd.dataloader.encodings import encode_event_list

    B, n, H, W = 2, 400, 32, 32
    torch.manual_seed(11)
    m = LIFFireNet(model_cfg()).to(DEV)
    m.train()
    opt = FlatAdam(m, lr=2e-4, clip=100.0)
    opt.zero_grad()
    d = encode_event_list(G(synthetic.event_list_batch(B, n, H, W, 123)), 2, (H, W), want=("cnt", "mask", "pol"))

    def backward_once():
        m.reset_states()
        out = m(None, d["event_cnt"])["flow"][0]
        (out * out).sum().backward()

    backward_once()
    assert float(opt.flat_grad.abs().sum()) > 0
    opt.step()
    assert float(opt.flat_grad.abs().sum()) == 0.0  # step() hands the buffer back cleared (documented)
    backward_once()  # NOT through train.window_backward
    assert float(opt.flat_grad.abs().sum()) > 0
    opt.zero_grad()
    assert float(opt.flat_grad.abs().sum()) == 0.0
    opt.step()
    opt.zero_grad()  # clean after step(): no fill needed, still zero
    assert float(opt.flat_grad.abs().sum()) == 0.0


def test_graphed_window_step_equals_eager_training():
    """train.GraphedWindowStep: fixed-shape windows copied into a static buffer and replayed from two alternating
    hipGraphs (copy-free recurrent-state hand-over between them).  Six different windows, states carried across
    windows, must give the losses and parameters of six eager train_window steps."""
    from event_flow_amd import synthetic
    from event_flow_amd.train import GraphedWindowStep, encode_passes

    B, n, H, W, P = 2, 600, 32, 64, 3
    wins = [[G(synthetic.event_list_batch(B, n, H, W, 9000 + 100 * w + k)) for k in range(P)] for w in range(6)]

    def make():
        torch.manual_seed(5)
        m = LIFFireNet(model_cfg()).to(DEV)
        with torch.no_grad():
            for k, p in m.named_parameters():
                if k.endswith("thresh"):
                    p.mul_(0.25)
        m.train()
        return m

    m1 = make()
    opt1 = FlatAdam(m1, lr=2e-4, clip=100.0, device_step=True)
    opt1.zero_grad()
    stepper = GraphedWindowStep(m1, hloss.EventWarping(loss_cfg(H, W), DEV), opt1, 2, (H, W), want=("cnt", "mask", "pol"))
    m2 = make()
    opt2 = FlatAdam(m2, lr=2e-4, clip=100.0)
    opt2.zero_grad()
    l2 = hloss.EventWarping(loss_cfg(H, W), DEV)
    got, ref = [], []
    for w, lists in enumerate(wins):
        got.append(float(stepper.step(lists)))
        passes = encode_passes(lists, 2, (H, W), want=("cnt", "mask", "pol"))
        for d in passes:
            d["event_voxel"] = None
        ref.append(float(train_window(m2, l2, opt2, passes)))
        if w == 3:  # after the first replay of either graph: the model's Python-side state follows the replays
            for a, b in zip(m1.states, m2.states):
                assert float((a[1] != b[1]).float().mean()) < 1e-3  # spikes
                assert float((a[0] - b[0]).abs().max()) < 1e-3      # membrane potentials
    assert stepper.graphs is not None and stepper.seen == 6
    # the first replays follow the eager trajectory to fp32 round-off; later the two Adam paths' ~3e-6 parameter
    # differences flip a borderline spike somewhere (either path lands on either branch, depending on the order of the
    # atomics): from then on only the scale is comparable
    np.testing.assert_allclose(got[:4], ref[:4], rtol=5e-4)
    np.testing.assert_allclose(got[4:], ref[4:], rtol=2e-2)
    for (k, p), (_, q) in zip(m1.named_parameters(), m2.named_parameters()):
        d = np.abs(N(p) - N(q))
        assert d.max() <= 6 * 2e-4 + 1e-6, k


def test_graphed_window_step_survives_device_synchronize():
    """A hipDeviceSynchronize between two replays (any user code that times or checkpoints does one).  On ROCm 7.2 the replays
    behind it returned inf / NaN losses while the captured step still held memset NODES (hipMemsetAsync in the binning and loss
    kernels' launchers, torch's zero_()) between its kernel nodes; fine with stream synchronizes only, fine with
    DEBUG_CLR_GRAPH_PACKET_CAPTURE=0.  The library and its host now clear buffers with a kernel (evf_memset): the benched shape
    (8 x 128 x 128, 10 passes), the window that showed it, device synchronizes before every step from the first replay on."""
    from event_flow_amd import synthetic
    from event_flow_amd.train import GraphedWindowStep, encode_passes

    B, n, H, W, P = 8, 1500, 128, 128, 10
    lists = [G(synthetic.event_list_batch(B, n, H, W, synthetic.seed_for(5, 0, k))) for k in range(P)]

    def make():
        torch.manual_seed(0)
        m = LIFFireNet(model_cfg()).to(DEV)
        m.train()
        return m

    m1, m2 = make(), make()
    opt1 = FlatAdam(m1, lr=2e-4, clip=100.0, device_step=True)
    opt1.zero_grad()
    opt2 = FlatAdam(m2, lr=2e-4, clip=100.0)
    opt2.zero_grad()
    stepper = GraphedWindowStep(m1, hloss.EventWarping(loss_cfg(H, W), DEV), opt1, 2, (H, W))  # (default `want`: with the voxel grid)
    l2 = hloss.EventWarping(loss_cfg(H, W), DEV)
    got, ref = [], []
    for w in range(8):
        if w >= 3:
            torch.cuda.synchronize()
        got.append(float(stepper.step(lists)))
    for w in range(8):
        passes = encode_passes(lists, 2, (H, W), want=("cnt", "mask", "pol"))
        for d in passes:
            d["event_voxel"] = None
        ref.append(float(train_window(m2, l2, opt2, passes)))
    assert np.all(np.isfinite(got)), got
    np.testing.assert_allclose(got[:5], ref[:5], rtol=5e-4)
    np.testing.assert_allclose(got[5:], ref[5:], rtol=2e-2)  # (a flipped borderline spike: see the test above)


@pytest.mark.parametrize("shape", [(1, 5, 7), (3, 9, 33), (2, 8, 64)])
def test_firenet_tiny_and_ragged_resolutions_vs_oracle(shape):
    """Sensor sizes below / across the 8-row x 32-pixel tiles of the fused kernels (H < 8, W < 32, W = 33):
    forward flow, states and parameter gradients over two passes against the CPU oracle."""
    B, H, W = shape
    torch.manual_seed(2)
    model = LIFFireNet(model_cfg()).to(DEV)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if k.endswith("thresh"):
                p.mul_(0.2)
    params = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    keys = [k for k, _ in model.named_parameters()]
    for k in keys:
        params[k].requires_grad_(True)
    xs = [(torch.rand(B, 2, H, W) < 0.6).float() * torch.randint(1, 4, (B, 2, H, W)).float() for _ in range(2)]
    states = [None] * 7
    tot, tot_ref = 0, 0
    model.train()
    for x in xs:
        f_ref, states = osnn.firenet_forward("LIFFireNet", params, x, states)
        f = model(x.to(DEV), x.to(DEV))["flow"][0]
        wgt = torch.arange(f_ref.numel()).view(f_ref.shape).remainder(5).float() - 2.0
        tot_ref = tot_ref + (f_ref * wgt).sum()
        tot = tot + (f * wgt.to(DEV)).sum()
    nflip = sum(int((N(model.states[li][1]) != states[li][1].detach().numpy()).sum()) for li in range(7))
    assert nflip == 0  # tiny images: no borderline neuron in these seeds
    np.testing.assert_allclose(N(f), f_ref.detach().numpy(), rtol=1e-4, atol=1e-7)
    for li in range(7):
        np.testing.assert_allclose(N(model.states[li][0]), states[li][0].detach().numpy(), rtol=1e-5, atol=2e-6)
    tot.backward()
    tot_ref.backward()
    for k, p in model.named_parameters():
        ref = params[k].grad
        ref = ref.numpy() if ref is not None else np.zeros(tuple(p.shape), np.float32)
        denom = max(np.linalg.norm(ref), 1e-12)
        assert np.linalg.norm(N(p.grad) - ref) <= 2e-3 * denom + 1e-9, (k, np.linalg.norm(N(p.grad) - ref) / denom)


@pytest.mark.parametrize("acts,hard", [(("superspike", "superspike"), True), (("trianglespike", "trianglespike"), True),
                                       (("mgspike", "mgspike"), True), (("arctanspike", "arctanspike"), False),
                                       (("superspike", "arctanspike"), False)])
def test_fused_path_other_surrogates_and_soft_reset_vs_oracle(acts, hard):
    """The fused backward kernels carry the reference's default neuron (arctan surrogate, hard reset) as a compile-time
    fast path; every other surrogate (models/spiking_util.py:28-93) and the soft reset (spiking_submodules.py:119-123)
    run the general instantiation: two passes of LIF-FireNet against the CPU oracle (flow, states, parameter gradients)."""
    B, H, W = 2, 24, 40
    neuron = dict(NEURON, hard_reset=hard)
    cfg = model_cfg(neuron)
    cfg["activations"] = list(acts)
    flips = {}
    for seed in (5, 6, 7, 8):  # a borderline spike (|v' - thresh| at fp32 round-off) makes the gradients incomparable: the next seed
        torch.manual_seed(seed)
        model = LIFFireNet(dict(cfg)).to(DEV)
        with torch.no_grad():
            for k, p in model.named_parameters():
                if k.endswith("thresh"):
                    p.mul_(0.2)
        params = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        for k, _ in model.named_parameters():
            params[k].requires_grad_(True)
        xs = [(torch.rand(B, 2, H, W) < 0.6).float() * torch.randint(1, 4, (B, 2, H, W)).float() for _ in range(2)]
        states = [None] * 7
        tot, tot_ref = 0, 0
        model.train()
        for x in xs:
            f_ref, states = osnn.firenet_forward("LIFFireNet", params, x, states, acts=acts, hard_reset=hard)
            f = model(x.to(DEV), x.to(DEV))["flow"][0]
            wgt = torch.arange(f_ref.numel()).view(f_ref.shape).remainder(5).float() - 2.0
            tot_ref = tot_ref + (f_ref * wgt).sum()
            tot = tot + (f * wgt.to(DEV)).sum()
        nflip = sum(int((N(model.states[li][1]) != states[li][1].detach().numpy()).sum()) for li in range(7))
        flips[seed] = nflip
        if nflip:
            continue
        np.testing.assert_allclose(N(f), f_ref.detach().numpy(), rtol=1e-4, atol=1e-7)
        for li in range(7):
            np.testing.assert_allclose(N(model.states[li][0]), states[li][0].detach().numpy(), rtol=1e-5, atol=2e-6)
        tot.backward()
        tot_ref.backward()
        for k, p in model.named_parameters():
            ref = params[k].grad
            ref = ref.numpy() if ref is not None else np.zeros(tuple(p.shape), np.float32)
            denom = max(np.linalg.norm(ref), 1e-12)
            assert np.linalg.norm(N(p.grad) - ref) <= 2e-3 * denom + 1e-9, (k, np.linalg.norm(N(p.grad) - ref) / denom)
        return
    raise AssertionError(f"borderline spikes at every seed tried: {flips} (no flip-free pair of passes to compare gradients on)")


def _two_passes_vs_oracle(cls, name, neuron, B, H, W, seed):
    """Two passes of a fused FireNet against the CPU oracle.  Returns (#flipped spikes over all passes and layers, the
    oracle's largest |v' - thresh| among the flipped neurons of the FIRST layer that has any (everything after it
    diverges legitimately), max |flow difference| of the last pass, worst parameter-gradient
    difference relative to max(|that gradient|, 1e-4 |largest gradient|))."""
    torch.manual_seed(seed)
    model = cls(model_cfg(neuron)).to(DEV)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if k.endswith("thresh"):
                p.mul_(0.2)
    params = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    for k, _ in model.named_parameters():
        params[k].requires_grad_(True)
    xs = [(torch.rand(B, 2, H, W) < 0.6).float() * torch.randint(1, 4, (B, 2, H, W)).float() for _ in range(2)]
    states, tot, tot_ref, nflip, margin = [None] * 7, 0, 0, 0, 0.0
    model.train()
    for x in xs:
        f_ref, states = osnn.firenet_forward(name, params, x, states)
        f = model(x.to(DEV), x.to(DEV))["flow"][0]
        wgt = torch.arange(f_ref.numel()).view(f_ref.shape).remainder(5).float() - 2.0
        tot_ref = tot_ref + (f_ref * wgt).sum()
        tot = tot + (f * wgt.to(DEV)).sum()
        for li, ln in enumerate(LAYERS):
            bad = N(model.states[li][1]) != states[li][1].detach().numpy()
            if bad.any():
                if nflip == 0:  # the FIRST layer (of the first pass) with a flip: its inputs were still identical
                    th = params[ln + ".thresh"].detach().clamp_min(0.01).numpy().reshape(1, -1, 1, 1)
                    margin = float(np.abs(states[li][0].detach().numpy() - th)[bad].max())
                nflip += int(bad.sum())
    ferr = float((f.detach().cpu() - f_ref.detach()).abs().max())
    tot.backward()
    tot_ref.backward()
    refs = {k: (params[k].grad.numpy() if params[k].grad is not None else np.zeros(tuple(p.shape), np.float32))
            for k, p in model.named_parameters()}
    scale = max(np.linalg.norm(r) for r in refs.values())
    worst = max(float(np.linalg.norm(N(p.grad) - refs[k]) / max(np.linalg.norm(refs[k]), 1e-4 * scale, 1e-12))
                for k, p in model.named_parameters())
    return nflip, margin, ferr, worst


@pytest.mark.parametrize("seed", [3, 7])
def test_fused_firenets_on_random_shapes_vs_oracle(seed):
    """A seeded sweep of sensor sizes (1..40 x 1..200, B 1..3; every third case PLIF) through the fused path: ragged
    tiles in both directions, one-pixel-wide images, odd unit counts of the fused backward.  Where no spike flips, flow
    and gradients must match the oracle; where one does, the oracle itself must sit within 1e-6 of the threshold there
    (a rounding-order flip, not an error) -- tools/debug/fuzz_firenet.py is the long form of this test."""
    rng = np.random.default_rng(seed)
    compared = 0
    for it in range(10):
        B, H, W = int(rng.integers(1, 4)), int(rng.integers(1, 41)), int(rng.integers(1, 201))
        cls, name, neuron = (LIFFireNet, "LIFFireNet", NEURON) if it % 3 else (PLIFFireNet, "PLIFFireNet", PLIF_NEURON)
        nflip, margin, ferr, worst = _two_passes_vs_oracle(cls, name, neuron, B, H, W, seed * 1000 + it)
        if nflip:
            assert margin <= 1e-6, (name, B, H, W, nflip, margin)
            continue
        compared += 1
        assert ferr <= 1e-4 and worst <= 2e-3, (name, B, H, W, ferr, worst)
    assert compared >= 6


def test_event_warping_with_an_empty_pass_and_iwe_of_nothing():
    """Ragged windows: a pass that contributes zero events, and an IWE of an empty event list."""
    from event_flow_amd.utils import iwe as hiwe

    B, H, W = 2, 16, 24
    flow = torch.zeros(B, 2, H, W, device=DEV)
    out = hiwe.compute_pol_iwe(flow, torch.zeros(B, 0, 4, device=DEV), (H, W), torch.zeros(B, 0, 1, device=DEV),
                               torch.zeros(B, 0, 1, device=DEV))
    assert tuple(out.shape) == (B, 2, H, W) and float(out.abs().sum()) == 0.0


def test_stream_replicas_step_equals_the_unsplit_step():
    """train.StreamReplicas: the batch as two micro-batches through two replicas (shared weights, own state / tape /
    gradient buffer) on two HIP streams, gradients summed before the one optimizer step == the unsplit step, over two
    consecutive windows (the second starts from the carried recurrent state and the updated weights).
    (`bench.py --streams 2` replays the same step as one graph per replica on its stream + a join graph.)"""
    from event_flow_amd.train import StreamReplicas, train_window

    g = load_golden("g7_liffirenet_train")
    B0 = passes_from_golden(g)[0]["event_cnt"].shape[0]
    assert B0 % 2 == 0
    H, W = passes_from_golden(g)[0]["event_cnt"].shape[2:]

    def slice_passes(passes, lo, hi):
        return [{k: (v[lo:hi].contiguous() if v is not None else None) for k, v in d.items()} for d in passes]

    def run(split):
        model = build_from_golden(g)
        model.train()
        lossf = hloss.EventWarping(loss_cfg(H, W), DEV)
        opt = FlatAdam(model, lr=2e-4, clip=100.0)
        opt.zero_grad()
        reps = StreamReplicas(model, lossf, opt, n=2) if split else None
        out = []
        for w in range(2):
            passes = passes_from_golden(g)
            if split:
                loss = reps.train_window([slice_passes(passes, 0, B0 // 2), slice_passes(passes, B0 // 2, B0)])
            else:
                loss = train_window(model, lossf, opt, passes)
            torch.cuda.synchronize()
            out.append((float(loss), {k: N(v).copy() for k, v in model.state_dict().items()}, opt.grad_norm()))
        if split:  # the replicas really share the weights
            for m in reps.models[1:]:
                for p, q in zip(model.parameters(), m.parameters()):
                    assert p.data_ptr() == q.data_ptr()
        return out

    ref, got = run(False), run(True)
    for (l0, p0, n0), (l1, p1, n1) in zip(ref, got):
        np.testing.assert_allclose(l1, l0, rtol=1e-5)
        np.testing.assert_allclose(n1, n0, rtol=1e-4)
        for k in p0:
            # (Adam's first steps move every weight by ~lr * sign(g): compare the bulk, as the golden-step test does)
            d = np.abs(p1[k] - p0[k])
            assert d.max() <= 2 * 2e-4 + 1e-6, k
            assert np.mean(d > 2e-5) <= 0.02, (k, np.mean(d > 2e-5))


def test_plif_cells_recorded_on_diagonals_are_bit_identical(monkeypatch):
    """PLIF cells (pre-synaptic trace) through the recorded forward -- k_fwd_diag_t<.., PLIF>: team M pools the input spike
    counts of its strip, team E carries the trace; the head layer's passes in one launch with potential AND trace in registers
    (k_head_lif_fwd_win<.., PLIF>) -- against one launch per cell (k_conv_lif_fwd_b3<.., true>, k_head_lif_fwd): flows of every
    pass, potentials, spike words, traces after the window BIT for bit; and a training window's loss / gradient norm like two
    cell-by-cell runs (backward: the head layer's cells of the window in one launch with the trace backward inside, the hidden
    cells with the trace backward in their streaming team, launched pass by pass)."""
    from event_flow_amd import train as htrain

    g = load_golden("g7_pliffirenet_train")
    H, W = passes_from_golden(g)[0]["event_cnt"].shape[2:]

    def forward_only(defer):
        model = build_from_golden(g, fix="g7_pliffirenet_train")
        model.train()
        model.defer_forward(defer)
        flows = [model(d["event_voxel"], d["event_cnt"])["flow"][0] for d in passes_from_golden(g)]
        if defer:
            assert _lib.raw("evf_fwd_defer_pending") == 7 * len(flows)  # nothing has run yet (the head cells: one launch at the flush)
        model.defer_forward(False)
        assert _lib.raw("evf_fwd_defer_pending") == 0
        return [N(f).copy() for f in flows], [N(s).copy() for s in model.states]

    (f0, s0), (f1, s1) = forward_only(False), forward_only(True)
    assert len(s0) == len(s1) and all(a.shape == b.shape for a, b in zip(s0, s1))
    for a, b in zip(f0 + s0, f1 + s1):
        assert np.array_equal(a, b)
    assert max(float(np.abs(s).max()) for s in s1) > 0  # (the states are not trivially zero)

    grads = {}

    def run(defer, tag=None):
        monkeypatch.setattr(htrain, "DEFER_FORWARD", defer)
        monkeypatch.setattr(htrain, "DEFER_BACKWARD", defer)  # (PLIF: the backward of the window layer by layer, see below)
        model = build_from_golden(g, fix="g7_pliffirenet_train")
        model.train()
        lossf = hloss.EventWarping(loss_cfg(H, W), DEV)
        opt = FlatAdam(model, lr=2e-4, clip=100.0, device_step=False)
        opt.zero_grad()
        seen = []
        if tag is not None:  # the gradient as the window's backward leaves it (before clip + Adam consume it)
            real_step = opt.step
            opt.step = lambda *a_, **k_: (seen.append(opt.flat_grad.detach().clone()), real_step(*a_, **k_))[1]
        loss = htrain.train_window(model, lossf, opt, passes_from_golden(g))
        torch.cuda.synchronize()
        if tag is not None:
            grads[tag] = (seen[0], "_lm_bufs" in model._engine.__dict__)
        return float(loss), opt.grad_norm()

    (l0, n0), (l1, n1) = run(False), run(True)
    np.testing.assert_allclose(l1, l0, rtol=1e-6)
    np.testing.assert_allclose(n1, n0, rtol=1e-4)
    # the pooling's adjoint (AvgPool3x3^T / 32 of dL/d(pooled activity)) inside the input-gradient kernels against a k_plif_box
    # launch per cell: the same sums in the same order
    from event_flow_amd.models import engine as heng

    monkeypatch.setattr(heng, "PLIF_BOX_IN_DGRAD", False)
    l2, n2 = run(True)
    np.testing.assert_allclose(l2, l1, rtol=1e-6)
    np.testing.assert_allclose(n2, n1, rtol=1e-6)
    # the backward of the window LAYER by layer (feed-forward hidden layers: all passes in one launch with the carries in registers,
    # evf_plif_bwd_wgrad_window) against pass by pass: the same cells in another order -- the whole gradient vector to round-off
    from event_flow_amd.models import engine as heng2

    monkeypatch.setattr(heng2, "PLIF_BOX_IN_DGRAD", True)
    run(True, "lm")
    monkeypatch.setattr(heng2, "PLIF_LAYER_MAJOR", False)
    l4, n4 = run(True, "pp")
    monkeypatch.setattr(heng2, "PLIF_LAYER_MAJOR", True)
    assert grads["lm"][1] and not grads["pp"][1]  # (the layer-major path did run / did not run)
    np.testing.assert_allclose(l4, l1, rtol=1e-6)
    ga, gb = grads["lm"][0], grads["pp"][0]
    assert float(gb.abs().max()) > 0 and float((ga - gb).norm() / gb.norm()) < 2e-6
    # ... with its input gradients from pre-split planes through k_dgrad_diag_dma, a layer's passes per launch
    monkeypatch.setattr(heng2, "PLIF_LM_DGRAD", "dma")
    run(True, "lm_dma")
    monkeypatch.setattr(heng2, "PLIF_LM_DGRAD", "ws")
    # (two runs of one setting differ in the last bits already: the loss backward accumulates with float atomics)
    assert grads["lm_dma"][1] and float((grads["lm_dma"][0] - grads["lm"][0]).norm() / grads["lm"][0].norm()) < 2e-6
    # a window of 18 passes (more than a window launch holds: the passes that waited are replayed pass by pass, in order)
    def run_long(lm):
        monkeypatch.setattr(heng2, "PLIF_LAYER_MAJOR", lm)
        model = build_from_golden(g, fix="g7_pliffirenet_train")
        model.train()
        lossf = hloss.EventWarping(loss_cfg(H, W), DEV)
        opt = FlatAdam(model, lr=2e-4, clip=100.0)
        opt.zero_grad()
        seen = []
        real_step = opt.step
        opt.step = lambda *a_, **k_: (seen.append(opt.flat_grad.detach().clone()), real_step(*a_, **k_))[1]
        loss = htrain.train_window(model, lossf, opt, passes_from_golden(g) * 6)
        torch.cuda.synchronize()
        return float(loss), seen[0]

    (la, ga2), (lb, gb2) = run_long(True), run_long(False)
    monkeypatch.setattr(heng2, "PLIF_LAYER_MAJOR", True)
    np.testing.assert_allclose(la, lb, rtol=1e-6)
    assert float(gb2.abs().max()) > 0 and float((ga2 - gb2).norm() / gb2.norm()) < 2e-6
    # the trace backward as a launch of its own per cell (evf_plif_trace_bwd) against the fused forms (team E of the hidden cells'
    # fused backward, the head layer's window launch): the same bits per element, the per-channel sums to round-off
    monkeypatch.setattr(heng, "PLIF_BOX_IN_DGRAD", True)
    monkeypatch.setattr(heng, "PLIF_TRACE_FUSED", False)
    l3, n3 = run(True)
    np.testing.assert_allclose(l3, l1, rtol=1e-6)
    np.testing.assert_allclose(n3, n1, rtol=1e-5)


def test_diagonal_launches_equal_cell_by_cell_launches(monkeypatch):
    """train.window_backward records the hidden cells of a window and launches them diagonal by diagonal (k_fwd_diag:
    cells (t, l) with equal t + l in ONE launch, engine.defer_forward).  Same kernel body: the flow of every pass and the
    recurrent state after the window must be BIT-identical to the cell-by-cell launches; a training step (whose backward
    sums with float atomics, i.e. is reproducible to rounding only) must agree like two cell-by-cell runs do."""
    from event_flow_amd import train as htrain

    g = load_golden("g7_liffirenet_train")
    H, W = passes_from_golden(g)[0]["event_cnt"].shape[2:]

    def forward_only(defer):
        model = build_from_golden(g)
        model.train()
        model.defer_forward(defer)
        flows = [model(d["event_voxel"], d["event_cnt"])["flow"][0] for d in passes_from_golden(g)]
        if defer:
            assert _lib.raw("evf_fwd_defer_pending") == 7 * len(flows)  # nothing has run yet (the head cells: one launch at the flush)
        model.defer_forward(False)
        assert _lib.raw("evf_fwd_defer_pending") == 0
        return [N(f).copy() for f in flows], [N(s).copy() for s in model.states]

    (f0, s0), (f1, s1) = forward_only(False), forward_only(True)
    for a, b in zip(f0 + s0, f1 + s1):
        assert np.array_equal(a, b)

    def run(defer):
        monkeypatch.setattr(htrain, "DEFER_FORWARD", defer)
        monkeypatch.setattr(htrain, "DEFER_BACKWARD", defer)  # the backward cells likewise (k_bwd_diag, k_dgrad_diag)
        model = build_from_golden(g)
        model.train()
        lossf = hloss.EventWarping(loss_cfg(H, W), DEV)
        opt = FlatAdam(model, lr=2e-4, clip=100.0)
        opt.zero_grad()
        out = []
        for w in range(2):
            loss = htrain.train_window(model, lossf, opt, passes_from_golden(g))
            torch.cuda.synchronize()
            out.append((float(loss), opt.grad_norm(), {k: N(v).copy() for k, v in model.state_dict().items()}))
        return out

    ref, got = run(False), run(True)
    assert _lib.raw("evf_fwd_defer_pending") == 0 and _lib.raw("evf_bwd_defer_pending") == 0
    # first window: same weights, bit-identical flows -> the same loss up to the order of the loss's own float atomics
    np.testing.assert_allclose(got[0][0], ref[0][0], rtol=1e-6)
    for (l0, n0, p0), (l1, n1, p1) in zip(ref, got):
        np.testing.assert_allclose(l1, l0, rtol=1e-5)
        np.testing.assert_allclose(n1, n0, rtol=1e-4)
        for k in p0:
            d = np.abs(p1[k] - p0[k])
            assert d.max() <= 2 * 2e-4 + 1e-6, k
            assert np.mean(d > 2e-5) <= 0.02, (k, np.mean(d > 2e-5))


@pytest.mark.parametrize("fix", ["g7_liffirenet_train", "g7_pliffirenet_train"])
@pytest.mark.parametrize("mode", ["1", "top"])
def test_layer_major_forward_is_bit_identical(monkeypatch, fix, mode):
    """The recorded forward LAYER by layer (engine._fwd_slots, EVF_FWD_LM): the passes of a feed-forward hidden layer recorded
    under one index and launched as a chain -- k_fwd_win_t: potential, trace and the pixel's previous spikes stay in team E's
    registers from pass to pass, the tape is written only -- against the diagonal schedule (one launch per index t + l - 1, the
    state read back every pass): flows of every pass, potentials, spike words and traces after the window BIT for bit, windows
    of 4 passes, of 18 (more than a chain holds: the recording is launched and reopened) and a second window that starts from
    the first one's state; a training window's loss and gradient like two runs of one schedule."""
    from event_flow_amd import train as htrain
    from event_flow_amd.models import engine as heng

    g = load_golden(fix)
    base = passes_from_golden(g)
    H, W = base[0]["event_cnt"].shape[2:]

    def forward_only(lm, reps):
        monkeypatch.setattr(heng, "FWD_LAYER_MAJOR", lm)
        model = build_from_golden(g, fix=fix)
        model.train()
        flows, states = [], []
        for w in range(2):  # (the second window starts from the first one's final state)
            model.defer_forward(True)
            fl = [model(d["event_voxel"], d["event_cnt"])["flow"][0] for d in base * reps]
            model.defer_forward(False)
            assert _lib.raw("evf_fwd_defer_pending") == 0
            flows += [N(f).copy() for f in fl]
            states += [N(s).copy() for s in model.states]
            model.detach_states()
        return flows, states

    for reps in (1, (18 + len(base) - 1) // len(base)):
        (f0, s0), (f1, s1) = forward_only("0", reps), forward_only(mode, reps)
        assert len(f0) == len(f1) and len(s0) == len(s1)
        for a, b in zip(f0 + s0, f1 + s1):
            assert np.array_equal(a, b)
        assert max(float(np.abs(s).max()) for s in s1) > 0

    def run(lm):
        monkeypatch.setattr(heng, "FWD_LAYER_MAJOR", lm)
        monkeypatch.setattr(htrain, "DEFER_FORWARD", True)
        monkeypatch.setattr(htrain, "DEFER_BACKWARD", True)
        model = build_from_golden(g, fix=fix)
        model.train()
        lossf = hloss.EventWarping(loss_cfg(H, W), DEV)
        opt = FlatAdam(model, lr=2e-4, clip=100.0, device_step=False)
        opt.zero_grad()
        seen = []
        real_step = opt.step
        opt.step = lambda *a_, **k_: (seen.append(opt.flat_grad.detach().clone()), real_step(*a_, **k_))[1]
        loss = htrain.train_window(model, lossf, opt, base)
        torch.cuda.synchronize()
        return float(loss), seen[0]

    (l0, g0), (l1, g1) = run("0"), run(mode)
    np.testing.assert_allclose(l1, l0, rtol=1e-6)
    assert float(g0.abs().max()) > 0 and float((g1 - g0).norm() / g0.norm()) < 2e-6


@pytest.mark.parametrize("cls_name", ["LIFFireNet", "PLIFFireNet"])
@pytest.mark.parametrize("shape,hard", [((2, 18, 44), True), ((1, 7, 33), False), ((3, 32, 64), False), ((1, 64, 96), True)])
def test_layer_major_forward_on_ragged_shapes_and_both_resets(monkeypatch, cls_name, shape, hard):
    """k_fwd_win_t on partial strips (odd heights, widths that are no multiple of 32), whole ones, hard and soft reset, a network
    made alive (thresholds x 0.3): flows of five passes and the final state bit for bit against the diagonal schedule."""
    from event_flow_amd import train as htrain
    from event_flow_amd.models import engine as heng

    B, H, W = shape
    n_ev, P = 40 * B * H * W // 64 + 50, 5
    gen = torch.Generator().manual_seed(23)
    lists = []
    for _ in range(P):
        ts = torch.sort(torch.rand(B, n_ev, generator=gen), dim=1).values
        ys = torch.randint(0, H, (B, n_ev), generator=gen).float()
        xs = torch.randint(0, W, (B, n_ev), generator=gen).float()
        ps = torch.randint(0, 2, (B, n_ev), generator=gen).float() * 2 - 1
        lists.append(torch.stack([ts, ys, xs, ps], dim=2).to(DEV))
    passes = htrain.encode_passes(lists, 2, (H, W))
    neuron = dict(NEURON if cls_name == "LIFFireNet" else FIXTURES["g7_pliffirenet_train"][1])
    neuron["hard_reset"] = hard
    cls = LIFFireNet if cls_name == "LIFFireNet" else PLIFFireNet

    def forward_only(lm):
        monkeypatch.setattr(heng, "FWD_LAYER_MAJOR", lm)
        torch.manual_seed(4)
        model = cls(model_cfg(neuron)).to(DEV)
        with torch.no_grad():
            for k, p in model.named_parameters():
                if k.endswith("thresh"):
                    p.mul_(0.3)
        model.train()
        model.defer_forward(True)
        fl = [model(d["event_voxel"], d["event_cnt"])["flow"][0] for d in passes]
        model.defer_forward(False)
        return [N(f).copy() for f in fl], [N(s).copy() for s in model.states]

    f0, s0 = forward_only("0")
    assert sum(float(np.abs(s[1]).sum()) for s in s0) > 0  # (spikes in the final state: the network is alive)
    for mode in ("1", "top"):
        f1, s1 = forward_only(mode)
        for a, b in zip(f0 + s0, f1 + s1):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("fix,reps", [("g7_liffirenet_train", 1), ("g7_liffirenet_lowthresh", 1), ("g7_liffirenet_lowthresh", 5)])
def test_lif_top_layers_backward_as_window_launches(monkeypatch, fix, reps):
    """LIF windows: the backward of the feed-forward layers above the last recurrent one (R2b under the prediction head, R2a) for
    ALL passes in one launch each ahead of the diagonals (engine._backward_window_top: evf_lif_bwd_wgrad_window + one
    evf_conv_dgrad_b3_multi launch per layer) against every hidden layer on the diagonals: the same cells in another order -- loss
    equal, the whole gradient vector to round-off; windows of the fixture's passes and of 5 x as many (more than a window launch
    holds: the passes that waited are replayed pass by pass)."""
    from event_flow_amd import train as htrain
    from event_flow_amd.models import engine as heng

    g = load_golden(fix)
    base = passes_from_golden(g)
    H, W = base[0]["event_cnt"].shape[2:]
    used = {}

    def run(top):
        monkeypatch.setattr(heng, "LIF_BWD_TOP", top)
        monkeypatch.setattr(htrain, "DEFER_FORWARD", True)
        monkeypatch.setattr(htrain, "DEFER_BACKWARD", True)
        model = build_from_golden(g, fix=fix)
        model.train()
        lossf = hloss.EventWarping(loss_cfg(H, W), DEV)
        opt = FlatAdam(model, lr=2e-4, clip=100.0, device_step=False)
        opt.zero_grad()
        seen = []
        real_step = opt.step
        opt.step = lambda *a_, **k_: (seen.append(opt.flat_grad.detach().clone()), real_step(*a_, **k_))[1]
        losses = []
        for w in range(2):  # (the second window starts from the first one's state and from updated weights)
            losses.append(float(htrain.train_window(model, lossf, opt, base * reps)))
        torch.cuda.synchronize()
        used[top] = "_lm_bufs" in model._engine.__dict__
        return losses, seen

    (l0, g0), (l1, g1) = run(False), run(True)
    assert not used[False] and used[True] == (len(base) * reps <= 16)
    np.testing.assert_allclose(l1[0], l0[0], rtol=1e-6)
    assert float(g0[0].abs().max()) > 0 and float((g1[0] - g0[0]).norm() / g0[0].norm()) < 2e-6
    # second window: after an Adam step on gradients that differ in the last bits
    np.testing.assert_allclose(l1[1], l0[1], rtol=1e-4)


@pytest.mark.parametrize("skip", [1, 5, 0])
def test_lif_top_window_launches_step_aside_for_a_pass_without_flow_gradient(monkeypatch, skip):
    """A window in which one pass's flow does not enter the loss (its backward node gets no dL/dflow): the passes that were
    waiting for the top layers' window launches are replayed pass by pass, in order, and the rest of the window follows the
    same way -- parameter gradients as with every hidden layer on the diagonals (EVF_LIF_BWD_TOP=0)."""
    from event_flow_amd.models import engine as heng

    g = load_golden("g7_liffirenet_lowthresh")
    base = passes_from_golden(g) * 2
    assert len(base) > skip

    def run(top):
        monkeypatch.setattr(heng, "LIF_BWD_TOP", top)
        model = build_from_golden(g, fix="g7_liffirenet_lowthresh")
        model.train()
        model.defer_forward(True)
        flows = [model(d["event_voxel"], d["event_cnt"])["flow"][0] for d in base]
        model.defer_forward(False)
        loss = sum((f * f).sum() * (1.0 + 0.1 * k) for k, f in enumerate(flows) if k != skip)
        model.defer_backward(True)
        loss.backward()
        model.defer_backward(False)
        torch.cuda.synchronize()
        assert _lib.raw("evf_bwd_defer_pending") == 0
        return float(loss.detach()), {k: N(p.grad).copy() for k, p in model.named_parameters() if p.grad is not None}

    (l0, g0), (l1, g1) = run(False), run(True)
    assert l0 == l1 and set(g0) == set(g1) and len(g0) > 10
    num = sum(float(((g1[k] - g0[k]) ** 2).sum()) for k in g0) ** 0.5
    den = sum(float((g0[k] ** 2).sum()) for k in g0) ** 0.5
    assert den > 0 and num / den < 2e-6, num / den


@pytest.mark.parametrize("P", [1, 2, 50])
def test_diagonal_launches_short_and_long_windows(monkeypatch, P):
    """Windows of 1 and 2 passes (diagonals of one cell) and of 50 passes (the backward index 2 (P - 1 - t) + step runs past
    the recorder's 96 slots: it launches what it holds and starts over): loss and gradient like a repeated cell-by-cell run."""
    from event_flow_amd import train as htrain

    B, H, W, n_ev = 2, 16, 40, 150
    gen = torch.Generator().manual_seed(11)
    lists = []
    for _ in range(P):
        ts = torch.sort(torch.rand(B, n_ev, generator=gen), dim=1).values
        ys = torch.randint(0, H, (B, n_ev), generator=gen).float()
        xs = torch.randint(0, W, (B, n_ev), generator=gen).float()
        ps = torch.randint(0, 2, (B, n_ev), generator=gen).float() * 2 - 1
        lists.append(torch.stack([ts, ys, xs, ps], dim=2).to(DEV))

    from event_flow_amd.models.engine import FireNetEngine

    midpass = []  # flushes issued while a backward pass was being recorded: (its tape is still held afterwards)
    plain_flush = FireNetEngine.flush_backward

    def watched_flush(self):
        plain_flush(self)
        cur = self.__dict__.get("_bdefer_cur")
        if cur is not None:
            midpass.append(any(e is cur for e in self._bdefer_keep))

    monkeypatch.setattr(FireNetEngine, "flush_backward", watched_flush)

    def run(defer):
        monkeypatch.setattr(htrain, "DEFER_FORWARD", defer)
        monkeypatch.setattr(htrain, "DEFER_BACKWARD", defer)
        torch.manual_seed(4)
        model = LIFFireNet(model_cfg()).to(DEV)
        with torch.no_grad():
            for k, p in model.named_parameters():
                if k.endswith("thresh"):
                    p.mul_(0.3)
        model.train()
        lossf = hloss.EventWarping(loss_cfg(H, W), DEV)
        opt = FlatAdam(model, lr=2e-4, clip=100.0)
        opt.zero_grad()
        passes = htrain.encode_passes(lists, 2, (H, W))
        loss = htrain.window_backward(model, lossf, opt, passes)
        torch.cuda.synchronize()
        assert _lib.raw("evf_fwd_defer_pending") == 0 and _lib.raw("evf_bwd_defer_pending") == 0
        return float(loss.detach()), N(opt.flat_grad).copy()

    (l0, g0), (l1, g1) = run(False), run(True)
    # the overflow flush of the 50-pass window happens in the MIDDLE of a pass: the cells that pass records afterwards point
    # into its tape, which must therefore stay referenced (a freed tape = silently corrupted gradients)
    assert all(midpass) and (len(midpass) >= 1) == (P == 50), midpass
    np.testing.assert_allclose(l1, l0, rtol=1e-6)  # (bit-identical flows; the loss sums its images with float atomics)
    assert np.isfinite(g0).all() and np.linalg.norm(g0) > 0
    assert np.linalg.norm(g1 - g0) <= 1e-4 * np.linalg.norm(g0)


def test_fused_step_tail_equals_the_step_by_step_tail(monkeypatch):
    """evf_grads_finalize (slab reduction + row sums + segment add in one launch) and evf_clip_adam_fused (squared norm, grid
    hand-shake, clip + Adam + zero_grad in one launch) against the launches they replace, over three optimizer steps with
    clipping active: every gradient and the norm to fp32 round-off (the window itself is reproducible to round-off only, and
    the per-channel sums take another order), the same parameters after every step, the gradient buffer handed back zeroed and
    the step counter advanced on the device."""
    from event_flow_amd import train as htrain
    from event_flow_amd.models import engine as heng

    g = load_golden("g7_liffirenet_lowthresh")
    H, W = passes_from_golden(g)[0]["event_cnt"].shape[2:]

    def run(fused):
        monkeypatch.setattr(heng, "FUSED_TAIL", fused)
        monkeypatch.setattr(htrain, "FUSED_ADAM", fused)
        model = build_from_golden(g, fix="g7_liffirenet_lowthresh")
        model.train()
        lossf = hloss.EventWarping(loss_cfg(H, W), DEV)
        opt = FlatAdam(model, lr=2e-4, clip=0.5, device_step=True)  # (clip well below the gradient norm: the coefficient matters)
        opt.zero_grad()
        out = []
        for _ in range(3):
            loss = htrain.window_backward(model, lossf, opt, passes_from_golden(g))
            grad = N(opt.flat_grad).copy()
            htrain.window_apply(model, lossf, opt, loss)
            torch.cuda.synchronize()
            assert float(opt.flat_grad.abs().max()) == 0.0  # zero_grad folded into the step
            out.append((float(loss.detach()), grad, opt.grad_norm(), N(opt.flat_param).copy(), float(opt.norm_ws[1])))
        if fused:
            assert float(opt.norm_ws[2:5].abs().max()) == 0.0  # (running sum and tickets handed back zeroed)
        return out, {k: (o, p.numel()) for (k, p), o in zip([(k, p) for k, p in model.named_parameters() if p.requires_grad],
                                                            np.cumsum([0] + [p.numel() for p in model.parameters() if p.requires_grad][:-1]))}

    (ref, layout), (got, _) = run(False), run(True)
    for step, ((l0, g0, n0, p0, c0), (l1, g1, n1, p1, c1)) in enumerate(zip(ref, got)):
        assert c0 == c1 == step + 1
        np.testing.assert_allclose(l1, l0, rtol=1e-5)
        np.testing.assert_allclose(n1, n0, rtol=1e-5)
        assert n0 > 0.5  # (clipping is active)
        if step == 0:  # same weights, same forward: the window's gradient itself
            for k, (off, n) in layout.items():
                # (two runs of the window differ in the last bits already: the loss sums its images with float atomics)
                np.testing.assert_allclose(g1[off:off + n], g0[off:off + n], rtol=2e-4, atol=2e-6 * np.abs(g0).max(), err_msg=k)
        d = np.abs(p1 - p0)
        assert d.max() <= 2 * 2e-4 + 1e-6 and np.mean(d > 2e-5) <= 0.02, (step, d.max(), np.mean(d > 2e-5))
