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
from event_flow_amd.models.model import LIFFireFlowNet, LIFFireNet  # noqa: E402
from event_flow_amd.train import FlatAdam, train_window  # noqa: E402
from oracle import snn as osnn  # noqa: E402
from oracle import train as otrain  # noqa: E402

DEV = "cuda:0"
LAYERS = ["head", "G1", "R1a", "R1b", "G2", "R2a", "R2b"]
NEURON = {"leak": [-4.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True, "learn_thresh": True, "hard_reset": True}


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


def build_from_golden(g, prefix="param0_", cls=LIFFireNet):
    model = cls(model_cfg()).to(DEV)
    sd = {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}
    model.load_state_dict(sd)  # reference state_dict keys load unchanged
    return model


def passes_from_golden(g):
    P = int(g["meta_P"])
    keys = ("event_cnt", "event_voxel", "event_list", "event_list_pol_mask", "event_mask")
    return [{k: G(g[f"p{i}_{k}"]) for k in keys} for i in range(P)]


@pytest.mark.parametrize("fix", ["g7_liffirenet_train", "g7_liffirenet_lowthresh"])
def test_forward_per_layer_and_flow(fix):
    g = load_golden(fix)
    model = build_from_golden(g)
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
            if nflip == 0:
                np.testing.assert_allclose(N(out["flow"][0]), g[f"p{i}_flow"], rtol=1e-4, atol=1e-7)
            assert set(out["activity"].keys()) == {"0:input", "1:head", "2:G1", "3:R1a", "4:R1b", "5:G2", "6:R2a", "7:R2b", "8:pred"}
    assert nflip <= 1e-5 * ntot, (nflip, ntot)


def _train_once(g, use_flat_adam):
    model = build_from_golden(g)
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


@pytest.mark.parametrize("fix", ["g7_liffirenet_train", "g7_liffirenet_lowthresh"])
@pytest.mark.parametrize("flat", [True, False])
def test_train_step_vs_golden(fix, flat):
    g = load_golden(fix)
    loss, grads, gn, newp = _train_once(g, flat)
    np.testing.assert_allclose(loss, float(g["loss"]), rtol=2e-4)
    np.testing.assert_allclose(gn, float(g["grad_norm"]), rtol=2e-3)
    for k, got in grads.items():
        ref = g["grad_" + k]
        # aggregate parity (spike-flip chaos forbids element-wise 1e-4 on every weight)
        denom = max(np.linalg.norm(ref), 1e-12)
        assert np.linalg.norm(got - ref) <= 2e-3 * denom + 1e-10, (k, np.linalg.norm(got - ref) / denom)
    for k, ref in ((k[len("param1_"):], g[k]) for k in g.files if k.startswith("param1_")):
        # the first Adam step moves every weight by ~lr*sign(g): weights whose gradient is at
        # the fp32 noise floor may move the other way (<= 2*lr apart); the bulk must agree
        d = np.abs(newp[k] - ref)
        assert d.max() <= 2 * 2e-4 + 1e-6, k
        assert np.mean(d > 2e-5) <= 0.02, (k, np.mean(d > 2e-5))


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
    for k, p in model.named_parameters():
        ref = ograds[k].numpy()
        denom = max(np.linalg.norm(ref), 1e-12)
        assert np.linalg.norm(N(p.grad) - ref) <= 1e-2 * denom + 1e-9, (k, np.linalg.norm(N(p.grad) - ref) / denom)


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
    from event_flow_amd.models.model import FireNet, PLIFFireNet

    model = LIFFireFlowNet(model_cfg()).to(DEV)
    x = torch.rand(1, 2, 16, 40, device=DEV).round()
    out = model(x, x)
    assert out["flow"][0].shape == (1, 2, 16, 40)
    with pytest.raises(NotImplementedError):
        m = PLIFFireNet(model_cfg({"leak_v": [-4.0, 0.1], "thresh": [0.8, 0.1]})).to(DEV)
        m(x, x)
    with pytest.raises(_lib.EvflowError):
        LIFFireNet(model_cfg())(x.cpu(), x.cpu())  # CPU tensors: no fallback
    with pytest.raises(AttributeError):
        bad = model_cfg()
        bad["encoding"] = "nope"
        LIFFireNet(bad).to(DEV)(x, x)
