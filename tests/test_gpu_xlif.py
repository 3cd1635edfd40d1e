"""XLIF and ALIF FireNets (reference models/model.py:660-681; cells spiking_submodules.py:230-435, :660-875) on the recorded 32-channel
window kernels: the PLIF kernels with the pre-synaptic trace in the THRESHOLD (t0 + t1 * pt') instead of in the current
(include/evflow.h: bit 1 of the PLIF entry points' reset / accumulate flag).  Against the CPU oracle (pinned by the single-cell
goldens G6) and against the same network chained cell by cell on the general path."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from event_flow_amd import synthetic  # noqa: E402
from event_flow_amd.dataloader.encodings import encode_event_list  # noqa: E402
from event_flow_amd.loss import flow as hloss  # noqa: E402
from event_flow_amd.models.model import ALIFFireNet, XLIFFireNet  # noqa: E402
from event_flow_amd.train import FlatAdam, window_backward  # noqa: E402
from oracle import snn as osnn  # noqa: E402
from oracle import train as otrain  # noqa: E402

DEV = "cuda:0"
XLIF_NEURON = {"leak_v": [-4.0, 0.1], "leak_pt": [-2.0, 0.1], "t0": [0.3, 0.05], "t1": [0.5, 0.1], "learn_leak": True,
               "learn_thresh": True, "hard_reset": True}
ALIF_NEURON = {"leak_v": [-4.0, 0.1], "leak_t": [-2.0, 0.1], "t0": [0.3, 0.05], "t1": [0.5, 0.1], "learn_leak": True,
               "learn_thresh": True, "hard_reset": True}
NETS = {"XLIFFireNet": (XLIFFireNet, XLIF_NEURON, "leak_pt"), "ALIFFireNet": (ALIFFireNet, ALIF_NEURON, "leak_t")}


def N(t):
    return t.detach().cpu().numpy()


def cfg(neuron=XLIF_NEURON):
    return {"num_bins": 2, "base_num_channels": 32, "kernel_size": 3, "encoding": "cnt", "norm_input": False,
            "mask_output": True, "activations": ["arctanspike", "arctanspike"], "spiking_neuron": dict(neuron)}


def loss_cfg(H, W):
    return {"loader": {"resolution": [H, W]}, "loss": {"flow_regul_weight": 0.001, "overwrite_intermediate": False},
            "model": {"mask_output": True}}


def _flip_census(model, ref_states):
    got = model.states
    nflip = sum(int((N(got[li][1]) != ref_states[li][1].numpy()).sum()) for li in range(7))
    return nflip, sum(ref_states[li][1].numel() for li in range(7))


@pytest.mark.parametrize("name", ["XLIFFireNet", "ALIFFireNet"])
@pytest.mark.parametrize("shape", [(2, 16, 20), (1, 37, 70)])
def test_xlif_firenet_on_the_fused_engine_vs_oracle(shape, name):
    """Three passes through plain autograd (one fused backward per pass, cell by cell): flows, every state tensor (potential, spikes,
    trace) and every parameter gradient -- t0, t1, both leaks, all weights -- against the oracle.  ALIF: the threshold trace is driven
    by the cell's own previous spikes, whose gradient reaches the pass before (the g_zx path of the fused backward kernels)."""
    B, H, W = shape
    cls, neuron, trace_leak = NETS[name]
    torch.manual_seed(5)
    model = cls(cfg(neuron)).to(DEV)
    assert model._fused() and model.compute_path[0] == "fused"
    params = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    for k, _ in model.named_parameters():
        params[k].requires_grad_(True)
    xs = [(torch.rand(B, 2, H, W) < 0.5).float() * torch.randint(1, 4, (B, 2, H, W)).float() for _ in range(3)]
    states = [None] * 7
    tot_ref, tot = 0, 0
    for x in xs:
        f_ref, states = osnn.firenet_forward(name, params, x, states, hard_reset=True)
        out = model(x.to(DEV), x.to(DEV))
        np.testing.assert_allclose(N(out["flow"][0]), f_ref.detach().numpy(), rtol=1e-4, atol=1e-7)
        tot_ref = tot_ref + (f_ref * torch.arange(f_ref.numel()).view(f_ref.shape).remainder(7)).sum()
        fl = out["flow"][0]
        tot = tot + (fl * torch.arange(fl.numel(), device=DEV).view(fl.shape).remainder(7)).sum()
    for li, st in enumerate(model.states):
        np.testing.assert_allclose(N(st), torch.stack(states[li]).detach().numpy(), rtol=1e-5, atol=2e-6)
    tot.backward()
    tot_ref.backward()
    for k, p in model.named_parameters():
        ref = params[k].grad
        ref = ref.numpy() if ref is not None else np.zeros(tuple(p.shape), np.float32)
        got = N(p.grad) if p.grad is not None else np.zeros_like(ref)
        denom = max(np.linalg.norm(ref), 1e-12)
        assert np.linalg.norm(got - ref) <= 2e-3 * denom + 1e-9, (k, np.linalg.norm(got - ref) / denom)
    for k in ("head.t1", "G1.t1", "R2b.t1", "head.t0", "G2." + trace_leak):  # (the adaptive threshold's own parameters carry signal)
        assert float(np.abs(N(dict(model.named_parameters())[k].grad)).max()) > 0, k


@pytest.mark.parametrize("name", ["XLIFFireNet", "ALIFFireNet"])
def test_xlif_recorded_window_matches_plain_autograd_the_general_path_and_the_oracle(monkeypatch, name):
    """One training window (4 passes, CM loss) three ways on the HIP side -- recorded (train.train_window: forward chains / diagonals,
    backward layer by layer with the window kernels), plain autograd on the fused kernels, and the general path (EVF_XLIF_FUSED=0:
    one conv + neuron kernel per cell) -- and through the oracle's train step: loss and the whole gradient."""
    B, n, H, W, P = 2, 900, 40, 70, 4
    cls, neuron, _ = NETS[name]
    XLIFFireNet = lambda c: cls(cfg(neuron))  # noqa: E731, N806  (the network under test, built from its own neuron configuration)
    torch.manual_seed(3)
    ref_model = XLIFFireNet(cfg()).to(DEV)
    sd = {k: v.detach().clone() for k, v in ref_model.state_dict().items()}
    passes = [encode_event_list(torch.from_numpy(synthetic.event_list_batch(B, n, H, W, 900 + 10 * k)).to(DEV), 2, (H, W)) for k in range(P)]

    def grads_of(model, how):
        model.train()
        lossf = hloss.EventWarping(loss_cfg(H, W), DEV)
        if how == "recorded":
            opt = FlatAdam(model)  # (the parameters' .grad are views of its flat gradient buffer)
            opt.zero_grad()
            loss = window_backward(model, lossf, opt, passes)  # first half of train_window: passes, loss, backward -- no step
            g = {k: N(p.grad).copy() for k, p in model.named_parameters()}
        else:
            for d in passes:
                out = model(d["event_voxel"], d["event_cnt"])
                lossf.event_flow_association(out["flow"], d["event_list"], d["event_list_pol_mask"], d["event_mask"])
            loss = lossf()
            loss.backward()
            g = {k: N(p.grad).copy() for k, p in model.named_parameters()}
        return float(loss.detach()), g

    fused = XLIFFireNet(cfg()).to(DEV)
    fused.load_state_dict(sd)
    assert fused._fused()
    l_plain, g_plain = grads_of(fused, "plain")
    rec = XLIFFireNet(cfg()).to(DEV)
    rec.load_state_dict(sd)
    l_rec, g_rec = grads_of(rec, "recorded")
    monkeypatch.setenv("EVF_XLIF_FUSED", "0")
    monkeypatch.setenv("EVF_PATH_NOTICE", "0")
    gen = XLIFFireNet(cfg()).to(DEV)
    gen.load_state_dict(sd)
    assert not gen._fused() and gen.compute_path[0] == "general"
    l_gen, g_gen = grads_of(gen, "plain")
    monkeypatch.delenv("EVF_XLIF_FUSED")

    params = {k: v.detach().cpu().clone() for k, v in sd.items()}
    keys = [k for k, _ in ref_model.named_parameters()]  # (learn_thresh=True here: t0 / t1 are parameters, not the kind's default buffers)
    opasses = [{k: v.detach().cpu() for k, v in d.items()} for d in passes]
    l_ref, g_ref, _, ostates = otrain.train_step(name, params, keys, opasses, [None] * 7, (H, W), {"step": 0, "m": {}, "v": {}},
                                                 loss_cfg={"flow_regul_weight": 0.001, "mask_output": True},
                                                 model_cfg={"hard_reset": True})
    # same cells, same per-element arithmetic, other launch shapes: recorded == plain to the float atomics of the loss
    assert abs(l_rec - l_plain) <= 1e-6 * abs(l_plain)
    gn = float(np.sqrt(sum(float((g ** 2).sum()) for g in g_plain.values())))
    err = float(np.sqrt(sum(float(((g_rec[k] - g_plain[k]) ** 2).sum()) for k in g_plain)))
    assert err <= 2e-5 * gn, err / gn
    # general path and oracle: other summation orders in the convolutions -- borderline spikes may flip; the census decides the bar
    states = [None] * 7
    with torch.no_grad():
        for d in opasses:
            _, states = osnn.firenet_forward(name, params, d["event_cnt"], states, hard_reset=True)
    nflip, ntot = _flip_census(fused, states)
    assert nflip <= 1e-4 * ntot, (nflip, ntot)
    tol = 2e-3 if nflip == 0 else 5e-2
    np.testing.assert_allclose(l_plain, l_ref, rtol=1e-4 if nflip == 0 else 1e-3)
    np.testing.assert_allclose(l_gen, l_ref, rtol=1e-3)
    gref = float(np.sqrt(sum(float((g.numpy() ** 2).sum()) for g in g_ref.values())))
    for name, g in (("fused", g_plain), ("general", g_gen)):
        e = float(np.sqrt(sum(float(((g[k] - g_ref[k].numpy()) ** 2).sum()) for k in g)))
        assert e <= tol * gref, (name, e / gref, nflip)


@pytest.mark.parametrize("name", ["XLIFFireNet", "ALIFFireNet"])
def test_xlif_alif_hipgraph_replay_is_bitwise_the_eager_step_under_a_deterministic_loss(name):
    """The whole train step of a fused XLIF / ALIF FireNet replayed from hipGraphs (what bench.py's `firenet_family_at_c3_shape` times
    through train.GraphedWindowStep): device-side Adam counter, static state buffers incl. the trace, recorded forward and layer-major
    backward inside the capture.  As tests/test_gpu_network.py::test_hipgraph_replay_is_bitwise_the_eager_step_under_a_deterministic_loss
    for the LIF network: with a loss whose backward is deterministic (the contrast loss sums with float atomics: through it graph and
    eager steps of an alive network drift apart like two eager runs do, 1e-3 .. 1e-2 of the loss after a few steps) two eager + four
    replayed steps leave EXACTLY the parameters, Adam moments and recurrent states (potential, spikes, trace) of six eager steps."""
    from test_gpu_network import _LinearWindowLoss

    from event_flow_amd.train import train_window

    cls, neuron, _ = NETS[name]
    B, n, H, W, P = 2, 600, 32, 64, 3
    pool = [[torch.from_numpy(synthetic.event_list_batch(B, n, H, W, 7100 + 100 * w + k)).to(DEV) for k in range(P)] for w in range(2)]
    gw = torch.Generator(device="cpu").manual_seed(9)
    wts = [(torch.randn(B, 2, H, W, generator=gw) * 0.02).to(DEV) for _ in range(P)]

    def make():
        torch.manual_seed(3)
        m = cls(cfg(neuron)).to(DEV)
        m.train()
        return m

    def step(model, lossf, opt, lists):
        passes = [encode_event_list(ev, 2, (H, W), want=("cnt", "mask", "pol")) for ev in lists]
        for d in passes:
            d["event_voxel"] = None
        return train_window(model, lossf, opt, passes)

    m1 = make()
    assert m1._fused()
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
        for i in range(4):
            graphs[i % 2].replay()
        torch.cuda.synchronize()
    m2 = make()
    opt2 = FlatAdam(m2, lr=2e-4, clip=100.0, device_step=True)
    opt2.zero_grad()
    m2.use_static_states(True)
    l2 = _LinearWindowLoss(wts)
    for i in range(6):
        step(m2, l2, opt2, pool[i % 2])
    torch.cuda.synchronize()
    assert float(opt2.norm_ws[0].sqrt()) < 100.0  # no clipping: the (atomically summed) norm does not enter the update
    assert float(opt1.norm_ws[1]) == 6.0 and float(opt2.norm_ws[1]) == 6.0
    assert torch.equal(opt1.flat_param, opt2.flat_param)
    assert torch.equal(opt1.m, opt2.m) and torch.equal(opt1.v, opt2.v)
    for a, b in zip(m1.states, m2.states):
        assert torch.equal(a, b)
    sd0 = make().state_dict()
    assert any(float((p.detach() - sd0[k].to(DEV)).abs().max()) > 0 for k, p in m1.named_parameters() if k.endswith(("t0", "t1")))  # (the adaptive threshold trained)
