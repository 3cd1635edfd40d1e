"""GPU parity of the event-list kernels (encodings, IWE, CM loss, metrics):
HIP path (through the C ABI) vs the reference-generated golden fixtures and
vs the CPU oracle on seeded inputs.  Integer results bit-exact."""

import numpy as np
import pytest
import torch

from conftest import golden_cases, load_golden

pytestmark = pytest.mark.gpu

from event_flow_amd import _lib, synthetic  # noqa: E402
from event_flow_amd.dataloader import encodings as enc  # noqa: E402
from event_flow_amd.loss import flow as hloss  # noqa: E402
from event_flow_amd.utils import iwe as hiwe  # noqa: E402
from oracle import encodings as oenc  # noqa: E402
from oracle import iwe as oiwe  # noqa: E402
from oracle import loss as oloss  # noqa: E402

DEV = "cuda:0"


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def N(t):
    return t.detach().cpu().numpy()


def cfg(H, W, mask=True, overwrite=False, weight=0.001):
    return {"loader": {"resolution": [H, W]}, "loss": {"flow_regul_weight": weight, "overwrite_intermediate": overwrite}, "model": {"mask_output": mask}}


# ------------------------------------------------------------------ encodings
def test_encodings_golden_bit_exact():
    g = load_golden("g1_encodings")
    xs, ys, ts, ps = (G(g[k]) for k in ("xs", "ys", "ts", "ps"))
    res = tuple(int(v) for v in g["sensor"])
    assert np.array_equal(N(enc.events_to_channels(xs, ys, ps, sensor_size=res)), g["cnt"])
    assert np.array_equal(N(enc.events_to_image(xs, ys, ps.abs(), sensor_size=res, accumulate=False)), g["mask"])
    assert np.array_equal(N(enc.events_to_image(xs, ys, ps, sensor_size=res, accumulate=True)), g["image_acc"])
    for nb in (2, 5):
        got = N(enc.events_to_voxel(xs, ys, ts, ps, nb, sensor_size=res, round_ts=True))
        assert np.array_equal(got, g[f"voxel_nb{nb}_r1"])  # rounded ts: integer valued -> exact
        got = N(enc.events_to_voxel(xs, ys, ts, ps, nb, sensor_size=res, round_ts=False))
        np.testing.assert_allclose(got, g[f"voxel_nb{nb}_r0"], rtol=0, atol=2e-6)  # fp32 atomics reorder the sum


def test_encode_event_list_batched_vs_oracle():
    B, n, H, W = 8, 15000, 128, 128
    ev = synthetic.event_list_batch(B, n, H, W, 1234)
    out = enc.encode_event_list(G(ev), 2, (H, W))
    for b in range(B):
        o = oenc.encode_window(ev[b, :, 2], ev[b, :, 1], ev[b, :, 0], ev[b, :, 3], 2, (H, W))
        assert np.array_equal(N(out["event_cnt"][b]), o["event_cnt"])
        assert np.array_equal(N(out["event_mask"][b]), o["event_mask"])
        assert np.array_equal(N(out["event_list_pol_mask"][b]), o["event_list_pol_mask"].T)
        np.testing.assert_allclose(N(out["event_voxel"][b]), o["event_voxel"], rtol=0, atol=1e-5)
    # checksum of checksums: every event counted exactly once
    assert float(out["event_cnt"].sum()) == B * n


def test_encode_event_lists_all_passes_in_one_launch():
    """encode_event_lists bins the P lists of a window as one batch of P*B samples: bit-identical to P separate
    encode_event_list calls; ragged lists fall back to the per-list path."""
    B, n, H, W = 4, 1500, 64, 48
    lists = [G(synthetic.event_list_batch(B, n, H, W, 900 + k)) for k in range(5)]
    many = enc.encode_event_lists(lists, 2, (H, W), want=("cnt", "mask", "pol"))
    assert len(many) == 5
    for ev, d in zip(lists, many):
        one = enc.encode_event_list(ev, 2, (H, W), want=("cnt", "mask", "pol"))
        assert set(d) == set(one)
        for k in one:
            assert d[k].shape == one[k].shape and d[k].is_contiguous()
            assert torch.equal(d[k], one[k]), k
    ragged = lists[:2] + [lists[2][:, :700].contiguous()]
    out = enc.encode_event_lists(ragged, 2, (H, W), want=("cnt",))
    assert float(out[2]["event_cnt"].sum()) == B * 700 and float(out[0]["event_cnt"].sum()) == B * n


def test_encode_window_in_place_views_equal_the_per_list_encodings():
    """Passes that are the slices [:, p] of one [B,P,N,4] buffer are binned by evf_encode_window: same bits as P
    separate encode_event_list calls; the network inputs are contiguous per pass, and the loss's window tensors
    ([B,P*N,4] events, [B,P*N,2] polarities, [B,P,H,W] masks) are the SAME memory, not copies."""
    from event_flow_amd.loss.flow import _WindowRecord

    B, P, n, H, W = 3, 4, 700, 40, 56
    window = G(np.stack([synthetic.event_list_batch(B, n, H, W, 300 + k) for k in range(P)], 1))  # [B,P,N,4]
    lists = [window[:, p] for p in range(P)]
    assert enc.window_base(lists) is not None
    many = enc.encode_event_lists(lists, 3, (H, W), want=("cnt", "mask", "voxel", "pol"))
    rec = _WindowRecord()
    for p, d in enumerate(many):
        one = enc.encode_event_list(lists[p].contiguous(), 3, (H, W), want=("cnt", "mask", "voxel", "pol"))
        assert set(d) == set(one)
        for k in one:
            assert d[k].shape == one[k].shape, k
            if k == "event_voxel":  # (float atomics: summation order)
                np.testing.assert_allclose(N(d[k]), N(one[k]), rtol=0, atol=1e-5)
            else:
                assert torch.equal(d[k], one[k]), k
        assert d["event_cnt"].is_contiguous() and d["event_voxel"].is_contiguous()
        rec.add([torch.zeros(B, 2, H, W, device=DEV)], d["event_list"], d["event_list_pol_mask"], d["event_mask"])
    ev, pol, ev_pass = rec.packed()
    assert ev.data_ptr() == window.data_ptr() and tuple(ev.shape) == (B, P * n, 4) and ev.is_contiguous()
    assert torch.equal(ev, torch.cat(lists, 1))
    assert pol.data_ptr() == many[0]["event_list_pol_mask"].data_ptr() and tuple(pol.shape) == (B, P * n, 2)
    ms = rec.mask_stack()
    assert ms.data_ptr() == many[0]["event_mask"].data_ptr() and tuple(ms.shape) == (B, P, H, W)
    assert torch.equal(ms, torch.cat([d["event_mask"] for d in many], 1))
    # separate tensors still take the copying path
    sep = [t.clone() for t in lists]
    assert enc.window_base(sep) is None
    rec2 = _WindowRecord()
    for t, d in zip(sep, many):
        rec2.add([torch.zeros(B, 2, H, W, device=DEV)], t, d["event_list_pol_mask"].clone(), d["event_mask"].clone())
    assert torch.equal(rec2.packed()[0], ev) and torch.equal(rec2.mask_stack(), ms)


def test_encodings_edge_cases():
    H, W = 16, 20
    empty = torch.zeros(0, device=DEV)
    assert float(enc.events_to_image(empty, empty, empty, sensor_size=(H, W)).abs().sum()) == 0.0
    # padded (p == 0) rows are ignored
    ev = synthetic.event_list_batch(2, 100, H, W, 5)
    ev[:, 50:, 3] = 0
    out = enc.encode_event_list(G(ev), 3, (H, W))
    assert float(out["event_cnt"].sum()) == 100.0
    with pytest.raises(AssertionError):
        enc.events_to_channels(torch.zeros(3, device=DEV), torch.zeros(2, device=DEV), torch.zeros(3, device=DEV))
    from event_flow_amd._lib import EvflowError
    with pytest.raises(EvflowError):
        enc.events_to_image(torch.zeros(3), torch.zeros(3), torch.zeros(3))  # CPU tensor: no fallback


# ------------------------------------------------------------------ IWE
@pytest.mark.parametrize("tref", [1, 3, 0])
@pytest.mark.parametrize("rnd", [0, 1])
@pytest.mark.parametrize("S", [16, 128])
def test_get_interpolation_golden_bit_exact(tref, rnd, S):
    g = load_golden("g2_interpolation")
    res = tuple(int(v) for v in g["res"])
    idx, w = hiwe.get_interpolation(G(g["events"]), G(g["flow"]), tref, res, S, round_idx=bool(rnd))
    assert np.array_equal(N(idx), g[f"idx_t{tref}_r{rnd}_s{S}"])
    assert np.array_equal(N(w), g[f"w_t{tref}_r{rnd}_s{S}"])


@pytest.mark.parametrize("tag", ["c1", "b2"])
@pytest.mark.parametrize("S", [128, 32])
def test_compute_pol_iwe_golden(tag, S):
    g = load_golden("g3_pol_iwe")
    ev, flow, pol = G(g[tag + "_events"]), G(g[tag + "_flow"]), G(g[tag + "_pol"])
    res = tuple(int(v) for v in g[tag + "_res"])
    got = hiwe.compute_pol_iwe(flow, ev, res, pol[:, :, 0:1], pol[:, :, 1:2], flow_scaling=S, round_idx=True)
    assert np.array_equal(N(got), g[f"{tag}_iwe_s{S}_r1"])  # integer histogram: bit exact
    got = hiwe.compute_pol_iwe(flow, ev, res, pol[:, :, 0:1], pol[:, :, 1:2], flow_scaling=S, round_idx=False)
    np.testing.assert_allclose(N(got), g[f"{tag}_iwe_s{S}_r0"], rtol=0, atol=3e-6)
    # materialising API gives the same image
    idx, w = hiwe.get_interpolation(ev, oloss_gather(flow, ev, res), 1, res, S, round_idx=True)
    pos = hiwe.interpolate(idx, w, res, polarity_mask=pol[:, :, 0:1])
    assert np.array_equal(N(pos[:, 0]), g[f"{tag}_iwe_s{S}_r1"][:, 0])


def oloss_gather(flow, ev, res):
    return G(oiwe.gather_event_flow(N(flow), N(ev), res))


def test_pol_iwe_full_size_vs_oracle_and_properties():
    """BASELINE config 2 shape: B=8, 128x128, 15k events."""
    B, n, H, W = 8, 15000, 128, 128
    ev = synthetic.event_list_batch(B, n, H, W, 2000)
    rng = np.random.default_rng(3)
    flow = rng.uniform(-0.1, 0.1, size=(B, 2, H, W)).astype(np.float32)
    pol = np.stack([(ev[:, :, 3] > 0), (ev[:, :, 3] < 0)], 2).astype(np.float32)
    gpol = G(pol)
    got = N(hiwe.compute_pol_iwe(G(flow), G(ev), (H, W), gpol[:, :, 0:1], gpol[:, :, 1:2], flow_scaling=128, round_idx=True))
    ref = oiwe.compute_pol_iwe(flow, ev, (H, W), pol[:, :, 0:1], pol[:, :, 1:2], flow_scaling=128, round_idx=True)
    assert np.array_equal(got, ref)
    # zero flow => IWE == event count image (loss/flow.py:491-494)
    z = N(hiwe.compute_pol_iwe(G(flow * 0), G(ev), (H, W), gpol[:, :, 0:1], gpol[:, :, 1:2], round_idx=True))
    cnt = N(enc.encode_event_list(G(ev), 2, (H, W))["event_cnt"])
    assert np.array_equal(z, cnt)
    # bilinear weights of in-image events sum to 1: total mass <= number of events, equal for zero flow
    bl = hiwe.compute_pol_iwe(G(flow * 0), G(ev), (H, W), gpol[:, :, 0:1], gpol[:, :, 1:2], round_idx=False)
    assert float(bl.sum()) == B * n


@pytest.mark.parametrize("one", ["1", "0"])
def test_pol_iwe_one_launch_kernel_small_batches_bit_exact(one, monkeypatch):
    """EVF_IWE_ONE=1 selects k_iwe_splat_one (csrc/evf_events.hip; opt-in: measured no faster than fill + scatter, which "0" --
    the default -- runs on the same cases): B * M < 400 k events, rounded indices -- one entry per event written into the
    output's own memory, the image built in LDS by the sample's last block, no zero-fill launch, no global atomics.  Against
    the oracle (utils/iwe.py:95-153) bit for bit on integer histograms: ragged event counts (M not a multiple of 1024, M < 1024,
    M = the 16 Ki limit), 1 and 2 channels, small and non-square images, an output buffer full of garbage, the same call
    repeated (the ticket words are left at zero), large flows (many events leave the image), zero flow; general (non 0 / 1)
    weights through its fp32 LDS planes (sums of multiples of 1/8: exact in any order)."""
    monkeypatch.setenv("EVF_IWE_ONE", one)  # (read by the library per call)
    rng = np.random.default_rng(11)
    for B, n, H, W, amp in [(8, 15000, 128, 128, 0.1), (3, 1000, 64, 64, 0.3), (1, 777, 32, 48, 0.05), (2, 16384, 128, 128, 2.0),
                            (5, 4097, 96, 160, 0.2)]:
        ev = synthetic.event_list_batch(B, n, H, W, 3100 + n)
        flow = rng.uniform(-amp, amp, size=(B, 2, H, W)).astype(np.float32)
        pol = np.stack([(ev[:, :, 3] > 0), (ev[:, :, 3] < 0)], 2).astype(np.float32)
        gpol, gev, gfl = G(pol), G(ev), G(flow)
        ref = oiwe.compute_pol_iwe(flow, ev, (H, W), pol[:, :, 0:1], pol[:, :, 1:2], flow_scaling=128, round_idx=True)
        for rep in range(3):  # (the caching allocator hands back the previous result's memory: garbage the kernel must not count)
            got = N(hiwe.compute_pol_iwe(gfl, gev, (H, W), gpol[:, :, 0:1], gpol[:, :, 1:2], flow_scaling=128, round_idx=True))
            assert np.array_equal(got, ref), (B, n, H, W, rep, int((got != ref).sum()))
        assert ref.sum() > 0 and (amp < 1.0 or ref.sum() < 0.9 * B * n)  # (large flows: events do leave the image)
        # one channel, no weights / a polarity mask (deblur_events)
        one = N(hiwe.deblur_events(gfl, gev, (H, W), flow_scaling=128, round_idx=True))
        assert np.array_equal(one, oiwe.deblur_events(flow, ev, (H, W), 128, True))
        onep = N(hiwe.deblur_events(gfl, gev, (H, W), flow_scaling=128, round_idx=True, polarity_mask=gpol[:, :, 0:1]))
        assert np.array_equal(onep, oiwe.deblur_events(flow, ev, (H, W), 128, True, pol[:, :, 0:1]))
        # zero flow: the event count image
        z = N(hiwe.compute_pol_iwe(G(flow * 0), gev, (H, W), gpol[:, :, 0:1], gpol[:, :, 1:2], round_idx=True))
        assert np.array_equal(z, N(enc.encode_event_list(gev, 2, (H, W))["event_cnt"]))
        # general weights (multiples of 1/8 in [0, 4)): the fp32 planes of the same kernel, exact sums
        wts = (rng.integers(0, 32, size=(B, n, 2)) / 8.0).astype(np.float32)
        gw = G(wts)
        gotw = N(hiwe.compute_pol_iwe(gfl, gev, (H, W), gw[:, :, 0:1], gw[:, :, 1:2], flow_scaling=128, round_idx=True))
        refw = oiwe.compute_pol_iwe(flow, ev, (H, W), wts[:, :, 0:1], wts[:, :, 1:2], flow_scaling=128, round_idx=True)
        assert np.array_equal(gotw, refw), (B, n, H, W)


# ------------------------------------------------------------------ CM loss
def _run_hip_loss(g, c, H, W):
    tag = c["tag"]
    lossf = hloss.EventWarping(cfg(H, W, c["mask"], c["overwrite"]), DEV)
    flows = []
    for k in range(c["P"]):
        fl = [G(g[f"{tag}_p{k}_flow{s}"]).requires_grad_(True) for s in range(c["scales"])]
        flows.append(fl)
        ev = G(g[f"{tag}_p{k}_event_list"])
        ev0 = ev.clone()
        lossf.event_flow_association(fl, ev, G(g[f"{tag}_p{k}_event_list_pol_mask"]), G(g[f"{tag}_p{k}_event_mask"]))
        assert torch.equal(ev, ev0)  # caller's tensor is not mutated
    if c["overwrite"]:
        lossf.overwrite_intermediate_flow(flows[-1])
    val = lossf()
    val.backward()
    return val, flows


@pytest.mark.parametrize("splat", ["atomics", "lds", "lds-one-launch-per-pass", "lds+bwd-stripes"])
def test_event_warping_golden_loss_and_grad(splat, monkeypatch):
    # both image-accumulation paths of evf_cm_loss_fwd: device-scope atomics, and LDS stripes over pre-warped events -- the
    # latter with the loss launches merged (default: 2 forward + 2 backward launches, the last splat block finishes the loss)
    # and one launch per pass (evf_cm_merge(0): fill, pre-warp, splat, reduce, smooth, finalize / 3 backward)
    monkeypatch.setattr(hloss, "CM_LDS_MIN_EVENTS", 1 if splat.startswith("lds") else 1 << 60)
    assert _lib.load().evf_cm_merge(0 if splat.endswith("per-pass") else 1) == 0
    # dL/dflow of the events: device-scope atomics (what these small cases take by size), or k_cm_event_bwd_lds forced
    assert _lib.load().evf_cm_bwd_lds(1 if splat.endswith("bwd-stripes") else 0) == 0
    try:
        _golden_loss_and_grad()
    finally:
        _lib.load().evf_cm_merge(1)
        _lib.load().evf_cm_bwd_lds(-1)


def _golden_loss_and_grad():
    g = load_golden("g4_event_warping")
    H, W = (int(v) for v in g["res"])
    for c in golden_cases(g):
        val, flows = _run_hip_loss(g, c, H, W)
        np.testing.assert_allclose(float(val.detach()), float(g[c["tag"] + "_loss"]), rtol=1e-5, err_msg=str(c))
        for k in range(c["P"]):
            for s in range(c["scales"]):
                ref = g[f"{c['tag']}_p{k}_gflow{s}"]
                got = flows[k][s].grad
                got = N(got) if got is not None else np.zeros_like(ref)
                # fp32 atomics reorder the image sums; the gradient divides by small IWE
                # values, so element-wise noise reaches ~1e-4 of the max (the reference's own
                # fp32 gradient is 1e-4..1e-2 away from a float64 evaluation on these cases)
                scale = max(np.abs(ref).max(), 1e-12)
                assert np.abs(got - ref).max() <= 1e-3 * scale + 1e-9, (c, k, s, np.abs(got - ref).max(), scale)
                assert np.linalg.norm(got - ref) <= 2e-4 * np.linalg.norm(ref) + 1e-9, (c, k, s)


@pytest.mark.parametrize("P,n,bwd", [(1, 15000, -1), (10, 1500, -1), (1, 15000, 0), (10, 1500, 1)])
def test_event_warping_full_size_vs_oracle(P, n, bwd):
    """config 2 shape (B=8, 128x128, 15k events per window) against the oracle.  bwd: evf_cm_bwd_lds -- by size ((1, 15000) takes
    the LDS stripes, (10, 1500) the atomics), and each case once more on the OTHER path."""
    _lib.load().evf_cm_bwd_lds(bwd)
    try:
        _full_size_vs_oracle(P, n)
    finally:
        _lib.load().evf_cm_bwd_lds(-1)


def test_event_gradient_stripes_match_atomics_on_ragged_shapes():
    """k_cm_event_bwd_lds against k_cm_event_bwd: ragged image sizes (last stripe short, W not a multiple of anything), several
    scales, per-pass maps and the overwritten single map, events crowded into a few rows (a queue fuller than a chunk's share)."""
    lib = _lib.load()
    for (B, H, W, P, n, S, overwrite, crowd) in ((3, 37, 53, 2, 9000, 2, False, False), (2, 70, 41, 3, 5000, 1, True, True),
                                                 (1, 256, 256, 1, 50000, 4, False, False), (2, 33, 300, 1, 12000, 1, False, True)):
        grads = []
        for mode in (0, 1):
            lib.evf_cm_bwd_lds(mode)
            try:
                rng = np.random.default_rng(5)
                c = cfg(H, W)
                c["loss"]["overwrite_intermediate"] = overwrite
                lossf = hloss.EventWarping(c, DEV)
                fls = []
                for k in range(P):
                    ev = synthetic.event_list_batch(B, n, H, W, 300 + k)
                    if crowd:
                        ev[:, :, 1] = np.floor(ev[:, :, 1] / H * 5.0) + (H - 6)  # rows H - 6 .. H - 2 only
                    ev = G(ev)
                    pol = torch.stack([(ev[:, :, 3] > 0).float(), (ev[:, :, 3] < 0).float()], 2).contiguous()
                    fl = [G(rng.uniform(-0.1, 0.1, size=(B, 2, H, W)).astype(np.float32)).requires_grad_(True) for _ in range(S)]
                    fls.append(fl)
                    lossf.event_flow_association(fl, ev, pol, torch.ones(B, 1, H, W, device=DEV))
                if overwrite:
                    lossf.overwrite_intermediate_flow(fls[-1])
                lossf().backward()
                grads.append([N(f.grad) if f.grad is not None else None for fl in fls for f in fl])
            finally:
                lib.evf_cm_bwd_lds(-1)
        for a, b in zip(*grads):
            assert (a is None) == (b is None)
            if a is not None:  # (the same per-event terms, summed in another order: float round-off of the sums)
                assert np.abs(a - b).max() <= 1e-5 * max(np.abs(a).max(), 1e-12), (B, H, W, P, n, S, np.abs(a - b).max(), np.abs(a).max())


def _full_size_vs_oracle(P, n):
    B, H, W = 8, 128, 128
    rng = np.random.default_rng(100 + P)
    lossf = hloss.EventWarping(cfg(H, W), DEV)
    win = oloss.Window((H, W))
    gflows, oflows = [], []
    for k in range(P):
        ev = synthetic.event_list_batch(B, n, H, W, 7000 + 100 * k)
        d = oenc.collate([oenc.encode_window(ev[b, :, 2], ev[b, :, 1], ev[b, :, 0], ev[b, :, 3], 2, (H, W)) for b in range(B)])
        f = rng.uniform(-0.05, 0.05, size=(B, 2, H, W)).astype(np.float32)
        gf, of = G(f).requires_grad_(True), torch.from_numpy(f).requires_grad_(True)
        gflows.append(gf)
        oflows.append(of)
        lossf.event_flow_association([gf], G(d["event_list"]), G(d["event_list_pol_mask"]), G(d["event_mask"]))
        win.add([of], torch.from_numpy(d["event_list"]), torch.from_numpy(d["event_list_pol_mask"]), torch.from_numpy(d["event_mask"]))
    val = lossf()
    val.backward()
    ref = oloss.event_warping_loss(win, max(H, W), 0.001)
    ref.backward()
    np.testing.assert_allclose(float(val.detach()), float(ref.detach()), rtol=2e-5)
    for gf, of in zip(gflows, oflows):
        r = of.grad.numpy()
        # element-wise fp32 noise (atomics order, division by small IWE values); tight in L2
        assert np.abs(N(gf.grad) - r).max() <= 1e-3 * np.abs(r).max()
        assert np.linalg.norm(N(gf.grad) - r) <= 2e-4 * np.linalg.norm(r)


def test_demo_iwe_known_answer():
    """tools/demo_iwe.py:69-91 idea: for events generated by one constant
    motion, the loss over a (u,v) grid is minimal at the true motion."""
    H = W = 64
    n, B = 4000, 1
    xs, ys, ts, ps, (u, v) = synthetic.moving_dots_events(n, H, W, 99, max_disp=6.0, k=60)
    d = oenc.collate([oenc.encode_window(xs, ys, ts, ps, 2, (H, W))])
    ev, pol, mask = G(d["event_list"]), G(d["event_list_pol_mask"]), G(d["event_mask"])
    best, arg = None, None
    grid = np.arange(-8, 9, 2.0)
    for uu in grid:
        for vv in grid:
            lossf = hloss.EventWarping(cfg(H, W, mask=False, weight=0.0), DEV)
            flow = torch.zeros(B, 2, H, W, device=DEV)
            flow[:, 0], flow[:, 1] = uu / 64.0, vv / 64.0
            lossf.event_flow_association([flow], ev, pol, mask)
            val = float(lossf())
            if best is None or val < best:
                best, arg = val, (uu, vv)
    assert abs(arg[0] - u) <= 2.0 and abs(arg[1] - v) <= 2.0, (arg, (u, v))


# ------------------------------------------------------------------ metrics
@pytest.mark.parametrize("ow", [0, 1])
def test_metrics_golden(ow):
    g = load_golden("g5_metrics")
    H, W = (int(v) for v in g["res"])
    P = int(g["P"])
    tag = f"ow{ow}"
    c = cfg(H, W, overwrite=bool(ow))
    ms = [hloss.FWL(c, DEV, flow_scaling=32), hloss.RSAT(c, DEV, flow_scaling=32)]
    last = None
    for k in range(P):
        last = G(g[f"{tag}_p{k}_flow"])
        inputs = {
            "event_list": G(g[f"{tag}_p{k}_event_list"]), "event_list_pol_mask": G(g[f"{tag}_p{k}_event_list_pol_mask"]),
            "event_mask": G(g[f"{tag}_p{k}_event_mask"]), "gtflow": G(g[f"{tag}_p{k}_gtflow"]),
            "dt_input": torch.tensor([1.0]), "dt_gt": torch.tensor([1.0]),
        }
        for m in ms:
            m.event_flow_association([last], inputs)
    if ow:
        for m in ms:
            m.overwrite_intermediate_flow([last])
    np.testing.assert_allclose(N(ms[0]()), g[tag + "_fwl"], rtol=1e-5)
    np.testing.assert_allclose(N(ms[1]()), g[tag + "_rsat"], rtol=1e-5)
    assert np.array_equal(N(ms[0].compute_window_events()), g[tag + "_window_events"])
    assert np.array_equal(N(ms[0].compute_window_iwe()), g[tag + "_window_iwe"])
    np.testing.assert_allclose(N(ms[0].compute_masked_window_flow()), g[tag + "_masked_flow"], rtol=1e-5, atol=1e-7)


def test_aee_golden():
    g = load_golden("g5_metrics")
    H, W = (int(v) for v in g["res"])
    m = hloss.AEE(cfg(H, W), DEV, flow_scaling=32)
    assert m.num_events == float("inf")
    inputs = {
        "event_list": torch.zeros(1, 4, 4, device=DEV), "event_list_pol_mask": torch.zeros(1, 4, 2, device=DEV),
        "event_mask": G(g["aee_event_mask"]), "gtflow": G(g["aee_gt"]),
        "dt_input": torch.tensor([float(g["aee_dt"][1])]), "dt_gt": torch.tensor([float(g["aee_dt"][0])]),
    }
    m.event_flow_association([G(g["aee_flow"])], inputs)
    a, p = m()
    np.testing.assert_allclose(N(a), g["aee_val"], rtol=1e-5)
    np.testing.assert_allclose(N(p), g["aee_outl"], rtol=1e-5)


def test_pol_iwe_large_batch_lds_path_bit_exact():
    """B*N >= 400k events selects the LDS-privatised splat kernel: same integer histogram,
    in both rounding modes, with and without polarity weights, on a non-square sensor."""
    B, n, H, W = 30, 15000, 96, 160
    ev = synthetic.event_list_batch(B, n, H, W, 9000)
    rng = np.random.default_rng(4)
    flow = rng.uniform(-0.15, 0.15, size=(B, 2, H, W)).astype(np.float32)
    pol = np.stack([(ev[:, :, 3] > 0), (ev[:, :, 3] < 0)], 2).astype(np.float32)
    gpol = G(pol)
    got = N(hiwe.compute_pol_iwe(G(flow), G(ev), (H, W), gpol[:, :, 0:1], gpol[:, :, 1:2], flow_scaling=128, round_idx=True))
    ref = oiwe.compute_pol_iwe(flow, ev, (H, W), pol[:, :, 0:1], pol[:, :, 1:2], flow_scaling=128, round_idx=True)
    assert np.array_equal(got, ref)
    got = N(hiwe.compute_pol_iwe(G(flow), G(ev), (H, W), gpol[:, :, 0:1], gpol[:, :, 1:2], flow_scaling=128, round_idx=False))
    ref = oiwe.compute_pol_iwe(flow, ev, (H, W), pol[:, :, 0:1], pol[:, :, 1:2], flow_scaling=128, round_idx=False)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5)
    one = N(hiwe.deblur_events(G(flow), G(ev), (H, W), flow_scaling=128, round_idx=True))
    assert np.array_equal(one[:, 0], got.shape and (oiwe.deblur_events(flow, ev, (H, W), 128, True))[:, 0])


@pytest.mark.parametrize("shape", [(64, 15000, 128, 128), (70, 9000, 96, 160)])
def test_pol_iwe_register_resident_kernel_bit_exact(shape):
    """B >= 64, one flow map, plane <= 64 KiB, <= 15 Ki events per sample selects k_iwe_splat_reg (events in
    registers, flow planes and the image through one reused LDS plane).  Binary polarity masks take the packed
    16-bit integer-atomic path, other weights the float path; both must reproduce the oracle's histogram exactly
    (weights that are multiples of 1/4 keep every partial sum exact), including events warped out of the image."""
    B, n, H, W = shape
    ev = synthetic.event_list_batch(B, n, H, W, 9100)
    rng = np.random.default_rng(5)
    flow = rng.uniform(-0.6, 0.6, size=(B, 2, H, W)).astype(np.float32)  # up to 77 px: many events leave the image
    pol = np.stack([(ev[:, :, 3] > 0), (ev[:, :, 3] < 0)], 2).astype(np.float32)
    gpol = G(pol)
    got = N(hiwe.compute_pol_iwe(G(flow), G(ev), (H, W), gpol[:, :, 0:1], gpol[:, :, 1:2], flow_scaling=128, round_idx=True))
    ref = oiwe.compute_pol_iwe(flow, ev, (H, W), pol[:, :, 0:1], pol[:, :, 1:2], flow_scaling=128, round_idx=True)
    assert np.array_equal(got, ref)
    assert got.sum() < B * n  # some events did leave
    wts = pol * rng.integers(1, 9, size=pol.shape).astype(np.float32) * 0.25
    gw = G(wts)
    got = N(hiwe.compute_pol_iwe(G(flow), G(ev), (H, W), gw[:, :, 0:1], gw[:, :, 1:2], flow_scaling=128, round_idx=True))
    ref = oiwe.compute_pol_iwe(flow, ev, (H, W), wts[:, :, 0:1], wts[:, :, 1:2], flow_scaling=128, round_idx=True)
    assert np.array_equal(got, ref)
    one = N(hiwe.deblur_events(G(flow), G(ev), (H, W), flow_scaling=128, round_idx=True))
    assert np.array_equal(one, oiwe.deblur_events(flow, ev, (H, W), 128, True))


def test_encode_window_edge_cases():
    """evf_encode_window: subsets of the outputs, a single pass, padding events (p = 0), out-of-image coordinates, no
    events at all, and loud failures for the wrong layout."""
    B, P, n, H, W = 2, 3, 50, 12, 16
    rng = np.random.default_rng(3)
    ev = np.zeros((B, P, n, 4), np.float32)
    ev[..., 0] = rng.uniform(0, 1, (B, P, n))
    ev[..., 1] = rng.integers(-2, H + 2, (B, P, n))  # some rows outside the image
    ev[..., 2] = rng.integers(0, W, (B, P, n))
    ev[..., 3] = rng.choice([-1.0, 0.0, 1.0], (B, P, n))  # p = 0: padding
    evd = G(ev)
    for want in (("cnt",), ("mask", "pol"), ("voxel",), ("cnt", "mask", "voxel", "pol")):
        many = enc.encode_window(evd, 4, (H, W), want=want)
        for p, d in enumerate(many):
            one = enc.encode_event_list(evd[:, p].contiguous(), 4, (H, W), want=want)
            assert set(d) == set(one), want
            for k in one:
                if k == "event_voxel":
                    np.testing.assert_allclose(N(d[k]), N(one[k]), rtol=0, atol=1e-5)
                else:
                    assert torch.equal(d[k], one[k]), (want, k)
    single = enc.encode_window(evd[:, :1].contiguous(), 2, (H, W), want=("cnt", "mask", "pol"))
    assert len(single) == 1 and torch.equal(single[0]["event_cnt"], enc.encode_event_list(evd[:, 0].contiguous(), 2, (H, W))["event_cnt"])
    empty = enc.encode_window(torch.zeros(B, P, 0, 4, device=DEV), 2, (H, W), want=("cnt", "mask", "pol"))
    assert float(empty[0]["event_cnt"].abs().sum()) == 0 and tuple(empty[2]["event_list_pol_mask"].shape) == (B, 0, 2)
    with pytest.raises(_lib.EvflowError):
        enc.encode_window(evd.permute(1, 0, 2, 3), 2, (H, W))  # pass-major view: not contiguous
    with pytest.raises(_lib.EvflowError):
        enc.encode_window(evd.cpu(), 2, (H, W))
