"""Kernel-level equivalences through the C ABI: every fused / merged entry point of the 32->32 spiking stack against
the separate entry points it replaces, on random tensors (incl. a ragged shape).  State tensors must agree bit for bit;
sums that are taken in another order (weight-gradient slabs, per-channel parameter gradients) to fp32 round-off."""

import numpy as np
import pytest
import torch

from event_flow_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
C = 32
P = lambda t: t.data_ptr() if t is not None else None  # noqa: E731


def _f(*s, scale=1.0):
    return torch.randn(*s, device=DEV) * scale


def _bits(B, H, W, rate=0.15):
    z = torch.rand(B, H, W, 32, device=DEV) < rate
    w = torch.zeros(B, H, W, dtype=torch.int64, device=DEV)
    for c in range(32):
        w |= z[..., c].long() << c
    return torch.where(w < 2**31, w, w - 2**32).to(torch.int32).contiguous()


def _planes(bits):
    B, H, W = bits.shape
    t = torch.empty(B, H, 32, (W + 31) // 32, dtype=torch.int32, device=DEV)
    _lib.call("evf_bits_transpose", P(bits), B, H, W, P(t))
    return t


def _packs():
    w = _f(32, 32, 3, 3, scale=0.1)
    fwd = torch.empty(54 * 1024, dtype=torch.uint8, device=DEV)
    bwd = torch.empty(54 * 1024, dtype=torch.uint8, device=DEV)
    _lib.call("evf_pack_conv_weight_b3", P(w), 32, 32, P(fwd))
    _lib.call("evf_pack_conv_weight_b3t", P(w), 32, 32, P(bwd))
    return fwd, bwd


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


SHAPES = [(8, 128, 128), (2, 37, 50)]


@pytest.mark.parametrize("shape", SHAPES)
def test_input_gradient_variants_are_bit_identical(shape):
    """evf_conv_dgrad_b3 (pre-split planes) == evf_conv_dgrad_b3_f32 (split while staging), through BOTH kernels behind the
    latter (0: one-phase-after-the-other LDS kernel, 1: wave-specialised producer / consumer kernel); the _pair form == two
    calls; the PLIF term (pooled-trace gradient on the input spikes) through both kernels."""
    B, H, W = shape
    torch.manual_seed(1)
    g = _f(B, H, W, C, scale=0.3)
    _, wt1 = _packs()
    _, wt2 = _packs()
    # the split the fused backward would have written (it also returns g_cur): reuse it from a dummy launch
    z = _bits(B, H, W)
    xT = _planes(z)
    nsl = _lib.load().evf_lif_bwd_wgrad_slabs(B, H, W)
    slab = torch.zeros(nsl, 9216, device=DEV)
    leak, thresh = _f(32, scale=0.1) - 1, _f(32, scale=0.1) + 0.8
    gsp = torch.empty(3, B, H, W, C, dtype=torch.bfloat16, device=DEV)
    gcur, gvp = torch.empty(B, H, W, C, device=DEV), torch.empty(B, H, W, C, device=DEV)
    gl, gt = torch.zeros(32, device=DEV), torch.zeros(32, device=DEV)
    _lib.call("evf_lif_bwd_wgrad", P(g), None, P(_f(B, H, W, C)), None, None, P(xT), None, P(leak), P(thresh), B, H, W, 1, 0, 10.0,
              P(gcur), P(gsp), P(gvp), P(gl), P(gt), P(slab), None, 0)
    base = _f(B, H, W, C)
    gP, xb = _f(B, H, W), _bits(B, H, W)
    try:
        for which in (0, 1):
            assert _lib.load().evf_conv_dgrad_select(which) == 0
            for acc in (0, 1):
                a, b = base.clone(), base.clone()
                _lib.call("evf_conv_dgrad_b3", P(gsp), P(wt1), P(a), acc, B, H, W, None, None)
                _lib.call("evf_conv_dgrad_b3_f32", P(gcur), P(wt1), P(b), acc, B, H, W, None, None)
                assert torch.equal(a, b), (which, acc)
                c, d = base.clone(), torch.empty(B, H, W, C, device=DEV)
                _lib.call("evf_conv_dgrad_b3_f32_pair", P(gcur), P(wt1), P(c), acc, P(wt2), P(d), B, H, W, None, None)
                e = torch.empty(B, H, W, C, device=DEV)
                _lib.call("evf_conv_dgrad_b3_f32", P(gcur), P(wt2), P(e), 0, B, H, W, None, None)
                assert torch.equal(c, b) and torch.equal(d, e), (which, acc)
                a, b = base.clone(), base.clone()
                _lib.call("evf_conv_dgrad_b3", P(gsp), P(wt1), P(a), acc, B, H, W, P(gP), P(xb))
                _lib.call("evf_conv_dgrad_b3_f32", P(gcur), P(wt1), P(b), acc, B, H, W, P(gP), P(xb))
                assert torch.equal(a, b), (which, acc, "plif")
    finally:
        _lib.load().evf_conv_dgrad_select(-1)
    assert _lib.load().evf_conv_dgrad_select(7) != 0  # bad argument: status, no change


def _gcur_and_split(B, H, W):
    """dL/d(current) as the fused backward writes it: the fp32 tensor AND its exact 3-way bf16 split (three planes)."""
    g = _f(B, H, W, C, scale=0.3)
    xT = _planes(_bits(B, H, W))
    nsl = _lib.load().evf_lif_bwd_wgrad_slabs(B, H, W)
    slab = torch.zeros(nsl, 9216, device=DEV)
    leak, thresh = _f(32, scale=0.1) - 1, _f(32, scale=0.1) + 0.8
    gsp = torch.empty(3, B, H, W, C, dtype=torch.bfloat16, device=DEV)
    gcur, gvp = torch.empty(B, H, W, C, device=DEV), torch.empty(B, H, W, C, device=DEV)
    gl, gt = torch.zeros(32, device=DEV), torch.zeros(32, device=DEV)
    _lib.call("evf_lif_bwd_wgrad", P(g), None, P(_f(B, H, W, C)), None, None, P(xT), None, P(leak), P(thresh), B, H, W, 1, 0, 10.0,
              P(gcur), P(gsp), P(gvp), P(gl), P(gt), P(slab), None, 0)
    return gcur, gsp


@pytest.mark.parametrize("shape", SHAPES + [(3, 20, 96)])
def test_recorded_input_gradient_cells_are_bit_identical(shape):
    """The input-gradient cells of a backward index, recorded (evf_bwd_defer_*) and launched together, against one direct
    launch per cell -- bit-identical outputs through all three dispatchers: k_dgrad_diag (the LDS kernel's body),
    k_dgrad_diag_ws (persistent producer / consumer blocks over the flat product list, fp32 gradient in) and
    k_dgrad_diag_dma (gradient pre-split by the fused backward, staged by LDS-DMA, one wave per SIMD, alternating
    accumulators, DPP-built middle taps).  Several cells per index, one- and two-product cells mixed, two indices, so that
    block ranges cross product boundaries (weight-set switches inside a block); ragged shapes (out-of-image halo = zero page)."""
    B, H, W = shape
    torch.manual_seed(5)
    L = _lib.load()
    cells = []  # (index, fp32 gradient, its split planes, weight set 1, weight set 2 or None)
    for d, pair in ((0, False), (0, True), (0, False), (1, True), (1, True), (1, False), (1, False), (1, True)):
        cells.append((d,) + _gcur_and_split(B, H, W) + (_packs()[1], _packs()[1] if pair else None))
    ref = []
    for _, g, _gs, w1, w2 in cells:
        a = torch.empty(B, H, W, C, device=DEV)
        _lib.call("evf_conv_dgrad_b3_f32", P(g), P(w1), P(a), 0, B, H, W, None, None)
        b = None
        if w2 is not None:
            b = torch.empty(B, H, W, C, device=DEV)
            _lib.call("evf_conv_dgrad_b3_f32", P(g), P(w2), P(b), 0, B, H, W, None, None)
        ref.append((a, b))
    try:
        for which, split in ((0, False), (1, False), (1, True), (2, True)):  # (2: the ring-halo kernel on the pre-split planes)
            assert L.evf_dgrad_diag_select(which) == 0
            outs = [(torch.full((B, H, W, C), 7.0, device=DEV), torch.full((B, H, W, C), 7.0, device=DEV)) for _ in cells]
            assert _lib.raw("evf_bwd_defer_begin") == 0
            try:
                for (d, g, gs, w1, w2), (a, b) in zip(cells, outs):
                    assert _lib.raw("evf_bwd_defer_slot", d) == 0
                    src = gs if split else g
                    if w2 is None:
                        _lib.call("evf_conv_dgrad_b3" if split else "evf_conv_dgrad_b3_f32", P(src), P(w1), P(a), 0, B, H, W, None, None)
                    else:
                        _lib.call("evf_conv_dgrad_b3_pair" if split else "evf_conv_dgrad_b3_f32_pair", P(src), P(w1), P(a), 0, P(w2), P(b),
                                  B, H, W, None, None)
                assert _lib.raw("evf_bwd_defer_pending") == len(cells)
            finally:
                _lib.call("evf_bwd_defer_flush")
            assert _lib.raw("evf_bwd_defer_pending") == 0
            torch.cuda.synchronize()
            for k, ((a, b), (ra, rb)) in enumerate(zip(outs, ref)):
                assert torch.equal(a, ra), (which, split, k)
                if rb is not None:
                    assert torch.equal(b, rb), (which, split, k, "second product")
    finally:
        L.evf_dgrad_diag_select(-1)
    assert L.evf_dgrad_diag_select(5) != 0


@pytest.mark.parametrize("shape", SHAPES)
def test_prediction_head_fused_into_its_neighbours(shape):
    """evf_conv_lif_fwd_b3_pred == evf_conv_lif_fwd_b3 + evf_pred_fwd; evf_lif_bwd_wgrad_top == evf_pred_bwd +
    evf_lif_bwd_wgrad."""
    B, H, W = shape
    torch.manual_seed(2)
    x, zp = _bits(B, H, W), _bits(B, H, W)
    wf, _ = _packs()
    leak, thresh = _f(32, scale=0.1) - 1, _f(32, scale=0.1) + 0.3
    v = _f(B, H, W, C, scale=0.5)
    pw, pb = _f(2, 32, scale=0.05), _f(2, scale=0.01)
    nW = (W + 31) // 32
    outs = []
    for fused in (False, True):
        vo, zo = torch.empty(B, H, W, C, device=DEV), torch.empty(B, H, W, dtype=torch.int32, device=DEV)
        zT = torch.empty(B, H, 32, nW, dtype=torch.int32, device=DEV)
        flow = torch.empty(B, 2, H, W, device=DEV)
        if fused:
            _lib.call("evf_conv_lif_fwd_b3_pred", P(x), P(wf), None, P(leak), P(thresh), P(v), P(zp), B, H, W, 1, P(vo), P(zo), P(zT),
                      P(pw), P(pb), P(flow))
        else:
            _lib.call("evf_conv_lif_fwd_b3", P(x), P(wf), None, P(leak), P(thresh), P(v), P(zp), B, H, W, 1, P(vo), P(zo), P(zT))
            _lib.call("evf_pred_fwd", P(zo), P(pw), P(pb), B, H, W, P(flow))
        outs.append((vo, zo, zT, flow))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    vo, zo, zT, flow = outs[0]
    assert float((zo != 0).float().mean()) > 0.5  # the layer does spike
    # backward
    g_flow, gv = _f(B, 2, H, W), _f(B, H, W, C, scale=0.1)
    xT = _planes(x)
    nsl = _lib.load().evf_lif_bwd_wgrad_slabs(B, H, W)
    res = []
    for fused in (False, True):
        slab = torch.zeros(nsl, 9216, device=DEV)
        gcur, gvp = torch.empty(B, H, W, C, device=DEV), torch.empty(B, H, W, C, device=DEV)
        gl, gt = torch.zeros(32, device=DEV), torch.zeros(32, device=DEV)
        dw, db = torch.zeros(2, 32, device=DEV), torch.zeros(2, device=DEV)
        if fused:
            _lib.call("evf_lif_bwd_wgrad_top", P(flow), P(g_flow), P(pw), P(zo), P(dw), P(db), P(gv), P(vo), P(v), P(zp), P(xT), P(leak),
                      P(thresh), B, H, W, 1, 0, 10.0, P(gcur), None, P(gvp), P(gl), P(gt), P(slab), 0)
        else:
            gz = torch.empty(B, H, W, C, device=DEV)
            _lib.call("evf_pred_bwd", P(zo), P(flow), P(g_flow), P(pw), B, H, W, P(gz), P(dw), P(db))
            _lib.call("evf_lif_bwd_wgrad", P(gz), P(gv), P(vo), P(v), P(zp), P(xT), None, P(leak), P(thresh), B, H, W, 1, 0, 10.0,
                      P(gcur), None, P(gvp), P(gl), P(gt), P(slab), None, 0)
        res.append((gcur, gvp, slab.sum(0), gl, gt, dw, db))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for k in range(2, 7):
        assert _rel(res[1][k], res[0][k]) < 5e-6, k




@pytest.mark.parametrize("shape,cin,hard,fresh", [((8, 128, 128), 2, 1, True), ((2, 37, 50), 2, 1, False), ((2, 37, 50), 4, 0, True),
                                                    ((1, 64, 96), 3, 1, False)])
def test_head_layer_of_a_window_in_one_launch_is_bit_identical_forward(shape, cin, hard, fresh):
    """The head layer's cells of a forward recording run as ONE launch at the flush (k_head_lif_fwd_win: membrane potential
    and spikes of a tile carried in registers from pass to pass) -- every pass's v', z' and channel-major planes must equal
    one evf_head_lif_fwd launch per pass bit for bit.  Full and ragged tiles, Cin 2 / 3 / 4, both reset rules, starting from
    the zero state and from a given state; 19 passes: more than one launch's worth (16), so a run is cut and continued."""
    B, H, W = shape
    npass = 19
    torch.manual_seed(11)
    w = _f(32, cin, 3, 3, scale=0.4)
    leak, thresh = _f(32, scale=0.3), _f(32, scale=0.1) + 0.5
    xs = [torch.poisson(torch.full((B, cin, H, W), 0.4, device=DEV)) for _ in range(npass)]
    v0 = None if fresh else _f(B, H, W, C, scale=0.5)
    z0 = None if fresh else _bits(B, H, W)

    def run(record):
        outs = []
        v, z = v0, z0
        if record:
            assert _lib.raw("evf_fwd_defer_begin") == 0
        try:
            for x in xs:
                vo = torch.full((B, H, W, C), 3.0, device=DEV)
                zo = torch.full((B, H, W), 5, dtype=torch.int32, device=DEV)
                zT = torch.full((B, H, C, (W + 31) // 32), 9, dtype=torch.int32, device=DEV)
                _lib.call("evf_head_lif_fwd", P(x), P(w), P(leak), P(thresh), P(v), P(z), B, cin, H, W, hard, P(vo), P(zo), P(zT))
                outs.append((vo, zo, zT))
                v, z = vo, zo
            if record:
                assert _lib.raw("evf_fwd_defer_pending") == npass
        finally:
            if record:
                _lib.call("evf_fwd_defer_flush")
        torch.cuda.synchronize()
        return outs

    ref, got = run(False), run(True)
    assert any(int((zo != 0).sum()) > 0 for _, zo, _ in ref)  # (the cells do spike)
    for t, (a, b) in enumerate(zip(ref, got)):
        for name, x, y in zip(("v", "z", "zT"), a, b):
            assert torch.equal(x, y), (t, name)


@pytest.mark.parametrize("shape,surrogate,hard", [((8, 128, 128), 0, 1), ((2, 37, 50), 0, 1), ((2, 37, 50), 2, 0), ((16, 128, 128), 0, 1)])
def test_head_layer_of_a_window_in_one_launch_is_bit_identical_backward(shape, surrogate, hard):
    """The head layer's backward cells of a recording run after the last index as ONE launch (k_head_bwd_win) when every
    pass has a dL/d(spikes) buffer of its own -- carried dL/dv and the block's partial sums re-read by the thread that wrote
    them (any shape / surrogate), or kept in registers with all loads of a pass in flight (<= 4 trips per block, the
    reference's default neuron).  Against the same cells with ONE shared dL/d(spikes) buffer, which the library then runs
    where they were recorded, one launch each: dL/dv of the window's start must agree bit for bit, the weight-gradient slabs
    and the per-block per-channel rows to fp32 round-off (the register form keeps its sums over all passes).  (16 x 128 x 128: 8 trips per block -- the through-memory form of the default
    neuron.)"""
    B, H, W = shape
    npass = 6
    torch.manual_seed(13)
    L = _lib.load()
    nsl = L.evf_head_lif_bwd_wgrad_slabs(B, H, W)
    leak, thresh = _f(32, scale=0.3), _f(32, scale=0.1) + 0.5
    xs = [torch.poisson(torch.full((B, 2, H, W), 0.4, device=DEV)) for _ in range(npass)]
    vs = [_f(B, H, W, C, scale=0.7) for _ in range(npass + 1)]  # vs[t]: v before pass t; vs[t + 1]: after
    zs = [_bits(B, H, W) for _ in range(npass)]
    gz = _f(B, H, W, C, scale=0.2)
    row_ld = 64

    def run(shared):
        gzs = [gz if shared else gz.clone() for _ in range(npass)]
        gv = torch.empty(B, H, W, C, device=DEV)
        slab = torch.full((nsl, 32 * 18), 7.0, device=DEV)
        rows = torch.zeros(nsl, row_ld, device=DEV)
        assert _lib.raw("evf_bwd_defer_begin") == 0
        try:
            for k in range(npass):  # backward pass k = forward pass npass - 1 - k
                t = npass - 1 - k
                assert _lib.raw("evf_bwd_defer_slot", 2 * k + 3) == 0
                _lib.call("evf_head_lif_bwd_wgrad", P(gzs[k]), P(gv) if k else None, P(vs[t + 1]), P(vs[t]) if t else None,
                          P(zs[t]) if t else None, P(xs[t]), P(leak), P(thresh), B, 2, H, W, hard, surrogate, 10.0, None, P(gv),
                          P(rows[:, :32]), P(rows[:, 32:]), P(slab), (1 if k else 0) | (row_ld << 8))
            assert _lib.raw("evf_bwd_defer_pending") == npass
        finally:
            _lib.call("evf_bwd_defer_flush")
        torch.cuda.synchronize()
        return gv, slab, rows

    ref, got = run(True), run(False)
    assert float(ref[1].abs().max()) > 0 and float(ref[2].abs().max()) > 0
    assert torch.equal(ref[0], got[0]), "g_v"
    for name, a, b in zip(("slab", "rows"), ref[1:], got[1:]):
        # (the register form sums over ALL passes before it reduces across the block's waves: fp32 round-off apart)
        assert _rel(b.sum(0), a.sum(0)) < 5e-6 and _rel(b, a) < 5e-5, (name, _rel(b, a))


@pytest.mark.parametrize("shape", [(8, 128, 128), (2, 37, 50), (4, 260, 346), (16, 128, 128)])
def test_plif_head_backward_of_a_window_with_the_trace_inside(shape):
    """PLIF head: evf_head_plif_bwd_wgrad (neuron backward, weight gradient and the presynaptic trace's backward in one pass)
    against evf_head_lif_bwd_wgrad + evf_plif_trace_bwd, one launch per pass and as the ONE launch a recording makes of the
    window's passes (k_head_bwd_win<.., PLIF>: dL/dv and dL/d(pt) carried in registers when a block makes <= 3 trips -- 8 x 128 x 128
    and, with the larger grid evf_head_lif_bwd_wgrad_slabs picks for it, 4 x 260 x 346 --, through memory otherwise: 16 x 128 x 128).
    evf_bwd_defer_hold_heads: a call that cannot be recorded in the middle launches what else is recorded, the head cells wait.
    dL/dv and dL/d(pt) entering the window bit for bit; slabs and per-channel sums (leak, thresh, leak_pt, add_pt) to round-off."""
    B, H, W = shape
    npass = 5
    torch.manual_seed(29)
    L = _lib.load()
    nsl = max(L.evf_head_lif_bwd_wgrad_slabs(B, H, W), 512)
    leak, thresh = _f(32, scale=0.3), _f(32, scale=0.1) + 0.5
    lpt, apt = _f(32, scale=0.5) - 1.0, _f(32, scale=0.5) - 2.0
    xs = [torch.poisson(torch.full((B, 2, H, W), 0.4, device=DEV)) for _ in range(npass)]
    vs = [_f(B, H, W, C, scale=0.7) for _ in range(npass + 1)]   # vs[t]: v before pass t
    pts = [_f(B, H, W, C, scale=0.3).abs() for _ in range(npass)]  # pts[t]: trace before pass t (pts[0] unused: zero state)
    Ps = [_f(B, H, W, scale=0.2).abs() for _ in range(npass)]
    zs = [_bits(B, H, W) for _ in range(npass)]
    gzs = [_f(B, H, W, C, scale=0.2) for _ in range(npass)]
    row_ld = 160

    def run(mode):  # "two": two calls per pass; "one": fused, a launch per pass; "win": fused, recorded (one launch)
        gv, gpt = torch.empty(B, H, W, C, device=DEV), torch.empty(B, H, W, C, device=DEV)
        gcur = torch.empty(B, H, W, C, device=DEV)
        slab = torch.full((nsl, 32 * 18), 7.0, device=DEV)
        rows = torch.zeros(nsl, row_ld, device=DEV)
        if mode == "win":
            assert _lib.raw("evf_bwd_defer_begin") == 0 and _lib.raw("evf_bwd_defer_hold_heads", 1) == 0
        try:
            for k in range(npass):
                t = npass - 1 - k
                flag = (1 if k else 0) | (row_ld << 8)
                head = (P(gzs[k]), P(gv) if k else None, P(vs[t + 1]), P(vs[t]) if t else None, P(zs[t]) if t else None, P(xs[t]),
                        P(leak), P(thresh), B, 2, H, W, 1, 0, 10.0)
                if mode == "two":
                    _lib.call("evf_head_lif_bwd_wgrad", *head, P(gcur), P(gv), P(rows[:, :32]), P(rows[:, 32:]), P(slab), flag)
                    _lib.call("evf_plif_trace_bwd", P(gcur), P(gpt) if k else None, P(pts[t]) if t else None, P(vs[0]), P(Ps[t]), P(lpt), P(apt),
                              B, H, W, P(gpt), P(torch.empty(B, H, W, device=DEV)), None, P(rows[:, 64:]), P(rows[:, 96:]), row_ld)
                else:
                    if mode == "win":
                        assert _lib.raw("evf_bwd_defer_slot", 2 * k) == 0
                    _lib.call("evf_head_plif_bwd_wgrad", *head, P(gv), P(rows[:, :32]), P(rows[:, 32:]), P(slab), flag, P(gpt) if k else None,
                              P(pts[t]) if t else None, P(Ps[t]), P(lpt), P(apt), P(gpt), P(rows[:, 64:]), P(rows[:, 96:]))
                    if mode == "win" and k == 2:
                        # a call that cannot be recorded (accumulating input gradient): launches what is recorded -- not the head cells
                        g = _f(B, H, W, C)
                        _lib.call("evf_conv_dgrad_b3_f32", P(g), P(_packs()[1]), P(torch.zeros(B, H, W, C, device=DEV)), 1, B, H, W, None, None)
                        assert _lib.raw("evf_bwd_defer_pending") == 3
            if mode == "win":
                assert _lib.raw("evf_bwd_defer_pending") == npass
        finally:
            if mode == "win":
                _lib.call("evf_bwd_defer_flush")
        torch.cuda.synchronize()
        return gv, gpt, slab, rows

    ref = run("two")
    assert float(ref[1].abs().max()) > 0 and float(ref[3][:, 64:128].sum(0).abs().min()) > 0
    for mode in ("one", "win"):
        got = run(mode)
        assert torch.equal(ref[0], got[0]), (mode, "g_v")
        assert torch.equal(ref[1], got[1]), (mode, "g_pt")
        for name, a, b in zip(("slab", "rows"), ref[2:], got[2:]):
            assert _rel(b.sum(0), a.sum(0)) < 2e-5, (mode, name, _rel(b.sum(0), a.sum(0)))


@pytest.mark.parametrize("shape", SHAPES + [(3, 20, 96), (4, 64, 160)])
def test_recorded_fused_backward_cells_match_the_one_cell_launches(shape):
    """The fused-backward cells of a backward index, recorded and launched together -- through k_bwd_diag (the one-cell kernel's
    body) and through k_bwd_diag_ws (four waves stream the tensors, do the neuron backward and stage the operands, four
    contract them) -- against one evf_lif_bwd_wgrad2 / _top launch per cell: dL/d(current) (fp32 and its three split planes)
    and dL/dv bit for bit; weight-gradient slabs (summed over their rows) and per-channel sums to fp32 round-off.  All three
    cell kinds in one index (feed-forward, recurrent with the second gradient part, under the prediction head), ragged shapes
    (partial units, a unit count that is odd per block), first touch and accumulation."""
    B, H, W = shape
    torch.manual_seed(17)
    L = _lib.load()
    nsl = L.evf_lif_bwd_wgrad_slabs(B, H, W)
    nW = (W + 31) // 32
    row_ld = 160

    def cell(kind):
        c = {"kind": kind, "gv": _f(B, H, W, C, scale=0.1), "vo": _f(B, H, W, C, scale=0.6), "vp": _f(B, H, W, C, scale=0.6),
             "zp": _bits(B, H, W), "xT": _planes(_bits(B, H, W)), "leak": _f(32, scale=0.3), "thresh": _f(32, scale=0.1) + 0.4}
        if kind == "top":
            c.update(flow=torch.tanh(_f(B, 2, H, W)), g_flow=_f(B, 2, H, W), pw=_f(2, 32, scale=0.05), zo=_bits(B, H, W, rate=0.4))
        else:
            c.update(gz=_f(B, H, W, C, scale=0.2), gz2=_f(B, H, W, C, scale=0.2) if kind == "rec" else None)
        if kind == "rec":
            c["zT"] = _planes(c["zp"])
        return c

    cells = [cell(k) for k in ("ff", "rec", "top", "rec", "ff")]

    def run(mode):  # None: one launch per cell; 0 / 1 / 2: recorded, k_bwd_diag / k_bwd_diag_ws<4> / k_bwd_diag_ws<8>
        outs = []
        for acc in (0, 1):  # first touch of the slabs, then accumulation (same inputs again)
            if mode is not None:
                assert L.evf_bwd_diag_select(mode) == 0 and _lib.raw("evf_bwd_defer_begin") == 0 and _lib.raw("evf_bwd_defer_slot", 4) == 0
            try:
                for n, c in enumerate(cells):
                    if acc == 0:
                        c["out"] = {"gcur": torch.full((B, H, W, C), 3.0, device=DEV), "gsp": torch.zeros(3, B, H, W, C, dtype=torch.bfloat16, device=DEV),
                                    "gvp": torch.full((B, H, W, C), 3.0, device=DEV), "rows": torch.zeros(nsl, row_ld, device=DEV),
                                    "sff": torch.full((nsl, 9216), 5.0, device=DEV), "srec": torch.full((nsl, 9216), 5.0, device=DEV)}
                    o = c["out"]
                    flag = acc | (row_ld << 8)
                    if c["kind"] == "top":
                        _lib.call("evf_lif_bwd_wgrad_top", P(c["flow"]), P(c["g_flow"]), P(c["pw"]), P(c["zo"]), P(o["rows"][:, 64:]),
                                  P(o["rows"][:, 128:]), P(c["gv"]), P(c["vo"]), P(c["vp"]), P(c["zp"]), P(c["xT"]), P(c["leak"]), P(c["thresh"]),
                                  B, H, W, 1, 0, 10.0, P(o["gcur"]), P(o["gsp"]), P(o["gvp"]), P(o["rows"][:, :32]), P(o["rows"][:, 32:]),
                                  P(o["sff"]), flag)
                    else:
                        rec = c["kind"] == "rec"
                        _lib.call("evf_lif_bwd_wgrad2", P(c["gz"]), P(c["gz2"]), P(c["gv"]), P(c["vo"]), P(c["vp"]), P(c["zp"]), P(c["xT"]),
                                  P(c["zT"]) if rec else None, P(c["leak"]), P(c["thresh"]), B, H, W, 1, 0, 10.0, P(o["gcur"]), P(o["gsp"]),
                                  P(o["gvp"]), P(o["rows"][:, :32]), P(o["rows"][:, 32:]), P(o["sff"]), P(o["srec"]) if rec else None, flag)
                if mode is not None:
                    assert _lib.raw("evf_bwd_defer_pending") == len(cells)
            finally:
                if mode is not None:
                    _lib.call("evf_bwd_defer_flush")
                    L.evf_bwd_diag_select(-1)
        torch.cuda.synchronize()
        for c in cells:
            o = c.pop("out")
            outs.append((o["gcur"], o["gsp"], o["gvp"], o["rows"].sum(0), o["sff"].sum(0), o["srec"].sum(0) if c["kind"] == "rec" else None))
        return outs

    ref = run(None)
    for mode in (0, 1, 2):
        got = run(mode)
        for n, (a, b) in enumerate(zip(ref, got)):
            for name, x, y in zip(("g_cur", "split", "g_v_prev"), a[:3], b[:3]):
                assert torch.equal(x, y), (mode, n, name)
            for name, x, y in zip(("rows", "slab_ff", "slab_rec"), a[3:], b[3:]):
                if x is not None:
                    assert float(x.abs().max()) > 0 and _rel(y, x) < 2e-5, (mode, n, name, _rel(y, x))
    assert L.evf_bwd_diag_select(3) != 0


@pytest.mark.parametrize("shape", SHAPES + [(3, 20, 96), (2, 33, 70)])
def test_plif_trace_backward_inside_the_fused_backward(shape):
    """PLIF hidden cells: evf_plif_bwd_wgrad2 / _top (the presynaptic trace's backward in team E of the fused backward, g_cur
    never read back) against evf_lif_bwd_wgrad2 / _top followed by evf_plif_trace_bwd: dL/d(current), its split planes, dL/dv,
    the trace carry and the raw map dL/d(pooled activity) bit for bit; slabs and per-channel sums (leak, thresh, leak_pt,
    add_pt) to fp32 round-off.  One launch per cell and recorded (three kinds in one index), with and without carry / previous
    trace, carry written in place, first touch and accumulation, ragged shapes."""
    B, H, W = shape
    torch.manual_seed(23)
    L = _lib.load()
    nsl = max(L.evf_lif_bwd_wgrad_slabs(B, H, W), 512)
    row_ld = 224

    def cell(kind, carry, prev):
        c = {"kind": kind, "gv": _f(B, H, W, C, scale=0.1), "vo": _f(B, H, W, C, scale=0.6), "vp": _f(B, H, W, C, scale=0.6),
             "zp": _bits(B, H, W), "xT": _planes(_bits(B, H, W)), "leak": _f(32, scale=0.3), "thresh": _f(32, scale=0.1) + 0.4,
             "gk": _f(B, H, W, C, scale=0.2) if carry else None, "pp": _f(B, H, W, C, scale=0.3).abs() if prev else None,
             "P": _f(B, H, W, scale=0.2).abs(), "lpt": _f(32, scale=0.5) - 1.0, "apt": _f(32, scale=0.5) - 2.0}
        if kind == "top":
            c.update(flow=torch.tanh(_f(B, 2, H, W)), g_flow=_f(B, 2, H, W), pw=_f(2, 32, scale=0.05), zo=_bits(B, H, W, rate=0.4))
        else:
            c.update(gz=_f(B, H, W, C, scale=0.2), gz2=_f(B, H, W, C, scale=0.2) if kind == "rec" else None)
        if kind == "rec":
            c["zT"] = _planes(c["zp"])
        return c

    cells = [cell("ff", True, True), cell("rec", True, True), cell("top", False, True), cell("rec", True, False), cell("ff", False, False)]

    def run(fused, recorded):
        outs = []
        for acc in (0, 1):
            if recorded:
                assert _lib.raw("evf_bwd_defer_begin") == 0 and _lib.raw("evf_bwd_defer_slot", 2) == 0
            try:
                for c in cells:
                    if acc == 0:
                        c["out"] = {"gcur": torch.full((B, H, W, C), 3.0, device=DEV), "gsp": torch.zeros(3, B, H, W, C, dtype=torch.bfloat16, device=DEV),
                                    "gvp": torch.full((B, H, W, C), 3.0, device=DEV), "rows": torch.zeros(nsl, row_ld, device=DEV),
                                    "sff": torch.full((nsl, 9216), 5.0, device=DEV), "srec": torch.full((nsl, 9216), 5.0, device=DEV),
                                    # the carry is read and written in place (the engine's buffers)
                                    "gpt": c["gk"].clone() if c["gk"] is not None else torch.full((B, H, W, C), 3.0, device=DEV),
                                    "gP": torch.full((B, H, W), 3.0, device=DEV)}
                    o = c["out"]
                    if acc == 1 and c["gk"] is not None:
                        o["gpt"].copy_(c["gk"])
                    flag = acc | (row_ld << 8)
                    carry = P(o["gpt"]) if c["gk"] is not None else None
                    tr = (carry, P(c["pp"]), P(c["P"]), P(c["lpt"]), P(c["apt"]), P(o["gpt"]), P(o["gP"]), P(o["rows"][:, 160:]), P(o["rows"][:, 192:]))
                    if c["kind"] == "top":
                        _lib.call("evf_plif_bwd_wgrad_top" if fused else "evf_lif_bwd_wgrad_top", P(c["flow"]), P(c["g_flow"]), P(c["pw"]), P(c["zo"]),
                                  P(o["rows"][:, 64:]), P(o["rows"][:, 128:]), P(c["gv"]), P(c["vo"]), P(c["vp"]), P(c["zp"]), P(c["xT"]), P(c["leak"]),
                                  P(c["thresh"]), B, H, W, 1, 0, 10.0, P(o["gcur"]), P(o["gsp"]), P(o["gvp"]), P(o["rows"][:, :32]),
                                  P(o["rows"][:, 32:]), P(o["sff"]), flag, *(tr if fused else ()))
                    else:
                        rec = c["kind"] == "rec"
                        _lib.call("evf_plif_bwd_wgrad2" if fused else "evf_lif_bwd_wgrad2", P(c["gz"]), P(c["gz2"]), P(c["gv"]), P(c["vo"]), P(c["vp"]),
                                  P(c["zp"]), P(c["xT"]), P(c["zT"]) if rec else None, P(c["leak"]), P(c["thresh"]), B, H, W, 1, 0, 10.0, P(o["gcur"]),
                                  P(o["gsp"]), P(o["gvp"]), P(o["rows"][:, :32]), P(o["rows"][:, 32:]), P(o["sff"]), P(o["srec"]) if rec else None,
                                  flag, *(tr if fused else ()))
                    if not fused:
                        _lib.call("evf_plif_trace_bwd", P(o["gcur"]), carry, P(c["pp"]), P(c["vo"]), P(c["P"]), P(c["lpt"]), P(c["apt"]), B, H, W,
                                  P(o["gpt"]), P(o["gP"]), None, P(o["rows"][:, 160:]), P(o["rows"][:, 192:]), row_ld)
                if recorded:
                    assert _lib.raw("evf_bwd_defer_pending") == len(cells)
            finally:
                if recorded:
                    _lib.call("evf_bwd_defer_flush")
        torch.cuda.synchronize()
        for c in cells:
            o = c.pop("out")
            outs.append((o["gcur"], o["gsp"], o["gvp"], o["gpt"], o["gP"], o["rows"].sum(0), o["sff"].sum(0),
                         o["srec"].sum(0) if c["kind"] == "rec" else None))
        return outs

    ref = run(False, False)
    assert all(float(r[3].abs().max()) > 0 and float(r[4].abs().max()) > 0 for r in ref)
    assert all(float(r[5][160:].abs().min()) > 0 for r in ref)  # (every trace parameter has a gradient)
    for recorded in (False, True):
        got = run(True, recorded)
        for n, (a, b) in enumerate(zip(ref, got)):
            for name, x, y in zip(("g_cur", "split", "g_v_prev", "g_pt_prev", "g_P"), a[:5], b[:5]):
                assert torch.equal(x, y), (recorded, n, name)
            for name, x, y in zip(("rows", "slab_ff", "slab_rec"), a[5:], b[5:]):
                if x is not None:
                    assert float(x.abs().max()) > 0 and _rel(y, x) < 2e-5, (recorded, n, name, _rel(y, x))
    # not the default neuron: the fused form refuses (the two-call path serves it)
    c = cells[0]
    o = torch.zeros(B, H, W, C, device=DEV)
    rc = L.evf_plif_bwd_wgrad2(P(c["gz"]), None, P(c["gv"]), P(c["vo"]), P(c["vp"]), P(c["zp"]), P(c["xT"]), None, P(c["leak"]), P(c["thresh"]), B, H,
                               W, 0, 0, 10.0, P(o), None, P(o.clone()), P(torch.zeros(nsl, row_ld, device=DEV)), P(torch.zeros(nsl, row_ld, device=DEV)),
                               P(torch.zeros(nsl, 9216, device=DEV)), None, 0, None, None, P(c["P"]), P(c["lpt"]), P(c["apt"]), P(o.clone()),
                               P(torch.zeros(B, H, W, device=DEV)), P(torch.zeros(64, device=DEV)), P(torch.zeros(64, device=DEV)), _lib.stream_ptr())
    assert rc == -95  # EVF_ENOTSUP


@pytest.mark.parametrize("shape", SHAPES + [(3, 21, 96), (1, 2, 32)])
@pytest.mark.parametrize("hard", [1, 0])
def test_recorded_forward_cells_are_bit_identical(shape, hard):
    """The forward cells of an index, recorded (evf_fwd_defer_*) and launched together, against one direct launch per cell --
    bit-identical potentials, spike words, channel-major bit planes and flow through all three dispatchers: k_fwd_diag (a tile
    per block), k_fwd_diag_p (persistent, a strip per wave) and k_fwd_diag_t (persistent, matrix team + element-wise team:
    full-line state I/O, spike words by DPP, bit planes by a butterfly transpose).  Feed-forward and recurrent cells mixed,
    one cell with the prediction head in its epilogue, cells without previous state (zero page), two indices (block ranges
    cross cell boundaries), ragged shapes (odd H: half strips; W not a multiple of 32: partial rows)."""
    B, H, W = shape
    torch.manual_seed(9)
    L = _lib.load()
    nW = (W + 31) // 32
    cells = []
    for d, rec, state, pred in ((0, False, True, False), (0, True, True, False), (0, True, False, False), (1, False, False, False),
                                (1, False, True, True), (1, True, True, False)):
        x = _bits(B, H, W, 0.3)
        wff, wrec = _packs()[0], (_packs()[0] if rec else None)
        leak, thresh = _f(32, scale=0.3) - 1, _f(32, scale=0.1) + 0.4
        v_prev = _f(B, H, W, C, scale=0.5) if state else None
        z_prev = _bits(B, H, W, 0.2) if state else None
        pw, pb = (_f(2, 32, scale=0.2), _f(2, scale=0.1)) if pred else (None, None)
        cells.append((d, x, wff, wrec, leak, thresh, v_prev, z_prev, pw, pb))

    def outs():
        return (torch.full((B, H, W, C), 7.0, device=DEV), torch.full((B, H, W), 5, dtype=torch.int32, device=DEV),
                torch.full((B, H, 32, nW), 5, dtype=torch.int32, device=DEV), torch.full((B, 2, H, W), 7.0, device=DEV))

    def launch(cell, o):
        d, x, wff, wrec, leak, thresh, v_prev, z_prev, pw, pb = cell
        v, z, zT, flow = o
        if pw is None:
            _lib.call("evf_conv_lif_fwd_b3", P(x), P(wff), P(wrec), P(leak), P(thresh), P(v_prev), P(z_prev), B, H, W, hard, P(v), P(z), P(zT))
        else:
            _lib.call("evf_conv_lif_fwd_b3_pred", P(x), P(wff), P(wrec), P(leak), P(thresh), P(v_prev), P(z_prev), B, H, W, hard, P(v), P(z),
                      P(zT), P(pw), P(pb), P(flow))

    ref = []
    for cell in cells:
        o = outs()
        launch(cell, o)
        ref.append(o)
    torch.cuda.synchronize()
    assert any(int((r[1] != 0).sum()) > 0 for r in ref)  # (the cells do spike)
    try:
        for which in (0, 1, 2):
            assert L.evf_fwd_diag_select(which) == 0
            got = [outs() for _ in cells]
            assert _lib.raw("evf_fwd_defer_begin") == 0
            try:
                for cell, o in zip(cells, got):
                    assert _lib.raw("evf_fwd_defer_slot", cell[0]) == 0
                    launch(cell, o)
                assert _lib.raw("evf_fwd_defer_pending") == len(cells)
            finally:
                _lib.call("evf_fwd_defer_flush")
            assert _lib.raw("evf_fwd_defer_pending") == 0
            torch.cuda.synchronize()
            for k, (o, r, cell) in enumerate(zip(got, ref, cells)):
                assert torch.equal(o[0], r[0]), (which, k, "v")
                assert torch.equal(o[1], r[1]), (which, k, "z")
                assert torch.equal(o[2], r[2]), (which, k, "zT")
                if cell[8] is not None:
                    assert torch.equal(o[3], r[3]), (which, k, "flow")
    finally:
        L.evf_fwd_diag_select(-1)
    assert L.evf_fwd_diag_select(7) != 0


@pytest.mark.parametrize("shape", [(8, 128, 128), (2, 33, 70), (4, 260, 346)])
def test_plif_hidden_cell_backward_of_a_window_in_one_launch(shape):
    """evf_plif_bwd_wgrad_window (all passes of a window of a feed-forward PLIF hidden cell in one launch: dL/dv, dL/d(pt) and the
    potential carried in registers) against one evf_plif_bwd_wgrad2 launch per pass with the carries through memory: dL/d(current)
    and dL/d(pooled activity) of every pass and the gradients on the window's entry state bit for bit; the weight-gradient slab and
    the per-channel sums to round-off.  First pass of the window without previous state, one pass without dL/d(spikes)."""
    import ctypes

    B, H, W = shape
    npass = 6
    torch.manual_seed(31)
    L = _lib.load()
    nsl = max(L.evf_lif_bwd_wgrad_slabs(B, H, W), 512)
    row_ld = 224
    leak, thresh = _f(32, scale=0.3), _f(32, scale=0.1) + 0.4
    lpt, apt = _f(32, scale=0.5) - 1.0, _f(32, scale=0.5) - 2.0
    vs = [None] + [_f(B, H, W, C, scale=0.6) for _ in range(npass)]        # vs[t]: potential before pass t; vs[t + 1]: after
    pts = [None] + [_f(B, H, W, C, scale=0.3).abs() for _ in range(npass - 1)]  # pts[t]: trace before pass t
    zs = [None] + [_bits(B, H, W) for _ in range(npass - 1)]
    xT = [_planes(_bits(B, H, W)) for _ in range(npass)]
    Ps = [_f(B, H, W, scale=0.2).abs() for _ in range(npass)]
    gzs = [_f(B, H, W, C, scale=0.2) for _ in range(npass)]
    gzs[2] = None

    def outs():
        return {"gcur": [torch.full((B, H, W, C), 3.0, device=DEV) for _ in range(npass)],
                "gP": [torch.full((B, H, W), 3.0, device=DEV) for _ in range(npass)],
                "gv": torch.full((B, H, W, C), 3.0, device=DEV), "gpt": torch.full((B, H, W, C), 3.0, device=DEV),
                "rows": torch.zeros(nsl, row_ld, device=DEV), "slab": torch.full((nsl, 9216), 5.0, device=DEV)}

    def per_pass():
        o = outs()
        for k in range(npass):  # backward pass k = forward pass t
            t = npass - 1 - k
            _lib.call("evf_plif_bwd_wgrad2", P(gzs[t]), None, P(o["gv"]) if k else None, P(vs[t + 1]), P(vs[t]), P(zs[t]), P(xT[t]), None,
                      P(leak), P(thresh), B, H, W, 1, 0, 10.0, P(o["gcur"][t]), None, P(o["gv"]), P(o["rows"][:, :32]), P(o["rows"][:, 32:]),
                      P(o["slab"]), None, (1 if k else 0) | (row_ld << 8), P(o["gpt"]) if k else None, P(pts[t]), P(Ps[t]), P(lpt), P(apt),
                      P(o["gpt"]), P(o["gP"][t]), P(o["rows"][:, 64:]), P(o["rows"][:, 96:]))
        return o

    def window():
        o = outs()
        order = list(range(npass - 1, -1, -1))  # index 0 = the last pass
        arr = lambda ts: (ctypes.c_void_p * npass)(*[P(x) for x in ts])  # noqa: E731
        _lib.call("evf_plif_bwd_wgrad_window", npass, arr([gzs[t] for t in order]), arr([vs[t + 1] for t in order]),
                  arr([vs[t] for t in order]), arr([zs[t] for t in order]), arr([xT[t] for t in order]), arr([o["gcur"][t] for t in order]), None,
                  arr([pts[t] for t in order]), arr([Ps[t] for t in order]), arr([o["gP"][t] for t in order]), P(leak), P(thresh), P(lpt),
                  P(apt), B, H, W, 10.0, P(o["gv"]), P(o["gpt"]), P(o["rows"][:, :32]), P(o["rows"][:, 32:]), P(o["rows"][:, 64:]),
                  P(o["rows"][:, 96:]), P(o["slab"]), 0 | (row_ld << 8))
        return o

    # the layer under the prediction head: the head's backward inside (evf_plif_bwd_wgrad_top per pass / _window_top)
    flows = [torch.tanh(_f(B, 2, H, W)) for _ in range(npass)]
    gfl = [_f(B, 2, H, W) for _ in range(npass)]
    zo = [_bits(B, H, W, rate=0.4) for _ in range(npass)]
    pw = _f(2, 32, scale=0.05)

    def top_per_pass():
        o = outs()
        for k in range(npass):
            t = npass - 1 - k
            _lib.call("evf_plif_bwd_wgrad_top", P(flows[t]), P(gfl[t]), P(pw), P(zo[t]), P(o["rows"][:, 128:]), P(o["rows"][:, 192:]),
                      P(o["gv"]) if k else None, P(vs[t + 1]), P(vs[t]), P(zs[t]), P(xT[t]), P(leak), P(thresh), B, H, W, 1, 0, 10.0,
                      P(o["gcur"][t]), None, P(o["gv"]), P(o["rows"][:, :32]), P(o["rows"][:, 32:]), P(o["slab"]), (1 if k else 0) | (row_ld << 8),
                      P(o["gpt"]) if k else None, P(pts[t]), P(Ps[t]), P(lpt), P(apt), P(o["gpt"]), P(o["gP"][t]), P(o["rows"][:, 64:]),
                      P(o["rows"][:, 96:]))
        return o

    def top_window():
        o = outs()
        order = list(range(npass - 1, -1, -1))
        arr = lambda ts: (ctypes.c_void_p * npass)(*[P(x) for x in ts])  # noqa: E731
        _lib.call("evf_plif_bwd_wgrad_window_top", npass, arr([flows[t] for t in order]), arr([gfl[t] for t in order]), P(pw),
                  arr([zo[t] for t in order]), P(o["rows"][:, 128:]), P(o["rows"][:, 192:]), arr([vs[t + 1] for t in order]),
                  arr([vs[t] for t in order]), arr([zs[t] for t in order]), arr([xT[t] for t in order]), arr([o["gcur"][t] for t in order]), None,
                  arr([pts[t] for t in order]), arr([Ps[t] for t in order]), arr([o["gP"][t] for t in order]), P(leak), P(thresh), P(lpt),
                  P(apt), B, H, W, 10.0, P(o["gv"]), P(o["gpt"]), P(o["rows"][:, :32]), P(o["rows"][:, 32:]), P(o["rows"][:, 64:]),
                  P(o["rows"][:, 96:]), P(o["slab"]), 0 | (row_ld << 8))
        return o

    for ref, got in ((per_pass(), window()), (top_per_pass(), top_window())):
        _check_window(ref, got, npass)


@pytest.mark.parametrize("shape", [(8, 128, 128), (2, 33, 70)])
def test_lif_hidden_cell_backward_of_a_window_in_one_launch(shape):
    """evf_lif_bwd_wgrad_window (LIF feed-forward hidden cell / the layer under the prediction head, all passes of a window in one
    launch) against evf_lif_bwd_wgrad2 / _top per pass: dL/d(current) as fp32 and as its three bf16 planes of every pass and the
    gradient on the window's entry state bit for bit; slab and per-channel sums to round-off."""
    import ctypes

    B, H, W = shape
    npass = 6
    torch.manual_seed(37)
    L = _lib.load()
    nsl = max(L.evf_lif_bwd_wgrad_slabs(B, H, W), 512)
    row_ld = 224
    leak, thresh = _f(32, scale=0.3), _f(32, scale=0.1) + 0.4
    vs = [None] + [_f(B, H, W, C, scale=0.6) for _ in range(npass)]
    zs = [None] + [_bits(B, H, W) for _ in range(npass - 1)]
    xT = [_planes(_bits(B, H, W)) for _ in range(npass)]
    gzs = [_f(B, H, W, C, scale=0.2) for _ in range(npass)]
    flows = [torch.tanh(_f(B, 2, H, W)) for _ in range(npass)]
    gfl = [_f(B, 2, H, W) for _ in range(npass)]
    zo = [_bits(B, H, W, rate=0.4) for _ in range(npass)]
    pw = _f(2, 32, scale=0.05)
    arr = lambda ts: (ctypes.c_void_p * npass)(*[P(x) for x in ts])  # noqa: E731
    order = list(range(npass - 1, -1, -1))

    def outs():
        return {"gcur": [torch.full((B, H, W, C), 3.0, device=DEV) for _ in range(npass)],
                "gsp": [torch.zeros(3, B, H, W, C, dtype=torch.bfloat16, device=DEV) for _ in range(npass)],
                "gv": torch.full((B, H, W, C), 3.0, device=DEV), "rows": torch.zeros(nsl, row_ld, device=DEV),
                "slab": torch.full((nsl, 9216), 5.0, device=DEV)}

    for top in (False, True):
        ref, got = outs(), outs()
        for k in range(npass):
            t = npass - 1 - k
            flag = (1 if k else 0) | (row_ld << 8)
            if top:
                _lib.call("evf_lif_bwd_wgrad_top", P(flows[t]), P(gfl[t]), P(pw), P(zo[t]), P(ref["rows"][:, 128:]), P(ref["rows"][:, 192:]),
                          P(ref["gv"]) if k else None, P(vs[t + 1]), P(vs[t]), P(zs[t]), P(xT[t]), P(leak), P(thresh), B, H, W, 1, 0, 10.0,
                          P(ref["gcur"][t]), P(ref["gsp"][t]), P(ref["gv"]), P(ref["rows"][:, :32]), P(ref["rows"][:, 32:]), P(ref["slab"]), flag)
            else:
                _lib.call("evf_lif_bwd_wgrad2", P(gzs[t]), None, P(ref["gv"]) if k else None, P(vs[t + 1]), P(vs[t]), P(zs[t]), P(xT[t]), None,
                          P(leak), P(thresh), B, H, W, 1, 0, 10.0, P(ref["gcur"][t]), P(ref["gsp"][t]), P(ref["gv"]), P(ref["rows"][:, :32]),
                          P(ref["rows"][:, 32:]), P(ref["slab"]), None, flag)
        _lib.call("evf_lif_bwd_wgrad_window", npass, None if top else arr([gzs[t] for t in order]),
                  arr([flows[t] for t in order]) if top else None, arr([gfl[t] for t in order]) if top else None, P(pw) if top else None,
                  arr([zo[t] for t in order]) if top else None, P(got["rows"][:, 128:]) if top else None,
                  P(got["rows"][:, 192:]) if top else None, arr([vs[t + 1] for t in order]), arr([vs[t] for t in order]),
                  arr([zs[t] for t in order]), arr([xT[t] for t in order]), arr([got["gcur"][t] for t in order]),
                  arr([got["gsp"][t] for t in order]), P(leak), P(thresh), B, H, W, 10.0, P(got["gv"]), P(got["rows"][:, :32]),
                  P(got["rows"][:, 32:]), P(got["slab"]), 0 | (row_ld << 8))
        torch.cuda.synchronize()
        for t in range(npass):
            assert torch.equal(ref["gcur"][t], got["gcur"][t]), (top, "g_cur", t)
            assert torch.equal(ref["gsp"][t], got["gsp"][t]), (top, "g_split", t)
        assert torch.equal(ref["gv"], got["gv"]) and float(ref["gv"].abs().max()) > 0
        for name in ("slab", "rows"):
            assert _rel(got[name].sum(0), ref[name].sum(0)) < 2e-5, (top, name, _rel(got[name].sum(0), ref[name].sum(0)))


@pytest.mark.parametrize("shape", [(8, 128, 128), (2, 33, 70), (4, 260, 346)])
def test_input_gradients_of_a_product_list_in_one_launch(shape):
    """evf_conv_dgrad_b3_multi (up to 16 products through k_dgrad_diag_dma in one launch, some with the PLIF trace term from the raw
    dL/dP map) against one evf_conv_dgrad_b3 call per product (k_conv_dgrad_ws, `accumulate | 2` for the trace term): bit for bit."""
    import ctypes

    B, H, W = shape
    torch.manual_seed(41)
    L = _lib.load()
    n = 7
    packs = [_packs()[1] for _ in range(3)]
    gsp = [(_f(3, B, H, W, C, scale=0.1)).to(torch.bfloat16) for _ in range(n)]
    gP = [_f(B, H, W, scale=0.3) if k % 3 != 2 else None for k in range(n)]
    xb = [_bits(B, H, W, rate=0.3) if gP[k] is not None else None for k in range(n)]
    ref = [torch.full((B, H, W, C), 7.0, device=DEV) for _ in range(n)]
    got = [torch.full((B, H, W, C), 9.0, device=DEV) for _ in range(n)]
    for k in range(n):
        _lib.call("evf_conv_dgrad_b3", P(gsp[k]), P(packs[k % 3]), P(ref[k]), 2 if gP[k] is not None else 0, B, H, W, P(gP[k]), P(xb[k]))
    arr = lambda ts: (ctypes.c_void_p * n)(*[P(x) for x in ts])  # noqa: E731
    _lib.call("evf_conv_dgrad_b3_multi", n, arr(gsp), arr([packs[k % 3] for k in range(n)]), arr(got), arr(gP), arr(xb), B, H, W)
    torch.cuda.synchronize()
    for k in range(n):
        assert float(ref[k].abs().max()) > 0 and torch.equal(ref[k], got[k]), k
    assert L.evf_conv_dgrad_b3_multi(17, arr(gsp), arr(gsp), arr(got), None, None, B, H, W, _lib.stream_ptr()) == -22


def _check_window(ref, got, npass):
    torch.cuda.synchronize()
    for t in range(npass):
        assert torch.equal(ref["gcur"][t], got["gcur"][t]), ("g_cur", t)
        assert torch.equal(ref["gP"][t], got["gP"][t]), ("g_P", t)
    assert torch.equal(ref["gv"], got["gv"]) and torch.equal(ref["gpt"], got["gpt"])
    assert float(ref["gpt"].abs().max()) > 0 and float(ref["rows"][:, 64:128].sum(0).abs().min()) > 0
    for name in ("slab", "rows"):
        assert _rel(got[name].sum(0), ref[name].sum(0)) < 2e-5, (name, _rel(got[name].sum(0), ref[name].sum(0)))


def test_evf_memset_is_a_kernel_fill_of_any_size():
    """evf_memset (what the library and the Python host use instead of hipMemsetAsync / zero_(): no memset nodes in a captured
    step): every byte of [dst, dst + bytes) set, nothing outside touched -- sizes around the 4- and 16-byte steps of the fill
    kernel, destinations that are 4- but not 16-byte aligned, the byte patterns the library uses (0x00, 0xFF)."""
    L = _lib.load()
    for value in (0x00, 0xFF, 0x5A):
        for nbytes in (0, 1, 3, 4, 5, 15, 16, 17, 4095, 4096, 4099, (1 << 20) + 7, (9 << 20) + 2):
            for off in (0, 4, 12):
                buf = torch.full((nbytes + 64,), 0x33, dtype=torch.uint8, device=DEV)
                rc = L.evf_memset(buf.data_ptr() + 16 + off, value, nbytes, _lib.stream_ptr())
                assert rc == 0
                got = buf.cpu().numpy()
                lo, hi = 16 + off, 16 + off + nbytes
                assert (got[:lo] == 0x33).all() and (got[hi:] == 0x33).all(), (value, nbytes, off)
                assert (got[lo:hi] == value).all(), (value, nbytes, off)
    assert L.evf_memset(None, 0, 16, _lib.stream_ptr()) == -22
    z = _lib.zeros((3, 5, 7), device=DEV)
    assert z.dtype == torch.float32 and float(z.abs().sum()) == 0.0
    t = torch.ones(1000, device=DEV)
    assert _lib.zero_(t[10:20]) is not None and float(t.sum()) == 990.0
