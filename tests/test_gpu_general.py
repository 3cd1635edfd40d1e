"""GPU parity of the general path (any channel count / stride): matrix-core
conv forward / input gradient / weight gradient, the four spiking cells as
stand-alone modules against the reference-generated single-step goldens (G6),
the ANN FireNet (G8), the spiking EV-FlowNet (G9) and the ALIF/XLIF FireNets
against the CPU oracle.

Tolerances: the conv kernels multiply and accumulate in fp32 (v_mfma_f32_32x32x2),
so they differ from the CPU fp32 convolution only by summation order: 1e-5
relative to the largest output.  Spikes are exact wherever the golden membrane
potential is further than 1e-5 from the threshold."""

import numpy as np
import pytest
import torch

from conftest import golden_cases, load_golden

pytestmark = pytest.mark.gpu

from event_flow_amd import _lib  # noqa: E402
from event_flow_amd.models import hip_ops  # noqa: E402
from event_flow_amd.models import spiking_submodules as cells  # noqa: E402
from event_flow_amd.models.model import ALIFFireNet, FireNet, SpikingRecEVFlowNet, XLIFFireNet  # noqa: E402
from oracle import snn as osnn  # noqa: E402

DEV = "cuda:0"


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def N(t):
    return t.detach().cpu().numpy()


def close(got, ref, rel, what=""):
    scale = max(float(np.abs(ref).max()), 1e-12)
    err = float(np.abs(got - ref).max())
    assert err <= rel * scale, (what, err, scale)


# ------------------------------------------------------------------ conv kernels
CONV_CASES = [
    # B, Cin, Cout, H, W, k, stride
    (2, 2, 8, 12, 10, 3, 2),
    (1, 4, 8, 9, 7, 3, 1),
    (2, 66, 16, 10, 12, 3, 1),
    (1, 130, 32, 20, 24, 3, 1),
    (2, 64, 128, 16, 16, 3, 2),
    (1, 32, 2, 17, 33, 1, 1),
    (2, 32, 2, 64, 96, 1, 1),  # streaming few-output 1x1 wgrad, several pixels per block
    (1, 64, 3, 9, 31, 1, 1),
    (3, 128, 4, 7, 5, 1, 1),
    (1, 8, 1, 40, 40, 1, 1),
    (1, 32, 5, 17, 33, 1, 1),  # matrix-core 1x1 wgrad with scalar g loads
    (1, 30, 2, 17, 33, 1, 1),
    (1, 256, 96, 5, 6, 3, 1),
    (3, 5, 7, 11, 13, 3, 2),
    (1, 8, 8, 13, 9, 3, 2),
    # 5x5 / 7x7 (models/unet.py:51: kernel_size defaults to 5)
    (2, 6, 8, 12, 10, 5, 1),
    (1, 32, 64, 20, 24, 5, 2),
    (2, 66, 40, 9, 11, 5, 1),
    (1, 4, 8, 15, 13, 7, 1),
    (1, 16, 16, 14, 14, 7, 2),
    # ragged spatial tiles (16 x 32) and channel groups of the tiled 3x3 kernel
    (2, 132, 64, 40, 70, 3, 1),
    (1, 32, 132, 33, 65, 3, 1),
    (1, 20, 96, 18, 34, 3, 1),
    (2, 512, 64, 8, 8, 3, 1),  # low resolution, many channels: the layers that split their contraction
    # images of at most 16 x 16 with many channels: the whole-image kernel (evf_conv_b3img.hip; forced in the b3tile modes)
    (2, 128, 192, 16, 16, 3, 1),
    (1, 96, 64, 13, 16, 3, 1),
    (3, 64, 40, 4, 7, 3, 1),
    (8, 512, 512, 16, 16, 3, 1),  # the spiking EV-FlowNet's 512-channel layers at their benched size (its own plan picks the kernel)
    # many output channels per input tile: the decoders' input gradients (evf_conv_b3n.hip; forced in the b3nstream modes):
    # 5 N tiles with a 4-channel remainder, 9 tiles in two chunks, ragged spatial tiles (8 x 32)
    (2, 132, 32, 24, 40, 3, 1),
    (1, 260, 64, 19, 33, 3, 1),
    (1, 32, 200, 9, 70, 3, 1),
]


@pytest.fixture
def conv_mode(request, monkeypatch):
    """b3: forward / input gradient on the bf16 matrix cores with exact splits (the default), general kernel only;
    b3tile: its spatially tiled 3x3 stride-1 kernel wherever the operands allow; f32: the fp32-MFMA kernels."""
    mode = request.param
    monkeypatch.setattr(hip_ops, "CONV_B3", mode != "f32")
    # b3nstream: the tile family forced AND its N-streaming member (evf_conv_b3n.hip) wherever the operands allow
    monkeypatch.setenv("EVF_CONV_NSTREAM", "2" if mode.startswith("b3nstream") else "1")
    assert _lib.load().evf_conv_tile_select(2 if mode.startswith(("b3tile", "b3nstream")) else 0) == 0
    assert _lib.load().evf_conv_split_select(3 if mode.endswith("split") else 0) == 0  # (deterministic slab reduction)
    yield mode
    _lib.load().evf_conv_tile_select(-1)
    _lib.load().evf_conv_split_select(0)


@pytest.mark.parametrize("conv_mode", ["b3", "b3tile", "b3split", "b3tilesplit", "b3nstream", "b3nstreamsplit", "f32"], indirect=True)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_forward_dgrad_wgrad_vs_cpu(case, conv_mode):
    B, Cin, Cout, H, W, k, s = case
    gen = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, k, k, generator=gen) * 0.2
    b = torch.randn(Cout, generator=gen)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y_ref = torch.nn.functional.conv2d(xr, wr, br, stride=s, padding=k // 2)
    gy = torch.randn(y_ref.shape, generator=gen)
    y_ref.backward(gy)

    class Owner:
        pass

    owner = Owner()
    xd, wd, bd = G(x.numpy()).requires_grad_(True), G(w.numpy()).requires_grad_(True), G(b.numpy()).requires_grad_(True)
    y = hip_ops.conv_act(owner, xd, wd, bd, stride=s, activation=None)
    assert tuple(y.shape) == tuple(y_ref.shape)
    close(N(y), y_ref.detach().numpy(), 1e-5, "fwd")
    y.backward(G(gy.numpy()))
    close(N(xd.grad), xr.grad.numpy(), 1e-5, "dgrad")
    close(N(wd.grad), wr.grad.numpy(), 2e-5, "wgrad")
    close(N(bd.grad), br.grad.numpy(), 2e-5, "bias grad")


@pytest.mark.parametrize("conv_mode", ["b3", "b3tile", "b3nstream"], indirect=True)
@pytest.mark.parametrize("case", [(2, 130, 32, 20, 24, 3, 1), (1, 64, 128, 16, 16, 3, 2), (2, 66, 40, 9, 11, 5, 1),
                                  (2, 132, 64, 40, 70, 3, 1)])
def test_conv2d_b3_spike_inputs_take_the_three_term_product_without_changing_results(case, conv_mode):
    """The wave vote of evf_conv_b3gen.hip: binary spikes / small integers / bilinear blends run 3 products per 16
    channels, real-valued channels 6 -- decoder-like inputs mix both (2 flow channels + spikes).  Either way the result
    is the fp32-rounded exact convolution: compare with float64.  (The tiled kernel votes per block and 16-channel
    group, the general one per wave and chunk.)"""
    B, Cin, Cout, H, W, k, s = case
    gen = torch.Generator().manual_seed(11)
    w = torch.randn(Cout, Cin, k, k, generator=gen) * 0.2
    spikes = (torch.rand(B, Cin, H, W, generator=gen) < 0.2).float()
    blend = spikes + torch.randint(0, 3, (B, Cin, H, W), generator=gen).float() / 16.0
    mixed = spikes.clone()
    mixed[:, :2] = torch.randn(B, 2, H, W, generator=gen)
    for name, x in (("spikes", spikes), ("blend", blend), ("mixed", mixed)):
        ref = torch.nn.functional.conv2d(x.double(), w.double(), None, stride=s, padding=k // 2).numpy()
        y = hip_ops.conv_act(object(), G(x.numpy()), G(w.numpy()), None, stride=s, activation=None)
        scale = np.abs(ref).max()
        assert np.abs(N(y) - ref).max() <= 4e-7 * scale * np.sqrt(Cin * k * k / 16.0), name
    # the vote is an optimisation, never a different result: an inexact value far away (another wave's pixels) leaves
    # every output outside its receptive field bit-identical
    a = spikes.clone()
    b = spikes.clone()
    b[0, 0, 0, 0] = 0.3
    ya = N(hip_ops.conv_act(object(), G(a.numpy()), G(w.numpy()), None, stride=s, activation=None))
    yb = N(hip_ops.conv_act(object(), G(b.numpy()), G(w.numpy()), None, stride=s, activation=None))
    r = k // 2 // s + 1
    assert np.array_equal(ya[:, :, r:, :], yb[:, :, r:, :]) and np.array_equal(ya[1:], yb[1:])
    assert not np.array_equal(ya[0, :, 0, 0], yb[0, :, 0, 0])


@pytest.mark.parametrize("conv_mode", ["b3", "b3tile", "b3split", "b3tilesplit", "b3nstream", "b3nstreamsplit"], indirect=True)
@pytest.mark.parametrize("case", [(2, 64, 96, 20, 36, 3, 1), (1, 132, 30, 17, 33, 3, 1), (2, 32, 64, 12, 12, 3, 2), (1, 48, 8, 9, 9, 1, 1),
                                  (2, 64, 96, 16, 12, 3, 1)])
def test_conv2d_b3_bias_and_accumulate_through_every_kernel(case, conv_mode):
    """y (+)= conv(x) + bias and g_x (+)= conv^T(g_y) through the general, tiled and split-K kernels: the accumulate and
    bias paths of their epilogues and of the slab reduction (recurrent cells accumulate the rec conv into the ff conv)."""
    B, Cin, Cout, H, W, k, s = case
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(B, H, W, Cin, generator=gen)
    w = torch.randn(Cout, Cin, k, k, generator=gen) * 0.2
    b = torch.randn(Cout, generator=gen)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), stride=s, padding=k // 2).permute(0, 2, 3, 1)
    y0 = torch.randn(ref.shape, generator=gen)
    xd, wd, bd = G(x.numpy()), G(w.numpy()), G(b.numpy())
    cache = hip_ops._PackCache()
    y = G(y0.numpy())
    hip_ops.conv_fwd(xd, cache.get(wd, 0, 0, Cin), bd, y, Cin, Cout, k, s, accumulate=1)
    close(N(y), (ref + y0.double()).numpy(), 1e-5, "fwd accumulate + bias")
    y = G(y0.numpy())
    hip_ops.conv_fwd(xd, cache.get(wd, 0, 0, Cin), None, y, Cin, Cout, k, s, accumulate=0)
    close(N(y), (ref - b.double()).numpy(), 1e-5, "fwd overwrite")
    # input gradient, accumulate into an existing gradient
    gy = torch.randn(ref.shape, generator=gen)
    xr = x.permute(0, 3, 1, 2).double().requires_grad_(True)
    torch.nn.functional.conv2d(xr, w.double(), None, stride=s, padding=k // 2).backward(gy.permute(0, 3, 1, 2).double())
    gx0 = torch.randn(B, H, W, Cin, generator=gen)
    gx = G(gx0.numpy())
    hip_ops.conv_dgrad(G(gy.numpy()), cache.get(wd, 1, 0, Cin), gx, Cin, Cout, k, s, accumulate=1)
    close(N(gx), (xr.grad.permute(0, 2, 3, 1) + gx0.double()).numpy(), 1e-5, "dgrad accumulate")


@pytest.mark.parametrize("case", [(2, 128, 64, 16, 16), (1, 64, 64, 32, 40), (2, 36, 32, 17, 23), (1, 512, 128, 8, 8), (8, 512, 512, 16, 16),
                                  (2, 64, 130, 24, 24)])
def test_wgrad_fused_slab_reduction_equals_the_three_launch_path(case, monkeypatch):
    """evf_conv2d_wgrad with accumulate bit 2 ("x exactly representable in bf16 by construction"): ONE launch -- the last block of
    every weight tile sums the pixel splits in index order -- against the verified path (bf16 kernel, fp32 redo pass,
    k_wgrad_reduce) and against float64; overwrite and accumulate forms, a channel offset inside a wider weight; and an x that
    breaks the promise gives NaN, never a rounded gradient."""
    B, Cin, Cout, H, W = case
    gen = torch.Generator().manual_seed(B * 1000 + Cin)
    # (both forms of the exact path are exercised: this process runs the default -- reduce launch kept, fp32 pass skipped --, the
    #  fused tail is measured slower and sits behind EVF_WGRAD_FUSE=1, read once per process: test_wgrad_fused_tail_in_a_subprocess)
    # spike-valued input: bilinear x2 blends of {0, 1, 2} sums (multiples of 1/16)
    lo = torch.randint(0, 3, (B, Cin, (H + 1) // 2, (W + 1) // 2), generator=gen).float()
    x = torch.nn.functional.interpolate(lo, scale_factor=2, mode="bilinear", align_corners=False)[:, :, :H, :W].contiguous()
    gy = torch.randn(B, Cout, H, W, generator=gen) * 0.1
    xd = G(x.permute(0, 2, 3, 1).contiguous().numpy())
    gd = G(gy.permute(0, 2, 3, 1).contiguous().numpy())
    ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, 3, 3), gy.double(), padding=1).numpy()
    ct, off = Cin + 8, 4  # the weight tensor is wider than this call's channel range
    L = _lib.load()
    ws = torch.empty(max(L.evf_conv2d_wgrad_ws(B, H, W, Cin, Cout, 3, 1), 1), device=DEV)

    def run(flags, base):
        g_w = base.clone()
        _lib.call("evf_conv2d_wgrad", _lib.ptr(xd), Cin, _lib.ptr(gd), Cout, _lib.ptr(g_w), None, B, H, W, Cin, Cout, 3, 1, ct, off,
                  flags, _lib.ptr(ws))
        torch.cuda.synchronize()
        return N(g_w)

    base = torch.randn(Cout, ct, 3, 3, generator=gen).to(DEV)
    three, fused = run(1, base), run(1 | 4, base)
    scale = np.abs(ref).max()
    assert np.abs(fused - three).max() <= 2e-6 * scale, np.abs(fused - three).max() / scale
    want = N(base).astype(np.float64)
    want[:, off:off + Cin] += ref
    assert np.abs(fused - want).max() <= 1e-5 * scale
    assert np.array_equal(fused[:, :off], N(base)[:, :off]) and np.array_equal(fused[:, off + Cin:], N(base)[:, off + Cin:])
    again = run(1 | 4, base)
    assert np.array_equal(again, fused)  # fixed summation order: the same bits whatever the block schedule
    # overwrite form (the call covers the whole weight)
    zero = torch.full((Cout, Cin, 3, 3), 7.0, device=DEV)
    g1, g2 = zero.clone(), zero.clone()
    for g_w, fl in ((g1, 0), (g2, 4)):
        _lib.call("evf_conv2d_wgrad", _lib.ptr(xd), Cin, _lib.ptr(gd), Cout, _lib.ptr(g_w), None, B, H, W, Cin, Cout, 3, 1, Cin, 0, fl,
                  _lib.ptr(ws))
    assert np.abs(N(g2) - N(g1)).max() <= 2e-6 * scale and np.abs(N(g2) - ref).max() <= 1e-5 * scale
    # a broken promise is loud
    xbad = xd.clone()
    xbad[0, 1, 1, 0] = 0.3
    g3 = zero.clone()
    _lib.call("evf_conv2d_wgrad", _lib.ptr(xbad), Cin, _lib.ptr(gd), Cout, _lib.ptr(g3), None, B, H, W, Cin, Cout, 3, 1, Cin, 0, 4,
              _lib.ptr(ws))
    if Cout % 4 == 0:  # (else the shape is not the bf16 kernel's: the fp32 kernel takes it, exactly, promise or not)
        assert np.isnan(N(g3)).any()
    g4 = zero.clone()  # ... and the tickets were handed back zeroed: the next exact call is clean again
    _lib.call("evf_conv2d_wgrad", _lib.ptr(xd), Cin, _lib.ptr(gd), Cout, _lib.ptr(g4), None, B, H, W, Cin, Cout, 3, 1, Cin, 0, 4,
              _lib.ptr(ws))
    assert np.array_equal(N(g4), N(g2))


@pytest.mark.parametrize("case", [(2, 128, 64, 16, 16), (1, 64, 64, 32, 40), (2, 72, 100, 17, 23), (8, 512, 512, 16, 16), (2, 256, 64, 64, 64),
                                  (1, 64, 128, 9, 130), (3, 68, 64, 8, 8)])
def test_wgrad_two_team_kernel_is_bit_identical_to_the_tap_per_wave_kernel(case):
    """k_wgrad9_b3v (four matrix waves x nine taps + four loader waves, csrc/evf_wgrad_b3gen.hip) against k_wgrad9_b3 (a wave per tap):
    the same products in the same order per accumulator -- the same bits --, and both against float64; bias gradient included;
    spike-valued and real-valued x (the latter raises the redo flags: the fp32 pass recomputes the tiles)."""
    B, Cin, Cout, H, W = case
    gen = torch.Generator().manual_seed(B * 77 + Cin + W)
    L = _lib.load()
    ws = torch.empty(max(L.evf_conv2d_wgrad_ws(B, H, W, Cin, Cout, 3, 1), 1), device=DEV)
    gy = torch.randn(B, Cout, H, W, generator=gen) * 0.1
    gd = G(gy.permute(0, 2, 3, 1).contiguous().numpy())
    for kind in ("spikes", "real"):
        if kind == "spikes":
            x = torch.randint(0, 3, (B, Cin, H, W), generator=gen).float() * (torch.rand(B, Cin, H, W, generator=gen) < 0.4)
        else:
            x = torch.randn(B, Cin, H, W, generator=gen)
        xd = G(x.permute(0, 2, 3, 1).contiguous().numpy())
        ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, 3, 3), gy.double(), padding=1).numpy()
        out = {}
        try:
            for mode in (0, 2):  # (2: the two-team kernel wherever its block shape fits, whatever the tile count)
                _lib.call("evf_wgrad_teams_select", mode)
                g_w = torch.full((Cout, Cin, 3, 3), 3.0, device=DEV)
                g_b = torch.full((Cout,), 5.0, device=DEV)
                _lib.call("evf_conv2d_wgrad", _lib.ptr(xd), Cin, _lib.ptr(gd), Cout, _lib.ptr(g_w), _lib.ptr(g_b), B, H, W, Cin, Cout, 3, 1,
                          Cin, 0, 0, _lib.ptr(ws))
                torch.cuda.synchronize()
                out[mode] = (N(g_w), N(g_b))
        finally:
            _lib.call("evf_wgrad_teams_select", 0)
        assert np.array_equal(out[0][0], out[2][0]), (kind, np.abs(out[0][0] - out[2][0]).max())
        scale = np.abs(ref).max()
        assert np.abs(out[2][0] - ref).max() <= 1e-5 * scale, kind
        bref = gy.double().sum((0, 2, 3)).numpy()
        for m in (0, 2):
            assert np.abs(out[m][1] - bref).max() <= 1e-5 * max(np.abs(bref).max(), 1.0), (kind, m)


@pytest.mark.parametrize("case", [(8, 512, 512), (4, 128, 256), (2, 64, 512), (8, 1024, 128), (8, 256, 512), (16, 128, 512)])
def test_small_image_exact_input_conv_equals_the_voting_kernels(case):
    """evf_conv2d_fwd_b3 with accumulate bit 2 ("x exactly representable in bf16 by construction") on 16 x 16 images: the
    two-images-per-block kernel of csrc/evf_conv_b3small.hip against the voting kernels (same exact products: fp32 round-off
    of another summation order) and against float64; overwrite / accumulate / bias; a broken promise gives NaN."""
    B, Cin, Cout = case
    H = W = 16
    gen = torch.Generator().manual_seed(B + Cin)
    x = torch.randint(0, 3, (B, Cin, H, W), generator=gen).float() * (torch.rand(B, Cin, H, W, generator=gen) < 0.3)
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) * 0.05
    bias = torch.randn(Cout, generator=gen)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), bias.double(), padding=1).permute(0, 2, 3, 1).numpy()
    xd = G(x.permute(0, 2, 3, 1).contiguous().numpy())
    wd, bd = G(w.numpy()), G(bias.numpy())
    L = _lib.load()
    wp = torch.empty(L.evf_conv2d_b3_packed_size(Cout, Cin, 3, 0), device=DEV)
    _lib.call("evf_pack_conv2d_weight_b3", _lib.ptr(wd), Cout, Cin, 3, 0, Cin, 0, _lib.ptr(wp))
    nws = L.evf_conv2d_b3_ws(B, H, W, Cout)
    ws = torch.empty(max(nws, 1), device=DEV)

    def run(flags, base):
        y = base.clone()
        _lib.call("evf_conv2d_fwd_b3", _lib.ptr(xd), Cin, _lib.ptr(wp), _lib.ptr(bd), _lib.ptr(y), Cout, B, H, W, Cin, Cout, 3, 1, flags,
                  _lib.ptr(ws), nws)
        torch.cuda.synchronize()
        return N(y)

    base = torch.randn(B, H, W, Cout, generator=gen).to(DEV)
    scale = np.abs(ref).max()
    v0, e0 = run(0, base), run(4, base)
    assert np.abs(e0 - v0).max() <= 2e-6 * scale and np.abs(e0 - ref).max() <= 1e-5 * scale
    v1, e1 = run(1, base), run(1 | 4, base)
    assert np.abs(e1 - v1).max() <= 2e-6 * scale and np.abs(e1 - (ref + N(base))).max() <= 1e-5 * scale
    assert np.array_equal(run(4, base), e0)  # deterministic split sums
    xd[0, 3, 5, 1] = 0.3  # not a multiple of 1/16 in 8 bits
    # (evf_conv3_b3s_plan: enough blocks, and no more K splits than the 8 slabs of scratch: else the voting kernels take the call, exactly)
    applies = (B // 2) * (Cout // 64) * (Cin // 64) >= 128 and Cin // 64 <= 8
    assert np.isnan(run(4, base)).any() == applies and not np.isnan(run(0, base)).any()


@pytest.mark.parametrize("case", [(16, 64, 64, 64, 64, 0), (2, 132, 32, 80, 130, 4), (16, 260, 64, 32, 32, 2), (2, 1024, 256, 32, 32, 0),
                                  (8, 96, 40, 48, 100, 0), (4, 520, 128, 32, 64, 4)])  # (shapes evf_conv3_b3x_plan accepts: >= 96 blocks)
def test_exact_input_conv_on_spatial_tiles_and_behind_a_real_valued_head(case):
    """csrc/evf_conv_b3small.hip, GEOM 1 (16 x 32 tiles of larger images) and the real-valued head: a decoder input whose first
    channels are flow values (exact_from = 2 / 4: channels 0..15 run as the exact 3-way split) -- against the voting kernels and
    float64; ragged tiles / channel counts; K splits; a broken promise behind exact_from gives NaN, real values IN the head do not."""
    B, Cin, Cout, H, W, ef = case
    gen = torch.Generator().manual_seed(Cin + W)
    x = torch.randint(0, 33, (B, Cin, H, W), generator=gen).float() / 16.0 * (torch.rand(B, Cin, H, W, generator=gen) < 0.4)
    if ef:
        x[:, :ef] = torch.randn(B, ef, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) * 0.05
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None, padding=1).permute(0, 2, 3, 1).numpy()
    xd = G(x.permute(0, 2, 3, 1).contiguous().numpy())
    wd = G(w.numpy())
    L = _lib.load()
    wp = torch.empty(L.evf_conv2d_b3_packed_size(Cout, Cin, 3, 0), device=DEV)
    _lib.call("evf_pack_conv2d_weight_b3", _lib.ptr(wd), Cout, Cin, 3, 0, Cin, 0, _lib.ptr(wp))
    nws = L.evf_conv2d_b3_ws(B, H, W, Cout)
    ws = torch.empty(max(nws, 1), device=DEV)

    def run(flags):
        y = torch.full((B, H, W, Cout), 5.0, device=DEV)
        _lib.call("evf_conv2d_fwd_b3", _lib.ptr(xd), Cin, _lib.ptr(wp), None, _lib.ptr(y), Cout, B, H, W, Cin, Cout, 3, 1, flags,
                  _lib.ptr(ws), nws)
        torch.cuda.synchronize()
        return N(y)

    scale = np.abs(ref).max()
    fl = 4 | (ef << 4)
    v0, e0 = run(0), run(fl)
    assert np.abs(e0 - v0).max() <= 3e-6 * scale and np.abs(e0 - ref).max() <= 1e-5 * scale, (np.abs(e0 - v0).max() / scale, np.abs(e0 - ref).max() / scale)
    assert np.array_equal(run(fl), e0)
    xd[0, H // 2, W // 2, Cin - 3] = 0.3  # behind exact_from: the promise is broken
    assert np.isnan(run(fl)).any() and not np.isnan(run(0)).any()


@pytest.mark.parametrize("case", [(2, 32, 24, 40), (1, 64, 17, 130), (2, 128, 9, 64), (1, 256, 5, 7)])
def test_wgrad_of_a_four_channel_input_streams(case):
    """3x3 stride-1 weight gradient with Cin = 4 (the real-valued head of a decoder input): k_wgrad9_fewin (a thread per output
    channel sliding a 3 x 3 float4 window along a row) against float64, inside a wider weight tensor, accumulating."""
    B, Cout, H, W = case
    gen = torch.Generator().manual_seed(Cout + W)
    Ctot = 12  # the activation is wider than the four channels of this call
    x = torch.randn(B, Ctot, H, W, generator=gen)
    gy = torch.randn(B, Cout, H, W, generator=gen) * 0.1
    xd = G(x.permute(0, 2, 3, 1).contiguous().numpy())
    gd = G(gy.permute(0, 2, 3, 1).contiguous().numpy())
    ref = torch.nn.grad.conv2d_weight(x[:, :4].double(), (Cout, 4, 3, 3), gy.double(), padding=1).numpy()
    L = _lib.load()
    ws = torch.empty(max(L.evf_conv2d_wgrad_ws(B, H, W, 4, Cout, 3, 1), 1), device=DEV)
    base = torch.randn(Cout, Ctot, 3, 3, generator=gen).to(DEV)
    g_w = base.clone()
    _lib.call("evf_conv2d_wgrad", _lib.ptr(xd), Ctot, _lib.ptr(gd), Cout, _lib.ptr(g_w), None, B, H, W, 4, Cout, 3, 1, Ctot, 0, 1 | 2,
              _lib.ptr(ws))
    want = N(base).astype(np.float64)
    want[:, :4] += ref
    assert np.abs(N(g_w) - want).max() <= 1e-5 * np.abs(ref).max()
    assert np.array_equal(N(g_w)[:, 4:], N(base)[:, 4:])


def test_concat_up2_in_one_kernel_equals_the_two_kernels():
    """hip_ops.concat_up2 = upsample2x_bilinear(concat_channels(parts, pad)) bit for bit (the same blend per element), its gradients
    to the parts equal too; odd channel counts fall back to the two kernels."""
    torch.manual_seed(3)
    B, H, W = 2, 9, 13
    parts = [torch.randn(B, c, H, W, device=DEV) for c in (2, 16, 16)]
    a = [p.clone().requires_grad_(True) for p in parts]
    b = [p.clone().requires_grad_(True) for p in parts]
    y1 = hip_ops.concat_up2(a, 2)
    y2 = hip_ops.upsample2x_bilinear(hip_ops.concat_channels(b, 2))
    assert y1.shape == y2.shape == (B, 36, 2 * H, 2 * W) and torch.equal(y1, y2)
    g = torch.randn_like(y1)
    y1.backward(g)
    y2.backward(g)
    for p, q in zip(a, b):
        assert torch.equal(p.grad, q.grad)
    ref = torch.nn.functional.interpolate(torch.cat(parts + [torch.zeros(B, 2, H, W, device=DEV)], 1), scale_factor=2, mode="bilinear",
                                          align_corners=False)
    assert torch.allclose(y1, ref, atol=1e-6)
    odd = [torch.randn(B, 3, H, W, device=DEV), torch.randn(B, 5, H, W, device=DEV)]
    assert torch.equal(hip_ops.concat_up2(odd), hip_ops.upsample2x_bilinear(hip_ops.concat_channels(odd)))


def test_wgrad_fused_tail_in_a_subprocess():
    """The same test with EVF_WGRAD_FUSE=1 (the last block of a weight tile reduces the pixel splits itself)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_general.py"), "-m", "gpu", "-x", "-q", "-k",
                          "wgrad_fused_slab_reduction"], capture_output=True, text=True, timeout=900, cwd=root,
                         env=dict(os.environ, EVF_WGRAD_FUSE="1"))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]


def test_spike_provenance_tags_follow_the_evflownet_dataflow():
    """hip_ops.spike_tag: cell outputs, residual sums, channel concatenations, zero paddings and ONE bilinear x2 blend keep the
    "exactly representable in bf16" provenance (which lets conv_wgrad skip its verification pass); a flow prediction in front of
    a concatenation moves `exact_from` behind it, a foreign op or a second blend drops the tag."""
    from event_flow_amd.models.model_util import _centred

    torch.manual_seed(0)
    cell = cells.ConvLIF(8, 16, 3).to(DEV)
    x = (torch.rand(1, 8, 12, 12, device=DEV) > 0.5).float()
    out, st = cell(x, None)
    assert hip_ops.spike_tag(out) == (1.0, 0) and hip_ops.spike_tag(x) is None
    cell2 = cells.ConvLIF(16, 16, 3).to(DEV)
    out2, _ = cell2(out, None, residual=out)
    assert hip_ops.spike_tag(out2) == (2.0, 0)
    flow = torch.randn(1, 2, 12, 12, device=DEV)
    cat = hip_ops.concat_channels([flow, out, out2], pad=2)
    assert hip_ops.spike_tag(cat) == (2.0, 2)
    up = hip_ops.upsample2x_bilinear(cat)
    assert hip_ops.spike_tag(up) == (2.0, 2)
    assert hip_ops.spike_tag(hip_ops.upsample2x_bilinear(up)) is None  # 1/256 steps: no longer 8 significant bits
    assert hip_ops.spike_tag(_centred(out, torch.zeros(1, 1, 13, 14))) == (1.0, 0)
    assert hip_ops.spike_tag(out * 1.0) is None  # a foreign op: the verified path
    # the values really are what the tag promises
    v = up[:, 2:].detach()
    assert torch.equal(v, v.bfloat16().float())


def test_conv_bad_arguments_fail_loudly():
    x = torch.zeros(1, 4, 8, 8, device=DEV)
    w = torch.zeros(4, 4, 4, 4, device=DEV)  # even kernel sizes do not exist in the reference (padding k/2) and are not built
    with pytest.raises(_lib.EvflowError):
        hip_ops.conv_act(object(), x, w, None)
    with pytest.raises(_lib.EvflowError):
        hip_ops.conv_act(object(), x.cpu(), torch.zeros(4, 4, 3, 3), None)  # CPU tensors: no fallback


def test_upsampling_matches_torch_semantics():
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(2, 6, 5, 7, generator=gen)
    xr = x.clone().requires_grad_(True)
    y_ref = torch.nn.functional.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=False)
    gy = torch.randn(y_ref.shape, generator=gen)
    y_ref.backward(gy)
    xd = G(x.numpy()).requires_grad_(True)
    y = hip_ops.upsample2x_bilinear(xd)
    np.testing.assert_allclose(N(y), y_ref.detach().numpy(), rtol=1e-6, atol=1e-6)
    y.backward(G(gy.numpy()))
    np.testing.assert_allclose(N(xd.grad), xr.grad.numpy(), rtol=1e-5, atol=1e-6)
    f = torch.randn(2, 2, 4, 6, generator=gen)
    fr = f.clone().requires_grad_(True)
    u_ref = torch.nn.functional.interpolate(fr, scale_factor=(4.0, 4.0))
    gu = torch.randn(u_ref.shape, generator=gen)
    u_ref.backward(gu)
    fd = G(f.numpy()).requires_grad_(True)
    u = hip_ops.upsample_nearest(fd, 4.0)
    assert np.array_equal(N(u), u_ref.detach().numpy())
    u.backward(G(gu.numpy()))
    np.testing.assert_allclose(N(fd.grad), fr.grad.numpy(), rtol=1e-6, atol=1e-6)


# ------------------------------------------------------------------ stand-alone cells (G6)
CELL_CLS = {
    ("lif", False): cells.ConvLIF, ("lif", True): cells.ConvLIFRecurrent,
    ("plif", False): cells.ConvPLIF, ("plif", True): cells.ConvPLIFRecurrent,
    ("alif", False): cells.ConvALIF, ("alif", True): cells.ConvALIFRecurrent,
    ("xlif", False): cells.ConvXLIF, ("xlif", True): cells.ConvXLIFRecurrent,
}


def _thresh_of(c, g, tag, new_ref):
    """Effective threshold of the golden step (to mask borderline neurons)."""
    if c["kind"] in ("lif", "plif"):
        return np.maximum(g[tag + "_param_thresh"], 0.01)[None]
    t0 = np.maximum(g[tag + "_param_t0"], 0.01)[None]
    t1 = np.maximum(g[tag + "_param_t1"], 0.0)[None]
    return t0 + t1 * new_ref[2]


@pytest.mark.parametrize("fix,n", [("g6_cells", 28), ("g15_cells_k5", 8), ("g18_cells_weightnorm", 3), ("g19_cells_groupnorm", 3)])
def test_g6_cells_forward_backward_on_gpu(fix, n):
    """G6: 3x3 cells; G15: the 5x5 (models/unet.py:51 default) and 7x7 kernels of the general conv path, stride 1 / 2; G18: LIF
    cells with norm="weight" (evf_weight_norm_fwd / _bwd under the reference's weight_g / weight_v parameters); G19: with
    norm="group" (hip_ops.group_norm1 on the input and -- recurrent cell -- on the previous spikes)."""
    g = load_golden(fix)
    cases = golden_cases(g)
    assert len(cases) == n
    for c in cases:
        tag = c["tag"]
        x_np, st_np = g[tag + "_x"], g[tag + "_state"]
        Cin, C = x_np.shape[1], st_np.shape[2]
        kw = dict(activation=c["act"], hard_reset=c["hard_reset"], act_width=float(g[tag + "_param_act_width"]))
        if c["kind"] in ("alif", "xlif"):
            kw["learn_thresh"] = True
        if not c["recurrent"] and c.get("stride", 1) != 1:
            kw["stride"] = c["stride"]
        if c.get("norm"):
            kw["norm"] = c["norm"]
        cell = CELL_CLS[(c["kind"], c["recurrent"])](Cin, C, c.get("ksz", 3), **kw).to(DEV)
        sd = {k[len(tag + "_param_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + "_param_")}
        cell.load_state_dict(sd)
        x = G(x_np).requires_grad_(True)
        st = G(st_np).requires_grad_(True)
        out, new = cell(x, st)
        new_ref = g[tag + "_new"]
        safe = np.abs(new_ref[0] - _thresh_of(c, g, tag, new_ref)) > 1e-5
        assert np.array_equal(N(out)[safe], g[tag + "_out"][safe]), c
        assert safe.mean() > 0.999
        assert np.array_equal(N(out), g[tag + "_out"]), c  # none of the golden cases is borderline
        # (5x5 / 7x7 on real-valued inputs: 25-49 x Cin products per output, each exact up to the terms below 2^-24 of its
        # leading one that the six-term bf16 product drops -- about twice the round-off of an fp32 accumulation)
        np.testing.assert_allclose(N(new), new_ref, rtol=1e-5, atol=2e-6 if c.get("ksz", 3) == 3 else 1e-5, err_msg=str(c))
        params = dict(cell.named_parameters())
        grads = torch.autograd.grad([out, new], [x, st] + list(params.values()), [G(g[tag + "_g_out"]), G(g[tag + "_g_new"])],
                                    allow_unused=True)
        close(N(grads[0]), g[tag + "_gx"], 2e-5, f"{c} gx")
        close(N(grads[1]), g[tag + "_gstate"], 2e-5, f"{c} gstate")
        for (pn, p), gr in zip(params.items(), grads[2:]):
            ref = g[f"{tag}_grad_{pn}"]
            got = N(gr) if gr is not None else np.zeros_like(ref)
            close(got, ref, 1e-4, f"{c} {pn}")


def test_cell_first_step_without_state_and_residual():
    torch.manual_seed(0)
    cell = cells.ConvLIFRecurrent(8, 8, 3, thresh=(0.3, 0.1)).to(DEV)
    x = (torch.rand(1, 8, 6, 9, device=DEV) < 0.4).float()
    res = torch.rand(1, 8, 6, 9, device=DEV).round()
    out, st = cell(x, None, residual=res)
    p = {"c." + k: v.detach().cpu() for k, v in cell.state_dict().items()}
    o_ref, st_ref = osnn.cell_step("lif", p, "c.", x.cpu(), None, recurrent=True, residual=res.cpu())
    assert np.array_equal(N(out), o_ref.numpy())
    np.testing.assert_allclose(N(st), torch.stack(st_ref).numpy(), rtol=1e-5, atol=1e-6)
    assert tuple(st.shape) == (2, 1, 8, 6, 9)


# ------------------------------------------------------------------ ANN FireNet (G8, BASELINE config 1)
def _ann_cfg():
    return {"num_bins": 2, "base_num_channels": 32, "kernel_size": 3, "encoding": "voxel", "norm_input": False,
            "mask_output": True, "activations": ["relu", None], "spiking_neuron": None}


def test_g8_firenet_ann_forward_and_backward():
    g = load_golden("g8_firenet_ann")
    model = FireNet(_ann_cfg()).to(DEV)
    sd = {k[len("param_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param_")}
    model.load_state_dict(sd)
    flows = []
    for i in range(2):
        out = model(G(g[f"p{i}_event_voxel"]), G(g[f"p{i}_event_cnt"]), log=(i == 1))
        flows.append(out["flow"][0])
        np.testing.assert_allclose(N(out["flow"][0]), g[f"p{i}_flow"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(N(model.states[1]), g[f"p{i}_state_G1"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(N(model.states[4]), g[f"p{i}_state_G2"], rtol=1e-4, atol=1e-6)
    assert set(out["activity"]) == {"0:input", "1:head", "2:G1", "3:R1a", "4:R1b", "5:G2", "6:R2a", "7:R2b", "8:pred"}
    # BPTT over the two passes against the CPU oracle (same parameters, same inputs)
    loss = sum((f * f).sum() for f in flows)
    loss.backward()
    params = {k: torch.from_numpy(g["param_" + k]).clone().requires_grad_(True) for k in sd}
    states = [None] * 7
    tot = 0
    for i in range(2):
        f, states = osnn.firenet_forward("FireNet", params, torch.from_numpy(g[f"p{i}_event_voxel"]), states)
        tot = tot + (f * f).sum()
    tot.backward()
    np.testing.assert_allclose(float(loss.detach()), float(tot.detach()), rtol=1e-5)
    for k, p in model.named_parameters():
        close(N(p.grad), params[k].grad.numpy(), 2e-4, k)


# ------------------------------------------------------------------ ANN comparison FireNets (G10)
ANN_VARIANTS = {
    "FireFlowNet": (("relu", "relu"), None),
    "RNNFireNet": (("relu", None), None),
    "LeakyFireNet": (("relu", None), {"leak": [-1.0, 0.5], "learn_leak": True}),
    "LeakyFireFlowNet": (("relu", "tanh"), {"leak": [-1.0, 0.5], "learn_leak": True}),
}


@pytest.mark.parametrize("name", sorted(ANN_VARIANTS))
def test_g10_ann_firenet_variants(name):
    """FireFlowNet / RNNFireNet / LeakyFireNet / LeakyFireFlowNet against the reference's outputs: flows of three
    passes, final states, BPTT gradients of every parameter (incl. the per-channel leaks)."""
    from event_flow_amd.models import model as M

    g = load_golden("g10_ann_firenets")
    acts, neuron = ANN_VARIANTS[name]
    cfg = {"num_bins": 2, "base_num_channels": 8, "kernel_size": 3, "encoding": "cnt", "norm_input": False,
           "mask_output": True, "activations": list(acts), "spiking_neuron": neuron}
    model = getattr(M, name)(cfg).to(DEV)
    pre = name + ".param_"
    sd = {k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}
    model.load_state_dict(sd)
    tot = 0
    for i in range(3):
        x = G(g[f"p{i}_event_cnt"])
        flow = model(x, x)["flow"][0]
        np.testing.assert_allclose(N(flow), g[f"{name}.p{i}_flow"], rtol=1e-4, atol=1e-6)
        tot = tot + flow.pow(2).sum() + flow.sum()
    states = model.states
    for li in range(7):
        key = f"{name}.state{li}"
        if key in g.files:
            np.testing.assert_allclose(N(states[li]), g[key], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(float(tot.detach()), float(g[f"{name}.loss"]), rtol=1e-5)
    tot.backward()
    for k, p in model.named_parameters():
        close(N(p.grad), g[f"{name}.grad_{k}"], 2e-4, k)
    model.detach_states()
    model.reset_states()


# ------------------------------------------------------------------ non-spiking EV-FlowNets (G11)
ANN_UNETS = {
    "EVFlowNet": None,
    "RecEVFlowNet": None,
    "RNNRecEVFlowNet": None,
    "LeakyRecEVFlowNet": {"leak": [-1.0, 0.5], "learn_leak": True},
}


@pytest.mark.parametrize("name", sorted(ANN_UNETS))
def test_g11_ann_evflownets(name):
    """EVFlowNet / RecEVFlowNet (ConvGRU) / RNNRecEVFlowNet / LeakyRecEVFlowNet against the reference's outputs:
    4 flow scales of two passes, final states, BPTT gradients of every parameter."""
    from event_flow_amd.models import model as M

    g = load_golden("g11_ann_unets")
    cfg = {"num_bins": 2, "base_num_channels": 4, "kernel_size": 3, "encoding": "cnt", "norm_input": False,
           "mask_output": True, "activations": ["relu", None], "spiking_neuron": ANN_UNETS[name]}
    model = getattr(M, name)(cfg).to(DEV)
    pre = name + ".param_"
    sd = {k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}
    model.load_state_dict(sd)
    tot = 0
    for i in range(2):
        x = G(g[f"p{i}_event_cnt"])
        flows = model(x, x)["flow"]
        assert len(flows) == 4
        for s_, f in enumerate(flows):
            close(N(f), g[f"{name}.p{i}_flow{s_}"], 1e-4, f"pass {i} scale {s_}")
            tot = tot + f.pow(2).sum() + f.sum()
    if name != "EVFlowNet":
        states = model.states
        for si, st in enumerate(states):
            ref = g[f"{name}.state{si}"]
            assert tuple(st.shape) == ref.shape
            close(N(st), ref, 1e-4, f"state {si}")
    np.testing.assert_allclose(float(tot.detach()), float(g[f"{name}.loss"]), rtol=2e-5)
    tot.backward()
    for k, p in model.named_parameters():
        close(N(p.grad), g[f"{name}.grad_{k}"], 3e-4, k)
    model.detach_states()
    model.reset_states()


# ------------------------------------------------------------------ E2VID (G12)
def test_g12_e2vid():
    """E2VID (ConvLSTM encoders, skip 'sum') against the reference's outputs: flows of three passes, final
    (hidden, cell) states, BPTT gradients of every parameter."""
    from event_flow_amd.models import model as M

    g = load_golden("g12_e2vid")
    cfg = {"num_bins": 2, "base_num_channels": 4, "kernel_size": 3, "encoding": "cnt", "norm_input": False,
           "mask_output": True, "activations": ["relu", None], "spiking_neuron": None}
    model = M.E2VID(cfg).to(DEV)
    sd = {k[len("param_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param_")}
    model.load_state_dict(sd)
    tot = 0
    for i in range(3):
        x = G(g[f"p{i}_event_cnt"])
        out = model(x, x)
        assert len(out["flow"]) == 1 and out["activity"] is None
        close(N(out["flow"][0]), g[f"p{i}_flow"], 1e-4, f"pass {i}")
        tot = tot + out["flow"][0].pow(2).sum() + out["flow"][0].sum()
    states = model.states
    assert len(states) == 3 and all(type(s_) is tuple and len(s_) == 2 for s_ in states)
    for si, (h, c) in enumerate(states):
        close(N(h), g[f"state{si}_hidden"], 1e-4, f"hidden {si}")
        close(N(c), g[f"state{si}_cell"], 1e-4, f"cell {si}")
    np.testing.assert_allclose(float(tot.detach()), float(g["loss"]), rtol=2e-5)
    tot.backward()
    for k, p in model.named_parameters():
        close(N(p.grad), g["grad_" + k], 3e-4, k)
    model.detach_states()
    assert all(not h.requires_grad for st in model.unetrecurrent.states for h in st)
    model.reset_states()
    assert model.unetrecurrent.states == [None] * 3


# ------------------------------------------------------------------ reference checkpoint (G14)
def test_g14_reference_checkpoint_gives_the_reference_flow():
    """A checkpoint pickled by the reference (whole model object under an MLflow run id) restored into the HIP model
    reproduces the flow the reference computed with it."""
    import os

    from event_flow_amd.models.model import LIFFireNet
    from event_flow_amd.utils.utils import load_model

    g = load_golden("g14_checkpoint")
    cfg = {"num_bins": 2, "base_num_channels": 8, "kernel_size": 3, "encoding": "cnt", "norm_input": False, "mask_output": True,
           "activations": ["arctanspike", "arctanspike"],
           "spiking_neuron": {"leak": [-4.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True, "learn_thresh": True, "hard_reset": True}}
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mlruns")
    model = load_model("0123456789abcdef0123456789abcdef", LIFFireNet(cfg).to(DEV), DEV, root=root)
    x = G(g["event_cnt"])
    with torch.no_grad():
        flow = model(x, x)["flow"][0]
    np.testing.assert_allclose(N(flow), g["flow"], rtol=1e-4, atol=1e-7)


# ------------------------------------------------------------------ spiking EV-FlowNet (G9, BASELINE config 4 architecture)
def _unet_cfg(C=4):
    return {"num_bins": 2, "base_num_channels": C, "kernel_size": 3, "encoding": "cnt", "norm_input": False,
            "mask_output": True, "activations": ["arctanspike", "arctanspike"],
            "spiking_neuron": {"leak": [-4.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True, "learn_thresh": True,
                               "hard_reset": True}}


def test_g9_spiking_evflownet_forward_states_and_gradients():
    g = load_golden("g9_spiking_unet")
    model = SpikingRecEVFlowNet(_unet_cfg(4)).to(DEV)
    sd = {k[len("param_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param_")}
    model.load_state_dict(sd)
    assert sum(p.numel() for p in model.parameters()) == sum(v.numel() for k, v in sd.items() if "act_width" not in k)
    for i in range(2):
        out = model(G(g[f"p{i}_event_cnt"]), G(g[f"p{i}_event_cnt"]))
        flows = out["flow"]
        assert len(flows) == 4 and out["activity"] is None
        states = model.states
        assert len(states) == 10
        for s in range(10):
            ref = g[f"p{i}_state{s}"]
            assert tuple(states[s].shape) == ref.shape
            close(N(states[s]), ref, 1e-5, f"pass {i} state {s}")  # v = sum of up to 9*130 fp32 products: summation-order noise
        for s, f in enumerate(flows):
            np.testing.assert_allclose(N(f), g[f"p{i}_flow{s}"], rtol=1e-4, atol=1e-6)
    tot = sum(f.pow(2).sum() for f in flows)
    tot.backward()
    for k, p in model.named_parameters():
        ref = g["grad_" + k]
        got = N(p.grad) if p.grad is not None else np.zeros_like(ref)
        assert np.abs(got - ref).max() <= 2e-4 * max(np.abs(ref).max(), 1e-10) + 1e-12, k
    with pytest.raises(NotImplementedError):
        model(G(g["p0_event_cnt"]), G(g["p0_event_cnt"]), log=True)  # reference model.py:523-524
    model.detach_states()
    model.reset_states()
    assert model.states == [None] * 10


def test_evflownet_cropping_and_odd_sizes():
    torch.manual_seed(1)
    model = SpikingRecEVFlowNet(_unet_cfg(4)).to(DEV)
    model.init_cropping(30, 20)  # width, height -> padded to 32 x 32
    x = (torch.rand(1, 2, 20, 30, device=DEV) < 0.3).float()
    out = model(x, x)
    assert all(tuple(f.shape) == (1, 2, 20, 30) for f in out["flow"])


# ------------------------------------------------------------------ ALIF / XLIF FireNets vs the CPU oracle
@pytest.mark.parametrize("name,cls", [("ALIFFireNet", ALIFFireNet), ("XLIFFireNet", XLIFFireNet)])
def test_adaptive_threshold_firenets_vs_oracle(name, cls):
    torch.manual_seed(5)
    neuron = {"leak_v": [-4.0, 0.1], "t0": [0.3, 0.05], "t1": [0.5, 0.1], "learn_leak": True, "learn_thresh": True}
    neuron["leak_t" if name == "ALIFFireNet" else "leak_pt"] = [-2.0, 0.1]
    cfg = {"num_bins": 2, "base_num_channels": 32, "kernel_size": 3, "encoding": "cnt", "norm_input": False,
           "mask_output": True, "activations": ["arctanspike", "arctanspike"], "spiking_neuron": neuron}
    model = cls(cfg).to(DEV)
    params = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    keys = [k for k, _ in model.named_parameters()]
    for k in keys:
        params[k].requires_grad_(True)
    xs = [(torch.rand(2, 2, 16, 20) < 0.5).float() * torch.randint(1, 4, (2, 2, 16, 20)).float() for _ in range(3)]
    states = [None] * 7
    tot_ref, tot = 0, 0
    for x in xs:
        f_ref, states = osnn.firenet_forward(name, params, x, states)
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


def test_weight_normalised_lif_firenet_vs_oracle():
    """LIFFireNet whose cells are built with norm="weight" (spiking_neuron kwargs reach the cells, models/model.py:179-186): the
    state_dict carries the reference's ff.weight_g / ff.weight_v (rec.* on the recurrent cells), the network runs cell by cell
    through the general path (the fused engine takes plain weights), flows / states / every parameter gradient -- weight_g and
    weight_v included -- against the CPU oracle over three passes."""
    from event_flow_amd.models.model import LIFFireNet

    torch.manual_seed(9)
    neuron = {"leak": [-4.0, 0.1], "thresh": [0.3, 0.05], "learn_leak": True, "learn_thresh": True, "hard_reset": True, "norm": "weight"}
    cfg = {"num_bins": 2, "base_num_channels": 32, "kernel_size": 3, "encoding": "cnt", "norm_input": False,
           "mask_output": True, "activations": ["arctanspike", "arctanspike"], "spiking_neuron": neuron}
    model = LIFFireNet(cfg).to(DEV)
    assert not model._fused()
    sd_keys = set(model.state_dict())
    assert {"head.ff.weight_g", "head.ff.weight_v", "G1.rec.weight_g", "G1.rec.weight_v", "R2b.ff.weight_v"} <= sd_keys
    assert "head.ff.weight" not in sd_keys
    with torch.no_grad():
        for k, p in model.named_parameters():
            if k.endswith("weight_g"):
                p.mul_(torch.rand_like(p) + 0.5)  # (at construction g = ||v||, i.e. w = v)
    params = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    keys = [k for k, _ in model.named_parameters()]
    for k in keys:
        params[k].requires_grad_(True)
    xs = [(torch.rand(2, 2, 16, 20) < 0.5).float() * torch.randint(1, 4, (2, 2, 16, 20)).float() for _ in range(3)]
    states = [None] * 7
    tot_ref, tot = 0, 0
    for x in xs:
        f_ref, states = osnn.firenet_forward("LIFFireNet", params, x, states)
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


def test_evflownet_at_config4_width_vs_oracle():
    """BASELINE config 4 architecture at its real width (base 32: 64..512 channels, 1024/514/258/130-channel decoder
    inputs, 20.4 M parameters) on a 128x128 crop, B=1, two passes: flows of all four scales and the parameter
    gradients against the CPU oracle.  (256x256 x B=8 runs in tools/bench_evflownet.py; the oracle needs minutes there.)"""
    torch.manual_seed(11)
    model = SpikingRecEVFlowNet(_unet_cfg(32)).to(DEV)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if k.endswith("thresh"):
                p.mul_(0.3)  # keep every layer active at this input rate
    params = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    keys = [k for k, _ in model.named_parameters()]
    for k in keys:
        params[k].requires_grad_(True)
    xs = [(torch.rand(1, 2, 128, 128) < 0.3).float() * torch.randint(1, 4, (1, 2, 128, 128)).float() for _ in range(2)]
    torch.set_num_threads(16)
    states = [None] * 10
    tot, tot_ref, nflip, ntot = 0, 0, 0, 0
    for x in xs:
        flows_ref, states = osnn.spiking_unet_forward("lif", params, x, states)
        out = model(x.to(DEV), x.to(DEV))
        for f, fr in zip(out["flow"], flows_ref):
            tot = tot + (f * f).sum()
            tot_ref = tot_ref + (fr * fr).sum()
    got = model.states
    for s in range(10):
        ref = states[s]
        ref = torch.stack([torch.stack(t) for t in ref]) if isinstance(ref[0], tuple) else torch.stack(ref)
        z_got, z_ref = N(got[s])[..., 1, :, :, :, :] if ref.dim() == 6 else N(got[s])[1], None
        z_ref = ref.detach().numpy()[..., 1, :, :, :, :] if ref.dim() == 6 else ref.detach().numpy()[1]
        nflip += int((z_got != z_ref).sum())
        ntot += z_ref.size
    assert nflip <= 1e-4 * ntot, (nflip, ntot)
    rel = 1e-4 if nflip == 0 else 5e-2
    for f, fr in zip(out["flow"], flows_ref):
        close(N(f), fr.detach().numpy(), rel if nflip == 0 else 0.5, "flow")
    tot.backward()
    tot_ref.backward()
    gn = float(np.sqrt(sum(float((params[k].grad.numpy() ** 2).sum()) for k in keys if params[k].grad is not None)))
    err = 0.0
    for k, p in model.named_parameters():
        ref = params[k].grad
        ref = ref.numpy() if ref is not None else np.zeros(tuple(p.shape), np.float32)
        g = N(p.grad) if p.grad is not None else np.zeros_like(ref)
        err += float(((g - ref) ** 2).sum())
    assert np.sqrt(err) <= (2e-3 if nflip == 0 else 1e-1) * gn, (np.sqrt(err) / gn, nflip)


@pytest.mark.parametrize("name", ["arctanspike", "superspike", "trianglespike", "mgspike"])
def test_standalone_spike_functions(name):
    """models/spiking_util.py API: forward Heaviside of (x - thresh), backward = the surrogate (oracle formulas)."""
    from event_flow_amd.models import spiking_util as su

    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 5, 7, generator=g)
    th = torch.tensor([0.2, -0.1, 0.4, 0.0, 0.3]).view(1, 5, 1)
    width = {"arctanspike": 10.0, "superspike": 10.0, "trianglespike": 1.0, "mgspike": 0.5}[name]
    xd = x.to(DEV).requires_grad_(True)
    z = getattr(su, name)(xd, th.to(DEV), torch.tensor(width))
    assert z.dtype == torch.float32 and np.array_equal(N(z), (x - th).gt(0).float().numpy())
    gy = torch.randn(x.shape, generator=g)
    z.backward(gy.to(DEV))
    ref = gy * osnn.surrogate(name, x - th, width)
    np.testing.assert_allclose(N(xd.grad), ref.numpy(), rtol=1e-5, atol=1e-7)
    z1 = getattr(su, name)(x.to(DEV))  # reference defaults: thresh = 1.0
    assert np.array_equal(N(z1), (x - 1.0).gt(0).float().numpy())
    with pytest.raises(_lib.EvflowError):
        getattr(su, name)(x)  # CPU tensor: no fallback


# ------------------------------------------------------------------ in-place parameter gradients
@pytest.mark.parametrize("which", ["evflownet", "ann_firenet", "alif_firenet"])
def test_direct_param_grads_equal_autograd_accumulation(which):
    """With hip_ops.DIRECT_PARAM_GRADS (set by FlatAdam) the backward kernels add weight / bias / neuron-parameter
    gradients straight into an already-bound fp32 .grad and return None to autograd.  Two identical backward
    passes must then give exactly what autograd's own accumulation gives: grad(second pass) = 2 x grad(first)."""
    torch.manual_seed(3)
    if which == "evflownet":
        model = SpikingRecEVFlowNet(_unet_cfg(8)).to(DEV)
        H = W = 32
    elif which == "ann_firenet":
        model = FireNet(_ann_cfg()).to(DEV)
        H = W = 24
    else:
        cfg = _unet_cfg(8)
        cfg["spiking_neuron"] = {"hard_reset": False}
        cfg["base_num_channels"] = 16
        model = ALIFFireNet(cfg).to(DEV)
        H = W = 24
    gen = torch.Generator().manual_seed(5)
    xs = [(torch.rand(2, 2, H, W, generator=gen) < 0.3).float().to(DEV) * 2 for _ in range(2)]

    def run():
        model.reset_states()
        tot = 0
        for x in xs:
            out = model(x, x)
            tot = tot + sum((f * f).sum() for f in out["flow"])
        tot.backward()

    old = hip_ops.DIRECT_PARAM_GRADS
    try:
        hip_ops.DIRECT_PARAM_GRADS = False
        run()
        first = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        assert len(first) > 4
        hip_ops.DIRECT_PARAM_GRADS = True
        run()  # .grad is bound now: the kernels accumulate in place
    finally:
        hip_ops.DIRECT_PARAM_GRADS = old
    for k, p in model.named_parameters():
        if k in first:
            close(N(p.grad), 2 * N(first[k]), 2e-5, k)


# ------------------------------------------------------------------ norm_input
def test_norm_input_matches_reference_formula():
    """models/model.py:247-252: the non-zero entries are standardised with their own mean / unbiased std."""
    gen = torch.Generator().manual_seed(11)
    x = torch.poisson(torch.full((3, 2, 37, 41), 0.7), generator=gen)
    x[:, 1] *= -0.5
    ref = x.clone()
    nz = ref != 0
    ref[nz] = (ref[nz] - ref[nz].mean()) / ref[nz].std()
    xd = G(x.numpy())
    got = hip_ops.norm_nonzero(xd)
    assert torch.equal(xd.cpu(), x)  # out of place
    np.testing.assert_allclose(N(got), ref.numpy(), rtol=2e-6, atol=2e-6)
    # through a model: FireNet with norm_input equals the same model fed the pre-normalised input
    cfg = _unet_cfg(32)
    torch.manual_seed(1)
    m1 = cells_model(cfg, norm=True)
    torch.manual_seed(1)
    m2 = cells_model(cfg, norm=False)
    a = m1(xd, xd)["flow"][0]
    b = m2(got, got)["flow"][0]
    assert torch.equal(a, b)


def cells_model(cfg, norm):
    from event_flow_amd.models.model import LIFFireNet

    c = dict(cfg)
    c["norm_input"] = norm
    return LIFFireNet(c).to(DEV)


@pytest.mark.parametrize("which", ["LIFFireNet", "SpikingRecEVFlowNet"])
def test_kernel_size_5_models_vs_oracle(which):
    """kernel_size = 5 (the default of the reference's UNets, models/unet.py:51; FireNets take it from the config,
    models/model.py:162-173): forward flows, states and every parameter gradient against the CPU oracle."""
    from event_flow_amd.models.model import LIFFireNet

    torch.manual_seed(21)
    C = 32 if which == "LIFFireNet" else 8
    cfg = dict(_unet_cfg(C), kernel_size=5)
    cfg["spiking_neuron"] = dict(cfg["spiking_neuron"], thresh=[0.25, 0.05])
    model = (LIFFireNet if which == "LIFFireNet" else SpikingRecEVFlowNet)(cfg).to(DEV)
    ks = [p.shape[-1] for k, p in model.named_parameters() if k.endswith("weight")]
    assert ks.count(5) >= 7 and set(ks) <= {1, 3, 5}  # (the residual blocks of the UNet are 3x3 whatever the config says)
    params = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    keys = [k for k, _ in model.named_parameters()]
    for k in keys:
        params[k].requires_grad_(True)
    xs = [(torch.rand(2, 2, 32, 32) < 0.3).float() * torch.randint(1, 4, (2, 2, 32, 32)).float() for _ in range(2)]
    states = [None] * (7 if which == "LIFFireNet" else 10)
    tot = tot_ref = 0
    for x in xs:
        if which == "LIFFireNet":
            f_ref, states = osnn.firenet_forward("LIFFireNet", params, x, states)
            flows_ref = [f_ref]
        else:
            flows_ref, states = osnn.spiking_unet_forward("lif", params, x, states)
        out = model(x.to(DEV), x.to(DEV))
        for f, fr in zip(out["flow"], flows_ref):
            close(N(f), fr.detach().numpy(), 1e-4, "flow")
            tot = tot + (f * f).sum()
            tot_ref = tot_ref + (fr * fr).sum()
    assert float(tot_ref) > 0
    tot.backward()
    tot_ref.backward()
    gn = float(np.sqrt(sum(float((params[k].grad.numpy() ** 2).sum()) for k in keys if params[k].grad is not None)))
    err = 0.0
    for k, p in model.named_parameters():
        ref = params[k].grad
        ref = ref.numpy() if ref is not None else np.zeros(tuple(p.shape), np.float32)
        got = N(p.grad) if p.grad is not None else np.zeros_like(ref)
        err += float(((got - ref) ** 2).sum())
    assert np.sqrt(err) <= 2e-3 * gn, np.sqrt(err) / gn


# ------------------------------------------------------------------ BN / IN layers, transposed-conv decoders (G16)
def test_g16_norm_and_transposed_conv_layers():
    """The reference's ANN layers with norm = 'BN' / 'IN', its transposed-conv decoder layer and a MultiResUNet built with
    norm='BN', use_upsample_conv=False (models/submodules.py:12-137, 140-185, 238-311; models/unet.py:196-311): outputs of two
    training-mode calls, input / parameter gradients, the running statistics they leave, and the eval-mode output."""
    from event_flow_amd.models import submodules as sub
    from event_flow_amd.models import unet

    g = load_golden("g16_norm_layers")
    cases = golden_cases(g)
    assert len(cases) == 10
    for c in cases:
        tag, cls, kw = c["tag"], c["cls"], c["kwargs"]
        m = (unet.MultiResUNet(dict(kw)) if cls == "MultiResUNet" else getattr(sub, cls)(**kw)).to(DEV)
        sd = {k[len(tag + "_param0_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + "_param0_")}
        m.load_state_dict(sd)  # the reference's state_dict keys (incl. norm buffers) load unchanged
        m.train()
        xs = [G(g[f"{tag}_x{k}"]).requires_grad_(True) for k in range(2)]
        tot = 0
        for k, x in enumerate(xs):
            y = m(x, None, G(g[f"{tag}_res{k}"]))[0] if cls == "ConvLayer_" else m(x)
            for j, yy in enumerate(y if isinstance(y, (list, tuple)) else [y]):
                close(N(yy), g[f"{tag}_y{k}_{j}"], 2e-5, f"{cls} {kw.get('norm')} y{k}_{j}")
                tot = tot + yy.pow(2).sum() + yy.sum()
        grads = torch.autograd.grad(tot, xs + list(m.parameters()), allow_unused=True)
        for k in range(2):  # (sum y^2 + sum y of an un-affine instance norm is nearly constant: tiny gradients, hence the atol)
            np.testing.assert_allclose(N(grads[k]), g[f"{tag}_gx{k}"], rtol=0, atol=2e-4 * float(np.abs(g[f"{tag}_gx{k}"]).max()) + 1e-5,
                                       err_msg=f"{cls} {kw.get('norm')} gx{k}")
        # (a bias in front of a normalisation has a mathematically zero gradient: round-off on both sides -- every tensor is
        #  held to 3e-4 of its own largest element or 1 % of the largest gradient element of the layer, whichever is larger)
        gmax = max(float(np.abs(g[f"{tag}_grad_{pn}"]).max()) for pn, _ in m.named_parameters())
        for (pn, prm), gr in zip(m.named_parameters(), grads[2:]):
            ref = g[f"{tag}_grad_{pn}"]
            got = N(gr) if gr is not None else np.zeros_like(ref)
            # (1e-3 for the instance norm: its backward projects out most of the upstream gradient, which amplifies round-off)
            tol = 1e-3 if kw.get("norm") == "IN" else 3e-4
            if float(np.abs(ref).max()) < 1e-2 * gmax:  # structurally zero (bias in front of a norm): both sides are summation noise
                assert np.abs(got - ref).max() <= 2e-3 * gmax, (cls, kw.get("norm"), pn, float(np.abs(got).max()))
                continue
            assert np.abs(got - ref).max() <= tol * max(float(np.abs(ref).max()), 1e-2 * gmax), (cls, kw.get("norm"), pn)
        for pn, v in m.state_dict().items():  # running mean / variance / batch counter after the two calls
            ref = g[f"{tag}_param1_{pn}"]
            np.testing.assert_allclose(N(v).astype(np.float64), ref.astype(np.float64), rtol=1e-5, atol=1e-6, err_msg=f"{cls} {pn}")
        m.eval()
        with torch.no_grad():
            ye = m(xs[0], None, G(g[f"{tag}_res0"]))[0] if cls == "ConvLayer_" else m(xs[0])
        for j, yy in enumerate(ye if isinstance(ye, (list, tuple)) else [ye]):
            close(N(yy), g[f"{tag}_yeval_{j}"], 2e-5, f"{cls} {kw.get('norm')} eval {j}")


def test_general_path_window_cycle_replays_without_state_copies():
    """train.capture_window_cycle: the training steps of two windows of a spiking EV-FlowNet as two hipGraphs replayed
    alternately.  The cells of the second graph write their new states straight into the tensors the first graph reads
    (hip_ops.route_states): no state tensor is copied, and the replayed steps give the losses of the same steps launched
    eagerly -- across device synchronizes (no memcpy / memset nodes in the graphs).  Learning rate 0: with the weights fixed
    every loss is a deterministic function of the windows so far THROUGH the recurrent state (the forward has no float
    atomics; the loss's own sum does: 1e-6) -- with a learning rate the atomics' round-off flips spikes and two EAGER runs
    already differ by 1e-3 after four steps."""
    from event_flow_amd import synthetic
    from event_flow_amd.loss.flow import EventWarping
    from event_flow_amd.train import FlatAdam, capture_window_cycle, encode_passes, train_window

    B, n, H, W = 2, 3000, 64, 64
    cfg = {"num_bins": 2, "base_num_channels": 8, "kernel_size": 3, "encoding": "cnt", "norm_input": False, "mask_output": True,
           "activations": ["arctanspike", "arctanspike"],
           "spiking_neuron": {"leak": [-4.0, 0.1], "thresh": [0.3, 0.05], "learn_leak": True, "learn_thresh": True, "hard_reset": True}}
    lc = {"loader": {"resolution": [H, W]}, "loss": {"flow_regul_weight": 0.001, "overwrite_intermediate": False}, "model": {"mask_output": True}}
    pool = [encode_passes([torch.from_numpy(synthetic.event_list_batch(B, n, H, W, 777 + 13 * w)).to(DEV)], 2, (H, W)) for w in range(2)]

    def run(graphed, nsteps=6, warm=2):
        torch.manual_seed(0)
        model = SpikingRecEVFlowNet(dict(cfg)).to(DEV)
        model.train()
        lossf = EventWarping(lc, DEV)
        opt = FlatAdam(model, lr=0.0, clip=100.0, device_step=True)
        opt.zero_grad()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        losses = []
        with torch.cuda.stream(side):
            for i in range(warm):
                losses.append(float(train_window(model, lossf, opt, pool[i % 2])))
            torch.cuda.synchronize()
            copied = None
            if graphed:
                graphs, copied = capture_window_cycle(model, lossf, opt, pool, side, route=graphed != "copy")  # (a capture launches nothing)
                torch.cuda.synchronize()
            for i in range(nsteps):
                if graphed:
                    graphs[i % 2][0].replay()
                    torch.cuda.synchronize()
                    losses.append(float(graphs[i % 2][1]))
                else:
                    losses.append(float(train_window(model, lossf, opt, pool[i % 2])))
        torch.cuda.synchronize()
        return losses, copied

    eager, _ = run(False)
    graphc, copiedc = run("copy")
    graph, copied = run(True)
    print("eager", eager, "\ngraph, states copied", graphc, copiedc, "\ngraph", graph, "state tensors copied", copied)
    assert copied == 0 and copiedc > 0, (copied, copiedc)
    # the state matters: the same window gives another loss every time it comes round
    assert all(np.isfinite(graph)) and len({round(v, 5) for v in eager[0::2]}) > 2, eager
    np.testing.assert_allclose(graphc, eager, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(graph, eager, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["ALIFFireNet", "XLIFFireNet", "FireNet"])
def test_general_path_window_cycle_of_the_firenet_family(name):
    """train.capture_window_cycle also takes the FireNet family where it is chained cell by cell (ALIF / XLIF / ANN: `model._states`):
    two windows of three passes each as two hipGraphs replayed alternately, the recurrent state crossing the replays, against the
    same steps launched eagerly (learning rate 0, as the EV-FlowNet test above)."""
    from event_flow_amd import synthetic
    from event_flow_amd.loss.flow import EventWarping
    from event_flow_amd.models import model as models
    from event_flow_amd.train import FlatAdam, capture_window_cycle, encode_passes, train_window

    B, n, H, W, P = 2, 1500, 32, 32, 3
    cfg = {"num_bins": 2, "base_num_channels": 32, "kernel_size": 3, "encoding": "cnt", "norm_input": False, "mask_output": True,
           "activations": ["relu", None] if name == "FireNet" else ["arctanspike", "arctanspike"]}
    lc = {"loader": {"resolution": [H, W]}, "loss": {"flow_regul_weight": 0.001, "overwrite_intermediate": False}, "model": {"mask_output": True}}
    pool = [encode_passes([torch.from_numpy(synthetic.event_list_batch(B, n, H, W, 31 + 7 * w + 1000 * k)).to(DEV) for k in range(P)], 2, (H, W))
            for w in range(2)]

    def run(graphed, nsteps=4, warm=2):
        torch.manual_seed(0)
        model = getattr(models, name)(dict(cfg)).to(DEV)
        model.train()
        assert model.compute_path[0] == ("general")
        lossf = EventWarping(lc, DEV)
        opt = FlatAdam(model, lr=0.0, clip=100.0, device_step=True)
        opt.zero_grad()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        losses = []
        with torch.cuda.stream(side):
            for i in range(warm):
                losses.append(float(train_window(model, lossf, opt, pool[i % 2])))
            torch.cuda.synchronize()
            if graphed:
                graphs, _ = capture_window_cycle(model, lossf, opt, pool, side)
                torch.cuda.synchronize()
            for i in range(nsteps):
                if graphed:
                    graphs[i % 2][0].replay()
                    torch.cuda.synchronize()
                    losses.append(float(graphs[i % 2][1]))
                else:
                    losses.append(float(train_window(model, lossf, opt, pool[i % 2])))
        torch.cuda.synchronize()
        opt.close()
        return losses

    eager, graph = run(False), run(True)
    print(name, "eager", eager, "\ngraph", graph)
    assert all(np.isfinite(graph)) and len({round(v, 5) for v in eager[0::2]}) > 1, eager  # (the state matters)
    np.testing.assert_allclose(graph, eager, rtol=1e-5, atol=1e-6)


def test_general_path_one_window_cycle_keeps_the_previous_state_for_the_backward():
    """A cycle of ONE window starts from the very tensors it has to end in: its cells must not be routed onto themselves (the
    neuron kernels would overwrite the previous state the backward pass and the recurrent weight gradient still read).  With a
    learning rate > 0 the replayed step has to make the update of the same step launched eagerly from the same state."""
    from event_flow_amd import synthetic
    from event_flow_amd.loss.flow import EventWarping
    from event_flow_amd.models import hip_ops
    from event_flow_amd.train import FlatAdam, _general_states, capture_window_cycle, encode_passes, train_window

    B, n, H, W = 2, 3000, 64, 64
    cfg = {"num_bins": 2, "base_num_channels": 8, "kernel_size": 3, "encoding": "cnt", "norm_input": False, "mask_output": True,
           "activations": ["arctanspike", "arctanspike"],
           "spiking_neuron": {"leak": [-4.0, 0.1], "thresh": [0.3, 0.05], "learn_leak": True, "learn_thresh": True, "hard_reset": True}}
    lc = {"loader": {"resolution": [H, W]}, "loss": {"flow_regul_weight": 0.001, "overwrite_intermediate": False}, "model": {"mask_output": True}}
    win = encode_passes([torch.from_numpy(synthetic.event_list_batch(B, n, H, W, 4242)).to(DEV)], 2, (H, W))

    # route_states itself refuses a pair that shares memory
    t = torch.zeros(1, 4, 8, 8, device=DEV)
    hip_ops.route_states([t, t[:, :2]], [t, t])
    try:
        assert hip_ops._routed(t, (1, 8, 8, 4)) is None and hip_ops._routed(t[:, :2], (1, 8, 8, 2)) is None
    finally:
        hip_ops.clear_state_routes()

    torch.manual_seed(0)
    model = SpikingRecEVFlowNet(dict(cfg)).to(DEV)
    model.train()
    lossf = EventWarping(lc, DEV)
    opt = FlatAdam(model, lr=1e-3, clip=100.0, device_step=True)
    opt.zero_grad()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            train_window(model, lossf, opt, win)
        torch.cuda.synchronize()
        home = _general_states(model)[1]
        snap = ([h.clone() for h in home], opt.flat_param.clone(), opt.m.clone(), opt.v.clone(), opt.norm_ws.clone())

        def restore():
            for h, s in zip(home, snap[0]):
                h.copy_(s)
            opt.flat_param.copy_(snap[1]); opt.m.copy_(snap[2]); opt.v.copy_(snap[3]); opt.norm_ws.copy_(snap[4])
            opt.step_invalidate()
            hip_ops.repack_all()

        l_eager = float(train_window(model, lossf, opt, win))
        p_eager, s_eager = opt.flat_param.clone(), [s.clone() for s in _general_states(model)[1]]
        torch.cuda.synchronize()
        # back to the snapshot, with the model's states being the `home` tensors again
        holder = _general_states(model)[0]
        restore()
        k = 0
        st = []
        for s in holder.states:
            if s is None:
                st.append(None)
            elif isinstance(s, (tuple, list)):
                st.append(type(s)(home[k : k + len(s)])); k += len(s)
            else:
                st.append(home[k]); k += 1
        holder.states = st
        torch.cuda.synchronize()
        graphs, copied = capture_window_cycle(model, lossf, opt, [win], side)
        torch.cuda.synchronize()
        restore()
        torch.cuda.synchronize()
        graphs[0][0].replay()
        torch.cuda.synchronize()
        l_graph = float(graphs[0][1])
        p_graph, s_graph = opt.flat_param.clone(), [h.clone() for h in home]
        graphs[0][0].replay()  # a second replay starts from the state the first one left in `home`
        torch.cuda.synchronize()
        assert np.isfinite(float(graphs[0][1]))
    assert copied == len(home), (copied, len(home))
    upd_e, upd_g = N(p_eager - snap[1]), N(p_graph - snap[1])
    print("one-window cycle: loss", l_eager, l_graph, "update rel-L2", float(np.linalg.norm(upd_g - upd_e) / np.linalg.norm(upd_e)))
    assert abs(l_graph - l_eager) <= 1e-5 * abs(l_eager) + 1e-6
    assert np.linalg.norm(upd_e) > 0 and np.linalg.norm(upd_g - upd_e) <= 1e-3 * np.linalg.norm(upd_e)
    for a, b in zip(s_eager, s_graph):  # the forward is deterministic: the new state is the eager step's bit for bit
        assert torch.equal(a, b)


@pytest.mark.parametrize("passes", [1, 2])
def test_cell_output_twins_give_the_gradients_of_autograd_accumulation(passes, monkeypatch):
    """hip_ops.FORK_TWIN (EVF_FORK_TWIN=1, opt-in): a cell output with two consumers hands the second one a twin view and the
    producing cell's backward receives the two gradients as two arguments (added inside the neuron kernel, or by one add when
    the state gradient holds that slot -- windows of several passes -- or the sum is the residual's gradient).  Same loss and
    the same parameter gradients (round-off of a + b in another place) as with autograd's own accumulation, on a spiking
    EV-FlowNet: encoders -> skip + next encoder, decoders -> prediction + next decoder, residual block inputs."""
    from event_flow_amd import synthetic
    from event_flow_amd.loss.flow import EventWarping
    from event_flow_amd.models import hip_ops
    from event_flow_amd.train import encode_passes

    B, n, H, W = 2, 3000, 64, 64
    cfg = {"num_bins": 2, "base_num_channels": 8, "kernel_size": 3, "encoding": "cnt", "norm_input": False, "mask_output": True,
           "activations": ["arctanspike", "arctanspike"],
           "spiking_neuron": {"leak": [-4.0, 0.1], "thresh": [0.3, 0.05], "learn_leak": True, "learn_thresh": True, "hard_reset": True}}
    lc = {"loader": {"resolution": [H, W]}, "loss": {"flow_regul_weight": 0.001, "overwrite_intermediate": False}, "model": {"mask_output": True}}
    win = encode_passes([torch.from_numpy(synthetic.event_list_batch(B, n, H, W, 900 + k)).to(DEV) for k in range(passes)], 2, (H, W))

    def run(twins):
        monkeypatch.setattr(hip_ops, "FORK_TWIN", twins)
        torch.manual_seed(0)
        model = SpikingRecEVFlowNet(dict(cfg)).to(DEV)
        model.train()
        lossf = EventWarping(lc, DEV)
        for d in win:
            out = model(d["event_voxel"], d["event_cnt"])
            lossf.event_flow_association(out["flow"], d["event_list"], d["event_list_pol_mask"], d["event_mask"])
        loss = lossf()
        loss.backward()
        torch.cuda.synchronize()
        return float(loss), {k: N(p.grad).copy() for k, p in model.named_parameters() if p.grad is not None}

    l0, g0 = run(False)
    l1, g1 = run(True)
    assert abs(l0 - l1) <= 1e-5 * abs(l0) and set(g0) == set(g1) and len(g0) > 20  # (the loss's own sum uses float atomics: 1e-6)
    num = np.sqrt(sum(float(((g1[k] - g0[k]) ** 2).sum()) for k in g0))
    den = np.sqrt(sum(float((g0[k] ** 2).sum()) for k in g0))
    print(f"[twins, {passes} pass(es)] loss {l0:.6f}, gradient rel-L2 between the two forms {num / den:.2e}")
    assert den > 0 and num <= 1e-4 * den  # (a gradient routed wrongly is O(1) off; the loss backward's float atomics: 1e-6)
