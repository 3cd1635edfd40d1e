"""BPTT engine for the FireNet family on the MI355X.

One forward pass of the network (reference models/model.py:255-265) is a single
autograd node that sequences the HIP kernels of libevflow_hip.so:

    forward   head: evf_head_lif_fwd, 6x evf_conv_lif_fwd, evf_pred_fwd     (8 launches)
    backward  evf_pred_bwd, per layer evf_lif_bwd + evf_conv_wgrad_bits /
              evf_head_wgrad + evf_conv_dgrad                                (<= 27 launches)

State (membrane potential [B,H,W,32] fp32, spikes bit-packed [B,H,W] int32)
and the saved activations live in a tape owned by the engine; PyTorch autograd
only sees the flow map and a scalar token that chains consecutive passes, so
that `loss.backward()` walks the passes in reverse time order (truncated BPTT,
train_flow.py:141-171) while state gradients are carried between the nodes in
engine buffers instead of materialised autograd edges.  Parameter gradients
are accumulated over the window and delivered by the node of the window's
first pass.
"""

import ctypes
import os

import torch

from .. import _lib
from . import hip_ops
from .spiking_util import SURROGATE_ID

C = 32  # channels of the accelerated kernels (base_num_channels)


def _i32(shape, dev):
    return torch.empty(shape, dtype=torch.int32, device=dev)


def _f32(shape, dev):
    return torch.empty(shape, dtype=torch.float32, device=dev)


# the input-gradient kernel takes the fp32 g_cur and splits it while staging (128 instead of 192 B/pixel written and read)
F32_DGRAD = os.environ.get("EVF_F32_DGRAD", "1") != "0"
# under the diagonal (recorded) backward the fused backward writes the exact bf16 split of g_cur (three planes, 192 B/pixel) instead
# of the fp32 tensor, and the persistent input-gradient launch stages it by LDS-DMA (k_dgrad_diag_dma); 0: fp32 g_cur, split by
# the input-gradient kernel's producer waves (k_dgrad_diag_ws)
SPLIT_DGRAD = os.environ.get("EVF_DGRAD_SPLIT", "1") != "0"
# the head layer's cells of a recorded window in ONE launch each way (k_head_lif_fwd_win, k_head_bwd_win): its state is per pixel
HEAD_WIN = os.environ.get("EVF_HEAD_WIN", "1") != "0"
PRED_FUSED = os.environ.get("EVF_PRED_FUSED", "1") != "0"  # prediction head in the epilogue of the last layer's forward
TOP_FUSED = os.environ.get("EVF_TOP_FUSED", "1") != "0"  # prediction-head backward inside the top layer's fused backward
PAIR_DGRAD = os.environ.get("EVF_PAIR_DGRAD", "1") != "0"
# PLIF: AvgPool3x3^T / 32 of dL/d(pooled activity) inside the input-gradient kernels (0: a k_plif_box launch per cell; A/B, tests)
PLIF_BOX_IN_DGRAD = os.environ.get("EVF_PLIF_BOX", "dgrad") != "kernel"
# PLIF hidden cells: the trace backward inside the fused backward's streaming team (evf_plif_bwd_wgrad2 / _top); 0: evf_plif_trace_bwd
# as a pass of its own behind evf_lif_bwd_wgrad2 (A/B, tests)
PLIF_TRACE_FUSED = os.environ.get("EVF_PLIF_TRACE_FUSED", "1") != "0"
# PLIF windows backward LAYER by layer: a feed-forward hidden layer's passes in ONE launch with dL/dv, dL/d(pt) and the potential in
# registers (evf_plif_bwd_wgrad_window: 640 instead of 1152 bytes per pixel and pass); recurrent layers and the layer under the
# prediction head pass by pass, the head layer's window last.  0: pass by pass (every cell a launch)
PLIF_LAYER_MAJOR = os.environ.get("EVF_PLIF_LAYER_MAJOR", "1") != "0"
# ... its input gradients: "ws" (default) = one k_conv_dgrad_ws launch per product from the fp32 dL/d(current); "dma" = from the
# pre-split planes through k_dgrad_diag_dma, a layer's passes (a recurrent cell's two products) per launch (evf_conv_dgrad_b3_multi)
# -- measured SLOWER for PLIF (12.4 against 11.3 ms per step): the trace term's ten loads per pixel sit in the matrix waves' issue
# stream there (59 against 47 us per product; without the term the kernel runs 43)
PLIF_LM_DGRAD = os.environ.get("EVF_PLIF_LM_DGRAD", "ws")
# LIF windows: the feed-forward layers ABOVE the last recurrent one (R2b under the prediction head, R2a) know their dL/d(spikes) of
# every pass before anything below them has run: their backward of the whole window runs first, one launch per layer with dL/dv and
# the potential in registers (evf_lif_bwd_wgrad_window: 17-19 us per cell at 8 x 128 x 128 against ~21 inside a diagonal launch of
# four), their input gradients as one launch per layer (evf_conv_dgrad_b3_multi); the layers below stay on the recorded diagonals.
# 0: every hidden layer on the diagonals
LIF_BWD_TOP = os.environ.get("EVF_LIF_BWD_TOP", "1") != "0"
# recorded FORWARD of a window LAYER by layer: a feed-forward hidden layer's passes are recorded under ONE index and launched as a
# chain (k_fwd_win_t: potential, trace and previous spikes in registers across the passes, the tape is written only); recurrent
# layers one pass per index.  "1": every hidden layer; "top": only the feed-forward layers above the last recurrent one (the
# layers below stay on the diagonals); "0": diagonals (cell (t, l) under index t + l - 1); "auto": "1" when ONE cell's rounds of
# strips fill the chip (>= FWD_LM_MIN_QUADS rounds of four strips: 4 x 260 x 346 has 1430, and its forward 2.35 -> 2.29 ms eager,
# 11.48 -> 11.20 ms per replayed step), else "top" (8 x 128 x 128: 512 rounds; single recurrent cells would not fill the chip there,
# 3.45 / 3.43 / 3.43 ms on diagonals, 3.42 / 3.40 / 3.41 with "top", 3.45 / 3.45 / 3.43 with "1" on one box)
FWD_LAYER_MAJOR = os.environ.get("EVF_FWD_LM", "auto")
FWD_LM_MIN_QUADS = int(os.environ.get("EVF_FWD_LM_MIN_QUADS", "1024"))
FWD_LM_PASSES = 16  # passes a recording holds in the layer-major forms (= FW_WIN_MAX, csrc/evf_fwd.h)
# window gradients -> the flat gradient buffer in one launch (evf_grads_finalize); 0: row sums, slab reduction, segment add one by one
FUSED_TAIL = os.environ.get("EVF_FUSED_TAIL", "1") != "0"
PARAM_ROWS = os.environ.get("EVF_PARAM_ROWS", "1") != "0"  # per-channel gradients through per-block rows (0: atomics)  # ff + rec input gradients of a recurrent cell in one launch


class _Window:
    """Gradient carries and parameter-gradient accumulators of one BPTT window."""

    def __init__(self, eng, B, H, W, dev):
        n = len(eng.cells)
        self.shape = (B, H, W)
        self.dev = dev
        self.gv = [None] * n  # dL/dv carried to the previous pass, per layer
        self.gz = [None] * n  # dL/d(output spikes) of the pass being processed
        self.gz_has = [False] * n
        self.g_cur = None
        self.g_split = None  # [3,B,H,W,32] bf16: exact 3-way split of g_cur (bf16x3 path)
        self.gpt = [None] * n  # PLIF: dL/d(trace) carried to the previous pass
        self.gpt_has = [False] * n
        self.gP = None  # PLIF: dL/d(pooled pre-synaptic activity) of the layer being processed [B,H,W]
        # small-parameter gradient accumulator: the engine's persistent buffer when FlatAdam owns the gradients (cleared by
        # the kernel that consumes it, _finalize), else fresh zeros
        self.small, self.small_persistent = eng._take_small(dev)
        # per-block partial sums of the per-channel gradients [blocks][small_size] (summed into `small` by _finalize)
        self.rows = eng._take_rows(B, H, W, dev)
        n_ = len(eng.cells)
        self.gzr, self.gzr_has = [None] * n_, [False] * n_  # recurrent part of dL/d(spikes), separate from gz (see _backward_pass)
        self.gcl = [None] * n_  # per-layer g_cur buffers (diagonal backward launches: several layers in flight)
        self.gsl = [None] * n_  # ... or per-layer split planes [3,B,H,W,32] bf16 (SPLIT_DGRAD)
        self.gz0 = []           # dL/d(spikes) of the head layer, one buffer per backward pass (HEAD_WIN)
        self.bwd_k = 0          # backward passes of this window so far
        self.lm = []            # PLIF_LAYER_MAJOR: (tape, dL/dflow, is_first) of the passes whose backward waits for the window's first
        self.slab_init = {}
        self.token = eng._token(dev)  # (a leaf whose value is never read: only its autograd edge chains the passes)
        self.n_passes = 0
        # flow maps of the window's passes in ONE buffer [cap,B,2,H,W]: the loss reads them in place (no torch.stack)
        self.flow_buf = _f32((eng._flow_cap, B, 2, H, W), dev)

    def buf(self, lst, l):
        if lst[l] is None:
            B, H, W = self.shape
            lst[l] = _f32((B, H, W, C), self.dev)
        return lst[l]


class _FireNetPass(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, win, is_first, x_in, token, *params):
        ctx.set_materialize_grads(False)
        flow, tape, new_states = eng._forward_pass(x_in, eng._states, record=True)
        eng._states = new_states
        ctx.eng, ctx.win, ctx.tape, ctx.is_first = eng, win, tape, is_first
        new_token = torch.empty((), dtype=torch.float32, device=flow.device)  # only its autograd edge is used
        return flow, new_token

    @staticmethod
    def backward(ctx, g_flow, g_token):
        eng, win = ctx.eng, ctx.win
        eng.flush_forward()
        if eng._lm_wanted(win, ctx.tape, g_flow):
            # layer-major backward of the window: nothing runs until the window's first pass has handed in its dL/dflow
            win.lm.append((ctx.tape, g_flow, ctx.is_first))
            if ctx.is_first:
                eng._backward_window_lm(win)
        elif eng._top_wanted(win, ctx.tape, g_flow):
            # LIF: the layers above the last recurrent one as window launches ahead of the diagonals (same stash)
            win.lm.append((ctx.tape, g_flow, ctx.is_first))
            if ctx.is_first:
                eng._backward_window_top(win)
        else:
            eng._lm_replay(win)  # (passes that waited for a layer-major backward this pass cannot join: pass by pass, in order)
            eng._backward_pass(win, ctx.tape, g_flow, ctx.is_first)
            win.bwd_k += 1
        if ctx.is_first:
            eng.flush_backward()  # (the recorded cells of all passes, diagonal by diagonal)
        ctx.tape = None  # (while a backward recording is open the engine holds the tape: _backward_pass, flush_backward)
        grads = eng._finalize(win) if ctx.is_first else (None,) * len(eng.params)
        return (None, None, None, None, g_token) + tuple(grads)


class FireNetEngine:
    """Sequences the kernels for head -> G1 -> R1a -> R1b -> G2 -> R2a -> R2b -> pred."""

    def __init__(self, cells, pred, num_bins, precision="bf16x3"):
        # "bf16x3": forward convs on the bf16 matrix cores with the exact 3-way weight split
        # (fp32-equivalent numerics); "fp32": v_mfma_f32_32x32x2_f32 everywhere
        if precision not in ("bf16x3", "fp32"):
            raise ValueError(precision)
        self.precision = precision
        self.cells = cells  # list of 7 spiking cell modules
        self.pred = pred
        self.num_bins = num_bins
        self.kind = cells[0].kind
        # XLIF cells (reference spiking_submodules.py:337-435, :771-875) run through the PLIF kernels: the same pre-synaptic trace, which
        # raises the THRESHOLD (t0 + t1 * pt') instead of being subtracted from the current.  Their t1 travels in the `add_pt` slot,
        # t0 in the `thresh` slot, and bit 1 of the entry points' reset / accumulate flag says so (include/evflow.h).
        # ALIF cells (:230-334, :660-768): the XLIF arithmetic with the threshold trace driven by the cell's OWN previous spikes (un-detached:
        # one more part of dL/d(spikes) of the pass before): mode 2 of the same entry points (bits 1-2 of the flag)
        self._plif = self.kind in ("plif", "xlif", "alif")
        self._xl = self.kind == "xlif"
        self._al = self.kind == "alif"
        self._xf = 2 if self._xl else (4 if self._al else 0)
        for i, c in enumerate(cells):
            if c.kind not in ("lif", "plif", "xlif", "alif") or c.kind != self.kind:
                raise NotImplementedError(f"{type(c).__name__}: LIF, PLIF, XLIF and ALIF cells are accelerated; there is no CPU fallback")
            if self._plif and precision != "bf16x3":
                raise NotImplementedError("PLIF / XLIF / ALIF cells are implemented on the bf16x3 path only")
            if c.kind in ("xlif", "alif") and not (c.hard_reset and c.activation == "arctanspike" and PLIF_TRACE_FUSED):
                raise NotImplementedError("fused XLIF / ALIF cells: hard reset, arctan surrogate, trace backward inside the fused backward")
            if c.hidden_size != C or c.kernel_size != 3 or c.stride != 1 or (i > 0 and c.input_size != C):
                raise NotImplementedError("accelerated FireNet kernels need base_num_channels=32, kernel_size=3")
            if i == 0 and c.recurrent:
                raise NotImplementedError("recurrent head cell")
        if cells[0].input_size > 8:
            raise NotImplementedError("head cell supports at most 8 input channels")
        # flat list of the tensors the autograd node depends on, with their owners
        self.params, self.pnames = [], []
        for i, c in enumerate(cells):
            self._reg(f"{i}.ff", c.ff.weight)
            if c.recurrent:
                self._reg(f"{i}.rec", c.rec.weight)
            if c.kind == "plif":
                self._reg(f"{i}.leak", c.leak_v)
                self._reg(f"{i}.leak_pt", c.leak_pt)
                self._reg(f"{i}.add_pt", c.add_pt)
            elif c.kind in ("xlif", "alif"):
                self._reg(f"{i}.leak", c.leak_v)
                self._reg(f"{i}.leak_pt", c.leak_pt if c.kind == "xlif" else c.leak_t)  # (the trace's leak: leak_t of an ALIF cell)
                self._reg(f"{i}.add_pt", c.t1)  # (the kernels' slot of the trace's weight: t1 here)
            else:
                self._reg(f"{i}.leak", c.leak)
            self._reg(f"{i}.thresh", c.t0 if c.kind in ("xlif", "alif") else c.thresh)
        self._reg("pred.w", pred.conv2d.weight)
        self._reg("pred.b", pred.conv2d.bias)
        # layout of the small-accumulator buffer
        off = 0
        self.small_off = {}
        for name, p in zip(self.pnames, self.params):
            if name.endswith(".ff") and not name.startswith("0.") or name.endswith(".rec"):
                continue  # 32x32x3x3 gradients come from the slab reduction
            self.small_off[name] = (off, p.numel())
            off += p.numel()
        self.small_size = off
        self._states = [None] * len(cells)
        self.static_states = False
        self._final_target = None  # state tensors the last pass of the window writes into (model.final_states_into)
        self._final_hint = False
        self._static = [None] * len(cells)
        self._win = None
        self._packed = {}
        self._packed_key = None
        self._slabs = {}
        self._small_buf = None  # persistent small-gradient accumulator (+ is it all zeros?)
        self._small_clean = False
        self._token0 = None
        self._flow_cap = 16  # passes per window the flow buffer is sized for (grows with the windows seen)
        self._flow_slot = None

    def _reg(self, name, t):
        self.params.append(t)
        self.pnames.append(name)

    def _take_small(self, dev):
        """-> (zeroed accumulator, persistent?).  Persistent only when every small parameter's gradient goes straight
        into FlatAdam's buffer through evf_add_segments, which clears what it consumes."""
        ok = hip_ops.DIRECT_PARAM_GRADS and all(
            p.requires_grad and p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous() and p.grad.is_cuda
            for name, p in zip(self.pnames, self.params) if name in self.small_off)
        if not ok:
            return _lib.zeros(self.small_size, dtype=torch.float32, device=dev), False
        if self._small_buf is None or self._small_buf.device != dev:
            self._small_buf = _lib.zeros(self.small_size, dtype=torch.float32, device=dev)
        elif not self._small_clean:  # (a window that never reached _finalize)
            _lib.zero_(self._small_buf)
        self._small_clean = False
        return self._small_buf, True

    def _take_rows(self, B, H, W, dev):
        """Persistent [blocks][small_size] buffer of per-block parameter-gradient partials (zero; evf_sum_rows hands it
        back zeroed).  The fused backward kernels add into their own row instead of 256-512 blocks adding atomically into
        the same 64 words."""
        if self.precision != "bf16x3" or not PARAM_ROWS:
            return None
        L = _lib.load()
        n = max(L.evf_lif_bwd_wgrad_slabs(B, H, W), L.evf_head_lif_bwd_wgrad_slabs(B, H, W), 512)
        buf = self.__dict__.get("_rows_buf")
        if buf is None or buf.device != dev or buf.shape[0] < n or not self.__dict__.get("_rows_clean", False):
            buf = _lib.zeros((n, self.small_size), dtype=torch.float32, device=dev)
            self._rows_buf = buf
        self._rows_clean = False
        return buf

    def _token(self, dev):
        if self._token0 is None or self._token0.device != dev:
            self._token0 = torch.zeros((), dtype=torch.float32, device=dev, requires_grad=True)  # (created once, outside any capture)
        return self._token0

    def _act_width(self, i):
        """Surrogate width as a host float; read back once per buffer version (a
        .item() in the hot loop would stall the launch queue on every call)."""
        t = self.cells[i].act_width
        key = (t.data_ptr(), t._version)
        cache = self.__dict__.setdefault("_aw_cache", {})
        if cache.get(i, (None, None))[0] != key:
            cache[i] = (key, float(t))
        return cache[i][1]

    # ------------------------------------------------------------------ state
    def reset_states(self):
        self._states = [None] * len(self.cells)
        self._win = None

    def detach_states(self):
        self._win = None  # next pass opens a new window; tensors in _states carry no graph
        if self.static_states:
            self._store_static()

    def _store_static(self):
        """Copy the current states into persistent buffers (fixed addresses), so a step
        captured in a hipGraph carries its final state into the next replay."""
        new = []
        for i, st in enumerate(self._states):
            if st is None:
                new.append(None)
                continue
            if self._static[i] is None:
                self._static[i] = tuple(torch.empty_like(t) for t in st)
            if st[0].data_ptr() != self._static[i][0].data_ptr():
                for dst, src in zip(self._static[i], st):
                    dst.copy_(src)
            new.append(self._static[i])
        self._states = new

    def get_states(self):
        """-> list of [2,B,C,H,W] float tensors (or None), the reference's layout."""
        self.flush_forward()  # (cells recorded for a diagonal launch write the states read here)
        out = []
        for st in self._states:
            if st is None:
                out.append(None)
                continue
            v, z = st[0], st[1]
            B, H, W, _ = v.shape
            vv, zz = _f32((B, C, H, W), v.device), _f32((B, C, H, W), v.device)
            _lib.call("evf_nhwc_to_nchw", _lib.ptr(v), B, C, H, W, _lib.ptr(vv))
            _lib.call("evf_bits_to_nchw", _lib.ptr(z), B, H, W, _lib.ptr(zz))
            parts = [vv, zz]
            if len(st) > 3:  # PLIF trace
                pp = _f32((B, C, H, W), v.device)
                _lib.call("evf_nhwc_to_nchw", _lib.ptr(st[3]), B, C, H, W, _lib.ptr(pp))
                parts.append(pp)
            out.append(torch.stack(parts))
        return out

    def set_states(self, states):
        new = []
        for st in states:
            if st is None:
                new.append(None)
                continue
            vv, zz = st[0].detach().float().contiguous(), st[1].detach().float().contiguous()
            B, _, H, W = vv.shape
            v, z = _f32((B, H, W, C), vv.device), _i32((B, H, W), vv.device)
            _lib.call("evf_nchw_to_nhwc", _lib.ptr(vv), B, C, H, W, _lib.ptr(v))
            _lib.call("evf_nchw_to_bits", _lib.ptr(zz), B, H, W, _lib.ptr(z))
            zT = _i32((B, H, C, (W + 31) // 32), vv.device)
            _lib.call("evf_bits_transpose", _lib.ptr(z), B, H, W, _lib.ptr(zT))
            if self._plif:
                pp = st[2].detach().float().contiguous()
                pt = _f32((B, H, W, C), vv.device)
                _lib.call("evf_nchw_to_nhwc", _lib.ptr(pp), B, C, H, W, _lib.ptr(pt))
                new.append((v, z, zT, pt))
            else:
                new.append((v, z, zT))
        self._states = new
        self._win = None

    # ------------------------------------------------------------------ weights
    def _prepare(self, dev):
        key = tuple((p.data_ptr(), p._version) for p in self.params)
        if key == self._packed_key:
            return
        b3_w, b3_f, b3_t = [], [], []
        for i, c in enumerate(self.cells):
            if i == 0:
                continue
            for nm, w in (("ff", c.ff.weight),) + ((("rec", c.rec.weight),) if c.recurrent else ()):
                wd = w.detach().float().contiguous()
                for tr in (0, 1) if self.precision == "fp32" else ():
                    k = (i, nm, tr)
                    if k not in self._packed:
                        self._packed[k] = _f32((9 * C * C,), dev)
                    _lib.call("evf_pack_conv_weight", _lib.ptr(wd), C, C, tr, _lib.ptr(self._packed[k]))
                for fmt in ("b3", "b3t"):
                    k = (i, nm, fmt)
                    if k not in self._packed:
                        self._packed[k] = torch.empty(54 * 1024, dtype=torch.uint8, device=dev)
                b3_w.append(wd)
                b3_f.append(self._packed[(i, nm, "b3")])
                b3_t.append(self._packed[(i, nm, "b3t")])
        for lo in range(0, len(b3_w), 16):  # both split layouts of all conv weights: one launch per 16 tensors
            n = min(16, len(b3_w) - lo)
            arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts[lo:lo + n]])  # noqa: E731
            _lib.call("evf_pack_conv_weights_b3_multi", arr(b3_w), arr(b3_f), arr(b3_t), n)
        self._flat = {}
        for name, p in zip(self.pnames, self.params):
            self._flat[name] = p.detach().float().contiguous().view(-1)
        self._packed_key = key

    # ------------------------------------------------------------------ forward
    def forward(self, x_in):
        _lib.require_gpu(x_in, "FireNet.forward")
        x_in = x_in.detach().float().contiguous()
        self._prepare(x_in.device)
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.params)
        if not needs_grad:
            flow, _, self._states = self._forward_pass(x_in, self._states, record=False)
            return flow
        B, _, H, W = x_in.shape
        is_first = self._win is None
        if is_first:
            self._win = _Window(self, B, H, W, x_in.device)
        win = self._win
        self._flow_slot = win.flow_buf[win.n_passes] if win.n_passes < win.flow_buf.shape[0] else None
        flow, token = _FireNetPass.apply(self, win, is_first, x_in, win.token, *self.params)
        self._flow_slot = None
        win.token = token
        win.n_passes += 1
        self._flow_cap = max(self._flow_cap, win.n_passes)
        return flow

    # -- deferred forward: the window's hidden cells launched diagonal by diagonal (csrc/evf_fwd_b3.hip, k_fwd_diag) --------
    def defer_forward(self, on=True):
        """While on, the hidden layers of every recorded (training) pass are RECORDED (cell (t, l) under index t + l - 1)
        and launched by flush_forward(): P + 5 launches of up to 6 independent cells instead of 6 P.  The head layer of a
        pass still launches at once (it needs only its own previous state).  Anything that reads the cells' outputs through
        this library flushes first (_lib.call); code that reads them through torch must call flush_forward() itself."""
        if not on:
            self.flush_forward()
        self._defer_on = bool(on)

    def flush_forward(self):
        if self.__dict__.get("_defer_open"):
            self._defer_open = False
            with torch.cuda.stream(self._defer_stream):  # the recording belongs to the stream it was opened on
                _lib.clear_defer_hook("fwd")
                _lib.call("evf_fwd_defer_flush")

    def defer_backward(self, on=True):
        """While on, the backward cells of the window (fused backward, input gradient, head backward of every pass) are
        recorded under the index 2 * (backward pass number) + step and launched index by index when the window's first
        pass has been recorded (csrc/evf_bwd_fused.hip, evf_bwd_defer_*): 15 + 15 + P launches instead of 13 P."""
        if not on:
            self.flush_backward()
        self._bdefer_on = bool(on)

    def flush_backward(self):
        if self.__dict__.get("_bdefer_open"):
            self._bdefer_open = False
            with torch.cuda.stream(self._bdefer_stream):  # the recording belongs to the stream it was opened on
                _lib.clear_defer_hook("bwd")
                _lib.call("evf_bwd_defer_flush")
        # the recorded cells hold raw pointers into the passes' tapes and upstream gradients: those tensors are kept here
        # until the cells have been launched (stream order then protects the memory like any other tensor's).  A flush in the
        # MIDDLE of a backward pass (index overflow in _bdefer_slot, the safety net of _lib.call) must not drop the entry of the
        # pass in progress: the cells it records after the flush point into the same tape.
        cur = self.__dict__.get("_bdefer_cur")
        self._bdefer_keep = [cur] if cur is not None else []

    def _bdefer_slot(self, win, step):
        """Index of step `step` (0 = top layer's fused backward, 1 = its input gradient, ...) of the current backward pass."""
        if not self.__dict__.get("_bdefer_open"):
            if _lib.raw("evf_bwd_defer_begin") != 0:
                raise _lib.EvflowError("evf_bwd_defer_begin: another backward recording is open on this stream (one engine per stream)")
            self._bdefer_open, self._bdefer_base = True, win.bwd_k
            self._bdefer_stream = torch.cuda.current_stream()
            _lib.set_defer_hook(self.flush_backward, _lib._DEFER_SAFE_BWD, "bwd")
        d = 2 * (win.bwd_k - self._bdefer_base) + step
        if d >= 96:  # (more than ~40 passes: launch what is recorded, start over)
            self.flush_backward()
            return self._bdefer_slot(win, step)
        if _lib.raw("evf_bwd_defer_slot", d) != 0:
            raise _lib.EvflowError("evf_bwd_defer_slot failed")

    def _defer_begin(self):
        rc = _lib.raw("evf_fwd_defer_begin")
        if rc != 0:
            raise _lib.EvflowError("evf_fwd_defer_begin: another forward recording is open on this stream (one engine per stream)")
        self._defer_open, self._defer_t = True, 0
        self._defer_stream = torch.cuda.current_stream()
        _lib.set_defer_hook(self.flush_forward, _lib._DEFER_SAFE_FWD, "fwd")

    def _fwd_mode(self, B, H, W):
        """The recorded forward's schedule at this shape: "0" diagonals, "1" layer by layer, "top" (FWD_LAYER_MAJOR)."""
        mode = FWD_LAYER_MAJOR
        if mode == "auto":
            quads = (B * ((H + 1) // 2) * ((W + 31) // 32) + 3) // 4
            mode = "1" if quads >= FWD_LM_MIN_QUADS else "top"
        if mode == "top" and not any(not c.recurrent for c in self.cells[1 + max([i for i, c in enumerate(self.cells) if c.recurrent], default=0):]):
            mode = "0"  # (no feed-forward layer above the last recurrent one)
        return mode if mode in ("1", "top") else "0"

    def _fwd_slots(self, B, H, W):
        """Index plan of a recorded forward: None = diagonals (cell (t, l) under t + l - 1), else per layer (base, stride):
        cell (t, l) is recorded under base + stride * t -- stride 0 = the layer's passes under one index (a chain, launched as one
        k_fwd_win_t), stride 1 = a pass per index."""
        mode = self._fwd_mode(B, H, W)
        if mode not in ("1", "top"):
            return None
        key = (mode, tuple(c.recurrent for c in self.cells))
        cache = self.__dict__.setdefault("_fwd_slot_cache", {})
        if key not in cache:
            n = len(self.cells)
            last_rec = max([i for i, c in enumerate(self.cells) if c.recurrent], default=0)
            plan, cur = [None] * n, 0
            if mode == "top":  # layers up to the last recurrent one on diagonals, the rest as chains behind them
                for i in range(1, last_rec + 1):
                    plan[i] = (i - 1, 1)
                cur = FWD_LM_PASSES + max(last_rec - 1, 0)
                for i in range(last_rec + 1, n):
                    plan[i] = (cur, 0)
                    cur += 1
            else:
                for i in range(1, n):
                    if self.cells[i].recurrent:
                        plan[i] = (cur, 1)
                        cur += FWD_LM_PASSES
                    else:
                        plan[i] = (cur, 0)
                        cur += 1
            cache[key] = plan if cur <= 96 else None
        return cache[key]

    def _flow_out(self, B, H, W, dev):
        slot, self._flow_slot = self._flow_slot, None
        return slot if slot is not None else _f32((B, 2, H, W), dev)

    def _forward_pass(self, x_in, states, record):
        B, Cin, H, W = x_in.shape
        dev = x_in.device
        layers = []
        new_states = []
        in_bits = in_bitsT = None
        flow = None  # written by the last layer's kernel when the prediction head is fused into it
        target = self._final_target if self._final_hint else None
        self._final_hint = False
        if target is not None:
            self._final_target = None
            live = {t.data_ptr() for st in states if st is not None for t in st if t is not None}
            ok = len(target) == len(self.cells) and all(
                tg is not None and tuple(tg[0].shape) == (B, H, W, C) and all(t.data_ptr() not in live for t in tg)
                for tg in target)
            if not ok:  # one-pass window starting from the target itself, or another geometry: fresh tensors
                target = None
        defer = (record and self.__dict__.get("_defer_on", False) and self.precision == "bf16x3" and self.kind in ("lif", "plif", "xlif", "alif")
                 and PRED_FUSED)
        if not defer:
            self.flush_forward()  # (a pass outside the recorded schedule, e.g. under no_grad: what is recorded runs first)
        if defer:
            slots = self._fwd_slots(B, H, W)
            if self.__dict__.get("_defer_open") and (self._defer_t + len(self.cells) - 2 >= 96 if slots is None
                                                     else self._defer_t >= FWD_LM_PASSES):
                self.flush_forward()
            if not self.__dict__.get("_defer_open"):
                self._defer_begin()
        for i, c in enumerate(self.cells):
            if defer and i > 0:
                d = self._defer_t + i - 1 if slots is None else slots[i][0] + slots[i][1] * self._defer_t
                if _lib.raw("evf_fwd_defer_slot", d) != 0:
                    raise _lib.EvflowError("evf_fwd_defer_slot failed")
            st = states[i]
            v_prev, z_prev, zT_prev = st[:3] if st is not None else (None, None, None)
            plif = self._plif
            xf = self._xf  # (bits 1-2 of the reset flag: 2 an XLIF cell, 4 an ALIF cell, include/evflow.h)
            pt_prev = st[3] if (plif and st is not None) else None
            if target is not None:
                pt_out = target[i][3] if plif else None
            else:
                pt_out = _f32((B, H, W, C), dev) if plif else None
            P_out = _f32((B, H, W), dev) if plif else None
            if v_prev is not None and tuple(v_prev.shape) != (B, H, W, C):
                raise _lib.EvflowError("state shape does not match the input; call reset_states()")
            if target is not None:
                v_out, z_out, zT_out = target[i][:3]
            else:
                v_out, z_out = _f32((B, H, W, C), dev), _i32((B, H, W), dev)
                zT_out = _i32((B, H, C, (W + 31) // 32), dev)  # channel-major bit planes for the weight gradients
            leak, thresh = self._flat[f"{i}.leak"], self._flat[f"{i}.thresh"]
            if plif and i == 0:
                _lib.call("evf_head_plif_fwd", _lib.ptr(x_in), _lib.ptr(self._flat["0.ff"]), _lib.ptr(leak),
                          _lib.ptr(self._flat["0.leak_pt"]), _lib.ptr(self._flat["0.add_pt"]), _lib.ptr(thresh), _lib.ptr(v_prev),
                          _lib.ptr(z_prev), _lib.ptr(pt_prev), B, Cin, H, W, (1 if c.hard_reset else 0) | xf, _lib.ptr(v_out),
                          _lib.ptr(z_out), _lib.ptr(zT_out), _lib.ptr(pt_out), _lib.ptr(P_out))
            elif plif:
                wrec = self._packed[(i, "rec", "b3")] if c.recurrent else None
                args = (_lib.ptr(in_bits), _lib.ptr(self._packed[(i, "ff", "b3")]), _lib.ptr(wrec),
                        _lib.ptr(leak), _lib.ptr(self._flat[f"{i}.leak_pt"]), _lib.ptr(self._flat[f"{i}.add_pt"]),
                        _lib.ptr(thresh), _lib.ptr(v_prev), _lib.ptr(z_prev), _lib.ptr(pt_prev), B, H, W,
                        (1 if c.hard_reset else 0) | xf, _lib.ptr(v_out), _lib.ptr(z_out), _lib.ptr(zT_out), _lib.ptr(pt_out),
                        _lib.ptr(P_out))
                if i == len(self.cells) - 1 and PRED_FUSED:  # last layer: the prediction head runs in this kernel's epilogue
                    flow = self._flow_out(B, H, W, dev)
                    _lib.call("evf_conv_plif_fwd_b3_pred", *args, _lib.ptr(self._flat["pred.w"]), _lib.ptr(self._flat["pred.b"]),
                              _lib.ptr(flow))
                else:
                    _lib.call("evf_conv_plif_fwd_b3", *args)
            elif i == 0:
                _lib.call("evf_head_lif_fwd", _lib.ptr(x_in), _lib.ptr(self._flat["0.ff"]), _lib.ptr(leak), _lib.ptr(thresh),
                          _lib.ptr(v_prev), _lib.ptr(z_prev), B, Cin, H, W, 1 if c.hard_reset else 0, _lib.ptr(v_out),
                          _lib.ptr(z_out), _lib.ptr(zT_out))
            else:
                fmt = "b3" if self.precision == "bf16x3" else 0
                wrec = self._packed[(i, "rec", fmt)] if c.recurrent else None
                if fmt == "b3" and i == len(self.cells) - 1 and PRED_FUSED:
                    # last layer: the prediction head runs in this kernel's epilogue
                    flow = self._flow_out(B, H, W, dev)
                    _lib.call("evf_conv_lif_fwd_b3_pred", _lib.ptr(in_bits), _lib.ptr(self._packed[(i, "ff", fmt)]), _lib.ptr(wrec),
                              _lib.ptr(leak), _lib.ptr(thresh), _lib.ptr(v_prev), _lib.ptr(z_prev), B, H, W,
                              1 if c.hard_reset else 0, _lib.ptr(v_out), _lib.ptr(z_out), _lib.ptr(zT_out),
                              _lib.ptr(self._flat["pred.w"]), _lib.ptr(self._flat["pred.b"]), _lib.ptr(flow))
                else:
                    _lib.call("evf_conv_lif_fwd_b3" if fmt == "b3" else "evf_conv_lif_fwd", _lib.ptr(in_bits),
                              _lib.ptr(self._packed[(i, "ff", fmt)]), _lib.ptr(wrec),
                              _lib.ptr(leak), _lib.ptr(thresh), _lib.ptr(v_prev), _lib.ptr(z_prev), B, H, W,
                              1 if c.hard_reset else 0, _lib.ptr(v_out), _lib.ptr(z_out), _lib.ptr(zT_out))
            if record:
                layers.append((in_bits, v_prev, z_prev, v_out, z_out, in_bitsT, zT_prev, pt_prev, pt_out, P_out))
            in_bits, in_bitsT = z_out, zT_out
            new_states.append((v_out, z_out, zT_out, pt_out) if plif else (v_out, z_out, zT_out))
        if flow is None:
            flow = self._flow_out(B, H, W, dev)
            _lib.call("evf_pred_fwd", _lib.ptr(in_bits), _lib.ptr(self._flat["pred.w"]), _lib.ptr(self._flat["pred.b"]), B, H, W,
                      _lib.ptr(flow))
        if defer:
            self._defer_t += 1
        tape = {"x_in": x_in, "layers": layers, "flow": flow} if record else None
        return flow, tape, new_states

    # ------------------------------------------------------------------ backward
    def _small(self, win, name):
        off, n = self.small_off[name]
        return win.small[off : off + n]

    def _rowed(self, win, name):
        """(tensor the kernel's per-channel output pointer refers to, row pitch): row 0 of the per-block rows when the
        window has them, else the dense accumulator (pitch 0 = atomics)."""
        off, n = self.small_off[name]
        if win.rows is None:
            return win.small[off : off + n], 0
        return win.rows[0, off : off + n], win.rows.shape[1]

    def _slab(self, key, nslab, dev):
        if key not in self._slabs or self._slabs[key].shape[0] != nslab or self._slabs[key].device != dev:
            self._slabs[key] = _f32((nslab, 9 * C * C), dev)
        return self._slabs[key]

    # ---- PLIF: backward of a window layer by layer ---------------------------------------------------------------------------
    def _lm_wanted(self, win, tape, g_flow):
        if not (PLIF_LAYER_MAJOR and self._plif and self.precision == "bf16x3" and g_flow is not None):
            return False
        n = len(self.cells)
        return (HEAD_WIN and PLIF_TRACE_FUSED and PLIF_BOX_IN_DGRAD and TOP_FUSED and F32_DGRAD and PAIR_DGRAD and PARAM_ROWS
                and self.__dict__.get("_bdefer_on", False) and win.rows is not None and n > 2 and not self.cells[n - 1].recurrent
                and tape["x_in"].shape[1] == 2 and len(win.lm) < 16 and win.bwd_k == 0
                and all(c.hard_reset and c.activation == "arctanspike" for c in self.cells))

    def _top_static(self, B, H, W):
        """The part of _top_wanted that depends on the network, the switches and the shape only (bench.py's accounting asks)."""
        if not (LIF_BWD_TOP and self.kind == "lif" and self.precision == "bf16x3"):
            return False
        n = len(self.cells)
        last_rec = max([i for i, c in enumerate(self.cells) if c.recurrent], default=0)
        return bool(HEAD_WIN and TOP_FUSED and F32_DGRAD and PAIR_DGRAD and PARAM_ROWS and SPLIT_DGRAD and n > 2 and 0 < last_rec < n - 1
                    and all(c.hard_reset and c.activation == "arctanspike" for c in self.cells)
                    and _lib.load().evf_conv_dgrad_b3_multi_fits(B, H, W) == 1)

    def _top_wanted(self, win, tape, g_flow):
        return bool(g_flow is not None and self.__dict__.get("_bdefer_on", False) and win.rows is not None
                    and tape["x_in"].shape[1] == 2 and len(win.lm) < 16 and win.bwd_k == 0 and self._top_static(*win.shape))

    def _backward_window_top(self, win):
        """LIF: the backward of the window's passes for the feed-forward layers above the last recurrent one, a launch per layer
        (k_bwd_win_lif[_top]: what evf_lif_bwd_wgrad_top / evf_lif_bwd_wgrad2 compute pass by pass, dL/dv and the potential carried in
        registers) + a launch for the layer's input gradients of all passes; then the passes' remaining layers are recorded on the
        diagonals as ever (_backward_pass without dL/dflow: the top layers have nothing pending).  Index s = 0 .. T-1 is the
        backward order (s = 0: the window's last pass).  Reference: autograd of train_flow.py:141-154 over models/model.py:255-265."""
        stash, win.lm = win.lm, []
        B, H, W = win.shape
        dev, n, T = win.dev, len(self.cells), len(stash)
        tapes = [st[0] for st in stash]
        gflows = [st[1].float().contiguous() for st in stash]
        firsts = [st[2] for st in stash]
        last_rec = max(i for i, c in enumerate(self.cells) if c.recurrent)
        nsl = _lib.load().evf_lif_bwd_wgrad_slabs(B, H, W)
        shp = (B, H, W, C)
        gz = lambda i, s_: self._lm_buf(("gz", i, s_), shp, dev)  # noqa: E731  dL/d(spikes) of layer i at pass s
        gsp = [self._lm_buf(("gsp", s_), (3, B, H, W, C), dev, torch.bfloat16) for s_ in range(T)]  # planes of dL/d(current)
        arr = lambda ts: (ctypes.c_void_p * len(ts))(*[_lib.ptr(t) for t in ts])  # noqa: E731
        rowp = lambda name: _lib.ptr(self._rowed(win, name)[0])  # noqa: E731
        row_ld = win.rows.shape[1]
        # (the launches below run at once, in stream order; the tapes and upstream gradients they read are kept like a recorded pass's)
        self.__dict__.setdefault("_bdefer_keep", []).extend((tp, None, gf) for tp, gf in zip(tapes, gflows))
        for i in range(n - 1, last_rec, -1):
            lay = [tp["layers"][i] for tp in tapes]  # (in_bits, v_prev, z_prev, v_out, z_out, in_bitsT, zT_prev, ...)
            top = i == n - 1
            kf = (i, "ff")
            _lib.call("evf_lif_bwd_wgrad_window", T, None if top else arr([gz(i, s_) for s_ in range(T)]),
                      arr([tp["flow"] for tp in tapes]) if top else None, arr(gflows) if top else None,
                      _lib.ptr(self._flat["pred.w"]) if top else None, arr([l_[4] for l_ in lay]) if top else None,
                      rowp("pred.w") if top else None, rowp("pred.b") if top else None, arr([l_[3] for l_ in lay]),
                      arr([l_[1] for l_ in lay]), arr([l_[2] for l_ in lay]), arr([l_[5] for l_ in lay]), None, arr(gsp),
                      _lib.ptr(self._flat[f"{i}.leak"]), _lib.ptr(self._flat[f"{i}.thresh"]), B, H, W, self._act_width(i), None,
                      rowp(f"{i}.leak"), rowp(f"{i}.thresh"), _lib.ptr(self._slab(kf, nsl, dev)),
                      (1 if win.slab_init.get(kf) else 0) | (row_ld << 8))
            win.slab_init[kf] = True
            _lib.call("evf_conv_dgrad_b3_multi", T, arr(gsp), arr([self._packed[(i, "ff", "b3t")]] * T),
                      arr([gz(i - 1, s_) for s_ in range(T)]), None, None, B, H, W)
        for s_ in range(T):  # the layers from the last recurrent one down: recorded, diagonal by diagonal
            win.gz[last_rec], win.gz_has[last_rec] = gz(last_rec, s_), True
            self._backward_pass(win, tapes[s_], None, firsts[s_])
            win.bwd_k += 1

    def _lm_replay(self, win):
        stash, win.lm = win.lm, []
        for tape, g_flow, is_first in stash:
            self._backward_pass(win, tape, g_flow, is_first)
            win.bwd_k += 1

    def _lm_buf(self, key, shape, dev, dtype=torch.float32):
        """Persistent scratch of the layer-major backward (per-pass gradient maps: reused by every window of this shape)."""
        d = self.__dict__.setdefault("_lm_bufs", {})
        k = (key, tuple(shape), str(dev))
        if k not in d:
            d[k] = torch.empty(shape, dtype=dtype, device=dev)
        return d[k]

    def _backward_window_lm(self, win):
        """The backward of all passes of a PLIF window, layer by layer from the top (reference: autograd of train_flow.py:141-154
        over models/model.py:255-265 -- the same cells, another order: cell (l, t) needs (l + 1, t) and (l, t + 1) only).  Index s =
        0 .. T-1 is the backward order (s = 0: the window's last pass)."""
        stash, win.lm = win.lm, []
        B, H, W = win.shape
        dev, n, T = win.dev, len(self.cells), len(stash)
        tapes = [st[0] for st in stash]
        gflows = [st[1].float().contiguous() for st in stash]
        firsts = [st[2] for st in stash]
        L = _lib.load()
        nsl = L.evf_lif_bwd_wgrad_slabs(B, H, W)
        shp = (B, H, W, C)
        gz = lambda i, s_: self._lm_buf(("gz", i, s_), shp, dev)  # noqa: E731  dL/d(spikes) of layer i at pass s
        # dL/d(current) of the layer in work, per pass: as its three bf16 planes (k_dgrad_diag_dma) or as the fp32 tensor (k_conv_dgrad_ws)
        split = PLIF_LM_DGRAD != "ws" and L.evf_conv_dgrad_b3_multi_fits(B, H, W) == 1
        if self._al:
            split = False  # (ALIF: a recurrent cell's input gradient is ADDED to g_zx -- the accumulating fp32 form, evf_conv_dgrad_b3_f32)
        trace_in = not self._al  # the trace depends on the cell's INPUT (PLIF / XLIF: its gradient rides in the input-gradient kernels)
        if split:
            gsp = [self._lm_buf(("gsp", s_), (3, B, H, W, C), dev, torch.bfloat16) for s_ in range(T)]
            gcur = [None] * T
        else:
            gsp = [None] * T
            gcur = [self._lm_buf(("gcur", s_), shp, dev) for s_ in range(T)]
        gPs = [self._lm_buf(("gP", s_), (B, H, W), dev) for s_ in range(T)]
        arr = lambda ts: (ctypes.c_void_p * len(ts))(*[_lib.ptr(t) for t in ts])  # noqa: E731
        arr_n = lambda ts: arr(ts) if ts[0] is not None else None  # noqa: E731
        rowp = lambda name: _lib.ptr(self._rowed(win, name)[0])  # noqa: E731
        row_ld = win.rows.shape[1]
        xf = self._xf  # (bits 1-2 of the reset / accumulate flag: XLIF / ALIF cells, include/evflow.h)
        for i in range(n - 1, 0, -1):
            c = self.cells[i]
            lay = [tp["layers"][i] for tp in tapes]  # (in_bits, v_prev, z_prev, v_out, z_out, in_bitsT, zT_prev, pt_prev, pt_out, P)
            kf, kr = (i, "ff"), (i, "rec")
            gv_i, gpt_i = self._lm_buf(("gv", i), shp, dev), self._lm_buf(("gpt", i), shp, dev)
            leak, thr = _lib.ptr(self._flat[f"{i}.leak"]), _lib.ptr(self._flat[f"{i}.thresh"])
            lpt, apt = _lib.ptr(self._flat[f"{i}.leak_pt"]), _lib.ptr(self._flat[f"{i}.add_pt"])
            width = self._act_width(i)
            if i == n - 1:  # under the prediction head: its backward inside, all passes in one launch
                _lib.call("evf_plif_bwd_wgrad_window_top", T, arr([tp["flow"] for tp in tapes]), arr(gflows), _lib.ptr(self._flat["pred.w"]),
                          arr([l_[4] for l_ in lay]), rowp("pred.w"), rowp("pred.b"), arr([l_[3] for l_ in lay]), arr([l_[1] for l_ in lay]),
                          arr([l_[2] for l_ in lay]), arr([l_[5] for l_ in lay]), arr_n(gcur), arr_n(gsp), arr([l_[7] for l_ in lay]),
                          arr([l_[9] for l_ in lay]), arr(gPs), leak, thr, lpt, apt, B, H, W, width, None, None, rowp(f"{i}.leak"),
                          rowp(f"{i}.thresh"), rowp(f"{i}.leak_pt"), rowp(f"{i}.add_pt"), _lib.ptr(self._slab(kf, nsl, dev)),
                          (1 if win.slab_init.get(kf) else 0) | xf | (row_ld << 8))
                win.slab_init[kf] = True
            elif not c.recurrent:  # feed-forward: all passes in one launch, the carries in registers
                _lib.call("evf_plif_bwd_wgrad_window", T, arr([gz(i, s_) for s_ in range(T)]), arr([l_[3] for l_ in lay]),
                          arr([l_[1] for l_ in lay]), arr([l_[2] for l_ in lay]), arr([l_[5] for l_ in lay]), arr_n(gcur), arr_n(gsp),
                          arr([l_[7] for l_ in lay]), arr([l_[9] for l_ in lay]), arr(gPs), leak, thr, lpt, apt, B, H, W, width, None, None,
                          rowp(f"{i}.leak"), rowp(f"{i}.thresh"), rowp(f"{i}.leak_pt"), rowp(f"{i}.add_pt"),
                          _lib.ptr(self._slab(kf, nsl, dev)), (1 if win.slab_init.get(kf) else 0) | xf | (row_ld << 8))
                win.slab_init[kf] = True
            if not c.recurrent and split:  # input gradients of all passes in one launch (the pooling's adjoint of dL/dP inside)
                wts = [self._packed[(i, "ff", "b3t")]] * T
                _lib.call("evf_conv_dgrad_b3_multi", T, arr(gsp), arr(wts), arr([gz(i - 1, s_) for s_ in range(T)]), arr(gPs),
                          arr([l_[0] for l_ in lay]), B, H, W)
                continue
            if not c.recurrent:  # ... or one launch per pass (accumulate | 2: the pooling's adjoint inside)
                for s_ in range(T):
                    _lib.call("evf_conv_dgrad_b3_f32", _lib.ptr(gcur[s_]), _lib.ptr(self._packed[(i, "ff", "b3t")]), _lib.ptr(gz(i - 1, s_)),
                              2 if trace_in else 0, B, H, W, _lib.ptr(gPs[s_]) if trace_in else None, _lib.ptr(lay[s_][0]) if trace_in else None)
                continue
            # recurrent: pass by pass (the cell reads its own recurrent input gradient of the pass after)
            gzr_i = self._lm_buf(("gzr", i), shp, dev)
            has_gzr = False
            for s_ in range(T):
                in_bits, v_prev, z_prev, v_out, _zo, in_bitsT, zT_prev, pt_prev, _pto, P_sav = lay[s_]
                use_rec = z_prev is not None
                acc = 1 if win.slab_init.get(kf) else 0
                if use_rec and bool(win.slab_init.get(kr)) != bool(acc):
                    _lib.zero_(self._slab(kr, nsl, dev))  # (first recurrent contribution later than the feed-forward one)
                # (ALIF: gzr_i also takes the cell's g_zx -- read as the second part of dL/d(spikes), then written, by the same thread)
                _lib.call("evf_plif_bwd_wgrad2", _lib.ptr(gz(i, s_)), _lib.ptr(gzr_i) if has_gzr else None, _lib.ptr(gv_i) if s_ else None,
                          _lib.ptr(v_out), _lib.ptr(v_prev), _lib.ptr(z_prev), _lib.ptr(in_bitsT), _lib.ptr(zT_prev) if use_rec else None,
                          leak, thr, B, H, W, 1 | xf, SURROGATE_ID[c.activation], width, _lib.ptr(gcur[s_]), _lib.ptr(gsp[s_]), _lib.ptr(gv_i),
                          rowp(f"{i}.leak"), rowp(f"{i}.thresh"), _lib.ptr(self._slab(kf, nsl, dev)),
                          _lib.ptr(self._slab(kr, nsl, dev)) if use_rec else None, acc | (row_ld << 8),
                          _lib.ptr(gpt_i) if s_ else None, _lib.ptr(pt_prev), _lib.ptr(P_sav), lpt, apt, _lib.ptr(gpt_i),
                          _lib.ptr(gzr_i if self._al else gPs[s_]), rowp(f"{i}.leak_pt"), rowp(f"{i}.add_pt"))
                win.slab_init[kf] = True
                if use_rec:
                    win.slab_init[kr] = True
                rec_grad = use_rec and not firsts[s_]
                if split:  # both products of the pass (feed-forward: with the trace term; recurrent: to the cell's own previous spikes)
                    np_ = 2 if rec_grad else 1
                    a2 = lambda ts: (ctypes.c_void_p * np_)(*[_lib.ptr(t) for t in ts[:np_]])  # noqa: E731
                    _lib.call("evf_conv_dgrad_b3_multi", np_, a2([gsp[s_], gsp[s_]]),
                              a2([self._packed[(i, "ff", "b3t")], self._packed[(i, "rec", "b3t")]]), a2([gz(i - 1, s_), gzr_i]),
                              a2([gPs[s_], None]), a2([in_bits, None]), B, H, W)
                elif rec_grad and trace_in:
                    _lib.call("evf_conv_dgrad_b3_f32_pair", _lib.ptr(gcur[s_]), _lib.ptr(self._packed[(i, "ff", "b3t")]),
                              _lib.ptr(gz(i - 1, s_)), 2, _lib.ptr(self._packed[(i, "rec", "b3t")]), _lib.ptr(gzr_i), B, H, W,
                              _lib.ptr(gPs[s_]), _lib.ptr(in_bits))
                else:
                    _lib.call("evf_conv_dgrad_b3_f32", _lib.ptr(gcur[s_]), _lib.ptr(self._packed[(i, "ff", "b3t")]), _lib.ptr(gz(i - 1, s_)),
                              2 if trace_in else 0, B, H, W, _lib.ptr(gPs[s_]) if trace_in else None, _lib.ptr(in_bits) if trace_in else None)
                    if rec_grad:  # (ALIF) ... and the recurrent product ADDED to the g_zx the cell has just stored
                        _lib.call("evf_conv_dgrad_b3_f32", _lib.ptr(gcur[s_]), _lib.ptr(self._packed[(i, "rec", "b3t")]), _lib.ptr(gzr_i),
                                  1, B, H, W, None, None)
                has_gzr = rec_grad or self._al
        # head layer: its passes recorded, one launch at the flush (k_head_bwd_win<.., PLIF>)
        c = self.cells[0]
        nslh = L.evf_head_lif_bwd_wgrad_slabs(B, H, W)
        key = (0, "ff")
        if key not in self._slabs or self._slabs[key].shape != (nslh, C * 18) or self._slabs[key].device != dev:
            self._slabs[key] = _f32((nslh, C * 18), dev)
        gv0, gpt0 = self._lm_buf(("gv", 0), shp, dev), self._lm_buf(("gpt", 0), shp, dev)
        self._bdefer_cur = None
        self.__dict__.setdefault("_bdefer_keep", []).extend((tp, None, None) for tp in tapes)
        self._bdefer_slot(win, 0)
        if _lib.raw("evf_bwd_defer_hold_heads", 1) != 0:
            raise _lib.EvflowError("evf_bwd_defer_hold_heads failed")
        for s_ in range(T):
            _in, v_prev, z_prev, v_out, _zo, _inT, _zT, pt_prev, _pto, P_sav = tapes[s_]["layers"][0]
            if _lib.raw("evf_bwd_defer_slot", s_) != 0:
                raise _lib.EvflowError("evf_bwd_defer_slot failed")
            _lib.call("evf_head_plif_bwd_wgrad", _lib.ptr(gz(0, s_)), _lib.ptr(gv0) if s_ else None, _lib.ptr(v_out), _lib.ptr(v_prev),
                      _lib.ptr(z_prev), _lib.ptr(tapes[s_]["x_in"]), _lib.ptr(self._flat["0.leak"]), _lib.ptr(self._flat["0.thresh"]), B, 2, H, W,
                      1 | xf, SURROGATE_ID[c.activation], self._act_width(0), _lib.ptr(gv0), rowp("0.leak"), rowp("0.thresh"),
                      _lib.ptr(self._slabs[key]), (1 if win.slab_init.get(key) else 0) | (row_ld << 8), _lib.ptr(gpt0) if s_ else None,
                      _lib.ptr(pt_prev), _lib.ptr(self._lm_buf(("gzx", 0), shp, dev) if self._al else P_sav),  # (ALIF: g_zx in the P slot)
                      _lib.ptr(self._flat["0.leak_pt"]), _lib.ptr(self._flat["0.add_pt"]), _lib.ptr(gpt0),
                      rowp("0.leak_pt"), rowp("0.add_pt"))
            win.slab_init[key] = True
        win.bwd_k += T
        # (the caller -- _FireNetPass.backward of the window's first pass -- flushes the recording and finalizes)

    def _backward_pass(self, win, tape, g_flow, is_first):
        B, H, W = win.shape
        dev = win.dev
        n = len(self.cells)
        layers = tape["layers"]
        nslab = _lib.load().evf_conv_wgrad_slabs(B, H, W)
        # the prediction head's backward runs inside the fused backward of the (non-recurrent) layer below it
        top_fused = (TOP_FUSED and g_flow is not None and self.precision == "bf16x3" and n > 1 and not self._al
                     and not self.cells[n - 1].recurrent and not win.gz_has[n - 1])  # (ALIF: evf_pred_bwd + the plain cell)
        g_flow_c = g_flow.float().contiguous() if g_flow is not None else None
        if self.__dict__.get("_bdefer_on", False):
            # recorded cells of this pass are launched later (flush_backward): its tape and the contiguous upstream gradient
            # must outlive this function (autograd drops ctx.tape and g_flow as soon as the node returns)
            self._bdefer_cur = (tape, g_flow, g_flow_c)
            self.__dict__.setdefault("_bdefer_keep", []).append(self._bdefer_cur)
        if g_flow is not None and not top_fused:
            gz_top = win.buf(win.gz, n - 1)
            _lib.call("evf_pred_bwd", _lib.ptr(layers[n - 1][4]), _lib.ptr(tape["flow"]),
                      _lib.ptr(g_flow_c), _lib.ptr(self._flat["pred.w"]), B, H, W, _lib.ptr(gz_top),
                      _lib.ptr(self._small(win, "pred.w")), _lib.ptr(self._small(win, "pred.b")))
            win.gz_has[n - 1] = True
        bdefer = (self.__dict__.get("_bdefer_on", False) and self.precision == "bf16x3" and self.kind == "lif" and F32_DGRAD
                  and PAIR_DGRAD and (top_fused or g_flow is None) and tape["x_in"].shape[1] == 2)
        # PLIF: only the head layer's cells wait for the window's end (their chain is per pixel and reads one dL/d(spikes) buffer per
        # pass): a recording is open for them, the hidden cells' input gradients (pooling adjoint: not recordable) flush the rest
        # pass by pass, and evf_bwd_defer_flush runs the head's passes in ONE launch with the trace backward inside
        plif_hw = (self._plif and HEAD_WIN and PLIF_TRACE_FUSED and self.__dict__.get("_bdefer_on", False)
                   and self.precision == "bf16x3" and n > 1 and tape["x_in"].shape[1] == 2 and self.cells[0].hard_reset
                   and self.cells[0].activation == "arctanspike" and (top_fused or g_flow is None))
        if plif_hw:
            self._bdefer_slot(win, 0)
            if _lib.raw("evf_bwd_defer_hold_heads", 1) != 0:
                raise _lib.EvflowError("evf_bwd_defer_hold_heads failed")
        elif not bdefer:
            self.flush_backward()  # (a pass outside the recorded schedule: what is recorded runs first)
        if win.g_cur is None:
            win.g_cur = _f32((B, H, W, C), dev)
            if self.precision == "bf16x3":
                if not F32_DGRAD:
                    win.g_split = torch.empty((3, B, H, W, C), dtype=torch.bfloat16, device=dev)
        for i in range(n - 1, -1, -1):
            c = self.cells[i]
            in_bits, v_prev, z_prev, v_out, _, in_bitsT, zT_prev, pt_prev, pt_out, P_sav = layers[i]
            plif = self._plif
            xf = self._xf
            g_z = win.gz[i] if win.gz_has[i] else None
            g_z2 = win.gzr[i] if win.gzr_has[i] else None  # from the cell's own recurrent input gradient (pass t + 1)
            g_v = win.gv[i]
            win.gz_has[i] = win.gzr_has[i] = False
            top = top_fused and i == n - 1
            if g_z is None and g_z2 is None and g_v is None and not top:
                continue  # no gradient reaches this layer at this pass
            use_rec = c.recurrent and z_prev is not None
            split = bdefer and SPLIT_DGRAD and i > 0
            if split:  # the fused backward writes the three bf16 planes of g_cur instead of the fp32 tensor
                if win.gsl[i] is None:
                    win.gsl[i] = torch.empty((3, B, H, W, C), dtype=torch.bfloat16, device=dev)
                g_cur_i, g_split_i = None, win.gsl[i]
            else:
                g_cur_i = win.buf(win.gcl, i) if bdefer else win.g_cur  # (several layers are in flight under diagonal launches)
                g_split_i = win.g_split
            if bdefer:
                self._bdefer_slot(win, 2 * (n - 1 - i))
            gv_out = win.buf(win.gv, i)
            leak_g, thr_g = self._small(win, f"{i}.leak"), self._small(win, f"{i}.thresh")
            (leak_r, row_ld), (thr_r, _) = self._rowed(win, f"{i}.leak"), self._rowed(win, f"{i}.thresh")
            # PLIF: the trace backward rides in the fused backward (default neuron, pooling adjoint inside the input-gradient kernels)
            trace_fused = (plif and PLIF_TRACE_FUSED and PLIF_BOX_IN_DGRAD and i > 0 and self.precision == "bf16x3"
                           and c.hard_reset and c.activation == "arctanspike")
            if (self._xl or self._al) and ((i > 0 and not trace_fused) or (i == 0 and tape["x_in"].shape[1] != 2)):
                raise _lib.EvflowError("fused XLIF / ALIF cells need the trace backward inside the fused backward kernels (EVF_PLIF_TRACE_FUSED, "
                                       "EVF_PLIF_BOX=dgrad, a two-channel input); EVF_XLIF_FUSED=0 serves the network on the general path")
            if trace_fused:
                if win.gP is None:
                    win.gP = _f32((B, H, W), dev)
                    win.gP_raw = _f32((B, H, W), dev)
                gpt_out = win.buf(win.gpt, i)
                # ALIF: the kernel's g_P_raw slot takes g_zx [B,H,W,32] -- what this pass sends into dL/d(spikes) of the pass before
                # through the threshold trace; it comes back as g_z_out2 (same buffer: read, then written, by the same thread)
                gzx_buf = win.buf(win.gzr, i) if self._al else None
                trace_args = (_lib.ptr(gpt_out if win.gpt_has[i] else None), _lib.ptr(pt_prev), _lib.ptr(P_sav),
                              _lib.ptr(self._flat[f"{i}.leak_pt"]), _lib.ptr(self._flat[f"{i}.add_pt"]), _lib.ptr(gpt_out),
                              _lib.ptr(gzx_buf if self._al else win.gP_raw), _lib.ptr(self._rowed(win, f"{i}.leak_pt")[0]),
                              _lib.ptr(self._rowed(win, f"{i}.add_pt")[0]))
                win.gpt_has[i] = not is_first
                if self._al:
                    win.gzr_has[i] = not is_first
            if i > 0 and self.precision == "bf16x3":
                # neuron backward + both weight gradients in one pass (evf_bwd_fused.hip)
                kf, kr = (i, "ff"), (i, "rec")
                nsl = _lib.load().evf_lif_bwd_wgrad_slabs(B, H, W)
                acc_flag = 1 if win.slab_init.get(kf) else 0
                if use_rec and bool(win.slab_init.get(kr)) != bool(acc_flag):
                    # first recurrent contribution arrives later than the ff one: start its slab at zero
                    _lib.zero_(self._slab(kr, nsl, dev))
                if top:
                    _lib.call("evf_plif_bwd_wgrad_top" if trace_fused else "evf_lif_bwd_wgrad_top", _lib.ptr(tape["flow"]), _lib.ptr(g_flow_c), _lib.ptr(self._flat["pred.w"]),
                              _lib.ptr(layers[i][4]), _lib.ptr(self._rowed(win, "pred.w")[0]), _lib.ptr(self._rowed(win, "pred.b")[0]),
                              _lib.ptr(g_v), _lib.ptr(v_out), _lib.ptr(v_prev), _lib.ptr(z_prev), _lib.ptr(in_bitsT),
                              _lib.ptr(self._flat[f"{i}.leak"]), _lib.ptr(self._flat[f"{i}.thresh"]), B, H, W,
                              (1 if c.hard_reset else 0) | (xf if trace_fused else 0), SURROGATE_ID[c.activation], self._act_width(i),
                              _lib.ptr(g_cur_i) if (plif or F32_DGRAD) else None, _lib.ptr(g_split_i), _lib.ptr(gv_out),
                              _lib.ptr(leak_r), _lib.ptr(thr_r), _lib.ptr(self._slab(kf, nsl, dev)), acc_flag | (row_ld << 8),
                              *(trace_args if trace_fused else ()))
                else:
                    _lib.call("evf_plif_bwd_wgrad2" if trace_fused else "evf_lif_bwd_wgrad2", _lib.ptr(g_z), _lib.ptr(g_z2), _lib.ptr(g_v), _lib.ptr(v_out), _lib.ptr(v_prev),
                          _lib.ptr(z_prev),
                          _lib.ptr(in_bitsT), _lib.ptr(zT_prev) if use_rec else None, _lib.ptr(self._flat[f"{i}.leak"]),
                          _lib.ptr(self._flat[f"{i}.thresh"]), B, H, W, (1 if c.hard_reset else 0) | (xf if trace_fused else 0), SURROGATE_ID[c.activation],
                          self._act_width(i), _lib.ptr(g_cur_i) if (plif or F32_DGRAD) else None, _lib.ptr(g_split_i),
                          _lib.ptr(gv_out),
                          _lib.ptr(leak_r), _lib.ptr(thr_r),
                          _lib.ptr(self._slab(kf, nsl, dev)), _lib.ptr(self._slab(kr, nsl, dev)) if use_rec else None,
                          acc_flag | (row_ld << 8), *(trace_args if trace_fused else ()))
                win.slab_init[kf] = True
                if use_rec:
                    win.slab_init[kr] = True
            elif i == 0 and tape["x_in"].shape[1] == 2:
                # head: neuron backward + weight gradient in one pass (per-block slabs, summed in _finalize)
                nsl = _lib.load().evf_head_lif_bwd_wgrad_slabs(B, H, W)
                key = (0, "ff")
                if key not in self._slabs or self._slabs[key].shape != (nsl, C * 18) or self._slabs[key].device != dev:
                    self._slabs[key] = _f32((nsl, C * 18), dev)
                if plif_hw or self._xl or self._al:  # ... and the trace backward in the same pass; recorded (see plif_hw above; XLIF / ALIF: always this form)
                    gpt_out = win.buf(win.gpt, 0)
                    _lib.call("evf_head_plif_bwd_wgrad", _lib.ptr(g_z), _lib.ptr(g_v), _lib.ptr(v_out), _lib.ptr(v_prev),
                              _lib.ptr(z_prev), _lib.ptr(tape["x_in"]), _lib.ptr(self._flat["0.leak"]),
                              _lib.ptr(self._flat["0.thresh"]), B, 2, H, W, 1 | xf, SURROGATE_ID[c.activation], self._act_width(0),
                              _lib.ptr(gv_out), _lib.ptr(leak_r), _lib.ptr(thr_r), _lib.ptr(self._slabs[key]),
                              (1 if win.slab_init.get(key) else 0) | (row_ld << 8),
                              _lib.ptr(gpt_out if win.gpt_has[0] else None), _lib.ptr(pt_prev),
                              _lib.ptr(win.buf(win.gzr, 0) if self._al else P_sav),  # (ALIF: the g_zx buffer in the P slot, include/evflow.h)
                              _lib.ptr(self._flat["0.leak_pt"]), _lib.ptr(self._flat["0.add_pt"]), _lib.ptr(gpt_out),
                              _lib.ptr(self._rowed(win, "0.leak_pt")[0]), _lib.ptr(self._rowed(win, "0.add_pt")[0]))
                    win.gpt_has[0] = not is_first
                    trace_fused = True  # (no evf_plif_trace_bwd below)
                else:
                    _lib.call("evf_head_lif_bwd_wgrad", _lib.ptr(g_z), _lib.ptr(g_v), _lib.ptr(v_out), _lib.ptr(v_prev),
                              _lib.ptr(z_prev), _lib.ptr(tape["x_in"]), _lib.ptr(self._flat["0.leak"]),
                              _lib.ptr(self._flat["0.thresh"]), B, 2, H, W, 1 if c.hard_reset else 0, SURROGATE_ID[c.activation],
                              self._act_width(0), _lib.ptr(win.g_cur) if plif else None, _lib.ptr(gv_out), _lib.ptr(leak_r),
                              _lib.ptr(thr_r), _lib.ptr(self._slabs[key]), (1 if win.slab_init.get(key) else 0) | (row_ld << 8))
                win.slab_init[key] = True
            else:
                _lib.call("evf_lif_bwd", _lib.ptr(g_z), _lib.ptr(g_v), _lib.ptr(v_out), _lib.ptr(v_prev), _lib.ptr(z_prev),
                          _lib.ptr(self._flat[f"{i}.leak"]), _lib.ptr(self._flat[f"{i}.thresh"]), B, H, W,
                          1 if c.hard_reset else 0, SURROGATE_ID[c.activation], self._act_width(i), _lib.ptr(win.g_cur),
                          _lib.ptr(gv_out), _lib.ptr(leak_g), _lib.ptr(thr_g))
                # weight gradients
                if i == 0:
                    _lib.call("evf_head_wgrad", _lib.ptr(tape["x_in"]), _lib.ptr(win.g_cur), B, tape["x_in"].shape[1], H, W,
                              _lib.ptr(self._small(win, "0.ff")))
                else:
                    k = (i, "ff")
                    _lib.call("evf_conv_wgrad_bits", _lib.ptr(in_bits), _lib.ptr(win.g_cur), B, H, W,
                              _lib.ptr(self._slab(k, nslab, dev)), 1 if win.slab_init.get(k) else 0)
                    win.slab_init[k] = True
                if use_rec:
                    k = (i, "rec")
                    _lib.call("evf_conv_wgrad_bits", _lib.ptr(z_prev), _lib.ptr(win.g_cur), B, H, W,
                              _lib.ptr(self._slab(k, nslab, dev)), 1 if win.slab_init.get(k) else 0)
                    win.slab_init[k] = True
            if plif and not trace_fused:  # trace backward: carries dL/dpt, yields dL/d(pooled activity) for the input-spike gradient
                if win.gP is None:
                    win.gP = _f32((B, H, W), dev)
                    win.gP_raw = _f32((B, H, W), dev)
                gpt_out = win.buf(win.gpt, i)
                carry = gpt_out if win.gpt_has[i] else None
                _lib.call("evf_plif_trace_bwd", _lib.ptr(win.g_cur), _lib.ptr(carry), _lib.ptr(pt_prev), _lib.ptr(pt_out),
                          _lib.ptr(P_sav), _lib.ptr(self._flat[f"{i}.leak_pt"]), _lib.ptr(self._flat[f"{i}.add_pt"]), B, H, W,
                          _lib.ptr(gpt_out), _lib.ptr(win.gP_raw), None if PLIF_BOX_IN_DGRAD else _lib.ptr(win.gP),
                          _lib.ptr(self._rowed(win, f"{i}.leak_pt")[0]),
                          _lib.ptr(self._rowed(win, f"{i}.add_pt")[0]), self._rowed(win, f"{i}.add_pt")[1])
                win.gpt_has[i] = not is_first
            if is_first:
                win.gv[i] = None  # the state entering the window is detached (train_flow.py:170)
            # input gradients: to the layer below (this pass) and to the own previous spikes (previous pass)
            rec_grad = use_rec and not is_first
            if i > 0:
                if bdefer:
                    self._bdefer_slot(win, 2 * (n - 1 - i) + 1)
                if (bdefer or plif_hw) and i == 1 and HEAD_WIN and not win.gz_has[0]:
                    # dL/d(spikes) of the head layer in a buffer of this pass's own: the head's backward cells of the whole window
                    # then run after the last diagonal, all passes in one launch (evf_hd_defer_launch_window)
                    while len(win.gz0) <= win.bwd_k:
                        win.gz0.append(_f32((B, H, W, C), dev))
                    win.gz[0] = win.gz0[win.bwd_k]
                ga = win.buf(win.gz, i - 1)
                acc_a = 1 if win.gz_has[i - 1] else 0
                # PLIF: the input-gradient kernel applies AvgPool3x3^T / 32 to the trace backward's raw map itself (accumulate | 2)
                trace_in = plif and not self._al  # (the trace depends on the INPUT: PLIF / XLIF; an ALIF cell's on its own spikes)
                gp_flag = 2 if (trace_in and PLIF_BOX_IN_DGRAD) else 0
                gp_map = (win.gP_raw if PLIF_BOX_IN_DGRAD else win.gP) if trace_in else None
                if self.precision == "bf16x3":
                    dg, gsrc = ("evf_conv_dgrad_b3_f32", g_cur_i) if F32_DGRAD else ("evf_conv_dgrad_b3", win.g_split)
                    if split:
                        dg, gsrc = "evf_conv_dgrad_b3", g_split_i
                    if rec_grad and (F32_DGRAD or split) and PAIR_DGRAD and not self._al:  # both input gradients of the recurrent cell in one launch
                        # the recurrent one goes to its OWN buffer (gzr): the cell's backward of the previous pass adds the
                        # two parts itself (evf_lif_bwd_wgrad2), so the input gradient of the layer above needs no
                        # accumulating form there, and those two launches may come in either order
                        gb = win.buf(win.gzr, i) if not plif else win.buf(win.gz, i)
                        _lib.call("evf_conv_dgrad_b3_pair" if split else "evf_conv_dgrad_b3_f32_pair", _lib.ptr(gsrc),
                                  _lib.ptr(self._packed[(i, "ff", "b3t")]),
                                  _lib.ptr(ga), acc_a | gp_flag, _lib.ptr(self._packed[(i, "rec", "b3t")]), _lib.ptr(gb), B, H, W,
                                  _lib.ptr(gp_map) if trace_in else None, _lib.ptr(in_bits) if trace_in else None)
                        if plif:
                            win.gz_has[i] = True
                        else:
                            win.gzr_has[i] = True
                    else:
                        _lib.call(dg, _lib.ptr(gsrc), _lib.ptr(self._packed[(i, "ff", "b3t")]), _lib.ptr(ga),
                                  acc_a | gp_flag, B, H, W, _lib.ptr(gp_map) if trace_in else None, _lib.ptr(in_bits) if trace_in else None)
                        if rec_grad and self._al:  # ... ADDED to g_zx, which the cell's backward has just stored there
                            _lib.call(dg, _lib.ptr(gsrc), _lib.ptr(self._packed[(i, "rec", "b3t")]),
                                      _lib.ptr(win.buf(win.gzr, i)), 1, B, H, W, None, None)
                        elif rec_grad:
                            gb = win.buf(win.gz, i)
                            _lib.call(dg, _lib.ptr(gsrc), _lib.ptr(self._packed[(i, "rec", "b3t")]),
                                      _lib.ptr(gb), 0, B, H, W, None, None)
                            win.gz_has[i] = True
                elif rec_grad:
                    gb = win.buf(win.gz, i)
                    _lib.call("evf_conv_dgrad", _lib.ptr(win.g_cur), _lib.ptr(self._packed[(i, "ff", 1)]), _lib.ptr(ga), acc_a,
                              _lib.ptr(self._packed[(i, "rec", 1)]), _lib.ptr(gb), 0, B, H, W)
                    win.gz_has[i] = True
                else:
                    _lib.call("evf_conv_dgrad", _lib.ptr(win.g_cur), _lib.ptr(self._packed[(i, "ff", 1)]), _lib.ptr(ga), acc_a,
                              None, None, 0, B, H, W)
                win.gz_has[i - 1] = True
        self._bdefer_cur = None  # (the pass is recorded completely; its entry stays in _bdefer_keep until the flush)

    def _finalize_fused(self, win, nslab):
        """Every partial sum of the window -> the optimizer's flat gradient buffer in ONE launch (csrc/evf_step_tail.hip); only
        when every trainable parameter's .grad is bound to that buffer (FlatAdam).  False: the caller takes the step-by-step path."""
        if not hip_ops.DIRECT_PARAM_GRADS:
            return False
        red_src, red_dst, seg_dst, seg_off, seg_n, seg_rows = [], [], [], [], [], []
        # rows of the per-block partials a segment's writers can have touched: the head layer's launches (evf_head_*_bwd_wgrad: one
        # row per block of evf_head_lif_bwd_wgrad_slabs) / the hidden cells' (fused backward: <= evf_lif_bwd_wgrad_slabs blocks;
        # evf_plif_trace_bwd: <= 512)
        L = _lib.load()
        rows_head = L.evf_head_lif_bwd_wgrad_slabs(*win.shape)
        rows_hidden = max(L.evf_lif_bwd_wgrad_slabs(*win.shape), 512)
        for name, p in zip(self.pnames, self.params):
            if not p.requires_grad:
                continue
            if not (p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous() and p.grad.is_cuda):
                return False
            if name in self.small_off:
                seg_dst.append(p.grad)
                seg_off.append(self.small_off[name][0])
                seg_n.append(self.small_off[name][1])
                # (the head layer's TRACE parameters can also be written by the stand-alone evf_plif_trace_bwd -- EVF_PLIF_TRACE_FUSED=0,
                # or a head cell the fused window kernel does not serve -- with up to 512 rows whatever the head launch's block count)
                trace_par = name.endswith(("leak_pt", "add_pt"))
                seg_rows.append((max(rows_head, 512) if trace_par else rows_head) if name.startswith("0.") else rows_hidden)
            else:
                i, nm = name.split(".")
                if win.slab_init.get((int(i), nm)):
                    red_src.append(self._slabs[(int(i), nm)])
                    red_dst.append(p.grad)
        if len(red_src) > 16 or len(seg_dst) > 32 or not seg_dst:
            return False
        head = self._slabs[(0, "ff")] if ("0.ff" in self.small_off and win.slab_init.get((0, "ff"))) else None
        rows = win.rows
        n1, n2 = len(red_src), len(seg_dst)
        _lib.call("evf_grads_finalize", (ctypes.c_void_p * max(n1, 1))(*[t.data_ptr() for t in red_src]),
                  (ctypes.c_void_p * max(n1, 1))(*[t.data_ptr() for t in red_dst]), n1, nslab, _lib.ptr(win.small),
                  1 if win.small_persistent else 0, _lib.ptr(rows), rows.shape[0] if rows is not None else 0,
                  rows.shape[1] if rows is not None else 0, _lib.ptr(head), head.shape[0] if head is not None else 0,
                  head.shape[1] if head is not None else 0, self.small_off["0.ff"][0] if head is not None else 0,
                  (ctypes.c_void_p * n2)(*[t.data_ptr() for t in seg_dst]), (ctypes.c_int * n2)(*seg_off), (ctypes.c_int * n2)(*seg_n),
                  (ctypes.c_int * n2)(*seg_rows), n2)
        if rows is not None:
            # the kernel hands back zeroed the columns it consumed: the buffer as a whole is clean only when the trainable segments
            # cover every column (a frozen per-channel parameter's column keeps its partial sums -> _take_rows starts afresh)
            self._rows_clean = sum(seg_n) == self.small_size
        if win.small_persistent:
            self._small_clean = True
        self._last_window = win
        return True

    def _finalize(self, win):
        """Window complete: reduce the weight-gradient slabs, hand all parameter
        gradients to autograd (in self.params order)."""
        if hip_ops.DIRECT_PARAM_GRADS:
            hip_ops.direct_grads_written()  # (the reductions below add straight into the optimizer's flat gradient buffer)
        B, H, W = win.shape
        nslab = (_lib.load().evf_lif_bwd_wgrad_slabs(B, H, W) if self.precision == "bf16x3"
                 else _lib.load().evf_conv_wgrad_slabs(B, H, W))
        if FUSED_TAIL and self._finalize_fused(win, nslab):
            return [None] * len(self.params)
        if win.rows is not None:  # per-block partials of the per-channel gradients -> the small accumulator (rows zeroed)
            _lib.call("evf_sum_rows", _lib.ptr(win.rows), win.rows.shape[0], win.rows.shape[1], 1 | 2, _lib.ptr(win.small))
            self._rows_clean = True
        grads = []
        seg_src, seg_dst, seg_n = [], [], []
        red_src, red_dst = [], []
        for name, p in zip(self.pnames, self.params):
            if not p.requires_grad:
                grads.append(None)
                continue
            # a bound fp32 .grad while FlatAdam is in charge (hip_ops.DIRECT_PARAM_GRADS; its flat buffer): accumulate into it in our
            # kernels and hand autograd nothing -- saves one add kernel per parameter tensor per step
            direct = hip_ops.DIRECT_PARAM_GRADS and p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous() and p.grad.is_cuda
            if name in self.small_off:
                if name == "0.ff" and win.slab_init.get((0, "ff")):  # head weight gradient: per-block partials
                    sl = self._slabs[(0, "ff")]
                    _lib.call("evf_sum_rows", _lib.ptr(sl), sl.shape[0], sl.shape[1], 1, _lib.ptr(self._small(win, name)))
                if direct:
                    seg_src.append(self.small_off[name][0])
                    seg_dst.append(p.grad)
                    seg_n.append(self.small_off[name][1])
                    grads.append(None)
                else:
                    grads.append(self._small(win, name).view(p.shape).to(p.dtype))
                continue
            i, nm = name.split(".")
            k = (int(i), nm)
            if direct:
                if win.slab_init.get(k):
                    red_src.append(self._slabs[k])
                    red_dst.append(p.grad)
                grads.append(None)
                continue
            g = _lib.zeros(p.shape, dtype=torch.float32, device=win.dev)
            if win.slab_init.get(k):
                _lib.call("evf_reduce_slabs", _lib.ptr(self._slabs[k]), nslab, 9 * C * C, 0, _lib.ptr(g))
            grads.append(g.to(p.dtype))
        for lo in range(0, len(red_src), 16):  # all conv-weight slabs in one launch (16 tensors per call)
            n = min(16, len(red_src) - lo)
            _lib.call("evf_reduce_slabs_multi", (ctypes.c_void_p * n)(*[t.data_ptr() for t in red_src[lo:lo + n]]),
                      (ctypes.c_void_p * n)(*[t.data_ptr() for t in red_dst[lo:lo + n]]), n, nslab, 9 * C * C)
        for lo in range(0, len(seg_src), 32):  # all small gradients in one launch (32 segments per call)
            hi = min(lo + 32, len(seg_src))
            ptrs = (ctypes.c_void_p * 32)(*[t.data_ptr() for t in seg_dst[lo:hi]])
            offs = (ctypes.c_int * 32)(*seg_src[lo:hi])
            lens = (ctypes.c_int * 32)(*seg_n[lo:hi])
            _lib.call("evf_add_segments", _lib.ptr(win.small), ptrs, offs, lens, hi - lo, 1 if win.small_persistent else 0)
        if win.small_persistent:
            self._small_clean = True
        self._last_window = win
        return grads
