"""Training-step glue on the MI355X -- counterpart of reference
train_flow.py:129-171: P forward passes, EventWarping over the window,
backward through all passes, global-norm clip + Adam, detach_states.

`FlatAdam` keeps the model parameters, their gradients and the Adam moments in
flat fp32 buffers (one all-reduce / one fused `evf_clip_adam_step` launch pair
per optimizer step instead of per-tensor torch kernels): every nn.Parameter
becomes a view into `flat_param`, every `.grad` a view into `flat_grad`.
"""

import os
import weakref

import torch

from . import _lib
from .dataloader.encodings import encode_event_list, encode_event_lists
from .models import hip_ops


FUSED_ADAM = os.environ.get("EVF_FUSED_ADAM", "1") != "0"


class FlatAdam:
    """clip_grad_norm_(max_norm) + Adam(lr, betas, eps) fused on flat buffers.
    Reference semantics: train_flow.py:157-163 with torch.optim.Adam defaults."""

    def __init__(self, model, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, clip=100.0, device_step=False):
        self.model = model
        self.device_step = device_step  # keep the step counter on the GPU (hipGraph replay)
        self.params = [p for p in model.parameters() if p.requires_grad]
        dev = self.params[0].device
        _lib.require_gpu(self.params[0], "FlatAdam")
        n = sum(p.numel() for p in self.params)
        self.n = n
        self.flat_param = torch.empty(n, dtype=torch.float32, device=dev)
        # gradient buffer + 2 tail floats (loss, new_seq flag): the unit of the DP all-reduce
        self.comm = torch.zeros(n + 2, dtype=torch.float32, device=dev)
        self.flat_grad = self.comm[:n]
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        # [0] squared gradient norm of the last step, [1] device-side step counter, [2..4] the fused kernel's running sum / tickets
        self.norm_ws = torch.zeros(8, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat_param[off : off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_param[off : off + k].view(p.shape)
            p.grad = self.flat_grad[off : off + k].view(p.shape)
            off += k
        self.lr, self.betas, self.eps, self.clip = lr, betas, eps, clip
        self.steps = 0
        # True between step() (which CLEARS the gradient buffer as it consumes it: `.grad` reads zeros after step()) and the next
        # write into the buffer.  Writers announce themselves: autograd's AccumulateGrad through the per-parameter hook below, our
        # kernels that add into the bound .grad directly through hip_ops.direct_grads_written() -- so a `loss.backward()` issued
        # anywhere between step() and zero_grad() (gradient accumulation, custom loops) is seen and zero_grad() then does clear.
        self._grad_clean = False
        # (weak: a hook that held the optimizer would keep it and its flat buffers alive with the model, and a second
        # FlatAdam on the same model would stack hooks; close() removes them)
        wself = weakref.ref(self)

        def _dirty(_p, _w=wself):
            o = _w()
            if o is not None:
                o.mark_grad_dirty()

        self._hook_handles = [p.register_post_accumulate_grad_hook(_dirty) for p in self.params]
        hip_ops.on_direct_grads(self)
        hip_ops.DIRECT_PARAM_GRADS = True  # every .grad is a view of flat_grad: the backward kernels add into it in place

    def close(self):
        """Detach from the model: remove the per-parameter hooks (the parameters keep their storage in the flat buffer)."""
        for h in self.__dict__.get("_hook_handles", []):
            h.remove()
        self._hook_handles = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def zero_grad(self):
        # the fused step clears the buffer as it consumes it: zero_grad() right after step() (train_flow.py:163-164)
        # needs no fill kernel.  A loop that runs a backward between step() and zero_grad() calls mark_grad_dirty().
        if not self._grad_clean:
            _lib.zero_(self.flat_grad)
        self._grad_clean = False
        if any(p.grad is None for p in self.params):  # keep .grad bound to the flat buffer
            self._rebind()

    def mark_grad_dirty(self):
        """A backward pass is about to add into the flat gradient buffer."""
        self._grad_clean = False

    def _rebind(self):
        off = 0
        for p in self.params:
            k = p.numel()
            p.grad = self.flat_grad[off : off + k].view(p.shape)
            off += k

    def step(self):
        self.steps += 1
        # one launch (squared norm, grid hand-shake, clip + Adam + zero_grad; csrc/evf_step_tail.hip); EVF_FUSED_ADAM=0: fill + two
        _lib.call("evf_clip_adam_fused" if FUSED_ADAM else "evf_clip_adam_step", _lib.ptr(self.flat_param), _lib.ptr(self.flat_grad), _lib.ptr(self.m),
                  _lib.ptr(self.v), self.n, float(self.clip) if self.clip is not None else 0.0, float(self.lr),
                  float(self.betas[0]), float(self.betas[1]), float(self.eps), 0 if self.device_step else self.steps,
                  _lib.ptr(self.norm_ws), 1)
        self._grad_clean = True
        # the kernel rewrote the parameters behind torch's version counters: drop the
        # engine's packed-weight cache explicitly
        for m in [self.model] + list(getattr(self, "extra_models", [])):  # (+ the stream replicas sharing these weights)
            if hasattr(m, "invalidate_weight_cache"):
                m.invalidate_weight_cache()
        hip_ops.invalidate_packed_weights()
        hip_ops.repack_all()  # (general path: all packed operands in one launch instead of two per layer, lazily)

    def step_invalidate(self):
        """The flat parameter buffer was rewritten from outside (broadcast, checkpoint): drop packed-weight caches."""
        if hasattr(self.model, "invalidate_weight_cache"):
            self.model.invalidate_weight_cache()
        hip_ops.invalidate_packed_weights()

    def grad_norm(self):
        """L2 norm of the (pre-clip) gradient of the last step."""
        return float(self.norm_ws[0].sqrt())


# EVF_DEFER_FWD=0: every (pass, layer) cell of the fused FireNets as its own launch (the A/B switch of the diagonal launches)
DEFER_FORWARD = os.environ.get("EVF_DEFER_FWD", "1") != "0"
DEFER_BACKWARD = os.environ.get("EVF_DEFER_BWD", "1") != "0"  # ... and of the backward cells
_UNIT = {}


def _unit_gradient(loss):
    """Cached `ones_like(loss)` (autograd would fill a fresh one per step).  Never created inside a graph capture: a tensor
    allocated there belongs to the capture's memory pool and must not be handed to later eager steps."""
    key = (loss.device, loss.dtype, tuple(loss.shape))
    if key not in _UNIT:
        if loss.is_cuda and torch.cuda.is_current_stream_capturing():
            return torch.ones_like(loss)  # (this capture's own tensor; the cache is filled by the first eager step)
        _UNIT[key] = torch.ones_like(loss)
    return _UNIT[key]


def window_backward(model, loss_function, optimizer, passes, dp=None):
    """First half of a window: the passes, the loss and its backward (train_flow.py:129-154).
    Leaves this rank's gradient in the optimizer's flat buffer and, with `dp`, the
    local loss in the buffer's tail, ready for the all-reduce."""
    defer = DEFER_FORWARD and hasattr(model, "defer_forward")
    if defer:
        model.defer_forward(True)  # fused FireNets: the window's hidden cells run diagonal by diagonal (engine.defer_forward)
    try:
        for k, d in enumerate(passes):
            if k == len(passes) - 1 and hasattr(model, "mark_last_pass"):
                model.mark_last_pass()  # lets a graph-captured step hand its final state over without a copy
            x = model(d["event_voxel"], d["event_cnt"])
            loss_function.event_flow_association(x["flow"], d["event_list"], d["event_list_pol_mask"], d["event_mask"])
    finally:
        if defer:
            model.defer_forward(False)  # (launches what was recorded)
    if loss_function.overwrite_intermediate:
        loss_function.overwrite_intermediate_flow(x["flow"])
    loss = loss_function()
    if hasattr(optimizer, "mark_grad_dirty"):
        optimizer.mark_grad_dirty()
    bdefer = DEFER_BACKWARD and hasattr(model, "defer_backward")
    if bdefer:
        model.defer_backward(True)  # fused FireNets: the backward cells of all passes run diagonal by diagonal
    try:
        loss.backward(_unit_gradient(loss))  # (autograd would fill a fresh ones_like(loss) per step)
    finally:
        if bdefer:
            model.defer_backward(False)
    if dp is not None and dp.active:
        dp.stage(optimizer.comm, loss)
    return loss


def window_forward_loss(model, loss_function, passes):
    """The passes of a window and its loss WITHOUT a backward pass and without an optimizer step (BASELINE.json configs[1]:
    forward + IWE loss; eval-style timing).  Runs in grad mode so that the hidden cells take the recorded diagonal launches
    like a training window; the window is dropped afterwards (states carried, graph discarded).  Returns the 0-d loss."""
    if not passes:
        raise _lib.EvflowError("window_forward_loss: an empty window (no passes)")
    defer = DEFER_FORWARD and hasattr(model, "defer_forward")
    if defer:
        model.defer_forward(True)
    try:
        for d in passes:
            x = model(d["event_voxel"], d["event_cnt"])
            loss_function.event_flow_association(x["flow"], d["event_list"], d["event_list_pol_mask"], d["event_mask"])
    finally:
        if defer:
            model.defer_forward(False)
    if loss_function.overwrite_intermediate:
        loss_function.overwrite_intermediate_flow(x["flow"])
    loss = loss_function().detach()
    model.detach_states()
    loss_function.reset()
    return loss


def window_apply(model, loss_function, optimizer, loss, dp=None):
    """Second half: clip + Adam on the (reduced) gradient, state detach, loss reset
    (train_flow.py:157-171).  Returns the 0-d (global) loss tensor."""
    staged = dp is not None and dp.active
    if staged:
        loss, _ = dp.staged(optimizer.comm)
    optimizer.step()
    optimizer.zero_grad()
    model.detach_states()
    loss_function.reset()
    # (the staged loss is a view of the communication buffer, rewritten by the next step: hand out a copy; the local
    # loss is a tensor of its own)
    return loss.detach().clone() if staged else loss.detach()


def train_window(model, loss_function, optimizer, passes, dp=None):
    """One truncated-BPTT window = one optimizer step (train_flow.py:129-171).
    `passes`: list of dicts with event_cnt, event_voxel, event_list,
    event_list_pol_mask, event_mask (GPU tensors).  Returns the 0-d loss tensor
    (no host sync).  With `dp` the flat gradient (+ loss) is SUM all-reduced over
    the ranks between the two halves (the loss sums over the batch)."""
    loss = window_backward(model, loss_function, optimizer, passes, dp)
    if dp is not None:
        dp.reduce(optimizer.comm)
    return window_apply(model, loss_function, optimizer, loss, dp)


class StreamReplicas:
    """Micro-batch pipelining inside ONE GPU: the batch is cut into `n` slices and each slice runs its window (passes,
    loss, backward) through its own replica of the model on its own HIP stream.  The replicas share the parameter
    storage (one FlatAdam) and own their recurrent state, BPTT tape, packed weights and gradient buffer; the slices'
    gradients (and losses) are summed into the optimizer's buffer before the ONE optimizer step.  Same result as the
    unsplit step up to summation order -- the loss sums over the batch (train_flow.py:141-154), exactly the argument of
    the data-parallel ranks.  Why: every kernel of the recurrent network is one short round of blocks (load, matrix phase,
    store, ~5 us of ramp / drain per launch); two half-sized kernels of different layers in flight fill those gaps
    (measured: 1490 -> 1730 windows/s at 8 x 128x128).

        reps = StreamReplicas(model, loss_function, optimizer, n=2)      # before the first forward pass
        loss = reps.train_window(passes_per_slice)                        # list of n pass lists
    """

    def __init__(self, model, loss_function, optimizer, n=2):
        import copy

        if not isinstance(optimizer, FlatAdam):
            raise _lib.EvflowError("StreamReplicas needs FlatAdam (one flat gradient buffer per replica)")
        self.n = int(n)
        self.opt = optimizer
        self.models, self.lossfs, self.streams, self.grads = [model], [loss_function], [None], [optimizer.flat_grad]
        dev = optimizer.flat_param.device
        for _ in range(1, self.n):
            eng = getattr(model, "_engine", None)
            if eng is not None:
                model._engine = None  # (built lazily; never copied)
            shadow = copy.deepcopy(model)
            if eng is not None:
                model._engine = eng
            g = _lib.zeros(optimizer.n, dtype=torch.float32, device=dev)
            off = 0
            for p, q in zip([p for p in model.parameters() if p.requires_grad], [q for q in shadow.parameters() if q.requires_grad]):
                k = p.numel()
                q.data = p.data  # the SAME storage: one set of weights
                q.grad = g[off:off + k].view(q.shape)
                off += k
            shadow.train(model.training)
            self.models.append(shadow)
            self.lossfs.append(copy.deepcopy(loss_function))
            self.streams.append(torch.cuda.Stream(device=dev))
            self.grads.append(g)
        optimizer.extra_models = self.models[1:]

    def stream(self, k):
        return self.streams[k] if self.streams[k] is not None else torch.cuda.current_stream()

    def backward_slice(self, k, passes):
        """Window of slice k on the CURRENT stream -> its 0-d loss; gradient left in grads[k]."""
        return window_backward(self.models[k], self.lossfs[k], self.opt, passes, dp=None)

    def combine(self, losses):
        """Sum the shadows' gradients (cleared on the way) and losses into slice 0's; on the current stream, after it
        has waited for the slice streams."""
        import ctypes

        for k in range(1, self.n):
            ptrs = (ctypes.c_void_p * 32)(self.grads[0].data_ptr())
            _lib.call("evf_add_segments", _lib.ptr(self.grads[k]), ptrs, (ctypes.c_int * 32)(0), (ctypes.c_int * 32)(self.opt.n), 1, 1)
            ptrs = (ctypes.c_void_p * 32)(losses[0].data_ptr())
            _lib.call("evf_add_segments", _lib.ptr(losses[k].detach().view(1)), ptrs, (ctypes.c_int * 32)(0), (ctypes.c_int * 32)(1), 1, 0)
        return losses[0]

    def apply(self, loss, dp=None):
        """clip + Adam on the summed gradient, detach / reset of every replica (window_apply)."""
        out = window_apply(self.models[0], self.lossfs[0], self.opt, loss, dp)
        for m, lf in zip(self.models[1:], self.lossfs[1:]):
            m.detach_states()
            lf.reset()
        return out

    def train_window(self, passes_per_slice, dp=None):
        """The whole step, eagerly (replicas on their streams, fork / join around them)."""
        main = torch.cuda.current_stream()
        losses = [None] * self.n
        for k in range(1, self.n):
            self.streams[k].wait_stream(main)
        for k in range(self.n):
            if k == 0:
                losses[0] = self.backward_slice(0, passes_per_slice[0])
            else:
                with torch.cuda.stream(self.streams[k]):
                    losses[k] = self.backward_slice(k, passes_per_slice[k])
        for k in range(1, self.n):
            main.wait_stream(self.streams[k])
        loss = self.combine(losses)
        if dp is not None and dp.active:
            dp.stage(self.opt.comm, loss)
            dp.reduce(self.opt.comm)
        return self.apply(loss, dp)


def encode_passes(event_lists, num_bins, res, want=("cnt", "mask", "voxel", "pol")):
    """[B,N,4] event lists (one per pass) -> the loader dicts, encoded on the GPU in one batched launch."""
    return encode_event_lists(event_lists, num_bins, res, want=want)


class _CellListStates:
    """`.states` of a FireNet-family network on the general path (models.model.FireNet._states: one entry per cell)."""

    def __init__(self, model):
        self.model = model

    @property
    def states(self):
        return self.model._states

    @states.setter
    def states(self, value):
        self.model._states = list(value)


def _general_states(model):
    """(holder, flat list of the state tensors) of a general-path recurrent model: the *EVFlowNet / E2VID families (models.model: a
    `multires_unetrec` / `unetrecurrent` whose `.states` is a list of tensors, tuples of tensors or None) and the FireNet family
    when it is chained cell by cell (ANN / ALIF / XLIF / residual / other widths: `model._states`)."""
    holder = getattr(model, "multires_unetrec", None) or getattr(model, "unetrecurrent", None)
    if holder is None and hasattr(model, "_states") and hasattr(model, "_fused") and not model._fused():
        holder = _CellListStates(model)
    if holder is None or not hasattr(holder, "states"):
        raise _lib.EvflowError("capture_window_cycle needs a general-path recurrent model (multires_unetrec / unetrecurrent states, or "
                               "a FireNet-family network off the fused engine); the fused LIF / PLIF FireNets replay through "
                               "GraphedWindowStep")
    flat = []
    for st in holder.states:
        if st is not None:
            flat += list(st) if isinstance(st, (tuple, list)) else [st]
    return holder, flat


def capture_window_cycle(model, loss_function, optimizer, windows, stream, capture_error_mode="global", route=True):
    """The training step of each window of `windows` (lists of encoded passes, fixed shapes) of a GENERAL-path recurrent model
    as a hipGraph; the graphs replay in order, again and again.  The warm-up must have run eagerly on `stream` (the model
    holds a state).  The recurrent state crosses replays WITHOUT copies: graph 0 reads the tensors the warm-up left (`home`),
    graph k the tensors graph k - 1 wrote (fixed addresses in its pool), and the cells of the LAST graph write their new states
    straight into `home` (hip_ops.route_states) -- a state tensor a cell could not be routed for (another shape or layout) is
    copied at the end of the last graph instead.  Returns ([(graph, loss tensor)], number of state tensors copied)."""
    from .models import hip_ops

    if not getattr(optimizer, "device_step", False):
        raise _lib.EvflowError("capture_window_cycle needs FlatAdam(..., device_step=True)")
    holder, home = _general_states(model)
    home_struct = list(holder.states)
    graphs, ncopied = [], 0
    try:
        for w, passes in enumerate(windows):
            last = w == len(windows) - 1
            # (route=False: every state tensor is copied at the end of the last graph -- A/B, tests.  A cycle of ONE window
            # starts from `home` itself: its cells would overwrite the previous state the backward pass still reads, so it
            # is copied as well; route_states refuses overlapping pairs on its own.)
            if last and route and len(windows) > 1:
                hip_ops.route_states(_general_states(model)[1], home)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream, capture_error_mode=capture_error_mode):
                loss = train_window(model, loss_function, optimizer, passes)
                if last:
                    left = _general_states(model)[1]
                    if len(left) != len(home):
                        raise _lib.EvflowError("the window changed the structure of the recurrent state")
                    for h, st in zip(home, left):
                        if st.data_ptr() != h.data_ptr():
                            h.copy_(st)
                            ncopied += 1
            graphs.append((g, loss))
    finally:
        hip_ops.clear_state_routes()
    holder.states = home_struct
    return graphs, ncopied


class GraphedWindowStep:
    """`train_window` for windows of a FIXED shape (P passes of [B,N,4] events) replayed from hipGraphs: one graph
    launch per optimizer step instead of ~260 kernel launches (the eager step is host bound at this size).

    New windows are copied into a static event buffer [B,P,N,4] (P*B*N*16 bytes) before each replay.  Two graphs are captured
    and replayed alternately so that the recurrent state crosses replays without copies (graph A starts from the
    buffers the warm-up left and ends in its own pool tensors, graph B starts from those and writes its last pass
    straight back; see FireNet.final_states_into).  With several ranks (`dp`) each step is two graphs around the one
    eager all-reduce.  Needs `FlatAdam(..., device_step=True)` (the Adam step counter must live on the device) and a
    model with the fused FireNet engine; the first `warmup` windows passed to `step` run eagerly.

        stepper = GraphedWindowStep(model, loss_function, optimizer, num_bins, resolution)
        for event_lists in windows:            # list of P tensors [B,N,4] (t, y, x, p)
            loss = stepper.step(event_lists)   # 0-d tensor, no host sync
    """

    def __init__(self, model, loss_function, optimizer, num_bins, res, dp=None, want=("cnt", "mask", "voxel", "pol"), warmup=2):
        if not getattr(optimizer, "device_step", False):
            raise _lib.EvflowError("GraphedWindowStep needs FlatAdam(..., device_step=True)")
        for name in ("use_static_states", "state_buffers", "final_states_into"):
            if not hasattr(model, name):
                raise _lib.EvflowError("GraphedWindowStep needs a model with the fused FireNet engine")
        self.model, self.lossf, self.opt, self.dp = model, loss_function, optimizer, dp
        self.num_bins, self.res, self.want = num_bins, res, want
        self.warmup, self.seen = max(int(warmup), 2), 0
        self.static_ev = None
        self.graphs = None
        self.stream = torch.cuda.Stream()
        model.use_static_states(True)

    def _passes(self):
        d = encode_event_lists(list(self.static_ev.unbind(1)), self.num_bins, self.res, want=self.want)
        for p in d:
            p.setdefault("event_voxel", None)
            p.setdefault("event_cnt", None)
        return d

    def _capture(self):
        mode = "thread_local" if (self.dp is not None and self.dp.active) else "global"
        home = self.model.state_buffers()
        self.model.use_static_states(False)
        self.graphs = []
        for gi in range(2):
            if gi == 1:
                self.model.final_states_into(home)
            pre, post = torch.cuda.CUDAGraph(), None
            if self.dp is None or not self.dp.active:
                with torch.cuda.graph(pre, stream=self.stream, capture_error_mode=mode):
                    loss = train_window(self.model, self.lossf, self.opt, self._passes(), dp=self.dp)
            else:
                post = torch.cuda.CUDAGraph()
                with torch.cuda.graph(pre, stream=self.stream, capture_error_mode=mode):
                    local = window_backward(self.model, self.lossf, self.opt, self._passes(), self.dp)
                with torch.cuda.graph(post, stream=self.stream, capture_error_mode=mode):
                    loss = window_apply(self.model, self.lossf, self.opt, local, self.dp)
            # where this graph leaves the recurrent state (graph 0: tensors of its own pool; graph 1: `home`)
            self.graphs.append((pre, post, loss, self.model.state_buffers()))

    def step(self, event_lists):
        # static window buffer [B,P,N,4] (batch-major: the binning kernel and the loss read the window in place)
        ev = torch.stack([e.to(torch.float32) for e in event_lists], 1)
        if self.static_ev is None:
            self.static_ev = torch.empty_like(ev)
        elif ev.shape != self.static_ev.shape:
            raise _lib.EvflowError(f"window shape changed: {tuple(ev.shape)} vs {tuple(self.static_ev.shape)}")
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.static_ev.copy_(ev)
            if self.seen < self.warmup:  # eager steps (they also fix the streams autograd will use)
                loss = train_window(self.model, self.lossf, self.opt, self._passes(), dp=self.dp)
            else:
                if self.graphs is None:
                    torch.cuda.synchronize()
                    self._capture()
                pre, post, loss, left = self.graphs[(self.seen - self.warmup) & 1]
                pre.replay()
                if post is not None:
                    self.dp.reduce(self.opt.comm)
                    post.replay()
                self.model.set_state_buffers(left)  # keep model.states / eager calls consistent with the replayed step
            self.seen += 1
        cur.wait_stream(self.stream)
        return loss
