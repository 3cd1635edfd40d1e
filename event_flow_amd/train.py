"""Training-step glue on the MI355X -- counterpart of reference
train_flow.py:129-171: P forward passes, EventWarping over the window,
backward through all passes, global-norm clip + Adam, detach_states.

`FlatAdam` keeps the model parameters, their gradients and the Adam moments in
flat fp32 buffers (one all-reduce / one fused `evf_clip_adam_step` launch pair
per optimizer step instead of per-tensor torch kernels): every nn.Parameter
becomes a view into `flat_param`, every `.grad` a view into `flat_grad`.
"""

import torch

from . import _lib
from .dataloader.encodings import encode_event_list, encode_event_lists
from .models import hip_ops


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
        self.norm_ws = torch.zeros(2, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat_param[off : off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_param[off : off + k].view(p.shape)
            p.grad = self.flat_grad[off : off + k].view(p.shape)
            off += k
        self.lr, self.betas, self.eps, self.clip = lr, betas, eps, clip
        self.steps = 0
        hip_ops.DIRECT_PARAM_GRADS = True  # every .grad is a view of flat_grad: the backward kernels add into it in place

    def zero_grad(self):
        self.flat_grad.zero_()
        if any(p.grad is None for p in self.params):  # keep .grad bound to the flat buffer
            self._rebind()

    def _rebind(self):
        off = 0
        for p in self.params:
            k = p.numel()
            p.grad = self.flat_grad[off : off + k].view(p.shape)
            off += k

    def step(self):
        self.steps += 1
        _lib.call("evf_clip_adam_step", _lib.ptr(self.flat_param), _lib.ptr(self.flat_grad), _lib.ptr(self.m),
                  _lib.ptr(self.v), self.n, float(self.clip) if self.clip is not None else 0.0, float(self.lr),
                  float(self.betas[0]), float(self.betas[1]), float(self.eps), 0 if self.device_step else self.steps,
                  _lib.ptr(self.norm_ws))
        # the kernel rewrote the parameters behind torch's version counters: drop the
        # engine's packed-weight cache explicitly
        if hasattr(self.model, "invalidate_weight_cache"):
            self.model.invalidate_weight_cache()
        hip_ops.invalidate_packed_weights()

    def grad_norm(self):
        """L2 norm of the (pre-clip) gradient of the last step."""
        return float(self.norm_ws[0].sqrt())


def window_backward(model, loss_function, optimizer, passes, dp=None):
    """First half of a window: the passes, the loss and its backward (train_flow.py:129-154).
    Leaves this rank's gradient in the optimizer's flat buffer and, with `dp`, the
    local loss in the buffer's tail, ready for the all-reduce."""
    for k, d in enumerate(passes):
        if k == len(passes) - 1 and hasattr(model, "mark_last_pass"):
            model.mark_last_pass()  # lets a graph-captured step hand its final state over without a copy
        x = model(d["event_voxel"], d["event_cnt"])
        loss_function.event_flow_association(x["flow"], d["event_list"], d["event_list_pol_mask"], d["event_mask"])
    if loss_function.overwrite_intermediate:
        loss_function.overwrite_intermediate_flow(x["flow"])
    loss = loss_function()
    loss.backward()
    if dp is not None:
        dp.stage(optimizer.comm, loss)
    return loss


def window_apply(model, loss_function, optimizer, loss, dp=None):
    """Second half: clip + Adam on the (reduced) gradient, state detach, loss reset
    (train_flow.py:157-171).  Returns the 0-d (global) loss tensor."""
    if dp is not None:
        loss, _ = dp.staged(optimizer.comm)
    optimizer.step()
    optimizer.zero_grad()
    model.detach_states()
    loss_function.reset()
    return loss.detach().clone()


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


def encode_passes(event_lists, num_bins, res, want=("cnt", "mask", "voxel", "pol")):
    """[B,N,4] event lists (one per pass) -> the loader dicts, encoded on the GPU in one batched launch."""
    return encode_event_lists(event_lists, num_bins, res, want=want)
