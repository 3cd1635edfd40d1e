"""ctypes binding of libevflow_hip.so (the C ABI declared in include/evflow.h).

There is NO fallback: if the shared library is missing or a call fails the
product path raises.  PyTorch only provides device memory (`tensor.data_ptr()`)
and the current HIP stream.
"""

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EVF_LIB") or os.path.join(_HERE, "libevflow_hip.so")  # EVF_LIB: A/B builds of the library

P = ctypes.c_void_p
I = ctypes.c_int
F = ctypes.c_float
L = ctypes.c_int64

# name -> argument types (all functions return int)
SIGNATURES = {
    "evf_version": [],
    "evf_device_count": [],
    "evf_events_to_image": [P, P, P, I, I, I, I, P, P],
    "evf_encode_events": [P, I, I, I, I, I, I, P, P, P, P, P],
    "evf_encode_window": [P, I, I, I, I, I, I, I, I, P, P, P],
    "evf_iwe_splat": [P, P, P, P, P, P, I, I, I, I, I, F, F, F, I, I, P, P],
    "evf_get_interpolation": [P, P, I, I, I, I, F, F, I, P, P, P],
    "evf_interpolate": [P, P, P, I, I, I, I, I, P, P],
    "evf_cm_smooth_blocks": [I, I, I, I],
    "evf_norm_nonzero": [P, ctypes.c_int64, P, P, P],
    "evf_cm_loss_ws": [I, I, I, I, I],
    "evf_cm_loss_fwd": [P, P, P, P, P, I, I, I, I, I, I, F, F, I, P, P, P, P, P, P],
    "evf_cm_loss_bwd": [P, P, P, P, P, I, I, I, I, I, I, F, F, I, P, P, P, P, P, P],
    "evf_image_variance": [P, I, I, P, P],
    "evf_avg_ts_ratio": [P, I, I, F, P, P],
    "evf_aee": [P, P, P, P, I, I, I, F, P, P],
    "evf_mask_union": [P, I, I, I, I, P, P],
    "evf_masked_flow_mean": [P, P, I, I, I, I, P, P],
}

# network entry points (added as the kernels land)
NETWORK_SIGNATURES = {
    "evf_pack_conv_weight": [P, I, I, I, P, P],
    "evf_unpack_conv_wgrad": [P, I, I, I, P, P],
    "evf_head_lif_fwd": [P, P, P, P, P, P, I, I, I, I, I, P, P, P, P],
    "evf_conv_lif_fwd": [P, P, P, P, P, P, P, I, I, I, I, P, P, P, P],
    "evf_pack_conv_weight_b3": [P, I, I, P, P],
    "evf_conv_lif_fwd_b3": [P, P, P, P, P, P, P, I, I, I, I, P, P, P, P],
    "evf_conv_lif_fwd_b3_pred": [P, P, P, P, P, P, P, I, I, I, I, P, P, P, P, P, P, P],
    "evf_fwd_defer_begin": [P],
    "evf_fwd_defer_slot": [I, P],
    "evf_fwd_defer_pending": [P],
    "evf_fwd_defer_flush": [P],
    "evf_defer_poison": [I],
    "evf_bwd_defer_begin": [P],
    "evf_bwd_defer_slot": [I, P],
    "evf_bwd_defer_hold_heads": [I, P],
    "evf_bwd_defer_pending": [P],
    "evf_bwd_defer_flush": [P],
    "evf_comm_load": [ctypes.c_char_p],
    "evf_comm_last_error": [],
    "evf_comm_version": [P],
    "evf_comm_unique_id": [P],
    "evf_comm_init": [P, I, I, P],
    "evf_comm_count": [P, P],
    "evf_comm_destroy": [P],
    "evf_allreduce_sum": [P, P, L, P],
    "evf_allreduce_max": [P, P, L, P],
    "evf_defer_profile": [I],
    "evf_defer_profile_read": [P, P],
    "evf_lif_bwd": [P, P, P, P, P, P, P, I, I, I, I, I, F, P, P, P, P, P],
    "evf_lif_bwd_wgrad_slabs": [I, I, I],
    "evf_head_lif_bwd_wgrad_slabs": [I, I, I],
    "evf_head_lif_bwd_wgrad": [P, P, P, P, P, P, P, P, I, I, I, I, I, I, F, P, P, P, P, P, I, P],
    "evf_head_plif_bwd_wgrad": [P, P, P, P, P, P, P, P, I, I, I, I, I, I, F, P, P, P, P, I, P, P, P, P, P, P, P, P, P],
    "evf_sum_rows": [P, I, I, I, P, P],
    "evf_add_segments": [P, P, P, P, I, I, P],
    "evf_pack_conv_weights_b3_multi": [P, P, P, I, P],
    "evf_reduce_slabs_multi": [P, P, I, I, I, P],
    "evf_lif_bwd_wgrad": [P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, P, P, P, P, P, P, P, I, P],
    "evf_lif_bwd_wgrad2": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, P, P, P, P, P, P, P, I, P],
    "evf_lif_bwd_wgrad_top": [P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, P, P, P, P, P, P, I, P],
    "evf_debug_poison_lds": [ctypes.c_uint32, P],
    "evf_memset": [P, I, ctypes.c_size_t, P],
    "evf_plif_bwd_wgrad2": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, P, P, P, P, P, P, P, I, P, P, P, P, P, P, P, P, P, P],
    "evf_conv_dgrad_b3_multi": [I, P, P, P, P, P, I, I, I, P],
    "evf_conv_dgrad_b3_multi_fits": [I, I, I],
    "evf_lif_bwd_wgrad_window": [I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, F, P, P, P, P, I, P],
    "evf_plif_bwd_wgrad_window_top": [I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, F, P, P, P, P, P, P, P, I, P],
    "evf_plif_bwd_wgrad_window": [I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, F, P, P, P, P, P, P, P, I, P],
    "evf_plif_bwd_wgrad_top": [P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, P, P, P, P, P, P, I,
                               P, P, P, P, P, P, P, P, P, P],
    "evf_pack_conv_weight_b3t": [P, I, I, P, P],
    "evf_conv_dgrad_b3": [P, P, P, I, I, I, I, P, P, P],
    "evf_conv_dgrad_b3_f32": [P, P, P, I, I, I, I, P, P, P],
    "evf_conv_dgrad_select": [I],
    "evf_dgrad_diag_select": [I],
    "evf_bwd_diag_select": [I],
    "evf_fwd_diag_select": [I],
    "evf_conv_dgrad_b3_f32_pair": [P, P, P, I, P, P, I, I, I, P, P, P],
    "evf_conv_dgrad_b3_pair": [P, P, P, I, P, P, I, I, I, P, P, P],
    "evf_conv_plif_fwd_b3": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, P, P, P, P, P, P],
    "evf_conv_plif_fwd_b3_pred": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, P, P, P, P, P, P, P, P, P],
    "evf_weight_norm_fwd": [P, P, I, I, P, P, P],
    "evf_weight_norm_bwd": [P, P, P, P, I, I, P, P, P],
    "evf_head_plif_fwd": [P, P, P, P, P, P, P, P, P, I, I, I, I, I, P, P, P, P, P, P],
    "evf_plif_trace_bwd": [P, P, P, P, P, P, P, I, I, I, P, P, P, P, P, I, P],
    "evf_conv_dgrad": [P, P, P, I, P, P, I, I, I, I, P],
    "evf_conv_wgrad_bits": [P, P, I, I, I, P, I, P],
    "evf_conv_wgrad_slabs": [I, I, I],
    "evf_reduce_slabs": [P, I, I, I, P, P],
    "evf_head_wgrad": [P, P, I, I, I, I, P, P],
    "evf_pred_fwd": [P, P, P, I, I, I, P, P],
    "evf_pred_bwd": [P, P, P, P, I, I, I, P, P, P, P],
    "evf_bits_to_nchw": [P, I, I, I, P, P],
    "evf_bits_transpose": [P, I, I, I, P, P],
    "evf_nchw_to_bits": [P, I, I, I, P, P],
    "evf_nhwc_to_nchw": [P, I, I, I, I, P, P],
    "evf_nchw_to_nhwc": [P, I, I, I, I, P, P],
    "evf_clip_adam_step": [P, P, P, P, L, F, F, F, F, F, I, P, I, P],
    "evf_cm_merge": [I],
    "evf_cm_bwd_lds": [I],
    "evf_clip_adam_fused": [P, P, P, P, L, F, F, F, F, F, I, P, I, P],
    "evf_grads_finalize": [P, P, I, I, P, I, P, I, I, P, I, I, I, P, P, P, P, I, P],
    # general path (any channel count, NHWC fp32)
    "evf_conv2d_packed_size": [I, I, I, I],
    "evf_pack_conv2d_weight": [P, I, I, I, I, I, I, P, P],
    "evf_conv2d_fwd": [P, I, P, P, P, I, I, I, I, I, I, I, I, I, P],
    "evf_conv2d_dgrad": [P, I, P, P, I, I, I, I, I, I, I, I, I, P],
    "evf_conv2d_b3_packed_size": [I, I, I, I],
    "evf_pack_conv2d_weight_b3": [P, I, I, I, I, I, I, P, P],
    "evf_pack_conv2d_weights_b3_multi": [P, P, P, I, P],
    "evf_conv2d_b3_ws": [I, I, I, I],
    "evf_conv2d_fwd_b3": [P, I, P, P, P, I, I, I, I, I, I, I, I, I, P, L, P],
    "evf_conv2d_dgrad_b3": [P, I, P, P, I, I, I, I, I, I, I, I, I, P, L, P],
    "evf_conv_tile_select": [I],
    "evf_conv_split_select": [I],
    "evf_wgrad_teams_select": [I],
    "evf_conv2d_wgrad_ws": [I, I, I, I, I, I, I],
    "evf_conv2d_wgrad": [P, I, P, I, P, P, I, I, I, I, I, I, I, I, I, I, P, P],
    "evf_neuron_fwd": [I, P, P, P, P, P, P, P, P, P, P, L, I, I, P, P, P, P, P],
    "evf_conv2d_fwd_b3_parts": [P, I, P, P, I, I, I, I, I, I, I, I, I, P, L, P, P],
    "evf_lif_fwd_parts": [P, I, L, P, I, L, P, P, P, P, P, L, I, I, P, P, P, P],
    "evf_neuron_bwd": [I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, L, I, I, I, F, P, P, P, P, P, P, P, P, P, P, P],
    "evf_pretrace_fwd": [P, I, I, I, I, I, I, I, P, P, P],
    "evf_pretrace_bwd": [P, I, P, I, I, I, I, I, I, P, I, I, P],
    "evf_concat_channels": [P, P, P, I, L, P, I, P],
    "evf_concat_up2_fwd": [P, P, P, I, I, I, I, P, I, P],
    "evf_head1x1_ws": [I, I],
    "evf_head1x1_fwd": [P, I, P, P, I, L, I, I, P, I, P],
    "evf_head1x1_bwd": [P, I, P, I, P, L, P, I, L, I, I, P, I, P, P, I, P, P],
    "evf_upsample2x_fwd": [P, I, I, I, I, P, P],
    "evf_upsample2x_bwd": [P, I, I, I, I, P, P],
    "evf_upsample_nearest_fwd": [P, L, I, I, I, P, P],
    "evf_upsample_nearest_bwd": [P, L, I, I, I, P, P],
    "evf_apply_pixel_mask": [P, P, I, I, I, I, P],
    "evf_chan_reduce": [P, I, P, I, P, P, I, I, L, I, P, P],
    "evf_chan_affine": [P, I, P, I, P, P, P, I, L, I, P, I, P],
    "evf_act_fwd": [I, P, P, L, P, P],
    "evf_act_bwd": [I, P, P, L, P, P],
    "evf_leaky_fwd": [P, P, P, P, I, L, I, P, P, P],
    "evf_lstm_fwd": [P, P, L, I, P, P, P],
    "evf_lstm_bwd": [P, P, P, P, P, L, I, P, P, P],
    "evf_leaky_bwd": [P, P, P, P, P, I, L, I, P, P, P, P],
    "evf_spike_fwd": [P, P, I, L, P, P],
    "evf_spike_bwd": [I, P, P, I, P, F, L, P, P],
    "evf_gru_gates_fwd": [P, P, P, L, P, P, P, P],
    "evf_gru_out_fwd": [P, P, P, L, P, P, P],
    "evf_gru_out_bwd": [P, P, P, P, L, P, P, P, P],
    "evf_gru_gates_bwd": [P, P, P, L, P, P, P],
}
RESTYPES = {"evf_head1x1_ws": ctypes.c_int64, "evf_comm_last_error": ctypes.c_char_p, "evf_conv2d_packed_size": ctypes.c_int64, "evf_conv2d_b3_packed_size": ctypes.c_int64, "evf_conv2d_b3_ws": ctypes.c_int64, "evf_conv2d_wgrad_ws": ctypes.c_int64, "evf_cm_loss_ws": ctypes.c_int64}
SIGNATURES.update(NETWORK_SIGNATURES)

_lib = None


class EvflowError(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  Raises EvflowError when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EvflowError(
            f"{LIB_PATH} not found: build it with `python -m event_flow_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.argtypes = argtypes
        fn.restype = RESTYPES.get(name, ctypes.c_int)
    _lib = lib
    return lib


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise EvflowError("evflow kernels need contiguous tensors")
    return t.data_ptr()


def ptr_strided(t):
    """Device pointer of an NHWC activation whose channel dimension may be a slice of a wider tensor: everything but the
    pixel stride must be dense (the kernels take the pixel stride as `ld`)."""
    if t is None:
        return None
    if t.dim() != 4 or t.stride(3) != 1 or t.stride(1) != t.shape[2] * t.stride(2) or t.stride(0) != t.shape[1] * t.stride(1):
        raise EvflowError("evflow kernels need NHWC tensors that are dense up to the pixel stride")
    return t.data_ptr()


# optional per-entry-point timing with HIP events on the launch stream (bench.py)
_prof = None
_PROF_VARIANT = {
    "evf_conv_lif_fwd": lambda a: "rec" if a[2] is not None else "ff",
    "evf_conv_lif_fwd_b3": lambda a: "rec" if a[2] is not None else "ff",
    "evf_conv_lif_fwd_b3_pred": lambda a: "rec" if a[2] is not None else "ff",
    "evf_conv_dgrad": lambda a: "two" if a[4] is not None else "one",
    "evf_lif_bwd_wgrad": lambda a: "rec" if a[6] is not None else "ff",
    "evf_lif_bwd_wgrad2": lambda a: ("rec" if a[7] is not None else "ff") + ("+2" if a[1] is not None else ""),
    "evf_conv_dgrad_b3_f32": lambda a: "acc" if a[3] else "",
    "evf_conv_dgrad_b3_f32_pair": lambda a: "acc" if a[3] else "",
    # general convs: the shape is the variant ("B,H,W,Cin,Cout,k,stride"): bench.py derives the FLOP of every launch from it
    "evf_conv2d_fwd": lambda a: ",".join(str(int(v)) for v in a[6:13]),
    "evf_conv2d_dgrad": lambda a: ",".join(str(int(v)) for v in a[5:12]),
    "evf_conv2d_fwd_b3": lambda a: ",".join(str(int(v)) for v in a[6:13]),
    "evf_conv2d_fwd_b3_parts": lambda a: ",".join(str(int(v)) for v in a[5:12]),
    "evf_conv2d_dgrad_b3": lambda a: ",".join(str(int(v)) for v in a[5:12]),
    "evf_conv2d_wgrad": lambda a: ",".join(str(int(v)) for v in a[6:13]),
}


_prof_cal = None
last_event_overhead_ms = 0.0
last_tiny_kernel_ms = 0.0  # duration of one trivial launch (64-element add) from the same two-point fit: the launch floor


def profile_start(names):
    """Time every call of the named entry points with a pair of HIP events on the launch stream.  A bracket
    reads more than the kernel it encloses (the command processor handles the two timestamp packets a few us
    apart).  That overhead is calibrated here by a two-point fit -- brackets around one and around two launches
    of the same tiny kernel, T1 = o + t and T2 = o + 2t, so o = 2 T1 - T2 -- and subtracted in profile_stop, so
    the durations agree with rocprofv3's kernel-trace averages."""
    global _prof, _prof_cal
    _prof_cal = []
    dummy = torch.zeros(64, device="cuda")
    for i in range(48):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(1 + (i & 1)):
            dummy.add_(1.0)
        e1.record()
        _prof_cal.append((1 + (i & 1), e0, e1))
    _prof = {n: [] for n in names}


def profile_stop():
    """-> {(name, variant): [ms, ...]} (empty-bracket overhead removed); synchronises."""
    global _prof, _prof_cal, last_event_overhead_ms, last_tiny_kernel_ms
    rec, _prof = _prof, None
    cal, _prof_cal = _prof_cal, None
    torch.cuda.synchronize()
    ovh = 0.0
    if cal:
        med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
        t1 = med([e0.elapsed_time(e1) for k, e0, e1 in cal[8:] if k == 1])  # first brackets: warm-up
        t2 = med([e0.elapsed_time(e1) for k, e0, e1 in cal[8:] if k == 2])
        ovh = max(2.0 * t1 - t2, 0.0)
        last_tiny_kernel_ms = max(t2 - t1, 0.0)
    last_event_overhead_ms = ovh
    out = {}
    for name, lst in (rec or {}).items():
        for variant, e0, e1 in lst:
            out.setdefault((name, variant), []).append(max(e0.elapsed_time(e1) - ovh, 0.0))
    return out


# Deferred forward cells (evf_fwd_defer_*, models/engine.py): while a recording is open, any OTHER entry point may read what
# the recorded cells write, so it launches them first.  The names below never do: they record or launch a cell themselves,
# or only produce network inputs.
# The recorders of the library are per STREAM (include/evflow.h, "CONTEXTS / THREADS"), and so is the hook that launches what
# has been recorded on a stream before any entry point outside the recorded schedule runs on it: _hooks[stream] = (callable
# set by the engine that opened the recording, the entry points that record themselves or flush inside the library).
# The library allows a forward AND a backward recording open on one stream at once, so a stream holds one hook per kind:
# _hooks[stream] = {"fwd": (flush, safe), "bwd": (flush, safe)}.
_hooks = {}


def set_defer_hook(flush, safe, kind):
    _hooks.setdefault(stream_ptr(), {})[kind] = (flush, safe)


def clear_defer_hook(kind):
    sp = stream_ptr()
    d = _hooks.get(sp)
    if d is not None:
        d.pop(kind, None)
        if not d:
            _hooks.pop(sp, None)


def raw(name, *args):
    """Status of an entry point called on torch's current stream (no hook, no exception): the evf_*_defer_* bookkeeping calls."""
    return getattr(load(), name)(*args, stream_ptr())

_DEFER_SAFE_FWD = {"evf_conv_lif_fwd_b3", "evf_conv_lif_fwd_b3_pred", "evf_head_lif_fwd", "evf_fwd_defer_flush",
                   "evf_conv_plif_fwd_b3", "evf_conv_plif_fwd_b3_pred",
                   "evf_head_plif_fwd",  # (recorded like evf_head_lif_fwd: the window's passes in one launch)
                   "evf_encode_window", "evf_encode_events", "evf_events_to_image"}
# backward recording (evf_bwd_defer_*): these record themselves, or flush inside the library when they cannot
_DEFER_SAFE_BWD = {"evf_lif_bwd_wgrad", "evf_lif_bwd_wgrad2", "evf_lif_bwd_wgrad_top", "evf_plif_bwd_wgrad2", "evf_plif_bwd_wgrad_top", "evf_conv_dgrad_b3_f32",
                   "evf_conv_dgrad_b3_f32_pair", "evf_conv_dgrad_b3", "evf_conv_dgrad_b3_pair", "evf_head_lif_bwd_wgrad", "evf_head_plif_bwd_wgrad",
                   "evf_bwd_defer_flush"}


def zero_(t):
    """t.zero_() as a KERNEL launch (evf_memset): torch's zero_() / torch.zeros are hipMemsetAsync, i.e. memset nodes inside a
    captured step, and graphs with memset nodes between kernel nodes replayed corrupted after a device synchronize on ROCm 7.2
    (include/evflow.h, evf_memset).  Dense CUDA tensors; anything else through torch."""
    if t.is_cuda and t.is_contiguous() and t.numel():
        rc = load().evf_memset(t.data_ptr(), 0, t.numel() * t.element_size(), stream_ptr())
        if rc != 0:
            raise EvflowError(f"evf_memset failed with status {rc}")
        return t
    return t.zero_()


def zeros(shape, dtype=torch.float32, device=None):
    """torch.zeros through zero_() above."""
    t = torch.empty(shape, dtype=dtype, device=device)
    return zero_(t)


_POISON_LDS = os.environ.get("EVF_DEBUG_POISON_LDS", "0") == "1"  # debugging: NaN in every CU's LDS before every entry point


def call(name, *args):
    """Invoke an entry point on torch's current stream; raise on error."""
    hooks = _hooks.get(stream_ptr()) if _hooks else None
    if hooks:
        for flush, safe in list(hooks.values()):
            if name not in safe:
                flush()
    if _POISON_LDS:
        load().evf_debug_poison_lds(0, stream_ptr())
    if _prof is not None and name in _prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(load(), name)(*args, stream_ptr())
        e1.record()
        _prof[name].append((_PROF_VARIANT.get(name, lambda a: "")(args), e0, e1))
    else:
        rc = getattr(load(), name)(*args, stream_ptr())
    if rc != 0:
        raise EvflowError(f"{name} failed with status {rc}")


def require_gpu(t, what):
    if not t.is_cuda:
        raise EvflowError(f"{what}: tensor is on {t.device}; the evflow path runs on the MI355X only (no CPU fallback)")
    if t.device.index != torch.cuda.current_device():
        # launches go to the CURRENT device's stream: a tensor on another GPU would be touched from the wrong context
        raise EvflowError(f"{what}: tensor is on {t.device} but the current device is cuda:{torch.cuda.current_device()}; "
                          "call torch.cuda.set_device(tensor.device) first (one process per GPU)")
    if t.dtype != torch.float32 and t.dtype != torch.int32:
        raise EvflowError(f"{what}: unsupported dtype {t.dtype}")
