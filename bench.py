#!/usr/bin/env python
"""Benchmark of the event_flow hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    N > 1 without a launcher (WORLD_SIZE unset): bench.py starts its own N ranks, one per GPU, by
    re-executing itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free port>`; launched under torch.distributed.run (or
    torchrun) by the caller it uses the ranks it was given.

Metric (BASELINE.json): event-windows/s of the LIF-FireNet train step at
128x128 with 15k events per window (10 passes x 1500 events, truncated BPTT
over the window, contrast-maximisation loss, clip + Adam), 8 windows per GPU,
pure data parallel (one RCCL all-reduce of the flat gradient per step).  One
"step" = one window per batch slot = one optimizer step.  Inputs (raw event
lists) are resident in HBM before the timed region; the timed region contains
the event binning, every forward pass, the loss, the whole backward, the
all-reduce and the optimizer -- nothing is skipped or cached.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel: algorithmic bytes per launch / mean launch
                duration against the 8 TB/s HBM peak (the bf16x3 kernels are
                HBM bound; --precision fp32: FLOP against the fp32-MFMA peak).
                Durations: HIP events captured INTO the replayed hipGraph
                (timestamp-kernel nodes around the diagonal launches of a
                second, instrumented capture), i.e. the kernels as they run
                inside the replayed step; the eager HIP-event figure is kept
                beside it (`frac_eager`)
  kernels       the same for every conv entry point
  iwe_warp      compute_pol_iwe (integer IWE) GB/s at the spec shape and at a
                bandwidth-saturating batch, against the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (PyTorch-CPU port of the reference path) timed
                on this box's host cores on a bounded sample
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

H = W = 128
B_PER_GPU = 8
PASSES = 10
EV_PER_PASS = 1500
FP32_MFMA_PEAK = 157.3  # TFLOP/s dense, MI355X_MICROARCH.md
BF16_MFMA_PEAK = 2500.0  # TFLOP/s dense (v_mfma_f32_32x32x16_bf16), MI355X_MICROARCH.md
HBM_PEAK = 8000.0  # GB/s spec
ATOMIC_RATE = 21.4e9  # random device-scope fp32 atomics per second, measured (tools/probes/atomic_probe.hip; u32: 27.4e9)

MODEL_CFG = {
    "name": "LIFFireNet", "encoding": "cnt", "round_encoding": False, "norm_input": False, "num_bins": 2,
    "base_num_channels": 32, "kernel_size": 3, "activations": ["arctanspike", "arctanspike"], "mask_output": True,
    "spiking_neuron": {"leak": [-4.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True, "learn_thresh": True, "hard_reset": True},
}
LOSS_CFG = {"loader": {"resolution": [H, W]}, "loss": {"flow_regul_weight": 0.001, "overwrite_intermediate": False, "clip_grad": 100.0},
            "model": {"mask_output": True}}

CONV_FLOP = 2 * 9 * 32 * 32  # per pixel per 32->32 3x3 conv

PLIF_NEURON = {"leak_v": [-4.0, 0.1], "leak_pt": [-4.0, 0.1], "add_pt": [-2.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True,
               "learn_thresh": True, "hard_reset": True}
WORKLOADS = {
    # BASELINE.json configs[2] (headline; per-GPU shard = configs[1] + backward): the default
    "c3": {"model": "LIFFireNet", "H": 128, "W": 128, "B": 8, "neuron": None,
           "text": "LIF-FireNet full train step (BPTT over 10 passes x 1500 events = 15k events/window, 128x128, CM loss, clip+Adam), "
                   "8 windows per GPU [BASELINE configs[2] per-GPU shard; superset of configs[1]]"},
    # BASELINE.json configs[4]: PLIF-FireNet on MVSEC-shaped windows, batch 32 over 8 GPUs = 4 per GPU
    "c5": {"model": "PLIFFireNet", "H": 260, "W": 346, "B": 4, "neuron": PLIF_NEURON,
           "text": "PLIF-FireNet full train step (BPTT over 10 passes x 1500 events, 260x346, CM loss, clip+Adam), 4 windows per GPU "
                   "[BASELINE configs[4] per-GPU shard]"},
}


CONFIG_ID = 3


def set_workload(name):
    """Point the module-level workload constants at one of WORKLOADS (c3 is the import-time default)."""
    global H, W, B_PER_GPU, MODEL_CFG, LOSS_CFG, CONFIG_ID
    wl = WORKLOADS[name]
    CONFIG_ID = int(name[1:])
    H, W, B_PER_GPU = wl["H"], wl["W"], wl["B"]
    MODEL_CFG = dict(MODEL_CFG, name=wl["model"])
    if wl["neuron"] is not None:
        MODEL_CFG["spiking_neuron"] = dict(wl["neuron"])
    LOSS_CFG = {"loader": {"resolution": [H, W]}, "loss": dict(LOSS_CFG["loss"]), "model": {"mask_output": True}}
    return wl


def source_hash():
    """sha1 over the kernel sources: PMC numbers measured on other sources are stale and are not reported."""
    import glob
    import hashlib

    h = hashlib.sha1()
    d = os.path.join(ROOT, "event_flow_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def make_windows(rank, n_pool, dev, slices=1, kind="uniform"):
    """n_pool windows, each PASSES event lists [B,1500,4] (platform-stable synthetic events; kind: synthetic.event_list_batch).
    slices > 1: every window as `slices` lists of pass lists over contiguous batch ranges (the micro-batches of
    train.StreamReplicas)."""
    from event_flow_amd import synthetic

    pool = []
    for wdx in range(n_pool):
        lists = []
        for k in range(PASSES):
            seed0 = synthetic.seed_for(CONFIG_ID, rank, 0) + 100000 * wdx + 1000 * k
            ev = synthetic.event_list_batch(B_PER_GPU, EV_PER_PASS, H, W, seed0, kind=kind)
            lists.append(ev[0] if isinstance(ev, tuple) else ev)  # (moving_dots also returns its ground-truth motion)
        # one resident buffer [B,P,N,4] per window, the passes are its slices [:, p]: the binning kernel and the loss read the
        # window in place (no torch.stack / torch.cat in the step)
        window = torch.from_numpy(np.ascontiguousarray(np.stack(lists, 1))).to(dev)
        if slices == 1:
            pool.append([window[:, k] for k in range(PASSES)])
        else:
            b = B_PER_GPU // slices
            parts = [window[j * b:(j + 1) * b].contiguous() for j in range(slices)]
            pool.append([[part[:, k] for k in range(PASSES)] for part in parts])
    return pool


def _encode(lists):
    from event_flow_amd.train import encode_passes

    passes = encode_passes(lists, 2, (H, W), want=("cnt", "mask", "pol"))  # all 10 passes binned in one launch
    for d in passes:
        d["event_voxel"] = None  # encoding = cnt
    return passes


def run_step(model, lossf, opt, dp, lists, reps=None):
    from event_flow_amd.train import train_window

    if reps is not None:  # lists: one pass list per micro-batch
        return reps.train_window([_encode(part) for part in lists], dp=dp)
    return train_window(model, lossf, opt, _encode(lists), dp=dp)


class StepGraph:
    """One training step of one input window as hipGraph replays.  The whole step is a single
    graph -- with several ranks too: the step's ONE all-reduce is evf_allreduce_sum
    (include/evflow.h) on the library's own RCCL communicator, a plain ncclAllReduce on the
    capture stream, i.e. one more kernel node -- with EVF_DP_NATIVE=1 (opt-in: the captured
    multi-rank collective has not run on hardware yet).  The default (and EVF_DP_TWO_GRAPHS=1, or a
    non-RCCL backend) is the earlier form: two graphs -- (binning, passes, loss, backward,
    loss staged into the flat buffer) and (clip+Adam, detach, reset) -- with the all-reduce
    launched eagerly between them."""

    def __init__(self, model, lossf, opt, dp, lists, stream, reps=None, capture_mode=None):
        from event_flow_amd.train import window_apply, window_backward

        self.dp, self.comm, self.reps = dp, opt.comm, reps
        # other threads (the process group's watchdog polls events) must not invalidate a capture of this thread
        mode = capture_mode or ("thread_local" if dp.active else "global")
        if reps is not None:
            # micro-batch pipelining (train.StreamReplicas): one graph per replica on its own stream, then the join:
            # gradients / losses summed, (all-reduce outside any capture,) clip+Adam, detach, reset
            self.main = stream
            self.slices, losses = [], []
            for k in range(reps.n):
                g = torch.cuda.CUDAGraph()
                st = stream if k == 0 else reps.streams[k]
                st.wait_stream(stream)
                with torch.cuda.graph(g, stream=st, capture_error_mode=mode):
                    losses.append(reps.backward_slice(k, _encode(lists[k])))
                stream.wait_stream(st)
                self.slices.append((g, st))
            self.pre = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.pre, stream=stream, capture_error_mode=mode):
                local = reps.combine(losses)
                if dp.active:
                    dp.stage(opt.comm, local)
                else:
                    self.loss = reps.apply(local, dp)
            self.post = None
            if dp.active:
                self.post = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.post, stream=stream, capture_error_mode=mode):
                    self.loss = reps.apply(local, dp)
            return
        if not dp.active or (dp.capturable and os.environ.get("EVF_DP_TWO_GRAPHS", "0") != "1"):
            # one rank -- or N ranks whose all-reduce is evf_allreduce_sum on the library's own RCCL communicator: a plain
            # ncclAllReduce on this stream, captured like every other launch -- the whole step is ONE graph
            self.pre = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.pre, stream=stream, capture_error_mode=mode):
                self.loss = run_step(model, lossf, opt, dp, lists)
            self.post = None
            return
        self.pre, self.post = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.pre, stream=stream, capture_error_mode=mode):
            local = window_backward(model, lossf, opt, _encode(lists), dp)
        with torch.cuda.graph(self.post, stream=stream, capture_error_mode=mode):
            self.loss = window_apply(model, lossf, opt, local, dp)

    def replay(self):
        if self.reps is not None:  # fork: every replica's graph on its stream; join on the main one
            for _g, st in self.slices[1:]:
                st.wait_stream(self.main)
            for g, st in self.slices:
                with torch.cuda.stream(st):
                    g.replay()
            for _g, st in self.slices[1:]:
                self.main.wait_stream(st)
        self.pre.replay()
        if self.post is not None:
            self.dp.reduce(self.comm)
            self.post.replay()
        return self.loss


def capture_step_graphs(model, lossf, opt, dp, pool, stream, reps=None, capture_mode=None):
    """One StepGraph per window of `pool` (the warm-up must have run eagerly on `stream` with
    model.use_static_states(True)).  The recurrent state crosses replays without a copy: graph 0 starts from the
    buffers the warm-up left, graph k from the tensors graph k-1's last pass wrote (fixed addresses in its capture
    pool), and the last graph's last pass writes straight back into the first buffers.  Replays cycle 0,1,0,1...
    `graph.left` = the state tensors a replay of that graph leaves behind (model.set_state_buffers)."""
    models = reps.models if reps is not None else [model]
    home = [m.state_buffers() for m in models]
    for m in models:
        m.use_static_states(False)
    graphs = []
    for gi, lists in enumerate(pool):
        if gi == len(pool) - 1:
            for m, h in zip(models, home):
                m.final_states_into(h)
        graphs.append(StepGraph(model, lossf, opt, dp, lists, stream, reps, capture_mode))
        graphs[-1].left = model.state_buffers() if reps is None else [m.state_buffers() for m in models]
    return graphs


def launch_floor_us():
    """Duration of a launch that does nothing worth mentioning (a 64-element add), from the two-point calibration every
    profile_start / profile_stop pair runs (T1 = o + t, T2 = o + 2t): the floor any single-kernel operator sits on."""
    from event_flow_amd import _lib

    return _lib.last_tiny_kernel_ms * 1e3, _lib.last_event_overhead_ms * 1e3


def iwe_warp_bandwidth(dev, B, reps=20):
    from event_flow_amd import synthetic
    from event_flow_amd.utils.iwe import compute_pol_iwe

    n = 15000
    H = W = 128  # the north-star shape of the IWE figure, whatever the benched workload
    g = np.random.default_rng(1)
    ev_small = synthetic.event_list_batch(min(B, 8), n, H, W, 4242)
    ev = np.concatenate([ev_small] * (B // ev_small.shape[0]), 0) if B > ev_small.shape[0] else ev_small
    flow = torch.from_numpy(g.uniform(-0.1, 0.1, size=(B, 2, H, W)).astype(np.float32)).to(dev)
    ev = torch.from_numpy(ev).to(dev)
    pol = torch.stack([(ev[:, :, 3] > 0).float(), (ev[:, :, 3] < 0).float()], 2).contiguous()
    from event_flow_amd import _lib

    for _ in range(3):
        compute_pol_iwe(flow, ev, (H, W), pol[:, :, 0:1], pol[:, :, 1:2], flow_scaling=128, round_idx=True)
    torch.cuda.synchronize()
    # kernel duration from HIP events recorded around the launch on its stream (the call is one
    # kernel; a host-paced loop would measure Python overhead at the small shape)
    _lib.profile_start(["evf_iwe_splat"])
    for _ in range(reps):
        compute_pol_iwe(flow, ev, (H, W), pol[:, :, 0:1], pol[:, :, 1:2], flow_scaling=128, round_idx=True)
    times = _lib.profile_stop()[("evf_iwe_splat", "")]
    ms = float(np.median(times))
    alg_bytes = B * (n * 28 + 2 * H * W * 4)  # 16 B event + 8 B flow gather + 4 B atomic dst, + final image
    return {"B": B, "events": n, "ms_per_call": ms, "algorithmic_MB": alg_bytes / 1e6, "GBps": alg_bytes / ms / 1e6,
            "frac_of_hbm_peak": alg_bytes / ms / 1e6 / HBM_PEAK}


_KERNEL_OF = {"evf_conv_dgrad_b3": "k_conv_dgrad_b3_lds<false, false, false, false>",
              "evf_conv_dgrad_b3_f32": "k_conv_dgrad_b3_lds<true, false, false, false>",
              "evf_conv_dgrad_b3_f32/acc": "k_conv_dgrad_b3_lds<true, true, false, false>",
              "evf_conv_dgrad_b3_f32_pair": "k_conv_dgrad_b3_lds<true, false, false, true>",
              "evf_conv_dgrad_b3_f32_pair/acc": "k_conv_dgrad_b3_lds<true, true, false, true>",
              "evf_conv_lif_fwd_b3_pred/ff": "k_conv_lif_fwd_b3<false, false>",
              "evf_lif_bwd_wgrad/ff": "k_lif_bwd_wgrad<false, false, true>", "evf_lif_bwd_wgrad_top": "k_lif_bwd_wgrad<false, true, true>",
              "evf_lif_bwd_wgrad/rec": "k_lif_bwd_wgrad<true, false, true>", "evf_lif_bwd_wgrad/rec+2": "k_lif_bwd_wgrad<true, false, true>",
              "evf_lif_bwd_wgrad/ff+2": "k_lif_bwd_wgrad<false, false, true>", "evf_conv_lif_fwd_b3/ff": "k_conv_lif_fwd_b3<false, false>",
              "evf_conv_lif_fwd_b3/rec": "k_conv_lif_fwd_b3<true, false>", "evf_head_lif_fwd": "k_head_lif_fwd<1>",
              "k_head_lif_fwd_win": "k_head_lif_fwd_win<1, false, 8>", "k_head_bwd_win": "k_head_bwd_win<true, 4, false, 0>",
              "evf_head_lif_bwd_wgrad": "k_head_bwd_mfma<true>", "evf_conv_dgrad/one": "k_conv_dgrad<false>",
              "evf_conv_dgrad/two": "k_conv_dgrad<true>", "k_fwd_diag": "k_fwd_diag_t<true, true, false, 0>", "k_fwd_win": "k_fwd_win_t<true, true, false, 0>", "k_bwd_win": "k_bwd_win_lif", "k_dgrad_multi": "k_dgrad_diag_dma<true, false>", "k_bwd_diag": "k_bwd_diag_ws<8>",
              "k_dgrad_diag": "k_dgrad_diag_dma<true, false>", "evf_conv_lif_fwd/ff": "k_conv_lif_fwd<false>",
              "evf_conv_lif_fwd/rec": "k_conv_lif_fwd<true>", "evf_conv_wgrad_bits": "k_conv_wgrad_bits"}


_PMC_FILE = "r*_bench_pmc_traffic.json"  # (config 5: r*_c5_pmc_traffic.json, see kernel_names_for)


def kernel_names_for(model_name, B, Hh, Ww):
    """The device kernels the library launches for THIS workload (rocprofv3's names): the table above is the LIF-FireNet at
    8 x 128 x 128.  PLIF cells run the PLIF instantiations; from 6 tiles of 4 x 32 pixels per CU on the stand-alone input
    gradient is the wave-specialised k_conv_dgrad_ws<ACC, PLIF, PAIR> (dg_launch, evf_dgrad_b3.hip) -- `acc` in the entry's
    variant means accumulate != 0, which for a PLIF cell is the raw-dL/dP flag (ACC = false)."""
    global _PMC_FILE
    plif = model_name == "PLIFFireNet"
    if plif:
        _PMC_FILE = "r*_c5_pmc_traffic.json"
        _KERNEL_OF.update({"k_fwd_diag": "k_fwd_diag_t<true, false, true, 0>", "k_fwd_win": "k_fwd_win_t<true, false, true, 0>",
                           "k_bwd_win": "k_bwd_win_plif", "k_bwd_diag": "k_bwd_diag_ws_plif",
                           "k_head_lif_fwd_win": "k_head_lif_fwd_win<1, true, 8>", "k_head_bwd_win": "k_head_bwd_win<true, 3, true, 0>"})
    if B * ((Hh + 3) // 4) * ((Ww + 31) // 32) >= 6 * 256:
        tf = lambda v: "true" if v else "false"  # noqa: E731
        for pair in (False, True):
            for acc in (False, True):
                ent = "evf_conv_dgrad_b3_f32" + ("_pair" if pair else "") + ("/acc" if acc else "")
                _KERNEL_OF[ent] = f"k_conv_dgrad_ws<{tf(acc and not plif)}, {tf(plif)}, {tf(pair)}>"


def _pmc(entry):
    """PMC figures of the kernel behind `entry` from the newest committed passes (rocprofv3 cannot run inside this
    process): profiles/r*_bench_pmc_traffic.json (config 5: r*_c5_pmc_traffic.json) -- HBM bytes per launch (FETCH_SIZE
    x2-corrected + WRITE_SIZE) and MFMA busy (SQ_VALU_MFMA_BUSY_CYCLES per SIMD / GRBM_GUI_ACTIVE per XCD).  The file carries the
    hash of the kernel sources it was measured on; on any other sources the numbers are stale and NOT reported (-> (None, reason))."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", _PMC_FILE)))
    if not files or entry not in _KERNEL_OF:
        return None, "no PMC pass for this kernel"
    d = json.load(open(files[-1]))
    if d.get("src_hash") != source_hash():
        return None, f"stale: {os.path.basename(files[-1])} was measured on sources {d.get('src_hash')}, these are {source_hash()}"
    t = d["per_launch"].get(_KERNEL_OF[entry])
    if not t or t["fetch_MB"] != t["fetch_MB"]:
        return None, "kernel not in the PMC pass"
    return {"MB_per_launch": round(t["fetch_MB"] + t["write_MB"], 2), "fetch_MB": t["fetch_MB"], "write_MB": t["write_MB"],
            "mfma_busy_pct": t.get("mfma_busy_pct"), "source": os.path.basename(files[-1]), "src_hash": d["src_hash"]}, None


# device kernels behind the general path's input-gradient entry point (evf_conv2d_dgrad_b3): the six-term tile kernels and the
# parity-class (stride-2) form of the gather kernel; forward products run k_conv3_b3x / k_conv2d_b3<.., false, ..>
def _c4_dgrad_kernel(name):
    import re

    return (name.startswith("k_conv3_b3t<") or name.startswith("k_conv3_b3n") or name.startswith("k_conv3_b3i") or
            re.match(r"k_conv2d_b3<\d+, \d+, true, ", name) is not None)


def _pmc_c4(entry):
    """PMC bytes per STEP of the device kernels behind a general-path entry point, from the newest committed config-4 passes
    (profiles/r*_c4_pmc_traffic.json: per-launch means x launches per step); refused on other kernel sources."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_c4_pmc_traffic.json")))
    if not files or entry != "evf_conv2d_dgrad_b3":
        return None, "no PMC pass for this entry point"
    d = json.load(open(files[-1]))
    if d.get("src_hash") != source_hash():
        return None, f"stale: {os.path.basename(files[-1])} was measured on sources {d.get('src_hash')}, these are {source_hash()}"
    ks = {k: v for k, v in d["per_launch"].items() if _c4_dgrad_kernel(k)}
    if not ks:
        return None, "kernels not in the PMC pass"
    tot = sum((v["fetch_MB"] + v["write_MB"]) * v["launches_per_step"] for v in ks.values())
    return {"MB_per_step": round(tot, 1), "kernels": ks, "source": os.path.basename(files[-1]), "src_hash": d["src_hash"]}, None


def gpu_forward_loss_line(wl, dev, pool, precision, reps_n=20, capture_mode="global"):
    """BASELINE configs[1]: LIF-FireNet forward + IWE (contrast-maximisation) loss, 128x128, 15k events / window, batch 8 -- the
    forward half of the headline step on the same windows (binning, 10 passes, loss; no backward, no optimizer step), on a model
    instance of its own, replayed from one hipGraph per window like the headline step and timed with the wall clock."""
    from event_flow_amd.loss.flow import EventWarping
    from event_flow_amd.models import model as models
    from event_flow_amd.train import window_forward_loss

    torch.manual_seed(0)
    net = getattr(models, wl["model"])(dict(MODEL_CFG)).to(dev)
    net.precision = precision
    net.train()
    lossf = EventWarping(LOSS_CFG, dev)
    net.use_static_states(True)  # (recurrent state at fixed addresses across replays; copied at the window boundary)
    cur = torch.cuda.current_stream()
    for i in range(3):
        window_forward_loss(net, lossf, _encode(pool[i % len(pool)]))
    torch.cuda.synchronize()
    graphs, losses, mode = [], [], "hipgraph"
    try:
        for lists in pool:
            g = torch.cuda.CUDAGraph()
            # (with a process group up, its watchdog thread polls events: it must not invalidate this thread's capture)
            with torch.cuda.graph(g, stream=cur, capture_error_mode=capture_mode):
                losses.append(window_forward_loss(net, lossf, _encode(lists)))
            graphs.append(g)
        for i in range(2):
            graphs[i % len(graphs)].replay()
    except Exception as e:  # noqa: BLE001 -- capture unsupported here: eager launches
        print(f"[bench] c2: hipGraph capture failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
        graphs, mode = [], "eager"
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = None
    for i in range(reps_n):
        if graphs:
            graphs[i % len(graphs)].replay()
            loss = losses[i % len(graphs)]
        else:
            loss = window_forward_loss(net, lossf, _encode(pool[i % len(pool)]))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps_n
    return {"workload": "LIF-FireNet forward + CM loss only (binning, 10 passes x 1500 events, 128x128, batch 8; BASELINE configs[1])",
            "value": B_PER_GPU / dt, "unit": "event-windows/s", "ms_per_window_batch": dt * 1e3, "launch": mode, "loss": float(loss)}


def parity_report_status():
    """The committed parity report (profiles/rNN_parity_report.txt, tools/parity_report.sh) is stamped with the hash of the
    kernel sources it measured: distances of other sources are STALE, and the line says so."""
    import glob
    import re

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_parity_report.txt")))
    if not files:
        return {"file": None}
    f = files[-1]
    m = re.search(r"^# kernel sources: ([0-9a-f]+)", open(f).read(), re.M)
    cur = source_hash()
    return {"file": os.path.relpath(f, ROOT), "kernel_sources": m.group(1) if m else None, "current_sources": cur,
            "stale": (m.group(1) != cur) if m else True}


def cpu_model():
    """CPU model string of the host (stated next to the core count)."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform

    return platform.processor() or platform.machine()


def other_config_line(cfg, steps=10, warmup=3, timeout=420, extra=None):
    """A short run of another BASELINE configuration (or of the headline one with other flags: `extra`) in its own process
    (own model, own HIP context), summarised for the headline line: value, ms_per_step and the roofline object of that run."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__)] + (["--config", cfg] if cfg else []) + ["--steps", str(steps), "--warmup", str(warmup),
           "--no-cpu-baseline", "--no-iwe", "--no-others"] + (["--no-graph-profile", "--repeat-blocks", "3"] + list(extra) if extra else [])
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
        lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": f"rc {r.returncode}: {r.stderr.strip()[-300:]}"}
        d = json.loads(lines[-1])
        roof = d.get("roofline") or {}
        return {"metric": d["metric"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                "warmup": d["warmup"], "dtype": d["dtype"], "launch": d["config"].get("launch"), "workload": d["config"]["workload"],
                "timing_blocks_ms_per_step": (d.get("timing_blocks") or {}).get("ms_per_step"), "loss": d["config"].get("loss"),
                "workload_activity": d.get("workload_activity"),
                "roofline": {k: roof.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic")}}
    except Exception as e:  # noqa: BLE001  (never costs the headline line)
        return {"error": f"{type(e).__name__}: {e}"}


def physical_cores():
    """Physical cores of the host (unique (socket, core) pairs of /proc/cpuinfo; SMT siblings count once)."""
    try:
        seen, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        if seen:
            return len(seen)
    except OSError:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def cpu_baseline(threads, max_seconds=60.0, name="LIFFireNet"):
    """Oracle (PyTorch-CPU port of the reference path) on a bounded sample: full train steps of the SAME per-GPU work as the GPU
    step (B = 8 windows of 10 passes x 1500 events; c5: 4 windows), at three FIXED thread counts -- 1, 16 and all physical cores
     -- each on at least one full step (>= 3 windows), all three reported (`by_threads`); `value` / `cores` = the
    fastest of the three."""
    from event_flow_amd import synthetic
    from oracle import encodings as oenc
    from oracle import snn as osnn
    from oracle import train as otrain

    gen = torch.Generator().manual_seed(0)
    neuron = ({"leak": (-4.0, 0.1), "thresh": (0.8, 0.1)} if name == "LIFFireNet" else
              {"leak_v": (-4.0, 0.1), "leak_pt": (-4.0, 0.1), "add_pt": (-2.0, 0.1), "thresh": (0.8, 0.1)})
    params0 = osnn.make_firenet_params(name, gen, neuron=neuron)
    keys = osnn.trainable_keys(params0)
    Bc = B_PER_GPU  # the same per-GPU batch of windows the GPU step processes
    passes = []
    for k in range(PASSES):
        ev = synthetic.event_list_batch(Bc, EV_PER_PASS, H, W, 555 + 10 * k)
        d = oenc.collate([oenc.encode_window(ev[b, :, 2], ev[b, :, 1], ev[b, :, 0], ev[b, :, 3], 2, (H, W)) for b in range(Bc)])
        passes.append({k2: torch.from_numpy(v) for k2, v in d.items()})
    lcfg = {"flow_regul_weight": 0.001, "mask_output": True}

    def one_step(ps, st, opt, pr):
        _, _, pr, st = otrain.train_step(name, pr, keys, ps, st, (H, W), opt, loss_cfg=lcfg)
        return pr, st

    phys = physical_cores()
    counts = sorted({t for t in (1, 16, phys) if 1 <= t <= max(threads, 1)} or {1})
    budget = {1: 1, 16: 20}  # full steps at most per thread count (1 thread: one step of Bc windows, ~10 s)
    by_threads, best = {}, None
    params = params0
    for t in counts:
        torch.set_num_threads(t)
        one_step(passes[:1], [None] * 7, {"step": 0, "m": {}, "v": {}}, params0)  # warm (thread pool, allocator)
        t0 = time.perf_counter()
        opt, states, n_done, params = {"step": 0, "m": {}, "v": {}}, [None] * 7, 0, params0
        while True:
            params, states = one_step(passes, states, opt, params)
            n_done += 1
            el = time.perf_counter() - t0
            if el > (10.0 if t > 1 else 0.0) or n_done >= budget.get(t, 20) or el > max_seconds:
                break
        el = time.perf_counter() - t0
        by_threads[str(t)] = {"windows_per_s": Bc * n_done / el, "steps": n_done, "windows": Bc * n_done, "seconds": round(el, 2)}
        if best is None or by_threads[str(t)]["windows_per_s"] > by_threads[str(best)]["windows_per_s"]:
            best = t
    best_t = best
    n_done, el = by_threads[str(best_t)]["steps"], by_threads[str(best_t)]["seconds"]
    # SURVEY 8(d) asks for more CPU figures beside the headline one; each is a bounded sample of its own
    extra = {}
    try:
        from oracle import iwe as oiwe

        torch.set_num_threads(best_t)
        with torch.no_grad():  # configuration 2: forward + loss of the 10-pass window, no backward
            t1 = time.perf_counter()
            otrain.forward_window(name, params, passes, [None] * 7, (H, W), loss_cfg=lcfg)
            extra["fwd_loss_windows_per_s"] = Bc / (time.perf_counter() - t1)
        extra["one_thread_windows_per_s"] = by_threads.get("1", {}).get("windows_per_s")  # per-core figure (a full step on one thread)
        ev = synthetic.event_list_batch(64, PASSES * EV_PER_PASS, H, W, 999)  # compute_pol_iwe, 64 windows of 15k events
        fl = np.random.default_rng(0).standard_normal((64, 2, H, W)).astype(np.float32)
        pos, neg = (ev[:, :, 3:4] > 0).astype(np.float32), (ev[:, :, 3:4] < 0).astype(np.float32)
        t1 = time.perf_counter()
        oiwe.compute_pol_iwe(fl, ev, (H, W), pos, neg, flow_scaling=128, round_idx=True)
        extra["iwe_warp_GBps"] = 64 * (28 * PASSES * EV_PER_PASS + 2 * H * W * 4) / (time.perf_counter() - t1) / 1e9
    except Exception as e:  # the extras never cost the headline baseline
        extra["error"] = f"{type(e).__name__}: {e}"
    return {"value": by_threads[str(best_t)]["windows_per_s"], "unit": "event-windows/s", "cores": best_t, "kind": "port",
            "cpu_model": cpu_model(), "host_cpus": os.cpu_count(), "physical_cores": phys, "by_threads": by_threads, "extra": extra,
            "sample": f"full train steps of {Bc} windows (B={Bc}, {PASSES} passes x {EV_PER_PASS} events, {H}x{W}) = the GPU step's "
                      f"per-GPU work, at FIXED thread counts {counts} (1, 16, all {phys} physical cores of the {os.cpu_count()}-CPU host): "
                      + ", ".join(f"{t} thr: {v['steps']} step(s) = {v['windows']} windows in {v['seconds']} s" for t, v in by_threads.items())
                      + f"; value = the fastest ({best_t} threads); oracle = PyTorch-CPU fp32 port of the reference path"}


def self_launch(n):
    """`python bench.py --gpus N` with no launcher: start N ranks (one per GPU) of this same command line under
    torch.distributed.run on 127.0.0.1 and a free port, relay their output (rank 0 prints the JSON line) and
    return the launcher's exit status."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL between processes)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def hip_ops_conv_b3():
    from event_flow_amd.models import hip_ops

    return hip_ops.CONV_B3


def main_c4(args):
    """BASELINE configs[3]: LIF-EV-FlowNet (SpikingRecEVFlowNet, base 32, 20.4 M parameters), 256x256, one window of 50 000
    events per sample, batch 8, 4 flow scales, full train step on the general path (bf16 MFMA with exact operand splits; replayed
    from two hipGraphs).  Same JSON shape as the headline line; the roofline object is the conv entry point with the largest total
    time: its ISSUED bf16 matrix work against the dense bf16 MFMA peak."""
    from event_flow_amd import _lib, synthetic
    from event_flow_amd.loss.flow import EventWarping
    from event_flow_amd.models.model import SpikingRecEVFlowNet
    from event_flow_amd.train import FlatAdam, encode_passes, train_window

    if args.gpus != 1:
        raise SystemExit("--config c4 is a single-GPU line (BASELINE configs[3]: 1 x MI355X)")
    dev = "cuda:0"
    torch.cuda.set_device(0)
    Hc = Wc = 256
    Bc, nev = 8, 50000
    torch.manual_seed(0)
    cfg = dict(MODEL_CFG, name="SpikingRecEVFlowNet")
    model = SpikingRecEVFlowNet(dict(cfg)).to(dev)
    if args.thresh_scale != 1.0:
        with torch.no_grad():
            for k, p in model.named_parameters():
                if k.endswith("thresh"):
                    p.mul_(args.thresh_scale)
    model.train()
    lossf = EventWarping({"loader": {"resolution": [Hc, Wc]}, "loss": dict(LOSS_CFG["loss"]), "model": {"mask_output": True}}, dev)
    use_graph = not args.no_graph
    opt = FlatAdam(model, lr=2e-4, clip=100.0, device_step=use_graph)
    opt.zero_grad()
    pool = [encode_passes([torch.from_numpy(synthetic.event_list_batch(Bc, nev, Hc, Wc, synthetic.seed_for(4, 0, 0) + 100000 * w)).to(dev)],
                          2, (Hc, Wc)) for w in range(2)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(side)
    for i in range(max(args.warmup, 2)):
        loss = train_window(model, lossf, opt, pool[i % 2])
    torch.cuda.synchronize()
    # the whole step (passes, 4-scale loss, backward, clip + Adam, state reset) of each of the two windows as a hipGraph; a
    # window of this workload is ONE pass, the recurrent state is reset per window (train_window detaches and the loss resets),
    # so the two graphs are independent and replay alternately.  Falls back to eager launches if the capture is refused.
    graphs, mode = None, "eager"

    state_copies = None
    if use_graph:
        try:
            # the recurrent state crosses the windows: graph 0 starts from the tensors the warm-up left, graph 1 from the tensors
            # graph 0 produces (fixed addresses in its pool), and graph 1's cells write their new states straight back into the
            # first ones (train.capture_window_cycle / hip_ops.route_states: no copy of the 410 MB of state, no memcpy nodes)
            from event_flow_amd.train import capture_window_cycle

            graphs, state_copies = capture_window_cycle(model, lossf, opt, pool, side, route=os.environ.get("EVF_STATE_ROUTE", "1") != "0")
            torch.cuda.synchronize()
            for w in range(2):
                graphs[w][0].replay()
            torch.cuda.synchronize()
            mode = "hipgraph"
        except Exception as e:  # noqa: BLE001
            print(f"[bench c4] hipGraph capture failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
            graphs = None
            torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if graphs is not None:
            graphs[i % 2][0].replay()
            loss = graphs[i % 2][1]
        else:
            loss = train_window(model, lossf, opt, pool[i % 2])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    block_ms = [elapsed / args.steps * 1e3]  # (the contract's block, then the same region repeated: median and spread)
    if graphs is not None:
        for _ in range(max(args.repeat_blocks, 1) - 1):
            torch.cuda.synchronize()
            tb = time.perf_counter()
            for i in range(args.steps):
                graphs[i % 2][0].replay()
            torch.cuda.synchronize()
            block_ms.append((time.perf_counter() - tb) / args.steps * 1e3)
        loss = graphs[(args.steps - 1) % 2][1]
    names = ["evf_conv2d_fwd", "evf_conv2d_dgrad", "evf_conv2d_fwd_b3", "evf_conv2d_fwd_b3_parts", "evf_lif_fwd_parts", "evf_conv2d_dgrad_b3", "evf_conv2d_wgrad", "evf_neuron_fwd", "evf_neuron_bwd", "evf_upsample2x_fwd",
             "evf_upsample2x_bwd", "evf_upsample_nearest_fwd", "evf_upsample_nearest_bwd", "evf_cm_loss_fwd", "evf_cm_loss_bwd",
             "evf_clip_adam_step", "evf_pack_conv2d_weight", "evf_pack_conv2d_weight_b3", "evf_pack_conv2d_weights_b3_multi", "evf_encode_events"]
    prof_steps = 2
    _lib.profile_start(names)
    for i in range(prof_steps):
        train_window(model, lossf, opt, pool[i % 2])
    prof = _lib.profile_stop()
    kernels = {}
    for (name, var), ms in prof.items():
        if name == "evf_conv2d_fwd_b3_parts":  # the same product, its K-split partial sums left to the neuron kernel
            name = "evf_conv2d_fwd_b3"
        ent = kernels.setdefault(name, {"launches": 0, "total_ms_per_step": 0.0, "flop_per_step": 0.0, "bytes_per_step": 0.0})
        ent["launches"] += len(ms) // prof_steps
        ent["total_ms_per_step"] += float(np.sum(ms)) / prof_steps
        if name.startswith("evf_conv2d_") and var:
            b, h, w, cin, cout, k, st = (int(v) for v in var.split(","))
            ho, wo = (h + 2 * (k // 2) - k) // st + 1, (w + 2 * (k // 2) - k) // st + 1
            ent["flop_per_step"] += 2.0 * k * k * cin * cout * b * ho * wo * len(ms) / prof_steps
            ent["bytes_per_step"] += 4.0 * b * (h * w * cin + ho * wo * cout) * len(ms) / prof_steps
    for name, ent in kernels.items():
        if ent["flop_per_step"]:
            ent["TFLOPs"] = ent["flop_per_step"] / (ent["total_ms_per_step"] * 1e-3) / 1e12
            ent["frac_of_fp32_mfma_peak"] = ent["TFLOPs"] / FP32_MFMA_PEAK
            wgrad_b3 = name == "evf_conv2d_wgrad" and os.environ.get("EVF_WGRAD", "b3") != "f32"
            if name.endswith("_b3") or wgrad_b3:
                # issued bf16 work lies between 3x (every wave's fragment exactly representable) and 6x the fp32-equivalent
                # FLOPs, by the per-wave vote of evf_conv_b3gen.hip; the input gradient always takes 6; the weight gradient
                # (evf_wgrad_b3gen.hip) issues 3x for the spike-valued channel tiles (stride-2 and flagged tiles: fp32 MFMA)
                lo_terms = 6 if "dgrad" in name else 3
                ent["issued_bf16_TFLOPs_range"] = [ent["TFLOPs"] * lo_terms, ent["TFLOPs"] * 6]
                ent["frac_of_bf16_peak_range"] = [ent["TFLOPs"] * lo_terms / BF16_MFMA_PEAK, ent["TFLOPs"] * 6 / BF16_MFMA_PEAK]
            ent["algorithmic_GBps"] = ent["bytes_per_step"] / (ent["total_ms_per_step"] * 1e-3) / 1e9
        ent["algorithmic_bytes_per_step"] = ent.pop("bytes_per_step")  # (fp32 input + output of every launch: 0 for the element-wise entries)
    dom_name = max((n for n in kernels if "TFLOPs" in kernels[n]), key=lambda n: kernels[n]["total_ms_per_step"])
    dom = kernels[dom_name]
    c4_traffic, c4_traffic_why = _pmc_c4(dom_name)
    out = {
        "metric": "event-windows/sec (train step, 256x256x50k ev, LIF-EV-FlowNet)", "value": Bc * args.steps / elapsed,
        "unit": "event-windows/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "LIF-EV-FlowNet (SpikingRecEVFlowNet, base 32) full train step, 256x256, 50k events/window, batch 8, "
                               "4 flow scales, CM loss, clip+Adam [BASELINE configs[3]]", "baseline_config": "c4", "global_batch": Bc,
                   "events_per_window": nev, "parallelism": "dp1", "launch": mode, "loss": float(loss),
                   "state_tensors_copied_per_cycle": state_copies,
                   "conv_precision": ("forward / input gradient: bf16 MFMA with exact 3-way operand splits, fp32 accumulation "
                                      "(3 products per 16 channels for spike-valued waves, 6 otherwise; EVF_CONV=f32 for the fp32 "
                                      "kernels); " if hip_ops_conv_b3() else "") +
                                     ("weight gradient: 3x3 stride 1 on the bf16 matrix cores (x one bf16 plane, g three; spike-valued "
                                      "channel tiles), the rest fp32 MFMA; " if os.environ.get("EVF_WGRAD", "b3") != "f32" else
                                      "weight gradient: fp32 MFMA (v_mfma_f32_32x32x2_f32); ") + "NHWC fp32 activations"},
        "roofline": ({"kernel": dom_name, "bound": "mfma", "achieved": dom["issued_bf16_TFLOPs_range"][0], "peak": BF16_MFMA_PEAK,
                      "unit": "TFLOP/s", "frac": dom["issued_bf16_TFLOPs_range"][0] / BF16_MFMA_PEAK,
                      "traffic": int(c4_traffic["MB_per_step"] * 1e6) if c4_traffic else None,
                      "traffic_unit": "bytes per STEP over all launches of the entry point's device kernels (PMC passes: FETCH_SIZE x2-corrected + "
                                      "WRITE_SIZE, per-launch means x launches per step)",
                      "traffic_detail": c4_traffic if c4_traffic else {"unavailable": c4_traffic_why},
                      "algorithmic_bytes_per_step": int(dom.get("algorithmic_bytes_per_step", 0)),
                      "launches_per_step": dom["launches"], "total_ms_per_step": dom["total_ms_per_step"],
                      "fp32_equivalent_TFLOPs": dom["TFLOPs"],
                      "note": "all launches of the entry point in a step together: sum of 2*k*k*Cin*Cout*B*Ho*Wo over the launches / "
                              "their summed HIP-event time = fp32-equivalent TFLOP/s; `achieved` = the bf16 matrix work ISSUED for it "
                              "(at least 3 exact-split products per fp32 product; 6 for real-valued operands), priced against the "
                              "dense bf16 MFMA peak"}
                     if "issued_bf16_TFLOPs_range" in dom else
                     {"kernel": dom_name, "bound": "mfma", "achieved": dom["TFLOPs"], "peak": FP32_MFMA_PEAK, "unit": "TFLOP/s",
                      "frac": dom["TFLOPs"] / FP32_MFMA_PEAK, "traffic": None,
                      "note": "all launches of the entry point in a step together: sum of 2*k*k*Cin*Cout*B*Ho*Wo over the launches / "
                              "their summed HIP-event time"}),
        "timing_blocks": {"ms_per_step": [round(v, 4) for v in block_ms], "steps_per_block": args.steps,
                          "median_ms_per_step": float(np.median(block_ms)),
                          "spread_pct": float(100.0 * (np.max(block_ms) - np.min(block_ms)) / np.median(block_ms))},
        "workload_activity": {"thresh_scale": args.thresh_scale, "events": "uniform"},
        "kernels": kernels,
        "kernel_timing": {"method": "HIP events around each launch over eager steps, bracket overhead removed",
                          "bracket_overhead_us": round(_lib.last_event_overhead_ms * 1e3, 2)},
    }
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-iwe", action="store_true")
    ap.add_argument("--precision", choices=["bf16x3", "fp32"], default="bf16x3",
                    help="matrix-core path of the 32->32 convs: exact bf16x3 split (default) or fp32 MFMA")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--profile-capture-mode", choices=["global", "thread_local", "relaxed"], default="thread_local",
                    help="stream-capture mode of the instrumented capture (timestamp-kernel nodes)")
    ap.add_argument("--no-graph-profile", action="store_true",
                    help="skip the second, instrumented capture (timestamp kernels around the diagonal launches) the per-kernel "
                         "durations of the replayed step come from; kernels[*] then carry the eager HIP-event figures only")
    ap.add_argument("--streams", type=int, default=1,
                    help="micro-batch pipelining (train.StreamReplicas): the per-GPU batch as this many slices on their own HIP "
                         "streams (default 1 = off).  Round 2 (one block per tile, launches of a few cells): 2 gave +6.6 %% windows/s at the "
                         "headline shape; with round 3's persistent one-block-per-CU launches it loses: 3.69 against 3.65 ms per step "
                         "(4 streams: 4.62)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--repeat-blocks", type=int, default=5,
                    help="timed blocks of --steps replays each: the first is the contract's timed region (value / ms_per_step), the others "
                         "follow it and give the median and the spread (timing_blocks)")
    ap.add_argument("--thresh-scale", type=float, default=1.0,
                    help="multiply every firing threshold of the model by this (0.25: an ALIVE network, every layer at 20-50 %% spike rate; the "
                         "default initialisation with synthetic uniform events is nearly silent above the third layer)")
    ap.add_argument("--events", choices=["uniform", "moving_dots"], default="uniform", help="synthetic event generator (synthetic.event_list_batch)")
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="multi-GPU readiness check without a timed region: start the N ranks exactly as the real run does, bind one "
                         "GPU per rank, bring the process group (and, with EVF_DP_NATIVE=1, the library's own RCCL communicator) up, "
                         "SUM all-reduce one known buffer, print ONE JSON record and exit.  Fails loudly when fewer than N GPUs are visible")
    ap.add_argument("--no-alive", action="store_true", help="skip the short run of the ALIVE workload the default invocation appends")
    ap.add_argument("--no-others", action="store_true",
                    help="skip the short c4 / c5 runs the default invocation appends as `other_configs` (after the c3 line's timed region)")
    ap.add_argument("--config", choices=["c3", "c4", "c5"], default="c3",
                    help="BASELINE.json workload: c3 = headline LIF-FireNet train step (default), c5 = PLIF-FireNet 260x346 (4 per GPU), "
                         "c4 = LIF-EV-FlowNet 256x256 x 50k events (general path)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if args.config == "c4":
            raise SystemExit("--config c4 is a single-GPU line (BASELINE configs[3]: 1 x MI355X)")
        if not os.environ.get("EVF_BENCH_SINGLE_DEVICE") and torch.cuda.device_count() < args.gpus:
            # fail HERE, with the count, not as N tracebacks out of the launcher (EVF_BENCH_SINGLE_DEVICE=1: the test hook that lets
            # several gloo ranks share one GPU)
            raise SystemExit(f"--gpus {args.gpus} needs {args.gpus} visible GPUs (one rank per GPU), this node shows "
                             f"{torch.cuda.device_count()}: nothing launched")
        raise SystemExit(self_launch(args.gpus))

    from event_flow_amd import _lib
    from event_flow_amd.loss.flow import EventWarping
    from event_flow_amd.models import model as models
    from event_flow_amd.parallel import DataParallel
    from event_flow_amd.train import FlatAdam

    _lib.load()  # fails loudly when the HIP library is missing
    if args.config == "c4":
        return main_c4(args)
    wl = set_workload(args.config)
    kernel_names_for(wl["model"], B_PER_GPU, H, W)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("EVF_BENCH_SINGLE_DEVICE"):  # test hook: several ranks share one GPU (with EVF_DP_BACKEND=gloo)
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {os.environ.get('RANK', '0')}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible "
                         f"(--gpus {args.gpus} needs one GPU per rank)")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    dp = DataParallel(device=dev)
    if dp.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={dp.world}: launch with torch.distributed.run")
    if dp.active:
        # fail fast, with the reason, when the process group did not come up with one rank per requested GPU (the bench line's
        # `ranks` below is what torch.distributed reports after init, never the environment's word)
        import torch.distributed as _dist

        ranks = _dist.get_world_size() if _dist.is_initialized() else 0
        if ranks != args.gpus:
            raise SystemExit(f"--gpus {args.gpus}: the {dp.backend} process group has {ranks} rank(s) after init (expected {args.gpus}); "
                             "check the launcher (one process per GPU, MASTER_ADDR=127.0.0.1) and HSA_ENABLE_IPC_MODE_LEGACY=0")

    if args.dry_run_launch:
        # everything the real run does up to the first step: ranks, devices, process group, (own communicator,) one collective
        t = torch.full((1024,), float(dp.rank + 1), dtype=torch.float32, device=dev)
        if dp.active:
            dp.reduce(t)
        torch.cuda.synchronize()
        want = dp.world * (dp.world + 1) / 2.0
        ok = bool((t == want).all().item()) if dp.active else True
        bad = dp.max_over_ranks(0.0 if ok else 1.0)
        if dp.rank == 0:
            import torch.distributed as _dist

            print(json.dumps({"dry_run_launch": True, "n_gpus": dp.world, "baseline_config": args.config,
                              "global_batch": B_PER_GPU * dp.world, "per_gpu_batch": B_PER_GPU, "resolution": [H, W],
                              "backend": dp.backend, "ranks_in_process_group": _dist.get_world_size() if _dist.is_initialized() else 1,
                              "devices_visible": torch.cuda.device_count(), "device_of_rank0": torch.cuda.get_device_name(local_rank),
                              "sum_allreduce_of_rank_plus_1": float(t[0]), "expected": want, "ok_all_ranks": bad == 0.0,
                              "native_requested": dp.native_requested, "native_comm_count": dp.native_ranks,
                              "native_fallback": dp.native_fallback, "capturable_collective": dp.capturable}), flush=True)
        dp.barrier()
        dp.close()
        raise SystemExit(0 if bad == 0.0 else 1)

    torch.manual_seed(0)  # identical replicas on every rank
    model = getattr(models, wl["model"])(dict(MODEL_CFG)).to(dev)
    model.precision = args.precision
    if args.thresh_scale != 1.0:
        with torch.no_grad():
            for k, p in model.named_parameters():
                if k.endswith("thresh"):
                    p.mul_(args.thresh_scale)
    model.train()
    lossf = EventWarping(LOSS_CFG, dev)
    use_graph = not args.no_graph
    opt = FlatAdam(model, lr=2e-4, clip=100.0, device_step=use_graph)
    opt.zero_grad()
    # micro-batch pipelining (train.StreamReplicas): the rank's batch as `--streams` slices, each through its own replica
    # (shared weights, own state / tape / gradient buffer) on its own HIP stream; gradients summed before the one step
    nstream = args.streams if (args.streams > 1 and B_PER_GPU % args.streams == 0) else 1
    reps = None
    if nstream > 1:
        from event_flow_amd.train import StreamReplicas

        reps = StreamReplicas(model, lossf, opt, n=nstream)
    if use_graph:
        for m in (reps.models if reps is not None else [model]):
            m.use_static_states(True)  # recurrent state must live at fixed addresses across replays
    pool = make_windows(dp.rank, 2, dev, slices=nstream, kind=args.events)
    names = ["evf_lif_bwd_wgrad2", "evf_conv_lif_fwd", "evf_conv_lif_fwd_b3", "evf_conv_lif_fwd_b3_pred", "evf_conv_dgrad", "evf_conv_dgrad_b3",
             "evf_conv_dgrad_b3_f32", "evf_conv_dgrad_b3_f32_pair", "evf_conv_wgrad_bits", "evf_lif_bwd_wgrad", "evf_lif_bwd_wgrad_top",
             "evf_lif_bwd", "evf_head_lif_bwd_wgrad", "evf_head_lif_fwd", "evf_head_wgrad", "evf_pred_bwd", "evf_reduce_slabs", "evf_reduce_slabs_multi", "evf_cm_loss_fwd",
             "evf_cm_loss_bwd", "evf_conv_plif_fwd_b3", "evf_head_plif_fwd", "evf_plif_trace_bwd", "evf_encode_events", "evf_clip_adam_step",
             ]
    from event_flow_amd import train as _train

    # diagonal launches (train.window_backward -> engine.defer_forward): the hidden forward cells of a window are recorded and
    # launched by ONE evf_fwd_defer_flush call (P + 5 k_fwd_diag launches); the per-cell entry points then launch nothing
    # (PLIF: the forward cells are recorded too -- k_fwd_diag_t<.., PLIF> --, the backward cells launch one by one)
    plif_net = wl["model"] == "PLIFFireNet"
    diag_fwd = _train.DEFER_FORWARD and getattr(model, "precision", "") == "bf16x3" and wl["model"] in ("LIFFireNet", "PLIFFireNet")
    diag_bwd = _train.DEFER_BACKWARD and getattr(model, "precision", "") == "bf16x3" and wl["model"] == "LIFFireNet"
    if diag_fwd:  # (these entry points then only record: timed inside the flush, evf_defer_profile)
        names = [n for n in names if n not in ("evf_conv_lif_fwd_b3", "evf_conv_lif_fwd_b3_pred", "evf_conv_plif_fwd_b3",
                                               "evf_conv_plif_fwd_b3_pred") and (plif_net or n != "evf_head_lif_fwd")]
    if diag_bwd:
        names = [n for n in names if n not in ("evf_lif_bwd_wgrad2", "evf_lif_bwd_wgrad", "evf_lif_bwd_wgrad_top", "evf_conv_dgrad_b3_f32",
                                               "evf_conv_dgrad_b3_f32_pair", "evf_conv_dgrad_b3", "evf_head_lif_bwd_wgrad")]

    # Everything runs on one side stream: warm-up (eager), then one whole training step
    # per input window is captured into a hipGraph on that same stream (autograd's
    # AccumulateGrad nodes remember the stream they were first used on), so that ~1100
    # kernel launches per step become a single graph launch.
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(side)
    for i in range(max(args.warmup, 2)):
        run_step(model, lossf, opt, dp, pool[i % len(pool)], reps)
    graphs = None
    mode = "eager"
    if use_graph:
        try:
            torch.cuda.synchronize()
            graphs = capture_step_graphs(model, lossf, opt, dp, pool, side, reps)
        except Exception as e:  # capture unsupported in this environment: eager launches
            print(f"[bench] rank {dp.rank}: hipGraph capture failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
            graphs = None
        torch.cuda.synchronize()
        # all ranks must issue the same collective sequence: graphs on every rank or on none
        if dp.active and dp.max_over_ranks(0.0 if graphs is not None else 1.0) != 0.0:
            graphs = None
        if graphs is not None:
            for i in range(2):  # replay warm-up
                graphs[i % len(graphs)].replay()
            torch.cuda.synchronize()
            mode = "hipgraph"

    dp.barrier()
    torch.cuda.synchronize()
    if graphs is None:
        _lib.profile_start(names)
        _lib.load().evf_defer_profile(1)
    if dp.active:
        dp.time_reduces(True)
    t0 = time.perf_counter()
    loss = None
    for i in range(args.steps):
        if graphs is not None:
            loss = graphs[i % len(graphs)].replay()
        else:
            loss = run_step(model, lossf, opt, dp, pool[i % len(pool)], reps)
    torch.cuda.synchronize()
    dp.barrier()
    elapsed = time.perf_counter() - t0
    reduce_ms = dp.reduce_times_ms() if dp.active else []
    if dp.active:
        dp.time_reduces(False)
    # Spread of the headline figure: the SAME timed region (barrier, K replays, synchronize, barrier) four more times, right
    # after the contract's own block.  `value` / `ms_per_step` stay the first block's (the contract: exactly K steps); the
    # blocks, their median and their spread are reported beside them (every rank runs the same number of replays).
    block_ms = [elapsed / args.steps * 1e3]
    if graphs is not None and args.repeat_blocks > 1:
        for _ in range(args.repeat_blocks - 1):
            dp.barrier()
            torch.cuda.synchronize()
            tb = time.perf_counter()
            for i in range(args.steps):
                loss_b = graphs[i % len(graphs)].replay()
            torch.cuda.synchronize()
            dp.barrier()
            block_ms.append(dp.max_over_ranks(time.perf_counter() - tb) / args.steps * 1e3)
        del loss_b
    one_graph = bool(dp.active and graphs is not None and graphs[0].post is None)
    if dp.active and not reduce_ms:
        # the collective sits inside the replayed graph (no bracket there): its own device time from 20 eager launches of the same
        # all-reduce on a scratch buffer of the same size, after the timed region
        scratch = torch.zeros_like(opt.comm)
        dp.time_reduces(True)
        for _ in range(20):
            dp.reduce(scratch)
        reduce_ms = dp.reduce_times_ms()[2:]
        dp.time_reduces(False)
    # the diagonal / head-window launches AS THEY RUN INSIDE A REPLAYED STEP: a second, instrumented capture of the same step
    # graphs whose flushes bracket every dispatcher launch with timestamp-kernel nodes (evf_defer_profile(2)); replayed after
    # the timed region, read once.  The timed graphs above carry no such nodes.
    gprof = None
    if graphs is not None and not args.no_graph_profile:
        import ctypes as _ct0

        try:
            L = _lib.load()
            L.evf_defer_profile(2)
            graphs_p, cap_err = None, None
            try:
                graphs_p = capture_step_graphs(model, lossf, opt, dp, pool, side, reps, capture_mode=args.profile_capture_mode)
            except Exception as e:  # noqa: BLE001
                cap_err = e
            finally:
                L.evf_defer_profile(0)
            torch.cuda.synchronize()
            # N ranks: the replays below hold the step's all-reduce -- every rank replays or none does (a rank that lost its
            # instrumented capture alone would leave the others waiting inside the collective)
            if dp.active and dp.max_over_ranks(0.0 if cap_err is None else 1.0) != 0.0:
                raise RuntimeError(f"instrumented capture failed on a rank ({cap_err})")
            if cap_err is not None:
                raise cap_err
            nrep = 3 * len(graphs_p)
            for i in range(nrep):
                graphs_p[i % len(graphs_p)].replay()
            torch.cuda.synchronize()
            _gms, _gcnt = (_ct0.c_float * 16)(), (_ct0.c_int * 16)()
            if L.evf_defer_profile_read(_gms, _gcnt) != 0:
                raise RuntimeError("evf_defer_profile_read failed")
            empty_ms = (_gms[7] / _gcnt[7]) if _gcnt[7] else 0.0
            gprof = {"empty_bracket_us": empty_ms * 1e3, "per_kind": {k: (_gms[k] / _gcnt[k], _gcnt[k]) for k in range(16) if _gcnt[k] and k != 7}}
            del graphs_p
        except Exception as e:  # noqa: BLE001 -- the instrumented capture never costs the headline line
            print(f"[bench] rank {dp.rank}: instrumented graph capture failed ({type(e).__name__}: {e}); eager kernel timing only", file=sys.stderr)
            gprof = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.synchronize()
    if graphs is None:
        prof = _lib.profile_stop()
        prof_steps = args.steps
    else:
        # per-kernel durations cannot be bracketed inside a graph replay: time the same
        # kernels with HIP events over a few eager steps of the same workload
        prof_steps = 3
        _lib.profile_start(names)
        _lib.load().evf_defer_profile(1)
        for i in range(prof_steps):
            run_step(model, lossf, opt, dp, pool[i % len(pool)], reps)
        prof = _lib.profile_stop()
    # the diagonal launches, timed per launch inside the flushes (HIP events in the library, same bracket overhead)
    import ctypes as _ct

    _ms, _cnt = (_ct.c_float * 16)(), (_ct.c_int * 16)()
    if _lib.load().evf_defer_profile_read(_ms, _cnt) != 0:
        raise RuntimeError("evf_defer_profile_read failed")
    prof_eager = {}
    for k, nm in enumerate([("k_fwd_diag", ""), ("k_bwd_diag", ""), ("k_dgrad_diag", ""), ("evf_head_lif_bwd_wgrad", ""),
                            ("k_head_lif_fwd_win", ""), ("k_head_bwd_win", ""), ("k_fwd_win", ""), ("-", ""), ("k_bwd_win", ""),
                            ("k_dgrad_multi", "")]):
        if _cnt[k] and k != 7:
            prof[nm] = [max(_ms[k] / _cnt[k] - _lib.last_event_overhead_ms, 0.0)] * _cnt[k]
            if gprof and "per_kind" in gprof and k in gprof["per_kind"]:
                # inside the replayed graph: mean bracket of the last replay minus the empty bracket of the same graph
                prof_eager[nm] = prof[nm][0]
                prof[nm] = [max(gprof["per_kind"][k][0] - gprof["empty_bracket_us"] * 1e-3, 0.0)] * _cnt[k]
    # evf_lif_bwd_wgrad2 = evf_lif_bwd_wgrad with dL/d(spikes) in two parts: one kernel, reported under the one name
    # (variant "+2": the second part present, +128 B/px)
    prof = {(("evf_lif_bwd_wgrad",) + k[1:] if k[0] == "evf_lif_bwd_wgrad2" else k): v for k, v in prof.items()}
    event_overhead_us = _lib.last_event_overhead_ms * 1e3
    elapsed = dp.max_over_ranks(elapsed)
    loss_val = float(loss)

    model_precision = model.precision
    _model_obj = model
    if dp.rank == 0:
        npix = (B_PER_GPU // nstream) * H * W  # pixels one LAUNCH covers (a micro-batch when the step is pipelined)
        # algorithmic work per launch: FLOP of the 3x3 32->32 contraction(s) and
        # compulsory HBM bytes (fp32 tensors 128 B/px, split-bf16 planes 192 B/px, spike words 4 B/px)
        model = {
            ("evf_conv_lif_fwd", "ff"): (CONV_FLOP * npix, 268 * npix), ("evf_conv_lif_fwd", "rec"): (2 * CONV_FLOP * npix, 268 * npix),
            ("evf_conv_lif_fwd_b3", "ff"): (CONV_FLOP * npix, 272 * npix), ("evf_conv_lif_fwd_b3", "rec"): (2 * CONV_FLOP * npix, 272 * npix),
            ("evf_conv_dgrad", "one"): (CONV_FLOP * npix, 256 * npix), ("evf_conv_dgrad", "two"): (2 * CONV_FLOP * npix, 384 * npix),
            ("evf_conv_dgrad_b3", ""): (CONV_FLOP * npix, 320 * npix),
            # fp32 g_cur in (128 B/px, halo not counted), fp32 gradient out; the pair form writes two outputs
            ("evf_conv_dgrad_b3_f32", ""): (CONV_FLOP * npix, 256 * npix), ("evf_conv_dgrad_b3_f32_pair", ""): (2 * CONV_FLOP * npix, 384 * npix),
            # accumulating forms: the previous g_x is read as well (+128 B/px)
            ("evf_conv_dgrad_b3_f32", "acc"): (CONV_FLOP * npix, 384 * npix), ("evf_conv_dgrad_b3_f32_pair", "acc"): (2 * CONV_FLOP * npix, 512 * npix),
            ("evf_conv_lif_fwd_b3_pred", "ff"): (CONV_FLOP * npix, 280 * npix),
            # top layer: g_v, v', v in, g_cur + g_v_prev out, flow / g_flow 16 B/px, spike words
            ("evf_lif_bwd_wgrad_top", ""): (CONV_FLOP * npix, 668 * npix),
            ("evf_conv_wgrad_bits", ""): (CONV_FLOP * npix, 132 * npix),
            # g_z, g_v, v', v in; g_cur (fp32, split later by the dgrad) + g_v_prev out; spike words / planes
            ("evf_lif_bwd_wgrad", "ff"): (CONV_FLOP * npix, 776 * npix), ("evf_lif_bwd_wgrad", "rec"): (2 * CONV_FLOP * npix, 780 * npix),
            ("evf_lif_bwd_wgrad", "ff+2"): (CONV_FLOP * npix, 904 * npix), ("evf_lif_bwd_wgrad", "rec+2"): (2 * CONV_FLOP * npix, 908 * npix),
            ("evf_lif_bwd", ""): (0, 772 * npix), ("evf_head_lif_bwd_wgrad", ""): (2 * 18 * 32 * npix, 652 * npix),
            ("evf_head_lif_fwd", ""): (2 * 18 * 32 * npix, 272 * npix),
        }
        # which roofline bounds the kernel: the fp32-MFMA convs are matrix-core bound; the bf16x3 kernels
        # need 1/5 of those cycles and are HBM bound, like the elementwise ones
        hbm_bound = {"evf_conv_lif_fwd_b3", "evf_conv_lif_fwd_b3_pred", "evf_conv_dgrad_b3", "evf_conv_dgrad_b3_f32",
                     "evf_conv_dgrad_b3_f32_pair", "evf_lif_bwd_wgrad", "evf_lif_bwd_wgrad_top", "evf_lif_bwd", "evf_head_lif_fwd",
                     "evf_head_lif_bwd_wgrad"}
        # matrix-core accounting: the bf16x3 kernels issue 3 (forward, weight gradient: binary operand x 3-way split) or 6
        # (input gradient: two real operands) bf16 MFMA products per algorithmic fp32 product
        bf16_terms = {"evf_conv_lif_fwd_b3": 3, "evf_conv_lif_fwd_b3_pred": 3, "evf_conv_plif_fwd_b3": 3, "evf_lif_bwd_wgrad": 3,
                      "evf_lif_bwd_wgrad_top": 3, "evf_conv_dgrad_b3": 6, "evf_conv_dgrad_b3_f32": 6, "evf_conv_dgrad_b3_f32_pair": 6}
        # PLIF: + previous trace in, new trace out (128 B/px each), pooled activity out (4)
        model[("evf_conv_plif_fwd_b3", "")] = (CONV_FLOP * npix, 532 * npix)
        model[("evf_head_plif_fwd", "")] = (2 * 18 * 32 * npix, 532 * npix)
        # g_cur, g_pt carry, pt_prev in; g_pt_prev out (fp32 [npix][32]: 4 x 128 B/px); P in, g_P_raw out [npix] (the new trace is
        # recomputed from pt_prev and P, the pooling's adjoint runs inside the input-gradient kernels)
        model[("evf_plif_trace_bwd", "")] = (0, 520 * npix)
        hbm_bound |= {"evf_conv_plif_fwd_b3", "evf_head_plif_fwd", "evf_plif_trace_bwd"}
        # contrast-maximisation loss: forward 88 B/event (24 B read + 2 directions x 4 corners x 2 images x 4 B) +
        # the 8 images read once for the reduction + the flow maps of the P passes for the smoothness term; backward 96 B/event
        # (24 B + 64 B image-gradient gather + 8 B atomic to dL/dflow) + the 8 images + dL/dflow of the P passes written
        nev_w = PASSES * EV_PER_PASS
        img = (8 + 2 * PASSES) * H * W * 4
        model[("evf_cm_loss_fwd", "")] = (0, (B_PER_GPU // nstream) * (nev_w * 88 + img))
        model[("evf_cm_loss_bwd", "")] = (0, (B_PER_GPU // nstream) * (nev_w * 96 + img))
        hbm_bound |= {"evf_cm_loss_fwd", "evf_cm_loss_bwd"}
        # diagonal launches: PASSES + 5 launches hold the window's cells; per LAUNCH = the window's total / (PASSES + 5)
        nl = PASSES + 5
        alt_bytes = {}
        # the recorded forward layer by layer (engine._fwd_slots: shapes where one cell fills the chip, EVF_FWD_LM): recurrent
        # layers one cell per launch, a feed-forward hidden layer's passes in ONE launch (k_fwd_win_t) that reads the input words
        # and writes the tape only -- per pixel and pass 4 + 128 + 8 (PLIF: + 128 trace + 4 pooled activity), the state before
        # the window once, the flow maps of the top layer (8 B per pass on one of the four layers)
        _eng_obj = getattr((reps.models[0] if reps is not None else _model_obj), "_engine", None)
        fwd_mode = _eng_obj._fwd_mode(B_PER_GPU // nstream, H, W) if (diag_fwd and _eng_obj is not None) else "0"
        fwd_lm = fwd_mode == "1"
        nl_f = nl  # launches that hold the diagonal cells of the forward
        if diag_fwd and fwd_mode in ("1", "top"):
            per_cell = 532 if plif_net else 272
            per_pass = 272 if plif_net else 140
            if fwd_lm:  # recurrent cells one per launch; four feed-forward layers as chains (the top one writes the flow maps: 8 B)
                model[("k_fwd_diag", "")] = (2 * CONV_FLOP * npix, per_cell * npix)
                per_pass += 2
            else:  # "top": G1, R1a, R1b, G2 on P + 3 diagonals (6 contractions per pass), R2a and R2b (+ flow maps) as chains
                nl_f = PASSES + 3
                model[("k_fwd_diag", "")] = (6 * PASSES * CONV_FLOP * npix / nl_f, PASSES * 4 * per_cell * npix / nl_f)
                per_pass += 4
            model[("k_fwd_win", "")] = (PASSES * CONV_FLOP * npix, (PASSES * per_pass + (260 if plif_net else 132)) * npix)
            hbm_bound |= {"k_fwd_diag", "k_fwd_win"}
            bf16_terms["k_fwd_diag"] = 3
            bf16_terms["k_fwd_win"] = 3
        elif diag_fwd:  # 6 hidden cells per pass (8 contractions: two recurrent cells), 272 B/px each, 280 under the prediction head
            per_cell = 532 if plif_net else 272  # (PLIF: + trace in / out + pooled activity)
            model[("k_fwd_diag", "")] = (8 * PASSES * CONV_FLOP * npix / nl, PASSES * (5 * per_cell + per_cell + 8) * npix / nl)
            hbm_bound |= {"k_fwd_diag"}
            bf16_terms["k_fwd_diag"] = 3
        # LIF: the two feed-forward layers above the last recurrent one run their backward of all passes ahead of the diagonals
        # (engine._backward_window_top): the diagonals then hold four layers in P + 3 launches each way
        bwd_top = bool(diag_bwd and _eng_obj is not None and _eng_obj._top_static(B_PER_GPU // nstream, H, W) and PASSES <= 16)
        nl_b, n_bwd_layers = (PASSES + 3, 4) if bwd_top else (nl, 6)
        if diag_bwd:
            # fused-backward cells: per pass 3 feed-forward cells (776 B/px), the top one (668), 2 recurrent ones (780; 908 with the
            # second gradient part: every pass but the last; in the first pass they have no previous state: 904)
            rec_b = 2 * ((PASSES - 2) * 908 + 780 + 904)
            by_b = ((2 * PASSES * 776 + rec_b) if bwd_top else (3 * PASSES * 776 + PASSES * 668 + rec_b)) * npix
            # input-gradient cells: per pass 4 with one weight set (256 B/px) and 2 with two (384); all six single in the first pass
            # (window launches on top: 2 with one weight set and 2 with two)
            n_one = 2 if bwd_top else 4
            by_d = ((PASSES - 1) * (n_one * 256 + 2 * 384) + (n_one + 2) * 256) * npix
            by_b32, by_d32 = by_b, by_d
            from event_flow_amd.models import engine as _eng

            if _eng.SPLIT_DGRAD:
                # dL/d(current) travels as its exact 3-way bf16 split (three planes, 192 B/px) instead of the fp32 tensor (128): the
                # fused backward writes +64 B/px per cell, the input gradient reads +64 B/px per cell (k_dgrad_diag_dma stages the
                # planes by LDS-DMA).  `frac_fp32_layout` below prices the same launch with the fp32 layout's bytes.
                by_b += n_bwd_layers * PASSES * 64 * npix
                by_d += n_bwd_layers * PASSES * 64 * npix
            n_contr = n_bwd_layers + 2  # (two recurrent cells: two contractions each)
            model[("k_bwd_diag", "")] = (n_contr * PASSES * CONV_FLOP * npix / nl_b, by_b / nl_b)
            model[("k_dgrad_diag", "")] = ((n_contr * (PASSES - 1) + n_bwd_layers) * CONV_FLOP * npix / nl_b, by_d / nl_b)
            alt_bytes = {"k_bwd_diag": by_b32 / nl_b, "k_dgrad_diag": by_d32 / nl_b}
            if bwd_top:
                # a layer's window launch (two per step: under the prediction head / below it), per pixel and pass: dL/d(spikes) 128 (top:
                # flow + dL/dflow 16), v_prev 128, spike words and planes 12, the three planes of dL/d(current) 192 written; v' of the
                # last pass once.  Its input gradients: P products per launch, planes in (192), gradient out (128)
                model[("k_bwd_win", "")] = (PASSES * CONV_FLOP * npix, (PASSES * (460 + 348) / 2 + 128) * npix)
                model[("k_dgrad_multi", "")] = (PASSES * CONV_FLOP * npix, PASSES * 320 * npix)
                hbm_bound |= {"k_bwd_win", "k_dgrad_multi"}
                bf16_terms["k_bwd_win"] = 3
                bf16_terms["k_dgrad_multi"] = 6
            hbm_bound |= {"k_bwd_diag", "k_dgrad_diag"}
            bf16_terms["k_bwd_diag"] = 3
            bf16_terms["k_dgrad_diag"] = 6
        # the head layer of a recorded window, one launch each way (per LAUNCH = per window).  Forward: per pass the 2-channel
        # input in (8 B/px), v', spike words and planes out (136); the starting state once (132).  Backward: per pass dL/d(spikes),
        # the previous potential, spike word and input in (268); v' of the last pass and the carried gradient in / out once (384)
        model[("k_head_lif_fwd_win", "")] = (PASSES * 2 * 18 * 32 * npix, (PASSES * 144 + 132) * npix)
        model[("k_head_bwd_win", "")] = (PASSES * 2 * 18 * 32 * npix, (PASSES * 268 + 384) * npix)
        hbm_bound |= {"k_head_lif_fwd_win", "k_head_bwd_win"}
        kernels = {}
        step_alg_bytes = 0.0
        recorded_only = []
        for key, ms in prof.items():
            ms = np.array(ms)
            name = "/".join(k for k in key if k)
            if ms.size and float(ms.mean()) * 1e3 < 8.0 and key[0] in ("evf_head_plif_fwd", "evf_head_lif_fwd", "evf_conv_lif_fwd_b3", "evf_conv_plif_fwd_b3",
                                                                          "evf_conv_lif_fwd_b3_pred", "evf_conv_plif_fwd_b3_pred", "evf_head_lif_bwd_wgrad",
                                                                          "evf_lif_bwd_wgrad", "evf_lif_bwd_wgrad_top", "evf_conv_dgrad_b3"):
                # an entry point that only RECORDS its cell while a window is being recorded (its kernel runs inside a window /
                # diagonal launch listed under k_*): an empty bracket is not a kernel time
                recorded_only.append(name)
                continue
            ent = {"launches": int(ms.size), "mean_us": float(ms.mean() * 1e3), "total_ms_per_step": float(ms.sum() / prof_steps)}
            if key in prof_eager:
                ent["timing"] = "inside the replayed hipGraph (timestamp kernels of an instrumented capture, empty bracket removed)"
                ent["mean_us_eager"] = prof_eager[key] * 1e3
            if key in model:
                fl, by = model[key]
                ent["algorithmic_MB"] = by / 1e6
                ent["GBps"] = by / (ms.mean() * 1e-3) / 1e9
                ent["frac_of_hbm_peak"] = ent["GBps"] / HBM_PEAK
                step_alg_bytes += by * ms.size / prof_steps
                if fl:
                    ent["fp32_equiv_TFLOPs"] = fl / (ms.mean() * 1e-3) / 1e12
                    if key[0] in bf16_terms:
                        ent["issued_bf16_TFLOPs"] = bf16_terms[key[0]] * ent["fp32_equiv_TFLOPs"]
                        ent["frac_of_bf16_peak"] = ent["issued_bf16_TFLOPs"] / BF16_MFMA_PEAK
                    else:
                        ent["frac_of_fp32_mfma_peak"] = ent["fp32_equiv_TFLOPs"] / FP32_MFMA_PEAK
                ent["bound"] = "hbm" if key[0] in hbm_bound else "mfma"
                pm, why = _pmc(name)
                ent["mfma_busy_pct"] = pm["mfma_busy_pct"] if pm else None  # SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE (PMC pass)
                if not pm:
                    ent["pmc"] = why
            if key[0] in ("evf_cm_loss_fwd", "evf_cm_loss_bwd"):
                ent["note"] = ("one call = 5 (forward) / 3 (backward) launches over the window's events and its 8 warped-event images: "
                               "latency of dependent launches at this size, not a bandwidth figure")
            if diag_bwd and key[0] in alt_bytes and alt_bytes[key[0]] != model[key][1]:
                ent["algorithmic_MB_fp32_layout"] = alt_bytes[key[0]] / 1e6
                ent["frac_fp32_layout"] = alt_bytes[key[0]] / (ms.mean() * 1e-3) / 1e9 / HBM_PEAK
            if key[0] == "k_fwd_diag" and fwd_lm:
                ent["note"] = "forward layer by layer: a launch = ONE recurrent cell (k_fwd_diag_t); the feed-forward layers: k_fwd_win"
            elif key[0] == "k_fwd_win":
                ent["note"] = f"a feed-forward hidden layer's {PASSES} passes in one launch (k_fwd_win_t): state in registers, tape written only"
            elif key[0] == "k_fwd_diag" and fwd_mode == "top":
                ent["note"] = (f"diagonal launches: the {4 * PASSES} cells of the four hidden layers up to the last recurrent one in {nl_f} "
                               "launches of 1..4 independent (pass, layer) cells; the two layers above them: k_fwd_win; per LAUNCH")
            elif key[0] in ("k_bwd_diag", "k_dgrad_diag") and bwd_top:
                ent["note"] = (f"diagonal launches: the {4 * PASSES} cells of this kind of the four hidden layers up to the last recurrent one in "
                               f"{nl_b} launches of 1..4 independent (pass, layer) cells; the two layers above them: evf_lif_bwd_wgrad_window / "
                               "evf_conv_dgrad_b3_multi, all passes per launch; mean_us / algorithmic_MB are per LAUNCH (window total / launches)")
            elif key[0] in ("k_fwd_diag", "k_bwd_diag", "k_dgrad_diag"):
                ent["note"] = (f"diagonal launches: the window's {6 * PASSES} cells of this kind in {nl} launches of 1..6 independent "
                               "(pass, layer) cells; mean_us / algorithmic_MB are per LAUNCH (window total / launches)")
            kernels[name] = ent
        # The roofline object is the DEVICE kernel with the largest total time per step: entries that launch the same kernel (the
        # input gradients on the diagonals and as product lists are both k_dgrad_diag_dma) count together; its figures are the
        # sums over those launches (algorithmic bytes / FLOP of all of them over their summed duration).
        groups = {}
        for k in prof:
            if k in model:
                nm = "/".join(x for x in k if x)
                groups.setdefault(_KERNEL_OF.get(nm, nm), []).append(k)
        dom_dev = max(groups, key=lambda g: sum(sum(prof[k]) for k in groups[g]))
        dom_keys = sorted(groups[dom_dev], key=lambda k: -sum(prof[k]))
        dom_key = dom_keys[0]
        dom = dict(kernels["/".join(k for k in dom_key if k)])
        if len(dom_keys) > 1:  # several entries, one kernel: aggregate per launch
            n_l = sum(len(prof[k]) for k in dom_keys)
            t_ms = sum(float(np.sum(prof[k])) for k in dom_keys)
            by_all = sum(model[k][1] * len(prof[k]) for k in dom_keys)
            fl_all = sum(model[k][0] * len(prof[k]) for k in dom_keys)
            dom.update({"launches": n_l, "mean_us": t_ms / n_l * 1e3, "total_ms_per_step": t_ms / prof_steps,
                        "algorithmic_MB": by_all / n_l / 1e6, "GBps": by_all / (t_ms * 1e-3) / 1e9,
                        "frac_of_hbm_peak": by_all / (t_ms * 1e-3) / 1e9 / HBM_PEAK})
            if all(k in prof_eager for k in dom_keys):
                dom["mean_us_eager"] = sum(prof_eager[k] * len(prof[k]) for k in dom_keys) / n_l * 1e3
            if fl_all:
                dom["fp32_equiv_TFLOPs"] = fl_all / (t_ms * 1e-3) / 1e12
                if dom_key[0] in bf16_terms:
                    dom["issued_bf16_TFLOPs"] = bf16_terms[dom_key[0]] * dom["fp32_equiv_TFLOPs"]
                    dom["frac_of_bf16_peak"] = dom["issued_bf16_TFLOPs"] / BF16_MFMA_PEAK
            if any(k[0] in alt_bytes for k in dom_keys):
                by_alt = sum(alt_bytes.get(k[0], model[k][1]) * len(prof[k]) for k in dom_keys)
                dom["algorithmic_MB_fp32_layout"] = by_alt / n_l / 1e6
                dom["frac_fp32_layout"] = by_alt / (t_ms * 1e-3) / 1e9 / HBM_PEAK
        # which side bounds it, by the figures themselves: the kernel sits on the matrix side when the bf16 matrix work it ISSUES is a
        # larger fraction of the dense bf16 peak than its algorithmic bytes are of the HBM peak (the PMC's MFMA-busy says the same)
        if dom.get("frac_of_bf16_peak") is not None and dom["frac_of_bf16_peak"] > dom.get("frac_fp32_layout", dom["frac_of_hbm_peak"]):
            dom["bound"] = "mfma_bf16"
        detail, why_not = _pmc("/".join(k for k in dom_key if k))
        traffic = int(detail["MB_per_launch"] * 1e6) if detail else None  # HBM bytes per launch (PMC), next to ...
        algo = int(dom["algorithmic_MB"] * 1e6) if "algorithmic_MB" in dom else None  # ... the algorithmic bytes per launch
        per_step = {"launches_per_step": dom["launches"] / prof_steps, "total_ms_per_step": dom["total_ms_per_step"],
                    "entries": ["/".join(x for x in k if x) for k in dom_keys], "device_kernel": dom_dev,
                    "selection": "arg-max of total time per step over the DEVICE kernels (entries launching one kernel count together)"}
        if dom["bound"] == "mfma_bf16":
            # the six-term input gradient: priced on the matrix side.  `achieved` = bf16 matrix work ISSUED (6 exact-split
            # products per fp32 product) per second against the dense bf16 peak; the algorithmic (fp32-equivalent) rate and the
            # HBM fraction of the same launches stand beside it
            hb = dom.get("frac_fp32_layout", dom["frac_of_hbm_peak"])
            roof = {"kernel": dom_dev, "bound": "mfma", "achieved": dom["issued_bf16_TFLOPs"], "peak": BF16_MFMA_PEAK, "unit": "TFLOP/s",
                    "frac": dom["frac_of_bf16_peak"], "mean_launch_us": dom["mean_us"],
                    "timing": dom.get("timing", "HIP events around eager launches, bracket overhead removed"),
                    "mean_launch_us_eager": dom.get("mean_us_eager"),
                    "issued_bf16_TFLOPs": dom["issued_bf16_TFLOPs"], "algorithmic_fp32_equiv_TFLOPs": dom["fp32_equiv_TFLOPs"],
                    "bf16_products_per_fp32_product": bf16_terms.get(dom_key[0]),
                    "traffic": traffic, "traffic_unit": "bytes/launch (PMC pass: the diagonal launches)",
                    "algorithmic_bytes": int(dom.get("algorithmic_MB_fp32_layout", dom["algorithmic_MB"]) * 1e6),
                    "algorithmic_bytes_layout": int(dom["algorithmic_MB"] * 1e6), "frac_of_hbm_peak": hb,
                    "frac_of_hbm_peak_layout": dom["frac_of_hbm_peak"],
                    "traffic_detail": detail if detail else {"unavailable": why_not},
                    "mfma_busy_pct": detail["mfma_busy_pct"] if detail else None, **per_step,
                    "note": "input gradient g_x = sum_tap g[pix + tap] * W^T: both operands real-valued, so the exact bf16 split needs 6 "
                            "products per fp32 product (DESIGN 4.1); 108 MFMAs per 4 x 32-pixel tile and wave. `frac` prices the ISSUED "
                            "bf16 work; the fp32-equivalent rate is algorithmic_fp32_equiv_TFLOPs"}
        elif dom["bound"] == "hbm":
            # `achieved` / `frac` price the launch with its COMPULSORY bytes (every tensor once in the fp32 layout:
            # dL/d(current) as 128 B/px).  What the kernel really moves in the layout it uses (that tensor as three bf16 planes,
            # 192 B/px) is reported beside it as *_layout.
            comp_frac = dom.get("frac_fp32_layout", dom["frac_of_hbm_peak"])
            comp_bytes = int(dom["algorithmic_MB_fp32_layout"] * 1e6) if "algorithmic_MB_fp32_layout" in dom else algo
            eager_scale = (dom["mean_us"] / dom["mean_us_eager"]) if dom.get("mean_us_eager") else None
            roof = {"kernel": dom_dev, **per_step, "bound": "hbm", "achieved": comp_frac * HBM_PEAK, "peak": HBM_PEAK,
                    "unit": "GB/s", "frac": comp_frac, "mean_launch_us": dom["mean_us"],
                    "timing": dom.get("timing", "HIP events around eager launches, bracket overhead removed"),
                    "frac_eager": (comp_frac * eager_scale) if eager_scale else None, "mean_launch_us_eager": dom.get("mean_us_eager"),
                    "traffic": traffic, "traffic_unit": "bytes/launch",
                    "algorithmic_bytes": comp_bytes, "frac_layout": dom["frac_of_hbm_peak"], "achieved_layout": dom["GBps"],
                    "algorithmic_bytes_layout": algo, "traffic_detail": detail if detail else {"unavailable": why_not},
                    "mfma_busy_pct": detail["mfma_busy_pct"] if detail else None,
                    "issued_bf16_TFLOPs": dom.get("issued_bf16_TFLOPs"), "frac_of_bf16_peak": dom.get("frac_of_bf16_peak"),
                    "note": "bf16x3 kernel (exact 3-way bf16 split, fp32 accumulate): matrix work is 1/5 of the fp32-MFMA form, "
                            "so the kernel sits on the memory side; algorithmic bytes per launch in kernels[*].algorithmic_MB"}
        else:
            roof = {"kernel": "/".join(k for k in dom_key if k), "bound": "mfma", "achieved": dom["fp32_equiv_TFLOPs"], "peak": FP32_MFMA_PEAK,
                    "unit": "TFLOP/s", "frac": dom["fp32_equiv_TFLOPs"] / FP32_MFMA_PEAK, "traffic": traffic, "traffic_unit": "bytes/launch",
                    "traffic_detail": detail if detail else {"unavailable": why_not}}
        # BASELINE's second target: MFMA utilisation of the ConvLIF stack = the PMC's MFMA-busy of every kernel that holds a 3 x 3
        # contraction (forward diagonals / chains, fused backward, window backward, input gradients, head windows), weighted with
        # the time that kernel takes inside the replayed step.  None when the committed PMC pass is of other sources.
        stack, stack_ms, stack_busy_ms, stack_missing = {}, 0.0, 0.0, []
        for nm, ent in kernels.items():
            if ent.get("fp32_equiv_TFLOPs") is None:
                continue
            if ent.get("mfma_busy_pct") is None:
                stack_missing.append(nm)
                continue
            stack[nm] = {"ms_per_step": round(ent["total_ms_per_step"], 4), "mfma_busy_pct": ent["mfma_busy_pct"]}
            stack_ms += ent["total_ms_per_step"]
            stack_busy_ms += ent["total_ms_per_step"] * ent["mfma_busy_pct"] / 100.0
        mfma_busy_stack = (100.0 * stack_busy_ms / stack_ms) if (stack_ms > 0 and not stack_missing) else None
        # ... and the matrix work ISSUED over the whole replayed step against the dense bf16 peak (no PMC needed)
        issued_flop_step = sum(ent.get("issued_bf16_TFLOPs", 0.0) * 1e12 * ent["total_ms_per_step"] * 1e-3 for ent in kernels.values())
        out = {
            "metric": f"event-windows/sec (train step, {W}x{H}x{PASSES * EV_PER_PASS // 1000}k ev)" + ("" if args.config == "c3" else f" [{wl['model']}, {args.config}]"),
            "value": B_PER_GPU * dp.world * args.steps / elapsed,
            "unit": "event-windows/s", "n_gpus": dp.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["text"], "baseline_config": args.config,
                       "global_batch": B_PER_GPU * dp.world, "events_per_window": PASSES * EV_PER_PASS,
                       "parallelism": f"dp{dp.world}", "launch": mode, "loss": loss_val,
                       "streams": nstream,
                       "forward_launches": ("layer by layer: recurrent hidden layers one cell per launch (k_fwd_diag_t), a feed-forward hidden "
                                            "layer's passes in one launch with the state in registers (k_fwd_win_t), the head layer of all "
                                            "passes in 1; EVF_FWD_LM=0: diagonals" if (diag_fwd and fwd_lm) else
                                            "diagonal + chains: the cells of the hidden layers up to the last recurrent one in P + 3 launches "
                                            "(k_fwd_diag_t), the two feed-forward layers above them one launch each for all passes with the "
                                            "state in registers (k_fwd_win_t), the head layer of all passes in 1 (k_head_lif_fwd_win); "
                                            "EVF_FWD_LM=0: P + 5 diagonals" if (diag_fwd and fwd_mode == "top") else
                                            "diagonal: the window's hidden forward cells in P + 5 launches (k_fwd_diag, cells "
                                            "(pass, layer) with equal pass + layer together), "
                                            + ("the PLIF head layer one launch per pass" if plif_net else
                                               "the head layer of all passes in 1 (k_head_lif_fwd_win)") + "; EVF_DEFER_FWD=0: one launch per cell"
                                            if diag_fwd else "one launch per (pass, layer) cell"),
                       "backward_launches": ("the two feed-forward layers above the last recurrent one first, all passes per launch with dL/dv "
                                             "and the potential in registers (k_bwd_win_lif_top, k_bwd_win_lif) + their input gradients as one "
                                             "launch per layer (k_dgrad_diag_dma over a product list); then diagonal: fused-backward cells of "
                                             "the four layers below in P + 3 launches (k_bwd_diag), their input-gradient cells in P + 3 "
                                             "(k_dgrad_diag), the head layer's backward of all passes in 1 (k_head_bwd_win); "
                                             "EVF_LIF_BWD_TOP=0: all six hidden layers on P + 5 diagonals" if (diag_bwd and bwd_top) else
                                             "diagonal: fused-backward cells in P + 5 launches (k_bwd_diag), input-gradient cells in "
                                             "P + 5 (k_dgrad_diag), the head layer's backward of all passes in 1 (k_head_bwd_win); EVF_DEFER_BWD=0: 13 launches per pass"
                                             if diag_bwd else "one launch per cell"),
                       "pipelining": (f"each rank's {B_PER_GPU} windows as {nstream} micro-batches of {B_PER_GPU // nstream} on {nstream} HIP "
                                      "streams (replicas sharing the weights; gradients summed before the one optimizer step): "
                                      "kernels[*] / roofline are per LAUNCH of a micro-batch, timed one at a time; in the replayed "
                                      "step two such launches are in flight" if nstream > 1 else None),
                       "collective": ({"backend": dp.backend, "library": "RCCL " + ".".join(str(v) for v in torch.cuda.nccl.version())
                                       if dp.backend == "nccl" else dp.backend,
                                       "ranks": torch.distributed.get_world_size() if torch.distributed.is_initialized() else dp.world,
                                       "per_step": "1 SUM all-reduce of [flat gradient | loss | new_seq] = %d bytes" % (opt.comm.numel() * 4),
                                       # the collective's own device time (HIP events on its stream around every all-reduce of
                                       # the timed region, this rank): what a scaling run has to compare its step time with
                                       "all_reduce_us": ({"mean": float(np.mean(reduce_ms) * 1e3), "max": float(np.max(reduce_ms) * 1e3),
                                                          "n": len(reduce_ms)} if reduce_ms else None),
                                       "mode": ("captured: evf_allreduce_sum (ncclAllReduce on the library's own RCCL communicator) is a node "
                                                "of the step's ONE hipGraph" if one_graph else
                                                ("evf_allreduce_sum, eager" if dp.capturable else "torch.distributed all_reduce, eager")
                                                + (" between the step's two hipGraphs" if graphs is not None else "")),
                                       # the library's own communicator (EVF_DP_NATIVE=1, opt-in): asked for / why not used / the
                                       # ranks RCCL itself counts on it (ncclCommCount) / RCCL's version code as that binding sees it
                                       "native_requested": bool(getattr(dp, "native_requested", False)),
                                       "native_fallback": getattr(dp, "native_fallback", None),
                                       "native_comm_count": getattr(dp, "native_ranks", None),
                                       "native_rccl_version": getattr(dp, "native_version", None),
                                       "forced_at_one_rank": dp.world == 1}
                                      if dp.active else
                                      {"backend": dp.backend, "library": ("RCCL " + ".".join(str(v) for v in torch.cuda.nccl.version())
                                                                          if torch.distributed.is_nccl_available() else None),
                                       "ranks": 1, "per_step": "none at one rank (the all-reduce of [flat gradient | loss | new_seq] = "
                                                               "%d bytes is issued from 2 ranks on)" % (opt.comm.numel() * 4)}),
                       "conv_precision": ("fp32 results via exact 3-way bf16 splits of the fp32 operands on the bf16 matrix cores, "
                                          "fp32 accumulation" if model_precision == "bf16x3" else "fp32 MFMA (v_mfma_f32_32x32x2_f32)")},
            "roofline": roof,
            "mfma_stack": {"mfma_busy_stack_pct": mfma_busy_stack, "kernels": stack, "kernels_without_pmc": stack_missing,
                           "issued_bf16_frac_of_peak_whole_step": issued_flop_step / (elapsed / args.steps) / 1e12 / BF16_MFMA_PEAK,
                           "note": "sum(ms_per_step x SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE) / sum(ms_per_step) over the kernels that hold "
                                   "a 3 x 3 contraction; PMC figures from the committed pass of the same kernel sources"},
            # all modelled kernels of a step together: algorithmic bytes / step time against the HBM peak
            "step_hbm_frac": step_alg_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK,
            # the contract's timed region (block 0 = value / ms_per_step) and the same region repeated right after it
            "timing_blocks": {"ms_per_step": [round(v, 4) for v in block_ms], "steps_per_block": args.steps,
                              "median_ms_per_step": float(np.median(block_ms)), "min_ms_per_step": float(np.min(block_ms)),
                              "max_ms_per_step": float(np.max(block_ms)),
                              "spread_pct": float(100.0 * (np.max(block_ms) - np.min(block_ms)) / np.median(block_ms)),
                              "windows_per_s_median": B_PER_GPU * dp.world / (float(np.median(block_ms)) * 1e-3)},
            "workload_activity": {"thresh_scale": args.thresh_scale, "events": args.events},
            "kernels": kernels, "entry_points_that_only_record": recorded_only,
            "kernel_timing": {"method": "diagonal / head-window launches: HIP events captured into an instrumented copy of the step graphs "
                                        "(timestamp-kernel nodes), read after its replays, minus the empty bracket of the same graph; "
                                        "every other entry: HIP events around each launch on its stream over eager steps, minus the bracket "
                                        "overhead o = 2 T(1 tiny kernel) - T(2 tiny kernels) calibrated in the same run",
                              "bracket_overhead_us": round(event_overhead_us, 2),
                              "graph_profile": ({k: v for k, v in gprof.items() if k != "per_kind"} if gprof else None)},
        }
        # the two side measurements must never cost the headline line: report their failure instead
        if not args.no_iwe:
            try:
                spec = iwe_warp_bandwidth(dev, 8)
                floor, bracket = launch_floor_us()
                # what bounds the call at the spec shape: one device-scope atomic per event (the image is far too small to
                # saturate anything else) at the measured random-atomic rate, on top of the floor of a launch
                spec["launch_floor_us"] = floor
                one = os.environ.get("EVF_IWE_ONE", "0") == "1"
                spec["launches_per_call"] = 1 if one else 2
                spec["kernel"] = "k_iwe_splat_one" if one else "k_evf_fill + k_iwe_splat<true>"
                if not one:
                    spec["atomic_floor_us"] = 8 * 15000 / ATOMIC_RATE * 1e6
                spec["hbm_time_us"] = spec["algorithmic_MB"] * 1e6 / (HBM_PEAK * 1e9) * 1e6
                spec["event_pass_floor_us"] = 4.2  # rocprofv3, round 6: every event read + flow gathered + ONE 4-byte store, nothing else
                spec["note"] = ("latency / atomic-rate bound at this size: 4.4 MB is 0.55 us of HBM time; an empty kernel under the same "
                                "HIP-event bracket takes launch_floor_us, a kernel that only reads every event, gathers its flow and stores "
                                "4 bytes 4.2-5.0 us by rocprofv3 (event_pass_floor_us).  The call is a zero-fill + one scatter kernel whose "
                                "120 k device-scope atomics alone need atomic_floor_us at the measured 21.4 G atomics/s (scope does not "
                                "matter: tools/probes/atomic_probe.hip).  EVF_IWE_ONE=1: the one-launch form (entries into the output's own "
                                "memory, image built in LDS by the sample's last block, no global atomics) -- measured 10.0-11.3 us against "
                                "9.1 + 1.8, not the default (csrc/evf_events.hip, k_iwe_splat_one)")
                out["iwe_warp"] = {"spec_shape": spec, "saturating": iwe_warp_bandwidth(dev, 512, reps=5),
                                   "saturating_2048": iwe_warp_bandwidth(dev, 2048, reps=3),
                                   "empty_launch_us": {"tiny_kernel": floor, "event_bracket_overhead": bracket}}
            except Exception as e:  # noqa: BLE001
                out["iwe_warp"] = {"error": f"{type(e).__name__}: {e}"}
        if dp.world == 1 and not args.no_cpu_baseline:
            threads = args.cpu_threads or (os.cpu_count() or 1)
            try:
                out["cpu_baseline"] = cpu_baseline(threads, name=wl["model"])
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        out["parity_report"] = parity_report_status()
        if dp.world == 1 and args.config == "c3":
            # BASELINE configs[1] on the GPU side (the CPU side: cpu_baseline.extra.fwd_loss_windows_per_s): forward passes + CM loss of
            # the same windows, no backward, no optimizer step; eager launches (the recorded diagonal forward), after the timed region
            try:
                out.setdefault("other_configs", {})["c2"] = (gpu_forward_loss_line(wl, dev, pool, model_precision, capture_mode="thread_local" if dp.active else "global") if reps is None else
                                                             {"skipped": "micro-batch pipelining is on (--streams)"})
            except Exception as e:  # noqa: BLE001
                out.setdefault("other_configs", {})["c2"] = {"error": f"{type(e).__name__}: {e}"}
        if dp.world == 1 and args.config == "c3" and not args.no_others:
            # BASELINE configs[3] / configs[4] next to the headline: short runs in their own processes AFTER everything of the
            # c3 line has been measured (this process only waits meanwhile)
            torch.cuda.synchronize()
            out.setdefault("other_configs", {}).update({"c4": other_config_line("c4"), "c5": other_config_line("c5"),
                                    "note": "`python bench.py --config c4|c5 --steps 10 --warmup 3`, one process each, run after the c3 "
                                            "line's timed region and side measurements; full lines: profiles/"})
        if dp.world == 1 and args.config == "c3" and not args.no_others and not args.no_alive and args.thresh_scale == 1.0:
            # The same step on an ALIVE network (thresholds x 0.25, moving-dots events: every layer at 20-50 % spike rate; parity of
            # exactly this workload: tests/test_gpu_teacher_forced.py).  The c3 kernels have no data-dependent path, so the time must
            # not depend on the activity -- measured here, not assumed; c4's per-wave 3- / 6-term vote IS data dependent.
            torch.cuda.synchronize()
            alive = {}
            for tag, extra in (("c3_thresh_x0.25_moving_dots", ["--thresh-scale", "0.25", "--events", "moving_dots"]),
                               ("c3_thresh_x0.25_uniform", ["--thresh-scale", "0.25"]),
                               ("c4_thresh_x0.25", ["--config", "c4", "--thresh-scale", "0.25"])):
                alive[tag] = other_config_line(None, steps=20 if "c4" not in tag else 10, extra=extra)
            out["alive_workloads"] = {**alive, "default_ms_per_step": out["ms_per_step"],
                                      "note": "`python bench.py <flags> --no-cpu-baseline --no-iwe --no-others`, one process each, after the "
                                              "headline line's measurements: the replayed step at another spike activity"}
            for tag, ent in alive.items():
                out["config"]["alive_" + tag.replace(".", "") + "_ms_per_step"] = round(ent["ms_per_step"], 3) if "ms_per_step" in ent else None
        if dp.world == 1 and args.config == "c3" and not args.no_others:
            # the other neuron models of the FireNet family on the same recorded window kernels (PLIF; XLIF / ALIF: modes of the PLIF
            # kernels), at the headline shape, one short process each (tools/debug/xlif_step.py: train.GraphedWindowStep)
            import re
            import subprocess

            fam = {}
            for nm in ("PLIFFireNet", "XLIFFireNet", "ALIFFireNet"):
                try:
                    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "debug", "xlif_step.py"), nm], capture_output=True,
                                       text=True, timeout=300, cwd=ROOT)
                    m = re.search(r"(fused|general path)[^:]*: ([0-9.]+) ms per step", r.stdout)
                    fam[nm] = {"path": m.group(1), "ms_per_step": float(m.group(2))} if m else {"error": (r.stderr or r.stdout)[-300:]}
                except Exception as e:  # noqa: BLE001
                    fam[nm] = {"error": f"{type(e).__name__}: {e}"}
            out["firenet_family_at_c3_shape"] = {**fam, "note": "train step (hard reset, arctan) at 8 x 128 x 128, 10 passes x 1500 events, "
                                                 "replayed from hipGraphs; LIFFireNet = the headline line"}
            for nm, ent in fam.items():
                out["config"][nm.lower() + "_c3_shape_ms_per_step"] = ent.get("ms_per_step")
        # the driver's record keeps the scalar entries of `config` (strings cut at 120 characters) and drops nested objects: the
        # figures of the other BASELINE configurations and of the collective are repeated there in compact form
        tb = out["timing_blocks"]
        out["config"]["ms_per_step_median_of_blocks"] = round(tb["median_ms_per_step"], 4)
        out["config"]["ms_per_step_spread_pct"] = round(tb["spread_pct"], 2)
        out["config"]["timing_blocks"] = len(tb["ms_per_step"])
        out["config"]["roofline_kernel"] = roof.get("kernel")
        out["config"]["roofline_bound"] = roof.get("bound")
        out["config"]["roofline_total_ms_per_step"] = round(roof.get("total_ms_per_step", 0.0), 4)
        out["config"]["mfma_busy_stack_pct"] = round(mfma_busy_stack, 2) if mfma_busy_stack is not None else None
        out["config"]["issued_bf16_frac_of_peak_whole_step"] = round(out["mfma_stack"]["issued_bf16_frac_of_peak_whole_step"], 4)
        iw = out.get("iwe_warp") or {}
        if "spec_shape" in iw:
            # BASELINE's second headline ("IWE-warp GB/s"): compute_pol_iwe at the spec shape (8 x 15k events, 128 x 128) and at a
            # bandwidth-saturating batch, as scalars the driver's record keeps
            out["config"]["iwe_warp_spec_GBps"] = round(iw["spec_shape"]["GBps"], 1)
            out["config"]["iwe_warp_spec_us"] = round(iw["spec_shape"]["ms_per_call"] * 1e3, 2)
            out["config"]["iwe_warp_spec_frac"] = round(iw["spec_shape"]["frac_of_hbm_peak"], 4)
            out["config"]["iwe_warp_spec_launches"] = iw["spec_shape"].get("launches_per_call")
            out["config"]["iwe_warp_sat_GBps"] = round(iw["saturating_2048"]["GBps"], 1)
            out["config"]["iwe_warp_sat_frac"] = round(iw["saturating_2048"]["frac_of_hbm_peak"], 4)
        oc = out.get("other_configs", {})
        for cname in ("c2", "c4", "c5"):
            ent = oc.get(cname) or {}
            if "value" in ent:
                out["config"][f"{cname}_windows_per_s"] = round(float(ent["value"]), 1)
                ms = ent.get("ms_per_step", ent.get("ms_per_window_batch"))
                if ms is not None:
                    out["config"][f"{cname}_ms_per_step"] = round(float(ms), 3)
            elif ent:
                out["config"][f"{cname}_windows_per_s"] = None
        col = out["config"].get("collective") or {}
        ar = col.get("all_reduce_us") or {}
        # scalars of the collective's preflight (the driver's record keeps scalars only): ranks torch's process group reports,
        # ranks RCCL counts on the library's own communicator (None: not requested, EVF_DP_NATIVE=1 asks), why it fell back
        out["config"]["collective_ranks"] = col.get("ranks")
        out["config"]["collective_native_requested"] = col.get("native_requested", False)
        out["config"]["collective_native_comm_count"] = col.get("native_comm_count")
        out["config"]["collective_native_fallback"] = col.get("native_fallback")
        out["config"]["collective_short"] = ("%s, %s rank(s), %s%s" % (
            col.get("library") or col.get("backend"), col.get("ranks"), col.get("mode", "eager all-reduce between two graphs") if dp.active else "none at one rank",
            (", all-reduce mean %.1f us max %.1f us (n=%d)" % (ar["mean"], ar["max"], ar["n"])) if ar else ""))[:118]
        # ... and the full objects move to the FRONT of the line (a truncated record keeps them)
        front = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "other_configs", "collective", "config", "roofline", "cpu_baseline")
        out["collective"] = out["config"].get("collective")
        out = {**{k: out[k] for k in front if k in out}, **{k: v for k, v in out.items() if k not in front}}
        print(json.dumps(out), flush=True)
    dp.barrier()
    dp.close()


if __name__ == "__main__":
    main()
