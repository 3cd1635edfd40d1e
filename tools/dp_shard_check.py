#!/usr/bin/env python
"""HIP data-parallel check (last bullet): N ranks, each running the HIP train step on its
contiguous slot range of ONE global batch, SUM all-reduced by `DataParallel` -- against the HIP gradient of the whole
global batch computed by a single replica.  Checked on every rank:

  * flat gradient after the all-reduce == unsharded gradient (rel-L2, the weight-gradient slabs are summed in a
    different order),
  * the loss in the buffer's tail == the unsharded loss (the reference loss sums over the batch, loss/flow.py:226,259,289),
  * this rank's recurrent states (v, spikes) == the [lo:hi) slot slice of the unsharded states, bit for bit,
  * parameters after clip + Adam on the reduced buffer == parameters after the unsharded step (train_flow.py:157-163).

    python tools/dp_shard_check.py --ranks 2            # starts its own ranks (torch.distributed.run, 127.0.0.1)

On a one-GPU box the ranks share the device (EVF_BENCH_SINGLE_DEVICE=1) and gloo carries the buffer
(EVF_DP_BACKEND=gloo); on a multi-GPU node the same code runs one rank per GPU over RCCL.
Prints one JSON line on rank 0; exit status != 0 when a rank's check fails."""

import argparse
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def launch(n):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--per-rank", type=int, default=2)
    ap.add_argument("--res", type=int, default=64)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--events", type=int, default=800)
    ap.add_argument("--thresh-scale", type=float, default=0.25)
    ap.add_argument("--model", default="LIFFireNet")
    ap.add_argument("--tol", type=float, default=1e-5)
    args = ap.parse_args()
    if args.ranks > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch(args.ranks))

    import numpy as np
    import torch

    from event_flow_amd import _lib, synthetic
    from event_flow_amd.loss.flow import EventWarping
    from event_flow_amd.models import model as models
    from event_flow_amd.parallel import DataParallel
    from event_flow_amd.train import FlatAdam, encode_passes, window_apply, window_backward

    _lib.load()
    local_rank = 0 if os.environ.get("EVF_BENCH_SINGLE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    dp = DataParallel(device=dev)
    assert dp.world == args.ranks, (dp.world, args.ranks)
    H = W = args.res
    G = args.per_rank * dp.world
    lo, hi = dp.shard(G)
    neuron = ({"leak": [-4.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True, "learn_thresh": True, "hard_reset": True}
              if args.model == "LIFFireNet" else
              {"leak_v": [-4.0, 0.1], "leak_pt": [-4.0, 0.1], "add_pt": [-2.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True,
               "learn_thresh": True, "hard_reset": True})
    cfg = {"num_bins": 2, "base_num_channels": 32, "kernel_size": 3, "encoding": "cnt", "norm_input": False, "mask_output": True,
           "activations": ["arctanspike", "arctanspike"], "spiking_neuron": neuron}
    lcfg = {"loader": {"resolution": [H, W]}, "loss": {"flow_regul_weight": 0.001, "overwrite_intermediate": False, "clip_grad": 100.0},
            "model": {"mask_output": True}}
    # the same global batch on every rank (seeded per slot); a rank trains on slots [lo, hi)
    lists = [torch.from_numpy(synthetic.event_list_batch(G, args.events, H, W, 7000 + 1000 * k)).to(dev) for k in range(args.passes)]

    def replica():
        torch.manual_seed(0)  # identical replicas
        m = getattr(models, args.model)(dict(cfg)).to(dev)
        with torch.no_grad():
            for k, p in m.named_parameters():
                if k.endswith("thresh"):
                    p.mul_(args.thresh_scale)
        m.train()
        opt = FlatAdam(m, lr=2e-4, clip=100.0)
        opt.zero_grad()
        return m, EventWarping(lcfg, dev), opt

    def passes_of(a, b):
        ps = encode_passes([ev[a:b].contiguous() for ev in lists], 2, (H, W), want=("cnt", "mask", "pol"))
        for d in ps:
            d["event_voxel"] = None
        return ps

    # --- sharded: this rank's slots, one SUM all-reduce -----------------------------------------------------------
    m_s, l_s, o_s = replica()
    local = window_backward(m_s, l_s, o_s, passes_of(lo, hi), dp)
    dp.reduce(o_s.comm)
    torch.cuda.synchronize()
    g_s = o_s.flat_grad.detach().cpu().numpy().copy()
    loss_s = float(o_s.comm[o_s.n])
    st_s = [s.detach().cpu().numpy().copy() for s in m_s.states]
    window_apply(m_s, l_s, o_s, local, dp)
    p_s = o_s.flat_param.detach().cpu().numpy().copy()
    gn_s = o_s.grad_norm()

    # --- unsharded: one replica, the whole global batch ---------------------------------------------------------------
    m_u, l_u, o_u = replica()
    loss_u_t = window_backward(m_u, l_u, o_u, passes_of(0, G), None)
    torch.cuda.synchronize()
    g_u = o_u.flat_grad.detach().cpu().numpy().copy()
    loss_u = float(loss_u_t.detach())
    st_u = [s.detach().cpu().numpy().copy() for s in m_u.states]
    window_apply(m_u, l_u, o_u, loss_u_t, None)
    p_u = o_u.flat_param.detach().cpu().numpy().copy()
    gn_u = o_u.grad_norm()

    rel = float(np.linalg.norm(g_s - g_u) / max(np.linalg.norm(g_u), 1e-30))
    # states: [2(+1), B, C, H, W]; this rank's slots against the slice of the unsharded run
    state_equal = all(np.array_equal(a, b[:, lo:hi]) for a, b in zip(st_s, st_u))
    nspk = int(sum(b[1].sum() for b in st_u))
    # the first Adam step moves a weight by ~lr*sign(g): weights whose gradient sits at the summation-order noise
    # floor may move the other way (<= 2 lr apart); everything else must agree to round-off
    dparam = float(np.abs(p_s - p_u).max())
    frac_moved = float(np.mean(np.abs(p_s - p_u) > 1e-7))
    res = {"ranks": dp.world, "backend": dp.backend, "global_batch": G, "shard": [lo, hi], "grad_rel_l2": rel, "loss_sharded": loss_s,
           "loss_unsharded": loss_u, "states_bit_equal": bool(state_equal), "spikes_in_last_state": nspk, "grad_norm": [gn_s, gn_u],
           "max_param_diff_after_step": dparam,
           "frac_params_differing": frac_moved, "model": args.model}
    ok = (rel <= args.tol and abs(loss_s - loss_u) <= 1e-5 * abs(loss_u) and state_equal and nspk > 0
          and abs(gn_s - gn_u) <= 1e-5 * gn_u and dparam <= 4.1e-4 and frac_moved <= 1e-3)
    bad = dp.max_over_ranks(0.0 if ok else 1.0)
    if dp.rank == 0:
        res["ok_all_ranks"] = bad == 0.0
        print(json.dumps(res), flush=True)
    elif not ok:
        print(f"[rank {dp.rank}] FAILED {json.dumps(res)}", file=sys.stderr, flush=True)
    dp.barrier()
    dp.close()
    raise SystemExit(0 if bad == 0.0 else 1)


if __name__ == "__main__":
    main()
