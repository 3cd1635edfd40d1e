"""Data-parallel path on CPU: world_size 2 over gloo.  Each rank computes the
(oracle) gradient of its batch shard; DataParallel.all_reduce_grads must give
the single-process gradient of the global batch (SUM semantics -- the reference
loss sums over the batch, loss/flow.py:226,259,289), the global loss and the
OR of the new_seq flags in ONE collective."""

import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

B, N, H, W, P = 4, 120, 16, 16, 2


def _problem():
    from event_flow_amd import synthetic
    from oracle import encodings as oenc
    from oracle import snn as osnn

    gen = torch.Generator().manual_seed(3)
    params = osnn.make_firenet_params("LIFFireNet", gen, neuron={"leak": (-4.0, 0.1), "thresh": (0.15, 0.03)})
    keys = osnn.trainable_keys(params)
    passes = []
    for k in range(P):
        ev = synthetic.event_list_batch(B, N, H, W, 77 + 10 * k)
        d = oenc.collate([oenc.encode_window(ev[b, :, 2], ev[b, :, 1], ev[b, :, 0], ev[b, :, 3], 2, (H, W)) for b in range(B)])
        passes.append({k2: torch.from_numpy(v) for k2, v in d.items()})
    return params, keys, passes


def _grad(params, keys, passes, lo, hi):
    from oracle import train as otrain

    sub = [{k: v[lo:hi] for k, v in d.items()} for d in passes]
    loss, grads, _, _ = otrain.train_step("LIFFireNet", params, keys, sub, [None] * 7, (H, W), {"step": 0, "m": {}, "v": {}},
                                          loss_cfg={"flow_regul_weight": 0.001, "mask_output": True})
    return loss, torch.cat([grads[k].reshape(-1) for k in keys])


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from event_flow_amd.parallel import DataParallel

    dp = DataParallel(backend="gloo")
    params, keys, passes = _problem()
    lo, hi = dp.shard(B)
    loss, flat = _grad(params, keys, passes, lo, hi)
    comm = torch.zeros(flat.numel() + DataParallel.TAIL)
    comm[: flat.numel()] = flat
    gl, flag = dp.all_reduce_grads(comm, torch.tensor(loss), new_seq=(rank == 1))
    t = dp.max_over_ranks(float(rank + 1))
    flags = dp.any_flags([rank == 1, False, rank == 0])  # loader events every rank must act on (train_flow.py driver)
    w = torch.full((3,), float(rank + 5))
    dp.broadcast(w)  # replicas start from rank 0's parameters
    dp.barrier()
    if rank == 0:
        q.put((comm.numpy().copy(), float(gl), float(flag), t, (lo, hi)))
    else:
        q.put(("r1", flags, w.tolist()))
    dp.close()


def test_two_rank_allreduce_equals_global_batch_gradient():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300), q.get(timeout=300)]
    comm, gl, flag, tmax, shard0 = next(g for g in got if len(g) == 5)
    _, flags1, w1 = next(g for g in got if len(g) == 3)
    assert flags1 == [True, False, True] and w1 == [5.0, 5.0, 5.0]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.set_num_threads(2)
    params, keys, passes = _problem()
    loss, flat = _grad(params, keys, passes, 0, B)
    ref = flat.numpy()
    assert shard0 == (0, 2)
    assert np.linalg.norm(comm[:-2] - ref) <= 1e-4 * np.linalg.norm(ref)
    np.testing.assert_allclose(gl, loss, rtol=1e-5)
    assert flag == 1.0 and tmax == 2.0


def test_single_process_dp_is_identity():
    from event_flow_amd.parallel import DataParallel

    os.environ.pop("WORLD_SIZE", None)
    os.environ.pop("RANK", None)
    dp = DataParallel(backend="gloo")
    comm = torch.arange(6, dtype=torch.float32)
    l, f = dp.all_reduce_grads(comm, torch.tensor(2.5))
    assert float(l) == 2.5 and float(f) == 0.0 and torch.equal(comm[:4], torch.arange(4, dtype=torch.float32))
    assert dp.shard(8) == (0, 8) and dp.any_flags([True, False]) == [True, False]
