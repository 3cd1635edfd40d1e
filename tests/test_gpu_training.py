"""End-to-end: the whole path (binning -> spiking forward -> contrast-maximisation loss -> BPTT -> clip+Adam)
learns.  Self-supervised training on synthetic moving dots with a known per-sample motion must lower the
loss and bring the predicted flow closer to the true motion than a zero-flow prediction."""

import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


@pytest.mark.parametrize("model", ["LIFFireNet", "PLIFFireNet"])
def test_training_on_moving_dots_learns_the_motion(model):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_demo.py"), "--model", model, "--steps", "1200"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["loss_last_100"] < 0.85 * res["loss_first_100"], res
    # (the trajectory is chaotic -- spike flips, atomic summation order: 0.68-0.90 px were observed for the same seed)
    assert res["aee_after"] < 0.85 * res["aee_zero_flow"], res
    assert res["aee_after"] < res["aee_before"], res


def _free_port():
    """A TCP port nobody listens on right now (fixed port numbers collided between tests: a rendezvous port of an earlier
    test still in TIME_WAIT made torch.distributed.run fail now and then)."""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_reference_shaped_drivers_run_end_to_end(tmp_path):
    """train_flow.py / eval_flow.py counterparts (reference train_flow.py:38-194, eval_flow.py:40-258) on the synthetic
    loader: training writes a checkpoint, evaluation loads it and reports FWL / RSAT / AEE."""
    w = str(tmp_path / "m.pth")
    for extra in (["--fused-optimizer"], []):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "train_flow.py"), "--synthetic", "--epochs", "2", "--out", w] + extra,
                             capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        res = json.loads(out.stdout.strip().splitlines()[-1])
        assert len(res["loss_per_epoch"]) == 2 and all(0 < v < 10 for v in res["loss_per_epoch"])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "eval_flow.py"), "--synthetic", "--weights", w, "--store", str(tmp_path)],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert {"FWL", "RSAT", "AEE", "iwe_variance"} <= set(res) and all(v == v for v in res.values() if isinstance(v, float))
    stored = list((tmp_path / "results" / "eval_0").glob("*/flow/*.png"))
    assert stored and len(stored) == len(list((tmp_path / "results" / "eval_0").glob("*/iwe/*.png")))


def test_drivers_on_sequence_files(tmp_path):
    """The same drivers fed by the sequence loader (dataloader/h5.py; `.npz` flavour of the reference's HDF5 layout):
    training over 5 moving-dots sequences with augmentation and the hot-pixel filter on, then AEE against the stored
    ground-truth flow maps in mode gtflow_dt1."""
    import numpy as np
    import yaml

    from event_flow_amd import synthetic
    from event_flow_amd.dataloader.h5 import write_npz_sequence

    H = W = 64
    data = tmp_path / "data"
    data.mkdir()
    for i in range(5):
        xs, ys, ts, ps, (u, v) = synthetic.moving_dots_events(9000, H, W, 300 + i, max_disp=36.0)
        ts = ts * 0.6 + 5.0
        stamps = 5.0 + 0.1 * np.arange(7)
        gt = np.zeros((2, H, W), np.float32)
        gt[0], gt[1] = u / 6, v / 6  # pixels per 0.1 s interval
        maps = [(f"{k:06d}", stamps[k], gt) for k in range(7)]
        write_npz_sequence(str(data / f"seq{i}.npz"), xs.astype(np.int16), ys.astype(np.int16), ts, (ps > 0).astype(np.int8),
                           flow_dt1=maps)
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "train_SNN.yml")))
    cfg["data"].update(path=str(data), window=1000, window_loss=3000)
    cfg["loader"].update(batch_size=2, n_epochs=2, augment=["Horizontal", "Vertical", "Polarity"], augment_prob=[0.5, 0.5, 0.5])
    cfg["hot_filter"] = {"enabled": True, "max_px": 100, "min_obvs": 5, "max_rate": 0.8}
    tcfg = str(tmp_path / "train.yml")
    yaml.safe_dump(cfg, open(tcfg, "w"))
    w = str(tmp_path / "m.pth")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "train_flow.py"), "--config", tcfg, "--epochs", "2", "--out", w,
                          "--fused-optimizer"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert len(res["loss_per_epoch"]) == 2 and all(0 < v < 10 for v in res["loss_per_epoch"])
    ecfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "eval_flow.yml")))
    ecfg["data"].update(path=str(data), mode="gtflow_dt1", window=1, window_eval=1000)
    ecfg["metrics"]["name"] = ["AEE"]
    epath = str(tmp_path / "eval.yml")
    yaml.safe_dump(ecfg, open(epath, "w"))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "eval_flow.py"), "--config", epath, "--train-config", tcfg,
                          "--weights", w], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["AEE"] == res["AEE"] and 0 <= res["AEE"] < 50 and 0 <= res["AEE_percent_outliers"] <= 1


def test_train_driver_two_ranks_on_sequence_files(tmp_path):
    """train_flow.py under torch.distributed.run with 2 ranks (sharing the test box's one GPU; gloo carries the
    collectives): the sequence files are sharded over the ranks, the ranks reset / end their epochs together and
    make one SUM all-reduce of the flat gradient per optimizer step."""
    import numpy as np
    import yaml

    from event_flow_amd import synthetic
    from event_flow_amd.dataloader.h5 import write_npz_sequence

    data = tmp_path / "data"
    data.mkdir()
    for i in range(5):  # 3 files for rank 0, 2 for rank 1, of different lengths: rank 1 finishes its pass first
        n = 6000 + 1500 * (i % 2)
        xs, ys, ts, ps, _ = synthetic.moving_dots_events(n, 64, 64, 400 + i, max_disp=30.0)
        write_npz_sequence(str(data / f"seq{i}.npz"), xs.astype(np.int16), ys.astype(np.int16), ts * 0.5 + 1.0, (ps > 0).astype(np.int8))
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "train_SNN.yml")))
    cfg["data"].update(path=str(data), window=1000, window_loss=2000)
    cfg["loader"].update(batch_size=1, n_epochs=2)
    tcfg = str(tmp_path / "train.yml")
    yaml.safe_dump(cfg, open(tcfg, "w"))
    w = str(tmp_path / "m.pth")
    env = dict(os.environ, EVF_DP_BACKEND="gloo", EVF_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "train_flow.py"), "--config", tcfg, "--epochs", "2", "--out", w,
           "--fused-optimizer"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["ranks"] == 2 and len(res["loss_per_epoch"]) == 2 and all(0 < v < 10 for v in res["loss_per_epoch"])
    assert os.path.exists(w)


def _bench_two_ranks(extra, port, warmup=2):
    """bench.py under torch.distributed.run with 2 ranks sharing the one GPU of the test box (gloo moves the
    flat gradient buffer; on a multi-GPU node the same code path runs over RCCL)."""
    env = dict(os.environ, EVF_DP_BACKEND="gloo", EVF_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", str(warmup),
           "--no-cpu-baseline", "--no-iwe"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_bench_data_parallel_step_two_ranks_graph_equals_eager():
    """The multi-rank step (two hipGraphs around one eager all-reduce) must give the loss of the eager
    multi-rank step: same windows, same replicas, gradients summed over both ranks."""
    g = _bench_two_ranks([], _free_port())
    e = _bench_two_ranks(["--no-graph"], _free_port(), warmup=4)  # graph mode adds 2 replay warm-up steps: same 7 updates
    assert g["n_gpus"] == 2 and g["config"]["launch"] == "hipgraph" and g["config"]["parallelism"] == "dp2", g
    assert e["config"]["launch"] == "eager", e
    assert g["config"]["global_batch"] == 16, g
    lg, le = g["config"]["loss"], e["config"]["loss"]
    assert lg == lg and abs(lg - le) <= 2e-3 * abs(le), (lg, le)


def _bench_rccl_one_rank(extra, port, **more_env):
    """bench.py under torch.distributed.run with ONE rank and EVF_DP_FORCE=1: backend "nccl" (= RCCL) on the one GPU."""
    env = dict(os.environ, EVF_DP_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **more_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "EVF_DP_BACKEND", "EVF_BENCH_SINGLE_DEVICE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
                          "--no-cpu-baseline", "--no-iwe", "--no-others"] + extra,
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, "\n".join(ln for ln in out.stderr.splitlines() if "frame #" not in ln)[-6000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_rccl_code_path_on_one_rank_graph_equals_eager_and_plain_step():
    """The code the multi-GPU run takes, on the one GPU there is: DataParallel(backend="nccl") initialised with device_id,
    the step as TWO hipGraphs (thread_local capture) with the RCCL all-reduce launched eagerly between them,
    barrier(device_ids), max_over_ranks -- forced at world size 1 (EVF_DP_FORCE=1).  A one-rank SUM all-reduce is the
    identity, so the loss after the same number of updates must equal the plain one-GPU run's (graph and eager)."""
    g = _bench_rccl_one_rank([], _free_port(), EVF_DP_NATIVE="1")
    e = _bench_rccl_one_rank(["--no-graph", "--warmup", "4"], _free_port(), EVF_DP_NATIVE="1")  # graph mode adds 2 replay warm-up steps: same 8 updates
    d = _bench_rccl_one_rank([], _free_port())  # the DEFAULT since round 6: torch.distributed's all-reduce between two graphs
    col = g["config"]["collective"]
    assert col["backend"] == "nccl" and col["library"].startswith("RCCL") and col["ranks"] == 1 and col["forced_at_one_rank"], col
    # (EVF_DP_NATIVE=1: the collective is evf_allreduce_sum on the library's own RCCL communicator, a node of the step's ONE graph)
    assert col["mode"].startswith("captured") and col["all_reduce_us"] and col["all_reduce_us"]["n"] > 0, col
    assert col["native_requested"] and col["native_comm_count"] == 1 and g["config"]["collective_native_comm_count"] == 1, col
    assert "evf_allreduce_sum" in e["config"]["collective"]["mode"], e["config"]["collective"]
    cd = d["config"]["collective"]
    assert cd["mode"].startswith("torch.distributed all_reduce, eager between") and not cd["native_requested"] and cd["native_fallback"] is None, cd
    assert cd["all_reduce_us"] and cd["all_reduce_us"]["n"] > 0 and d["config"]["collective_ranks"] == 1, cd
    assert g["config"]["launch"] == "hipgraph" and e["config"]["launch"] == "eager", (g["config"], e["config"])
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "EVF_DP_FORCE", "EVF_DP_BACKEND"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-iwe",
                          "--no-others"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    p = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert p["config"]["collective"]["ranks"] == 1 and not p["config"]["collective"].get("forced_at_one_rank"), p["config"]
    lg, le, lp, ld = g["config"]["loss"], e["config"]["loss"], p["config"]["loss"], d["config"]["loss"]
    print("loss: rccl graph", lg, "rccl eager", le, "plain", lp, "default (torch all-reduce between two graphs)", ld)
    assert lg == lg and abs(lg - le) <= 2e-3 * abs(le) and abs(lg - lp) <= 2e-3 * abs(lp) and abs(ld - lp) <= 2e-3 * abs(lp), (lg, le, lp, ld)


@pytest.mark.parametrize("stage", ["capture"])
def test_own_rccl_communicator_that_fails_its_preflight_falls_back_to_torch(stage):
    """parallel.DataParallel._init_native votes after every stage (load, ncclCommInitRank, an eager SUM of known values, the same
    SUM as a node of a replayed hipGraph): a failure anywhere (injected here) leaves EVERY rank on torch.distributed's
    all_reduce between the step's two graphs -- the run goes on, the bench line says why, the loss is the captured run's."""
    g = _bench_rccl_one_rank([], _free_port(), EVF_DP_NATIVE="1")
    f = _bench_rccl_one_rank([], _free_port(), EVF_DP_NATIVE="1", EVF_DP_NATIVE_INJECT=stage)
    cg, cf = g["config"]["collective"], f["config"]["collective"]
    assert cg["mode"].startswith("captured") and cg["native_fallback"] is None, cg
    assert cf["mode"].startswith("torch.distributed all_reduce, eager between") and stage in cf["native_fallback"], cf
    lg, lf = g["config"]["loss"], f["config"]["loss"]
    assert lg == lg and abs(lg - lf) <= 2e-3 * abs(lg), (lg, lf)


def test_two_graph_rccl_step_is_bitwise_the_one_graph_step():
    """What a rank replays in a multi-GPU run, forced at world size 1 -- (b) ONE hipGraph with evf_allreduce_sum captured as a node,
    (c) two hipGraphs around that collective launched eagerly, (d) two hipGraphs around torch.distributed's all_reduce -- leaves
    EXACTLY the parameters, Adam moments and recurrent states of the single-GPU one-graph step (deterministic loss, no clipping):
    tools/dp_two_graph_check.py in its own process (nccl process group)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "EVF_DP_FORCE", "EVF_DP_BACKEND"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dp_two_graph_check.py")], capture_output=True, text=True,
                         timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    print(res)
    assert res["one_graph_step_is_one_graph"] and res["forced_step_is_two_graphs"] and res["backend"] == "nccl", res
    # the N-rank step with evf_allreduce_sum (the library's own RCCL communicator) captured: ONE graph; torch's collective is not capturable
    assert res["captured_rccl_step_is_one_graph"] and res["torch_path_not_capturable"] and res["rccl_version"], res
    assert res["updates"] == [6.0] * 4 and max(res["grad_norm"]) < 100.0 and res["trained"], res  # (2 eager + 4 replayed, unclipped)
    assert res["params_bitwise_equal"] and res["moments_bitwise_equal"] and res["states_bitwise_equal"], res


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (how the driver invokes it) re-executes itself under
    torch.distributed.run with one rank per GPU and prints ONE JSON line with n_gpus = 2."""
    env = dict(os.environ, EVF_DP_BACKEND="gloo", EVF_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                          "--no-cpu-baseline", "--no-iwe"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["parallelism"] == "dp2" and res["config"]["global_batch"] == 16, res
    assert res["config"]["collective"]["ranks"] == 2 and res["value"] > 0


@pytest.mark.parametrize("ranks,model", [(2, "LIFFireNet"), (4, "LIFFireNet"), (2, "PLIFFireNet")])
def test_hip_sharded_gradient_equals_hip_global_batch_gradient(ranks, model):
    """N ranks each running the HIP step on their slot range + ONE SUM all-reduce == the HIP gradient of the
    global batch on one replica (rel-L2 <= 1e-5), same loss, bit-equal per-slot states, same parameters after clip+Adam."""
    env = dict(os.environ, EVF_DP_BACKEND="gloo", EVF_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dp_shard_check.py"), "--ranks", str(ranks), "--model", model],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    print(res)
    assert res["ok_all_ranks"] and res["ranks"] == ranks and res["grad_rel_l2"] <= 1e-5 and res["states_bit_equal"], res


@pytest.mark.parametrize("model_name,H,W,per_rank", [("LIFFireNet", 128, 128, 8), ("PLIFFireNet", 260, 346, 4)])
def test_eight_emulated_ranks_sum_to_the_global_batch_gradient_at_the_baseline_shapes(model_name, H, W, per_rank):
    """SURVEY 8(e), last bullet, at BASELINE configs[2] / configs[4]: 8 fake ranks on one device -- each runs the HIP window
    (10 passes x 1500 events) on its contiguous slot range of ONE global batch (8 x 8 = 64 windows at 128 x 128 for the
    LIF-FireNet, 8 x 4 = 32 at 260 x 346 for the PLIF-FireNet), gradients and losses accumulated with SUM as the all-reduce
    would -- against the HIP step of the whole global batch on one replica: gradient rel-L2 <= 1e-5, same loss, the slots'
    recurrent states bit for bit, the same parameters after clip + Adam on the summed buffer (reference: loss sums over the
    batch, loss/flow.py:226,259,289; clip + Adam train_flow.py:157-163)."""
    from event_flow_amd import synthetic
    from event_flow_amd.loss.flow import EventWarping
    from event_flow_amd.models import model as models
    from event_flow_amd.train import FlatAdam, encode_passes, window_apply, window_backward

    ranks, P, n = 8, 10, 1500
    Gb = ranks * per_rank
    neuron = ({"leak": [-4.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True, "learn_thresh": True, "hard_reset": True}
              if model_name == "LIFFireNet" else
              {"leak_v": [-4.0, 0.1], "leak_pt": [-4.0, 0.1], "add_pt": [-2.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True,
               "learn_thresh": True, "hard_reset": True})
    cfg = {"num_bins": 2, "base_num_channels": 32, "kernel_size": 3, "encoding": "cnt", "norm_input": False, "mask_output": True,
           "activations": ["arctanspike", "arctanspike"], "spiking_neuron": neuron}
    lcfg = {"loader": {"resolution": [H, W]}, "loss": {"flow_regul_weight": 0.001, "overwrite_intermediate": False, "clip_grad": 100.0},
            "model": {"mask_output": True}}
    # per-slot seeds as bench.make_windows builds a rank's windows: the global batch is the ranks' batches side by side
    lists = [torch.from_numpy(np.concatenate([synthetic.event_list_batch(per_rank, n, H, W, synthetic.seed_for(3, r, 0) + 1000 * k)
                                              for r in range(ranks)], 0)).to(DEV) for k in range(P)]

    def replica():
        torch.manual_seed(0)
        m = getattr(models, model_name)(dict(cfg)).to(DEV)
        with torch.no_grad():
            for k, p in m.named_parameters():
                if k.endswith("thresh"):
                    p.mul_(0.25)  # an alive network: every layer spikes, most weights carry gradient
        m.train()
        opt = FlatAdam(m, lr=2e-4, clip=100.0)
        opt.zero_grad()
        return m, EventWarping(lcfg, DEV), opt

    def passes_of(a, b):
        ps = encode_passes([ev[a:b].contiguous() for ev in lists], 2, (H, W), want=("cnt", "mask", "pol"))
        for d in ps:
            d["event_voxel"] = None
        return ps

    m_u, l_u, o_u = replica()
    loss_u = float(window_backward(m_u, l_u, o_u, passes_of(0, Gb), None).detach())
    torch.cuda.synchronize()
    g_u = o_u.flat_grad.clone()
    st_u = [s.clone() for s in m_u.states]
    window_apply(m_u, l_u, o_u, torch.zeros((), device=DEV), None)
    p_u, gn_u = o_u.flat_param.clone(), o_u.grad_norm()
    del m_u, l_u, o_u

    m_s, l_s, o_s = replica()
    g_sum = torch.zeros_like(g_u, dtype=torch.float64)
    loss_s, states_equal = 0.0, True
    for r in range(ranks):
        lo, hi = r * per_rank, (r + 1) * per_rank
        m_s.reset_states()
        o_s.zero_grad()
        loss_s += float(window_backward(m_s, l_s, o_s, passes_of(lo, hi), None).detach())
        torch.cuda.synchronize()
        g_sum += o_s.flat_grad.double()
        states_equal = states_equal and all(bool(torch.equal(a, b[:, lo:hi])) for a, b in zip(m_s.states, st_u))
        m_s.detach_states()
        l_s.reset()
    rel = float((g_sum - g_u.double()).norm() / g_u.double().norm())
    nspk = int(sum(float(b[1].sum()) for b in st_u))
    # clip + Adam on the summed buffer (what every rank does after the all-reduce)
    o_s.zero_grad()
    o_s.mark_grad_dirty()
    o_s.flat_grad.copy_(g_sum.float())
    window_apply(m_s, l_s, o_s, torch.zeros((), device=DEV), None)
    dparam = float((o_s.flat_param - p_u).abs().max())
    frac = float(((o_s.flat_param - p_u).abs() > 1e-7).float().mean())
    print(f"[{model_name} 8 x {per_rank} = {Gb} windows at {H}x{W}] gradient rel-L2 {rel:.2e}, loss {loss_s:.6f} vs {loss_u:.6f}, "
          f"spikes in the last state {nspk}, grad norm {o_s.grad_norm():.5f} vs {gn_u:.5f}, max parameter difference {dparam:.2e}, "
          f"fraction of parameters differing {frac:.2e}")
    assert nspk > 0 and states_equal
    assert rel <= 1e-5 and abs(loss_s - loss_u) <= 1e-5 * abs(loss_u), (rel, loss_s, loss_u)
    assert abs(o_s.grad_norm() - gn_u) <= 1e-5 * gn_u and dparam <= 4.1e-4 and frac <= 1e-3, (dparam, frac)


@pytest.mark.parametrize("cfg,batch", [("c3", 64), ("c5", 32)])
def test_bench_eight_rank_launch_path_dry_run(cfg, batch):
    """`python bench.py --gpus 8 --config c3|c5 --dry-run-launch`: the argument / launcher path of the 8-GPU scaling run up to
    the first collective.  On this one-GPU box it must (a) REFUSE loudly, naming the device count, without starting anything,
    and (b) with the test hook that lets ranks share the device (gloo) start 8 ranks, bring the process group up with 8 ranks,
    SUM all-reduce a known buffer (36 = 1 + .. + 8 everywhere) and report BASELINE's global batch (8 x 8 = 64 / 8 x 4 = 32)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "EVF_DP_BACKEND", "EVF_BENCH_SINGLE_DEVICE", "EVF_DP_FORCE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--config", cfg, "--dry-run-launch"]
    if torch.cuda.device_count() < 8:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
        assert out.returncode != 0 and "needs 8 visible GPUs" in out.stderr and "nothing launched" in out.stderr, (out.returncode, out.stderr[-500:])
    env.update(EVF_DP_BACKEND="gloo", EVF_BENCH_SINGLE_DEVICE="1", OMP_NUM_THREADS="2")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    if out.returncode != 0:  # (eight fresh processes rendezvous on a just-freed port: one more attempt on a new port before it counts)
        print("first attempt failed:", out.stderr[-1500:])
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    res = json.loads(lines[0])
    print(res)
    assert res["dry_run_launch"] and res["n_gpus"] == 8 and res["ranks_in_process_group"] == 8 and res["global_batch"] == batch, res
    assert res["ok_all_ranks"] and res["sum_allreduce_of_rank_plus_1"] == 36.0 and res["baseline_config"] == cfg, res
