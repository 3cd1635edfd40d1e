"""End-to-end: the whole path (binning -> spiking forward -> contrast-maximisation loss -> BPTT -> clip+Adam)
learns.  Self-supervised training on synthetic moving dots with a known per-sample motion must lower the
loss and bring the predicted flow closer to the true motion than a zero-flow prediction."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    g = _bench_rccl_one_rank([], _free_port())
    e = _bench_rccl_one_rank(["--no-graph", "--warmup", "4"], _free_port())  # graph mode adds 2 replay warm-up steps: same 8 updates
    col = g["config"]["collective"]
    assert col["backend"] == "nccl" and col["library"].startswith("RCCL") and col["ranks"] == 1 and col["forced_at_one_rank"], col
    # (round 5: the collective is evf_allreduce_sum on the library's own RCCL communicator, a node of the step's ONE graph)
    assert col["mode"].startswith("captured") and col["all_reduce_us"] and col["all_reduce_us"]["n"] > 0, col
    assert "evf_allreduce_sum" in e["config"]["collective"]["mode"], e["config"]["collective"]
    assert g["config"]["launch"] == "hipgraph" and e["config"]["launch"] == "eager", (g["config"], e["config"])
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "EVF_DP_FORCE", "EVF_DP_BACKEND"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-iwe",
                          "--no-others"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    p = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert p["config"]["collective"]["ranks"] == 1 and not p["config"]["collective"].get("forced_at_one_rank"), p["config"]
    lg, le, lp = g["config"]["loss"], e["config"]["loss"], p["config"]["loss"]
    print("loss: rccl graph", lg, "rccl eager", le, "plain", lp)
    assert lg == lg and abs(lg - le) <= 2e-3 * abs(le) and abs(lg - lp) <= 2e-3 * abs(lp), (lg, le, lp)


@pytest.mark.parametrize("stage", ["capture"])
def test_own_rccl_communicator_that_fails_its_preflight_falls_back_to_torch(stage):
    """parallel.DataParallel._init_native votes after every stage (load, ncclCommInitRank, an eager SUM of known values, the same
    SUM as a node of a replayed hipGraph): a failure anywhere (injected here) leaves EVERY rank on torch.distributed's
    all_reduce between the step's two graphs -- the run goes on, the bench line says why, the loss is the captured run's."""
    g = _bench_rccl_one_rank([], _free_port())
    f = _bench_rccl_one_rank([], _free_port(), EVF_DP_NATIVE_INJECT=stage)
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
