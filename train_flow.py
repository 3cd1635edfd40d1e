"""Training driver -- counterpart of the reference's train_flow.py:38-194 on the MI355X path: the same YAML keys,
the same loop (forward per input window, `event_flow_association`, loss / backward / clip / optimizer step once
`window_loss` events are collected, `detach_states`, `reset`), the models / loss of `event_flow_amd`.

Differences, all host side: MLflow and the visualiser are optional extras of the reference and not used here (metrics
go to stdout and `--out` receives the checkpoint).  Data: the sequence files under `data.path` (HDF5 through h5py, or
the `.npz` flavour of the same layout, event_flow_amd/dataloader/h5.py), or `--synthetic` windows (moving dots with
known motion).  `--fused-optimizer` swaps `clip_grad_norm_` + `torch.optim.Adam` for the fused flat-buffer kernel
(same update rule).

  python train_flow.py --config configs/train_SNN.yml --synthetic [--epochs 5] [--out model.pth]
"""

import argparse
import json
import os
import time

import torch

from event_flow_amd.configs.parser import YAMLParser
from event_flow_amd.loss.flow import EventWarping
from event_flow_amd.models.model import MODELS
from event_flow_amd.parallel import DataParallel
from event_flow_amd.train import FlatAdam
from event_flow_amd.utils.utils import load_model


def train(args, config_parser):
    config = config_parser.config
    if config["data"]["mode"] == "frames":
        print("Config error: Training pipeline not compatible with frames mode.")
        raise AttributeError
    config = config_parser.combine_entries(config)
    device = config_parser.device
    if torch.device(device).type == "cuda":
        torch.cuda.set_device(device)  # the evf_* launches go to the current device's stream (loader.gpu may not be 0)

    # data parallel (one process per GPU under torch.distributed.run): sequence files sharded over the ranks, one SUM
    # all-reduce of the flat gradient per optimizer step (event_flow_amd/parallel.py)
    dp = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        if not args.fused_optimizer:
            raise SystemExit("data-parallel training reduces the flat gradient buffer: add --fused-optimizer")
        if not os.environ.get("EVF_BENCH_SINGLE_DEVICE"):  # (test hook: all ranks on the one GPU of the box)
            device = torch.device(f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}")
            torch.cuda.set_device(device)
        dp = DataParallel(device=device)
    rank, world = (dp.rank, dp.world) if dp else (0, 1)

    if args.synthetic:
        from event_flow_amd.dataloader.synthetic_loader import SyntheticLoader

        data = SyntheticLoader(config, config["model"]["num_bins"], config["model"].get("round_encoding", False), device=device,
                               seed_offset=1000 * rank)
    else:
        from event_flow_amd.dataloader.h5 import H5Loader

        data = H5Loader(config, config["model"]["num_bins"], config["model"].get("round_encoding", False), device=device,
                        rank=rank, world_size=world)

    loss_function = EventWarping(config, device)
    model = MODELS[config["model"]["name"]](config["model"].copy()).to(device)
    if args.prev:  # a state_dict file, a reference checkpoint (pickled model) or its MLflow run id under ./mlruns
        model = load_model(args.prev, model, device)
    model.train()

    clip = config["loss"].get("clip_grad", None)
    if args.fused_optimizer:
        optimizer = FlatAdam(model, lr=config["optimizer"]["lr"], clip=clip)
    else:
        optimizer = getattr(torch.optim, config["optimizer"]["name"])(model.parameters(), lr=config["optimizer"]["lr"])
    if dp:  # every rank starts from rank 0's parameters
        dp.broadcast(optimizer.flat_param)
        optimizer.step_invalidate()
    optimizer.zero_grad()

    n_epochs = args.epochs if args.epochs is not None else config["loader"]["n_epochs"]
    best_loss, history = 1.0e6, []
    t_start = time.time()
    data.shuffle()  # once, before the first epoch (train_flow.py:92)
    for epoch in range(n_epochs):
        train_loss, samples, steps = torch.zeros((), device=device), 0, 0
        batches = iter(data)
        while True:
            inputs = next(batches, None)
            new_seq, done = bool(inputs is not None and data.new_seq), inputs is None
            if dp:  # the ranks reset and end their epochs together
                new_seq, done = dp.any_flags([new_seq, done])
            if done:
                break
            if new_seq:  # any slot restarted: reset everything (train_flow.py:100-105)
                data.new_seq = False
                loss_function.reset()
                model.reset_states()
                optimizer.zero_grad()

            x = model(inputs["event_voxel"], inputs["event_cnt"])
            loss_function.event_flow_association(x["flow"], inputs["event_list"], inputs["event_list_pol_mask"],
                                                 inputs["event_mask"])

            # The optimizer step contains the step's collective, so with several ranks the decision to step is itself
            # made collectively: `num_events` is a rank-local count (it differs between ranks in the time / gtflow
            # modes), and a rank entering the all-reduce alone would pair it with another rank's flag exchange.
            step_now = loss_function.num_events >= config["data"]["window_loss"]
            if dp:
                step_now = dp.any_flags([step_now])[0]
            if step_now:
                if config["loss"]["overwrite_intermediate"]:
                    loss_function.overwrite_intermediate_flow(x["flow"])
                loss = loss_function()
                samples += config["loader"]["batch_size"] * world
                steps += 1
                loss.backward()
                if dp:  # gradient (+ loss) summed over the ranks: the step below is the global-batch step
                    loss = dp.all_reduce_grads(optimizer.comm, loss)[0]
                train_loss += loss.detach()  # no host sync inside the loop
                if args.fused_optimizer:
                    optimizer.step()  # clip + Adam in one kernel
                else:
                    if clip is not None:
                        torch.nn.utils.clip_grad.clip_grad_norm_(model.parameters(), clip)
                    optimizer.step()
                optimizer.zero_grad()
                model.detach_states()
                loss_function.reset()
        if hasattr(batches, "close"):
            batches.close()  # a rank that stops early (another rank ran out of files) ends its reader thread here
        epoch_loss = float(train_loss) / max(samples, 1)
        history.append(epoch_loss)
        if config["vis"].get("verbose", False) and rank == 0:
            print("Train Epoch: {:04d}  Loss: {:.6f}  ({} optimizer steps, {:.1f} s)".format(epoch, epoch_loss, steps,
                                                                                             time.time() - t_start))
        if epoch_loss < best_loss:
            best_loss = epoch_loss
            if args.out and rank == 0:
                torch.save(model.state_dict(), args.out)
    if dp:
        dp.barrier()
        dp.close()
    if rank == 0:
        print(json.dumps({"model": config["model"]["name"], "epochs": n_epochs, "loss_per_epoch": history, "best_loss": best_loss,
                          "ranks": world}))
    return history


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", default="configs/train_SNN.yml", help="training configuration")
    parser.add_argument("--synthetic", action="store_true", help="synthetic moving-dots windows instead of HDF5 files")
    parser.add_argument("--prev", default="", help="state_dict to resume from (reference: --prev_runid)")
    parser.add_argument("--out", default="", help="where to save the best state_dict")
    parser.add_argument("--epochs", type=int, default=None)
    parser.add_argument("--fused-optimizer", action="store_true")
    args = parser.parse_args()
    train(args, YAMLParser(args.config))
