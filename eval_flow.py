"""Evaluation driver -- counterpart of the reference's eval_flow.py:40-258 on the MI355X path: runs a model over
sequences, builds the per-polarity image of warped events (`compute_pol_iwe`) and accumulates FWL / RSAT / AEE
exactly as the reference loop does (association per input window, metric once `window_eval` events are collected).
No MLflow / visualiser; results are printed as JSON.

  python eval_flow.py --config configs/eval_flow.yml --train-config configs/train_SNN.yml --synthetic [--weights model.pth]
"""

import argparse
import json

import numpy as np
import torch

from event_flow_amd.configs.parser import YAMLParser
from event_flow_amd.loss import flow as metrics_mod
from event_flow_amd.models.model import MODELS
from event_flow_amd.utils.iwe import compute_pol_iwe
from event_flow_amd.utils.utils import load_model


def test(args, config_parser):
    # the reference overlays the eval YAML on the training run's stored parameters (merge_configs); here the
    # training YAML plays the role of the stored run
    train_cfg = YAMLParser(args.train_config).config
    config = config_parser.config
    merged = {k: (dict(v) if isinstance(v, dict) else v) for k, v in train_cfg.items()}
    for key, val in config.items():
        if isinstance(val, dict):
            merged.setdefault(key, {}).update(val)
        else:
            merged[key] = val
    config = config_parser.combine_entries(merged)
    config["data"].setdefault("window_eval", config["data"]["window"])
    config["loss"] = config.get("loss", {"overwrite_intermediate": False})
    device = config_parser.device
    if torch.device(device).type == "cuda":
        torch.cuda.set_device(device)  # the evf_* launches go to the current device's stream (loader.gpu may not be 0)

    # configuration checks of the reference driver (eval_flow.py:54-73)
    names_cfg = config.get("metrics", {}).get("name", [])
    if "AEE" in names_cfg and not args.synthetic:  # (the synthetic loader carries one ground-truth map per input window)
        assert config["data"]["mode"] in ("gtflow_dt1", "gtflow_dt4"), "AEE computation not possible without ground truth mode"
        assert config["data"]["window"] <= 1, "AEE computation not compatible with window > 1"
        assert np.isclose((1.0 / config["data"]["window"]) % 1.0, 0.0), \
            "AEE computation not compatible with windows whose inverse is not a round number"
    if config["data"]["mode"] == "frames":
        if config["data"]["window"] <= 1.0:
            assert np.isclose((1.0 / config["data"]["window"]) % 1.0, 0.0), \
                "Frames mode not compatible with < 1 windows whose inverse is not a round number"
        else:
            assert np.isclose(config["data"]["window"] % 1.0, 0.0), "Frames mode not compatible with > 1 fractional windows"

    model = MODELS[config["model"]["name"]](config["model"].copy()).to(device)
    if args.weights:  # a state_dict file, a reference checkpoint (pickled model) or its MLflow run id under ./mlruns
        model = load_model(args.weights, model, device)
    model.eval()

    names = config.get("metrics", {}).get("name", [])
    criteria = [getattr(metrics_mod, m)(config, device, flow_scaling=config["metrics"]["flow_scaling"]) for m in names]

    if args.synthetic:
        from event_flow_amd.dataloader.synthetic_loader import SyntheticLoader

        data = SyntheticLoader(config, config["model"]["num_bins"], device=device, num_sequences=args.sequences)
    else:
        from event_flow_amd.dataloader.h5 import H5Loader

        data = H5Loader(config, config["model"]["num_bins"], device=device)

    vis = None
    if args.store:  # stored PNGs in the reference's folder layout (utils/visualization.py:120-226); batch size 1
        from event_flow_amd.utils.visualization import Visualization

        vis = Visualization(config, eval_id=0, path_results=args.store.rstrip("/") + "/")
    results = {m: {"metric": 0.0, "it": 0, **({"percent": 0.0} if m == "AEE" else {})} for m in names}
    iwe_sharpness = []
    idx_AEE = 0  # sub-windows since the last AEE evaluation (eval_flow.py:117,171-176)
    aee_every = 1 if args.synthetic else int(np.round(1.0 / config["data"]["window"]))
    with torch.no_grad():
        for inputs in data:
            if data.new_seq:
                data.new_seq = False
                model.reset_states()
            if getattr(data, "pass_done", False):  # every sequence was visited (eval_flow.py:124-127)
                break
            x = model(inputs["event_voxel"], inputs["event_cnt"])
            iwe = compute_pol_iwe(x["flow"][-1], inputs["event_list"], config["loader"]["resolution"],
                                  inputs["event_list_pol_mask"][:, :, 0:1], inputs["event_list_pol_mask"][:, :, 1:2],
                                  flow_scaling=config["metrics"]["flow_scaling"], round_idx=True)
            iwe_sharpness.append(iwe.sum(1).var(dim=(1, 2)).mean())
            if vis is not None and iwe.shape[0] == 1:
                flow_vis = x["flow"][-1] * inputs["event_mask"] if config["model"].get("mask_output", True) else x["flow"][-1]
                vis.store(inputs, flow_vis, iwe, "seq%03d" % getattr(data, "seq_num", 0), ts=getattr(data, "last_proc_timestamp", None))
            for metric in criteria:
                metric.event_flow_association(x["flow"], inputs)
            for i, name in enumerate(names):
                if criteria[i].num_events >= config["data"]["window_eval"]:
                    if config["loss"].get("overwrite_intermediate", False):
                        criteria[i].overwrite_intermediate_flow(x["flow"])
                    if name == "AEE":
                        # no ground truth interval yet: skip WITHOUT resetting; with window < 1 the ground-truth map
                        # spans round(1/window) input windows, so the criterion keeps accumulating until the last
                        # of them and is evaluated (and reset) only then (eval_flow.py:169-176)
                        if float(torch.as_tensor(inputs["dt_gt"]).reshape(-1)[0]) <= 0.0:
                            continue
                        idx_AEE += 1
                        if idx_AEE != aee_every:
                            continue
                    val = criteria[i]()
                    if name == "AEE":
                        idx_AEE = 0
                    results[name]["it"] += 1
                    if name == "AEE":
                        results[name]["metric"] += float(val[0].mean())
                        results[name]["percent"] += float(val[1].mean())
                    else:
                        results[name]["metric"] += float(val.mean())
                    criteria[i].reset()
    out = {"model": config["model"]["name"], "iwe_variance": float(torch.stack(iwe_sharpness).mean())}
    for name, r in results.items():
        out[name] = r["metric"] / max(r["it"], 1)
        if name == "AEE":
            out["AEE_percent_outliers"] = r["percent"] / max(r["it"], 1)
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", default="configs/eval_flow.yml")
    parser.add_argument("--train-config", default="configs/train_SNN.yml", help="configuration the model was trained with")
    parser.add_argument("--weights", default="", help="state_dict (reference: the run id)")
    parser.add_argument("--synthetic", action="store_true")
    parser.add_argument("--sequences", type=int, default=4)
    parser.add_argument("--store", default="", help="directory that receives results/eval_0/<sequence>/{events,flow,iwe,...}/*.png")
    args = parser.parse_args()
    test(args, YAMLParser(args.config))
