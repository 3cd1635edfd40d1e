"""Checkpoint helpers -- counterpart of reference utils/utils.py:8-37 without MLflow.

The reference stores WHOLE pickled model objects (`mlflow.pytorch.log_model`, `<run>/artifacts/model/data/model.pth`,
utils/utils.py:36-37) whose classes live in its top-level `models.*` modules, and restores them with
`model.load_state_dict(torch.load(path).state_dict())` (:18-19).  `load_model` reads such a file on a machine that has
only this package: while unpickling, the reference's module names resolve to the mirror modules of
`event_flow_amd.models` (same class names, same parameter names), and only the state_dict is taken over."""

import contextlib
import importlib
import os
import sys

import torch

_MIRRORS = ("base", "model", "model_util", "spiking_submodules", "spiking_util", "submodules", "unet")


@contextlib.contextmanager
def reference_module_names():
    """Temporarily expose event_flow_amd.models.* under the reference's names (`models`, `models.model`, ...)."""
    pkg = importlib.import_module("event_flow_amd.models")
    added = {}
    try:
        for name, mod in [("models", pkg)] + [(f"models.{m}", importlib.import_module(f"event_flow_amd.models.{m}")) for m in _MIRRORS]:
            if name not in sys.modules:
                sys.modules[name] = mod
                added[name] = mod
        yield
    finally:
        for name in added:
            sys.modules.pop(name, None)


def checkpoint_path(prev_runid, root="mlruns"):
    """A file path as is; otherwise the MLflow run id's `artifacts/model/data/model.pth` under `root`/<experiment>/."""
    if os.path.isfile(prev_runid):
        return prev_runid
    if os.path.isdir(root):
        for exp in sorted(os.listdir(root)):
            cand = os.path.join(root, exp, prev_runid, "artifacts", "model", "data", "model.pth")
            if os.path.isfile(cand):
                return cand
    return None


def load_model(prev_runid, model, device, root="mlruns"):
    """Restore `model` from a reference checkpoint (pickled model object) or a plain state_dict file.
    Like the reference (:8-25): an unknown run leaves the model untouched."""
    path = checkpoint_path(prev_runid, root) if prev_runid else None
    if path is None:
        print("No model found at" + str(prev_runid) + "\n")
        return model
    with reference_module_names():
        loaded = torch.load(path, map_location=device, weights_only=False)
    state = loaded.state_dict() if isinstance(loaded, torch.nn.Module) else loaded
    model.load_state_dict(state)
    print("Model restored from " + str(prev_runid) + "\n")
    return model


def save_model(model, path):
    """Plain state_dict (what `load_model` and the reference's `load_state_dict` both take)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(model.state_dict(), path)


def create_model_dir(path_results, runid):
    """Reference :28-33."""
    path_results += runid + "/"
    if not os.path.exists(path_results):
        os.makedirs(path_results)
    print("Results stored at " + path_results + "\n")
    return path_results
