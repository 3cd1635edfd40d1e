"""YAML configuration loader -- counterpart of reference configs/parser.py:6-127 (same class name, properties and
defaults, so the reference's YAML files load unchanged).  MLflow merging (`merge_configs`) keeps its semantics for
runs exported as plain dicts; the MLflow client itself is not a dependency."""

import ast
import copy

import numpy as np
import torch
import yaml

# reference defaults, parser.py:31-63
_DEFAULTS = {
    "experiment": "Default",
    "data": {"mode": "events", "window": 5000},
    "loader": {"resolution": [180, 240], "batch_size": 1, "augment": [], "gpu": 0, "seed": 0},
    "hot_filter": {"enabled": True, "max_px": 100, "min_obvs": 5, "max_rate": 0.8},
    "model": {},
    "spiking_neuron": {},
    "vis": {"bars": False},
}


def _merge(dst, src):
    """Recursive update: nested dicts are merged key by key, leaves overwritten (parser.py:69-78)."""
    for key, val in src.items():
        if isinstance(val, dict):
            _merge(dst.setdefault(key, {}), val)
        else:
            dst[key] = val
    return dst


class YAMLParser:
    """YAML parser for optical-flow config files."""

    def __init__(self, config):
        self.update(config)
        self.get_device()
        self.init_seeds()

    config = property(lambda self: self._config)
    device = property(lambda self: self._device)
    loader_kwargs = property(lambda self: self._loader_kwargs)

    def reset_config(self):
        self._config = copy.deepcopy(_DEFAULTS)

    def parse_config(self, file):
        with open(file) as fid:
            self.parse_dict(yaml.load(fid, Loader=yaml.FullLoader))

    def parse_dict(self, input_dict, parent=None):
        _merge(self._config if parent is None else parent, input_dict)

    def update(self, config):
        self.reset_config()
        self.parse_config(config)

    def get_device(self):
        cuda = torch.cuda.is_available()
        self._device = torch.device(f"cuda:{self._config['loader']['gpu']}" if cuda else "cpu")
        self._loader_kwargs = {"num_workers": 0, "pin_memory": True} if cuda else {}

    @staticmethod
    def worker_init_fn(worker_id):
        np.random.seed(np.random.get_state()[1][0] + worker_id)

    def init_seeds(self):
        seed = self._config["loader"]["seed"]
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed)

    def merge_configs(self, run):
        """Overlay this config on the (stringified) parameters of a stored run (parser.py:99-115)."""
        config = {}
        for key, val in run.items():
            config[key] = ast.literal_eval(val) if len(val) > 0 and val[0] == "{" else val
        self.parse_dict(self._config, config)
        return self.combine_entries(config)

    @staticmethod
    def combine_entries(config):
        """`spiking_neuron` travels as its own top-level block (MLflow's length limit, parser.py:117-127) and
        belongs inside `model`."""
        if "spiking_neuron" in config:
            config["model"]["spiking_neuron"] = config.pop("spiking_neuron")
        return config
