"""CPU-side checks of the host mirror of the reference model API (no compute
calls: there is no CPU path).  The reference's state_dict keys and shapes, as
captured in the golden fixtures, must load unchanged; the state API keeps the
reference's semantics; calling a model without the MI355X fails loudly."""

import numpy as np
import pytest
import torch

from conftest import load_golden

from event_flow_amd import _lib
from event_flow_amd.models import model as M
from event_flow_amd.models import spiking_submodules as cells
from event_flow_amd.models.model_util import CropParameters, copy_states, skip_concat

NEURON = {"leak": [-4.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True, "learn_thresh": True, "hard_reset": True}


def cfg(C=32, neuron=NEURON, acts=("arctanspike", "arctanspike"), encoding="cnt"):
    return {"name": "x", "num_bins": 2, "base_num_channels": C, "kernel_size": 3, "encoding": encoding, "round_encoding": False,
            "norm_input": False, "mask_output": True, "activations": list(acts),
            "spiking_neuron": dict(neuron) if neuron else None}


def _load(model, g, prefix):
    sd = {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}
    missing, unexpected = model.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return sd


def test_reference_state_dicts_load_unchanged():
    _load(M.LIFFireNet(cfg()), load_golden("g7_liffirenet_train"), "param0_")
    plif = {"leak_v": [-4.0, 0.1], "leak_pt": [-4.0, 0.1], "add_pt": [-2.0, 0.1], "thresh": [0.8, 0.1]}
    _load(M.PLIFFireNet(cfg(neuron=plif)), load_golden("g7_pliffirenet_train"), "param0_")
    _load(M.FireNet(cfg(neuron=None, acts=("relu", None), encoding="voxel")), load_golden("g8_firenet_ann"), "param_")
    g10 = load_golden("g10_ann_firenets")
    leaky = {"leak": [-1.0, 0.5], "learn_leak": True}
    _load(M.FireFlowNet(cfg(C=8, neuron=None, acts=("relu", "relu"))), g10, "FireFlowNet.param_")
    _load(M.RNNFireNet(cfg(C=8, neuron=None, acts=("relu", None))), g10, "RNNFireNet.param_")
    _load(M.LeakyFireNet(cfg(C=8, neuron=leaky, acts=("relu", None))), g10, "LeakyFireNet.param_")
    _load(M.LeakyFireFlowNet(cfg(C=8, neuron=leaky, acts=("relu", "tanh"))), g10, "LeakyFireFlowNet.param_")
    _load(M.E2VID(cfg(C=4, neuron=None, acts=("relu", None))), load_golden("g12_e2vid"), "param_")
    g11 = load_golden("g11_ann_unets")
    for name in ("EVFlowNet", "RecEVFlowNet", "RNNRecEVFlowNet"):
        _load(getattr(M, name)(cfg(C=4, neuron=None, acts=("relu", None))), g11, name + ".param_")
    _load(M.LeakyRecEVFlowNet(cfg(C=4, neuron=leaky, acts=("relu", None))), g11, "LeakyRecEVFlowNet.param_")
    sd = _load(M.SpikingRecEVFlowNet(cfg(C=4)), load_golden("g9_spiking_unet"), "param_")
    assert sd["multires_unetrec.decoders.1.conv2d.ff.weight"].shape == (16, 66, 3, 3)  # cat(pred, x, skip)


def test_parameter_counts_match_the_reference():
    # 74 818 (LIF-FireNet), 148 450 (FireNet ANN), 20 400 840 (LIF-EV-FlowNet) parameters
    n = lambda m: sum(p.numel() for p in m.parameters())  # noqa: E731
    assert n(M.LIFFireNet(cfg())) == 74818
    assert n(M.FireNet(cfg(neuron=None, acts=("relu", None)))) == 148450
    assert n(M.SpikingRecEVFlowNet(cfg())) == 20400840


def test_model_zoo_names_and_ctor_does_not_mutate_config():
    for name in ("FireNet", "FireFlowNet", "RNNFireNet", "LeakyFireNet", "LeakyFireFlowNet", "E2VID", "EVFlowNet", "RecEVFlowNet",
                 "RNNRecEVFlowNet", "LeakyRecEVFlowNet", "LIFFireNet", "PLIFFireNet", "ALIFFireNet", "XLIFFireNet", "LIFFireFlowNet", "SpikingRecEVFlowNet",
                 "PLIFRecEVFlowNet", "ALIFRecEVFlowNet", "XLIFRecEVFlowNet"):
        assert name in M.MODELS and getattr(M, name) is M.MODELS[name]
    c = cfg(C=4)
    before = {k: (list(v) if isinstance(v, list) else v) for k, v in c.items()}
    M.SpikingRecEVFlowNet(c)
    assert {k: (list(v) if isinstance(v, list) else v) for k, v in c.items()} == before  # reference quirk q3 not inherited
    a, b = M.LIFFireNet(cfg()), M.PLIFFireNet(cfg(neuron={"leak_v": [-4.0, 0.1]}))
    assert hasattr(a.head, "leak") and hasattr(b.head, "leak_v")  # no shared class-level kwargs (quirk q2)


def test_compute_path_of_the_adaptive_threshold_firenets(monkeypatch):
    """Which HIP path serves a network is host logic: XLIF / ALIF FireNets go to the recorded window kernels with the reference's
    TRAINING neuron (configs/train_SNN.yml: hard reset, arctan) and stay on the general path with the cells' own default (soft
    reset), another surrogate, or EVF_XLIF_FUSED=0; LIF / PLIF FireNets are fused either way."""
    monkeypatch.setenv("EVF_PATH_NOTICE", "0")
    for name, leak in (("XLIFFireNet", "leak_pt"), ("ALIFFireNet", "leak_t")):
        hard = {"leak_v": [-4.0, 0.1], leak: [-4.0, 0.1], "t0": [0.3, 0.0], "t1": [0.5, 0.0], "hard_reset": True}
        assert M.MODELS[name](cfg(neuron=hard)).compute_path == ("fused", "")
        path, why = M.MODELS[name](cfg(neuron={k: v for k, v in hard.items() if k != "hard_reset"})).compute_path
        assert path == "general" and "soft reset" in why
        path, why = M.MODELS[name](cfg(neuron=hard, acts=("superspike", "superspike"))).compute_path
        assert path == "general"
        monkeypatch.setenv("EVF_XLIF_FUSED", "0")
        assert M.MODELS[name](cfg(neuron=hard)).compute_path == ("general", "EVF_XLIF_FUSED=0")
        monkeypatch.delenv("EVF_XLIF_FUSED")
    assert M.LIFFireNet(cfg()).compute_path[0] == "fused"


def test_cell_constructors_mirror_reference_defaults():
    c = cells.ConvALIF(4, 8, 3)
    assert not c.hard_reset and c.kind == "alif" and "t0" in dict(c.named_buffers())  # learn_thresh=False -> buffers
    c = cells.ConvLIFRecurrent(8, 8, 3)
    assert c.hard_reset and c.recurrent and set(dict(c.named_parameters())) == {"leak", "thresh", "ff.weight", "rec.weight"}
    with pytest.raises(AssertionError):
        cells.ConvLIF(4, 8, 3, activation=None)  # spiking_submodules.py:78-81
    with pytest.raises(AttributeError):
        cells.ConvLIF(4, 8, 3, activation="nospike")
    blk = cells.SpikingRecurrentConvLayer(2, 8, 3, stride=2, recurrent_block_type="plif")
    assert type(blk.conv).__name__ == "ConvPLIF" and type(blk.recurrent_block).__name__ == "ConvPLIFRecurrent"


def test_state_api_without_a_forward():
    m = M.LIFFireNet(cfg())
    assert m.states == [None] * 7
    m.reset_states()
    m.detach_states()
    u = M.SpikingRecEVFlowNet(cfg(C=4))
    assert u.states == [None] * 10 and u.multires_unetrec.num_states == 10
    u.detach_states()
    u.reset_states()
    st = [torch.zeros(2, 1, 4, 3, 3), None]
    assert copy_states([None, 1]) == [None, 1]
    cp = copy_states([st[0], st[0]])
    assert cp[0] is not st[0] and torch.equal(cp[0], st[0])


def test_no_cpu_execution_of_models_and_cells():
    x = torch.zeros(1, 2, 16, 16)
    for m in (M.LIFFireNet(cfg()), M.FireNet(cfg(neuron=None, acts=("relu", None))), M.SpikingRecEVFlowNet(cfg(C=4)),
              M.ALIFFireNet(cfg(neuron={"leak_v": [-4.0, 0.1]}))):
        with pytest.raises(_lib.EvflowError):
            m(x, x)
    with pytest.raises(_lib.EvflowError):
        cells.ConvLIF(2, 8, 3)(x, None)
    with pytest.raises(AttributeError):
        bad = cfg()
        bad["encoding"] = "nope"
        M.SpikingRecEVFlowNet(bad)(x, x)


def test_crop_parameters_and_skip_concat_shapes():
    cp = CropParameters(346, 260, 4)  # MVSEC resolution: width, height
    assert (cp.width_crop_size, cp.height_crop_size) == (352, 272)
    padded = cp.pad(torch.zeros(1, 2, 260, 346))
    assert tuple(padded.shape) == (1, 2, 272, 352)
    assert tuple(padded[:, :, cp.iy0 : cp.iy1, cp.ix0 : cp.ix1].shape) == (1, 2, 260, 346)
    a, b = torch.zeros(1, 3, 7, 9), torch.zeros(1, 5, 8, 10)
    assert tuple(skip_concat(a, b).shape) == (1, 8, 8, 10)
    assert np.isclose(float(skip_concat(torch.ones(1, 1, 2, 2), torch.zeros(1, 1, 4, 4)).sum()), 4.0)


def test_yaml_parser_reads_reference_style_configs(tmp_path):
    """configs/parser.py counterpart: defaults, nested merge, spiking_neuron folded into model (parser.py:117-127)."""
    from event_flow_amd.configs.parser import YAMLParser

    f = tmp_path / "c.yml"
    f.write_text(
        "data:\n    window: 1000\n    window_loss: 10000\nmodel:\n    name: LIFFireNet\n    num_bins: 2\n"
        "spiking_neuron:\n    leak: [-4.0, 0.1]\n    hard_reset: True\nloader:\n    batch_size: 8\n    resolution: [128, 128]\n"
    )
    p = YAMLParser(str(f))
    c = p.config
    assert c["experiment"] == "Default" and c["data"]["mode"] == "events" and c["data"]["window"] == 1000
    assert c["loader"]["gpu"] == 0 and c["loader"]["seed"] == 0 and c["loader"]["batch_size"] == 8
    assert c["hot_filter"] == {"enabled": True, "max_px": 100, "min_obvs": 5, "max_rate": 0.8}
    c = p.combine_entries(c)
    assert "spiking_neuron" not in c and c["model"]["spiking_neuron"]["leak"] == [-4.0, 0.1]
    merged = YAMLParser(str(f)).merge_configs({"model": "{'name': 'PLIFFireNet'}", "experiment": "x"})
    assert merged["model"]["name"] == "LIFFireNet" and merged["experiment"] == "Default"  # the config wins (parser.py:112)
    # the repo's own driver configs parse and name an accelerated model
    own = YAMLParser(str(__import__("pathlib").Path(__file__).resolve().parents[1] / "configs" / "train_SNN.yml")).config
    assert own["model"]["name"] in M.MODELS and own["data"]["window_loss"] % own["data"]["window"] == 0


def test_h5_files_without_any_hdf5_reader_fail_loudly(tmp_path, monkeypatch):
    """`.h5` sequences go through h5py or, without it, through the HDF5 C library (dataloader/hdf5_ctypes.py); a file that is
    not HDF5 -- or a box with neither -- fails loudly instead of yielding empty batches."""
    from event_flow_amd.dataloader import hdf5_ctypes
    from event_flow_amd.dataloader.h5 import H5Loader

    (tmp_path / "a.h5").write_bytes(b"not an hdf5 file")
    cfg = {"data": {"path": str(tmp_path), "mode": "events", "window": 10},
           "loader": {"batch_size": 1, "resolution": [8, 8], "augment": []}, "hot_filter": {"enabled": False}}
    with pytest.raises((OSError, ImportError)):
        H5Loader(cfg, 2)
    try:
        import h5py  # noqa: F401
    except ImportError:
        monkeypatch.setattr(hdf5_ctypes, "_lib", None)
        monkeypatch.setattr(hdf5_ctypes, "_CANDIDATES", ("/nonexistent/libhdf5.so",))
        with pytest.raises(hdf5_ctypes.Hdf5Error):
            H5Loader(cfg, 2)



def test_reference_pickled_checkpoint_restores(tmp_path):
    """The reference stores whole pickled model objects (utils/utils.py:36-37) and restores through
    `load_state_dict(torch.load(path).state_dict())` (:18-19): such a file, made by the reference itself, loads here
    by run id or by path although its classes live in the reference's `models.*` modules."""
    import os
    import sys

    from event_flow_amd.utils.utils import load_model, save_model

    g = load_golden("g14_checkpoint")
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mlruns")
    runid = "0123456789abcdef0123456789abcdef"
    for ref in (runid, os.path.join(root, "0", runid, "artifacts", "model", "data", "model.pth")):
        model = M.LIFFireNet(cfg(C=8))
        before = {k: v.clone() for k, v in model.state_dict().items()}
        load_model(ref, model, "cpu", root=root)
        for k, v in model.state_dict().items():
            assert np.array_equal(v.numpy(), g["param_" + k]), k
        assert any(not torch.equal(before[k], v) for k, v in model.state_dict().items())
    assert "models" not in sys.modules and "models.model" not in sys.modules  # the aliases are gone again
    untouched = M.LIFFireNet(cfg(C=8))
    sd = {k: v.clone() for k, v in untouched.state_dict().items()}
    load_model("no-such-run", untouched, "cpu", root=root)  # unknown run: model unchanged (reference :9-12)
    assert all(torch.equal(sd[k], v) for k, v in untouched.state_dict().items())
    save_model(model, str(tmp_path / "x" / "m.pth"))
    again = load_model(str(tmp_path / "x" / "m.pth"), M.LIFFireNet(cfg(C=8)), "cpu")
    assert all(np.array_equal(v.numpy(), g["param_" + k]) for k, v in again.state_dict().items())


def test_recorded_forward_index_plans(monkeypatch):
    """engine._fwd_slots (host logic of the recorded forward, EVF_FWD_LM): diagonals, every hidden layer by layer, or only the
    feed-forward layers above the last recurrent one as chains -- a chain's index lies behind every index of the layer it reads,
    a recurrent layer gets an index per pass, everything fits the recorder's 96 indices, the choice follows the shape."""
    from event_flow_amd.models import engine as heng

    eng = M.LIFFireNet(cfg())._eng()  # head, G1 (rec), R1a, R1b, G2 (rec), R2a, R2b
    assert [c.recurrent for c in eng.cells] == [False, True, False, False, True, False, False]
    P = heng.FWD_LM_PASSES
    monkeypatch.setattr(heng, "FWD_LAYER_MAJOR", "0")
    assert eng._fwd_mode(8, 128, 128) == "0" and eng._fwd_slots(8, 128, 128) is None
    monkeypatch.setattr(heng, "FWD_LAYER_MAJOR", "top")
    assert eng._fwd_slots(8, 128, 128) == [None, (0, 1), (1, 1), (2, 1), (3, 1), (P + 3, 0), (P + 4, 0)]
    monkeypatch.setattr(heng, "FWD_LAYER_MAJOR", "1")
    assert eng._fwd_slots(8, 128, 128) == [None, (0, 1), (P, 0), (P + 1, 0), (P + 2, 1), (2 * P + 2, 0), (2 * P + 3, 0)]
    for mode in ("top", "1"):
        monkeypatch.setattr(heng, "FWD_LAYER_MAJOR", mode)
        plan = eng._fwd_slots(2, 32, 32)
        idx = lambda i, t: plan[i][0] + plan[i][1] * t  # noqa: E731
        for t in range(P):
            for i in range(1, 7):
                assert 0 <= idx(i, t) < 96
                if i > 1:  # the layer below at the same pass comes first
                    assert idx(i - 1, t) < idx(i, t)
                if t > 0 and plan[i][1]:  # the cell's own previous pass comes first (a chain holds both under one index)
                    assert idx(i, t - 1) < idx(i, t)
    monkeypatch.setattr(heng, "FWD_LAYER_MAJOR", "auto")
    assert eng._fwd_mode(8, 128, 128) == "top" and eng._fwd_mode(4, 260, 346) == "1" and eng._fwd_mode(1, 32, 32) == "top"
    # a network without recurrent layers: every hidden layer is a chain in both forms
    ff = M.LIFFireFlowNet(cfg())._eng()
    assert not any(c.recurrent for c in ff.cells)
    for mode in ("top", "1"):
        monkeypatch.setattr(heng, "FWD_LAYER_MAJOR", mode)
        plan = ff._fwd_slots(2, 32, 32)
        assert all(p[1] == 0 for p in plan[1:]) and [p[0] for p in plan[1:]] == sorted({p[0] for p in plan[1:]})
