"""Sequence loader end to end on the MI355X: batches of H5Loader (event windows read on the host, encodings binned
on the GPU by custom_collate) against the batches the reference's BaseDataLoader code makes from the same raw
sequences (fixture G13: augmentation + hot-pixel filter on), with and without the reader thread."""

import numpy as np
import pytest
import torch

from test_host_loader import _cfg, _timed_sequence, g13_loader

from event_flow_amd.dataloader.h5 import H5Loader

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("prefetch", [0, 2])
def test_batches_match_the_reference_collate(tmp_path, prefetch):
    g, ld, (H, W, win, nwin, nb) = g13_loader(tmp_path, prefetch=prefetch)
    seen = 0
    for w, batch in enumerate(ld):
        if w < nwin:
            assert not ld.new_seq and not ld.pass_done
            assert set(batch) == {"event_cnt", "event_voxel", "event_mask", "event_list", "event_list_pol_mask", "dt_gt", "dt_input"}
            for k, v in batch.items():
                ref = g[f"w{w}_{k}"]
                assert v.is_cuda and tuple(v.shape) == ref.shape and str(v.dtype).split(".")[-1] == str(ref.dtype), k
                if k == "event_voxel":  # sums of fp32 temporal weights: equal up to the order of the atomic adds
                    np.testing.assert_allclose(v.cpu().numpy(), ref, rtol=0, atol=2e-6, err_msg=f"{w} {k}")
                    assert np.array_equal(v.cpu().numpy() == 0, ref == 0)
                else:  # integer-valued images and fp32 copies: bit for bit
                    assert np.array_equal(v.cpu().numpy(), ref), (w, k)
        else:  # both slots ran out of events: first windows of the next files, fresh augmentation flags
            assert ld.new_seq and ld.pass_done and w == nwin
            assert tuple(batch["event_list"].shape) == (2, win, 4)
        seen += 1
    assert seen == nwin + 1 and ld.epoch == 1 and ld.seq_num == 0 and ld.samples == 2 * (nwin + 1) and not ld.new_seq


def test_ragged_time_windows_and_ground_truth(tmp_path):
    """mode time: windows of different event counts in one batch are padded with p = 0 rows that every encoding
    ignores; mode gtflow_dt1 carries the flow map and its time base."""
    _timed_sequence(tmp_path / "a.npz", seed=5)
    _timed_sequence(tmp_path / "b.npz", seed=6, n=3000)
    ld = H5Loader(_cfg(tmp_path, "time", 0.25, 2, (16, 20)), 2, prefetch=0)
    samples = [ld[0], ld[1]]
    n0, n1 = (s["event_list"].shape[1] for s in samples)
    assert n0 != n1
    batch = ld.custom_collate(samples)
    N = max(n0, n1)
    assert tuple(batch["event_list"].shape) == (2, N, 4)
    for b, n in enumerate((n0, n1)):
        ev = samples[b]["event_list"]
        cnt = np.zeros((2, 16, 20), np.float32)
        np.add.at(cnt[0], (ev[1].astype(int), ev[2].astype(int)), (ev[3] > 0).astype(np.float32))
        np.add.at(cnt[1], (ev[1].astype(int), ev[2].astype(int)), (ev[3] < 0).astype(np.float32))
        assert np.array_equal(batch["event_cnt"][b].cpu().numpy(), cnt)
        assert np.array_equal(batch["event_mask"][b, 0].cpu().numpy(), (cnt.sum(0) > 0).astype(np.float32))
        assert float(batch["event_list_pol_mask"][b, n:].abs().sum()) == 0.0
        assert float(batch["event_list_pol_mask"][b, :n].sum()) == n
    (tmp_path / "b.npz").unlink()
    _timed_sequence(tmp_path / "a.npz", seed=5, maps=1)
    ld = H5Loader(_cfg(tmp_path, "gtflow_dt1", 1, 1, (16, 20)), 2)
    batches = list(ld)
    assert len(batches) == 4 and ld.epoch == 1  # 3 intervals + the wrap-around batch
    b0 = batches[0]
    assert tuple(b0["gtflow"].shape) == (1, 2, 16, 20) and b0["gtflow"].is_cuda and b0["dt_gt"].dtype == torch.float64
    np.testing.assert_allclose(float(b0["dt_gt"][0]), 0.4, rtol=1e-12)


def test_reference_style_dataloader_wrapping(tmp_path):
    """The reference wraps the dataset in torch.utils.data.DataLoader(collate_fn=data.custom_collate), num_workers=0
    (train_flow.py:66-73): same batches, `new_seq` raised by `__getitem__` itself."""
    g, ld, (H, W, win, nwin, nb) = g13_loader(tmp_path)
    dl = torch.utils.data.DataLoader(ld, drop_last=True, batch_size=2, collate_fn=ld.custom_collate)
    for w, batch in enumerate(dl):
        if w == nwin:
            assert ld.new_seq and ld.seq_num >= len(ld.files)
            break
        assert not ld.new_seq
        for k in ("event_cnt", "event_mask", "event_list", "event_list_pol_mask", "dt_input"):
            assert np.array_equal(batch[k].cpu().numpy(), g[f"w{w}_{k}"]), (w, k)


def test_empty_windows_flow_through_model_and_loss(tmp_path):
    """A window with <= 10 events is handed on as an EMPTY one (dataloader/h5.py:240-245): the batch then carries
    zero-length event lists and all-zero encodings; model, loss association and the loss itself must take them."""
    from event_flow_amd.loss.flow import EventWarping
    from event_flow_amd.models.model import LIFFireNet

    rng = np.random.Generator(np.random.PCG64(3))
    n = 2000
    ts = np.concatenate([np.sort(rng.random(n - 5)) * 0.5, 0.5 + 0.3 + 0.01 * np.arange(5)]) + 10.0  # 5 events in [0.75, 1.0)
    from event_flow_amd.dataloader.h5 import write_npz_sequence

    write_npz_sequence(str(tmp_path / "a.npz"), rng.integers(0, 32, n), rng.integers(0, 32, n), ts, rng.integers(0, 2, n))
    ld = H5Loader(_cfg(tmp_path, "time", 0.25, 1, (32, 32)), 2, prefetch=0)
    samples = [ld[i] for i in range(3)]
    assert [s["event_list"].shape[1] > 0 for s in samples] == [True, True, False]
    empty = ld.custom_collate([samples[2]])
    assert tuple(empty["event_list"].shape) == (1, 0, 4) and tuple(empty["event_list_pol_mask"].shape) == (1, 0, 2)
    assert float(empty["event_cnt"].abs().sum()) == 0 and float(empty["event_mask"].sum()) == 0 and float(empty["dt_input"]) == 0
    cfg = {"num_bins": 2, "base_num_channels": 32, "kernel_size": 3, "encoding": "cnt", "norm_input": False, "mask_output": True,
           "activations": ["arctanspike", "arctanspike"],
           "spiking_neuron": {"leak": [-4.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True, "learn_thresh": True, "hard_reset": True}}
    model = LIFFireNet(cfg).to("cuda:0")
    lossf = EventWarping({"loader": {"resolution": [32, 32]}, "loss": {"flow_regul_weight": 0.001, "overwrite_intermediate": False},
                          "model": {"mask_output": True}}, "cuda:0")
    full = ld.custom_collate([samples[0]])
    for batch in (full, empty):
        out = model(batch["event_voxel"], batch["event_cnt"])
        lossf.event_flow_association(out["flow"], batch["event_list"], batch["event_list_pol_mask"], batch["event_mask"])
    assert lossf.num_events == samples[0]["event_list"].shape[1]
    loss = lossf()
    loss.backward()
    assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


def test_hdf5_sequence_batches_equal_npz_batches(tmp_path):
    """A real HDF5 sequence file (tests/golden/seq_fixture.h5, written by h5py in the reference's layout) through the loader
    + on-GPU binning gives bit for bit the batches of its `.npz` twin, with the reader thread on (dataloader/h5.py:45-94,249)."""
    import os
    import shutil

    from test_host_loader import GOLDEN, _h5_reader_available

    from event_flow_amd.dataloader.h5 import write_npz_sequence

    if not _h5_reader_available():
        pytest.skip("neither h5py nor an HDF5 C library on this box")
    tw = np.load(os.path.join(GOLDEN, "seq_fixture_twin.npz"))
    for d in ("h5", "npz"):
        (tmp_path / d).mkdir()
    shutil.copy(os.path.join(GOLDEN, "seq_fixture.h5"), tmp_path / "h5" / "a.h5")
    groups = {g: [(k[len(g) + 1:], float(tw[f"{g}_ts/{k[len(g) + 1:]}"]), tw[k]) for k in sorted(tw.files) if k.startswith(g + "/")]
              for g in ("flow_dt1",)}
    write_npz_sequence(str(tmp_path / "npz" / "a.npz"), tw["events/xs"], tw["events/ys"], tw["events/ts"], tw["events/ps"], **groups)
    for mode, window in (("events", 500), ("gtflow_dt1", 1)):
        la = H5Loader(_cfg(tmp_path / "h5", mode, window, 1, (16, 20)), 5, prefetch=2)
        lb = H5Loader(_cfg(tmp_path / "npz", mode, window, 1, (16, 20)), 5, prefetch=0)
        n = 0
        for ba, bb in zip(la, lb):
            assert ba.keys() == bb.keys()
            for k in ba:
                if k == "event_voxel":  # sums of fp32 temporal weights: equal up to the order of the atomic adds
                    assert torch.allclose(ba[k], bb[k], rtol=0, atol=2e-6), (mode, n, k)
                else:
                    assert ba[k].is_cuda and torch.equal(ba[k], bb[k]), (mode, n, k)
            n += 1
            if n == 3:
                break
        assert n == 3
