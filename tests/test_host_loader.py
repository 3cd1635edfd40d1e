"""Host side of the sequence loader (event_flow_amd/dataloader/{base,h5}.py) without a GPU: windowing in every input
mode, sequence changes, augmentation, hot-pixel masks -- against the reference-made fixture G13 where the reference's
code could produce one (its BaseDataLoader; the H5Loader itself needs h5py) and against brute-force expectations for
the index arithmetic of dataloader/h5.py:136-229."""

import numpy as np
import pytest
import torch

from conftest import load_golden

from event_flow_amd import _lib
from event_flow_amd.dataloader.h5 import H5Loader, write_npz_sequence


def _cfg(path, mode, window, B, res, augment=(), hot=False):
    return {"data": {"path": str(path), "mode": mode, "window": window},
            "loader": {"batch_size": B, "resolution": list(res), "augment": list(augment), "augment_prob": [0.5] * len(augment)},
            "hot_filter": {"enabled": hot, "max_px": 100, "min_obvs": 2, "max_rate": 0.8}, "vis": {"bars": False}}


def _g13_files(tmp_path):
    g = load_golden("g13_loader")
    H, W, win, nwin, nb = (int(v) for v in g["meta_HW_win_nwin_nb"])
    for b in range(2):
        write_npz_sequence(str(tmp_path / f"seq{b}.npz"), g[f"seq{b}_xs"], g[f"seq{b}_ys"], g[f"seq{b}_ts"], g[f"seq{b}_ps"])
    return g, (H, W, win, nwin, nb)


def g13_loader(tmp_path, **kw):
    g, (H, W, win, nwin, nb) = _g13_files(tmp_path)
    ld = H5Loader(_cfg(tmp_path, "events", win, 2, (H, W), ("Horizontal", "Vertical", "Polarity"), hot=True), nb, **kw)
    ld.batch_augmentation = {k: [bool(v) for v in g["aug_" + k]] for k in ("Horizontal", "Vertical", "Polarity")}
    return g, ld, (H, W, win, nwin, nb)


def test_count_windows_match_the_reference_samples(tmp_path):
    g, ld, (H, W, win, nwin, nb) = g13_loader(tmp_path)
    assert ld.get_iters(0) == (win * nwin + 37) // win
    for w in range(nwin):
        for b in range(2):
            s = ld[2 * w + b]
            assert s["event_list"].shape == (4, win) and s["event_list"].dtype == np.float32
            assert np.array_equal(s["event_list"].T, g[f"w{w}_event_list"][b])  # rows (ts, y, x, p), bit for bit
            assert float(s["dt_input"]) == float(g[f"w{w}_dt_input"][b]) and float(s["dt_gt"]) == 0.0
            hm = s["hot_mask"].numpy()
            # the reference-made masked encodings tell where its mask was zero: pixels with events but event_mask == 0
            ev = s["event_list"]
            occupied = np.zeros((H, W), bool)
            occupied[ev[1].astype(int), ev[2].astype(int)] = True
            assert np.array_equal(hm == 0, occupied & (g[f"w{w}_event_mask"][b, 0] == 0))
            assert ((hm == 0).sum() >= 3) if w >= 2 else ((hm == 0).sum() == 0)  # planted pixels (+ chance), once min_obvs passed
    assert not ld.new_seq and ld.seq_num == 0
    # the 37-event tail is shorter than a window: both slots move on to the next file (h5.py:232-275)
    s = ld[2 * nwin]
    assert ld.new_seq and ld.seq_num == 1 and ld.batch_idx == [2, 1] and ld.batch_row[0] == win
    assert s["event_list"].shape == (4, win) and ld.hot_idx[0] == 1
    ld[2 * nwin + 1]
    assert ld.seq_num == 2 and ld.batch_idx == [2, 3]


def _timed_sequence(path, n=4000, H=16, W=20, seed=5, maps=0, frames=0):
    rng = np.random.Generator(np.random.PCG64(seed))
    ts = np.sort(rng.random(n)) * 2.0 + 100.0
    xs, ys, ps = rng.integers(0, W, n), rng.integers(0, H, n), rng.integers(0, 2, n)
    groups = {}
    stamps = 100.0 + 0.4 * np.arange(1, 5) + 0.01
    if maps:
        for g_ in ("flow_dt1", "flow_dt4"):
            groups[g_] = [(f"{k:09d}", stamps[k], rng.standard_normal((2, H, W)).astype(np.float32)) for k in range(4)]
    if frames:
        groups["images"] = [(f"{k:09d}", stamps[k], rng.integers(0, 255, (H, W)).astype(np.uint8)) for k in range(4)]
    write_npz_sequence(str(path), xs, ys, ts, ps, **groups)
    return xs, ys, ts, ps, stamps, groups


def test_time_windows(tmp_path):
    xs, ys, ts, ps, _, _ = _timed_sequence(tmp_path / "a.npz")
    ld = H5Loader(_cfg(tmp_path, "time", 0.25, 1, (16, 20)), 2)
    t0 = ts[0]
    for w in range(3):
        s = ld[w]
        lo, hi = np.searchsorted(ts, t0 + 0.25 * w), np.searchsorted(ts, t0 + 0.25 * (w + 1))
        assert s["event_list"].shape[1] == hi - lo
        assert np.array_equal(s["event_list"][2], xs[lo:hi].astype(np.float32))
        assert np.array_equal(s["event_list"][3], ps[lo:hi].astype(np.float32) * 2 - 1)
        rel = (ts[lo:hi] - t0).astype(np.float32)  # fp32 BEFORE the normalisation, like base.py:80-85
        np.testing.assert_array_equal(s["event_list"][0], (rel - rel[0]) / (rel[-1] - rel[0]))
        assert float(s["dt_input"]) == (ts[hi - 1] - t0) - (ts[lo] - t0)
    assert ld.get_iters(0) == (ts[-1] - ts[0]) // 0.25


@pytest.mark.parametrize("mode,window", [("gtflow_dt1", 1), ("gtflow_dt4", 0.25)])
def test_ground_truth_windows(tmp_path, mode, window):
    xs, ys, ts, ps, stamps, groups = _timed_sequence(tmp_path / "a.npz", maps=1)
    ld = H5Loader(_cfg(tmp_path, mode, window, 1, (16, 20), ("Horizontal", "Vertical")), 2)
    ld.batch_augmentation = {"Horizontal": [True], "Vertical": [False]}
    maps = groups["flow_dt1" if mode == "gtflow_dt1" else "flow_dt4"]
    n_ok = 3 if window == 1 else 12  # rows until ceil(row + window) reaches the last stamp (h5.py:196-203)
    for i in range(n_ok):
        row = i * window
        s = ld[i]
        k0 = int(np.floor(row))
        k1 = int(np.ceil(row + window))
        if window < 1 and k1 - k0 > 1:
            k0 += k1 - k0 - 1
        lo, hi = np.searchsorted(ts, stamps[k0]), np.searchsorted(ts, stamps[k1])
        if window < 1:
            d = hi - lo
            lo, hi = int(lo + (row - k0) * d), int(lo + (row + window - k0) * d)
        assert s["event_list"].shape[1] == hi - lo, (i, lo, hi)
        assert np.array_equal(s["event_list"][2], 20 - 1 - xs[lo:hi].astype(np.float32))  # horizontal flip
        assert np.array_equal(s["event_list"][1], ys[lo:hi].astype(np.float32))
        ref = np.flip(maps[k1][2], 2).copy()
        ref[0] *= -1
        assert np.array_equal(s["gtflow"], ref)
        assert float(s["dt_gt"]) == stamps[k1] - stamps[k1 - 1]
    assert ld.seq_num == 0
    ld[n_ok]  # runs past the last map: same file again (only one), sequence counter up
    assert ld.seq_num == 1 and ld.new_seq and ld.batch_row[0] == window


def test_frame_windows_and_errors(tmp_path):
    xs, ys, ts, ps, stamps, groups = _timed_sequence(tmp_path / "a.npz", frames=1)
    ld = H5Loader(_cfg(tmp_path, "frames", 1, 1, (16, 20), ("Vertical",)), 2)
    ld.batch_augmentation = {"Vertical": [True]}
    s = ld[0]
    assert s["frames"].shape == (2, 16, 20) and s["frames"].dtype == np.uint8
    assert np.array_equal(s["frames"][0], np.flip(groups["images"][0][2], 0))
    assert np.array_equal(s["frames"][1], np.flip(groups["images"][1][2], 0))
    with pytest.raises(AttributeError):
        H5Loader(_cfg(tmp_path, "bogus", 1, 1, (16, 20)), 2)
    with pytest.raises(FileNotFoundError):
        H5Loader(_cfg(tmp_path, "events", 100, 2, (16, 20)), 2)  # one file, two batch slots
    if not torch.cuda.is_available():
        ld = H5Loader(_cfg(tmp_path, "events", 100, 1, (16, 20)), 2)
        with pytest.raises(_lib.EvflowError):
            ld.custom_collate([ld[0]])  # encodings are made on the MI355X only


def test_ranks_read_disjoint_files(tmp_path):
    for i in range(5):
        _timed_sequence(tmp_path / f"s{i}.npz", n=500, seed=20 + i)
    cfg = _cfg(tmp_path, "events", 100, 1, (16, 20))
    shards = [H5Loader(cfg, 2, rank=r, world_size=2).files for r in range(2)]
    assert [len(f) for f in shards] == [3, 2] and not set(shards[0]) & set(shards[1])
    assert sorted(shards[0] + shards[1]) == sorted(str(tmp_path / f"s{i}.npz") for i in range(5))
    with pytest.raises(FileNotFoundError):
        H5Loader(_cfg(tmp_path, "events", 100, 3, (16, 20)), 2, rank=1, world_size=2)  # 2 files for 3 slots
