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


# ------------------------------------------------------------------ real HDF5 files (fixture written by h5py)
GOLDEN = __import__("os").path.join(__import__("os").path.dirname(__file__), "golden")


def _h5_reader_available():
    try:
        import h5py  # noqa: F401
        return True
    except ImportError:
        from event_flow_amd.dataloader import hdf5_ctypes
        return hdf5_ctypes.available()


needs_hdf5 = pytest.mark.skipif(not _h5_reader_available(), reason="neither h5py nor an HDF5 C library on this box")


@needs_hdf5
def test_hdf5_file_written_by_h5py_reads_back_exactly():
    """tests/golden/seq_fixture.h5 was written by h5py 3.3 / libhdf5 1.10 in the reference's layout (tools/make_h5_fixture.py:
    resizable chunked event datasets appended in pieces, boolean polarities, gzip-compressed flow maps, timestamp attributes);
    seq_fixture_twin.npz holds the same arrays.  The sequence object the loader uses must give back exactly those arrays:
    whole datasets, slices, single elements, file and dataset attributes, group members in name order."""
    from event_flow_amd.dataloader.h5 import open_sequence

    tw = np.load(GOLDEN + "/seq_fixture_twin.npz")
    seq = open_sequence(GOLDEN + "/seq_fixture.h5")
    assert float(seq.attrs["t0"]) == float(tw["t0"]) and float(seq.attrs["duration"]) == float(tw["duration"])
    for name in ("xs", "ys", "ts", "ps"):
        d = seq.events(name)
        ref = tw["events/" + name]
        assert len(d) == len(ref) == 4000
        assert np.array_equal(np.asarray(d[:]).astype(ref.dtype), ref)
        assert np.array_equal(np.asarray(d[1490:1510]).astype(ref.dtype), ref[1490:1510])  # across the append / chunk boundary
        assert d[-1] == ref[-1] and d[0] == ref[0] and d[2777] == ref[2777]
        assert np.asarray(d[4000:4000]).shape == (0,)
    for g in ("images", "flow_dt1", "flow_dt4"):
        names, stamps = seq.group(g)
        want = sorted(k[len(g) + 1:] for k in tw.files if k.startswith(g + "/"))
        assert names == want and len(names) == 4
        for n_, st_ in zip(names, stamps):
            assert float(st_) == float(tw[f"{g}_ts/{n_}"])
            got = seq.read(g, n_)
            assert got.dtype == tw[f"{g}/{n_}"].dtype and np.array_equal(got, tw[f"{g}/{n_}"])
    seq.close()


@needs_hdf5
@pytest.mark.parametrize("mode,window", [("events", 300), ("time", 0.25), ("gtflow_dt1", 1), ("gtflow_dt4", 0.25), ("frames", 1)])
def test_loader_on_hdf5_equals_loader_on_npz(tmp_path, mode, window):
    """The reference's H5Loader logic (dataloader/h5.py:136-295) over the real HDF5 file gives sample for sample what it
    gives over the `.npz` flavour of the same sequence (whose windowing the tests above pin)."""
    import shutil

    tw = np.load(GOLDEN + "/seq_fixture_twin.npz")
    (tmp_path / "h5").mkdir()
    (tmp_path / "npz").mkdir()
    shutil.copy(GOLDEN + "/seq_fixture.h5", tmp_path / "h5" / "a.h5")
    groups = {g: [(k[len(g) + 1:], float(tw[f"{g}_ts/{k[len(g) + 1:]}"]), tw[k]) for k in sorted(tw.files) if k.startswith(g + "/")]
              for g in ("images", "flow_dt1", "flow_dt4")}
    write_npz_sequence(str(tmp_path / "npz" / "a.npz"), tw["events/xs"], tw["events/ys"], tw["events/ts"], tw["events/ps"], **groups)
    a = H5Loader(_cfg(tmp_path / "h5", mode, window, 1, (16, 20), ("Horizontal", "Polarity")), 2)
    b = H5Loader(_cfg(tmp_path / "npz", mode, window, 1, (16, 20), ("Horizontal", "Polarity")), 2)
    a.batch_augmentation = {"Horizontal": [True], "Polarity": [False]}
    b.batch_augmentation = {"Horizontal": [True], "Polarity": [False]}
    assert a.get_iters(0) == b.get_iters(0)
    for i in range({"gtflow_dt1": 3, "frames": 2}.get(mode, 4)):  # (stay inside the sequence: a restart re-draws the augmentation)
        sa, sb = a[i], b[i]
        assert sa.keys() == sb.keys()
        for k in sa:
            va, vb = np.asarray(sa[k]), np.asarray(sb[k])
            assert va.dtype == vb.dtype and np.array_equal(va, vb), (mode, i, k)
    assert (a.seq_num, a.batch_row) == (b.seq_num, b.batch_row)
