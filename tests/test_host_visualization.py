"""Stored visualisations (event_flow_amd/utils/visualization.py): known-answer colours and the reference's folder
layout.  The reference's own renderer needs cv2 / matplotlib, so there is no reference-made fixture for this module."""

import struct
import zlib

import numpy as np
import torch

from event_flow_amd.utils.visualization import Visualization, events_to_image, flow_to_image, minmax_norm


def _read_png(path):
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    w, h, depth, color = struct.unpack(">IIBB", b[16:26])
    assert depth == 8
    i, data = 8, b""
    while i < len(b):
        n, tag = struct.unpack(">I", b[i:i + 4])[0], b[i + 4:i + 8]
        assert struct.unpack(">I", b[i + 8 + n:i + 12 + n])[0] == zlib.crc32(tag + b[i + 8:i + 8 + n]) & 0xFFFFFFFF
        if tag == b"IDAT":
            data += b[i + 8:i + 8 + n]
        i += 12 + n
    ch = 3 if color == 2 else 1
    raw = np.frombuffer(zlib.decompress(data), np.uint8).reshape(h, 1 + w * ch)
    assert not raw[:, 0].any()  # filter type 0 on every row
    return raw[:, 1:].reshape(h, w, ch) if ch == 3 else raw[:, 1:]


def test_flow_colour_wheel():
    fx = np.array([[1.0, -1.0, 0.0], [0.0, 0.5, 0.0]])
    fy = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0]])
    img = flow_to_image(fx, fy)
    assert img.shape == (2, 3, 3) and img.dtype == np.uint8
    assert tuple(img[0, 0]) == (0, 255, 255)      # +x: hue 0.5
    assert tuple(img[0, 1]) == (255, 0, 0)        # -x: hue 0 (== 1)
    assert tuple(img[0, 2]) == (127, 0, 255)      # +y: hue 0.75
    assert tuple(img[1, 0]) == (127, 255, 0)      # -y: hue 0.25
    assert tuple(img[1, 1]) == (0, 127, 127)      # half magnitude
    assert tuple(img[1, 2]) == (0, 0, 0)          # no flow
    assert not flow_to_image(np.zeros((2, 2)), np.zeros((2, 2))).any()


def test_event_images_and_norm():
    cnt = np.zeros((4, 5, 2))
    cnt[0, 0, 0], cnt[1, 1, 1], cnt[2, 2] = 2, 2, (2, 2)
    img = events_to_image(cnt)
    assert tuple(img[0, 0]) == (0, 1, 0) and tuple(img[1, 1]) == (0, 0, 1) and tuple(img[2, 2]) == (0, 1, 1) and not img[3, 3].any()
    g = events_to_image(cnt, "gray")
    assert g[0, 0] == 1.0 and g[1, 1] == 0.0 and g[2, 2] == 0.5 and g[3, 3] == 0.5
    x = np.arange(101, dtype=np.float64)
    n = minmax_norm(x)
    assert n.min() == 0 and n.max() == 1 and abs(n[50] - 0.5) < 1e-12


def test_store_layout_and_png_round_trip(tmp_path):
    vis = Visualization({"vis": {"px": 400}}, eval_id=3, path_results=str(tmp_path) + "/")
    cnt = torch.zeros(1, 2, 6, 8)
    cnt[0, 0, 1, 2], cnt[0, 1, 3, 4] = 3, 3
    flow = torch.zeros(1, 2, 6, 8)
    flow[0, 0, 2, 2] = 1.0
    frames = torch.arange(2 * 6 * 8, dtype=torch.uint8).view(1, 2, 6, 8)
    for k in range(2):
        vis.store({"event_cnt": cnt, "gtflow": flow, "frames": frames}, flow, cnt, "seqA", ts=0.5 * k)
    base = tmp_path / "results" / "eval_3" / "seqA"
    assert sorted(p.name for p in base.iterdir()) == sorted(list(Visualization.FOLDERS) + ["timestamps.txt"])
    assert (base / "timestamps.txt").read_text() == "0.0\n0.5\n"
    for sub in ("events", "flow", "gtflow", "frames", "iwe"):
        assert sorted(p.name for p in (base / sub).iterdir()) == ["000000000.png", "000000001.png"]
    ev = _read_png(base / "events" / "000000000.png")
    assert tuple(ev[1, 2]) == (0, 255, 0) and tuple(ev[3, 4]) == (255, 0, 0)  # positive green, negative red
    fl = _read_png(base / "flow" / "000000001.png")
    assert tuple(fl[2, 2]) == (0, 255, 255) and not fl[0, 0].any()
    fr = _read_png(base / "frames" / "000000000.png")
    assert np.array_equal(fr, frames[0, 1].numpy())


def test_colour_coding_against_the_reference_functions():
    """Fixture G17 = the reference's own Visualization.flow_to_image / minmax_norm / events_to_image
    (utils/visualization.py:230-315, matplotlib's hsv_to_rgb inside) on seeded inputs, incl. axis-aligned / zero / constant
    flow and single-polarity counts (tools/make_vis_fixture.py)."""
    import os

    from event_flow_amd.utils import visualization as vis

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g17_visualization.npz"))
    for k in range(4):
        got = vis.flow_to_image(g[f"flow{k}_x"], g[f"flow{k}_y"])
        ref = g[f"flow{k}_rgb"]
        assert got.shape == ref.shape and got.dtype == np.uint8
        d = np.abs(got.astype(int) - ref.astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 0.01, (k, d.max(), (d > 0).mean())  # 255 * x truncated: one level at exact ties
    for k in range(3):
        np.testing.assert_allclose(vis.events_to_image(g[f"cnt{k}"], "green_red"), g[f"cnt{k}_green_red"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(vis.events_to_image(g[f"cnt{k}"], "gray"), g[f"cnt{k}_gray"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(vis.minmax_norm(g[f"mm{k}_in"]), g[f"mm{k}_out"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(vis.minmax_norm(g["mm_const_in"]), g["mm_const_out"], rtol=0, atol=0)
