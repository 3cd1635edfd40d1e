"""Stored visualisations -- counterpart of the rendering half of reference utils/visualization.py (colour coding
:230-315, folder layout and file names of `Visualization.store` :120-226) with numpy + zlib only (the reference needs
cv2 and matplotlib).  The colour coding (flow_to_image, minmax_norm, events_to_image) is pinned by fixture G17 = the
reference's own functions run on seeded inputs (tools/make_vis_fixture.py; matplotlib exists in the image's conda
interpreter); what goes through cv2 in the reference -- resizing, BGR file writing, the live window (`update`, :28-118) --
has no counterpart to compare with here: the PNG container and the folder layout are checked structurally only.

Host-side by nature: tensors are fetched once per stored frame; nothing here is on the training path."""

import os
import struct
import zlib

import numpy as np


def write_png(path, img):
    """8-bit grey [H,W] or RGB [H,W,3] PNG."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape[:2]
    color = 2 if img.ndim == 3 else 0
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def _hsv_to_rgb(h, s, v):
    i = np.floor(h * 6.0)
    f = h * 6.0 - i
    p, q, t = v * (1.0 - s), v * (1.0 - s * f), v * (1.0 - s * (1.0 - f))
    i = i.astype(np.int64) % 6
    r = np.choose(i, [v, q, p, p, t, v])
    g = np.choose(i, [t, v, v, q, p, p])
    b = np.choose(i, [p, p, t, v, v, q])
    return np.stack([r, g, b], axis=-1)


def flow_to_image(flow_x, flow_y):
    """[H,W] x / y flow -> [H,W,3] uint8 RGB: hue = direction, value = magnitude stretched to the image's range
    (reference :230-255)."""
    mag = np.hypot(flow_x, flow_y)
    lo, span = mag.min(), mag.max() - mag.min()
    val = (mag - lo) / span if span != 0.0 else mag - lo
    hue = (np.arctan2(flow_y, flow_x) + np.pi) / (2.0 * np.pi)
    return (255 * _hsv_to_rgb(hue, np.ones_like(hue), val)).astype(np.uint8)


def minmax_norm(x):
    """Robust min-max normalisation between the 1st and 99th percentile (reference :257-267)."""
    lo, hi = np.percentile(x, 1), np.percentile(x, 99)
    if hi - lo != 0:
        x = (x - lo) / (hi - lo)
    return np.clip(x, 0, 1)


def events_to_image(event_cnt, color_scheme="green_red"):
    """[H,W,2] per-polarity counts -> [H,W] grey or [H,W,3] image in [0,1]; green_red is in the reference's channel
    order (index 1 = positive, index 2 = negative; it is written through cv2, i.e. as B,G,R) (reference :269-315)."""
    pos, neg = event_cnt[:, :, 0].astype(np.float64), event_cnt[:, :, 1].astype(np.float64)
    top = max(np.percentile(pos, 99), np.percentile(neg, 99))
    out = []
    for img in (pos, neg):
        lo = np.percentile(img, 1)
        out.append(np.clip((img - lo) / (top - lo) if lo != top else img, 0, 1))
    pos, neg = out
    if color_scheme == "gray":
        return 0.5 + 0.5 * pos - 0.5 * neg
    image = np.zeros(event_cnt.shape[:2] + (3,))
    image[:, :, 1] = pos
    image[:, :, 2] = neg
    return image


def _hwc(t, channels):
    a = t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
    return a.transpose(0, 2, 3, 1).reshape(a.shape[2], a.shape[3], channels)


def _u8(x):
    return np.clip(np.rint(x), 0, 255).astype(np.uint8)


class Visualization:
    """`store()` of the reference (:120-226): one numbered PNG per element and call under
    <path_results>results/eval_<id>/<sequence>/{events,events_window,flow,flow_window,gtflow,frames,iwe,iwe_window}/
    plus timestamps.txt.  Batch size 1, as in the reference."""

    FOLDERS = ("events", "events_window", "flow", "flow_window", "gtflow", "frames", "iwe", "iwe_window")

    def __init__(self, kwargs, eval_id=-1, path_results=None):
        self.img_idx = 0
        self.px = kwargs.get("vis", {}).get("px", 400)
        self.color_scheme = "green_red"
        self.store_dir = self.store_file = None
        if eval_id >= 0 and path_results is not None:
            self.store_dir = os.path.join(path_results, "results", "eval_" + str(eval_id)) + "/"
            os.makedirs(self.store_dir, exist_ok=True)

    def _events_png(self, path, cnt):
        img = events_to_image(cnt, self.color_scheme) * 255
        write_png(path, _u8(img[:, :, ::-1] if img.ndim == 3 else img))  # cv2 order B,G,R -> PNG order R,G,B

    def store(self, inputs, flow, iwe, sequence, events_window=None, masked_window_flow=None, iwe_window=None, ts=None):
        if self.store_dir is None:
            raise ValueError("Visualization(..., eval_id, path_results) is needed to store images")
        path_to = self.store_dir + sequence + "/"
        if not os.path.exists(path_to):  # new sequence
            for sub in self.FOLDERS:
                os.makedirs(path_to + sub)
            if self.store_file is not None:
                self.store_file.close()
            self.store_file = open(path_to + "timestamps.txt", "w")
            self.img_idx = 0
        name = "/%09d.png" % self.img_idx
        for sub, cnt in (("events", inputs.get("event_cnt")), ("events_window", events_window), ("iwe", iwe), ("iwe_window", iwe_window)):
            if cnt is not None:
                self._events_png(path_to + sub + name, _hwc(cnt, 2))
        for sub, fl in (("flow", flow), ("flow_window", masked_window_flow), ("gtflow", inputs.get("gtflow"))):
            if fl is not None:
                f = _hwc(fl, 2)
                write_png(path_to + sub + name, flow_to_image(f[:, :, 0], f[:, :, 1]))
        if inputs.get("frames") is not None:
            write_png(path_to + "frames" + name, _u8(_hwc(inputs["frames"], 2)[:, :, 1]))
        if ts is not None:
            self.store_file.write(str(ts) + "\n")
            self.store_file.flush()
        self.img_idx += 1
