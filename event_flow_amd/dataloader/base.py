"""Host side of the event loader -- mirror of reference dataloader/base.py:10-283 (BaseDataLoader): per-slot
augmentation flags, sequence reset, timestamp normalisation, flips of events / frames / flow maps, the hot-pixel
statistics and the collate step.

What differs from the reference is WHERE the encodings are made: the reference bins every sample on the CPU inside
`__getitem__` (create_cnt/mask/voxel_encoding, dataloader/h5.py:282-286) and stacks CPU tensors; here `__getitem__`
hands back the formatted event rows and `custom_collate` uploads ONE [B,N,4] block and bins the whole batch with one
`evf_encode_events` launch on the MI355X (ragged windows are padded with p = 0 rows, which the kernel ignores).
The batch dict has the reference's keys, shapes and values."""

import random

import numpy as np
import torch

from .. import _lib
from .encodings import encode_event_list, get_hot_event_mask


class BaseDataLoader(torch.utils.data.Dataset):
    def __init__(self, config, num_bins, round_encoding=False, device=None):
        self.config = config
        self.epoch = 0
        self.seq_num = 0
        self.samples = 0
        self.new_seq = False
        self.num_bins = num_bins
        self.round_encoding = round_encoding
        self.device = device if device is not None else "cuda:0"
        self.batch_size = self.config["loader"]["batch_size"]
        self.res = tuple(self.config["loader"]["resolution"])

        # batch-specific data augmentation mechanisms (reference :24-32)
        self.batch_augmentation = {}
        for mechanism in self.config["loader"]["augment"]:
            self.batch_augmentation[mechanism] = [False for _ in range(self.batch_size)]
        for i, mechanism in enumerate(self.config["loader"]["augment"]):
            for batch in range(self.batch_size):
                if np.random.random() < self.config["loader"]["augment_prob"][i]:
                    self.batch_augmentation[mechanism][batch] = True

        # hot pixels (reference :34-39)
        if self.config["hot_filter"]["enabled"]:
            self.hot_idx = [0 for _ in range(self.batch_size)]
            self.hot_events = [torch.zeros(self.res) for _ in range(self.batch_size)]

    def __getitem__(self, index):
        raise NotImplementedError

    def reset_sequence(self, batch):
        """Reset the sequence-specific variables of one batch slot (reference :49-65)."""
        self.seq_num += 1
        if self.config["hot_filter"]["enabled"]:
            self.hot_idx[batch] = 0
            self.hot_events[batch] = torch.zeros(self.res)
        for i, mechanism in enumerate(self.config["loader"]["augment"]):
            self.batch_augmentation[mechanism][batch] = bool(np.random.random() < self.config["loader"]["augment_prob"][i])

    @staticmethod
    def event_formatting(xs, ys, ts, ps):
        """fp32 arrays, polarity {0,1} -> {-1,+1}, timestamps normalised to [0,1] (reference :67-86)."""
        xs = np.asarray(xs).astype(np.float32)
        ys = np.asarray(ys).astype(np.float32)
        ts = np.asarray(ts).astype(np.float32)
        ps = np.asarray(ps).astype(np.float32) * 2 - 1
        if ts.shape[0] > 0:
            ts = (ts - ts[0]) / (ts[-1] - ts[0])
        return xs, ys, ts, ps

    def _flipped(self, mechanism, batch):
        flags = self.batch_augmentation.get(mechanism)
        return bool(flags) and bool(flags[batch])

    def augment_events(self, xs, ys, ps, batch):
        """Horizontal / vertical / polarity flips of one slot's events (reference :88-116)."""
        H, W = self.res
        if self._flipped("Horizontal", batch):
            xs = (W - 1) - xs
        if self._flipped("Vertical", batch):
            ys = (H - 1) - ys
        if self._flipped("Polarity", batch):
            ps = -ps
        return xs, ys, ps

    def augment_frames(self, img, batch):
        """The same spatial flips on an APS frame [H,W] (reference :118-131)."""
        axes = [ax for ax, mech in ((1, "Horizontal"), (0, "Vertical")) if self._flipped(mech, batch)]
        return np.flip(img, axes) if axes else img

    def augment_flowmap(self, flowmap, batch):
        """... and on a [2,H,W] (x, y) flow map, where a flipped axis also negates its flow component
        (reference :133-148)."""
        out = np.array(flowmap, dtype=np.float32, copy=True)
        for comp, axis, mech in ((0, 2, "Horizontal"), (1, 1, "Vertical")):
            if self._flipped(mech, batch):
                out = np.flip(out, axis).copy()
                out[comp] *= -1.0
        return out

    @staticmethod
    def create_list_encoding(xs, ys, ts, ps):
        """[4,N] rows (ts, ys, xs, ps) (reference :197-208)."""
        return np.stack([ts, ys, xs, ps]).astype(np.float32)

    def create_hot_mask(self, xs, ys, ps, batch):
        """Binary [H,W] mask that removes pixels with a high event rate (reference :224-243).  The reference derives
        the per-window occupancy from the count image; the events' own pixels give the same set."""
        hot_update = torch.zeros(self.res)
        if len(xs):
            keep = ps != 0
            hot_update[torch.from_numpy(ys[keep].astype(np.int64)), torch.from_numpy(xs[keep].astype(np.int64))] = 1
        self.hot_events[batch] += hot_update
        self.hot_idx[batch] += 1
        event_rate = self.hot_events[batch] / self.hot_idx[batch]
        return get_hot_event_mask(event_rate, self.hot_idx[batch], max_px=self.config["hot_filter"]["max_px"],
                                  min_obvs=self.config["hot_filter"]["min_obvs"], max_rate=self.config["hot_filter"]["max_rate"])

    def __len__(self):
        return 1000  # not used (reference :245-246)

    def custom_collate(self, batch):
        """List of `__getitem__` samples -> the reference's batch dict (reference :248-265), on the GPU:
        event_cnt [B,2,H,W], event_voxel [B,nb,H,W], event_mask [B,1,H,W], event_list [B,N,4],
        event_list_pol_mask [B,N,2], dt_gt [B], dt_input [B] (+ gtflow [B,2,H,W], frames [B,2,H,W])."""
        B = len(batch)
        n = max(s["event_list"].shape[1] for s in batch)
        ev = torch.zeros((B, max(n, 1), 4), dtype=torch.float32).pin_memory() if torch.cuda.is_available() else \
            torch.zeros((B, max(n, 1), 4), dtype=torch.float32)
        for b, s in enumerate(batch):
            k = s["event_list"].shape[1]
            if k:
                ev[b, :k] = torch.from_numpy(np.ascontiguousarray(s["event_list"].T))
        _lib.load()  # no CPU path: fail before touching the device when the library or the GPU is missing
        if not torch.cuda.is_available():
            raise _lib.EvflowError("custom_collate: the encodings are made on the MI355X only (no CPU fallback)")
        ev = ev.to(self.device, non_blocking=True)
        out = encode_event_list(ev, self.num_bins, self.res, round_ts=self.round_encoding)
        if n == 0:  # only padding rows: hand back empty lists like the reference would
            out["event_list"] = out["event_list"][:, :0]
            out["event_list_pol_mask"] = out["event_list_pol_mask"][:, :0]
        if self.config["hot_filter"]["enabled"]:
            hot = torch.stack([s["hot_mask"] for s in batch]).to(self.device, non_blocking=True).contiguous()
            H, W = self.res
            for key, C in (("event_voxel", self.num_bins), ("event_cnt", 2), ("event_mask", 1)):
                _lib.call("evf_apply_pixel_mask", _lib.ptr(out[key]), _lib.ptr(hot), B, C, H, W)
        for key in ("dt_gt", "dt_input"):
            out[key] = torch.from_numpy(np.stack([np.asarray(s[key], dtype=np.float64) for s in batch])).to(self.device)
        if "gtflow" in batch[0]:
            out["gtflow"] = torch.from_numpy(np.stack([s["gtflow"] for s in batch])).to(self.device)
        if "frames" in batch[0]:
            out["frames"] = torch.from_numpy(np.stack([s["frames"] for s in batch])).to(self.device)
        return out

    def shuffle(self, flag=True):
        """Shuffle the sequence order (reference :267-273)."""
        if flag:
            random.shuffle(self.files)
