"""Synthetic stand-in for the reference's HDF5 loader (dataloader/h5.py:45-343, not portable here: no h5py and no
datasets on the box).  It honours the same contract towards the drivers:

  * iterating yields the batch dict of `H5Loader.__getitem__` after `custom_collate` (dataloader/base.py:248-265):
    event_cnt [B,2,H,W], event_voxel [B,nb,H,W], event_mask [B,1,H,W], event_list [B,N,4] (ts,y,x,p),
    event_list_pol_mask [B,N,2], plus gtflow [B,2,H,W] and dt_gt / dt_input for evaluation;
  * `.new_seq` is raised when a batch slot starts a new sequence (train_flow.py:100-105), `.shuffle()`, `.seq_num`.

Events come from event_flow_amd.synthetic (moving dots with a known motion, or uniform noise) and are encoded on the
GPU by one `evf_encode_events` launch per batch -- the on-device binning hand-off SURVEY.md section 8(f)2 asks for."""

import numpy as np
import torch

from .. import synthetic
from .encodings import encode_event_list


class SyntheticLoader:
    def __init__(self, config, num_bins, round_encoding=False, device="cuda:0", kind="moving_dots", windows_per_seq=20,
                 num_sequences=8, max_disp=6.0, seed_offset=0):
        self.config = config
        self.num_bins = num_bins
        self.round_encoding = round_encoding
        self.res = tuple(config["loader"]["resolution"])
        self.batch_size = config["loader"]["batch_size"]
        self.n_events = int(config["data"]["window"])
        self.device = device
        self.kind = kind
        self.windows_per_seq = windows_per_seq
        self.num_sequences = num_sequences
        self.max_disp = max_disp
        self.seed_offset = seed_offset  # data-parallel ranks draw different sequences
        self.new_seq = False
        self.seq_num = 0
        self._order = np.arange(num_sequences)
        self.samples = 0

    def shuffle(self, seed=0):
        self._order = np.random.default_rng(seed + self.samples).permutation(self.num_sequences)

    def __len__(self):
        return self.num_sequences * self.windows_per_seq // self.batch_size

    def _window(self, seq, w):
        H, W = self.res
        total = self.n_events * self.windows_per_seq
        if self.kind == "uniform":
            xs, ys, ts, ps = synthetic.uniform_events(total, H, W, 100 + self.seed_offset + seq)
            uv = (0.0, 0.0)
        else:
            xs, ys, ts, ps, uv = synthetic.moving_dots_events(total, H, W, 100 + self.seed_offset + seq, max_disp=self.max_disp * self.windows_per_seq)
        sl = slice(w * self.n_events, (w + 1) * self.n_events)
        t = ts[sl].astype(np.float64)
        t = (t - t[0]) / max(t[-1] - t[0], 1e-12)  # event_formatting: ts normalised per window (base.py:84-85)
        ev = np.stack([t.astype(np.float32), ys[sl], xs[sl], ps[sl]], 1)
        return ev, (uv[0] / self.windows_per_seq, uv[1] / self.windows_per_seq)

    def __iter__(self):
        H, W = self.res
        B = self.batch_size
        slots = [(int(self._order[b % self.num_sequences]), 0) for b in range(B)]
        nxt = B
        for _ in range(len(self)):
            evs, gts = [], []
            self.new_seq = False
            for b in range(B):
                seq, w = slots[b]
                if w >= self.windows_per_seq:  # this slot's sequence is over: start the next one
                    seq, w = int(self._order[nxt % self.num_sequences]), 0
                    nxt += 1
                    self.seq_num += 1
                    self.new_seq = True
                ev, uv = self._window(seq, w)
                slots[b] = (seq, w + 1)
                evs.append(ev)
                gts.append(uv)
            ev = torch.from_numpy(np.stack(evs)).to(self.device)
            out = encode_event_list(ev, self.num_bins, (H, W), round_ts=self.round_encoding)
            gt = torch.zeros(B, 2, H, W, device=self.device)
            for b, (u, v) in enumerate(gts):
                gt[b, 0], gt[b, 1] = u, v
            out["gtflow"] = gt
            out["dt_gt"] = torch.ones(B, device=self.device)
            out["dt_input"] = torch.ones(B, device=self.device)
            self.samples += B
            yield out
