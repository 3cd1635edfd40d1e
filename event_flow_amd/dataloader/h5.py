"""Event-sequence reader -- mirror of reference dataloader/h5.py:45-343 (H5Loader): one open sequence per batch slot,
input windows by event count / time / APS frame / ground-truth flow map ("events", "time", "frames", "gtflow_dt1",
"gtflow_dt4"), sequence change when a slot runs out, the reference's per-sample dict.

MI355X hand-off: `__getitem__` only slices and formats the raw (x, y, t, p) arrays on the
host; the encodings of the whole batch are binned on the GPU by `custom_collate` (dataloader/base.py) in one
`evf_encode_events` launch, and `__iter__` keeps `prefetch` host batches in flight on a reader thread so that file
reads overlap the previous step's kernels.

Storage: the reference's HDF5 layout (`events/{xs,ys,ts,ps}`, attrs `t0` / `duration`, groups `images`, `flow_dt1`,
`flow_dt4` whose datasets carry a `timestamp` attribute; dataloader/h5.py:24-42,68,127-131) through h5py when it is
installed, through the HDF5 C library + ctypes otherwise (dataloader/hdf5_ctypes.py), and the same layout flattened into `.npz` archives (`events/xs` ..., `t0`, `duration`,
`<group>/<name>` + `<group>_ts/<name>`) otherwise -- `write_npz_sequence` produces them.  Both go through one
`_Sequence` interface, so the windowing logic is tested without h5py."""

import os
import queue
import threading

import numpy as np

from .base import BaseDataLoader
from .encodings import binary_search_array

GROUPS = ("images", "flow_dt1", "flow_dt4")


class _Sequence:
    """What the loader needs from a sequence file."""

    def events(self, name):  # -> array-like with len(), slicing and [-1]
        raise NotImplementedError

    def group(self, group):  # -> (names, timestamps) in the file's visiting order
        raise NotImplementedError

    def read(self, group, name):  # -> ndarray
        raise NotImplementedError

    def close(self):
        pass


class _H5Sequence(_Sequence):
    def __init__(self, path):
        import h5py  # noqa: F401  (not in this image; present wherever the reference runs)

        try:
            import hdf5plugin  # noqa: F401  (compression filters of the public datasets)
        except ImportError:
            pass
        self.f = h5py.File(path, "r")
        self.attrs = {"t0": self.f.attrs["t0"], "duration": self.f.attrs["duration"]}

    def events(self, name):
        return self.f["events/" + name]

    def group(self, group):
        names, ts = [], []

        def visit(name, obj):
            if hasattr(obj, "dtype") and name not in names:
                names.append(name)
                ts.append(obj.attrs["timestamp"])

        self.f[group].visititems(visit)
        return names, ts

    def read(self, group, name):
        return self.f[group][name][:]

    def close(self):
        self.f.close()


class _LibHdf5Sequence(_Sequence):
    """The same `.h5` files through the HDF5 C library and ctypes (dataloader/hdf5_ctypes.py) -- used when h5py is not
    installed for this interpreter."""

    def __init__(self, path):
        from . import hdf5_ctypes

        self.f = hdf5_ctypes.File(path)
        self.attrs = {"t0": self.f.attr("t0"), "duration": self.f.attr("duration")}

    def events(self, name):
        return self.f.dataset("events/" + name)

    def group(self, group):
        names = self.f.names(group)  # name order = the order h5py's visititems reports
        # (one attribute read per frame, nothing kept open: a real sequence has tens of thousands of frame datasets)
        return names, [self.f.dataset_attr(f"{group}/{n}", "timestamp") for n in names]

    def read(self, group, name):
        return self.f.read(f"{group}/{name}")

    def close(self):
        self.f.close()


class _NpzSequence(_Sequence):
    def __init__(self, path):
        self.z = np.load(path, mmap_mode="r", allow_pickle=False)
        self.attrs = {"t0": self.z["t0"].item(), "duration": self.z["duration"].item()}
        self._ev = {}

    def events(self, name):
        if name not in self._ev:
            self._ev[name] = self.z["events/" + name]
        return self._ev[name]

    def group(self, group):
        names = sorted(k[len(group) + 1:] for k in self.z.files if k.startswith(group + "/"))  # h5py visits by name
        return names, [self.z[f"{group}_ts/{n}"].item() for n in names]

    def read(self, group, name):
        return np.asarray(self.z[f"{group}/{name}"])

    def close(self):
        self.z.close()


def open_sequence(path):
    if path.endswith(".npz"):
        return _NpzSequence(path)
    try:
        import h5py  # noqa: F401
    except ImportError:
        return _LibHdf5Sequence(path)  # (raises hdf5_ctypes.Hdf5Error when no HDF5 library can be loaded either)
    return _H5Sequence(path)


def write_npz_sequence(path, xs, ys, ts, ps, t0=None, duration=None, **groups):
    """Write one sequence in the `.npz` flavour of the reference's HDF5 layout.  ps in {0,1} like the datasets;
    groups: images / flow_dt1 / flow_dt4 = list of (name, timestamp, array)."""
    ts = np.asarray(ts, dtype=np.float64)
    a = {"events/xs": np.asarray(xs), "events/ys": np.asarray(ys), "events/ts": ts, "events/ps": np.asarray(ps),
         "t0": np.asarray(ts[0] if t0 is None else t0, dtype=np.float64),
         "duration": np.asarray((ts[-1] - ts[0]) if duration is None else duration, dtype=np.float64)}
    for g, items in groups.items():
        assert g in GROUPS, g
        for name, stamp, arr in items:
            a[f"{g}/{name}"] = np.asarray(arr)
            a[f"{g}_ts/{name}"] = np.asarray(stamp, dtype=np.float64)
    np.savez(path, **a)


class H5Loader(BaseDataLoader):
    def __init__(self, config, num_bins, round_encoding=False, device=None, prefetch=2, rank=0, world_size=1):
        """rank / world_size: data-parallel sharding -- rank r reads the sequence files r, r + world_size, ... with
        its own `loader.batch_size` slots (batch slots are independent sequences)."""
        super().__init__(config, num_bins, round_encoding, device)
        self.last_proc_timestamp = 0
        self.prefetch = prefetch
        self._threaded = self._restarted = self.pass_done = False
        self.mode = self.config["data"]["mode"]
        self.window = self.config["data"]["window"]
        if self.mode not in ("events", "time", "frames", "gtflow_dt1", "gtflow_dt4"):
            print("DataLoader error: Unknown mode.")
            raise AttributeError

        # "memory" that goes from one forward pass to the next (reference :50-55)
        self.batch_idx = list(range(self.batch_size))  # event sequence per slot
        self.batch_row = [0 for _ in range(self.batch_size)]  # event_idx / time_idx / frame_idx / gt_idx per slot

        self.files = []
        for root, _dirs, files in os.walk(config["data"]["path"]):
            for file in sorted(files):
                if file.endswith(".h5") or file.endswith(".npz"):
                    self.files.append(os.path.join(root, file))
        self.files = sorted(self.files)[rank::world_size]
        if len(self.files) < self.batch_size:
            raise FileNotFoundError(f"{config['data']['path']}: {len(self.files)} sequence file(s) for batch_size "
                                    f"{self.batch_size} (the reference opens one file per batch slot, h5.py:64-68)")

        self.open_files, self.batch_last_ts, self.open_files_stamps = [], [], []
        for batch in range(self.batch_size):
            self.open_files.append(None)
            self.batch_last_ts.append(0)
            self.open_files_stamps.append(None)
            self._open(batch, self.files[batch])

    # ------------------------------------------------------------------ files
    def _stamp_group(self):
        return {"frames": "images", "gtflow_dt1": "flow_dt1", "gtflow_dt4": "flow_dt4"}.get(self.mode)

    def _open(self, batch, path):
        if self.open_files[batch] is not None:
            self.open_files[batch].close()
        seq = open_sequence(path)
        self.open_files[batch] = seq
        self.batch_last_ts[batch] = seq.events("ts")[-1] - seq.attrs["t0"]
        g = self._stamp_group()
        self.open_files_stamps[batch] = seq.group(g) if g else None

    def get_iters(self, batch):
        """Number of forward passes of the slot's sequence for the input mode and window (reference :95-112)."""
        if self.mode == "events":
            max_iters = len(self.open_files[batch].events("xs"))
        elif self.mode == "time":
            max_iters = self.open_files[batch].attrs["duration"]
        else:
            max_iters = len(self.open_files_stamps[batch][1]) - 1
        return max_iters // self.window

    def get_events(self, file, idx0, idx1):
        """All events between two indices, timestamps relative to the sequence start (reference :114-134)."""
        xs = np.asarray(file.events("xs")[idx0:idx1])
        ys = np.asarray(file.events("ys")[idx0:idx1])
        ts = np.asarray(file.events("ts")[idx0:idx1], dtype=np.float64) - file.attrs["t0"]
        ps = np.asarray(file.events("ps")[idx0:idx1])
        if ts.shape[0] > 0:
            self.last_proc_timestamp = ts[-1]
        return xs, ys, ts, ps

    def find_ts_index(self, file, timestamp):
        """Closest event index of a timestamp, by binary search (reference :178-183)."""
        return binary_search_array(file.events("ts"), timestamp)

    def _stamp_rows(self, batch, window):
        idx0 = int(np.floor(self.batch_row[batch]))
        idx1 = int(np.ceil(self.batch_row[batch] + window))
        if window < 1.0 and idx1 - idx0 > 1:
            idx0 += idx1 - idx0 - 1
        return idx0, idx1

    def get_event_index(self, batch, window=0):
        """Event indices of the slot's next input window (reference :136-176)."""
        f = self.open_files[batch]
        if self.mode == "events":
            return self.batch_row[batch], self.batch_row[batch] + window
        if self.mode == "time":
            t = self.batch_row[batch] + f.attrs["t0"]
            return self.find_ts_index(f, t), self.find_ts_index(f, t + window)
        idx0, idx1 = self._stamp_rows(batch, window)
        stamps = self.open_files_stamps[batch][1]
        return self.find_ts_index(f, stamps[idx0]), self.find_ts_index(f, stamps[idx1])

    # ------------------------------------------------------------------ one sample
    def __getitem__(self, index):
        """One input window of batch slot `index % batch_size` (reference :185-343): formatted event rows and
        scalars on the host; `custom_collate` makes the encodings on the GPU."""
        stamped = self.mode in ("frames", "gtflow_dt1", "gtflow_dt4")
        while True:
            batch = index % self.batch_size
            f = self.open_files[batch]
            restart = False
            if stamped and int(np.ceil(self.batch_row[batch] + self.window)) >= len(self.open_files_stamps[batch][1]):
                restart = True

            xs = ys = ts = ps = np.zeros((0))
            if not restart:
                idx0, idx1 = self.get_event_index(batch, window=self.window)
                if stamped and self.window < 1.0:  # a fraction of the interval between two stamps (reference :214-229)
                    floor_row, _ = self._stamp_rows(batch, self.window)
                    idx0_change = self.batch_row[batch] - floor_row
                    idx1_change = self.batch_row[batch] + self.window - floor_row
                    delta_idx = idx1 - idx0
                    idx1 = int(idx0 + idx1_change * delta_idx)
                    idx0 = int(idx0 + idx0_change * delta_idx)
                xs, ys, ts, ps = self.get_events(f, idx0, idx1)

            if (self.mode == "events" and xs.shape[0] < self.window) or (
                self.mode == "time" and self.batch_row[batch] + self.window >= self.batch_last_ts[batch]
            ):
                restart = True

            if xs.shape[0] <= 10:  # very few events: an empty window (reference :240-245)
                xs = ys = ts = ps = np.empty([0])

            if restart:  # next sequence for this slot (reference :247-275)
                self._restarted = True
                if not self._threaded:  # under __iter__'s reader thread the flag travels with its batch instead
                    self.new_seq = True
                self.reset_sequence(batch)
                self.batch_row[batch] = 0
                self.batch_idx[batch] = max(self.batch_idx) + 1
                self._open(batch, self.files[self.batch_idx[batch] % len(self.files)])
                continue

            dt_input = np.asarray(0.0)
            if ts.shape[0] > 0:
                dt_input = np.asarray(ts[-1] - ts[0])
            xs, ys, ts, ps = self.event_formatting(xs, ys, ts, ps)
            xs, ys, ps = self.augment_events(xs, ys, ps, batch)

            output = {"event_list": self.create_list_encoding(xs, ys, ts, ps)}
            if self.config["hot_filter"]["enabled"]:
                output["hot_mask"] = self.create_hot_mask(xs, ys, ps, batch)

            if self.mode == "frames":
                curr_idx = int(np.floor(self.batch_row[batch]))
                next_idx = int(np.ceil(self.batch_row[batch] + self.window))
                names = self.open_files_stamps[batch][0]
                frames = np.zeros((2, self.res[0], self.res[1]))
                frames[0] = self.augment_frames(f.read("images", names[curr_idx]), batch)
                frames[1] = self.augment_frames(f.read("images", names[next_idx]), batch)
                output["frames"] = frames.astype(np.uint8)

            dt_gt = 0.0
            if self.mode in ("gtflow_dt1", "gtflow_dt4"):
                idx = int(np.ceil(self.batch_row[batch] + self.window))
                names, stamps = self.open_files_stamps[batch]
                output["gtflow"] = self.augment_flowmap(f.read(self._stamp_group(), names[idx]), batch)
                if idx > 0:
                    dt_gt = stamps[idx] - stamps[idx - 1]
            output["dt_gt"] = np.asarray(dt_gt)
            output["dt_input"] = dt_input

            self.batch_row[batch] += self.window
            return output

    # ------------------------------------------------------------------ batches
    def _host_batches(self):
        """Batches of samples in DataLoader order (sample i belongs to slot i % B) with the flags that belong to
        them: (a slot restarted, every file has been started once).  The pass ends with the batch that wraps around
        (train_flow.py:106-123 trains on it, eval_flow.py:124-127 drops it)."""
        index = 0
        while True:
            self._restarted = False
            samples = [self[index + b] for b in range(self.batch_size)]
            index += self.batch_size
            done = self.seq_num >= len(self.files)
            yield samples, (self._restarted, done)
            if done:
                return

    def _deliver(self, samples, flags):
        batch = self.custom_collate(samples)
        self.new_seq, self.pass_done = flags
        self.samples += self.batch_size
        return batch

    def _end_pass(self):
        self.new_seq = False
        if self.seq_num >= len(self.files):
            self.seq_num %= len(self.files)
            self.epoch += 1

    def __iter__(self):
        """Yields the collated GPU batch dicts of one pass over the files.  While a batch is out, `self.new_seq`
        (a slot of it started a new sequence) and `self.pass_done` (it is the wrap-around batch) describe THAT
        batch, although the reader thread is already `prefetch` batches ahead."""
        self._threaded = bool(self.prefetch)
        try:
            if not self.prefetch:
                for samples, flags in self._host_batches():
                    yield self._deliver(samples, flags)
                return
            q = queue.Queue(maxsize=self.prefetch)
            stop = threading.Event()

            def put(item):
                """Blocking put that gives up once the consumer has closed the iterator (a full queue nobody
                reads would otherwise hold the thread, and the consumer's join, forever) -> delivered?"""
                while not stop.is_set():
                    try:
                        q.put(item, timeout=0.1)
                        return True
                    except queue.Full:
                        continue
                return False

            def produce():
                try:
                    for item in self._host_batches():
                        if not put(item):
                            return
                    put(None)
                except BaseException as e:  # surfaced in the consumer
                    put(e)

            th = threading.Thread(target=produce, daemon=True)
            th.start()
            try:
                while True:
                    item = q.get()
                    if item is None:
                        break
                    if isinstance(item, BaseException):
                        raise item
                    yield self._deliver(*item)
            finally:
                stop.set()
                th.join()
        finally:
            self._threaded = False
            self._end_pass()
