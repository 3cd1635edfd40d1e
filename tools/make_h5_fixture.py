#!/opt/conda/bin/python3.9
"""Write tests/golden/seq_fixture.h5 (+ its twin seq_fixture_twin.npz) with h5py -- run with an interpreter that HAS h5py
(this image: /opt/conda/bin/python3.9 tools/make_h5_fixture.py; the main /usr/bin/python3 has none).

The file follows the layout the reference's loader reads (dataloader/h5.py:24-42,68,127-131) the way its converters write it
(resizable, chunked 1-D event datasets; one dataset per frame / flow map with a `timestamp` attribute; file attributes `t0`,
`duration`): events/{xs,ys,ts,ps}, images/image%09d, flow_dt1/flow_dt1_%09d, flow_dt4/flow_dt4_%09d.  The twin `.npz` holds
the same arrays under flat names -- what the HDF5 read path is compared with."""
import os

import h5py
import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
H, W, n = 16, 20, 4000
rng = np.random.Generator(np.random.PCG64(77))
ts = np.sort(rng.random(n)) * 2.0 + 100.0
xs = rng.integers(0, W, n).astype(np.int16)
ys = rng.integers(0, H, n).astype(np.int16)
ps = rng.integers(0, 2, n).astype(np.bool_)
stamps = 100.0 + 0.4 * np.arange(1, 5) + 0.01
twin = {"events/xs": xs, "events/ys": ys, "events/ts": ts, "events/ps": ps.astype(np.uint8), "t0": np.float64(ts[0]),
        "duration": np.float64(ts[-1] - ts[0])}
with h5py.File(os.path.join(OUT, "seq_fixture.h5"), "w") as f:
    for name, arr in (("xs", xs), ("ys", ys), ("ts", ts), ("ps", ps)):
        d = f.create_dataset("events/" + name, (0,), dtype=arr.dtype, maxshape=(None,), chunks=True)  # as the packagers do
        for lo in range(0, n, 1500):  # appended in pieces
            hi = min(lo + 1500, n)
            d.resize(hi, axis=0)
            d[lo:hi] = arr[lo:hi]
    f.attrs["t0"] = ts[0]
    f.attrs["duration"] = ts[-1] - ts[0]
    f.attrs["sensor_resolution"] = (H, W)
    for k in range(4):
        img = rng.integers(0, 255, (H, W)).astype(np.uint8)
        d = f.create_dataset(f"images/image{k:09d}", data=img)
        d.attrs["timestamp"] = stamps[k]
        d.attrs["size"] = img.shape
        twin[f"images/image{k:09d}"], twin[f"images_ts/image{k:09d}"] = img, np.float64(stamps[k])
        for g in ("flow_dt1", "flow_dt4"):
            fl = rng.standard_normal((2, H, W)).astype(np.float32)
            d = f.create_dataset(f"{g}/{g}_{k:09d}", data=fl, compression="gzip" if g == "flow_dt4" else None)
            d.attrs["timestamp"] = stamps[k]
            twin[f"{g}/{g}_{k:09d}"], twin[f"{g}_ts/{g}_{k:09d}"] = fl, np.float64(stamps[k])
np.savez(os.path.join(OUT, "seq_fixture_twin.npz"), **twin)
print("wrote", os.path.getsize(os.path.join(OUT, "seq_fixture.h5")), "bytes of HDF5 (h5py", h5py.__version__, ", libhdf5", h5py.version.hdf5_version, ")")
