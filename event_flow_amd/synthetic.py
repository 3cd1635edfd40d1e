"""Platform-stable synthetic event windows (numpy PCG64; identical bytes on
every box).  Stands in for the HDF5 reader (reference dataloader/h5.py) in the
benchmark and the parity tests: produces the formatted event tuple
(xs, ys, ts in [0,1], ps in {-1,+1}) that reference
dataloader/base.py:66-86 (`event_formatting`) hands to the encodings.

Two generators:
  uniform      x,y uniform over the sensor, t sorted uniform, p = +-1
  moving_dots  K dots translating with one constant (u,v) per sample; gives a
               ground-truth flow for AEE and the argmin known-answer test.
"""

import numpy as np


def _rng(seed):
    return np.random.Generator(np.random.PCG64(int(seed)))


def uniform_events(n, H, W, seed):
    """-> xs, ys, ts, ps float32 [n]; ts sorted, ts[0]=0, ts[-1]=1."""
    g = _rng(seed)
    xs = g.integers(0, W, size=n).astype(np.float32)
    ys = g.integers(0, H, size=n).astype(np.float32)
    ts = np.sort(g.random(n)).astype(np.float64)
    if n > 1:
        ts = (ts - ts[0]) / (ts[-1] - ts[0])
    ps = (g.integers(0, 2, size=n) * 2 - 1).astype(np.float32)
    return xs, ys, ts.astype(np.float32), ps


def moving_dots_events(n, H, W, seed, max_disp=20.0, k=200):
    """Events emitted along the trajectories of k dots that all translate by
    (u, v) pixels over the window.  -> xs, ys, ts, ps, (u, v)."""
    g = _rng(seed)
    u, v = g.uniform(-max_disp, max_disp, size=2)
    x0 = g.uniform(0, W, size=k)
    y0 = g.uniform(0, H, size=k)
    pol = (g.integers(0, 2, size=k) * 2 - 1).astype(np.float32)
    ts = np.sort(g.random(n))
    if n > 1:
        ts = (ts - ts[0]) / (ts[-1] - ts[0])
    d = g.integers(0, k, size=n)
    x = x0[d] + u * ts + g.integers(-1, 2, size=n)
    y = y0[d] + v * ts + g.integers(-1, 2, size=n)
    xs = np.clip(np.rint(x), 0, W - 1).astype(np.float32)
    ys = np.clip(np.rint(y), 0, H - 1).astype(np.float32)
    return xs, ys, ts.astype(np.float32), pol[d], (float(u), float(v))


def event_list_batch(B, n, H, W, seed0, kind="uniform"):
    """[B,n,4] float32 rows (t, y, x, p) -- the layout of `event_list` after
    reference custom_collate (dataloader/base.py:197-208,248-265)."""
    out = np.empty((B, n, 4), dtype=np.float32)
    gt = []
    for b in range(B):
        if kind == "uniform":
            xs, ys, ts, ps = uniform_events(n, H, W, seed0 + b)
        else:
            xs, ys, ts, ps, uv = moving_dots_events(n, H, W, seed0 + b)
            gt.append(uv)
        out[b, :, 0], out[b, :, 1], out[b, :, 2], out[b, :, 3] = ts, ys, xs, ps
    return (out, gt) if kind != "uniform" else out


def seed_for(config_id, rank, sample):
    """seed = 1000*config + 17*rank + sample."""
    return 1000 * int(config_id) + 17 * int(rank) + int(sample)
