"""k_bwd_win_plif (a feed-forward PLIF hidden cell, the passes of a window in one launch) against one evf_plif_bwd_wgrad2 launch per
pass at the config-5 shape:  python tools/debug/plif_window_bench.py [B H W passes]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from event_flow_amd import _lib  # noqa: E402

B, H, W, T = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (4, 260, 346, 10)
dev, C = "cuda:0", 32
P = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
f = lambda *s: torch.randn(*s, device=dev) * 0.3  # noqa: E731
L = _lib.load()
nsl = max(L.evf_lif_bwd_wgrad_slabs(B, H, W), 512)
row_ld = 160
leak, thresh, lpt, apt = f(32), f(32) + 0.5, f(32) - 1, f(32) - 2
vs = [f(B, H, W, C) for _ in range(T + 1)]
pts = [f(B, H, W, C).abs() for _ in range(T)]
zs = [torch.randint(0, 2 ** 31 - 1, (B, H, W), dtype=torch.int32, device=dev) for _ in range(T)]
nW = (W + 31) // 32
xT = [torch.randint(0, 2 ** 31 - 1, (B, H, 32, nW), dtype=torch.int32, device=dev) for _ in range(T)]
Ps = [f(B, H, W).abs() for _ in range(T)]
gzs = [f(B, H, W, C) for _ in range(T)]
gcur = [torch.empty(B, H, W, C, device=dev) for _ in range(T)]
gP = [torch.empty(B, H, W, device=dev) for _ in range(T)]
gv, gpt = torch.empty(B, H, W, C, device=dev), torch.empty(B, H, W, C, device=dev)
rows, slab = torch.zeros(nsl, row_ld, device=dev), torch.zeros(nsl, 9216, device=dev)


def per_pass():
    for k in range(T):
        t = T - 1 - k
        _lib.call("evf_plif_bwd_wgrad2", P(gzs[t]), None, P(gv) if k else None, P(vs[t + 1]), P(vs[t]), P(zs[t]), P(xT[t]), None, P(leak), P(thresh),
                  B, H, W, 1, 0, 10.0, P(gcur[t]), None, P(gv), P(rows[:, :32]), P(rows[:, 32:]), P(slab), None, 1 | (row_ld << 8),
                  P(gpt) if k else None, P(pts[t]), P(Ps[t]), P(lpt), P(apt), P(gpt), P(gP[t]), P(rows[:, 64:]), P(rows[:, 96:]))


order = list(range(T - 1, -1, -1))
arr = lambda ts: (ctypes.c_void_p * T)(*[P(x) for x in ts])  # noqa: E731


def window():
    _lib.call("evf_plif_bwd_wgrad_window", T, arr([gzs[t] for t in order]), arr([vs[t + 1] for t in order]), arr([vs[t] for t in order]),
              arr([zs[t] for t in order]), arr([xT[t] for t in order]), arr([gcur[t] for t in order]), None, arr([pts[t] for t in order]),
              arr([Ps[t] for t in order]), arr([gP[t] for t in order]), P(leak), P(thresh), P(lpt), P(apt), B, H, W, 10.0, P(gv), P(gpt),
              P(rows[:, :32]), P(rows[:, 32:]), P(rows[:, 64:]), P(rows[:, 96:]), P(slab), 1 | (row_ld << 8))


for name, fn in (("one launch per pass", per_pass), ("window launch", window)):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-22s %.1f us per window of %d passes (%.1f per pass)" % (name, e0.elapsed_time(e1) * 100, T, e0.elapsed_time(e1) * 100 / T))
