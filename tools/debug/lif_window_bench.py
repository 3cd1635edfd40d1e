"""k_bwd_win_lif[_top] (LIF feed-forward cell, the passes of a window in one launch; split planes out) at the config-3 shape:
  python tools/debug/lif_window_bench.py [B H W passes]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from event_flow_amd import _lib  # noqa: E402

B, H, W, T = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (8, 128, 128, 10)
dev, C = "cuda:0", 32
P = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
f = lambda *s: torch.randn(*s, device=dev) * 0.3  # noqa: E731
L = _lib.load()
nsl = max(L.evf_lif_bwd_wgrad_slabs(B, H, W), 512)
row_ld = 224
leak, thresh = f(32), f(32) + 0.5
vs = [f(B, H, W, C) for _ in range(T + 1)]
zs = [torch.randint(0, 2 ** 31 - 1, (B, H, W), dtype=torch.int32, device=dev) for _ in range(T)]
nW = (W + 31) // 32
xT = [torch.randint(0, 2 ** 31 - 1, (B, H, 32, nW), dtype=torch.int32, device=dev) for _ in range(T)]
gzs = [f(B, H, W, C) for _ in range(T)]
flows = [torch.tanh(f(B, 2, H, W)) for _ in range(T)]
gfl = [f(B, 2, H, W) for _ in range(T)]
pw = f(2, 32)
gsp = [torch.empty(3, B, H, W, C, dtype=torch.bfloat16, device=dev) for _ in range(T)]
gv = torch.empty(B, H, W, C, device=dev)
rows, slab = torch.zeros(nsl, row_ld, device=dev), torch.zeros(nsl, 9216, device=dev)
order = list(range(T - 1, -1, -1))
arr = lambda ts: (ctypes.c_void_p * T)(*[P(x) for x in ts])  # noqa: E731


def window(top):
    _lib.call("evf_lif_bwd_wgrad_window", T, None if top else arr([gzs[t] for t in order]), arr([flows[t] for t in order]) if top else None,
              arr([gfl[t] for t in order]) if top else None, P(pw) if top else None, arr([zs[t] for t in order]) if top else None,
              P(rows[:, 128:]) if top else None, P(rows[:, 192:]) if top else None, arr([vs[t + 1] for t in order]), arr([vs[t] for t in order]),
              arr([zs[t] for t in order]), arr([xT[t] for t in order]), None, arr([gsp[t] for t in order]), P(leak), P(thresh), B, H, W, 10.0,
              None, P(rows[:, :32]), P(rows[:, 32:]), P(slab), 1 | (row_ld << 8))


for name, fn in (("hidden cell, window", lambda: window(False)), ("top cell, window", lambda: window(True))):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-22s %.1f us per window of %d passes (%.1f per pass)" % (name, e0.elapsed_time(e1) * 50, T, e0.elapsed_time(e1) * 50 / T))
