"""evf_grads_finalize at the c3 step's sizes, its two parts apart:  python tools/debug/finalize_bench.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from event_flow_amd import _lib  # noqa: E402

dev = "cuda:0"
L = _lib.load()
nslab, nt = 256, 8
slabs = [torch.randn(nslab, 9216, device=dev) for _ in range(nt)]
dst = [torch.zeros(32, 32, 3, 3, device=dev) for _ in range(nt)]
ncols, nrows, nh = 1024 + 66, 1024, 1024
rows = torch.randn(nrows, ncols, device=dev)
head = torch.randn(nh, 576, device=dev)
small = torch.zeros(ncols, device=dev)
# segments: 7 layers x (leak 32, thresh 32) + pred.w 64 + pred.b 2 + head weight 576
offs, ns = [], []
o = 0
for n in [576] + [32, 32] * 7 + [64, 2]:
    offs.append(o), ns.append(n)
    o += n
segd = [torch.zeros(n, device=dev) for n in ns]
P = lambda ts: (ctypes.c_void_p * max(len(ts), 1))(*[t.data_ptr() for t in ts])  # noqa: E731


def run(with_slabs, with_segs, seg_rows=None):
    n1 = nt if with_slabs else 0
    n2 = len(ns) if with_segs else 0
    sr = (ctypes.c_int * len(ns))(*seg_rows) if seg_rows else None
    def go():
        rc = L.evf_grads_finalize(P(slabs), P(dst), n1, nslab, small.data_ptr(), 1, rows.data_ptr(), nrows, ncols, head.data_ptr(), nh, 576, 0,
                                  P(segd), (ctypes.c_int * len(ns))(*offs), (ctypes.c_int * len(ns))(*ns), sr, n2, _lib.stream_ptr())
        assert rc == 0, rc
    for _ in range(5):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        go()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50 * 1e3


print("slabs only      %.1f us" % run(True, False))
print("segments only   %.1f us" % run(False, True))
print("segments, 512 rows for the hidden ones %.1f us" % run(False, True, [1024] + [512] * 16))
print("both            %.1f us" % run(True, True, [1024] + [512] * 16))
