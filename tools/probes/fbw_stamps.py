"""Phase stamps of k_bwd_diag_ws (debug build with -DFB_STAMPS through EVF_LIB): runs a few bench-shaped training steps, then
reads the stamps of the LAST k_bwd_diag_ws launch: per block, team E (wave 0) and team M (wave 4): cycles of work before each
barrier of the unit loop and cycles spent at it."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from event_flow_amd import _lib

sys.argv = ["bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-iwe", "--no-others", "--no-graph"]
try:
    bench.main()
except SystemExit:
    pass
torch.cuda.synchronize()
buf = np.zeros(16 * 2 * 96, np.uint64)
lib = _lib.load()
lib.evf_debug_fb_stamps.argtypes = [ctypes.c_void_p]
assert lib.evf_debug_fb_stamps(buf.ctypes.data) == 0
st = buf.reshape(16, 2, 96)
for b in (0, 1, 7, 15):
    for w, name in ((0, "E"), (1, "M")):
        v = st[b, w]
        v = v[v > 0].astype(np.int64)
        if v.size < 4:
            continue
        d = np.diff(v)
        # stamps: [pre-barrier, post-barrier] * ...: even diffs = wait at the barrier, odd diffs = work
        wait, work = d[0::2], d[1::2]
        print(f"block {b:2d} team {name}: n={v.size} total {int(v[-1] - v[0])}  work/unit median {int(np.median(work[:-1]))} "
              f"wait/unit median {int(np.median(wait))}")
        print("      work:", " ".join(str(int(x)) for x in work[:24]))
        print("      wait:", " ".join(str(int(x)) for x in wait[:24]))
