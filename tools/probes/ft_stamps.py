"""Phase stamps of k_fwd_diag_t (debug build with -DFT_STAMPS=<cells per launch> through EVF_LIB): runs a few bench-shaped training
steps, then prints the stamps of the LAST launch with that many cells: per block, team M (wave 0) and team E (wave 4).
Stamps per cell segment: [after staging]; team M per round: start, before the matrix phase, after it, before barrier B (tile
written; the drain round: start, before B); team E: fill round (start, before B), then per round >= 1: start, after barrier C,
after the element loop, after the word / flow part, after the bit planes, before barrier B."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from event_flow_amd import _lib

CFG = os.environ.get("FT_STAMPS_CONFIG", "c3")  # c5: the PLIF instantiation at 260 x 346
sys.argv = ["bench.py", "--config", CFG, "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-iwe", "--no-others", "--no-graph"]
try:
    bench.main()
except SystemExit:
    pass
torch.cuda.synchronize()
NW = 12
buf = np.zeros(4 * NW * 128, np.uint64)
lib = _lib.load()
lib.evf_debug_ft_stamps.argtypes = [ctypes.c_void_p]
assert lib.evf_debug_ft_stamps(buf.ctypes.data) == 0
st = buf.reshape(4, NW, 128)
t0 = min(int(x) for x in st[st > 0].ravel())
for b in (0, 1, 2, 3):  # blocks 0, 80, 160, 240
    for w in range(NW):
        v = st[b, w]
        v = v[v > 0].astype(np.int64)
        if v.size < 4:
            continue
        print(f"block {b} wave {w:2d} ({'M' if w < 4 else 'E'}): n={v.size} first {int(v[0]) - t0} total {int(v[-1] - v[0])}")
        print("      diffs:", " ".join(str(int(x)) for x in np.diff(v)[:60]))
