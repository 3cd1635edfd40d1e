"""Phase stamps of k_dgrad_diag_dma (build with -DWD_STAMPS, loaded through EVF_LIB): per block and wave the cycles of every
item's matrix phase (incl. the DMA pieces / previous epilogue riding behind its MFMAs), the wait for the own DMA pieces and
the barrier.    EVF_LIB=event_flow_amd/libevflow_wmstamps.so python tools/probes/wm_stamps.py [ncells] [npair]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from event_flow_amd import _lib
B, H, W = 8, 128, 128
dev = "cuda:0"
P = lambda t: t.data_ptr()
ncell = int(sys.argv[1]) if len(sys.argv) > 1 else 4
npair = int(sys.argv[2]) if len(sys.argv) > 2 else 2
L = _lib.load()


def pack():
    w = torch.randn(32, 32, 3, 3, device=dev) * 0.1
    t = torch.empty(54 * 1024, dtype=torch.uint8, device=dev)
    _lib.call("evf_pack_conv_weight_b3t", P(w), 32, 32, P(t))
    return t


def planes():
    g = torch.randn(B, H, W, 32, device=dev)
    hi = g.to(torch.bfloat16)
    r = g - hi.float()
    mid = r.to(torch.bfloat16)
    lo = (r - mid.float()).to(torch.bfloat16)
    return torch.stack([hi, mid, lo]).contiguous()


cells = [(planes(), pack(), pack() if k < npair else None, torch.empty(B, H, W, 32, device=dev), torch.empty(B, H, W, 32, device=dev))
         for k in range(ncell)]
for rep in range(4):
    assert _lib.raw("evf_bwd_defer_begin") == 0
    assert _lib.raw("evf_bwd_defer_slot", 0) == 0
    for g, w1, w2, a, b in cells:
        if w2 is None:
            _lib.call("evf_conv_dgrad_b3", P(g), P(w1), P(a), 0, B, H, W, None, None)
        else:
            _lib.call("evf_conv_dgrad_b3_pair", P(g), P(w1), P(a), 0, P(w2), P(b), B, H, W, None, None)
    _lib.call("evf_bwd_defer_flush")
torch.cuda.synchronize()
buf = np.zeros(16 * 4 * 128, np.uint64)
L.evf_debug_wm_stamps.argtypes = [ctypes.c_void_p]
assert L.evf_debug_wm_stamps(buf.ctypes.data) == 0
st = buf.reshape(16, 4, 128)
for b in (0, 1, 8, 15):
    t0 = st[b, :2, 0].min()
    for team, name in ((0, "matrix wave 0"), (1, "loader wave 4")):
        v = st[b, team]
        v = v[v > 0]
        rel = (v - t0).astype(np.int64)
        print(f"block {b:2d} {name}: n={len(rel)} prologue {int(rel[1])} total {int(rel[-1])}")
        if team == 0:  # [start, before first barrier, (item start, matrix done, before barrier) x n, end]
            it = rel[2:2 + 3 * ((len(rel) - 3) // 3)].reshape(-1, 3)
            print("      matrix phase per item:", " ".join(str(int(x)) for x in (it[:, 1] - it[:, 0])[:26]))
            print("      barrier + item setup: ", " ".join(str(int(x)) for x in (it[1:, 0] - it[:-1, 2])[:26]))
        else:  # [start, prologue done, (item start, pieces issued, pieces landed) x n, end]
            it = rel[2:2 + 3 * ((len(rel) - 3) // 3)].reshape(-1, 3)
            print("      issue 10 pieces:      ", " ".join(str(int(x)) for x in (it[:, 1] - it[:, 0])[:26]))
            print("      wait until landed:    ", " ".join(str(int(x)) for x in (it[:, 2] - it[:, 1])[:26]))
            print("      barrier wait:         ", " ".join(str(int(x)) for x in (it[1:, 0] - it[:-1, 2])[:26]))
