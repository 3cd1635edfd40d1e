"""Phase stamps of k_dgrad_diag_ws (build with -DWD_STAMPS through event_flow_amd.build.build_variant, loaded through EVF_LIB):
per block the cycle offsets of consumer wave 0 and producer wave 4, and the per-item deltas.
    EVF_LIB=event_flow_amd/libevflow_wdstamps.so python tools/probes/wd_stamps.py [ncells] [npair]"""
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


cells = [(torch.randn(B, H, W, 32, device=dev), pack(), pack() if k < npair else None, torch.empty(B, H, W, 32, device=dev),
          torch.empty(B, H, W, 32, device=dev)) for k in range(ncell)]
for rep in range(4):
    assert _lib.raw("evf_bwd_defer_begin") == 0
    assert _lib.raw("evf_bwd_defer_slot", 0) == 0
    for g, w1, w2, a, b in cells:
        if w2 is None:
            _lib.call("evf_conv_dgrad_b3_f32", P(g), P(w1), P(a), 0, B, H, W, None, None)
        else:
            _lib.call("evf_conv_dgrad_b3_f32_pair", P(g), P(w1), P(a), 0, P(w2), P(b), B, H, W, None, None)
    _lib.call("evf_bwd_defer_flush")
torch.cuda.synchronize()
buf = np.zeros(16 * 2 * 128, np.uint64)
L.evf_debug_wd_stamps.argtypes = [ctypes.c_void_p]
assert L.evf_debug_wd_stamps(buf.ctypes.data) == 0
st = buf.reshape(16, 2, 128)
for b in (0, 1, 8, 15):
    for team, name in ((0, "consumer"), (1, "producer")):
        v = st[b, team]
        v = v[v > 0]
        rel = (v - st[b, :, 0].min()).astype(np.int64)
        d = np.diff(rel)
        print(f"block {b:2d} {name}: n={len(rel)} total {int(rel[-1])}  first 10: " + " ".join(str(int(x)) for x in rel[:10]))
        if team == 0 and len(rel) > 12:  # consumer: [start, wait-done, (item start, matrix done, item end) ...]
            it = rel[2:2 + 3 * ((len(rel) - 3) // 3)].reshape(-1, 3)
            print("      matrix phase per item:", " ".join(str(int(x)) for x in (it[:, 1] - it[:, 0])[:26]))
            print("      epilogue per item:    ", " ".join(str(int(x)) for x in (it[:, 2] - it[:, 1])[:26]))
            print("      barrier wait to next: ", " ".join(str(int(x)) for x in (it[1:, 0] - it[:-1, 2])[:26]))
        if team == 1 and len(rel) > 12:  # producer: [start, prologue done, (step start, fetch issued, split done) ...]
            it = rel[2:2 + 3 * ((len(rel) - 3) // 3)].reshape(-1, 3)
            print("      fetch issue per item: ", " ".join(str(int(x)) for x in (it[:, 1] - it[:, 0])[:26]))
            print("      split+store per item: ", " ".join(str(int(x)) for x in (it[:, 2] - it[:, 1])[:26]))
            print("      barrier wait to next: ", " ".join(str(int(x)) for x in (it[1:, 0] - it[:-1, 2])[:26]))
