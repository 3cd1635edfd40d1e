"""Phase stamps of k_fwd_diag_p (build with -DFP_STAMPS, loaded through EVF_LIB): per block, for wave 0 and wave 4, the cycles
of every strip's parts.   EVF_LIB=event_flow_amd/libevflow_fpstamps.so python tools/probes/fp_stamps.py [ncells] [nrec]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from event_flow_amd import _lib
B, H, W = 8, 128, 128
dev = "cuda:0"
P = lambda t: t.data_ptr()
ncell = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nrec = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L = _lib.load()


def pack():
    w = torch.randn(32, 32, 3, 3, device=dev) * 0.1
    t = torch.empty(54 * 1024, dtype=torch.uint8, device=dev)
    _lib.call("evf_pack_conv_weight_b3", P(w), 32, 32, P(t))
    return t


def bits():
    return torch.randint(-2**31, 2**31 - 1, (B, H, W), dtype=torch.int32, device=dev) & torch.randint(-2**31, 2**31 - 1, (B, H, W), dtype=torch.int32, device=dev)


leak, thresh = torch.randn(32, device=dev) * 0.1 - 4, torch.randn(32, device=dev) * 0.1 + 0.8
cells = []
for k in range(ncell):
    cells.append(dict(x=bits(), wff=pack(), wrec=pack() if k < nrec else None, v=torch.randn(B, H, W, 32, device=dev), z=bits(),
                      vo=torch.empty(B, H, W, 32, device=dev), zo=torch.empty(B, H, W, dtype=torch.int32, device=dev),
                      zT=torch.empty(B, H, 32, (W + 31) // 32, dtype=torch.int32, device=dev)))
for rep in range(4):
    assert _lib.raw("evf_fwd_defer_begin") == 0
    assert _lib.raw("evf_fwd_defer_slot", 0) == 0
    for c in cells:
        _lib.call("evf_conv_lif_fwd_b3", P(c["x"]), P(c["wff"]), P(c["wrec"]) if c["wrec"] is not None else None, P(leak), P(thresh), P(c["v"]),
                  P(c["z"]), B, H, W, 1, P(c["vo"]), P(c["zo"]), P(c["zT"]))
    _lib.call("evf_fwd_defer_flush")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for mode in ("persistent",):
    ts = []
    for rep in range(6):
        assert _lib.raw("evf_fwd_defer_begin") == 0
        assert _lib.raw("evf_fwd_defer_slot", 0) == 0
        for c in cells:
            _lib.call("evf_conv_lif_fwd_b3", P(c["x"]), P(c["wff"]), P(c["wrec"]) if c["wrec"] is not None else None, P(leak), P(thresh), P(c["v"]),
                      P(c["z"]), B, H, W, 1, P(c["vo"]), P(c["zo"]), P(c["zT"]))
        e0.record()
        _lib.call("evf_fwd_defer_flush")
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"flush of {ncell} cells ({nrec} recurrent) [{os.environ.get('EVF_FWD_DIAG', 'persistent')}]: " + " ".join(f"{t:.1f}" for t in ts) + " us")
buf = np.zeros(16 * 2 * 128, np.uint64)
L.evf_debug_fp_stamps.argtypes = [ctypes.c_void_p]
assert L.evf_debug_fp_stamps(buf.ctypes.data) == 0
st = buf.reshape(16, 2, 128)
for b in (0, 1, 8, 15):
    t0 = st[b, :, 0].min()
    for team in (0, 1):
        v = st[b, team]
        v = v[v > 0]
        rel = (v - t0).astype(np.int64)
        print(f"block {b:2d} wave {4 * team}: n={len(rel)} total {int(rel[-1])}; all stamps: " + " ".join(str(int(x)) for x in np.diff(rel)[:60]))
