"""Phase stamps of k_lif_bwd_wgrad (debug build with -DFB_STAMPS through EVF_LIB).  Per loop iteration: start | loads issued |
MFMAs done | commit done (then the barrier)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from event_flow_amd import _lib
B, H, W, C = 8, 128, 128, 32
dev = "cuda:0"
P = lambda t: t.data_ptr()
f = lambda *s: torch.randn(*s, device=dev)
rec = len(sys.argv) > 1 and sys.argv[1] == "rec"
leak, thresh = f(32) * 0.1 - 4, f(32) * 0.1 + 0.8
nsl = _lib.load().evf_lif_bwd_wgrad_slabs(B, H, W)
slab, slab2 = torch.zeros(nsl, 9216, device=dev), torch.zeros(nsl, 9216, device=dev)
gl, gt = torch.zeros(32, device=dev), torch.zeros(32, device=dev)
sets = []
for _ in range(6):
    sets.append(dict(g1=f(B, H, W, C), g2=f(B, H, W, C), vo=f(B, H, W, C), v=f(B, H, W, C), g3=f(B, H, W, C), g4=f(B, H, W, C),
                     z=torch.randint(-2**31, 2**31 - 1, (B, H, W), dtype=torch.int32, device=dev),
                     xT=torch.randint(-2**31, 2**31 - 1, (B, H, 32, (W + 31) // 32), dtype=torch.int32, device=dev)))
for k in range(12):
    d = sets[k % 6]
    _lib.call("evf_lif_bwd_wgrad", P(d["g1"]), P(d["g2"]), P(d["vo"]), P(d["v"]), P(d["z"]), P(d["xT"]), P(d["xT"]) if rec else None, P(leak),
              P(thresh), B, H, W, 1, 0, 10.0, P(d["g3"]), None, P(d["g4"]), P(gl), P(gt), P(slab), P(slab2) if rec else None, 1)
torch.cuda.synchronize()
buf = np.zeros(16 * 2 * 96, np.uint64)
lib = _lib.load()
lib.evf_debug_fb_stamps.argtypes = [ctypes.c_void_p]
assert lib.evf_debug_fb_stamps(buf.ctypes.data) == 0
st = buf.reshape(16, 2, 96)
for b in (0, 5, 15):
    for w, name in ((0, "waveA"), (1, "waveB")):
        v = st[b, w]
        v = v[v > 0].astype(np.int64)
        rel = v - v[0]
        print(f"block {b:2d} {name}: " + " ".join(str(int(x)) for x in np.diff(np.concatenate([[0], rel]))), " total", int(rel[-1]))
