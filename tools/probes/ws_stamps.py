"""Phase stamps of k_conv_dgrad_ws (debug build with -DWS_STAMPS, loaded through EVF_LIB): prints per block the cycle
offsets of the consumer wave 0 and the producer wave 4.   EVF_LIB=/path/libevflow_stamps.so python tools/probes/ws_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from event_flow_amd import _lib
B, H, W = 8, 128, 128
dev = "cuda:0"
P = lambda t: t.data_ptr()
w1 = torch.randn(32, 32, 3, 3, device=dev) * 0.1
wt1 = torch.empty(54 * 1024, dtype=torch.uint8, device=dev)
_lib.call("evf_pack_conv_weight_b3t", P(w1), 32, 32, P(wt1))
pair = len(sys.argv) > 1 and sys.argv[1] == "pair"
gs = [torch.randn(B, H, W, 32, device=dev) for _ in range(12)]
oa = [torch.empty(B, H, W, 32, device=dev) for _ in range(12)]
ob = [torch.empty(B, H, W, 32, device=dev) for _ in range(12)]
for k in range(12):
    if pair:
        _lib.call("evf_conv_dgrad_b3_f32_pair", P(gs[k]), P(wt1), P(oa[k]), 0, P(wt1), P(ob[k]), B, H, W, None, None)
    else:
        _lib.call("evf_conv_dgrad_b3_f32", P(gs[k]), P(wt1), P(oa[k]), 0, B, H, W, None, None)
torch.cuda.synchronize()
buf = np.zeros(16 * 2 * 64, np.uint64)
lib = _lib.load()
lib.evf_debug_ws_stamps.argtypes = [ctypes.c_void_p]
assert lib.evf_debug_ws_stamps(buf.ctypes.data) == 0
st = buf.reshape(16, 2, 64)
t0 = st[:, :, 0].min()
for b in (0, 1, 8, 15):
    for team, name in ((0, "consumer"), (1, "producer")):
        v = st[b, team]
        v = v[v > 0]
        rel = (v - st[b, :, 0].min()).astype(np.int64)
        print(f"block {b:2d} {name}: start +{int(st[b, team, 0] - t0)}  " + " ".join(str(int(x)) for x in rel))
