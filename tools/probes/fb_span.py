"""Per-block start / end (100 MHz chip-wide counter) of k_lif_bwd_wgrad -- how much of the kernel is tail?
   EVF_BWD=fused EVF_LIB=.../libevflow_fbspan.so python tools/probes/fb_span.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from event_flow_amd import _lib
B, H, W, C = 8, 128, 128, 32
dev = "cuda:0"
P = lambda t: t.data_ptr()
f = lambda *s: torch.randn(*s, device=dev)
leak, thresh = f(32) * 0.1 - 4, f(32) * 0.1 + 0.8
nsl = _lib.load().evf_lif_bwd_wgrad_slabs(B, H, W)
slab = torch.zeros(nsl, 9216, device=dev)
gl, gt = torch.zeros(32, device=dev), torch.zeros(32, device=dev)
sets = []
for _ in range(6):
    sets.append(dict(g1=f(B, H, W, C), g2=f(B, H, W, C), vo=f(B, H, W, C), v=f(B, H, W, C), g3=f(B, H, W, C), g4=f(B, H, W, C),
                     z=torch.randint(-2**31, 2**31 - 1, (B, H, W), dtype=torch.int32, device=dev),
                     xT=torch.randint(-2**31, 2**31 - 1, (B, H, 32, (W + 31) // 32), dtype=torch.int32, device=dev)))
lib = _lib.load()
lib.evf_debug_fb_span.argtypes = [ctypes.c_void_p]
for trial in range(3):
    for k in range(8):
        d = sets[k % 6]
        _lib.call("evf_lif_bwd_wgrad", P(d["g1"]), P(d["g2"]), P(d["vo"]), P(d["v"]), P(d["z"]), P(d["xT"]), None, P(leak),
                  P(thresh), B, H, W, 1, 0, 10.0, P(d["g3"]), None, P(d["g4"]), P(gl), P(gt), P(slab), None, 1)
    torch.cuda.synchronize()
    buf = np.zeros(2 * 1024, np.uint64)
    assert lib.evf_debug_fb_span(buf.ctypes.data) == 0
    sp = buf.reshape(1024, 2)[:nsl].astype(np.int64)
    t0 = sp[:, 0].min()
    st, en = (sp[:, 0] - t0) / 100.0, (sp[:, 1] - t0) / 100.0  # us
    dur = en - st
    print(f"blocks {nsl}: start spread {st.max():.2f} us; block life min/median/max {dur.min():.2f}/{np.median(dur):.2f}/{dur.max():.2f} us; "
          f"first end {en.min():.2f}, median end {np.median(en):.2f}, last end {en.max():.2f} us")
    xcd = np.arange(nsl) % 8
    print("   median block life per XCD:", " ".join(f"{np.median(dur[xcd == x]):.1f}" for x in range(8)),
          "| last end per XCD:", " ".join(f"{en[xcd == x].max():.1f}" for x in range(8)))
