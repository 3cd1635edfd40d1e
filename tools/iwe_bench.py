"""compute_pol_iwe at a saturating shape: python tools/iwe_bench.py [B]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_flow_amd import _lib, synthetic
from event_flow_amd.utils.iwe import compute_pol_iwe
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
H = W = 128; n = 15000; dev = "cuda:0"
g = np.random.default_rng(1)
ev_small = synthetic.event_list_batch(8, n, H, W, 4242)
ev = torch.from_numpy(np.concatenate([ev_small] * (B // 8), 0)).to(dev)
flow = torch.from_numpy(g.uniform(-0.1, 0.1, size=(B, 2, H, W)).astype(np.float32)).to(dev)
pol = torch.stack([(ev[:, :, 3] > 0).float(), (ev[:, :, 3] < 0).float()], 2).contiguous()
for _ in range(3):
    compute_pol_iwe(flow, ev, (H, W), pol[:, :, 0:1], pol[:, :, 1:2], flow_scaling=128, round_idx=True)
torch.cuda.synchronize()
_lib.profile_start(["evf_iwe_splat"])
for _ in range(20):
    compute_pol_iwe(flow, ev, (H, W), pol[:, :, 0:1], pol[:, :, 1:2], flow_scaling=128, round_idx=True)
t = _lib.profile_stop()[("evf_iwe_splat", "")]
ms = float(np.median(t)); alg = B * (n * 28 + 2 * H * W * 4)
print(f"B={B} rows={os.environ.get('EVF_IWE_ROWS','auto')} {ms*1e3:.1f} us  {alg/ms/1e6:.0f} GB/s  {alg/ms/1e6/8000:.3f} of HBM peak")
