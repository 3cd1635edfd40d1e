"""Per-block start / end (100 MHz chip-wide counter) of the LAST forward and input-gradient launch of an eager train step.
   EVF_LIB=tools/probes/bin/libevflow_span.so python tools/probes/span_step.py      (library built with -DEVF_SPAN)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from event_flow_amd import _lib
from event_flow_amd.loss.flow import EventWarping
from event_flow_amd.models import model as models
from event_flow_amd.parallel import DataParallel
from event_flow_amd.train import FlatAdam

dev = "cuda:0"
dp = DataParallel(device=dev)
torch.manual_seed(0)
model = models.LIFFireNet(dict(bench.MODEL_CFG)).to(dev)
model.train()
lossf = EventWarping(bench.LOSS_CFG, dev)
opt = FlatAdam(model, lr=2e-4, clip=100.0)
opt.zero_grad()
pool = bench.make_windows(0, 2, dev)
lib = _lib.load()
for name in ("evf_debug_dg_span", "evf_debug_fw_span"):
    getattr(lib, name).argtypes = [ctypes.c_void_p]
for i in range(4):
    bench.run_step(model, lossf, opt, dp, pool[i % 2])
    torch.cuda.synchronize()
for name, nblk in (("evf_debug_fw_span", 512), ("evf_debug_dg_span", 256)):
    buf = np.zeros(2 * 4096, np.uint64)
    assert getattr(lib, name)(buf.ctypes.data) == 0
    sp = buf.reshape(4096, 2).astype(np.int64)
    sp = sp[(sp[:, 0] > 0) & (sp[:, 1] > 0)]
    # keep the blocks of the last launch (the array holds one entry per block id; all ids are rewritten by every launch)
    t0 = sp[:, 0].min()
    st, en = (sp[:, 0] - t0) / 100.0, (sp[:, 1] - t0) / 100.0
    dur = en - st
    print(f"{name}: {len(sp)} blocks; start spread {st.max():.2f} us; block life min/median/max {dur.min():.2f}/{np.median(dur):.2f}/{dur.max():.2f} us; "
          f"first end {en.min():.2f}, median end {np.median(en):.2f}, last end {en.max():.2f} us (= span of all blocks)")
