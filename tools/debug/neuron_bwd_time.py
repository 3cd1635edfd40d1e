import sys
import torch
sys.path.insert(0, ".")
from event_flow_amd import _lib
dev = torch.device("cuda:0")
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (npix, C) in ((8 * 128 * 128, 64), (8 * 256 * 256, 32), (8 * 16 * 16, 512)):
    mk = lambda: torch.randn(npix, C, device=dev)
    gz, vo, v, z, gc, gp = mk(), mk(), mk(), (mk() > 1).float(), mk(), mk()
    p0, p1 = torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5
    g0, g1 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    def call(withg, prev=True):
        _lib.call("evf_neuron_bwd", 0, None, _lib.ptr(gz), None, None, _lib.ptr(vo), None, _lib.ptr(v) if prev else None, _lib.ptr(z) if prev else None, None,
                  None, _lib.ptr(p0), _lib.ptr(p1), None, None, npix, C, 1, 0, 10.0, _lib.ptr(gc), None, None, None, None,
                  _lib.ptr(g0) if withg else None, _lib.ptr(g1) if withg else None, None, None, _lib.ptr(ws) if withg == 2 else None)
    ws = torch.zeros(32 * 4096 + 64, device=dev)
    call(1); a = g0.clone(); g0.zero_(); g1.zero_(); call(2); torch.cuda.synchronize()
    print(npix, C, "param grads by direct atomics %.1f us, through replicas %.1f us, none %.1f us; bytes %.0f MB; replicas vs direct rel diff %.1e, scratch left %g" % (
        t(lambda: call(1)), t(lambda: call(2)), t(lambda: call(0)), npix * C * 4 * 5 / 1e6, float((g0 - a).abs().max() / a.abs().max()), float(ws.abs().sum())))
