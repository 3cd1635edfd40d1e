"""Which torch (aten) kernels run inside one bench-shaped training step, and from where: torch.profiler over eager steps,
grouped by op name with the Python stack of the first occurrence."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity

sys.argv = ["bench.py", "--steps", "2", "--warmup", "2", "--no-cpu-baseline", "--no-iwe", "--no-others", "--no-graph"]
orig = bench.run_step
state = {"n": 0}


def wrapped(*a, **k):
    state["n"] += 1
    if state["n"] == 4:  # (after the warm-up steps)
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
            out = orig(*a, **k)
            torch.cuda.synchronize()
        rows = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
        import collections
        cnt = collections.Counter(e.name[:80] for e in rows)
        print("== device kernels of one step (name, count)")
        for n, c in cnt.most_common(60):
            print(f"   {c:4d}  {n}")
        print("== aten ops launching kernels (op, count, first stack)")
        seen = collections.OrderedDict()
        for e in prof.events():
            if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith("aten::"):
                continue
            if not any(k.device_type == torch.autograd.DeviceType.CUDA for k in getattr(e, "kernels", [])) and not getattr(e, "kernels", None):
                continue
            key = e.name
            st = [s for s in (e.stack or []) if "event_flow_amd" in s or "bench.py" in s][:3]
            seen.setdefault((key, tuple(st)), 0)
            seen[(key, tuple(st))] += 1
        for (k, st), c in sorted(seen.items(), key=lambda kv: -kv[1])[:40]:
            print(f"   {c:4d}  {k}   <- {' | '.join(s.strip()[-90:] for s in st)}")
        return out
    return orig(*a, **k)


bench.run_step = wrapped
try:
    bench.main()
except SystemExit:
    pass
