"""Debug: captured step graphs created, replayed, destroyed and created again (different batch sizes)."""
import sys, gc
import torch
sys.path.insert(0, ".")
sys.argv = [sys.argv[0]]
exec(open("tools/debug/two_streams.py").read().split("one = build(8, 0)")[0])
for B in (4, 2, 2, 8, 2):
    x = build(B, 7)
    print("built", B, flush=True)
    print(B, timeit([x], steps=5), flush=True)
    del x
    gc.collect()
    torch.cuda.synchronize()
print("lifecycle ok")
q = [build(2, 10 + i) for i in range(4)]
print("4 built", flush=True)
print(timeit(q[:1], steps=5), flush=True)
print(timeit(q[:2], steps=5), flush=True)
print(timeit(q, steps=5), flush=True)
