import sys, os
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import importlib
os.environ.setdefault("EVF_PARAM_ROWS", "1")
import test_gpu_network as T
g = T.load_golden("g7_liffirenet_train")
loss, grads, gn, newp = T._train_once(g, True)
print("loss", loss, float(g["loss"]), "gn", gn, float(g["grad_norm"]))
for k, got in grads.items():
    ref = g["grad_" + k]
    print(f"{k:24s} rel {np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-12):.3e}  |ref| {np.linalg.norm(ref):.3e} |got| {np.linalg.norm(got):.3e}")
