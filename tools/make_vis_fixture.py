#!/opt/conda/bin/python3.9
"""Golden vectors for the colour coding of the stored visualisations: the reference's OWN static functions
Visualization.flow_to_image / minmax_norm / events_to_image (utils/visualization.py:230-315) run on seeded inputs.

Needs matplotlib (the reference's hsv_to_rgb) -- in this image only /opt/conda/bin/python3.9 has it:
    /opt/conda/bin/python3.9 tools/make_vis_fixture.py
The reference module also imports cv2 at its top, which no interpreter here has; the three functions pinned here never touch it
(it serves the live window and imwrite), so an EMPTY placeholder module is registered under that name for the import only --
nothing of it is executed, and nothing that would need it (Visualization.update / .store) is part of this fixture."""
import os
import sys
import types

import numpy as np

sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
from utils.visualization import Visualization as R  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
rng = np.random.Generator(np.random.PCG64(2024))
a = {}
H, W = 24, 30
for k in range(4):
    fx, fy = rng.standard_normal((H, W)) * (k + 0.5), rng.standard_normal((H, W)) * 2.0
    if k == 2:  # axis-aligned and zero vectors: hue sector boundaries, zero magnitude
        fx[:, :10], fy[:, :10] = 0.0, np.abs(fy[:, :10])
        fx[:5, 10:20], fy[:5, 10:20] = 0.0, 0.0
        fy[5:, 10:20] = 0.0
    if k == 3:  # constant flow: zero magnitude range
        fx[:], fy[:] = 1.5, -0.5
    a[f"flow{k}_x"], a[f"flow{k}_y"], a[f"flow{k}_rgb"] = fx, fy, R.flow_to_image(fx, fy)
for k in range(3):
    cnt = rng.poisson(0.6 * (k + 1), size=(H, W, 2)).astype(np.float64)
    if k == 2:
        cnt[:, :, 1] = 0  # one polarity only
    a[f"cnt{k}"] = cnt
    a[f"cnt{k}_green_red"] = R.events_to_image(cnt.copy(), "green_red")
    a[f"cnt{k}_gray"] = R.events_to_image(cnt.copy(), "gray")
    x = rng.standard_normal((H, W, 1)) * 3 + k
    a[f"mm{k}_in"], a[f"mm{k}_out"] = x, R.minmax_norm(x.copy())
a["mm_const_in"] = np.full((H, W, 1), 2.0)
a["mm_const_out"] = R.minmax_norm(a["mm_const_in"].copy())
np.savez_compressed(os.path.join(OUT, "g17_visualization.npz"), **a)
print("g17_visualization:", len(a), "arrays")
