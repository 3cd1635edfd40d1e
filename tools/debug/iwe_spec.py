import sys, json
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from event_flow_amd import _lib
dev = torch.device("cuda:0")
r = bench.iwe_warp_bandwidth(dev, 8)
print(json.dumps(r)[:600])
