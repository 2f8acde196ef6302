#!/usr/bin/env python3
"""Mean duration of one kernel of a workload's forward, timed with events around the preprocessing layer alone (no MLP).
   python tools/prof_one.py P1 [label]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from molann_amd import workloads as wl
from molann_amd.ann import last_launch_info
w = wl.get_workload(sys.argv[1])
dev = torch.device("cuda:0")
model = wl.build_model(w, dev)
pp = getattr(model, "preprocessing_layer", model)
n = int(os.environ.get("FRAMES", 1 << 19))
x = w.make_frames(n, device=dev)
with torch.no_grad():
    for _ in range(3): pp(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): pp(x)
    b.record(); b.synchronize()
print("%.1f us per %d frames  %s" % (a.elapsed_time(b) * 100, n, last_launch_info(pp)[:150]))
