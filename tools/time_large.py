#!/usr/bin/env python3
"""Diagnostic: the 5000-atom workloads stage by stage (features kernel alone, whole forward).
   python tools/time_large.py [C4|C5] [frames]      MOLANN_NO_RING=1 -> frames_wave_kernel instead of frames_ring_kernel"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from molann_amd import workloads as wl
from molann_amd.ann import last_launch_info
dev = torch.device("cuda:0")
w = wl.get_workload(sys.argv[1] if len(sys.argv) > 1 else "C5")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
model = wl.build_model(w, dev).requires_grad_(False)
pp = model.preprocessing_layer
xs = [w.make_frames(n, device=dev, seed=i) for i in range(2)]
def t(m):
    with torch.no_grad():
        for i in range(3): m(xs[i % 2])
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(10): m(xs[i % 2])
        b.record(); b.synchronize()
    return a.elapsed_time(b) / 10
for name, m in (("features", pp), ("forward", model)):
    ms = t(m)
    print("%s %-9s %d frames: %.3f ms  %.3g frames/s  %.1f us per 1k frames  [%s]" % (w.name, name, n, ms, n / ms * 1e3, ms * 1e3 / (n / 1000), last_launch_info(m)[:100]))
