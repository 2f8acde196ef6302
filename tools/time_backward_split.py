#!/usr/bin/env python3
"""Diagnostic: forward + backward of C3 as a whole and of its preprocessing alone (Kabsch + features, no MLP), per 1 M frames.
   FRAMES=4194304 python tools/time_backward_split.py   (more frames per step: less of the host's per-step time in the figure)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from molann_amd import workloads as wl
dev = torch.device("cuda:0")
w = wl.get_workload("C3")
model = wl.build_model(w, dev)
pp = model.preprocessing_layer
N = int(os.environ.get("FRAMES", w.frames))
xs = [w.make_frames(N, device=dev, seed=i) for i in range(3)]
def run(m, cols, need_x, fwd_only=False):
    G = torch.randn((N, cols), device=dev)
    def step(x):
        x = x.detach().requires_grad_(need_x)
        out = m(x)
        if not fwd_only: out.backward(G)
    for i in range(3): step(xs[i % 3])
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(10): step(xs[i % 3])
    b.record(); b.synchronize()
    return a.elapsed_time(b) / 10 * 1e3 * (1048576.0 / N)
print("C3 model  fwd+bwd, params + x grads: %.1f us" % run(model, 8, True))
print("C3 model  fwd+bwd, params only     : %.1f us" % run(model, 8, False))
print("C3 preprocessing fwd+bwd, x grads  : %.1f us" % run(pp, 6, True))
print("C3 preprocessing fwd only (grad mode): %.1f us" % run(pp, 6, True, fwd_only=True))
