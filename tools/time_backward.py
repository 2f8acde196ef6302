#!/usr/bin/env python3
"""Time a training-style step (forward + backward through the HIP kernels) on a workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from molann_amd import workloads as wl
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
dev = torch.device("cuda:0")
w = wl.get_workload(name)
model = wl.build_model(w, dev)
xs = [w.make_frames(w.frames, device=dev, seed=i) for i in range(3)]
G = torch.randn((w.frames, w.n_atoms, 3) if w.kind == "align" else (w.frames, w.out_dim()), device=dev)
def step(x, need_x):
    x = x.detach().requires_grad_(need_x)
    out = model(x)
    out.backward(G)
for need_x in (False, True):
    if not need_x and not any(p.requires_grad for p in model.parameters()):
        continue
    for i in range(3): step(xs[i % 3], need_x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(10): step(xs[i % 3], need_x)
    b.record(); b.synchronize()
    print("%s forward+backward (%s): %.1f us per %d frames" % (name, "params + x grads" if need_x else "params grads only", a.elapsed_time(b) / 10 * 1e3, w.frames))
