#!/usr/bin/env python3
"""Diagnostic: the two halves of the backward on their own (C ABI through ctypes), C3 shapes.
   python tools/time_mlp_bwd.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from molann_amd import workloads as wl
dev = torch.device("cuda:0")
w = wl.get_workload(os.environ.get("WL", "C3"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else w.frames
model = wl.build_model(w, dev)
x = w.make_frames(n, device=dev, seed=1)
x.requires_grad_(True)
out = model(x)                      # builds the plan, packs the weights
out.sum().backward()
plan = model.plan_for(x)
f = torch.empty((n, plan.feature_dim), device=dev)
plan.features(x.detach(), f)
g = torch.randn((n, plan.out_dim), device=dev)
gf = torch.empty_like(f)
gp = torch.zeros(plan.grad_params_size(), device=dev)
gx = torch.empty_like(x.detach())
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps * 1e3
s = 1e6 / n
print("per 1M frames (n = %d)" % n)
print("features (twin of the fused kernel) : %7.1f us   %s" % (timeit(lambda: plan.features(x.detach(), f)) * s, plan.last_launch_info()[:60]))
o2 = torch.empty((n, plan.out_dim), device=dev)
print("forward_packed                      : %7.1f us" % (timeit(lambda: plan.forward_packed(x.detach(), o2)) * s))
print("forward_train (keeps the features)  : %7.1f us" % (timeit(lambda: plan.forward_train(x.detach(), o2, f)) * s))
print("mlp_packed                          : %7.1f us" % (timeit(lambda: plan.mlp_packed(f, g.new_empty((n, plan.out_dim)))) * s))
print("mlp_backward  grad_f + grad_params  : %7.1f us" % (timeit(lambda: plan.mlp_backward(f, g, gf, gp)) * s))
print("mlp_backward  grad_params only      : %7.1f us" % (timeit(lambda: plan.mlp_backward(f, g, None, gp)) * s))
print("mlp_backward  grad_f only           : %7.1f us" % (timeit(lambda: plan.mlp_backward(f, g, gf, None)) * s))
print("features_backward                   : %7.1f us" % (timeit(lambda: plan.features_backward(x.detach(), gf, gx)) * s))
print("backward (all three, recompute)     : %7.1f us" % (timeit(lambda: plan.backward(x.detach(), g, gx, gp)) * s))
