#!/usr/bin/env python3
"""Diagnostic: molann_backward_f32 alone (C ABI), per 1 M frames.   [WL=C3] [DIMS=6,32,32,8] [MOLANN_NO_RING_BWD=1] python tools/time_bwd_only.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from molann_amd import workloads as wl
dev = torch.device("cuda:0")
w = wl.get_workload(os.environ.get("WL", "C3"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else w.frames
model = wl.build_model(w, dev)
if os.environ.get("DIMS"):          # another MLP behind the workload's features: DIMS=6,32,32,8
    from molann_amd.ann import MolANN, create_sequential_nn
    dims = [int(v) for v in os.environ["DIMS"].split(",")]
    model = MolANN(model.preprocessing_layer, create_sequential_nn(dims).to(dev))
    w.mlp_dims = dims
x = w.make_frames(n, device=dev, seed=1).requires_grad_(True)
model(x).sum().backward()
def find_plan(m):
    if hasattr(m, "_fast_state"):
        p = m.plan_for(x)
        if p is not None and p.supports_backward():
            return p
    for mod in m.modules():
        for e in getattr(mod, "_plans", lambda: {})().values():
            if hasattr(e, "plan") and e.plan.supports_backward():
                return e.plan
plan = find_plan(model)
cols = plan.out_dim if plan.grad_params_size() > 0 else plan.feature_dim
g = torch.randn((n, cols), device=dev)
xd = x.detach()
gx = torch.empty_like(xd)
gp = torch.zeros(max(1, plan.grad_params_size()), device=dev)
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps * 1e3 * (1048576.0 / n)
has_p = plan.grad_params_size() > 0
print("backward x + params : %7.1f us   %s" % (timeit(lambda: plan.backward(xd, g, gx, gp if has_p else None)), plan.last_launch_info()[:110]))
print("backward x only     : %7.1f us" % timeit(lambda: plan.backward(xd, g, gx, None)))
if has_p:
    print("backward params only: %7.1f us" % timeit(lambda: plan.backward(xd, g, None, gp)))
if has_p:
    o = torch.empty((n, cols), device=dev)
    print("forward_packed      : %7.1f us   %s" % (timeit(lambda: plan.forward_packed(xd, o)), plan.last_launch_info()[:100]))
