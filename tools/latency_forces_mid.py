#!/usr/bin/env python3
"""Values + forces of ONE frame of a mid-size model (P1) per MD step: eager autograd, and the plan's entry points called back to back."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from molann_amd import workloads as wl
name = sys.argv[1] if len(sys.argv) > 1 else "P1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
w = wl.get_workload(name)
dev = torch.device("cuda:0")
model = wl.build_model(w, dev)
for p in model.parameters(): p.requires_grad_(False)
x = w.make_frames(n, device=dev)
dy = torch.ones(n, w.out_dim(), device=dev)
def autograd_step():
    xx = x.detach().requires_grad_(True)
    y = model(xx)
    (gx,) = torch.autograd.grad(y, xx, dy)
    return y, gx
def timed(fn, reps=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
print("%s, %d frame(s): eager autograd values + forces %.1f us per step" % (name, n, timed(autograd_step)))
plan = model.plan_for(x)
y = torch.empty(n, w.out_dim(), device=dev); f = torch.empty(n, plan.feature_dim, device=dev)
gf = torch.empty_like(f); gx = torch.empty_like(x)
def abi_step():
    plan.forward_train(x, y, f)
    plan.mlp_backward(f, dy, gf, None)
    plan.features_backward(x, gf, gx)
print("%s, %d frame(s): forward_train + mlp_backward + features_backward (ctypes) %.1f us per step  [backward kind %d]" % (name, n, timed(abi_step), plan.backward_kind()))
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        abi_step()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        abi_step()
    print("%s, %d frame(s): the same three calls as one hipGraph replay %.1f us per step" % (name, n, timed(g.replay)))
    yy, gg = autograd_step()
    g.replay(); torch.cuda.synchronize()
    print("graph vs autograd: max |dy| %.3g max |dgx| %.3g" % (float((yy - y).abs().max()), float((gg - gx).abs().max())))
except Exception as e:
    print("capture failed:", repr(e)[:300])
from molann_amd.graph import GraphedForces
gf_ = GraphedForces(model, x)
def gstep():
    gf_(x); gf_.vjp(dy)
print("%s, %d frame(s): GraphedForces (copy + forward replay, copy + backward replay) %.1f us per step" % (name, n, timed(gstep)))
def gstep2():
    gf_.graph.replay(); gf_.bwd_graph.replay()
print("%s, %d frame(s): its two replays alone %.1f us per step" % (name, n, timed(gstep2)))
print("%s, %d frame(s): forward replay alone %.1f us, backward replay alone %.1f us" % (name, n, timed(gf_.graph.replay), timed(gf_.bwd_graph.replay)))
