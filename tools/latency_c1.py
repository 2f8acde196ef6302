#!/usr/bin/env python3
"""C1 (1024-frame batches): per-call latency eager vs HIP-graph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from molann_amd import workloads as wl
from molann_amd.graph import GraphedForward
dev = torch.device("cuda:0")
w = wl.get_workload("C1")
model = wl.build_model(w, dev).requires_grad_(False)
x = w.make_frames(1024, device=dev)
g = GraphedForward(model, x)
def timeit(fn, n=2000):
    with torch.no_grad():
        for _ in range(50): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6
print("eager  : %.2f us per 1024-frame forward" % timeit(lambda: model(x)))
print("graph  : %.2f us per 1024-frame forward (replay incl. input copy)" % timeit(lambda: g(x)))
print("replay : %.2f us (graph.replay only)" % timeit(lambda: g.graph.replay()))
scripted = torch.jit.script(model)
print("script : %.2f us per 1024-frame forward (TorchScript module -> molann::run)" % timeit(lambda: scripted(x)))
x1 = x[:1].contiguous()
print("eager  : %.2f us per 1-frame forward (the MD-engine call pattern)" % timeit(lambda: model(x1)))
print("script : %.2f us per 1-frame forward" % timeit(lambda: scripted(x1)))
xg = x1.clone().requires_grad_(True)
def force():
    y = scripted(xg)
    return torch.autograd.grad(y.sum(), [xg])[0]
def timeit_grad(fn, n=1000):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6
print("script : %.2f us per 1-frame forward + forces (autograd through molann::run_backward)" % timeit_grad(force))
