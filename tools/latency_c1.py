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
