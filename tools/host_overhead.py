#!/usr/bin/env python3
"""Diagnostic: host time of one training step (forward + backward through autograd) on a tiny batch, with a profile."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from molann_amd import workloads as wl
dev = torch.device("cuda:0")
w = wl.get_workload("C3")
model = wl.build_model(w, dev)
x0 = w.make_frames(256, device=dev, seed=1)
G = torch.randn((256, 8), device=dev)
def step(need_x):
    x = x0.detach().requires_grad_(need_x)
    model(x).backward(G)
for need_x in (True, False):
    for _ in range(20): step(need_x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): step(need_x)
    torch.cuda.synchronize()
    print("need_x=%s: %.1f us per step (host-bound batch of 256 frames)" % (need_x, (time.perf_counter() - t0) / 200 * 1e6))
import warnings
warnings.simplefilter("ignore")
scripted = torch.jit.script(model)
def step_s(need_x):
    x = x0.detach().requires_grad_(need_x)
    scripted(x).backward(G)
for need_x in (True, False):
    for _ in range(20): step_s(need_x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): step_s(need_x)
    torch.cuda.synchronize()
    print("scripted (C++ autograd node) need_x=%s: %.1f us per step" % (need_x, (time.perf_counter() - t0) / 200 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(200): step(False)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
