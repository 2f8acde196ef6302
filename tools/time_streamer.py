#!/usr/bin/env python3
"""PCIe-inclusive throughput: frames start (and results end) in host memory."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from molann_amd import workloads as wl
from molann_amd.stream import stream_forward
dev = torch.device("cuda:0")
w = wl.get_workload(sys.argv[1] if len(sys.argv) > 1 else "C3")
model = wl.build_model(w, dev).requires_grad_(False)
n = 1 << 22
x = w.make_frames(n, seed=1).numpy()
stream_forward(model, x[:1 << 18], device=dev)
for chunk in (1 << 17, 1 << 18, 1 << 19):
    t = time.perf_counter(); y = stream_forward(model, x, chunk_frames=chunk, device=dev); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("%s host->host, %d frames, chunk %d: %.1f ms  = %.3g frames/s  (%.1f GB/s over the link)" % (w.name, n, chunk, dt * 1e3, n / dt, n * w.n_atoms * 12 / dt / 1e9))
xp = torch.from_numpy(x).pin_memory()
for chunk in (1 << 18, 1 << 19):
    t = time.perf_counter(); y2 = stream_forward(model, xp, chunk_frames=chunk, device=dev); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("%s pinned host->host, %d frames, chunk %d: %.1f ms  = %.3g frames/s  (%.1f GB/s over the link)" % (w.name, n, chunk, dt * 1e3, n / dt, n * w.n_atoms * 12 / dt / 1e9))
assert np.array_equal(y, y2)
