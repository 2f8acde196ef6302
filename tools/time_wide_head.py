#!/usr/bin/env python3
"""C3's preprocessing with MLP heads outside the fused lane kernel (two-stream path): whole-model time per 1 M frames."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from molann_amd import workloads as wl
from molann_amd.ann import MolANN, create_sequential_nn, last_launch_info
dev = torch.device("cuda:0")
w = wl.get_workload("C3")
base = wl.build_model(w, dev)
xs = [w.make_frames(1 << 20, device=dev, seed=i) for i in range(3)]
for dims in ([6, 32, 8], [6, 64, 64, 8], [6, 128, 128, 8]):
    model = MolANN(base.preprocessing_layer, create_sequential_nn(dims).to(dev)).requires_grad_(False)
    with torch.no_grad():
        for i in range(3): model(xs[i % 3])
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(20): model(xs[i % 3])
        b.record(); b.synchronize()
    print("%-18s %.1f us per 1M frames   %s" % (dims, a.elapsed_time(b) / 20 * 1e3, last_launch_info(model)[:110]))
