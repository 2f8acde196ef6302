#!/usr/bin/env python3
"""Aligned coordinates of mid-size frames two ways: AlignmentLayer.forward, and a position feature over all atoms behind the
same alignment (the same numbers as a [N, 3 n] feature row).   python tools/time_position_all.py [n_atoms] [n_align_step]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from molann_amd import workloads as wl
from molann_amd.ann import AlignmentLayer, FeatureLayer, PreprocessingANN, last_launch_info
from molann_amd.atomgroup import Universe
from molann_amd.feature import Feature
n_atoms = int(sys.argv[1]) if len(sys.argv) > 1 else 166
step = int(sys.argv[2]) if len(sys.argv) > 2 else 4
xyz = wl.synthetic_chain(n_atoms=n_atoms, step=1.4, seed=11)
u = Universe(xyz)
dev = torch.device("cuda:0")
al = AlignmentLayer(u.atoms_by_number(list(range(2, n_atoms + 1, step))), u.atoms)
pp = PreprocessingANN(al, FeatureLayer([Feature("all", "position", u.atoms)], u.atoms, False)).to(dev)
al = al.to(dev)
n = min(1 << 20, (4 << 30) // (12 * n_atoms))
x = torch.from_numpy(xyz).to(dev).unsqueeze(0) + 0.1 * torch.randn((n, n_atoms, 3), device=dev)
for name, m in (("AlignmentLayer", al), ("position feature", pp)):
    with torch.no_grad():
        for _ in range(2): y = m(x)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): y = m(x)
        b.record(); b.synchronize()
    ms = a.elapsed_time(b) / 5
    print("%-18s %d atoms, %d frames: %.3f ms  %.0f GB/s  %s" % (name, n_atoms, n, ms, 24.0 * n_atoms * n / ms / 1e6, last_launch_info(m)[:120]))
    if name == "AlignmentLayer": ya = y.reshape(n, -1).clone()
print("max difference between the two:", float((ya - y).abs().max()))
