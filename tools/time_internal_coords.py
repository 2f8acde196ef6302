#!/usr/bin/env python3
"""All consecutive bonds, angles and dihedrals of a chain (the usual internal-coordinate featurisation): features only, with and without an alignment in front.
   python tools/time_internal_coords.py [n_atoms]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from molann_amd import workloads as wl
from molann_amd.ann import AlignmentLayer, FeatureLayer, PreprocessingANN, last_launch_info
from molann_amd.atomgroup import Universe
from molann_amd.feature import Feature
n_atoms = int(sys.argv[1]) if len(sys.argv) > 1 else 166
xyz = wl.synthetic_chain(n_atoms=n_atoms, step=1.4, seed=11)
u = Universe(xyz)
feats = []
for k, name in ((2, "bond"), (3, "angle"), (4, "dihedral")):
    for s in range(1, n_atoms - k + 2):
        feats.append(Feature("%s%d" % (name, s), name, u.atoms_by_number(list(range(s, s + k)))))
dev = torch.device("cuda:0")
n = min(1 << 20, (3 << 30) // (12 * n_atoms))
x = torch.from_numpy(xyz).to(dev).unsqueeze(0) + 0.1 * torch.randn((n, n_atoms, 3), device=dev)
for with_align in (False, True):
    al = AlignmentLayer(u.atoms_by_number(list(range(2, n_atoms + 1, 4))), u.atoms) if with_align else None
    pp = PreprocessingANN(al, FeatureLayer(feats, u.atoms, False)).to(dev)
    with torch.no_grad():
        for _ in range(2): y = pp(x)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): y = pp(x)
        b.record(); b.synchronize()
    ms = a.elapsed_time(b) / 5
    bytes_ = n * (12.0 * n_atoms + 4.0 * y.shape[1])
    print("%d atoms, %d features (d=%d), align=%s, %d frames: %.3f ms  %.0f GB/s (frame in + features out)  %s" % (n_atoms, len(feats), y.shape[1], with_align, n, ms, bytes_ / ms / 1e6, last_launch_info(pp)[:110]))
    del y
