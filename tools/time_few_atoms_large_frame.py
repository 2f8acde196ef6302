#!/usr/bin/env python3
"""A plan that touches ~20 atoms of a 5000-atom frame: the specialised lane kernel (compact tile, independent of
the frame size) against the wave-per-frame kernel (MOLANN_NO_JIT=1).  python tools/time_few_atoms_large_frame.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from molann_amd import workloads as wl
from molann_amd.ann import AlignmentLayer, FeatureLayer, MolANN, PreprocessingANN, create_sequential_nn, last_launch_info
from molann_amd.atomgroup import Universe
from molann_amd.feature import Feature
dev = torch.device("cuda:0")
n_inp, n = 5000, 1 << 17
xyz = wl.synthetic_chain()
u = Universe(xyz)
lig = list(range(2400, 2412))                                     # a 12-atom "ligand" in the middle of the frame
feats = [Feature("d%d" % i, "dihedral", u.atoms_by_number([a + 1 for a in lig[i:i + 4]])) for i in range(0, 8, 2)] + \
        [Feature("b", "bond", u.atoms_by_number([lig[0] + 1, lig[11] + 1]))]
align = [a + 1 for a in lig[:8]]
pp = PreprocessingANN(AlignmentLayer(u.atoms_by_number(align), u.atoms), FeatureLayer(feats, u.atoms, False))
model = MolANN(pp, create_sequential_nn([pp.output_dimension(), 32, 8])).to(dev).requires_grad_(False)
x = torch.from_numpy(xyz).to(dev).unsqueeze(0) + 0.1 * torch.randn((n, n_inp, 3), device=dev)
with torch.no_grad():
    for _ in range(3): model(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): model(x)
    b.record(); b.synchronize()
ms = a.elapsed_time(b) / 10
print("%s\n  %d frames of %d atoms: %.3f ms = %.3g frames/s" % (last_launch_info(model), n, n_inp, ms, n / ms * 1e3))
