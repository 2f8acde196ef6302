#!/usr/bin/env python3
"""Standalone rate of the wide-MLP kernels (molann_mlp_packed_f32) on random features.
   python tools/time_mlp.py [bf16|f32] [n_frames] [dims...]      (MOLANN_NO_JIT=1 -> the generic kernel)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from molann_amd import _capi

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
dims = [int(v) for v in sys.argv[3:]] or [341, 512, 256, 16]
dev = torch.device("cuda:0")
k, r = divmod(dims[0], 3)
feats = [(_capi.FEAT_POSITION, list(range(k)))]
if r == 1:
    feats.append((_capi.FEAT_BOND, [k, k + 1]))
elif r == 2:
    feats.append((_capi.FEAT_DIHEDRAL, [k, k + 1, k + 2, k + 3]))
plan = _capi.Plan(k + 8, features=feats, layer_dims=dims, activation=int(os.environ.get("ACT", _capi.ACT_TANH)),
                  mlp_precision=_capi.MLP_BF16 if prec == "bf16" else _capi.MLP_F32)
ws = [torch.randn(j, i, device=dev) / i ** 0.5 for i, j in zip(dims[:-1], dims[1:])]
bs = [torch.zeros(j, device=dev) for j in dims[1:]]
plan.update_mlp(ws, bs)
f = torch.randn(n, dims[0], device=dev)
out = torch.empty(n, dims[-1], device=dev)
for _ in range(3):
    plan.mlp_packed(f, out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    plan.mlp_packed(f, out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
flop = 2.0 * n * sum(i * j for i, j in zip(dims[:-1], dims[1:]))
print("%s  %s  %d frames: %.3f ms, %.3g frames/s, %.1f TFLOP/s" % (plan.last_launch_info(), dims, n, ms, n / ms * 1e3, flop / ms / 1e9))
