#!/usr/bin/env python3
"""AlignmentLayer.forward over frame sizes: frames_align_regs_kernel against frames_wave_kernel (MOLANN_NO_RING=1), GB/s of read + write."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from molann_amd.ann import AlignmentLayer, last_launch_info
from molann_amd.atomgroup import Universe
dev = torch.device("cuda:0")
sizes = [int(a) for a in sys.argv[1:]] or [100, 301, 1000, 2503, 5000, 12000]
for n_inp in sizes:
    rng = np.random.default_rng(n_inp)
    xyz = np.cumsum(rng.normal(size=(n_inp, 3)) * 0.9, axis=0).astype(np.float32)
    u = Universe(xyz)
    align = sorted(rng.choice(n_inp, size=max(3, n_inp // 16), replace=False).tolist())
    al = AlignmentLayer(u.atoms_by_number([a + 1 for a in align]), u.atoms).to(dev)
    n = int(min(1 << 20, (6 << 30) // (12 * n_inp)))
    x = torch.from_numpy(xyz).to(dev).unsqueeze(0) + 0.2 * torch.randn((n, n_inp, 3), device=dev)
    res = []
    for env in ((None,) if os.environ.get("ONLY_REGS") else (None, "1")):
        if env: os.environ["MOLANN_NO_RING"] = env
        else: os.environ.pop("MOLANN_NO_RING", None)
        with torch.no_grad():
            for _ in range(2): al(x)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5): y = al(x)
            b.record(); b.synchronize()
        ms = a.elapsed_time(b) / 5
        res.append("%7.3f ms %6.0f GB/s  %s" % (ms, 24.0 * n_inp * n / ms / 1e6, last_launch_info(al)[:46]))
        del y
    print("n_inp %5d, %7d frames: %s" % (n_inp, n, " | ".join(res)))
    del x
