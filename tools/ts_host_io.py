"""Helper of tools/check_torchscript_host.sh: `prepare` writes model.pt / frames.bin without touching the GPU,
`check` compares the C++ host's out.bin with the eager model (and the fp64 oracle) on the GPU."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from molann_amd import workloads as wl  # noqa: E402

mode, d, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
w = wl.get_workload("C3")
model = wl.build_model(w, None, 0)
x = w.make_frames(n, seed=21)
if mode == "prepare":
    torch.jit.script(model).save(os.path.join(d, "model.pt"))
    x.numpy().astype(np.float32).tofile(os.path.join(d, "frames.bin"))
    print("wrote model.pt and %d frames" % n)
else:
    from build_util import oracle_for_workload
    dev = torch.device("cuda:0")
    model = model.to(dev)
    xg = x.to(dev).requires_grad_(True)
    y = model(xg)
    (g,) = torch.autograd.grad(y.sum(), [xg])
    raw = np.fromfile(os.path.join(d, "out.bin"), dtype=np.float32)
    d_out = y.shape[1]
    y_host = torch.from_numpy(raw[:n * d_out].reshape(n, d_out))
    g_host = torch.from_numpy(raw[n * d_out:].reshape(n, 22, 3))
    want = oracle_for_workload(w, model, x, torch.float64)
    e_y = float((y_host - y.detach().cpu()).abs().max())
    e_g = float((g_host - g.cpu()).abs().max())
    e_o = float((y_host.double() - want).abs().max())
    print("C++ host vs eager: out %.3g, forces %.3g; C++ host vs fp64 oracle: %.3g" % (e_y, e_g, e_o))
    assert e_y == 0.0 and e_g == 0.0 and e_o <= 1e-5
    print("ok")
