#!/usr/bin/env python3
"""C1 (1024 frames) and 1-frame calls: where the host time of an eager forward goes.  Each line is the per-call time of a loop of
back-to-back calls (the GPU work, ~3-4 us, overlaps the host's next call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from molann_amd import workloads as wl
dev = torch.device("cuda:0")
w = wl.get_workload(sys.argv[1] if len(sys.argv) > 1 else "C1")
model = wl.build_model(w, dev).requires_grad_(False)
def timeit(fn, n=3000):
    with torch.no_grad():
        for _ in range(100): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6
for nfr in (1024, 1):
    x = w.make_frames(nfr, device=dev)
    with torch.no_grad():
        model(x)
    st = model._fast_state(x)
    lins = st["linears"]
    al = st["al"]
    ref = al.ref_x if al is not None else st["no_ref"]
    ws, bs = [l.weight for l in lins], [l.bias for l in lins]
    desc = st["desc"]
    print("--- %d frame(s)" % nfr)
    print("model(x)                                  %.2f us" % timeit(lambda: model(x)))
    print("torch.ops.molann.run(x, desc, ...)        %.2f us" % timeit(lambda: torch.ops.molann.run(x, desc, ref, ws, bs)))
    if hasattr(torch.ops.molann, "run_h"):
        h = st.get("handle")
        print("torch.ops.molann.run_h(x, handle, ...)    %.2f us" % timeit(lambda: torch.ops.molann.run_h(x, h, ref, ws, bs)))
    plan = model.plan_for(x)
    out = torch.empty((nfr, w.out_dim()), device=dev)
    print("plan.forward_packed(x, out) (ctypes)      %.2f us" % timeit(lambda: plan.forward_packed(x, out)))
    print("torch.empty((n, d_out))                   %.2f us" % timeit(lambda: torch.empty((nfr, w.out_dim()), device=dev)))
    print("x.add_(0) (one ATen kernel launch)        %.2f us" % timeit(lambda: x.add_(0)))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        ts = []
        for _ in range(200):
            a.record(); model(x); b.record(); b.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    print("one synchronised forward, event to event  %.2f us (median), %.2f (min)" % (ts[len(ts) // 2], ts[0]))
