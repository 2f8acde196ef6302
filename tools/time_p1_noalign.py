import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molann_amd import workloads as wl
from molann_amd.ann import last_launch_info
w = wl.get_workload("P1")
w.align = None; w.rigid_motion = False
dev = torch.device("cuda:0")
model = wl.build_model(w, dev)
x = w.make_frames(1 << 20, device=dev)
with torch.no_grad():
    for _ in range(3): model(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): model(x)
    b.record(); b.synchronize()
print("P1 without alignment: %.1f us per 1M frames  %s" % (a.elapsed_time(b) * 100, last_launch_info(model)))
