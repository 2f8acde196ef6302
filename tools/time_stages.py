import sys, torch
sys.path.insert(0,'/root/repo')
from molann_amd import workloads as wl
from molann_amd.ann import last_launch_info
dev=torch.device('cuda:0')
w=wl.get_workload('C3')
model=wl.build_model(w,dev).requires_grad_(False)
pp=model.preprocessing_layer
xs=[w.make_frames(w.frames,device=dev,seed=i) for i in range(5)]
def t(m):
    with torch.no_grad():
        for i in range(5): m(xs[i%5])
        torch.cuda.synchronize()
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(50): m(xs[i%5])
        b.record(); b.synchronize()
    return a.elapsed_time(b)/50*1e3
print('C3 full      %.1f us'%t(model), last_launch_info(model))
print('C3 features  %.1f us'%t(pp), last_launch_info(pp))
