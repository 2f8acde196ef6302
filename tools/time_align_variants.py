import sys, torch
sys.path.insert(0,'/root/repo')
from molann_amd import workloads as wl
from molann_amd.ann import AlignmentLayer, FeatureLayer, PreprocessingANN, last_launch_info
from molann_amd.atomgroup import Universe
from molann_amd.feature import Feature
dev=torch.device('cuda:0')
w=wl.get_workload('A3')
U=Universe(wl.ALA_DIPEPTIDE_XYZ)
al=AlignmentLayer(U.atoms_by_number(w.align), U.atoms).to(dev)
pp=PreprocessingANN(AlignmentLayer(U.atoms_by_number(w.align), U.atoms), FeatureLayer([Feature('p','position',U.atoms)], U.atoms)).to(dev)
xs=[w.make_frames(w.frames,device=dev,seed=i) for i in range(4)]
def t(m):
    with torch.no_grad():
        for i in range(4): m(xs[i%4])
        torch.cuda.synchronize()
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(40): m(xs[i%4])
        b.record(); b.synchronize()
    return a.elapsed_time(b)/40*1e3
print('AlignmentLayer.forward      %.1f us'%t(al), last_launch_info(al))
print('align + position features   %.1f us'%t(pp), last_launch_info(pp))
with torch.no_grad():
    print('max diff', float((al(xs[0]).reshape(-1,66)-pp(xs[0])).abs().max()))
