import sys, time, itertools, torch
sys.path.insert(0,'/root/repo')
from molann_amd import workloads as wl
from molann_amd.ann import AlignmentLayer, FeatureLayer, MolANN, PreprocessingANN, create_sequential_nn, last_launch_info
from molann_amd.atomgroup import Universe
from molann_amd.feature import Feature
dev=torch.device('cuda:0')
U=Universe(wl.ALA_DIPEPTIDE_XYZ)
heavy=[2,5,6,7,9,11,15,16,17,19]
pairs=list(itertools.combinations(heavy,2))[:45]
dihs=[(5,7,9,15),(7,9,15,17),(2,5,7,9),(9,15,17,19)]
feats=[Feature('b%d'%i,'bond',U.atoms_by_number(list(p))) for i,p in enumerate(pairs)]+[Feature('d%d'%i,'dihedral',U.atoms_by_number(list(d))) for i,d in enumerate(dihs)]
for n_feat in (40, 49):
    fl=FeatureLayer(feats[:n_feat],U.atoms,False)
    pp=PreprocessingANN(AlignmentLayer(U.atoms_by_number([2,5,7,9,15,17,19]),U.atoms),fl)
    t0=time.time()
    model=MolANN(pp,create_sequential_nn([pp.output_dimension(),30,30,2])).to(dev).requires_grad_(False)
    w=wl.get_workload('C3')
    xs=[w.make_frames(1<<20,device=dev,seed=i) for i in range(3)]
    with torch.no_grad():
        model(xs[0]); torch.cuda.synchronize(); t1=time.time()
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(20): model(xs[i%3])
        b.record(); b.synchronize()
    print(n_feat,'features d=%d'%pp.output_dimension(), '%.1f us per 1M frames'%(a.elapsed_time(b)/20*1e3), 'first call %.1fs'%(t1-t0), last_launch_info(model)[:60])
