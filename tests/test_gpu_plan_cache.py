"""The plans behind the modules are cached state: these tests pin what the cache must never do - serve one model with
another model's weights or reference, miss an update of a live tensor, leak plans without bound, let two streams
race on a plan-owned workspace - and that a second differentiation of the kernel backward is refused, not wrong."""

import gc
import io

import pytest
import torch

from build_util import oracle_for_workload, workload_model
from molann_amd import workloads as wl

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _err(y, w, model, x):
    return float((y.cpu().double() - oracle_for_workload(w, model, x, torch.float64)).abs().max())


def _fresh(w, dev, seed):
    """A model of workload `w` whose weights AND alignment reference differ from seed to seed."""
    model = workload_model(w, torch.device("cpu"), seed=seed)
    al = getattr(model.preprocessing_layer, "align_layer", None)
    if hasattr(al, "ref_x"):
        g = torch.Generator().manual_seed(1000 + seed)
        q = torch.randn(1, 4, generator=g)
        rot = wl.quaternion_to_matrix(q / q.norm())[0]
        with torch.no_grad():
            al.ref_x.copy_(al.ref_x @ rot)
    return model.to(dev)


@pytest.mark.parametrize("name", ["C3", "C1"])
def test_second_model_of_the_same_architecture_gets_its_own_weights(name, hip_device):
    """model A runs, is freed, model B of the same description is built - the caching allocator
    hands B's parameters A's addresses with equal version counters.  B must compute with B's tensors."""
    w = wl.get_workload(name)
    x = w.make_frames(515, seed=3).to(hip_device)
    for seed in range(6):
        model = _fresh(w, hip_device, seed)
        with torch.no_grad():
            y = model(x)
        torch.cuda.synchronize()
        assert _err(y, w, model, x) <= TOL, seed
        del model, y
        gc.collect()
        torch.cuda.synchronize()


def test_two_live_models_of_one_architecture_interleaved(hip_device):
    w = wl.get_workload("C3")
    x = w.make_frames(300, seed=4).to(hip_device)
    a, b = _fresh(w, hip_device, 11), _fresh(w, hip_device, 12)
    sa, sb = (torch.jit.script(m) for m in (a, b))
    with torch.no_grad():
        for _ in range(3):
            for m, s in ((a, sa), (b, sb)):
                y = m(x)
                assert _err(y, w, m, x) <= TOL
                assert torch.equal(s(x), y)
    assert float((a(x) - b(x)).abs().max()) > 1e-3     # they really are different models


def test_loaded_copies_of_one_file_do_not_serve_each_other(hip_device):
    """Two torch.jit.load()s of the same file carry the same description (instance id included) and so share a plan:
    the identity check on the live tensors is what keeps them apart."""
    w = wl.get_workload("C3")
    model = _fresh(w, hip_device, 5)
    buf = io.BytesIO()
    torch.jit.save(torch.jit.script(model), buf)
    one = torch.jit.load(io.BytesIO(buf.getvalue()), map_location=hip_device)
    two = torch.jit.load(io.BytesIO(buf.getvalue()), map_location=hip_device)
    x = w.make_frames(200, seed=6).to(hip_device)
    with torch.no_grad():
        for p in two.parameters():
            p.mul_(1.5)
        y1, y2, y1b = one(x), two(x), one(x)
    assert torch.equal(y1, y1b) and _err(y1, w, model, x) <= TOL
    assert float((y1 - y2).abs().max()) > 1e-3


def test_data_edits_need_refresh_parameters_and_get_it(hip_device):
    w = wl.get_workload("C3")
    model = _fresh(w, hip_device, 7)
    x = w.make_frames(128, seed=8).to(hip_device)
    with torch.no_grad():
        y0 = model(x)
        for p in model.parameters():
            p.data.mul_(0.5)            # invisible to the version counter of p
        model.refresh_parameters()
        y1 = model(x)
    assert _err(y1, w, model, x) <= TOL and float((y0 - y1).abs().max()) > 1e-3
    # the ctypes plan (the path under grad mode) honours it as well
    xg = x.clone().requires_grad_(True)
    model(xg).sum().backward()
    with torch.no_grad():
        for p in model.parameters():
            p.data.mul_(2.0)
    model.refresh_parameters()
    y2 = model(xg)
    assert float((y2.detach() - y0).abs().max()) <= 2e-6


def test_replaced_parameter_is_seen(hip_device):
    w = wl.get_workload("C3")
    model = _fresh(w, hip_device, 9)
    x = w.make_frames(128, seed=8).to(hip_device)
    with torch.no_grad():
        model(x)
        lin = model.ann_layers[0]
        old = lin.weight
        lin.weight = torch.nn.Parameter(old.detach() * 0.25)
        del old
        gc.collect()
        y = model(x)
    assert _err(y, w, model, x) <= TOL


def test_cache_is_bounded_and_released(hip_device):
    import molann_amd.script as script
    script.load_ops()
    w = wl.get_workload("C1")
    x = w.make_frames(64, seed=1).to(hip_device)
    torch.ops.molann.drop_plans()
    models = []
    for seed in range(5):
        m = _fresh(w, hip_device, seed)
        with torch.no_grad():
            m(x)
        models.append(m)
    assert torch.ops.molann.cached_plans() == 5
    del models, m
    gc.collect()
    assert torch.ops.molann.cached_plans() == 0           # weakref.finalize -> molann::release
    scripted = [torch.jit.script(_fresh(w, hip_device, s)) for s in range(70)]
    with torch.no_grad():
        for s in scripted:
            s(x)
    assert torch.ops.molann.cached_plans() <= 64          # LRU bound (MOLANN_PLAN_CACHE_SIZE)
    with torch.no_grad():
        y = scripted[0](x)                                # evicted long ago: rebuilt, still right
    assert torch.isfinite(y).all()
    torch.ops.molann.drop_plans()


def test_double_backward_gives_a_graph_and_third_order_is_refused(hip_device):
    """ADVICE r1 (medium) / VERDICT r2 item 9: the reference differentiates twice through plain autograd.  create_graph=True now
    returns gradients WITH a graph (tests/test_gpu_backward.py::test_double_backward_matches_reference_autograd checks the values);
    differentiating those a third time is refused, not wrong."""
    w = wl.get_workload("C3")
    model = _fresh(w, hip_device, 2)
    x = w.make_frames(32, seed=2).to(hip_device).requires_grad_(True)
    for m in (model, torch.jit.script(model)):
        y = m(x)
        (gx,) = torch.autograd.grad(y.sum(), x, create_graph=True)
        assert gx.requires_grad
        (g2,) = torch.autograd.grad((gx * gx).sum(), x, retain_graph=True)
        assert torch.isfinite(g2).all()
        with pytest.raises(RuntimeError):                  # a graph for the second-order gradients = third order: refused
            torch.autograd.grad((gx * gx).sum(), x, create_graph=True)
        (gx1,) = torch.autograd.grad(m(x).sum(), x)       # first order still fine afterwards
        assert torch.isfinite(gx1).all() and float((gx1 - gx.detach()).abs().max()) <= 1e-4 * max(1.0, float(gx1.abs().max()))


def test_unfused_forward_from_two_streams(hip_device):
    """A plan with a wide MLP owns a workspace, a side stream and events: forwards of one model
    issued from two streams must not race on them."""
    w = wl.get_workload("C4")
    model = workload_model(w, hip_device)
    xs = [w.make_frames(96, seed=20 + i).to(hip_device) for i in range(2)]
    with torch.no_grad():
        want = [model(x).clone() for x in xs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=hip_device) for _ in range(2)]
    outs = [[], []]
    with torch.no_grad():
        for rep in range(6):
            for i, st in enumerate(streams):
                with torch.cuda.stream(st):
                    outs[i].append(model(xs[i]))
    torch.cuda.synchronize()
    for i in range(2):
        for y in outs[i]:
            assert torch.equal(y, want[i])


def test_second_model_gets_its_own_reference(hip_device):
    """The same through `ref_x`, where it is visible: aligned positions (C3p), scripted modules freed and rebuilt."""
    from oracle import molann_oracle as mo
    w = wl.get_workload("C3p")
    x = w.make_frames(257, seed=9)
    feats = [(t, [a - 1 for a in atoms]) for t, atoms in w.features]
    al = [a - 1 for a in w.align]
    for seed in range(5):
        model = _fresh_pp(w, hip_device, seed)
        s = torch.jit.script(model)
        with torch.no_grad():
            y = s(x.to(hip_device))
            ye = model(x.to(hip_device))
        ref = model.align_layer.ref_x.detach().cpu().double()
        want = mo.preprocessing_forward(x.double(), feats, w.use_angle_value, al, ref)
        assert float((y.cpu().double() - want).abs().max()) <= 2e-5, seed
        assert torch.equal(y, ye)
        del model, s, y, ye
        gc.collect()


def _fresh_pp(w, dev, seed):
    model = workload_model(w, torch.device("cpu"))
    g = torch.Generator().manual_seed(1000 + seed)
    q = torch.randn(1, 4, generator=g)
    rot = wl.quaternion_to_matrix(q / q.norm())[0]
    with torch.no_grad():
        model.align_layer.ref_x.copy_(model.align_layer.ref_x @ rot)
    return model.to(dev)


def test_a_captured_graph_keeps_its_plan_through_lru_eviction(hip_device):
    """a HIP graph captured through `torch.ops.molann.run` holds raw pointers into the cached plan and a
    replay never touches the cache's LRU clock.  With room for two plans: capture, build three other models (evictions),
    replay - the graph's plan must still be there - and compare with an eager forward.  Child process: the cache size is
    read once per process."""
    import os
    import subprocess
    import sys
    code = r'''
import gc, sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from build_util import workload_model
from molann_amd import workloads as wl
from molann_amd.graph import GraphedForward
dev = torch.device("cuda:0")
w = wl.get_workload("C3")
model = workload_model(w, dev, seed=1).requires_grad_(False)
x0 = w.make_frames(256, seed=1).to(dev)
g = GraphedForward(model, x0)
assert torch.ops.molann.cached_plans() >= 1
others = []
for seed in (2, 3, 4):
    for name in ("C1", "C3"):
        m = workload_model(wl.get_workload(name), dev, seed=seed).requires_grad_(False)
        with torch.no_grad():
            m(wl.get_workload(name).make_frames(64, seed=seed).to(dev))
        others.append(m)
torch.cuda.synchronize()
assert torch.ops.molann.cached_plans() <= 3, torch.ops.molann.cached_plans()    # two + the pinned one at most
x = w.make_frames(256, seed=9).to(dev)
got = g(x).clone()
torch.cuda.synchronize()
with torch.no_grad():
    want = model(x)
assert torch.equal(got, want)
# a model dropped right behind its launch: the plan's code objects and memory outlive the queued kernel
m = workload_model(w, dev, seed=7).requires_grad_(False)
big = w.make_frames(1 << 18, device=dev, seed=3)
with torch.no_grad():
    y = m(big)
del m
gc.collect()
torch.cuda.synchronize()
assert torch.isfinite(y).all()
del g
gc.collect()
print("graph-pin ok")
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MOLANN_PLAN_CACHE_SIZE="2")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0 and "graph-pin ok" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])


def test_graphed_forces_refuses_the_three_launch_backward(hip_device, monkeypatch):
    """GraphedForces captures one launch of the one-pass backward; a plan whose backward is the
    three-launch path (plan-owned events and workspace) is refused instead of captured."""
    from molann_amd.graph import GraphedForces
    monkeypatch.setenv("MOLANN_NO_RING_BWD", "1")
    w = wl.get_workload("C3")
    model = workload_model(w, hip_device, seed=5).requires_grad_(False)
    with pytest.raises(NotImplementedError):
        GraphedForces(model, w.make_frames(8, seed=1).to(hip_device))
