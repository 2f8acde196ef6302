"""Backward of the HIP path (molann_backward_f32 through torch.autograd) against (i) the reference's own
autograd results (tests/golden/grad_*.npz, written by oracle/gen_golden.py --grads) and (ii) torch autograd
through the fp64 oracle on fresh batches."""

import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN_DIR
from molann_amd import workloads as wl
from molann_amd.ann import AlignmentLayer, FeatureLayer, MolANN, PreprocessingANN, create_sequential_nn
from molann_amd.atomgroup import Universe
from molann_amd.feature import Feature
from oracle import molann_oracle as mo

pytestmark = pytest.mark.gpu
U = Universe(wl.ALA_DIPEPTIDE_XYZ)


def _model_from_golden(d, dev):
    U = Universe(d["ref_xyz"]) if "ref_xyz" in d else globals()["U"]       # a system of its own (the 166-atom chain of P1)
    ptr = d["feat_ptr"]
    feats = [Feature("f%d" % i, wl.TYPE_NAMES[int(t)], U.atoms_by_number(d["feat_numbers"][ptr[i]:ptr[i + 1]].tolist()))
             for i, t in enumerate(d["feat_types"].tolist())]
    al = AlignmentLayer(U.atoms_by_number(d["align_numbers"].tolist()), U.atoms) if "align_numbers" in d else None
    pp = PreprocessingANN(al, FeatureLayer(feats, U.atoms, bool(d["use_angle_value"])))
    if "mlp_dims" not in d:
        return pp.to(dev)
    nn = create_sequential_nn(d["mlp_dims"].tolist())
    lins = [m for m in nn if isinstance(m, torch.nn.Linear)]
    with torch.no_grad():
        for i, lin in enumerate(lins):
            lin.weight.copy_(torch.from_numpy(d["W%d" % i]))
            lin.bias.copy_(torch.from_numpy(d["b%d" % i]))
    return MolANN(pp, nn).to(dev)


def _close(got, ref32, ref64, what):
    """within 1e-4 (relative to the gradient's scale) of the fp64 reference, or no further from it than twice
    the reference's own fp32 run"""
    scale = max(1e-3, float(np.abs(ref64).max()))
    err = float(np.abs(got.astype(np.float64) - ref64).max())
    own = float(np.abs(ref32.astype(np.float64) - ref64).max())
    assert err <= max(1e-4 * scale, 2.0 * own), (what, err, own, scale)


@pytest.mark.parametrize("name", ["grad_molann_C1", "grad_molann_C3", "grad_features_C2", "grad_features_C3_val",
                                  "grad_features_C3p", "grad_molann_P1", "grad_features_P1"])
def test_gradients_match_reference_autograd(name, hip_device):
    d = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    model = _model_from_golden(d, hip_device)
    x = torch.from_numpy(d["x"]).to(hip_device).requires_grad_(True)
    out = model(x)
    assert out.requires_grad
    _close(out.detach().cpu().numpy(), d["out_f32"], d["out_f64"], "forward")
    (out * torch.from_numpy(d["G"]).to(hip_device)).sum().backward()
    torch.cuda.synchronize()
    _close(x.grad.cpu().numpy(), d["gx_f32"], d["gx_f64"], "grad_x")
    for i, p in enumerate(model.parameters()):
        _close(p.grad.cpu().numpy(), d["gp%d_f32" % i], d["gp%d_f64" % i], "param %d" % i)


def _oracle_grads(w, model, x, G, act="tanh"):
    feats = [(t, [a - 1 for a in atoms]) for t, atoms in w.features]
    al = [a - 1 for a in w.align] if w.align is not None else None
    ref_x = mo.center_reference(torch.from_numpy(w.ref_xyz[[a - 1 for a in w.align]])).double() if al else None
    xx = x.detach().cpu().double().requires_grad_(True)
    if w.mlp_dims:
        lins = [m for m in model.ann_layers if isinstance(m, torch.nn.Linear)]
        ws = [l.weight.detach().cpu().double().requires_grad_(True) for l in lins]
        bs = [l.bias.detach().cpu().double().requires_grad_(True) for l in lins]
        out = mo.molann_forward(xx, feats, ws, bs, w.use_angle_value, al, ref_x, act)
        prm = [t for pair in zip(ws, bs) for t in pair]
    else:
        out = mo.preprocessing_forward(xx, feats, w.use_angle_value, al, ref_x)
        prm = []
    (out * G.double()).sum().backward()
    return xx.grad, [p.grad for p in prm]


@pytest.mark.parametrize("cfg", ["C1", "C2", "C3", "C3p"])
@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000])
def test_gradients_vs_fp64_oracle(cfg, n, hip_device):
    w = wl.get_workload(cfg)
    model = wl.build_model(w, hip_device)
    x = w.make_frames(n, seed=300 + n).to(hip_device).requires_grad_(True)
    g = torch.Generator().manual_seed(n)
    G = torch.randn((n, w.out_dim()), generator=g)
    out = model(x)
    (out * G.to(hip_device)).sum().backward()
    gx_want, gp_want = _oracle_grads(w, model, x, G)
    scale = max(1e-3, float(gx_want.abs().max()))
    assert float((x.grad.cpu().double() - gx_want).abs().max()) <= 2e-4 * scale
    untouched = [i for i in range(w.n_atoms) if (i + 1) not in w.touched_atoms()]
    if untouched:
        assert float(x.grad[:, untouched].abs().max()) == 0.0      # atoms the plan never reads
    for p, want in zip(model.parameters(), gp_want):
        s = max(1e-3, float(want.abs().max()))
        assert float((p.grad.cpu().double() - want).abs().max()) <= 2e-4 * s


@pytest.mark.parametrize("act", [torch.nn.ReLU, torch.nn.Sigmoid, torch.nn.SiLU, torch.nn.LeakyReLU])
def test_gradients_other_activations(act, hip_device):
    w = wl.get_workload("C3")
    base = wl.build_model(w, hip_device)
    torch.manual_seed(4)
    model = MolANN(base.preprocessing_layer, create_sequential_nn([6, 20, 12, 4], activation=act()).to(hip_device))
    x = w.make_frames(500, seed=8).to(hip_device).requires_grad_(True)
    G = torch.randn((500, 4), generator=torch.Generator().manual_seed(1))
    (model(x) * G.to(hip_device)).sum().backward()
    # torch autograd on the same device modules as the check of the MLP part: features from the HIP path (no grad)
    with torch.no_grad():
        f = base.preprocessing_layer(x.detach())
    f = f.double().cpu().requires_grad_(True)
    import copy
    nn64 = copy.deepcopy(model.ann_layers).double().cpu()
    (nn64(f) * G.double()).sum().backward()
    for p, q in zip(model.ann_layers.parameters(), nn64.parameters()):
        s = max(1e-3, float(q.grad.abs().max()))
        assert float((p.grad.cpu().double() - q.grad).abs().max()) <= 2e-4 * s, act.__name__


def test_training_step_decreases_loss(hip_device):
    """A few optimiser steps through the HIP forward + backward (weights repacked after every step)."""
    w = wl.get_workload("C3")
    model = wl.build_model(w, hip_device)
    x = w.make_frames(4096, seed=2).to(hip_device)
    target = torch.randn((4096, 8), generator=torch.Generator().manual_seed(3)).to(hip_device) * 0.1
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    losses = []
    for _ in range(20):
        opt.zero_grad()
        loss = ((model(x) - target) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < 0.7 * losses[0], losses


def test_alignment_layer_gradient(hip_device):
    """AlignmentLayer on its own: gradient of the aligned frame w.r.t. x (through the rotation)."""
    w = wl.get_workload("A3")
    al = wl.build_model(w, hip_device)
    x = w.make_frames(300, seed=17).to(hip_device).requires_grad_(True)
    G = torch.randn((300, 22, 3), generator=torch.Generator().manual_seed(2))
    y = al(x)
    with torch.no_grad():
        assert torch.allclose(y, al(x.detach()), atol=1e-6)       # same values as the no-grad kernel
    (y * G.to(hip_device)).sum().backward()
    xx = x.detach().cpu().double().requires_grad_(True)
    ref_x = mo.center_reference(torch.from_numpy(w.ref_xyz[[a - 1 for a in w.align]])).double()
    (mo.align_forward(xx, [a - 1 for a in w.align], ref_x) * G.double()).sum().backward()
    scale = float(xx.grad.abs().max())
    assert float((x.grad.cpu().double() - xx.grad).abs().max()) <= 2e-4 * scale


@pytest.mark.parametrize("cfg", ["C4", "C5"])
def test_large_frames_gradients(cfg, hip_device):
    """Wave-per-frame plans: dL/dx from frames_wave_bwd_gather_kernel (features) chained with the torch MLP, parameter
    gradients from torch; against autograd through the fp64 oracle."""
    big = wl.get_workload(cfg)
    model = wl.build_model(big, hip_device)
    if cfg == "C5":
        model.mlp_precision = "f32"                 # compare like with like: the oracle is not a bf16 model
    n = 6
    x = big.make_frames(n, seed=3)
    xg = x.to(hip_device).requires_grad_(True)
    G = torch.randn((n, big.out_dim()), generator=torch.Generator().manual_seed(4))
    y = model(xg)
    (y * G.to(hip_device)).sum().backward()
    from molann_amd.ann import last_launch_info
    assert "frames_wave_bwd_gather_kernel" in last_launch_info(model.preprocessing_layer)
    lins = [m for m in model.ann_layers if isinstance(m, torch.nn.Linear)]
    xx = x.double().requires_grad_(True)
    ws = [l.weight.detach().cpu().double().requires_grad_(True) for l in lins]
    bs = [l.bias.detach().cpu().double().requires_grad_(True) for l in lins]
    al = [a - 1 for a in big.align]
    ref_x = mo.center_reference(torch.from_numpy(big.ref_xyz[al])).double()
    want = mo.molann_forward(xx, [(t, [a - 1 for a in atoms]) for t, atoms in big.features], ws, bs,
                             big.use_angle_value, al, ref_x)
    (want * G.double()).sum().backward()
    assert float((y.detach().cpu().double() - want.detach()).abs().max()) <= 1e-4
    for got, ref in [(xg.grad, xx.grad)] + [(l.weight.grad, wt.grad) for l, wt in zip(lins, ws)]:
        scale = max(1e-6, float(ref.abs().max()))
        assert float((got.cpu().double() - ref).abs().max()) <= 2e-4 * scale, cfg
    touched = set(al) | {a - 1 for _, atoms in big.features for a in atoms}
    untouched = torch.tensor(sorted(set(range(big.n_atoms)) - touched)[:200])
    assert float(xg.grad[:, untouched.to(hip_device)].abs().max()) == 0.0     # the row was zeroed, nothing else written


def test_alignment_layer_gradient_on_large_frames(hip_device):
    big = wl.get_workload("C4")
    al_atoms = [a - 1 for a in big.align]
    U5 = Universe(wl.synthetic_chain())
    layer = AlignmentLayer(U5.atoms_by_number(big.align), U5.atoms).to(hip_device)
    x = big.make_frames(3, seed=8)
    xg = x.to(hip_device).requires_grad_(True)
    G = torch.randn(x.shape, generator=torch.Generator().manual_seed(6))
    (layer(xg) * G.to(hip_device)).sum().backward()
    xx = x.double().requires_grad_(True)
    ref_x = mo.center_reference(torch.from_numpy(big.ref_xyz[al_atoms])).double()
    (mo.align_forward(xx, al_atoms, ref_x) * G.double()).sum().backward()
    scale = float(xx.grad.abs().max())
    assert float((xg.grad.cpu().double() - xx.grad).abs().max()) <= 2e-4 * scale


@pytest.mark.parametrize("act", [torch.nn.Tanh, torch.nn.GELU])
def test_wide_mlp_trains_through_hip_features(act, hip_device):
    """MLP wider than the fused backward kernel covers: feature gradients from the HIP kernels, the MLP and its
    gradient from torch; all of it against autograd through the fp64 oracle."""
    w = wl.get_workload("C3")
    feats = [Feature("f%d" % i, wl.TYPE_NAMES[t], U.atoms_by_number(atoms)) for i, (t, atoms) in enumerate(w.features)]
    al = AlignmentLayer(U.atoms_by_number(w.align), U.atoms)
    pp = PreprocessingANN(al, FeatureLayer(feats, U.atoms, False))
    torch.manual_seed(5)
    model = MolANN(pp, create_sequential_nn([6, 64, 48, 4], activation=act())).to(hip_device)
    x = w.make_frames(500, seed=9)
    xg = x.to(hip_device).requires_grad_(True)
    G = torch.randn((500, 4), generator=torch.Generator().manual_seed(1))
    y = model(xg)
    (y * G.to(hip_device)).sum().backward()
    lins = [m for m in model.ann_layers if isinstance(m, torch.nn.Linear)]
    xx = x.double().requires_grad_(True)
    ws = [l.weight.detach().cpu().double().requires_grad_(True) for l in lins]
    bs = [l.bias.detach().cpu().double().requires_grad_(True) for l in lins]
    ref_x = mo.center_reference(torch.from_numpy(w.ref_xyz[[a - 1 for a in w.align]])).double()
    f = mo.preprocessing_forward(xx, [(t, [a - 1 for a in atoms]) for t, atoms in w.features], False,
                                 [a - 1 for a in w.align], ref_x)
    h = f
    for i, (wt, b) in enumerate(zip(ws, bs)):
        h = h @ wt.T + b
        if i + 1 < len(ws):
            h = act()(h)
    (h * G.double()).sum().backward()
    assert float((y.detach().cpu().double() - h.detach()).abs().max()) <= 1e-4
    for got, want in [(xg.grad, xx.grad)] + [(l.weight.grad, wt.grad) for l, wt in zip(lins, ws)] + \
                     [(l.bias.grad, b.grad) for l, b in zip(lins, bs)]:
        scale = max(1e-6, float(want.abs().max()))
        assert float((got.cpu().double() - want).abs().max()) <= 2e-4 * scale


# ---- the halves of the backward through the C ABI (molann_forward_train_f32, molann_mlp_backward_f32,
# molann_features_backward_f32) -------------------------------------------------------------------------------------------
def _c3_plan(model, x):
    """The ctypes plan of a MolANN, weights packed."""
    return model.plan_for(x)


@pytest.mark.parametrize("dims,act", [([6, 32, 8], torch.nn.Tanh), ([6, 32, 32, 3], torch.nn.Tanh), ([6, 17, 5, 9, 2], torch.nn.Sigmoid),
                                      ([6, 4], torch.nn.Tanh), ([6, 30, 31], torch.nn.SiLU), ([6, 24, 24, 1], torch.nn.ReLU),
                                      ([6, 32, 32, 32, 32], torch.nn.LeakyReLU)])
@pytest.mark.parametrize("n", [1, 64, 777])
def test_mlp_backward_kernel_vs_fp64_autograd(dims, act, n, hip_device):
    """molann_mlp_backward_f32 alone: grad_f and the parameter gradients of ann_layers on given features."""
    import copy
    w = wl.get_workload("C3")
    base = wl.build_model(w, hip_device)
    torch.manual_seed(len(dims) * 100 + n)
    model = MolANN(base.preprocessing_layer, create_sequential_nn(dims, activation=act()).to(hip_device))
    x = w.make_frames(n, seed=5).to(hip_device).requires_grad_(True)
    plan = _c3_plan(model, x)
    f = (torch.randn((n, dims[0]), generator=torch.Generator().manual_seed(2)) * 1.5).to(hip_device)
    g = torch.randn((n, dims[-1]), generator=torch.Generator().manual_seed(3)).to(hip_device)
    gf = torch.full_like(f, float("nan"))
    gp = torch.zeros(plan.grad_params_size(), device=hip_device)
    plan.mlp_backward(f, g, gf, gp)
    gp2 = torch.ones_like(gp)                               # accumulated into, grad_f optional
    plan.mlp_backward(f, g, None, gp2)
    gf2 = torch.empty_like(f)
    plan.mlp_backward(f, g, gf2, None)                      # ... and so are the parameter gradients
    f64 = f.double().cpu().requires_grad_(True)
    nn64 = copy.deepcopy(model.ann_layers).double().cpu()
    (nn64(f64) * g.double().cpu()).sum().backward()
    s = max(1e-3, float(f64.grad.abs().max()))
    assert float((gf.cpu().double() - f64.grad).abs().max()) <= 2e-4 * s
    assert torch.equal(gf, gf2)
    want = torch.cat([t.grad.reshape(-1) for lin in [m for m in nn64 if isinstance(m, torch.nn.Linear)] for t in (lin.weight, lin.bias)])
    s = max(1e-3, float(want.abs().max()))
    assert float((gp.cpu().double() - want).abs().max()) <= 2e-4 * s
    assert float((gp2.cpu().double() - 1.0 - want).abs().max()) <= 2e-4 * s + 1e-6


def test_forward_train_keeps_features_and_output_bits(hip_device):
    w = wl.get_workload("C3")
    model = wl.build_model(w, hip_device)
    n = 4099
    x = w.make_frames(n, seed=11).to(hip_device).requires_grad_(True)
    plan = _c3_plan(model, x)
    xd = x.detach()
    out0, out1 = torch.empty((n, plan.out_dim), device=hip_device), torch.empty((n, plan.out_dim), device=hip_device)
    f0, f1 = torch.empty((n, plan.feature_dim), device=hip_device), torch.empty((n, plan.feature_dim), device=hip_device)
    plan.forward_packed(xd, out0)
    plan.features(xd, f0)                                   # the features-only twin of the fused kernel
    assert "molann_lane_jit<NL=0>" in plan.last_launch_info()
    plan.forward_train(xd, out1, f1)
    assert torch.equal(out0, out1) and torch.equal(f0, f1)
    with torch.no_grad():
        assert torch.equal(f0, model.preprocessing_layer(xd))


def test_split_backward_equals_recomputing_backward(hip_device, monkeypatch):
    """Without the one-pass kernel: molann_backward_f32 (features recomputed into the plan's workspace, chunked) against
    the saved-features pair."""
    monkeypatch.setenv("MOLANN_NO_RING_BWD", "1")
    w = wl.get_workload("C3")
    model = wl.build_model(w, hip_device)
    n = 5000
    x = w.make_frames(n, seed=12).to(hip_device).requires_grad_(True)
    plan = _c3_plan(model, x)
    assert plan.backward_kind() == 1
    xd = x.detach()
    g = torch.randn((n, plan.out_dim), generator=torch.Generator().manual_seed(4)).to(hip_device)
    gx0, gp0 = torch.empty_like(xd), torch.zeros(plan.grad_params_size(), device=hip_device)
    plan.backward(xd, g, gx0, gp0)
    assert "molann_mlp_bwd" in plan.last_launch_info() and "molann_lane_bwd" in plan.last_launch_info()
    out, f = torch.empty((n, plan.out_dim), device=hip_device), torch.empty((n, plan.feature_dim), device=hip_device)
    plan.forward_train(xd, out, f)
    gf, gx1, gp1 = torch.empty_like(f), torch.empty_like(xd), torch.zeros_like(gp0)
    plan.mlp_backward(f, g, gf, gp1)
    plan.features_backward(xd, gf, gx1)
    assert torch.equal(gx0, gx1)
    assert float((gp0 - gp1).abs().max()) <= 1e-5 * max(1.0, float(gp1.abs().max()))   # block sums added in another order
    side = torch.cuda.Stream(device=hip_device)             # the workspace is the plan's: another stream waits for the previous call
    gx2, gp2 = torch.empty_like(xd), torch.zeros_like(gp0)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        plan.backward(xd, g, gx2, gp2)
    plan.backward(xd, g, gx0, gp0.zero_())
    torch.cuda.synchronize()
    assert torch.equal(gx0, gx2) and torch.equal(gx0, gx1)


@pytest.mark.parametrize("cfg", ["C1", "C3", "C2"])
def test_one_pass_backward_against_the_two_kernel_path(cfg, hip_device, monkeypatch):
    """molann_bwd_ring (what molann_backward_f32 launches by default) against the split path on the same inputs, and
    its workspace protocol across streams."""
    w = wl.get_workload(cfg)
    n = 70001
    x = w.make_frames(n, seed=13).to(hip_device).requires_grad_(True)
    model = wl.build_model(w, hip_device)
    G = torch.randn((n, w.out_dim()), generator=torch.Generator().manual_seed(5)).to(hip_device)
    (model(x) * G).sum().backward()
    info = model.last_launch_info() if hasattr(model, "last_launch_info") else ""
    got = [x.grad.clone()] + [p.grad.clone() for p in model.parameters()]
    monkeypatch.setenv("MOLANN_NO_RING_BWD", "1")
    model2 = wl.build_model(w, hip_device)
    model2.load_state_dict(model.state_dict())
    x2 = x.detach().clone().requires_grad_(True)
    (model2(x2) * G).sum().backward()
    want = [x2.grad] + [p.grad for p in model2.parameters()]
    for a, b in zip(got, want):
        s = max(1e-3, float(b.abs().max()))
        assert float((a - b).abs().max()) <= 2e-5 * s
    monkeypatch.delenv("MOLANN_NO_RING_BWD")
    if w.mlp_dims:
        plan = model.plan_for(x)
        assert plan.backward_kind() == 2
        xd = x.detach()
        gx0, gp0 = torch.empty_like(xd), torch.zeros(plan.grad_params_size(), device=hip_device)
        plan.backward(xd, G, gx0, gp0)
        assert "molann_bwd_ring" in plan.last_launch_info()
        side = torch.cuda.Stream(device=hip_device)
        gx1, gp1 = torch.empty_like(xd), torch.zeros_like(gp0)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            plan.backward(xd, G, gx1, gp1)
        plan.backward(xd, G, gx0, gp0.zero_())
        torch.cuda.synchronize()
        assert torch.equal(gx0, gx1) and torch.equal(gx0, got[0])
        assert float((gp0 - gp1).abs().max()) <= 1e-5 * max(1.0, float(gp0.abs().max()))


@pytest.mark.parametrize("dims", [[6, 32, 32, 8], [6, 32, 32, 32, 4], [6, 16, 16, 16, 2]])
def test_one_pass_backward_with_deeper_mlps(dims, hip_device, monkeypatch):
    """Three and four Linear layers: the one-pass kernel keeps the weight fragments in an LDS image (FRAG_LDS) where they do
    not fit the registers; against the two-kernel path and, for the parameters, fp64 autograd of ann_layers."""
    import copy
    w = wl.get_workload("C3")
    base = wl.build_model(w, hip_device)
    torch.manual_seed(sum(dims))
    model = MolANN(base.preprocessing_layer, create_sequential_nn(dims).to(hip_device))
    n = 4100
    x = w.make_frames(n, seed=14).to(hip_device).requires_grad_(True)
    G = torch.randn((n, dims[-1]), generator=torch.Generator().manual_seed(6)).to(hip_device)
    (model(x) * G).sum().backward()
    assert model.plan_for(x).backward_kind() == 2
    monkeypatch.setenv("MOLANN_NO_RING_BWD", "1")
    model2 = MolANN(base.preprocessing_layer, copy.deepcopy(model.ann_layers))
    x2 = x.detach().clone().requires_grad_(True)
    (model2(x2) * G).sum().backward()
    assert model2.plan_for(x2).backward_kind() == 1
    s = max(1e-3, float(x2.grad.abs().max()))
    assert float((x.grad - x2.grad).abs().max()) <= 2e-5 * s
    with torch.no_grad():
        f = base.preprocessing_layer(x.detach())
    f64 = f.double().cpu().requires_grad_(True)
    nn64 = copy.deepcopy(model.ann_layers).double().cpu()
    (nn64(f64) * G.double().cpu()).sum().backward()
    for p, p2, q in zip(model.ann_layers.parameters(), model2.ann_layers.parameters(), nn64.parameters()):
        s = max(1e-3, float(q.grad.abs().max()))
        assert float((p.grad.cpu().double() - q.grad).abs().max()) <= 2e-4 * s
        assert float((p2.grad.cpu().double() - q.grad).abs().max()) <= 2e-4 * s


def test_large_frames_backward_with_and_without_atomics(hip_device, monkeypatch):
    """frames_wave_bwd_gather_kernel (contributions gathered per touched atom) against frames_wave_bwd_kernel (float atomics)."""
    from molann_amd.ann import last_launch_info
    big = wl.get_workload("C5")
    pp = wl.build_model(big, hip_device).preprocessing_layer
    n = 9
    x = big.make_frames(n, seed=5).to(hip_device)
    G = torch.randn((n, pp.output_dimension()), generator=torch.Generator().manual_seed(6)).to(hip_device)
    grads = []
    for atomics in (False, True):
        if atomics:
            monkeypatch.setenv("MOLANN_BWD_ATOMICS", "1")
        xg = x.clone().requires_grad_(True)
        (pp(xg) * G).sum().backward()
        assert ("frames_wave_bwd_kernel" if atomics else "frames_wave_bwd_gather_kernel") in last_launch_info(pp)
        grads.append(xg.grad)
    s = max(1e-6, float(grads[1].abs().max()))
    assert float((grads[0] - grads[1]).abs().max()) <= 1e-5 * s


def test_a_nan_frame_poisons_only_itself(hip_device):
    """The one-pass backward reuses one LDS buffer per consumer for the MLP scratch and the gradient tile: a frame of NaNs must
    not leak into the frames that buffer serves afterwards (x gradients; the parameter sums are NaN, as torch's would be)."""
    w = wl.get_workload("C3")
    model = wl.build_model(w, hip_device)
    n = 64 * 2048 + 7                      # 8 tiles per block: the consumers that meet the NaN frames go on to other tiles
    x = w.make_frames(n, seed=15).to(hip_device)
    G = torch.randn((n, w.out_dim()), generator=torch.Generator().manual_seed(7)).to(hip_device)
    xa = x.clone().requires_grad_(True)
    (model(xa) * G).sum().backward()
    bad = [3, 64 * 17 + 5]
    xb = x.clone()
    xb[bad] = float("nan")
    xb.requires_grad_(True)
    for p in model.parameters():
        p.grad = None
    (model(xb) * G).sum().backward()
    keep = torch.ones(n, dtype=torch.bool, device=hip_device)
    keep[bad] = False
    assert torch.isfinite(xb.grad[keep]).all()
    assert torch.equal(xb.grad[keep], xa.grad[keep])
    assert torch.isnan(xb.grad[bad][:, [a - 1 for a in w.touched_atoms()]]).all()


@pytest.mark.parametrize("cfg", ["C1", "C2", "C3", "C3p"])
@pytest.mark.parametrize("n", [1, 8, 64, 65, 1000])
def test_value_and_vjp_in_one_launch(cfg, n, hip_device):
    """molann_value_and_vjp_f32 (the one-pass backward that also stores the forward's outputs): values against the forward
    kernel, the vector-Jacobian product against eager autograd, through the module method, the ctypes plan and `into=`."""
    from build_util import workload_model
    from molann_amd import workloads as wl
    from molann_amd.ann import MolANN
    w = wl.get_workload(cfg)
    model = workload_model(w, hip_device).requires_grad_(False)
    x = w.make_frames(n, seed=50 + n).to(hip_device)
    dy = torch.randn((n, w.out_dim()), generator=torch.Generator().manual_seed(n)).to(hip_device)
    xe = x.clone().requires_grad_(True)
    ye = model(xe)
    (dxe,) = torch.autograd.grad(ye, xe, dy)
    if isinstance(model, MolANN):
        y, dx = model.value_and_vjp(x, dy)
        assert "molann_bwd_ring<values>" in model.last_launch_info(), model.last_launch_info()
        plan = model.plan_for(x)
    else:       # a PreprocessingANN: through its plan
        plan = model._plans()[("features", hip_device.index)].plan
        y, dx = torch.empty_like(ye), torch.empty_like(x)
        with torch.cuda.device(hip_device):
            plan.value_and_vjp(x, dy, y, dx)
    scale = max(1.0, float(dxe.abs().max()))
    assert float((y - ye.detach()).abs().max()) <= 2e-6 * max(1.0, float(ye.abs().max()))
    assert float((dx - dxe).abs().max()) <= 1e-6 * scale
    y2, dx2 = torch.full_like(y, float("nan")), torch.full_like(dx, float("nan"))
    with torch.cuda.device(hip_device):
        plan.value_and_vjp(x, dy, y2, dx2)
    torch.cuda.synchronize()
    assert torch.equal(y2, y) and torch.equal(dx2, dx)
    if isinstance(model, MolANN):
        y3, dx3 = torch.empty_like(y), torch.empty_like(dx)
        r = model.value_and_vjp(x, dy, into=(y3, dx3))
        assert r[0].data_ptr() == y3.data_ptr() and torch.equal(y3, y) and torch.equal(dx3, dx)


def test_value_and_vjp_latency_and_jacobian(hip_device):
    """The MD-step use: values + forces of 1 .. 64 frames per call (host time per call printed), and the Jacobian of one frame's
    values from one launch on d_out copies of it."""
    import time
    from build_util import workload_model
    from molann_amd import workloads as wl
    from molann_amd.graph import GraphedForces
    w = wl.get_workload("C3")
    model = workload_model(w, hip_device).requires_grad_(False)
    for n in (1, 64):
        x = w.make_frames(n, seed=3).to(hip_device)
        dy = torch.randn((n, w.out_dim()), generator=torch.Generator().manual_seed(5)).to(hip_device)
        y, dx = torch.empty((n, w.out_dim()), device=hip_device), torch.empty_like(x)
        for _ in range(20):
            model.value_and_vjp(x, dy, into=(y, dx))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(500):
            model.value_and_vjp(x, dy, into=(y, dx))
        torch.cuda.synchronize()
        t_into = (time.perf_counter() - t0) / 500 * 1e6
        t0 = time.perf_counter()
        for _ in range(500):
            model.value_and_vjp(x, dy)
        torch.cuda.synchronize()
        t_fresh = (time.perf_counter() - t0) / 500 * 1e6
        g = GraphedForces(model, x)
        for _ in range(20):
            g(x); g.vjp(dy)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(500):
            g(x); g.vjp(dy)
        torch.cuda.synchronize()
        t_graph = (time.perf_counter() - t0) / 500 * 1e6
        print("values + forces, %d frame(s): %.1f us per call (one launch, caller's buffers), %.1f (fresh tensors), %.1f (two graph replays)"
              % (n, t_into, t_fresh, t_graph))
        yg, dxg = g.value_and_vjp(x, dy)
        assert torch.equal(yg, y) and torch.equal(dxg, dx)
    d_out = w.out_dim()
    x1 = w.make_frames(1, seed=9).to(hip_device)
    yj, J = model.value_and_vjp(x1.expand(d_out, -1, -1).contiguous(), torch.eye(d_out, device=hip_device))
    xe = x1.clone().requires_grad_(True)
    ye = model(xe)
    assert float((yj[0] - ye.detach()[0]).abs().max()) <= 2e-6
    for k in range(d_out):
        (gk,) = torch.autograd.grad(ye[0, k], xe, retain_graph=True)
        assert float((J[k] - gk[0]).abs().max()) <= 1e-6 * max(1.0, float(gk.abs().max()))


@pytest.mark.parametrize("name", ["grad2_molann_C3", "grad2_features_C3p", "grad2_features_C2"])
@pytest.mark.parametrize("dtype", ["float32", "float64", "scripted"])
def test_double_backward_matches_reference_autograd(name, dtype, hip_device):
    """create_graph=True (round 3): a loss on forces.  E = sum(model(x) * G), F = dE/dx with create_graph=True, L = sum(F * F);
    dL/dx and dL/d(parameters) against the REFERENCE's autograd through its SVD (tests/golden/grad2_*.npz, written by
    oracle/gen_golden.py --round3).  The second-order terms are central differences of the float64 kernels along the cotangent
    (molann_amd/ann.py: _FeatBackward64; csrc/molann_torch.cpp: FeatBackward64Fn): 1e-6 of scale in float64, and no further from
    the float64 reference than a few times the reference's own float32 run in float32.  Eager float32 (the operator's autograd
    node), eager float64 (the Python functions) and a scripted model."""
    d = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    model = _model_from_golden(d, hip_device)
    x = torch.from_numpy(d["x"]).to(hip_device)
    G = torch.from_numpy(d["G"]).to(hip_device)
    if dtype == "float64":
        model, x, G = model.double(), x.double(), G.double()
    run = torch.jit.script(model) if dtype == "scripted" else model
    x.requires_grad_(True)
    out = run(x)
    (F,) = torch.autograd.grad((out * G).sum(), x, create_graph=True)
    assert F.requires_grad
    L = (F * F).sum()
    L.backward()
    torch.cuda.synchronize()
    scale_f = max(1e-3, float(np.abs(d["F_f64"]).max()))
    assert float(np.abs(F.detach().cpu().numpy().astype(np.float64) - d["F_f64"]).max()) <= (1e-9 if dtype == "float64" else 1e-4) * scale_f
    def close(got, key):
        ref64, ref32 = d[key + "_f64"], d[key + "_f32"]
        scale = max(1e-3, float(np.abs(ref64).max()))
        err = float(np.abs(got.astype(np.float64) - ref64).max())
        own = float(np.abs(ref32.astype(np.float64) - ref64).max())
        bound = 1e-6 * scale if dtype == "float64" else max(2e-4 * scale, 4.0 * own)
        assert err <= bound, (key, err, own, scale)
    close(x.grad.cpu().numpy(), "gx2")
    params = list(model.parameters())
    for i, p in enumerate(params):
        got = p.grad.cpu().numpy() if p.grad is not None else np.zeros(tuple(p.shape), dtype=np.float32)
        close(got, "gp2_%d" % i)


def test_small_head_behind_wave_per_frame_kernels_trains_through_the_hip_kernels(hip_device):
    """P1 (166 atoms: the features come from frames_ring_kernel, the head [16,32,8] from mlp_lane_kernel): requires_grad goes
    through the plan's own backward - the kept features, molann_mlp_bwd on the matrix cores and the wave-per-frame
    preprocessing backward - not through torch's Linear layers; gradients against torch autograd of the float64 oracle."""
    from molann_amd.ann import last_launch_info
    w = wl.get_workload("P1")
    model = wl.build_model(w, hip_device)
    n = 3000
    x = w.make_frames(n, seed=4).to(hip_device).requires_grad_(True)
    out = model(x)
    assert "frames_ring_kernel" in last_launch_info(model) and "mlp_lane_kernel" in last_launch_info(model), last_launch_info(model)
    G = torch.randn(n, w.out_dim(), generator=torch.Generator().manual_seed(1)).to(hip_device)
    (out * G).sum().backward()
    info = last_launch_info(model)
    assert "frames_group_bwd_kernel<B=8>" in info, info
    st = model._fast_state(x)
    assert st["entry"]().plan.backward_kind() == 1
    lins = [m for m in model.ann_layers if isinstance(m, torch.nn.Linear)]
    xx = x.detach().cpu().double().requires_grad_(True)
    ws = [l.weight.detach().cpu().double().requires_grad_(True) for l in lins]
    bs = [l.bias.detach().cpu().double().requires_grad_(True) for l in lins]
    feats = [(t, [a - 1 for a in atoms]) for t, atoms in w.features]
    al = [a - 1 for a in w.align]
    ref_x = mo.center_reference(torch.from_numpy(w.ref_xyz[al])).double()
    want = mo.molann_forward(xx, feats, ws, bs, w.use_angle_value, al, ref_x)
    (want * G.cpu().double()).sum().backward()
    assert float((out.detach().cpu().double() - want.detach()).abs().max()) <= 1e-5
    sx = float(xx.grad.abs().max())
    assert float((x.grad.cpu().double() - xx.grad).abs().max()) <= 2e-4 * sx
    for lin, wr, br in zip(lins, ws, bs):
        assert float((lin.weight.grad.cpu().double() - wr.grad).abs().max()) <= 2e-4 * max(1.0, float(wr.grad.abs().max()))
        assert float((lin.bias.grad.cpu().double() - br.grad).abs().max()) <= 2e-4 * max(1.0, float(br.grad.abs().max()))


@pytest.mark.parametrize("n_inp,kernel", [(90, "frames_align_bwd_regs_kernel<W=1,U=1>"), (166, "frames_align_bwd_regs_kernel<W=1,U=1>"),
                                          (700, "frames_align_bwd_regs_kernel<W=1,U=3>"), (1537, "frames_align_bwd_regs_kernel<W=2,U=4>"),
                                          (5000, "frames_align_bwd_regs_kernel<W=4,U=5>"), (7001, "frames_align_bwd_regs_kernel<W=8,U=4>"),
                                          (12288, "frames_align_bwd_regs_kernel<W=8,U=6>"), (12400, "frames_wave_bwd")])
def test_dense_alignment_gradient(n_inp, kernel, hip_device, monkeypatch):
    """AlignmentLayer.forward under autograd on frames the lane kernels do not serve: the forward through the alignment kernels,
    the gradient through frames_align_bwd_regs_kernel (the cotangent in registers; up to 12 288 atoms, the gather kernel
    beyond), against torch autograd of the float64 oracle and against the gather kernel (MOLANN_NO_DENSE_ALIGN=1); batch sizes
    around the grid's, the input never written, atoms outside the alignment set included."""
    from molann_amd.ann import last_launch_info
    rng = np.random.default_rng(n_inp)
    xyz = np.cumsum(rng.normal(size=(n_inp, 3)) * 0.9, axis=0).astype(np.float32)
    xyz -= xyz.mean(axis=0, keepdims=True)
    u = Universe(xyz)
    align = sorted(rng.choice(n_inp, size=int(rng.choice([3, 40, min(300, n_inp // 2)])), replace=False).tolist())
    layer = AlignmentLayer(u.atoms_by_number([a + 1 for a in align]), u.atoms).to(hip_device)
    ref_x = mo.center_reference(torch.from_numpy(xyz[align])).double()
    g = torch.Generator().manual_seed(n_inp)
    for n in (1, 37, 600 if n_inp <= 1537 else 90):
        x = (torch.from_numpy(xyz).unsqueeze(0) + 0.2 * torch.randn((n, n_inp, 3), generator=g)).float().contiguous()
        G = torch.randn(x.shape, generator=g)
        xg = x.to(hip_device).requires_grad_(True)
        y = layer(xg)
        fwd_info = last_launch_info(layer)
        (y * G.to(hip_device)).sum().backward()
        info = last_launch_info(layer)
        assert kernel in info, info
        if "frames_align_bwd_regs_kernel" in kernel:
            assert "frames_align_batch_kernel" in fwd_info or "frames_align_regs_kernel" in fwd_info, fwd_info
        xx = x.double().requires_grad_(True)
        want = mo.align_forward(xx, align, ref_x)
        (want * G.double()).sum().backward()
        assert float((y.detach().cpu().double() - want.detach()).abs().max()) <= 1e-5 + 1e-6 * float(want.detach().abs().max())
        scale = float(xx.grad.abs().max())
        assert float((xg.grad.cpu().double() - xx.grad).abs().max()) <= 2e-4 * scale, (n, info)
        assert torch.equal(xg.detach().cpu(), x)
    if "frames_align_bwd_regs_kernel" in kernel:
        monkeypatch.setenv("MOLANN_NO_DENSE_ALIGN", "1")
        layer2 = AlignmentLayer(u.atoms_by_number([a + 1 for a in align]), u.atoms).to(hip_device)
        x2 = x.to(hip_device).requires_grad_(True)
        (layer2(x2) * G.to(hip_device)).sum().backward()
        assert "frames_align_bwd_regs_kernel" not in last_launch_info(layer2)
        assert float((x2.grad - xg.grad).abs().max()) <= 2e-5 * scale
