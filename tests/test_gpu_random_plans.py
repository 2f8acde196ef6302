"""Randomly drawn plans (input group size, touched atoms, feature mix, alignment set, MLP) against the fp64 oracle:
exercises the plan-specialised kernel's compact staging (which 16-byte windows of a frame are copied, windows
clamped at the end of the frame, odd window counts, tiny frames) and its generic fallback on the same plans."""

import numpy as np
import pytest
import torch

from molann_amd import workloads as wl
from molann_amd.ann import (AlignmentLayer, FeatureLayer, MolANN, PreprocessingANN, create_sequential_nn,
                            last_launch_info)
from molann_amd.atomgroup import Universe
from molann_amd.feature import Feature
from oracle import molann_oracle as mo

pytestmark = pytest.mark.gpu
NEED = {0: 3, 1: 2, 2: 4, 3: None}     # atoms per feature type (position: any number)


def _draw(rng, case):
    n_inp = int(rng.choice([1, 2, 3, 4, 5, 7, 12, 22, 31, 40]))
    xyz = (rng.normal(size=(n_inp, 3)) * 2.5).astype(np.float32)
    u = Universe(xyz)
    feats, spec = [], []
    for i in range(int(rng.integers(1, 7))):
        t = int(rng.choice([0, 1, 2, 3]))
        k = NEED[t] if NEED[t] is not None else int(rng.integers(1, min(n_inp, 4) + 1))
        if k > n_inp:
            t, k = (1, 2) if n_inp >= 2 else (3, 1)
        # bias towards the ends of the frame (first / last atoms: clamped windows)
        pool = np.arange(n_inp)
        w = np.ones(n_inp); w[:2] += 2; w[-2:] += 3
        atoms = rng.choice(pool, size=k, replace=False, p=w / w.sum()).tolist()
        feats.append(Feature("f%d" % i, wl.TYPE_NAMES[t], u.atoms_by_number([a + 1 for a in atoms])))
        spec.append((t, atoms))
    uav = bool(rng.integers(0, 2))
    align = None
    if n_inp >= 3 and case % 3 != 0:
        align = sorted(rng.choice(np.arange(n_inp), size=int(rng.integers(3, min(n_inp, 8) + 1)), replace=False).tolist())
    al = AlignmentLayer(u.atoms_by_number([a + 1 for a in align]), u.atoms) if align is not None else None
    pp = PreprocessingANN(al, FeatureLayer(feats, u.atoms, uav))
    model = pp
    if case % 2 == 0:
        d = pp.output_dimension()
        dims = [d] + [int(v) for v in rng.integers(2, 33, size=int(rng.integers(1, 3)))]
        torch.manual_seed(case)
        model = MolANN(pp, create_sequential_nn(dims))
    return xyz, spec, uav, align, model


def _oracle(model, x, spec, uav, align, xyz):
    ref_x = mo.center_reference(torch.from_numpy(xyz[align])).double() if align is not None else None
    xd = x.double()
    if isinstance(model, MolANN):
        lins = [m for m in model.ann_layers if isinstance(m, torch.nn.Linear)]
        return mo.molann_forward(xd, spec, [l.weight.detach().cpu().double() for l in lins],
                                 [l.bias.detach().cpu().double() for l in lins], uav, align, ref_x)
    return mo.preprocessing_forward(xd, spec, uav, align, ref_x)


@pytest.mark.parametrize("nojit", ["0", "1"], ids=["specialised", "generic"])
def test_random_plans_match_oracle(nojit, hip_device, monkeypatch):
    monkeypatch.setenv("MOLANN_NO_JIT", nojit)
    rng = np.random.default_rng(2024)
    kinds = set()
    for case in range(60):
        xyz, spec, uav, align, model = _draw(rng, case)
        model = model.to(hip_device)
        n = int(rng.choice([1, 63, 64, 130, 777]))
        g = torch.Generator().manual_seed(case)
        x = torch.from_numpy(xyz).unsqueeze(0) + 0.3 * torch.randn((n, xyz.shape[0], 3), generator=g)
        if align is not None:                                        # random rigid motion per frame
            q = torch.randn((n, 4), generator=g)
            q = q / q.norm(dim=1, keepdim=True)
            x = torch.einsum("nij,nkj->nki", wl.quaternion_to_matrix(q), x) + 2.0 * torch.randn((n, 1, 3), generator=g)
        x = x.float().contiguous()
        with torch.no_grad():
            got = model(x.to(hip_device)).cpu()
        want = _oracle(model, x, spec, uav, align, xyz)
        info = last_launch_info(model if isinstance(model, MolANN) else model)
        kinds.add(info.split("<")[0].split(" ")[0])
        assert got.shape == want.shape, (case, info)
        err = float((got.double() - want).abs().max())
        # ill-conditioned alignment sets (3 random atoms may be nearly collinear) amplify fp32 rounding of
        # position features; invariant features do not see the rotation at all
        has_pos = any(t == 3 for t, _ in spec)
        assert err <= (2e-4 if (has_pos and align is not None) else 2e-5), (case, err, info, spec, align)
    assert ("molann_lane_jit" in kinds) == (nojit == "0"), kinds   # (a 1-atom frame has no 16-byte window: generic kernel)


def test_random_plans_gradients(hip_device):
    """The plan-specialised backward kernel on drawn plans: dL/dx (and parameter gradients) against autograd
    through the fp64 oracle."""
    rng = np.random.default_rng(77)
    checked = 0
    for case in range(40):
        xyz, spec, uav, align, model = _draw(rng, case)
        if xyz.shape[0] < 2:
            continue
        if align is not None:
            if len(align) < 4:
                continue
            ref = xyz[align] - xyz[align].mean(0)
            sv = np.linalg.svd(ref.T @ ref, compute_uv=False)
            if sv[2] / sv[0] < 0.05:                                  # nearly planar reference: derivative ill-conditioned
                continue
        model = model.to(hip_device)
        n = int(rng.choice([1, 65, 300]))
        g = torch.Generator().manual_seed(1000 + case)
        x = (torch.from_numpy(xyz).unsqueeze(0) + 0.15 * torch.randn((n, xyz.shape[0], 3), generator=g)).float().contiguous()
        xg = x.to(hip_device).requires_grad_(True)
        y = model(xg)
        G = torch.randn(y.shape, generator=g)
        (y * G.to(hip_device)).sum().backward()
        xx = x.double().requires_grad_(True)
        ref_x = mo.center_reference(torch.from_numpy(xyz[align])).double() if align is not None else None
        params = []
        if isinstance(model, MolANN):
            lins = [m for m in model.ann_layers if isinstance(m, torch.nn.Linear)]
            ws = [l.weight.detach().cpu().double().requires_grad_(True) for l in lins]
            bs = [l.bias.detach().cpu().double().requires_grad_(True) for l in lins]
            want = mo.molann_forward(xx, spec, ws, bs, uav, align, ref_x)
            params = [(l.weight.grad, w.grad) for l, w in zip(lins, ws)] + [(l.bias.grad, b.grad) for l, b in zip(lins, bs)]
        else:
            want = mo.preprocessing_forward(xx, spec, uav, align, ref_x)
        (want * G.double()).sum().backward()
        if isinstance(model, MolANN):
            params = [(l.weight.grad, w.grad) for l, w in zip(lins, ws)] + [(l.bias.grad, b.grad) for l, b in zip(lins, bs)]
        # angle / dihedral VALUES have unbounded derivatives at the poles: skip frames sitting on one
        if not torch.isfinite(xx.grad).all() or float(xx.grad.abs().max()) > 1e4:
            continue
        for got, ref in [(xg.grad, xx.grad)] + params:
            scale = max(1e-6, float(ref.abs().max()))
            assert float((got.cpu().double() - ref).abs().max()) <= 5e-4 * scale, (case, spec, align, uav)
        checked += 1
    assert checked >= 15, checked


@pytest.mark.parametrize("n_inp", [48, 64, 85, 86, 100, 150, 213, 300, 1000])
def test_frame_sizes_across_the_lane_wave_boundary(n_inp, hip_device):
    """Input groups around the size where plans move from the lane-per-frame to the wave-per-frame kernel."""
    rng = np.random.default_rng(n_inp)
    xyz = np.cumsum(rng.normal(size=(n_inp, 3)) * 0.9, axis=0).astype(np.float32)
    u = Universe(xyz)
    spec = [(2, [0, 1, 2, 3]), (1, [n_inp - 2, n_inp - 1]), (0, [n_inp // 2, n_inp // 2 + 1, n_inp // 2 + 3]),
            (3, [n_inp - 1, 0]), (2, [n_inp - 4, n_inp - 3, n_inp - 2, n_inp - 1])]
    feats = [Feature("f%d" % i, wl.TYPE_NAMES[t], u.atoms_by_number([a + 1 for a in atoms])) for i, (t, atoms) in enumerate(spec)]
    align = sorted(rng.choice(np.arange(n_inp), size=min(n_inp, 24), replace=False).tolist())
    pp = PreprocessingANN(AlignmentLayer(u.atoms_by_number([a + 1 for a in align]), u.atoms), FeatureLayer(feats, u.atoms, False))
    torch.manual_seed(n_inp)
    model = MolANN(pp, create_sequential_nn([pp.output_dimension(), 16, 4])).to(hip_device)
    n = 193
    g = torch.Generator().manual_seed(n_inp)
    x = torch.from_numpy(xyz).unsqueeze(0) + 0.2 * torch.randn((n, n_inp, 3), generator=g)
    q = torch.randn((n, 4), generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    x = (torch.einsum("nij,nkj->nki", wl.quaternion_to_matrix(q), x) + torch.randn((n, 1, 3), generator=g)).float().contiguous()
    with torch.no_grad():
        got = model(x.to(hip_device)).cpu()
        aligned = pp.align_layer(x.to(hip_device)).cpu()
    want = _oracle(model, x, spec, False, align, xyz)
    assert float((got.double() - want).abs().max()) <= 1e-4, last_launch_info(model)   # position items behind a 24-atom fit
    ref_x = mo.center_reference(torch.from_numpy(xyz[align])).double()
    assert float((aligned.double() - mo.align_forward(x.double(), align, ref_x)).abs().max()) <= 1e-4


@pytest.mark.parametrize("n_inp", [100, 1000, 5000])
def test_few_touched_atoms_of_a_large_frame_take_the_lane_kernel(n_inp, hip_device, monkeypatch):
    """The specialised lane kernel stages only touched windows, so its tile does not grow with the frame: a plan
    touching <= 32 atoms of a 5000-atom frame is a lane-per-frame plan (and a wave-per-frame one without hipRTC)."""
    rng = np.random.default_rng(n_inp)
    xyz = np.cumsum(rng.normal(size=(n_inp, 3)) * 0.9, axis=0).astype(np.float32)
    u = Universe(xyz)
    spec = [(2, [0, 1, 2, 3]), (1, [n_inp - 2, n_inp - 1]), (0, [n_inp // 2, n_inp // 2 + 1, n_inp // 2 + 3]),
            (3, [n_inp - 1, 7]), (2, [n_inp - 4, n_inp - 3, n_inp - 2, n_inp - 1])]
    feats = [Feature("f%d" % i, wl.TYPE_NAMES[t], u.atoms_by_number([a + 1 for a in atoms])) for i, (t, atoms) in enumerate(spec)]
    align = sorted(rng.choice(np.arange(n_inp), size=8, replace=False).tolist())
    n = 300
    g = torch.Generator().manual_seed(n_inp)
    x = (torch.from_numpy(xyz).unsqueeze(0) + 0.2 * torch.randn((n, n_inp, 3), generator=g)).float().contiguous()
    outs = {}
    for nojit in ("0", "1"):
        monkeypatch.setenv("MOLANN_NO_JIT", nojit)
        pp = PreprocessingANN(AlignmentLayer(u.atoms_by_number([a + 1 for a in align]), u.atoms), FeatureLayer(feats, u.atoms, False))
        torch.manual_seed(n_inp)
        model = MolANN(pp, create_sequential_nn([pp.output_dimension(), 16, 4])).to(hip_device)
        with torch.no_grad():
            outs[nojit] = model(x.to(hip_device)).cpu()
        info = last_launch_info(model)
        assert ("molann_lane_jit" in info) == (nojit == "0"), info
        want = _oracle(model, x, spec, False, align, xyz)
        assert float((outs[nojit].double() - want).abs().max()) <= 1e-4, info
    # gradients w.r.t. x: the wave-per-frame backward serves the feature plan, the MLP is torch's (composition)
    monkeypatch.setenv("MOLANN_NO_JIT", "0")
    pp = PreprocessingANN(AlignmentLayer(u.atoms_by_number([a + 1 for a in align]), u.atoms), FeatureLayer(feats, u.atoms, False))
    torch.manual_seed(n_inp)
    model = MolANN(pp, create_sequential_nn([pp.output_dimension(), 16, 4])).to(hip_device)
    xg = x[:40].to(hip_device).requires_grad_(True)
    model(xg).sum().backward()
    xx = x[:40].double().requires_grad_(True)
    _oracle(model, xx, spec, False, align, xyz).sum().backward()
    scale = float(xx.grad.abs().max())
    assert float((xg.grad.cpu().double() - xx.grad).abs().max()) <= 5e-4 * scale


@pytest.mark.parametrize("n_feat", [33, 40, 49])
def test_mlp_behind_33_to_64_features_is_fused(n_feat, hip_device):
    """Feature dimensions 33..64 in front of a narrow MLP: 16 k-steps in layer 0 of the specialised kernel's MFMA
    MLP (the ahead-of-time kernel stops at 32 and leaves such an MLP to mlp_mfma_kernel)."""
    import itertools
    u = Universe(wl.ALA_DIPEPTIDE_XYZ)
    heavy = [2, 5, 6, 7, 9, 11, 15, 16, 17, 19]
    pairs = list(itertools.combinations(heavy, 2))[:45]
    dihs = [(5, 7, 9, 15), (7, 9, 15, 17), (2, 5, 7, 9), (9, 15, 17, 19)]
    spec = [(1, [a - 1 for a in p]) for p in pairs] + [(2, [a - 1 for a in d]) for d in dihs]
    spec = spec[:n_feat] if n_feat <= 45 else spec
    feats = [Feature("f%d" % i, wl.TYPE_NAMES[t], u.atoms_by_number([a + 1 for a in atoms])) for i, (t, atoms) in enumerate(spec)]
    align = [1, 4, 6, 8, 14, 16, 18]
    pp = PreprocessingANN(AlignmentLayer(u.atoms_by_number([a + 1 for a in align]), u.atoms), FeatureLayer(feats, u.atoms, False))
    torch.manual_seed(n_feat)
    model = MolANN(pp, create_sequential_nn([pp.output_dimension(), 30, 30, 2])).to(hip_device)
    w = wl.get_workload("C3")
    x = w.make_frames(1000, seed=n_feat)
    with torch.no_grad():
        got = model(x.to(hip_device)).cpu()
    info = last_launch_info(model)
    assert "molann_lane_jit<NL=3>" in info, info
    want = _oracle(model, x, spec, False, align, wl.ALA_DIPEPTIDE_XYZ.astype(np.float32))
    assert float((got.double() - want).abs().max()) <= 2e-5, info
    xg = x[:100].to(hip_device).requires_grad_(True)            # no fused backward at this width: composition
    model(xg).sum().backward()
    xx = x[:100].double().requires_grad_(True)
    _oracle(model, xx, spec, False, align, wl.ALA_DIPEPTIDE_XYZ.astype(np.float32)).sum().backward()
    assert float((xg.grad.cpu().double() - xx.grad).abs().max()) <= 5e-4 * float(xx.grad.abs().max())


@pytest.mark.parametrize("seed", range(6))
def test_random_large_frame_plans_through_the_ring_kernel(seed, hip_device, monkeypatch):
    """frames_ring_kernel (large frames: loader / consumer waves around an LDS ring of per-frame window images):
    random frame sizes, touched-atom counts from a handful to > 2048 windows (where the plan falls back to
    frames_wave_kernel), alignment sets, item mixes incl. positions at both ends of the frame, and batch sizes around
    the ring's and the grid's sizes (1 frame, fewer frames than CUs, many frames per block) - against the fp64 oracle,
    and against frames_wave_kernel on the same plan (bit for bit with one frame per ring entry, within 2e-6 with several)."""
    rng = np.random.default_rng(1000 + seed)
    n_inp = int(rng.choice([120, 333, 1000, 2500, 5000]))
    xyz = np.cumsum(rng.normal(size=(n_inp, 3)) * 0.9, axis=0).astype(np.float32)
    u = Universe(xyz)
    n_feat = int(rng.choice([3, 40, 150, 700]))
    spec, feats = [(3, [n_inp - 1, 0])], []
    for i in range(n_feat):
        t = int(rng.choice([0, 1, 2]))
        k = NEED[t]
        s0 = int(rng.integers(0, n_inp - k + 1))
        atoms = list(range(s0, s0 + k)) if rng.random() < 0.7 else sorted(rng.choice(n_inp, size=k, replace=False).tolist())
        spec.append((t, atoms))
    feats = [Feature("f%d" % i, wl.TYPE_NAMES[t], u.atoms_by_number([a + 1 for a in atoms])) for i, (t, atoms) in enumerate(spec)]
    uav = bool(rng.integers(0, 2))
    align = sorted(rng.choice(np.arange(n_inp), size=int(rng.choice([3, 40, 300 if n_inp >= 333 else 50])), replace=False).tolist()) \
        if seed % 3 != 2 else None
    al = AlignmentLayer(u.atoms_by_number([a + 1 for a in align]), u.atoms) if align is not None else None
    pp = PreprocessingANN(al, FeatureLayer(feats, u.atoms, uav)).to(hip_device)
    g = torch.Generator().manual_seed(seed)
    for n in (1, 37, 300, 2111):
        x = (torch.from_numpy(xyz).unsqueeze(0) + 0.2 * torch.randn((n, n_inp, 3), generator=g)).float().contiguous()
        with torch.no_grad():
            got = pp(x.to(hip_device)).cpu()
        info = last_launch_info(pp)
        want = _oracle(pp, x, spec, uav, align, xyz)
        scale = max(1.0, float(want.abs().max()))
        assert float((got.double() - want).abs().max()) <= 2e-5 * scale, (n, info)
    assert "frames_ring_kernel" in info or "frames_wave_kernel" in info or "molann_lane_jit" in info, info
    if "frames_ring_kernel" in info:
        monkeypatch.setenv("MOLANN_NO_RING", "1")
        with torch.no_grad():
            old = pp(x.to(hip_device)).cpu()
        assert "frames_wave_kernel" in last_launch_info(pp)
        if "B=" in info:                      # several frames per entry: the covariance sums are taken in another order
            assert float((old - got).abs().max()) <= 2e-6 * scale, info
        else:
            assert torch.equal(old, got)      # the same per-frame arithmetic on the same values
