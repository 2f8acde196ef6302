"""The __host__ __device__ arithmetic the kernels are built from (molann_amd/csrc/molann_math.h),
run on the CPU through the molann_selftest_* hooks and checked against the oracle / golden vectors."""

import ctypes

import numpy as np
import pytest
import torch

from golden_util import Case
from molann_amd import _capi
from oracle import molann_oracle as mo


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def _feature(type_id, uav, atoms):
    a = np.zeros(12, np.float32)
    a[:atoms.size] = atoms.reshape(-1)
    out = np.zeros(3, np.float32)
    w = _capi.lib().molann_selftest_feature(type_id, int(uav), _ptr(a), _ptr(out))
    assert w > 0
    return out[:w]


@pytest.mark.parametrize("type_id,n_atoms", [(0, 3), (1, 2), (2, 4), (3, 1)])
@pytest.mark.parametrize("uav", [False, True])
def test_feature_math_matches_oracle(type_id, n_atoms, uav):
    g = torch.Generator().manual_seed(10 * type_id + uav)
    x = torch.randn((200, n_atoms, 3), generator=g) * 1.5
    want = mo.feature_forward(x.double(), type_id, list(range(n_atoms)), uav)
    for i in range(x.shape[0]):
        got = _feature(type_id, uav, x[i].numpy())
        w = want[i].numpy()
        tol = 2e-6 if not (uav and type_id == 0) else 2e-6 + 4e-7 / max(1e-3, np.sqrt(1 - min(1.0, float(np.cos(w[0]) ** 2))))
        assert np.allclose(got, w, atol=tol, rtol=2e-6), (i, got, w)


def test_feature_math_pdb_anchors():
    pdb = np.load(Case("fmap_pdbframe_hist").__dict__ and __import__("os").path.join(
        __import__("golden_util").GOLDEN_DIR, "ala_dipeptide_pdb.npz"))["xyz"]
    assert abs(_feature(1, False, pdb[[4, 5]])[0] - 1.23003721) < 1e-6
    cs = _feature(2, False, pdb[[0, 2, 1, 3]])
    assert np.allclose(cs, [-0.50046289, 0.86575800], atol=1e-6)
    assert abs(_feature(2, True, pdb[[0, 2, 1, 3]])[0] - 2.09492970) < 1e-6


@pytest.mark.parametrize("code,fn", [
    (0, torch.tanh), (1, torch.relu), (2, torch.sigmoid), (3, lambda t: t),
    (4, torch.nn.functional.elu), (5, torch.nn.functional.silu), (6, torch.nn.functional.softplus),
    (7, torch.nn.functional.leaky_relu), (8, torch.nn.functional.gelu)])
def test_activations_match_torch(code, fn):
    xs = torch.cat([torch.linspace(-12, 12, 2001), torch.tensor([0.0, 1e-8, -1e-8, 0.124, 0.126, -0.125, 30.0, -30.0, 88.0, -88.0])])
    want = fn(xs.double()).numpy()
    L = _capi.lib()
    got = np.array([L.molann_selftest_activation(code, float(v)) for v in xs.tolist()])
    assert np.allclose(got, want, atol=3e-7, rtol=3e-7), np.abs(got - want).max()


def _rotation_from_hook(P, ref):
    """P, ref: centred [a,3] float arrays -> R via the host build of kabsch_rotation."""
    # as in the kernels: fp32 coordinates, covariance accumulated in fp64
    H = np.ascontiguousarray((P.astype(np.float32).astype(np.float64).T @ ref.astype(np.float32).astype(np.float64)).reshape(9))
    e0 = float(0.5 * ((P * P).sum() + (ref * ref).sum()) * 1.0001)
    R = np.zeros(9, np.float32)
    assert _capi.lib().molann_selftest_kabsch_rotation(_ptr(H), e0, _ptr(R)) == 0
    return R.reshape(3, 3)


@pytest.mark.parametrize("name", ["align_125_centred", "align_backbone_centred", "align_backbone_rigid",
                                  "align_backbone_far", "align_125_rigid", "align_sidechain_reflect",
                                  "align_all22_rigid", "align_pdbframe", "align_subset_input"])
def test_kabsch_rotation_on_golden_align_cases(name):
    """Rotation from the quaternion/Newton solver reproduces the reference's aligned frames."""
    c = Case(name)
    x = c.x.double().numpy()
    ref = c.ref_x.double().numpy()
    worst = 0.0
    for i in range(x.shape[0]):
        sel = x[i][c.align_local]
        cen = sel.mean(0)
        R = _rotation_from_hook((sel - cen).astype(np.float32), ref.astype(np.float32)).astype(np.float64)
        got = (x[i] - cen) @ R
        worst = max(worst, np.abs(got - c.out_f64[i].numpy()).max())
    own = float((c.out_f32.double() - c.out_f64).abs().max())
    assert worst <= max(2e-6, own), (worst, own)   # at least as close to fp64 truth as the fp32 reference


def test_kabsch_rotation_degenerate_inputs_are_finite():
    L = _capi.lib()
    for H in (np.zeros(9), np.full(9, np.nan),
              np.array([1., 0, 0, 0, 0, 0, 0, 0, 0]),            # rank 1
              np.array([1e-30, 0, 0, 0, 1e-30, 0, 0, 0, 1e-30])):
        R = np.zeros(9, np.float32)
        assert L.molann_selftest_kabsch_rotation(_ptr(H), 1.0, _ptr(R)) == 0
        assert np.isfinite(R).all()
        assert abs(np.linalg.det(R.reshape(3, 3).astype(np.float64)) - 1.0) < 1e-5


def test_kabsch_rotation_f32_instantiation():
    """The fp32 instantiation (plans whose items are all invariant): always a proper rotation, and on
    well-conditioned covariances within fp32 rounding (times the conditioning) of the fp64 one."""
    L = _capi.lib()
    g = np.random.default_rng(11)
    worst = 0.0
    for i in range(400):
        a = int(g.integers(3, 9))
        ref = g.normal(size=(a, 3)) * 2
        ref -= ref.mean(0)
        q = g.normal(size=4); q /= np.linalg.norm(q)
        w, x, y, z = q
        Q = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        P = ref @ Q + 0.2 * g.normal(size=(a, 3))
        P -= P.mean(0)
        H = np.ascontiguousarray((P.T @ ref).reshape(9))
        e0 = 0.5 * ((P * P).sum() + (ref * ref).sum()) * 1.0001
        R64, R32 = np.zeros(9, np.float32), np.zeros(9, np.float32)
        assert L.molann_selftest_kabsch_rotation(_ptr(H), e0, _ptr(R64)) == 0
        H32 = H.astype(np.float32)
        assert L.molann_selftest_kabsch_rotation_f32(_ptr(H32), np.float32(e0), _ptr(R32)) == 0
        M = R32.reshape(3, 3).astype(np.float64)
        assert np.abs(M @ M.T - np.eye(3)).max() < 2e-6 and abs(np.linalg.det(M) - 1.0) < 2e-6
        sv = np.linalg.svd(H.reshape(3, 3), compute_uv=False)
        d = np.sign(np.linalg.det(H.reshape(3, 3)))
        if (sv[1] + d * sv[2]) / sv[0] > 0.2:               # well-defined rotation
            worst = max(worst, np.abs(R32 - R64).max())
    assert worst < 2e-5, worst
    for H in (np.zeros(9, np.float32), np.full(9, np.nan, np.float32), np.array([1, 0, 0, 0, 0, 0, 0, 0, 0], np.float32)):
        R = np.zeros(9, np.float32)
        assert L.molann_selftest_kabsch_rotation_f32(_ptr(H), np.float32(1.0), _ptr(R)) == 0
        assert np.isfinite(R).all() and abs(np.linalg.det(R.reshape(3, 3).astype(np.float64)) - 1.0) < 1e-5


# ---- reverse mode: the analytic gradients against torch autograd of the oracle (fp64) ----------------------
@pytest.mark.parametrize("type_id,n_atoms", [(0, 3), (1, 2), (2, 4), (3, 1)])
@pytest.mark.parametrize("uav", [False, True])
def test_feature_backward_matches_autograd(type_id, n_atoms, uav):
    g = torch.Generator().manual_seed(100 + 10 * type_id + uav)
    L = _capi.lib()
    for i in range(100):
        x = (torch.randn((1, n_atoms, 3), generator=g) * 1.5).double().requires_grad_(True)
        f = mo.feature_forward(x, type_id, list(range(n_atoms)), uav)
        gf = torch.randn(f.shape, generator=g).double()
        (f * gf).sum().backward()
        want = x.grad[0].numpy()
        a = np.zeros(12, np.float32); a[:3 * n_atoms] = x.detach().numpy().reshape(-1)
        g3 = np.zeros(3, np.float32); g3[:gf.numel()] = gf.numpy().reshape(-1)
        ga = np.zeros(12, np.float32)
        assert L.molann_selftest_feature_backward(type_id, int(uav), _ptr(a), _ptr(g3), _ptr(ga)) == 0
        got = ga.reshape(4, 3)[:n_atoms]
        scale = max(1.0, float(np.abs(want).max()))
        assert np.allclose(got, want, atol=2e-5 * scale, rtol=2e-5), (i, got, want)


def test_kabsch_backward_matches_autograd():
    """G_H from the closed form equals autograd through the SVD formula (ann.py:188-195) in fp64."""
    g = torch.Generator().manual_seed(5)
    L = _capi.lib()
    for i in range(200):
        a = int(torch.randint(3, 9, (1,), generator=g))
        ref = torch.randn((a, 3), generator=g).double() * 2
        ref = ref - ref.mean(0)
        P = (ref @ wl_q(torch.randn((1, 4), generator=g))[0].double() + 0.2 * torch.randn((a, 3), generator=g).double())
        if i % 4 == 1:
            P[:, 2] *= -1                      # reflection branch
        P = P - P.mean(0)
        H = (P.T @ ref).requires_grad_(True)
        u, s, vh = torch.linalg.svd(H)
        d = torch.sign(torch.linalg.det(u @ vh)).detach()
        R = u @ torch.diag(torch.stack([torch.ones(()).double(), torch.ones(()).double(), d])) @ vh
        GR = torch.randn((3, 3), generator=g).double()
        (R * GR).sum().backward()
        sv = s.detach().numpy()
        if (sv[1] + float(d) * sv[2]) / sv[0] < 5e-2:
            continue                           # nearly ill-defined rotation: derivative blows up, not compared
        Hn = np.ascontiguousarray(H.detach().numpy().reshape(9))
        Rn = np.ascontiguousarray(R.detach().numpy().reshape(9).astype(np.float32))
        GRn = np.ascontiguousarray(GR.numpy().reshape(9).astype(np.float32))
        GH = np.zeros(9, np.float32)
        assert L.molann_selftest_kabsch_backward(_ptr(Hn), _ptr(Rn), _ptr(GRn), _ptr(GH)) == 0
        want = H.grad.numpy().reshape(9)
        assert np.allclose(GH, want, atol=1e-5 * max(1.0, np.abs(want).max()), rtol=1e-4), (i, GH, want)


def wl_q(q):
    from molann_amd import workloads as wl
    return wl.quaternion_to_matrix(q / q.norm(dim=1, keepdim=True))


@pytest.mark.parametrize("code,fn", [(0, torch.tanh), (1, torch.relu), (2, torch.sigmoid), (5, torch.nn.functional.silu),
                                     (7, torch.nn.functional.leaky_relu)])
def test_activation_derivatives(code, fn):
    zs = torch.linspace(-6, 6, 241).double()
    zs = zs[zs.abs() > 1e-3].requires_grad_(True)
    fn(zs).sum().backward()
    L = _capi.lib()
    got = np.array([L.molann_selftest_act_derivative(code, float(v)) for v in zs.detach().tolist()])
    assert np.allclose(got, zs.grad.numpy(), atol=2e-6)
