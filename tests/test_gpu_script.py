"""TorchScript export on the GPU: the scripted / saved / reloaded module runs the same plans as the eager
module (bit-identical outputs), matches the reference's golden outputs, re-reads live tensors and is
differentiable through molann::run's autograd kernel."""

import io

import pytest
import torch

from build_util import build_modules, oracle_for_workload, workload_model
from golden_util import Case
from molann_amd import workloads as wl

pytestmark = pytest.mark.gpu
F32_TOL = 1e-5


def _reload(model, device):
    buf = io.BytesIO()
    torch.jit.save(torch.jit.script(model), buf)
    buf.seek(0)
    return torch.jit.load(buf, map_location=device)


@pytest.mark.parametrize("name", ["align_backbone_rigid", "align_subset_input", "features_C2", "features_C3p", "features_C3_val",
                                  "flayer_test_mixed", "pp_align123_dihedral", "pp_noalign_pos12", "molann_test",
                                  "molann_C1", "molann_C1s", "molann_C3", "molann_C3_relu", "molann_C4_small", "molann_C5_small"])
def test_scripted_matches_golden_and_eager(name, hip_device):
    c = Case(name)
    model = build_modules(c, hip_device)
    loaded = _reload(model, hip_device)
    x = c.x.to(hip_device)
    with torch.no_grad():
        eager = model(x)
        got = loaded(x)
    torch.cuda.synchronize()
    assert torch.equal(got, eager)                         # same plan, same kernels
    assert float((got.cpu() - c.out_f32).abs().max()) <= c.tolerance_vs_f32()


def test_scripted_module_moves_with_to_and_rereads_live_tensors(hip_device):
    w = wl.get_workload("C3")
    model = workload_model(w, torch.device("cpu"))
    loaded = _reload(model, "cpu").to(hip_device)          # scripted on the CPU, moved afterwards
    model = model.to(hip_device)
    x = w.make_frames(777, seed=5).to(hip_device)
    with torch.no_grad():
        assert torch.equal(loaded(x), model(x))
        for p in loaded.parameters():                      # in-place update: version counter changes, storage does not
            p.mul_(0.5)
        for p in model.parameters():
            p.mul_(0.5)
        loaded.ref_x.copy_(torch.roll(loaded.ref_x, 1, 0))
        model.preprocessing_layer.align_layer.ref_x.copy_(loaded.ref_x)
        y = loaded(x)
        assert torch.equal(y, model(x))
    want = oracle_for_workload(w, model, x, torch.float64)
    assert float((y.cpu().double() - want).abs().max()) <= F32_TOL


def test_scripted_empty_batch_and_bad_shapes(hip_device):
    w = wl.get_workload("C1")
    loaded = _reload(workload_model(w, hip_device), hip_device)
    assert loaded(torch.zeros(0, 22, 3, device=hip_device)).shape == (0, 3)
    with pytest.raises(RuntimeError, match="Input should be a 3d torch tensor"):
        loaded(torch.zeros(4, 21, 3, device=hip_device))
    with pytest.raises(RuntimeError, match="float64|same dtype"):     # a float32 model on a float64 input: mixed dtypes, as in the
        loaded(torch.zeros(4, 22, 3, device=hip_device, dtype=torch.float64))   # reference (under grad mode torch's own message)
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="float64"):
            loaded(torch.zeros(4, 22, 3, device=hip_device, dtype=torch.float64))


@pytest.mark.parametrize("cfg", ["C1", "C3", "C2", "A3"])
def test_scripted_autograd_matches_eager_autograd(cfg, hip_device):
    """forces from a scripted collective variable: d(sum w*y)/dx and parameter gradients through molann::run."""
    w = wl.get_workload(cfg)
    model = workload_model(w, hip_device)
    scripted = torch.jit.script(model)                     # shares the parameters with `model`
    x = w.make_frames(333, seed=11).to(hip_device)
    xa = x.clone().requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    ya = model(xa)
    yb = scripted(xb)
    assert yb.requires_grad and torch.equal(ya.detach(), yb.detach())
    wgt = torch.randn(ya.shape, device=hip_device, generator=torch.Generator(device=hip_device).manual_seed(3))
    params = [p for p in model.parameters()]
    ga = torch.autograd.grad((ya * wgt).sum(), [xa] + params)
    gb = torch.autograd.grad((yb * wgt).sum(), [xb] + params)
    for a, b in zip(ga, gb):
        scale = max(1.0, float(a.abs().max()))
        assert float((a - b).abs().max()) <= 1e-5 * scale  # same kernel; float atomics reorder the parameter sums


def test_scripted_grad_x_only_under_frozen_parameters(hip_device):
    w = wl.get_workload("C3")
    model = workload_model(w, hip_device)
    for p in model.parameters():
        p.requires_grad_(False)
    loaded = _reload(model, hip_device)
    x = w.make_frames(200, seed=2).to(hip_device).requires_grad_(True)
    y = loaded(x)
    (gx,) = torch.autograd.grad(y[:, 0].sum(), [x])
    xe = x.detach().clone().requires_grad_(True)
    (ge,) = torch.autograd.grad(model(xe)[:, 0].sum(), [xe])
    assert torch.equal(gx, ge)


def test_scripted_large_frame_model_is_differentiable(hip_device):
    """A fused plan without a backward kernel (5000-atom frames): the operator composes the feature plan's
    backward with the MLP as ATen ops, as the eager module does."""
    w = wl.get_workload("C4")
    model = workload_model(w, hip_device)
    scripted = torch.jit.script(model)
    x = w.make_frames(5, seed=2).to(hip_device)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = model(xa), scripted(xb)
    assert float((ya - yb).detach().abs().max()) <= 1e-5 * max(1.0, float(ya.detach().abs().max()))
    params = list(model.parameters())
    ga = torch.autograd.grad(ya.sum(), [xa] + params)
    gb = torch.autograd.grad(yb.sum(), [xb] + params)
    for a, b in zip(ga, gb):
        assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max()))
    with torch.no_grad():                                     # inference keeps the fused two-kernel path
        assert torch.equal(scripted(x), model(x))
