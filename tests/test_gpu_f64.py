"""`model.double()(x.double())` (the reference's modules follow x.dtype, ann.py:187-197): the float64
kernels against the output of the REFERENCE's own `.double()` run stored in every golden fixture (`out_f64`), to 1e-10
relative to the output's scale, and the dtype / device errors the reference raises for mixed inputs."""

import pytest
import torch

from build_util import build_modules
from golden_util import Case, case_names

pytestmark = pytest.mark.gpu
ALL = [n for n in case_names() if not n.endswith("bf16w")]


@pytest.mark.parametrize("name", ALL)
def test_float64_forward_matches_reference_float64(name, hip_device):
    c = Case(name)
    model = build_modules(c, hip_device).double()
    x = c.x.double().to(hip_device)
    with torch.no_grad():
        got = model(x)
    torch.cuda.synchronize()
    assert got.dtype == torch.float64 and got.shape == c.out_f64.shape
    if got.numel() == 0:
        return
    scale = max(1.0, float(c.out_f64.abs().max()))
    err = float((got.cpu() - c.out_f64).abs().max())
    assert err <= 1e-10 * scale, (err, scale)


def test_float64_and_float32_share_a_model(hip_device):
    """.double() and back: the plan is the same one, the live tensors are re-read in their dtype."""
    c = Case("molann_C3")
    model = build_modules(c, hip_device)
    x = c.x.to(hip_device)
    with torch.no_grad():
        y32 = model(x)
        y64 = model.double()(x.double())
        y32b = model.float()(x)
    assert torch.equal(y32, y32b)
    assert float((y64.cpu() - c.out_f64).abs().max()) <= 1e-10
    assert float((y32.cpu().double() - y64.cpu()).abs().max()) <= 1e-5


def test_mixed_dtypes_raise_like_the_reference(hip_device):
    c = Case("molann_C3")
    model = build_modules(c, hip_device)
    with torch.no_grad():
        with pytest.raises(RuntimeError):
            model(c.x.double().to(hip_device))           # float32 module, float64 input
        model = model.double()
        with pytest.raises(RuntimeError):
            model(c.x.to(hip_device))                    # float64 module, float32 input
    with pytest.raises(NotImplementedError):
        model(c.x.double().to(hip_device).requires_grad_(True))   # float64 is forward only


@pytest.mark.parametrize("name", ["align_backbone_rigid", "features_C3p", "molann_C3", "molann_C4_small"])
def test_scripted_float64(name, hip_device):
    """torch.jit.script(model.double()) runs the same float64 kernels through molann::run."""
    c = Case(name)
    model = build_modules(c, hip_device).double()
    scripted = torch.jit.script(model)
    x = c.x.double().to(hip_device)
    with torch.no_grad():
        got = scripted(x)
        eager = model(x)
    assert torch.equal(got, eager)
    scale = max(1.0, float(c.out_f64.abs().max()))
    assert float((got.cpu() - c.out_f64).abs().max()) <= 1e-10 * scale
    with pytest.raises(RuntimeError):
        scripted(x.clone().requires_grad_(True))
