"""The plan-specialised wide bf16 MLP (csrc/molann_mlp_jit.inc: activations chained through MFMA accumulators,
weights streamed through LDS slabs) against a CPU emulation of the same arithmetic model - bf16 weights,
bf16 activations between layers, fp32 accumulation - and against the generic bf16 kernel it replaces."""

import pytest
import torch

from molann_amd import _capi

pytestmark = pytest.mark.gpu

ACTS = {_capi.ACT_TANH: torch.tanh, _capi.ACT_RELU: torch.relu, _capi.ACT_SIGMOID: torch.sigmoid}


def _plan(dims, act, n_inp=128, precision=_capi.MLP_BF16):
    """A plan whose feature dimension is dims[0]: positions of dims[0]//3 atoms (+ a bond / a dihedral)."""
    k, r = divmod(dims[0], 3)
    feats = [(_capi.FEAT_POSITION, list(range(k)))] if k else []
    if r == 1:
        feats.append((_capi.FEAT_BOND, [k, k + 1]))
    elif r == 2:
        feats.append((_capi.FEAT_DIHEDRAL, [k, k + 1, k + 2, k + 3]))
    return _capi.Plan(n_inp, features=feats, layer_dims=dims, activation=act, mlp_precision=precision)


def _bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _emulate(f, ws, bs, act):
    h = _bf16(f.cpu()).double()
    for i, (w, b) in enumerate(zip(ws, bs)):
        z = (h @ _bf16(w.cpu()).double().T + b.cpu().double()).float()
        if i + 1 == len(ws):
            return z
        h = _bf16(ACTS[act](z)).double()


def _params(dims, dev, seed):
    g = torch.Generator().manual_seed(seed)
    ws = [(torch.randn(j, k, generator=g) / k ** 0.5).to(dev) for k, j in zip(dims[:-1], dims[1:])]
    bs = [(0.1 * torch.randn(j, generator=g)).to(dev) for j in dims[1:]]
    return ws, bs


@pytest.mark.parametrize("dims,act", [
    ([341, 512, 256, 16], _capi.ACT_TANH),       # C5
    ([85, 128, 64, 8], _capi.ACT_TANH),          # C4's shape on the bf16 path
    ([40, 33, 7], _capi.ACT_SIGMOID),            # nothing a multiple of 16; padded units see sigmoid(0) = 0.5
    ([70, 5], _capi.ACT_TANH),                   # one layer: a lone P
    ([64, 48, 3], _capi.ACT_RELU),               # one (P, C) pair
    ([50, 64, 32, 48, 5], _capi.ACT_TANH),       # two pairs
    ([33, 40, 24, 36, 20, 9], _capi.ACT_RELU),   # two pairs and a lone P, odd block counts
])
@pytest.mark.parametrize("n", [1, 191, 1000])
def test_chain_kernel_matches_bf16_emulation(dims, act, n, hip_device):
    plan = _plan(dims, act)
    ws, bs = _params(dims, hip_device, 7)
    plan.update_mlp(ws, bs)
    f = torch.randn(n, dims[0], device=hip_device, generator=torch.Generator(device=hip_device).manual_seed(n))
    out = torch.full((n, dims[-1]), float("nan"), device=hip_device)
    plan.mlp_packed(f, out)
    torch.cuda.synchronize()
    assert "molann_mlp_chain" in plan.last_launch_info(), plan.last_launch_info()
    want = _emulate(f, ws, bs, act)
    scale = max(1.0, float(want.abs().max()))
    err = float((out.cpu() - want).abs().max())
    # fp32 accumulation order differs and an activation that lands on a bf16 rounding boundary may flip by one ulp
    assert err <= 4e-3 * scale, (err, scale)


def test_chain_kernel_against_generic_bf16_kernel(hip_device, monkeypatch):
    dims, act = [341, 512, 256, 16], _capi.ACT_TANH
    ws, bs = _params(dims, hip_device, 3)
    f = torch.randn(5000, dims[0], device=hip_device, generator=torch.Generator(device=hip_device).manual_seed(1))
    outs = []
    for nojit in ("0", "1"):
        monkeypatch.setenv("MOLANN_NO_JIT", nojit)
        plan = _plan(dims, act)
        plan.update_mlp(ws, bs)
        out = torch.empty(5000, dims[-1], device=hip_device)
        plan.mlp_packed(f, out)
        torch.cuda.synchronize()
        assert ("molann_mlp_chain" in plan.last_launch_info()) == (nojit == "0")
        outs.append(out.cpu())
    assert float((outs[0] - outs[1]).abs().max()) <= 4e-3 * max(1.0, float(outs[1].abs().max()))


def test_chain_kernel_rereads_updated_weights(hip_device):
    dims, act = [96, 64, 4], _capi.ACT_TANH
    plan = _plan(dims, act)
    f = torch.randn(300, dims[0], device=hip_device)
    for seed in (1, 2):
        ws, bs = _params(dims, hip_device, seed)
        plan.update_mlp(ws, bs)
        out = torch.empty(300, dims[-1], device=hip_device)
        plan.mlp_packed(f, out)
        torch.cuda.synchronize()
        want = _emulate(f, ws, bs, act)
        assert float((out.cpu() - want).abs().max()) <= 4e-3 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("dims,act", [
    ([85, 128, 64, 8], _capi.ACT_TANH),          # C4
    ([6, 64, 64, 8], _capi.ACT_TANH),            # a wide MLP behind C3's six features
    ([341, 512, 256, 16], _capi.ACT_TANH),
    ([40, 33, 7], _capi.ACT_SIGMOID),
    ([70, 5], _capi.ACT_TANH),
    ([50, 64, 32, 48, 5], _capi.ACT_RELU),
    ([33, 40, 24, 36, 20, 9], _capi.ACT_TANH),
])
@pytest.mark.parametrize("n", [1, 255, 1000])
def test_fp32_chain_kernel_is_fp32(dims, act, n, hip_device):
    """The same kernel on the fp32 MFMA (exact fp32 products, fp32 accumulation): within 1e-5 of a float64 MLP."""
    plan = _plan(dims, act, precision=_capi.MLP_F32)
    ws, bs = _params(dims, hip_device, 11)
    plan.update_mlp(ws, bs)
    f = torch.randn(n, dims[0], device=hip_device, generator=torch.Generator(device=hip_device).manual_seed(n))
    out = torch.full((n, dims[-1]), float("nan"), device=hip_device)
    plan.mlp_packed(f, out)
    torch.cuda.synchronize()
    assert "molann_mlp_chain<f32" in plan.last_launch_info(), plan.last_launch_info()
    h = f.cpu().double()
    for i, (w, b) in enumerate(zip(ws, bs)):
        h = h @ w.cpu().double().T + b.cpu().double()
        if i + 1 < len(ws):
            h = ACTS[act](h)
    assert float((out.cpu().double() - h).abs().max()) <= 1e-5 * max(1.0, float(h.abs().max()))
