"""Parity of the HIP path (through the module API -> ctypes -> C ABI -> gfx950 kernels) with the
reference's own outputs (tests/golden) and with the oracle run on this box.  All tests need a GPU."""

import numpy as np
import pytest
import torch

from build_util import build_modules, oracle_for_workload, workload_model
from golden_util import Case, case_names
from molann_amd import workloads as wl
from molann_amd.ann import last_launch_info

pytestmark = pytest.mark.gpu

ALL = case_names()
F32_TOL = 1e-5   # BASELINE.json: "within 1e-5 fp32" of the reference CPU path


def _run(model, x):
    with torch.no_grad():
        y = model(x)
    torch.cuda.synchronize()
    return y.cpu()


@pytest.mark.parametrize("name", [n for n in ALL if not n.endswith("bf16w")])
def test_golden_case(name, hip_device):
    c = Case(name)
    model = build_modules(c, hip_device)
    got = _run(model, c.x.to(hip_device))
    assert got.shape == c.out_f32.shape
    if got.numel() == 0:
        return
    assert torch.isfinite(got).all()
    err32 = float((got - c.out_f32).abs().max())
    err64 = float((got.double() - c.out_f64).abs().max())
    own = float((c.out_f32.double() - c.out_f64).abs().max())
    # within 1e-5 of the reference fp32 run, except where that run itself is further than that from
    # its own fp64 run (ill-conditioned alignment sets): there fp64 arbitrates
    assert err32 <= c.tolerance_vs_f32(), (err32, err64, own)
    assert err64 <= max(F32_TOL, own), (err32, err64, own)


def test_bf16_mlp_case(hip_device):
    """C5-shaped model on the bf16 MFMA path against the reference run with bf16-rounded weights."""
    c = Case("molann_C5_small_bf16w")
    model = build_modules(c, hip_device, mlp_precision="bf16")
    got = _run(model, c.x.to(hip_device))
    scale = float(c.out_f64.abs().max())
    err = float((got.double() - c.out_f64).abs().max())
    # measured 1.18e-3 (output scale 0.35): twice that; tests/test_gpu_large_batches.py holds the kernel to its arithmetic model
    assert err <= 2.5e-3 * max(1.0, scale), (err, scale)   # bf16 activations: 8-bit mantissa through 3 layers


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 63, 64, 65, 127, 128, 129, 1000, 4099])
@pytest.mark.parametrize("cfg", ["C1", "C2", "C3", "C3p"])
def test_batch_sizes_vs_oracle(cfg, n, hip_device):
    w = wl.get_workload(cfg)
    model = workload_model(w, hip_device)
    x = w.make_frames(max(n, 1), seed=100 + n)[:n]
    got = _run(model, x.to(hip_device))
    if n == 0:
        assert got.shape == (0, w.out_dim())
        return
    want = oracle_for_workload(w, model, x, torch.float64)
    assert got.shape == want.shape
    assert float((got.double() - want).abs().max()) <= 2e-5


def test_align_batch_sizes_vs_oracle(hip_device):
    from build_util import Universe  # noqa: F401
    from molann_amd.ann import AlignmentLayer
    from molann_amd.atomgroup import Universe as U
    from oracle import molann_oracle as mo
    w = wl.get_workload("C3")
    u = U(w.ref_xyz)
    al = AlignmentLayer(u.atoms_by_number(w.align), u.atoms).to(hip_device)
    ref_x = mo.center_reference(torch.from_numpy(w.ref_xyz[[a - 1 for a in w.align]])).double()
    for n in (1, 2, 3, 5, 64, 65, 200, 1025):
        x = w.make_frames(n, seed=7 + n)
        got = _run(al, x.to(hip_device))
        want = mo.align_forward(x.double(), [a - 1 for a in w.align], ref_x)
        assert float((got.double() - want).abs().max()) <= 1e-5, n


def test_unaligned_and_noncontiguous_inputs(hip_device):
    w = wl.get_workload("C3")
    model = workload_model(w, hip_device)
    x = w.make_frames(300, seed=5).to(hip_device)
    base = _run(model, x)
    # a view starting one frame in: data_ptr is only 8-byte aligned (264 B frames) -> narrow load path
    got = _run(model, x[1:])
    assert x[1:].data_ptr() % 16 != 0
    assert torch.equal(got, base[1:])
    # non-contiguous (strided) input
    xs = torch.empty((300, 22, 6), device=hip_device)[:, :, :3]
    xs.copy_(x)
    assert not xs.is_contiguous()
    assert torch.equal(_run(model, xs), base)
    # input is never mutated
    x0 = x.clone()
    _run(model, x)
    assert torch.equal(x, x0)


def test_rigid_motion_invariance_full_size(hip_device):
    """Size-independent property at the full C3 size: the output does not change when every frame is
    moved by its own random rotation + translation (features see the aligned frame)."""
    w = wl.get_workload("C3")
    model = workload_model(w, hip_device)
    n = 1 << 20
    x = w.make_frames(n, device=hip_device, seed=3)
    g = torch.Generator(device=hip_device).manual_seed(11)
    q = torch.randn((n, 4), generator=g, device=hip_device)
    q = q / q.norm(dim=1, keepdim=True)
    x2 = torch.matmul(x, wl.quaternion_to_matrix(q)) + 5.0 * torch.randn((n, 1, 3), generator=g, device=hip_device)
    with torch.no_grad():
        y1, y2 = model(x), model(x2)
    assert torch.isfinite(y1).all()
    assert float((y1 - y2).abs().max()) <= 2e-5
    # and a sample of it against the oracle
    idx = torch.randint(0, n, (4096,), generator=torch.Generator().manual_seed(1))
    want = oracle_for_workload(w, model, x[idx.to(hip_device)], torch.float64)
    assert float((y1[idx.to(hip_device)].cpu().double() - want).abs().max()) <= 1e-5


def test_alignment_is_idempotent_and_centres(hip_device):
    """Aligning an aligned frame changes nothing; the align atoms end up centred on the origin."""
    from molann_amd.ann import AlignmentLayer
    from molann_amd.atomgroup import Universe as U
    w = wl.get_workload("C3")
    u = U(w.ref_xyz)
    al = AlignmentLayer(u.atoms_by_number(w.align), u.atoms).to(hip_device)
    x = w.make_frames(1 << 16, device=hip_device, seed=9)
    with torch.no_grad():
        a1 = al(x)
        a2 = al(a1)
    assert float((a1 - a2).abs().max()) <= 1e-5
    cen = a1[:, [a - 1 for a in w.align], :].mean(dim=1)
    assert float(cen.abs().max()) <= 1e-5


def test_live_parameters_and_buffers_are_reread(hip_device):
    w = wl.get_workload("C3")
    model = workload_model(w, hip_device)
    x = w.make_frames(256, seed=2).to(hip_device)
    y0 = _run(model, x)
    lins = [m for m in model.ann_layers if isinstance(m, torch.nn.Linear)]
    with torch.no_grad():
        lins[-1].bias.add_(1.0)          # in-place update, as an optimiser step does
    y1 = _run(model, x)
    assert torch.allclose(y1, y0 + 1.0, atol=1e-6)
    want = oracle_for_workload(w, model, x.cpu(), torch.float64)
    assert float((y1.double() - want).abs().max()) <= 1e-5
    # replacing the buffer through load_state_dict is seen too
    sd = model.state_dict()
    key = "preprocessing_layer.align_layer.ref_x"
    rot = wl.quaternion_to_matrix(torch.tensor([[0.5, 0.5, 0.5, 0.5]])).to(hip_device)[0]
    sd[key] = sd[key] @ rot
    model.load_state_dict(sd)
    y2 = _run(model, x)
    assert float((y2 - y1).abs().max()) <= 2e-5   # C3's features are rotation invariant ...
    pp = model.preprocessing_layer
    a = _run(pp.align_layer, x)                    # ... the aligned coordinates are not
    from oracle import molann_oracle as mo
    want_a = mo.align_forward(x.cpu().double(), pp.align_layer._local_align_atom_indices, sd[key].cpu().double())
    assert float((a.double() - want_a).abs().max()) <= 1e-5


def test_generic_ann_layers_and_other_activations(hip_device):
    """An ann_layers module the kernels do not recognise still gets GPU-computed features; ELU/GELU/
    Softplus MLPs take the MFMA MLP kernel."""
    from molann_amd.ann import MolANN, create_sequential_nn
    w = wl.get_workload("C3")
    base = workload_model(w, hip_device)
    x = w.make_frames(500, seed=4).to(hip_device)
    f = _run(base.preprocessing_layer, x)
    for act in (torch.nn.ELU(), torch.nn.GELU(), torch.nn.Softplus(), torch.nn.SiLU(), torch.nn.LeakyReLU()):
        torch.manual_seed(1)
        nn = create_sequential_nn([6, 24, 12, 5], activation=act).to(hip_device)
        m = MolANN(base.preprocessing_layer, nn)
        got = _run(m, x)
        with torch.no_grad():
            want = nn.double().cpu()(f.double())
        assert float((got.double() - want).abs().max()) <= 1e-5, type(act).__name__
    class Odd(torch.nn.Module):
        def forward(self, t):
            return t.sum(dim=1, keepdim=True)
    got = _run(MolANN(base.preprocessing_layer, Odd()), x)
    assert torch.allclose(got, f.sum(dim=1, keepdim=True))


def test_wide_fp32_mlp_on_small_frames(hip_device):
    """22-atom frames with an MLP too wide for the fused lane kernel: features -> MFMA fp32 MLP."""
    from molann_amd.ann import MolANN, create_sequential_nn
    w = wl.get_workload("C3")
    base = workload_model(w, hip_device)
    x = w.make_frames(3000, seed=6).to(hip_device)
    f = _run(base.preprocessing_layer, x)
    torch.manual_seed(2)
    nn = create_sequential_nn([6, 100, 70, 3]).to(hip_device)
    got = _run(MolANN(base.preprocessing_layer, nn), x)
    with torch.no_grad():
        want = nn.double().cpu()(f.double())
    assert float((got.double() - want).abs().max()) <= 1e-5


@pytest.mark.parametrize("dims", [[6, 64, 64, 8], [6, 48, 33, 5], [6, 128, 128, 8], [6, 100, 70, 3], [6, 40, 8]])
def test_wide_heads_run_inside_the_lane_kernel(dims, hip_device, monkeypatch):
    """Hidden widths 33 .. 128 behind the 22-atom preprocessing: ONE launch of the lane kernel's WIDE_MLP build (the chain
    MLP's arithmetic on a weight stream resident in LDS) - against a float64 MLP on the kernel's own features, bit for bit
    against the two-kernel path it replaces (the same accumulation order), across batch sizes incl. ragged tiles, and with
    live parameter updates."""
    from molann_amd.ann import MolANN, create_sequential_nn
    w = wl.get_workload("C3")
    base = workload_model(w, hip_device)
    torch.manual_seed(sum(dims))
    nn = create_sequential_nn(dims).to(hip_device)
    model = MolANN(base.preprocessing_layer, nn).requires_grad_(False)
    for n in (1, 63, 64, 65, 1000, 70001):
        x = w.make_frames(n, seed=n).to(hip_device)
        got = _run(model, x)
        assert "molann_lane_jit<NL=%d,wide>" % (len(dims) - 1) in last_launch_info(model), last_launch_info(model)
        f = _run(base.preprocessing_layer, x)
        h = f.double()
        lins = [m for m in nn if isinstance(m, torch.nn.Linear)]
        for i, lin in enumerate(lins):
            h = h @ lin.weight.detach().cpu().double().T + lin.bias.detach().cpu().double()
            if i + 1 < len(lins):
                h = torch.tanh(h)
        assert float((got.double() - h).abs().max()) <= 1e-5, n
    monkeypatch.setenv("MOLANN_NO_WIDE_FUSED", "1")
    model2 = MolANN(base.preprocessing_layer, nn).requires_grad_(False)
    got2 = _run(model2, x)
    assert "molann_mlp_chain<f32" in last_launch_info(model2), last_launch_info(model2)
    assert torch.equal(got2, got)
    monkeypatch.delenv("MOLANN_NO_WIDE_FUSED")
    with torch.no_grad():
        lins[-1].bias.add_(0.5)
    assert torch.allclose(_run(model, x), got + 0.5, atol=1e-6)


def test_cpu_tensor_and_grad_mode_fail_loudly(hip_device):
    w = wl.get_workload("C3")
    model = workload_model(w, hip_device)
    x = w.make_frames(4)
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            model(x)                                  # CPU tensor: no fallback
    y = model(x.to(hip_device))                       # grad mode on: served by the backward-capable path
    assert y.requires_grad
    with pytest.raises(RuntimeError):                  # float64 input, float32 module: mixed dtypes, as in the reference
        with torch.no_grad():
            model(x.to(hip_device).double())
    with pytest.raises(TypeError):                     # neither float32 nor float64
        with torch.no_grad():
            model(x.to(hip_device).half())


def test_capi_direct_forward_f32(hip_device):
    """molann_forward_f32 (live W/b pointers) called straight through ctypes."""
    import ctypes
    from molann_amd import _capi
    w = wl.get_workload("C1")
    model = workload_model(w, hip_device)
    x = w.make_frames(777, seed=8).to(hip_device)
    lins = [m for m in model.ann_layers if isinstance(m, torch.nn.Linear)]
    feats = [(t, [a - 1 for a in atoms]) for t, atoms in w.features]
    plan = _capi.Plan(22, features=feats, layer_dims=w.mlp_dims)
    out = torch.empty((777, 3), device=hip_device)
    W = (ctypes.c_void_p * 2)(*[l.weight.data_ptr() for l in lins])
    B = (ctypes.c_void_p * 2)(*[l.bias.data_ptr() for l in lins])
    rc = _capi.lib().molann_forward_f32(plan._handle, ctypes.c_void_p(x.data_ptr()), 777, W, B,
                                        ctypes.c_void_p(out.data_ptr()), plan._stream())
    assert rc == 0
    torch.cuda.synchronize()
    want = oracle_for_workload(w, model, x.cpu(), torch.float64)
    assert float((out.cpu().double() - want).abs().max()) <= 1e-5
    assert "lane" in plan.last_launch_info()      # generic or plan-specialised lane kernel
    # stage errors
    assert _capi.lib().molann_align_f32(plan._handle, ctypes.c_void_p(x.data_ptr()), 777,
                                        ctypes.c_void_p(out.data_ptr()), plan._stream()) == _capi.E_STAGE


@pytest.mark.parametrize("env", [{"MOLANN_NO_JIT": "1"}, {"MOLANN_NO_JIT": "1", "MOLANN_NO_REGS": "1"}],
                         ids=["generic_regs", "generic_lds"])
@pytest.mark.parametrize("name", ["molann_C1", "molann_C3", "features_C2", "features_C3p", "features_C3_val",
                                  "flayer_permuted_input", "pp_align123_pos12", "molann_C3_relu", "molann_C3_sigmoid"])
def test_generic_lane_kernels(name, env, hip_device, monkeypatch):
    """The ahead-of-time lane kernel (what runs when hipRTC is missing), in both of its feature modes."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    c = Case(name)
    model = build_modules(c, hip_device)
    got = _run(model, c.x.to(hip_device))
    pp = model if c.kind == "features" else model.preprocessing_layer
    info = last_launch_info(model if c.kind != "features" else pp)
    assert "frames_lane_kernel" in info, info
    assert ("features_lds" in info) == ("MOLANN_NO_REGS" in env) or "features_regs" not in info
    assert float((got - c.out_f32).abs().max()) <= c.tolerance_vs_f32()


def test_specialised_kernel_is_used_by_default(hip_device):
    c = Case("molann_C3")
    model = build_modules(c, hip_device)
    _run(model, c.x.to(hip_device))
    info = last_launch_info(model)
    assert "molann_lane_jit" in info, info


def test_align_workload_full_size_properties(hip_device):
    """AlignmentLayer.forward at 1M frames: internal distances are preserved (rigid map) and the align
    atoms' centroid lands on the origin."""
    w = wl.get_workload("A3")
    al = wl.build_model(w, hip_device)
    x = w.make_frames(1 << 20, device=hip_device, seed=12)
    with torch.no_grad():
        a = al(x)
    d0 = (x[:, 4] - x[:, 18]).norm(dim=1)
    d1 = (a[:, 4] - a[:, 18]).norm(dim=1)
    assert float((d0 - d1).abs().max()) <= 2e-5
    cen = a[:, [i - 1 for i in w.align], :].mean(dim=1)
    assert float(cen.abs().max()) <= 1e-5


def test_hip_graph_replay_matches_eager(hip_device):
    from molann_amd.graph import GraphedForward
    for cfg in ("C1", "C3"):
        w = wl.get_workload(cfg)
        model = workload_model(w, hip_device).requires_grad_(False)
        x0 = w.make_frames(1024, seed=1).to(hip_device)
        g = GraphedForward(model, x0)
        for seed in (2, 3):
            x = w.make_frames(1024, seed=seed).to(hip_device)
            want = _run(model, x)
            got = g(x).clone()
            torch.cuda.synchronize()
            assert torch.equal(got.cpu(), want), cfg


@pytest.mark.parametrize("cfg", ["C3", "P1"])
def test_hip_graph_forces_match_autograd(cfg, hip_device):
    """GraphedForces: the values and the vector-Jacobian product of a small batch as two HIP graph replays (a collective variable
    differentiated at every MD step), against eager autograd; with its latency per call.  C3: the one-pass backward kernel; P1
    (166 atoms, wave-per-frame preprocessing): the forward that keeps its features and the two-launch backward on them."""
    import time
    from molann_amd.graph import GraphedForces
    w = wl.get_workload(cfg)
    model = workload_model(w, hip_device).requires_grad_(False)
    for n in (1, 64):
        g = GraphedForces(model, w.make_frames(n, seed=1).to(hip_device))
        for seed in (2, 3):
            x = w.make_frames(n, seed=seed).to(hip_device)
            dy = torch.randn((n, w.out_dim()), generator=torch.Generator().manual_seed(seed)).to(hip_device)
            y = g(x).clone()
            dx = g.vjp(dy).clone()
            xe = x.clone().requires_grad_(True)
            ye = model(xe)
            (dxe,) = torch.autograd.grad(ye, xe, dy)
            assert torch.equal(y, ye.detach())
            assert float((dx - dxe).abs().max()) <= 1e-6 * max(1.0, float(dxe.abs().max()))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            g(x)
            g.vjp(dy)
        torch.cuda.synchronize()
        print("GraphedForces, %s, %d frame(s): %.1f us per values + forces" % (cfg, n, (time.perf_counter() - t0) / 200 * 1e6))
        y2, dx2 = g.value_and_vjp(x, dy)
        assert float((y2 - ye.detach()).abs().max()) <= 1e-6 and float((dx2 - dxe).abs().max()) <= 1e-6 * max(1.0, float(dxe.abs().max()))
    # the whole Jacobian of one frame's values: a batch of d_out copies of the frame, the identity as cotangent
    d_out = w.out_dim()
    x1 = w.make_frames(1, seed=9).to(hip_device)
    g = GraphedForces(model, x1.expand(d_out, -1, -1).contiguous())
    y = g(x1.expand(d_out, -1, -1).contiguous())[0].clone()
    J = g.vjp(torch.eye(d_out, device=hip_device)).clone()
    xe = x1.clone().requires_grad_(True)
    ye = model(xe)
    assert torch.equal(y, ye.detach()[0])
    for k in range(d_out):
        (gk,) = torch.autograd.grad(ye[0, k], xe, retain_graph=True)
        assert float((J[k] - gk[0]).abs().max()) <= 1e-6 * max(1.0, float(gk.abs().max()))


def test_host_trajectory_streamer(hip_device):
    """Pinned double-buffered H2D/D2H around the forward gives the same rows as one resident batch."""
    from molann_amd.stream import stream_forward
    w = wl.get_workload("C3")
    model = workload_model(w, hip_device).requires_grad_(False)
    x = w.make_frames(10000, seed=31)
    want = _run(model, x.to(hip_device)).numpy()
    for chunk in (1024, 4096, 10000, 1 << 20):
        got = stream_forward(model, x.numpy(), chunk_frames=chunk, device=hip_device)
        assert got.shape == want.shape and np.array_equal(got, want), chunk
    assert stream_forward(model, x.numpy()[:0], device=hip_device).shape == (0, 8)
    # a pinned torch tensor crosses the link from where it lies (no staging copy); a pageable one is staged
    assert np.array_equal(stream_forward(model, x.clone().pin_memory(), chunk_frames=3000, device=hip_device), want)
    assert np.array_equal(stream_forward(model, x, chunk_frames=3000, device=hip_device), want)


def test_hip_graph_replay_of_the_two_stream_path(hip_device):
    """An MLP outside the fused lane kernel: feature kernel on the caller's stream, chain MLP kernel on the plan's
    side stream, fork/join by events - all of it capturable into one HIP graph."""
    from molann_amd.graph import GraphedForward
    from molann_amd.ann import MolANN, create_sequential_nn, last_launch_info
    w = wl.get_workload("C3")
    base = workload_model(w, hip_device)
    torch.manual_seed(3)
    # (a head whose weight stream does not fit LDS: narrower ones run inside the lane kernel, test_wide_heads_run_inside_the_lane_kernel)
    model = MolANN(base.preprocessing_layer, create_sequential_nn([6, 256, 256, 8]).to(hip_device)).requires_grad_(False)
    x = w.make_frames(300000, seed=41).to(hip_device)        # more than one workspace chunk
    want = _run(model, x)
    assert "molann_mlp_chain<f32" in last_launch_info(model), last_launch_info(model)
    g = GraphedForward(model, x)
    for seed in (42, 43):
        x2 = w.make_frames(300000, seed=seed).to(hip_device)
        got = g(x2).cpu()
        assert torch.equal(got, _run(model, x2))
    ref = oracle_for_workload(w, model, x[:2000].cpu(), torch.float64)
    assert float((want[:2000].double() - ref).abs().max()) <= F32_TOL
