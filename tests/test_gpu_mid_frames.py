"""Frames between the two BASELINE systems (a few dozen to a few hundred atoms, more than 32 of them touched): the ring kernel
with several frames per ring entry - one rotation solve per entry, lane b solving frame b (frames_ring_kernel<ND, B>)."""

import re

import numpy as np
import pytest
import torch

from build_util import workload_model
from molann_amd import workloads as wl
from molann_amd.ann import last_launch_info
from oracle import molann_oracle as mo

pytestmark = pytest.mark.gpu


def _oracle_rows(w, model, xs):
    feats = [(t, [a - 1 for a in atoms]) for t, atoms in w.features]
    al = [a - 1 for a in w.align]
    ref_x = mo.center_reference(torch.from_numpy(w.ref_xyz[al])).double()
    want_f = mo.preprocessing_forward(xs.double(), feats, w.use_angle_value, al, ref_x)
    lins = [m for m in model.ann_layers if isinstance(m, torch.nn.Linear)]
    ws = [l.weight.detach().cpu().double() for l in lins]
    bs = [l.bias.detach().cpu().double() for l in lins]
    return want_f, mo.molann_forward(xs.double(), feats, ws, bs, w.use_angle_value, al, ref_x)


@pytest.mark.parametrize("n", [1, 7, 8, 9, 1000, 2048 * 8 + 3, 70001])
def test_peptide_frames_batched_solve(n, hip_device, monkeypatch):
    """P1 (166 atoms, Kabsch on 42, 8 dihedrals, MLP [16,32,8]): eight frames per ring entry (eight lanes per frame); features
    against one frame per entry (MOLANN_RING_BATCH=1) and against frames_wave_kernel within 2e-6 (the covariance sums are taken
    in another order: the last bits of the centroid differ), those two bit for bit, features and outputs within 1e-5 of the
    float64 oracle; batch sizes around the entry size, a short last entry, rings that wrap."""
    w = wl.get_workload("P1")
    model = workload_model(w, hip_device).requires_grad_(False)
    pp = model.preprocessing_layer
    x = w.make_frames(n, device=hip_device, seed=3)
    with torch.no_grad():
        f8 = pp(x)
        info = last_launch_info(pp)
        y8 = model(x)
    torch.cuda.synchronize()
    m = re.search(r"frames_ring_kernel<ND=1,B=8> \((\d+) consumer \+ (\d+) loader waves, ring of (\d+) entries of 8 frames", info)
    assert m, info
    monkeypatch.setenv("MOLANN_RING_BATCH", "1")
    with torch.no_grad():
        f1 = pp(x)
        assert "frames_ring_kernel<ND=1>" in last_launch_info(pp)
    monkeypatch.setenv("MOLANN_NO_RING", "1")
    with torch.no_grad():
        fw = pp(x)
        assert "frames_wave_kernel" in last_launch_info(pp)
    torch.cuda.synchronize()
    assert torch.equal(f1, fw)
    assert float((f8 - fw).abs().max()) <= 2e-6
    idx = torch.unique(torch.cat([torch.arange(0, min(n, 24)), torch.arange(max(0, n - 24), n),
                                  torch.from_numpy(np.random.default_rng(1).integers(0, n, size=200))]))
    want_f, want_y = _oracle_rows(w, model, x[idx.to(hip_device)].cpu())
    err_f = float((f8[idx.to(hip_device)].cpu().double() - want_f).abs().max())
    err_y = float((y8[idx.to(hip_device)].cpu().double() - want_y).abs().max())
    print("P1, %d frames: features max err %.3g, outputs %.3g" % (n, err_f, err_y))
    assert err_f <= 1e-5 and err_y <= 1e-5


@pytest.mark.parametrize("n_inp,n_align,n_feat,want", [(80, 40, 30, "ND=1,B=8"), (166, 120, 20, "ND=2,B=4"), (300, 150, 30, "ND=3,B=2"), (300, 230, 20, "ND=4,B=2"),
                                                       (400, 380, 10, "ND=6>")])
def test_entry_sizes(n_inp, n_align, n_feat, want, hip_device, monkeypatch):
    """The other entry sizes (4 frames of up to 128 windows, 2 of up to 256, 1 beyond) on chains with large alignment sets, against
    frames_wave_kernel (within 2e-6 of the features' scale; bit for bit with one frame per entry) and the float64 oracle within 1e-5."""
    from test_gpu_large_batches import _chain_plan
    xyz, feats, al, model = _chain_plan(n_inp, n_feat, n_align, [16, 4], hip_device, n_inp + n_align)
    pp = model.preprocessing_layer
    n = 5003
    g = torch.Generator().manual_seed(n_inp)
    x = (torch.from_numpy(xyz).unsqueeze(0) + 0.15 * torch.randn((n, n_inp, 3), generator=g)).to(hip_device)
    with torch.no_grad():
        f = pp(x)
        info = last_launch_info(pp)
    assert "frames_ring_kernel<" + want in info, info
    monkeypatch.setenv("MOLANN_NO_RING", "1")
    with torch.no_grad():
        fw = pp(x)
    torch.cuda.synchronize()
    if "B=" in info:
        assert float((f - fw).abs().max()) <= 2e-6 * max(1.0, float(fw.abs().max()))
    else:
        assert torch.equal(f, fw)
    ref_x = mo.center_reference(torch.from_numpy(xyz[al])).double()
    want_f = mo.preprocessing_forward(x[:256].cpu().double(), feats, False, al, ref_x)
    assert float((f[:256].cpu().double() - want_f).abs().max()) <= 1e-5 * max(1.0, float(want_f.abs().max()))
