"""The oracle (oracle/molann_oracle.py) against the reference's own outputs (tests/golden)."""

import numpy as np
import pytest
import torch

from golden_util import Case, case_names, load_meta

ALL = case_names()


def test_golden_inventory():
    assert len(ALL) >= 60
    kinds = {Case(n).kind for n in ("align_backbone_rigid", "features_C2", "molann_C3")}
    assert kinds == {"align", "features", "forward"}


@pytest.mark.parametrize("name", ALL)
def test_oracle_matches_reference_fp32(name):
    c = Case(name)
    with torch.no_grad():
        got = c.oracle(torch.float32)
    assert got.shape == c.out_f32.shape and got.dtype == torch.float32
    # same ATen ops in the same order: equal up to LAPACK/BLAS threading differences
    err = float((got - c.out_f32).abs().max()) if got.numel() else 0.0
    assert err <= 2e-6, err


@pytest.mark.parametrize("name", ALL)
def test_oracle_matches_reference_fp64(name):
    c = Case(name)
    if name.endswith("bf16w"):
        pytest.skip("weights are stored rounded; covered by the fp32 check")
    with torch.no_grad():
        got = c.oracle(torch.float64)
    assert got.dtype == torch.float64
    err = float((got - c.out_f64).abs().max()) if got.numel() else 0.0
    assert err <= 1e-10, err


def test_oracle_pdb_anchors():
    """Numbers quoted in SURVEY.md section 4 for the PDB frame."""
    c = Case("fmap_pdbframe_hist")
    o = c.oracle()[0].numpy()
    # columns: d(5,7,9,15) cs, d(7,9,15,17) cs, b(2,5), b(5,6), a(20,19,21), a(16,15,17), d(1,3,2,4) cs, d(1,2,3,4) cs
    assert np.allclose(o[0:2], [-1.0, 0.0], atol=1e-6) and np.allclose(o[2:4], [-1.0, 0.0], atol=1e-6)
    assert abs(o[4] - 1.5297) < 1e-4 and abs(o[5] - 1.23003721) < 1e-6
    assert abs(o[6] + 0.3328) < 1e-4 and abs(o[7] + 0.5423) < 1e-4
    assert np.allclose(o[8:10], [-0.50046289, 0.86575800], atol=1e-6)
    assert np.allclose(o[10:12], [-0.50046289, -0.86575800], atol=1e-6)


def test_oracle_feature_dims():
    from oracle import molann_oracle as mo
    for name in ALL:
        c = Case(name)
        if c.features:
            dims = [mo.feature_dim(t, len(idx), c.use_angle_value) for t, idx in c.features]
            assert dims == c.feat_dims and sum(dims) == c.feature_dim


def test_reference_meta_has_error_table():
    meta = load_meta()
    assert meta["errors"]["align_2d_input"] == "AssertionError"
    assert meta["errors"]["feature_repeated_atoms"] == "IndexError"
    assert meta["molann_state_dict_keys"][0] == "preprocessing_layer.align_layer.ref_x"
