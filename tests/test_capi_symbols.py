"""The C-ABI library loads and exports every symbol include/molann_hip.h declares (no GPU needed)."""

import ctypes

import pytest

from molann_amd import _capi


def test_library_exports_every_declared_symbol():
    L = _capi.lib()
    names = _capi.declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), "libmolann_hip.so does not export %s" % n
    assert L.molann_abi_version() == _capi.ABI_VERSION


def test_error_strings():
    assert _capi.error_string(0) == "ok"
    for code in range(-1, -10, -1):
        assert _capi.error_string(code) not in ("", "unknown error")


def _desc(**kw):
    d = _capi.PlanDesc()
    d.abi_version = _capi.ABI_VERSION
    d.n_inp = 22
    for k, v in kw.items():
        setattr(d, k, v)
    return d


def _create(d):
    h = ctypes.c_void_p()
    return _capi.lib().molann_plan_create(ctypes.byref(d), ctypes.byref(h))


def test_plan_validation_rejects_bad_descriptions():
    """Validation runs before any device call, so it is checkable without a GPU."""
    i32 = ctypes.c_int32
    assert _capi.lib().molann_plan_create(None, ctypes.byref(ctypes.c_void_p())) == _capi.E_NULL
    assert _create(_desc()) == _capi.E_DESC                        # no stage at all
    assert _create(_desc(abi_version=99, n_features=1)) == _capi.E_DESC
    ft, fp, fi = (i32 * 1)(1), (i32 * 2)(0, 2), (i32 * 2)(4, 22)
    assert _create(_desc(n_features=1, feat_type=ft, feat_ptr=fp, feat_idx=fi)) == _capi.E_INDEX
    fi3 = (i32 * 3)(4, 5, 6)
    assert _create(_desc(n_features=1, feat_type=ft, feat_ptr=(i32 * 2)(0, 3), feat_idx=fi3)) == _capi.E_FEATURE
    assert _create(_desc(n_features=1, feat_type=(i32 * 1)(7), feat_ptr=fp, feat_idx=(i32 * 2)(4, 5))) == _capi.E_FEATURE
    assert _create(_desc(n_align=2, align_idx=(i32 * 2)(0, 30), ref_x=(ctypes.c_float * 6)())) == _capi.E_INDEX
    assert _create(_desc(n_align=2)) == _capi.E_NULL
    dims = (i32 * 2)(3, 0)
    assert _create(_desc(n_layers=1, layer_dims=dims)) == _capi.E_DESC


def test_launches_reject_null_plan():
    L = _capi.lib()
    assert L.molann_align_f32(None, None, 0, None, None) == _capi.E_NULL
    assert L.molann_features_f32(None, None, 0, None, None) == _capi.E_NULL
    assert L.molann_forward_packed_f32(None, None, 0, None, None) == _capi.E_NULL
    assert L.molann_plan_destroy(None) == 0
