"""The plan-specialised lane kernel: source generation and hipRTC compilation need no GPU."""

import ctypes

import pytest

from molann_amd import _capi, workloads as wl


_desc = _capi.workload_desc


@pytest.mark.parametrize("name", ["C1", "C2", "C3", "C3p"])
def test_specialised_kernel_compiles(name):
    d, keep = _desc(wl.get_workload(name))
    buf = ctypes.create_string_buffer(1 << 20)
    rc = _capi.lib().molann_debug_jit(ctypes.byref(d), 1, buf, 1 << 20)
    assert rc > 1000, (rc, buf.value.decode()[:2000])        # source length; a compile failure returns the log
    src = buf.value.decode()
    w = wl.get_workload(name)
    assert "constexpr int N_INP = %d;" % w.n_atoms in src
    assert "constexpr int N_ALIGN = %d;" % (len(w.align) if w.align else 0) in src
    assert "molann_lane_jit" in src


@pytest.mark.parametrize("mode,entry", [(3, "molann_lane_bwd"), (11, "molann_mlp_bwd"), (19, "molann_bwd_ring"), (83, "molann_bwd_ring")])
@pytest.mark.parametrize("name", ["C1", "C2", "C3", "C3p"])
def test_backward_kernels_compile(name, mode, entry):
    """The plan-specialised backward kernels (preprocessing half, MLP half, one pass) cross-compile for gfx950."""
    w = wl.get_workload(name)
    d, keep = _desc(w)
    buf = ctypes.create_string_buffer(1 << 21)
    rc = _capi.lib().molann_debug_jit(ctypes.byref(d), mode, buf, 1 << 21)
    if mode == 11 and not w.mlp_dims:
        assert rc == _capi.E_STAGE
        return
    assert rc > 1000, (rc, buf.value.decode()[:3000])
    assert entry in buf.value.decode()


@pytest.mark.parametrize("dims,act", [([6, 32, 8], 0), ([6, 17, 5, 9, 2], 2), ([6, 30, 31], 5), ([6, 32, 32, 32, 4], 7), ([6, 4], 0),
                                      ([6, 24, 24, 1], 1), ([6, 32, 32, 8], 5), ([6, 1, 1], 3)])
@pytest.mark.parametrize("mode", [11, 19])
def test_backward_kernels_compile_for_other_mlps(dims, act, mode):
    """Host geometry (scratch rows, fragment image, LDS layout) against the kernels' own static_asserts, over layer counts, widths
    and activations (SiLU keeps the pre-activations as well): the MLP half and the one-pass backward cross-compile."""
    d, keep = _desc(wl.get_workload("C3"))
    ld = (ctypes.c_int32 * len(dims))(*dims)
    d.n_layers, d.layer_dims, d.activation = len(dims) - 1, ld, act
    buf = ctypes.create_string_buffer(1 << 21)
    rc = _capi.lib().molann_debug_jit(ctypes.byref(d), mode, buf, 1 << 21)
    assert rc > 1000, (rc, buf.value.decode()[:3000])


def test_code_object_cache_on_disk(tmp_path, monkeypatch):
    """MOLANN_JIT_CACHE_DIR: the second build of the same kernel is read from disk; a damaged file is rebuilt, not loaded."""
    import os
    import time
    os.chmod(tmp_path, 0o700)
    monkeypatch.setenv("MOLANN_JIT_CACHE_DIR", str(tmp_path))
    d, keep = _desc(wl.get_workload("C2"))
    buf = ctypes.create_string_buffer(1 << 20)
    t0 = time.perf_counter()
    assert _capi.lib().molann_debug_jit(ctypes.byref(d), 1, buf, 1 << 20) > 1000
    t1 = time.perf_counter()
    files = [f for f in os.listdir(tmp_path) if f.endswith(".hsaco")]
    assert len(files) == 1 and open(os.path.join(tmp_path, files[0]), "rb").read(8) == b"MOLANNCO"    # header, then the ELF
    assert _capi.lib().molann_debug_jit(ctypes.byref(d), 1, buf, 1 << 20) > 1000
    t2 = time.perf_counter()
    assert t2 - t1 < 0.5 * (t1 - t0)                     # no compile the second time
    open(os.path.join(tmp_path, files[0]), "wb").write(b"not a code object")
    assert _capi.lib().molann_debug_jit(ctypes.byref(d), 1, buf, 1 << 20) > 1000
    assert open(os.path.join(tmp_path, files[0]), "rb").read(36)[32:] == b"\x7fELF"    # rebuilt and replaced
    d2, keep2 = _desc(wl.get_workload("C3"))              # another plan: another file
    assert _capi.lib().molann_debug_jit(ctypes.byref(d2), 1, buf, 1 << 20) > 1000
    assert len([f for f in os.listdir(tmp_path) if f.endswith(".hsaco")]) == 2


def test_large_frames_are_not_specialised():
    d, keep = _desc(wl.get_workload("C4"))
    rc = _capi.lib().molann_debug_jit(ctypes.byref(d), 0, None, 0)
    assert rc == _capi.E_UNSUPPORTED


def test_jit_cache_verifies_what_it_loads(tmp_path, monkeypatch):
    """the on-disk code-object cache is used only from a directory of this user that nobody else can write,
    a cached file carries its length and a hash of its contents, and one that does not verify is deleted and rebuilt."""
    import os
    d, keep = _desc(wl.get_workload("C2"))
    buf = ctypes.create_string_buffer(1 << 20)
    cache = tmp_path / "cache"
    cache.mkdir(mode=0o700)
    monkeypatch.setenv("MOLANN_JIT_CACHE_DIR", str(cache))
    assert _capi.lib().molann_debug_jit(ctypes.byref(d), 1, buf, 1 << 20) > 1000
    files = list(cache.iterdir())
    assert len(files) == 1 and files[0].name.endswith(".hsaco")
    blob = files[0].read_bytes()
    assert blob[:8] == b"MOLANNCO" and blob[32:36] == b"\x7fELF"
    assert _capi.lib().molann_debug_jit(ctypes.byref(d), 1, buf, 1 << 20) > 1000          # served from the cache
    # a flipped byte in the code object: not loaded, deleted, rebuilt
    files[0].write_bytes(blob[:200] + bytes([blob[200] ^ 0xff]) + blob[201:])
    assert _capi.lib().molann_debug_jit(ctypes.byref(d), 1, buf, 1 << 20) > 1000
    assert files[0].read_bytes() == blob
    # a truncated file
    files[0].write_bytes(blob[:1000])
    assert _capi.lib().molann_debug_jit(ctypes.byref(d), 1, buf, 1 << 20) > 1000
    assert files[0].read_bytes() == blob
    # a directory others may write is not used
    loose = tmp_path / "loose"
    loose.mkdir()
    os.chmod(loose, 0o777)
    monkeypatch.setenv("MOLANN_JIT_CACHE_DIR", str(loose))
    assert _capi.lib().molann_debug_jit(ctypes.byref(d), 1, buf, 1 << 20) > 1000
    assert list(loose.iterdir()) == []


@pytest.mark.parametrize("dims,act", [([6, 64, 64, 8], 0), ([6, 48, 33, 5], 2), ([6, 128, 128, 8], 0), ([6, 100, 70, 3], 4), ([6, 40, 8], 8), ([6, 33, 33, 33, 2], 1)])
def test_wide_fused_forward_compiles(dims, act):
    """The WIDE_MLP build of the lane kernel (hidden widths beyond 32: the chain MLP's arithmetic on an LDS-resident weight stream)
    cross-compiles for gfx950 over widths, layer counts and activations; heads whose stream does not fit are refused."""
    d, keep = _desc(wl.get_workload("C3"))
    ld = (ctypes.c_int32 * len(dims))(*dims)
    d.n_layers, d.layer_dims, d.activation = len(dims) - 1, ld, act
    buf = ctypes.create_string_buffer(1 << 22)
    rc = _capi.lib().molann_debug_jit(ctypes.byref(d), 129, buf, 1 << 22)
    assert rc > 1000, (rc, buf.value.decode()[:3000])
    assert "constexpr bool WIDE_MLP = true;" in buf.value.decode()
    big = (ctypes.c_int32 * 4)(6, 256, 256, 8)
    d.n_layers, d.layer_dims = 3, big
    assert _capi.lib().molann_debug_jit(ctypes.byref(d), 129, buf, 1 << 22) == _capi.E_UNSUPPORTED
