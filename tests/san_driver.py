"""Run in a CHILD process by tests/test_sanitized_host.py with the ASan runtime preloaded and MOLANN_SAN_LIB=1: drives
the host half of libmolann_hip (descriptor validation, feature-item expansion, compact windows, ring geometry, kernel
source generation, the host instantiation of the per-frame math) through every entry point that needs no GPU, on
valid, random and deliberately malformed descriptions.  Any ASan / UBSan report aborts the process."""
import ctypes
import random
import sys

from molann_amd import _capi

L = _capi.lib()
assert _capi.build_kind() in ("release", "diagnostics")
I = lambda v: (ctypes.c_int32 * max(1, len(v)))(*v)  # noqa: E731


def desc(n_inp, align, feats, uav, dims, act=0, prec=0, abi=_capi.ABI_VERSION):
    d = _capi.PlanDesc()
    d.abi_version, d.n_inp = abi, n_inp
    keep = []
    if align:
        a, r = I(align), (ctypes.c_float * (3 * len(align)))(*[0.1 * i for i in range(3 * len(align))])
        d.n_align, d.align_idx, d.ref_x = len(align), a, r
        keep += [a, r]
    if feats:
        ptr, flat = [0], []
        for _, idx in feats:
            flat += list(idx)
            ptr.append(len(flat))
        ft, fp, fi = I([t for t, _ in feats]), I(ptr), I(flat)
        d.n_features, d.feat_type, d.feat_ptr, d.feat_idx = len(feats), ft, fp, fi
        keep += [ft, fp, fi]
    d.use_angle_value = uav
    if dims:
        ld = I(dims)
        d.n_layers, d.layer_dims = len(dims) - 1, ld
        keep.append(ld)
    d.activation, d.mlp_precision = act, prec
    return d, keep


buf = ctypes.create_string_buffer(1 << 21)
rng = random.Random(7)
n_ok = n_rej = 0
ARITY = {0: 3, 1: 2, 2: 4}
for trial in range(400):
    n_inp = rng.choice([1, 2, 3, 5, 22, 22, 22, 64, 85, 300, 5000])
    feats = []
    for _ in range(rng.randint(0, 12)):
        t = rng.choice([0, 1, 2, 3])
        k = ARITY.get(t, rng.randint(1, 6))
        if k <= n_inp:
            feats.append((t, rng.sample(range(n_inp), k)))
    align = rng.sample(range(n_inp), rng.randint(0, min(n_inp, 9))) if rng.random() < 0.6 else []
    uav = rng.randint(0, 1)
    dims = None
    if feats and rng.random() < 0.6:
        d_feat = sum(3 * len(ix) if t == 3 else (2 if (t == 2 and not uav) else 1) for t, ix in feats)
        dims = [d_feat] + [rng.choice([1, 2, 5, 8, 16, 32, 33, 64]) for _ in range(rng.randint(1, 4))]
    # one in four descriptions is broken on purpose
    broken = rng.random() < 0.25
    if broken:
        kind = rng.randint(0, 5)
        if kind == 0 and feats:
            feats[0] = (feats[0][0], [n_inp + 3] + list(feats[0][1][1:]))     # index out of range
        elif kind == 1 and feats:
            feats[0] = (7, feats[0][1])                                       # unknown type
        elif kind == 2 and feats:
            feats[0] = (1, list(range(min(3, n_inp))))                        # bond with 3 atoms
        elif kind == 3 and dims:
            dims[0] += 1                                                      # layer_dims[0] != feature dim
        elif kind == 4 and align:
            align[0] = -1
        else:
            dims = [3] * 40 if feats else dims                                # too many layers
    d, keep = desc(n_inp, align, feats, uav, dims, act=rng.randint(0, 8), prec=rng.randint(0, 1))
    # source only (compiling is hipRTC's, third-party code): forward, backward of the preprocessing, of the MLP, in one pass
    for mode, name in ((0, b"molann_lane_jit"), (2, b"molann_lane_bwd"), (10, b"molann_mlp_bwd"), (18, b"molann_bwd_ring")):
        rc = L.molann_debug_jit(ctypes.byref(d), mode, buf, 1 << 21)
        if rc > 0:
            n_ok += 1
            assert name in buf.value
        else:
            n_rej += 1
    h = ctypes.c_void_p()
    rc = L.molann_plan_create(ctypes.byref(d), ctypes.byref(h))   # no GPU here: validation + expansion, then a HIP error
    assert rc != 0 and not h.value
# malformed ABI, null pointers
d, keep = desc(22, [1, 4], [(1, [4, 5])], 0, None, abi=99)
assert L.molann_debug_jit(ctypes.byref(d), 0, buf, 1 << 21) < 0
assert L.molann_plan_create(None, None) != 0
assert L.molann_plan_destroy(None) == 0 and L.molann_plan_feature_dim(None) < 0
for code in range(-12, 3):
    assert isinstance(L.molann_error_string(code), bytes)
# the host instantiation of the per-frame math
H = (ctypes.c_double * 9)(1.0, 0.2, -0.1, 0.3, 0.9, 0.05, -0.2, 0.1, 1.1)
R = (ctypes.c_float * 9)()
assert L.molann_selftest_kabsch_rotation(H, ctypes.c_double(3.0), R) == 0
Hf = (ctypes.c_float * 9)(*list(H))
assert L.molann_selftest_kabsch_rotation_f32(Hf, ctypes.c_float(3.0), R) == 0
for bad in ((0.0,) * 9, (float("nan"),) * 9, (1e300,) * 9):
    Hb = (ctypes.c_double * 9)(*bad)
    assert L.molann_selftest_kabsch_rotation(Hb, ctypes.c_double(1.0), R) == 0
atoms = (ctypes.c_float * 12)(*[0.3 * i * ((-1) ** i) for i in range(12)])
out3, g3, ga = (ctypes.c_float * 3)(), (ctypes.c_float * 3)(1.0, 0.5, -2.0), (ctypes.c_float * 12)()
for t in range(-1, 7):
    for uav in (0, 1):
        L.molann_selftest_feature(t, uav, atoms, out3)
        L.molann_selftest_feature_backward(t, uav, atoms, g3, ga)
for act in range(-1, 10):
    for v in (-30.0, -1.0, 0.0, 1e-8, 2.0, 50.0, float("inf"), float("nan")):
        L.molann_selftest_activation(act, ctypes.c_float(v))
        L.molann_selftest_act_derivative(act, ctypes.c_float(v))
GR, GH = (ctypes.c_float * 9)(*[0.1 * i for i in range(9)]), (ctypes.c_float * 9)()
assert L.molann_selftest_kabsch_backward(H, R, GR, GH) == 0
print("san_driver ok: %d sources generated, %d descriptions rejected" % (n_ok, n_rej))
sys.exit(0)
