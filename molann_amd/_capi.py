"""ctypes binding of ``libmolann_hip.so`` (the C ABI declared in ``include/molann_hip.h``).

Nothing here computes anything: it marshals index lists into a ``molann_plan_desc``, keeps the opaque
plan alive, and passes raw device pointers + the current HIP stream to the launch functions.  If the
library is missing the import of this module fails loudly; there is no CPU fallback.
"""

import ctypes
import os
import re
import subprocess

import torch  # imported first so that libamdhip64.so.7 is torch's copy (same SONAME, loaded once)

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_CSRC, "libmolann_hip.so")
# the diagnostics build (honours MOLANN_DEBUG_* / MOLANN_ELIDE_INVARIANT_ALIGNMENT): tools/ only, opt-in by name
DIAG_LIB_PATH = os.path.join(_CSRC, "libmolann_hip_diag.so")
SAN_LIB_PATH = os.path.join(_CSRC, "libmolann_hip_san.so")   # `make san`: host code under ASan + UBSan (CPU tests only)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "molann_hip.h")

ABI_VERSION = 1
FEAT_ANGLE, FEAT_BOND, FEAT_DIHEDRAL, FEAT_POSITION = 0, 1, 2, 3
ACT_TANH, ACT_RELU, ACT_SIGMOID, ACT_IDENTITY, ACT_ELU, ACT_SILU, ACT_SOFTPLUS, ACT_LEAKY_RELU, ACT_GELU = range(9)
MLP_F32, MLP_BF16 = 0, 1
MAX_LAYERS = 16

E_NULL, E_DESC, E_INDEX, E_FEATURE, E_STAGE, E_ALIGNMENT, E_UNSUPPORTED, E_NOT_PACKED, E_DEVICE = range(-1, -10, -1)


class MolannHipError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        super(MolannHipError, self).__init__("%s failed: [%d] %s" % (what, code, error_string(code)))


class PlanDesc(ctypes.Structure):
    _fields_ = [
        ("abi_version", ctypes.c_int32), ("n_inp", ctypes.c_int32),
        ("n_align", ctypes.c_int32), ("align_idx", ctypes.POINTER(ctypes.c_int32)),
        ("ref_x", ctypes.POINTER(ctypes.c_float)),
        ("n_features", ctypes.c_int32), ("feat_type", ctypes.POINTER(ctypes.c_int32)),
        ("feat_ptr", ctypes.POINTER(ctypes.c_int32)), ("feat_idx", ctypes.POINTER(ctypes.c_int32)),
        ("use_angle_value", ctypes.c_int32),
        ("n_layers", ctypes.c_int32), ("layer_dims", ctypes.POINTER(ctypes.c_int32)),
        ("activation", ctypes.c_int32), ("mlp_precision", ctypes.c_int32),
    ]


def build_library(force=False):
    """Compile csrc/: libmolann_hip.so (hipcc, gfx950, seconds) and libmolann_torch.so (g++, the TorchScript
    operators over the same ABI).  make decides what is stale.  Used by __graft_entry__.build()."""
    if force:
        subprocess.check_call(["make", "-C", _CSRC, "clean"])
    subprocess.check_call(["make", "-C", _CSRC, "all"])
    return LIB_PATH


_lib = None


def lib():
    """The loaded library (raises if it has not been built: no silent fallback)."""
    global _lib
    if _lib is None:
        path = DIAG_LIB_PATH if os.environ.get("MOLANN_DIAG_LIB") == "1" else LIB_PATH
        if os.environ.get("MOLANN_SAN_LIB") == "1":   # tests/test_sanitized_host.py: the ASan + UBSan build of the host half
            path = SAN_LIB_PATH
        if not os.path.exists(path):
            raise ImportError("%s not found: run `make -C %s%s` (or __graft_entry__.build())"
                              % (path, _CSRC, " diag" if path == DIAG_LIB_PATH else (" san" if path == SAN_LIB_PATH else "")))
        L = ctypes.CDLL(path)
        vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
        sigs = {
            "molann_abi_version": (i32, []),
            "molann_error_string": (ctypes.c_char_p, [i32]),
            "molann_build_kind": (ctypes.c_char_p, []),
            "molann_plan_create": (i32, [ctypes.POINTER(PlanDesc), ctypes.POINTER(vp)]),
            "molann_plan_destroy": (i32, [vp]),
            "molann_plan_feature_dim": (i32, [vp]),
            "molann_plan_out_dim": (i32, [vp]),
            "molann_plan_kernel_family": (i32, [vp]),
            "molann_plan_update_ref": (i32, [vp, vp, vp]),
            "molann_plan_update_mlp": (i32, [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), vp]),
            "molann_align_f32": (i32, [vp, vp, i64, vp, vp]),
            "molann_features_f32": (i32, [vp, vp, i64, vp, vp]),
            "molann_forward_packed_f32": (i32, [vp, vp, i64, vp, vp]),
            "molann_forward_f32": (i32, [vp, vp, i64, ctypes.POINTER(vp), ctypes.POINTER(vp), vp, vp]),
            "molann_mlp_packed_f32": (i32, [vp, vp, i64, vp, vp]),
            "molann_plan_update_ref_f64": (i32, [vp, vp, vp]),
            "molann_align_f64": (i32, [vp, vp, i64, vp, vp]),
            "molann_features_f64": (i32, [vp, vp, i64, vp, vp]),
            "molann_mlp_f64": (i32, [vp, vp, i64, ctypes.POINTER(vp), ctypes.POINTER(vp), vp, vp]),
            "molann_forward_f64": (i32, [vp, vp, i64, ctypes.POINTER(vp), ctypes.POINTER(vp), vp, vp, vp]),
            "molann_plan_last_launch_info": (i32, [vp, ctypes.c_char_p, i32]),
            "molann_plan_grad_params_size": (i32, [vp]),
            "molann_plan_supports_backward": (i32, [vp]),
            "molann_plan_backward_kind": (i32, [vp]),
            "molann_backward_f32": (i32, [vp, vp, vp, i64, vp, vp, vp]),
            "molann_value_and_vjp_f32": (i32, [vp, vp, vp, i64, vp, vp, vp]),
            "molann_forward_train_f32": (i32, [vp, vp, i64, vp, vp, vp]),
            "molann_features_backward_f64": (i32, [vp, vp, vp, i64, vp, vp]),
            "molann_features_backward_f32": (i32, [vp, vp, vp, i64, vp, vp]),
            "molann_mlp_backward_f32": (i32, [vp, vp, vp, i64, vp, vp, vp]),
            "molann_debug_read_stamps": (i32, [vp]),
            "molann_debug_jit": (i32, [ctypes.POINTER(PlanDesc), i32, ctypes.c_char_p, i32]),
            "molann_selftest_kabsch_rotation": (i32, [vp, ctypes.c_double, vp]),
            "molann_selftest_kabsch_rotation_f32": (i32, [vp, f32, vp]),
            "molann_selftest_feature": (i32, [i32, i32, vp, vp]),
            "molann_selftest_activation": (f32, [i32, f32]),
            "molann_selftest_feature_backward": (i32, [i32, i32, vp, vp, vp]),
            "molann_selftest_kabsch_backward": (i32, [vp, vp, vp, vp]),
            "molann_selftest_act_derivative": (f32, [i32, f32]),
        }
        for name, (res, args) in sigs.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        if L.molann_abi_version() != ABI_VERSION:
            raise ImportError("libmolann_hip.so ABI %d != binding ABI %d" % (L.molann_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


def declared_symbols():
    """Every function name declared in include/molann_hip.h."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(molann_[a-z0-9_]+)\s*\(", text)))


def build_kind():
    """"release" or "diagnostics" (see include/molann_hip.h)."""
    return lib().molann_build_kind().decode()


def error_string(code):
    return lib().molann_error_string(int(code)).decode()


def _check(code, what):
    if code != 0:
        raise MolannHipError(code, what)


def _i32_array(values):
    arr = (ctypes.c_int32 * max(1, len(values)))(*[int(v) for v in values])
    return arr


class Plan(object):
    """Owns one ``molann_plan``.  All index lists are positions inside the n_inp axis."""

    def __init__(self, n_inp, align_idx=None, ref_x=None, features=None, use_angle_value=False,
                 layer_dims=None, activation=ACT_TANH, mlp_precision=MLP_F32):
        self._handle = None
        L = lib()
        d = PlanDesc()
        d.abi_version = ABI_VERSION
        d.n_inp = int(n_inp)
        keep = []
        if align_idx is not None and len(align_idx) > 0:
            a = _i32_array(align_idx)
            r = torch.as_tensor(ref_x, dtype=torch.float32).detach().cpu().contiguous().reshape(-1)
            if r.numel() != 3 * len(align_idx):
                raise ValueError("ref_x must be [n_align, 3]")
            rbuf = (ctypes.c_float * r.numel())(*r.tolist())
            d.n_align, d.align_idx, d.ref_x = len(align_idx), a, rbuf
            keep += [a, rbuf]
        features = list(features or [])
        if features:
            ft = _i32_array([t for t, _ in features])
            ptr, flat = [0], []
            for _, idx in features:
                flat.extend(int(i) for i in idx)
                ptr.append(len(flat))
            fp, fi = _i32_array(ptr), _i32_array(flat)
            d.n_features, d.feat_type, d.feat_ptr, d.feat_idx = len(features), ft, fp, fi
            keep += [ft, fp, fi]
        d.use_angle_value = 1 if use_angle_value else 0
        if layer_dims:
            ld = _i32_array(layer_dims)
            d.n_layers, d.layer_dims = len(layer_dims) - 1, ld
            keep.append(ld)
        d.activation = int(activation)
        d.mlp_precision = int(mlp_precision)
        h = ctypes.c_void_p()
        _check(L.molann_plan_create(ctypes.byref(d), ctypes.byref(h)), "molann_plan_create")
        self._handle = h
        self.n_inp = int(n_inp)
        self.n_align = int(d.n_align)
        self.n_layers = int(d.n_layers)
        self.activation = int(activation)
        self.feature_dim = L.molann_plan_feature_dim(h)
        self.out_dim = L.molann_plan_out_dim(h)
        self.kernel_family = L.molann_plan_kernel_family(h)
        self.device = torch.cuda.current_device()

    def close(self):
        if self._handle is not None and _lib is not None:
            _lib.molann_plan_destroy(self._handle)
        self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    @staticmethod
    def _stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def update_ref(self, ref_x):
        _check(lib().molann_plan_update_ref(self._handle, ctypes.c_void_p(ref_x.data_ptr()), self._stream()),
               "molann_plan_update_ref")

    def update_ref_f64(self, ref_x):
        _check(lib().molann_plan_update_ref_f64(self._handle, ctypes.c_void_p(ref_x.data_ptr()), self._stream()),
               "molann_plan_update_ref_f64")

    def align_f64(self, x, out):
        return self._run("molann_align_f64", x, out, x.shape[0])

    def features_f64(self, x, out):
        return self._run("molann_features_f64", x, out, x.shape[0])

    def forward_f64(self, x, weights, biases, work, out):
        n = len(weights)
        W = (ctypes.c_void_p * n)(*[w.data_ptr() for w in weights])
        B = (ctypes.c_void_p * n)(*[b.data_ptr() for b in biases])
        _check(lib().molann_forward_f64(self._handle, x.data_ptr(), x.shape[0], W, B, work.data_ptr(), out.data_ptr(),
                                        self._stream()), "molann_forward_f64")
        return out

    def features_backward_f64(self, x, grad_f, grad_x):
        code = _lib.molann_features_backward_f64(self._handle, x.data_ptr(), grad_f.data_ptr(), x.shape[0], grad_x.data_ptr(),
                                                 torch.cuda.current_stream().cuda_stream)
        if code != 0:
            raise MolannHipError(code, "molann_features_backward_f64")

    def mlp_f64(self, f, weights, biases, out):
        n = len(weights)
        W = (ctypes.c_void_p * n)(*[w.data_ptr() for w in weights])
        B = (ctypes.c_void_p * n)(*[b.data_ptr() for b in biases])
        _check(lib().molann_mlp_f64(self._handle, f.data_ptr(), f.shape[0], W, B, out.data_ptr(), self._stream()), "molann_mlp_f64")
        return out

    def update_mlp(self, weights, biases):
        n = len(weights)
        W = (ctypes.c_void_p * n)(*[w.data_ptr() for w in weights])
        B = (ctypes.c_void_p * n)(*[b.data_ptr() for b in biases])
        _check(lib().molann_plan_update_mlp(self._handle, W, B, self._stream()), "molann_plan_update_mlp")

    def _run(self, fn_name, x, out, n):
        code = getattr(_lib, fn_name)(self._handle, x.data_ptr(), n, out.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream)
        if code != 0:
            raise MolannHipError(code, fn_name)
        return out

    def align(self, x, out):
        return self._run("molann_align_f32", x, out, x.shape[0])

    def features(self, x, out):
        return self._run("molann_features_f32", x, out, x.shape[0])

    def forward_packed(self, x, out):
        return self._run("molann_forward_packed_f32", x, out, x.shape[0])

    def mlp_packed(self, f, out):
        return self._run("molann_mlp_packed_f32", f, out, f.shape[0])

    def supports_backward(self):
        return lib().molann_plan_supports_backward(self._handle) == 1

    def backward_kind(self):
        """2: `backward` is one pass over x; 1: several launches (keep the features of the forward: `forward_train`); 0: none."""
        return lib().molann_plan_backward_kind(self._handle)

    def grad_params_size(self):
        return lib().molann_plan_grad_params_size(self._handle)

    def backward(self, x, grad_out, grad_x, grad_params):
        """grad_x / grad_params may be None; grad_params is accumulated into."""
        code = _lib.molann_backward_f32(self._handle, x.data_ptr(), grad_out.data_ptr(), x.shape[0],
                                        grad_x.data_ptr() if grad_x is not None else None,
                                        grad_params.data_ptr() if grad_params is not None else None,
                                        torch.cuda.current_stream().cuda_stream)
        if code != 0:
            raise MolannHipError(code, "molann_backward_f32")

    def value_and_vjp(self, x, grad_out, out, grad_x):
        """out = forward(x) and grad_x = the vector-Jacobian product for `grad_out`, one launch (plans with backward_kind() == 2)."""
        code = _lib.molann_value_and_vjp_f32(self._handle, x.data_ptr(), grad_out.data_ptr(), x.shape[0], out.data_ptr(), grad_x.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream)
        if code != 0:
            raise MolannHipError(code, "molann_value_and_vjp_f32")
        return out, grad_x

    def forward_train(self, x, out, features):
        """`forward_packed` that also keeps the features (for `mlp_backward` + `features_backward`)."""
        code = _lib.molann_forward_train_f32(self._handle, x.data_ptr(), x.shape[0], out.data_ptr(), features.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream)
        if code != 0:
            raise MolannHipError(code, "molann_forward_train_f32")
        return out

    def features_backward(self, x, grad_f, grad_x):
        """dL/dx of `features` for the same x."""
        code = _lib.molann_features_backward_f32(self._handle, x.data_ptr(), grad_f.data_ptr(), x.shape[0], grad_x.data_ptr(),
                                                 torch.cuda.current_stream().cuda_stream)
        if code != 0:
            raise MolannHipError(code, "molann_features_backward_f32")

    def mlp_backward(self, f, grad_out, grad_f, grad_params):
        """Backward of `mlp_packed` for the same f; grad_f / grad_params may be None; grad_params is accumulated into."""
        code = _lib.molann_mlp_backward_f32(self._handle, f.data_ptr(), grad_out.data_ptr(), f.shape[0],
                                            grad_f.data_ptr() if grad_f is not None else None,
                                            grad_params.data_ptr() if grad_params is not None else None,
                                            torch.cuda.current_stream().cuda_stream)
        if code != 0:
            raise MolannHipError(code, "molann_mlp_backward_f32")

    def last_launch_info(self):
        buf = ctypes.create_string_buffer(256)
        lib().molann_plan_last_launch_info(self._handle, buf, 256)
        return buf.value.decode()


def workload_desc(w):
    """``(PlanDesc, keep)`` for a `molann_amd.workloads.Workload` (0-based indices, zero reference coordinates): what
    `molann_debug_jit` cross-compiles a plan's specialised kernels from without a device.  `keep` holds the ctypes arrays
    the description points into."""
    d = PlanDesc()
    d.abi_version, d.n_inp = ABI_VERSION, w.n_atoms
    keep = []
    I = lambda v: (ctypes.c_int32 * max(1, len(v)))(*v)  # noqa: E731
    if w.align:
        a, r = I([x - 1 for x in w.align]), (ctypes.c_float * (3 * len(w.align)))()
        d.n_align, d.align_idx, d.ref_x = len(w.align), a, r
        keep += [a, r]
    ptr, flat = [0], []
    for _, atoms in w.features:
        flat += [x - 1 for x in atoms]
        ptr.append(len(flat))
    ft, fp, fi = I([t for t, _ in w.features]), I(ptr), I(flat)
    d.n_features, d.feat_type, d.feat_ptr, d.feat_idx = len(w.features), ft, fp, fi
    keep += [ft, fp, fi]
    if w.mlp_dims:
        ld = I(w.mlp_dims)
        d.n_layers, d.layer_dims = len(w.mlp_dims) - 1, ld
        keep.append(ld)
    return d, keep
