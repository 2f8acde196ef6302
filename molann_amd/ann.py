"""`molann.ann`'s module API on the MI355X: same classes, constructor signatures, attributes,
error behaviour and state_dict keys; `forward` runs hand-written gfx950 kernels through the C ABI
in ``libmolann_hip.so`` (see ``include/molann_hip.h``).

    AlignmentLayer(align_atom_group, input_atom_group)           ann.py:69-199
    FeatureMap(feature, input_atom_group, use_angle_value)       ann.py:201-356
    FeatureLayer(feature_list, input_atom_group, use_angle_value) ann.py:358-474
    PreprocessingANN(align_layer, feature_layer)                 ann.py:476-565
    MolANN(preprocessing_layer, ann_layers)                      ann.py:567-624
    create_sequential_nn(layer_dims, activation)                 ann.py:37-67

There is no CPU or composite-PyTorch fallback: a forward on anything but a float32 or float64 tensor that
lives on a HIP device raises.  float64 (`model.double()(x.double())`, which the reference supports because its
modules follow x.dtype) runs the `molann_*_f64` kernels; under grad mode its features are differentiated by
`molann_features_backward_f64` and its MLP runs as the torch module it is.  float32 gradients (w.r.t. x and the Linear parameters) come from hand-written backward
kernels: fused with the MLP for the plans the lane-per-frame kernel serves (22-atom class, MLP widths <= 32),
features only on large frames (one wave per frame).  With an MLP outside the fused kernel (wider, ELU / GELU /
Softplus, or any MLP on large frames) a forward under grad mode takes features and their gradient from the
kernels and runs ``ann_layers`` as the torch module it is.
"""

import torch
import pandas as pd

from . import _capi

_ACT_CODES = (
    (torch.nn.Tanh, _capi.ACT_TANH, lambda m: True),
    (torch.nn.ReLU, _capi.ACT_RELU, lambda m: True),
    (torch.nn.Sigmoid, _capi.ACT_SIGMOID, lambda m: True),
    (torch.nn.Identity, _capi.ACT_IDENTITY, lambda m: True),
    (torch.nn.ELU, _capi.ACT_ELU, lambda m: m.alpha == 1.0),
    (torch.nn.SiLU, _capi.ACT_SILU, lambda m: True),
    (torch.nn.Softplus, _capi.ACT_SOFTPLUS, lambda m: m.beta == 1.0 and m.threshold == 20.0),
    (torch.nn.LeakyReLU, _capi.ACT_LEAKY_RELU, lambda m: m.negative_slope == 0.01),
    (torch.nn.GELU, _capi.ACT_GELU, lambda m: getattr(m, "approximate", "none") == "none"),
)


def create_sequential_nn(layer_dims, activation=torch.nn.Tanh()):
    """Feed-forward network ``Linear -> act -> ... -> Linear`` (no activation after the last layer).

    Module names follow the reference (`ann.py:63-65`) so that state_dict keys are interchangeable:
    ``'{i}th_layer'`` and ``'activation of {i}th_layer'``; one activation object is shared.
    """
    assert len(layer_dims) >= 2, 'Error: at least 2 layers are needed to define a neural network (length={})!'.format(len(layer_dims))
    net = torch.nn.Sequential()
    n_linear = len(layer_dims) - 1
    for i in range(1, n_linear + 1):
        net.add_module('%dth_layer' % i, torch.nn.Linear(layer_dims[i - 1], layer_dims[i]))
        if i < n_linear:
            net.add_module('activation of %dth_layer' % i, activation)
    return net


def _activation_code(module):
    for cls, code, ok in _ACT_CODES:
        if type(module) is cls and ok(module):
            return code
    return None


def recognise_mlp(ann_layers):
    """``(linears, activation_code)`` if ``ann_layers`` is a Sequential of the shape
    `create_sequential_nn` builds with an activation the kernels implement, else ``None``.
    Walks ``_modules`` (``children()`` de-duplicates the shared activation object)."""
    if not isinstance(ann_layers, torch.nn.Sequential):
        return None
    mods = list(ann_layers._modules.values())
    if not mods or len(mods) % 2 == 0:
        return None
    linears, code = [], None
    for i, m in enumerate(mods):
        if i % 2 == 0:
            if type(m) is not torch.nn.Linear or m.bias is None:
                return None
            linears.append(m)
        else:
            c = _activation_code(m)
            if c is None or (code is not None and c != code):
                return None
            code = c
    for a, b in zip(linears[:-1], linears[1:]):
        if a.out_features != b.in_features:
            return None
    if len(linears) > _capi.MAX_LAYERS:
        return None
    return linears, (_capi.ACT_IDENTITY if code is None else code)


_RUN_OP = []


def _run_op():
    """``torch.ops.molann.run`` if csrc/libmolann_torch.so has been built, else None (then ctypes serves every call)."""
    if not _RUN_OP:
        import os
        if os.environ.get("MOLANN_DIAG_LIB") == "1":   # tools/ on the diagnostics build: the operator library is
            _RUN_OP.append(None)                        # linked against the release build, so ctypes serves every call
            return None
        try:
            from . import script
            script.load_ops()
            _RUN_OP.append(torch.ops.molann.run)
        except ImportError:
            _RUN_OP.append(None)
    return _RUN_OP[0]


def _local_indices(input_indices, wanted, what):
    try:
        return [input_indices.index(int(i)) for i in wanted]
    except ValueError:
        raise ValueError(what)


def _check_input(x, input_atom_num):
    assert isinstance(x, torch.Tensor), 'Input x is not a torch tensor'
    assert x.size(1) == input_atom_num and x.size(2) == 3, \
        f'Input should be a 3d torch tensor, with sizes [*, {input_atom_num}, 3]. Actual sizes: {x.shape}'


def _wants_grad(x, grad_sources=()):
    return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in grad_sources))


def _device_input(x, grad_sources=(), backward_ok=False):
    """The tensor the kernels read: float32 or float64, on a HIP device, contiguous.  Everything else raises."""
    if not x.is_cuda:
        raise RuntimeError("molann_amd runs on the MI355X only: got a %s tensor (no CPU path; move x and the "
                           "module to a HIP device)" % x.device.type)
    if x.dtype not in (torch.float32, torch.float64):
        raise TypeError("molann_amd kernels are float32 / float64; got %s" % x.dtype)
    if not backward_ok and _wants_grad(x, grad_sources):
        raise NotImplementedError("no backward kernel for this module / plan yet: call it under "
                                  "torch.no_grad() (or freeze the parameters)")
    return x if x.is_contiguous() else x.contiguous()


def _release_plans(desc, device):
    """weakref.finalize callback of a MolANN: its plans in the operator library's cache go with it."""
    try:
        torch.ops.molann.release(desc, device)
    except Exception:   # interpreter shutdown / library gone
        pass


class _PlanFunction(torch.autograd.Function):
    """forward = one fused launch of the plan; backward = molann_backward_f32: one pass over x that recomputes the forward
    per frame (nothing but x is saved), runs the MLP's backward on the matrix cores and the analytic reverse mode of
    the preprocessing.  Where that kernel could not be built (`backward_kind() == 1`) the forward also keeps the
    features and the backward is two launches, molann_mlp_backward_f32 and molann_features_backward_f32.
    `params` are the Linear weights/biases in layer order (may be empty)."""

    @staticmethod
    def forward(ctx, x, entry, with_mlp, *params):
        plan = entry.plan
        out = torch.empty((x.shape[0], plan.out_dim if with_mlp else plan.feature_dim), dtype=torch.float32, device=x.device)
        feat = None
        # What the backward will need is known here.  Gradients for x: one pass over x does everything (backward_kind 2),
        # nothing to keep.  Parameters only (x is data): the MLP's backward alone, on the features this forward keeps
        # (C3: 63 us per 1 M frames instead of 104 for the pass over x).
        if with_mlp and entry.backward_kind() != 1 and ctx.needs_input_grad[0]:
            plan.forward_packed(x, out)
        elif with_mlp:
            feat = torch.empty((x.shape[0], plan.feature_dim), dtype=torch.float32, device=x.device)
            try:
                plan.forward_train(x, out, feat)
            except _capi.MolannHipError as e:       # no feature-keeping twin of this plan's kernel: the backward recomputes
                if e.code != _capi.E_UNSUPPORTED:
                    raise
                feat = None
                plan.forward_packed(x, out)
        else:
            plan.features(x, out)
        if feat is None:
            ctx.save_for_backward(x)
        else:
            ctx.save_for_backward(x, feat)
        ctx.entry, ctx.shapes = entry, [tuple(p.shape) for p in params]
        ctx.with_mlp, ctx.params = with_mlp, params      # (create_graph=True: the backward is then rebuilt as a differentiable composition)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        # Grad mode is on inside a backward only under create_graph=True: the caller wants to differentiate these gradients
        # again (a loss on forces; the reference can, through plain autograd incl. its SVD, ann.py:188-197).  The kernels'
        # results carry no graph, so the gradient is rebuilt as a DIFFERENTIABLE composition: the float64 features with their
        # double-differentiable backward (_FeatBackward64), the MLP as ATen ops on the live parameters.
        if torch.is_grad_enabled():
            return _double_backward(ctx, grad_out)
        x = ctx.saved_tensors[0]
        feat = ctx.saved_tensors[1] if len(ctx.saved_tensors) > 1 else None
        plan = ctx.entry.plan
        need_x = ctx.needs_input_grad[0]
        need_p = any(ctx.needs_input_grad[3:])
        gx = torch.empty_like(x) if need_x else None
        gp = torch.zeros(plan.grad_params_size(), dtype=torch.float32, device=x.device) if need_p else None
        g = grad_out.contiguous()
        if g.dtype != torch.float32:
            g = g.float()
        with torch.cuda.device(x.device):
            if feat is None:
                plan.backward(x, g, gx, gp)
            else:
                gf = torch.empty_like(feat) if need_x else None
                plan.mlp_backward(feat, g, gf, gp)
                if need_x:
                    plan.features_backward(x, gf, gx)
        grads, off = [], 0
        for i, shp in enumerate(ctx.shapes):
            n = 1
            for d in shp:
                n *= d
            grads.append(gp[off:off + n].view(shp) if (need_p and ctx.needs_input_grad[3 + i]) else None)
            off += n
        return (gx, None, None) + tuple(grads)


class _PlanFunction64(torch.autograd.Function):
    """The float64 features of a plan (`model.double()`): forward = molann_features_f64, backward =
    molann_features_backward_f64 (everything recomputed in double from x).  The MLP of a float64 model is torch's."""

    @staticmethod
    def forward(ctx, x, entry):
        out = torch.empty((x.shape[0], entry.plan.feature_dim), dtype=torch.float64, device=x.device)
        entry.plan.features_f64(x, out)
        ctx.save_for_backward(x)
        ctx.entry = entry
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (x,) = ctx.saved_tensors
        g = grad_out.contiguous()
        if g.dtype != torch.float64:
            g = g.double()
        if torch.is_grad_enabled():          # create_graph=True: the same product, as a node that can be differentiated again
            return _FeatBackward64.apply(x, g, ctx.entry), None
        gx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            ctx.entry.plan.features_backward_f64(x, g, gx)
        return gx, None


class _FeatBackward64(torch.autograd.Function):
    """``gx = J(x)^T g`` of the float64 features (`molann_features_backward_f64`) as a node that can itself be differentiated -
    what ``create_graph=True`` needs (second-order terms of a loss on forces; the reference gets them from autograd through its
    SVD, `ann.py:188-197`).  For a cotangent ``v`` on ``gx`` its backward needs ``d/dx [v . J(x)^T g]`` and ``d/dg [v . J(x)^T g] = J(x) v``:
    both are directional derivatives along ``v`` - of the first-order kernel's own output and of the features - taken as CENTRAL
    DIFFERENCES of the float64 kernels, per frame with step ``h = 6e-6 * max(1, |x|_max) / |v|_max`` (the optimum for a
    second-order formula in double: truncation ~ h^2, rounding ~ 1e-16 / h; agreement with the reference's analytic double
    backward ~1e-9 of scale, `tests/test_gpu_backward.py::test_double_backward_*`).  Four launches; itself first-order."""

    @staticmethod
    def forward(ctx, x, g, entry):
        gx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            entry.plan.features_backward_f64(x, g, gx)
        ctx.save_for_backward(x, g)
        ctx.entry = entry
        return gx

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, v):
        x, g = ctx.saved_tensors
        plan = ctx.entry.plan
        v = v.contiguous().double()
        vmax = v.abs().amax(dim=(1, 2), keepdim=True)
        xmax = x.abs().amax(dim=(1, 2), keepdim=True).clamp_(min=1.0)
        h = torch.where(vmax > 0, 6e-6 * xmax / vmax.clamp(min=1e-300), torch.zeros_like(vmax))
        xp, xm = (x + h * v).contiguous(), (x - h * v).contiguous()
        inv = torch.where(h > 0, 0.5 / h.clamp(min=1e-300), torch.zeros_like(h))
        gxp, gxm = torch.empty_like(x), torch.empty_like(x)
        fp = torch.empty((x.shape[0], plan.feature_dim), dtype=torch.float64, device=x.device)
        fm = torch.empty_like(fp)
        with torch.cuda.device(x.device):
            if ctx.needs_input_grad[0]:
                plan.features_backward_f64(xp, g, gxp)
                plan.features_backward_f64(xm, g, gxm)
            if ctx.needs_input_grad[1]:
                plan.features_f64(xp, fp)
                plan.features_f64(xm, fm)
        grad_x = (gxp - gxm) * inv if ctx.needs_input_grad[0] else None
        grad_g = (fp - fm) * inv.view(-1, 1) if ctx.needs_input_grad[1] else None
        return grad_x, grad_g, None


_ACT_FNS = {
    _capi.ACT_TANH: torch.tanh, _capi.ACT_RELU: torch.relu, _capi.ACT_SIGMOID: torch.sigmoid, _capi.ACT_IDENTITY: (lambda t: t),
    _capi.ACT_ELU: torch.nn.functional.elu, _capi.ACT_SILU: torch.nn.functional.silu, _capi.ACT_SOFTPLUS: torch.nn.functional.softplus,
    _capi.ACT_LEAKY_RELU: torch.nn.functional.leaky_relu, _capi.ACT_GELU: torch.nn.functional.gelu,
}


def _double_backward(ctx, grad_out):
    """The gradients of a `_PlanFunction` node as a differentiable composition (create_graph=True): features in float64 through
    `_PlanFunction64` (whose backward is `_FeatBackward64`), the MLP as ATen ops on the live parameters, `torch.autograd.grad`
    with ``create_graph=True`` over it."""
    x = ctx.saved_tensors[0]
    entry, params = ctx.entry, list(ctx.params)
    with torch.enable_grad():
        y = _PlanFunction64.apply(x.double(), entry).to(x.dtype)
        if ctx.with_mlp:
            act = _ACT_FNS[entry.plan.activation]
            n = len(params) // 2
            for l in range(n):
                y = torch.nn.functional.linear(y, params[2 * l], params[2 * l + 1])
                if l + 1 < n:
                    y = act(y)
        wanted = [(0, x)] if ctx.needs_input_grad[0] else []
        wanted += [(3 + i, p) for i, p in enumerate(params) if ctx.needs_input_grad[3 + i]]
        got = torch.autograd.grad(y, [t for _, t in wanted], grad_out, create_graph=True, allow_unused=True) if wanted else ()
    out = [None] * (3 + len(params))
    for (i, _), gi in zip(wanted, got):
        out[i] = gi
    return tuple(out)


def _device_buffer(ref_x, x):
    """The module's `ref_x` buffer must live where x lives and have its dtype (the reference's matmul, ann.py:187,
    raises RuntimeError for mixed devices and for mixed dtypes too)."""
    if ref_x.device != x.device:
        raise RuntimeError("Expected all tensors to be on the same device: ref_x is on %s, x on %s "
                           "(move the module with .to(x.device))" % (ref_x.device, x.device))
    if ref_x.dtype != x.dtype:
        raise RuntimeError("expected ref_x and x to have the same dtype, but got ref_x %s and x %s (call .double() / "
                           ".float() on the module)" % (ref_x.dtype, x.dtype))
    return ref_x


class _PlanOwner(object):
    """Mixin: per-device cache of C-ABI plans, dropped on copy / pickling (plans hold device memory)."""

    def _plans(self):
        cache = self.__dict__.get("_plan_cache")
        if cache is None:
            cache = {}
            self.__dict__["_plan_cache"] = cache
        return cache

    def refresh_parameters(self):
        """Re-read `ref_x` and the Linear parameters at the next forward.  Needed only after writing them in a way
        their version counters do not show (`p.data.mul_()`, `p.data.copy_()`, a detached alias): ordinary in-place
        ops, optimizer steps, load_state_dict, .to() and replaced Parameters are seen without it."""
        for e in self._plans().values():
            if isinstance(e, _PlanEntry):
                e.invalidate()
        st = self.__dict__.get("_fast")
        if st is not None and st.get("desc") is not None and st.get("op") is not None:
            torch.ops.molann.invalidate(st["desc"], st["sig"][5])
        for m in self.children():
            if isinstance(m, _PlanOwner):
                m.refresh_parameters()

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_plan_cache", None)
        state.pop("_fast", None)
        state.pop("_fp", None)
        return state

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k not in ("_plan_cache", "_fast", "_fp"):
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new


class _TensorKey(object):
    """Which tensor was packed: the OBJECT (weak reference: an address the allocator reuses cannot pass for it), its
    storage address and its version counter.  In-place writes through `.data` bump neither: `refresh_parameters()`."""
    __slots__ = ("ref", "ptr", "version")

    def __init__(self, t):
        import weakref
        self.ref, self.ptr, self.version = weakref.ref(t), t.data_ptr(), t._version

    def matches(self, t):
        return self.ref() is t and self.ptr == t.data_ptr() and self.version == t._version


class _PlanEntry(object):
    """A plan plus the versions of the live tensors (ref_x buffer, Linear parameters) packed in it."""

    def __init__(self, plan):
        self.plan = plan
        self.ref_key = None
        self.mlp_key = None
        self._bwd_kind = None

    def backward_kind(self):
        if self._bwd_kind is None:
            self._bwd_kind = self.plan.backward_kind()     # a property of the plan: asked (and built) once
        return self._bwd_kind

    def invalidate(self):
        self.ref_key = self.mlp_key = None

    def sync_ref(self, ref_x):
        if self.ref_key is None or not self.ref_key.matches(ref_x):
            if ref_x.dtype == torch.float64:      # the buffer of a `.double()` model: kept in double
                r = ref_x.contiguous()
                self.plan.update_ref_f64(r)
            else:
                r = ref_x if (ref_x.dtype == torch.float32 and ref_x.is_contiguous()) else ref_x.float().contiguous()
                self.plan.update_ref(r)
            self.ref_key = _TensorKey(ref_x)
            self._ref_hold = r

    def sync_mlp(self, linears):
        params = [p for lin in linears for p in (lin.weight, lin.bias)]
        if self.mlp_key is None or len(self.mlp_key) != len(params) or \
                not all(k.matches(p) for k, p in zip(self.mlp_key, params)):
            ws = [lin.weight.detach().contiguous() for lin in linears]
            bs = [lin.bias.detach().contiguous() for lin in linears]
            self.plan.update_mlp(ws, bs)
            self.mlp_key = [_TensorKey(p) for p in params]


def _feature_spec(feature_layer_or_map):
    maps = feature_layer_or_map.feature_map_list if isinstance(feature_layer_or_map, FeatureLayer) else [feature_layer_or_map]
    spec = [(fm.type_id, list(fm._local_atom_indices)) for fm in maps]
    uav = bool(maps[0].use_angle_value)
    return spec, uav


def _get_entry(owner, x, tag, build):
    key = (tag, x.device.index)
    cache = owner._plans()
    entry = cache.get(key)
    if entry is None:
        with torch.cuda.device(x.device):
            entry = _PlanEntry(build())
        cache[key] = entry
    return entry


def last_launch_info(module):
    """Launch info of the plan a molann_amd module used last ('' before its first forward)."""
    if isinstance(module, MolANN) and module.__dict__.get("_fast", {}).get("op") is not None:
        return module.last_launch_info()
    cache = module._plans()
    entries = [e for e in cache.values() if isinstance(e, _PlanEntry)]
    return entries[-1].plan.last_launch_info() if entries else ""


class AlignmentLayer(_PlanOwner, torch.nn.Module):
    r"""Kabsch superposition of every frame onto the (centred) coordinates of ``align_atom_group``:
    :math:`x \mapsto (x - c(x)) R(x)`, all ``n_inp`` atoms returned (`ann.py:157-199`)."""

    def __init__(self, align_atom_group, input_atom_group):
        super(AlignmentLayer, self).__init__()
        self.align_atom_indices = align_atom_group.ix.tolist()
        self.input_atom_indices = input_atom_group.ix.tolist()
        self.input_atom_num = len(input_atom_group)
        ref_x = torch.from_numpy(align_atom_group.positions)
        self.register_buffer('ref_x', ref_x)
        self.ref_x = self.ref_x - torch.mean(self.ref_x, 0)       # centred once (ann.py:140-141)
        self._local_align_atom_indices = _local_indices(self.input_atom_indices, self.align_atom_indices,
                                                        "Atoms used for alignment must be among the input")

    def show_info(self):
        print(f'\n{self.input_atom_num} atoms used for input, (0-based) global indices: \n', self.input_atom_indices)
        print(f'\n{len(self._local_align_atom_indices)} atoms used for alignment, with (0-based) global indices: \n',
              self.align_atom_indices)
        print('local indices\n', self._local_align_atom_indices)
        print('\ncoordinates of reference state used in aligment:\n', self.ref_x.cpu().numpy())

    def __prepare_scriptable__(self):
        """`torch.jit.script(align)` (`test/test_molann.py:46`) compiles this instead, see molann_amd/script.py."""
        from . import script
        return script.ScriptPlan(script.make_desc(script.KIND_ALIGN, self.input_atom_num,
                                                  align_idx=self._local_align_atom_indices), ref_x=self.ref_x)

    def _entry(self, x):
        entry = _get_entry(self, x, "align", lambda: _capi.Plan(
            self.input_atom_num, align_idx=self._local_align_atom_indices, ref_x=self.ref_x))
        entry.sync_ref(_device_buffer(self.ref_x, x))
        return entry

    def forward(self, x):
        _check_input(x, self.input_atom_num)
        x = _device_input(x, backward_ok=True)
        if x.shape[0] == 0:
            return torch.empty_like(x)
        if _wants_grad(x):
            # the aligned frame == alignment + one position item per atom: that plan has a backward kernel
            def build():
                return _capi.Plan(self.input_atom_num, align_idx=self._local_align_atom_indices, ref_x=self.ref_x,
                                  features=[(_capi.FEAT_POSITION, list(range(self.input_atom_num)))])
            entry = _get_entry(self, x, "align_grad", build)
            with torch.cuda.device(x.device):
                entry.sync_ref(_device_buffer(self.ref_x, x))
                if x.dtype == torch.float64:
                    return _PlanFunction64.apply(x, entry).view(x.shape[0], self.input_atom_num, 3)
                if not entry.plan.supports_backward():
                    raise NotImplementedError("no backward kernel for this alignment plan (large frames): use torch.no_grad()")
                return _PlanFunction.apply(x, entry, False).view(x.shape[0], self.input_atom_num, 3)
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            plan = self._entry(x).plan
            if x.dtype == torch.float64:
                plan.align_f64(x, out)
            else:
                plan.align(x, out)
        return out


class FeatureMap(_PlanOwner, torch.nn.Module):
    """One feature (angle / bond / dihedral / position) of every frame (`ann.py:288-356`)."""

    def __init__(self, feature, input_atom_group, use_angle_value=False):
        super(FeatureMap, self).__init__()
        self.feature = feature
        self.type_id = feature.get_type_id()
        self.use_angle_value = use_angle_value
        self.input_atom_indices = input_atom_group.ix.tolist()
        self.input_atom_num = len(input_atom_group)
        self._local_atom_indices = _local_indices(self.input_atom_indices, feature.get_atom_indices() - 1,
                                                  "Atoms used in feature must be among the input")

    def dim(self):
        """1 for angle / bond / dihedral value, 2 for dihedral (cos, sin), 3k for k positions."""
        if self.type_id in (0, 1):
            return 1
        if self.type_id == 2:
            return 1 if self.use_angle_value == True else 2  # noqa: E712 (mirrors the reference's comparison)
        if self.type_id == 3:
            return 3 * len(self.feature.get_atom_indices())
        return 0

    def forward(self, x):
        _check_input(x, self.input_atom_num)
        return _run_features(self, x, None)

    def __prepare_scriptable__(self):
        return _script_features(self, None)


class FeatureLayer(_PlanOwner, torch.nn.Module):
    """All features of a list, concatenated column-wise in list order (`ann.py:454-474`)."""

    def __init__(self, feature_list, input_atom_group, use_angle_value=False):
        super(FeatureLayer, self).__init__()
        assert len(feature_list) > 0, 'Error: feature list is empty!'
        self.feature_list = feature_list
        self.feature_map_list = torch.nn.ModuleList([FeatureMap(f, input_atom_group, use_angle_value) for f in feature_list])
        self.input_atom_num = len(input_atom_group)

    def get_feature_info(self):
        return pd.concat([f.get_feature_info() for f in self.feature_list], ignore_index=True)

    def get_feature(self, idx):
        return self.feature_list[idx]

    def output_dimension(self):
        return sum([f_map.dim() for f_map in self.feature_map_list])

    def forward(self, x):
        _check_input(x, self.input_atom_num)
        return _run_features(self, x, None)

    def __prepare_scriptable__(self):
        return _script_features(self, None)


def _script_features(feature_owner, align_layer):
    """The ScriptPlan of a FeatureMap / FeatureLayer, optionally behind an AlignmentLayer."""
    from . import script
    spec, uav = _feature_spec(feature_owner)
    if align_layer is None:
        return script.ScriptPlan(script.make_desc(script.KIND_FEATURES, feature_owner.input_atom_num, features=spec,
                                                  use_angle_value=uav))
    return script.ScriptPlan(script.make_desc(script.KIND_FEATURES, feature_owner.input_atom_num,
                                              align_idx=align_layer._local_align_atom_indices, features=spec,
                                              use_angle_value=uav), ref_x=align_layer.ref_x)


def _run_features(feature_owner, x, align_layer, plan_owner=None):
    """features (optionally of the aligned frame) through one fused launch."""
    x = _device_input(x, backward_ok=True)
    spec, uav = _feature_spec(feature_owner)
    owner = plan_owner if plan_owner is not None else feature_owner

    def build():
        if align_layer is None:
            return _capi.Plan(feature_owner.input_atom_num, features=spec, use_angle_value=uav)
        return _capi.Plan(feature_owner.input_atom_num, align_idx=align_layer._local_align_atom_indices,
                          ref_x=align_layer.ref_x, features=spec, use_angle_value=uav)

    entry = _get_entry(owner, x, "features", build)
    if x.shape[0] == 0:
        return torch.empty((0, entry.plan.feature_dim), dtype=x.dtype, device=x.device)
    if x.dtype == torch.float64:
        with torch.cuda.device(x.device):
            if align_layer is not None:
                entry.sync_ref(_device_buffer(align_layer.ref_x, x))
            if _wants_grad(x):
                return _PlanFunction64.apply(x, entry)
            out = torch.empty((x.shape[0], entry.plan.feature_dim), dtype=torch.float64, device=x.device)
            entry.plan.features_f64(x, out)
        return out
    with torch.cuda.device(x.device):
        if align_layer is not None:
            entry.sync_ref(_device_buffer(align_layer.ref_x, x))
        if _wants_grad(x):
            if not entry.plan.supports_backward():
                raise NotImplementedError("no backward kernel for this plan (large frames): use torch.no_grad()")
            return _PlanFunction.apply(x, entry, False)
        out = torch.empty((x.shape[0], entry.plan.feature_dim), dtype=torch.float32, device=x.device)
        entry.plan.features(x, out)
    return out


class PreprocessingANN(_PlanOwner, torch.nn.Module):
    """``feature_layer(align_layer(x))``; ``align_layer=None`` means no alignment (`ann.py:533-565`)."""

    def __init__(self, align_layer, feature_layer):
        super(PreprocessingANN, self).__init__()
        self.align_layer = align_layer if align_layer is not None else torch.nn.Identity()
        self.feature_layer = feature_layer

    def output_dimension(self):
        return self.feature_layer.output_dimension()

    def _fusable(self):
        return isinstance(self.feature_layer, FeatureLayer) and \
            (isinstance(self.align_layer, AlignmentLayer) or type(self.align_layer) is torch.nn.Identity)

    def forward(self, x):
        if not self._fusable():
            return self.feature_layer(self.align_layer(x))
        al = self.align_layer if isinstance(self.align_layer, AlignmentLayer) else None
        if al is not None:
            _check_input(x, al.input_atom_num)
            assert al.input_atom_num == self.feature_layer.input_atom_num, \
                f'Input should be a 3d torch tensor, with sizes [*, {self.feature_layer.input_atom_num}, 3]. Actual sizes: {x.shape}'
        _check_input(x, self.feature_layer.input_atom_num)
        return _run_features(self.feature_layer, x, al, plan_owner=self)

    def __prepare_scriptable__(self):
        if not self._fusable():
            return torch.nn.Sequential(self.align_layer, self.feature_layer)
        al = self.align_layer if isinstance(self.align_layer, AlignmentLayer) else None
        return _script_features(self.feature_layer, al)


class MolANN(_PlanOwner, torch.nn.Module):
    """``ann_layers(preprocessing_layer(x))`` (`ann.py:606-624`).

    When ``ann_layers`` is a Sequential of Linear layers with one of the supported activations (what
    `create_sequential_nn` builds) the whole forward is one fused plan; any other module receives the
    features computed on the GPU.  ``mlp_precision='bf16'`` selects bf16 weights/activations with fp32
    accumulation on the bf16 MFMA for wide MLPs (the reference has no such mode).
    """

    def __init__(self, preprocessing_layer, ann_layers, mlp_precision="f32"):
        super(MolANN, self).__init__()
        self.preprocessing_layer = preprocessing_layer
        self.ann_layers = ann_layers
        assert mlp_precision in ("f32", "bf16")
        self.mlp_precision = mlp_precision

    def get_preprocessing_layer(self):
        return self.preprocessing_layer

    def __prepare_scriptable__(self):
        """`torch.jit.script(molann).save(...)` (`test/test_molann.py:114`, `README.rst:49`): the fused plan as
        one operator call when ann_layers is recognised, else the scripted preprocessing followed by ann_layers."""
        from . import script
        pp = self.preprocessing_layer
        rec = recognise_mlp(self.ann_layers)
        if rec is None or not (isinstance(pp, PreprocessingANN) and pp._fusable()):
            return torch.nn.Sequential(pp, self.ann_layers)
        linears, act = rec
        al = pp.align_layer if isinstance(pp.align_layer, AlignmentLayer) else None
        spec, uav = _feature_spec(pp.feature_layer)
        dims = [linears[0].in_features] + [lin.out_features for lin in linears]
        assert dims[0] == pp.feature_layer.output_dimension(), \
            'ann_layers expects %d inputs but the feature layer produces %d' % (dims[0], pp.feature_layer.output_dimension())
        desc = script.make_desc(script.KIND_FORWARD, pp.feature_layer.input_atom_num,
                                align_idx=al._local_align_atom_indices if al is not None else None, features=spec,
                                use_angle_value=uav, layer_dims=dims, activation=act,
                                mlp_precision=_capi.MLP_BF16 if self.mlp_precision == "bf16" else _capi.MLP_F32)
        return script.ScriptPlan(desc, ref_x=al.ref_x if al is not None else None, linears=linears)

    def plan_for(self, x):
        """The C-ABI plan (ctypes, `_capi.Plan`) of this model on x's device with `ref_x` and the Linear parameters packed:
        what tools/ and the tests of single entry points work on.  None if the model is not served by one fused plan."""
        st = self._fast_state(x)
        if not st["fused"]:
            return None
        entry = st["entry"]()
        with torch.cuda.device(x.device):
            if st["al"] is not None:
                entry.sync_ref(_device_buffer(st["al"].ref_x, x))
            entry.sync_mlp(st["linears"])
        return entry.plan

    def value_and_vjp(self, x, grad_out, into=None):
        """``(y, dx)`` with ``y = self(x)`` and ``dx = sum_k grad_out[:, k] d y[:, k] / d x`` in ONE kernel launch
        (`molann_value_and_vjp_f32`: the one-pass backward recomputes the forward anyway and, in this build, stores it too).
        For a caller that needs a collective variable and its forces at every step (`README.rst:49`): ~half the host time of a
        forward plus a backward.  The Jacobian of one frame: ``x.expand(d_out, -1, -1)`` with ``torch.eye(d_out)`` as cotangent.
        No autograd graph is recorded (parameters are data); ``into=(y, dx)`` reuses the caller's buffers.  Models served by one
        fused plan whose backward is the one-pass kernel, float32."""
        st = self._fast_state(x) if isinstance(x, torch.Tensor) and x.is_cuda else None
        if st is None or not st["fused"]:
            raise NotImplementedError("value_and_vjp needs a model served by one fused plan on a HIP device")
        al, fl = st["al"], st["fl"]
        _check_input(x, fl.input_atom_num)
        if x.dtype != torch.float32:
            raise TypeError("value_and_vjp is float32 only; got %s" % x.dtype)
        x = x.detach()
        x = x if x.is_contiguous() else x.contiguous()
        lins = st["linears"]
        if st["op"] is not None:
            y, dx = st["op_vjp"](x, st["handle"], _device_buffer(al.ref_x, x) if al is not None else st["no_ref"],
                                 [lin.weight for lin in lins], [lin.bias for lin in lins], grad_out, list(into) if into is not None else [])
            return y, dx
        entry = st["entry"]()
        with torch.cuda.device(x.device):
            if al is not None:
                entry.sync_ref(_device_buffer(al.ref_x, x))
            entry.sync_mlp(lins)
            y, dx = into if into is not None else (torch.empty((x.shape[0], st["out_dim"]), dtype=torch.float32, device=x.device), torch.empty_like(x))
            g = grad_out if (grad_out.dtype == torch.float32 and grad_out.is_contiguous()) else grad_out.float().contiguous()
            entry.plan.value_and_vjp(x, g, y, dx)
        return y, dx

    def last_launch_info(self):
        """Name + geometry of the kernels the last forward launched (bench / profiles / tests)."""
        st = self.__dict__.get("_fast")
        if st is not None and st.get("fused") and st.get("op") is not None:
            return torch.ops.molann.launch_info(st["desc"], st["sig"][5])
        return last_launch_info(self)

    def _fast_state(self, x):
        """Everything about this model that does not change from call to call (which modules it is made
        of, the recognised MLP, the plan), rebuilt only when the module tree or the device changes."""
        pp = self.preprocessing_layer
        nn = self.ann_layers
        fl = getattr(pp, "feature_layer", None)
        al = getattr(pp, "align_layer", None)
        sig = (id(pp), id(nn), id(fl), id(al), len(getattr(nn, "_modules", ())), x.device.index, self.mlp_precision)
        st = self.__dict__.get("_fast")
        if st is not None and st["sig"] == sig:
            return st
        rec = recognise_mlp(nn)
        st = {"sig": sig, "fused": False}
        if rec is not None and isinstance(pp, PreprocessingANN) and pp._fusable():
            linears, act = rec
            al = al if isinstance(al, AlignmentLayer) else None
            spec, uav = _feature_spec(fl)
            dims = [linears[0].in_features] + [lin.out_features for lin in linears]
            assert dims[0] == fl.output_dimension(), \
                'ann_layers expects %d inputs but the feature layer produces %d' % (dims[0], fl.output_dimension())

            def build():
                return _capi.Plan(fl.input_atom_num,
                                  align_idx=al._local_align_atom_indices if al is not None else None,
                                  ref_x=al.ref_x if al is not None else None,
                                  features=spec, use_angle_value=uav, layer_dims=dims, activation=act,
                                  mlp_precision=_capi.MLP_BF16 if self.mlp_precision == "bf16" else _capi.MLP_F32)

            tag = ("forward", tuple(dims), act, self.mlp_precision)
            st.update(fused=True, linears=linears, al=al, fl=fl, out_dim=dims[-1],
                      entry=lambda: _get_entry(self, x, tag, build),   # the ctypes plan, built when first needed
                      params=[p for lin in linears for p in (lin.weight, lin.bias)])
            # inference calls go through the dispatcher operator of csrc/molann_torch.cpp when that library is
            # built (same C ABI, same plan cache logic in C++): 8.5 us per call instead of 15.6 us through
            # ctypes for a 1024-frame batch (tools/latency_c1.py)
            st["op"] = _run_op()
            if st["op"] is not None:
                st["op_vjp"] = torch.ops.molann.value_and_vjp_h
                from . import script
                st["desc"] = script.make_desc(script.KIND_FORWARD, fl.input_atom_num,
                                              align_idx=al._local_align_atom_indices if al is not None else None,
                                              features=spec, use_angle_value=uav, layer_dims=dims, activation=act,
                                              mlp_precision=_capi.MLP_BF16 if self.mlp_precision == "bf16" else _capi.MLP_F32)
                st["no_ref"] = torch.zeros(0, 3)
                st["handle"] = torch.ops.molann.register_desc(st["desc"])
                import weakref
                weakref.finalize(self, _release_plans, st["desc"], x.device.index)
                # the inference fast path of forward(): everything it compares or passes, looked up once
                self.__dict__["_fp"] = (x.device, (fl.input_atom_num, 3), pp, nn, al, fl, len(nn._modules), st["handle"], torch.ops.molann.run_h,
                                        st["no_ref"], [lin._parameters for lin in linears], self.mlp_precision)
        if not (st["fused"] and st.get("op") is not None):
            self.__dict__.pop("_fp", None)
        self.__dict__["_fast"] = st
        return st

    def forward(self, x):
        # ---- inference fast path (a 1024-frame call is ~3 us of kernel: the host side is what a caller waits for; tools/latency_breakdown.py).
        # Taken only when nothing it skips could matter: the same module objects as when the plan was made, a float32 [N > 0, n_inp, 3]
        # tensor on the plan's device, nothing to record for autograd.  Everything else takes the general path below, checks and all.
        fp = self.__dict__.get("_fp")
        if fp is not None and type(x) is torch.Tensor and x.dtype is torch.float32 and x.device == fp[0] and x.dim() == 3 \
                and tuple(x.shape[1:]) == fp[1] and x.shape[0] > 0:
            mods = self._modules
            pp, nn = fp[2], fp[3]
            if mods["preprocessing_layer"] is pp and mods["ann_layers"] is nn and len(nn._modules) == fp[6] and self.mlp_precision == fp[11] \
                    and pp._modules["feature_layer"] is fp[5] and (pp._modules["align_layer"] is fp[4] or fp[4] is None and type(pp._modules["align_layer"]) is torch.nn.Identity):
                plist = fp[10]
                grad = torch.is_grad_enabled() and (x.requires_grad or any(d["weight"].requires_grad or d["bias"].requires_grad for d in plist))
                ref = fp[4]._buffers["ref_x"] if fp[4] is not None else fp[9]
                w0 = plist[0]["weight"]
                if not grad and w0.dtype is torch.float32 and w0.device == fp[0] and (fp[4] is None or (ref.dtype is torch.float32 and ref.device == fp[0])):
                    return fp[8](x, fp[7], ref, [d["weight"] for d in plist], [d["bias"] for d in plist])
        assert isinstance(x, torch.Tensor), 'Input x is not a torch tensor'
        st = self._fast_state(x) if x.is_cuda else None
        if st is None or not st["fused"]:
            if recognise_mlp(self.ann_layers) is None or not (isinstance(self.preprocessing_layer, PreprocessingANN)
                                                              and self.preprocessing_layer._fusable()):
                return self.ann_layers(self.preprocessing_layer(x))
            _check_input(x, self.preprocessing_layer.feature_layer.input_atom_num)
            _device_input(x)          # raises: not a device tensor
        al, fl = st["al"], st["fl"]
        if al is not None:
            _check_input(x, al.input_atom_num)
        _check_input(x, fl.input_atom_num)
        x = _device_input(x, grad_sources=st["params"], backward_ok=True)
        if x.shape[0] == 0:
            return torch.empty((0, st["out_dim"]), dtype=x.dtype, device=x.device)
        if x.dtype == torch.float64 and _wants_grad(x, st["params"]):
            # float64 training / forces: features and their gradient from the float64 kernels, ann_layers as the torch module it is
            return self.ann_layers(self.preprocessing_layer(x))
        if x.dtype == torch.float64:
            # `model.double()(x.double())`: the float64 kernels, the Linear parameters read as they are
            lins = st["linears"]
            w0 = lins[0].weight
            if w0.device != x.device or w0.dtype != torch.float64:
                raise RuntimeError("ann_layers must be float64 on %s for a float64 input (got %s on %s): call .double()"
                                   % (x.device, w0.dtype, w0.device))
            entry = st["entry"]()
            with torch.cuda.device(x.device):
                if al is not None:
                    entry.sync_ref(_device_buffer(al.ref_x, x))
                work = torch.empty((x.shape[0], entry.plan.feature_dim), dtype=torch.float64, device=x.device)
                out = torch.empty((x.shape[0], st["out_dim"]), dtype=torch.float64, device=x.device)
                entry.plan.forward_f64(x, [lin.weight.detach().contiguous() for lin in lins],
                                       [lin.bias.detach().contiguous() for lin in lins], work, out)
            return out
        if _wants_grad(x, st["params"]):
            w0 = st["linears"][0].weight
            if w0.device != x.device or w0.dtype != torch.float32:
                raise RuntimeError("ann_layers must be float32 on %s (got %s on %s)" % (x.device, w0.dtype, w0.device))
            if st.get("fused_bwd") is None:           # asked once: the answer is a property of the plan
                with torch.cuda.device(x.device):
                    if st["op"] is not None:
                        st["fused_bwd"] = bool(torch.ops.molann.supports_backward(
                            x, st["desc"], _device_buffer(al.ref_x, x) if al is not None else st["no_ref"]))
                    else:
                        st["fused_bwd"] = bool(st["entry"]().plan.supports_backward())
            if st["fused_bwd"] and st["op"] is not None:
                # the dispatcher operator's autograd node (csrc/molann_torch.cpp): the same kernels and the same choice between the
                # one-pass backward and the MLP's backward on kept features when x is data.  A training step costs ~100 us of
                # host time through it against ~195 through the Python autograd.Function below (tools/host_overhead.py)
                lins = st["linears"]
                return st["op"](x, st["desc"], _device_buffer(al.ref_x, x) if al is not None else st["no_ref"],
                                [lin.weight for lin in lins], [lin.bias for lin in lins])
            if st["fused_bwd"]:
                entry = st["entry"]()
                with torch.cuda.device(x.device):
                    if al is not None:
                        entry.sync_ref(_device_buffer(al.ref_x, x))
                    entry.sync_mlp(st["linears"])
                    return _PlanFunction.apply(x, entry, True, *st["params"])
            # No fused backward kernel (MLP wider than 32 / ELU, GELU, Softplus).  Training still works when the
            # preprocessing has one (small frames): features and their gradient from the HIP kernels, the MLP and
            # its gradient as the user's own torch module on the device - what already happens for ann_layers
            # this file does not recognise.  Large frames raise inside the preprocessing layer.
            return self.ann_layers(self.preprocessing_layer(x))
        w0 = st["linears"][0].weight
        if w0.device != x.device or w0.dtype != torch.float32:
            raise RuntimeError("ann_layers must be float32 on %s (got %s on %s)" % (x.device, w0.dtype, w0.device))
        if st["op"] is not None:
            lins = st["linears"]   # nothing here requires grad under an enabled grad mode: the operator records nothing
            return st["op"](x, st["desc"], _device_buffer(al.ref_x, x) if al is not None else st["no_ref"],
                            [lin.weight for lin in lins], [lin.bias for lin in lins])
        entry = st["entry"]()
        out = torch.empty((x.shape[0], st["out_dim"]), dtype=torch.float32, device=x.device)
        if x.device.index == torch.cuda.current_device():
            if al is not None:
                entry.sync_ref(_device_buffer(al.ref_x, x))
            entry.sync_mlp(st["linears"])
            entry.plan.forward_packed(x, out)
        else:
            with torch.cuda.device(x.device):
                if al is not None:
                    entry.sync_ref(_device_buffer(al.ref_x, x))
                entry.sync_mlp(st["linears"])
                entry.plan.forward_packed(x, out)
        return out
