"""TorchScript export of the molann_amd modules.

The reference's models leave Python as TorchScript: every case of its test file ends with
``torch.jit.script(module).save(name)`` (`test/test_molann.py:36,46,62,75,101,114`) and `README.rst:49` loads the
file in an MD engine.  The modules of `molann_amd.ann` launch their kernels through ctypes, which TorchScript
cannot compile, so each of them implements ``__prepare_scriptable__`` (the hook `torch.jit.script` calls
first) and hands over a :class:`ScriptPlan` instead: a module whose ``forward`` is one call of the dispatcher
operator ``molann::run`` from ``csrc/libmolann_torch.so`` (`csrc/molann_torch.cpp`), which binds the same C ABI.

    scripted = torch.jit.script(model)          # model: any molann_amd.ann module
    scripted.save('model.pt')
    ...
    import molann_amd.script; molann_amd.script.load_ops()     # or dlopen libmolann_torch.so in a C++ host
    model = torch.jit.load('model.pt').to('cuda')

The scripted module shares the parameters of the eager one, is differentiable where the eager one is
(``molann::run`` has an autograd kernel over ``molann_backward_f32``: forces from a collective variable come
out of ``torch.autograd.grad`` as in the reference) and, like it, runs on float32 HIP tensors only.
Its state_dict is flat: ``ref_x`` and ``linears.{i}.weight / bias``.
"""

import os
from typing import List

import torch

from . import _capi

TORCH_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libmolann_torch.so")

KIND_ALIGN, KIND_FEATURES, KIND_FORWARD = 0, 1, 2
_DESC_LAYOUT = 2
_loaded = False


def new_instance_id():
    """A process-unique, practically world-unique id for one model: plans (packed weights, workspaces) are cached per
    description, and two models of the same architecture must not share one."""
    import random
    return random.SystemRandom().getrandbits(62) | 1


def load_ops():
    """Register ``torch.ops.molann.*`` (idempotent).  Raises if the library has not been built."""
    global _loaded
    if not _loaded:
        _capi.lib()
        if not os.path.exists(TORCH_LIB_PATH):
            raise ImportError("%s not found: run `make -C %s libmolann_torch.so` (or __graft_entry__.build())"
                              % (TORCH_LIB_PATH, os.path.dirname(TORCH_LIB_PATH)))
        torch.ops.load_library(TORCH_LIB_PATH)
        _loaded = True


def make_desc(kind, n_inp, align_idx=None, features=None, use_angle_value=False, layer_dims=None,
              activation=_capi.ACT_TANH, mlp_precision=_capi.MLP_F32, instance=None):
    """The integer list ``molann::run`` builds its plan from (layout: `csrc/molann_torch.cpp`).
    ``features`` is ``[(type_id, [local atom indices]), ...]`` in output-column order."""
    align_idx = [int(i) for i in (align_idx or [])]
    features = list(features or [])
    layer_dims = [int(d) for d in (layer_dims or [])]
    n_layers = len(layer_dims) - 1 if layer_dims else 0
    desc = [_DESC_LAYOUT, int(kind), int(n_inp), len(align_idx), len(features), 1 if use_angle_value else 0,
            n_layers, int(activation), int(mlp_precision), int(new_instance_id() if instance is None else instance)]
    desc += align_idx
    desc += [int(t) for t, _ in features]
    if features:
        ptr, flat = [0], []
        for _, idx in features:
            flat += [int(i) for i in idx]
            ptr.append(len(flat))
        desc += ptr + flat
    if n_layers > 0:
        desc += layer_dims
    return desc


class ScriptPlan(torch.nn.Module):
    """What `torch.jit.script` compiles in place of a molann_amd module: the plan description as constants,
    the live tensors (``ref_x`` buffer, Linear layers) as module state, one operator call as ``forward``."""

    desc: List[int]

    def __init__(self, desc, ref_x=None, linears=()):
        super(ScriptPlan, self).__init__()
        load_ops()
        self.desc = [int(v) for v in desc]
        self.register_buffer("ref_x", ref_x if ref_x is not None else torch.zeros(0, 3))
        self.linears = torch.nn.ModuleList(list(linears))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        weights: List[torch.Tensor] = []
        biases: List[torch.Tensor] = []
        for lin in self.linears:
            weights.append(lin.weight)
            biases.append(lin.bias)
        return torch.ops.molann.run(x, self.desc, self.ref_x, weights, biases)
