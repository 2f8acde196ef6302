"""CPU oracle for the molann per-frame forward path.

TEST INFRASTRUCTURE ONLY.  This file is a CPU restatement, in composite PyTorch
ops, of the reference's algorithm (zwpku/molann v1.1.7, `molann/ann.py`).  It
is the checker the HIP path is compared against and the "port" CPU baseline
timed by `bench.py`; it is never the product path.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.

Parity status: PINNED.  `oracle/gen_golden.py` imports the reference itself
from /root/reference in the build container and writes its inputs/outputs to
`tests/golden/*.npz`; `tests/test_oracle_golden.py` checks every function here
against those vectors (the reference's own tests hold no numeric vectors,
SURVEY.md section 4).

Every function takes plain tensors and index lists (no AtomGroup), works for
float32 and float64 (the dtype of ``x``), and follows the reference op by op so
the fp32 rounding behaviour is the same:

* ``align_forward``        <- AlignmentLayer.forward      ann.py:157-199
* ``center_reference``     <- AlignmentLayer.__init__     ann.py:135-141
* ``feature_forward``      <- FeatureMap.forward          ann.py:288-356
* ``feature_dim``          <- FeatureMap.dim              ann.py:265-286
* ``feature_layer_forward``<- FeatureLayer.forward        ann.py:454-474
* ``mlp_forward``          <- create_sequential_nn        ann.py:37-67
* ``molann_forward``       <- MolANN.forward              ann.py:620-624

One deliberate difference: the reference calls ``torch.cross`` without ``dim``
(ann.py:342-343), which picks the FIRST size-3 dimension, so a batch of exactly
three frames produces wrong dihedrals there.  The oracle always crosses over
the xyz axis (dim=1); golden vectors avoid N == 3.
"""

import torch

ANGLE, BOND, DIHEDRAL, POSITION = 0, 1, 2, 3  # feature.py:87-97 type ids

_ACTIVATIONS = {
    "tanh": torch.tanh,
    "relu": torch.relu,
    "sigmoid": torch.sigmoid,
    "identity": lambda t: t,
}


def center_reference(ref_positions):
    """ann.py:135-141: the reference coordinates are shifted to zero mean once."""
    ref_x = torch.as_tensor(ref_positions)
    return ref_x - torch.mean(ref_x, 0)


def align_forward(x, local_align_idx, ref_x):
    """Kabsch superposition of every frame onto ``ref_x`` (ann.py:179-197).

    x: [N, n_inp, 3]; local_align_idx: positions of the align atoms inside the
    n_inp axis (ann.py:144); ref_x: [a, 3] already centred.  Returns [N, n_inp, 3].
    """
    idx = list(local_align_idx)
    sel = x[:, idx, :]                                   # :179
    x_c = torch.mean(sel, 1, True)                       # :181
    x_notran = sel - x_c                                 # :183
    xtmp = x_notran.permute((0, 2, 1))                   # :185
    prod = torch.matmul(xtmp, ref_x.to(x.dtype))         # :187
    u, s, vh = torch.linalg.svd(prod)                    # :188
    diag = torch.diag(torch.ones(3)).unsqueeze(0).repeat(x.size(0), 1, 1).to(x.device, dtype=u.dtype)  # :190
    sign_vec = torch.sign(torch.linalg.det(torch.matmul(u, vh))).detach()  # :192
    diag[:, 2, 2] = sign_vec                             # :193
    rot = torch.bmm(torch.bmm(u, diag), vh)              # :195
    return torch.matmul(x - x_c, rot)                    # :197


def feature_dim(type_id, n_atoms, use_angle_value):
    """ann.py:265-286."""
    if type_id in (ANGLE, BOND):
        return 1
    if type_id == DIHEDRAL:
        return 1 if use_angle_value else 2
    if type_id == POSITION:
        return 3 * n_atoms
    raise ValueError(type_id)


def feature_forward(x, type_id, idx, use_angle_value=False):
    """One feature of every frame (ann.py:323-354). idx are local atom positions."""
    idx = list(idx)
    if type_id == ANGLE:                                 # :323-332
        r21 = x[:, idx[0], :] - x[:, idx[1], :]
        r23 = x[:, idx[2], :] - x[:, idx[1], :]
        r21l = torch.norm(r21, dim=1, keepdim=True)
        r23l = torch.norm(r23, dim=1, keepdim=True)
        cos_angle = (r21 * r23).sum(dim=1, keepdim=True) / (r21l * r23l)
        return torch.acos(cos_angle) if use_angle_value else cos_angle
    if type_id == BOND:                                  # :334-336
        r12 = x[:, idx[1], :] - x[:, idx[0], :]
        return torch.norm(r12, dim=1, keepdim=True)
    if type_id == DIHEDRAL:                              # :338-351
        r12 = x[:, idx[1], :] - x[:, idx[0], :]
        r23 = x[:, idx[2], :] - x[:, idx[1], :]
        r34 = x[:, idx[3], :] - x[:, idx[2], :]
        n1 = torch.cross(r12, r23, dim=1)
        n2 = torch.cross(r23, r34, dim=1)
        cos_phi = (n1 * n2).sum(dim=1, keepdim=True)
        sin_phi = (n1 * r34).sum(dim=1, keepdim=True) * torch.norm(r23, dim=1, keepdim=True)
        radius = torch.sqrt(cos_phi ** 2 + sin_phi ** 2)
        if use_angle_value:
            return torch.atan2(sin_phi, cos_phi)
        return torch.cat((cos_phi / radius, sin_phi / radius), dim=1)
    if type_id == POSITION:                              # :353-354
        return x[:, idx, :].reshape((-1, len(idx) * 3))
    raise ValueError(type_id)


def feature_layer_forward(x, features, use_angle_value=False):
    """Column-concatenation of all features in list order (ann.py:473).

    features: list of (type_id, [local atom positions]).
    """
    return torch.cat([feature_forward(x, t, idx, use_angle_value) for t, idx in features], dim=1)


def mlp_forward(f, weights, biases, activation="tanh"):
    """Linear -> act -> ... -> Linear, no activation after the last (ann.py:60-65).

    weights[i]: [d_{i+1}, d_i] (torch.nn.Linear layout), biases[i]: [d_{i+1}].
    """
    act = _ACTIVATIONS[activation]
    h = f
    for i, (w, b) in enumerate(zip(weights, biases)):
        h = torch.nn.functional.linear(h, w.to(h.dtype), b.to(h.dtype))
        if i + 1 < len(weights):
            h = act(h)
    return h


def preprocessing_forward(x, features, use_angle_value=False, local_align_idx=None, ref_x=None):
    """PreprocessingANN.forward (ann.py:553-565); align is Identity when absent (:539-542)."""
    if local_align_idx is not None:
        x = align_forward(x, local_align_idx, ref_x)
    return feature_layer_forward(x, features, use_angle_value)


def molann_forward(x, features, weights, biases, use_angle_value=False,
                   local_align_idx=None, ref_x=None, activation="tanh"):
    """MolANN.forward (ann.py:620-624)."""
    f = preprocessing_forward(x, features, use_angle_value, local_align_idx, ref_x)
    return mlp_forward(f, weights, biases, activation)
