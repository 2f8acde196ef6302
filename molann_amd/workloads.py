"""The benchmark / parity configurations C1..C5 of BASELINE.json and SURVEY.md section 8(d).

Each workload names the system (atom count, reference coordinates), the index
lists (1-based atom numbers, as in the reference's `test/feature.txt`), the MLP
and the synthetic-frame distribution.  `bench.py`, the tests and
`oracle/gen_golden.py` all build their inputs from here so that they agree.
"""


import numpy as np
import torch

ANGLE, BOND, DIHEDRAL, POSITION = 0, 1, 2, 3
TYPE_NAMES = {ANGLE: "angle", BOND: "bond", DIHEDRAL: "dihedral", POSITION: "position"}

# Coordinates (Angstrom) of the 22 atoms of ACE-ALA-NME, the reference's test
# system (`test/alanine-dipeptide-vacuum.pdb:2-23`), stored as numbers.
ALA_DIPEPTIDE_XYZ = np.array([
    [2.000, 1.000, -0.000], [2.000, 2.090, 0.000], [1.486, 2.454, 0.890], [1.486, 2.454, -0.890],
    [3.427, 2.641, -0.000], [4.391, 1.877, -0.000], [3.555, 3.970, -0.000], [2.733, 4.556, -0.000],
    [4.853, 4.614, -0.000], [5.408, 4.316, 0.890], [5.661, 4.221, -1.232], [5.123, 4.521, -2.131],
    [6.630, 4.719, -1.206], [5.809, 3.141, -1.241], [4.713, 6.129, 0.000], [3.601, 6.653, 0.000],
    [5.846, 6.835, 0.000], [6.737, 6.359, -0.000], [5.846, 8.284, 0.000], [4.819, 8.648, 0.000],
    [6.360, 8.648, 0.890], [6.360, 8.648, -0.890]], dtype=np.float32)

ALA_BACKBONE = (2, 5, 7, 9, 15, 17, 19)   # heavy backbone atoms, 1-based


class Workload(object):
    """One configuration: system + index lists (1-based) + MLP + frame generator."""

    def __init__(self, name, ref_xyz, features, align=None, mlp_dims=None, use_angle_value=False,
                 frames=1 << 20, noise=0.1, rigid_motion=False, translation=3.0, seed=1234,
                 mlp_dtype="f32", description="", kind="forward"):
        self.name = name
        self.ref_xyz = np.ascontiguousarray(ref_xyz, dtype=np.float32)
        self.n_atoms = int(self.ref_xyz.shape[0])
        self.features = [(int(t), tuple(int(a) for a in atoms)) for t, atoms in features]
        self.align = tuple(int(a) for a in align) if align is not None else None
        self.mlp_dims = list(mlp_dims) if mlp_dims is not None else None
        self.use_angle_value = bool(use_angle_value)
        self.frames = int(frames)
        self.noise = float(noise)
        self.rigid_motion = bool(rigid_motion)
        self.translation = float(translation)
        self.seed = int(seed)
        self.mlp_dtype = mlp_dtype
        self.description = description
        self.kind = kind          # "forward" (features [+ MLP]) or "align" (AlignmentLayer.forward alone)

    # ---- derived sizes -------------------------------------------------------------------
    def feature_dim(self):
        d = 0
        for t, atoms in self.features:
            if t in (ANGLE, BOND):
                d += 1
            elif t == DIHEDRAL:
                d += 1 if self.use_angle_value else 2
            else:
                d += 3 * len(atoms)
        return d

    def out_dim(self):
        if self.kind == "align":
            return 3 * self.n_atoms
        return self.mlp_dims[-1] if self.mlp_dims else self.feature_dim()

    def touched_atoms(self):
        s = set(self.align or ())
        for _, atoms in self.features:
            s.update(atoms)
        return sorted(s)

    def algorithmic_bytes_per_frame(self):
        """12 * |align U feature atoms| + sizeof(out) * d_out (align: every atom in and out)."""
        if self.kind == "align":
            return 24 * self.n_atoms
        return 12 * len(self.touched_atoms()) + 4 * self.out_dim()

    def dense_bytes_per_frame(self):
        return 12 * self.n_atoms + 4 * self.out_dim()

    # ---- synthetic frames ----------------------------------------------------------------
    def make_frames(self, n_frames=None, device="cpu", seed=None, chunk=1 << 16):
        """``[n, n_atoms, 3]`` fp32 frames: reference + noise (+ random rigid motion).

        Same distribution on every device; the random STREAM differs between the CPU
        and the GPU generator, so parity batches are made on the CPU and copied.
        """
        n = self.frames if n_frames is None else int(n_frames)
        seed = self.seed if seed is None else int(seed)
        dev = torch.device(device)
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        ref = torch.from_numpy(self.ref_xyz).to(dev)
        out = torch.empty((n, self.n_atoms, 3), dtype=torch.float32, device=dev)
        for s in range(0, n, chunk):
            m = min(chunk, n - s)
            x = ref.unsqueeze(0) + self.noise * torch.randn((m, self.n_atoms, 3), generator=gen, device=dev)
            if self.rigid_motion:
                q = torch.randn((m, 4), generator=gen, device=dev)
                q = q / q.norm(dim=1, keepdim=True)
                rot = quaternion_to_matrix(q)
                t = self.translation * torch.randn((m, 1, 3), generator=gen, device=dev)
                x = torch.matmul(x, rot) + t
            out[s:s + m] = x
        return out

    def make_mlp(self, seed=0):
        """Weights of `create_sequential_nn(mlp_dims)` under torch's default Linear init."""
        if not self.mlp_dims:
            return None, None
        torch.manual_seed(seed)
        ws, bs = [], []
        for i in range(len(self.mlp_dims) - 1):
            lin = torch.nn.Linear(self.mlp_dims[i], self.mlp_dims[i + 1])
            ws.append(lin.weight.detach().clone())
            bs.append(lin.bias.detach().clone())
        return ws, bs


def quaternion_to_matrix(q):
    """Unit quaternions [m,4] (w,x,y,z) -> proper rotation matrices [m,3,3]."""
    w, x, y, z = q.unbind(dim=1)
    rot = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1)
    return rot.reshape(-1, 3, 3)


def synthetic_chain(n_atoms=5000, step=1.5, seed=7):
    """Random-walk chain used by C4/C5: numpy default_rng(7), step 1.5 A."""
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((n_atoms, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    xyz = np.cumsum(step * d, axis=0)
    xyz -= xyz.mean(axis=0, keepdims=True)
    return xyz.astype(np.float32)


def chain_features(n_atoms, n_features, seed):
    """bond/angle/dihedral cycling on consecutive-index windows at seeded random offsets."""
    rng = np.random.default_rng(seed)
    feats = []
    kinds = [(BOND, 2), (ANGLE, 3), (DIHEDRAL, 4)]
    starts = rng.choice(n_atoms - 4, size=n_features, replace=False)
    for i in range(n_features):
        t, k = kinds[i % 3]
        s = int(starts[i]) + 1
        feats.append((t, tuple(range(s, s + k))))
    return feats


def _c1():
    return Workload("C1", ALA_DIPEPTIDE_XYZ, [(BOND, (5, 6)), (DIHEDRAL, (1, 3, 2, 4))],
                    mlp_dims=[3, 5, 3], frames=1024,
                    description="22 atoms, bond 5-6 + dihedral 1-3-2-4 (literal order), MLP [3,5,3]")


def _c1_sorted():
    return Workload("C1s", ALA_DIPEPTIDE_XYZ, [(BOND, (5, 6)), (DIHEDRAL, (1, 2, 3, 4))],
                    mlp_dims=[3, 5, 3], frames=1024,
                    description="C1 with the dihedral atoms as select_atoms returns them (sorted)")


def _c2():
    return Workload("C2", ALA_DIPEPTIDE_XYZ, [(BOND, (5, 6)), (DIHEDRAL, (1, 3, 2, 4))],
                    frames=1 << 20, description="22 atoms, FeatureLayer only (2 features, d=3)")


def _c3():
    feats = [(DIHEDRAL, (5, 7, 9, 15)), (DIHEDRAL, (7, 9, 15, 17)), (BOND, (5, 6)), (ANGLE, (16, 15, 17))]
    return Workload("C3", ALA_DIPEPTIDE_XYZ, feats, align=ALA_BACKBONE, mlp_dims=[6, 32, 8],
                    frames=1 << 20, rigid_motion=True,
                    description="22 atoms, Kabsch on 7 backbone atoms + 4 features (d=6) + MLP [6,32,8]")


def _c3p():
    return Workload("C3p", ALA_DIPEPTIDE_XYZ, [(POSITION, tuple(range(1, 23)))], align=ALA_BACKBONE,
                    frames=1 << 20, rigid_motion=True,
                    description="22 atoms, Kabsch + identity (position) feature over all atoms (d=66)")


def _c4():
    xyz = synthetic_chain()
    align = tuple(range(9, 5001, 16))
    return Workload("C4", xyz, chain_features(5000, 64, 41), align=align, mlp_dims=[85, 128, 64, 8],
                    frames=1 << 19,
                    description="5000-atom chain, Kabsch on 312 'CA' + 64 features (d=85) + MLP [85,128,64,8]")


def _c5():
    xyz = synthetic_chain()
    align = tuple(range(9, 5001, 16))
    return Workload("C5", xyz, chain_features(5000, 256, 42), align=align, mlp_dims=[341, 512, 256, 16],
                    frames=1 << 20, mlp_dtype="bf16",
                    description="5000-atom chain, Kabsch on 312 'CA' + 256 features (d=341) + bf16 MLP [341,512,256,16]")


def _a3():
    return Workload("A3", ALA_DIPEPTIDE_XYZ, [], align=ALA_BACKBONE, frames=1 << 20, rigid_motion=True, kind="align",
                    description="22 atoms, AlignmentLayer.forward alone (Kabsch on 7 backbone atoms, all 22 atoms written back)")


def _a4():
    xyz = synthetic_chain()
    return Workload("A4", xyz, [], align=tuple(range(9, 5001, 16)), frames=1 << 18, rigid_motion=True, kind="align",
                    description="5000-atom chain, AlignmentLayer.forward alone (Kabsch on 312 'CA', all 5000 atoms written back)")


def _peptide_xyz():
    return synthetic_chain(n_atoms=166, step=1.4, seed=11)


def _p1():
    # the size class between the two BASELINE systems: a 166-atom peptide (chignolin's size), Kabsch on every fourth atom
    feats = [(DIHEDRAL, tuple(range(s, s + 4))) for s in range(5, 160, 20)]
    return Workload("P1", _peptide_xyz(), feats, align=tuple(range(2, 167, 4)), mlp_dims=[16, 32, 8], frames=1 << 20,
                    rigid_motion=True, description="166-atom chain, Kabsch on 42 atoms + 8 dihedrals (d=16) + MLP [16,32,8]")


def _p2():
    # the reference's own use of the alignment: aligned POSITIONS of a subset of atoms as the encoder's input (README.rst: feature type 'position')
    sel = tuple(range(2, 167, 4))
    return Workload("P2", _peptide_xyz(), [(POSITION, sel)], align=sel, mlp_dims=[126, 64, 32, 2], frames=1 << 20, rigid_motion=True,
                    description="166-atom chain, Kabsch on 42 atoms + their aligned positions (d=126) + MLP [126,64,32,2]")


def _a5():
    return Workload("A5", _peptide_xyz(), [], align=tuple(range(2, 167, 4)), frames=1 << 20, rigid_motion=True, kind="align",
                    description="166-atom chain, AlignmentLayer.forward alone (Kabsch on 42 atoms, all 166 atoms written back)")


_FACTORIES = {"A3": _a3, "A4": _a4, "A5": _a5, "P1": _p1, "P2": _p2, "C1": _c1, "C1s": _c1_sorted, "C2": _c2, "C3": _c3, "C3p": _c3p, "C4": _c4, "C5": _c5}


def get_workload(name):
    if name not in _FACTORIES:
        raise KeyError("unknown workload %r (have %s)" % (name, ", ".join(sorted(_FACTORIES))))
    return _FACTORIES[name]()


def workload_names():
    return list(_FACTORIES)


def build_model(w, device=None, seed=0):
    """The product model (molann_amd.ann modules) of a workload; input group = all atoms."""
    from .ann import AlignmentLayer, FeatureLayer, MolANN, PreprocessingANN, create_sequential_nn
    from .atomgroup import Universe
    from .feature import Feature
    u = Universe(w.ref_xyz)
    input_ag = u.atoms
    alayer = AlignmentLayer(u.atoms_by_number(w.align), input_ag) if w.align is not None else None
    if w.kind == "align":
        return alayer.to(device) if device is not None else alayer
    feats = [Feature("f%d" % i, TYPE_NAMES[t], u.atoms_by_number(atoms)) for i, (t, atoms) in enumerate(w.features)]
    pp = PreprocessingANN(alayer, FeatureLayer(feats, input_ag, w.use_angle_value))
    if not w.mlp_dims:
        model = pp
    else:
        torch.manual_seed(seed)
        model = MolANN(pp, create_sequential_nn(w.mlp_dims), mlp_precision=("bf16" if w.mlp_dtype == "bf16" else "f32"))
    return model.to(device) if device is not None else model
