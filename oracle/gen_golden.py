#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

It imports `molann.ann` / `molann.feature` from /root/reference (they need only
torch + pandas), feeds them duck-typed atom groups (MDAnalysis is not installed;
the reference only uses ``.ix``, ``.positions``, ``len``, iteration, ``+``) and
stores, per case, the inputs, the index lists, the weights and the reference's
fp32 output plus the output of its ``.double()`` copy.  Only DATA is written:
no reference source travels.  The GPU box has no /root/reference; tests there
read these files.

Batches never have exactly 3 frames: the reference's `torch.cross` without
``dim`` (ann.py:342-343) is wrong for N == 3.
"""

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

from molann.ann import (AlignmentLayer, FeatureMap, FeatureLayer, PreprocessingANN, MolANN,  # noqa: E402
                        create_sequential_nn)
from molann.feature import Feature  # noqa: E402

from molann_amd.atomgroup import Universe  # noqa: E402
from molann_amd import workloads as wl  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
TYPE_NAMES = wl.TYPE_NAMES


def read_pdb_xyz(path):
    xyz = []
    for line in open(path):
        if line.startswith("ATOM"):
            xyz.append([float(line[30:38]), float(line[38:46]), float(line[46:54])])
    return np.asarray(xyz, dtype=np.float32)


def csr(lists):
    ptr = np.zeros(len(lists) + 1, dtype=np.int64)
    flat = []
    for i, l in enumerate(lists):
        flat.extend(int(v) for v in l)
        ptr[i + 1] = len(flat)
    return np.asarray(flat, dtype=np.int64), ptr


def build_reference_model(u, input_numbers, features, align, mlp_dims, use_angle_value, seed=0):
    """Reference modules for one configuration (all numbers 1-based, order preserved)."""
    input_ag = u.atoms_by_number(input_numbers)
    feats = [Feature("f%d" % i, TYPE_NAMES[t], u.atoms_by_number(atoms)) for i, (t, atoms) in enumerate(features)]
    flayer = FeatureLayer(feats, input_ag, use_angle_value) if feats else None
    alayer = AlignmentLayer(u.atoms_by_number(align), input_ag) if align is not None else None
    nn = None
    if mlp_dims:
        torch.manual_seed(seed)
        nn = create_sequential_nn(list(mlp_dims))
    return input_ag, feats, flayer, alayer, nn


def run_case(name, u, x, input_numbers, features=(), align=None, mlp_dims=None, use_angle_value=False,
             kind="forward", weight_transform=None, extra=None, x_recipe=None, store_weights=True):
    """Run the reference on x (fp32) and on x.double(); save everything needed to replay."""
    input_ag, feats, flayer, alayer, nn = build_reference_model(u, input_numbers, list(features), align,
                                                                mlp_dims, use_angle_value)
    if nn is not None and weight_transform is not None:
        with torch.no_grad():
            for p in nn.parameters():
                p.copy_(weight_transform(p))
    if kind == "align":
        model = alayer
    elif kind == "features":
        model = PreprocessingANN(alayer, flayer)
    elif kind == "forward":
        model = MolANN(PreprocessingANN(alayer, flayer), nn)
    else:
        raise ValueError(kind)
    x = torch.as_tensor(x, dtype=torch.float32)
    with torch.no_grad():
        out32 = model(x)
        import copy
        out64 = copy.deepcopy(model).double()(x.double())
    rec = {
        "kind": np.array(kind),
        "n_inp": np.int64(len(input_ag)),
        "input_ix": np.asarray(input_ag.ix, dtype=np.int64),
        "use_angle_value": np.bool_(use_angle_value),
        "out_f32": out32.numpy(),
        "out_f64": out64.numpy(),
    }
    if x_recipe is None:
        rec["x"] = x.numpy()
    else:
        # big frames are not stored: (workload name, n_frames, seed) regenerates them with
        # molann_amd.workloads.Workload.make_frames on the CPU; the checksum guards against drift
        rec["x_recipe"] = np.array(json.dumps(x_recipe))
        rec["x_checksum"] = np.float64(x.double().sum().item())
        rec["x_first"] = x[0, :4, :].numpy()
    if alayer is not None:
        rec["align_numbers"] = np.asarray(align, dtype=np.int64)
        rec["align_local"] = np.asarray(alayer._local_align_atom_indices, dtype=np.int64)
        rec["ref_pos"] = np.asarray(u.atoms_by_number(align).positions, dtype=np.float32)
        rec["ref_x"] = alayer.ref_x.numpy()
    if flayer is not None:
        rec["feat_types"] = np.asarray([t for t, _ in features], dtype=np.int64)
        flat, ptr = csr([atoms for _, atoms in features])
        rec["feat_numbers"], rec["feat_ptr"] = flat, ptr
        lflat, _ = csr([fm._local_atom_indices for fm in flayer.feature_map_list])
        rec["feat_local"] = lflat
        rec["feat_dims"] = np.asarray([fm.dim() for fm in flayer.feature_map_list], dtype=np.int64)
        rec["feature_dim"] = np.int64(flayer.output_dimension())
    if nn is not None and kind == "forward":
        rec["mlp_dims"] = np.asarray(mlp_dims, dtype=np.int64)
        lins = [m for m in nn if isinstance(m, torch.nn.Linear)]
        if store_weights:
            for i, lin in enumerate(lins):
                rec["W%d" % i] = lin.weight.detach().numpy()
                rec["b%d" % i] = lin.bias.detach().numpy()
        rec["state_dict_keys"] = np.array(list(model.state_dict().keys()))
    if extra:
        rec.update(extra)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print("%-28s x%s -> out%s  max|f32-f64|=%.3g" % (name, tuple(x.shape), tuple(out32.shape),
                                                     float((out32.double() - out64).abs().max()) if out32.numel() else 0.0))
    return out32


def noisy(ref_xyz, n, sigma, seed, rigid=False, translation=3.0, reflect_every=0):
    g = torch.Generator().manual_seed(seed)
    ref = torch.from_numpy(np.asarray(ref_xyz, dtype=np.float32))
    x = ref.unsqueeze(0) + sigma * torch.randn((n,) + tuple(ref.shape), generator=g)
    if reflect_every:
        x[::reflect_every, :, 2] *= -1.0       # mirror image: forces det(U Vh) < 0 (ann.py:192-193)
    if rigid:
        q = torch.randn((n, 4), generator=g)
        q = q / q.norm(dim=1, keepdim=True)
        x = torch.matmul(x, wl.quaternion_to_matrix(q)) + translation * torch.randn((n, 1, 3), generator=g)
    return x


def grad_case(name, u, x, input_numbers, features, align, mlp_dims, use_angle_value=False, seed=11, extra=None):
    """Gradients of sum(out * G) from the REFERENCE's autograd (fp32 and its .double() copy)."""
    import copy
    input_ag, feats, flayer, alayer, nn = build_reference_model(u, input_numbers, list(features), align, mlp_dims, use_angle_value)
    model = MolANN(PreprocessingANN(alayer, flayer), nn) if nn is not None else PreprocessingANN(alayer, flayer)
    g = torch.Generator().manual_seed(seed)
    x = torch.as_tensor(x, dtype=torch.float32)
    rec = {}
    G = None
    for tag, m, xx in (("f32", model, x.clone()), ("f64", copy.deepcopy(model).double(), x.double())):
        xx.requires_grad_(True)
        out = m(xx)
        if G is None:
            G = torch.randn(out.shape, generator=g)
        (out * G.to(out.dtype)).sum().backward()
        rec["gx_" + tag] = xx.grad.numpy()
        for i, prm in enumerate(m.parameters()):
            rec["gp%d_%s" % (i, tag)] = prm.grad.numpy()
        rec["out_" + tag] = out.detach().numpy()
    rec.update(x=x.numpy(), G=G.numpy(), n_inp=np.int64(len(input_ag)), use_angle_value=np.bool_(use_angle_value),
               feat_types=np.asarray([t for t, _ in features], dtype=np.int64))
    flat, ptr = csr([a for _, a in features])
    rec["feat_numbers"], rec["feat_ptr"] = flat, ptr
    if align is not None:
        rec["align_numbers"] = np.asarray(align, dtype=np.int64)
    if nn is not None:
        rec["mlp_dims"] = np.asarray(mlp_dims, dtype=np.int64)
        for i, lin in enumerate([m for m in nn if isinstance(m, torch.nn.Linear)]):
            rec["W%d" % i] = lin.weight.detach().numpy()
            rec["b%d" % i] = lin.bias.detach().numpy()
    if extra:
        rec.update(extra)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print("%-24s gx %s  max|gx32-gx64| = %.3g" % (name, rec["gx_f32"].shape, np.abs(rec["gx_f32"] - rec["gx_f64"]).max()))


def grad_main():
    """Gradient golden vectors only (files grad_*.npz); the forward cases are left untouched."""
    os.makedirs(OUT, exist_ok=True)
    pdb = read_pdb_xyz("/root/reference/test/alanine-dipeptide-vacuum.pdb")
    u = Universe(pdb)
    all22 = list(range(1, 23))
    for cname in ("C1", "C3"):
        w = wl.get_workload(cname)
        grad_case("grad_molann_%s" % cname, u, w.make_frames(128, seed=21), all22, w.features, w.align, w.mlp_dims)
    w = wl.get_workload("C3")
    grad_case("grad_features_C3_val", u, w.make_frames(128, seed=22), all22, w.features, w.align, None, use_angle_value=True)
    w = wl.get_workload("C3p")
    grad_case("grad_features_C3p", u, w.make_frames(128, seed=23), all22, w.features, w.align, None)
    w = wl.get_workload("C2")
    grad_case("grad_features_C2", u, w.make_frames(128, seed=24), all22, w.features, None, None)


def main():
    os.makedirs(OUT, exist_ok=True)
    pdb = read_pdb_xyz("/root/reference/test/alanine-dipeptide-vacuum.pdb")
    assert pdb.shape == (22, 3)
    assert np.array_equal(pdb, wl.ALA_DIPEPTIDE_XYZ), "workloads.ALA_DIPEPTIDE_XYZ drifted from the PDB"
    u = Universe(pdb)
    all22 = list(range(1, 23))
    np.savez_compressed(os.path.join(OUT, "ala_dipeptide_pdb.npz"), xyz=pdb)

    x256 = noisy(pdb, 256, 0.1, 1234)
    x_pdb1 = torch.from_numpy(pdb).unsqueeze(0)

    # ---- FeatureMap: 4 types x 2 use_angle_value, index lists of test/feature.txt + test_molann.py
    fm_cases = {
        "angle_20_19_21": (wl.ANGLE, (20, 19, 21)), "angle_16_15_17": (wl.ANGLE, (16, 15, 17)),
        "angle_1_3_2": (wl.ANGLE, (1, 3, 2)),
        "bond_2_5": (wl.BOND, (2, 5)), "bond_5_6": (wl.BOND, (5, 6)), "bond_1_3": (wl.BOND, (1, 3)),
        "dihedral_5_7_9_15": (wl.DIHEDRAL, (5, 7, 9, 15)), "dihedral_7_9_15_17": (wl.DIHEDRAL, (7, 9, 15, 17)),
        "dihedral_1_3_2_4": (wl.DIHEDRAL, (1, 3, 2, 4)), "dihedral_1_2_3_4": (wl.DIHEDRAL, (1, 2, 3, 4)),
        "position_resid2": (wl.POSITION, tuple(range(7, 17))), "position_1_3_2": (wl.POSITION, (1, 3, 2)),
        "position_all": (wl.POSITION, tuple(all22)),
    }
    for nm, feat in fm_cases.items():
        for uav in (False, True):
            run_case("fmap_%s_%s" % (nm, "val" if uav else "cs"), u, x256, all22, [feat],
                     use_angle_value=uav, kind="features")
    # the single PDB frame (batch 1), anchors quoted in SURVEY.md section 4
    run_case("fmap_pdbframe_hist", u, x_pdb1, all22,
             [(wl.DIHEDRAL, (5, 7, 9, 15)), (wl.DIHEDRAL, (7, 9, 15, 17)), (wl.BOND, (2, 5)), (wl.BOND, (5, 6)),
              (wl.ANGLE, (20, 19, 21)), (wl.ANGLE, (16, 15, 17)), (wl.DIHEDRAL, (1, 3, 2, 4)),
              (wl.DIHEDRAL, (1, 2, 3, 4))], kind="features")

    # ---- FeatureLayer as in test_FeatureLayer (input group = atoms 1..5, mixed list, identity layer)
    x5 = x256[:, :5, :].contiguous()
    run_case("flayer_test_mixed", u, x5, [1, 2, 3, 4, 5],
             [(wl.DIHEDRAL, (1, 3, 2, 4)), (wl.BOND, (1, 3)), (wl.ANGLE, (1, 3, 2))], kind="features")
    run_case("flayer_test_mixed_val", u, x5, [1, 2, 3, 4, 5],
             [(wl.DIHEDRAL, (1, 3, 2, 4)), (wl.BOND, (1, 3)), (wl.ANGLE, (1, 3, 2))], use_angle_value=True,
             kind="features")
    run_case("flayer_identity5", u, x5, [1, 2, 3, 4, 5], [(wl.POSITION, (1, 2, 3, 4, 5))], kind="features")
    # input group in a permuted order: local index != global index (ann.py:261)
    perm = [9, 2, 17, 5, 7, 15, 6, 16, 19]
    xperm = x256[:, [p - 1 for p in perm], :].contiguous()
    run_case("flayer_permuted_input", u, xperm, perm,
             [(wl.DIHEDRAL, (5, 7, 9, 15)), (wl.BOND, (5, 6)), (wl.ANGLE, (16, 15, 17)), (wl.POSITION, (19, 2))],
             kind="features")

    # ---- AlignmentLayer
    bb = list(wl.ALA_BACKBONE)
    run_case("align_125_centred", u, x256, all22, align=[1, 2, 5], kind="align")          # test_AlignmentLayer
    run_case("align_backbone_centred", u, x256, all22, align=bb, kind="align")
    run_case("align_backbone_rigid", u, noisy(pdb, 256, 0.1, 77, rigid=True), all22, align=bb, kind="align")
    run_case("align_backbone_far", u, noisy(pdb, 256, 0.1, 78, rigid=True, translation=30.0), all22, align=bb,
             kind="align")
    run_case("align_125_rigid", u, noisy(pdb, 256, 0.2, 79, rigid=True), all22, align=[1, 2, 5], kind="align")
    run_case("align_sidechain_reflect", u, noisy(pdb, 256, 0.1, 80, rigid=True, reflect_every=2), all22,
             align=[9, 10, 11, 15, 7], kind="align")                                      # non-planar set, mirrored frames
    run_case("align_all22_rigid", u, noisy(pdb, 128, 0.3, 81, rigid=True), all22, align=all22, kind="align")
    run_case("align_pdbframe", u, x_pdb1, all22, align=[1, 2, 5], kind="align")
    run_case("align_subset_input", u, x256[:, [0, 1, 2, 4], :].contiguous(), [1, 2, 3, 5], align=[1, 2, 5],
             kind="align")
    for n in (1, 2, 4, 63, 64, 65):
        run_case("align_backbone_n%d" % n, u, noisy(pdb, n, 0.1, 100 + n, rigid=True), all22, align=bb, kind="align")

    # ---- PreprocessingANN as in test_PreprocessingANN
    run_case("pp_align123_dihedral", u, x5, [1, 2, 3, 4, 5], [(wl.DIHEDRAL, (1, 3, 2, 4))], align=[1, 2, 3],
             kind="features")
    run_case("pp_align123_pos12", u, noisy(pdb[:5], 256, 0.1, 5, rigid=True), [1, 2, 3, 4, 5], [(wl.POSITION, (1, 2))],
             align=[1, 2, 3], kind="features")
    run_case("pp_noalign_pos12", u, x5, [1, 2, 3, 4, 5], [(wl.POSITION, (1, 2))], kind="features")

    # ---- MolANN: test_MolANN shape, then the BASELINE configs at parity size
    run_case("molann_test", u, x5, [1, 2, 3, 4, 5], [(wl.DIHEDRAL, (1, 3, 2, 4))], mlp_dims=[2, 5, 3])
    for cname in ("C1", "C1s", "C3"):
        w = wl.get_workload(cname)
        run_case("molann_%s" % cname, u, w.make_frames(1024 if cname != "C3" else 512, seed=w.seed), all22,
                 w.features, align=w.align, mlp_dims=w.mlp_dims)
    w = wl.get_workload("C2")
    run_case("features_C2", u, w.make_frames(1024), all22, w.features, kind="features")
    w = wl.get_workload("C3p")
    run_case("features_C3p", u, w.make_frames(256), all22, w.features, align=w.align, kind="features")
    w = wl.get_workload("C3")
    run_case("features_C3_val", u, w.make_frames(256), all22, w.features, align=w.align, use_angle_value=True,
             kind="features")
    for act_name, act in (("relu", torch.nn.ReLU()), ("sigmoid", torch.nn.Sigmoid())):
        input_ag, feats, flayer, alayer, _ = build_reference_model(u, all22, w.features, w.align, None, False)
        torch.manual_seed(3)
        nn = create_sequential_nn([6, 16, 16, 4], activation=act)
        model = MolANN(PreprocessingANN(alayer, flayer), nn)
        xa = w.make_frames(256, seed=9)
        import copy
        with torch.no_grad():
            o32 = model(xa)
            o64 = copy.deepcopy(model).double()(xa.double())
        lins = [m for m in nn if isinstance(m, torch.nn.Linear)]
        flat, ptr = csr([a for _, a in w.features])
        rec = dict(kind=np.array("forward"), x=xa.numpy(), n_inp=np.int64(22), input_ix=np.arange(22),
                   use_angle_value=np.bool_(False), out_f32=o32.numpy(), out_f64=o64.numpy(),
                   align_numbers=np.asarray(w.align), align_local=np.asarray(alayer._local_align_atom_indices),
                   ref_pos=np.asarray(u.atoms_by_number(w.align).positions), ref_x=alayer.ref_x.numpy(),
                   feat_types=np.asarray([t for t, _ in w.features]), feat_numbers=flat, feat_ptr=ptr,
                   feat_local=flat - 1, feat_dims=np.asarray([fm.dim() for fm in flayer.feature_map_list]),
                   feature_dim=np.int64(6), mlp_dims=np.asarray([6, 16, 16, 4]), activation=np.array(act_name))
        for i, lin in enumerate(lins):
            rec["W%d" % i] = lin.weight.detach().numpy()
            rec["b%d" % i] = lin.bias.detach().numpy()
        np.savez_compressed(os.path.join(OUT, "molann_C3_%s.npz" % act_name), **rec)
        print("molann_C3_%s" % act_name, tuple(o32.shape))

    # ---- 5000-atom configs at parity size (x is regenerated from the seed: 16 frames = 960 KB each)
    for cname, nfr in (("C4", 16), ("C5", 16)):
        w = wl.get_workload(cname)
        uc = Universe(w.ref_xyz)
        xa = w.make_frames(nfr, seed=w.seed)
        allc = list(range(1, w.n_atoms + 1))
        recipe = {"workload": cname, "frames": nfr, "seed": w.seed}
        run_case("molann_%s_small" % cname, uc, xa, allc, w.features, align=w.align, mlp_dims=w.mlp_dims,
                 x_recipe=recipe)
        if cname == "C5":
            # the reference run with bf16-ROUNDED weights (what the bf16 MFMA path holds), fp32 math
            # (weights = those of molann_C5_small rounded to bf16; not stored twice)
            run_case("molann_C5_small_bf16w", uc, xa, allc, w.features, align=w.align, mlp_dims=w.mlp_dims,
                     weight_transform=lambda p: p.to(torch.bfloat16).to(torch.float32), x_recipe=recipe,
                     store_weights=False)

    # ---- error paths: record the exception TYPE the reference raises
    errs = {}

    def rec_err(key, fn):
        try:
            fn()
            errs[key] = "none"
        except BaseException as e:  # noqa: BLE001
            errs[key] = type(e).__name__
    input_ag = u.atoms_by_number(all22)
    al = AlignmentLayer(u.atoms_by_number([1, 2, 5]), input_ag)
    fl = FeatureLayer([Feature("b", "bond", u.atoms_by_number([5, 6]))], input_ag)
    fmap = FeatureMap(Feature("b", "bond", u.atoms_by_number([5, 6])), input_ag)
    x22 = torch.from_numpy(pdb)
    rec_err("align_not_tensor", lambda: al(pdb))
    rec_err("align_2d_input", lambda: al(x22))
    rec_err("align_wrong_natoms", lambda: al(x22[:21].unsqueeze(0)))
    rec_err("align_wrong_last", lambda: al(torch.zeros(4, 22, 2)))
    rec_err("flayer_not_tensor", lambda: fl(pdb))
    rec_err("flayer_2d_input", lambda: fl(x22))
    rec_err("flayer_wrong_natoms", lambda: fl(x22[:21].unsqueeze(0)))
    rec_err("fmap_wrong_natoms", lambda: fmap(x22[:21].unsqueeze(0)))
    rec_err("align_atom_not_in_input", lambda: AlignmentLayer(u.atoms_by_number([1, 2, 5]), u.atoms_by_number([1, 2, 3])))
    rec_err("feature_atom_not_in_input", lambda: FeatureMap(Feature("b", "bond", u.atoms_by_number([5, 6])),
                                                              u.atoms_by_number([1, 2, 3, 4, 5])))
    rec_err("flayer_empty_list", lambda: FeatureLayer([], input_ag))
    rec_err("nn_one_dim", lambda: create_sequential_nn([10]))
    rec_err("feature_unknown_type", lambda: Feature("q", "torsion", u.atoms_by_number([1, 2])))
    rec_err("feature_repeated_atoms", lambda: Feature("q", "bond", u.atoms_by_number([1, 1])))
    rec_err("feature_bond_3atoms", lambda: Feature("q", "bond", u.atoms_by_number([1, 2, 3])))
    rec_err("feature_angle_2atoms", lambda: Feature("q", "angle", u.atoms_by_number([1, 2])))
    rec_err("feature_dihedral_3atoms", lambda: Feature("q", "dihedral", u.atoms_by_number([1, 2, 3])))
    rec_err("empty_batch_align", lambda: al(torch.zeros(0, 22, 3)))
    rec_err("empty_batch_flayer", lambda: fl(torch.zeros(0, 22, 3)))
    with torch.no_grad():
        e_al = al(torch.zeros(0, 22, 3))
        e_fl = fl(torch.zeros(0, 22, 3))
    meta = {
        "errors": errs,
        "empty_batch_shapes": {"align": list(e_al.shape), "flayer": list(e_fl.shape)},
        "molann_state_dict_keys": list(MolANN(PreprocessingANN(al, fl), create_sequential_nn([1, 4, 2])).state_dict().keys()),
        "sequential_module_names": list(create_sequential_nn([3, 5, 4, 2])._modules.keys()),
        "torch_version": torch.__version__,
        "reference_version": "molann 1.1.7 (setup.cfg)",
    }
    with open(os.path.join(OUT, "reference_meta.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)
    print(json.dumps(meta, indent=1))


def grad2_case(name, u, x, input_numbers, features, align, mlp_dims, use_angle_value=False, seed=13):
    """Second-order golden vectors from the REFERENCE's autograd: E = sum(out * G), forces F = dE/dx with create_graph=True,
    L = sum(F * F); stored: dL/dx and dL/d(parameters) (float64 model; the float32 model's dL/dx as well)."""
    import copy
    input_ag, feats, flayer, alayer, nn = build_reference_model(u, input_numbers, list(features), align, mlp_dims, use_angle_value)
    model = MolANN(PreprocessingANN(alayer, flayer), nn) if nn is not None else PreprocessingANN(alayer, flayer)
    g = torch.Generator().manual_seed(seed)
    x = torch.as_tensor(x, dtype=torch.float32)
    rec = {}
    G = None
    for tag, m, xx in (("f32", model, x.clone()), ("f64", copy.deepcopy(model).double(), x.double())):
        xx.requires_grad_(True)
        out = m(xx)
        if G is None:
            G = torch.randn(out.shape, generator=g)
        (F,) = torch.autograd.grad((out * G.to(out.dtype)).sum(), xx, create_graph=True)
        L = (F * F).sum()
        L.backward()
        rec["F_" + tag] = F.detach().numpy()
        rec["gx2_" + tag] = xx.grad.numpy()
        for i, prm in enumerate(m.parameters()):
            rec["gp2_%d_%s" % (i, tag)] = (prm.grad if prm.grad is not None else torch.zeros_like(prm)).numpy()   # (the forces do not depend on the last bias)
    rec.update(x=x.numpy(), G=G.numpy(), n_inp=np.int64(len(input_ag)), use_angle_value=np.bool_(use_angle_value),
               feat_types=np.asarray([t for t, _ in features], dtype=np.int64))
    flat, ptr = csr([a for _, a in features])
    rec["feat_numbers"], rec["feat_ptr"] = flat, ptr
    if align is not None:
        rec["align_numbers"] = np.asarray(align, dtype=np.int64)
    if nn is not None:
        rec["mlp_dims"] = np.asarray(mlp_dims, dtype=np.int64)
        for i, lin in enumerate([m_ for m_ in nn if isinstance(m_, torch.nn.Linear)]):
            rec["W%d" % i] = lin.weight.detach().numpy()
            rec["b%d" % i] = lin.bias.detach().numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print("%-24s gx2 %s  max|gx2_32-gx2_64| = %.3g of %.3g" % (name, rec["gx2_f64"].shape, np.abs(rec["gx2_f32"] - rec["gx2_f64"]).max(), np.abs(rec["gx2_f64"]).max()))


def round3_main():
    """Round 3 additions only (the other files are left untouched): AlignmentLayer.forward on large frames - the 5000-atom
    chain of workload A4 (x regenerated from the seed) and a 301-atom chain whose frame is not a multiple of 16 bytes - and
    a MolANN with hidden widths 33..64 on the 22-atom system."""
    os.makedirs(OUT, exist_ok=True)
    w = wl.get_workload("A4")
    uc = Universe(w.ref_xyz)
    allc = list(range(1, w.n_atoms + 1))
    # (x is stored: the rigid motion of A4's frames is a batched matmul whose rounding differs between hosts)
    run_case("align_chain5000", uc, w.make_frames(6, seed=w.seed), allc, align=list(w.align), kind="align")
    xyz301 = w.ref_xyz[:301] - w.ref_xyz[:301].mean(axis=0, keepdims=True)
    u301 = Universe(xyz301)
    run_case("align_chain301", u301, noisy(xyz301, 70, 0.1, 301, rigid=True), list(range(1, 302)), align=list(range(3, 302, 7)),
             kind="align", extra={"ref_xyz": xyz301})
    pdb = read_pdb_xyz("/root/reference/test/alanine-dipeptide-vacuum.pdb")
    u = Universe(pdb)
    w3 = wl.get_workload("C3")
    run_case("molann_C3_wide64", u, w3.make_frames(512, seed=64), list(range(1, 23)), w3.features, align=w3.align, mlp_dims=[6, 64, 64, 8])
    run_case("molann_C3_wide48", u, w3.make_frames(300, seed=48), list(range(1, 23)), w3.features, align=w3.align, mlp_dims=[6, 48, 33, 5])
    # second-order gradients (create_graph=True): the reference's autograd through its SVD (files grad2_*.npz)
    all22 = list(range(1, 23))
    grad2_case("grad2_molann_C3", u, w3.make_frames(48, seed=31), all22, w3.features, w3.align, w3.mlp_dims)
    wp = wl.get_workload("C3p")
    grad2_case("grad2_features_C3p", u, wp.make_frames(48, seed=32), all22, wp.features, wp.align, None)
    w2 = wl.get_workload("C2")
    grad2_case("grad2_features_C2", u, w2.make_frames(48, seed=33), all22, w2.features, None, None)


def round3b_main():
    """Round 3, second batch (the other files are left untouched): frames between the two BASELINE systems - workload P1 (166-atom
    chain, Kabsch on 42 atoms, 8 dihedrals, MLP [16, 32, 8]; 99 frames: a short last ring entry), its AlignmentLayer alone (A5) and a
    300-atom chain aligned on 200 atoms (two frames per ring entry)."""
    os.makedirs(OUT, exist_ok=True)
    w = wl.get_workload("P1")
    u = Universe(w.ref_xyz)
    alln = list(range(1, w.n_atoms + 1))
    run_case("molann_P1", u, w.make_frames(99, seed=5), alln, w.features, align=list(w.align), mlp_dims=w.mlp_dims,
             extra={"ref_xyz": w.ref_xyz})
    run_case("align_P1", u, w.make_frames(41, seed=6), alln, align=list(w.align), kind="align", extra={"ref_xyz": w.ref_xyz})
    xyz = wl.synthetic_chain(n_atoms=300, step=1.4, seed=13)
    u3 = Universe(xyz)
    rng = np.random.default_rng(8)
    align = sorted((rng.choice(300, size=200, replace=False) + 1).tolist())
    feats = wl.chain_features(300, 30, 9)
    run_case("molann_chain300", u3, noisy(xyz, 51, 0.1, 300, rigid=True), list(range(1, 301)), feats, align=align,
             mlp_dims=[sum(1 if t in (wl.BOND, wl.ANGLE) else 2 for t, _ in feats), 32, 8], extra={"ref_xyz": xyz})


def round3c_main():
    """Round 3, third batch: gradients (the reference's autograd, fp32 and fp64) of the mid-size model P1 - the training path of a
    small head behind the wave-per-frame kernels - and of its features alone."""
    os.makedirs(OUT, exist_ok=True)
    w = wl.get_workload("P1")
    u = Universe(w.ref_xyz)
    alln = list(range(1, w.n_atoms + 1))
    grad_case("grad_molann_P1", u, w.make_frames(70, seed=21), alln, w.features, list(w.align), w.mlp_dims, extra={"ref_xyz": w.ref_xyz})
    grad_case("grad_features_P1", u, w.make_frames(33, seed=22), alln, w.features, list(w.align), None, extra={"ref_xyz": w.ref_xyz})


if __name__ == "__main__":
    if "--round3c" in sys.argv:
        round3c_main()
        sys.exit(0)
    if "--round3b" in sys.argv:
        round3b_main()
        sys.exit(0)
    if "--round3" in sys.argv:
        round3_main()
        sys.exit(0)
    if "--grads" in sys.argv:
        grad_main()
    else:
        main()
