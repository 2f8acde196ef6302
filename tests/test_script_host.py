"""TorchScript export (molann_amd/script.py, csrc/molann_torch.cpp) - what runs without a GPU: scripting,
saving and loading every module the reference's test file scripts (test/test_molann.py:36,46,62,75,101,114),
the plan description baked into the graph, and the loud failure on CPU tensors."""

import io

import pytest
import torch

from build_util import workload_model
from molann_amd import script, workloads as wl
from molann_amd.ann import AlignmentLayer, FeatureLayer, FeatureMap, MolANN, PreprocessingANN, create_sequential_nn
from molann_amd.atomgroup import Universe
from molann_amd.feature import Feature

U = Universe(wl.ALA_DIPEPTIDE_XYZ)


def _roundtrip(scripted):
    buf = io.BytesIO()
    torch.jit.save(scripted, buf)
    buf.seek(0)
    return torch.jit.load(buf)


def test_op_library_loads_and_registers_both_operators():
    script.load_ops()
    assert "run" in str(torch.ops.molann.run) and "run_backward" in str(torch.ops.molann.run_backward)
    schema = str(torch.ops.molann.run.default._schema)
    assert schema == "molann::run(Tensor x, int[] desc, Tensor ref_x, Tensor[] weights, Tensor[] biases) -> Tensor"


def test_make_desc_layout():
    d = script.make_desc(script.KIND_FORWARD, 22, align_idx=[1, 4], features=[(2, [0, 2, 1, 3]), (1, [4, 5])],
                         use_angle_value=True, layer_dims=[2, 5, 3], activation=1, mlp_precision=0, instance=77)
    assert d == [2, 2, 22, 2, 2, 1, 2, 1, 0, 77, 1, 4, 2, 1, 0, 4, 6, 0, 2, 1, 3, 4, 5, 2, 5, 3]
    assert script.make_desc(script.KIND_ALIGN, 5, align_idx=[0, 1, 4], instance=5) == [2, 0, 5, 3, 0, 0, 0, 0, 0, 5, 0, 1, 4]
    # two models of one architecture get descriptions (= cached plans, packed weights, workspaces) of their own
    a, b = (script.make_desc(script.KIND_ALIGN, 5, align_idx=[0, 1, 4]) for _ in range(2))
    assert a[9] != b[9] and a[:9] == b[:9] and a[10:] == b[10:]


def _reference_test_modules():
    """The modules of the reference's test file, on its 5-atom input group."""
    input_ag = U.select_atoms('bynum 1 2 3 4 5')
    f_dih = Feature('name', 'dihedral', U.select_atoms('bynum 1 3 2 4'))
    f_ang = Feature('name', 'angle', U.select_atoms('bynum 1 3 2'))
    f_bond = Feature('name', 'bond', U.select_atoms('bynum 1 3'))
    f_pos = Feature('name', 'position', U.select_atoms('bynum 1 2'))
    align = AlignmentLayer(U.select_atoms('bynum 1 2 3'), input_ag)
    fl = FeatureLayer([f_dih], input_ag, use_angle_value=False)
    pp = PreprocessingANN(None, fl)
    return {
        "feature_map": FeatureMap(f_dih, input_ag, use_angle_value=False),
        "align": AlignmentLayer(U.select_atoms('bynum 1 2 5'), U.atoms),
        "feature_layer": FeatureLayer([f_dih, f_bond, f_ang], input_ag, use_angle_value=False),
        "identity_feature_layer": FeatureLayer([Feature('identity', 'position', input_ag)], input_ag, use_angle_value=False),
        "pp_layer": PreprocessingANN(None, FeatureLayer([f_pos], input_ag, use_angle_value=False)),
        "pp_layer_aligned": PreprocessingANN(align, fl),
        "ann_layer": MolANN(pp, create_sequential_nn([pp.output_dimension(), 5, 3])),
    }


@pytest.mark.parametrize("name", ["feature_map", "align", "feature_layer", "identity_feature_layer", "pp_layer",
                                  "pp_layer_aligned", "ann_layer"])
def test_reference_test_modules_script_and_save(name, tmp_path):
    m = _reference_test_modules()[name]
    scripted = torch.jit.script(m)
    path = str(tmp_path / (name + ".pt"))
    scripted.save(path)                                   # what every reference test does last
    loaded = torch.jit.load(path)
    assert "molann::run" in str(loaded.graph)
    assert list(loaded.desc) == list(scripted.desc)
    assert loaded.desc[2] == m.input_atom_num if hasattr(m, "input_atom_num") else True
    for k, v in scripted.state_dict().items():
        assert torch.equal(v, loaded.state_dict()[k])


def test_scripted_model_shares_parameters_and_keeps_reference_buffer():
    w = wl.get_workload("C3")
    m = workload_model(w, torch.device("cpu"))
    s = torch.jit.script(m)
    lin0 = [x for x in m.ann_layers if isinstance(x, torch.nn.Linear)][0]
    assert s.linears[0].weight.data_ptr() == lin0.weight.data_ptr()
    assert torch.equal(s.ref_x, m.preprocessing_layer.align_layer.ref_x)
    assert list(s.state_dict().keys()) == ['ref_x', 'linears.0.weight', 'linears.0.bias', 'linears.1.weight', 'linears.1.bias']
    assert s.desc[:9] == [2, script.KIND_FORWARD, 22, 7, 4, 0, 2, 0, 0] and s.desc[9] != 0


def test_unrecognised_ann_layers_script_as_a_chain():
    """ann_layers that are not what create_sequential_nn builds: scripted preprocessing, then the module itself."""
    w = wl.get_workload("C2")
    pp = workload_model(w, torch.device("cpu"))
    head = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Dropout(0.0), torch.nn.Linear(4, 1))
    s = _roundtrip(torch.jit.script(MolANN(pp, head)))
    g = str(s.inlined_graph)
    assert "molann::run" in g and "aten::linear" in g


def test_cpu_tensor_raises_no_fallback():
    w = wl.get_workload("C1")
    s = _roundtrip(torch.jit.script(workload_model(w, torch.device("cpu"))))
    with pytest.raises((RuntimeError, NotImplementedError), match="CPU"):
        s(torch.zeros(4, 22, 3))
