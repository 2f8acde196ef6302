"""molann's per-frame forward path (Kabsch alignment -> bond/angle/dihedral/position features -> MLP)
on the AMD Instinct MI355X: hand-written gfx950 kernels behind the `molann.ann` module API.

    from molann_amd.ann import AlignmentLayer, FeatureLayer, PreprocessingANN, MolANN, create_sequential_nn
    from molann_amd.feature import Feature, FeatureFileReader
    from molann_amd.atomgroup import Universe            # MDAnalysis-free atom groups

Importing the package does not load the HIP library; the first forward (or `molann_amd._capi.lib()`)
does, and fails loudly if `molann_amd/csrc/libmolann_hip.so` has not been built.
"""

__version__ = "0.1.0"
