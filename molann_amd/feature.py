"""Feature descriptors (setup time): same classes and semantics as `molann/feature.py`.

`Feature` validates a (name, type, atom group) triple exactly as the reference does
(`feature.py:79-102`): unknown type -> NotImplementedError (:82), repeated atoms ->
IndexError (:85), wrong atom count -> AssertionError (:88,91,94).  Atom groups are
duck-typed (``.ix``, ``len``, iteration, ``+``): MDAnalysis groups and
:class:`molann_amd.atomgroup.AtomGroup` both work.
"""

import pandas as pd

_TYPE_IDS = {"angle": 0, "bond": 1, "dihedral": 2, "position": 3}
_ATOM_COUNTS = {"angle": 3, "bond": 2, "dihedral": 4}
_COUNT_MSG = {
    "angle": "3 atoms are needed to define an angle feature, {} provided",
    "bond": "2 atoms are needed to define a bond length feature, {} provided",
    "dihedral": "4 atoms are needed to define a dihedral angle feature, {} provided",
}


class Feature(object):
    """One feature of the system: name, type ('angle' | 'bond' | 'dihedral' | 'position'), atoms."""

    def __init__(self, name, feature_type, atom_group):
        if feature_type not in _TYPE_IDS:
            raise NotImplementedError(f'feature {feature_type} not implemented!')
        if len(set(atom_group)) < len(atom_group):
            raise IndexError('atom group contains repeated elements!')
        if feature_type in _ATOM_COUNTS:
            assert len(atom_group) == _ATOM_COUNTS[feature_type], _COUNT_MSG[feature_type].format(len(atom_group))
        self.name = name
        self.type_name = feature_type
        self.atom_group = atom_group
        self.type_id = _TYPE_IDS[feature_type]

    def get_name(self):
        return self.name

    def get_type(self):
        return self.type_name

    def get_atom_indices(self):
        """1-based global indices of the atoms, in group order (`feature.py:123`)."""
        return self.atom_group.ix + 1

    def get_type_id(self):
        return self.type_id

    def get_feature_info(self):
        return pd.DataFrame({'name': self.name, 'type': self.type_name, 'type_id': self.type_id,
                             'atom indices (1-based)': [self.get_atom_indices()]})


class FeatureFileReader(object):
    """Reads one ``[section] ... [End]`` block of a feature file (`feature.py:147-194, 224-265`).

    Each feature line is ``name, type, selector[, selector ...]``; the selectors are passed to
    ``universe.select_atoms`` and concatenated in order.  ``universe`` is an MDAnalysis Universe or a
    :class:`molann_amd.atomgroup.Universe`.
    """

    def __init__(self, feature_file, section_name, universe):
        self.feature_file = feature_file
        self.section_name = section_name
        self.u = universe
        self.feature_list = []

    def read(self):
        self.feature_list = []
        in_section = False
        with open(self.feature_file, "r") as cfg:
            for line in cfg:
                line = line.strip()
                if not line or line.startswith("#"):
                    continue
                if line.startswith("["):
                    if line.strip('[]') == self.section_name:
                        in_section = True
                        continue
                    if in_section and line.strip('[]') == 'End':
                        break
                if in_section:
                    name, ftype, *selectors = line.split(',')
                    ag = None
                    for sel in selectors:
                        part = self.u.select_atoms(sel)
                        ag = part if ag is None else ag + part
                    self.feature_list.append(Feature(name.strip(), ftype.strip(), ag))
        return self.feature_list

    def get_feature_list(self):
        return self.feature_list

    def get_num_of_features(self):
        return len(self.feature_list)

    def get_feature_info(self):
        df = pd.DataFrame()
        for f in self.feature_list:
            df = pd.concat([df, f.get_feature_info()], ignore_index=True)
        return df
