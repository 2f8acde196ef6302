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


def _tokenise(path):
    """The feature file as a list of ``(line number, kind, text)`` with kind ``'header'`` (text = the name between
    the brackets) or ``'entry'``; blank lines and ``#`` comment lines are dropped (`feature.py:147-194`)."""
    tokens = []
    with open(path, "r") as fh:
        for lineno, raw in enumerate(fh, 1):
            text = raw.strip()
            if not text or text.startswith("#"):
                continue
            if text.startswith("["):
                tokens.append((lineno, "header", text.strip("[]")))
            else:
                tokens.append((lineno, "entry", text))
    return tokens


def _section_entries(path, section_name):
    """The entry lines of the FIRST ``[section_name]`` block: from its header to the next ``[End]`` (or the end of
    the file).  Everything before that header is skipped whatever it is, as the reference does
    (`feature.py:241-247`); a section that does not exist gives no entries.  One difference, on files the reference
    cannot read either: another ``[Header]`` inside the open block - which the reference would hand to
    ``str.split(',')`` and die of an unpacking ValueError - is rejected by name."""
    tokens = _tokenise(path)
    start = next((i for i, (_, kind, text) in enumerate(tokens) if kind == "header" and text == section_name), None)
    if start is None:
        return []
    entries = []
    for lineno, kind, text in tokens[start + 1:]:
        if kind == "header":
            if text == "End":
                break
            if text == section_name:      # the reference treats a repeated header of the open section as a no-op
                continue
            raise ValueError("%s:%d: section [%s] opened inside section [%s] (missing [End]?)"
                             % (path, lineno, text, section_name))
        entries.append((lineno, text))
    return entries


class FeatureFileReader(object):
    """Reads one ``[section] ... [End]`` block of a feature file (`feature.py:147-194, 224-265`).

    Each feature line is ``name, type, selector[, selector ...]``; the selectors are passed to
    ``universe.select_atoms`` and concatenated in order.  ``universe`` is an MDAnalysis Universe or a
    :class:`molann_amd.atomgroup.Universe`.  A section that does not exist yields an empty list, as in the reference.
    """

    def __init__(self, feature_file, section_name, universe):
        self.feature_file = feature_file
        self.section_name = section_name
        self.u = universe
        self.feature_list = []

    def _feature_from_line(self, lineno, text):
        fields = text.split(",")
        if len(fields) < 2:
            raise ValueError("%s:%d: expected 'name, type, selector[, selector ...]', got %r"
                             % (self.feature_file, lineno, text))
        name, ftype, selectors = fields[0].strip(), fields[1].strip(), fields[2:]
        group = None
        for selector in selectors:
            part = self.u.select_atoms(selector)
            group = part if group is None else group + part
        return Feature(name, ftype, group)

    def read(self):
        self.feature_list = [self._feature_from_line(lineno, text)
                             for lineno, text in _section_entries(self.feature_file, self.section_name)]
        return self.feature_list

    def get_feature_list(self):
        return self.feature_list

    def get_num_of_features(self):
        return len(self.feature_list)

    def get_feature_info(self):
        frames = [f.get_feature_info() for f in self.feature_list]
        return pd.concat(frames, ignore_index=True) if frames else pd.DataFrame()
