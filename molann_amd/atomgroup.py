"""Minimal AtomGroup / Universe stand-ins so the path can be set up without MDAnalysis.

The reference duck-types its atom groups: `molann/ann.py:131-135,255-258` and
`molann/feature.py:84,123` only use ``.ix`` (0-based global indices, ndarray),
``.positions`` (float32 ``[k, 3]``), ``len()``, iteration and ``+``
(concatenation, `feature.py:258`).  Real MDAnalysis AtomGroups satisfy the same
protocol and are accepted everywhere an ``AtomGroup`` is accepted here.
"""

import numpy as np


class Atom(object):
    """One atom of an :class:`AtomGroup`; hashable on its global index so that
    ``len(set(group))`` detects repeats as `feature.py:84` expects."""

    __slots__ = ("ix", "position")

    def __init__(self, ix, position):
        self.ix = int(ix)
        self.position = position

    def __hash__(self):
        return hash(self.ix)

    def __eq__(self, other):
        return isinstance(other, Atom) and other.ix == self.ix

    def __repr__(self):
        return "<Atom %d>" % (self.ix + 1)


class AtomGroup(object):
    """Ordered list of atoms: global 0-based indices + their coordinates."""

    def __init__(self, ix, positions):
        self.ix = np.asarray(ix, dtype=np.int64).reshape(-1)
        self.positions = np.ascontiguousarray(np.asarray(positions, dtype=np.float32).reshape(-1, 3))
        if self.ix.shape[0] != self.positions.shape[0]:
            raise ValueError("ix and positions must have the same length")

    def __len__(self):
        return int(self.ix.shape[0])

    def __iter__(self):
        for i, p in zip(self.ix, self.positions):
            yield Atom(i, p)

    def __add__(self, other):
        return AtomGroup(np.concatenate([self.ix, np.asarray(other.ix)]),
                         np.concatenate([self.positions, np.asarray(other.positions, dtype=np.float32)]))

    def __getitem__(self, item):
        if isinstance(item, (int, np.integer)):
            return Atom(self.ix[item], self.positions[item])
        return AtomGroup(self.ix[item], self.positions[item])

    def __repr__(self):
        return "<AtomGroup with %d atoms>" % len(self)


class Universe(object):
    """All atoms of a system, with the small subset of the MDAnalysis selection
    language that the reference's own files use (`test/feature.txt`,
    `test/test_molann.py`): ``bynum`` (1-based, single numbers and ``a:b``
    ranges), ``resid``, ``name``, ``type`` and ``all``, joined by ``or``.

    As in MDAnalysis, one ``select_atoms`` call returns atoms SORTED by index
    whatever order they were written in (`feature.py:62-69`); to keep a literal
    order concatenate single selections (``sel('bynum 2') + sel('bynum 1')``),
    or use :meth:`atoms_by_number`, which preserves the order given.
    """

    def __init__(self, positions, names=None, resids=None, types=None):
        positions = np.asarray(positions, dtype=np.float32).reshape(-1, 3)
        n = positions.shape[0]
        self.atoms = AtomGroup(np.arange(n), positions)
        self.names = list(names) if names is not None else [""] * n
        self.resids = list(resids) if resids is not None else [1] * n
        self.types = list(types) if types is not None else [nm.lstrip("0123456789")[:1] for nm in self.names]

    @classmethod
    def from_pdb(cls, filename):
        """Read ATOM/HETATM records (columns per the PDB format) of one model."""
        pos, names, resids = [], [], []
        with open(filename, "r") as fh:
            for line in fh:
                if line.startswith("ENDMDL"):
                    break
                if not (line.startswith("ATOM") or line.startswith("HETATM")):
                    continue
                names.append(line[12:16].strip())
                resids.append(int(line[22:26]))
                pos.append([float(line[30:38]), float(line[38:46]), float(line[46:54])])
        return cls(pos, names=names, resids=resids)

    def atoms_by_number(self, numbers):
        """Atoms with the given 1-based numbers, in the ORDER given."""
        idx = np.asarray(list(numbers), dtype=np.int64) - 1
        if idx.size and (idx.min() < 0 or idx.max() >= len(self.atoms)):
            raise IndexError("atom number out of range")
        return self.atoms[idx]

    def _select_one(self, tokens):
        key, vals = tokens[0], tokens[1:]
        n = len(self.atoms)
        mask = np.zeros(n, dtype=bool)
        if key == "all":
            mask[:] = True
        elif key in ("bynum", "resid"):
            wanted = set()
            for v in vals:
                if ":" in v or "-" in v:
                    a, b = v.replace("-", ":").split(":")
                    wanted.update(range(int(a), int(b) + 1))
                else:
                    wanted.add(int(v))
            if key == "bynum":
                for w in wanted:
                    if 1 <= w <= n:
                        mask[w - 1] = True
            else:
                mask = np.array([r in wanted for r in self.resids], dtype=bool)
        elif key == "name":
            mask = np.array([nm in vals for nm in self.names], dtype=bool)
        elif key == "type":
            mask = np.array([t in vals for t in self.types], dtype=bool)
        else:
            raise NotImplementedError("selection keyword '%s' is not supported" % key)
        return mask

    def select_atoms(self, selector):
        clauses = [c.split() for c in selector.replace(" and ", " ").split(" or ")]
        mask = np.zeros(len(self.atoms), dtype=bool)
        for tokens in clauses:
            if tokens:
                mask |= self._select_one(tokens)
        return self.atoms[np.nonzero(mask)[0]]
