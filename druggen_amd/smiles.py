"""SMILES -> padded molecular graph, without RDKit.

The reference builds its inputs with RDKit + PyG (``src/data/dataset.py:119-160,
280-316`` and the encoder tables of ``src/data/utils.py:70-126``); neither is in
this image, so the part of that pipeline the hot path depends on is restated
here for the SMILES dialect the reference's own result files use (organic-subset
atoms, bracket atoms with charge / H count / chirality, ``- = # :`` bonds,
branches, ``1-9`` and ``%nn`` ring closures):

* atoms keep SMILES order, one node per heavy atom, label = index of the atomic
  number in ``sorted({0} | atomic numbers seen)`` (PAD first -- utils.py:98-106);
* bond label = index in ``[ZERO] + sorted(bond types seen)`` with RDKit's enum
  order SINGLE < DOUBLE < TRIPLE < AROMATIC (utils.py:101-108); a bond without a
  symbol is AROMATIC between two aromatic (lower-case) atoms, SINGLE otherwise;
* a molecule is kept when it has <= ``max_atom`` atoms and every atom has a bond
  (dataset.py:104,138-139 ``connected=True``);
* per-molecule tensors follow dataset.py:300-316: ``x`` one-hot ``[max_atom, m_dim]``,
  ``edge_index`` = row-major non-zeros of the dense label matrix, ``edge_attr`` the
  labels; ``collate`` offsets node ids by ``i * max_atom`` like a PyG ``Batch``.

Parity note: RDKit re-perceives aromaticity on parse; for SMILES *written by
RDKit* (the reference's CSVs are) the lower-case flags round-trip, which is the
only case claimed here.  Explicit ``[H]`` atoms and ``.``-separated fragments are
rejected rather than guessed at.
"""
from __future__ import annotations

import re
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

__all__ = ["SINGLE", "DOUBLE", "TRIPLE", "AROMATIC", "parse_smiles", "build_encoders", "molecule_graph",
           "collate", "SmilesError", "MolGraph", "GraphBatch"]

# RDKit BondType enum values (only their order matters: it fixes the label order)
SINGLE, DOUBLE, TRIPLE, AROMATIC = 1, 2, 3, 12

_ELEMENTS = ("H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr "
             "Rb Sr Y Zr Nb Mo Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe").split()
_Z = {sym: i + 1 for i, sym in enumerate(_ELEMENTS)}
_ORGANIC = ("Cl", "Br", "B", "C", "N", "O", "P", "S", "F", "I")
_AROMATIC_ORGANIC = ("b", "c", "n", "o", "p", "s")
_BRACKET = re.compile(r"^(\d+)?(se|as|[a-z]|[A-Z][a-z]?)(@{0,2}(?:TH\d|AL\d|SP\d|TB\d+|OH\d+)?)?(H\d*)?([+-]+\d*|[+-]\d+)?(:\d+)?$")
_BOND_SYMBOLS = {"-": SINGLE, "=": DOUBLE, "#": TRIPLE, ":": AROMATIC, "/": SINGLE, "\\": SINGLE}


class SmilesError(ValueError):
    pass


@dataclass
class MolGraph:
    smiles: str
    x: np.ndarray            # [max_atom, m_dim] one-hot float32
    edge_index: np.ndarray   # [2, nnz] int64
    edge_attr: np.ndarray    # [nnz] int64
    num_atoms: int


@dataclass
class GraphBatch:
    """The attributes ``load_molecules`` reads from a PyG ``Batch`` (utils.py:128-137)."""
    x: object
    edge_index: object
    edge_attr: object
    batch: object

    def to(self, device):
        return GraphBatch(*(t.to(device) for t in (self.x, self.edge_index, self.edge_attr, self.batch)))


def parse_smiles(smiles: str) -> Tuple[List[int], List[bool], List[Tuple[int, int, int]]]:
    """-> (atomic numbers, aromatic flags, bonds as (i, j, bond type)) in SMILES order."""
    atoms: List[int] = []
    arom: List[bool] = []
    bonds: List[Tuple[int, int, int]] = []
    stack: List[int] = []
    rings: Dict[int, Tuple[int, Optional[int]]] = {}
    prev: Optional[int] = None
    pending: Optional[int] = None
    i, n = 0, len(smiles)

    def bond_type(a: int, b: int, explicit: Optional[int]) -> int:
        if explicit is not None:
            return explicit
        return AROMATIC if arom[a] and arom[b] else SINGLE

    def add_atom(z: int, aromatic: bool):
        nonlocal prev, pending
        atoms.append(z)
        arom.append(aromatic)
        idx = len(atoms) - 1
        if prev is not None:
            bonds.append((prev, idx, bond_type(prev, idx, pending)))
        pending = None
        prev = idx

    while i < n:
        ch = smiles[i]
        if ch == "[":
            j = smiles.find("]", i)
            if j < 0:
                raise SmilesError(f"unclosed bracket atom in {smiles!r}")
            m = _BRACKET.match(smiles[i + 1:j])
            if not m:
                raise SmilesError(f"cannot parse bracket atom {smiles[i:j + 1]!r} in {smiles!r}")
            sym = m.group(2)
            aromatic = sym[0].islower()
            el = sym.capitalize()
            if el not in _Z:
                raise SmilesError(f"unknown element {sym!r} in {smiles!r}")
            if el == "H":
                raise SmilesError(f"explicit hydrogen atoms are not supported ({smiles!r})")
            add_atom(_Z[el], aromatic)
            i = j + 1
        elif ch in "-=#:/\\":
            pending = _BOND_SYMBOLS[ch]
            i += 1
        elif ch == "(":
            if prev is None:
                raise SmilesError(f"branch before any atom in {smiles!r}")
            stack.append(prev)
            i += 1
        elif ch == ")":
            if not stack:
                raise SmilesError(f"unbalanced ')' in {smiles!r}")
            prev = stack.pop()
            i += 1
        elif ch.isdigit() or ch == "%":
            if ch == "%":
                if not smiles[i + 1:i + 3].isdigit():
                    raise SmilesError(f"bad ring closure in {smiles!r}")
                label, i = int(smiles[i + 1:i + 3]), i + 3
            else:
                label, i = int(ch), i + 1
            if prev is None:
                raise SmilesError(f"ring closure before any atom in {smiles!r}")
            if label in rings:
                other, opened = rings.pop(label)
                if other == prev:
                    raise SmilesError(f"ring closure onto itself in {smiles!r}")
                explicit = pending if pending is not None else opened
                bonds.append((other, prev, bond_type(other, prev, explicit)))
            else:
                rings[label] = (prev, pending)
            pending = None
        elif ch == ".":
            raise SmilesError(f"disconnected fragments are not supported ({smiles!r})")
        else:
            for sym in _ORGANIC:
                if smiles.startswith(sym, i):
                    add_atom(_Z[sym], False)
                    i += len(sym)
                    break
            else:
                if ch in _AROMATIC_ORGANIC:
                    add_atom(_Z[ch.upper()], True)
                    i += 1
                else:
                    raise SmilesError(f"unexpected character {ch!r} at {i} in {smiles!r}")
    if stack:
        raise SmilesError(f"unbalanced '(' in {smiles!r}")
    if rings:
        raise SmilesError(f"unclosed ring {sorted(rings)} in {smiles!r}")
    if not atoms:
        raise SmilesError("empty SMILES")
    return atoms, arom, bonds


def build_encoders(smiles: Iterable[str], max_atom: int):
    """utils.py:70-126 -> (atom_encoder, atom_decoder, bond_encoder, bond_decoder, kept smiles, max_length).
    Unparseable strings are skipped like ``MolFromSmiles(...) is None`` (utils.py:85-86)."""
    atom_labels, bond_labels, kept, max_length = {0}, set(), [], 0
    for s in smiles:
        try:
            atoms, _, bonds = parse_smiles(s)
        except SmilesError:
            continue
        if len(atoms) > max_atom:
            continue
        kept.append(s)
        atom_labels.update(atoms)
        bond_labels.update(b[2] for b in bonds)
        max_length = max(max_length, len(atoms))
    atom_order = sorted(atom_labels)
    bond_order = [0] + sorted(bond_labels)
    atom_encoder = {z: i for i, z in enumerate(atom_order)}
    bond_encoder = {t: i for i, t in enumerate(bond_order)}
    return (atom_encoder, {i: z for z, i in atom_encoder.items()}, bond_encoder,
            {i: t for t, i in bond_encoder.items()}, kept, max_length)


def molecule_graph(smiles: str, atom_encoder: Dict[int, int], bond_encoder: Dict[int, int], max_atom: int) -> Optional[MolGraph]:
    """One dataset entry (dataset.py:296-316), or None when the reference would drop the molecule
    (too many atoms, unknown atom / bond type, an atom without bonds)."""
    atoms, _, bonds = parse_smiles(smiles)
    if len(atoms) > max_atom:
        return None
    if any(z not in atom_encoder for z in atoms) or any(t not in bond_encoder for _, _, t in bonds):
        return None
    adj = np.zeros((max_atom, max_atom), dtype=np.int64)
    for a, b, t in bonds:
        adj[a, b] = adj[b, a] = bond_encoder[t]
    if not (adj[:len(atoms), :len(atoms)].sum(-1) > 0).all():
        return None
    labels = np.array([atom_encoder[z] for z in atoms] + [0] * (max_atom - len(atoms)), dtype=np.int64)
    x = np.zeros((max_atom, len(atom_encoder)), dtype=np.float32)
    x[np.arange(max_atom), labels] = 1.0
    src, dst = np.nonzero(adj)
    return MolGraph(smiles, x, np.stack([src, dst]).astype(np.int64), adj[src, dst], len(atoms))


def collate(graphs: Sequence[MolGraph]):
    """PyG-style batch of padded graphs as torch tensors (what ``DataLoader`` hands ``load_molecules``)."""
    import torch
    if not graphs:
        raise ValueError("empty batch")
    n = graphs[0].x.shape[0]
    x = torch.from_numpy(np.concatenate([g.x for g in graphs], 0))
    edge_index = torch.from_numpy(np.concatenate([g.edge_index + i * n for i, g in enumerate(graphs)], 1))
    edge_attr = torch.from_numpy(np.concatenate([g.edge_attr for g in graphs], 0))
    batch = torch.arange(len(graphs)).repeat_interleave(n)
    return GraphBatch(x, edge_index, edge_attr, batch)
