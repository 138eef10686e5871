"""Numeric problem description handed across the C-ABI (include/ksfd_hip.h: ksfd_config).

The reference carries the same information in SolutionParameters / LigandGroups
(KSFD/ksfdsoln.py:104-161, KSFD/ksfdligand.py:306-388, 527-547); the operators only ever see the
already-expanded ligand table ps.Vgroups.ligands() and ps.values(t), which is what this holds.
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

MAX_LIG = 12


class CConfig(C.Structure):
    """Mirror of `ksfd_config` in include/ksfd_hip.h (field order must match)."""
    _fields_ = [('dim', C.c_int32), ('nlig', C.c_int32), ('ngroups', C.c_int32), ('cap_kind', C.c_int32),
                ('n', C.c_int64 * 3), ('L', C.c_double * 3),
                ('s2', C.c_double), ('rhomax', C.c_double), ('cushion', C.c_double),
                ('maxscale', C.c_double), ('rhomin', C.c_double), ('Umin', C.c_double),
                ('lig_group', C.POINTER(C.c_int32)),
                ('lig_w', C.POINTER(C.c_double)), ('lig_s', C.POINTER(C.c_double)),
                ('lig_gamma', C.POINTER(C.c_double)), ('lig_D', C.POINTER(C.c_double)),
                ('grp_alpha', C.POINTER(C.c_double)), ('grp_beta', C.POINTER(C.c_double))]


@dataclass
class ProblemConfig:
    dim: int
    n: tuple                      # (nx, ny, nz), unused axes = 1
    L: tuple                      # (width, height, depth)
    lig_group: np.ndarray         # [nlig] 0-based group of each ligand
    lig_w: np.ndarray
    lig_s: np.ndarray
    lig_gamma: np.ndarray
    lig_D: np.ndarray
    grp_alpha: np.ndarray         # [ngroups]
    grp_beta: np.ndarray
    s2: float = 0.02357 ** 2 / 2
    rhomax: float = 28000.0
    cushion: float = 2000.0
    maxscale: float = 2.0
    rhomin: float = 1e-7
    Umin: float = 1e-7
    cap_kind: int = 0             # 0 tophat, 1 witch  (--cappotential, ksfdsolver2.py:387-389)
    _keep: list = field(default_factory=list, repr=False)

    def __post_init__(self):
        n = [int(x) for x in self.n] + [1] * (3 - len(self.n))
        L = [float(x) for x in self.L] + [1.0] * (3 - len(self.L))
        for a in range(self.dim, 3):
            n[a] = 1
        self.n, self.L = tuple(n), tuple(L)
        self.lig_group = np.ascontiguousarray(self.lig_group, dtype=np.int32)
        for k in ('lig_w', 'lig_s', 'lig_gamma', 'lig_D', 'grp_alpha', 'grp_beta'):
            setattr(self, k, np.ascontiguousarray(getattr(self, k), dtype=np.float64))
        if not (1 <= self.dim <= 3):
            raise ValueError('dim must be 1, 2 or 3 (KSFD/ksfdgrid.py:159-162)')
        if self.nlig > MAX_LIG:
            raise ValueError('at most %d ligands supported' % MAX_LIG)
        if self.nlig and (self.lig_group.min() < 0 or self.lig_group.max() >= self.ngroups):
            raise ValueError('lig_group out of range')

    @property
    def nlig(self):
        return int(self.lig_group.size)

    @property
    def ngroups(self):
        return int(self.grp_alpha.size)

    @property
    def F(self):
        return self.nlig + 1

    @property
    def N(self):
        return self.n[0] * self.n[1] * self.n[2]

    @property
    def spacing(self):
        return tuple(self.L[a] / self.n[a] for a in range(self.dim))

    def as_ctypes(self):
        c = CConfig()
        c.dim, c.nlig, c.ngroups, c.cap_kind = self.dim, self.nlig, self.ngroups, int(self.cap_kind)
        for a in range(3):
            c.n[a] = self.n[a]
            c.L[a] = self.L[a]
        for k in ('s2', 'rhomax', 'cushion', 'maxscale', 'rhomin', 'Umin'):
            setattr(c, k, float(getattr(self, k)))
        c.lig_group = self.lig_group.ctypes.data_as(C.POINTER(C.c_int32))
        for k in ('lig_w', 'lig_s', 'lig_gamma', 'lig_D', 'grp_alpha', 'grp_beta'):
            setattr(c, k, getattr(self, k).ctypes.data_as(C.POINTER(C.c_double)))
        return c

    @classmethod
    def from_golden(cls, z):
        """Build from the meta keys of a tests/golden/*.npz file."""
        return cls(dim=int(z['dim']), n=tuple(int(x) for x in z['n']), L=tuple(float(x) for x in z['L']),
                   lig_group=z['lig_group'], lig_w=z['lig_w'], lig_s=z['lig_s'], lig_gamma=z['lig_gamma'],
                   lig_D=z['lig_D'], grp_alpha=z['grp_alpha'], grp_beta=z['grp_beta'],
                   s2=float(z['s2']), rhomax=float(z['rhomax']), cushion=float(z['cushion']),
                   maxscale=float(z['maxscale']), rhomin=float(z['rhomin']), Umin=float(z['Umin']),
                   cap_kind=int(z['cap_kind']))

    @classmethod
    def standard(cls, dim, n, L=None, nlig=1, cap_kind=0):
        """The ligand sets of SURVEY.md 8d / options84:32-46: attractant (+ repellent for nlig=2)."""
        n = tuple(n)
        L = tuple(L) if L is not None else (1.0,) * dim
        if nlig == 1:
            kw = dict(lig_group=[0], lig_w=[1.0], lig_s=[0.01], lig_gamma=[0.01], lig_D=[1e-6],
                      grp_alpha=[1500.0], grp_beta=[5.56e-4])
        elif nlig == 2:
            kw = dict(lig_group=[0, 1], lig_w=[1.0, 1.0], lig_s=[0.01, 0.001], lig_gamma=[0.01, 0.001],
                      lig_D=[1e-6, 1e-5], grp_alpha=[1500.0, 1500.0], grp_beta=[5.56e-4, -5.56e-4])
        else:
            raise ValueError('standard() knows nlig 1 or 2')
        return cls(dim=dim, n=n, L=L, cap_kind=cap_kind, **kw)
