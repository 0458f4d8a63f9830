"""Minimal stand-in for ``jarvis.core.atoms`` (jarvis-tools is not installable here; alignn pins
``jarvis-tools>=2021.07.19`` in setup.py).  Test infrastructure only.

Just enough of ``Atoms`` for the reference's graph builder (alignn/graphs.py:155-264) to run unmodified on a POSCAR:
``lattice`` (``a, b, c``, ``cart_coords(frac)``), ``frac_coords``, ``cart_coords``, ``elements`` and
``get_all_neighbors(r)``, which restates jarvis' published algorithm (itself adapted from pymatgen): every periodic
image of every atom within ``r`` of a site, as ``[site index, neighbour index, distance, image]`` rows, self-distances
and anything closer than ``bond_tol`` dropped.  The image shift is evaluated as explicit elementwise float64 operations
(``i*a + j*b + k*c + cart``, squares summed x + y + z, one sqrt) so that the distances - and with them the tie decisions
at the shell of the 12th neighbour - are reproducible bit for bit by the builders under test.
"""

import itertools
import math

import numpy as np


class Lattice:
    def __init__(self, lattice_mat):
        self.matrix = np.array(lattice_mat, dtype=np.float64)

    @property
    def abc(self):
        return [float(np.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])) for v in self.matrix]

    a = property(lambda self: self.abc[0])
    b = property(lambda self: self.abc[1])
    c = property(lambda self: self.abc[2])

    def cart_coords(self, frac_coords):
        f = np.asarray(frac_coords, dtype=np.float64)
        m = self.matrix
        return f[..., 0:1] * m[0] + f[..., 1:2] * m[1] + f[..., 2:3] * m[2]

    def reciprocal_lattice_abc(self):
        rec = 2 * math.pi * np.linalg.inv(self.matrix).T
        return np.linalg.norm(rec, axis=1)


class Atoms:
    def __init__(self, lattice_mat=None, coords=None, elements=None, cartesian=False):
        self.lattice_mat = np.array(lattice_mat, dtype=np.float64)
        self.lattice = Lattice(self.lattice_mat)
        coords = np.array(coords, dtype=np.float64)
        self.frac_coords = coords @ np.linalg.inv(self.lattice_mat) if cartesian else coords
        self.cart_coords = self.lattice.cart_coords(self.frac_coords)
        self.elements = list(elements)
        self.num_atoms = len(self.elements)

    @staticmethod
    def from_poscar(path):
        """VASP 5 POSCAR: comment, scale, 3 lattice rows, species, counts, direct|cartesian, positions."""
        with open(path) as f:
            ln = [x.strip() for x in f.read().splitlines() if x.strip()]
        scale = float(ln[1])
        lat = np.array([[float(x) for x in ln[i].split()[:3]] for i in (2, 3, 4)]) * scale
        species, counts = ln[5].split(), [int(x) for x in ln[6].split()]
        k = 7
        if ln[k][0] in "sS":  # selective dynamics
            k += 1
        cartesian = ln[k][0] in "cCkK"
        n = sum(counts)
        pos = np.array([[float(x) for x in ln[k + 1 + i].split()[:3]] for i in range(n)])
        if cartesian:
            pos = pos * scale
        elements = [s for s, c in zip(species, counts) for _ in range(c)]
        return Atoms(lattice_mat=lat, coords=pos, elements=elements, cartesian=cartesian)

    def get_all_neighbors(self, r=5, bond_tol=0.15):
        recp_len = self.lattice.reciprocal_lattice_abc()
        maxr = np.ceil((r + bond_tol) * recp_len / (2 * math.pi))
        nmin = np.floor(np.min(self.frac_coords, axis=0)) - maxr
        nmax = np.ceil(np.max(self.frac_coords, axis=0)) + maxr
        all_ranges = [np.arange(x, y) for x, y in zip(nmin, nmax)]
        m = self.lattice_mat
        site = self.cart_coords
        n = len(site)
        neighbors = [list() for _ in range(n)]
        for image in itertools.product(*all_ranges):
            shift = image[0] * m[0] + image[1] * m[1] + image[2] * m[2]
            coords = site + shift  # image of every atom j
            d = coords[:, None, :] - site[None, :, :]  # d[j, i] = image(j) - site(i)
            sq = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2]
            dist = np.sqrt(sq)
            jj, ii = np.nonzero((dist <= r) & (dist > 1e-8) & (dist > bond_tol))
            for j, i in zip(jj, ii):
                neighbors[i].append([i, j, dist[j, i], image])
        return neighbors


def get_supercell_dims(*a, **k):
    raise NotImplementedError
