"""Synthetic JARVIS-DFT-shaped periodic crystals -> (g, L(g)) in the reference's input layout.

No jarvis / DGL needed.  Restates, in vectorised numpy, the input contract the
reference's graph builder produces:

* k-nearest-neighbour edges with ``max_neighbors=12`` inside ``cutoff=8`` A, every
  neighbour tied with the 12th kept (reference ``alignn/graphs.py:155-227``);
* canonised undirected multigraph: each bond keyed ``(min id, max id, image)``
  and emitted as the consecutive pair ``(s->t, +d), (t->s, -d)``
  (``graphs.py:128-153`` and ``:230-264``);
* ``r`` = Cartesian displacement src -> dst (``graphs.py:550``);
* line graph: node ``i`` of L(g) is edge ``i`` of g, ``e1 -> e2`` iff
  ``dst(e1) == src(e2)`` and ``e1 != e2`` (``g.line_graph(shared=True)``, ``graphs.py:588``);
* bond-angle cosine ``h = clamp(-r[e1].r[e2] / (|r[e1]||r[e2]|), -1, 1)`` (``graphs.py:847-864``);
* batching = disjoint union with cumulative offsets (``lmdb_dataset.py:87-108``).

The generator recipe (density 0.05 atoms/A^3, sheared cell, 1.6 A rejection,
seed ``1234 + i`` for graph ``i``) is the one fixed in SURVEY.md section 8(d), so the
bench, the tests and the CPU baseline all see identical batches.
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np

__all__ = ["RawGraph", "make_crystal", "make_molecule", "knn_multigraph", "line_graph_coo", "make_batch", "batch_raw"]


@dataclass
class RawGraph:
    """One (g, L(g)) pair (or a batch of them) as plain numpy COO arrays.

    ``u, v``         int64 [E]   bond graph edges, src -> dst
    ``r``            f32   [E,3] Cartesian displacement src -> dst
    ``atom_features``f32   [N,F]
    ``lg_u, lg_v``   int64 [T]   line-graph edges (ids of g's edges)
    ``h``            f32   [T]   bond-angle cosines
    ``batch_num_nodes / batch_num_edges / batch_num_triplets`` int64 [B]
    ``lattice``      f32   [B,3,3]
    """

    u: np.ndarray
    v: np.ndarray
    r: np.ndarray
    atom_features: np.ndarray
    lg_u: np.ndarray
    lg_v: np.ndarray
    h: np.ndarray
    batch_num_nodes: np.ndarray
    batch_num_edges: np.ndarray
    batch_num_triplets: np.ndarray
    lattice: np.ndarray

    @property
    def num_nodes(self) -> int:
        return int(self.atom_features.shape[0])

    @property
    def num_edges(self) -> int:
        return int(self.u.shape[0])

    @property
    def num_triplets(self) -> int:
        return int(self.lg_u.shape[0])

    @property
    def batch_size(self) -> int:
        return int(self.batch_num_nodes.shape[0])


# --------------------------------------------------------------------------
# structures
# --------------------------------------------------------------------------
def make_crystal(n_atoms: int, seed: int, density: float = 0.05, min_dist: float = 1.6):
    """Random periodic cell: returns (lattice[3,3] rows = a,b,c; frac[n,3]; Z[n])."""
    rng = np.random.default_rng(seed)
    edge = (n_atoms / density) ** (1.0 / 3.0)
    lat = np.diag(rng.uniform(0.85, 1.15, 3) * edge)
    lat[1, 0] = rng.uniform(-1.0, 1.0)
    lat[2, 1] = rng.uniform(-1.0, 1.0)
    # rescale so the density is exact
    lat *= (n_atoms / density / abs(np.linalg.det(lat))) ** (1.0 / 3.0)
    frac = np.empty((n_atoms, 3))
    k = 0
    tries = 0
    while k < n_atoms:
        cand = rng.uniform(0.0, 1.0, 3)
        tries += 1
        if k:
            d = frac[:k] - cand
            d -= np.round(d)
            if np.min(np.linalg.norm(d @ lat, axis=1)) < min_dist and tries < 200000:
                continue
        frac[k] = cand
        k += 1
    z = rng.integers(1, 93, n_atoms)
    return lat, frac, z


def make_molecule(n_atoms: int, seed: int, box: float = 500.0, spread: float = 1.4):
    """Non-periodic cluster in a huge box (QM9-shaped stress case): no image neighbours."""
    rng = np.random.default_rng(seed)
    lat = np.eye(3) * box
    radius = spread * n_atoms ** (1.0 / 3.0) * 1.2
    pts = np.empty((n_atoms, 3))
    k = 0
    while k < n_atoms:
        cand = rng.uniform(-radius, radius, 3)
        if k and np.min(np.linalg.norm(pts[:k] - cand, axis=1)) < 1.0:
            continue
        pts[k] = cand
        k += 1
    frac = pts / box + 0.5
    z = rng.integers(1, 10, n_atoms)
    return lat, frac, z


# --------------------------------------------------------------------------
# bond graph
# --------------------------------------------------------------------------
def _all_neighbors(lat, frac, cutoff):
    """(src, dst, image[3], dist) of every periodic neighbour within ``cutoff``."""
    n = frac.shape[0]
    # number of images needed along each lattice vector: cutoff / plane spacing
    inv = np.linalg.inv(lat)
    spacing = 1.0 / np.linalg.norm(inv, axis=0)
    reach = np.ceil(cutoff / spacing).astype(int)
    rng_ = [np.arange(-k, k + 1) for k in reach]
    images = np.stack(np.meshgrid(*rng_, indexing="ij"), -1).reshape(-1, 3)
    # distances as explicit elementwise float64 operations in a FIXED order (no BLAS, no fused multiply-add): the tie
    # decisions at the shell of the 12th neighbour then agree bit for bit with the oracle's neighbour list
    # (oracle/shims/jarvis/core/atoms.py) and with the device builder (alignn_amd/neighbors.py)
    cart = frac[:, 0:1] * lat[0] + frac[:, 1:2] * lat[1] + frac[:, 2:3] * lat[2]
    imf = images.astype(np.float64)
    shift = imf[:, 0:1] * lat[0] + imf[:, 1:2] * lat[1] + imf[:, 2:3] * lat[2]  # [I,3]
    # d[i, j, I] = (cart[j] + shift[I]) - cart[i]
    d = (cart[None, :, None, :] + shift[None, None, :, :]) - cart[:, None, None, :]
    dist = np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2])
    mask = (dist <= cutoff) & (dist > 1e-8)
    src, dst, img = np.nonzero(mask)
    return src, dst, images[img], dist[src, dst, img], n


def knn_multigraph(lat, frac, cutoff: float = 8.0, max_neighbors: int = 12):
    """k-NN canonised undirected multigraph; returns (u, v, r) with paired directions."""
    n = frac.shape[0]
    k_eff = min(max_neighbors, max(n - 1, 1)) if np.isinf(cutoff) else max_neighbors
    while True:
        src, dst, img, dist, _ = _all_neighbors(lat, frac, cutoff)
        counts = np.bincount(src, minlength=n)
        if counts.min() >= k_eff:
            break
        # too few neighbours somewhere: widen the search sphere (graphs.py:170-188)
        lengths = np.linalg.norm(lat, axis=1)
        cutoff = float(lengths.max()) if cutoff < lengths.max() else 2.0 * cutoff
    # per-site: keep everything out to the shell of the k-th neighbour
    order = np.lexsort((dist, src))
    src, dst, img, dist = src[order], dst[order], img[order], dist[order]
    first = np.concatenate(([0], np.cumsum(counts)[:-1]))
    kth = dist[first + k_eff - 1]
    keep = dist <= kth[src]
    src, dst, img = src[keep], dst[keep], img[keep]
    # canonical key: smaller id first, image measured from the first vertex
    swap = dst < src
    a = np.where(swap, dst, src)
    b = np.where(swap, src, dst)
    im = np.where(swap[:, None], -img, img)
    key = np.concatenate([a[:, None], b[:, None], im], 1)
    key = np.unique(key, axis=0)  # set semantics + deterministic (sorted) order
    a, b, im = key[:, 0], key[:, 1], key[:, 2:5]
    d = (frac[b] + im - frac[a]) @ lat
    u = np.stack([a, b], 1).reshape(-1)
    v = np.stack([b, a], 1).reshape(-1)
    r = np.stack([d, -d], 1).reshape(-1, 3)
    return u.astype(np.int64), v.astype(np.int64), r


def molecule_graph(lat, frac, max_neighbors: int = 12):
    """k-NN graph of an isolated cluster (k capped at n-1), same canonisation."""
    n = frac.shape[0]
    cart = frac @ lat
    d = cart[None] - cart[:, None]  # d[i,j] = cart[j]-cart[i]
    dist = np.linalg.norm(d, axis=-1)
    np.fill_diagonal(dist, np.inf)
    k = min(max_neighbors, n - 1)
    kth = np.sort(dist, axis=1)[:, k - 1]
    src, dst = np.nonzero(dist <= kth[:, None])
    a, b = np.minimum(src, dst), np.maximum(src, dst)
    key = np.unique(np.stack([a, b], 1), axis=0)
    a, b = key[:, 0], key[:, 1]
    dd = cart[b] - cart[a]
    u = np.stack([a, b], 1).reshape(-1)
    v = np.stack([b, a], 1).reshape(-1)
    r = np.stack([dd, -dd], 1).reshape(-1, 3)
    return u.astype(np.int64), v.astype(np.int64), r


# --------------------------------------------------------------------------
# line graph
# --------------------------------------------------------------------------
def line_graph_coo(u: np.ndarray, v: np.ndarray, n_nodes: int):
    """(e1, e2) with v[e1] == u[e2], e1 != e2; emitted e1-major (order is unspecified in DGL)."""
    m = u.shape[0]
    order = np.argsort(u, kind="stable")  # edges grouped by source
    out_deg = np.bincount(u, minlength=n_nodes)
    ptr = np.concatenate(([0], np.cumsum(out_deg)))
    succ = out_deg[v]  # successors of every e1 = out-edges of its destination
    e1 = np.repeat(np.arange(m), succ)
    first = np.cumsum(succ) - succ
    within = np.arange(e1.shape[0]) - first[e1]
    e2 = order[ptr[v][e1] + within]
    keep = e1 != e2
    return e1[keep].astype(np.int64), e2[keep].astype(np.int64)


def bond_cosines(r: np.ndarray, e1: np.ndarray, e2: np.ndarray):
    r1 = -r[e1]
    r2 = r[e2]
    c = np.sum(r1 * r2, 1) / (np.linalg.norm(r1, axis=1) * np.linalg.norm(r2, axis=1))
    return np.clip(c, -1.0, 1.0)


# --------------------------------------------------------------------------
# batches
# --------------------------------------------------------------------------
def _one(n_atoms, seed, kind, n_features):
    if kind == "crystal":
        lat, frac, z = make_crystal(n_atoms, seed)
        u, v, r = knn_multigraph(lat, frac)
    elif kind == "molecule":
        lat, frac, z = make_molecule(n_atoms, seed)
        u, v, r = molecule_graph(lat, frac)
    else:  # pragma: no cover
        raise ValueError(kind)
    # float32 displacement is what the reference stores (graphs.py:550); the
    # cosines are then computed from that float32 ``r`` (graphs.py:589).
    r = r.astype(np.float32)
    e1, e2 = line_graph_coo(u, v, n_atoms)
    h = bond_cosines(r.astype(np.float32), e1, e2).astype(np.float32)
    feats = np.zeros((n_atoms, n_features), np.float32)
    feats[np.arange(n_atoms), (z - 1) % n_features] = 1.0
    # cgcnn-style rows are multi-hot; add a few deterministic extra bits
    feats[np.arange(n_atoms), (z * 7 + 3) % n_features] = 1.0
    return RawGraph(
        u,
        v,
        r,
        feats,
        e1,
        e2,
        h,
        np.array([n_atoms], np.int64),
        np.array([u.shape[0]], np.int64),
        np.array([e1.shape[0]], np.int64),
        lat.astype(np.float32)[None],
    )


def batch_raw(graphs) -> RawGraph:
    n_off = np.cumsum([0] + [g.num_nodes for g in graphs])
    e_off = np.cumsum([0] + [g.num_edges for g in graphs])
    return RawGraph(
        np.concatenate([g.u + n_off[i] for i, g in enumerate(graphs)]),
        np.concatenate([g.v + n_off[i] for i, g in enumerate(graphs)]),
        np.concatenate([g.r for g in graphs]),
        np.concatenate([g.atom_features for g in graphs]),
        np.concatenate([g.lg_u + e_off[i] for i, g in enumerate(graphs)]),
        np.concatenate([g.lg_v + e_off[i] for i, g in enumerate(graphs)]),
        np.concatenate([g.h for g in graphs]),
        np.concatenate([g.batch_num_nodes for g in graphs]),
        np.concatenate([g.batch_num_edges for g in graphs]),
        np.concatenate([g.batch_num_triplets for g in graphs]),
        np.concatenate([g.lattice for g in graphs]),
    )


def make_batch(
    batch_size: int = 64,
    n_atoms: int | tuple[int, int] = 60,
    seed0: int = 1234,
    kind: str = "crystal",
    n_features: int = 92,
) -> RawGraph:
    """``batch_size`` graphs; graph ``i`` uses seed ``seed0 + i``.

    ``n_atoms`` may be an inclusive ``(lo, hi)`` range (ragged batches / cfg 5).
    """
    return batch_raw(make_graphs(batch_size, n_atoms, seed0, kind, n_features))


def make_graphs(batch_size: int = 64, n_atoms: int | tuple[int, int] = 60, seed0: int = 1234, kind: str = "crystal",
                n_features: int = 92) -> list:
    """The single graphs of ``make_batch`` (graph ``i`` uses seed ``seed0 + i``), e.g. to shard a global batch by cost."""
    graphs = []
    for i in range(batch_size):
        if isinstance(n_atoms, tuple):
            n = int(np.random.default_rng(seed0 + i + 7919).integers(n_atoms[0], n_atoms[1] + 1))
        else:
            n = int(n_atoms)
        graphs.append(_one(n, seed0 + i, kind, n_features))
    return graphs
