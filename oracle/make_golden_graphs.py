"""Edge-set goldens for the kNN graph builder, written by the REFERENCE's own
``nearest_neighbor_edges`` / ``build_undirected_edgedata`` / ``canonize_edge`` (alignn/graphs.py:128-264), imported
unmodified on ``oracle/shims`` (jarvis' ``Atoms`` is the shim's minimal restatement, see its header), over the
reference's own example structures ``alignn/examples/sample_data/*.vasp``.

    python oracle/make_golden_graphs.py          (authoring container only: needs /root/reference)

-> tests/golden/graphs_sample_data.npz: per structure the lattice, fractional coordinates and the reference's
(u, v, image, r) arrays in the reference's own (dict-insertion) order.  Test infrastructure only.
"""

import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "shims"))
sys.path.insert(0, "/root/reference")

from jarvis.core.atoms import Atoms  # noqa: E402  (shim)
from alignn.graphs import build_undirected_edgedata, nearest_neighbor_edges  # noqa: E402  (the reference)

if __name__ == "__main__":
    out, names = {}, []
    files = sorted(glob.glob("/root/reference/alignn/examples/sample_data/*.vasp"))
    for i, f in enumerate(files):
        atoms = Atoms.from_poscar(f)
        edges, _ = nearest_neighbor_edges(atoms=atoms, cutoff=8.0, max_neighbors=12, use_canonize=True)
        u, v, r, images = build_undirected_edgedata(atoms, edges)
        names.append(os.path.basename(f))
        out[f"{i}.lat"] = atoms.lattice_mat
        out[f"{i}.frac"] = atoms.frac_coords
        out[f"{i}.u"] = u.numpy().astype(np.int32)
        out[f"{i}.v"] = v.numpy().astype(np.int32)
        out[f"{i}.image"] = images.numpy().astype(np.int8)
        out[f"{i}.r"] = r.numpy().astype(np.float32)
        print(names[-1], "atoms", atoms.num_atoms, "edges", len(u), "E/N", round(len(u) / atoms.num_atoms, 2))
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "graphs_sample_data.npz"), **out)
    print(len(names), "structures")
