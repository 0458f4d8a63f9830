"""Edge-list goldens for the RADIUS graph builder (second half), written by the REFERENCE's
own ``radius_graph`` (alignn/graphs.py:267-364, imported unmodified on ``oracle/shims``; jarvis' ``Atoms`` is the shim's
minimal restatement) over the reference's own example structures ``alignn/examples/sample_data/*.vasp`` - the strategy the
reference's force-field configs select (alignn/examples/sample_data_ff/config_example_atomwise.json:6, cutoff 4.0) and
``alignn/ff/calculators.py:280-291`` therefore runs at every MD step.

    python oracle/make_golden_radius.py          (authoring container only: needs /root/reference)

-> tests/golden/radius_sample_data.npz: per structure and cutoff (3.0: sparse cells widen their cutoff; 4.0: the FF
example's; 5.0: the function's default) the reference's (u, v, image) arrays IN THE REFERENCE'S ORDER (torch.where order:
by source atom, then periodic image, then destination atom) and its bond vectors; for cutoff 8.0 only the edge count and
an order-dependent checksum (the lists are ~100 bonds per atom).  Test infrastructure only.
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
from alignn.graphs import radius_graph  # noqa: E402  (the reference)


def checksum(u, v, img):
    """order-dependent 64-bit mix of the edge list (tests/test_radius_graph.py evaluates the same expression)"""
    k = np.arange(1, len(u) + 1, dtype=np.uint64)
    key = (u.astype(np.uint64) * np.uint64(1000003) + v.astype(np.uint64)) * np.uint64(1000003)
    key = key + ((img[:, 0].astype(np.int64) + 64) * 16384 + (img[:, 1].astype(np.int64) + 64) * 128 + (img[:, 2].astype(np.int64) + 64)).astype(np.uint64)
    with np.errstate(over="ignore"):
        return int(np.sum(key * (k * np.uint64(2654435761) + np.uint64(12345)), dtype=np.uint64))


if __name__ == "__main__":
    out, names = {}, []
    files = sorted(glob.glob("/root/reference/alignn/examples/sample_data/*.vasp"))
    for i, f in enumerate(files):
        atoms = Atoms.from_poscar(f)
        names.append(os.path.basename(f))
        out[f"{i}.lat"] = atoms.lattice_mat
        out[f"{i}.frac"] = atoms.frac_coords
        for cut in (3.0, 4.0, 5.0, 8.0):
            u, v, r, images = radius_graph(atoms, cutoff=cut)
            u, v = u.numpy().astype(np.int32), v.numpy().astype(np.int32)
            img = np.rint(images.numpy()).astype(np.int8)
            tag = f"{i}.c{cut:g}"
            out[tag + ".n"] = np.int64(len(u))
            out[tag + ".sum"] = np.uint64(checksum(u, v, img))
            if cut < 8.0:
                out[tag + ".u"], out[tag + ".v"], out[tag + ".image"] = u.astype(np.int16), v.astype(np.int16), img
                out[tag + ".r"] = r.numpy().astype(np.float32)
        print(names[-1], "atoms", atoms.num_atoms, {c: int(out[f"{i}.c{c:g}.n"]) for c in (3.0, 4.0, 5.0, 8.0)})
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "radius_sample_data.npz"), **out)
    print(len(names), "structures")
