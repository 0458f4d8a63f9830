"""Batch staging fast path.

The reference moves a batch to the device as three pickled DGL objects per step: ``dgl.batch`` of the crystals'
graphs, ``dgl.batch`` of their line graphs and the lattices (``alignn/lmdb_dataset.py:56-108``), then
``g.to(device)``, ``lg.to(device)`` (``alignn/train.py:264-270``) - for a 64-crystal batch that is the T-sized
int64 COO list of L(g) (10.8 MB) plus its cosines, on the critical path of every step, and the model cannot start
before the copy is done.  At >3 000 graphs/s (20 ms per 64-crystal step) that path is the bottleneck.

Here a batch crosses PCIe as ONE pinned buffer holding only what cannot be derived on the device:

    int32  u[E], v[E]            bond graph COO (both directions, caller's order)
    int32  batch_num_nodes[B]
    f32    r[E,3]                bond vectors
    f32    lattice[B,3,3]
    f32    target[B]             (optional)
    f32    atom_features[N,F]    - or, index-compressed, int32 species[N] into a device-resident [Z,F] table

(2.4 MB, or 1.0 MB with the species table, instead of ~15 MB).  Everything else is rebuilt on the GPU on a STAGING
stream while the previous step computes: canonical CSR of g (``graph.build_csr``), L(g) straight from that CSR
(``graph.line_graph_of`` - no T-sized list is ever sent or sorted) and the bond-angle cosines
(``alignn_bond_cosine_fwd``, the reference's ``compute_bond_cosines``, ``alignn/graphs.py:847-864``).  The result
is the same canonical ``GraphBatch`` that ``GraphBatch.from_coo`` builds from the caller's explicit line graph.

``PrefetchLoader`` keeps ``depth`` batches in flight: copy + staging of batch i+1 overlap the training step of
batch i; the consumer's stream waits on the staging event only, never on the host.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, Iterator, Optional

import numpy as np
import torch

from . import ops
from .graph import GraphBatch, _ptr_from_counts, build_csr, line_graph_of

__all__ = ["PackedBatch", "pack", "pack_raw", "stage", "PrefetchLoader"]

_ALIGN = 64  # bytes: every section of the buffer starts 64-byte aligned (float4 / int4 device views)


def _up(n: int) -> int:
    return (n + _ALIGN - 1) // _ALIGN * _ALIGN


@dataclass
class PackedBatch:
    """One batch as a single (pinned when possible) host byte buffer plus the section table."""

    buf: torch.Tensor  # uint8 [nbytes], pinned host memory
    sections: dict  # name -> (byte offset, torch dtype, shape)
    num_nodes: int
    num_edges: int
    batch_size: int
    num_triplets: int = -1  # rows of L(g), derived from the bond list when the batch is packed (-1: unknown)
    max_in_degree: int = 0  # largest number of bonds arriving at one atom (CSRGraph.dense_max_src of the line graph)

    @property
    def nbytes(self) -> int:
        return int(self.buf.numel())

    def host(self, name: str) -> torch.Tensor:
        off, dtype, shape = self.sections[name]
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        return self.buf[off:off + n].view(dtype).view(shape)


def pack(u, v, batch_num_nodes, r, lattice, atom_features=None, species=None, target=None, pin: Optional[bool] = None) -> PackedBatch:
    """Pack numpy/torch arrays of one batch (reference layout: ``u, v`` int64 COO of the batched bond graph,
    ``r`` [E,3], ``lattice`` [B,3,3], per-atom ``atom_features`` [N,F] or ``species`` [N]) into one buffer.
    Indices are narrowed to int32 (a batch has far fewer than 2^31 atoms or bonds)."""
    if (atom_features is None) == (species is None):
        raise ValueError("give exactly one of atom_features / species")
    a = lambda x, dt: np.ascontiguousarray(np.asarray(x), dtype=dt)  # noqa: E731
    parts = [("u", a(u, np.int32)), ("v", a(v, np.int32)), ("batch_num_nodes", a(batch_num_nodes, np.int32)),
             ("r", a(r, np.float32)), ("lattice", a(lattice, np.float32))]
    if target is not None:
        parts.append(("target", a(target, np.float32)))
    if species is not None:
        parts.append(("species", a(species, np.int32)))
    else:
        parts.append(("atom_features", a(atom_features, np.float32)))
    E = parts[0][1].shape[0]
    if parts[1][1].shape[0] != E or parts[3][1].shape != (E, 3):
        raise ValueError("u, v, r disagree on the number of bonds")
    N = int(parts[2][1].sum())
    if int(parts[0][1].max(initial=-1)) >= N or int(parts[1][1].max(initial=-1)) >= N:
        raise ValueError("bond endpoint out of range")
    # what the device-side staging needs to know BEFORE it runs (sizes of its outputs, the dense-block bound of the
    # line-graph backward) and what is cheaper here than as device operations: functions of the bond list alone
    uu, vv, bnn = parts[0][1].astype(np.int64), parts[1][1].astype(np.int64), parts[2][1].astype(np.int64)
    din = np.bincount(vv, minlength=N)
    n_triplets = int(din[uu].sum() - np.count_nonzero(uu == vv))  # rows of L(g): (in-edges of src(e2)) - (e2 itself if self image)
    max_in = int(din[uu].max(initial=0))  # largest in-degree among atoms that have a bond leaving them (= line_graph_of's bound)
    gp = np.zeros(bnn.shape[0] + 1, dtype=np.int32)
    np.cumsum(bnn, out=gp[1:])
    parts.append(("graph_ptr", gp))
    lat64 = np.asarray(lattice, dtype=np.float64).reshape(-1, 3, 3)
    parts.append(("volume", np.abs(np.linalg.det(lat64)).astype(np.float32)))
    sections, off = {}, 0
    tdt = {np.dtype(np.int32): torch.int32, np.dtype(np.float32): torch.float32}
    for name, arr in parts:
        sections[name] = (off, tdt[arr.dtype], tuple(arr.shape))
        off = _up(off + arr.nbytes)
    if pin is None:
        pin = torch.cuda.is_available()
    buf = torch.empty(max(off, _ALIGN), dtype=torch.uint8, pin_memory=bool(pin))
    nb = buf.numpy()
    for name, arr in parts:
        o = sections[name][0]
        nb[o:o + arr.nbytes] = arr.reshape(-1).view(np.uint8)
    return PackedBatch(buf, sections, N, E, int(parts[2][1].shape[0]), n_triplets, max_in)


def pack_raw(raw, target=None, pin: Optional[bool] = None) -> PackedBatch:
    """From an ``alignn_amd.synthetic.RawGraph`` - its explicit line graph and cosines are simply not sent."""
    return pack(raw.u, raw.v, raw.batch_num_nodes, raw.r, raw.lattice, atom_features=raw.atom_features, target=target,
                pin=pin)


def stage(packed: PackedBatch, device, feature_table: Optional[torch.Tensor] = None, cosines: bool = True):
    """Copy the buffer to ``device`` (one async memcpy on the current stream) and rebuild the canonical batch there.
    Returns ``(GraphBatch, target or None)``.  ``feature_table`` [Z,F]: device-resident per-species features for a
    ``species`` batch.  ``cosines=False`` leaves ``h`` unset (models with ``lg_on_fly`` compute it themselves)."""
    dev = torch.device(device)
    d = packed.buf.to(dev, non_blocking=True)

    def view(name):
        off, dtype, shape = packed.sections[name]
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        return d[off:off + n].view(dtype).view(shape)

    if STAGE_HIP and dev.type == "cuda" and packed.num_triplets >= 0 and packed.num_edges > 0:
        return _stage_hip(packed, dev, d, view, feature_table, cosines)
    g = build_csr(view("u"), view("v"), packed.num_nodes)
    lg = line_graph_of(g)
    bnn = view("batch_num_nodes").to(torch.int64)
    gp = _ptr_from_counts(bnn).to(torch.int32)
    batch = GraphBatch(g=g, lg=lg, graph_ptr=gp, batch_size=packed.batch_size)
    if "species" in packed.sections:
        if feature_table is None:
            raise ValueError("a species batch needs the device feature table")
        batch.atom_features = feature_table.to(dev)[view("species").long()].contiguous()
    else:
        batch.atom_features = view("atom_features").contiguous()
    batch.r = view("r")[g.perm].contiguous()
    batch.volume = torch.linalg.det(view("lattice").double()).abs().float()
    if cosines:
        if dev.type != "cuda":
            raise RuntimeError("bond cosines are computed by the HIP kernel: stage(..., cosines=True) needs a GPU device")
        batch.h = ops.bond_cosines(batch.r, lg.src, lg.dst)
    target = view("target").clone() if "target" in packed.sections else None
    return batch, target


STAGE_HIP = True  # one C call (csrc/stage.hip) instead of ~60 torch index operations; tests flip it to compare (same arrays)
_I32, _I64, _F32 = torch.int32, torch.int64, torch.float32


def _stage_hip(packed: PackedBatch, dev, d, view, feature_table, cosines):
    """``stage`` through ``alignn_stage_batch``: ONE device allocation carved into every index array of (g, L(g)), the
    canonical bond vectors and the cosines, ONE C call (two stable radix sorts of E keys, two prefix sums, index
    arithmetic for the T rows) - the consumer's thread spends microseconds, not the ~5 ms of host time the chain of torch
    operations in ``graph.build_csr`` / ``graph.line_graph_of`` costs.  Same arrays, bit for bit (tests)."""
    from . import _lib

    lib = _lib.load()
    N, E, T = packed.num_nodes, packed.num_edges, packed.num_triplets
    want = [("seg_ptr", _I32, N + 1), ("src", _I32, E), ("dst", _I32, E), ("out_ptr", _I32, N + 1), ("out_slot", _I32, E),
            ("perm", _I64, E), ("inv", _I64, E), ("r", _F32, 3 * E), ("lg_seg_ptr", _I32, E + 1), ("lg_src", _I32, T),
            ("lg_dst", _I32, T), ("lg_out_ptr", _I32, E + 1), ("lg_out_slot", _I32, T), ("seg_rank", _I32, T),
            ("ident", _I64, T), ("h", _F32, T if cosines else 0)]
    offs, off = {}, 0
    for name, dt, n in want:
        offs[name] = off
        off = _up(off + n * (8 if dt is _I64 else 4))
    ws_bytes = lib.alignn_stage_batch_workspace(N, E)
    out = torch.empty(off + ws_bytes, dtype=torch.uint8, device=dev)
    base = out.data_ptr()
    a = {name: (out[offs[name]:offs[name] + n * (8 if dt is _I64 else 4)].view(dt) if n else None) for name, dt, n in want}
    if T == 0:  # (no bond pair shares an atom: empty arrays, as graph.csr_and_line_graph gives - never None inside a CSRGraph)
        for name, dt, _n in want:
            if a[name] is None and name != "h":
                a[name] = torch.empty(0, dtype=dt, device=dev)
    with _lib.device_guard(out):  # (_lib.stream() is the CURRENT device's stream: stage(packed, "cuda:1") while cuda:0 is current)
        rc = lib.alignn_stage_batch(
        view("u").data_ptr(), view("v").data_ptr(), view("r").data_ptr(), N, E, T,
        base + offs["seg_ptr"], base + offs["src"], base + offs["dst"], base + offs["out_ptr"], base + offs["out_slot"],
        base + offs["perm"], base + offs["inv"], base + offs["r"], base + offs["lg_seg_ptr"], base + offs["lg_src"],
        base + offs["lg_dst"], base + offs["lg_out_ptr"], base + offs["lg_out_slot"], base + offs["seg_rank"],
        base + offs["ident"], (base + offs["h"]) if cosines else None, None, base + off, ws_bytes, _lib.stream())
    _lib.check(rc, "stage_batch")
    from .graph import CSRGraph

    g = CSRGraph(n_nodes=N, n_edges=E, seg_ptr=a["seg_ptr"], seg_node=None, src=a["src"], dst=a["dst"], out_ptr=a["out_ptr"],
                 out_slot=a["out_slot"], perm=a["perm"], inv=a["inv"])
    lg = CSRGraph(n_nodes=E, n_edges=T, seg_ptr=a["lg_seg_ptr"], seg_node=a["out_slot"], src=a["lg_src"], dst=a["lg_dst"],
                  out_ptr=a["lg_out_ptr"], out_slot=a["lg_out_slot"], perm=a["ident"], inv=a["ident"],
                  grp_seg_ptr=a["out_ptr"], grp_src_ptr=a["seg_ptr"], dense_max_src=packed.max_in_degree, seg_rank=a["seg_rank"])
    batch = GraphBatch(g=g, lg=lg, graph_ptr=view("graph_ptr"), batch_size=packed.batch_size)
    batch.cache["staged_block"] = (out, d)  # the two device blocks every view above lives in
    if "species" in packed.sections:
        if feature_table is None:
            raise ValueError("a species batch needs the device feature table")
        batch.atom_features = feature_table.to(dev)[view("species").long()].contiguous()
    else:
        batch.atom_features = view("atom_features")
    batch.r = a["r"].view(E, 3)
    batch.volume = view("volume")
    if cosines:
        batch.h = a["h"]
    target = view("target") if "target" in packed.sections else None
    return batch, target


def _batch_tensors(b: GraphBatch):
    blk = b.cache.get("staged_block")
    if blk is not None:  # (every array is a view of these two allocations)
        yield from blk
        if b.atom_features is not None and b.atom_features.data_ptr() not in range(blk[1].data_ptr(), blk[1].data_ptr() + blk[1].numel()):
            yield b.atom_features
        return
    for csr in (b.g, b.lg):
        if csr is None:
            continue
        for t in (csr.seg_ptr, csr.seg_node, csr.src, csr.dst, csr.out_ptr, csr.out_slot, csr.perm, csr.inv,
                  csr.grp_seg_ptr, csr.grp_src_ptr, csr.seg_rank):
            if t is not None:
                yield t
    for t in (b.graph_ptr, b.atom_features, b.r, b.h, b.volume):
        if t is not None:
            yield t


class PrefetchLoader:
    """Iterate over ``PackedBatch``es, yielding staged ``(GraphBatch, target)`` pairs; the copy and the device-side
    rebuild of the next ``depth`` batches run on a staging stream underneath the consumer's compute."""

    def __init__(self, packed: Iterable[PackedBatch], device, depth: int = 2, feature_table: Optional[torch.Tensor] = None,
                 cosines: bool = True):
        self.source, self.device, self.depth = packed, torch.device(device), max(1, depth)
        self.table, self.cosines = feature_table, cosines
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    def _stage(self, p):
        if self.stream is None:
            return stage(p, self.device, self.table, self.cosines), None
        # no wait on the consumer's stream: staging must overlap the step that is computing (the feature table is
        # long-lived and assumed ready)
        with torch.cuda.stream(self.stream):
            out = stage(p, self.device, self.table, self.cosines)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return out, ev

    def __iter__(self) -> Iterator:
        it = iter(self.source)
        queue = []
        for p in it:
            queue.append(self._stage(p))
            if len(queue) > self.depth:
                yield self._hand_over(queue.pop(0))
        while queue:
            yield self._hand_over(queue.pop(0))

    def _hand_over(self, item):
        (batch, target), ev = item
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            # the batch was allocated on the staging stream but lives (and dies) under the consumer's stream
            for t in _batch_tensors(batch):
                t.record_stream(cur)
            if target is not None:
                target.record_stream(cur)
        return batch, target
