"""Canonical device layout of a batched (g, L(g)) pair.

The reference hands the model two DGL graphs in arbitrary COO order
(``alignn/train.py:264-270``; built by ``alignn/graphs.py:472-592`` and batched by
``alignn/lmdb_dataset.py:87-108``).  The kernels want *segments*: all in-edges of one destination
stored contiguously, so that one wavefront owns one destination and edge-feature row ``k`` is CSR
slot ``k`` (see include/alignn_hip.h).  This module converts once per batch (pure index work,
device agnostic so it is unit-testable on CPU) and the result is cached on the batch.

Layout chosen for locality on MI355X:

* ``g``: edges sorted by destination atom -> the in-edges of atom ``j`` are consecutive rows.
* ``L(g)``: its nodes are g's edges *in that canonical order*; its segments (one per bond
  ``e2 = j->k``) are ordered by the bond's source atom ``j``.  All bonds leaving atom ``j`` gather
  the same block of rows (the in-edges of ``j``), so consecutive wavefronts re-read the same
  ~13 KiB-per-tensor block from L1/L2 instead of HBM.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import torch


@dataclass
class CSRGraph:
    """One graph in canonical segment order (all index tensors int32, on the compute device)."""

    n_nodes: int
    n_edges: int
    seg_ptr: torch.Tensor  # [n_nodes+1] segment s covers slots [seg_ptr[s], seg_ptr[s+1])
    seg_node: Optional[torch.Tensor]  # [n_nodes] node of segment s (None: identity)
    src: torch.Tensor  # [m] source node of slot k
    dst: torch.Tensor  # [m] destination node of slot k
    out_ptr: torch.Tensor  # [n_nodes+1] by-source grouping ...
    out_slot: torch.Tensor  # [m] ... of slot ids
    perm: torch.Tensor  # [m] int64: slot k holds the caller's edge perm[k]
    inv: torch.Tensor  # [m] int64: caller's edge e lives in slot inv[e]
    # line graphs only: one dense (sources x segments) block per centre atom j of the parent graph -
    # segment ranks [grp_seg_ptr[j], grp_seg_ptr[j+1]) and source nodes [grp_src_ptr[j], grp_src_ptr[j+1])
    grp_seg_ptr: Optional[torch.Tensor] = None
    grp_src_ptr: Optional[torch.Tensor] = None
    # > 0: every block is DENSE and source-sorted - segment s of atom j lists ALL sources of j in ascending order,
    # minus s itself when the bond is a self-image - so the row of (source q, segment s) is pure index arithmetic
    # (see alignn_egc_bwd_lg_dense); the value is the largest number of sources of any atom.  0: not established.
    dense_max_src: int = 0
    # [m] int32, line graphs: the segment (rank in seg_ptr order) of every edge row - what the edge-gate projection's
    # epilogue gathers the destination term with (ops.gemm_nt_f16x3_gather with a segment-ordered table); built by
    # ``segment_rank()`` when the batch is staged
    seg_rank: Optional[torch.Tensor] = None

    def segment_rank(self) -> torch.Tensor:
        if self.seg_rank is None:
            sp = self.seg_ptr.long()
            self.seg_rank = torch.repeat_interleave(torch.arange(self.n_nodes, device=sp.device, dtype=torch.int32),
                                                    sp[1:] - sp[:-1], output_size=self.n_edges)
        return self.seg_rank


def _ptr_from_counts(counts: torch.Tensor) -> torch.Tensor:
    out = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=counts.device)
    out[1:] = torch.cumsum(counts, 0)
    return out


def build_csr(u: torch.Tensor, v: torch.Tensor, n_nodes: int, seg_order: Optional[torch.Tensor] = None,
              sort_sources: bool = False) -> CSRGraph:
    """Canonicalise COO edges ``u -> v``.

    ``seg_order`` (optional, int64 [n_nodes]) lists the nodes in the order their segments should be
    stored; default is node order.  Within a segment, edges keep the caller's relative order
    (stable), so the summation order inside a segment is a deterministic function of the input;
    ``sort_sources`` orders them by source node instead (line graphs: makes the blocks index-addressable).
    """
    u = u.to(torch.int64)
    v = v.to(torch.int64)
    m = int(u.numel())
    dev = u.device
    counts = torch.bincount(v, minlength=n_nodes)
    if seg_order is None:
        key = v
        seg_node = None
        seg_counts = counts
    else:
        rank = torch.empty(n_nodes, dtype=torch.int64, device=dev)
        rank[seg_order] = torch.arange(n_nodes, device=dev)
        key = rank[v]
        seg_node = seg_order.to(torch.int32)
        seg_counts = counts[seg_order]
    if sort_sources:
        key = key * n_nodes + u
    perm = torch.argsort(key, stable=True)
    inv = torch.empty(m, dtype=torch.int64, device=dev)
    inv[perm] = torch.arange(m, device=dev)
    src = u[perm]
    dst = v[perm]
    out_slot = torch.argsort(src, stable=True)
    out_ptr = _ptr_from_counts(torch.bincount(src, minlength=n_nodes))
    return CSRGraph(
        n_nodes=n_nodes,
        n_edges=m,
        seg_ptr=_ptr_from_counts(seg_counts).to(torch.int32),
        seg_node=seg_node,
        src=src.to(torch.int32),
        dst=dst.to(torch.int32),
        out_ptr=out_ptr.to(torch.int32),
        out_slot=out_slot.to(torch.int32),
        perm=perm,
        inv=inv,
    )


def line_graph_of(g: CSRGraph) -> CSRGraph:
    """Canonical line graph of a canonical bond graph, built on the device the graph lives on - the
    ``g.line_graph(shared=True)`` of the reference (alignn/graphs.py:588; rebuilt inside the forward at
    alignn/models/alignn_atomwise.py:376-386) without ever materialising a COO list in caller order.

    L(g) node i is g's slot i; ``e1 -> e2`` iff ``dst(e1) == src(e2)`` and ``e1 != e2`` (backtracking kept).  In
    the canonical layout this is one dense block per atom j: segments = out-edges of j in ``out_slot`` order,
    each listing the in-edges of j (the contiguous slots ``[seg_ptr[j], seg_ptr[j+1])``) minus itself.
    """
    dev = g.src.device
    m = g.n_edges
    seg_node = g.out_slot.long()  # e2 of every segment: out-edges grouped by source atom, slot order within
    atom = g.src.long()[seg_node]  # centre atom j of the segment
    sp = g.seg_ptr.long()
    din = (sp[1:] - sp[:-1])[atom]  # candidates e1 per segment (before the e1 != e2 exclusion)
    seg_of = torch.repeat_interleave(torch.arange(m, device=dev), din)
    first = torch.cumsum(din, 0) - din
    e1 = sp[atom][seg_of] + (torch.arange(seg_of.numel(), device=dev) - first[seg_of])
    e2 = seg_node[seg_of]
    keep = e1 != e2
    e1, e2, seg_of = e1[keep], e2[keep], seg_of[keep]
    counts = torch.bincount(seg_of, minlength=m)
    t = int(e1.numel())
    out_slot = torch.argsort(e1, stable=True)
    out_ptr = _ptr_from_counts(torch.bincount(e1, minlength=m))
    ident = torch.arange(t, device=dev)
    return CSRGraph(
        n_nodes=m,
        n_edges=t,
        seg_ptr=_ptr_from_counts(counts).to(torch.int32),
        seg_node=seg_node.to(torch.int32),
        src=e1.to(torch.int32),
        dst=e2.to(torch.int32),
        out_ptr=out_ptr.to(torch.int32),
        out_slot=out_slot.to(torch.int32),
        perm=ident,
        inv=ident,
        grp_seg_ptr=g.out_ptr,
        grp_src_ptr=g.seg_ptr,
        dense_max_src=int(din.max()) if m > 0 else 0,  # dense and source-sorted by construction (= _dense_blocks)
        seg_rank=seg_of.to(torch.int32),
    )


STAGE_HIP = True  # device COO -> (g, L(g)) through ONE C call (csrc/stage.hip) where it applies; tests flip it to compare


def csr_and_line_graph(u: torch.Tensor, v: torch.Tensor, n_nodes: int, r: Optional[torch.Tensor] = None, lg_edges=None):
    """``(build_csr(u, v, n), line_graph_of(g), r[g.perm])`` - the canonical bond graph, its canonical line graph and the
    bond vectors in slot order - for a bond list that already lives on the device.  On a HIP device this is ONE host read
    (T = rows of L(g) and the largest in-degree, both functions of the bond list) + ONE C call, ``alignn_stage_batch``
    (two radix sorts of E keys, two prefix sums, index arithmetic for the T rows: csrc/stage.hip) instead of the ~45 torch
    index operations of the two builders, whose arrays it reproduces bit for bit (tests/test_gpu_stage.py) - what an MD
    step pays between the neighbour search and the model (alignn/ff/calculators.py:280-291 rebuilds the graph every step).
    Elsewhere (CPU, empty graphs, the switch off): the two builders.

    ``lg_edges = (lg_u, lg_v)``: the caller's OWN edge list of L(g) (the ``lg`` of the reference's ``(g, lg)`` pair,
    alignn/graphs.py:588,995 ``g.line_graph(shared=True)``, in the caller's ids of g's edges).  The canonical rows are the
    same; ``lg.perm`` / ``lg.inv`` then map the caller's edge order onto them (``alignn_map_line_graph_rows``: index
    arithmetic, second host read = its error count).  Returns ``None`` when that list is NOT the line graph of g (a
    filtered one, duplicates, a different size) or the device path does not apply - the caller takes the generic builder."""
    dev = u.device
    E = int(u.numel())
    if not (STAGE_HIP and dev.type == "cuda" and E > 0 and n_nodes > 0):
        if lg_edges is not None:
            return None
        g = build_csr(u, v, n_nodes)
        return g, line_graph_of(g), (None if r is None else r[g.perm].contiguous())
    from . import _lib

    lib = _lib.load()
    u32, v32 = u.to(torch.int32).contiguous(), v.to(torch.int32).contiguous()
    # in-degrees by a scatter-add over CLAMPED endpoints (no hidden host read as in torch.bincount, no device-side assert on a
    # bad index); the range of the endpoints travels in the same host read as T and is checked right after it - the radix
    # sorts below use bits_for(N) key bits and would silently build a corrupted CSR from an endpoint outside [0, N)
    uc, vc = u32.long().clamp(0, n_nodes - 1), v32.long().clamp(0, n_nodes - 1)
    din = torch.zeros(n_nodes, dtype=torch.int64, device=dev).scatter_add_(0, vc, torch.ones_like(vc))
    din_u = din[uc]
    # the one host read: T = sum over bonds e2 of (in-degree of src(e2)) - [e2 is a self image]; the dense-block bound
    T, max_in, lo, hi = torch.stack([din_u.sum() - (u32 == v32).sum(), din_u.max(), torch.minimum(u32.min(), v32.min()).long(),
                                     torch.maximum(u32.max(), v32.max()).long()]).tolist()
    if lo < 0 or hi >= n_nodes:
        raise ValueError(f"edge endpoint out of range: ids span [{int(lo)}, {int(hi)}] but the graph has {int(n_nodes)} nodes")
    T, N = int(T), int(n_nodes)
    if lg_edges is not None and (T == 0 or int(lg_edges[0].numel()) != T or int(lg_edges[1].numel()) != T):
        return None
    i32, i64, f32 = torch.int32, torch.int64, torch.float32
    want = [("seg_ptr", i32, N + 1), ("src", i32, E), ("dst", i32, E), ("out_ptr", i32, N + 1), ("out_slot", i32, E),
            ("perm", i64, E), ("inv", i64, E), ("r", f32, 3 * E if r is not None else 0), ("lg_seg_ptr", i32, E + 1),
            ("lg_src", i32, T), ("lg_dst", i32, T), ("lg_out_ptr", i32, E + 1), ("lg_out_slot", i32, T), ("seg_rank", i32, T),
            ("ident", i64, T if lg_edges is None else 0), ("out_rank", i32, E if lg_edges is not None else 0)]
    offs, off = {}, 0
    for name, dt, n in want:
        offs[name] = off
        off = (off + n * (8 if dt is i64 else 4) + 63) // 64 * 64
    ws_bytes = lib.alignn_stage_batch_workspace(N, E)
    out = torch.empty(off + ws_bytes, dtype=torch.uint8, device=dev)
    base = out.data_ptr()
    a = {name: (out[offs[name]:offs[name] + n * (8 if dt is i64 else 4)].view(dt) if n else None) for name, dt, n in want}
    rr = None if r is None else r.to(f32).contiguous()
    with _lib.device_guard(out):
        _lib.check(lib.alignn_stage_batch(
            u32.data_ptr(), v32.data_ptr(), None if rr is None else rr.data_ptr(), N, E, T,
            base + offs["seg_ptr"], base + offs["src"], base + offs["dst"], base + offs["out_ptr"], base + offs["out_slot"],
            base + offs["perm"], base + offs["inv"], (base + offs["r"]) if rr is not None else None, base + offs["lg_seg_ptr"],
            base + offs["lg_src"], base + offs["lg_dst"], base + offs["lg_out_ptr"], base + offs["lg_out_slot"],
            base + offs["seg_rank"], (base + offs["ident"]) if lg_edges is None else None, None,
            (base + offs["out_rank"]) if lg_edges is not None else None, base + off, ws_bytes, _lib.stream()), "stage_batch")
        perm_lg = inv_lg = None
        if lg_edges is not None:
            lu, lv = (torch.as_tensor(x).to(dev).to(i64).contiguous() for x in lg_edges)
            perm_lg = torch.full((T,), -1, dtype=i64, device=dev)
            inv_lg = torch.empty(T, dtype=i64, device=dev)
            bad = torch.zeros(1, dtype=i32, device=dev)
            _lib.check(lib.alignn_map_line_graph_rows(
                lu.data_ptr(), lv.data_ptr(), base + offs["inv"], base + offs["seg_ptr"], base + offs["src"], base + offs["dst"],
                base + offs["out_rank"], base + offs["lg_seg_ptr"], E, T, perm_lg.data_ptr(), inv_lg.data_ptr(), bad.data_ptr(),
                _lib.stream()), "map_line_graph_rows")
            if int(bad) != 0:
                return None
    g = CSRGraph(n_nodes=N, n_edges=E, seg_ptr=a["seg_ptr"], seg_node=None, src=a["src"], dst=a["dst"], out_ptr=a["out_ptr"],
                 out_slot=a["out_slot"], perm=a["perm"], inv=a["inv"])
    ident = perm_lg if perm_lg is not None else (a["ident"] if T else torch.empty(0, dtype=i64, device=dev))
    z32 = torch.empty(0, dtype=i32, device=dev)
    lg = CSRGraph(n_nodes=E, n_edges=T, seg_ptr=a["lg_seg_ptr"], seg_node=a["out_slot"], src=a["lg_src"] if T else z32,
                  dst=a["lg_dst"] if T else z32, out_ptr=a["lg_out_ptr"], out_slot=a["lg_out_slot"] if T else z32, perm=ident,
                  inv=inv_lg if inv_lg is not None else ident, grp_seg_ptr=a["out_ptr"], grp_src_ptr=a["seg_ptr"], dense_max_src=int(max_in),
                  seg_rank=a["seg_rank"] if T else z32)
    return g, lg, (None if r is None else a["r"].view(E, 3))


def _dense_blocks(g: CSRGraph, lg: CSRGraph) -> int:
    """Largest source count of any block if L(g)'s blocks are dense and source-sorted (CSRGraph.dense_max_src), else 0.

    Segment s (bond e2 = seg_node[s], centre atom j = src(e2)) must list the in-edges of j, i.e. the L(g) nodes
    [g.seg_ptr[j], g.seg_ptr[j+1]), in ascending order, all of them except e2 itself (present only for self-images)."""
    if lg.n_edges == 0:
        return 0
    dev = lg.src.device
    sp = lg.seg_ptr.long()
    seglen = sp[1:] - sp[:-1]
    e2 = lg.seg_node.long() if lg.seg_node is not None else torch.arange(lg.n_nodes, device=dev)
    atom = g.src.long()[e2]
    gsp = g.seg_ptr.long()
    p_beg, n_src = gsp[atom], gsp[atom + 1] - gsp[atom]
    self_q = e2 - p_beg
    has_self = (self_q >= 0) & (self_q < n_src)
    if not bool((seglen == n_src - has_self.long()).all()):
        return 0
    seg_of = torch.repeat_interleave(torch.arange(lg.n_nodes, device=dev), seglen)
    r = torch.arange(lg.n_edges, device=dev) - sp[:-1][seg_of]
    q = r + (has_self[seg_of] & (r >= self_q[seg_of])).long()
    if not bool((lg.src.long() == p_beg[seg_of] + q).all()):
        return 0
    return int(n_src.max())


@dataclass
class GraphBatch:
    """Canonical (g, L(g)) batch plus the canonically ordered inputs."""

    g: CSRGraph
    lg: Optional[CSRGraph]
    graph_ptr: torch.Tensor  # int32 [B+1] node offsets per crystal
    batch_size: int
    atom_features: Optional[torch.Tensor] = None  # [N, F]
    r: Optional[torch.Tensor] = None  # [E, 3] canonical g-slot order
    h: Optional[torch.Tensor] = None  # [T]    canonical lg-slot order
    volume: Optional[torch.Tensor] = None  # [B] cell volumes (g.ndata["V"] of each crystal's first atom)
    extra_features: Optional[torch.Tensor] = None  # [N, k] g.ndata["extra_features"] (ALIGNNConfig.extra_features != 0)
    r_from_positions: Optional[torch.Tensor] = None  # [E, 3] bond vectors recomputed from frac_coords + images (batch_stress=False)
    cache: dict = field(default_factory=dict)  # derived index structures (built once per batch, outside graph capture)

    @property
    def edge_graph_ptr(self) -> torch.Tensor:
        """int32 [B+1]: bond-slot offsets per crystal (bonds are sorted by destination atom, atoms by crystal)."""
        return self.g.seg_ptr[self.graph_ptr.long()]

    @property
    def device(self):
        return self.graph_ptr.device

    def cell_volumes(self) -> torch.Tensor:
        """``volume`` as the kernels take it by raw pointer: float32, contiguous, on the batch's device, ONE value per crystal.
        Anything else a caller may have put there (float64, a CPU tensor, a [B, 1] view) is converted once and cached; a
        tensor of another length (e.g. the per-atom g.ndata["V"]) is an error, not garbage stresses."""
        v = self.volume
        if v is None:
            raise ValueError("stress needs the cell volumes: g.ndata['V'] (or GraphBatch.volume)")
        if v.numel() != self.batch_size:
            raise ValueError(f"GraphBatch.volume holds {v.numel()} values for {self.batch_size} crystals: one cell volume per crystal is needed")
        dev = self.graph_ptr.device
        if v.dtype == torch.float32 and v.device == dev and v.is_contiguous() and v.dim() == 1 and not v.requires_grad:
            return v
        key = ("volume_f32", v.data_ptr(), v._version)
        hit = self.cache.get("volume_f32")
        if hit is None or hit[0] != key:
            hit = (key, v.detach().to(device=dev, dtype=torch.float32).reshape(-1).contiguous())
            self.cache["volume_f32"] = hit
        return hit[1]

    def topology_only(self) -> "GraphBatch":
        """The index structures without any feature tensor (what may be cached on a caller's graph object: features
        are re-read on every forward, like the reference does, so in-place edits of ``g.edata['r']`` etc. are seen)."""
        return GraphBatch(g=self.g, lg=self.lg, graph_ptr=self.graph_ptr, batch_size=self.batch_size)

    @staticmethod
    def from_coo(u, v, n_nodes, batch_num_nodes, lg_u=None, lg_v=None, atom_features=None, r=None, h=None, device=None,
                 volume=None, build_line_graph=False, topology: Optional["GraphBatch"] = None):
        """Build from raw COO tensors (caller's edge order).  ``build_line_graph``: no ``lg_u/lg_v`` given - derive
        L(g) on the device (``line_graph_of``); the cosines ``h`` are then left to the model (``lg_on_fly``).
        ``topology``: a batch built earlier from the SAME graphs - its index structures are reused and only the
        features are gathered into canonical order."""
        dev = torch.device(device) if device is not None else u.device
        if topology is not None:
            g, lg = topology.g, topology.lg
            out = GraphBatch(g=g, lg=lg, graph_ptr=topology.graph_ptr, batch_size=topology.batch_size, cache=topology.cache)
            return GraphBatch._attach(out, dev, atom_features, r, h, volume)
        u = torch.as_tensor(u).to(dev)
        v = torch.as_tensor(v).to(dev)
        staged = None
        if lg_u is not None:  # the caller's own L(g): one device pass if it is the line graph of g, else the generic builder
            staged = csr_and_line_graph(u, v, int(n_nodes), lg_edges=(lg_u, lg_v))
        if staged is not None:
            g, lg, _ = staged
        elif lg_u is None and build_line_graph:
            g, lg, _ = csr_and_line_graph(u, v, int(n_nodes))
        else:
            g = build_csr(u, v, int(n_nodes))
            lg = None
        if lg_u is not None and staged is None:
            e1 = g.inv[torch.as_tensor(lg_u).to(dev).to(torch.int64)]
            e2 = g.inv[torch.as_tensor(lg_v).to(dev).to(torch.int64)]
            m = g.n_edges
            # order L(g)'s segments (bonds e2) by the bond's source atom, then by bond id
            seg_order = torch.argsort(g.src.to(torch.int64) * m + torch.arange(m, device=dev), stable=True)
            lg = build_csr(e1, e2, m, seg_order, sort_sources=True)
            # every edge of a true line graph joins an in-edge and an out-edge of the same atom; then the canonical
            # layout is block structured (see CSRGraph.grp_*) and the fused backward kernel applies.  A filtered
            # line graph (eALIGNN) would fail this check and simply use the generic two-pass backward.
            if bool((g.dst[lg.src.long()] == g.src[lg.dst.long()]).all()):
                lg.grp_seg_ptr, lg.grp_src_ptr = g.out_ptr, g.seg_ptr
                lg.dense_max_src = _dense_blocks(g, lg)
            lg.segment_rank()
        bnn = torch.as_tensor(batch_num_nodes).to(dev).to(torch.int64)
        gp = _ptr_from_counts(bnn).to(torch.int32)
        out = GraphBatch(g=g, lg=lg, graph_ptr=gp, batch_size=int(bnn.numel()))
        return GraphBatch._attach(out, dev, atom_features, r, h, volume)

    @staticmethod
    def _attach(out, dev, atom_features, r, h, volume):
        g, lg = out.g, out.lg
        if atom_features is not None:
            out.atom_features = torch.as_tensor(atom_features).to(dev).contiguous()
        if r is not None:
            out.r = torch.as_tensor(r).to(dev)[g.perm].contiguous()
        if h is not None and lg is not None:
            out.h = torch.as_tensor(h).to(dev)[lg.perm].contiguous()
        if volume is not None:
            out.volume = torch.as_tensor(volume).to(dev).to(torch.float32).contiguous()
        return out

    @staticmethod
    def from_raw(raw, device=None, dtype=torch.float32):
        """From an ``alignn_amd.synthetic.RawGraph`` (numpy)."""
        t = torch.from_numpy
        return GraphBatch.from_coo(
            t(raw.u),
            t(raw.v),
            raw.num_nodes,
            t(raw.batch_num_nodes),
            t(raw.lg_u),
            t(raw.lg_v),
            t(raw.atom_features).to(dtype),
            t(raw.r).to(dtype),
            t(raw.h).to(dtype),
            device=device,
            volume=torch.linalg.det(t(raw.lattice).double()).abs().float(),
        )

    @staticmethod
    def from_dgl(g, lg=None, device=None, build_line_graph=False, topology: Optional["GraphBatch"] = None):
        """From DGL-like graphs: anything exposing ``edges()``, ``num_nodes()``, ``batch_num_nodes()``
        and ``ndata`` / ``edata`` dicts (a real ``dgl.DGLGraph`` or the oracle shim).  Reads
        ``g.ndata['atom_features']``, ``g.edata['r']``, ``lg.edata['h']`` exactly as
        ``ALIGNN.forward`` does (alignn/models/alignn.py:298,307,313) without popping them.  ``topology``: see
        ``from_coo`` (the features are read afresh, the index structures reused)."""
        u, v = (None, None) if topology is not None else g.edges()
        dev = torch.device(device) if device is not None else (topology.device if topology is not None else u.device)
        kw = {}
        if lg is not None:
            if topology is None:
                kw["lg_u"], kw["lg_v"] = lg.edges()
            if "h" in lg.edata:
                kw["h"] = lg.edata["h"]
        if "atom_features" in g.ndata:
            kw["atom_features"] = g.ndata["atom_features"]
        extra = g.ndata["extra_features"] if "extra_features" in g.ndata else None
        if "r" in g.edata:
            kw["r"] = g.edata["r"]
        if "V" in g.ndata:  # per-atom copy of the cell volume (alignn/graphs.py:553); take each crystal's first atom
            bnn = torch.as_tensor(g.batch_num_nodes()).to(torch.int64)
            first = torch.cumsum(bnn, 0) - bnn
            kw["volume"] = g.ndata["V"][first.to(g.ndata["V"].device)]
        out = GraphBatch.from_coo(u, v, g.num_nodes(), g.batch_num_nodes(), device=dev,
                                  build_line_graph=build_line_graph and lg is None, topology=topology, **kw)
        if extra is not None:
            out.extra_features = torch.as_tensor(extra).to(dev).to(torch.float32).contiguous()
        return out


def cached_dgl_batch(gg, lg, dev, build_line_graph=False) -> GraphBatch:
    """``GraphBatch.from_dgl`` with the TOPOLOGY (CSR structures, permutations) cached on the caller's graph object -
    the training loop passes the same DGL graphs every epoch - and the features gathered afresh on every call.  The
    cache is keyed on the device, the line graph object that was passed and the node / edge counts."""
    ent = getattr(gg, "_alignn_amd_topology", None)
    topo = None
    if ent is not None:
        t, lg_ref, built = ent
        same_lg = (lg_ref is None and lg is None) or (lg_ref is not None and lg_ref() is lg)
        if (t.device == dev and same_lg and (built or not build_line_graph) and t.g.n_nodes == gg.num_nodes()
                and t.g.n_edges == gg.num_edges()):
            topo = t
    batch = GraphBatch.from_dgl(gg, lg, device=dev, build_line_graph=build_line_graph, topology=topo)
    if topo is None:
        import weakref

        try:
            lg_ref = None if lg is None else weakref.ref(lg)
        except TypeError:  # an object that cannot be weakly referenced: hold it (g and lg travel together anyway)
            lg_ref = (lambda o: (lambda: o))(lg)
        try:
            gg._alignn_amd_topology = (batch.topology_only(), lg_ref, bool(build_line_graph and lg is None))
        except Exception:
            pass
    return batch
