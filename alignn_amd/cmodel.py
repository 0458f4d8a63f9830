"""Whole-model launch path: ``ALIGNN.forward`` / its backward as ONE C call each (``csrc/model.hip``).

The reference's loop hands the model a new ``(g, lg)`` every iteration (``alignn/train.py:258-270``); a step on a batch
nobody has seen before cannot be replayed from a hipGraph, so its ~350 kernels are enqueued again - and when
``alignn_amd.ops`` sequences them from Python (95 autograd nodes, ~100 torch glue calls, an allocation per tensor) the host
needs 9-23 ms for what the GPU finishes in 15.8 ms.  ``alignn_model_fwd`` / ``alignn_model_bwd`` issue the SAME launches
(same kernels, same arguments: bit-identical parameters gradients, tests/test_gpu_cmodel.py) from C over one workspace
block planned from (N, E, T); Python keeps the module tree (``state_dict`` keys are the reference's), ONE
``torch.autograd.Function`` whose inputs are the parameters, and the loss.

Used by ``ALIGNN.forward`` whenever it applies (``applicable``): BatchNorm flavour in training mode on float32 HIP tensors,
every parameter trainable, every kernel-choice switch of ``ops`` at its default (a test that flips one gets the
per-operator path it wants to compare).  Anything else - eval mode, frozen parameters, the LayerNorm models, extra
features, kernel choices the C side does not carry (``hipErrorNotSupported``) - takes the per-operator path as before.
"""

from __future__ import annotations

import ctypes as C
import os
import time
import weakref

import torch

from . import _lib, ops

ENABLED = os.environ.get("ALIGNN_AMD_CMODEL", "1") != "0"
# a convolution's edge input gradient written over its own dead gate pre-activation (csrc/model.hip edge_grad_buffer): 2.8 GB of
# the 19.3 GB workspace at the benchmark batch.  The tape then does not survive the backward: a SECOND backward of a retained
# graph raises (set cmodel.REUSE_TAPE = False for loss.backward(retain_graph=True) followed by another backward).
REUSE_TAPE = os.environ.get("ALIGNN_AMD_REUSE_TAPE", "1") != "0"
GRAD_SINK = os.environ.get("ALIGNN_AMD_GRAD_SINK", "1") != "0"  # backward writes into FlatAdamW's packed gradient buffer
STATS = {"fwd": 0, "bwd": 0, "plans": 0, "rebuilds": 0, "arena_bytes": 0}
_NOT_SUPPORTED = 801  # hipErrorNotSupported
TIMING = None  # tools/host_profile_c.py: {"cfwd": s, "cbwd": s, "bwd_py": s} accumulated host seconds

_p, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class MlpParams(C.Structure):
    _fields_ = [(n, _p) for n in ("W", "b", "gamma", "beta", "rm", "rv", "gW", "gb", "red", "img", "img_t", "w_amax")] + [
        ("in_", _i32), ("out", _i32)]


class ConvParams(C.Structure):
    _fields_ = [(n, _p) for n in (
        "wcat", "bcat", "w_eg", "b_eg", "n_gamma", "n_beta", "e_gamma", "e_beta", "n_rm", "n_rv", "e_rm", "e_rv",
        "g_wcat", "g_bcat", "g_weg", "g_beg", "n_red", "e_red", "wcat_img", "wcat_img_t", "weg_img", "weg_img_t",
        "wcat_amax", "weg_amax")]


class GraphCSR(C.Structure):
    _fields_ = [(n, _p) for n in ("seg_ptr", "seg_node", "src", "dst", "out_ptr", "out_slot", "grp_seg_ptr", "grp_src_ptr",
                                  "seg_rank")] + [("n", _i64), ("m", _i64), ("n_groups", _i64), ("dense_max_src", _i32),
                                                  ("pad_", _i32)]


class ModelBatch(C.Structure):
    _fields_ = [("g", GraphCSR), ("lg", GraphCSR), ("graph_ptr", _p), ("atom_features", _p), ("r", _p), ("h", _p),
                ("B", _i32), ("pad_", _i32)]


class ModelDesc(C.Structure):
    _fields_ = [(n, _i32) for n in ("alignn_layers", "gcn_layers", "H", "out_features", "atom_in", "edge_bins", "angle_bins",
                                    "embed")] + [
        ("edge_gamma", _f32), ("angle_gamma", _f32), ("eps", _f32), ("momentum", _f32),
        ("edge_centers", _p), ("angle_centers", _p),
        ("atom", MlpParams), ("edge1", MlpParams), ("edge2", MlpParams), ("angle1", MlpParams), ("angle2", MlpParams),
        ("convs", C.POINTER(ConvParams)),
        ("fc_W", _p), ("fc_b", _p), ("g_fc_W", _p), ("g_fc_b", _p),
        ("weight_descs", _p), ("weight_amax", _p), ("bump_ptrs", _p),
        ("n_weights", _i32), ("n_bump", _i32), ("x6_min_tiles", _i32), ("bd_segment_table", _i32),
        ("angle_fused", _i32), ("norm", _i32), ("reuse_tape", _i32), ("dw_fused", _i32),
        ("amax_min_rows", _i64), ("lane_min_rows", _i64), ("side_min_rows", _i64),
        ("lane_T", _p), ("side", _p), ("aux", _p)]


class FFDesc(C.Structure):
    """alignn_ff_desc: the force / stress head's switches (ALIGNNAtomWiseConfig) + the cell volumes of the batch"""
    _fields_ = [(n, _i32) for n in ("lg_on_fly", "add_reverse_forces", "force_mult_natoms", "energy_mult_natoms", "has_stress",
                                    "use_penalty", "dense_lg_reverse", "pad_")] + [
        (n, _f32) for n in ("grad_multiplier", "stress_multiplier", "penalty_factor", "penalty_threshold")] + [("volume", _p)]


_SIGS_DONE = False

# Per-model state (the Binding with its ctypes blocks and helper streams, the cached module walks, the weight-image
# preparation) lives HERE, keyed weakly by the model - not in ``model.__dict__``: ctypes structs with pointer fields and
# stream objects cannot be pickled, and ``copy.deepcopy(model)`` / ``torch.save(model)`` / ``pickle.dumps(model)`` (EMA / SWA
# copies, best-model snapshots) copy ``__dict__``.  Nothing in a value may hold the model strongly.
_PER_MODEL = weakref.WeakKeyDictionary()


def model_cache(model) -> dict:
    c = _PER_MODEL.get(model)
    if c is None:
        c = _PER_MODEL[model] = {}
    return c



def _lib_model():
    global _SIGS_DONE
    lib = _lib.load()
    if not _SIGS_DONE:
        for which, st in enumerate((MlpParams, ConvParams, GraphCSR, ModelBatch, ModelDesc)):
            if lib.alignn_model_sizeof(which) != C.sizeof(st):
                raise RuntimeError(f"alignn_model struct {which}: library says {lib.alignn_model_sizeof(which)} bytes, "
                                   f"the binding lays out {C.sizeof(st)}")
        if lib.alignn_ff_desc_sizeof() != C.sizeof(FFDesc):
            raise RuntimeError("alignn_ff_desc: the binding's layout differs from the library's")
        _SIGS_DONE = True
    return lib


def _flags_default() -> bool:
    """Every kernel-choice switch of the per-operator path at the value the C side hard-wires."""
    o = ops
    return (o.F16X3 and o.GATHER_FUSED and o.STATS_FUSED and o.BNRED_FUSED and o.SPLIT_BOTH and o.BATCHED_WEIGHT_PREP
            and o.NN_SPLIT and o.APPLY_SUM and o.FUSED_LG_BACKWARD and o.DENSE_LG_BACKWARD and o.COMPOSITE
            and o.KERNEL_TIMER is None and o.FOLD_ABOVE == 1024 and o._PARAM_GRADS["on"])


class disabled:
    """``with cmodel.disabled():`` - the per-operator path (tests that count its registries, tools that time it)."""

    def __enter__(self):
        global ENABLED
        self.prev, ENABLED = ENABLED, False

    def __exit__(self, *exc):
        global ENABLED
        ENABLED = self.prev
        return False


def _ptr(t):
    return None if t is None else t.data_ptr()


class Binding:
    """Everything about one model on one device that does not change from batch to batch: the parameter blocks of the C
    description, the gradient layout, the helper streams, the workspace."""

    def __init__(self, model):
        from .alignn import EdgeGatedGraphConv, MLPLayer

        self._model = weakref.ref(model)
        cfg = model.config
        dev = model.fc.weight.device
        self.device = dev
        self.convs = []
        for layer in model.alignn_layers:
            self.convs += [layer.node_update, layer.edge_update]
        self.convs += list(model.gcn_layers)
        self.mlps = [model.atom_embedding, model.edge_embedding[1], model.edge_embedding[2], model.angle_embedding[1],
                     model.angle_embedding[2]]
        self.norm = 1 if getattr(self.convs[0], "_norm", "batch") == "layer" else 0  # alignn_model_desc.norm
        self.norms = [] if self.norm else ([mod.layer[1] for mod in self.mlps] +
                                           [bn for cv in self.convs for bn in (cv.bn_nodes, cv.bn_edges)])
        self.dead_edge = {2 * cfg.alignn_layers - 1, len(self.convs) - 1}  # convs whose edge output nobody reads
        assert all(isinstance(m, EdgeGatedGraphConv) for m in self.convs) and all(isinstance(m, MLPLayer) for m in self.mlps)
        self.desc = ModelDesc()
        self.conv_arr = (ConvParams * len(self.convs))()
        self.desc.convs = C.cast(self.conv_arr, C.POINTER(ConvParams))
        with torch.cuda.device(dev):
            _lib.check(_lib_model().alignn_model_init(), "model_init")
            self.streams = [torch.cuda.Stream(device=dev) for _ in range(3)]  # lane T, side, aux
        self.sig = self.psig = self.prep_sig = None
        self.desc_addr = C.addressof(self.desc)
        self.arena = None
        self.arena_busy = False
        self.arena_gen = 0  # bumped by every forward that takes the shared block (a retained graph's second backward checks it)
        self.arena_stream = None
        self.pinned = []
        self.spare = None  # the unpinned block of eager steps while the shared one belongs to captured graphs
        self.plans = {}
        self.bump = None
        self._sink_key = self._sink_plan = None
        self._layout()

    @property
    def model(self):
        return self._model()

    # ---- gradient layout: one flat buffer per backward, every parameter's gradient a view of it
    def _layout(self):
        m = self.model
        entries = []  # (parameter or None (padding), floats)
        self.grad_fields = []  # (struct, field name, float offset)
        off = 0

        self.grad_blocks = []  # per grad field: the parameters whose gradients form that contiguous block, in order

        def place(owner, field, params):
            nonlocal off
            pad = -off % 64
            if pad:
                entries.append((None, pad))
                off += pad
            self.grad_fields.append((owner, field, off))
            self.grad_blocks.append(list(params))
            for p in params:
                entries.append((p, p.numel()))
                off += p.numel()

        d = self.desc
        for blk, mod in zip((d.atom, d.edge1, d.edge2, d.angle1, d.angle2), self.mlps):
            lin, bn = mod.layer[0], mod.layer[1]
            place(blk, "gW", [lin.weight])
            place(blk, "gb", [lin.bias])
            place(blk, "red", [bn.bias, bn.weight])
        for i, cv in enumerate(self.convs):
            blk = self.conv_arr[i]
            ws, bs = cv._fused_parameter_groups()
            place(blk, "g_wcat", ws)
            place(blk, "g_bcat", bs)
            place(blk, "g_weg", [cv.edge_gate.weight])
            place(blk, "g_beg", [cv.edge_gate.bias])
            place(blk, "n_red", [cv.bn_nodes.bias, cv.bn_nodes.weight])
            place(blk, "e_red", [cv.bn_edges.bias, cv.bn_edges.weight])
        place(d, "g_fc_W", [m.fc.weight])
        place(d, "g_fc_b", [m.fc.bias])
        pad = -off % 4  # (alignn_ff_grad adds the tangent halves with float4 accesses)
        if pad:
            entries.append((None, pad))
            off += pad
        self.grad_floats = off
        self.params = [p for p, _ in entries if p is not None]
        self.sizes = [n for _, n in entries]
        self.keep = [i for i, (p, _) in enumerate(entries) if p is not None]
        self.shapes = [tuple(p.shape) for p in self.params]
        dead = set()
        for i in self.dead_edge:
            dead.update((id(self.convs[i].bn_edges.bias), id(self.convs[i].bn_edges.weight)))
        self.no_grad = [id(p) in dead for p in self.params]

    # ---- the optimizer's packed gradient buffer as the destination of the backward (alignn_amd.optim.FlatAdamW registers
    # itself as the model's gradient sink): every block whose parameters sit adjacent and in order in that buffer is
    # written THERE - no 124-view gather in step(), and the data-parallel all-reduce runs on the buffer the backward wrote
    def sink(self):
        """The optimizer registered as this model's gradient sink (alignn_amd.optim.FlatAdamW), or None"""
        ref = model_cache(self.model).get("grad_sink")
        return ref() if ref is not None else None

    def sink_plan(self):
        """-> (per grad field: device address inside the sink's buffer or None, per parameter: its slot view or None) or None.
        None also while an earlier backward of this step already wrote into the buffer (``sink.sink_in_flight``, cleared by the
        optimizer's ``step()`` / ``zero_grad()``): a second autograd node of the same graph - ``(l(model(b1)) + l(model(b2)))
        .backward()`` - runs before AccumulateGrad has filled any ``p.grad``, would find them all None and overwrite the first
        node's gradients in place; it takes its private buffer instead and autograd adds the two."""
        sink = self.sink()
        if sink is None or getattr(sink, "_inner", None) is None or getattr(sink, "sink_in_flight", False):
            return None
        key = (id(sink), id(sink._grad_all), sink._grad_all.data_ptr())
        if self._sink_key != key:
            fields, views = [], {}
            for params in self.grad_blocks:
                slots = [sink.gradient_slot(p) for p in params]
                ok = all(sl is not None and sl.is_contiguous() and sl.device == self.device for sl in slots)
                if ok:
                    for a, b, pa in zip(slots, slots[1:], params):
                        ok = ok and b.data_ptr() == a.data_ptr() + 4 * pa.numel()
                if ok:
                    fields.append(slots[0].data_ptr())
                    for p_, sl in zip(params, slots):
                        views[id(p_)] = sl
                else:
                    fields.append(None)
            self._sink_key, self._sink_plan = key, (fields, [views.get(id(p_)) for p_ in self.params])
        return self._sink_plan

    def param_sig(self):
        """Where every parameter AND every BatchNorm buffer the C side writes lives (a reassigned ``running_mean`` /
        ``running_var`` / ``num_batches_tracked`` must not leave the kernels a stale pointer)."""
        sig = [p.data_ptr() for p in self.params]
        for bn in self.norms:
            sig += [bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr()]
        return tuple(sig)

    # ---- parameter blocks (rebuilt when a parameter moved: .to(), FlatAdamW re-homing, load_state_dict(assign=True))
    def refresh(self):
        m, d = self.model, self.desc
        mc = model_cache(m)
        prep = mc.get("weight_prep")
        if prep is None:
            prep = mc["weight_prep"] = ops.WeightPrep()
        psig = self.param_sig()
        if psig == self.psig and prep.sig is not None and prep.sig == self.prep_sig:
            return  # nothing moved since the blocks were filled (the fused buffers alias the parameters)
        # fused node projections first: asking for them may re-home the four Linear parameters of a convolution
        fused = [cv._fused_node_projection() for cv in self.convs]
        weights = []
        for (wcat, _b), cv in zip(fused, self.convs):
            weights += [wcat, cv.edge_gate.weight]
        mlp_w = [mod.layer[0].weight for mod in self.mlps if (mod.layer[0].weight.shape[0] >= 128
                                                              and mod.layer[0].weight.shape[0] % 16 == 0
                                                              and mod.layer[0].weight.shape[1] % 16 == 0)]
        weights += mlp_w
        have_images = prep.ensure(weights)
        self.psig = self.param_sig()  # (after any re-homing above)
        self.prep_sig = prep.sig if have_images else None
        sig = (self.psig, tuple(w.data_ptr() for w in weights), have_images and prep.desc.data_ptr())
        if sig == self.sig:
            return
        STATS["rebuilds"] += 1
        self.sig = sig
        self.weights = weights
        img = {}
        if have_images:
            for w, (sw, sw_t) in zip(weights, prep.images):
                img[id(w)] = (sw.buf.data_ptr(), sw_t.buf.data_ptr(), sw.amax.data_ptr())
        cfg = m.config
        d.alignn_layers, d.gcn_layers, d.H, d.out_features = cfg.alignn_layers, cfg.gcn_layers, cfg.hidden_features, m.fc.weight.shape[0]
        d.atom_in, d.edge_bins, d.angle_bins, d.embed = (cfg.atom_input_features, cfg.edge_input_features,
                                                         cfg.triplet_input_features, cfg.embedding_features)
        rbf_e, rbf_a = m.edge_embedding[0], m.angle_embedding[0]
        d.edge_gamma, d.angle_gamma, d.eps, d.momentum = rbf_e.gamma, rbf_a.gamma, ops.BN_EPS, ops.BN_MOMENTUM
        d.edge_centers, d.angle_centers = rbf_e.centers.data_ptr(), rbf_a.centers.data_ptr()
        bumps = []
        for blk, mod in zip((d.atom, d.edge1, d.edge2, d.angle1, d.angle2), self.mlps):
            lin, bn = mod.layer[0], mod.layer[1]
            blk.W, blk.b, blk.gamma, blk.beta = lin.weight.data_ptr(), lin.bias.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr()
            blk.rm, blk.rv = (None, None) if self.norm else (bn.running_mean.data_ptr(), bn.running_var.data_ptr())
            i3 = img.get(id(lin.weight), (None, None, None))
            blk.img, blk.img_t, blk.w_amax = i3
            blk.in_, blk.out = lin.weight.shape[1], lin.weight.shape[0]
            if not self.norm:
                bumps.append(bn.num_batches_tracked)
        for i, (cv, (wcat, bcat)) in enumerate(zip(self.convs, fused)):
            blk = self.conv_arr[i]
            blk.wcat, blk.bcat = wcat.data_ptr(), bcat.data_ptr()
            blk.w_eg, blk.b_eg = cv.edge_gate.weight.data_ptr(), cv.edge_gate.bias.data_ptr()
            blk.n_gamma, blk.n_beta = cv.bn_nodes.weight.data_ptr(), cv.bn_nodes.bias.data_ptr()
            blk.e_gamma, blk.e_beta = cv.bn_edges.weight.data_ptr(), cv.bn_edges.bias.data_ptr()
            if not self.norm:
                blk.n_rm, blk.n_rv = cv.bn_nodes.running_mean.data_ptr(), cv.bn_nodes.running_var.data_ptr()
                blk.e_rm, blk.e_rv = cv.bn_edges.running_mean.data_ptr(), cv.bn_edges.running_var.data_ptr()
            blk.wcat_img, blk.wcat_img_t, blk.wcat_amax = img.get(id(wcat), (None, None, None))
            blk.weg_img, blk.weg_img_t, blk.weg_amax = img.get(id(cv.edge_gate.weight), (None, None, None))
            if not self.norm:
                bumps += [cv.bn_nodes.num_batches_tracked, cv.bn_edges.num_batches_tracked]
        d.fc_W, d.fc_b = m.fc.weight.data_ptr(), m.fc.bias.data_ptr()
        if have_images:
            d.weight_descs, d.weight_amax, d.n_weights = prep.desc.data_ptr(), prep.amax.data_ptr(), len(weights)
        else:
            d.weight_descs, d.weight_amax, d.n_weights = None, None, 0
        self.bump_keep = bumps
        self.bump = torch.tensor([t.data_ptr() for t in bumps], dtype=torch.int64).to(self.device) if bumps else None
        d.bump_ptrs, d.n_bump = (self.bump.data_ptr() if bumps else None), len(bumps)
        d.norm = self.norm
        d.x6_min_tiles, d.bd_segment_table = ops.X6_MIN_TILES, int(ops.BD_SEGMENT_TABLE)
        d.amax_min_rows = ops.AMAX_MIN_ROWS if ops.F16X3 else (1 << 62)
        self.plans = {}

    def set_mode(self):
        """Per-call stream choices (mirror ops.lanes / on_side_stream / FORK_DGRAD: the helper streams cost the host a few
        microseconds per event here, not ~80, so outside a capture they stay ON unless switched off)."""
        d = self.desc
        capturing = torch.cuda.is_current_stream_capturing()
        lanes = ops._LANE["enabled"] not in ("0", False)
        d.lane_T = self.streams[0].cuda_stream if lanes else None
        d.lane_min_rows = ops._LANE["min_rows"]
        d.side = self.streams[1].cuda_stream if ops._SIDE["enabled"] else None
        d.side_min_rows = 0 if capturing else ops._SIDE["min_rows"]
        d.aux = self.streams[2].cuda_stream if ops.FORK_DGRAD != "0" else None
        d.angle_fused = int(ops.ANGLE_FUSED)
        d.reuse_tape = int(REUSE_TAPE)
        d.dw_fused = int(ops.DW_MIN_ROWS) if ops.DGRAD_WGRAD_FUSED else 0
        return capturing

    def batch_struct(self, b):
        mb = ModelBatch()
        for dst, g in ((mb.g, b.g), (mb.lg, b.lg)):
            dst.seg_ptr, dst.seg_node, dst.src, dst.dst = _ptr(g.seg_ptr), _ptr(g.seg_node), _ptr(g.src), _ptr(g.dst)
            dst.out_ptr, dst.out_slot = _ptr(g.out_ptr), _ptr(g.out_slot)
            dst.grp_seg_ptr, dst.grp_src_ptr, dst.seg_rank = _ptr(g.grp_seg_ptr), _ptr(g.grp_src_ptr), _ptr(g.seg_rank)
            dst.n, dst.m = g.n_nodes, g.n_edges
            dst.n_groups = (g.grp_seg_ptr.numel() - 1) if g.grp_seg_ptr is not None else 0
            dst.dense_max_src = int(g.dense_max_src)
        mb.graph_ptr, mb.B = b.graph_ptr.data_ptr(), b.batch_size
        return mb

    def plan(self, mb):
        key = (mb.g.n, mb.g.m, mb.lg.m, mb.B, mb.lg.dense_max_src, bool(mb.lg.grp_seg_ptr), bool(mb.lg.seg_rank),
               bool(mb.g.seg_node), bool(self.desc.lane_T), bool(self.desc.side), bool(self.desc.aux),
               self.desc.side_min_rows, self.desc.lane_min_rows, self.desc.angle_fused, self.desc.reuse_tape, self.desc.dw_fused)
        hit = self.plans.get(key)
        if hit is None:
            fwd, tot = C.c_size_t(0), C.c_size_t(0)
            rc = _lib_model().alignn_model_plan(self.desc_addr, C.addressof(mb), C.addressof(fwd), C.addressof(tot))
            STATS["plans"] += 1
            if rc == _NOT_SUPPORTED:
                hit = (None, None)
            else:
                _lib.check(rc, "model_plan")
                hit = (fwd.value, tot.value)
            self.plans[key] = hit
        return hit

    def take_arena(self, nbytes, capturing, need_backward):
        """The workspace of one forward (+ backward) -> (block, owns the shared block, generation).  Eagerly launched steps
        share ONE grow-only block (a fresh 15 GB allocation per step would fragment the caching allocator when no two
        batches are alike); a forward that finds it still held by an earlier one whose backward has not run, and every
        forward inside a stream capture whose block would have to grow (it then has to belong to the graph's own memory
        pool), allocates its own.  A block a captured graph replays into is never leased to an eager step that needs a
        backward (a replay between that forward and its backward would overwrite the tape): such steps get another shared
        block.  Forwards issued from different streams are ordered on the block by a stream wait."""
        if self.arena_busy:
            return torch.empty(nbytes, dtype=torch.uint8, device=self.device), False, 0
        is_pinned = lambda a: a is not None and any(a is q for q in self.pinned)  # noqa: E731
        if (not capturing) and need_backward and is_pinned(self.arena):
            # park the graphs' block, take the unpinned one of the previous eager step back (alternating captures and eager
            # steps swap the same two blocks instead of leaving one more block behind per alternation: ADVICE r05)
            self.arena, self.spare = self.spare, self.arena
            if is_pinned(self.arena):
                self.arena = None
            self.arena_stream = None
        elif capturing and not is_pinned(self.arena):
            # prefer a block some graph already replays into over pinning one more
            best = min((a for a in self.pinned if a.numel() >= nbytes), key=lambda a: a.numel(), default=None)
            if best is not None:
                if self.arena is not None and (self.spare is None or self.spare.numel() < self.arena.numel() or is_pinned(self.spare)):
                    self.spare = self.arena
                self.arena = best
                self.arena_stream = None
        if self.arena is None or self.arena.numel() < nbytes:
            if capturing:  # (the shared block would have to be allocated outside the capture: let the graph's pool own one)
                return torch.empty(nbytes, dtype=torch.uint8, device=self.device), False, 0
            self.arena = None
            self.arena = torch.empty(int(nbytes * 1.05) + 4096, dtype=torch.uint8, device=self.device)
            self.arena_stream = None
            STATS["arena_bytes"] = self.arena.numel()
        if capturing and not is_pinned(self.arena):
            # a captured graph replays into THIS block for as long as it lives: never free it (a later, larger batch
            # allocates a new shared block; eager forwards in between only overwrite what every replay recomputes) - until the
            # caller says the graphs are gone: release_captured_workspaces(model)
            self.pinned.append(self.arena)
        cur = torch.cuda.current_stream(self.device)
        if self.arena_stream is not None and self.arena_stream != cur and not capturing:
            cur.wait_stream(self.arena_stream)  # (the last user of the block ran on another stream)
        self.arena_stream = cur
        self.arena_busy = need_backward
        self.arena_gen += 1
        return self.arena, need_backward, self.arena_gen


class _Lease:
    """Holds the shared workspace for one forward until its backward ran - or until the autograd graph that owned it was
    dropped without one (validation in training mode, an exception)."""

    def __init__(self, bind, owns):
        self.bind, self.owns = bind, owns

    def release(self):
        if self.owns:
            self.owns = False
            self.bind.arena_busy = False

    __del__ = release


def release_captured_workspaces(model) -> int:
    """Tell the binding that every hipGraph captured over ``model`` so far has been destroyed: the workspace blocks those graphs
    replayed into are no longer kept alive (-> number of blocks released).  Without this call a block is kept for the life of
    the model - a replay into freed memory would be silent corruption, and the binding cannot see a graph die."""
    b = model_cache(model).get("binding")
    if b is None:
        return 0
    n = len(b.pinned)
    if b.arena is not None and any(b.arena is a for a in b.pinned) and b.spare is not None and not b.arena_busy:
        b.arena, b.spare = b.spare, None
    b.pinned = []
    return n


_SLOT = [0]


class slot:
    """``with cmodel.slot(k):`` - forwards issued inside use the model's k-th binding: its own workspace block and its own helper
    streams (lane T, side, aux).  Two forwards of ONE model under different slots, each on its own torch stream, overlap on the GPU
    - what alignn_amd/microbatch.py uses to run the bond-row chains of one part of a batch under the triplet-row kernels of
    another.  (Slot 0 is the default; a backward finds its binding through the forward's context, not through the slot.)"""

    def __init__(self, k):
        self.k, self.prev = int(k), 0

    def __enter__(self):
        self.prev, _SLOT[0] = _SLOT[0], self.k
        return self

    def __exit__(self, *exc):
        _SLOT[0] = self.prev
        return False


def binding_of(model) -> Binding:
    mc = model_cache(model)
    key = "binding" if _SLOT[0] == 0 else f"binding{_SLOT[0]}"
    b = mc.get(key)
    if b is None or b.device != model.fc.weight.device:
        b = mc[key] = Binding(model)
    return b


class _ModelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bind, mb, keep, arena, arena_bytes, lease, *params):
        lib = _lib_model()
        out = torch.empty(mb.B, bind.desc.out_features, dtype=torch.float32, device=bind.device)
        t0 = time.perf_counter() if TIMING is not None else 0.0
        _lib.check(lib.alignn_model_fwd(bind.desc_addr, C.addressof(mb), arena.data_ptr(), arena_bytes, out.data_ptr(),
                                        _lib.stream()), "model_fwd")
        if TIMING is not None:
            TIMING["cfwd"] = TIMING.get("cfwd", 0.0) + time.perf_counter() - t0
        STATS["fwd"] += 1
        owns, gen = lease
        ctx.bind, ctx.mb, ctx.keep, ctx.arena, ctx.arena_bytes, ctx.lease = bind, mb, keep, arena, arena_bytes, _Lease(bind, owns)
        ctx.shared_gen = gen if arena is bind.arena else 0  # (0: a block of its own - nobody else writes into it)
        ctx.sig = bind.sig
        return out

    @staticmethod
    def backward(ctx, g_out):
        bind = ctx.bind
        lib = _lib_model()
        if ctx.arena is None:
            raise RuntimeError("alignn_amd.cmodel: backward called twice on a forward that ran in a workspace of its own "
                               "(released after the first backward)")
        if getattr(ctx, "tape_spent", False):
            raise RuntimeError("alignn_amd.cmodel: second backward through the same forward - the first one reused parts of the "
                               "forward's workspace (cmodel.REUSE_TAPE); set alignn_amd.cmodel.REUSE_TAPE = False (or "
                               "ALIGNN_AMD_REUSE_TAPE=0) for loss.backward(retain_graph=True) followed by another backward")
        if ctx.shared_gen and (bind.arena is not ctx.arena or bind.arena_gen != ctx.shared_gen or
                               (bind.arena_busy and not ctx.lease.owns)):
            # loss.backward(retain_graph=True) followed by another backward is fine as long as the forward's tape is intact
            # (the backward only reads it); once another forward has taken the shared workspace it is not
            raise RuntimeError("alignn_amd.cmodel: backward through a forward whose workspace another forward has reused "
                               "(retain_graph across model calls: set ALIGNN_AMD_CMODEL=0 for the per-operator path)")
        if bind.sig != ctx.sig:
            raise RuntimeError("alignn_amd.cmodel: the model's parameters moved between forward and backward")
        t_in = time.perf_counter() if TIMING is not None else 0.0
        g_out = g_out.contiguous()
        gflat = torch.empty(bind.grad_floats, dtype=torch.float32, device=bind.device)
        base = gflat.data_ptr()
        # straight into the optimizer's packed gradient buffer where it has one and this is the first gradient of the step
        # (p.grad is None everywhere: otherwise autograd ACCUMULATES and the slot already holds the earlier gradient)
        plan = bind.sink_plan() if GRAD_SINK else None
        if plan is not None and any(p.grad is not None for p in bind.params):
            plan = None
        for k, (owner, field, off) in enumerate(bind.grad_fields):
            dest = plan[0][k] if plan is not None else None
            setattr(owner, field, dest if dest is not None else base + 4 * off)
        STATS["sink"] = STATS.get("sink", 0) + (plan is not None)
        if plan is not None:
            bind.sink().sink_in_flight = True
        bind.set_mode()
        t0 = time.perf_counter() if TIMING is not None else 0.0
        try:
            _lib.check(lib.alignn_model_bwd(bind.desc_addr, C.addressof(ctx.mb), ctx.arena.data_ptr(), ctx.arena_bytes,
                                            g_out.data_ptr(), _lib.stream()), "model_bwd")
        finally:
            if TIMING is not None:
                TIMING["cbwd"] = TIMING.get("cbwd", 0.0) + time.perf_counter() - t0
            ctx.lease.release()  # (the next forward may take the block; a second backward of THIS graph re-checks above)
            if not ctx.shared_gen:
                ctx.arena = None  # a block of its own (up to 20 GB): do not keep it for as long as the graph object lives
        STATS["bwd"] += 1
        ctx.tape_spent = bool(bind.desc.reuse_tape)
        pieces = gflat.split_with_sizes(bind.sizes)
        grads = []
        slots = plan[1] if plan is not None else None
        for j, (i, shape, dead) in enumerate(zip(bind.keep, bind.shapes, bind.no_grad)):
            if dead:
                grads.append(None)
            elif slots is not None and slots[j] is not None:
                # (a FRESH view object: autograd's AccumulateGrad adopts an incoming gradient only if nobody else holds it -
                # a cached view would be cloned, param by param)
                grads.append(slots[j].view(shape))
            else:
                grads.append(pieces[i] if len(shape) == 1 else pieces[i].view(shape))
        if TIMING is not None:
            TIMING["bwd_py"] = TIMING.get("bwd_py", 0.0) + time.perf_counter() - t_in
        return (None,) * 6 + tuple(grads)


def applicable(model, b) -> bool:
    """Can the training-mode forward of this (model, batch) go through alignn_model_fwd / _bwd?"""
    return model.training and _structure_ok(model, b, torch.is_grad_enabled())


def infer_applicable(model, b) -> bool:
    """... and the eval-mode forward without autograd through alignn_model_infer?"""
    return (not model.training) and (not torch.is_grad_enabled()) and ops.INFER_FUSED and _structure_ok(model, b, False)


def _atomwise_cfg_ok(cfg) -> bool:
    """ALIGNNAtomWise configurations the C side carries: the plain energy (+ force / stress) head"""
    return (cfg.extra_features == 0 and cfg.output_features == 1 and not cfg.classification and cfg.link == "identity"
            and cfg.additional_output_features == 0 and not (cfg.atomwise_output_features > 0 and cfg.atomwise_weight != 0)
            and not cfg.use_cutoff_function and not cfg.include_pos_deriv)


def _structure_ok(model, b, need_grad, flavour="ALIGNN") -> bool:
    if not (ENABLED and _flags_default()):
        return False
    cfg = model.config
    if cfg.extra_features != 0 or cfg.alignn_layers < 1 or cfg.gcn_layers < 1 or cfg.hidden_features % 4:
        return False
    w = model.fc.weight
    if not w.is_cuda or w.dtype != torch.float32 or b.lg is None or b.atom_features is None or b.r is None:
        return False
    if type(model).__name__ != flavour:
        return False
    need_h = flavour == "ALIGNN" or not cfg.lg_on_fly
    if need_h and b.h is None:
        return False
    for t in (b.atom_features, b.r, b.h if need_h else None):
        if t is not None and (t.dtype != torch.float32 or t.requires_grad or not t.is_cuda):
            return False
    if flavour == "ALIGNNAtomWise" and not _atomwise_cfg_ok(cfg):
        return False
    want_norm = "batch" if flavour == "ALIGNN" else "layer"
    mc = model_cache(model)
    ok = mc.get("static_ok")
    if ok is None:
        from .alignn import EdgeGatedGraphConv, MLPLayer

        ok = all(getattr(m, "_norm", "batch") == want_norm and (not isinstance(m, EdgeGatedGraphConv) or m.residual)
                 for m in model.modules() if isinstance(m, (EdgeGatedGraphConv, MLPLayer)))
        mc["static_ok"] = ok
    if not ok:
        return False
    # hooks on a layer (feature extraction, the activation checks of tests/test_gpu_full_size.py, DDP-style wrappers of a
    # submodule) fire from the layers' own forward: the per-operator path calls those, one C call for the whole model does not
    subs = mc.get("submodules")
    if subs is None:
        subs = mc["submodules"] = [m for m in model.modules() if m is not model]
    import torch.nn.modules.module as _tm

    if _tm._global_forward_hooks or _tm._global_forward_pre_hooks or _tm._global_backward_hooks:
        return False
    mode = model.training
    for m in subs:
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks:
            return False
        # a model in train() with some blocks in eval() (frozen-statistics fine-tuning): every layer of the per-operator path
        # follows ITS OWN flag (batch statistics vs running statistics, which counters are bumped); one C call runs one mode
        if m.training != mode:
            return False
    slots = mc.get("slots")
    if slots is None:  # [(module._parameters, name, parameter)]: a replaced Parameter object is noticed without walking the tree
        slots = mc["slots"] = [(mod._parameters, name, p) for mod in model.modules()
                               for name, p in mod._parameters.items() if p is not None]
    for d, name, p in slots:
        if d.get(name) is not p:
            sink = mc.get("grad_sink")
            mc.clear()
            if sink is not None:  # (the optimizer registers itself once, in _build: a replaced Parameter has no slot in its
                mc["grad_sink"] = sink  # buffer and sink_plan() routes that block to the private buffer - the rest stays)
            return _structure_ok(model, b, need_grad, flavour)
        if need_grad and not p.requires_grad:
            return False
    if any(p.dtype != torch.float32 for p in (model.atom_embedding.layer[0].weight, model.fc.bias)):
        return False
    return True


def forward(model, b, h=None):
    """-> ``fc(AvgPooling(...))`` [B, out_features] of a training-mode forward, or None when the C side does not carry a
    kernel choice this (model, batch) needs (the caller then takes the per-operator path).  ``h``: bond-angle cosines to use
    instead of ``b.h`` (ALIGNNAtomWise with lg_on_fly recomputes them from the bond vectors)."""
    bind = binding_of(model)
    with _lib.device_guard(model.fc.weight):
        bind.refresh()
        capturing = bind.set_mode()
        mb = bind.batch_struct(b)
        af, r, h = b.atom_features.contiguous(), b.r.contiguous(), (b.h if h is None else h).contiguous()
        if af.shape != (b.g.n_nodes, bind.desc.atom_in) or r.shape != (b.g.n_edges, 3) or h.numel() != b.lg.n_edges:
            raise ValueError("feature rows do not match the graphs")
        mb.atom_features, mb.r, mb.h = af.data_ptr(), r.data_ptr(), h.data_ptr()
        fwd_bytes, total = bind.plan(mb)
        if total is None:
            return None
        need_bwd = torch.is_grad_enabled()
        nbytes = total if need_bwd else fwd_bytes
        arena, owns, gen = bind.take_arena(nbytes, capturing, need_bwd)
        ops.new_weight_generation()
        try:
            return _ModelFn.apply(bind, mb, (b, af, r, h), arena, nbytes, (owns, gen), *bind.params)
        except BaseException:
            if owns:  # (the forward raised before a lease existed: hand the shared workspace back)
                bind.arena_busy = False
            raise


def infer(model, b):
    """-> ``fc(AvgPooling(...))`` [B, out_features] of an eval-mode forward without autograd (one C call), or None when the C
    side does not carry a kernel choice this (model, batch) needs."""
    bind = binding_of(model)
    lib = _lib_model()
    with _lib.device_guard(model.fc.weight):
        bind.refresh()
        capturing = bind.set_mode()
        mb = bind.batch_struct(b)
        af, r, h = b.atom_features.contiguous(), b.r.contiguous(), b.h.contiguous()
        if af.shape != (b.g.n_nodes, bind.desc.atom_in) or r.shape != (b.g.n_edges, 3) or h.numel() != b.lg.n_edges:
            raise ValueError("feature rows do not match the graphs")
        mb.atom_features, mb.r, mb.h = af.data_ptr(), r.data_ptr(), h.data_ptr()
        key = ("infer", mb.g.n, mb.g.m, mb.lg.m, mb.B, bool(bind.desc.lane_T), bind.desc.lane_min_rows, bind.desc.angle_fused)
        nbytes = bind.plans.get(key)
        if nbytes is None:
            nbytes = bind.plans[key] = lib.alignn_model_infer_workspace(bind.desc_addr, C.addressof(mb))
        if not nbytes:
            return None
        arena, _owns, _gen = bind.take_arena(nbytes, capturing, False)
        out = torch.empty(mb.B, bind.desc.out_features, dtype=torch.float32, device=bind.device)
        _lib.check(lib.alignn_model_infer(bind.desc_addr, C.addressof(mb), arena.data_ptr(), nbytes, out.data_ptr(), _lib.stream()),
                   "model_infer")
        STATS["infer"] = STATS.get("infer", 0) + 1
        return out


# ---------------------------------------------------------------------------------------------------------------------
# ALIGNNAtomWise with the force / stress head: alignn_ff_eval / alignn_ff_grad (csrc/model.hip, csrc/ff.hip)
# ---------------------------------------------------------------------------------------------------------------------
def atomwise_applicable(model, b, need_param_grads) -> bool:
    """Can this (ALIGNNAtomWise, batch) go through the whole-model C entry points?  (LayerNorm flavour, the plain energy
    / force / stress head - ``_atomwise_cfg_ok`` -, float32 on a HIP device, every parameter trainable when gradients are wanted)"""
    return _structure_ok(model, b, need_param_grads, "ALIGNNAtomWise")


def _ff_desc(model, b):
    cfg = model.config
    f = FFDesc()
    f.lg_on_fly, f.add_reverse_forces = int(cfg.lg_on_fly), int(cfg.add_reverse_forces)
    f.force_mult_natoms, f.energy_mult_natoms = int(cfg.force_mult_natoms), int(cfg.energy_mult_natoms)
    f.has_stress = int(cfg.stresswise_weight != 0)
    f.use_penalty = int(cfg.use_penalty)
    from . import ff2

    f.dense_lg_reverse = int(ff2.DENSE_LG_REVERSE and ops.FUSED_LG_BACKWARD and ops.DENSE_LG_BACKWARD)
    f.grad_multiplier, f.stress_multiplier = float(cfg.grad_multiplier), float(cfg.stress_multiplier)
    f.penalty_factor, f.penalty_threshold = float(cfg.penalty_factor), float(cfg.penalty_threshold)
    vol = None
    if f.has_stress:
        vol = b.cell_volumes()  # float32, contiguous, on the device, one per crystal (or ValueError)
        f.volume = vol.data_ptr()
    return f, vol


def _ff_prepare(model, b, need_grad):
    """-> (bind, mb, ffd, keep-alive tuple, arena, bytes, (owns, gen)) or None (kernel choice not carried)"""
    bind = binding_of(model)
    bind.refresh()
    capturing = bind.set_mode()
    mb = bind.batch_struct(b)
    af, r = b.atom_features.contiguous(), b.r.contiguous()
    h = None if model.config.lg_on_fly else b.h.contiguous()
    if af.shape != (b.g.n_nodes, bind.desc.atom_in) or r.shape != (b.g.n_edges, 3) or (h is not None and h.numel() != b.lg.n_edges):
        raise ValueError("feature rows do not match the graphs")
    mb.atom_features, mb.r, mb.h = af.data_ptr(), r.data_ptr(), _ptr(h)
    ffd, vol = _ff_desc(model, b)
    key = ("ff", mb.g.n, mb.g.m, mb.lg.m, mb.B, mb.lg.dense_max_src, bool(mb.lg.grp_seg_ptr), bool(mb.lg.seg_rank),
           bool(bind.desc.lane_T), bool(bind.desc.side), bool(bind.desc.aux), bind.desc.side_min_rows, bind.desc.lane_min_rows,
           ffd.lg_on_fly, ffd.has_stress, ffd.dense_lg_reverse)
    hit = bind.plans.get(key)
    if hit is None:
        ev, tot = C.c_size_t(0), C.c_size_t(0)
        rc = _lib_model().alignn_ff_plan(bind.desc_addr, C.addressof(mb), C.addressof(ffd), C.addressof(ev), C.addressof(tot))
        STATS["plans"] += 1
        if rc == _NOT_SUPPORTED:
            hit = (None, None)
        else:
            _lib.check(rc, "ff_plan")
            hit = (ev.value, tot.value)
        bind.plans[key] = hit
    if hit[1] is None:
        return None
    nbytes = hit[1] if need_grad else hit[0]
    arena, owns, gen = bind.take_arena(nbytes, capturing, need_grad)
    ops.new_weight_generation()
    return bind, mb, ffd, (b, af, r, h, vol), arena, nbytes, (owns, gen)


def _ff_outputs(bind, mb, ffd):
    dev = bind.device
    out = torch.empty(mb.B, dtype=torch.float32, device=dev)
    forces = torch.empty(mb.g.n, 3, dtype=torch.float32, device=dev)
    stress = torch.empty(mb.B, 3, 3, dtype=torch.float32, device=dev) if ffd.has_stress else None
    return out, forces, stress


class _FFFn(torch.autograd.Function):
    """(energies, forces, stresses) of ALIGNNAtomWise as ONE autograd node over two C calls: forward = alignn_ff_eval (values;
    the tape stays in the workspace), backward = alignn_ff_grad (the loss gradient THROUGH the forces as one reverse pass over a
    tangent-carrying forward pass - alignn_amd/ff2.py's scheme, issued from C)."""

    @staticmethod
    def forward(ctx, bind, mb, ffd, keep, arena, arena_bytes, lease, *params):
        lib = _lib_model()
        out, forces, stress = _ff_outputs(bind, mb, ffd)
        _lib.check(lib.alignn_ff_eval(bind.desc_addr, C.addressof(mb), C.addressof(ffd), arena.data_ptr(), arena_bytes,
                                      out.data_ptr(), forces.data_ptr(), _ptr(stress), _lib.stream()), "ff_eval")
        STATS["ff_eval"] = STATS.get("ff_eval", 0) + 1
        owns, gen = lease
        ctx.bind, ctx.mb, ctx.ffd, ctx.keep, ctx.arena, ctx.arena_bytes, ctx.lease = bind, mb, ffd, keep, arena, arena_bytes, _Lease(bind, owns)
        ctx.shared_gen = gen if arena is bind.arena else 0
        ctx.sig = bind.sig
        ctx.has_stress = stress is not None
        if stress is None:
            stress = torch.empty(1, device=out.device)
            ctx.mark_non_differentiable(stress)
        return out, forces, stress

    @staticmethod
    def backward(ctx, g_out, g_forces, g_stress):
        bind = ctx.bind
        lib = _lib_model()
        if ctx.arena is None:
            raise RuntimeError("alignn_amd.cmodel: backward called twice on a forward that ran in a workspace of its own")
        if ctx.shared_gen and (bind.arena is not ctx.arena or bind.arena_gen != ctx.shared_gen):
            raise RuntimeError("alignn_amd.cmodel: backward through a forward whose workspace another forward has reused "
                               "(retain_graph across model calls: set ALIGNN_AMD_CMODEL=0 for the per-operator path)")
        if bind.sig != ctx.sig:
            raise RuntimeError("alignn_amd.cmodel: the model's parameters moved between forward and backward")
        g_out = None if g_out is None else g_out.reshape(-1).to(torch.float32).contiguous()
        g_forces = None if g_forces is None else g_forces.reshape(-1, 3).to(torch.float32).contiguous()
        g_stress = None if (g_stress is None or not ctx.has_stress) else g_stress.reshape(-1, 3, 3).to(torch.float32).contiguous()
        gflat = torch.empty(2, bind.grad_floats, dtype=torch.float32, device=bind.device)
        base = gflat.data_ptr()
        # straight into the optimizer's packed gradient buffer where it has one (as _ModelFn.backward; the tangent halves of the
        # weight gradients of those blocks go to a scratch twin of the buffer and are added in place by the C call)
        plan = bind.sink_plan() if GRAD_SINK else None
        sink_buf = sink_t = None
        if plan is not None and not any(p.grad is not None for p in bind.params):
            ref = model_cache(bind.model).get("grad_sink")
            sink = ref() if ref is not None else None
            sink_buf = getattr(sink, "_grad_all", None)
            if (sink_buf is None or sink_buf.dtype != torch.float32 or sink_buf.numel() % 4 or sink_buf.data_ptr() % 16
                    or not sink_buf.is_contiguous()):
                sink_buf = None
        if sink_buf is None:
            plan = None
        else:
            sink_t = torch.empty_like(sink_buf)
        for k, (owner, field, off) in enumerate(bind.grad_fields):
            dest = plan[0][k] if plan is not None else None
            setattr(owner, field, dest if dest is not None else base + 4 * off)
        STATS["ff_sink"] = STATS.get("ff_sink", 0) + (plan is not None)
        if plan is not None:
            bind.sink().sink_in_flight = True
        bind.set_mode()
        try:
            _lib.check(lib.alignn_ff_grad(bind.desc_addr, C.addressof(ctx.mb), C.addressof(ctx.ffd), ctx.arena.data_ptr(),
                                          ctx.arena_bytes, _ptr(g_out), _ptr(g_forces), _ptr(g_stress), gflat[0].data_ptr(),
                                          gflat[1].data_ptr(), bind.grad_floats, _ptr(sink_buf), _ptr(sink_t),
                                          sink_buf.numel() if sink_buf is not None else 0, _lib.stream()), "ff_grad")
        finally:
            ctx.lease.release()
            if not ctx.shared_gen:
                ctx.arena = None
        STATS["ff_grad"] = STATS.get("ff_grad", 0) + 1
        pieces = gflat[0].split_with_sizes(bind.sizes)
        grads = []
        slots = plan[1] if plan is not None else None
        for j, (i, shape, dead) in enumerate(zip(bind.keep, bind.shapes, bind.no_grad)):
            if dead:
                grads.append(None)
            elif slots is not None and slots[j] is not None:
                grads.append(slots[j].view(shape))  # (a fresh view object: AccumulateGrad adopts it instead of cloning)
            else:
                grads.append(pieces[i] if len(shape) == 1 else pieces[i].view(shape))
        return (None,) * 7 + tuple(grads)


def ff_train(model, b):
    """-> (out [B], forces [N, 3], stresses [B, 3, 3] or a placeholder) differentiable w.r.t. the parameters, or None when the C
    side does not carry a kernel choice this (model, batch) needs."""
    with _lib.device_guard(model.fc.weight):
        prep = _ff_prepare(model, b, True)
        if prep is None:
            return None
        bind, mb, ffd, keep, arena, nbytes, lease = prep
        try:
            return _FFFn.apply(bind, mb, ffd, keep, arena, nbytes, lease, *bind.params)
        except BaseException:
            if lease[0]:
                bind.arena_busy = False
            raise


def ff_eval(model, b):
    """-> (out, forces, stresses or None) as plain values (MD / calculators: alignn/ff/calculators.py:280-291), ONE C call; None
    when the C side does not carry a kernel choice this (model, batch) needs."""
    with _lib.device_guard(model.fc.weight):
        prep = _ff_prepare(model, b, False)
        if prep is None:
            return None
        bind, mb, ffd, _keep, arena, nbytes, _lease = prep
        out, forces, stress = _ff_outputs(bind, mb, ffd)
        _lib.check(_lib_model().alignn_ff_eval(bind.desc_addr, C.addressof(mb), C.addressof(ffd), arena.data_ptr(), nbytes,
                                               out.data_ptr(), forces.data_ptr(), _ptr(stress), _lib.stream()), "ff_eval")
        STATS["ff_eval"] = STATS.get("ff_eval", 0) + 1
        return out, forces, stress
