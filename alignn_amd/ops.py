"""torch.autograd glue over the C ABI of libalignn_hip.so.

Every forward/backward here is a fixed sequence of kernel launches on the current HIP stream;
torch only provides the tensors (caching allocator) and the autograd tape.  Nothing in this file
computes on the CPU or through torch kernels (a handful of tiny ``torch.empty`` / ``torch.cat``
bookkeeping calls aside), and nothing falls back if the library is missing.
"""

from __future__ import annotations

import weakref

import torch

from . import _lib
from ._lib import check, ptr, require_f32, stream
from .graph import CSRGraph

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
FUSED_LG_BACKWARD = True  # tests flip this to compare against the generic two-pass backward
INFER_FUSED = True  # eval-mode convs without autograd take the BatchNorm-folded gate pass (tests flip it to compare)
DENSE_LG_BACKWARD = True  # ... and this one to compare the dense-block kernel against the by-source fused kernel


# Force-field inference differentiates the energy w.r.t. the geometry only: inside ``no_param_grad()`` the autograd
# nodes built here skip their weight / bias / norm-parameter gradients (the engine would otherwise compute them - the
# parameters require grad - and throw them away).
_PARAM_GRADS = {"on": True}


class no_param_grad:
    def __enter__(self):
        self.prev = _PARAM_GRADS["on"]
        _PARAM_GRADS["on"] = False

    def __exit__(self, *exc):
        _PARAM_GRADS["on"] = self.prev


def _empty(*shape, like):
    return torch.empty(*shape, dtype=torch.float32, device=like.device)


# ---------------------------------------------------------------------------------------------
# side stream for weight gradients
#
# dW = G^T X is MFMA-bound and nothing in the rest of the backward pass depends on it, while the kernels
# that follow it on the critical path (BatchNorm reductions, gate backward) are HBM-bound.  Issuing the
# weight-gradient GEMMs and bias column-sums on a second HIP stream lets the two kinds of work share the
# CUs.  The main stream joins the side stream once, in a callback the autograd engine runs at the end of
# backward(), so callers see ordinary semantics (grads are ready for optimizer.step()).
# ---------------------------------------------------------------------------------------------
import os as _os

_SIDE = {"enabled": _os.environ.get("ALIGNN_AMD_SIDE_STREAM", "1") != "0", "streams": {}, "armed": False,
         "min_rows": int(_os.environ.get("ALIGNN_AMD_SIDE_MIN_ROWS", "32768"))}

# Parameters whose owner promises to read ``.grad`` only after backward() has returned (alignn_amd.ddp.FlatGradSync
# marks its parameters): under an initialised process group everything else is assumed to sit under
# DistributedDataParallel, whose reducer copies gradients into its buckets from C++ hooks DURING backward.
GRAD_READ_AFTER_BACKWARD = "_alignn_grad_read_after_backward"


# Forward passes that used a leaf since the last end-of-backward join.  Two autograd nodes of ONE backward that share a
# parameter - (l(model(b1)) + l(model(b2))).backward() - make the engine ADD their gradients as soon as the second one
# arrives, on the main stream: a gradient still being computed on the side stream would be read early (found in round 6:
# 0.3 % of the gradient scale off, tools history in DESIGN section 4b).  A leaf seen by more than one forward joins at once.
_FWD_USES = {}


def _note_forward(leaves):
    # (called from autograd.Function.forward, where grad mode is always off: count every forward that can have a backward)
    for p in leaves:
        if p is not None and p.requires_grad:
            _FWD_USES[id(p)] = _FWD_USES.get(id(p), 0) + 1


def _clear_forward_uses():
    _FWD_USES.clear()
    _SIDE["uses_armed"] = False


def _join_side_streams():
    _SIDE["armed"] = False
    for dev, side in _SIDE["streams"].items():
        torch.cuda.current_stream(dev).wait_stream(side)
    for dev, lane in _LANE["streams"].items():
        torch.cuda.current_stream(dev).wait_stream(lane)


def _deferred_join_is_safe(params):
    """May the gradients ``params`` receive stay on the side stream until the end of backward()?

    The autograd engine believes every output of a node was produced on the node's own (= the main) stream.  That is
    harmless only if nothing launches a kernel on those gradients before the end-of-backward join: AccumulateGrad of a
    leaf with ``grad is None`` and grad mode off just keeps the tensor.  Anything else - an existing ``.grad`` to add
    to (gradient accumulation, ``zero_grad(set_to_none=False)``), tensor / post-accumulate hooks, ``create_graph``
    (clone), a non-leaf weight (more autograd nodes downstream), DistributedDataParallel's reducer hooks - reads them
    on the main stream right away, so then the main stream joins the side stream before the node returns."""
    if _FWD_USES and not _SIDE.get("uses_armed"):  # (forget the forward passes of this step when its backward is over)
        _SIDE["uses_armed"] = True
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_clear_forward_uses)
        except RuntimeError:  # not inside backward()
            _clear_forward_uses()
    if torch.is_grad_enabled():
        return False
    ddp_possible = torch.distributed.is_available() and torch.distributed.is_initialized()
    for p in params:
        if p is None:
            continue
        if (not p.is_leaf or p.grad is not None or p._backward_hooks or _FWD_USES.get(id(p), 0) > 1
                or getattr(p, "_post_accumulate_grad_hooks", None)
                or (ddp_possible and not getattr(p, GRAD_READ_AFTER_BACKWARD, False))):
            return False
    return True


def on_side_stream(fn, inputs, params=(), wait=()):
    """Run ``fn()`` (kernel launches only) on the side stream; returns its tensors.  ``inputs`` are the tensors it
    reads (kept alive for the allocator until the side stream has consumed them); ``params`` the leaves whose
    gradients it computes - they decide whether the join may wait until the end of backward()
    (``_deferred_join_is_safe``) or has to happen before this call returns."""
    if not _SIDE["enabled"]:
        return fn()
    dev = inputs[0].device
    main = torch.cuda.current_stream(dev)
    # Eagerly launched, the fork costs the HOST ~80 us per call (stream switch, an event, a record_stream per input) - more
    # than the GPU gets back when the product is small and the step is launch-bound anyway (8 crystals: 7.9-8.7 ms with every
    # weight gradient forked, 6.3-6.7 ms with none).  So outside a stream capture only products over many rows go to the side
    # stream; inside a capture (host time irrelevant) all of them do.
    if inputs[0].dim() > 0 and inputs[0].shape[0] < _SIDE["min_rows"] and not torch.cuda.is_current_stream_capturing():
        for ev in wait:
            if ev is not None:
                main.wait_event(ev)
        return fn()
    side = _SIDE["streams"].get(dev)
    if side is None:
        side = _SIDE["streams"][dev] = torch.cuda.Stream(device=dev)
    side.wait_stream(main)
    for ev in wait:  # work of lane T that ``fn`` reads
        if ev is not None:
            side.wait_event(ev)
    with torch.cuda.stream(side):
        outs = fn()
    for t in inputs:
        if t is not None:
            t.record_stream(side)
    for o in outs:
        if o is not None:
            o.record_stream(main)
    if not _deferred_join_is_safe(params):
        main.wait_stream(side)
        return outs
    if not _SIDE["armed"]:
        _SIDE["armed"] = True
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_join_side_streams)
        except RuntimeError:  # not inside backward(): join right away
            _join_side_streams()
    return outs


# ---------------------------------------------------------------------------------------------
# two lanes: the triplet-row (line-graph edge) kernels on their own stream
#
# A training step is two interleaved chains.  The T-row chain (T = rows of the line graph's edge features, 676 k at the
# benchmark batch) is a dozen HBM-bound kernels of 0.3-0.4 ms per line-graph convolution; the atom / bond chain (N and
# E rows: the bond-graph convolutions, the node half of the line-graph convolutions, their projections and norms) is
# ~400 launches of 5-100 us that are latency-bound and leave most of the chip idle.  The two meet only at a few points
# per layer (the gate pass needs the node projection P; the node norm needs the gate pass' sums; in backward the
# block kernel needs gS1 / gS0 and hands back GP), so inside ``lanes()`` the T-row kernels go to a second stream
# ("lane T") and the small ones stay on the caller's stream, with events at exactly those meeting points.  Both
# lanes are part of a hipGraph capture (fork / join by events).  The caller's stream joins lane T at the end of the
# model forward and at the end of backward (same engine callback as the weight-gradient side stream), or right away
# where a gradient computed on lane T could be read earlier (``_deferred_join_is_safe``).
# ---------------------------------------------------------------------------------------------
# ALIGNN_AMD_LANES: "auto" (default) = only while the step is being captured into a hipGraph - there the fork / join
# events cost nothing at replay; eagerly launched steps are bound by the host's enqueue rate on most hosts and the extra
# event / stream calls (+1-4 ms per step) cost more than the overlap returns (-0.5 ms) -, "1" = always, "0" = never.
_LANE = {"enabled": _os.environ.get("ALIGNN_AMD_LANES", "auto"), "min_rows": int(_os.environ.get("ALIGNN_AMD_LANE_MIN_ROWS", "131072")),
         "active": False,
         "streams": {}, "main": None, "T": None, "priority": 0}
_ON_T = {}  # id -> weakref: tensors whose producer kernel ran on lane T (consumers on lane T need no event)


def _mark_on_T(t):
    if t is not None:
        k = id(t)
        _ON_T[k] = weakref.ref(t, lambda _r, k=k: _ON_T.pop(k, None))
    return t


def _is_on_T(t):
    e = _ON_T.get(id(t))
    return e is not None and e() is t


class lanes:
    """``with ops.lanes(device):`` around a whole-model forward: T-row kernels of the convolutions / embeddings inside
    run on lane T (see above).  Nesting and ``ALIGNN_AMD_LANES=0`` make it a no-op."""

    def __init__(self, device):
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.dev = dev
        self.on = False

    def __enter__(self):
        mode = _LANE["enabled"]
        if mode in ("0", False) or _LANE["active"] or self.dev.type != "cuda":
            return self
        if mode == "auto" and not torch.cuda.is_current_stream_capturing():
            return self
        self.on = True
        main = torch.cuda.current_stream(self.dev)
        T = _LANE["streams"].get(self.dev)
        if T is None:
            T = _LANE["streams"][self.dev] = torch.cuda.Stream(device=self.dev, priority=_LANE["priority"])
        _LANE.update(active=True, main=main, T=T)
        new_amax_arena(self.dev)  # zero-filled on the caller's stream, before the fork
        T.wait_stream(main)  # parameters (optimizer step), inputs, the arena
        return self

    def __exit__(self, *exc):
        if self.on:
            _LANE["main"].wait_stream(_LANE["T"])
            _LANE["active"] = False
        return False


def _lane_for(rows, *tensors):
    """-> (main, T) if a kernel over ``rows`` rows should go to lane T now, else None.  In backward the lanes context is
    gone: the Function remembers (``ctx.lane``) and passes ``force=True`` through ``_lane_streams``."""
    if _LANE["active"] and rows >= _LANE["min_rows"]:
        return _LANE["main"], _LANE["T"]
    return None


def _lane_streams(dev):
    """(main, T) for a backward node whose forward ran with lanes: main = the stream the engine runs the node on."""
    main, T = torch.cuda.current_stream(dev), _LANE["streams"][dev]
    _LANE.update(main=main, T=T)
    return main, T


def _event_after(stream):
    ev = torch.cuda.Event()
    ev.record(stream)
    return ev


class _on_T:
    """Run the enclosed launches (and allocations) on lane T.  ``wait``: events lane T waits for first; tensors in
    ``reads`` that were not produced on lane T make it wait for the main stream's current position instead."""

    def __init__(self, main, T, wait=(), reads=()):
        self.main, self.T, self.wait, self.reads = main, T, wait, reads

    def __enter__(self):
        if any(t is not None and not _is_on_T(t) for t in self.reads):
            self.T.wait_event(_event_after(self.main))
        for ev in self.wait:
            if ev is not None:
                self.T.wait_event(ev)
        for t in self.reads:
            if t is not None and not _is_on_T(t):
                t.record_stream(self.T)
        self.ctx = torch.cuda.stream(self.T)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        return self.ctx.__exit__(*exc)


def _main_reads(*tensors):
    """The caller's stream is about to read these: wait for lane T if one of them was produced there."""
    if _ON_T and any(t is not None and _is_on_T(t) for t in tensors):
        T = _LANE["T"]
        cur = torch.cuda.current_stream(T.device)
        cur.wait_event(_event_after(T))
        for t in tensors:
            if t is not None and _is_on_T(t):
                t.record_stream(cur)  # (allocated in lane T's pool)


def main_reads(*tensors):
    """Public form of ``_main_reads`` for model code: call it on any tensor a lane-aware layer (MLPLayer,
    EdgeGatedGraphConv) returned BEFORE handing it to a consumer that is not lane-aware (a plain torch operation such as
    ``y * c_off`` or ``y[perm]``) - inside ``lanes()`` that tensor may have been produced on lane T, and a consumer enqueued
    on the caller's stream without this event races with its producer (and is missing from a captured hipGraph as a
    dependency).  A no-op when nothing is marked."""
    _main_reads(*tensors)
    return tensors[0] if len(tensors) == 1 else tensors


def _arm_backward_join():
    if not _SIDE["armed"]:
        _SIDE["armed"] = True
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_join_side_streams)
        except RuntimeError:  # not inside backward(): join right away
            _join_side_streams()


# ---------------------------------------------------------------------------------------------
# thin launch wrappers
# ---------------------------------------------------------------------------------------------
def gemm_nt(a, w, bias=None, addend=None, out=None):
    """out[M,N] = a[M,K] @ w[N,K]^T (+bias) (+addend)."""
    lib = _lib.load()
    require_f32(a, w, bias, addend)
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = _empty(M, N, like=a)
    check(
        lib.alignn_gemm_nt(ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(addend),
                           addend.stride(0) if addend is not None else 0, ptr(out), out.stride(0), M, N, K, stream()),
        "gemm_nt",
    )
    return out


X6_MIN_TILES = 256  # 64x256 tiles below which the fp32-MFMA kernel's finer tiles win (100-512 measured the same)
F16X3 = True  # use the three-product fp16 scheme wherever max|A| is known (False: always bf16x6)

# max|x| of activations, tracked by the kernels that produce them (one device float per tensor).  Keyed by object
# identity with a weak reference, so an entry dies with its tensor and can never describe recycled memory, and
# stamped with the tensor's version counter, so an in-place edit after registration silently drops the bound (the
# projection then takes the range-free bf16x6 scheme).
_AMAX = {}

# Hits and misses of the identity-keyed side-band registries (``_AMAX``, ``_W_IMG_T`` / ``_W_AMAX``, ``_NORM_SRC``,
# ``_PRE_RED``).  A miss is always SAFE (the consumer measures / reduces / slices for itself, or takes the range-free
# bf16x6 scheme) but it is a slower step, and anything that clones a tensor between two autograd nodes (a hook,
# ``torch.utils.checkpoint``, DistributedDataParallel's bucket views) causes one silently - so they are counted
# (tests/test_gpu_round3.py pins the counts of a default-config step, plain and DDP-wrapped) and the two expensive
# fall-backs warn once per process.
REGISTRY_STATS = {"amax_hit": 0, "amax_miss": 0, "wimg_hit": 0, "wimg_miss": 0, "norm_src_hit": 0, "norm_src_miss": 0,
                  "pre_red_hit": 0, "pre_red_miss": 0, "bf16x6_fallback": 0, "separate_bn_reduce": 0}
_WARNED = set()


def reset_registry_stats():
    for k in REGISTRY_STATS:
        REGISTRY_STATS[k] = 0
    BNRED_STATS["fused"] = BNRED_STATS["used"] = 0


def _warn_once(key, msg):
    if key not in _WARNED:
        _WARNED.add(key)
        import warnings

        warnings.warn(msg, RuntimeWarning, stacklevel=3)


def set_amax(t, amax):
    k = id(t)
    _AMAX[k] = (weakref.ref(t, lambda _r, k=k: _AMAX.pop(k, None)), amax, t._version)
    return t


def get_amax(t):
    e = _AMAX.get(id(t))
    hit = e is not None and e[0]() is t and e[2] == t._version
    if t.dim() == 2 and _track(t.shape[0]):  # (only tensors a producer WOULD have tracked count as misses)
        REGISTRY_STATS["amax_hit" if hit else "amax_miss"] += 1
    return e[1] if hit else None


_AMAX_ARENA = {"buf": None, "next": 0}


def reset_amax_arena():
    """Start a fresh arena (hipGraph capture calls this so that the zero-fill is part of the captured graph)."""
    _AMAX_ARENA["buf"] = None


def new_amax_arena(device):
    """Zero-fill a fresh arena.  With two lanes the fill goes to the main stream and lane T is made to wait for it, so
    that whichever lane's kernel raises a slot first finds it zeroed."""
    a = _AMAX_ARENA
    main, T = _LANE["main"], _LANE["T"]
    if main is not None and main.device == torch.device(device) and (_LANE["active"] or torch.cuda.current_stream(device) == T):
        with torch.cuda.stream(main):
            a["buf"] = torch.zeros(512, dtype=torch.float32, device=device)
        a["buf"].record_stream(T)
        T.wait_event(_event_after(main))
    else:
        a["buf"] = torch.zeros(512, dtype=torch.float32, device=device)
    a["next"] = 0


def new_amax(like):
    """A zeroed device scalar for a producer kernel to atomicMax into: a slot of a 512-float arena, so a training
    step pays one or two fill kernels instead of ~60."""
    a = _AMAX_ARENA
    if a["buf"] is None or a["next"] >= a["buf"].numel() or a["buf"].device != like.device:
        new_amax_arena(like.device)
    i = a["next"]
    a["next"] = i + 1
    return a["buf"][i:i + 1]


AMAX_MIN_ROWS = 4096  # below this no projection of the tensor can reach X6_MIN_TILES 64-row tiles (N <= 1024): skip tracking


def _track(rows):
    return F16X3 and rows >= AMAX_MIN_ROWS


def absmax(x, arena=False):
    """max|x| of a 2-D fp32 tensor as a device scalar (stand-alone pass; the fused producers avoid it).  ``arena``: into
    a zeroed slot of the step's amax arena - no reset launch of its own."""
    lib = _lib.load()
    require_f32(x)
    if arena:
        out = new_amax(x)
        check(lib.alignn_absmax_raise(ptr(x), x.stride(0), x.shape[0], x.shape[1], ptr(out), stream()), "absmax_raise")
        return out
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    check(lib.alignn_absmax(ptr(x), x.stride(0), x.shape[0], x.shape[1], ptr(out), stream()), "absmax")
    return out


class SplitWeight:
    """A weight pre-sliced into the bf16x6 kernel's DMA image (opaque device buffer + logical shape)."""

    def __init__(self, buf, n, k):
        self.buf, self.n, self.k = buf, n, k


def split_bf16x3(w, transpose=False):
    """Slice ``w`` [N,K] (or ``w^T`` of a stored [K,N] matrix) into three truncated-bf16 planes."""
    lib = _lib.load()
    require_f32(w)
    n, k = (w.shape[1], w.shape[0]) if transpose else (w.shape[0], w.shape[1])
    buf = torch.empty(lib.alignn_split_bf16x3_bytes(n, k), dtype=torch.uint8, device=w.device)
    check(lib.alignn_split_bf16x3(ptr(w), w.stride(0), n, k, int(transpose), ptr(buf), stream()), "split_bf16x3")
    return SplitWeight(buf, n, k)


_W_AMAX = {}
_W_IMG_T = {}  # id(w) -> (weakref, stamp, SplitWeight of w^T): made with the forward image, used by the way back
# Validity stamp of those two caches.  ``w._version`` alone is not enough: the fused node projection ``wcat`` is a plain
# tensor whose storage is updated THROUGH the four Parameters that alias it (``.data`` views), and under FlatAdamW the
# update goes through the flat buffer - neither bumps ``wcat._version`` (or the Parameters').  So an entry also carries
# the weight GENERATION: bumped after every optimizer step (torch's global step hook - any torch.optim.Optimizer,
# FlatAdamW included) and at the start of every whole-model forward (``new_weight_generation``).  A backward whose own
# forward did not slice the weight (e.g. it took the bf16x6 path because max|x| was unknown) can then never meet the
# image or max|w| of an earlier step.
_WGEN = [0]


def new_weight_generation(*_a, **_k):
    _WGEN[0] += 1


def _wstamp(w):
    return (w._version, _WGEN[0])


from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_step_hook  # noqa: E402

_reg_step_hook(new_weight_generation)
SPLIT_BOTH = True  # both slice images of a weight from one launch (tests flip it: same bits)


_W_IMG = {}  # id(w) -> (weakref, stamp, SplitWeight of w): images made ahead of the step by WeightPrep (below)
BATCHED_WEIGHT_PREP = True  # tests flip it: same bits


WEIGHT_PREP_STATS = {"runs": 0, "weights": 0}


class WeightPrep:
    """max|W| and both slice images of all split-product weights of a model in ONE call per step
    (alignn_prepare_weights: a memset + two launches instead of two launches per weight, 26 weights at the default
    model).  The image buffers and the descriptor table are persistent (rebuilt when a weight moved); every run files the
    images in the registries ``split_f16x2`` consults, stamped with the current weight generation - a later lookup by
    another forward (another generation) misses and slices for itself."""

    def __init__(self):
        self.sig, self.keep, self.refs, self.modules = None, [], [], None

    def _rebuild(self, weights, sig):
        import numpy as np

        lib = _lib.load()
        dev = weights[0].device
        nw = len(weights)
        self.amax = torch.zeros(nw, dtype=torch.float32, device=dev)
        self.images, self.keep = [], list(weights)
        rec = np.zeros(nw, dtype=np.dtype([("W", "<u8"), ("ldw", "<i8"), ("N", "<i4"), ("K", "<i4"), ("amax", "<u8"),
                                           ("out", "<u8"), ("outT", "<u8")]))
        assert rec.dtype.itemsize == 48
        for i, w in enumerate(weights):
            n, k = w.shape
            buf = torch.empty(lib.alignn_split_f16x2_bytes(n, k), dtype=torch.uint8, device=dev)
            buf_t = torch.empty(lib.alignn_split_f16x2_bytes(k, n), dtype=torch.uint8, device=dev)
            sw, sw_t = SplitWeight(buf, n, k), SplitWeight(buf_t, k, n)
            sw.amax = sw_t.amax = self.amax[i:i + 1]
            self.images.append((sw, sw_t))
            rec[i] = (w.data_ptr(), w.stride(0), n, k, sw.amax.data_ptr(), buf.data_ptr(), buf_t.data_ptr())
        self.desc = torch.from_numpy(rec.view(np.uint8)).to(dev)
        self.sig = sig

    def ensure(self, weights) -> bool:
        """Image buffers and the descriptor table for ``weights`` (rebuilt when one of them moved); False when the batched
        preparation does not apply to this list.  Launches nothing."""
        if not (BATCHED_WEIGHT_PREP and F16X3 and SPLIT_BOTH):
            return False
        sig = tuple([w.data_ptr() for w in weights])
        if sig != self.sig or len(weights) != len(self.keep) or any(a is not b for a, b in zip(weights, self.keep)):
            ok = [w for w in weights if (w.dim() == 2 and w.is_cuda and w.dtype == torch.float32 and w.stride(1) == 1
                                         and w.stride(0) % 4 == 0 and w.shape[0] % 16 == 0 and w.shape[1] % 16 == 0)]
            if len(ok) != len(weights) or not ok:
                self.sig = None
                return False
            self._rebuild(weights, sig)
            self.refs = [(id(w), weakref.ref(w, lambda _r, k_=id(w): (_W_IMG.pop(k_, None), _W_IMG_T.pop(k_, None),
                                                                      _W_AMAX.pop(k_, None)))) for w in weights]
        return True

    def run(self, weights):
        """``weights``: the same list (same tensor objects, same order) every step while nothing moved - the cheap path."""
        if not self.ensure(weights):
            return
        check(_lib.load().alignn_prepare_weights(ptr(self.desc), len(weights), ptr(self.amax), stream()), "prepare_weights")
        WEIGHT_PREP_STATS["runs"] += 1
        WEIGHT_PREP_STATS["weights"] += len(weights)
        gen = _WGEN[0]
        for w, (k_, ref), (sw, sw_t) in zip(weights, self.refs, self.images):
            st = (w._version, gen)
            _W_IMG[k_] = (ref, st, sw)
            _W_IMG_T[k_] = (ref, st, sw_t)
            _W_AMAX[k_] = (ref, st, sw.amax)


def split_f16x2(w, transpose=False):
    """Slice ``w * 2^s`` into two fp16 planes (``s`` from max|w|, kept with the image)."""
    lib = _lib.load()
    require_f32(w)
    n, k = (w.shape[1], w.shape[0]) if transpose else (w.shape[0], w.shape[1])
    if not transpose:
        img = _W_IMG.get(id(w))  # made ahead of this step by WeightPrep
        if img is not None and img[0]() is w and img[1] == _wstamp(w):
            return img[2]
    if transpose:
        # the image of w^T was made together with the forward image of the same weight version (below)
        img = _W_IMG_T.get(id(w))
        if img is not None and img[0]() is w and img[1] == _wstamp(w):
            REGISTRY_STATS["wimg_hit"] += 1
            return img[2]
        REGISTRY_STATS["wimg_miss"] += 1
    elif (SPLIT_BOTH and w.dim() == 2 and w.stride(1) == 1 and w.stride(0) % 4 == 0 and w.shape[0] % 16 == 0
          and w.shape[1] % 16 == 0):
        # forward product: max|w| into an arena slot (no reset launch), then this image and the input-gradient image in
        # ONE launch - two launches per weight and step instead of four (autograd.Function.forward runs with grad mode
        # off, so "will there be a way back" is not known here; without one the second image is a few microseconds
        # inside the same launch).  As with the cached maximum below, only the way BACK reuses anything - every
        # forward measures and slices afresh.
        amax = absmax(w, arena=True)
        buf = torch.empty(lib.alignn_split_f16x2_bytes(n, k), dtype=torch.uint8, device=w.device)
        buf_t = torch.empty(lib.alignn_split_f16x2_bytes(k, n), dtype=torch.uint8, device=w.device)
        check(lib.alignn_split_f16x2_both(ptr(w), w.stride(0), n, k, ptr(amax), ptr(buf), ptr(buf_t), stream()), "split_f16x2_both")
        sw, sw_t = SplitWeight(buf, n, k), SplitWeight(buf_t, k, n)
        sw.amax = sw_t.amax = amax
        k_ = id(w)
        _W_IMG_T[k_] = (weakref.ref(w, lambda _r, k_=k_: _W_IMG_T.pop(k_, None)), _wstamp(w), sw_t)
        _W_AMAX[k_] = (weakref.ref(w, lambda _r, k_=k_: _W_AMAX.pop(k_, None)), _wstamp(w), amax)
        return sw
    # max|w| is shared by the forward (W) and the input-gradient (W^T) images of one step: cache it per weight stamp
    # (version + generation, see _WGEN)
    hit = _W_AMAX.get(id(w))
    # reuse only on the way BACK (transpose=True: the input-gradient image of the weight the forward just sliced) -
    # every forward measures afresh, so an in-place edit that bypasses the version counter (w.data.mul_) between
    # steps can never meet a stale maximum
    if transpose and hit is not None and hit[0]() is w and hit[1] == _wstamp(w):
        amax = hit[2]
    else:
        amax = absmax(w if w.stride(0) % 4 == 0 and w.shape[1] % 4 == 0 else w.contiguous().view(1, -1))
        k_ = id(w)
        _W_AMAX[k_] = (weakref.ref(w, lambda _r, k_=k_: _W_AMAX.pop(k_, None)), _wstamp(w), amax)
    buf = torch.empty(lib.alignn_split_f16x2_bytes(n, k), dtype=torch.uint8, device=w.device)
    check(lib.alignn_split_f16x2(ptr(w), w.stride(0), n, k, int(transpose), ptr(amax), ptr(buf), stream()), "split_f16x2")
    sw = SplitWeight(buf, n, k)
    sw.amax = amax
    return sw


# Force training (alignn_amd/ff2.py): while this is a dict, the LayerNorm-flavoured layers file what their forward computed
# under the address of their weight - MLPLayerFn: ("mlp", x, pre, y); EdgeGatedConvFn: ("conv", x, y, P, M, xpre, s0, hh,
# x_out, y_out) - so that the dual pass that follows the force evaluation computes TANGENTS only instead of repeating the values
FORWARD_TAPE = None

KERNEL_TIMER = None  # bench.py: {"min_rows": r, "events": []} -> HIP events around every f16x3 NT launch of >= r rows


def _timed(label, M, N, K, launch):
    """Run ``launch()``; with bench.py's KERNEL_TIMER armed and M large enough, bracket it with HIP events."""
    t = KERNEL_TIMER
    if t is None or M < t["min_rows"]:
        return launch()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    try:
        return launch()
    finally:
        ev1.record()
        t["events"].append((label, N, K, ev0, ev1))


def gemm_nt_f16x3(a, a_amax, ws, bias=None, addend=None, out=None):
    """out[M,N] = a[M,K] @ W[N,K]^T with W pre-sliced by ``split_f16x2`` and ``a_amax`` >= max|a| (device scalar)."""
    lib = _lib.load()
    require_f32(a, bias, addend)
    M, K = a.shape
    N = ws.n
    if K != ws.k:
        raise ValueError(f"reduction length mismatch: {K} vs {ws.k}")
    if out is None:
        out = _empty(M, N, like=a)
    return _timed("addend" if addend is not None else "plain", M, N, K,
                  lambda: _gemm_nt_f16x3_launch(lib, a, a_amax, ws, bias, addend, out, M, N, K))


def _gemm_nt_f16x3_launch(lib, a, a_amax, ws, bias, addend, out, M, N, K):
    check(
        lib.alignn_gemm_nt_f16x3(ptr(a), a.stride(0), ptr(a_amax), ptr(ws.buf), ptr(ws.amax), ptr(bias), ptr(addend),
                                 addend.stride(0) if addend is not None else 0, ptr(out), out.stride(0), M, N, K,
                                 stream()),
        "gemm_nt_f16x3",
    )
    return out


def gemm_nt_f16x3_bnred(a, a_amax, ws, xn, nstat, bias=None, addend=None, out=None):
    """``gemm_nt_f16x3`` whose output is the gradient of ``r + silu(BatchNorm(xn))``: also returns ``red`` [2,N] =
    (sum gz, sum gz*xhat) over the rows - the reductions BatchNorm's backward starts with (``_bn_silu_bwd_reduce``) -
    taken in the product's epilogue while the tile is still in registers (one pass over ``xn`` instead of two passes
    over ``xn`` and the gradient)."""
    lib = _lib.load()
    require_f32(a, bias, addend, xn, nstat)
    M, K = a.shape
    N = ws.n
    if K != ws.k or xn.shape != (M, N):
        raise ValueError(f"shape mismatch: a {tuple(a.shape)}, W [{N},{ws.k}], xn {tuple(xn.shape)}")
    if out is None:
        out = _empty(M, N, like=a)
    tiles = lib.alignn_gemm_nt_x6_row_tiles(M, N, K)
    partial = _empty(tiles + 1, 2, N, like=a)  # (+ the kernel's scratch slab)
    _timed("bnred_addend" if addend is not None else "bnred", M, N, K, lambda: check(
        lib.alignn_gemm_nt_f16x3_bnred(ptr(a), a.stride(0), ptr(a_amax), ptr(ws.buf), ptr(ws.amax), ptr(bias), ptr(addend),
                                       addend.stride(0) if addend is not None else 0, ptr(out), out.stride(0), M, N, K,
                                       ptr(xn), xn.stride(0), ptr(nstat), ptr(partial), stream()),
        "gemm_nt_f16x3_bnred",
    ))
    red = _empty(2, N, like=a)
    partial, tiles = _fold(partial, tiles, 2 * N)
    check(lib.alignn_bn_bwd_finalize(ptr(partial), tiles, N, ptr(red), stream()), "bn_bwd_finalize")
    return out, red


DGRAD_WGRAD_FUSED = _os.environ.get("ALIGNN_AMD_DW_FUSED", "1") != "0"  # csrc/gemm_dw.hip (tests / A-B runs flip it)
DW_MIN_ROWS = int(_os.environ.get("ALIGNN_AMD_DW_MIN_ROWS", "65536"))  # edge rows from which a convolution takes it
DW_STATS = {"fused": 0}


def dgrad_wgrad_applies(M, N, K, g_amax, y_amax):
    """Would the backward of a ``[M, K] -> [M, N]`` Linear run as ONE pass over its output gradient (csrc/gemm_dw.hip)?"""
    return bool(DGRAD_WGRAD_FUSED and M >= DW_MIN_ROWS and F16X3 and g_amax is not None and y_amax is not None
                and _lib.load().alignn_gemm_dgrad_wgrad_supported(M, N, K))


def gemm_dgrad_wgrad(gm, g_amax, y, y_amax, wt, addend=None, xn=None, nstat=None, out=None):
    """Backward of ``m = y W^T`` in one pass over ``gm`` = dL/dm [M, 256]: -> (g_y = gm W (+ addend), dW = gm^T y, red or None).
    ``wt`` = ``split_f16x2(W, True)``.  With (xn, nstat) the BatchNorm-backward sums of g_y against the pre-activation ``xn`` come
    along as ``red`` [2, 256] (see ``gemm_nt_f16x3_bnred``).  Replaces grad_input + grad_weight of the edge_gate Linear
    (alignn/models/alignn.py:101), which each streamed ``gm`` from HBM."""
    lib = _lib.load()
    require_f32(gm, y, addend, xn, nstat)
    M, N = gm.shape
    K = y.shape[1]
    if y.shape[0] != M or wt.n != K or wt.k != N:
        raise ValueError(f"shape mismatch: gm {tuple(gm.shape)}, y {tuple(y.shape)}, W^T image [{wt.n},{wt.k}]")
    if out is None:
        out = _empty(M, K, like=gm)
    dW = _empty(N, K, like=gm)
    slabs = lib.alignn_gemm_dgrad_wgrad_slabs(M)
    partial = _empty(2 * slabs, 2, K, like=gm) if xn is not None else None
    nbytes = lib.alignn_gemm_dgrad_wgrad_workspace(M)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=gm.device)
    label = ("dw_bnred" if xn is not None else "dw") + ("_addend" if addend is not None else "")
    _timed(label, M, K, N, lambda: check(
        lib.alignn_gemm_dgrad_wgrad_f16x3(ptr(gm), gm.stride(0), ptr(g_amax), ptr(y), y.stride(0), ptr(y_amax), ptr(wt.buf),
                                          ptr(wt.amax), ptr(addend), addend.stride(0) if addend is not None else 0, ptr(out),
                                          out.stride(0), ptr(xn), xn.stride(0) if xn is not None else 0, ptr(nstat), ptr(partial),
                                          ptr(dW), dW.stride(0), M, ptr(ws), nbytes, stream()),
        "gemm_dgrad_wgrad_f16x3",
    ))
    DW_STATS["fused"] += 1
    red = None
    if xn is not None:
        red = _empty(2, K, like=gm)
        check(lib.alignn_bn_bwd_finalize(ptr(partial), 2 * slabs, K, ptr(red), stream()), "bn_bwd_finalize")
    return out, dW, red


GATHER_FUSED = True  # EdgeGatedGraphConv: u_add_v folded into the edge-gate projection (tests flip it to compare: same bits)
STATS_FUSED = True  # BatchNorm statistics from the epilogue of the projection that writes the tensor (tests flip it)


def _f16x3_applies(a, a_amax, N, K):
    """Would ``project`` run this product on the three-product fp16 kernel?"""
    lib = _lib.load()
    M = a.shape[0]
    return (F16X3 and a_amax is not None and a.stride(0) % 4 == 0 and ((M + 63) // 64) * ((N + 255) // 256) >= X6_MIN_TILES
            and bool(lib.alignn_gemm_nt_x6_supported(M, N, K)))


BD_SEGMENT_TABLE = True  # line graphs: destination term from a segment-ordered copy (tests flip it: same bits)
BD_TABLE_STATS = {"used": 0}  # edge-gate projections that read the table (tests)


def segment_ordered_bd(P, graph, H):
    """The Bd block of the node projection P [n, 4H] with its rows in SEGMENT order (row s = Bd[seg_node[s]]) - the table
    ``gemm_nt_f16x3_gather(..., bd2=, rank=)`` reads: consecutive edge rows then gather consecutive table rows.  None when the
    graph's segments are in node order already (bond graphs) or the switch is off."""
    if not BD_SEGMENT_TABLE or graph.seg_node is None or graph.seg_rank is None:
        return None
    lib = _lib.load()
    n = P.shape[0]
    bd2 = _empty(n, H, like=P)
    BD_TABLE_STATS["used"] += 1
    check(lib.alignn_gather_rows_ld(P.data_ptr() + 4 * H, P.stride(0), ptr(graph.seg_node), ptr(bd2), H, n, H, stream()),
          "gather_rows_ld")
    return bd2


def gemm_nt_f16x3_gather(a, a_amax, ws, bias, P, src, dst, out=None, want_stats=False, bd2=None, rank=None):
    """out[e] = a[e] @ W^T + bias + P[src[e], 0:N] + P[dst[e], N:2N]  (edge-gate projection + DGL u_add_v in one pass).
    ``bd2`` / ``rank``: the second term as bd2[rank[e]] instead (``segment_ordered_bd``: same values, cache-friendly order).
    ``want_stats``: -> (out, partial [tiles,2,N], tiles): per-tile column sums of out and out^2 (alignn_bn_finalize's slabs)."""
    lib = _lib.load()
    require_f32(a, bias, P, bd2)
    M, K = a.shape
    N = ws.n
    if out is None:
        out = _empty(M, N, like=a)
    tiles = lib.alignn_gemm_nt_x6_row_tiles(M, N, K) if want_stats else 0
    partial = _empty(tiles + 1, 2, N, like=a) if want_stats else None  # (+ the kernel's scratch slab)
    if bd2 is not None:
        _timed("gather", M, N, K, lambda: check(
            lib.alignn_gemm_nt_f16x3_gather2(ptr(a), a.stride(0), ptr(a_amax), ptr(ws.buf), ptr(ws.amax), ptr(bias), ptr(out),
                                             out.stride(0), M, N, K, ptr(P), P.stride(0), ptr(src), ptr(bd2), bd2.stride(0),
                                             ptr(rank), ptr(partial), stream()),
            "gemm_nt_f16x3_gather2",
        ))
        return (out, partial, tiles) if want_stats else out
    _timed("gather", M, N, K, lambda: check(
        lib.alignn_gemm_nt_f16x3_gather(ptr(a), a.stride(0), ptr(a_amax), ptr(ws.buf), ptr(ws.amax), ptr(bias), ptr(out),
                                        out.stride(0), M, N, K, ptr(P), P.stride(0), ptr(src), ptr(dst), ptr(partial),
                                        stream()),
        "gemm_nt_f16x3_gather",
    ))
    return (out, partial, tiles) if want_stats else out


def gemm_nt_f16x3_stats(a, a_amax, ws, bias=None, out=None):
    """-> (out = a @ W^T + bias, partial [tiles,2,N], tiles): the projection with the column sums of its output and of its
    square per row tile - the BatchNorm statistics of ``out`` without another pass over it."""
    lib = _lib.load()
    require_f32(a, bias)
    M, K = a.shape
    N = ws.n
    if out is None:
        out = _empty(M, N, like=a)
    tiles = lib.alignn_gemm_nt_x6_row_tiles(M, N, K)
    partial = _empty(tiles + 1, 2, N, like=a)  # (+ the kernel's scratch slab)
    _timed("stats", M, N, K, lambda: check(
        lib.alignn_gemm_nt_f16x3_stats(ptr(a), a.stride(0), ptr(a_amax), ptr(ws.buf), ptr(ws.amax), ptr(bias), ptr(out),
                                       out.stride(0), M, N, K, ptr(partial), stream()),
        "gemm_nt_f16x3_stats",
    ))
    return out, partial, tiles


def gemm_nt_x6(a, ws, bias=None, addend=None, out=None):
    """out[M,N] = a[M,K] @ W[N,K]^T with W pre-sliced by ``split_bf16x3`` (fp32-grade accuracy, bf16 MFMA)."""
    lib = _lib.load()
    require_f32(a, bias, addend)
    M, K = a.shape
    N = ws.n
    if K != ws.k:
        raise ValueError(f"reduction length mismatch: {K} vs {ws.k}")
    if out is None:
        out = _empty(M, N, like=a)
    check(
        lib.alignn_gemm_nt_x6(ptr(a), a.stride(0), ptr(ws.buf), ptr(bias), ptr(addend),
                              addend.stride(0) if addend is not None else 0, ptr(out), out.stride(0), M, N, K, stream()),
        "gemm_nt_x6",
    )
    return out


NN_SPLIT = True  # split-reduction input gradients on atom rows (tests flip it)


def project(a, w, bias=None, addend=None, transpose_w=False, a_amax=None):
    """a @ W^T (transpose_w: a @ W) choosing the kernel: the split-product kernels for the wide, deep products
    that dominate the step (three fp16 products when max|a| is known - ``a_amax`` or the producer registry -, six
    bf16 products otherwise), the exact-fp32 MFMA kernel for the rest."""
    lib = _lib.load()
    M = a.shape[0]
    N, K = (w.shape[1], w.shape[0]) if transpose_w else (w.shape[0], w.shape[1])
    # the split-product kernels work in 128 x 256 (short or shallow products: 64 x 256) tiles and need enough of them
    x6_tiles = ((M + 63) // 64) * ((N + 255) // 256)
    if x6_tiles >= X6_MIN_TILES and a.stride(0) % 4 == 0 and lib.alignn_gemm_nt_x6_supported(M, N, K):
        if a_amax is None:
            a_amax = get_amax(a)
        if F16X3 and a_amax is not None:
            return gemm_nt_f16x3(a, a_amax, split_f16x2(w, transpose_w), bias, addend)
        if F16X3:
            REGISTRY_STATS["bf16x6_fallback"] += 1
            _warn_once("bf16x6", f"alignn_amd: a [{M},{K}] x [{K},{N}] projection runs the six-product bf16 scheme because max|a| "
                                 "of its input is unknown (the tensor did not come from a kernel that tracks it, or was cloned / "
                                 "edited in between): correct, but ~1.6x slower than the three-product fp16 scheme")
        return gemm_nt_x6(a, split_bf16x3(w, transpose_w), bias, addend)
    if transpose_w:
        if NN_SPLIT and bias is None and lib.alignn_gemm_nn_split_workspace(M, w.shape[0], w.shape[1]):
            return gemm_nn(a, w, addend)  # few output tiles, long reduction: reduction slabs side by side
        if w.shape[0] % 4 == 0 and w.shape[1] >= 16:
            return gemm_nt(a, w.t().contiguous(), bias, addend)
        return gemm_nn(a, w, addend)
    return gemm_nt(a, w, bias, addend)


def gemm_nn(g, w, addend=None, out=None):
    """out[M,K] = g[M,N] @ w[N,K] (+addend)."""
    lib = _lib.load()
    require_f32(g, w, addend)
    M, N = g.shape
    K = w.shape[1]
    if out is None:
        out = _empty(M, K, like=g)
    nbytes = lib.alignn_gemm_nn_split_workspace(M, N, K) if NN_SPLIT else 0
    if nbytes and g.stride(0) % 4 == 0 and w.stride(0) % 4 == 0 and (addend is None or addend.stride(0) % 4 == 0):
        # few output tiles, long reduction (atom rows x [4H, H]): reduction slabs side by side + one fixed-order sum
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=g.device)
        check(lib.alignn_gemm_nn_split(ptr(g), g.stride(0), ptr(w), w.stride(0), ptr(addend),
                                       addend.stride(0) if addend is not None else 0, ptr(out), out.stride(0), M, N, K,
                                       ptr(ws), nbytes, stream()), "gemm_nn_split")
        return out
    check(
        lib.alignn_gemm_nn(ptr(g), g.stride(0), ptr(w), w.stride(0), ptr(addend),
                           addend.stride(0) if addend is not None else 0, ptr(out), out.stride(0), M, N, K, stream()),
        "gemm_nn",
    )
    return out


def _dgrad(g, w, addend=None, g_amax=None):
    """Input gradient g[M,N] @ w[N,K] through the reduction-contiguous (NT) kernels on w^T."""
    return project(g, w, None, addend, transpose_w=True, a_amax=g_amax)


# BatchNorm-backward reductions taken in the epilogue of the projection that PRODUCES the gradient (see
# alignn_gemm_nt_f16x3_bnred).  A layer whose output is y = r + silu(BatchNorm(xn)) registers (xn, stat) under y; the layer
# that consumes y looks the record up in its forward and, in its backward, lets the input-gradient projection of y also
# reduce gz / gz*xhat over the rows; the result travels to the producer's backward under the identity of the gradient
# tensor.  Any mismatch (another consumer of y, gradients summed by autograd, a different kernel path) simply finds
# no record and the producer reduces on its own.
BNRED_FUSED = True
BNRED_STATS = {"fused": 0, "used": 0}  # (tests: how often the fused reduction was produced / consumed)
_NORM_SRC = {}  # id(y)  -> (weakref(y), xn, stat)
_PRE_RED = {}  # id(gy) -> (weakref(gy), xn, red)


def _register_norm_src(y, xn, stat):
    if y is not None and BNRED_FUSED:
        k = id(y)
        _NORM_SRC[k] = (weakref.ref(y, lambda _r, k=k: _NORM_SRC.pop(k, None)), xn, stat)


def _norm_src_of(y):
    e = _NORM_SRC.get(id(y))
    hit = e is not None and e[0]() is y
    if BNRED_FUSED:
        REGISTRY_STATS["norm_src_hit" if hit else "norm_src_miss"] += 1
    return (e[1], e[2]) if hit else None


def _take_pre_red(gy, xn):
    if gy is None:
        return None
    e = _PRE_RED.pop(id(gy), None)
    if e is not None and e[0]() is gy and e[1] is xn:
        BNRED_STATS["used"] += 1
        REGISTRY_STATS["pre_red_hit"] += 1
        return e[2]
    REGISTRY_STATS["pre_red_miss"] += 1
    # a record for THIS pre-activation whose gradient tensor is gone, or filed under another gradient tensor: the gradient
    # was cloned / re-wrapped between the two autograd nodes (a hook, checkpointing, a bucket view) - the fused reduction
    # was paid for and is lost (NOT reused: the hook may have changed the values)
    lost = _PRE_RED_ORPHANS.pop(id(xn), None) is not None
    for k, rec in list(_PRE_RED.items()):
        if rec[1] is xn:
            _PRE_RED.pop(k, None)
            lost = True
    if lost:
        REGISTRY_STATS["separate_bn_reduce"] += 1
        _warn_once("pre_red", "alignn_amd: BatchNorm-backward sums taken in a projection's epilogue were not picked up by the "
                              "layer they belong to (its incoming gradient is a different tensor object than the one the "
                              "projection wrote: hook / clone / checkpoint in between?) - reducing again in a separate pass")
    return None


_PRE_RED_ORPHANS = {}  # id(xn) -> weakref(xn): records whose gradient tensor died before its producer's backward asked


def _orphan_pre_red(k):
    rec = _PRE_RED.pop(k, None)
    if rec is not None:
        xn = rec[1]
        kx = id(xn)
        _PRE_RED_ORPHANS[kx] = weakref.ref(xn, lambda _r, kx=kx: _PRE_RED_ORPHANS.pop(kx, None))


def _dgrad_bnred(g, w, addend, g_amax, src):
    """``_dgrad`` that also leaves the BatchNorm-backward reductions of ``src = (xn, stat)`` for the producer of the
    tensor this gradient belongs to - when the product takes the f16x3 kernel; otherwise plain ``_dgrad``."""
    lib = _lib.load()
    M, N, K = g.shape[0], w.shape[1], w.shape[0]
    ok = (src is not None and BNRED_FUSED and F16X3 and g_amax is not None and g.stride(0) % 4 == 0
          and ((M + 63) // 64) * ((N + 255) // 256) >= X6_MIN_TILES and lib.alignn_gemm_nt_x6_supported(M, N, K)
          and tuple(src[0].shape) == (M, N) and src[0].stride(0) % 4 == 0)
    if not ok:
        return _dgrad(g, w, addend, g_amax)
    out, red = gemm_nt_f16x3_bnred(g, g_amax, split_f16x2(w, True), src[0], src[1], None, addend)
    BNRED_STATS["fused"] += 1
    k = id(out)
    _PRE_RED[k] = (weakref.ref(out, lambda _r, k=k: _orphan_pre_red(k)), src[0], red)
    return out


def gemm_tn(g, a, g_amax=None, a_amax=None):
    """dW[N,K] = g[M,N]^T @ a[M,K] (deterministic split over M).  With both maxima known (device scalars) the large
    aligned shapes run the three-product fp16 scheme."""
    if not F16X3 or g_amax is None or a_amax is None:
        g_amax = a_amax = None
    lib = _lib.load()
    require_f32(g, a)
    M, N = g.shape
    K = a.shape[1]
    out = _empty(N, K, like=g)
    nbytes = lib.alignn_gemm_tn_workspace(M, N, K)
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=g.device)
    check(
        lib.alignn_gemm_tn(ptr(g), g.stride(0), ptr(g_amax), ptr(a), a.stride(0), ptr(a_amax), ptr(out), out.stride(0),
                           M, N, K, ptr(ws), nbytes, stream()),
        "gemm_tn",
    )
    return out


def col_sum(x):
    lib = _lib.load()
    rows, F = x.shape
    out = _empty(F, like=x)
    ws = _empty(lib.alignn_col_stats_slabs(rows) * 2 * F, like=x)
    check(lib.alignn_col_sum(ptr(x), x.stride(0), rows, F, ptr(out), ptr(ws), stream()), "col_sum")
    return out


FOLD_ABOVE = 1024  # more slabs than this (one per row tile of a T-row projection): fold to 64 first


def _fold(partial, slabs, width):
    """(partial, slabs) with at most FOLD_ABOVE slabs: many-slab inputs are pre-summed to 64 slabs (alignn_slab_fold)."""
    if slabs <= FOLD_ABOVE:
        return partial, slabs
    lib = _lib.load()
    g = lib.alignn_slab_fold_slabs()
    out = _empty(g, width, like=partial)
    check(lib.alignn_slab_fold(ptr(partial), slabs, width, ptr(out), stream()), "slab_fold")
    return out, g


def _welford_slabs(slabs, F, like):
    """Storage for ``slabs`` pivot slabs [3][F] + their counts (alignn_col_stats_welford, the gate passes)."""
    return _empty(slabs * (3 * F + 1), like=like)


def _bn_finalize(partial, slabs, rows, gamma, beta, running_mean, running_var, update_running, welford=False):
    """-> stat [4,F] = mean, rstd, scale, shift.  slabs == 0: evaluation mode (running statistics).  ``welford``: the slabs
    are (sum, M2) + counts (well-conditioned: column statistics pass, gate passes) instead of plain sums."""
    lib = _lib.load()
    F = gamma.numel()
    stat = _empty(4, F, like=gamma)
    rm = running_mean if (update_running or slabs == 0) else None
    rv = running_var if (update_running or slabs == 0) else None
    if welford and slabs:
        check(lib.alignn_bn_finalize_welford(ptr(partial), slabs, rows, F, ptr(gamma), ptr(beta), BN_EPS, BN_MOMENTUM, ptr(rm),
                                             ptr(rv), ptr(stat), stream()), "bn_finalize_welford")
        return stat
    if slabs:
        partial, slabs = _fold(partial, slabs, 2 * F)
    check(
        lib.alignn_bn_finalize(ptr(partial), slabs, rows, F, ptr(gamma), ptr(beta), BN_EPS, BN_MOMENTUM, ptr(rm),
                               ptr(rv), ptr(stat), stream()),
        "bn_finalize",
    )
    return stat


def _bn_silu_fwd(x, res, stat):
    lib = _lib.load()
    rows, F = x.shape
    y = _empty(rows, F, like=x)
    amax = new_amax(x) if _track(rows) else None  # max|y|, tracked by the kernel for the next projection of y
    check(
        lib.alignn_bn_silu_fwd(ptr(x), x.stride(0), ptr(res), res.stride(0) if res is not None else 0, ptr(stat),
                               ptr(y), y.stride(0), rows, F, ptr(amax), stream()),
        "bn_silu_fwd",
    )
    return set_amax(y, amax) if amax is not None else y


def _bn_silu_bwd_reduce(gy, x, stat):
    """-> red [2,F]: sum gz (= dbeta), sum gz*xhat (= dgamma)."""
    lib = _lib.load()
    rows, F = x.shape
    slabs = lib.alignn_col_stats_slabs(rows)
    partial = _empty(slabs, 2, F, like=x)
    check(
        lib.alignn_bn_silu_bwd_reduce(ptr(gy), gy.stride(0), ptr(x), x.stride(0), ptr(stat), rows, F, ptr(partial),
                                      stream()),
        "bn_silu_bwd_reduce",
    )
    red = _empty(2, F, like=x)
    check(lib.alignn_bn_bwd_finalize(ptr(partial), slabs, F, ptr(red), stream()), "bn_bwd_finalize")
    return red


def _bn_silu_bwd_apply(gy, x, stat, gamma, red, eval_mode, out, amax=None):
    """``amax`` (optional device scalar, zeroed by the caller) is raised to max|out|."""
    lib = _lib.load()
    rows, F = x.shape
    check(
        lib.alignn_bn_silu_bwd_apply(ptr(gy), gy.stride(0), ptr(x), x.stride(0), ptr(stat), ptr(gamma), ptr(red),
                                     int(eval_mode), ptr(out), out.stride(0), rows, F, ptr(amax), stream()),
        "bn_silu_bwd_apply",
    )
    return out


LN_EPS = 1e-5


def _ln_silu_fwd(x, res, gamma, beta, want_stats=True):
    """Y = res + silu(LayerNorm(x)); returns (Y, stats[rows,2] = mean, rstd)."""
    lib = _lib.load()
    rows, F = x.shape
    y = _empty(rows, F, like=x)
    stats = _empty(rows, 2, like=x) if want_stats else None
    amax = new_amax(x) if _track(rows) else None
    check(
        lib.alignn_ln_silu_fwd(ptr(x), x.stride(0), ptr(res), res.stride(0) if res is not None else 0, ptr(gamma),
                               ptr(beta), LN_EPS, ptr(y), y.stride(0), ptr(stats), rows, F, ptr(amax), stream()),
        "ln_silu_fwd",
    )
    if amax is not None:
        set_amax(y, amax)
    return y, stats


def _ln_silu_bwd(gy, x, gamma, beta, stats, out, amax=None, node=None):
    """LayerNorm/SiLU backward into ``out`` (``amax``: raised to max|out|); returns red [2,F] = (dbeta, dgamma).
    ``node`` = (S0, HH, GS1, GS0): ``x`` is the node pre-activation of an edge-gated convolution - the adjoints of its two segment
    sums are written in the same pass (alignn_ln_silu_bwd_node)."""
    lib = _lib.load()
    rows, F = x.shape
    slabs = lib.alignn_ln_slabs(rows)
    partial = _empty(slabs, 2, F, like=x)
    if node is not None:
        s0, hh, gs1, gs0 = node
        check(
            lib.alignn_ln_silu_bwd_node(ptr(gy), gy.stride(0), ptr(x), x.stride(0), ptr(gamma), ptr(beta), ptr(stats), ptr(out),
                                        out.stride(0), ptr(partial), rows, F, ptr(amax), ptr(s0), ptr(hh), ptr(gs1), ptr(gs0),
                                        stream()),
            "ln_silu_bwd_node",
        )
    else:
        check(
            lib.alignn_ln_silu_bwd(ptr(gy), gy.stride(0), ptr(x), x.stride(0), ptr(gamma), ptr(beta), ptr(stats), ptr(out),
                                   out.stride(0), ptr(partial), rows, F, ptr(amax), stream()),
            "ln_silu_bwd",
        )
    red = _empty(2, F, like=x)
    check(lib.alignn_bn_bwd_finalize(ptr(partial), slabs, F, ptr(red), stream()), "ln_bwd_finalize")
    return red


def _bond_cosines_raw(r, e1, e2):
    lib = _lib.load()
    require_f32(r)
    h = _empty(e1.numel(), like=r)
    check(lib.alignn_bond_cosine_fwd(ptr(r), ptr(e1), ptr(e2), ptr(h), e1.numel(), stream()), "bond_cosine_fwd")
    return h


def _segment_sum_raw(vals, seg_ptr, slot, node, n_out):
    """out[node[s] or s] = sum of vals[slot[k] or k] over k in segment s (any width; fixed order)."""
    lib = _lib.load()
    vals = vals.contiguous()
    out = _empty(n_out, vals.shape[1], like=vals)
    check(lib.alignn_segment_sum(ptr(vals), vals.stride(0), ptr(seg_ptr), ptr(slot), ptr(node), ptr(out), out.stride(0),
                                 seg_ptr.numel() - 1, vals.shape[1], stream()), "segment_sum")
    return out


class BondCosFn(torch.autograd.Function):
    """compute_bond_cosines with its FIRST derivative w.r.t. the bond vectors (force-field inference); the training
    path that differentiates through the forces uses alignn_amd.ff.bond_cosines instead."""

    @staticmethod
    def forward(ctx, r, lg):
        r = r.contiguous()
        ctx.save_for_backward(r)
        ctx.lg = lg
        return _bond_cosines_raw(r, lg.src, lg.dst)

    @staticmethod
    def backward(ctx, gh):
        lib = _lib.load()
        (r,) = ctx.saved_tensors
        lg = ctx.lg
        T = lg.n_edges
        ga, gb = _empty(T, 3, like=r), _empty(T, 3, like=r)
        check(lib.alignn_bond_cosine_bwd(ptr(r), ptr(lg.src), ptr(lg.dst), ptr(gh.contiguous()), ptr(ga), ptr(gb), T,
                                         stream()), "bond_cosine_bwd")
        # dh/dr[e]: the triplets where e is the first bond (by source of L(g)) + those where it is the second (by dst)
        gr = _segment_sum_raw(ga, lg.out_ptr, lg.out_slot, None, lg.n_nodes)
        gr = gr + _segment_sum_raw(gb, lg.seg_ptr, None, lg.seg_node, lg.n_nodes)
        return gr, None


def bond_cosines(r, e1, e2=None):
    """compute_bond_cosines on the canonical line graph (alignn/graphs.py:847-864).  ``bond_cosines(r, lg)`` with a
    CSRGraph is differentiable w.r.t. ``r`` (first order); the index form ``(r, e1, e2)`` is forward only."""
    if isinstance(e1, CSRGraph):
        if r.requires_grad:
            return BondCosFn.apply(r, e1)
        return _bond_cosines_raw(r.contiguous(), e1.src, e1.dst)
    if r.requires_grad:
        raise NotImplementedError("pass the line graph itself (bond_cosines(r, lg)) to differentiate w.r.t. r")
    return _bond_cosines_raw(r.contiguous(), e1, e2)


# ---------------------------------------------------------------------------------------------
# Linear
# ---------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """nn.Linear on the MFMA GEMMs (alignn/models/alignn.py:341 ``fc`` and stand-alone projections)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous()
        w = w.contiguous()
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        ctx.param_grads = _PARAM_GRADS["on"]
        return project(x, w, b)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        gx = _dgrad(gy, w) if ctx.needs_input_grad[0] else None
        gw = gemm_tn(gy, x) if (ctx.needs_input_grad[1] and ctx.param_grads) else None
        gb = None
        if ctx.has_bias and ctx.needs_input_grad[2] and ctx.param_grads:
            if gy.shape[1] % 4 == 0:
                gb = col_sum(gy)
            else:  # e.g. the 1-wide readout fc: gb[n] = (gy^T @ ones)[n]
                ones = torch.ones(gy.shape[0], 1, dtype=torch.float32, device=gy.device)
                gb = gemm_tn(gy, ones).reshape(-1)
        return gx, gw, gb


def linear(x, w, b=None):
    return LinearFn.apply(x, w, b)


# ---------------------------------------------------------------------------------------------
# MLPLayer = Linear + BatchNorm1d + SiLU   (alignn/models/alignn.py:170-184)
# ---------------------------------------------------------------------------------------------
APPLY_SUM = True  # bias gradient from the norm-backward pass (tests flip it)


class MLPLayerFn(torch.autograd.Function):
    """``norm`` = "batch" (alignn/models/alignn.py:170-184) or "layer" (alignn/models/utils.py:277-292).  T-row layers
    (the angle embedding) run on lane T inside ``lanes()``."""

    @staticmethod
    def _fwd(ctx, x, w, b, gamma, beta, running_mean, running_var, training, norm):
        lib = _lib.load()
        ctx.x_amax = get_amax(x)
        fused_stats = (norm == "batch" and training and STATS_FUSED
                       and _f16x3_applies(x, ctx.x_amax, w.shape[0], w.shape[1]))
        if fused_stats:  # the projection's epilogue leaves the column sums BatchNorm needs
            pre, partial, slabs = gemm_nt_f16x3_stats(x, ctx.x_amax, split_f16x2(w), b)
        else:
            pre = project(x, w, b, a_amax=ctx.x_amax)
        rows, F = pre.shape
        if norm == "layer":
            y, stat = _ln_silu_fwd(pre, None, gamma, beta)
        else:
            if training:
                if not fused_stats:
                    slabs = lib.alignn_col_stats_slabs(rows)
                    partial = _welford_slabs(slabs, F, pre)
                    check(lib.alignn_col_stats_welford(ptr(pre), pre.stride(0), rows, F, ptr(partial), stream()), "col_stats")
                stat = _bn_finalize(partial, slabs, rows, gamma, beta, running_mean, running_var, True, welford=not fused_stats)
            else:
                stat = _bn_finalize(None, 0, rows, gamma, beta, running_mean, running_var, False)
            y = _bn_silu_fwd(pre, None, stat)
            _register_norm_src(y, pre, stat)
        return y, pre, stat

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, running_mean, running_var, training, norm="batch"):
        x = x.contiguous()
        w = w.contiguous()
        lane = _lane_for(x.shape[0])
        ctx.lane = lane is not None
        ctx.x_on_T = _is_on_T(x)  # then x's producer is lane-aware and takes its gradient on lane T
        ctx.x_src = _norm_src_of(x)  # x = silu(BatchNorm(.)) of the previous layer: see _dgrad_bnred
        if lane is not None:
            with _on_T(*lane, reads=(x,)):
                y, pre, stat = MLPLayerFn._fwd(ctx, x, w, b, gamma, beta, running_mean, running_var, training, norm)
            _mark_on_T(y)
        else:
            _main_reads(x)
            y, pre, stat = MLPLayerFn._fwd(ctx, x, w, b, gamma, beta, running_mean, running_var, training, norm)
        ctx.save_for_backward(x, w, pre, stat, gamma, beta)
        ctx.training = training
        ctx.norm = norm
        ctx.param_grads = _PARAM_GRADS["on"]
        ctx.wb = (w, b)  # the leaves whose gradients come off the side stream (see _deferred_join_is_safe)
        _note_forward(ctx.wb)
        if FORWARD_TAPE is not None and norm == "layer":
            FORWARD_TAPE[w.data_ptr()] = ("mlp", x, pre, y)
        return y

    @staticmethod
    def _bwd(ctx, gy, x, w, pre, stat, gamma, beta):
        gpre = torch.empty_like(pre)
        gb_part = None
        g_amax = new_amax(pre) if _track(pre.shape[0]) else None
        if ctx.norm == "layer":
            red = _ln_silu_bwd(gy, pre, gamma, beta, stat, gpre, g_amax)
        else:
            red = _take_pre_red(gy, pre)
            if red is None:
                red = _bn_silu_bwd_reduce(gy, pre, stat)
            if APPLY_SUM and getattr(ctx, "param_grads", False) and pre.shape[1] <= 1024:
                # the pass that writes gpre also sums its columns (the Linear's bias gradient): no col_sum pass over gpre
                lib = _lib.load()
                rows, F = pre.shape
                slabs = lib.alignn_col_stats_slabs(rows)
                gb_part = _empty(slabs, F, like=pre)
                check(lib.alignn_bn_silu_bwd_apply_sum(ptr(gy), gy.stride(0), ptr(pre), pre.stride(0), ptr(stat), ptr(red),
                                                       int(not ctx.training), ptr(gpre), gpre.stride(0), rows, F, ptr(g_amax),
                                                       ptr(gb_part), stream()), "bn_silu_bwd_apply_sum")
                gb_part = (gb_part, slabs)
            else:
                _bn_silu_bwd_apply(gy, pre, stat, gamma, red, not ctx.training, gpre, g_amax)
        gx = _dgrad_bnred(gpre, w, None, g_amax, ctx.x_src) if ctx.needs_input_grad[0] else None
        return gpre, g_amax, red, gx, gb_part

    @staticmethod
    def backward(ctx, gy):
        x, w, pre, stat, gamma, beta = ctx.saved_tensors
        gy = gy.contiguous()
        ev = None
        if ctx.lane:
            main, T = _lane_streams(gy.device)
            with _on_T(main, T, reads=(gy,)):
                gpre, g_amax, red, gx, gb_part = MLPLayerFn._bwd(ctx, gy, x, w, pre, stat, gamma, beta)
            ev = _event_after(T)
            # dgamma / dbeta (``red``) come off lane T: same rule as the side-stream gradients; the input gradient stays
            # on lane T only for a producer that is lane-aware itself
            if (gx is None or ctx.x_on_T) and (not ctx.param_grads or _deferred_join_is_safe((gamma, beta))):
                _arm_backward_join()
                _mark_on_T(gx)
            else:
                main.wait_event(ev)
                for t in (gx, red):
                    if t is not None:
                        t.record_stream(main)
        else:
            _main_reads(gy)
            gpre, g_amax, red, gx, gb_part = MLPLayerFn._bwd(ctx, gy, x, w, pre, stat, gamma, beta)
        dbeta, dgamma = red[0], red[1]
        if not ctx.param_grads:
            return gx, None, None, None, None, None, None, None, None
        x_amax = ctx.x_amax
        def bias_grad():
            if gb_part is None:
                return col_sum(gpre)
            out = _empty(gpre.shape[1], like=gpre)
            check(_lib.load().alignn_slab_sum(ptr(gb_part[0]), gb_part[1], gpre.shape[1], ptr(out), stream()), "slab_sum")
            return out

        gw, gb = on_side_stream(lambda: (gemm_tn(gpre, x, g_amax, x_amax), bias_grad()),
                                [gpre, x, g_amax, x_amax, gb_part[0] if gb_part is not None else None], ctx.wb, wait=(ev,))
        return gx, gw, gb, dgamma, dbeta, None, None, None, None


# ---------------------------------------------------------------------------------------------
# The bond-angle embedding as one node: RBFExpansion -> MLPLayer -> MLPLayer (alignn/models/alignn.py:215-222) on T rows
# ---------------------------------------------------------------------------------------------
ANGLE_FUSED = True  # csrc/angle.hip where its shapes allow (training mode, BatchNorm flavour); tests flip it to compare


def angle_fused_applies(h, rbf, l1, l2, training) -> bool:
    """Shapes / modes csrc/angle.hip carries: float32 cosines without gradient on a HIP device, BatchNorm layers with running
    statistics - in training mode, or in evaluation mode without autograd -, bins < 48 -> 64 -> 256 features, no hooks on the
    three modules."""
    if not training and (torch.is_grad_enabled() or not INFER_FUSED):
        return False
    if not (ANGLE_FUSED and h.is_cuda and h.dtype == torch.float32 and not h.requires_grad and h.dim() == 1
            and h.numel() > 0):
        return False
    lin1, bn1, lin2, bn2 = l1.layer[0], l1.layer[1], l2.layer[0], l2.layer[1]
    if not (isinstance(bn1, torch.nn.BatchNorm1d) and isinstance(bn2, torch.nn.BatchNorm1d)):
        return False
    for bn in (bn1, bn2):
        if bn.running_mean is None or bn.momentum is None or bn.weight is None or abs(bn.momentum - BN_MOMENTUM) > 0 or bn.eps != BN_EPS:
            return False
    for m in (rbf, l1, l2, lin1, bn1, lin2, bn2, l1.layer, l2.layer):
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks:
            return False
    if lin1.weight.dtype != torch.float32 or lin1.bias is None or lin2.bias is None:
        return False
    return bool(_lib.load().alignn_angle_embed_supported(lin1.weight.shape[1], lin1.weight.shape[0], lin2.weight.shape[0])) and (
        lin1.weight.shape[1] == rbf.centers.numel() and lin2.weight.shape[1] == lin1.weight.shape[0])


def _angle_args(h, centers, gamma, p1, p2, bufs, stat1, stat2, scal):
    from .angle import AngleArgs
    from .cmodel import MlpParams

    a = AngleArgs()
    a.h, a.rows = ptr(h), h.numel()
    a.centers, a.gamma, a.bins = ptr(centers), float(gamma), centers.numel()
    for dst, (w, b, g, be), (rm, rv) in ((a.l1, p1, bufs[0]), (a.l2, p2, bufs[1])):
        dst.W, dst.b, dst.gamma, dst.beta, dst.rm, dst.rv = ptr(w), ptr(b), ptr(g), ptr(be), ptr(rm), ptr(rv)
        dst.in_, dst.out = w.shape[1], w.shape[0]
    a.eps, a.momentum = BN_EPS, BN_MOMENTUM
    a.stat1, a.stat2, a.scal = ptr(stat1), ptr(stat2), ptr(scal)
    return a


class AngleEmbedFn(torch.autograd.Function):
    """z [T, 256] from the cosines h [T] and the eight parameter tensors of the two layers; backward = their gradients (h
    gets none).  Forward and backward each are ONE C call (``alignn_angle_embed_fwd / _bwd``); inside ``lanes()`` both run on
    lane T like the T-row layers they replace."""

    @staticmethod
    def forward(ctx, h, centers, gamma, w1, b1, g1, be1, w2, b2, g2, be2, rm1, rv1, rm2, rv2):
        import ctypes as C

        lib = _lib.load()
        h = h.contiguous()
        T = h.numel()
        lane = _lane_for(T)
        ctx.lane = lane is not None

        def run():
            z = _empty(T, w2.shape[0], like=h)
            amax = new_amax(z) if _track(T) else None
            stat1, stat2, scal = _empty(4 * w1.shape[0], like=h), _empty(4 * w2.shape[0], like=h), _empty(lib.alignn_angle_embed_scal_floats(), like=h)
            a = _angle_args(h, centers, gamma, (w1, b1, g1, be1), (w2, b2, g2, be2), ((rm1, rv1), (rm2, rv2)), stat1, stat2, scal)
            nbytes = lib.alignn_angle_embed_workspace(T, centers.numel(), 0)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=h.device)
            a.z, a.z_amax, a.workspace, a.workspace_bytes = ptr(z), ptr(amax), ptr(ws), nbytes
            check(lib.alignn_angle_embed_fwd(C.byref(a), stream()), "angle_embed_fwd")
            return z, amax, stat1, stat2, scal

        if lane is not None:
            with _on_T(*lane, reads=(h,)):
                z, amax, stat1, stat2, scal = run()
            _mark_on_T(z)
        else:
            z, amax, stat1, stat2, scal = run()
        if amax is not None:
            set_amax(z, amax)
        ctx.save_for_backward(h, centers, w1, b1, g1, be1, w2, b2, g2, be2, rm1, rv1, rm2, rv2, stat1, stat2, scal)
        ctx.gamma = gamma
        ctx.param_grads = _PARAM_GRADS["on"]
        return z

    @staticmethod
    def backward(ctx, gz):
        import ctypes as C

        lib = _lib.load()
        h, centers, w1, b1, g1, be1, w2, b2, g2, be2, rm1, rv1, rm2, rv2, stat1, stat2, scal = ctx.saved_tensors
        none = (None,) * 15
        if not ctx.param_grads:
            return none
        gz = gz.contiguous()
        T = h.numel()

        def run():
            out = [(_empty(*w.shape, like=w), _empty(w.shape[0], like=w), _empty(2, w.shape[0], like=w)) for w in (w1, w2)]
            a = _angle_args(h, centers, ctx.gamma, (w1, b1, g1, be1), (w2, b2, g2, be2), ((rm1, rv1), (rm2, rv2)), stat1, stat2, scal)
            for dst, (gw, gb, red) in ((a.l1, out[0]), (a.l2, out[1])):
                dst.gW, dst.gb, dst.red = ptr(gw), ptr(gb), ptr(red)
            nbytes = lib.alignn_angle_embed_workspace(T, centers.numel(), 1)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=h.device)
            a.gz, a.workspace, a.workspace_bytes = ptr(gz), ptr(ws), nbytes
            check(lib.alignn_angle_embed_bwd(C.byref(a), stream()), "angle_embed_bwd")
            return out

        params = (w1, b1, g1, be1, w2, b2, g2, be2)
        if ctx.lane:
            main, T_ = _lane_streams(gz.device)
            with _on_T(main, T_, reads=(gz,)):
                out = run()
            ev = _event_after(T_)
            if _deferred_join_is_safe(params):  # the gradients stay on lane T until the end of backward()
                _arm_backward_join()
            else:
                main.wait_event(ev)
                for grp in out:
                    for t in grp:
                        t.record_stream(main)
        else:
            _main_reads(gz)
            out = run()
        (gw1, gb1, red1), (gw2, gb2, red2) = out
        return (None, None, None, gw1, gb1, red1[1], red1[0], gw2, gb2, red2[1], red2[0], None, None, None, None)


def angle_embed_infer(h, rbf, l1, l2):
    """Evaluation mode without autograd: one pass (``alignn_angle_embed_infer``)."""
    import ctypes as C

    lib = _lib.load()
    lin1, bn1, lin2, bn2 = l1.layer[0], l1.layer[1], l2.layer[0], l2.layer[1]
    h = h.contiguous()
    T = h.numel()
    lane = _lane_for(T)

    def run():
        z = _empty(T, lin2.weight.shape[0], like=h)
        amax = new_amax(z) if _track(T) else None
        stat1, stat2, scal = _empty(4 * lin1.weight.shape[0], like=h), _empty(4 * lin2.weight.shape[0], like=h), _empty(lib.alignn_angle_embed_scal_floats(), like=h)
        a = _angle_args(h, rbf.centers, rbf.gamma, (lin1.weight, lin1.bias, bn1.weight, bn1.bias),
                        (lin2.weight, lin2.bias, bn2.weight, bn2.bias),
                        ((bn1.running_mean, bn1.running_var), (bn2.running_mean, bn2.running_var)), stat1, stat2, scal)
        a.z, a.z_amax = ptr(z), ptr(amax)
        check(lib.alignn_angle_embed_infer(C.byref(a), stream()), "angle_embed_infer")
        return z, amax

    if lane is not None:
        with _on_T(*lane, reads=(h,)):
            z, amax = run()
        _mark_on_T(z)
    else:
        z, amax = run()
    if amax is not None:
        set_amax(z, amax)
    return z


def angle_embed(h, rbf, l1, l2):
    """``l2(l1(rbf(h)))`` through ``AngleEmbedFn`` (the caller checked ``angle_fused_applies``)."""
    lin1, bn1, lin2, bn2 = l1.layer[0], l1.layer[1], l2.layer[0], l2.layer[1]
    return AngleEmbedFn.apply(h, rbf.centers, rbf.gamma, lin1.weight, lin1.bias, bn1.weight, bn1.bias, lin2.weight, lin2.bias,
                              bn2.weight, bn2.bias, bn1.running_mean, bn1.running_var, bn2.running_mean, bn2.running_var)


# ---------------------------------------------------------------------------------------------
# EdgeGatedGraphConv   (alignn/models/alignn.py:78-129)
# ---------------------------------------------------------------------------------------------
# Composite entry points (csrc/composite.hip): the launches of a convolution's forward / backward issued by ONE C call each
# instead of ~11 / ~20 (same kernels, same arguments, same order: bit-identical - tests flip this flag).  Used for the
# BatchNorm flavour in training mode on one stream; lanes (hipGraph capture: host cost is irrelevant there), LayerNorm,
# eval mode and the rare kernel choices the composites do not carry take the per-kernel path below.
COMPOSITE = True
# composite backward: node input gradient on a second stream beside the edge one.  "auto": inside a stream capture only (like the
# lanes: -0.15 ms per replayed step at 64 crystals, -0.24 ms at 8; eagerly launched the four event calls per convolution cost
# the host more than the overlap returns: 16.96 vs 16.81 ms); "1": always; "0": never.
FORK_DGRAD = _os.environ.get("ALIGNN_AMD_FORK", "auto")
_AUX = {}  # device -> torch.cuda.Stream of the fork (None: the library's events could not be made yet)


def _aux_stream(dev):
    """Raw handle of the stream the composite backward forks its node input gradient onto, or None (switch off, or the
    first call on this device happens inside a stream capture - the library's event pair has to be created outside one)."""
    if FORK_DGRAD == "0":
        return None
    capturing = torch.cuda.is_current_stream_capturing()
    s = _AUX.get(dev)
    if s is None:
        if capturing:
            return None
        with torch.cuda.device(dev):
            check(_lib.load().alignn_fork_events_init(), "fork_events_init")
            s = _AUX[dev] = torch.cuda.Stream(device=dev)
    if FORK_DGRAD == "auto" and not capturing:
        return None
    return s.cuda_stream
COMPOSITE_STATS = {"fwd": 0, "bwd": 0, "wgrad": 0}


def _x6_shape_ok(a, N, K):
    """``project``'s test for the split-product kernels (shape part)."""
    M = a.shape[0]
    return (((M + 63) // 64) * ((N + 255) // 256) >= X6_MIN_TILES and a.stride(0) % 4 == 0
            and bool(_lib.load().alignn_gemm_nt_x6_supported(M, N, K)))


def _dgrad_kind(g, w, g_amax):
    """Which kernel ``_dgrad(g, w)`` (= ``project(g, w, transpose_w=True)``) runs: 1 f16x3 on the W^T image, 2 split-
    reduction NN, 3 plain NN, 4 NT on a transposed copy; None: one the composites do not carry (bf16x6)."""
    lib = _lib.load()
    M, N, K = g.shape[0], w.shape[1], w.shape[0]
    if _x6_shape_ok(g, N, K):
        return 1 if (F16X3 and g_amax is not None) else None
    if NN_SPLIT and lib.alignn_gemm_nn_split_workspace(M, w.shape[0], w.shape[1]):
        return 2 if (g.stride(0) % 4 == 0 and w.stride(0) % 4 == 0) else 3
    if w.shape[0] % 4 == 0 and w.shape[1] >= 16:
        return 4
    return 3


def _P(t):
    return 0 if t is None else t.data_ptr()


class _Shape:
    """What the kernel-choice helpers need of a not-yet-allocated contiguous [rows, cols] fp32 matrix."""

    def __init__(self, r, c):
        self.shape = (r, c)

    def stride(self, d):
        return self.shape[1] if d == 0 else 1


class EdgeGatedConvFn(torch.autograd.Function):
    """Whole convolution as one autograd node with a hand-written backward.

    Inputs are in the canonical segment order of ``graph`` (edge row k == CSR slot k).
    ``wcat`` = cat(src_gate, dst_gate, dst_update, src_update).weight  [4H,H];  ``bcat`` likewise: plain tensors
    (no grad) that ALIAS the storage of the eight leaves ``w4`` / ``b4`` (EdgeGatedGraphConv keeps its four node
    projections in one fused buffer); the leaves are passed only so that autograd routes their gradients - four row
    blocks of one [4H,H] weight-gradient GEMM.

    Inside ``lanes()`` a convolution with many edge rows (the line graph) splits over the two lanes: edge projection,
    gate pass and edge norm on lane T, node projection and node norm on the caller's stream, meeting at the gate pass
    (needs P) and after it (the node norm needs its sums); backward mirrors that around the block kernel.
    """

    @staticmethod
    def forward(ctx, graph: CSRGraph, x, y, wcat, bcat, w_sg, w_dg, w_du, w_su, b_sg, b_dg, b_du, b_su, w_eg, b_eg,
                n_gamma, n_beta, n_rm, n_rv, e_gamma, e_beta, e_rm, e_rv, training: bool, residual: bool,
                need_y: bool = True, norm: str = "batch"):
        lib = _lib.load()
        ctx.set_materialize_grads(False)
        x = x.contiguous()
        y = y.contiguous()
        n, H = x.shape
        m = y.shape[0]
        if n != graph.n_nodes or m != graph.n_edges:
            raise ValueError(f"feature rows ({n},{m}) do not match graph ({graph.n_nodes},{graph.n_edges})")
        lane = _lane_for(m)
        ctx.lane = lane is not None
        ctx.y_on_T = _is_on_T(y)
        ctx.y_src = _norm_src_of(y)  # y = r + silu(BatchNorm(.)) of the previous layer: see _dgrad_bnred
        ctx.x_amax, ctx.y_amax = get_amax(x), get_amax(y)  # tracked by the kernels that produced x and y
        bn_train = training and norm == "batch"
        if COMPOSITE and bn_train and lane is None and KERNEL_TIMER is None:
            done = EdgeGatedConvFn._forward_composite(ctx, graph, x, y, wcat, bcat, w_eg, b_eg, n_gamma, n_beta, n_rm, n_rv,
                                                      e_gamma, e_beta, e_rm, e_rv, residual, need_y)
            if done is not None:
                ctx.leaves = (w_sg, w_dg, w_du, w_su, b_sg, b_dg, b_du, b_su, w_eg, b_eg)
                _note_forward(ctx.leaves)
                return done
        slabs = lib.alignn_egc_slabs(n)
        _main_reads(x, None if lane is not None else y)
        # ---- node side, part 1 (caller's stream): P = [A | Bd | Bh | Ux]
        P = project(x, wcat, bcat, a_amax=ctx.x_amax)  # [n,4H]
        xpre = _empty(n, H, like=x)
        s0 = _empty(n, H, like=x)
        hh = _empty(n, H, like=x)
        n_part = _welford_slabs(slabs, H, x) if bn_train else None  # (written by the gate pass: Welford slabs)

        # u_add_v inside the edge projection's epilogue when that projection runs on the f16x3 kernel (the T- and E-row
        # convolutions of a real batch): M leaves the GEMM as m = A[u] + Bd[v] + C and the gate pass only reads it
        pre_added = GATHER_FUSED and w_eg.shape[0] == H and _f16x3_applies(y, ctx.y_amax, H, w_eg.shape[1])

        # ... and with BatchNorm the same epilogue leaves the column sums of m, so the gate pass can normalise right away:
        # it writes y' = y + silu(BN(m)) itself and the separate 3-row normalise / activate pass disappears
        fuse_norm = pre_added and norm == "batch" and STATS_FUSED

        def edge_side():
            bd2 = segment_ordered_bd(P, graph, H) if pre_added else None
            if fuse_norm and training:
                M, e_part, e_slabs = gemm_nt_f16x3_gather(y, ctx.y_amax, split_f16x2(w_eg), b_eg, P, graph.src, graph.dst,
                                                          want_stats=True, bd2=bd2, rank=graph.seg_rank)
                return M, (e_part, e_slabs)
            if pre_added:
                M = gemm_nt_f16x3_gather(y, ctx.y_amax, split_f16x2(w_eg), b_eg, P, graph.src, graph.dst, bd2=bd2,
                                         rank=graph.seg_rank)
            else:
                M = project(y, w_eg, b_eg, a_amax=ctx.y_amax)  # [m,H]  -> m_pre in place
            e_part = _welford_slabs(slabs, H, x) if (bn_train and not fuse_norm) else None
            return M, e_part

        fused_out = {}

        def gate(M, e_part):
            if fuse_norm:
                if training:
                    e_stat = _bn_finalize(e_part[0], e_part[1], m, e_gamma, e_beta, e_rm, e_rv, True)
                else:
                    e_stat = _bn_finalize(None, 0, m, e_gamma, e_beta, e_rm, e_rv, False)
                fused_out["e_stat"] = e_stat
                if need_y:
                    y_o = _empty(m, H, like=x)
                    y_amax = new_amax(x) if _track(m) else None
                    check(
                        lib.alignn_egc_gate_fwd_pre_norm(ptr(P), ptr(M), ptr(graph.seg_ptr), ptr(graph.seg_node),
                                                         ptr(graph.src), n, m, H, ptr(xpre), ptr(s0), ptr(hh), ptr(n_part),
                                                         ptr(e_stat), ptr(y) if residual else None, ptr(y_o), ptr(y_amax),
                                                         stream()),
                        "egc_gate_fwd_pre_norm",
                    )
                    fused_out["y"] = set_amax(y_o, y_amax) if y_amax is not None else y_o
                    return
                e_part = None  # dead edge output: plain pre-added gate pass, the statistics are already in e_stat
            if norm == "layer" and pre_added and need_y and lib.alignn_egc_ln_fused_supported(H, m):
                # LayerNorm flavour on a line graph: the edge LayerNorm inside the gate pass (csrc/convln.hip)
                y_o = _empty(m, H, like=x)
                y_amax = new_amax(x) if _track(m) else None
                e_rows = _empty(m, 2, like=x)
                check(
                    lib.alignn_egc_gate_fwd_pre_ln(ptr(P), ptr(M), ptr(graph.seg_ptr), ptr(graph.seg_node), ptr(graph.src), n, m,
                                                   H, ptr(xpre), ptr(s0), ptr(hh), ptr(e_gamma), ptr(e_beta), LN_EPS,
                                                   ptr(y) if residual else None, ptr(y_o), ptr(e_rows), ptr(y_amax), stream()),
                    "egc_gate_fwd_pre_ln",
                )
                fused_out["ln"] = (set_amax(y_o, y_amax) if y_amax is not None else y_o, e_rows)
                return
            fn = lib.alignn_egc_gate_fwd_pre if pre_added else lib.alignn_egc_gate_fwd
            check(
                fn(ptr(P), ptr(M), ptr(graph.seg_ptr), ptr(graph.seg_node), ptr(graph.src), n, m, H, ptr(xpre), ptr(s0),
                   ptr(hh), ptr(e_part), ptr(n_part), stream()),
                "egc_gate_fwd",
            )

        def edge_norm(M, e_part):
            if fuse_norm:
                y_o = fused_out.get("y")
                _register_norm_src(y_o, M, fused_out["e_stat"])
                return y_o, fused_out["e_stat"]
            if norm == "layer":  # LayerNorm flavour (alignn_atomwise.py:151,155): per-row statistics, no global barrier
                if "ln" in fused_out:
                    return fused_out["ln"]
                if need_y:
                    return _ln_silu_fwd(M, y if residual else None, e_gamma, e_beta)
                return None, _empty(1, 2, like=x)
            e_stat = (_bn_finalize(e_part, slabs, m, e_gamma, e_beta, e_rm, e_rv, True, welford=True) if training
                      else _bn_finalize(None, 0, m, e_gamma, e_beta, e_rm, e_rv, False))
            # need_y == False: the caller discards the edge output (last layer) - skip the pass, keep the
            # statistics side effect (running_mean/var of bn_edges are updated exactly as in the reference)
            y_o = _bn_silu_fwd(M, y if residual else None, e_stat) if need_y else None
            _register_norm_src(y_o, M, e_stat)
            return y_o, e_stat

        if lane is not None:
            main, T = lane
            ev_p = _event_after(main)
            for t in (P, xpre, s0, hh, n_part):
                if t is not None:
                    t.record_stream(T)
            with _on_T(main, T, reads=(y,)):
                if pre_added:
                    T.wait_event(ev_p)  # (the projection itself reads P now)
                M, e_part = edge_side()
                T.wait_event(ev_p)
                gate(M, e_part)
                ev_g = _event_after(T)
                y_out, e_stat = edge_norm(M, e_part)
            _mark_on_T(y_out)
            main.wait_event(ev_g)
        else:
            M, e_part = edge_side()
            gate(M, e_part)
            y_out, e_stat = edge_norm(M, e_part)
        # ---- node side, part 2 (caller's stream)
        if norm == "layer":
            x_out, n_stat = _ln_silu_fwd(xpre, x if residual else None, n_gamma, n_beta)
        else:
            n_stat = (_bn_finalize(n_part, slabs, n, n_gamma, n_beta, n_rm, n_rv, True, welford=True) if training
                      else _bn_finalize(None, 0, n, n_gamma, n_beta, n_rm, n_rv, False))
            x_out = _bn_silu_fwd(xpre, x if residual else None, n_stat)
        ctx.graph = graph
        ctx.training = training
        ctx.residual = residual
        ctx.norm = norm
        ctx.param_grads = _PARAM_GRADS["on"]
        ctx.leaves = (w_sg, w_dg, w_du, w_su, b_sg, b_dg, b_du, b_su, w_eg, b_eg)
        _note_forward(ctx.leaves)
        ctx.save_for_backward(x, y, wcat, w_eg, P, M, xpre, s0, hh, n_stat, e_stat, n_gamma, e_gamma, n_beta, e_beta)
        if FORWARD_TAPE is not None and norm == "layer":
            FORWARD_TAPE[w_eg.data_ptr()] = ("conv", x, y, P, M, xpre, s0, hh, x_out, y_out, e_stat)
        return x_out, y_out

    @staticmethod
    def _forward_composite(ctx, graph, x, y, wcat, bcat, w_eg, b_eg, n_gamma, n_beta, n_rm, n_rv, e_gamma, e_beta, e_rm, e_rv,
                           residual, need_y):
        """The BatchNorm / training forward through alignn_egc_conv_fwd; None if a kernel choice is not covered."""
        lib = _lib.load()
        n, Kin = x.shape
        m = y.shape[0]
        H = w_eg.shape[0]
        if wcat.shape != (4 * H, Kin) or y.shape[1] != Kin or (residual and Kin != H):
            return None
        require_f32(x, y, wcat, bcat, w_eg, b_eg, n_gamma, n_beta, e_gamma, e_beta)
        x_amax, y_amax = ctx.x_amax, ctx.y_amax
        # node projection: the kernel ``project(x, wcat, bcat)`` would take
        wcat_img = None
        if _x6_shape_ok(x, 4 * H, Kin):
            if not (F16X3 and x_amax is not None):
                return None
            node_kind = 1
        else:
            node_kind = 0
        # edge projection
        weg_img = None
        if GATHER_FUSED and STATS_FUSED and _f16x3_applies(y, y_amax, H, Kin):
            edge_kind = 1
        elif _x6_shape_ok(y, H, Kin):
            return None  # a split-product projection without the fused epilogues: per-kernel path
        else:
            edge_kind = 0
        _main_reads(x, y)
        if node_kind == 1:
            wcat_img = split_f16x2(wcat)
        if edge_kind == 1:
            weg_img = split_f16x2(w_eg)
            if BD_SEGMENT_TABLE and graph.seg_node is not None and graph.seg_rank is not None:
                edge_kind = 2  # (segment_ordered_bd's table, built and read inside the C call)
                BD_TABLE_STATS["used"] += 1
        P = _empty(n, 4 * H, like=x)
        M = _empty(m, H, like=x)
        xpre, s0, hh = _empty(n, H, like=x), _empty(n, H, like=x), _empty(n, H, like=x)
        stats = _empty(2, 4, H, like=x)
        n_stat, e_stat = stats[0], stats[1]
        x_out = _empty(n, H, like=x)
        y_out = _empty(m, H, like=x) if need_y else None
        xo_amax = new_amax(x) if _track(n) else None
        yo_amax = new_amax(x) if (need_y and _track(m)) else None
        nbytes = lib.alignn_egc_conv_fwd_scratch(n, m, H, Kin, edge_kind)
        scratch = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
        args = _lib.EGC_FWD_ARGS.pack(
            _P(graph.seg_ptr), _P(graph.seg_node), _P(graph.src), _P(graph.dst), _P(graph.seg_rank), n, m,
            H, Kin, node_kind, edge_kind, int(residual), 0, BN_EPS, BN_MOMENTUM,
            _P(x), _P(y), _P(x_amax), _P(y_amax),
            _P(wcat), _P(bcat), _P(wcat_img.amax if wcat_img is not None else None), _P(wcat_img.buf if wcat_img is not None else None),
            _P(w_eg), _P(b_eg), _P(weg_img.amax if weg_img is not None else None), _P(weg_img.buf if weg_img is not None else None),
            _P(n_gamma), _P(n_beta), _P(e_gamma), _P(e_beta), _P(n_rm), _P(n_rv), _P(e_rm), _P(e_rv),
            _P(P), _P(M), _P(xpre), _P(s0), _P(hh), _P(n_stat), _P(e_stat), _P(x_out), _P(y_out), _P(xo_amax), _P(yo_amax),
            _P(scratch), nbytes)
        check(lib.alignn_egc_conv_fwd(args, stream()), "egc_conv_fwd")
        COMPOSITE_STATS["fwd"] += 1
        if xo_amax is not None:
            set_amax(x_out, xo_amax)
        if yo_amax is not None:
            set_amax(y_out, yo_amax)
        _register_norm_src(y_out, M, e_stat)
        ctx.graph = graph
        ctx.training = True
        ctx.residual = residual
        ctx.norm = "batch"
        ctx.param_grads = _PARAM_GRADS["on"]
        ctx.save_for_backward(x, y, wcat, w_eg, P, M, xpre, s0, hh, n_stat, e_stat, n_gamma, e_gamma, n_beta, e_beta)
        return x_out, y_out

    @staticmethod
    def _backward_composite(ctx, gx_out, gy_out):
        """The BatchNorm / training backward through alignn_egc_conv_bwd (+ alignn_egc_conv_wgrad on the side stream);
        None if a kernel choice is not covered (the per-kernel path then runs)."""
        lib = _lib.load()
        graph: CSRGraph = ctx.graph
        x, y, wcat, w_eg, P, M, xpre, s0, hh, n_stat, e_stat, n_gamma, e_gamma, n_beta, e_beta = ctx.saved_tensors
        n, Kin = x.shape
        m = y.shape[0]
        H = w_eg.shape[0]
        if wcat.shape != (4 * H, Kin) or w_eg.shape[1] != Kin:
            return None
        if ctx.param_grads and Kin == H and dgrad_wgrad_applies(m, H, Kin, True if _track(m) else None, ctx.y_amax):
            return None  # (input + weight gradient of the edge-gate projection in one pass: the per-kernel path below)
        # kernel choices of the two input-gradient products, decided on shapes and on which maxima will exist (the arena
        # slots themselves are drawn only once the composite is certain to run)
        has_gp, has_gm = (True if _track(n) else None), (True if _track(m) else None)
        dx_kind = _dgrad_kind(_Shape(n, 4 * H), wcat, has_gp)
        if dx_kind is None:
            return None
        y_src = ctx.y_src
        bnred = (y_src is not None and BNRED_FUSED and F16X3 and has_gm is not None and _x6_shape_ok(_Shape(m, H), Kin, H)
                 and tuple(y_src[0].shape) == (m, Kin) and y_src[0].stride(0) % 4 == 0)
        if bnred:
            dy_kind = 1
        else:
            dy_kind = _dgrad_kind(_Shape(m, H), w_eg, has_gm)
            if dy_kind == 1:
                dy_kind = 5
            elif dy_kind == 2 or dy_kind is None:
                return None
        gp_amax = new_amax(x) if has_gp else None
        gm_amax = new_amax(x) if has_gm else None
        if gx_out is None:
            gx_out = torch.zeros_like(x)
        _main_reads(gx_out, gy_out)
        gx_out = gx_out.contiguous()
        if gy_out is not None:
            gy_out = gy_out.contiguous()
        lg_blocks = graph.grp_seg_ptr is not None and FUSED_LG_BACKWARD
        dense = lg_blocks and DENSE_LG_BACKWARD and lib.alignn_egc_bwd_lg_dense_supported(graph.dense_max_src)
        gate_mode = 2 if dense else (1 if lg_blocks else 0)
        gslabs = (graph.grp_seg_ptr.numel() - 1) if lg_blocks else lib.alignn_egc_slabs(n)
        e_red_in = _take_pre_red(gy_out, M) if gy_out is not None else None
        wcat_t_img = split_f16x2(wcat, True) if dx_kind == 1 else None
        weg_t_img = split_f16x2(w_eg, True) if dy_kind in (1, 5) else None
        wcat_t = wcat.t().contiguous() if dx_kind == 4 else None
        weg_t = w_eg.t().contiguous() if dy_kind == 4 else None
        GP = _empty(n, 4 * H, like=x)
        GM = _empty(m, H, like=x)
        gs = _empty(2, n, H, like=x)
        gb_part = _empty(gslabs, H, like=x)
        n_red = _empty(2, H, like=x)
        e_red = e_red_in if e_red_in is not None else (_empty(2, H, like=x) if gy_out is not None else None)
        src_red = _empty(2, Kin, like=x) if dy_kind == 1 else None
        g_x = _empty(n, Kin, like=x)
        g_y = _empty(m, Kin, like=x)
        nbytes = lib.alignn_egc_conv_bwd_scratch(n, m, H, Kin, dx_kind, dy_kind)
        scratch = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
        args = _lib.EGC_BWD_ARGS.pack(
            _P(graph.seg_ptr), _P(graph.seg_node), _P(graph.src), _P(graph.dst), _P(graph.out_ptr), _P(graph.out_slot),
            _P(graph.grp_seg_ptr), _P(graph.grp_src_ptr), n, m, gslabs if lg_blocks else 0,
            H, Kin, gate_mode, int(graph.dense_max_src) if dense else 0, dx_kind, dy_kind, int(ctx.residual), 0,
            _P(gx_out), _P(gy_out),
            _P(P), _P(M), _P(xpre), _P(s0), _P(hh), _P(n_stat), _P(e_stat), _P(n_gamma), _P(e_gamma),
            _P(e_red_in),
            _P(wcat), _P(wcat_t), _P(wcat_t_img.amax if wcat_t_img is not None else None),
            _P(wcat_t_img.buf if wcat_t_img is not None else None),
            _P(w_eg), _P(weg_t), _P(weg_t_img.amax if weg_t_img is not None else None),
            _P(weg_t_img.buf if weg_t_img is not None else None),
            _P(y_src[0] if dy_kind == 1 else None), _P(y_src[1] if dy_kind == 1 else None),
            y_src[0].stride(0) if dy_kind == 1 else 0,
            _P(GP), _P(GM), _P(gs[0]), _P(gs[1]), _P(n_red), _P(e_red if e_red_in is None else None), _P(gb_part), _P(g_x),
            _P(g_y), _P(src_red), _P(gp_amax), _P(gm_amax), _aux_stream(x.device) or 0,
            _P(scratch), nbytes)
        check(lib.alignn_egc_conv_bwd(args, stream()), "egc_conv_bwd")
        COMPOSITE_STATS["bwd"] += 1
        if dy_kind == 1:
            BNRED_STATS["fused"] += 1
            k = id(g_y)
            _PRE_RED[k] = (weakref.ref(g_y, lambda _r, k=k: _orphan_pre_red(k)), y_src[0], src_red)
        if not ctx.param_grads:
            return (None, g_x, g_y) + (None,) * 24
        x_amax, y_amax = ctx.x_amax, ctx.y_amax
        tn_gm = (gm_amax, y_amax) if (F16X3 and gm_amax is not None and y_amax is not None) else (None, None)
        tn_gp = (gp_amax, x_amax) if (F16X3 and gp_amax is not None and x_amax is not None) else (None, None)

        def _wgrads():
            g_weg, g_beg = _empty(H, Kin, like=x), _empty(H, like=x)
            g_wcat, g_bcat = _empty(4 * H, Kin, like=x), _empty(4 * H, like=x)
            wb = lib.alignn_egc_conv_wgrad_scratch(n, m, H, Kin)
            ws = torch.empty(wb // 4, dtype=torch.float32, device=x.device)
            wargs = _lib.EGC_WGRAD_ARGS.pack(n, m, H, Kin, gslabs, 0, _P(GM), _P(GP), _P(x), _P(y), _P(gb_part), _P(tn_gm[0]),
                                             _P(tn_gp[0]), _P(tn_gp[1]), _P(tn_gm[1]), _P(g_weg), _P(g_beg), _P(g_wcat), _P(g_bcat),
                                             _P(ws), wb)
            check(lib.alignn_egc_conv_wgrad(wargs, stream()), "egc_conv_wgrad")
            COMPOSITE_STATS["wgrad"] += 1
            return g_weg, g_beg, g_wcat, g_bcat

        g_weg, g_beg, g_wcat, g_bcat = on_side_stream(_wgrads, [GM, y, GP, x, gb_part, gm_amax, gp_amax, x_amax, y_amax],
                                                      ctx.leaves)
        dn_gamma, dn_beta = n_red[1], n_red[0]
        de_gamma = e_red[1] if e_red is not None else None
        de_beta = e_red[0] if e_red is not None else None
        gw4 = tuple(g_wcat[i * H:(i + 1) * H] for i in range(4))
        gb4 = tuple(g_bcat[i * H:(i + 1) * H] for i in range(4))
        return (None, g_x, g_y, None, None) + gw4 + gb4 + (g_weg, g_beg, dn_gamma, dn_beta, None, None, de_gamma,
                                                            de_beta, None, None, None, None, None, None)

    @staticmethod
    def backward(ctx, gx_out, gy_out):
        lib = _lib.load()
        if (COMPOSITE and ctx.norm == "batch" and ctx.training and not ctx.lane and KERNEL_TIMER is None):
            done = EdgeGatedConvFn._backward_composite(ctx, gx_out, gy_out)
            if done is not None:
                return done
        graph: CSRGraph = ctx.graph
        x, y, wcat, w_eg, P, M, xpre, s0, hh, n_stat, e_stat, n_gamma, e_gamma, n_beta, e_beta = ctx.saved_tensors
        n, H = x.shape
        m = y.shape[0]
        ev = not ctx.training
        layer = ctx.norm == "layer"
        if gx_out is None:
            gx_out = torch.zeros_like(x)
        _main_reads(gx_out, None if ctx.lane else gy_out)
        gx_out = gx_out.contiguous()
        GP = _empty(n, 4 * H, like=x)
        # max|GP| (all four blocks: every kernel that writes a block raises the same scalar) and max|GM|, so that
        # the input- and weight-gradient projections below can run the three-product scheme
        gp_amax = new_amax(x) if _track(n) else None
        gm_amax = new_amax(x) if _track(m) else None
        # ---- node branch (caller's stream): SiLU/norm backward -> g_xpre (stored as the Ux block of GP)
        g_xpre = GP[:, 3 * H:]
        gs1 = _empty(n, H, like=x)
        gs0 = _empty(n, H, like=x)
        if layer:  # (LayerNorm backward and the quotient's adjoints in one pass)
            n_red = _ln_silu_bwd(gx_out, xpre, n_gamma, n_beta, n_stat, g_xpre, gp_amax, node=(s0, hh, gs1, gs0))
        else:
            n_red = _bn_silu_bwd_reduce(gx_out, xpre, n_stat)
        if not layer:  # (norm backward and the quotient's adjoints in one pass)
            check(lib.alignn_bn_silu_bwd_apply_node(ptr(gx_out), gx_out.stride(0), ptr(xpre), xpre.stride(0), ptr(n_stat),
                                                    ptr(n_gamma), ptr(n_red), int(ev), ptr(g_xpre), g_xpre.stride(0), n, H,
                                                    ptr(gp_amax), ptr(s0), ptr(hh), ptr(gs1), ptr(gs0), stream()),
                  "bn_silu_bwd_apply_node")
        if gy_out is not None:
            gy_out = gy_out.contiguous()
        lg_blocks = graph.grp_seg_ptr is not None and FUSED_LG_BACKWARD
        dense = lg_blocks and DENSE_LG_BACKWARD and lib.alignn_egc_bwd_lg_dense_supported(graph.dense_max_src)
        ln_inside = bool(layer and dense and gy_out is not None and lib.alignn_egc_ln_fused_supported(H, m))  # (csrc/convln.hip)
        ln_dst = bool(layer and not lg_blocks and gy_out is not None and lib.alignn_egc_ln_dst_supported(H))  # (the bond graph)
        fused_red = {}

        def edge_branch():
            """norm backward of the edge output, then the gate backward: writes GM [m,H], the A | Bd | Bh blocks of GP and
            the column-sum slabs of GM.  Everything here touches T rows."""
            e_red = None
            g_branch, e_stat_arg = gy_out, e_stat
            if gy_out is not None:
                if ln_inside or ln_dst:
                    pass  # (LayerNorm backward inside the gate backward: gate_backward below)
                elif layer:
                    # LayerNorm: finish the normalised-branch gradient here, hand it over as-is (e_stat = NULL)
                    g_branch = _empty(m, H, like=x)
                    e_red = _ln_silu_bwd(gy_out, M, e_gamma, e_beta, e_stat, g_branch)
                    e_stat_arg = None
                else:
                    e_red = _take_pre_red(gy_out, M)  # left by the layer that produced gy_out, if it could
                    if e_red is None:
                        e_red = _bn_silu_bwd_reduce(gy_out, M, e_stat)
            return e_red, g_branch, e_stat_arg

        def gate_backward(e_red, g_branch, e_stat_arg):
            GM = _empty(m, H, like=x)
            if ln_inside:
                gslabs = graph.grp_seg_ptr.numel() - 1
                gb_part = _empty(gslabs, H, like=x)
                ln_part = _empty(gslabs, 2, H, like=x)
                check(
                    lib.alignn_egc_bwd_lg_dense_ln(ptr(gy_out), ptr(M), ptr(P), ptr(gs1), ptr(gs0), ptr(e_gamma), ptr(e_beta),
                                                   ptr(e_stat), m, ptr(graph.grp_seg_ptr), ptr(graph.grp_src_ptr), gslabs,
                                                   graph.dense_max_src, ptr(graph.seg_ptr), ptr(graph.seg_node), H, ptr(GM),
                                                   ptr(GP), ptr(gb_part), ptr(ln_part), ptr(gm_amax), ptr(gp_amax), stream()),
                    "egc_bwd_lg_dense_ln",
                )
                red = _empty(2, H, like=x)
                check(lib.alignn_bn_bwd_finalize(ptr(ln_part), gslabs, H, ptr(red), stream()), "ln_finalize")
                fused_red["e"] = red
            elif dense:
                # line graph with dense, source-sorted blocks: one pass, rows addressed by index arithmetic
                gslabs = graph.grp_seg_ptr.numel() - 1
                gb_part = _empty(gslabs, H, like=x)
                check(
                    lib.alignn_egc_bwd_lg_dense(ptr(g_branch), ptr(M), ptr(P), ptr(gs1), ptr(gs0), ptr(e_stat_arg),
                                                ptr(e_red), int(ev), m, ptr(graph.grp_seg_ptr), ptr(graph.grp_src_ptr),
                                                gslabs, graph.dense_max_src, ptr(graph.seg_ptr), ptr(graph.seg_node), H,
                                                ptr(GM), ptr(GP), ptr(gb_part), ptr(gm_amax), ptr(gp_amax), stream()),
                    "egc_bwd_lg_dense",
                )
            elif lg_blocks:
                # line graph: destination- and source-ordered passes in one kernel, one workgroup per centre atom
                gslabs = graph.grp_seg_ptr.numel() - 1
                gb_part = _empty(gslabs, H, like=x)
                check(
                    lib.alignn_egc_bwd_lg_fused(ptr(g_branch), ptr(M), ptr(P), ptr(gs1), ptr(gs0), ptr(e_stat_arg),
                                                ptr(e_red), int(ev), m, ptr(graph.grp_seg_ptr), ptr(graph.grp_src_ptr),
                                                gslabs, ptr(graph.seg_ptr), ptr(graph.seg_node), ptr(graph.dst),
                                                ptr(graph.out_ptr), ptr(graph.out_slot), H, ptr(GM), ptr(GP), ptr(gb_part),
                                                ptr(gm_amax), ptr(gp_amax), stream()),
                    "egc_bwd_lg_fused",
                )
            elif ln_dst:
                gslabs = lib.alignn_egc_ln_dst_slabs(n)
                gb_part = _empty(gslabs, H, like=x)
                ln_part = _empty(gslabs, 2, H, like=x)
                check(
                    lib.alignn_egc_bwd_dst_ln(ptr(gy_out), ptr(M), ptr(P), ptr(gs1), ptr(gs0), ptr(e_gamma), ptr(e_beta), ptr(e_stat),
                                              ptr(graph.seg_ptr), ptr(graph.seg_node), ptr(graph.src), n, H, ptr(GM), ptr(GP),
                                              ptr(gb_part), ptr(ln_part), ptr(gm_amax), ptr(gp_amax), stream()),
                    "egc_bwd_dst_ln",
                )
                red = _empty(2, H, like=x)
                check(lib.alignn_bn_bwd_finalize(ptr(ln_part), gslabs, H, ptr(red), stream()), "ln_finalize")
                fused_red["e"] = red
                check(
                    lib.alignn_egc_bwd_src(ptr(GM), ptr(M), ptr(gs1), ptr(graph.out_ptr), ptr(graph.out_slot),
                                           ptr(graph.dst), n, H, ptr(GP), ptr(gp_amax), stream()),
                    "egc_bwd_src",
                )
            else:
                gslabs = lib.alignn_egc_slabs(n)
                gb_part = _empty(gslabs, H, like=x)
                check(
                    lib.alignn_egc_bwd_dst(ptr(g_branch), ptr(M), ptr(P), ptr(gs1), ptr(gs0), ptr(e_stat_arg), ptr(e_gamma),
                                           ptr(e_red), int(ev), m, ptr(graph.seg_ptr), ptr(graph.seg_node), ptr(graph.src),
                                           n, H, ptr(GM), ptr(GP), ptr(gb_part), ptr(gm_amax), ptr(gp_amax), stream()),
                    "egc_bwd_dst",
                )
                check(
                    lib.alignn_egc_bwd_src(ptr(GM), ptr(M), ptr(gs1), ptr(graph.out_ptr), ptr(graph.out_slot),
                                           ptr(graph.dst), n, H, ptr(GP), ptr(gp_amax), stream()),
                    "egc_bwd_src",
                )
            return GM, gb_part, gslabs

        fused_dw = {}

        def edge_dgrad(GM):
            addend = gy_out if (ctx.residual and gy_out is not None) else None
            Kin_ = y.shape[1]
            if (ctx.param_grads and Kin_ == H and w_eg.shape == (H, Kin_) and GM.stride(0) % 4 == 0 and y.stride(0) % 4 == 0
                    and dgrad_wgrad_applies(m, H, Kin_, gm_amax, ctx.y_amax) and _x6_shape_ok(_Shape(m, H), Kin_, H)):
                # input gradient + weight gradient of the edge-gate projection in one pass over GM (csrc/gemm_dw.hip)
                src = ctx.y_src if (ctx.y_src is not None and BNRED_FUSED and not layer and tuple(ctx.y_src[0].shape) == (m, Kin_)
                                    and ctx.y_src[0].stride(0) % 4 == 0) else None
                out, dW, red = gemm_dgrad_wgrad(GM, gm_amax, y, ctx.y_amax, split_f16x2(w_eg, True), addend,
                                                src[0] if src is not None else None, src[1] if src is not None else None)
                fused_dw["g_weg"] = dW
                if src is not None:
                    BNRED_STATS["fused"] += 1
                    k = id(out)
                    _PRE_RED[k] = (weakref.ref(out, lambda _r, k=k: _orphan_pre_red(k)), src[0], red)
                return out
            return _dgrad_bnred(GM, w_eg, addend, gm_amax, ctx.y_src)

        ev_d = ev_w = None
        if ctx.lane:
            main, T = _lane_streams(x.device)
            ev_n = _event_after(main)  # gs1, gs0, the Ux block of GP and its amax are ready
            for t in (GP, gs1, gs0, gp_amax, gm_amax, P):
                if t is not None:
                    t.record_stream(T)
            with _on_T(main, T, reads=(gy_out,)):
                e_red, g_branch, e_stat_arg = edge_branch()  # (independent of the node branch: no wait yet)
                T.wait_event(ev_n)
                GM, gb_part, gslabs = gate_backward(e_red, g_branch, e_stat_arg)
                e_red = fused_red.get("e", e_red)
                ev_d = _event_after(T)
                g_y = edge_dgrad(GM)
                if "g_weg" in fused_dw:
                    ev_w = _event_after(T)  # (the weight gradient came out of the same pass: the side stream hands it on)
            main.wait_event(ev_d)  # GP complete: the node input gradient below reads it
            # de_gamma / de_beta (e_red) and g_y come off lane T: they may stay there if y's producer takes its gradient
            # on lane T itself and nothing reads the two norm gradients before the end-of-backward join
            if ctx.y_on_T and (not ctx.param_grads or _deferred_join_is_safe((e_gamma, e_beta))):
                _arm_backward_join()
                _mark_on_T(g_y)
            else:
                main.wait_stream(T)
                for t in (g_y, e_red):
                    if t is not None:
                        t.record_stream(main)
        else:
            e_red, g_branch, e_stat_arg = edge_branch()
            GM, gb_part, gslabs = gate_backward(e_red, g_branch, e_stat_arg)
            e_red = fused_red.get("e", e_red)
        # projections: weight gradients on the side stream, input gradients (critical path) on the main one
        def _wgrads():
            g_beg_ = _empty(H, like=x)  # column sum of GM, accumulated inside the destination-order pass
            check(lib.alignn_slab_sum(ptr(gb_part), gslabs, H, ptr(g_beg_), stream()), "slab_sum")
            g_weg_ = fused_dw["g_weg"] if "g_weg" in fused_dw else gemm_tn(GM, y, gm_amax, y_amax)
            return g_weg_, g_beg_, gemm_tn(GP, x, gp_amax, x_amax), col_sum(GP)

        x_amax, y_amax = ctx.x_amax, ctx.y_amax

        # order matters: the side stream starts where the main stream stands at the on_side_stream() call.  Issue
        # the input-gradient GEMMs first, so the weight-gradient GEMMs (LDS-heavy, cannot share a CU with the x6
        # tiles) run beside the NEXT layer's HBM-bound kernels instead of fighting these GEMMs for whole CUs.
        g_x = _dgrad(GP, wcat, addend=gx_out if ctx.residual else None, g_amax=gp_amax)
        if not ctx.lane:
            g_y = edge_dgrad(GM)
        if not ctx.param_grads:
            return (None, g_x, g_y) + (None,) * 24
        g_weg, g_beg, g_wcat, g_bcat = on_side_stream(_wgrads, [GM, y, GP, x, gb_part, gm_amax, gp_amax, x_amax, y_amax],
                                                      ctx.leaves, wait=(ev_d, ev_w))
        dn_gamma, dn_beta = n_red[1], n_red[0]
        de_gamma = e_red[1] if e_red is not None else None
        de_beta = e_red[0] if e_red is not None else None
        gw4 = tuple(g_wcat[i * H:(i + 1) * H] for i in range(4))  # row blocks of the fused gradient (views, no copy)
        gb4 = tuple(g_bcat[i * H:(i + 1) * H] for i in range(4))
        return (None, g_x, g_y, None, None) + gw4 + gb4 + (g_weg, g_beg, dn_gamma, dn_beta, None, None, de_gamma,
                                                            de_beta, None, None, None, None, None, None)


def edge_gated_conv_cat(graph: CSRGraph, x, y, wcat, bcat, w_eg, b_eg, n_gamma, n_beta, n_rm, n_rv, e_gamma, e_beta, e_rm,
                        e_rv, training: bool, residual: bool, need_y: bool = True, norm: str = "batch"):
    """EdgeGatedConvFn with the fused node projection given as ONE [4H,K] weight and ONE [4H] bias that may themselves
    require grad (stand-alone use, kernel tests): their row blocks are handed over as views, so autograd adds the four
    block gradients back into ``wcat.grad`` / ``bcat.grad``."""
    H = wcat.shape[0] // 4
    w4 = tuple(wcat[i * H:(i + 1) * H] for i in range(4))
    b4 = tuple(bcat[i * H:(i + 1) * H] for i in range(4))
    return EdgeGatedConvFn.apply(graph, x, y, wcat.detach(), bcat.detach(), *w4, *b4, w_eg, b_eg, n_gamma, n_beta, n_rm,
                                 n_rv, e_gamma, e_beta, e_rm, e_rv, training, residual, need_y, norm)


def edge_gated_conv_infer(graph: CSRGraph, x, y, wcat, bcat, w_eg, b_eg, n_gamma, n_beta, n_rm, n_rv, e_gamma, e_beta,
                          e_rm, e_rv, residual: bool, need_y: bool = True):
    """EdgeGatedGraphConv.forward for inference (eval mode, no autograd): BatchNorm is the affine map of its running
    statistics, so the edge output comes straight out of the gate pass (alignn_egc_gate_infer) - m_pre is never
    written and nothing is saved.  Same values as the training-capable path in eval mode."""
    lib = _lib.load()
    x, y = x.contiguous(), y.contiguous()
    n, H = x.shape
    m = y.shape[0]
    if n != graph.n_nodes or m != graph.n_edges:
        raise ValueError(f"feature rows ({n},{m}) do not match graph ({graph.n_nodes},{graph.n_edges})")
    P = project(x, wcat, bcat)
    C = project(y, w_eg, b_eg)
    n_stat = _bn_finalize(None, 0, n, n_gamma, n_beta, n_rm, n_rv, False)
    e_stat = _bn_finalize(None, 0, m, e_gamma, e_beta, e_rm, e_rv, False)
    xpre = _empty(n, H, like=x)
    y_out = _empty(m, H, like=x) if need_y else None
    y_amax = new_amax(x) if (need_y and _track(m)) else None
    check(
        lib.alignn_egc_gate_infer(ptr(P), ptr(C), ptr(graph.seg_ptr), ptr(graph.seg_node), ptr(graph.src), n, m, H,
                                  ptr(xpre), ptr(e_stat), ptr(y) if residual else None, ptr(y_out), ptr(y_amax), stream()),
        "egc_gate_infer",
    )
    if y_amax is not None:
        set_amax(y_out, y_amax)
    x_out = _bn_silu_fwd(xpre, x if residual else None, n_stat)
    return x_out, y_out


# ---------------------------------------------------------------------------------------------
# featurisation / readout
# ---------------------------------------------------------------------------------------------
def _rbf_raw(d, centers, gamma):
    lib = _lib.load()
    require_f32(d, centers)
    rows, bins = d.numel(), centers.numel()
    out = _empty(rows, bins, like=d)
    check(lib.alignn_rbf_fwd(ptr(d), ptr(centers), float(gamma), ptr(out), rows, bins, stream()), "rbf_fwd")
    return out


class RbfFn(torch.autograd.Function):
    """RBF expansion with its first derivative w.r.t. the distance / cosine (force-field inference)."""

    @staticmethod
    def forward(ctx, d, centers, gamma):
        d = d.contiguous()
        ctx.save_for_backward(d, centers)
        ctx.gamma = float(gamma)
        return _rbf_raw(d, centers, gamma)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        d, centers = ctx.saved_tensors
        g = g.contiguous()
        gd = _empty(d.numel(), like=d)
        check(lib.alignn_rbf_bwd(ptr(d), ptr(centers), ctx.gamma, ptr(g), ptr(gd), d.numel(), centers.numel(), stream()),
              "rbf_bwd")
        return gd.reshape(d.shape), None, None


def rbf_expand(d, centers, gamma):
    """RBFExpansion.forward (alignn/models/utils.py:40-44); differentiable (first order) w.r.t. ``d``."""
    if d.requires_grad:
        return RbfFn.apply(d, centers, gamma)
    return _rbf_raw(d.contiguous(), centers, gamma)


def _bond_length_raw(r):
    lib = _lib.load()
    require_f32(r)
    out = _empty(r.shape[0], like=r)
    check(lib.alignn_norm3_fwd(ptr(r), ptr(out), r.shape[0], stream()), "norm3_fwd")
    return out


class BondLengthFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, r):
        r = r.contiguous()
        ctx.save_for_backward(r)
        return _bond_length_raw(r)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (r,) = ctx.saved_tensors
        gr = torch.empty_like(r)
        check(lib.alignn_norm3_bwd(ptr(r), ptr(g.contiguous()), ptr(gr), r.shape[0], stream()), "norm3_bwd")
        return gr


def bond_length(r):
    """torch.norm(r, dim=1) (alignn/models/alignn.py:313); differentiable (first order) w.r.t. ``r``."""
    if r.requires_grad:
        return BondLengthFn.apply(r)
    return _bond_length_raw(r.contiguous())


class AvgPoolFn(torch.autograd.Function):
    """dgl.nn.AvgPooling (alignn/models/alignn.py:325): per-crystal mean over atoms."""

    @staticmethod
    def forward(ctx, x, graph_ptr):
        lib = _lib.load()
        x = x.contiguous()
        B = graph_ptr.numel() - 1
        H = x.shape[1]
        out = _empty(B, H, like=x)
        check(lib.alignn_segment_mean_fwd(ptr(x), ptr(graph_ptr), ptr(out), B, H, stream()), "segment_mean_fwd")
        ctx.save_for_backward(graph_ptr)
        ctx.n = x.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (graph_ptr,) = ctx.saved_tensors
        g = g.contiguous()
        B, H = g.shape
        gx = _empty(ctx.n, H, like=g)
        check(lib.alignn_segment_mean_bwd(ptr(g), ptr(graph_ptr), ptr(gx), B, H, stream()), "segment_mean_bwd")
        return gx, None
