"""The bond-angle embedding as ONE pair of C calls (csrc/angle.hip): ``alignn_angle_embed_fwd`` / ``alignn_angle_embed_bwd``
for callers outside the whole-model path (tests, tools).  ``z = MLPLayer(64 -> 256)(MLPLayer(bins -> 64)(RBFExpansion(h)))``
(alignn/models/alignn.py:215-222) in training mode, without the [T, bins] / [T, 64] / [T, 256] intermediates.  No fallback:
the library is required."""

import ctypes as C

import torch

from . import _lib
from .cmodel import MlpParams

_p, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class AngleArgs(C.Structure):
    _fields_ = [("h", _p), ("rows", _i64), ("centers", _p), ("gamma", _f32), ("bins", _i32), ("l1", MlpParams), ("l2", MlpParams),
                ("eps", _f32), ("momentum", _f32), ("stat1", _p), ("stat2", _p), ("scal", _p), ("z", _p), ("z_amax", _p),
                ("gz", _p), ("workspace", _p), ("workspace_bytes", C.c_size_t)]


def _mlp(lin, bn, grads):
    m = MlpParams()
    m.W, m.b, m.gamma, m.beta = (t.data_ptr() for t in (lin.weight, lin.bias, bn.weight, bn.bias))
    m.rm, m.rv = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
    if grads is not None:
        m.gW, m.gb, m.red = (t.data_ptr() for t in grads)
    m.in_, m.out = lin.weight.shape[1], lin.weight.shape[0]
    return m


class AngleEmbedding:
    """Holds the buffers one forward + backward of the embedding needs.  ``layers`` = the two MLPLayer modules."""

    def __init__(self, centers: torch.Tensor, gamma: float, layers, eps=1e-5, momentum=0.1):
        self.lib = _lib.load()
        assert self.lib.alignn_angle_args_sizeof() == C.sizeof(AngleArgs)
        self.centers, self.gamma, self.layers, self.eps, self.momentum = centers.contiguous(), float(gamma), layers, eps, momentum
        dev = centers.device
        self.stat1 = torch.empty(4 * 64, device=dev)
        self.stat2 = torch.empty(4 * 256, device=dev)
        self.scal = torch.empty(self.lib.alignn_angle_embed_scal_floats(), device=dev)
        (l1, b1), (l2, b2) = ((m.layer[0], m.layer[1]) for m in layers)
        self.grads = [(torch.empty_like(l.weight), torch.empty_like(l.bias), torch.empty(2 * l.weight.shape[0], device=dev))
                      for l in (l1, l2)]
        assert self.lib.alignn_angle_embed_supported(l1.weight.shape[1], l1.weight.shape[0], l2.weight.shape[0])

    def _args(self, h):
        a = AngleArgs()
        a.h, a.rows = h.data_ptr(), h.numel()
        a.centers, a.gamma, a.bins = self.centers.data_ptr(), self.gamma, self.centers.numel()
        (l1, b1), (l2, b2) = ((m.layer[0], m.layer[1]) for m in self.layers)
        a.l1, a.l2 = _mlp(l1, b1, self.grads[0]), _mlp(l2, b2, self.grads[1])
        a.eps, a.momentum = self.eps, self.momentum
        a.stat1, a.stat2, a.scal = self.stat1.data_ptr(), self.stat2.data_ptr(), self.scal.data_ptr()
        return a

    def forward(self, h: torch.Tensor):
        h = h.contiguous()
        a = self._args(h)
        z = torch.empty(h.numel(), 256, device=h.device)
        amax = torch.zeros(1, device=h.device)
        nbytes = self.lib.alignn_angle_embed_workspace(h.numel(), a.bins, 0)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=h.device)
        a.z, a.z_amax, a.workspace, a.workspace_bytes = z.data_ptr(), amax.data_ptr(), ws.data_ptr(), nbytes
        with _lib.device_guard(h):
            _lib.check(self.lib.alignn_angle_embed_fwd(C.byref(a), _lib.stream()), "angle_embed_fwd")
        self._h = h
        return z, amax

    def backward(self, gz: torch.Tensor):
        """-> ((gW1, gb1, red1), (gW2, gb2, red2)), red = [dbeta | dgamma]"""
        h, gz = self._h, gz.contiguous()
        a = self._args(h)
        nbytes = self.lib.alignn_angle_embed_workspace(h.numel(), a.bins, 1)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=h.device)
        a.gz, a.workspace, a.workspace_bytes = gz.data_ptr(), ws.data_ptr(), nbytes
        with _lib.device_guard(h):
            _lib.check(self.lib.alignn_angle_embed_bwd(C.byref(a), _lib.stream()), "angle_embed_bwd")
        return self.grads
