"""Readout layers of the DGL shim (test infrastructure only)."""

import torch
from torch import nn


def _segment_ids(g):
    counts = g.batch_num_nodes()
    return torch.repeat_interleave(torch.arange(counts.numel(), device=counts.device), counts), counts


class SumPooling(nn.Module):
    def forward(self, g, feat):
        seg, counts = _segment_ids(g)
        out = torch.zeros((counts.numel(),) + tuple(feat.shape[1:]), dtype=feat.dtype, device=feat.device)
        return out.index_add(0, seg, feat)


class AvgPooling(nn.Module):
    def forward(self, g, feat):
        seg, counts = _segment_ids(g)
        out = torch.zeros((counts.numel(),) + tuple(feat.shape[1:]), dtype=feat.dtype, device=feat.device)
        out = out.index_add(0, seg, feat)
        shape = (-1,) + (1,) * (feat.dim() - 1)
        return out / counts.to(feat.dtype).reshape(shape)
