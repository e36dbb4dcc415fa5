"""Global (per-mesh) CNN features of the PDE fields: `src/feature_extractors.py:6-34` and their use in
`GNN.forward` (`src/GNN.py:240-268`).

A caller either side of the message-passing path, not part of it: four 3x3 convolutions with
SELU on the n x n grid of a field, global average pool -> `global_feat_dim` numbers per mesh, repeated for every node of
that mesh and appended to the node features.  The convolutions are torch `nn.Conv1d/2d` modules (MIOpen on ROCm), with
the reference's parameter names so `state_dict`s interchange; gradients reach them through the encoder input
(`functional.grand_euler_block` returns d/dx0 when x0 requires grad).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


class GlobalFeatureExtractorCNN(nn.Module):
    """`convs.{k}.{weight,bias}`: in -> mid -> ... -> mid -> out, all 3x3 (or 3), stride 1, padding 1."""

    def __init__(self, in_channels: int, mid_channels: int, out_channels: int, dim: int = 2, num_layers: int = 4):
        super().__init__()
        conv = {1: nn.Conv1d, 2: nn.Conv2d}[dim]
        widths = [in_channels] + [mid_channels] * (num_layers - 1) + [out_channels]
        self.convs = nn.ModuleList([conv(a, b, kernel_size=3, stride=1, padding=1) for a, b in zip(widths[:-1], widths[1:])])
        self.global_avg_pool = {1: nn.AdaptiveAvgPool1d(1), 2: nn.AdaptiveAvgPool2d((1, 1))}[dim]

    def forward(self, u: torch.Tensor) -> torch.Tensor:
        u = u / torch.max(torch.abs(u))                     # one scale for the whole batch (feature_extractors.py:29)
        for conv in self.convs:
            u = F.selu(conv(u))
        return self.global_avg_pool(u).flatten(1)           # [B, out_channels]


def field_to_grid(field: torch.Tensor, mapping: Optional[torch.Tensor], mesh_dims: Sequence[int], batch_size: int,
                  dim: int) -> torch.Tensor:
    """Node-ordered field [B*n^d] -> CNN grid [B, n] / [B, n, n] (`src/utils_data.py:125-141`): per mesh, nodes are
    picked in `mapping` order, laid out row-major, then transposed and flipped along the first grid axis."""
    per_mesh = field.reshape(batch_size, -1)
    if dim == 1:
        return per_mesh
    if mapping is not None:
        per_mesh = per_mesh.index_select(1, mapping.to(per_mesh.device))
    return per_mesh.reshape(batch_size, mesh_dims[0], mesh_dims[1]).transpose(1, 2).flip(1)


def expand_to_nodes(per_mesh: torch.Tensor, batch: torch.Tensor) -> torch.Tensor:
    """[B, k] -> [N, k]: every node gets its mesh's row (`repeat_interleave(bincount(batch))`, GNN.py:252-253).  `batch` is the
    sorted mesh index of every node, so the repeat IS the row gather `per_mesh[batch]` - which, unlike `bincount` +
    `repeat_interleave` (data-dependent output size: a host synchronisation), can sit inside a captured hipGraph."""
    return per_mesh.index_select(0, batch)
