"""Flat-bucket Adam + data-parallel gradient all-reduce for the training loop's step.

`torch.optim.Adam(model.parameters(), lr, weight_decay)` followed by `.step()` is what the
reference runs after backward (`src/run_GNN.py:88,126-131`).  Here the parameters that receive
gradients live in ONE contiguous fp32 bucket (their `.data` / `.grad` are views of it), so a
step is one native kernel (`gadapt_adam_step`) and, under data parallelism, one RCCL
all-reduce of the bucket (2(C^2+C) floats with shared weights) with the
1/world scaling folded into the Adam kernel.  Parameters that never receive a gradient
(`lin_skip.weight`, `enc.weight`) stay outside the bucket, exactly as torch's Adam skips
`grad is None`.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch

from ._native import check, current_stream, lib, ptr


def shard_range(n_items: int, rank: int, world: int):
    """[lo, hi) of the meshes rank `rank` owns: contiguous, sizes differ by at most one."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class FlatAdam:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, process_group=None, reduce_op: str = 'mean'):
        seen, self.params = set(), []
        for p in params:                                   # shared convs repeat the same Parameter
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                self.params.append(p)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.group, self.reduce_op = process_group, reduce_op
        self.step_count = 0
        self.bucket: Optional[torch.Tensor] = None
        self.grad_bucket: Optional[torch.Tensor] = None
        self.active: List[torch.nn.Parameter] = []

    def _build(self):
        self.active = [p for p in self.params if p.grad is not None]
        if not self.active:
            raise RuntimeError("FlatAdam.step() before any backward()")
        dev = self.active[0].device
        n = sum(p.numel() for p in self.active)
        self.bucket = torch.empty(n, device=dev, dtype=torch.float32)
        self.grad_bucket = torch.empty(n, device=dev, dtype=torch.float32)
        off = 0
        for p in self.active:
            k = p.numel()
            self.bucket[off:off + k].copy_(p.data.reshape(-1))
            self.grad_bucket[off:off + k].copy_(p.grad.reshape(-1))
            p.data = self.bucket[off:off + k].view_as(p)
            p.grad = self.grad_bucket[off:off + k].view_as(p)
            off += k
        self.exp_avg = torch.zeros_like(self.bucket)
        self.exp_avg_sq = torch.zeros_like(self.bucket)

    def zero_grad(self, set_to_none: bool = False):
        if self.grad_bucket is None:
            for p in self.params:
                p.grad = None
        else:
            self.grad_bucket.zero_()                       # views stay attached: autograd accumulates in place

    def all_reduce(self):
        """SUM over ranks on the flat bucket (one RCCL collective over xGMI)."""
        import torch.distributed as dist
        if self.group is not None or (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            dist.all_reduce(self.grad_bucket, op=dist.ReduceOp.SUM, group=self.group)
            return dist.get_world_size(self.group)
        return 1

    def step(self):
        if self.bucket is None:
            self._build()
        world = self.all_reduce()
        scale = 1.0 / world if self.reduce_op == 'mean' else 1.0       # 'sum' for the modular pseudo-loss (run_GNN.py:118)
        self.step_count += 1
        b = self.bucket
        check(lib().gadapt_adam_step(ptr(b), ptr(self.grad_bucket), ptr(self.exp_avg), ptr(self.exp_avg_sq), b.numel(),
                                     self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                     self.step_count, scale, current_stream(b.device)), 'gadapt_adam_step')
