"""Flat-bucket Adam + data-parallel gradient all-reduce for the training loop's step.

`torch.optim.Adam(model.parameters(), lr, weight_decay)` followed by `.step()` is what the
reference runs after backward (`src/run_GNN.py:88,126-131`).  Here the parameters that receive
gradients live in ONE contiguous fp32 bucket (their `.data` are views of it, their `.grad` are views
of the flat gradient tensor the block op returns), so a
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
    """Adam over ONE flat fp32 bucket.

    The block op returns its weight gradients as slices of one freshly allocated flat tensor
    (`functional._GrandEulerBlock.backward`), which autograd installs as `.grad` without a copy when `.grad` is
    None (zero_grad's default).  `step()` then finds every gradient at its expected offset of one storage and
    hands that memory to the kernel as the gradient bucket: no per-parameter accumulate kernels, no zero fill.
    Any other situation (gradients accumulated over several backward calls, parameters from elsewhere) is
    handled by gathering the gradients into an own bucket first.

    Restriction against `torch.optim.Adam` (which keeps a step count and moments PER parameter and simply skips a parameter
    whose `.grad` is None on a given step): one flat bucket shares ONE step count.  A parameter that drops out of the
    gradient set (e.g. `zero_grad(set_to_none=True)` plus a branch that does not use it this step) is dropped from the
    bucket with its moments, and its return raises `RuntimeError` instead of silently giving it another parameter's bias
    correction.  Every parameter the reference's `GNN` trains (`lin_query/lin_key` weights and biases, `steps`,
    `sm_temp_a`, the CNN extractors) receives a gradient on every step, so the reference's loop never meets this; a model
    with conditionally used parameters needs `torch.optim.Adam` (or one FlatAdam per always-together parameter set).
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, process_group=None, reduce_op: str = 'mean', capturable: bool = False,
                 data_parallel: bool = True):
        seen, self.params = set(), []
        for p in params:                                   # shared convs repeat the same Parameter
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                self.params.append(p)
        # torch.optim-style view of the hyper-parameters: LR schedulers and user code read / write `param_groups[0]['lr']`
        self.param_groups = [{'params': self.params, 'lr': lr, 'betas': tuple(betas), 'eps': eps, 'weight_decay': weight_decay}]
        self.defaults = {'lr': lr, 'betas': tuple(betas), 'eps': eps, 'weight_decay': weight_decay}
        self.group, self.reduce_op = process_group, reduce_op
        self.data_parallel = bool(data_parallel)        # False: never all-reduce, even inside an initialised process group
        self.step_count = 0
        # capturable: the step count lives on the device (gadapt_adam_step_dev), so step() can be captured in a hipGraph
        # with forward and backward and replayed (torch.optim.Adam(capturable=True) does the same)
        self.capturable = bool(capturable)
        self._dev_state: Optional[torch.Tensor] = None
        self.bucket: Optional[torch.Tensor] = None
        self.grad_bucket: Optional[torch.Tensor] = None    # the flat gradient the last step()/_build() used
        self._own_grad: Optional[torch.Tensor] = None
        self.active: List[torch.nn.Parameter] = []
        self.offsets: List[int] = []
        self._active_ids = frozenset()

    # hyper-parameters live in param_groups[0] (one group: the reference builds Adam(model.parameters(), lr, weight_decay))
    lr = property(lambda self: self.param_groups[0]['lr'], lambda self, v: self.param_groups[0].__setitem__('lr', v))
    betas = property(lambda self: self.param_groups[0]['betas'], lambda self, v: self.param_groups[0].__setitem__('betas', tuple(v)))
    eps = property(lambda self: self.param_groups[0]['eps'], lambda self, v: self.param_groups[0].__setitem__('eps', v))
    weight_decay = property(lambda self: self.param_groups[0]['weight_decay'],
                            lambda self, v: self.param_groups[0].__setitem__('weight_decay', v))

    @staticmethod
    def _shared_offsets(params) -> Optional[List[int]]:
        """Element offsets of every .grad inside one storage if they tile a gap-free range of it, else None."""
        g0 = params[0].grad
        if g0 is None:
            return None
        store = g0.untyped_storage().data_ptr()
        offs = []
        for p in params:
            g = p.grad
            if (g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape
                    or g.untyped_storage().data_ptr() != store):
                return None
            offs.append(g.storage_offset())
        order = sorted(range(len(params)), key=lambda i: offs[i])
        pos = offs[order[0]]
        for i in order:
            if offs[i] != pos:
                return None                                # gap or overlap
            pos += params[i].numel()
        return offs

    def _build(self, carry=None):
        """Lay the parameters that have gradients out in one bucket.  `carry`: (old active list, old offsets, exp_avg,
        exp_avg_sq) of a previous layout whose moments are kept for the parameters that stay."""
        self.active = [p for p in self.params if p.grad is not None]
        if not self.active:
            raise RuntimeError("FlatAdam.step() before any backward()")
        dev = self.active[0].device
        shared = self._shared_offsets(self.active)
        if shared is not None:                             # bucket laid out like the gradient storage
            base = min(shared)
            self.active = [p for _, p in sorted(zip(shared, self.active), key=lambda t: t[0])]
            self.offsets = sorted(o - base for o in shared)
        else:
            self.offsets, off = [], 0
            for p in self.active:
                self.offsets.append(off)
                off += p.numel()
        n = sum(p.numel() for p in self.active)
        self.bucket = torch.empty(n, device=dev, dtype=torch.float32)
        for p, off in zip(self.active, self.offsets):
            k = p.numel()
            self.bucket[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.bucket[off:off + k].view_as(p)
        self.exp_avg = torch.zeros_like(self.bucket)
        self.exp_avg_sq = torch.zeros_like(self.bucket)
        if carry is not None:
            old_active, old_offsets, old_m, old_v = carry
            where = {id(p): off for p, off in zip(old_active, old_offsets)}
            for p, off in zip(self.active, self.offsets):
                o = where.get(id(p))
                if o is not None:
                    k = p.numel()
                    self.exp_avg[off:off + k].copy_(old_m[o:o + k])
                    self.exp_avg_sq[off:off + k].copy_(old_v[o:o + k])
        self._own_grad = None
        self._active_ids = frozenset(id(p) for p in self.active)
        self.grad_bucket = self._flat_grad()

    def _flat_grad(self) -> torch.Tensor:
        """The gradients as one flat tensor in bucket order: a view when they already are one, else a gathered copy."""
        g0 = self.active[0].grad
        if g0 is not None:
            store, start = g0.untyped_storage().data_ptr(), g0.storage_offset() - self.offsets[0]
            if start >= 0 and all(
                    p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous() and p.grad.shape == p.shape
                    and p.grad.untyped_storage().data_ptr() == store and p.grad.storage_offset() == start + off
                    for p, off in zip(self.active, self.offsets)):
                return g0.as_strided((self.bucket.numel(),), (1,), start)
        if self._own_grad is None:
            self._own_grad = torch.zeros_like(self.bucket)
        for p, off in zip(self.active, self.offsets):
            dst = self._own_grad[off:off + p.numel()]
            dst.copy_(p.grad.reshape(-1))
        return self._own_grad

    def state_dict(self) -> dict:
        """Checkpointable state (the reference keeps none: `src/run_GNN.py:140-152` tracks the model only): step count,
        both moments and the bucket layout by position in the constructor's parameter list."""
        if self.capturable and self._dev_state is not None:
            self.step_count = int(self._dev_state[0].item())
        index = {id(p): k for k, p in enumerate(self.params)}
        return {'step': self.step_count,
                'param_groups': [{k: v for k, v in self.param_groups[0].items() if k != 'params'}],
                'active': [index[id(p)] for p in self.active], 'offsets': list(self.offsets),
                'exp_avg': None if self.bucket is None else self.exp_avg.detach().clone(),
                'exp_avg_sq': None if self.bucket is None else self.exp_avg_sq.detach().clone()}

    def load_state_dict(self, state: dict):
        self.param_groups[0].update(state['param_groups'][0])
        self.step_count = int(state['step'])
        if state['exp_avg'] is None:
            return
        want = [self.params[k] for k in state['active']]
        dev = want[0].device
        # lay the bucket out exactly as it was saved; the parameters' current values move into it
        self.active, self.offsets = want, list(state['offsets'])
        n = sum(p.numel() for p in want)
        self.bucket = torch.empty(n, device=dev, dtype=torch.float32)
        for p, off in zip(self.active, self.offsets):
            k = p.numel()
            self.bucket[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.bucket[off:off + k].view_as(p)
        self.exp_avg = state['exp_avg'].to(dev, torch.float32).clone()
        self.exp_avg_sq = state['exp_avg_sq'].to(dev, torch.float32).clone()
        self._own_grad, self.grad_bucket = None, None
        self._active_ids = frozenset(id(p) for p in self.active)
        if self.capturable:
            self._dev_state = torch.tensor([self.step_count, 0, 0, 0], device=dev, dtype=torch.int32)

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if set_to_none or p.grad is None:
                p.grad = None
            else:
                p.grad.zero_()

    def all_reduce(self):
        """SUM over ranks on the flat bucket (one RCCL collective over xGMI)."""
        import torch.distributed as dist
        if not self.data_parallel:
            return 1
        if self.group is not None or (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            dist.all_reduce(self.grad_bucket, op=dist.ReduceOp.SUM, group=self.group)
            return dist.get_world_size(self.group)
        return 1

    def step(self):
        if self.bucket is None:
            self._build()
        elif torch.cuda.is_current_stream_capturing() if torch.cuda.is_available() else False:
            self.grad_bucket = self._flat_grad()           # a captured step replays one fixed layout
        else:
            have = frozenset(id(p) for p in self.params if p.grad is not None)
            if have != self._active_ids:
                # The set of parameters with gradients changed (a module unfrozen later, `steps` / temperature switched on,
                # a parameter that got no gradient this time).  torch.optim.Adam updates exactly the parameters whose
                # .grad is not None, each with its own step count; one flat bucket has ONE step count, so the layout is
                # rebuilt for the new set (moments of the parameters that stay are kept) - loudly if that would give a
                # newcomer the bias correction of an older step.
                if not have:
                    raise RuntimeError("FlatAdam.step(): no parameter has a gradient")
                if self.capturable:
                    raise RuntimeError("FlatAdam(capturable=True): the set of parameters with gradients changed after the "
                                       "bucket was laid out; a captured step cannot follow it")
                if self.step_count > 0 and not have <= self._active_ids:
                    raise RuntimeError("FlatAdam: a parameter entered the gradient set after step %d (its first gradient, or "
                                       "its return after a step without one); one flat bucket shares one step count (bias "
                                       "correction), so build a new optimizer for the new parameter set" % self.step_count)
                self._build(carry=(self.active, self.offsets, self.exp_avg, self.exp_avg_sq))
            else:
                self.grad_bucket = self._flat_grad()
        world = self.all_reduce()
        scale = 1.0 / world if self.reduce_op == 'mean' else 1.0       # 'sum' for the modular pseudo-loss (run_GNN.py:118)
        self.step_count += 1
        b = self.bucket
        if self.capturable:
            if self._dev_state is None:
                self._dev_state = torch.zeros(4, device=b.device, dtype=torch.int32)    # {steps, ticket / barrier counters}: gadapt_hip.h
            check(lib().gadapt_adam_step_dev(ptr(b), ptr(self.grad_bucket), ptr(self.exp_avg), ptr(self.exp_avg_sq), b.numel(),
                                             self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                             ptr(self._dev_state), scale, current_stream(b.device)), 'gadapt_adam_step_dev')
            return
        check(lib().gadapt_adam_step(ptr(b), ptr(self.grad_bucket), ptr(self.exp_avg), ptr(self.exp_avg_sq), b.numel(),
                                     self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                     self.step_count, scale, current_stream(b.device)), 'gadapt_adam_step')
