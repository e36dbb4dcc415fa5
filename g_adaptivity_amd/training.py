"""The reference's training iteration as one replayed hipGraph, for batches that change every step.

`src/run_GNN.py:99-131` runs, per batch:  optimizer.zero_grad(); out = model(data); loss = loss_fn(out, data.x_phys);
loss.backward(); optimizer.step().  Every one of those is a handful of short launches issued from Python; at the sizes the
path runs (0.3 ms of GPU work per step on the metric workload) the eager loop is bound by the host's launch path (three times the
step time `bench.py` reports for its replayed graph).  `GraphedTrainStep` captures that iteration ONCE per batch topology over
static input buffers (`x_comp`, `f_tensor`, `uu_tensor` and the target `x_phys`); every later call copies the new batch's node
fields into those buffers and replays.  What a replay executes is exactly the launch sequence of the eager iteration, so N
replayed steps leave bit-identical parameters to N eager steps (`tests/test_gpu_training.py`).

The topology (edge list, masks, node count) is fixed per captured graph: a batch with another topology - e.g. the short last batch
of an epoch - gets its own capture, keyed like the model's CSR cache (`graph.content_fingerprint`, memoised per tensor object, so
loaders that share the topology tensors between batches pay a dictionary lookup).  `inference.GraphedForward` is the evaluation
counterpart.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch

from . import graph as _graph_mod
from .functional import mse_loss, unit_gradient
from .optim import FlatAdam

INPUT_FIELDS = ('x_comp', 'f_tensor', 'uu_tensor')
TOPOLOGY_FIELDS = ('edge_index', 'to_boundary_edge_mask', 'to_corner_nodes_mask', 'diff_boundary_edges_mask', 'batch')


class _Captured:
    __slots__ = ('static', 'graph', 'loss', 'out', 'grads')


class GraphedTrainStep:
    """step = GraphedTrainStep(model, optimizer);  for data in loader: loss = step(data)

    `optimizer` is a `FlatAdam(capturable=True)` (step count on the device).  `loss_fn(out, target)` is traced at capture time
    (default: the one-launch native `mse_loss`); `target_field` names the batch attribute holding the target (`x_phys`,
    `src/run_GNN.py:106`).  `capture_optimizer=False` leaves `optimizer.step()` (gradient all-reduce + Adam) outside the graph
    and issues it eagerly after every replay - for process groups whose collectives cannot be stream-captured (gloo); with RCCL
    the collective is captured with the step.

    Returns the STATIC 0-d loss tensor of the captured graph (overwritten by the next call: accumulate or clone it).

    Under data parallelism every rank must call the step once per iteration (as with any all-reduce); WHEN a rank captures is its
    own business - a capture's warm-up issues no collective.  Hyper-parameters of the captured Adam launch (lr, betas, eps,
    weight_decay) are kernel arguments: a change of `optimizer.param_groups[0]` drops the captures and the next call re-captures.
    """

    def __init__(self, model, optimizer: FlatAdam, loss_fn: Callable = mse_loss, target_field: str = 'x_phys', warmup: int = 2,
                 capture_optimizer: bool = True, max_graphs: int = 4):
        if not isinstance(optimizer, FlatAdam) or not optimizer.capturable:
            raise TypeError("GraphedTrainStep needs FlatAdam(capturable=True): the step count must live on the device")
        if not model.training:
            raise RuntimeError("GraphedTrainStep captures the training iteration: call model.train() first")
        import torch.distributed as dist
        if (capture_optimizer and optimizer.data_parallel and dist.is_available() and dist.is_initialized() and dist.get_world_size(optimizer.group) > 1
                and dist.get_backend(optimizer.group) != 'nccl'):
            # a collective that synchronises with the host invalidates the capture, and that is not recoverable in-process on this
            # ROCm (tools/capture_recovery_probe.py): refuse up front instead
            raise ValueError(f"GraphedTrainStep(capture_optimizer=True): the '{dist.get_backend(optimizer.group)}' all-reduce of optimizer.step() "
                             "cannot be captured in a hipGraph (only RCCL collectives can); pass capture_optimizer=False")
        self.model, self.optimizer, self.loss_fn = model, optimizer, loss_fn
        self.target_field, self.warmup = target_field, max(int(warmup), 1)
        self.capture_optimizer = bool(capture_optimizer)
        self.device = torch.device(model.opt['device'])
        self.max_graphs = max_graphs
        self._captured: Dict[Tuple, _Captured] = {}          # insertion order = recency (see _lookup)
        self._hyper_captured: Optional[Tuple] = None
        self._side = torch.cuda.Stream(device=self.device)
        # the native one-launch losses take the preallocated root gradient (functional.unit_gradient): two launches fewer
        from .functional import l1_loss
        self._root = unit_gradient(self.device) if loss_fn in (mse_loss, l1_loss) else None
        self._pool = None
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)   # captured on a side stream by design

    # ------------------------------------------------------------------ the iteration (what eager code would run)
    def _iteration(self, data):
        self.optimizer.zero_grad()
        out = self.model(data)
        loss = self.loss_fn(out, getattr(data, self.target_field))
        if self._root is not None:
            loss.backward(gradient=self._root)
        else:
            loss.backward()
        if self.capture_optimizer:
            self.optimizer.step()
        return out, loss

    def eager(self, data) -> torch.Tensor:
        """The same iteration as plain launches (no capture): what `__call__` is compared with."""
        _, loss = self._iteration(data)
        if not self.capture_optimizer:
            self.optimizer.step()
        return loss

    # ------------------------------------------------------------------ capture
    def _snapshot(self):
        o = self.optimizer
        params = [p.detach().clone() for p in o.params]
        state = None
        if o.bucket is not None:
            state = (o.exp_avg.clone(), o.exp_avg_sq.clone(), None if o._dev_state is None else o._dev_state.clone(), o.step_count)
        return params, state

    def _restore(self, snap):
        """Undo the warm-up iterations: parameters, moments and step count as they were (a capture must not train)."""
        o = self.optimizer
        params, state = snap
        with torch.no_grad():
            for p, s_ in zip(o.params, params):
                p.data.copy_(s_)
            if state is None:                                # the optimizer was laid out by the warm-up itself: fresh state
                o.exp_avg.zero_(); o.exp_avg_sq.zero_()
                if o._dev_state is not None:
                    o._dev_state.zero_()
                o.step_count = 0
            else:
                o.exp_avg.copy_(state[0]); o.exp_avg_sq.copy_(state[1])
                if state[2] is not None:
                    o._dev_state.copy_(state[2])
                o.step_count = state[3]

    def _key(self, data) -> Tuple:
        tensors = [getattr(data, k) for k in TOPOLOGY_FIELDS if getattr(data, k, None) is not None]
        corners = getattr(data, 'corner_nodes', None)
        n_corner = 0 if corners is None else sum(len(c) for c in corners) if isinstance(corners, (list, tuple)) else len(corners)
        return (int(data.x_comp.shape[0]), n_corner, _graph_mod.content_fingerprint(tensors))

    def _capture(self, data) -> _Captured:
        dev = self.device
        c = _Captured()
        c.static = data.clone().to(dev)                      # the graph reads these tensors by address
        # clone() drops '_'-prefixed attributes; without the graph count the model would read `data.batch.max().item()` - a host
        # synchronisation - inside the capture (1-D models with global features have no corner list to count instead)
        if getattr(c.static, '_num_graphs', None) is None and getattr(data, 'batch', None) is not None:
            c.static._num_graphs = int(data.num_graphs)      # (one synchronisation, here, outside the capture)
        cur = torch.cuda.current_stream(dev)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            snap = self._snapshot()
            # The warm-up runs WITHOUT the gradient all-reduce: its results are discarded (_restore), and a rank that meets a new
            # topology - or re-captures one that was evicted - must not issue collectives its peers (replaying a cached graph)
            # do not issue: they would pair with the wrong step.  Collectives stay one per call of __call__ on every rank.
            dp, self.optimizer.data_parallel = self.optimizer.data_parallel, False
            try:
                for _ in range(self.warmup):                 # CSR cache, flat bucket, allocator: outside the capture
                    self._iteration(c.static)
                    if not self.capture_optimizer:
                        self.optimizer.step()
            finally:
                self.optimizer.data_parallel = dp
            self._restore(snap)
        cur.wait_stream(self._side)
        torch.cuda.synchronize(dev)
        c.graph = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad()
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        # other threads of the process (the RCCL watchdog) make HIP calls while this one captures: thread_local keeps them out
        mode = 'thread_local' if multi else 'global'
        with torch.cuda.graph(c.graph, stream=self._side, pool=self._pool, capture_error_mode=mode):
            c.out, c.loss = self._iteration(c.static)
        # the .grad tensors this graph writes (views of its flat gradient tensor): an eager optimizer.step() after a replay must
        # read THESE, not the ones a later capture of another topology left on the parameters
        c.grads = [(p, p.grad) for p in self.optimizer.params]
        if self._pool is None:
            self._pool = c.graph.pool()                      # later captures share the private pool (one step runs at a time)
        return c

    # ------------------------------------------------------------------ per step
    def _hyper(self) -> Tuple:
        """Optimizer hyper-parameters a captured Adam launch carries BY VALUE (kernel arguments of gadapt_adam_step_dev)."""
        g = self.optimizer.param_groups[0]
        return (float(g['lr']), tuple(float(b) for b in g['betas']), float(g['eps']), float(g['weight_decay']))

    def _lookup(self, data) -> _Captured:
        """The capture of `data`'s topology: least-recently-used eviction at `max_graphs`; captures whose Adam launch was recorded
        with other hyper-parameters than the optimizer has now (an LR scheduler stepped, `param_groups` edited) are dropped first -
        a replay would silently keep the old values."""
        if self.capture_optimizer and self._captured:
            hyper = self._hyper()
            if hyper != self._hyper_captured:
                self._captured.clear()
        key = self._key(data)
        c = self._captured.pop(key, None)
        if c is None:
            while len(self._captured) >= max(self.max_graphs, 1):
                self._captured.pop(next(iter(self._captured)))       # the least recently used one
            self._hyper_captured = self._hyper()
            c = self._capture(data)
        self._captured[key] = c                                      # (re-)inserted last = most recently used
        return c

    def __call__(self, data) -> torch.Tensor:
        c = self._lookup(data)
        if data is not c.static:
            for name in INPUT_FIELDS + (self.target_field,):
                src = getattr(data, name, None)
                dst = getattr(c.static, name, None)
                if src is None or dst is None or src is dst:
                    continue
                if src.data_ptr() != dst.data_ptr():         # loaders bound with `static_batch` write in place
                    dst.copy_(src.reshape(dst.shape), non_blocking=True)
        c.graph.replay()
        for p, g in c.grads:                                 # `.grad` shows what THIS replay computed (each capture has its own tensors)
            p.grad = g
        if not self.capture_optimizer:
            self.optimizer.step()                            # all-reduce + fused Adam
        return c.loss

    def static_batch(self, data):
        """The static batch object of `data`'s topology (captured on first use).  A loader that writes its node fields straight
        into these tensors (`DeviceMeshLoader(..., into=step.static_batch)`) saves the per-step copies."""
        return self._lookup(data).static

    def owns(self, static) -> bool:
        """True while `static` is the static batch of a live capture (a loader holding an evicted one must ask again)."""
        return any(c.static is static for c in self._captured.values())
