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

import time
from typing import Callable, Dict, Optional, Tuple

import torch

from . import graph as _graph_mod
from .functional import mse_loss, unit_gradient
from .optim import FlatAdam

import torch.nn as nn

from . import _native
from ._native import check, current_stream, lib, ptr

INPUT_FIELDS = ('x_comp', 'f_tensor', 'uu_tensor')
TOPOLOGY_FIELDS = ('edge_index', 'to_boundary_edge_mask', 'to_corner_nodes_mask', 'diff_boundary_edges_mask', 'batch')


class FusedIteration:
    """The training iteration of `src/run_GNN.py:99-131` for a weight-shared GRAND / GRAND_plus model behind the identity encoder as
    13 or 14 launches on preallocated buffers, straight over the C-ABI (no autograd): `gadapt_block_forward_loss` (layer 0 reads the node
    fields, the last layer's launch produces the loss derivative and the loss partials), `gadapt_block_backward`, `gadapt_step_tail`
    (slab sums; chain rule + Adam; the NEXT step's composite coefficients - unless the next forward computes them itself: hidden 64 on
    graphs the wide forward kernel takes, `coeffs_in_forward`).  The per-layer launches are the ones the autograd path
    issues, on the same values in the same order, and the Adam / coefficient arithmetic is shared code: parameters and moments are
    bit-identical to the eager loop; the loss value is summed in another (fixed) order.

    Built by `GraphedTrainStep` (and `bench.py`) when `eligible` says the model, optimizer, loss and batch qualify; everything else
    keeps the captured autograd iteration.  Bound to ONE batch object (its field tensors are read by address).  `coeffs`: the (A, p0)
    buffers shared by all the plans of one model - the tail of every step leaves the coefficients of the updated weights there;
    `refresh_coeffs()` recomputes them after the weights changed by any other means."""

    @staticmethod
    def eligible(model, optimizer, loss_fn, data, target_field: str) -> Optional[str]:
        """None when the fused iteration applies, else the reason it does not."""
        from .functional import l1_loss
        o = model.opt
        dev = torch.device(o['device'])
        if loss_fn not in (mse_loss, l1_loss):
            return 'loss is not the native mse_loss / l1_loss'
        if not (model._fusable() and o['share_conv']):
            return 'not a fusable weight-shared block'
        if o.get('learn_step') or o.get('softmax_temp_type') == 'learnable_a' or o['loss_type'] != 'mesh_loss':
            return 'learnable steps / temperature or a loss other than mesh_loss'
        if o.get('gnn_inc_glob_feat_f') or o.get('gnn_inc_glob_feat_uu') or o.get('gnn_normalize'):
            return 'global features / field normalisation'
        if not (isinstance(model.enc, nn.Linear) and model.enc.bias is None and not model.enc.weight.requires_grad
                and model.enc.weight.shape[1] <= 4 and isinstance(model.dec, nn.Identity)):
            return 'encoder is not a frozen bias-free Linear of at most 4 columns (or the decoder not Identity)'
        x_comp, tgt = data.x_comp, getattr(data, target_field, None)
        if not (torch.is_tensor(x_comp) and x_comp.is_cuda and x_comp.dtype == torch.float32 and x_comp.dim() == 2 and x_comp.is_contiguous()
                and x_comp.shape[1] == model.dim and torch.is_tensor(tgt) and tgt.is_cuda and tgt.dtype == torch.float32 and tgt.is_contiguous()
                and tgt.shape == (x_comp.shape[0], model.dim)):
            return 'x_comp / target are not dense fp32 [N,dim] device tensors'
        for flag, name in (('gnn_inc_feat_f', 'f_tensor'), ('gnn_inc_feat_uu', 'uu_tensor')):
            if o[flag]:
                t = getattr(data, name, None)
                if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.shape == (x_comp.shape[0],) and t.is_contiguous()):
                    return f'{name} is not a dense fp32 [N] device tensor'
        conv = model.conv_layers[0]
        want = [conv.lin_query.weight, conv.lin_query.bias, conv.lin_key.weight, conv.lin_key.bias]
        c = o['hidden_dim']
        if not (isinstance(optimizer, FlatAdam) and optimizer.capturable and optimizer.bucket is not None and optimizer._dev_state is not None
                and optimizer.reduce_op == 'mean' and len(optimizer.active) == 4 and all(a is b for a, b in zip(optimizer.active, want))
                and list(optimizer.offsets) == [0, c * c, c * c + c, 2 * c * c + c]):
            return 'optimizer is not a laid-out FlatAdam(capturable=True) over exactly [Wq | bq | Wk | bk]'
        graph = model._graph(data, x_comp.shape[0], dev)
        with torch.enable_grad():
            plan = model._small_plan(data, graph, x_comp, getattr(data, 'f_tensor', None) if o['gnn_inc_feat_f'] else None,
                                     getattr(data, 'uu_tensor', None) if o['gnn_inc_feat_uu'] else None)
        if plan is not None:                                            # small-mesh batch: the one-launch pair, 4 launches a step
            if plan['part'][1] * 16 > lib().gadapt_loss_partials_max():
                return 'small-mesh batch of more meshes than the loss partials (one per wave) allow'
            return None
        if not (o.get('compact_slots', True) and o['num_layers'] >= 2 and o['hidden_dim'] >= 8 and model._enc_is_zero_pad()):
            return 'per-layer kernels: not >= 2 layers at hidden >= 8 with compact slots behind the zero-pad identity encoder'
        return None

    def __init__(self, model, optimizer: FlatAdam, loss_fn, data, target_field: str, coeffs=None):
        from .functional import l1_loss
        o = model.opt
        self.model, self.optimizer = model, optimizer
        self.l1 = loss_fn is l1_loss
        dev = self.device = torch.device(o['device'])
        self.x_comp, self.target = data.x_comp, getattr(data, target_field)
        self.f = data.f_tensor if o['gnn_inc_feat_f'] else None
        self.uu = data.uu_tensor if o['gnn_inc_feat_uu'] else None
        n, c, L = int(self.x_comp.shape[0]), int(o['hidden_dim']), int(o['num_layers'])
        self.n, self.c, self.L, self.d = n, c, L, int(model.dim)
        self.graph = model._graph(data, n, dev)
        e = max(self.graph.num_edges, 1)
        f32 = dict(device=dev, dtype=torch.float32)
        self.lp = model._layer_params(dev).contiguous()
        self._fw = self._bw = self._tl = None                           # argument lists of the C-ABI calls (built on first use: _plans)
        self._bucket_ptr = None
        with torch.enable_grad():
            self.small = model._small_plan(data, self.graph, self.x_comp, self.f, self.uu)
        if self.small is not None:
            self._init_small(optimizer, f32)
            return
        # activations: slot l = input of layer l (slot 0: the compact [N,4] rows at its start); the last layer writes the head only
        self.x_all = torch.empty(L, n, c, **f32)
        self.alpha = torch.empty(L, e, **f32)
        self.x_top4 = torch.empty(n, 4, **f32)
        self.seed = torch.empty(n, self.d, **f32)
        self.partials = torch.zeros(lib().gadapt_loss_partials_max(), **f32)
        self.loss = torch.zeros((), **f32)
        self.g_ws, self.dxd_ws, self.edge_ws = torch.empty(2, n, c, **f32), torch.empty(n, c, **f32), torch.empty(e, 2, **f32)
        self.slab_rows = lib().gadapt_backward_slab_rows(n, c)
        self.slab = torch.empty(lib().gadapt_backward_slab_floats(n, c), **f32)
        self.scratch = torch.empty(32 * (c * c + c), **f32)
        self.flat = torch.empty(2 * c * c + 2 * c, **f32)                # [dWq | dbq | dWk | dbk]: the parameters' .grad are views of it
        self.coeffs = coeffs if coeffs is not None else (torch.empty(c, c, **f32), torch.empty(c, **f32))
        cuts = [0, c * c, c * c + c, 2 * c * c + c, 2 * c * c + 2 * c]
        self.grads = [(p, self.flat[cuts[k]:cuts[k + 1]].view_as(p)) for k, p in enumerate(optimizer.active)]
        self.out = self.x_top4[:, :self.d]
        # hidden 64 on a graph the wide forward takes: the layer-0 launch computes (A, p0) from the live weights itself - no coefficient
        # launch in the step (13 launches), and nothing to refresh when the weights change behind the step's back
        self.coeffs_in_forward = bool(lib().gadapt_forward_computes_coeffs(self.graph.c_ref, c))

    def _init_small(self, optimizer, f32):
        """Small-mesh batches (csrc/gadapt_smallmesh.inc): the whole forward is ONE launch - encoder, composite coefficients, every
        layer, the head, and with `gadapt_small_forward_loss` the loss derivative and partial sums - the backward another, then the slab
        sums and the chain rule + Adam launch: 4 launches a step (the captured autograd iteration: 7)."""
        n, c, L, pl = self.n, self.c, self.L, self.small
        e = max(self.graph.num_edges, 1)
        n_meshes = pl['part'][1]
        self.x_all, self.alpha = torch.empty(L, n, c, **f32), torch.empty(L, e, **f32)
        self.out = torch.empty(n, self.d, **f32)
        self.seed = torch.empty(n, self.d, **f32)
        self.partials = torch.zeros(lib().gadapt_loss_partials_max(), **f32)
        self.loss = torch.zeros((), **f32)
        self.slab_rows, self.slab = n_meshes, torch.empty(n_meshes * (c * c + c), **f32)
        self.scratch = torch.empty(32 * (c * c + c), **f32)
        self.flat = torch.empty(2 * c * c + 2 * c, **f32)
        cuts = [0, c * c, c * c + c, 2 * c * c + c, 2 * c * c + 2 * c]
        self.grads = [(p, self.flat[cuts[k]:cuts[k + 1]].view_as(p)) for k, p in enumerate(optimizer.active)]
        self.coeffs, self.coeffs_in_forward = None, True               # the forward launch computes them: nothing to keep or refresh
        self.enc_w = pl['enc_w']
        if pl['store']:                                                 # GRAND_plus.py:253-256, :381: the layers show the stored attention
            for l, layer in enumerate(self.model.conv_layers):
                layer.stored_ei, layer._stored = self.graph.edge_index, (self.graph, self.alpha[l])

    def refresh_coeffs(self):
        """(A, p0) of the parameters as they are NOW (one launch): before the first step, and after any change of the weights that did
        not come from `finish()` (an eager optimizer step, `load_state_dict`, a restored snapshot)."""
        if self.small is not None:
            return
        b, c = self.optimizer.bucket, self.c
        check(lib().gadapt_coeffs_forward(ptr(b), ptr(b[c * c:]), ptr(b[c * c + c:]), ptr(self.coeffs[0]), ptr(self.coeffs[1]), c, current_stream(self.device)),
              'gadapt_coeffs_forward')

    def _world(self) -> int:
        import torch.distributed as dist
        o = self.optimizer
        if o.data_parallel and (o.group is not None or (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)):
            return dist.get_world_size(o.group)
        return 1

    def _plans(self):
        """The argument lists of the step's C-ABI calls, built once: every pointer in them is fixed for the life of this object (its own
        buffers, the static batch's fields, the optimizer's laid-out bucket and moments) - per call only the stream, the number of loss
        partials and the optimizer's hyper-parameters are filled in.  (Issued rather than replayed, a step's host cost is what is left of
        the launch path: 34 us with the lists rebuilt per call, `data_ptr()` by `data_ptr()`.)"""
        o, c, L, L_ = self.optimizer, self.c, self.L, lib()
        b = o.bucket                                                    # [Wq | bq | Wk | bk]: the live parameters
        self._bucket_ptr = b.data_ptr()
        if self.small is not None:
            mesh_ptr, n_meshes, max_nodes, max_edges = self.small['part']
            wq, bq, wk = ptr(b), ptr(b[c * c:]), ptr(b[c * c + c:])
            self._fw = (L_.gadapt_small_forward_loss, [self.graph.c_ref, ptr(mesh_ptr[0]), ptr(mesh_ptr[1]), n_meshes, max_nodes, max_edges,
                                                       ptr(self.x_comp), self.d, ptr(self.f), ptr(self.uu), ptr(self.enc_w), self.enc_w.shape[1],
                                                       wq, bq, wk, 0, 0, ptr(self.lp), L, ptr(self.out), self.d, ptr(self.alpha), ptr(self.x_all),
                                                       ptr(self.target), int(self.l1), ptr(self.seed), ptr(self.partials), c, None], 'gadapt_small_forward_loss')
            self._bw = (L_.gadapt_small_backward, [self.graph.c_ref, ptr(mesh_ptr[0]), ptr(mesh_ptr[1]), n_meshes, max_nodes, max_edges, ptr(self.x_all),
                                                   ptr(self.alpha), ptr(self.seed), self.d, wq, bq, wk, 0, 0, ptr(self.lp), L, ptr(self.slab), c, None],
                        'gadapt_small_backward')
            a_ptr = p0_ptr = None
        else:
            a, p0 = self.coeffs
            self._fw = (L_.gadapt_block_forward_loss, [self.graph.c_ref, ptr(self.x_all), ptr(self.x_comp), self.d, ptr(self.f), ptr(self.uu), L, ptr(a), ptr(p0),
                                                       ptr(b) if self.coeffs_in_forward else None, ptr(self.lp), ptr(self.alpha), ptr(self.x_top4),
                                                       ptr(self.target), self.d, int(self.l1), ptr(self.seed), ptr(self.partials), c, None],
                        'gadapt_block_forward_loss')
            self._bw = (L_.gadapt_block_backward, [self.graph.c_ref, ptr(self.x_all), 4, ptr(self.alpha), ptr(self.seed), self.d, L, ptr(a), 0, ptr(p0), 0,
                                                   ptr(self.lp), ptr(self.g_ws), ptr(self.dxd_ws), ptr(self.edge_ws), ptr(self.slab), None, 0, None, c, None],
                        'gadapt_block_backward')
            a_ptr, p0_ptr = (None, None) if self.coeffs_in_forward else (ptr(a), ptr(p0))
        # gadapt_step_tail in its three forms: (slab given, moments given) = the whole tail; (slab, no moments) = this rank's gradient;
        # (no slab, moments) = the gradient is given (all-reduced)
        def tail(slab, moments):
            return [ptr(self.slab) if slab else None, self.slab_rows, ptr(self.scratch), ptr(b), ptr(self.flat),
                    ptr(o.exp_avg) if moments else None, ptr(o.exp_avg_sq) if moments else None, 0.0, 0.0, 0.0, 0.0, 0.0, ptr(o._dev_state), 1.0,
                    a_ptr, p0_ptr, ptr(self.partials) if slab else None, 0, ptr(self.loss), self.n * self.d, c, None]
        self._tl = {(False, False): tail(True, True), (True, False): tail(True, False), (False, True): tail(False, True)}
        self._fn_tail = L_.gadapt_step_tail

    def forward_backward(self):
        """zero_grad + model(data) + loss + backward: 4 + 7 launches at 4 layers (small-mesh batches: 1 + 1; the gradient of the conv
        parameters is still in the slab: `finish()` sums it).  Data parallel: also the slab sums + chain rule, so that `flat` holds this
        rank's gradient."""
        if self._fw is None or self._bucket_ptr != self.optimizer.bucket.data_ptr():
            self._plans()
        st = current_stream(self.device)
        fn, args, name = self._fw
        args[-1] = st
        self.n_part = fn(*args)
        if self.n_part < 0:
            check(self.n_part, name)
        fn, args, name = self._bw
        args[-1] = st
        rc = fn(*args)
        if rc:
            check(rc, name)
        self.model.end_MLmodel = time.time()                   # GNN.py:301
        if self.optimizer.data_parallel and self._world() > 1:
            self._tail(stop_after_gradient=True)

    def _tail(self, stop_after_gradient=False, gradient_given=False, scale=1.0):
        if self._fw is None or self._bucket_ptr != self.optimizer.bucket.data_ptr():
            self._plans()
        g = self.optimizer.param_groups[0]
        args = self._tl[(stop_after_gradient, gradient_given)]
        args[7], (args[8], args[9]), args[10], args[11], args[13], args[17] = g['lr'], g['betas'], g['eps'], g['weight_decay'], scale, self.n_part
        args[-1] = current_stream(self.device)
        rc = self._fn_tail(*args)
        if rc:
            check(rc, 'gadapt_step_tail')

    def finish(self):
        """optimizer.step(): one GPU - slab sums + chain rule + Adam + next coefficients (2 launches); data parallel - all-reduce of
        the flat gradient, then Adam + next coefficients (1 launch)."""
        world = self._world() if self.optimizer.data_parallel else 1
        if world == 1:
            self._tail()
        else:
            import torch.distributed as dist
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.optimizer.group)
            self._tail(gradient_given=True, scale=1.0 / world)
        self.optimizer.step_count += 1
        self.optimizer.grad_bucket = self.flat

    def run(self):
        self.forward_backward()
        self.finish()
        return self.out, self.loss


class _Captured:
    __slots__ = ('static', 'graph', 'loss', 'out', 'grads', 'fused', 'eager', 'replay_ab')


class GraphedTrainStep:
    """step = GraphedTrainStep(model, optimizer);  for data in loader: loss = step(data)

    `optimizer` is a `FlatAdam(capturable=True)` (step count on the device).  `loss_fn(out, target)` is traced at capture time
    (default: the one-launch native `mse_loss`); `target_field` names the batch attribute holding the target (`x_phys`,
    `src/run_GNN.py:106`).  `capture_optimizer=False` leaves `optimizer.step()` (gradient all-reduce + Adam) outside the graph
    and issues it eagerly after every replay - for process groups whose collectives cannot be stream-captured (gloo); with RCCL
    the collective is captured with the step.

    Returns the STATIC 0-d loss tensor of the captured graph (overwritten by the next call: accumulate or clone it).

    `replay`: how a FUSED step is issued once it is captured.  A fused iteration is three C-ABI calls (13 launches), so the host stays far
    ahead of a step of 100 us and more without a graph, while every hipGraph replay costs about 8 us of idle GPU on this platform
    (docs/measurements.md K: BASELINE config 2 231.6k meshes/s replayed, 242.5k issued).  'auto' (default): one GPU - both ways are timed on
    the static batch when a topology is captured (parameters and optimizer state restored afterwards) and the faster one is kept;
    data parallel - the rule the timing confirms on one GPU, the same on every rank: issue the per-layer form, replay the four-launch
    small-mesh form (its host cost is its GPU time).  'graph' / 'eager' force one.  The autograd iteration is always replayed.

    Under data parallelism every rank must call the step once per iteration (as with any all-reduce); WHEN a rank captures is its
    own business - a capture's warm-up issues no collective.  Hyper-parameters of the captured Adam launch (lr, betas, eps,
    weight_decay) are kernel arguments: a change of `optimizer.param_groups[0]` drops the captures and the next call re-captures.
    """

    def __init__(self, model, optimizer: FlatAdam, loss_fn: Callable = mse_loss, target_field: str = 'x_phys', warmup: int = 2,
                 capture_optimizer: bool = True, max_graphs: int = 4, fused: bool = True, replay: str = 'auto'):
        if replay not in ('auto', 'graph', 'eager'):
            raise ValueError("replay: 'auto', 'graph' or 'eager'")
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
        # fused=True: topologies whose step qualifies (FusedIteration.eligible) are captured as the 13-launch fused iteration instead of
        # the autograd one (same parameters bit for bit).  The composite coefficients (A, p0) then live across steps - every step's
        # tail leaves those of the updated weights - in buffers shared by all captures; `_coeffs_stale` marks them for recomputation
        # whenever the weights change by any other route (eager steps, a capture's warm-up, refresh()).
        self.fused = bool(fused)
        self.replay = replay
        self._coeffs = None
        self._coeffs_stale = True
        self.fused_reason: Optional[str] = None              # why the last capture did not take the fused route (None: it did)
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
        self._coeffs_stale = True
        return loss

    def refresh(self):
        """Tell the step that the model's weights were changed behind its back (`load_state_dict`, an optimizer step issued elsewhere,
        manual edits): the fused captures recompute their composite coefficients before the next replay."""
        self._coeffs_stale = True

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
            self._coeffs_stale = True
            # the fused 13-launch iteration when this topology qualifies (the warm-up has laid the optimizer's bucket out by now)
            c.fused = None
            if self.fused:
                self.fused_reason = FusedIteration.eligible(self.model, self.optimizer, self.loss_fn, c.static, self.target_field)
                if self.fused_reason is None:
                    if self._coeffs is None:
                        cc = int(self.model.opt['hidden_dim'])
                        self._coeffs = (torch.empty(cc, cc, device=dev), torch.empty(cc, device=dev))
                    c.fused = FusedIteration(self.model, self.optimizer, self.loss_fn, c.static, self.target_field, coeffs=self._coeffs)
                    c.fused.refresh_coeffs()
                    for _ in range(self.warmup):             # first launches of these entry points (LDS attributes, ...) outside the capture
                        c.fused.forward_backward()
                        if self.capture_optimizer:
                            dp, self.optimizer.data_parallel = self.optimizer.data_parallel, False
                            try:
                                c.fused.finish()
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
        if c.fused is not None:
            # (coefficients: left OUT of the graph - the first replay and every replay after a foreign weight change is preceded
            # by refresh_coeffs(), all others read what the previous step's tail wrote)
            with torch.cuda.graph(c.graph, stream=self._side, pool=self._pool, capture_error_mode=mode):
                c.fused.forward_backward()
                if self.capture_optimizer:
                    c.fused.finish()
            c.out, c.loss, c.grads = c.fused.out, c.fused.loss, c.fused.grads
        else:
            with torch.cuda.graph(c.graph, stream=self._side, pool=self._pool, capture_error_mode=mode):
                c.out, c.loss = self._iteration(c.static)
            # the .grad tensors this graph writes (views of its flat gradient tensor): an eager optimizer.step() after a replay must
            # read THESE, not the ones a later capture of another topology left on the parameters
            c.grads = [(p, p.grad) for p in self.optimizer.params]
        if self._pool is None:
            self._pool = c.graph.pool()                      # later captures share the private pool (one step runs at a time)
        c.eager, c.replay_ab = False, None
        if c.fused is not None and self.replay != 'graph':
            if self.replay == 'eager' or multi:
                c.eager = self.replay == 'eager' or c.fused.small is None
            else:
                c.eager = self._issue_beats_replay(c)
        return c

    def _run_captured(self, c):
        """One step of a capture: the replay, or - fused steps that are faster issued (`replay`) - its launches."""
        if c.eager:
            c.fused.forward_backward()
            if self.capture_optimizer:
                c.fused.finish()
        else:
            c.graph.replay()

    def _issue_beats_replay(self, c, n: int = 12) -> bool:
        """Times n steps replayed and n steps issued on the static batch (wall clock around a synchronisation: the host's share is the
        point), restores parameters and optimizer state, returns True when issuing is at least 1 % faster (the timings repeat to about 0.3 %) and queues a step in at most 60 % of its GPU time."""
        dev = self.device
        snap = self._snapshot()
        dp, self.optimizer.data_parallel = self.optimizer.data_parallel, False
        times, host_share = {}, 1.0
        try:
            for eager in (False, True, False, True):
                c.eager = eager
                if not c.fused.coeffs_in_forward:
                    c.fused.refresh_coeffs()
                self._run_captured(c)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(n):
                    self._run_captured(c)
                    if not self.capture_optimizer:
                        c.fused.finish()
                t_issue = time.perf_counter() - t0                # the host's part: everything is queued
                torch.cuda.synchronize(dev)
                times[eager] = min(times.get(eager, 1e9), (time.perf_counter() - t0) / n)
                if eager:
                    host_share = min(host_share, t_issue / max(time.perf_counter() - t0, 1e-9))
        finally:
            self.optimizer.data_parallel = dp
            c.eager = False
            self._restore(snap)
            self._coeffs_stale = True
        c.replay_ab = {'replayed_ms_per_step': round(times[False] * 1e3, 4), 'issued_ms_per_step': round(times[True] * 1e3, 4),
                       'issued_host_share': round(host_share, 3)}
        # ... and only while the host has room left for what else a loop does per batch (a loader's gather, logging): issuing a step
        # costs the host ~4 us per launch, a replay ~8 us per step
        return times[True] < 0.99 * times[False] and host_share <= 0.6

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
        if c.fused is not None and self._coeffs_stale and not c.fused.coeffs_in_forward:
            c.fused.refresh_coeffs()                         # (A, p0) of the weights as they are now; from here on the tails keep them current
        self._run_captured(c)
        for p, g in c.grads:                                 # `.grad` shows what THIS replay computed (each capture has its own tensors)
            p.grad = g
        if c.fused is not None:
            if not self.capture_optimizer:
                c.fused.finish()                             # all-reduce + Adam (+ next coefficients)
            # a step whose forward computes the coefficients itself leaves none behind for a capture that needs them given
            self._coeffs_stale = c.fused.coeffs_in_forward
        elif not self.capture_optimizer:
            self.optimizer.step()                            # all-reduce + fused Adam
            self._coeffs_stale = True
        else:
            self._coeffs_stale = True
        return c.loss

    def static_batch(self, data):
        """The static batch object of `data`'s topology (captured on first use).  A loader that writes its node fields straight
        into these tensors (`DeviceMeshLoader(..., into=step.static_batch)`) saves the per-step copies."""
        return self._lookup(data).static

    def owns(self, static) -> bool:
        """True while `static` is the static batch of a live capture (a loader holding an evicted one must ask again)."""
        return any(c.static is static for c in self._captured.values())
