"""`GNN(dataset, opt)` with the reference's surface (`src/GNN.py:144-342`), MI355X-native inside.

Same constructor, same `state_dict` keys (`enc.weight`, `conv_layers.{l}.lin_*`, `steps.{l}`),
same `forward(data) -> x_phys [N,dim]`, same `end_MLmodel` stamp - but the per-forward host
work of the reference (two `.item()` syncs, mask surgery, corner-loop Python, `src/GNN.py:192-218`)
is done once per batch topology (`graph.GraphCache`), and the whole residual loop
(`src/GNN.py:273-291`) is one differentiable call into the fused HIP kernels.
"""
from __future__ import annotations

import time
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as Fn
from .conv import GAT_conv, GAT_plus, GCN_conv, GRAND_conv, GRAND_plusConv, TRANS_conv
from .features import GlobalFeatureExtractorCNN, expand_to_nodes, field_to_grid
from . import graph as _graph_mod
from .graph import MeshGraph, prepare_edge_index


def get_nonlin(nonlin_type):
    table = {'relu': nn.ReLU, 'elu': nn.ELU, 'selu': nn.SELU, 'tanh': nn.Tanh, 'sigmoid': nn.Sigmoid,
             'leaky_relu': nn.LeakyReLU, 'identity': nn.Identity}
    if nonlin_type not in table:
        raise NotImplementedError                                        # GNN.py:64
    return table[nonlin_type]()


def get_mlp(in_dim, hid_dim, out_dim, nonlin_type):
    act = get_nonlin(nonlin_type)
    return nn.Sequential(nn.Linear(in_dim, hid_dim), act, nn.Linear(hid_dim, out_dim), act)


def get_enc(opt, in_dim, out_dim, hid_dim=None, nonlin_type='relu'):
    kind = opt['enc']
    if kind == 'identity':
        # frozen selector: copies the first min(in,out) inputs, zero elsewhere (GNN.py:75-90)
        lin = nn.Linear(in_dim, out_dim, bias=False)
        w = torch.zeros(out_dim, in_dim)
        k = min(in_dim, out_dim)
        w[torch.arange(k), torch.arange(k)] = 1.0
        lin.weight.data = w
        lin.weight.requires_grad = False
        return lin
    if kind == 'lin_layer':
        return nn.Linear(in_dim, out_dim)
    if kind == 'MLP':
        return get_mlp(in_dim, in_dim if hid_dim is None else hid_dim, out_dim, nonlin_type)
    raise NotImplementedError                                            # GNN.py:98


def get_dec(opt, in_dim, out_dim, hid_dim=None, nonlin_type='relu'):
    if opt['enc'] == 'identity':                                         # keyed on 'enc' in the reference too (GNN.py:102)
        return nn.Identity()
    return None


def get_conv(opt, conv_type, in_dim, out_dim, feat_dim=None):
    if conv_type == 'GRAND':
        return GRAND_conv(opt, in_dim, out_dim, heads=1)
    if conv_type == 'GRAND_plus':
        return GRAND_plusConv(opt, in_dim, out_dim, global_feat_dim=feat_dim, heads=1, concat=False, beta=False,
                              dropout=0.0, edge_dim=None, bias=False, root_weight=False)
    if conv_type == 'TRANS':
        return TRANS_conv(opt, in_dim, out_dim, heads=1)                 # GNN.py:112-113
    if conv_type == 'GCN':
        return GCN_conv(opt, in_dim, out_dim)                            # GNN.py:109-110
    if conv_type == 'GAT':
        return GAT_conv(opt, in_dim, out_dim, heads=1)                   # GNN.py:110-111
    if conv_type == 'GAT_plus':
        return GAT_plus(opt, in_dim, out_dim)                            # GNN.py:120-121
    raise NotImplementedError                                            # GNN.py:124


def build_conv_list(opt):
    shared = get_conv(opt, opt['conv_type'], opt['hidden_dim'], opt['hidden_dim'], opt['global_feat_dim']) \
        if opt['share_conv'] else None
    layers = [shared if opt['share_conv'] else
              get_conv(opt, opt['conv_type'], opt['hidden_dim'], opt['hidden_dim'], opt['global_feat_dim'])
              for _ in range(opt['num_layers'])]
    return nn.ModuleList(layers)


class MLP(nn.Module):
    """The reference's node-wise baseline model (`src/GNN.py:16-45`, built by `run_GNN.get_model` for `opt['model'] == 'MLP'`,
    `src/run_GNN.py:57-58`): encoder, two residual dense layers on `x_comp`, decoder - no graph, so nothing for the HIP path;
    dense torch layers (rocBLAS on the GPU), here so that `from g_adaptivity_amd import GNN, MLP` replaces the reference's import."""

    def __init__(self, dataset, opt):
        super().__init__()
        self.opt = opt
        in_dim = dataset.data.x_comp.shape[1]
        hid_dim = opt['hidden_dim']
        out_dim = dataset.data.x_comp.shape[1]
        self.enc = get_enc(opt, in_dim, hid_dim, nonlin_type=opt['non_lin'])
        self.non_lin = get_nonlin(opt['non_lin'])
        self.dec = get_dec(opt, hid_dim, out_dim, nonlin_type=opt['non_lin'])
        self.fc1 = nn.Linear(hid_dim, hid_dim)
        self.fc2 = nn.Linear(hid_dim, hid_dim)

    def forward(self, data):
        o = self.opt
        x = self.enc(data.x_comp)
        for fc in (self.fc1, self.fc2):                                   # GNN.py:33-43
            x = x + o['time_step'] * fc(x) if o['residual'] else fc(x)
            x = self.non_lin(F.dropout(x, o['dropout'], training=self.training))
        return self.dec(x)


class GNN(nn.Module):
    def __init__(self, dataset, opt):
        super().__init__()
        self.dataset = dataset
        self.opt = opt
        self.dim = dataset.num_x_comp_features
        md = opt['mesh_dims']
        self.mesh_dims = md if isinstance(md[0], int) else eval(md[0])    # params.get_arg_list (params.py:190-196)
        self.in_dims = [self.dim]
        for key, width in (('gnn_inc_feat_f', 1), ('gnn_inc_feat_uu', 1),
                           ('gnn_inc_glob_feat_f', opt.get('global_feat_dim')),
                           ('gnn_inc_glob_feat_uu', opt.get('global_feat_dim'))):
            if opt.get(key):
                self.in_dims.append(width)
        opt['hidden_dims_list'] = self.in_dims                            # GNN.py:161
        in_dim, hid = sum(self.in_dims), opt['hidden_dim']
        self.enc = get_enc(opt, in_dim, hid, nonlin_type=opt['non_lin'])
        self.conv_layers = build_conv_list(opt)
        self.non_lin = get_nonlin(opt['non_lin'])
        self.dec = get_dec(opt, hid, self.dim, nonlin_type=opt['non_lin'])
        if opt.get('gnn_inc_glob_feat_f'):                                # GNN.py:170-175
            self.global_out_dim = opt['global_feat_dim']
            self.global_feature_extractor_cnn_f = GlobalFeatureExtractorCNN(1, hid, self.global_out_dim, dim=self.dim)
        if opt.get('gnn_inc_glob_feat_uu'):
            self.global_out_dim = opt['global_feat_dim']
            self.global_feature_extractor_cnn_uu = GlobalFeatureExtractorCNN(1, hid, self.global_out_dim, dim=self.dim)
        if opt.get('learn_step'):
            self.steps = nn.ParameterList([nn.Parameter(torch.tensor([opt['time_step']]))
                                           for _ in range(opt['num_layers'])])       # GNN.py:179-180
        q = torch.linspace(0, 1, opt.get('eval_quad_points', 101))
        self.quad_points = q if self.dim == 1 else list(torch.meshgrid(q, q, indexing='ij'))
        self.end_MLmodel = None
        self._graphs: Dict[Tuple, MeshGraph] = {}
        self._dt_const = None
        self._lp_const = None
        self._sc_const = None

    # ------------------------------------------------------------------ graph cache
    def _graph(self, data, num_nodes: int, device) -> MeshGraph:
        """CSR of the edge list the conv layers see (`src/GNN.py:206-223`), built once per batch topology.  The cache key
        holds a content fingerprint of `edge_index` and of the three masks (`graph.content_fingerprint`): batches with
        equal node / edge counts but another connectivity, edge order or boundary marking get their own graph, as in the
        reference, which redoes the surgery on every forward."""
        corners = getattr(data, 'corner_nodes', None)
        if corners is None:
            clist = None
        elif isinstance(corners, (list, tuple)):
            clist = list(corners)
        else:
            clist = [np.asarray(corners)]                               # a single un-collated sample (data is not modified)
        # (memoised per corner-list OBJECT and the identity of its entries' ends: 4 entries per mesh walked in Python cost ~10 us per
        # forward of a 32-mesh batch; a list edited in place in the middle is not noticed - collation builds a new list per batch)
        sig = None if clist is None else (len(clist), id(clist[0]) if clist else 0, id(clist[-1]) if clist else 0)
        memo = getattr(self, '_ckey_memo', None)
        if memo is not None and memo[0] is corners and memo[1] == sig:
            ckey = memo[2]
        else:
            ckey = () if clist is None else tuple(int(v) for a in clist for v in np.asarray(a).reshape(-1))
            self._ckey_memo = (corners, sig, ckey)
        fix = bool(self.opt['fix_boundary'])
        tensors = [data.edge_index]
        if fix:
            # the loop nodes appended by GNN.py:209-218 come from `batch` (mesh count / per-mesh node counts)
            tensors += [data.to_boundary_edge_mask, data.to_corner_nodes_mask, data.diff_boundary_edges_mask, data.batch]
        key = (num_nodes, self.dim, ckey, fix, bool(self.opt.get('self_loops')), str(device),
               _graph_mod.content_fingerprint(tensors), _graph_mod.WIDE_KERNELS, _graph_mod.WIDE_MIN_NODES, _graph_mod.WIDE_HALF_MAX_NODES)
        g = self._graphs.get(key)
        if g is None:
            ei = prepare_edge_index(data, self.dim, self.mesh_dims[0], fix, bool(self.opt.get('self_loops')), num_nodes,
                                    corner_nodes=clist if self.dim == 2 else None)
            g = MeshGraph(ei.to(device), num_nodes, device)
            if len(self._graphs) >= 32:
                self._graphs.pop(next(iter(self._graphs)))
            self._graphs[key] = g
        return g

    def _layer_params(self, device) -> torch.Tensor:
        """[L,2] = (dt_l, score_scale_l) on device; differentiable wrt `sm_temp_a`.  With `learn_step` the [L] score scales
        alone: the block op takes the step parameters themselves (`steps=`) and packs the pairs (functional._GrandEulerBlock)."""
        L = self.opt['num_layers']
        learn_a = self.opt.get('softmax_temp_type') == 'learnable_a'
        if not (self.opt.get('learn_step') or learn_a):
            if self._lp_const is None or self._lp_const.device != device:
                sc = Fn.score_scale(self.opt['hidden_dim'], self.conv_layers[0]._temperature())
                self._lp_const = torch.tensor([[float(self.opt['time_step']), sc]] * L, device=device, dtype=torch.float32)
            return self._lp_const
        if learn_a:
            scales = torch.stack([layer._scale(device) for layer in self.conv_layers])
        else:
            if self._sc_const is None or self._sc_const.device != device or self._sc_const.numel() != L:
                sc = [float(layer._scale(device)) for layer in self.conv_layers]      # constants: read once
                self._sc_const = torch.tensor(sc, device=device, dtype=torch.float32)
            scales = self._sc_const
        if self.opt.get('learn_step'):
            return scales
        if self._dt_const is None or self._dt_const.device != device or self._dt_const.numel() != L:
            self._dt_const = torch.full((L,), float(self.opt['time_step']), device=device)
        return torch.stack([self._dt_const, scales], dim=1)

    def _enc_is_zero_pad(self) -> bool:
        """True when the frozen encoder weight is the identity zero-pad eye(C, F) that `get_enc('identity')` installs
        (checked once per weight version: a loaded state_dict may carry anything)."""
        w = self.enc.weight
        key = (w.data_ptr(), w._version, tuple(w.shape))
        if getattr(self, '_zero_pad_key', None) != key:
            self._zero_pad_key = key
            self._zero_pad = bool(torch.equal(w.detach(), torch.eye(w.shape[0], w.shape[1], device=w.device, dtype=w.dtype)))
        return self._zero_pad

    def _fusable(self) -> bool:
        o = self.opt
        plain = o['conv_type'] == 'GRAND_plus' or (o['conv_type'] == 'GRAND' and o['non_lin'] == 'identity')   # GNN.py:284-286
        native = o['hidden_dim'] in Fn._native.SUPPORTED_HIDDEN and not (o.get('reg_skew') and self.dim == 2) \
            and o.get('softmax_temp_type') != 'learnable_v'
        return bool(o['residual']) and plain and native and not (self.training and o.get('dropout', 0.0) > 0)

    def _gat_plus_fusable(self) -> bool:
        o = self.opt
        return (o['conv_type'] == 'GAT_plus' and o['hidden_dim'] in Fn._native.SUPPORTED_HIDDEN and not o.get('learn_step')
                and o.get('gat_plus_type') in ('GAT_res_lap', 'GAT_lin') and o['non_lin'] in Fn.NONLIN_CODES
                and o.get('fused_gat_plus', True) and not (self.training and o.get('dropout', 0.0) > 0))

    # ------------------------------------------------------------------ one-launch forward of small-mesh batches
    def _small_plan(self, data, graph, x_comp, f, uu):
        """Everything `functional.small_forward` / `small_block` need besides the node fields, or None when the batch / the model
        state does not qualify: fusable conv, frozen bias-free Linear encoder on fp32 fields, hidden <= 32, meshes that are
        contiguous node ranges and fit a workgroup (graph.mesh_partition, functional.small_forward_fits).  With autograd on
        (training) also: fixed steps and temperature (the one-launch backward returns the conv gradients only)."""
        o = self.opt
        # training = autograd is on AND some conv parameter wants a gradient: an eval-mode model called outside no_grad with frozen
        # weights is an evaluation (nothing to keep for a backward, the evaluation policy's sizes apply)
        train = torch.is_grad_enabled() and any(p.requires_grad for p in self.conv_layers.parameters())
        if not x_comp.is_cuda:                                             # (the per-layer path raises the no-CPU-fallback error)
            return None
        if not (Fn.SMALL_MESH_FORWARD and self._fusable() and o['hidden_dim'] <= 32
                and isinstance(self.enc, nn.Linear) and self.enc.bias is None and not self.enc.weight.requires_grad and x_comp.dtype == torch.float32
                and all(t is None or (t.dtype == torch.float32 and t.dim() == 1 and t.is_contiguous() and not t.requires_grad) for t in (f, uu))
                and not x_comp.requires_grad and o['loss_type'] in ('mesh_loss', 'modular')):
            return None
        if train and (o.get('learn_step') or o.get('softmax_temp_type') == 'learnable_a'):
            return None
        part = graph.mesh_partition(getattr(data, 'batch', None))
        if not (Fn.small_backward_fits if train else Fn.small_forward_fits)(graph, part, o['hidden_dim']):
            return None
        dev = x_comp.device
        first = self.conv_layers[0]
        keep = (lambda t: t) if train else (lambda t: t.detach())
        if o['share_conv']:
            wq, bq, wk, bk = (keep(t).unsqueeze(0) for t in (first.lin_query.weight, first.lin_query.bias, first.lin_key.weight, first.lin_key.bias))
        else:
            wq = torch.stack([keep(l.lin_query.weight) for l in self.conv_layers])
            bq = torch.stack([keep(l.lin_query.bias) for l in self.conv_layers])
            wk = torch.stack([keep(l.lin_key.weight) for l in self.conv_layers])
            bk = torch.stack([keep(l.lin_key.bias) for l in self.conv_layers])
        lp = self._layer_params(dev)
        if o.get('learn_step'):                                            # _layer_params returned the [L] scales alone
            lp = torch.stack([torch.cat([s_.detach().reshape(1) for s_ in self.steps]), lp.detach()], dim=1)
        store = o['conv_type'] == 'GRAND' or isinstance(o.get('show_mesh_evol_plots'), bool)
        ident = isinstance(self.dec, nn.Identity)
        return {'graph': graph, 'part': part, 'wq': wq.contiguous(), 'bq': bq.contiguous(), 'wk': wk.contiguous(), 'bk': bk, 'lp': lp.detach().contiguous(),
                'enc_w': self.enc.weight.detach().contiguous(), 'store': store, 'ident': ident, 'out_cols': self.dim if ident else o['hidden_dim'],
                'train': train}

    def _small_run(self, plan, x_comp, f, uu):
        if plan['train']:
            x, alpha = Fn.small_block(plan['graph'], plan['part'], x_comp, f, uu, plan['enc_w'], plan['wq'], plan['bq'], plan['wk'], plan['bk'],
                                      plan['lp'], self.opt['num_layers'], plan['out_cols'])
        else:
            x, alpha = Fn.small_forward(plan['graph'], plan['part'], x_comp, f, uu, plan['enc_w'], plan['wq'], plan['bq'], plan['wk'], plan['lp'],
                                        self.opt['num_layers'], plan['out_cols'], want_alpha=plan['store'])
        if plan['store']:                                                  # GRAND_plus.py:253-256, :381
            for l, layer in enumerate(self.conv_layers):
                layer.stored_ei, layer._stored = plan['graph'].edge_index, (plan['graph'], alpha[l])
        if not plan['ident']:
            x = self.dec(x) if self.dec is not None else x                 # GNN.py:298
            x = x[:, :self.dim]                                            # GNN.py:299
        return x

    # ------------------------------------------------------------------ forward
    def forward(self, data):
        o = self.opt
        dev = torch.device(o['device'])
        x_comp = data.x_comp.to(dev, non_blocking=True)
        if self.dim == 1 and x_comp.dim() == 1:
            x_comp = x_comp.unsqueeze(-1)                                  # GNN.py:225-226
        f = uu = None
        if o['gnn_inc_feat_f']:
            f = data.f_tensor.to(dev, non_blocking=True)
            f = f / torch.max(f) if o.get('gnn_normalize') else f          # GNN.py:230-233
        if o['gnn_inc_feat_uu']:
            uu = data.uu_tensor.to(dev, non_blocking=True)
            uu = uu / torch.max(uu) if o.get('gnn_normalize') else uu      # GNN.py:235-238
        n = x_comp.shape[0]
        graph = self._graph(data, n, dev)
        glob = []                                                          # per-mesh CNN features, GNN.py:240-268
        if o.get('gnn_inc_glob_feat_f') or o.get('gnn_inc_glob_feat_uu'):
            batch = data.batch.to(dev)
            # meshes in the batch: what collation recorded (`_num_graphs`, PyG's `Batch.num_graphs`) or the corner list's length -
            # a host synchronisation on `batch.max()` only for a batch object that carries neither (it would break a hipGraph capture)
            n_meshes = getattr(data, '_num_graphs', None)
            if n_meshes is None and isinstance(getattr(data, 'corner_nodes', None), (list, tuple)) and self.dim == 2:
                n_meshes = len(data.corner_nodes)
            if n_meshes is None:
                n_meshes = int(data.batch.max().item()) + 1
            mapping = getattr(data, 'mapping_tensor', None) if o.get('data_type') == 'randg_mix' else \
                getattr(self.dataset, 'mapping_tensor', None)
            for flag, name, extractor in (('gnn_inc_glob_feat_f', 'f_tensor', 'global_feature_extractor_cnn_f'),
                                          ('gnn_inc_glob_feat_uu', 'uu_tensor', 'global_feature_extractor_cnn_uu')):
                if o.get(flag):
                    field = getattr(data, name).to(dev).float()
                    if o.get('gnn_normalize'):
                        field = field / torch.max(field)
                    dims = self.mesh_dims
                    if o.get('data_type') == 'randg_mix' and self.dim == 2:
                        side = int(np.sqrt(x_comp.shape[0]))               # GNN.py:247,261: one mesh per batch for this data type
                        dims = [side, side]
                    grid = field_to_grid(field, mapping, dims, n_meshes, self.dim)
                    glob.append(expand_to_nodes(getattr(self, extractor)(grid.unsqueeze(1)), batch))

        def features():                                                    # the concatenated matrix, only where a caller needs it
            return torch.cat([x_comp] + [t.unsqueeze(-1) for t in (f, uu) if t is not None] + glob, dim=1).float()

        fusable = self._fusable()
        x_all, sliced, x0_cols = None, False, 0
        feats, coeffs = None, None
        first = self.conv_layers[0]
        # ---- small meshes (evaluation, and training at the sizes functional.small_training_policy takes): encoder + all layers + head as
        # ONE launch, one workgroup per mesh (the reference's own
        # sizes: params.py:37,56,107,130-134; utils_eval.py:193-201, utils_eval_Burgers.py:282-300)
        if fusable and not glob:
            plan = self._small_plan(data, graph, x_comp, f, uu)
            if plan is not None:
                x = self._small_run(plan, x_comp, f, uu)
                if not self.training and not torch.cuda.is_current_stream_capturing():
                    torch.cuda.current_stream(dev).synchronize()           # the stamp is read as a latency (utils_eval.py:201)
                self.end_MLmodel = time.time()                             # GNN.py:301
                return x
        # weight-shared conv (GNN.py:131-140): its composite coefficients ride in the encoder's launch
        conv_w = (first.lin_query.weight, first.lin_query.bias, first.lin_key.weight) \
            if (fusable and o['share_conv'] and o['hidden_dim'] in Fn._native.SUPPORTED_HIDDEN) else None
        if glob:                                                           # differentiable wrt the CNN parameters
            feats = features()
            x = F.linear(feats, self.enc.weight, self.enc.bias) if isinstance(self.enc, nn.Linear) else self.enc(feats)
        elif isinstance(self.enc, nn.Linear) and not self.enc.weight.requires_grad and self.enc.bias is None:
            native_in = all(t is None or (t.dtype == torch.float32 and t.dim() == 1) for t in (f, uu)) and x_comp.dtype == torch.float32 \
                and o['hidden_dim'] in Fn._native.SUPPORTED_HIDDEN         # the encoder kernel's row layouts; other widths: dense GEMM
            if (fusable or self._gat_plus_fusable()) and not (self.training and o.get('dropout', 0.0) > 0):
                # encoder output lands in slot 0 of the block's activation buffer: no copy
                x_all = torch.empty(o['num_layers'] + 1, n, o['hidden_dim'], device=dev, dtype=torch.float32)
                out0 = x_all[0]
            else:
                out0 = None
            if (fusable and o.get('compact_slots', True) and native_in and x_all is not None and o['num_layers'] >= 2 and o['hidden_dim'] >= 8
                    and self.enc.weight.shape[1] <= 4 and self._enc_is_zero_pad()):
                # identity encoder = zero-pad (GNN.py:75-82): layer 0 reads the compact [N,4] features, the padded
                # [N,C] matrix is never written (nor read back by the layer-0 backward)
                x0_cols = 4
                x = Fn.encode_features(x_comp, f, uu, self.enc.weight[:4], out=x_all[0].view(-1)[:4 * n].view(n, 4), conv=conv_w)
                if conv_w is not None:
                    x, coeffs = x
            elif native_in:                                                # GNN.py:225-239 + :270 in one launch
                x = Fn.encode_features(x_comp, f, uu, self.enc.weight, out=out0, conv=conv_w if x_all is not None else None)
                if conv_w is not None and x_all is not None:
                    x, coeffs = x
            elif o['hidden_dim'] in Fn._native.SUPPORTED_HIDDEN:
                feats = features()
                x = Fn.encode_linear(feats, self.enc.weight, out=out0)
            else:
                feats = features()
                x = F.linear(feats, self.enc.weight)
        else:
            feats = features()
            x = self.enc(feats)
        if x_all is None:
            x = F.dropout(x, o.get('dropout', 0.0), training=self.training)               # GNN.py:271

        if fusable:
            if o['share_conv']:
                wq, bq = first.lin_query.weight.unsqueeze(0), first.lin_query.bias.unsqueeze(0)
                wk, bk = first.lin_key.weight.unsqueeze(0), first.lin_key.bias.unsqueeze(0)
            else:
                wq = torch.stack([l.lin_query.weight for l in self.conv_layers])
                bq = torch.stack([l.lin_query.bias for l in self.conv_layers])
                wk = torch.stack([l.lin_key.weight for l in self.conv_layers])
                bk = torch.stack([l.lin_key.bias for l in self.conv_layers])
            store = o['conv_type'] == 'GRAND' or isinstance(o.get('show_mesh_evol_plots'), bool)
            x, alpha = Fn.grand_euler_block(x, wq, bq, wk, bk, self._layer_params(dev), graph,
                                            o['num_layers'], want_alpha=store, x_all=x_all,
                                            out_cols=self.dim if (isinstance(self.dec, nn.Identity) and o.get('compact_slots', True)) else None,
                                            x0_cols=x0_cols, coeffs=coeffs,
                                            steps=list(self.steps) if o.get('learn_step') else None)
            sliced = isinstance(self.dec, nn.Identity) and o.get('compact_slots', True)
            if store:                                                      # GRAND_plus.py:253-256, :381
                for l, layer in enumerate(self.conv_layers):
                    layer.stored_ei, layer._stored = graph.edge_index, (graph, alpha[l])
        elif self._gat_plus_fusable():
            # conv_type='GAT_plus' (GNN.py:120-121): the L layers with their update (GNN.py:284-296) as one block op - per layer one
            # fused forward launch, two in backward (csrc/gadapt_gat.inc) - on the self-looped graph GATConv builds
            looped = graph.with_self_loops()
            distinct = [self.conv_layers[0]] if o['share_conv'] else list(self.conv_layers)
            att_src = torch.stack([l.att_src.reshape(-1) for l in distinct]) if len(distinct) > 1 else distinct[0].att_src.reshape(1, -1)
            att_dst = torch.stack([l.att_dst.reshape(-1) for l in distinct]) if len(distinct) > 1 else distinct[0].att_dst.reshape(1, -1)
            sliced = self.dec is None or isinstance(self.dec, nn.Identity)
            x, alpha = Fn.gat_plus_block(x, att_src, att_dst, looped, o['num_layers'], float(o['time_step']), bool(o['residual']),
                                         o['gat_plus_type'] == 'GAT_res_lap', o['non_lin'], out_cols=self.dim if sliced else None,
                                         x_all=x_all if (x_all is not None and x.data_ptr() == x_all.data_ptr()) else None)
            for l, layer in enumerate(self.conv_layers):                  # GRAND_plus.py:403-404,413-414: stored_ei / stored_alpha
                layer.stored_ei, layer._stored = looped.edge_index, (looped, alpha[l])
        else:
            mesh = getattr(self.dataset, 'mesh', None)
            if o.get('data_type') == 'randg_mix' and isinstance(getattr(data, 'mesh', None), (list, tuple)) and data.mesh:
                mesh = data.mesh[-1]                                       # what the loop of GNN.py:277-281 leaves in `mesh`
            for i, layer in enumerate(self.conv_layers):                  # GNN.py:273-296, layer by layer
                if o['residual'] and o['conv_type'] == 'GRAND_plus':
                    res = layer(x, graph.edge_index, feats if feats is not None else features(), mesh, graph=graph)
                else:
                    res = layer(x, graph.edge_index, graph=graph)
                    res = self.non_lin(F.dropout(res, o.get('dropout', 0.0), training=self.training))
                if o['residual']:
                    x = x + (self.steps[i] if o.get('learn_step') else o['time_step']) * res
                else:
                    x = res

        if sliced:
            x_phys = x                                                     # dec = Identity and x[:, :dim] done inside the block op
        else:
            x = self.dec(x) if self.dec is not None else x                 # GNN.py:298
            x_phys = x[:, :self.dim]                                       # GNN.py:299
        if not self.training and not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(dev).synchronize()                  # the stamp is read as a latency (utils_eval.py:201)
        self.end_MLmodel = time.time()                                    # GNN.py:301
        if o['loss_type'] in ('mesh_loss', 'modular'):
            return x_phys
        raise NotImplementedError("loss_type='pde_loss': the differentiable-FEM tail (src/GNN.py:307-342) is outside "
                                  "the message-passing path (SURVEY.md §2 row 2)")
