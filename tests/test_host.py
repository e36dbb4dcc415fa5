"""Host-side logic and the C-ABI surface (no GPU): mesh graphs, collation, edge surgery, CSR build,
exported symbols, module surface, loud failure without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from g_adaptivity_amd import (GNN, GRAND_conv, GRAND_plusConv, MeshDataset, MeshGraph, collate, get_conv, hot_path_opt,
                              interval_mesh, prepare_edge_index, square_mesh)
from g_adaptivity_amd import _native
from oracle.pyg_restatement import masked_edge_index

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n,N,E0,E", [(11, 121, 640, 562), (15, 225, 1232, 1122), (32, 1024, 5890, 5644), (64, 4096, 24066, 23564)])
def test_square_mesh_counts(n, N, E0, E):
    """node/edge counts and the in-degree histogram {6: (n-2)^2, 2: 4(n-2), 1: 4}."""
    m = square_mesh(n)
    ds = collate([m])
    ds.corner_nodes = [m.corner_nodes]
    ei = prepare_edge_index(ds, 2, n, True, False, N)
    assert m.x_comp.shape[0] == N and m.edge_index.shape[1] == E0 and ei.shape[1] == E
    hist = torch.bincount(torch.bincount(ei[1], minlength=N))
    assert hist[6] == (n - 2) ** 2 and hist[2] == 4 * (n - 2) and hist[1] == 4
    assert sorted(m.corner_nodes.tolist()) == [0, n - 1, n * (n - 1), n * n - 1]


def test_interval_mesh():
    m = interval_mesh(32)
    assert m.edge_index.shape[1] == 62 and len(m.corner_nodes) == 0
    d = collate([m, m])
    ei = prepare_edge_index(d, 1, 32, True, False, 64)
    assert ei.shape[1] == 2 * 62
    deg = torch.bincount(ei[1], minlength=64)
    assert deg[[0, 31, 32, 63]].tolist() == [1, 1, 1, 1] and (deg[1:31] == 2).all()


def test_collate_rules_and_edge_surgery_match_oracle():
    ds = MeshDataset([9, 9], 3, seed=1)
    d = collate(ds.samples)
    assert d.x_comp.shape == (243, 2) and d.batch.tolist() == sum(([b] * 81 for b in range(3)), [])
    assert d.edge_index.max().item() == 242 and isinstance(d.corner_nodes, list) and len(d.corner_nodes) == 3
    ours = prepare_edge_index(d, 2, 9, True, False, 243)
    ref = masked_edge_index(d, 2, 9)
    assert torch.equal(ours, ref)
    with_loops = prepare_edge_index(d, 2, 9, True, True, 243)
    assert (with_loops[0] == with_loops[1]).sum().item() == 243


def test_csr_build_host():
    d = collate(MeshDataset([7, 7], 2, seed=0).samples)
    ei = prepare_edge_index(d, 2, 7, True, False, 98)
    g = MeshGraph(ei, 98, 'cpu')
    src, dst = ei
    rt, rs = g.rowptr_t.long(), g.rowptr_s.long()
    assert rt[-1] == ei.shape[1] == rs[-1]
    assert torch.equal(g.col_t.long(), src[g.eid_t.long()])
    for i in range(98):
        ids = g.eid_t[rt[i]:rt[i + 1]].long()
        assert (dst[ids] == i).all() and (ids[1:] > ids[:-1]).all()          # grouped by target, input order kept
        slots = g.perm_s[rs[i]:rs[i + 1]].long()
        assert (src[g.eid_t[slots].long()] == i).all()
        assert torch.equal(g.col_s[rs[i]:rs[i + 1]].long(), dst[g.eid_t[slots].long()])
    assert torch.equal(g.tpos_s[g.perm_s.long()].long(), torch.arange(ei.shape[1]))
    alpha_t = torch.arange(ei.shape[1], dtype=torch.float32)
    assert torch.equal(g.alpha_to_edge_order(alpha_t)[g.eid_t.long()], alpha_t)


def test_csr_build_rejects_out_of_range():
    ei = torch.tensor([[0, 5], [1, 0]])
    with pytest.raises(_native.NativeError):
        MeshGraph(ei, 3, 'cpu')


def test_every_declared_symbol_is_exported():
    header = open(os.path.join(ROOT, 'include', 'gadapt_hip.h')).read()
    declared = set(re.findall(r'\b(gadapt_[a-z0-9_]+)\s*\(', header))
    lib = C.CDLL(_native.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/gadapt_hip.h but not exported"
    assert declared == set(_native.PROTOTYPES), declared ^ set(_native.PROTOTYPES)
    assert _native.lib().gadapt_abi_version() == 9
    # argument checks need no GPU: the gradient-exchange entry point refuses a null communicator before it looks for RCCL
    assert _native.lib().gadapt_allreduce_flat(None, None, 0, 0, None) == -1
    assert b'allreduce_flat' in _native.lib().gadapt_last_error()
    assert [c for c in (3, 4, 8, 16, 32, 64, 128, 256) if _native.lib().gadapt_supported_hidden_dim(c)] == [4, 8, 16, 32, 64, 128]


def test_ctypes_prototypes_follow_the_header():
    """Every prototype `_native.py` binds has the parameter list `include/gadapt_hip.h` declares: the same count, pointers where the
    header has pointers (the graph structure by reference), 64-bit integers, ints and floats where it has those, and the return type -
    a ctypes call with one argument too few or an `int` where the header has `int64_t` passes garbage without a word."""
    header = open(os.path.join(ROOT, 'include', 'gadapt_hip.h')).read()
    header = re.sub(r'/\*.*?\*/', ' ', header, flags=re.S)
    header = re.sub(r'//[^\n]*', ' ', header)
    decls = re.findall(r'([A-Za-z_][A-Za-z0-9_ ]*?[\s\*]+)(gadapt_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;', header)
    assert len(decls) >= 40
    seen = set()

    def kind(param: str):
        param = param.strip()
        if '*' in param:
            return 'graph' if 'gadapt_graph' in param else 'ptr'
        t = param.rsplit(' ', 1)[0] if ' ' in param else param
        t = t.replace('const', '').replace('unsigned', '').strip()
        return {'int': 'int', 'int32_t': 'int', 'int64_t': 'long', 'float': 'float', 'size_t': 'long'}.get(t, t)

    want = {C.c_void_p: 'ptr', C.c_char_p: 'ptr', C.c_int: 'int', C.c_int64: 'long', C.c_float: 'float', _native._G: 'graph'}
    for ret, name, params in decls:
        seen.add(name)
        restype, argtypes = _native.PROTOTYPES[name]
        plist = [] if params.strip() in ('', 'void') else [x for x in params.split(',')]
        assert len(plist) == len(argtypes), (name, len(plist), len(argtypes))
        for k, (par, at) in enumerate(zip(plist, argtypes)):
            got = want.get(at, 'ptr' if hasattr(at, 'contents') or issubclass(at, C._Pointer) else None)   # POINTER(c_double), ...: a pointer
            assert kind(par) == got, (name, k, par.strip(), at)
        rk = 'ptr' if '*' in ret else kind(ret.strip() + ' x')
        assert rk == want[restype], (name, ret.strip(), restype)
    assert seen == set(_native.PROTOTYPES), seen ^ set(_native.PROTOTYPES)


def test_bad_arguments_return_codes_not_aborts():
    lib = _native.lib()
    assert lib.gadapt_backward_slab_rows(0, 64) < 0
    assert lib.gadapt_backward_slab_rows(1000, 7) < 0 and b'hidden_dim' in lib.gadapt_last_error()
    assert lib.gadapt_backward_slab_rows(131072, 64) == 512
    assert lib.gadapt_coeffs_forward(None, None, None, None, None, 8, None) == -1


def test_module_surface_and_state_dict_keys():
    opt = hot_path_opt(mesh_dims=[7, 7], hidden_dim=8, num_layers=3, learn_step=True)
    m = GNN(MeshDataset([7, 7], 1), opt)
    keys = set(m.state_dict())
    assert {'enc.weight', 'conv_layers.0.lin_key.weight', 'conv_layers.0.lin_key.bias', 'conv_layers.2.lin_query.weight',
            'conv_layers.2.lin_query.bias', 'conv_layers.1.lin_skip.weight', 'steps.0', 'steps.2'} <= keys
    assert m.conv_layers[0] is m.conv_layers[2]                       # share_conv (GNN.py:131-141)
    assert not m.enc.weight.requires_grad and torch.equal(m.enc.weight[:4], torch.eye(4)) and m.enc.weight[4:].abs().sum() == 0
    assert opt['hidden_dims_list'] == [2, 1, 1]
    m.epoch = 3; m.plot_evol_flag = True                              # arbitrary attribute writes (run_GNN.py:97)
    sep = GNN(MeshDataset([7, 7], 1), hot_path_opt(mesh_dims=[7, 7], share_conv=False, num_layers=2))
    assert sep.conv_layers[0] is not sep.conv_layers[1]
    assert isinstance(get_conv(opt, 'GRAND', 8, 8), GRAND_conv) and isinstance(get_conv(opt, 'GRAND_plus', 8, 8, 8), GRAND_plusConv)


def test_reference_error_conventions():
    opt = hot_path_opt()
    for bad in ('nope', 'Laplacian'):
        with pytest.raises(NotImplementedError):
            get_conv(opt, bad, 8, 8)
    with pytest.raises(NotImplementedError):
        GNN(MeshDataset([7, 7], 1), hot_path_opt(mesh_dims=[7, 7], non_lin='swish'))
    with pytest.raises(NotImplementedError):
        GNN(MeshDataset([7, 7], 1), hot_path_opt(mesh_dims=[7, 7], enc='bogus'))


def test_no_cpu_fallback():
    """The product path must fail loudly without the GPU instead of computing on the CPU."""
    opt = hot_path_opt(mesh_dims=[7, 7])
    ds = MeshDataset([7, 7], 1)
    m = GNN(ds, opt)
    with pytest.raises(_native.NativeError):
        m(collate(ds.samples))
    # the conv variants and the generic primitives are no exception
    from g_adaptivity_amd import sparse_ops as Sp
    d = collate(ds.samples)
    for conv in ('GAT_plus', 'GAT', 'GCN', 'TRANS'):
        mv = GNN(ds, hot_path_opt(mesh_dims=[7, 7], conv_type=conv))
        with pytest.raises(_native.NativeError):
            mv(d)
    g = MeshGraph(prepare_edge_index(d, 2, 7, True, False, 49), 49, 'cpu')
    with pytest.raises(_native.NativeError):
        Sp.spmm(g, None, torch.zeros(49, 8))
    with pytest.raises(_native.NativeError):
        Sp.edge_softmax(g, torch.zeros(g.num_edges))


def test_variant_state_dict_keys_match_the_oracle():
    """Parameter names of every conv variant behind get_conv (`src/GNN.py:108-124`) = PyG's / the reference's, so state_dicts
    interchange with the oracle's holders (strict load)."""
    from oracle.pyg_restatement import OracleGNN
    ds = MeshDataset([7, 7], 2, seed=0)
    want = {'GAT_plus': {'att_src', 'att_dst'}, 'GAT': {'att_src', 'att_dst', 'bias', 'lin_src.weight', 'lin_dst.weight'},
            'GCN': {'lin.weight', 'bias'}, 'TRANS': {'lin_key.weight', 'lin_key.bias', 'lin_query.weight', 'lin_query.bias',
                                                     'lin_value.weight', 'lin_value.bias', 'lin_skip.weight', 'lin_skip.bias'}}
    for conv, names in want.items():
        opt = hot_path_opt(mesh_dims=[7, 7], hidden_dim=8, num_layers=2, conv_type=conv, non_lin='relu')
        model, oracle = GNN(ds, dict(opt)), OracleGNN(ds, dict(opt))
        keys = set(model.state_dict())
        assert {k.split('conv_layers.0.')[1] for k in keys if k.startswith('conv_layers.0.')} == names, conv
        assert keys == set(oracle.state_dict()), conv
        model.load_state_dict(oracle.state_dict(), strict=True)
    # learnable_v: the parameter exists under the reference's name (src/GRAND_plus.py:159)
    lv = GNN(ds, hot_path_opt(mesh_dims=[7, 7], hidden_dim=8, softmax_temp_type='learnable_v'))
    assert lv.state_dict()['conv_layers.0.sm_temp_v.weight'].shape == (1, 8)


def test_global_cnn_feature_modules_have_the_reference_keys():
    """gnn_inc_glob_feat_*: `global_feature_extractor_cnn_{f,uu}.convs.{k}.{weight,bias}` (SURVEY.md §8(b) state_dict row)."""
    from g_adaptivity_amd import GNN, MeshDataset, hot_path_opt
    from oracle.pyg_restatement import OracleGNN
    opt = hot_path_opt(mesh_dims=[7, 7], hidden_dim=16, num_layers=2, gnn_inc_glob_feat_f=True, gnn_inc_glob_feat_uu=True)
    ds = MeshDataset([7, 7], 2, seed=0)
    model, oracle = GNN(ds, dict(opt)), OracleGNN(ds, dict(opt))
    keys = set(model.state_dict())
    for tag in ('f', 'uu'):
        for k in range(4):
            assert f'global_feature_extractor_cnn_{tag}.convs.{k}.weight' in keys
            assert f'global_feature_extractor_cnn_{tag}.convs.{k}.bias' in keys
    assert model.enc.weight.shape == (16, 2 + 1 + 1 + 8 + 8)
    assert keys == set(oracle.state_dict())
    model.load_state_dict(oracle.state_dict(), strict=True)


def test_field_to_grid_matches_the_gather_transpose_flip_recipe():
    import torch
    from g_adaptivity_amd.features import expand_to_nodes, field_to_grid
    from oracle.pyg_restatement import fd_tensor_to_grid
    torch.manual_seed(0)
    B, n = 3, 5
    u = torch.randn(B * n * n)
    mapping = torch.randperm(n * n)
    assert torch.equal(field_to_grid(u, mapping, [n, n], B, 2), fd_tensor_to_grid(u, mapping, [n, n], B, 2))
    assert torch.equal(field_to_grid(u, None, [n, n], B, 2)[1, 0], u.reshape(B, n, n)[1, :, n - 1])   # transpose + flip
    batch = torch.arange(B).repeat_interleave(torch.tensor([2, 1, 3]))
    per_mesh = torch.arange(6.).reshape(3, 2)
    assert torch.equal(expand_to_nodes(per_mesh, batch), per_mesh[batch])


def test_device_loader_batches_equal_cpu_collation():
    """DeviceMeshLoader (device='cpu' here): same batch contents as collate() of the same samples, topology shared."""
    import torch
    from g_adaptivity_amd import DeviceMeshLoader, MeshDataset, collate
    ds = MeshDataset([6, 6], 7, seed=3)
    loader = DeviceMeshLoader(ds, batch_size=3, shuffle=False, device='cpu')
    assert len(loader) == 3
    batches = list(loader)
    assert [b.num_graphs for b in batches] == [3, 3, 1]
    for k, b in enumerate(batches):
        ref = collate(ds.samples[3 * k:3 * k + 3])
        for key in ('x_comp', 'x_phys', 'f_tensor', 'uu_tensor', 'edge_index', 'batch', 'to_boundary_edge_mask',
                    'to_corner_nodes_mask', 'diff_boundary_edges_mask'):
            assert torch.equal(getattr(b, key), getattr(ref, key)), key
        assert all((a == c).all() for a, c in zip(b.corner_nodes, ref.corner_nodes))
    assert batches[0].edge_index.data_ptr() == batches[1].edge_index.data_ptr()      # one topology object per batch size
    g = torch.Generator().manual_seed(0)
    shuffled = list(DeviceMeshLoader(ds, batch_size=7, shuffle=True, device='cpu', generator=g))[0]
    assert sorted(shuffled.idx.tolist()) == list(range(7)) and shuffled.idx.tolist() != list(range(7))


def test_ell_copy_and_wide_eligibility():
    """gadapt_ell_build_host: ELL-8 copy of a CSR orientation (-1 padding, rows padded to 256) and the rule that
    sends a graph to the wide kernels: rows of at most 8 entries, every neighbour of node i inside rows
    [256*(i/256) - 64, 256*(i/256) + 320)."""
    from oracle.pyg_restatement import masked_edge_index
    for n_side, batch, want in ((64, 3, 6), (23, 7, 6), (128, 1, 0)):     # 128-wide rows: neighbours 128 apart -> tiled kernels
        ds = MeshDataset([n_side, n_side], batch, seed=1)
        ei = masked_edge_index(collate(ds.samples), 2, n_side)
        n = batch * n_side * n_side
        g = MeshGraph(ei, n, 'cpu')
        assert g.wide_deg == {'t': want, 's': want}
        if want:
            ell = g._ells['t'].view(-1, 8)
            assert ell.shape[0] == (n + 255) // 256 * 256 and (ell[n:] == -1).all()
            rt = g.rowptr_t.long()
            for i in (0, 1, n_side, n // 2, n - 1):
                d = int(rt[i + 1] - rt[i])
                assert torch.equal(ell[i, :d], g.col_t[rt[i]:rt[i + 1]]) and (ell[i, d:] == -1).all()
    gen = torch.Generator().manual_seed(0)
    rnd = torch.stack([torch.randint(0, 1000, (3000,), generator=gen), torch.randint(0, 1000, (3000,), generator=gen)])
    assert MeshGraph(rnd, 1000, 'cpu').wide_deg == {'t': 0, 's': 0}       # random graph: far neighbours
    star = torch.stack([torch.arange(1, 12), torch.zeros(11, dtype=torch.int64)])
    assert MeshGraph(star, 12, 'cpu').wide_deg['t'] == 0                   # a row of 11 in-edges
    from g_adaptivity_amd import graph as graph_mod
    ds = MeshDataset([16, 16], 2, seed=1)
    small = masked_edge_index(collate(ds.samples), 2, 16)
    graph_mod.WIDE_KERNELS = False
    try:
        assert MeshGraph(small, 512, 'cpu').wide_deg == {'t': 0, 's': 0}
    finally:
        graph_mod.WIDE_KERNELS = True
    graph_mod.WIDE_MIN_NODES = 24576                                        # the production default: small batches stay tiled
    try:
        assert MeshGraph(small, 512, 'cpu').wide_deg == {'t': 0, 's': 0}
    finally:
        graph_mod.WIDE_MIN_NODES = 0


def test_mixed_size_batches_and_batch_dict():
    """`data_type='randg_mix'` (`src/data_mixed_loader.py:6-35`, `src/GNN.py:199-218`): a batch mixes mesh sizes; excluded keys ride
    in `batch.batch_dict[i]`, non-tensor attributes collate into lists, and the edge surgery (corner offsets from the per-mesh
    node counts) matches the oracle's restatement."""
    from g_adaptivity_amd import Mixed_DataLoader, MixedMeshDataset
    ds = MixedMeshDataset([9, 12, 7], 6, seed=0)
    loader = Mixed_DataLoader(ds, batch_size=3, shuffle=False, follow_batch=[],
                              exclude_keys=['boundary_nodes_dict', 'mapping_dict', 'node_boundary_map', 'eval_errors', 'pde_params'])
    batches = list(loader)
    assert len(batches) == 2
    b = batches[0]
    n = 81 + 144 + 49
    assert b.x_comp.shape == (n, 2) and torch.bincount(b.batch).tolist() == [81, 144, 49]
    assert not hasattr(b, 'pde_params') and set(b.batch_dict) == {0, 1, 2} and 'centers' in b.batch_dict[1]['pde_params']
    assert isinstance(b.mesh, list) and len(b.mesh) == 3 and b.mesh[1].coordinates.cell_node_map().values.shape == (2 * 11 * 11, 3)
    ours = prepare_edge_index(b, 2, 9, True, False, n)
    assert torch.equal(ours, masked_edge_index(b, 2, 9))
    loops = ours[:, ours[0] == ours[1]]
    assert loops.shape[1] == 12 and sorted(loops[0].tolist())[4:8] == [81, 81 + 11, 81 + 132, 81 + 143]   # second mesh's corners, offset by 81
    g = MeshGraph(ours, n, 'cpu')
    assert g.num_edges == ours.shape[1]


def test_mlp_baseline_model():
    """`MLP(dataset, opt)` (`src/GNN.py:16-45`): enc -> x + dt fc1(x) -> non_lin -> x + dt fc2(x) -> non_lin -> dec, on `x_comp`;
    a literal restatement with the same parameters gives the same numbers, and the identity encoder / decoder pair makes the
    output [N, hidden] (the reference's `get_dec` is an Identity, `src/GNN.py:101-105`)."""
    from g_adaptivity_amd import MLP
    ds = MeshDataset([6, 6], 2, seed=0)
    opt = hot_path_opt(mesh_dims=[6, 6], hidden_dim=8, non_lin='tanh', time_step=0.1)
    torch.manual_seed(0)
    m = MLP(ds, opt)
    d = collate(ds.samples)
    out = m(d)
    x = torch.nn.functional.linear(d.x_comp, m.enc.weight)
    x = torch.tanh(x + 0.1 * m.fc1(x))
    x = torch.tanh(x + 0.1 * m.fc2(x))
    assert out.shape == (72, 8) and torch.allclose(out, x, atol=1e-7)
    assert set(m.state_dict()) == {'enc.weight', 'fc1.weight', 'fc1.bias', 'fc2.weight', 'fc2.bias'}


def test_fingerprint_separates_diagonal_flip():
    """flipping the diagonal of ONE grid quad in place - (a, a+n+1) -> (a+1, a+n), both directions - keeps every linear
    checksum (sum of v_i, sum of v_i * (i+1)) because both endpoint pairs sum to 2a+n+1; the cache key must still change."""
    from g_adaptivity_amd.graph import content_fingerprint
    n, a = 8, 10
    base = square_mesh(n).edge_index.clone()
    # make sure the quad's (a, a+n+1) diagonal is present in both directions, in place
    ei = base.clone()
    pos_fw = ((ei[0] == a + 1) & (ei[1] == a + n)).nonzero()
    pos_bw = ((ei[0] == a + n) & (ei[1] == a + 1)).nonzero()
    assert len(pos_fw) == 1 and len(pos_bw) == 1, "left-diagonal triangulation: (a+1, a+n) is an edge"
    flipped = ei.clone()
    flipped[:, pos_fw[0, 0]] = torch.tensor([a, a + n + 1])
    flipped[:, pos_bw[0, 0]] = torch.tensor([a + n + 1, a])
    # the linear checksums really do collide (this is the case the old key missed)
    w = torch.arange(1, ei.numel() + 1)
    assert int((ei.reshape(-1) * w).sum()) == int((flipped.reshape(-1) * w).sum()) and int(ei.sum()) == int(flipped.sum())
    assert not torch.equal(ei, flipped)
    assert content_fingerprint([ei]) != content_fingerprint([flipped])
    # and equal content in a fresh tensor object gives an equal key
    assert content_fingerprint([ei]) == content_fingerprint([ei.clone()])
    # an order change of the same edge set is a different key as well (summation order follows the caller's edge order)
    perm = torch.randperm(ei.shape[1], generator=torch.Generator().manual_seed(0))
    assert content_fingerprint([ei]) != content_fingerprint([ei[:, perm].contiguous()])


def test_wide_window_locality_test():
    """gadapt_wide_window_host: row-major meshes with up to 64 nodes per mesh row fit the 384-row window (steps of 256 nodes, halo 64) and
    the 256-row one of the four-wave form (steps of 128, halo 64; a 48-wide mesh, whose rows do not divide the step, included), 65..128
    only the 512-row one (halo 128); a mesh 129 wide fits none; the ELL copy is complete either way."""
    import ctypes as C
    lib = _native.lib()
    for n, want64, want128, want_half in ((32, True, True, True), (48, True, True, True), (64, True, True, True), (65, False, True, False),
                                          (128, False, True, False), (129, False, False, False)):
        m = square_mesh(n)
        d = collate([m]); d.corner_nodes = [m.corner_nodes]
        ei = prepare_edge_index(d, 2, n, True, False, n * n)
        g = MeshGraph(ei, n * n, 'cpu')
        got = {}
        for key, step, halo, rows in ((64, 256, 64, 8), (128, 256, 128, 7), ('half', 128, 64, 8)):
            md = C.c_int32(0)
            assert lib.gadapt_wide_window_host(g.rowptr_t.data_ptr(), g.col_t.data_ptr(), n * n, step, halo, rows, C.addressof(md)) == 0
            got[key] = md.value
        assert (got[64] > 0) == want64 and (got[128] > 0) == want128 and (got['half'] > 0) == want_half, (n, got)
        assert got[128] in (0, 6)
        # brute force: the rule itself, for the four-wave form
        rp, cl = g.rowptr_t.tolist(), g.col_t.tolist()
        ok = all(128 * (i // 128) - 64 <= j < 128 * (i // 128) + 192 for i in range(n * n) for j in cl[rp[i]:rp[i + 1]])
        assert ok == (got['half'] > 0), n
        ell = g._ells['t'].view(-1, 8)[:n * n]
        deg = (g.rowptr_t[1:] - g.rowptr_t[:-1]).long()
        assert torch.equal((ell >= 0).sum(1), deg)                              # complete even where the 384-row test fails
        assert torch.equal(ell[ell >= 0].long(), g.col_t[:int(deg.sum())].long())


def test_integration_md_ctypes_stub_matches_the_library():
    """INTEGRATION.md section 2 shows the ctypes stub a maintainer of the reference would add.  Its `Graph` structure must BE
    `struct gadapt_graph` (field for field, like `_native.GadaptGraph`), and its `build_graph` must run against the built library and
    fill the structure the way `MeshGraph` does (host-side calls only: no GPU here)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, 'INTEGRATION.md')).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = next(b for b in blocks if 'class Graph(C.Structure)' in b)
    ns, cwd = {}, os.getcwd()
    os.chdir(root)                                                   # the stub opens the library by its in-tree path
    try:
        exec(compile(stub, 'INTEGRATION.md', 'exec'), ns)
    finally:
        os.chdir(cwd)
    assert [(n, t) for n, t in ns['Graph']._fields_] == [(n, t) for n, t in _native.GadaptGraph._fields_]
    assert C.sizeof(ns['Graph']) == C.sizeof(_native.GadaptGraph)
    for mesh_n, batch in ((16, 2), (70, 1)):                         # 64-row window (and the four-wave one) / the 512-row window
        ds = MeshDataset([mesh_n, mesh_n], batch, seed=0)
        d = collate(ds.samples)
        ei = masked_edge_index(d, 2, mesh_n)
        n = d.x_comp.shape[0]
        from g_adaptivity_amd import graph as graph_mod
        keep = graph_mod.WIDE_MIN_NODES
        graph_mod.WIDE_MIN_NODES = 0
        try:
            mg = MeshGraph(ei, n, 'cpu')
        finally:
            graph_mod.WIDE_MIN_NODES = keep
        g, _bufs = ns['build_graph'](ei.to(torch.int64).cpu().contiguous(), n, 'cpu')
        ref = mg.c_struct
        for name in ('n_nodes', 'n_edges', 'wide_deg_t', 'wide_deg_s', 'wide_big_deg_t', 'wide_half_deg_t'):
            assert getattr(g, name) == getattr(ref, name), (mesh_n, name, getattr(g, name), getattr(ref, name))
        assert (g.wide_deg_t > 0) == (mesh_n <= 64) and (g.wide_half_deg_t > 0) == (mesh_n <= 64) and (g.wide_big_deg_t > 0) == (mesh_n > 64)
        for name, ref_t in (('rowptr_t', mg.rowptr_t), ('col_t', mg.col_t), ('rowptr_s', mg.rowptr_s), ('tpos_s', mg.tpos_s)):
            got = torch.frombuffer((C.c_int32 * ref_t.numel()).from_address(getattr(g, name)), dtype=torch.int32)
            assert torch.equal(got, ref_t.cpu()), name


def test_grand_plus_conv_option_surface():
    """Constructor options of `GRAND_plusConv` beyond what `get_conv` passes (`src/GRAND_plus.py:114-183`): parameter names and
    shapes follow the reference (`lin_skip` [H C, in] with concat, `lin_beta` [1, 3 H C], `sm_temp_a` [1,H,1]); `edge_dim` and the
    shape errors of the reference are refused at construction."""
    opt = hot_path_opt(softmax_temp_type='learnable_a')
    conv = GRAND_plusConv(opt, 32, 8, heads=4, concat=True, beta=True, root_weight=True, bias=True)
    sd = conv.state_dict()
    assert sd['lin_query.weight'].shape == (32, 32) and sd['lin_skip.weight'].shape == (32, 32) and sd['lin_skip.bias'].shape == (32,)
    assert sd['lin_beta.weight'].shape == (1, 96) and sd['sm_temp_a'].shape == (1, 4, 1)
    plain = GRAND_plusConv(hot_path_opt(), 8, 8, heads=1, concat=False, root_weight=False, bias=False)
    assert 'lin_beta.weight' not in plain.state_dict() and plain.state_dict()['lin_skip.weight'].shape == (8, 8) and not plain._general
    edged = GRAND_plusConv(opt, 8, 8, edge_dim=3)                   # lin_edge = Linear(edge_dim, H C, bias=False), GRAND_plus.py:165-166
    assert edged.state_dict()['lin_edge.weight'].shape == (8, 3) and edged._general and 'lin_edge.weight' not in plain.state_dict()
    with pytest.raises(NotImplementedError):
        GRAND_plusConv(opt, 8, 8, heads=2)                          # Identity(x).view(-1, 2, 8) needs 16 input channels
    with pytest.raises(ValueError):
        GRAND_plusConv(opt, 16, 8, heads=2, concat=False)           # `out - x`: [N,8] - [N,16]


def test_device_mesh_loader_fields_and_static_sink_on_cpu():
    """`DeviceMeshLoader(fields=..., into=...)` without a GPU (torch's index_select path): only the requested node fields are
    carried; from the second batch of a size on, the gathers land in the static batch the sink returned and that object is
    yielded; every batch equals the host collation of the same samples."""
    import torch
    from g_adaptivity_amd import DeviceMeshLoader, MeshDataset, collate
    ds = MeshDataset([7, 7], 10, seed=2)
    sinks = {}

    def sink(batch):
        s = batch.clone()
        sinks[int(batch.x_comp.shape[0])] = s
        return s

    gen = torch.Generator(); gen.manual_seed(3)
    loader = DeviceMeshLoader(ds, batch_size=4, shuffle=True, device='cpu', generator=gen, fields=('x_comp', 'x_phys', 'uu_tensor'), into=sink)
    seen_static = 0
    for epoch in range(2):
        for b in loader:
            assert not hasattr(b, 'f_tensor') and not hasattr(b, 'u_true_tensor')
            want = collate([ds.samples[i] for i in b.idx.tolist()])
            for k in ('x_comp', 'x_phys', 'uu_tensor'):
                assert torch.equal(getattr(b, k), getattr(want, k)), k
            assert torch.equal(b.edge_index, want.edge_index)
            seen_static += int(any(b is s for s in sinks.values()))
    assert set(sinks) == {4 * 49, 2 * 49}                           # one static batch per batch size (10 = 4 + 4 + 2)
    assert seen_static == 6 - 2                                     # every batch but the first of each size IS the static object