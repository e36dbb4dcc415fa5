"""MI355X-native message-passing hot path of g-adaptivity (GNN.py / GRAND_plus.py).

Public surface mirrors the reference modules:
    from g_adaptivity_amd import GNN, get_conv, GRAND_plusConv, GRAND_conv
    from g_adaptivity_amd import mse_loss, l1_loss          # the training loop's loss_fn, one launch each
The arithmetic lives in `libgadapt_hip.so` (csrc/, C-ABI in include/gadapt_hip.h).
"""
from .conv import GAT_conv, GAT_plus, GCN_conv, GRAND_conv, GRAND_plusConv, TRANS_conv
from .functional import l1_loss, mse_loss, unit_gradient
from .gnn import GNN, MLP, build_conv_list, get_conv, get_dec, get_enc, get_mlp, get_nonlin
from .graph import GraphCache, MeshGraph, prepare_edge_index
from .mesh_graph import (DeviceMeshLoader, MeshData, MeshDataset, MeshLoader, Mixed_DataLoader, MixedMeshDataset, collate, interval_mesh,
                         square_mesh, synthetic_batch)
from .params import hot_path_opt
from .training import GraphedTrainStep

__all__ = ['GNN', 'MLP', 'get_conv', 'build_conv_list', 'get_enc', 'get_dec', 'get_mlp', 'get_nonlin',
           'GRAND_plusConv', 'GRAND_conv', 'TRANS_conv', 'GAT_plus', 'GAT_conv', 'GCN_conv', 'MeshGraph', 'GraphCache', 'prepare_edge_index',
           'MeshData', 'MeshDataset', 'MeshLoader', 'DeviceMeshLoader', 'MixedMeshDataset', 'Mixed_DataLoader', 'collate', 'interval_mesh', 'square_mesh',
           'synthetic_batch', 'hot_path_opt', 'GraphedTrainStep', 'mse_loss', 'l1_loss', 'unit_gradient']
