"""ORACLE - test infrastructure only.  Independent dense restatement (fp64).

Pins `oracle/pyg_restatement.py` from a second direction (SURVEY.md §8(c) row
"independent checks"): the same layer written as dense linear algebra,
    S = Q K^T / sqrt(C)  masked by the adjacency (with edge multiplicity),
    A = row-softmax(S),   res = A x - x,
with no gather/scatter code shared with the sparse restatement.  O(N^2): small
graphs only.  Follows `src/GRAND_plus.py:225-267,279,333` mathematically.
"""
from __future__ import annotations

import math

import torch


def dense_attention(x, edge_index, w_query, b_query, w_key, b_key, temperature=None):
    """Returns the dense [N,N] attention matrix A with A[i,j] = sum of alpha over edges j->i."""
    n, c = x.shape
    x = x.double()
    q = x @ w_query.double().T + b_query.double()
    k = x @ w_key.double().T + b_key.double()
    s = (q @ k.T) / math.sqrt(c)
    if temperature is not None:
        s = s / float(temperature)
    mult = torch.zeros(n, n, dtype=torch.float64)
    mult.index_put_((edge_index[1], edge_index[0]), torch.ones(edge_index.shape[1], dtype=torch.float64),
                    accumulate=True)
    has = mult > 0
    s_masked = torch.where(has, s, torch.full_like(s, float('-inf')))
    row_max = s_masked.max(dim=1, keepdim=True).values
    row_max = torch.where(torch.isfinite(row_max), row_max, torch.zeros_like(row_max))
    e = torch.where(has, torch.exp(s - row_max), torch.zeros_like(s)) * mult
    return e / (e.sum(dim=1, keepdim=True) + 1e-16)


def dense_residual(x, edge_index, w_query, b_query, w_key, b_key, temperature=None):
    a = dense_attention(x, edge_index, w_query, b_query, w_key, b_key, temperature)
    return a @ x.double() - x.double()


def dense_euler_block(x, edge_index, w_query, b_query, w_key, b_key, num_layers, time_step, temperature=None):
    """L explicit-Euler steps with shared weights (`src/GNN.py:273-291`), fp64."""
    x = x.double()
    for _ in range(num_layers):
        x = x + time_step * dense_residual(x, edge_index, w_query, b_query, w_key, b_key, temperature)
    return x
