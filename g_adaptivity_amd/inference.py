"""Replay of the evaluation forward as one hipGraph.

The reference's evaluation drivers call the model once per sample (batch 1, `src/utils_eval.py:193-201`) and the
Burgers rollout re-invokes it every time step on the same mesh with a new `uu_tensor`
(`src/utils_eval_Burgers.py:282-300`).  At those sizes a forward is a handful of short launches, so the cost is the
host's launch path.  `GraphedForward` captures `model(data)` once (eval mode, no autograd) into a hipGraph over static
input buffers; every later call copies the new node fields into those buffers and replays the graph: one launch from the
host, no Python in between.  The topology (edge list, masks, node count) is fixed by the captured batch.
"""
from __future__ import annotations

import time
from typing import Optional

import torch

FIELDS = ('x_comp', 'f_tensor', 'uu_tensor')


class GraphedForward:
    def __init__(self, model, data, warmup: int = 2):
        if model.training:
            raise RuntimeError("GraphedForward captures the evaluation forward: call model.eval() first")
        self.model = model
        dev = torch.device(model.opt['device'])
        self.static = data.clone().to(dev)                 # the graph reads these tensors by address
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(warmup):                        # builds the CSR cache and warms the allocator outside the capture
                model(self.static)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph, stream=side):
            self.out = model(self.static)
        self.device = dev
        self._last = {}                                    # field -> (source tensor, its version) of the last copy

    @torch.no_grad()
    def __call__(self, data=None, *, sync: bool = True, **fields) -> torch.Tensor:
        """New node fields either as a batch object (`x_comp`, `f_tensor`, `uu_tensor` are read) or as keywords.
        Returns the static output tensor (overwritten by the next call; clone it to keep it)."""
        for name in FIELDS:
            src = fields.get(name, getattr(data, name, None) if data is not None else None)
            if src is None:
                continue
            # a field that is the very tensor copied last time, unmodified since (same object, same version counter), is
            # already in the static buffer: a rollout changes uu_tensor only (utils_eval_Burgers.py:288-289), and every
            # copy skipped is ~10 us of host launch path per call
            last = self._last.get(name)
            if last is not None and last[0] is src and last[1] == src._version:
                continue
            getattr(self.static, name).copy_(src.reshape(getattr(self.static, name).shape), non_blocking=True)
            self._last[name] = (src, src._version)
        self.graph.replay()
        if sync:
            torch.cuda.current_stream(self.device).synchronize()
        self.model.end_MLmodel = time.time()               # same stamp the eager forward leaves (GNN.py:301)
        return self.out
