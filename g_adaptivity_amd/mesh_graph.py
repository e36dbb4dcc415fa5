"""Mesh graphs without Firedrake: topology, boundary masks, fields, batching.

The reference builds its model input offline with Firedrake + PyG
(`src/data.py:424-502` firedrake_mesh_to_PyG, `src/data.py:140-160` random
Gaussians, PyG `Batch.from_data_list` collation used by `src/run_GNN.py:76`).
Neither Firedrake nor PyG exists on the GPU box, so this module produces the
same *schema*  from first principles:

* `interval_mesh(n)`      - UnitIntervalMesh(n-1): n nodes on [0,1].
* `square_mesh(n)`        - UnitSquareMesh(n-1,n-1), Firedrake's default
                            "left" diagonal; node id = ix*n + iy.
* `MeshData`              - attribute bag with the PyG `Data` duck type the
                            model reads (`.to`, attribute assignment).
* `collate(list)`         - PyG collation rules: node tensors cat on dim 0,
                            `edge_index` cat on dim 1 with node offsets,
                            per-edge masks cat on dim 0, non-tensors -> lists,
                            plus the `batch` vector.
* `MeshLoader`            - minimal DataLoader (batch_size, shuffle).

Edge order inside one mesh is sorted (src, dst); the reference's order is the
iteration order of a Python set (`src/data.py:430-441`) and is therefore not
reproducible - results depend on it only through fp32 reassociation.
"""
from __future__ import annotations

import copy
from typing import Iterable, List, Optional, Sequence

import numpy as np
import torch


class MeshData:
    """Attribute bag standing in for `torch_geometric.data.Data` / `Batch`."""

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    def keys(self):
        return [k for k in self.__dict__ if not k.startswith('_')]

    def to(self, device, non_blocking: bool = False):
        for k in self.keys():
            v = getattr(self, k)
            if torch.is_tensor(v):
                setattr(self, k, v.to(device, non_blocking=non_blocking))
        return self

    def clone(self):
        out = MeshData()
        for k in self.keys():
            v = getattr(self, k)
            out.__dict__[k] = v.clone() if torch.is_tensor(v) else copy.deepcopy(v)
        if '_num_graphs' in self.__dict__:                   # the collation's graph count travels with the copy (a clone without it
            out.__dict__['_num_graphs'] = self.__dict__['_num_graphs']   # falls back to batch.max(): a host synchronisation)
        return out

    @property
    def num_graphs(self) -> int:
        b = getattr(self, 'batch', None)
        return 1 if b is None else int(getattr(self, '_num_graphs', int(b.max()) + 1))

    @property
    def num_nodes(self) -> int:
        return int(self.x_comp.shape[0])

    def __repr__(self):
        parts = []
        for k in self.keys():
            v = getattr(self, k)
            parts.append(f"{k}={list(v.shape)}" if torch.is_tensor(v) else f"{k}=...")
        return f"MeshData({', '.join(parts)})"


# --------------------------------------------------------------------------
# topology
# --------------------------------------------------------------------------

def _masks_from_topology(edge_index: np.ndarray, boundary_ids: dict):
    """Three per-edge masks + corner list, as `src/data.py:457-494` defines them."""
    node_ids = {}
    for bid, nodes in boundary_ids.items():
        for v in nodes:
            node_ids.setdefault(int(v), []).append(bid)
    boundary = np.zeros(int(edge_index.max()) + 1, dtype=bool)
    boundary[list(node_ids.keys())] = True
    corner_nodes = np.array(sorted(v for v, ids in node_ids.items() if len(ids) > 1), dtype=np.int64)
    is_corner = np.zeros_like(boundary)
    is_corner[corner_nodes] = True

    src, dst = edge_index
    to_boundary = boundary[dst] & ~boundary[src]            # data.py:465
    to_corner = is_corner[dst]                              # data.py:468
    # data.py:480-494: both ends on the boundary, different side lists, no corner involved
    side_key = np.full(boundary.shape[0], -1, dtype=np.int64)
    for v, ids in node_ids.items():
        side_key[v] = sum(1 << int(b) for b in ids)
    diff_boundary = (boundary[src] & boundary[dst] & (side_key[src] != side_key[dst])
                     & ~is_corner[src] & ~is_corner[dst])
    return boundary, corner_nodes, to_boundary, to_corner, diff_boundary


def interval_mesh(n: int) -> MeshData:
    """n nodes on the unit interval; edges (i,i+1) in both directions."""
    assert n >= 3
    i = np.arange(n - 1)
    und = np.stack([i, i + 1])
    ei = np.concatenate([und, und[::-1]], axis=1)
    ei = ei[:, np.lexsort((ei[1], ei[0]))]
    # UnitIntervalMesh markers: 1 -> x=0, 2 -> x=1; no node lies on two of them,
    # so `corner_nodes` is empty in 1-D.
    boundary, corners, m_b, m_c, m_d = _masks_from_topology(ei, {1: [0], 2: [n - 1]})
    x = np.linspace(0.0, 1.0, n, dtype=np.float64)
    return MeshData(
        x_comp=torch.tensor(x, dtype=torch.float32),           # [n] in 1-D (GNN unsqueezes)
        x_phys=torch.tensor(x, dtype=torch.float32),
        edge_index=torch.from_numpy(ei.astype(np.int64)),
        boundary_nodes=torch.from_numpy(boundary),
        corner_nodes=corners,
        to_boundary_edge_mask=torch.from_numpy(m_b),
        to_corner_nodes_mask=torch.from_numpy(m_c),
        diff_boundary_edges_mask=torch.from_numpy(m_d),
    )


def square_mesh(n: int) -> MeshData:
    """n x n nodes on the unit square, every cell cut by the "left" diagonal.

    Quad (ix,iy) has vertices v0=(ix,iy) v1=(ix,iy+1) v2=(ix+1,iy+1) v3=(ix+1,iy)
    and triangles (v0,v1,v3),(v1,v2,v3): the diagonal joins v1 and v3.
    """
    assert n >= 3
    ix, iy = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing='ij')
    ix, iy = ix.ravel(), iy.ravel()
    v0 = ix * n + iy
    v1 = ix * n + iy + 1
    v2 = (ix + 1) * n + iy + 1
    v3 = (ix + 1) * n + iy
    tris = np.concatenate([np.stack([v0, v1, v3], 1), np.stack([v1, v2, v3], 1)], 0)
    pairs = np.concatenate([tris[:, [0, 1]], tris[:, [1, 2]], tris[:, [0, 2]]], 0)
    pairs = np.concatenate([pairs, pairs[:, ::-1]], 0)
    pairs = np.unique(pairs, axis=0)                          # sorted (src,dst), deduplicated
    ei = pairs.T.copy()

    node = np.arange(n * n)
    nx_, ny_ = node // n, node % n
    sides = {1: node[nx_ == 0], 2: node[nx_ == n - 1], 3: node[ny_ == 0], 4: node[ny_ == n - 1]}
    boundary, corners, m_b, m_c, m_d = _masks_from_topology(ei, sides)
    lin = np.linspace(0.0, 1.0, n, dtype=np.float64)
    xy = np.stack([lin[nx_], lin[ny_]], 1)
    return MeshData(
        x_comp=torch.tensor(xy, dtype=torch.float32),
        x_phys=torch.tensor(xy, dtype=torch.float32),
        edge_index=torch.from_numpy(ei.astype(np.int64)),
        boundary_nodes=torch.from_numpy(boundary),
        corner_nodes=corners,
        to_boundary_edge_mask=torch.from_numpy(m_b),
        to_corner_nodes_mask=torch.from_numpy(m_c),
        diff_boundary_edges_mask=torch.from_numpy(m_d),
        cells=torch.from_numpy(tris.astype(np.int64)),
    )


# --------------------------------------------------------------------------
# fields
# --------------------------------------------------------------------------

def gaussian_fields(x: np.ndarray, centers: Sequence[np.ndarray], scales: Sequence[np.ndarray]):
    """u = sum_k exp(-sum_d (x_d-c_d)^2/s_d^2) and f = -laplace(u), evaluated at nodes.

    u follows `firedrake_difFEM/difFEM_2d.py:268-277`; f is the Poisson forcing of
    that u (`firedrake_difFEM/solve_poisson.py:145-147`). x is [N,d] float64.
    """
    u = np.zeros(x.shape[0])
    f = np.zeros(x.shape[0])
    for c, s in zip(centers, scales):
        c = np.asarray(c, dtype=np.float64)
        s = np.asarray(s, dtype=np.float64)
        g = np.exp(-(((x - c) / s) ** 2).sum(1))
        lap = g * ((4.0 * (x - c) ** 2 / s ** 4) - 2.0 / s ** 2).sum(1)
        u += g
        f -= lap
    return u, f


def attach_random_fields(mesh: MeshData, rng: np.random.Generator, num_gauss: int = 2,
                         target_noise: float = 0.01) -> MeshData:
    """One dataset sample: the shared mesh + its own Gaussians (`src/data.py:147-158`).

    `uu` stands in for the coarse FEM solve (= u_true at the nodes); the target
    `x_phys` stands in for the MA/MMPDE5 mesh (x_comp + small interior noise).
    """
    d = mesh.clone()
    x = d.x_comp.double().numpy()
    if x.ndim == 1:
        x = x[:, None]
    dim = x.shape[1]
    centers = [rng.uniform(0.0, 1.0, dim).astype('f') for _ in range(num_gauss)]
    scales = [rng.uniform(0.1, 0.5, dim).astype('f') for _ in range(num_gauss)]
    u, f = gaussian_fields(x, centers, scales)
    d.u_true_tensor = torch.tensor(u, dtype=torch.float32)
    d.uu_tensor = torch.tensor(u, dtype=torch.float32)
    d.f_tensor = torch.tensor(f, dtype=torch.float32)
    noise = rng.standard_normal(x.shape) * target_noise
    noise[d.boundary_nodes.numpy()] = 0.0
    tgt = torch.tensor(x + noise, dtype=torch.float32)
    d.x_phys = tgt[:, 0] if d.x_comp.dim() == 1 else tgt
    d.pde_params = {'centers': centers, 'scales': scales}
    return d


# --------------------------------------------------------------------------
# collation
# --------------------------------------------------------------------------

_EDGE_KEYS = ('to_boundary_edge_mask', 'to_corner_nodes_mask', 'diff_boundary_edges_mask')


def collate(samples: Sequence[MeshData], exclude_keys: Optional[Sequence[str]] = None) -> MeshData:
    """PyG `Batch.from_data_list` rules for the keys the model reads (`exclude_keys`: attributes left out of the batch)."""
    out = MeshData()
    offsets = np.cumsum([0] + [s.num_nodes for s in samples])
    keys = [k for k in samples[0].keys() if not (exclude_keys and k in exclude_keys)]
    for k in keys:
        vals = [getattr(s, k) for s in samples]
        if k == 'edge_index' or k == 'cells':
            cat_dim = 1 if k == 'edge_index' else 0
            out.__dict__[k] = torch.cat([v + int(o) for v, o in zip(vals, offsets[:-1])], dim=cat_dim)
        elif torch.is_tensor(vals[0]):
            out.__dict__[k] = torch.cat(vals, dim=0)
        else:
            out.__dict__[k] = list(vals)                       # e.g. corner_nodes, pde_params
    out.batch = torch.repeat_interleave(torch.arange(len(samples)),
                                        torch.tensor([s.num_nodes for s in samples]))
    out._num_graphs = len(samples)
    return out


class _CellNodeMap:
    def __init__(self, values):
        self.values = values


class _Coordinates:
    def __init__(self, cells):
        self._map = _CellNodeMap(cells)

    def cell_node_map(self):
        return self._map


class MeshTopology:
    """The part of a Firedrake mesh object the model reads: `mesh.coordinates.cell_node_map().values` -> [T,3] node ids
    per triangle (`src/GRAND_plus.py:281`, used by `reg_skew` only)."""

    def __init__(self, cells: np.ndarray):
        self.coordinates = _Coordinates(np.asarray(cells))


class MeshDataset:
    """In-memory list of samples over one shared mesh (`MeshInMemoryDataset` duck type).

    Exposes what `GNN.__init__` reads (`src/GNN.py:149`): `num_x_comp_features`,
    plus `x_comp_shared` and `mesh_dims`.
    """

    def __init__(self, mesh_dims: Sequence[int], num_data: int, seed: int = 0, num_gauss: int = 2):
        self.mesh_dims = list(mesh_dims)
        self.dim = len(self.mesh_dims)
        if self.dim == 1:
            base = interval_mesh(self.mesh_dims[0])
        else:
            assert self.mesh_dims[0] == self.mesh_dims[1], "square meshes only"
            base = square_mesh(self.mesh_dims[0])
        rng = np.random.default_rng(seed)
        self.base = base
        self.data = base                                     # `dataset.data.x_comp` (read by the MLP baseline, src/GNN.py:20-22)
        self.x_comp_shared = base.x_comp
        self.num_x_comp_features = self.dim
        self.samples: List[MeshData] = [attach_random_fields(base, rng, num_gauss) for _ in range(num_data)]
        # no Firedrake mesh object here: a stand-in with the one attribute chain the model follows (2-D only)
        self.mesh = MeshTopology(base.cells.numpy()) if self.dim == 2 else None

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, i):
        if isinstance(i, (list, np.ndarray, torch.Tensor)):
            sub = copy.copy(self)
            sub.samples = [self.samples[int(j)] for j in i]
            return sub
        return self.samples[i]


class MixedMeshDataset(MeshDataset):
    """`data_type='randg_mix'`: samples over SEVERAL meshes (`src/data_mixed.py`), so a batch mixes node counts.  Every
    sample carries its own `mesh` stand-in and `mapping_tensor` (the model reads `data.mesh[i]` / `data.mapping_tensor` for
    this data type, `src/GNN.py:247-248,279`) and its `pde_params`."""

    def __init__(self, mesh_sizes: Sequence[int], num_data: int, seed: int = 0, num_gauss: int = 2):
        self.mesh_sizes = list(mesh_sizes)
        self.mesh_dims = [self.mesh_sizes[0], self.mesh_sizes[0]]
        self.dim = 2
        self.num_x_comp_features = 2
        rng = np.random.default_rng(seed)
        bases = {n: square_mesh(n) for n in self.mesh_sizes}
        self.samples = []
        for k in range(num_data):
            n = self.mesh_sizes[k % len(self.mesh_sizes)]
            d = attach_random_fields(bases[n], rng, num_gauss)
            d.mesh = MeshTopology(bases[n].cells.numpy())
            d.mapping_tensor = torch.arange(n * n)
            self.samples.append(d)
        self.base = bases[self.mesh_sizes[0]]
        self.x_comp_shared = self.base.x_comp
        self.mesh = MeshTopology(self.base.cells.numpy())


class Mixed_DataLoader:
    """`Mixed_DataLoader(dataset, batch_size, shuffle, follow_batch, exclude_keys)` (`src/data_mixed_loader.py:29-35`): the
    keys in `exclude_keys` stay out of the collated batch and ride along per sample in `batch.batch_dict[i]`
    (`M2NCustomCollater`, `:6-25`); `GNN.forward` reads `data.batch_dict[i]['pde_params']` for `randg_mix` (`src/GNN.py:199-202`)."""

    def __init__(self, dataset, batch_size: int = 1, shuffle: bool = False, follow_batch=None, exclude_keys=None,
                 generator: Optional[torch.Generator] = None, **kwargs):
        self.dataset, self.batch_size, self.shuffle, self.generator = dataset, batch_size, shuffle, generator
        self.follow_batch, self.exclude_keys = follow_batch, list(exclude_keys or [])

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def collater(self, batch: Sequence[MeshData]) -> MeshData:
        side = {i: {k: getattr(d, k) for k in d.keys() if k in self.exclude_keys} for i, d in enumerate(batch)}
        out = collate(batch, exclude_keys=self.exclude_keys)
        out.batch_dict = side
        return out

    def __iter__(self) -> Iterable[MeshData]:
        n = len(self.dataset)
        if self.shuffle:
            gen = self.generator if (self.generator is not None and self.generator.device.type == 'cpu') else None
            order = torch.randperm(n, generator=gen).tolist()
        else:
            order = list(range(n))
        for s in range(0, n, self.batch_size):
            yield self.collater([self.dataset[i] for i in order[s:s + self.batch_size]])


class MeshLoader:
    """`DataLoader(dataset, batch_size, shuffle)` for MeshDataset (`src/run_GNN.py:76`)."""

    def __init__(self, dataset: MeshDataset, batch_size: int = 1, shuffle: bool = False,
                 generator: Optional[torch.Generator] = None):
        self.dataset, self.batch_size, self.shuffle, self.generator = dataset, batch_size, shuffle, generator

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterable[MeshData]:
        n = len(self.dataset)
        if self.shuffle:
            gen = self.generator if (self.generator is not None and self.generator.device.type == 'cpu') else None
            order = torch.randperm(n, generator=gen).tolist()
        else:
            order = list(range(n))
        for s in range(0, n, self.batch_size):
            yield collate([self.dataset[i] for i in order[s:s + self.batch_size]])


class DeviceMeshLoader:
    """Device-resident collation for a fixed-mesh dataset.

    Every sample of a `MeshDataset` shares one mesh (`src/data.py:290`), so a batch of B samples always has the same
    topology: `edge_index`, the edge masks, `corner_nodes` and `batch` are those of ANY B samples.  They are collated
    once per batch size; the per-sample node fields live stacked on the device ([S, n, ...]) and a batch is one
    `index_select` per field - no host work, no H2D copy per step, and the model's CSR cache is hit every time.
    Yields the same duck-typed batch objects as `MeshLoader` / PyG's `DataLoader` (`src/run_GNN.py:76`).
    """

    NODE_FIELDS = ('x_comp', 'x_phys', 'f_tensor', 'uu_tensor', 'u_true_tensor')

    def __init__(self, dataset: MeshDataset, batch_size: int = 1, shuffle: bool = False, device='cuda',
                 generator: Optional[torch.Generator] = None, fields: Optional[Sequence[str]] = None, into=None):
        """`fields`: the node fields a batch carries (default: all of NODE_FIELDS the samples have).  `into`: a callable mapping a
        batch to the STATIC batch object of its topology (`training.GraphedTrainStep.static_batch`): from the second batch of a
        size on, the gathers write straight into that object's tensors and the object itself is yielded - no copy between the
        loader and a captured step.  (A yielded static batch is overwritten by the next one: do not hold on to it.)"""
        self.dataset, self.batch_size, self.shuffle = dataset, batch_size, shuffle
        self.device, self.generator = torch.device(device), generator
        self.fields = {k: torch.stack([getattr(s, k) for s in dataset.samples]).to(self.device)
                       for k in (fields or self.NODE_FIELDS) if hasattr(dataset.samples[0], k)}
        self._templates = {}
        self.into, self._static = into, {}
        self._gather_plans = {}

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def _template(self, b: int) -> MeshData:
        t = self._templates.get(b)
        if t is None:
            t = collate(self.dataset.samples[:b])
            for k in self.NODE_FIELDS + ('pde_params',):
                t.__dict__.pop(k, None)
            t = self._templates[b] = t.to(self.device)
        return t

    def _gather(self, idx: torch.Tensor, dsts, cache_key=None) -> None:
        """dst_k[b] = field_k[idx[b]] for every field: ONE launch (`gadapt_gather_fields`) for the fp32 fields on the GPU, torch's
        index_select for anything else.  `cache_key`: the destinations are the tensors of a static batch - their argument arrays are
        built once (a captured small-mesh step is four short launches: the loop is bound by what the host does per batch)."""
        plan = self._gather_plans.get(cache_key) if cache_key is not None else None
        if plan is not None and plan[6] != [d_.data_ptr() for d_ in dsts]:   # the static batch was rebuilt: other tensors behind the key
            plan = None
        if plan is None:
            native = self.device.type == 'cuda' and all(s_.dtype == torch.float32 and s_.is_contiguous() and d_.is_contiguous()
                                                         for s_, d_ in zip(self.fields.values(), dsts)) and len(dsts) <= 8
            if not native:
                for stacked, dst in zip(self.fields.values(), dsts):
                    torch.index_select(stacked, 0, idx, out=dst.view(idx.numel(), *stacked.shape[1:]))
                return
            import ctypes as C
            from . import _native
            k = len(dsts)
            plan = (k, (C.c_void_p * k)(*[s_.data_ptr() for s_ in self.fields.values()]), (C.c_void_p * k)(*[d_.data_ptr() for d_ in dsts]),
                    (C.c_int64 * k)(*[s_[0].numel() for s_ in self.fields.values()]), _native.lib().gadapt_gather_fields, _native.current_stream,
                    [d_.data_ptr() for d_ in dsts])
            if cache_key is not None:
                self._gather_plans[cache_key] = plan
        k, src, dst, rows, fn, stream_of, _ = plan
        if idx.dtype != torch.int64 or not idx.is_contiguous():
            idx = idx.to(torch.int64).contiguous()
        rc = fn(k, src, dst, rows, idx.data_ptr(), int(idx.numel()), stream_of(self.device))
        if rc != 0:
            from . import _native
            _native.check(rc, 'gadapt_gather_fields')

    def __iter__(self) -> Iterable[MeshData]:
        n = len(self.dataset)
        if self.shuffle:
            gen = self.generator if (self.generator is not None and self.generator.device == self.device) else None
            order = torch.randperm(n, device=self.device, generator=gen)
        else:
            order = torch.arange(n, device=self.device)
        for s in range(0, n, self.batch_size):
            idx = order[s:s + self.batch_size]
            b = int(idx.numel())
            static = self._static.get(b)
            owner = getattr(self.into, '__self__', None)              # GraphedTrainStep.static_batch: the step may have evicted the capture
            if static is not None and owner is not None and hasattr(owner, 'owns') and not owner.owns(static):
                static = self._static.pop(b, None) and None           # gather into fresh tensors below and ask the step again
            if static is not None:                                    # gathers land in the captured step's input buffers
                self._gather(idx, [getattr(static, k) for k in self.fields], cache_key=b)
                static.idx = idx
                yield static
                continue
            out = copy.copy(self._template(b))                        # shares the topology tensors
            out.__dict__ = dict(out.__dict__)
            dsts = [torch.empty(b * stacked.shape[1], *stacked.shape[2:], device=self.device, dtype=stacked.dtype)
                    for stacked in self.fields.values()]
            self._gather(idx, dsts)
            for k, dst in zip(self.fields, dsts):
                out.__dict__[k] = dst
            out.idx = idx
            if self.into is not None:
                self._static[b] = self.into(out)                      # captured on first use, initialised with this batch's fields
            yield out


def synthetic_batch(mesh_dims: Sequence[int], batch_size: int, seed: int = 0, num_gauss: int = 2) -> MeshData:
    """Convenience: one collated batch of `batch_size` samples."""
    ds = MeshDataset(mesh_dims, batch_size, seed=seed, num_gauss=num_gauss)
    return collate(ds.samples)
