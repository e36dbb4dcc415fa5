#!/usr/bin/env python
"""bench.py - meshes/sec, forward+backward(+optimizer step), 2-D Poisson 64x64 mesh graphs, batch 32 per GPU.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched with
torch.distributed.run, one rank per GPU (RCCL).  Rank 0 prints ONE JSON line.

A step = the reference's training iteration on one batch (`src/run_GNN.py:99-131`, mesh_loss):
    optimizer.zero_grad(); out = model(data); loss = F.mse_loss(out, data.x_phys); loss.backward(); optimizer.step()
with the batch resident in HBM.  Per-GPU work is fixed (32 meshes/rank): weak scaling; the only
collective is the all-reduce of the flat gradient bucket.

Extra objects on the JSON line:
  roofline     - dominant hot kernel: algorithmic bytes per launch / its measured
                 duration (HIP events on the launch stream, second instrumented pass of K steps).
  cpu_baseline - the CPU oracle (PyG-equivalent op sequence in plain torch) on the host cores, rank 0, N=1.
  windows      - `value` / `ms_per_step` are the MEDIAN of --windows (5) timed windows of exactly K steps; min / max beside it.
  value_dense_slots / ms_per_step_dense_slots - the same run's timing of the literal dense-slot flow.
  train_loop   - N=1: the reference's training loop (src/run_GNN.py:95-131) on SHUFFLED, CHANGING batches through
                 training.GraphedTrainStep + DeviceMeshLoader (the headline steps on one static batch).
  config.launch / config.launch_ab - how the step is issued: as one hipGraph replay, or (the fused step, --launch auto, where a short A/B
                 in set-up finds it faster on this box; N > 1: always) as its three C-ABI calls - the same launches on the same buffers.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (mesh n, meshes per GPU, layers, hidden, conv_type, feature flags)
    'poisson2d_64x64_b32_L4_C64': dict(n=64, batch=32, layers=4, hidden=64, conv='GRAND_plus', f=True, uu=True),
    'poisson2d_32x32_b32_L4_C64': dict(n=32, batch=32, layers=4, hidden=64, conv='GRAND_plus', f=True, uu=True),
    'burgers2d_64x64_b32_L6_C128': dict(n=64, batch=32, layers=6, hidden=128, conv='GRAND', f=False, uu=True),
    'euler20_128x128_b16_C64': dict(n=128, batch=16, layers=20, hidden=64, conv='GRAND_plus', f=True, uu=True),
    # not a BASELINE.json shape: the other hidden sizes of the fused kernels on the metric mesh batch (tuning runs)
    'poisson2d_64x64_b32_L4_C32': dict(n=64, batch=32, layers=4, hidden=32, conv='GRAND_plus', f=True, uu=True),
    'poisson2d_64x64_b32_L4_C8': dict(n=64, batch=32, layers=4, hidden=8, conv='GRAND_plus', f=True, uu=True),
    # the two halves of what separates config 5 from the metric workload (docs/measurements.md G): twice the nodes at the metric's
    # 64-node mesh rows, and the metric's node count at config 5's 128-node mesh rows
    'poisson2d_64x64_b64_L4_C64': dict(n=64, batch=64, layers=4, hidden=64, conv='GRAND_plus', f=True, uu=True),
    'poisson2d_128x128_b8_L4_C64': dict(n=128, batch=8, layers=4, hidden=64, conv='GRAND_plus', f=True, uu=True),
    'poisson2d_128x128_b16_L4_C64': dict(n=128, batch=16, layers=4, hidden=64, conv='GRAND_plus', f=True, uu=True),
    # the metric mesh batch through the other trainable paths of the operator surface:
    # learn_step=True (GNN.py:179-180,288-289: the SUMS instantiations of the target pass + the d dt reduction) and
    # conv_type='GAT_plus' (GRAND_plus.py:386-416: the generic CSR primitives of gadapt_sparse.inc)
    'poisson2d_64x64_b32_L4_C64_learn_step': dict(n=64, batch=32, layers=4, hidden=64, conv='GRAND_plus', f=True, uu=True, learn_step=True),
    'poisson2d_64x64_b32_L4_C64_GAT_plus': dict(n=64, batch=32, layers=4, hidden=64, conv='GAT_plus', f=True, uu=True),
}
# BASELINE.json configs 2, 4 and 5: timed on the default line beside the headline (config 3's per-GPU shard)
OTHER_BASELINE_WORKLOADS = ('poisson2d_32x32_b32_L4_C64', 'burgers2d_64x64_b32_L6_C128', 'euler20_128x128_b16_C64')
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X fp32 matrix (= vector) peak, SURVEY.md §8(d)
MIN_WINDOW_S = 0.05             # a timed window lasts at least this long (the K-step block is repeated)


def algorithmic_bytes_gat(kernel, n_nodes, n_edges, c, variant=0):
    """Fused GAT_plus layer launches (csrc/gadapt_gat.inc), counted the way SURVEY.md 8(d) counts the GRAND layer: every node
    matrix a launch must read or write once, int32 CSR, per-edge and per-node scalars (alpha, d z, a, b) as 4 bytes each."""
    csr = 4 * (n_edges + n_nodes + 1)
    dense = 4 * n_nodes * c
    if kernel == 'forward':            # read x, write x'; alpha out, a / b in and out
        return 2 * dense + csr + 4 * n_edges + 16 * n_nodes
    if kernel == 'backward_target':    # read x and g, write g_r; alpha in, d z out, a / b in, d b out
        return 3 * dense + csr + 8 * n_edges + 12 * n_nodes
    if kernel == 'backward_source':    # read x (partials), g and g_r, write d x; alpha and d z in (through perm_s), d b in
        if variant & 16:               # layer 0 without d x0: parameter partials only
            return dense + csr + 8 * n_edges + 4 * n_nodes
        return 4 * dense + csr + 12 * n_edges + 4 * n_nodes
    raise KeyError(kernel)


def algorithmic_bytes(kernel, n_nodes, n_edges, c, variant=0, dim=2):
    """compulsory traffic of one layer launch over the whole batch (fp32, int32 CSR).

    variant (gadapt_profile_variants): bit 0 = compact upstream gradient [N,dim], bit 1 = compact layer input [N,4],
    bit 2 = head-only output [N,4], bit 3 = 4-column backward output (the layer above a compact layer 0) - the matrices such a
    launch really reads / writes are what is counted."""
    csr = 4 * (n_edges + n_nodes + 1)
    dense = 4 * n_nodes * c
    x_in = 16 * n_nodes if variant & 2 else dense
    if kernel == 'forward':            # read x, write x'
        return x_in + (16 * n_nodes if variant & 4 else dense) + csr
    if kernel == 'backward_target':    # read g, read saved x  (+ CSR by target)
        # compact layer input: d alpha = dt <g_i, x_k> contracts over the 4 live columns - 16 bytes of each g row are read
        return (4 * n_nodes * dim if variant & 1 else (16 * n_nodes if variant & 2 else dense)) + x_in + csr
    if kernel == 'backward_source':    # write dx               (+ CSR by source)
        return (16 * n_nodes if variant & 8 else dense) + csr       # out4: only columns 0..3 of dx are produced (layer above a compact layer 0)
    raise KeyError(kernel)


VARIANT_NAMES = {16: 'partials_only', 0: 'dense', 1: 'compact_g', 2: 'compact_x', 4: 'head_only_out', 6: 'compact_x+head_only_out', 8: 'out4', 9: 'compact_g+out4'}


def load_pmc(workload):
    """Per-kernel counter summary of this workload from profiles/pmc.json (tools/profile_all.sh + tools/make_pmc_json.py:
    separate rocprofv3 --pmc passes); {} when the file or the workload is missing."""
    path = os.path.join(ROOT, 'profiles', 'pmc.json')
    try:
        return json.load(open(path)).get(workload, {})
    except Exception:
        return {}


def load_profile_avg_us(workload, kernel, variant):
    """Average duration (us) of this kernel variant in the committed rocprofv3 --kernel-trace --stats summary of this workload
    (profiles/<round>_<workload>_rocprofv3_kernel_stats.csv, newest round first); (None, None) when there is none.  The kernel
    name carries the variant as template arguments (tools/make_pmc_json.py has the same mapping)."""
    import csv, glob
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    try:
        from make_pmc_json import name as kname, variant as kvariant
    except Exception:
        return None, None
    # rNN_bench_<workload>_... = bench.py itself under rocprofv3 (tools/bench_all.sh); rNN_<workload>_... = tools/profile_step.py
    paths = sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r*_bench_{workload}_rocprofv3_kernel_stats.csv')), reverse=True) + \
            sorted((f for f in glob.glob(os.path.join(ROOT, 'profiles', f'r*_{workload}_rocprofv3_kernel_stats.csv')) if '_bench_' not in os.path.basename(f)),
                   reverse=True)
    for path in paths:
        tot, calls = 0.0, 0
        for row in csv.DictReader(open(path)):
            if kname(row['Name']) == kernel and kvariant(row['Name']) == variant:     # None (the name cannot tell the variant): skipped
                tot += float(row['TotalDurationNs']); calls += int(row['Calls'])
        if calls:
            return tot / calls / 1e3, os.path.relpath(path, ROOT)
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--workload', default='poisson2d_64x64_b32_L4_C64', choices=list(WORKLOADS))
    ap.add_argument('--torch-loss', action='store_true', help='torch F.mse_loss instead of the one-launch native loss')
    ap.add_argument('--plain-backward', action='store_true', help='loss.backward() without the preallocated root gradient')
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of hipGraph replay')
    ap.add_argument('--launch', choices=['auto', 'graph', 'eager'], default='auto',
                    help="how the fused step is issued: replayed as one hipGraph, as its 3 C-ABI calls (13 launches), or (auto) whichever a "
                         "short A/B in set-up finds faster on one GPU / issued under data parallelism (nothing to capture around the all-reduce)")
    ap.add_argument('--no-fused-step', action='store_true',
                    help='the captured autograd iteration (16 launches) instead of the fused 13-launch iteration (training.FusedIteration)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dense-slots', action='store_true',
                    help='materialise the zero-padded encoder output, the full last-layer output and the padded top gradient '
                         '(DESIGN.md §4) instead of their compact forms')
    ap.add_argument('--cpu-seconds', type=float, default=20.0)
    ap.add_argument('--windows', type=int, default=5,
                    help='timed windows of --steps steps each; value / ms_per_step are the MEDIAN window (min and max reported beside it)')
    ap.add_argument('--no-train-loop', action='store_true',
                    help='skip the third measurement: the reference training loop on SHUFFLED, changing batches (GraphedTrainStep)')
    ap.add_argument('--no-gat-plus', action='store_true',
                    help="skip the extra timing of conv_type='GAT_plus' on the same mesh batch (default workload, one GPU only)")
    ap.add_argument('--no-other-workloads', action='store_true',
                    help='skip the timing of the other single-GPU BASELINE.json configurations (default workload, one GPU only)')
    ap.add_argument('--no-companion', action='store_true',
                    help='skip the second timing of the other slot flow (dense when the headline is compact and vice versa)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
            # plain `python bench.py --gpus N`: start the N ranks as a CHILD job (one process per GPU under torch.distributed.run),
            # relay its one JSON line and its exit code.  Before anything here has touched a GPU, and never through exec: this
            # process only waits.
            import socket
            import subprocess
            with socket.socket() as so:
                so.bind(('127.0.0.1', 0))
                port = so.getsockname()[1]
            cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
                   '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
            env = dict(os.environ)
            env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
            child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
            lines = [ln for ln in child.stdout.splitlines() if ln.startswith('{')]
            if lines:
                print(lines[-1])
            sys.stdout.flush()
            raise SystemExit(child.returncode if child.returncode else (0 if lines else 1))
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, or plainly (the ranks are then started from here)")
    import torch.distributed as dist
    # rehearsal of the N > 1 code path on a box with fewer GPUs than ranks (never used by the driver): GADAPT_BENCH_SHARE_GPU=1
    # maps the ranks onto the GPUs that exist, GADAPT_BENCH_BACKEND=gloo moves the collectives off RCCL (which needs one GPU per rank)
    if os.environ.get('GADAPT_BENCH_SHARE_GPU') == '1':
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    backend = os.environ.get('GADAPT_BENCH_BACKEND', 'nccl')
    # Capturing the gradient all-reduce with the step (N > 1, RCCL): rehearsed first in CHILD processes, one per rank, before this
    # process touches its GPU (g_adaptivity_amd/rccl_probe.py: a failed capture is not recoverable in-process on this ROCm).
    # GADAPT_BENCH_CAPTURE_ALLREDUCE=1 / 0 force the choice without the rehearsal.
    env_cap = os.environ.get('GADAPT_BENCH_CAPTURE_ALLREDUCE')
    probe_ok = None
    # (--launch auto, the default: the fused step under data parallelism is issued, its all-reduce is an ordinary stream-ordered
    # collective - no capture of a collective, so no rehearsal either; --launch graph asks for both)
    if world > 1 and backend == 'nccl' and env_cap not in ('0', '1') and args.launch == 'graph':
        from g_adaptivity_amd.rccl_probe import rehearse
        probe_ok = rehearse()
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from g_adaptivity_amd import GNN, MeshDataset, collate, hot_path_opt, _native
    from g_adaptivity_amd.optim import FlatAdam
    from g_adaptivity_amd import mse_loss as native_mse_loss, unit_gradient
    loss_fn = F.mse_loss if args.torch_loss else native_mse_loss
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)   # the step is captured on a side stream by design

    w = WORKLOADS[args.workload]
    def rccl_capture_ok():
        if backend != 'nccl':
            if env_cap == '1' and rank == 0:
                # only RCCL collectives can be stream-captured; a capture that a synchronising collective invalidates is NOT
                # recoverable in-process (tools/capture_recovery_probe.py), so it is never attempted
                print(f"[bench] GADAPT_BENCH_CAPTURE_ALLREDUCE=1 ignored: backend '{backend}' cannot be captured", file=sys.stderr)
            return False
        if env_cap in ('0', '1'):
            return env_cap == '1'
        if probe_ok is None:                                          # not rehearsed (--launch auto / eager): collectives stay outside captures
            return False
        t = torch.tensor([1.0 if probe_ok else 0.0], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)                      # every rank's rehearsal must have passed
        return bool(t.item() == 1.0)

    def make_batch(w_):
        ds_ = MeshDataset([w_['n'], w_['n']], w_['batch'], seed=rank)    # every rank owns its own shard of meshes
        return ds_, collate(ds_.samples).to(dev)

    ds, data = make_batch(w)
    # one GPU: the optimizer step is captured with forward and backward (step count on the device).  N > 1: forward + loss +
    # backward are the replayed graph, the RCCL all-reduce of the 33 KB gradient bucket and the Adam launch (step count on
    # the device as well: no host-side value changes from step to step) follow it on the same stream; with the RCCL backend the
    # collective and Adam are captured too once a one-rank capture probe has passed on every rank (see rccl_capture_ok).
    capture_all = world == 1 or rccl_capture_ok()
    # loss.backward() with the root gradient handed over instead of created per step (g_adaptivity_amd.unit_gradient: same
    # value, two launches fewer - the one-element fill and the multiplication by it); --plain-backward: the literal call
    root = None if (args.plain_backward or args.torch_loss) else unit_gradient(dev)

    def build_runner(dense_slots, w=w, batch=None):
        """model + optimizer + step() for one slot flow; the step is a replayed hipGraph unless --no-graph / capture fails."""
        ds_, data_ = batch if batch is not None else (ds, data)
        target = data_.x_phys
        opt = hot_path_opt(mesh_dims=[w['n'], w['n']], hidden_dim=w['hidden'], num_layers=w['layers'], conv_type=w['conv'],
                           gnn_inc_feat_f=w['f'], gnn_inc_feat_uu=w['uu'], device=str(dev), loss_type='mesh_loss',
                           show_mesh_evol_plots='False', compact_slots=not dense_slots, learn_step=bool(w.get('learn_step', False)))
        torch.manual_seed(0)                                          # identical replicas
        model = GNN(ds_, opt).to(dev)
        model.train()
        optim = FlatAdam(model.parameters(), lr=opt['lr'], weight_decay=opt['decay'], capturable=True)

        def fwd_bwd():
            out = model(data_)
            loss = loss_fn(out, target)
            if root is None:
                loss.backward()
            else:
                loss.backward(gradient=root)
            return loss

        def eager_step():
            optim.zero_grad()
            loss = fwd_bwd()
            optim.step()                                              # all-reduce (N>1) + fused Adam
            return loss

        for _ in range(2):                                            # first steps eagerly: builds the CSR cache and the flat bucket
            eager_step()
        torch.cuda.synchronize()
        # The fused 13-launch iteration (training.FusedIteration: node fields read by layer 0, loss in the last layer's launch, Adam +
        # next-step coefficients in the tail) when model, optimizer, loss and batch qualify - the same step, bit-identical parameters
        # (tests/test_gpu_training.py); --no-fused-step / any other configuration: the autograd iteration.
        fused, fused_reason = None, 'disabled (--no-fused-step / --torch-loss / --plain-backward / --dense-slots)'
        if not (args.no_fused_step or args.torch_loss or args.plain_backward or dense_slots):
            from g_adaptivity_amd.training import FusedIteration
            fused_reason = FusedIteration.eligible(model, optim, loss_fn, data_, 'x_phys')
            if fused_reason is None:
                fused = FusedIteration(model, optim, loss_fn, data_, 'x_phys')
                fused.refresh_coeffs()

        def fwd_bwd_run():                                            # zero_grad + forward + loss + backward of the route taken
            if fused is not None:
                fused.forward_backward()
            else:
                optim.zero_grad(); fwd_bwd()

        def optim_run():
            if fused is not None:
                fused.finish()
            else:
                optim.step()

        graph, cap_all = None, capture_all
        # the fused step under data parallelism: issued (3 C-ABI calls + the all-reduce per step), unless --launch graph asks for the capture
        want_graph = not (args.no_graph or args.launch == 'eager' or (fused is not None and world > 1 and args.launch == 'auto'))
        launch_ab = None
        if not want_graph:
            cap_all = world == 1
        if want_graph:
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    fwd_bwd_run()
                    if cap_all:
                        optim_run()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                optim.zero_grad()
                # N > 1: other threads of the process make CUDA calls while this one captures (the RCCL process group's watchdog
                # polls events); "thread_local" keeps those from invalidating the capture.  GADAPT_BENCH_CAPTURE_MODE overrides.
                mode = os.environ.get('GADAPT_BENCH_CAPTURE_MODE', 'thread_local' if world > 1 else 'global')
                with torch.cuda.graph(g, stream=side, capture_error_mode=mode):
                    fwd_bwd_run()
                    if cap_all:
                        optim_run()
                graph = g
            except Exception as e:                                    # stay correct: fall back to eager launches
                print(f"[bench] hipGraph capture unavailable ({type(e).__name__}: {e}); using eager launches", file=sys.stderr)
                graph, cap_all = None, world == 1
                try:
                    torch.cuda.synchronize()
                except Exception:
                    pass
                _native.clear_error()                                 # the invalidated capture left HIP's per-thread last error set

        def step_replayed():
            graph.replay()                                            # forward + loss + backward (+ all-reduce + Adam when captured)
            if not cap_all:
                optim_run()                                           # all-reduce + fused Adam

        def step_issued():
            if fused is not None:
                fused.run()
            else:
                eager_step()

        # One GPU, fused step, --launch auto: a fused iteration is 3 C-ABI calls, so the host stays ahead of the GPU without a graph, and a
        # replay costs ~8 us of idle GPU per launch of the graph (docs/measurements.md K): time both ways (set-up, before the warm-up) and
        # keep the faster.  The steps are the same launches on the same buffers either way.
        if graph is not None and fused is not None and world == 1 and args.launch == 'auto':
            t_ab = {}
            for name, fn in (('replayed', step_replayed), ('issued', step_issued)) * 2:
                fn(); torch.cuda.synchronize()
                n_ab = 48
                t0 = time.perf_counter()
                for _ in range(n_ab):
                    fn()
                torch.cuda.synchronize()
                t_ab[name] = min(t_ab.get(name, 1e9), (time.perf_counter() - t0) / n_ab)
            launch_ab = {'replayed_ms_per_step': round(1e3 * t_ab['replayed'], 4), 'issued_ms_per_step': round(1e3 * t_ab['issued'], 4)}
            if t_ab['issued'] < 0.99 * t_ab['replayed']:                # (the two timings repeat to about 0.3 %)
                graph = None

        def step():
            if graph is not None:
                step_replayed()
            else:
                step_issued()

        return {'opt': opt, 'model': model, 'optim': optim, 'step': step, 'fwd_bwd': (fused.forward_backward if fused is not None else fwd_bwd), 'graph': graph,
                'capture_all': cap_all, 'fused': fused is not None, 'fused_reason': fused_reason, 'launch_ab': launch_ab}

    def barrier():
        if world > 1:
            dist.barrier()

    def timed_block(step_fn, blocks):
        """`blocks` x K steps bracketed by barrier + torch.cuda.synchronize() on both sides, MAX over ranks: seconds."""
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(blocks * args.steps):
            step_fn()
        torch.cuda.synchronize(); barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    def timed_windows(step_fn, n_windows=None):
        """W warm-up steps, then `--windows` timed windows.  A window is the K-step block (`--steps`) repeated R times back to back, R the
        smallest count that makes a window at least MIN_WINDOW_S long (K = 20 steps of 0.3 ms are 6 ms: too short for a clock or a
        utilisation sampler to corroborate - VERDICT r5 weak 11); R comes from one untimed K-step block and is the same on every rank
        (MAX over ranks).  Returns (per-window seconds PER K-STEP BLOCK, R)."""
        for _ in range(args.warmup):
            step_fn()
        # (the shorter of two probes, and 20 % of headroom: a K-step block of a few steps is mostly its two synchronisations, so a probe
        # over-estimates the block and the windows came out at 45 instead of 50 ms)
        probe = min(timed_block(step_fn, 1), timed_block(step_fn, 1))
        blocks = max(1, min(4096, int(1.2 * MIN_WINDOW_S / max(probe, 1e-6)) + 1))
        return [timed_block(step_fn, blocks) / blocks for _ in range(max(n_windows or args.windows, 1))], blocks

    def summarise(timed):
        times, blocks = timed
        srt = sorted(times)
        med = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
        per = lambda t: round(1e3 * t / args.steps, 4)                # noqa: E731
        return med, {'n': len(times), 'steps_per_window': args.steps * blocks, 'blocks_per_window': blocks, 'min_window_ms': round(1e3 * MIN_WINDOW_S, 1),
                     'window_ms_median': round(1e3 * med * blocks, 2), 'ms_per_step_min': per(srt[0]), 'ms_per_step_median': per(med),
                     'ms_per_step_max': per(srt[-1])}

    main_run = build_runner(args.dense_slots)
    model, optim, fwd_bwd, opt = main_run['model'], main_run['optim'], main_run['fwd_bwd'], main_run['opt']
    graph, capture_all = main_run['graph'], main_run['capture_all']
    elapsed, windows = summarise(timed_windows(main_run['step']))     # the headline: the MEDIAN window
    meshes = w['batch'] * world * args.steps
    value = meshes / elapsed

    # N > 1: what the gradient exchange costs per step (latency-bound, reported as us/step) - the collective
    # the step issues (FlatAdam.all_reduce on the flat gradient bucket), alone on the stream, event pair around it (RCCL) or host
    # clock around call + synchronisation (other backends); median of 30
    allreduce_us = None
    if world > 1 and optim.grad_bucket is not None:
        samples = []
        for _ in range(35):
            barrier(); torch.cuda.synchronize()
            if backend == 'nccl':
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); optim.all_reduce(); e1.record()
                torch.cuda.synchronize()
                samples.append(1e3 * e0.elapsed_time(e1))
            else:
                t0 = time.perf_counter(); optim.all_reduce(); torch.cuda.synchronize()
                samples.append(1e6 * (time.perf_counter() - t0))
        samples = sorted(samples[5:])
        t = torch.tensor([samples[len(samples) // 2]], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        allreduce_us = round(float(t.item()), 2)

    # the other slot flow in the same run: the headline's compact slots follow the identity encoder's
    # zero-pad contract; the dense flow is the literal one SURVEY.md 8(d)'s bytes describe.  Fused GRAND kernels only.
    companion = None
    if not args.no_companion and w['conv'] in ('GRAND', 'GRAND_plus'):
        other = build_runner(not args.dense_slots)
        c_el, c_win = summarise(timed_windows(other['step']))
        companion = {'slots': 'compact' if args.dense_slots else 'dense', 'value': round(meshes / c_el, 1),
                     'ms_per_step': round(1e3 * c_el / args.steps, 4), 'windows': c_win,
                     'launch': 'hipgraph' if other['graph'] is not None else 'eager'}
        del other
        torch.cuda.synchronize()

    # the same mesh batch through the other trainable conv of the operator factory (GNN.py:120-121: conv_type='GAT_plus', the fused block
    # of csrc/gadapt_gat.inc), same step, same timing protocol, fewer windows: on the default line so the driver's record carries it
    gat_plus = None
    if rank == 0 and world == 1 and not args.no_gat_plus and args.workload == 'poisson2d_64x64_b32_L4_C64' and not args.dense_slots:
        wg = WORKLOADS['poisson2d_64x64_b32_L4_C64_GAT_plus']
        other = build_runner(False, wg)
        g_el, g_win = summarise(timed_windows(other['step'], min(args.windows, 3)))
        gat_plus = {'workload': 'poisson2d_64x64_b32_L4_C64_GAT_plus', 'value': round(wg['batch'] * args.steps / g_el, 1), 'unit': 'meshes/s',
                    'ms_per_step': round(1e3 * g_el / args.steps, 4), 'windows': g_win, 'launch': 'hipgraph+adam' if other['graph'] is not None else 'eager'}
        del other
        torch.cuda.synchronize()

    def instrument(run, w, workload):
        """Second, instrumented pass of `run` (eager launches, HIP events on the launch stream around every hot-kernel launch):
        (kernels, roofline) - per-kernel, per-variant medians and the roofline object of the dominant kernel.  Rank 0 only."""
        roofline, kernels = None, {}
        model, optim, fwd_bwd = run['model'], run['optim'], run['fwd_bwd']
        import ctypes as C
        lib = _native.lib()
        lib.gadapt_profile_reset(); lib.gadapt_profile_enable(1)
        def local_step():                                             # rank 0 only: no collective in here
            optim.zero_grad()
            fwd_bwd()                                                 # (the fused route: its forward + backward launches, no optimizer)
        for _ in range(args.steps):
            local_step()
        # dispatch share of an event pair: empty launches queued behind real work, bracketed the same way
        for _ in range(4):
            local_step()
            lib.gadapt_profile_calibrate(16, _native.current_stream(dev))
        torch.cuda.synchronize()
        lib.gadapt_profile_enable(0)
        def median_of(kid):
            cbuf = (C.c_double * 256)()
            ccnt = lib.gadapt_profile_samples(kid, cbuf, 256)
            cal = sorted(cbuf[i] for i in range(max(ccnt, 0)))
            return cal[len(cal) // 2] if cal else 0.0
        p1, p2 = median_of(3), median_of(4)
        event_overhead_ms = min(max(2.0 * p1 - p2, 0.0), p1)          # D = 2 p1 - p2 (see gadapt_profile_calibrate)
        graph_obj = next(iter(model._graphs.values()))
        if w['conv'] == 'GAT_plus':
            graph_obj = graph_obj.with_self_loops()                   # the graph the GAT_plus kernels walk (GATConv's self-loops)
        n_nodes, n_edges = graph_obj.num_nodes, graph_obj.num_edges
        for kid, name in ((0, 'forward'), (1, 'backward_target'), (2, 'backward_source')):
            cap = 4 * args.steps * w['layers'] + 16
            buf, vbuf = (C.c_double * cap)(), (C.c_int * cap)()
            cnt = lib.gadapt_profile_samples(kid, buf, cap)
            vcnt = lib.gadapt_profile_variants(kid, vbuf, cap)
            xs = [buf[i] for i in range(max(cnt, 0))]
            if not xs or vcnt != cnt:
                continue
            # launches differ by layer (compact top gradient, compact layer-0 input, head-only last output): median over
            # the steps for each position within a step (drops launches that waited on the host); positions of the
            # same variant are averaged, and every variant is priced with the bytes IT moves
            per_step = max(1, cnt // (args.steps + 4))
            groups = {}
            for k in range(per_step):
                col = sorted(xs[k::per_step])
                groups.setdefault(vbuf[k], []).append(col[len(col) // 2])
            variants = {}
            for v, meds in sorted(groups.items()):
                # priced with the RAW event-pair median: an event pair also times the dispatch of the launch it brackets, so
                # this is an upper bound of the kernel's own duration (conservative for the roofline).  The calibrated dispatch
                # share is reported beside it (`net_us`), never subtracted from what prices `frac`.
                raw_ms = max(sum(meds) / len(meds), 1e-6)
                by = (algorithmic_bytes_gat if w['conv'] == 'GAT_plus' else algorithmic_bytes)(name, n_nodes, n_edges, w['hidden'], v)
                variants[VARIANT_NAMES.get(v, str(v))] = {'launches_per_step': len(meds), 'avg_us': round(raw_ms * 1e3, 2),
                                                          'net_us': round(max(raw_ms - event_overhead_ms, 1e-6) * 1e3, 2),
                                                          'alg_bytes_per_launch': by, 'achieved_GBs': round(by / (raw_ms * 1e-3) / 1e9, 1)}
            tot = sum(v['avg_us'] * v['launches_per_step'] for v in variants.values())
            kernels[name] = {'launches_per_step': per_step, 'avg_us': round(tot / per_step, 2), 'variants': variants}
        lib.gadapt_profile_reset()
        if kernels:
            # dominant kernel = the hot kernel with the largest share of the step (all its launches); it is priced on its DENSE
            # launches - full-width input, full-width output, the bytes SURVEY.md 8(d) counts - not on the cheaper compact ones
            dom = max(kernels, key=lambda k: kernels[k]['avg_us'] * kernels[k]['launches_per_step'])
            dvar = 'dense' if 'dense' in kernels[dom]['variants'] else max(
                kernels[dom]['variants'], key=lambda v: kernels[dom]['variants'][v]['avg_us'] * kernels[dom]['variants'][v]['launches_per_step'])
            kd = kernels[dom]['variants'][dvar]
            traffic = None
            pmc = load_pmc(workload)                                 # filled from separate rocprofv3 --pmc passes
            ent = pmc.get(f'{dom}:{dvar}', pmc.get(dom))
            if isinstance(ent, dict):
                traffic = ent.get('traffic_bytes')
            prof_us, prof_file = load_profile_avg_us(workload, dom, dvar)
            roofline = {'kernel': dom, 'variant': dvar, 'bound': 'hbm', 'achieved': kd['achieved_GBs'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': round(kd['achieved_GBs'] / HBM_PEAK_GBS, 4), 'traffic': traffic,
                        'traffic_note': 'fabric-side bytes per launch of this kernel variant, (2 FETCH_SIZE + WRITE_SIZE) KiB, read from the '
                                        'committed profiles/pmc.json (separate rocprofv3 --pmc passes, tools/profile_all.sh); not measured in this run',
                        'avg_launch_us': kd['avg_us'], 'alg_bytes_per_launch': kd['alg_bytes_per_launch'],
                        'duration_note': 'avg_launch_us = raw HIP-event-pair median on the launch stream (includes the dispatch share of the '
                                         'pair); frac = alg_bytes_per_launch / avg_launch_us / peak',
                        'event_pair_overhead_us': round(event_overhead_ms * 1e3, 2), 'avg_launch_us_net_of_overhead': kd['net_us'],
                        # the same quantity from the committed rocprofv3 --kernel-trace --stats summary of this workload
                        'profile': None if prof_us is None else {'file': prof_file, 'avg_us': round(prof_us, 2),
                                                                 'frac': round(kd['alg_bytes_per_launch'] / (prof_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}}
        return kernels, roofline

    roofline, kernels = None, {}
    if rank == 0:
        kernels, roofline = instrument(main_run, w, args.workload)

    # ---- the other single-GPU BASELINE.json configurations on the same line: same step, same timing protocol, two
    # windows, their own instrumented pass; no second slot flow, no CPU leg.  Default workload, one GPU only.
    other_workloads = None
    if rank == 0 and world == 1 and not args.no_other_workloads and args.workload == 'poisson2d_64x64_b32_L4_C64' and not args.dense_slots:
        other_workloads = {}
        for name in OTHER_BASELINE_WORKLOADS:
            wo = WORKLOADS[name]
            run = build_runner(False, wo, make_batch(wo))
            o_el, o_win = summarise(timed_windows(run['step'], 2))
            o_kernels, o_roof = instrument(run, wo, name)
            keep = ('kernel', 'variant', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_us', 'alg_bytes_per_launch', 'profile')
            other_workloads[name] = {'value': round(wo['batch'] * args.steps / o_el, 1), 'unit': 'meshes/s', 'ms_per_step': round(1e3 * o_el / args.steps, 4),
                                     'windows': o_win, 'launch': 'hipgraph+adam' if run['graph'] is not None else ('issued: 3 C-ABI calls per step' if run['fused'] else 'eager'), 'launch_ab': run.get('launch_ab'),
                                     'config': {'mesh': f"{wo['n']}x{wo['n']}", 'meshes_per_gpu': wo['batch'], 'mp_layers': wo['layers'], 'hidden': wo['hidden'],
                                                'conv_type': wo['conv']},
                                     'roofline': None if o_roof is None else {k_: o_roof[k_] for k_ in keep},
                                     'kernels_avg_us': {k_: v_['avg_us'] for k_, v_ in o_kernels.items()}}
            del run
            torch.cuda.synchronize()

    # ---- secondary roofline: the projections, priced as the reference formulation's GEMM flops
    # (12 N C^2 per layer: Q, K forward + dX, dW backward) against the fp32 matrix peak, whole step
    roofline_mfma = None
    if rank == 0 and kernels:
        per_gpu = value / world
        gemm_flops_per_mesh = 12.0 * (w['n'] ** 2) * w['hidden'] ** 2 * w['layers']
        tf = gemm_flops_per_mesh * per_gpu / 1e12
        # measured matrix-pipe utilisation per hot kernel (SQ_VALU_MFMA_BUSY_CYCLES over all SIMD cycles of the launch, from the
        # committed PMC passes) and its time-weighted mean over the hot kernels of a step
        pmc = load_pmc(args.workload)
        measured, wsum, tsum = {}, 0.0, 0.0
        for kname, kd in kernels.items():
            u = pmc.get(kname, {}).get('mfma_util') if isinstance(pmc.get(kname), dict) else None
            if u is not None:
                measured[kname] = u
                t = kd['avg_us'] * kd['launches_per_step']
                wsum, tsum = wsum + u * t, tsum + t
        tw = round(wsum / tsum, 4) if tsum else None
        # frac = MEASURED matrix-pipe utilisation of the hot kernels (time-weighted over a step's launches); the reference
        # formulation's GEMM flops against the fp32 matrix peak are reported under their own name - they are not a utilisation
        # (the kernels execute 8 N C^2 per layer on the bf16 / f16 pipes, not 12 N C^2 in fp32).
        roofline_mfma = {'bound': 'mfma', 'frac': tw, 'unit': 'fraction of matrix-pipe cycles busy',
                         'measured': {'matrix_pipe_busy_frac': measured, 'hot_kernels_time_weighted': tw,
                                      'source': 'profiles/pmc.json: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), '
                                                'committed rocprofv3 --pmc passes (not measured in this run)'},
                         'reference_flops_vs_fp32_peak': round(tf / FP32_MATRIX_PEAK_TFLOPS, 4),
                         'reference_flops': {'achieved_TFLOPs': round(tf, 2), 'peak_TFLOPs': FP32_MATRIX_PEAK_TFLOPS,
                                             'flops_per_mesh': gemm_flops_per_mesh,
                                             'note': 'SURVEY.md 8(d): 12 N C^2 L GEMM flops of the reference formulation x meshes/s per GPU '
                                                     'against the fp32 matrix peak; the kernels execute 8 N C^2 per layer (composite '
                                                     'A = Wk^T Wq) as split bf16 / f16 products'}}

    # ---- the reference's training loop on CHANGING batches (src/run_GNN.py:95-131): a dataset of 8 batches'
    # worth of meshes on the device, shuffled every epoch by DeviceMeshLoader, every iteration one replay of the captured step
    # (training.GraphedTrainStep), the loader gathering the next batch into the captured buffers.  Reported beside the headline,
    # which replays ONE static batch: the difference is the batch assembly (one launch) and the Python loop.
    train_loop = None
    if rank == 0 and world == 1 and not args.no_train_loop and not args.no_graph and w['conv'] in ('GRAND', 'GRAND_plus', 'GAT_plus'):
        from g_adaptivity_amd import DeviceMeshLoader, GraphedTrainStep
        n_batches = 8
        tds = MeshDataset([w['n'], w['n']], n_batches * w['batch'], seed=1)
        torch.manual_seed(0)
        tmodel = GNN(tds, opt).to(dev).train()
        toptim = FlatAdam(tmodel.parameters(), lr=opt['lr'], weight_decay=opt['decay'], capturable=True)
        tstep = GraphedTrainStep(tmodel, toptim, loss_fn=native_mse_loss)
        loader = DeviceMeshLoader(tds, batch_size=w['batch'], shuffle=True, device=dev, fields=('x_comp', 'x_phys', 'f_tensor', 'uu_tensor'),
                                  into=tstep.static_batch)
        total = torch.zeros((), device=dev)
        for batch in loader:                                          # first epoch: capture
            total += tstep(batch)
        torch.cuda.synchronize()
        # warm-up: the GPU sat idle while the 256-mesh dataset was generated on the host, and its clocks take tens of milliseconds
        # of work to come back (a 24-step timing right after the capture was seen at 3 ms per step, one run in five)
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.1:
            for batch in loader:
                total += tstep(batch)
            torch.cuda.synchronize()
        epochs = max(2, (args.steps * max(args.windows, 1) + n_batches - 1) // n_batches)
        from g_adaptivity_amd import graph as _gm
        hashed0, epoch_ms = _gm.FP_STATS['hashed'], []
        t0 = time.perf_counter()
        for _ in range(epochs):
            te = time.perf_counter()
            for batch in loader:
                total += tstep(batch)                                 # the loop's own per-step work: accumulate the loss on the device
            epoch_ms.append(round(1e3 * (time.perf_counter() - te), 3))   # (host issue time of the epoch: no synchronisation inside)
        torch.cuda.synchronize()
        tl = time.perf_counter() - t0
        train_loop = {'value': round(epochs * len(tds) / tl, 1), 'unit': 'meshes/s', 'ms_per_step': round(1e3 * tl / (epochs * n_batches), 4),
                      'steps': epochs * n_batches, 'dataset_meshes': len(tds), 'loader': 'DeviceMeshLoader(shuffle=True, into=step.static_batch)',
                      'step': 'GraphedTrainStep: zero_grad+forward+mse+backward+adam, one replay per batch',
                      'captures': len(tstep._captured), 'topology_tensors_rehashed': _gm.FP_STATS['hashed'] - hashed0, 'epoch_issue_ms': epoch_ms[:8]}
        del tstep, tmodel, toptim, loader

    # ---- CPU baseline: the oracle on the host cores, same workload, bounded sample.  A quarter of the batch per step (the
    # cost is linear in the meshes), a short sweep over thread counts (index_add_ / scatter ops stop scaling early), then
    # >= 10 timed steps at the best count.
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.pyg_restatement import OracleGNN
        sub = max(1, w['batch'] // 4)
        cdata = collate(ds.samples[:sub])
        ctarget = cdata.x_phys
        copt = dict(opt); copt['device'] = 'cpu'
        torch.manual_seed(0)
        oracle = OracleGNN(ds, copt)
        oracle.train()

        def cpu_step():
            oracle.zero_grad()
            F.mse_loss(oracle(cdata), ctarget).backward()

        def timed(k):
            t1 = time.perf_counter()
            for _ in range(k):
                cpu_step()
            return (time.perf_counter() - t1) / k

        default_threads = torch.get_num_threads()
        # (torch's default - every core of the box, 128 here - is not a candidate: index_add_ / scatter are oversubscribed by an
        # order of magnitude there: 0.5 meshes/s against 25 at 16 threads, and the one sample would eat the whole budget)
        cands = sorted({t for t in (8, 16, 32, 64) if t <= max(default_threads, 1)} or {max(default_threads, 1)})
        budget = max(args.cpu_seconds, 4.0)
        sweep = {}
        t_start = time.perf_counter()
        for t in cands:
            if sweep and time.perf_counter() - t_start > 0.3 * budget:
                break
            torch.set_num_threads(t)
            cpu_step()                                                # warm-up at this thread count
            sweep[t] = timed(1)
        best = min(sweep, key=sweep.get)
        torch.set_num_threads(best)
        left = budget - (time.perf_counter() - t_start)
        # the timed sample runs the FULL batch (the GPU leg's batch) when at least 5 steps of it fit what is left of the budget,
        # else the quarter batch the sweep used (the cost is linear in the meshes)
        full_est = sweep[best] * w['batch'] / sub
        n_timed = sub
        if n_timed < w['batch'] and 6 * full_est <= left:
            n_timed = w['batch']
            cdata = collate(ds.samples[:n_timed])
            ctarget = cdata.x_phys
            cpu_step()                                                # warm-up at this size
            left = budget - (time.perf_counter() - t_start)
        step_est = sweep[best] * n_timed / sub
        iters = max(5 if n_timed == w['batch'] else 10, min(60, int(left / max(step_est, 1e-3))))
        per = timed(iters)
        torch.set_num_threads(default_threads)
        cpu = {'value': round(n_timed / per, 2), 'unit': 'meshes/s', 'cores': best, 'host_cpus': os.cpu_count(), 'kind': 'port',
               'threads_sweep_meshes_per_s': {str(t): round(sub / v, 2) for t, v in sweep.items()},
               'sample': f"{iters} fwd+bwd steps of a {n_timed}-mesh batch of the same workload ({iters * per:.1f} s) at {best} threads "
                         f"(the best of {sorted(sweep)}, swept on a {sub}-mesh batch), CPU restatement of the reference path "
                         f"(PyG-equivalent op sequence), no optimizer step"}

    if rank == 0:
        line = {
            'metric': 'meshes/sec fwd+bwd, 2D Poisson 64x64 mesh graph, batch 32, 1/2/4/8 GPU',
            'value': round(value, 1), 'unit': 'meshes/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 4), 'windows': windows, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': args.workload, 'mesh': f"{w['n']}x{w['n']}", 'meshes_per_gpu': w['batch'],
                       'global_batch': w['batch'] * world, 'mp_layers': w['layers'], 'hidden': w['hidden'],
                       'conv_type': w['conv'], 'parallelism': f'dp{world}',
                       'step': 'zero_grad+forward+mse+backward+allreduce+adam', 'fused_step': main_run['fused'], 'fused_step_off_reason': main_run['fused_reason'], 'loss': 'torch' if args.torch_loss else 'native', 'root_gradient': 'created per step (loss.backward())' if root is None else 'preallocated (unit_gradient)', 'slots': 'dense' if args.dense_slots else 'compact', 'slots_note': None if args.dense_slots else 'identity encoder = zero-pad (GNN.py:75-82): layer 0 reads the [N,4] encoder output in forward and backward, layer 1 hands it the 4 gradient columns it reads, the last layer writes the [N,4] head the model returns (GNN.py:299) and takes the compact top gradient; --dense-slots runs the literal dense flow', 'launch': ('hipgraph+adam' if world == 1 else ('hipgraph+allreduce+adam' if capture_all else 'hipgraph, then allreduce+adam')) if graph is not None else ('issued: 3 C-ABI calls (13 launches)' + (' + all-reduce' if world > 1 else '') + ' per step' if main_run['fused'] else 'eager'), 'launch_ab': main_run.get('launch_ab')},
            'roofline': roofline, 'roofline_mfma': roofline_mfma, 'kernels': kernels, 'cpu_baseline': cpu, 'train_loop': train_loop,
            'gat_plus': gat_plus, 'other_workloads': other_workloads,
        }
        if world > 1:
            line['allreduce_us_per_step'] = allreduce_us
            line['rccl'] = {'world': world, 'backend': backend, 'captured': bool(graph is not None and capture_all),
                            'bucket_bytes': None if optim.grad_bucket is None else 4 * optim.grad_bucket.numel(),
                            'timing': 'HIP event pair around the collective alone on the stream' if backend == 'nccl' else 'host clock around call + synchronize'}
        if companion is not None:                                     # same run, the other slot flow (see above)
            key = 'dense_slots' if companion['slots'] == 'dense' else 'compact_slots'
            line['value_' + key], line['ms_per_step_' + key] = companion['value'], companion['ms_per_step']
            line['windows_' + key] = companion['windows']
        print(json.dumps(line))
    if world > 1:
        dist.barrier()                                                # rank 0 was still measuring: leave together
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
