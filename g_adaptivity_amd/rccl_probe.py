"""Rehearsal: can this process layout capture an RCCL all-reduce in a hipGraph?

The data-parallel step's only collective is one all-reduce of the flat gradient bucket.  Captured in the
step's hipGraph it costs nothing on the host; issued eagerly after every replay it is two more launches on the critical path.
RCCL collectives are stream-capturable, but a capture that goes wrong is NOT recoverable in-process on this ROCm (every later
launch of the thread fails with hipErrorStreamCaptureInvalidated, `tools/capture_recovery_probe.py`).  So callers rehearse in
CHILD processes first: `python -m g_adaptivity_amd.rccl_probe` is launched once per rank (same RANK / WORLD_SIZE / LOCAL_RANK,
its own rendezvous port) BEFORE the parent touches its GPU; the children form their own RCCL group, capture
`copy -> all_reduce -> scale` in a graph, replay it on changing inputs and check the sums.  Exit code 0 and the line
`RCCL_CAPTURE_OK` mean the layout captures; anything else (failure, crash, timeout) means "issue the collective eagerly".

`bench.py` and `GraphedTrainStep` users call `rehearse()`; `tests/test_gpu_callers.py` runs the one-rank case on the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys

PORT_OFFSET = 17


def _child() -> int:
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if os.environ.get('GADAPT_BENCH_SHARE_GPU') == '1':
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    import datetime
    dist.init_process_group('nccl', device_id=dev, timeout=datetime.timedelta(seconds=int(os.environ.get('GADAPT_PROBE_TIMEOUT', 60))))
    n = 8320                                                      # the bucket of the metric workload: 2 (C^2 + C) floats at C = 64
    src = torch.zeros(n, device=dev)
    buf = torch.zeros(n, device=dev)
    dist.all_reduce(buf)                                          # communicator set-up outside the capture
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        buf.copy_(src); dist.all_reduce(buf); buf.mul_(1.0 / world)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode='thread_local'):
        buf.copy_(src)
        dist.all_reduce(buf)
        buf.mul_(1.0 / world)
    ok = True
    for k in range(4):
        src.copy_(torch.arange(n, device=dev, dtype=torch.float32) * (k + 1) + rank)
        g.replay()
        torch.cuda.synchronize()
        want = torch.arange(n, device=dev, dtype=torch.float32) * (k + 1) + (world - 1) / 2.0
        ok = ok and bool(torch.allclose(buf, want, rtol=1e-6, atol=1e-4))
    flag = torch.tensor([1.0 if ok else 0.0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.barrier()
    dist.destroy_process_group()
    if flag.item() == 1.0:
        print("RCCL_CAPTURE_OK", flush=True)
        return 0
    print("RCCL_CAPTURE_WRONG_RESULT", flush=True)
    return 1


def rehearse(timeout: float = 120.0) -> bool:
    """Run the probe for THIS rank in a child process (call it on every rank, before the caller initialises its GPU or its
    process group).  True when the child reports success; ranks must still agree among themselves afterwards (all-reduce MIN of
    the flags) - a child that died on another rank makes its peers time out and report False too."""
    # (a child of a torch.distributed.run worker must not inherit the agent's rendezvous: with TORCHELASTIC_USE_AGENT_STORE set,
    # rank 0 would wait for a store the agent hosts at the ORIGINAL port instead of opening one at the probe's port)
    env = {k: v for k, v in os.environ.items() if not k.startswith('TORCHELASTIC_')}
    env.setdefault('MASTER_ADDR', '127.0.0.1')
    env['MASTER_PORT'] = str(int(env.get('MASTER_PORT', '29500')) + PORT_OFFSET)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env['PYTHONPATH'] = root + os.pathsep + env.get('PYTHONPATH', '')
    try:
        r = subprocess.run([sys.executable, '-m', 'g_adaptivity_amd.rccl_probe'], env=env, capture_output=True, text=True, timeout=timeout)
    except Exception:
        return False
    return r.returncode == 0 and 'RCCL_CAPTURE_OK' in r.stdout


if __name__ == '__main__':
    sys.exit(_child())
