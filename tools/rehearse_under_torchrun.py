"""Diagnostic: g_adaptivity_amd.rccl_probe.rehearse() called from torch.distributed.run workers (what bench.py --gpus N does before
it touches its GPU).  Prints per rank the verdict and the seconds it took.  On a one-GPU box with two ranks RCCL cannot form the
group (one GPU per rank), so the expected verdict is False - quickly, not after a rendezvous timeout."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g_adaptivity_amd.rccl_probe import rehearse
t0 = time.time()
ok = rehearse(timeout=float(os.environ.get('REHEARSE_TIMEOUT', 150)))
print(f"rank {os.environ.get('RANK')} rehearse -> {ok} in {time.time() - t0:.1f} s", flush=True)
