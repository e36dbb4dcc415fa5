"""gadapt_allreduce_flat (C-ABI, caller-owned ncclComm_t) inside a captured hipGraph, one rank.

Run as a child process by tests/test_gpu_callers.py: a capture that fails cannot be recovered from in-process on this ROCm
(tools/capture_recovery_probe.py), so it must not share a process with other tests.  Prints CABI_CAPTURE_OK on success."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g_adaptivity_amd import _native   # noqa: E402


class UniqueId(C.Structure):
    _fields_ = [('internal', C.c_char * 128)]


def main() -> int:
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    rccl = C.CDLL('librccl.so')
    uid, comm = UniqueId(), C.c_void_p()
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    lib = _native.lib()
    n = 2 * (64 * 64 + 64)
    src = torch.randn(n, device=dev)
    bucket = torch.zeros(n, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                                   # communicator warm-up outside the capture
        assert lib.gadapt_allreduce_flat(comm, bucket.data_ptr(), n, 1, side.cuda_stream) == 0
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        bucket.copy_(src)
        bucket.mul_(3.0)
        rc = lib.gadapt_allreduce_flat(comm, bucket.data_ptr(), n, 1, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.gadapt_last_error()
    ok = True
    for k in range(3):
        src.fill_(float(k + 1))
        g.replay()
        torch.cuda.synchronize()
        ok = ok and bool(torch.equal(bucket, torch.full_like(bucket, 3.0 * (k + 1))))
    rccl.ncclCommDestroy(comm)
    print("CABI_CAPTURE_OK" if ok else "CABI_CAPTURE_WRONG_RESULT", flush=True)
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
