#!/usr/bin/env python
"""Diagnostic (GPU box): what it takes to launch again after a hipGraph capture was invalidated (a synchronising call inside the
capture, as a non-capturable collective makes).  python tools/capture_recovery_probe.py <strategy>"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g_adaptivity_amd import _native   # noqa: E402

strategy = sys.argv[1] if len(sys.argv) > 1 else 'A'
dev = torch.device('cuda:0')
x = torch.randn(1024, device=dev)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=side, capture_error_mode='thread_local'):
        y = x * 2
        torch.cuda.synchronize()                      # invalidates the capture
        z = y + 1
except Exception as e:
    print('capture failed:', type(e).__name__, str(e).splitlines()[0])
hip = C.CDLL('libamdhip64.so')
if strategy in ('B', 'C'):
    graph_out = C.c_void_p()
    rc = hip.hipStreamEndCapture(C.c_void_p(side.cuda_stream), C.byref(graph_out))
    print('hipStreamEndCapture rc', rc)
if strategy == 'C':
    print('hipDeviceSynchronize rc', hip.hipDeviceSynchronize())
for k in range(3):
    print('hipGetLastError', hip.hipGetLastError())
print('clear_error ->', _native.clear_error())
try:
    torch.cuda.synchronize()
    print('sync ok')
except Exception as e:
    print('sync failed', str(e).splitlines()[0])
lib = _native.lib()
for k in range(2):
    rc = lib.gadapt_profile_calibrate(1, _native.current_stream(dev))
    print('native launch rc', rc, lib.gadapt_last_error())
try:
    w = (x + 1).sum().item()
    print('torch launch ok', w)
except Exception as e:
    print('torch launch failed', str(e).splitlines()[0])
