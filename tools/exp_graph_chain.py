#!/usr/bin/env python
"""GPU box: what a dependent chain of trivial kernels costs inside a hipGraph replay (per kernel), against the same chain issued eagerly."""
import time, torch
dev = torch.device('cuda:0')
x = torch.zeros(64, device=dev)
for K in (1, 4, 8, 16, 32):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            for _ in range(K): x.add_(1.0)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(K): x.add_(1.0)
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(200): g.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 200)
    print(f"{K:3d} kernels per replay: {best * 1e6:7.1f} us per replay = {best * 1e6 / K:5.2f} us per kernel", flush=True)
