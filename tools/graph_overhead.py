#!/usr/bin/env python3
"""Per-node cost of hipGraph replay for tiny kernels on this box (how much of the step is launch overhead)."""
import torch
dev = torch.device("cuda:0")
x = torch.zeros(64, device=dev)
y = torch.zeros(1 << 20, device=dev)
def timeit(fn, n):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 / n * 1e3
print("1000 dependent tiny adds (64 floats):   %.2f us per node" % timeit(lambda: x.add_(1.0), 1000))
print("1000 dependent 4 MB adds:               %.2f us per node" % timeit(lambda: y.add_(1.0), 1000))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(10): x.add_(1.0)
torch.cuda.synchronize(); e0.record()
for _ in range(1000): x.add_(1.0)
e1.record(); torch.cuda.synchronize()
print("1000 eager tiny adds:                   %.2f us per launch" % (e0.elapsed_time(e1)))
