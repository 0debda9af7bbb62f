"""Empirical HBM ceilings on the box: pure read, pure write, 1:1 and 2:1 read:write mixes,
measured with plain torch ops on 4 GiB operands (HIP events, median of 10).  Used to put the
mixed-traffic kernels (history-saving forward, adjoint) in context; see DESIGN.md."""
import json

import torch

dev = torch.device('cuda:0')
n = 1 << 30  # fp32 elements = 4 GiB
x = torch.randn(n, device=dev)
y = torch.empty_like(x)
z = torch.empty_like(x)
w = torch.empty_like(x)


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3)
    ts.sort()
    return ts[len(ts) // 2]


B = 4 * n
out = {}
out['read (sum)'] = B / timeit(lambda: x.sum())
out['write (fill)'] = B / timeit(lambda: y.fill_(1.0))
out['copy 1R:1W'] = 2 * B / timeit(lambda: y.copy_(x))
out['add 2R:1W'] = 3 * B / timeit(lambda: torch.add(x, z, out=w))
print(json.dumps({k: round(v / 1e12, 3) for k, v in out.items()}), 'TB/s')
